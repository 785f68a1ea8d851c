// ubench2.hip -- single-instruction issue rates on gfx950 via inline asm (nothing for the compiler
// to fold).  8 independent chains per lane, 8 waves per SIMD.  Reports lane-ops/clk/CU at 2.4 GHz
// nominal (128 = full rate, 64 = half, 32 = quarter) -- compare ratios, the clock floats.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define ITER 16384
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

#define DEFK(NAME, ASM)                                                                    \
    __global__ __launch_bounds__(256) void NAME(uint32_t* out, uint32_t seed)             \
    {                                                                                      \
        uint32_t x[8], y = seed ^ threadIdx.x, z = seed * 3 + blockIdx.x;                  \
        for (int c = 0; c < 8; c++) x[c] = threadIdx.x * 977 + c + seed;                   \
        for (int i = 0; i < ITER; i++)                                                     \
        {                                                                                  \
            _Pragma("unroll") for (int c = 0; c < 8; c++)                                  \
                asm volatile(ASM : "+v"(x[c]) : "v"(y), "v"(z));                           \
        }                                                                                  \
        uint32_t acc = 0;                                                                  \
        for (int c = 0; c < 8; c++) acc ^= x[c];                                           \
        out[blockIdx.x * blockDim.x + threadIdx.x] = acc;                                  \
    }

DEFK(k_xor,      "v_xor_b32 %0, %0, %1")
DEFK(k_bitop3,   "v_bitop3_b32 %0, %0, %1, %2 bitop3:0xd2")
DEFK(k_bitop3_2, "v_bitop3_b32 %0, %0, %1, %1 bitop3:0x66")
DEFK(k_bfi,      "v_bfi_b32 %0, %0, %1, %2")
DEFK(k_alignbit, "v_alignbit_b32 %0, %0, %1, 7")
DEFK(k_alignbitv,"v_alignbit_b32 %0, %0, %1, %2")
DEFK(k_alignbyte,"v_alignbyte_b32 %0, %0, %1, 1")
DEFK(k_perm,     "v_perm_b32 %0, %0, %1, %2")
DEFK(k_and_or,   "v_and_or_b32 %0, %0, %1, %2")
DEFK(k_lshl_or,  "v_lshl_or_b32 %0, %0, 3, %1")
DEFK(k_lshl_add, "v_lshl_add_u32 %0, %0, 3, %1")
DEFK(k_add3,     "v_add3_u32 %0, %0, %1, %2")
DEFK(k_xad,      "v_xad_u32 %0, %0, %1, %2")
DEFK(k_add,      "v_add_u32 %0, %0, %1")
DEFK(k_sub,      "v_sub_u32 %0, %0, %1")
DEFK(k_min,      "v_min_u32 %0, %0, %1")
DEFK(k_lshl,     "v_lshlrev_b32 %0, 3, %0")
DEFK(k_mul_lo,   "v_mul_lo_u32 %0, %0, %1")
DEFK(k_mul_hi,   "v_mul_hi_u32 %0, %0, %1")
DEFK(k_mul_u24,  "v_mul_u32_u24 %0, %0, %1")
DEFK(k_mad_u24,  "v_mad_u32_u24 %0, %0, %1, %2")
DEFK(k_mul_hi24, "v_mul_hi_u32_u24 %0, %0, %1")
DEFK(k_bcnt,     "v_bcnt_u32_b32 %0, %0, %1")
DEFK(k_cndmask,  "v_cndmask_b32 %0, %0, %1, vcc")
DEFK(k_mov_dpp,  "v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
DEFK(k_fma32,    "v_fma_f32 %0, %0, %1, %2")
// round 4: the rest of the opcodes the product kernels issue (for the opcode-weighted VALU floor, tools/valu_mix.py)
DEFK(k_and,      "v_and_b32 %0, %0, %1")
DEFK(k_or,       "v_or_b32 %0, %0, %1")
DEFK(k_not,      "v_not_b32 %0, %0")
DEFK(k_mov,      "v_mov_b32 %0, %1")
DEFK(k_subrev,   "v_subrev_u32 %0, %0, %1")
DEFK(k_lshr,     "v_lshrrev_b32 %0, 3, %0")
DEFK(k_ashr,     "v_ashrrev_i32 %0, 3, %0")
DEFK(k_max,      "v_max_u32 %0, %0, %1")
DEFK(k_or3,      "v_or3_b32 %0, %0, %1, %2")
DEFK(k_bfe,      "v_bfe_u32 %0, %0, 3, 7")
DEFK(k_add_co,   "v_add_co_u32 %0, vcc, %0, %1")
DEFK(k_addc_co,  "v_addc_co_u32 %0, vcc, %0, %1, vcc")
DEFK(k_add_f32,  "v_add_f32 %0, %0, %1")
DEFK(k_mul_f32,  "v_mul_f32 %0, %0, %1")
DEFK(k_xor_e64,  "v_xor_b32_e64 %0, %0, %1")
DEFK(k_add_e64,  "v_add_u32_e64 %0, %0, %1")
DEFK(k_xor_sdwa, "v_xor_b32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD")
DEFK(k_accw,     "v_accvgpr_write_b32 a0, %0")

// A Keccak-like MIX on independent chains: one v_xor, one v_bitop3, one v_alignbit per chain and iteration.  If
// single-opcode rates composed, this would run at the weighted mean of the three; what it measures is what a mixed
// stream really gets (the Keccak kernels sit ~10 % above their opcode-weighted bound).
#define DEFKSEQ(NAME, ASM, NOPS)                                                           \
    __global__ __launch_bounds__(256) void NAME(uint32_t* out, uint32_t seed)             \
    {                                                                                      \
        uint32_t x[8], y = seed ^ threadIdx.x, z = seed * 3 + blockIdx.x;                  \
        for (int c = 0; c < 8; c++) x[c] = threadIdx.x * 977 + c + seed;                   \
        for (int i = 0; i < ITER / NOPS; i++)                                              \
        {                                                                                  \
            _Pragma("unroll") for (int c = 0; c < 8; c++)                                  \
                asm volatile(ASM : "+v"(x[c]) : "v"(y), "v"(z));                           \
        }                                                                                  \
        uint32_t acc = 0;                                                                  \
        for (int c = 0; c < 8; c++) acc ^= x[c];                                           \
        out[blockIdx.x * blockDim.x + threadIdx.x] = acc;                                  \
    }
DEFKSEQ(k_mix_keccak, "v_xor_b32 %0, %0, %1\n\tv_bitop3_b32 %0, %0, %1, %2 bitop3:0xd2\n\tv_alignbit_b32 %0, %0, %1, 7", 3)
DEFKSEQ(k_mix_xor_align, "v_xor_b32 %0, %0, %1\n\tv_alignbit_b32 %0, %0, %1, 7", 2)
DEFKSEQ(k_mix_ntt, "v_sub_u32 %0, %0, %1\n\tv_min_u32 %0, %0, %1\n\tv_mul_hi_u32 %0, %0, %2\n\tv_mul_lo_u32 %0, %0, %1\n\tv_add3_u32 %0, %0, %1, %2", 5)

// The same three opcodes in RUNS: 8 (or 4 + 4) instructions of one kind back to back, then the next kind -- does the
// fast class need homogeneous neighbours in the instruction stream of ONE wave?
#define DEFKRUNS(NAME, A1, A2, A3)                                                         \
    __global__ __launch_bounds__(256) void NAME(uint32_t* out, uint32_t seed)             \
    {                                                                                      \
        uint32_t x[8], y = seed ^ threadIdx.x, z = seed * 3 + blockIdx.x;                  \
        for (int c = 0; c < 8; c++) x[c] = threadIdx.x * 977 + c + seed;                   \
        for (int i = 0; i < ITER / 3; i++)                                                 \
        {                                                                                  \
            _Pragma("unroll") for (int c = 0; c < 8; c++) asm volatile(A1 : "+v"(x[c]) : "v"(y), "v"(z)); \
            _Pragma("unroll") for (int c = 0; c < 8; c++) asm volatile(A2 : "+v"(x[c]) : "v"(y), "v"(z)); \
            _Pragma("unroll") for (int c = 0; c < 8; c++) asm volatile(A3 : "+v"(x[c]) : "v"(y), "v"(z)); \
        }                                                                                  \
        uint32_t acc = 0;                                                                  \
        for (int c = 0; c < 8; c++) acc ^= x[c];                                           \
        out[blockIdx.x * blockDim.x + threadIdx.x] = acc;                                  \
    }
DEFKRUNS(k_runs8_keccak, "v_xor_b32 %0, %0, %1", "v_bitop3_b32 %0, %0, %1, %2 bitop3:0xd2", "v_alignbit_b32 %0, %0, %1, 7")
DEFKRUNS(k_runs8_xxa, "v_xor_b32 %0, %0, %1", "v_xor_b32 %0, %0, %2", "v_alignbit_b32 %0, %0, %1, 7")
DEFKRUNS(k_runs8_xxx, "v_xor_b32 %0, %0, %1", "v_xor_b32 %0, %0, %2", "v_xor_b32 %0, %0, %1")

// Which of the two is it -- homogeneous neighbours, or INDEPENDENT neighbours?  k_mixind: the three opcodes
// alternate instruction by instruction, but adjacent instructions work on different chains (independent);
// k_runsN: runs of N independent instructions of one kind; k_dep1: ONE dependent chain per lane.
#define OPSEL(K) ((K) % 3 == 0 ? "v_xor_b32 %0, %0, %1" : (K) % 3 == 1 ? "v_bitop3_b32 %0, %0, %1, %2 bitop3:0xd2" : "v_alignbit_b32 %0, %0, %1, 7")
__global__ __launch_bounds__(256) void k_mixind(uint32_t* out, uint32_t seed)
{
    uint32_t x[8], y = seed ^ threadIdx.x, z = seed * 3 + blockIdx.x;
    for (int c = 0; c < 8; c++) x[c] = threadIdx.x * 977 + c + seed;
    for (int i = 0; i < ITER / 3; i++)
    {
#define ONE(C, K) if ((K) % 3 == 0) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x[C]) : "v"(y), "v"(z)); \
                  else if ((K) % 3 == 1) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0xd2" : "+v"(x[C]) : "v"(y), "v"(z)); \
                  else asm volatile("v_alignbit_b32 %0, %0, %1, 7" : "+v"(x[C]) : "v"(y), "v"(z));
        _Pragma("unroll") for (int r = 0; r < 3; r++)
        {
            ONE(0, 0 + r) ONE(1, 1 + r) ONE(2, 2 + r) ONE(3, 0 + r) ONE(4, 1 + r) ONE(5, 2 + r) ONE(6, 0 + r) ONE(7, 1 + r)
        }
#undef ONE
    }
    uint32_t acc = 0;
    for (int c = 0; c < 8; c++) acc ^= x[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
#define DEFKRUNSN(NAME, N)                                                                 \
    __global__ __launch_bounds__(256) void NAME(uint32_t* out, uint32_t seed)             \
    {                                                                                      \
        uint32_t x[8], y = seed ^ threadIdx.x, z = seed * 3 + blockIdx.x;                  \
        for (int c = 0; c < 8; c++) x[c] = threadIdx.x * 977 + c + seed;                   \
        for (int i = 0; i < ITER / 3; i++)                                                 \
        {                                                                                  \
            _Pragma("unroll") for (int g = 0; g < 8 / N; g++)                              \
            {                                                                              \
                _Pragma("unroll") for (int c = 0; c < N; c++) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x[g * N + c]) : "v"(y), "v"(z)); \
                _Pragma("unroll") for (int c = 0; c < N; c++) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0xd2" : "+v"(x[g * N + c]) : "v"(y), "v"(z)); \
                _Pragma("unroll") for (int c = 0; c < N; c++) asm volatile("v_alignbit_b32 %0, %0, %1, 7" : "+v"(x[g * N + c]) : "v"(y), "v"(z)); \
            }                                                                              \
        }                                                                                  \
        uint32_t acc = 0;                                                                  \
        for (int c = 0; c < 8; c++) acc ^= x[c];                                           \
        out[blockIdx.x * blockDim.x + threadIdx.x] = acc;                                  \
    }
DEFKRUNSN(k_runs1, 1)
DEFKRUNSN(k_runs2, 2)
DEFKRUNSN(k_runs4, 4)
__global__ __launch_bounds__(256) void k_dep1_xor(uint32_t* out, uint32_t seed)
{
    uint32_t x = seed ^ threadIdx.x, y = seed * 3 + blockIdx.x;
    for (int i = 0; i < ITER; i++)
    {
        _Pragma("unroll") for (int c = 0; c < 8; c++) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x) : "v"(y));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
}
__global__ __launch_bounds__(256) void k_dep1_alignbit(uint32_t* out, uint32_t seed)
{
    uint32_t x = seed ^ threadIdx.x, y = seed * 3 + blockIdx.x;
    for (int i = 0; i < ITER; i++)
    {
        _Pragma("unroll") for (int c = 0; c < 8; c++) asm volatile("v_alignbit_b32 %0, %0, %1, 7" : "+v"(x) : "v"(y));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
}

// 64-bit ops on register pairs
#define DEFK64(NAME, ASM)                                                                  \
    __global__ __launch_bounds__(256) void NAME(uint32_t* out, uint32_t seed)             \
    {                                                                                      \
        uint64_t x[8], y = seed ^ threadIdx.x, z = seed * 3 + blockIdx.x;                  \
        for (int c = 0; c < 8; c++) x[c] = threadIdx.x * 977 + c + seed;                   \
        for (int i = 0; i < ITER; i++)                                                     \
        {                                                                                  \
            _Pragma("unroll") for (int c = 0; c < 8; c++)                                  \
                asm volatile(ASM : "+v"(x[c]) : "v"(y), "v"(z));                           \
        }                                                                                  \
        uint32_t acc = 0;                                                                  \
        for (int c = 0; c < 8; c++) acc ^= (uint32_t)x[c] ^ (uint32_t)(x[c] >> 32);        \
        out[blockIdx.x * blockDim.x + threadIdx.x] = acc;                                  \
    }
DEFK64(k_lshl64,  "v_lshlrev_b64 %0, 3, %0")
DEFK64(k_fma64,   "v_fma_f64 %0, %0, %1, %2")
DEFK64(k_mul64,   "v_mul_f64 %0, %0, %1")
DEFK64(k_add64,   "v_add_f64 %0, %0, %1")
DEFK64(k_lshladd64,"v_lshl_add_u64 %0, %0, 3, %1")
DEFK64(k_max64,   "v_max_f64 %0, %0, %1")
DEFK64(k_rndne64, "v_rndne_f64 %0, %0")

// 64-bit accumulator with 32-bit operands / conversions between the two widths
#define DEFKMIX(NAME, ASM)                                                                 \
    __global__ __launch_bounds__(256) void NAME(uint32_t* out, uint32_t seed)             \
    {                                                                                      \
        uint64_t x[8];                                                                     \
        uint32_t y = seed ^ threadIdx.x, z = seed * 3 + blockIdx.x, w[8];                  \
        for (int c = 0; c < 8; c++) x[c] = threadIdx.x * 977 + c + seed, w[c] = c + seed;  \
        for (int i = 0; i < ITER; i++)                                                     \
        {                                                                                  \
            _Pragma("unroll") for (int c = 0; c < 8; c++)                                  \
                asm volatile(ASM : "+v"(x[c]), "+v"(w[c]) : "v"(y), "v"(z));               \
        }                                                                                  \
        uint32_t acc = 0;                                                                  \
        for (int c = 0; c < 8; c++) acc ^= (uint32_t)x[c] ^ (uint32_t)(x[c] >> 32) ^ w[c]; \
        out[blockIdx.x * blockDim.x + threadIdx.x] = acc;                                  \
    }
DEFKMIX(k_mad6432, "v_mad_u64_u32 %0, vcc, %2, %3, %0")
DEFKMIX(k_cvt_i32_f64, "v_cvt_i32_f64 %1, %0")
DEFKMIX(k_cvt_f64_i32, "v_cvt_f64_i32 %0, %1")
DEFKMIX(k_cmp_gt, "v_cmp_gt_u32 vcc, %1, %2")
DEFKMIX(k_cmp_cnd, "v_cmp_gt_u32 vcc, %1, %2\n\tv_cndmask_b32 %1, %1, %3, vcc")

template <typename K>
static void run(const char* name, K kern, uint32_t* d)
{
    const int blocks = 256 * 8;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, 17u); hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 5; r++) { hipEventRecord(a); hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, 17u); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms; }
    double ops = (double)blocks * 256 * ITER * 8;
    printf("%-14s %7.3f ms  %7.1f lane-ops/clk/CU\n", name, best, ops / (best * 1e-3) / 256 / 2.4e9);
}

int main()
{
    uint32_t* d; hipMalloc(&d, 64 << 20);
#define R(K) run(#K, K, d);
    R(k_xor) R(k_bitop3) R(k_bitop3_2) R(k_bfi) R(k_alignbit) R(k_alignbitv) R(k_alignbyte) R(k_perm)
    R(k_and_or) R(k_lshl_or) R(k_lshl_add) R(k_add3) R(k_xad) R(k_add) R(k_sub) R(k_min) R(k_lshl)
    R(k_mul_lo) R(k_mul_hi) R(k_mul_u24) R(k_mad_u24) R(k_mul_hi24) R(k_bcnt) R(k_cndmask) R(k_mov_dpp) R(k_fma32)
    R(k_lshl64) R(k_fma64) R(k_mul64) R(k_add64) R(k_lshladd64)
    R(k_and) R(k_or) R(k_not) R(k_mov) R(k_subrev) R(k_lshr) R(k_ashr) R(k_max) R(k_or3) R(k_bfe) R(k_add_co) R(k_addc_co)
    R(k_add_f32) R(k_mul_f32) R(k_xor_e64) R(k_add_e64) R(k_xor_sdwa) R(k_accw)
    R(k_mix_keccak) R(k_mix_xor_align) R(k_mix_ntt) R(k_runs8_keccak) R(k_runs8_xxa) R(k_runs8_xxx)
    R(k_mixind) R(k_runs1) R(k_runs2) R(k_runs4) R(k_dep1_xor) R(k_dep1_alignbit)
    R(k_max64) R(k_rndne64) R(k_mad6432) R(k_cvt_i32_f64) R(k_cvt_f64_i32) R(k_cmp_gt) R(k_cmp_cnd)
    hipFree(d);
    return 0;
}
