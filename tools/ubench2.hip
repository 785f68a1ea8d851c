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
    hipFree(d);
    return 0;
}
