// ubench.hip -- instruction-rate micro-benchmarks that decide the arithmetic choices of the path
// (SURVEY H4/H4b): integer multiply flavours, FP64 ops, 3-input bit ops, funnel shifts, and the
// Keccak-f[1600] permutation itself.  Build: hipcc --offload-arch=gfx950 -O3 tools/ubench.hip -o tools/ubench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include "../seal-embedded_amd/csrc/kernels/keccak.cuh"
#include "../seal-embedded_amd/csrc/kernels/modarith.cuh"

#define ITER 4096
#define CHAINS 8

template <int OP>
__global__ __launch_bounds__(256) void k_rate(uint32_t* out, uint32_t seed)
{
    uint32_t x[CHAINS], y = seed ^ threadIdx.x, z = seed * 3 + blockIdx.x;
    double d[CHAINS]; double dy = 1.0000001 + 1e-9 * threadIdx.x, dz = 0.5;
    uint64_t w[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; c++) { x[c] = threadIdx.x * 977 + c + seed; d[c] = 1.0 + c * 1e-3; w[c] = x[c]; }
    for (int i = 0; i < ITER; i++)
    {
#pragma unroll
        for (int c = 0; c < CHAINS; c++)
        {
            if (OP == 0) x[c] = x[c] * y + z;                                   // v_mad_u32_u24? mul_lo + add
            if (OP == 1) x[c] = __umulhi(x[c], y) + z;                          // v_mul_hi_u32
            if (OP == 2) w[c] = (uint64_t)(uint32_t)w[c] * y + w[c];            // v_mad_u64_u32
            if (OP == 3) d[c] = __fma_rn(d[c], dy, dz);                         // v_fma_f64
            if (OP == 4) d[c] = __dmul_rn(d[c], dy);                            // v_mul_f64
            if (OP == 5) d[c] = __dadd_rn(d[c], dz);                            // v_add_f64
            if (OP == 6) x[c] = __builtin_amdgcn_bitop3_b32(x[c], y, z, 0xD2);  // v_bitop3_b32
            if (OP == 7) x[c] = __builtin_amdgcn_alignbit(x[c], y, 7);          // v_alignbit_b32
            if (OP == 8) x[c] = x[c] ^ y;                                       // v_xor_b32
            if (OP == 9) x[c] = min(x[c], x[c] - y);                            // sub + min
            if (OP == 10) x[c] = x[c] * y;                                      // v_mul_lo_u32
            if (OP == 11) x[c] = __popc(x[c]) + y;                              // v_bcnt
        }
    }
    uint32_t acc = 0;
#pragma unroll
    for (int c = 0; c < CHAINS; c++) acc ^= x[c] ^ (uint32_t)w[c] ^ (uint32_t)(w[c] >> 32) ^ (uint32_t)__double2ll_rz(d[c] * 1e3);
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

__global__ __launch_bounds__(64) void k_keccak(uint32_t* out, int perms)
{
    seamd::KeccakState s;
    uint32_t seed[16];
    for (int i = 0; i < 16; i++) seed[i] = threadIdx.x * 31 + i + blockIdx.x;
    seamd::prng_absorb(s, seed, blockIdx.x);
    for (int p = 0; p < perms; p++) seamd::keccak_f1600(s);
    uint32_t acc = 0;
    for (int i = 0; i < 25; i++) acc ^= s.lo[i] ^ s.hi[i];
    out[blockIdx.x * 64 + threadIdx.x] = acc;
}

__global__ __launch_bounds__(256) void k_butterfly(uint32_t* out, uint32_t q)
{
    uint32_t x[8], y[8], w = 12345 + threadIdx.x, wp = (uint32_t)(((uint64_t)w << 32) / q);
    for (int c = 0; c < 8; c++) { x[c] = threadIdx.x + c; y[c] = blockIdx.x + 7 * c; }
    for (int i = 0; i < ITER; i++)
#pragma unroll
        for (int c = 0; c < 8; c++) seamd::ct_butterfly(x[c], y[c], w, wp, 0u - q, q << 1);
    uint32_t acc = 0;
    for (int c = 0; c < 8; c++) acc ^= x[c] ^ y[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <typename F>
static float time_ms(F f, int reps = 5)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < reps; r++) { hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms; }
    return best;
}

int main()
{
    uint32_t* d; hipMalloc(&d, 64 << 20);
    const int blocks = 256 * 8;  // 8 blocks of 256 threads per CU
    const char* names[] = {"mul_lo+add u32", "mul_hi_u32+add", "mad_u64_u32", "fma_f64", "mul_f64", "add_f64",
                           "bitop3_b32", "alignbit_b32", "xor_b32", "sub+min u32", "mul_lo_u32", "bcnt+add"};
    double lane_ops = (double)blocks * 256 * ITER * CHAINS;
#define RUN(OP) { float ms = time_ms([&] { hipLaunchKernelGGL(k_rate<OP>, dim3(blocks), dim3(256), 0, 0, d, 17u); }); \
                  printf("%-16s %8.3f ms  %8.2f Gop/s (lane-ops)  = %6.2f ops/clk/CU @2.4GHz\n", names[OP], ms, lane_ops / ms / 1e6, lane_ops / (ms * 1e-3) / 256 / 2.4e9); }
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9) RUN(10) RUN(11)
    for (int wpc : {4, 8, 16})
    {
        int kb = 256 * wpc, perms = 200;
        float ms = time_ms([&] { hipLaunchKernelGGL(k_keccak, dim3(kb), dim3(64), 0, 0, d, perms); });
        double total = (double)kb * 64 * perms;
        printf("keccak-f1600 %2d waves/CU: %8.3f ms  %8.2f Mperm/s  (%.0f clk per wave-perm/SIMD)\n", wpc, ms, total / ms / 1e3,
               ms * 1e-3 * 2.4e9 / ((double)kb * perms / 1024.0));
    }
    {
        float ms = time_ms([&] { hipLaunchKernelGGL(k_butterfly, dim3(blocks), dim3(256), 0, 0, d, 1053818881u); });
        double total = (double)blocks * 256 * ITER * 8;
        printf("harvey butterfly: %8.3f ms  %8.2f Gbfly/s\n", ms, total / ms / 1e6);
    }
    hipFree(d);
    return 0;
}
