// ubench4 -- latency of ONE sponge chain: lane-per-state Keccak-f[1600] against the pair-cooperative form
// (keccak.cuh, KeccakHalf), one wave per SIMD, and their outputs compared word for word.
//   hipcc --offload-arch=gfx950 -O3 -I seal-embedded_amd/csrc tools/ubench4.hip -o tools/ubench4
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
#include "se_types.h"
#include "kernels/keccak.cuh"
using namespace seamd;

__global__ __launch_bounds__(64) void k_full(const uint32_t *seeds, uint32_t *out, int steps)
{
    const int t = blockIdx.x * 64 + threadIdx.x;
    uint32_t seed[16];
    for (int i = 0; i < 16; i++) seed[i] = seeds[t * 16 + i];
    KeccakState st;
    prng_absorb(st, seed, 7);
    uint32_t acc = 0;
    for (int s = 0; s < steps; s++)
    {
        keccak_f1600(st);
#pragma unroll
        for (int i = 0; i < 17; i++) acc = acc * 31u + st.lo[i], acc = acc * 31u + st.hi[i];
    }
    out[t] = acc;
}

__global__ __launch_bounds__(64) void k_pair(const uint32_t *seeds, uint32_t *out, int steps)
{
    const int t    = blockIdx.x * 64 + threadIdx.x;
    const int ct   = t >> 1;
    const uint32_t part = t & 1;
    uint32_t seed[16];
    for (int i = 0; i < 16; i++) seed[i] = seeds[ct * 16 + i];
    KeccakHalf st;
    prng_absorb_half(st, seed, 7, part);
    uint32_t acc = 0;
    for (int s = 0; s < steps; s++)
    {
        keccak_half_f1600(st, part);
        // same digest as k_full: acc over (lo0, hi0, lo1, hi1, ...) -- done on the even lane with the partner's words
#pragma unroll
        for (int i = 0; i < 17; i++)
        {
            const uint32_t other = pair_swap(st.w[i]);
            const uint32_t lo = part ? other : st.w[i], hi = part ? st.w[i] : other;
            acc = acc * 31u + lo, acc = acc * 31u + hi;
        }
    }
    if (!part) out[ct] = acc;
}

int main()
{
    const int steps = 121;
    for (int waves : {256, 1024, 2048})
    {
        const int cts = waves * 64;
        std::vector<uint32_t> h(cts * 16);
        for (size_t i = 0; i < h.size(); i++) h[i] = (uint32_t)(i * 2654435761u + 12345u);
        uint32_t *d_seed, *d_a, *d_b;
        hipMalloc(&d_seed, h.size() * 4), hipMalloc(&d_a, cts * 4), hipMalloc(&d_b, cts * 4);
        hipMemcpy(d_seed, h.data(), h.size() * 4, hipMemcpyHostToDevice);
        hipEvent_t e0, e1;
        hipEventCreate(&e0), hipEventCreate(&e1);
        float ms_full = 0, ms_pair = 0;
        for (int rep = 0; rep < 3; rep++)
        {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k_full, dim3(waves), dim3(64), 0, 0, d_seed, d_a, steps);
            hipEventRecord(e1), hipEventSynchronize(e1), hipEventElapsedTime(&ms_full, e0, e1);
            hipEventRecord(e0);
            hipLaunchKernelGGL(k_pair, dim3(2 * waves), dim3(64), 0, 0, d_seed, d_b, steps);
            hipEventRecord(e1), hipEventSynchronize(e1), hipEventElapsedTime(&ms_pair, e0, e1);
        }
        std::vector<uint32_t> a(cts), b(cts);
        hipMemcpy(a.data(), d_a, cts * 4, hipMemcpyDeviceToHost), hipMemcpy(b.data(), d_b, cts * 4, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int i = 0; i < cts; i++) bad += a[i] != b[i];
        printf("%5d states x64 (%d full-lane waves): full %.3f ms = %.2f us/perm   pair %.3f ms = %.2f us/perm   mismatches %d\n",
               waves, waves, ms_full, ms_full * 1e3 / steps, ms_pair, ms_pair * 1e3 / steps, bad);
        hipFree(d_seed), hipFree(d_a), hipFree(d_b);
    }
    return 0;
}
