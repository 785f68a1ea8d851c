// ubench5 -- the wave-cooperative Keccak-f[1600] (keccak.cuh, WaveKeccak: one state per wave) against the
// lane-per-state form: final states compared word for word, and the latency of a 121-permutation chain.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I seal-embedded_amd/csrc tools/ubench5.hip -o tools/ubench5
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
#include "se_types.h"
#include "kernels/keccak.cuh"
using namespace seamd;

__global__ __launch_bounds__(64) void k_full(const uint8_t *seeds, uint32_t *out, int steps)
{
    const int t = blockIdx.x * 64 + threadIdx.x;
    uint32_t seed[16];
    for (int i = 0; i < 16; i++) seed[i] = reinterpret_cast<const uint32_t *>(seeds)[t * 16 + i];
    KeccakState st;
    prng_absorb(st, seed, 7 + t);
    for (int s = 0; s < steps; s++) keccak_f1600(st);
#pragma unroll
    for (int i = 0; i < 25; i++) out[t * 50 + 2 * i] = st.lo[i], out[t * 50 + 2 * i + 1] = st.hi[i];
}

__global__ __launch_bounds__(64) void k_wave(const uint8_t *seeds, uint32_t *out, int steps)
{
    const int ct = blockIdx.x, lane = threadIdx.x;
    WaveKeccak k;
    wave_keccak_init(k, lane);
    wave_prng_absorb(k, seeds + (size_t)ct * 64, 7 + ct, lane);
    for (int s = 0; s < steps; s++) wave_keccak_f1600(k);
    if (k.index >= 0) out[ct * 50 + 2 * k.index] = k.lo, out[ct * 50 + 2 * k.index + 1] = k.hi;
}

int main()
{
    for (int steps : {1, 121})
        for (int cts : {64, 1024, 4096})
        {
            std::vector<uint8_t> h((size_t)cts * 64);
            for (size_t i = 0; i < h.size(); i++) h[i] = (uint8_t)(i * 131u + (i >> 8) * 7u + 3u);
            uint8_t *d_seed;
            uint32_t *d_a, *d_b;
            hipMalloc(&d_seed, h.size()), hipMalloc(&d_a, cts * 200), hipMalloc(&d_b, cts * 200);
            hipMemcpy(d_seed, h.data(), h.size(), hipMemcpyHostToDevice);
            hipMemset(d_b, 0, cts * 200);
            hipEvent_t e0, e1;
            hipEventCreate(&e0), hipEventCreate(&e1);
            float ms_full = 0, ms_wave = 0;
            for (int rep = 0; rep < 3; rep++)
            {
                hipEventRecord(e0);
                hipLaunchKernelGGL(k_full, dim3(cts / 64), dim3(64), 0, 0, d_seed, d_a, steps);
                hipEventRecord(e1), hipEventSynchronize(e1), hipEventElapsedTime(&ms_full, e0, e1);
                hipEventRecord(e0);
                hipLaunchKernelGGL(k_wave, dim3(cts), dim3(64), 0, 0, d_seed, d_b, steps);
                hipEventRecord(e1), hipEventSynchronize(e1), hipEventElapsedTime(&ms_wave, e0, e1);
            }
            std::vector<uint32_t> a(cts * 50), b(cts * 50);
            hipMemcpy(a.data(), d_a, cts * 200, hipMemcpyDeviceToHost), hipMemcpy(b.data(), d_b, cts * 200, hipMemcpyDeviceToHost);
            int bad = 0, first = -1;
            for (int i = 0; i < cts * 50; i++)
                if (a[i] != b[i])
                {
                    if (first < 0) first = i;
                    bad++;
                }
            printf("steps %3d states %5d: lane-per-state %.3f ms (%.2f us/perm)   wave-per-state %.3f ms (%.2f us/perm)   "
                   "mismatching words %d (first %d)\n", steps, cts, ms_full, ms_full * 1e3 / steps, ms_wave,
                   ms_wave * 1e3 / steps, bad, first);
            (void)hipFree(d_seed), (void)hipFree(d_a), (void)hipFree(d_b);
        }
    return 0;
}
