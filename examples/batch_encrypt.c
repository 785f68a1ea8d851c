/*
 * batch_encrypt.c -- the batched entry point next to the reference API:
 * B plaintexts per call through se_encrypt_batch on the handle se_setup returned.
 * Prints the FNV-1a-64 digest of the ciphertext records in the reference's callback order
 * (c0_j then c1_j per prime, per ciphertext) and the rate of the call (PCIe inclusive).
 *
 *   gcc examples/batch_encrypt.c -Iinclude -Lseal-embedded_amd/lib -lseal_embedded_amd \
 *       -Wl,-rpath,$PWD/seal-embedded_amd/lib -o batch_encrypt
 *   ./batch_encrypt 4096 3 8192
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "seal_embedded_amd.h"

static uint64_t fnv(uint64_t h, const void *data, size_t nbytes)
{
    const uint8_t *p = (const uint8_t *)data;
    for (size_t i = 0; i < nbytes; i++)
    {
        h ^= p[i];
        h *= 0x100000001b3ull;
    }
    return h;
}

int main(int argc, char **argv)
{
    size_t n       = argc > 1 ? (size_t)atol(argv[1]) : 4096;
    size_t nprimes = argc > 2 ? (size_t)atol(argv[2]) : 3;
    size_t B       = argc > 3 ? (size_t)atol(argv[3]) : 1024;
    SE_PARMS *parms = se_setup(n, nprimes, 0.0, SE_SYM_ENCR);

    float *values   = (float *)malloc(B * (n / 2) * sizeof(float));
    uint8_t *share  = (uint8_t *)malloc(B * 64), *seeds = (uint8_t *)malloc(B * 64);
    uint32_t *c0    = (uint32_t *)malloc(B * nprimes * n * sizeof(uint32_t));
    uint32_t *c1    = (uint32_t *)malloc(B * nprimes * n * sizeof(uint32_t));
    for (size_t b = 0; b < B; b++)
    {
        for (size_t i = 0; i < n / 2; i++)
            values[b * (n / 2) + i] = (float)((double)((((uint64_t)(i + b)) * 2654435761ull) % 100000ull) / 1000 - 50);
        for (int k = 0; k < 64; k++)
        {
            share[b * 64 + k] = (uint8_t)(k + b);
            seeds[b * 64 + k] = (uint8_t)(255 - k + 3 * b);
        }
    }
    memset(c0, 0, B * nprimes * n * sizeof(uint32_t));  /* touch the pages before timing */
    memset(c1, 0, B * nprimes * n * sizeof(uint32_t));

    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    int failed = se_encrypt_batch(parms, values, B, share, seeds, c0, c1);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    double sec = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);

    uint64_t h = 0xcbf29ce484222325ull, h0 = 0;
    for (size_t b = 0; b < B; b++)
    {
        for (size_t j = 0; j < nprimes; j++)
        {
            h = fnv(h, c0 + (b * nprimes + j) * n, n * sizeof(uint32_t));
            h = fnv(h, c1 + (b * nprimes + j) * n, n * sizeof(uint32_t));
        }
        if (b == 0) h0 = h;
    }
    printf("failed=%d B=%zu first=%016llx all=%016llx seconds=%.4f ct_per_s=%.0f\n", failed, B,
           (unsigned long long)h0, (unsigned long long)h, sec, (double)B / sec);
    se_cleanup(parms);
    free(values); free(share); free(seeds); free(c0); free(c1);
    return failed == 0 ? 0 : 1;
}
