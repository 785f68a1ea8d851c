/*
 * api_digest.c -- a caller written against the reference's public API only
 * (device/lib/seal_embedded.h:91-130), linked against libseal_embedded_amd.so instead of
 * libseal_embedded.a.  It encrypts one plaintext through se_setup / se_encrypt_seeded and prints
 * the FNV-1a-64 digest of the byte stream handed to the send callback (c0_j, c1_j per prime).
 *
 *   gcc examples/api_digest.c -Iinclude -Lseal-embedded_amd/lib -lseal_embedded_amd \
 *       -Wl,-rpath,$PWD/seal-embedded_amd/lib -o api_digest
 *   SE_AMD_REFERENCE_C1_ALIAS=1 ./api_digest 4096 3 sym     # run where adapter_output_data/ lives
 *
 * With SE_AMD_REFERENCE_C1_ALIAS=1 the digest equals the one the compiled reference produces for
 * the same input and seeds (tests/golden/golden_digests.json, "api_fnv1a64_*").
 * Input: values[i] = ((i * 2654435761) mod 100000) / 1000 - 50, share_seed[k] = k,
 * seed[k] = 255 - k; key files are read from adapter_output_data/ (or $SE_AMD_DATA_PATH).
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "seal_embedded_amd.h"

static uint64_t g_hash = 0xcbf29ce484222325ull;
static size_t g_calls = 0, g_bytes = 0;

static size_t send_cb(void *data, size_t nbytes)
{
    const uint8_t *p = (const uint8_t *)data;
    for (size_t i = 0; i < nbytes; i++)
    {
        g_hash ^= p[i];
        g_hash *= 0x100000001b3ull;
    }
    g_calls++;
    g_bytes += nbytes;
    return nbytes;
}

int main(int argc, char **argv)
{
    size_t n        = argc > 1 ? (size_t)atol(argv[1]) : 4096;
    size_t nprimes  = argc > 2 ? (size_t)atol(argv[2]) : 3;
    int asym        = argc > 3 && strcmp(argv[3], "asym") == 0;
    SE_PARMS *parms = se_setup(n, nprimes, 0.0 /* chosen by the parameter set */,
                               asym ? SE_ASYM_ENCR : SE_SYM_ENCR);

    float *v = (float *)malloc(n / 2 * sizeof(float));
    for (size_t i = 0; i < n / 2; i++)
        v[i] = (float)((double)(((uint64_t)i * 2654435761ull) % 100000ull) / 1000 - 50);
    uint8_t share_seed[64], seed[64];
    for (int k = 0; k < 64; k++)
    {
        share_seed[k] = (uint8_t)k;
        seed[k]       = (uint8_t)(255 - k);
    }

    bool ok = se_encrypt_seeded(share_seed, seed, send_cb, v, n / 2 * sizeof(float), false, parms);
    printf("ok=%d callbacks=%zu bytes=%zu fnv1a64=%016llx\n", (int)ok, g_calls, g_bytes,
           (unsigned long long)g_hash);

    /* optional: latency of the single-ciphertext call (argv[4] = repetitions) */
    int reps = argc > 4 ? atoi(argv[4]) : 0;
    if (reps > 0)
    {
        struct timespec t0, t1;
        clock_gettime(CLOCK_MONOTONIC, &t0);
        for (int r = 0; r < reps; r++)
            ok = se_encrypt_seeded(share_seed, seed, send_cb, v, n / 2 * sizeof(float), false, parms) && ok;
        clock_gettime(CLOCK_MONOTONIC, &t1);
        double ms = 1e3 * (double)(t1.tv_sec - t0.tv_sec) + 1e-6 * (double)(t1.tv_nsec - t0.tv_nsec);
        printf("latency_ms_per_call=%.3f over %d calls\n", ms / reps, reps);
    }
    se_cleanup(parms);
    free(v);
    return ok ? 0 : 1;
}
