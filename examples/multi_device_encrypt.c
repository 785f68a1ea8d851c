/*
 * multi_device_encrypt.c -- the device-resident multi-GPU entry from plain C:
 * one group of contexts (one per HIP device of the node, or the ordinals given on the command line), the
 * batch cut into contiguous blocks with inputs and outputs resident on each block's device, and the final
 * gather of the records into the root device's slab by peer-to-peer copies.
 *
 * Prints the FNV-1a-64 digest of the GATHERED records in the reference's callback order (c0_j then c1_j
 * per prime, per ciphertext; seal_embedded.c:196-203) -- which must not depend on how many devices the
 * batch was cut over -- and the rate of the call with everything resident.
 *
 * The gathered slab is then CHECKED on the box: the root device encrypts the whole batch once more by itself
 * (one context, one device, no peer copy) and the two slabs are compared word for word (gather_verified=1);
 * `distinct_devices` says over how many different PCI devices the group really ran.
 *
 *   gcc examples/multi_device_encrypt.c -Iinclude -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ \
 *       -Lseal-embedded_amd/lib -lseal_embedded_amd -L/opt/rocm/lib -lamdhip64 \
 *       -Wl,-rpath,$PWD/seal-embedded_amd/lib -o multi_device_encrypt
 *   ./multi_device_encrypt 4096 3 8192 0,1,2,3,4,5,6,7        (no list: all visible devices)
 */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "seal_embedded_amd.h"

#define CHECK_HIP(call)                                                                  \
    do                                                                                   \
    {                                                                                    \
        hipError_t e_ = (call);                                                          \
        if (e_ != hipSuccess)                                                            \
        {                                                                                \
            fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_));                   \
            exit(2);                                                                     \
        }                                                                                \
    } while (0)
#define CHECK_SE(call)                                                                   \
    do                                                                                   \
    {                                                                                    \
        int rc_ = (call);                                                                \
        if (rc_ != SE_SUCCESS)                                                           \
        {                                                                                \
            fprintf(stderr, "%s: %d (%s)\n", #call, rc_, se_amd_last_error());           \
            exit(2);                                                                     \
        }                                                                                \
    } while (0)

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
    int devices[64];
    size_t ndev = 0;
    if (argc > 4)
        for (char *tok = strtok(argv[4], ","); tok && ndev < 64; tok = strtok(NULL, ",")) devices[ndev++] = atoi(tok);
    const char *sk_path = argc > 5 ? argv[5] : NULL;   /* sk_<n>.dat; default: all-zero codes = s of -1s */

    se_amd_group *g;
    CHECK_SE(se_amd_group_create(&g, n, nprimes, ndev ? devices : NULL, ndev));
    ndev = se_amd_group_size(g);
    uint8_t *sk = (uint8_t *)calloc(n / 4, 1);
    if (sk_path)
    {
        FILE *f = fopen(sk_path, "rb");
        if (!f || fread(sk, 1, n / 4, f) != n / 4)
        {
            fprintf(stderr, "cannot read %s\n", sk_path);
            return 2;
        }
        fclose(f);
    }
    CHECK_SE(se_amd_group_set_secret_key(g, sk));
    CHECK_SE(se_amd_group_reserve(g, B));

    /* the whole batch on the host, then block i onto device i */
    const size_t rec = nprimes * n;
    float *values    = (float *)malloc(B * (n / 2) * sizeof(float));
    uint8_t *share = (uint8_t *)malloc(B * 64), *seeds = (uint8_t *)malloc(B * 64);
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
    size_t first[64], count[64];
    CHECK_SE(se_amd_group_partition(g, B, first, count));
    const int root = 0;
    uint32_t *c0_all, *c1_all;
    CHECK_HIP(hipSetDevice(se_amd_group_device(g, root)));
    CHECK_HIP(hipMalloc((void **)&c0_all, B * rec * 4));
    CHECK_HIP(hipMalloc((void **)&c1_all, B * rec * 4));
    CHECK_HIP(hipMemset(c0_all, 0xEE, B * rec * 4));
    CHECK_HIP(hipMemset(c1_all, 0xEE, B * rec * 4));
    const float *d_values[64];
    const uint8_t *d_share[64], *d_seeds[64];
    uint32_t *d_c0[64], *d_c1[64];
    uint8_t *d_status[64];
    for (size_t i = 0; i < ndev; i++)
    {
        const size_t cnt = count[i] ? count[i] : 1;
        void *v, *s1, *s2, *st;
        CHECK_HIP(hipSetDevice(se_amd_group_device(g, i)));
        CHECK_HIP(hipMalloc(&v, cnt * (n / 2) * sizeof(float)));
        CHECK_HIP(hipMalloc(&s1, cnt * 64));
        CHECK_HIP(hipMalloc(&s2, cnt * 64));
        CHECK_HIP(hipMalloc(&st, cnt));
        CHECK_HIP(hipMemcpy(v, values + first[i] * (n / 2), count[i] * (n / 2) * sizeof(float), hipMemcpyHostToDevice));
        CHECK_HIP(hipMemcpy(s1, share + first[i] * 64, count[i] * 64, hipMemcpyHostToDevice));
        CHECK_HIP(hipMemcpy(s2, seeds + first[i] * 64, count[i] * 64, hipMemcpyHostToDevice));
        CHECK_HIP(hipMemset(st, 0, cnt));
        d_values[i] = (const float *)v, d_share[i] = (const uint8_t *)s1, d_seeds[i] = (const uint8_t *)s2;
        d_status[i] = (uint8_t *)st;
        if ((int)i == root)
        {
            /* the root produces its block in place inside the gathered slab: no copy for it */
            d_c0[i] = c0_all + first[i] * rec, d_c1[i] = c1_all + first[i] * rec;
        }
        else
        {
            CHECK_HIP(hipMalloc((void **)&d_c0[i], cnt * rec * 4));
            CHECK_HIP(hipMalloc((void **)&d_c1[i], cnt * rec * 4));
        }
    }

    /* warm-up (scratch, peer mappings), then the timed call */
    CHECK_SE(se_amd_encrypt_sym_multi_device(g, B, d_values, d_share, d_seeds, d_c0, d_c1, d_status, root, c0_all, c1_all));
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    CHECK_SE(se_amd_encrypt_sym_multi_device(g, B, d_values, d_share, d_seeds, d_c0, d_c1, d_status, root, c0_all, c1_all));
    clock_gettime(CLOCK_MONOTONIC, &t1);
    const double sec = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
    clock_gettime(CLOCK_MONOTONIC, &t0);
    CHECK_SE(se_amd_encrypt_sym_multi_device(g, B, d_values, d_share, d_seeds, d_c0, d_c1, d_status, -1, NULL, NULL));
    clock_gettime(CLOCK_MONOTONIC, &t1);
    const double sec_res = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);

    /* digest of the gathered slab on the root, record order = batch order */
    uint32_t *h0 = (uint32_t *)malloc(B * rec * 4), *h1 = (uint32_t *)malloc(B * rec * 4);
    CHECK_HIP(hipSetDevice(se_amd_group_device(g, root)));
    CHECK_HIP(hipMemcpy(h0, c0_all, B * rec * 4, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(h1, c1_all, B * rec * 4, hipMemcpyDeviceToHost));
    uint64_t h = 0xcbf29ce484222325ull;
    for (size_t b = 0; b < B; b++)
        for (size_t j = 0; j < nprimes; j++)
        {
            h = fnv(h, h0 + (b * nprimes + j) * n, n * 4);
            h = fnv(h, h1 + (b * nprimes + j) * n, n * 4);
        }
    /* ---- did the bytes arrive?  One single-device pass over the whole batch on the root, compared word for word */
    int gather_verified = 0;
    size_t first_bad    = (size_t)-1;
    {
        se_amd_ctx *solo;
        void *v, *s1, *s2;
        uint32_t *r0, *r1;
        const int rdev = se_amd_group_device(g, root);
        CHECK_HIP(hipSetDevice(rdev));
        CHECK_SE(se_amd_create(&solo, n, nprimes, rdev));
        CHECK_SE(se_amd_set_secret_key(solo, sk));
        CHECK_HIP(hipMalloc(&v, B * (n / 2) * sizeof(float)));
        CHECK_HIP(hipMalloc(&s1, B * 64));
        CHECK_HIP(hipMalloc(&s2, B * 64));
        CHECK_HIP(hipMalloc((void **)&r0, B * rec * 4));
        CHECK_HIP(hipMalloc((void **)&r1, B * rec * 4));
        CHECK_HIP(hipMemcpy(v, values, B * (n / 2) * sizeof(float), hipMemcpyHostToDevice));
        CHECK_HIP(hipMemcpy(s1, share, B * 64, hipMemcpyHostToDevice));
        CHECK_HIP(hipMemcpy(s2, seeds, B * 64, hipMemcpyHostToDevice));
        CHECK_SE(se_amd_encrypt_sym_device(solo, (const float *)v, B, (const uint8_t *)s1, (const uint8_t *)s2, r0, r1,
                                           NULL, NULL, NULL, NULL));
        CHECK_HIP(hipDeviceSynchronize());
        uint32_t *g0 = (uint32_t *)malloc(B * rec * 4), *g1 = (uint32_t *)malloc(B * rec * 4);
        CHECK_HIP(hipMemcpy(g0, r0, B * rec * 4, hipMemcpyDeviceToHost));
        CHECK_HIP(hipMemcpy(g1, r1, B * rec * 4, hipMemcpyDeviceToHost));
        gather_verified = 1;
        for (size_t w = 0; w < B * rec; w++)
            if (g0[w] != h0[w] || g1[w] != h1[w])
            {
                gather_verified = 0, first_bad = w / rec;
                break;
            }
        free(g0), free(g1);
        CHECK_HIP(hipFree(v));
        CHECK_HIP(hipFree(s1));
        CHECK_HIP(hipFree(s2));
        CHECK_HIP(hipFree(r0));
        CHECK_HIP(hipFree(r1));
        se_amd_destroy(solo);
    }
    /* how many different PCI devices did the group span? */
    size_t distinct = 0;
    char (*bus)[32] = (char (*)[32])calloc(ndev ? ndev : 1, 32);   /* the same device may be listed many times */
    for (size_t i = 0; i < ndev; i++)
    {
        char id[32] = "?";
        (void)hipDeviceGetPCIBusId(id, (int)sizeof id, se_amd_group_device(g, i));
        size_t k = 0;
        while (k < distinct && strcmp(bus[k], id)) k++;
        if (k == distinct) strcpy(bus[distinct++], id);
    }
    free(bus);
    int failed = 0;
    for (size_t i = 0; i < ndev; i++)
    {
        uint8_t *st = (uint8_t *)malloc(count[i] ? count[i] : 1);
        CHECK_HIP(hipSetDevice(se_amd_group_device(g, i)));
        CHECK_HIP(hipMemcpy(st, d_status[i], count[i], hipMemcpyDeviceToHost));
        for (size_t k = 0; k < count[i]; k++) failed += st[k] != 1;
        free(st);
    }
    printf("failed=%d B=%zu devices=%zu distinct_devices=%zu gather_verified=%d all=%016llx seconds_with_gather=%.4f "
           "ct_per_s_with_gather=%.0f seconds_resident=%.4f ct_per_s_resident=%.0f\n",
           failed, B, ndev, distinct, gather_verified, (unsigned long long)h, sec, (double)B / sec, sec_res,
           (double)B / sec_res);
    if (!gather_verified)
        fprintf(stderr, "gathered slab differs from the single-device pass at record %zu\n", first_bad);
    se_amd_group_destroy(g);
    return failed == 0 && gather_verified ? 0 : 1;
}
