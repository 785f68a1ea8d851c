/*
 * lower_asym_caller.c -- the public-key counterpart of lower_sym_caller.c, in the shape of
 * device/test/ckks_tests_asym.c:120-208: gen_pk per prime from fixed seeds (the reference's own
 * key-generation helper), then ckks_encode_base -> ckks_asym_init -> per prime
 * ckks_encode_encrypt_asym / ckks_next_prime_asym, through the reference's LOWER interface only.
 *
 *   gcc tests/c/lower_asym_caller.c -Iinclude -Iinclude/compat -Lseal-embedded_amd/lib \
 *       -lseal_embedded_amd -Wl,-rpath,$PWD/seal-embedded_amd/lib -o lower_asym_caller
 *   ./lower_asym_caller 4096 3 out.bin pk_seed.bin ep_seed.bin   # CWD: adapter_output_data/sk_<n>.dat
 *
 * Output file: pk0[np][n], pk1[np][n], int64 pte[n] (m + e0), u packed [n/4], e1[n] int8, then per
 * prime c0[n], c1[n], ntt_u_save[n], ntt_e1_save[n], ntt(m+e0)[n]; then the PRNG counter after init.
 */
#include <complex.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ckks_asym.h"
#include "ckks_common.h"
#include "ckks_sym.h"
#include "defines.h"
#include "fileops.h"
#include "parameters.h"
#include "rng.h"
#include "sample.h"

static void read_seed(const char *path, uint8_t *seed)
{
    FILE *f = fopen(path, "rb");
    if (!f || fread(seed, 1, SE_PRNG_SEED_BYTE_COUNT, f) != SE_PRNG_SEED_BYTE_COUNT) exit(9);
    fclose(f);
}

int main(int argc, char **argv)
{
    if (argc < 6) return 1;
    size_t n       = (size_t)atol(argv[1]);
    size_t nprimes = (size_t)atol(argv[2]);
    FILE *out      = fopen(argv[3], "wb");
    uint8_t pk_seed[SE_PRNG_SEED_BYTE_COUNT], ep_seed[SE_PRNG_SEED_BYTE_COUNT];
    read_seed(argv[4], pk_seed);
    read_seed(argv[5], ep_seed);
    if (!out) return 2;

    Parms parms;
    memset(&parms, 0, sizeof(parms));
    parms.is_asymmetric = true;
    parms.pk_from_file  = false;
    parms.sample_s      = false;
    parms.small_s       = true;
    parms.small_u       = true;

    ZZ *mempool = ckks_mempool_setup_asym(n);
    SE_PTRS p;
    ckks_set_ptrs_asym(n, mempool, &p);
    ckks_setup(n, nprimes, p.index_map_ptr, &parms);

    /* ---- key generation (ckks_tests_asym.c:174-208): ep once, gen_pk per prime ---- */
    ZZ *sk        = calloc(n / 16, sizeof(ZZ));
    ZZ *pk0       = calloc(nprimes * n, sizeof(ZZ));
    ZZ *pk1       = calloc(nprimes * n, sizeof(ZZ));
    ZZ *s_save    = calloc(n, sizeof(ZZ));
    ZZ *ntt_ep    = calloc(n, sizeof(ZZ));
    int8_t *ep    = calloc(n, 1);
    SE_PRNG prng, shareable_prng;
    load_sk(&parms, sk);
    prng_randomize_reset(&prng, ep_seed);
    sample_poly_cbd_generic_prng_16(n, &prng, ep);
    ckks_reset_primes(&parms);
    for (size_t j = 0; j < nprimes; j++)
    {
        uint8_t seedbuf[SE_PRNG_SEED_BYTE_COUNT];
        memcpy(seedbuf, pk_seed, sizeof(seedbuf));
        gen_pk(&parms, sk, p.ntt_roots_ptr, seedbuf, &shareable_prng, s_save, ep, ntt_ep, pk0 + j * n,
               pk1 + j * n);
        if (j + 1 < nprimes) next_modulus(&parms);
    }
    fwrite(pk0, sizeof(ZZ), nprimes * n, out);
    fwrite(pk1, sizeof(ZZ), nprimes * n, out);

    /* ---- encode + encrypt ---- */
    flpt *v     = p.values;
    size_t vlen = n / 2;
    for (size_t i = 0; i < vlen; i++)
        v[i] = (flpt)((double)(((uint64_t)i * 2654435761ull) % 100000ull) / 1000 - 50);
    uint8_t seed[SE_PRNG_SEED_BYTE_COUNT];
    for (int k = 0; k < SE_PRNG_SEED_BYTE_COUNT; k++) seed[k] = (uint8_t)(255 - k);

    ZZ *ntt_u_save  = calloc(n, sizeof(ZZ));
    ZZ *ntt_e1_save = calloc(n, sizeof(ZZ));
    ckks_reset_primes(&parms);
    if (!ckks_encode_base(&parms, v, vlen, p.index_map_ptr, p.ifft_roots, p.conj_vals)) return 3;
    ckks_asym_init(&parms, seed, &prng, p.conj_vals_int_ptr, p.ternary, p.e1_ptr);
    fwrite(p.conj_vals_int_ptr, sizeof(int64_t), n, out);
    fwrite(p.ternary, 1, n / 4, out);
    fwrite(p.e1_ptr, 1, n, out);
    for (size_t j = 0; j < parms.nprimes; j++)
    {
        memcpy(p.c0_ptr, pk0 + j * n, n * sizeof(ZZ));
        memcpy(p.c1_ptr, pk1 + j * n, n * sizeof(ZZ));
        ckks_encode_encrypt_asym(&parms, p.conj_vals_int_ptr, p.ternary, p.e1_ptr, p.ntt_roots_ptr,
                                 p.ntt_pte_ptr, ntt_u_save, ntt_e1_save, p.c0_ptr, p.c1_ptr);
        fwrite(p.c0_ptr, sizeof(ZZ), n, out);
        fwrite(p.c1_ptr, sizeof(ZZ), n, out);
        fwrite(ntt_u_save, sizeof(ZZ), n, out);
        fwrite(ntt_e1_save, sizeof(ZZ), n, out);
        fwrite(p.ntt_pte_ptr, sizeof(ZZ), n, out);
        bool more = ckks_next_prime_asym(&parms, p.ternary);
        if (more != (j + 1 < parms.nprimes)) return 6;
    }
    fwrite(&prng.counter, sizeof(uint64_t), 1, out);
    fclose(out);
    printf("lower_asym_caller: n=%zu primes=%zu ok, counter after init %llu\n", n, nprimes,
           (unsigned long long)prng.counter);
    return 0;
}
