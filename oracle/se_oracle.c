/*
 * se_oracle.c -- CPU restatement of SEAL-Embedded's device/lib encode+encrypt path.
 *
 * TEST INFRASTRUCTURE ONLY (see se_oracle.h).  Parity status: PINNED against the compiled
 * reference (oracle/_ref), the reference's own KATs and committed golden vectors.
 *
 * Each function cites the reference file:line (relative to /root/reference/) it restates.
 * This is a re-derivation in plain C99, not a copy: flat POD parameters, explicit counters,
 * no memory-pool aliasing, no compile-time configuration matrix.
 *
 * Build: gcc -O2 -ffp-contract=off -fPIC -shared (see oracle/Makefile).  No -ffast-math and
 * no -march=native: the FP64 encode must round exactly like the reference's x86-64 build.
 */
#define _GNU_SOURCE
#include "se_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------
 * Parameter tables.  Values are protocol constants.
 *   primes / scale rule : device/lib/parameters.c:129-230
 *   const_ratio pairs   : device/lib/modulus.c:23-56
 *   2n-th roots psi     : device/lib/ntt.c:199-291
 * ---------------------------------------------------------------------------------------- */
typedef struct
{
    uint32_t q, cr_hi, cr_lo;
} seo_modrow;

static const seo_modrow k_mod27[3] = {
    {134012929u, 0x20u, 0xc84dfe5u},
    {134111233u, 0x20u, 0x6814e43u},
    {134176769u, 0x20u, 0x2802e03u},
};

static const seo_modrow k_mod30[13] = {
    {1053818881u, 0x4u, 0x135bf4bau}, {1054015489u, 0x4u, 0x132a2218u},
    {1054212097u, 0x4u, 0x12f85437u}, {1055260673u, 0x4u, 0x11ef051eu},
    {1056178177u, 0x4u, 0x11074e88u}, {1056440321u, 0x4u, 0x10c52d4au},
    {1058209793u, 0x4u, 0xf07a84au},  {1060175873u, 0x4u, 0xd1a6142u},
    {1060700161u, 0x4u, 0xc9725e9u},  {1060765697u, 0x4u, 0xc86c0d4u},
    {1061093377u, 0x4u, 0xc34cf30u},  {1062469633u, 0x4u, 0xadd3267u},
    {1062535169u, 0x4u, 0xaccdb49u},
};

typedef struct
{
    uint32_t n, q, psi;
} seo_rootrow;

static const seo_rootrow k_roots[] = {
    {1024, 134012929u, 142143u},   {2048, 134012929u, 85250u},
    {4096, 134012929u, 7470u},     {4096, 134111233u, 3856u},
    {4096, 134176769u, 24149u},    {4096, 1053818881u, 503422u},
    {4096, 1054015489u, 16768u},   {4096, 1054212097u, 7305u},
    {8192, 1053818881u, 374229u},  {8192, 1054015489u, 123363u},
    {8192, 1054212097u, 79941u},   {8192, 1055260673u, 38869u},
    {8192, 1056178177u, 162146u},  {8192, 1056440321u, 81884u},
    {16384, 1053818881u, 13040u},  {16384, 1054015489u, 507u},
    {16384, 1054212097u, 1595u},   {16384, 1055260673u, 68507u},
    {16384, 1056178177u, 3073u},   {16384, 1056440321u, 6854u},
    {16384, 1058209793u, 44467u},  {16384, 1060175873u, 16117u},
    {16384, 1060700161u, 27607u},  {16384, 1060765697u, 222391u},
    {16384, 1061093377u, 105471u}, {16384, 1062469633u, 310222u},
    {16384, 1062535169u, 2005u},
};

int seo_params_init(seo_params *p, size_t n, size_t nprimes)
{
    memset(p, 0, sizeof(*p));
    const seo_modrow *tab;
    size_t maxp;
    double scale;
    switch (n)
    { /* parameters.c:190-226 */
        case 1024: tab = k_mod27; maxp = 1; scale = 1048576.0; break;  /* 2^20 */
        case 2048: tab = k_mod27; maxp = 1; scale = 33554432.0; break; /* 2^25 */
        case 4096: tab = k_mod30; maxp = 3; scale = 33554432.0; break;
        case 8192: tab = k_mod30; maxp = 6; scale = 33554432.0; break;
        case 16384: tab = k_mod30; maxp = 13; scale = 33554432.0; break;
        default: return -1;
    }
    if (nprimes < 1 || nprimes > maxp) return -2;
    p->n       = n;
    p->nprimes = nprimes;
    p->scale   = scale;
    size_t l   = 0;
    while (((size_t)1 << l) < n) l++;
    p->logn = l;
    for (size_t j = 0; j < nprimes; j++)
    {
        p->q[j]     = tab[j].q;
        p->cr_hi[j] = tab[j].cr_hi;
        p->cr_lo[j] = tab[j].cr_lo;
        p->psi[j]   = 0;
        for (size_t r = 0; r < sizeof(k_roots) / sizeof(k_roots[0]); r++)
            if (k_roots[r].n == n && k_roots[r].q == tab[j].q) p->psi[j] = k_roots[r].psi;
        if (!p->psi[j]) return -3;
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * Word arithmetic
 * ---------------------------------------------------------------------------------------- */

/* modulo.h:21-32 (shift_result): [0,2q) -> [0,q) */
static inline uint32_t seo_shift(uint32_t x, uint32_t q)
{
    return x >= q ? x - q : x;
}

/* modulo.h:43-75: x mod q through the high word of floor(2^64/q) */
uint32_t seo_barrett32(uint32_t x, const seo_params *p, size_t j)
{
    uint32_t est = (uint32_t)(((uint64_t)x * p->cr_hi[j]) >> 32);
    uint32_t r   = x - est * p->q[j];
    return seo_shift(r, p->q[j]);
}

/* modulo.h:84-116: 64-bit input, only the third word of x*floor(2^64/q) is formed */
uint32_t seo_barrett64(uint64_t x, const seo_params *p, size_t j)
{
    uint32_t x0 = (uint32_t)x, x1 = (uint32_t)(x >> 32);
    uint32_t c0 = p->cr_lo[j], c1 = p->cr_hi[j];

    uint32_t carry_in = (uint32_t)(((uint64_t)x0 * c0) >> 32);
    uint64_t mid      = (uint64_t)x0 * c1;
    uint64_t acc      = (uint64_t)carry_in + (uint32_t)mid; /* low word + carry-out */
    uint32_t mid_lo   = (uint32_t)acc;
    uint32_t mid_hi   = (uint32_t)(mid >> 32) + (uint32_t)(acc >> 32);

    uint64_t mid2    = (uint64_t)x1 * c0;
    uint64_t acc2    = (uint64_t)mid_lo + (uint32_t)mid2;
    uint32_t mid2_hi = (uint32_t)(mid2 >> 32) + (uint32_t)(acc2 >> 32);

    uint32_t quot = x1 * c1 + mid_hi + mid2_hi;
    uint32_t r    = x0 - quot * p->q[j];
    return seo_shift(r, p->q[j]);
}

/* uintmodarith.h:123-128 */
uint32_t seo_mul_mod(uint32_t a, uint32_t b, const seo_params *p, size_t j)
{
    return seo_barrett64((uint64_t)a * b, p, j);
}

/* uintmodarith.h:26-41 (requires a+b <= 2q-1) */
uint32_t seo_add_mod(uint32_t a, uint32_t b, uint32_t q)
{
    return seo_shift(a + b, q);
}

/* uintmodarith.h:62-71 */
uint32_t seo_neg_mod(uint32_t a, uint32_t q)
{
    return a ? q - a : 0;
}

/* uintmodarith.h:94-99 */
uint32_t seo_sub_mod(uint32_t a, uint32_t b, uint32_t q)
{
    return seo_add_mod(a, seo_neg_mod(b, q), q);
}

/* ------------------------------------------------------------------------------------------
 * SHAKE256 / Keccak-f[1600] (FIPS 202).  Reference: shake256/keccakf1600.c:51-316 (unrolled),
 * shake256/fips202.c:51-128.  Restated in the compact rho/pi-table form.
 * ---------------------------------------------------------------------------------------- */
static const uint64_t k_rc[24] = {
    0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL,
    0x000000000000808bULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
    0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
    0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
    0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
    0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
static const int k_rho[24] = {1,  3,  6,  10, 15, 21, 28, 36, 45, 55, 2,  14,
                              27, 41, 56, 8,  25, 43, 62, 18, 39, 61, 20, 44};
static const int k_pi[24]  = {10, 7,  11, 17, 18, 3, 5,  16, 8,  21, 24, 4,
                              15, 23, 19, 13, 12, 2, 20, 14, 22, 9,  6,  1};

static inline uint64_t rol64(uint64_t x, int s)
{
    return (x << s) | (x >> (64 - s));
}

void seo_keccak_f1600(uint64_t st[25])
{
    for (int round = 0; round < 24; round++)
    {
        uint64_t c[5], t;
        for (int x = 0; x < 5; x++) c[x] = st[x] ^ st[x + 5] ^ st[x + 10] ^ st[x + 15] ^ st[x + 20];
        for (int x = 0; x < 5; x++)
        {
            t = c[(x + 4) % 5] ^ rol64(c[(x + 1) % 5], 1);
            for (int y = 0; y < 25; y += 5) st[y + x] ^= t;
        }
        t = st[1];
        for (int i = 0; i < 24; i++)
        {
            int dst     = k_pi[i];
            uint64_t nx = st[dst];
            st[dst]     = rol64(t, k_rho[i]);
            t           = nx;
        }
        for (int y = 0; y < 25; y += 5)
        {
            for (int x = 0; x < 5; x++) c[x] = st[y + x];
            for (int x = 0; x < 5; x++) st[y + x] = c[x] ^ (~c[(x + 1) % 5] & c[(x + 2) % 5]);
        }
        st[0] ^= k_rc[round];
    }
}

#define SEO_RATE 136

static void st_xor_bytes(uint64_t st[25], const uint8_t *m, size_t len)
{
    for (size_t i = 0; i < len; i++) st[i >> 3] ^= (uint64_t)m[i] << (8 * (i & 7));
}

static void st_extract(const uint64_t st[25], uint8_t *out, size_t len)
{
    for (size_t i = 0; i < len; i++) out[i] = (uint8_t)(st[i >> 3] >> (8 * (i & 7)));
}

void seo_shake256(uint8_t *out, size_t outlen, const uint8_t *in, size_t inlen)
{
    uint64_t st[25];
    memset(st, 0, sizeof(st));
    while (inlen >= SEO_RATE)
    {
        st_xor_bytes(st, in, SEO_RATE);
        seo_keccak_f1600(st);
        in += SEO_RATE;
        inlen -= SEO_RATE;
    }
    uint8_t last[SEO_RATE];
    memset(last, 0, sizeof(last));
    memcpy(last, in, inlen);
    last[inlen] = 0x1F;
    last[SEO_RATE - 1] |= 0x80;
    st_xor_bytes(st, last, SEO_RATE);
    while (outlen)
    {
        seo_keccak_f1600(st);
        size_t take = outlen < SEO_RATE ? outlen : SEO_RATE;
        st_extract(st, out, take);
        out += take;
        outlen -= take;
    }
}

/* rng.h:78-91: block(ctr) = SHAKE256(seed || ctr as 8 little-endian bytes) */
void seo_prng_block(const uint8_t seed[SEO_SEED_BYTES], uint64_t ctr, uint8_t *out, size_t outlen)
{
    uint8_t msg[SEO_SEED_BYTES + 8];
    memcpy(msg, seed, SEO_SEED_BYTES);
    for (int i = 0; i < 8; i++) msg[SEO_SEED_BYTES + i] = (uint8_t)(ctr >> (8 * i));
    seo_shake256(out, outlen, msg, sizeof(msg));
}

/* ------------------------------------------------------------------------------------------
 * CKKS encode
 * ---------------------------------------------------------------------------------------- */

/* fft.h:48-55 */
size_t seo_bitrev(size_t x, size_t nbits)
{
    size_t r = 0;
    for (size_t b = 0; b < nbits; b++) r |= ((x >> b) & 1) << (nbits - 1 - b);
    return r;
}

/* ckks_common.c:32-68: orbit of 3 in Z_2n^*, both conjugate slots, merged with the bit reversal */
void seo_index_map(size_t n, size_t logn, uint16_t *map)
{
    uint64_t m = 2 * (uint64_t)n, pos = 1;
    for (size_t i = 0; i < n / 2; i++)
    {
        size_t i1      = (size_t)((pos - 1) / 2);
        size_t i2      = n - 1 - i1;
        map[i]         = (uint16_t)seo_bitrev(i1, logn);
        map[i + n / 2] = (uint16_t)seo_bitrev(i2, logn);
        pos            = (pos * 3) & (m - 1);
    }
}

/* fft.c:27-45 + :129: W[t] = conj(e^{2 pi i bitrev(t)/2n}) for t = h + j.
 * The expression order 2*M_PI*k/m is evaluated left to right exactly as calc_angle does. */
void seo_ifft_twiddles(size_t n, size_t logn, double *w)
{
    size_t m = n << 1;
    for (size_t t = 0; t < n; t++)
    {
        size_t k     = seo_bitrev(t, logn) & (m - 1);
        double angle = 2 * M_PI * (double)k / (double)(m);
        /* The reference writes cos(angle), sin(angle) (fft.c:43-44); its Release build (gcc -O3)
         * fuses them into one glibc sincos() call, whose results differ from sin()/cos() by 1 ulp
         * at a few angles.  The golden vectors come from that build, so sincos is explicit here
         * (independent of this file's optimisation level). */
        double sn, cs;
        sincos(angle, &sn, &cs);
        w[2 * t]     = cs;
        w[2 * t + 1] = -sn;
    }
}

/* one cached table per n (values identical to the reference's on-the-fly roots) */
static double *g_tw[5];
static pthread_mutex_t g_tw_lock = PTHREAD_MUTEX_INITIALIZER;
static const double *twiddles_for(size_t n, size_t logn)
{
    size_t slot = logn - 10;
    pthread_mutex_lock(&g_tw_lock);
    if (!g_tw[slot])
    {
        double *w = (double *)malloc(2 * n * sizeof(double));
        seo_ifft_twiddles(n, logn, w);
        g_tw[slot] = w;
    }
    pthread_mutex_unlock(&g_tw_lock);
    return g_tw[slot];
}

/* The reference multiplies `double complex` values with the C operator (fft.c:139, :205).  gcc (default
 * -fcx-... settings of the Release build: no -ffast-math) expands that into the Annex-G form
 *     x = ac - bd,  y = ad + bc          (one rounding per operation)
 * and, when x or y comes out NaN, calls libgcc's __muldc3, which recomputes x and y the same way and -- only
 * when BOTH are NaN -- "recovers infinities" (C99 G.5.1): an infinite factor is boxed to (+-1 | +-0), NaNs in the
 * other factor become signed zeros, and the product is INFINITY * (ac - bd), INFINITY * (ad + bc).  On finite
 * data none of this triggers; on non-finite plaintext values (NaN, +-Inf floats are legal inputs of
 * ckks_encode_base) it decides which coefficients are NaN (accepted, converted to INT64_MIN by x86's
 * cvttsd2si) and which are infinite (the overflow `return false` of ckks_common.c:195-204).
 * Restated from the published libgcc algorithm (libgcc2.c, __mulMODE3); pinned against the compiled
 * reference by tests/test_oracle.py::test_encode_nonfinite_matches_reference. */
static inline void seo_cmul(double a, double b, double c, double d, double *xo, double *yo)
{
    double ac = a * c, bd = b * d, ad = a * d, bc = b * c;
    double x = ac - bd, y = ad + bc;
    if (isnan(x) && isnan(y))
    {
        int recalc = 0;
        if (isinf(a) || isinf(b))
        {
            a = copysign(isinf(a) ? 1.0 : 0.0, a);
            b = copysign(isinf(b) ? 1.0 : 0.0, b);
            if (isnan(c)) c = copysign(0.0, c);
            if (isnan(d)) d = copysign(0.0, d);
            recalc = 1;
        }
        if (isinf(c) || isinf(d))
        {
            c = copysign(isinf(c) ? 1.0 : 0.0, c);
            d = copysign(isinf(d) ? 1.0 : 0.0, d);
            if (isnan(a)) a = copysign(0.0, a);
            if (isnan(b)) b = copysign(0.0, b);
            recalc = 1;
        }
        if (!recalc && (isinf(ac) || isinf(bd) || isinf(ad) || isinf(bc)))
        {
            if (isnan(a)) a = copysign(0.0, a);
            if (isnan(b)) b = copysign(0.0, b);
            if (isnan(c)) c = copysign(0.0, c);
            if (isnan(d)) d = copysign(0.0, d);
            recalc = 1;
        }
        if (recalc)
        {
            x = INFINITY * (a * c - b * d);
            y = INFINITY * (a * d + b * c);
        }
    }
    *xo = x;
    *yo = y;
}

/* fft.c:69-144: rounds tt = 1,2,..,n/2; (u,v) -> (u+v, (u-v)*s); no 1/n.
 * Complex product in C99 Annex-G operand order: (ac - bd, ad + bc), one rounding per op (seo_cmul). */
void seo_ifft_inpl(double *x, size_t n, size_t logn)
{
    const double *w = twiddles_for(n, logn);
    size_t tt = 1, h = n / 2;
    for (size_t r = 0; r < logn; r++, tt *= 2, h /= 2)
    {
        for (size_t j = 0, k0 = 0; j < h; j++, k0 += 2 * tt)
        {
            double c = w[2 * (h + j)], d = w[2 * (h + j) + 1];
            for (size_t k = k0; k < k0 + tt; k++)
            {
                double ur = x[2 * k], ui = x[2 * k + 1];
                double vr = x[2 * (k + tt)], vi = x[2 * (k + tt) + 1];
                double a = ur - vr, b = ui - vi;
                x[2 * k]            = ur + vr;
                x[2 * k + 1]        = ui + vi;
                seo_cmul(a, b, c, d, &x[2 * (k + tt)], &x[2 * (k + tt) + 1]);
            }
        }
    }
}

/* ckks_common.c:105-215.  Returns the index of the first coefficient that fails the overflow test
 * (:195-204; the reference returns false there, leaving out[0 .. index) converted), or n when none does.
 * A NaN coefficient does NOT fail (fabs(NaN) > 2^63 is false): the reference stores (int64_t)NaN, which
 * its x86-64 build evaluates with cvttsd2si to the "integer indefinite" INT64_MIN; so does +-2^63. */
size_t seo_encode_ex(const seo_params *p, const float *values, size_t values_len, const uint16_t *map,
                     int64_t *out)
{
    size_t n  = p->n;
    double *x = (double *)calloc(2 * n, sizeof(double));
    for (size_t i = 0; i < values_len; i++)
    {
        double v             = (double)values[i];
        x[2 * map[i]]        = v; /* imaginary parts stay 0 (:148-150) */
        x[2 * map[i + n / 2]] = v;
        x[2 * map[i] + 1]         = 0.0;
        x[2 * map[i + n / 2] + 1] = 0.0;
    }
    seo_ifft_inpl(x, n, p->logn);
    double n_inv = p->scale / (double)n; /* :183 */
    size_t i;
    for (i = 0; i < n; i++)
    {
        double coeff = round(x[2 * i] * n_inv);
        if (fabs(coeff) > 9223372036854775808.0) break; /* :195 MAX_INT_64_DOUBLE == 2^63 after conversion */
        /* (int64_t)coeff is undefined in C for NaN and 2^63; the reference's build yields INT64_MIN */
        out[i] = (coeff >= -9223372036854775808.0 && coeff < 9223372036854775808.0) ? (int64_t)coeff : INT64_MIN;
    }
    free(x);
    return i;
}

int seo_encode(const seo_params *p, const float *values, size_t values_len, const uint16_t *map,
               int64_t *out)
{
    return seo_encode_ex(p, values, values_len, map, out) == p->n;
}

/* ------------------------------------------------------------------------------------------
 * Samplers
 * ---------------------------------------------------------------------------------------- */

/* sample.c:263-284 */
static inline int popcnt8(unsigned v)
{
    return __builtin_popcount(v & 0xFFu);
}
static inline int cbd_from6(const uint8_t *x)
{
    return popcnt8(x[0]) + popcnt8(x[1]) + popcnt8(x[2] & 0x1F) - popcnt8(x[3]) - popcnt8(x[4]) -
           popcnt8(x[5] & 0x1F);
}

/* sample.c:347-356 */
void seo_cbd_add(int64_t *poly, size_t n, const uint8_t seed[64], uint64_t *ctr)
{
    uint8_t buf[96];
    for (size_t j = 0; j < n; j += 16)
    {
        seo_prng_block(seed, (*ctr)++, buf, 96);
        for (size_t i = 0; i < 16; i++) poly[j + i] += cbd_from6(buf + 6 * i);
    }
}

/* sample.c:311-321 */
void seo_cbd_int8(int8_t *poly, size_t n, const uint8_t seed[64], uint64_t *ctr)
{
    uint8_t buf[96];
    for (size_t j = 0; j < n; j += 16)
    {
        seo_prng_block(seed, (*ctr)++, buf, 96);
        for (size_t i = 0; i < 16; i++) poly[j + i] = (int8_t)cbd_from6(buf + 6 * i);
    }
}

/* sample.c:39-57: one 4n-byte block, then one 4-byte block per rejection draw, in index order */
void seo_sample_uniform(const seo_params *p, size_t j, const uint8_t seed[64], uint64_t *ctr,
                        uint32_t *poly)
{
    size_t n       = p->n;
    uint32_t bound = 0xFFFFFFFFu - seo_barrett32(0xFFFFFFFFu, p, j) - 1u;
    uint8_t *bytes = (uint8_t *)malloc(4 * n);
    seo_prng_block(seed, (*ctr)++, bytes, 4 * n);
    for (size_t i = 0; i < n; i++)
    {
        uint32_t x = (uint32_t)bytes[4 * i] | ((uint32_t)bytes[4 * i + 1] << 8) |
                     ((uint32_t)bytes[4 * i + 2] << 16) | ((uint32_t)bytes[4 * i + 3] << 24);
        while (x >= bound)
        {
            uint8_t w[4];
            seo_prng_block(seed, (*ctr)++, w, 4);
            x = (uint32_t)w[0] | ((uint32_t)w[1] << 8) | ((uint32_t)w[2] << 16) |
                ((uint32_t)w[3] << 24);
        }
        poly[i] = seo_barrett32(x, p, j);
    }
    free(bytes);
}

/* modulo.h:150-164 */
static inline uint8_t mod3_u8(uint8_t r)
{
    return (uint8_t)(r % 3u);
}

/* sample.c:218-242 + :61-87: 96 coefficients per 96-byte block, byte-level rejection at 0xFE,
 * 2-bit codes packed MSB-first within each byte */
void seo_sample_ternary_small(size_t n, const uint8_t seed[64], uint64_t *ctr, uint8_t *packed)
{
    memset(packed, 0, n / 4);
    for (size_t j = 0; j < n; j += 96)
    {
        uint8_t buf[96];
        seo_prng_block(seed, (*ctr)++, buf, 96);
        size_t stop = (j + 95 < n) ? 96 : (n - j);
        for (size_t i = 0; i < stop; i++)
        {
            uint8_t r = buf[i];
            while (r >= 0xFE) seo_prng_block(seed, (*ctr)++, &r, 1);
            size_t idx = i + j;
            packed[idx / 4] |= (uint8_t)(mod3_u8(r) << (6 - 2 * (idx % 4)));
        }
    }
}

/* sample.c:98-129: code 0 -> q-1, 1 -> 0, 2 -> 1 */
void seo_expand_ternary(const uint8_t *packed, size_t n, uint32_t q, uint32_t *out)
{
    for (size_t i = 0; i < n; i++)
    {
        uint32_t v = (packed[i / 4] >> (6 - 2 * (i % 4))) & 3u;
        out[i]     = v + (v == 0 ? q : 0) - 1u;
    }
}

/* ------------------------------------------------------------------------------------------
 * NTT
 * ---------------------------------------------------------------------------------------- */

/* ntt.c:40-52: roots[bitrev(i)] = psi^i */
void seo_ntt_roots(const seo_params *p, size_t j, uint32_t *roots)
{
    uint32_t power = p->psi[j];
    roots[0]       = 1;
    for (size_t i = 1; i < p->n; i++)
    {
        roots[seo_bitrev(i, p->logn)] = power;
        power                          = seo_mul_mod(power, p->psi[j], p, j);
    }
}

/* ntt.c:124-165: (u,v) -> (u + v*s, u - v*s), rounds h = 1..n/2, bit-reversed output */
void seo_ntt_inpl(const seo_params *p, size_t j, const uint32_t *roots, uint32_t *vec)
{
    size_t n   = p->n;
    uint32_t q = p->q[j];
    for (size_t h = 1, tt = n / 2; h < n; h *= 2, tt /= 2)
    {
        for (size_t g = 0, k0 = 0; g < h; g++, k0 += 2 * tt)
        {
            uint32_t s = roots[h + g];
            for (size_t k = k0; k < k0 + tt; k++)
            {
                uint32_t u  = vec[k];
                uint32_t v  = seo_mul_mod(vec[k + tt], s, p, j);
                vec[k]      = seo_add_mod(u, v, q);
                vec[k + tt] = seo_sub_mod(u, v, q);
            }
        }
    }
}

/* ckks_common.c:224-245: |x| mod q, then q - r for negative x (returns q, not 0, when r == 0) */
void seo_reduce_pte(const seo_params *p, size_t j, const int64_t *in, uint32_t *out)
{
    uint32_t q = p->q[j];
    for (size_t i = 0; i < p->n; i++)
    {
        int64_t x    = in[i];
        uint64_t mag = x < 0 ? (uint64_t)0 - (uint64_t)x : (uint64_t)x;
        uint32_t r   = seo_barrett64(mag, p, j);
        out[i]       = x < 0 ? q - r : r;
    }
}

/* ckks_common.c:259-265 */
void seo_reduce_e_small(const seo_params *p, size_t j, const int8_t *e, uint32_t *out)
{
    uint32_t q = p->q[j];
    for (size_t i = 0; i < p->n; i++) out[i] = (e[i] < 0 ? q : 0u) + (uint32_t)(int32_t)e[i];
}

/* ------------------------------------------------------------------------------------------
 * Whole path
 * ---------------------------------------------------------------------------------------- */

/* seal_embedded.c:98-215 (sym branch) -> ckks_sym.c:181-301 */
int seo_encrypt_sym(const seo_params *p, const uint16_t *map, const float *values,
                    size_t values_len, const uint8_t share_seed[64], const uint8_t seed[64],
                    const uint8_t *sk_packed, uint32_t *c0, uint32_t *c1, int64_t *pte_out,
                    uint32_t *ntt_pte_out, uint64_t *end_ctr)
{
    size_t n        = p->n;
    float *vals     = (float *)calloc(n / 2, sizeof(float));
    int64_t *pte    = (int64_t *)malloc(n * sizeof(int64_t));
    uint32_t *roots = (uint32_t *)malloc(n * sizeof(uint32_t));
    uint32_t *tmp   = (uint32_t *)malloc(n * sizeof(uint32_t));
    if (values_len > n / 2) values_len = n / 2;
    memcpy(vals, values, values_len * sizeof(float)); /* seal_embedded.c:108-111 (zero fill) */

    int ok = seo_encode(p, vals, n / 2, map, pte);
    if (ok)
    {
        uint64_t ectr = 0, actr = 0;
        seo_cbd_add(pte, n, seed, &ectr); /* ckks_sym.c:196 */
        if (pte_out) memcpy(pte_out, pte, n * sizeof(int64_t));
        for (size_t j = 0; j < p->nprimes; j++)
        {
            uint32_t *c0j = c0 + j * n, *c1j = c1 + j * n;
            uint32_t q = p->q[j];
            seo_sample_uniform(p, j, share_seed, &actr, c1j); /* :220 */
            seo_expand_ternary(sk_packed, n, q, c0j);         /* :255 */
            seo_ntt_roots(p, j, roots);                       /* :265 */
            seo_ntt_inpl(p, j, roots, c0j);                   /* :266 */
            for (size_t i = 0; i < n; i++)                    /* :273-277 */
                c0j[i] = seo_neg_mod(seo_mul_mod(c0j[i], c1j[i], p, j), q);
            seo_reduce_pte(p, j, pte, tmp);  /* :286 */
            seo_ntt_inpl(p, j, roots, tmp);  /* :292 */
            if (ntt_pte_out) memcpy(ntt_pte_out + j * n, tmp, n * sizeof(uint32_t));
            for (size_t i = 0; i < n; i++) c0j[i] = seo_add_mod(c0j[i], tmp[i], q); /* :300 */
        }
        if (end_ctr) *end_ctr = actr;
    }
    free(vals);
    free(pte);
    free(roots);
    free(tmp);
    return ok;
}

/* seal_embedded.c:98-215 (asym branch) -> ckks_asym.c:173-286 */
int seo_encrypt_asym(const seo_params *p, const uint16_t *map, const float *values,
                     size_t values_len, const uint8_t seed[64], const uint32_t *pk0,
                     const uint32_t *pk1, uint32_t *c0, uint32_t *c1, int64_t *pte_out,
                     uint8_t *u_out, int8_t *e1_out, uint64_t *end_ctr)
{
    size_t n        = p->n;
    float *vals     = (float *)calloc(n / 2, sizeof(float));
    int64_t *pte    = (int64_t *)malloc(n * sizeof(int64_t));
    uint32_t *roots = (uint32_t *)malloc(n * sizeof(uint32_t));
    uint32_t *tmp   = (uint32_t *)malloc(n * sizeof(uint32_t));
    uint8_t *u      = (uint8_t *)malloc(n / 4);
    int8_t *e1      = (int8_t *)malloc(n);
    if (values_len > n / 2) values_len = n / 2;
    memcpy(vals, values, values_len * sizeof(float));

    int ok = seo_encode(p, vals, n / 2, map, pte);
    if (ok)
    {
        uint64_t ctr = 0;
        seo_sample_ternary_small(n, seed, &ctr, u); /* ckks_asym.c:188 */
        seo_cbd_add(pte, n, seed, &ctr);            /* :200 */
        seo_cbd_int8(e1, n, seed, &ctr);            /* :201 */
        if (pte_out) memcpy(pte_out, pte, n * sizeof(int64_t));
        if (u_out) memcpy(u_out, u, n / 4);
        if (e1_out) memcpy(e1_out, e1, n);
        if (end_ctr) *end_ctr = ctr;
        for (size_t j = 0; j < p->nprimes; j++)
        {
            uint32_t *c0j = c0 + j * n, *c1j = c1 + j * n;
            uint32_t q = p->q[j];
            seo_expand_ternary(u, n, q, tmp); /* :235 */
            seo_ntt_roots(p, j, roots);
            seo_ntt_inpl(p, j, roots, tmp); /* :241 */
            for (size_t i = 0; i < n; i++)
            { /* :251-255 */
                c1j[i] = seo_mul_mod(pk1[j * n + i], tmp[i], p, j);
                c0j[i] = seo_mul_mod(pk0[j * n + i], tmp[i], p, j);
            }
            seo_reduce_e_small(p, j, e1, tmp); /* :263 */
            seo_ntt_inpl(p, j, roots, tmp);
            for (size_t i = 0; i < n; i++) c1j[i] = seo_add_mod(c1j[i], tmp[i], q); /* :272 */
            seo_reduce_pte(p, j, pte, tmp);                                         /* :280 */
            seo_ntt_inpl(p, j, roots, tmp);
            for (size_t i = 0; i < n; i++) c0j[i] = seo_add_mod(c0j[i], tmp[i], q); /* :284 */
        }
    }
    free(vals);
    free(pte);
    free(roots);
    free(tmp);
    free(u);
    free(e1);
    return ok;
}

/* ckks_asym.c:159-171 with conj_vals_int == NULL and ep_small given (ckks_sym.c:281-283) */
void seo_gen_pk(const seo_params *p, const uint8_t *sk_packed, const uint8_t pk_seed[64],
                const uint8_t ep_seed[64], uint32_t *pk0, uint32_t *pk1)
{
    size_t n        = p->n;
    int8_t *ep      = (int8_t *)malloc(n);
    uint32_t *roots = (uint32_t *)malloc(n * sizeof(uint32_t));
    uint32_t *tmp   = (uint32_t *)malloc(n * sizeof(uint32_t));
    uint64_t ectr   = 0;
    seo_cbd_int8(ep, n, ep_seed, &ectr);
    for (size_t j = 0; j < p->nprimes; j++)
    {
        uint32_t *p0 = pk0 + j * n, *p1 = pk1 + j * n;
        uint32_t q   = p->q[j];
        uint64_t ctr = 0; /* gen_pk re-seeds the shareable PRNG for every prime */
        seo_sample_uniform(p, j, pk_seed, &ctr, p1);
        seo_expand_ternary(sk_packed, n, q, p0);
        seo_ntt_roots(p, j, roots);
        seo_ntt_inpl(p, j, roots, p0);
        for (size_t i = 0; i < n; i++) p0[i] = seo_neg_mod(seo_mul_mod(p0[i], p1[i], p, j), q);
        seo_reduce_e_small(p, j, ep, tmp);
        seo_ntt_inpl(p, j, roots, tmp);
        for (size_t i = 0; i < n; i++) p0[i] = seo_add_mod(p0[i], tmp[i], q);
    }
    free(ep);
    free(roots);
    free(tmp);
}

/* ------------------------------------------------------------------------------------------
 * Verification side: inverse NTT, forward FFT, pseudo-decrypt, decode
 * ---------------------------------------------------------------------------------------- */
static uint32_t pow_mod(uint32_t base, uint64_t e, const seo_params *p, size_t j)
{
    uint32_t r = 1;
    while (e)
    {
        if (e & 1) r = seo_mul_mod(r, base, p, j);
        base = seo_mul_mod(base, base, p, j);
        e >>= 1;
    }
    return r;
}

/* intt.c:144-222: Gentleman-Sande rounds tt = 1,2,..,n/4 with s = psi^-bitrev(h+j); the last
 * round is merged with the 1/n scaling: (u+v)*inv_n, (u-v)*last_inv_sn. */
void seo_intt_inpl(const seo_params *p, size_t j, uint32_t *vec)
{
    size_t n = p->n, logn = p->logn;
    uint32_t q        = p->q[j];
    uint32_t inv_psi  = pow_mod(p->psi[j], (uint64_t)q - 2, p, j);
    uint32_t inv_n    = pow_mod((uint32_t)(n % q), (uint64_t)q - 2, p, j);
    size_t tt = 1, h = n / 2;
    for (size_t r = 0; r + 1 < logn; r++, tt *= 2, h /= 2)
    {
        for (size_t g = 0, k0 = 0; g < h; g++, k0 += 2 * tt)
        {
            uint32_t s = pow_mod(inv_psi, seo_bitrev(h + g, logn), p, j);
            for (size_t k = k0; k < k0 + tt; k++)
            {
                uint32_t u = vec[k], v = vec[k + tt];
                vec[k]      = seo_add_mod(u, v, q);
                vec[k + tt] = seo_mul_mod(seo_sub_mod(u, v, q), s, p, j);
            }
        }
    }
    /* h == 1 here: s_last = psi^-bitrev(1) = psi^-(n/2); last_inv_sn = s_last * inv_n */
    uint32_t last_inv_sn = seo_mul_mod(pow_mod(inv_psi, n / 2, p, j), inv_n, p, j);
    for (size_t i = 0; i < n / 2; i++)
    {
        uint32_t u = vec[i], v = vec[i + n / 2];
        vec[i]         = seo_mul_mod(seo_add_mod(u, v, q), inv_n, p, j);
        vec[i + n / 2] = seo_mul_mod(seo_sub_mod(u, v, q), last_inv_sn, p, j);
    }
}

/* fft.c:146-213; roots = conj of the cached IFFT table (cos, +sin) */
void seo_fft_inpl(double *x, size_t n, size_t logn)
{
    const double *w = twiddles_for(n, logn);
    for (size_t h = 1, tt = n / 2; h < n; h *= 2, tt /= 2)
    {
        for (size_t g = 0, k0 = 0; g < h; g++, k0 += 2 * tt)
        {
            double c = w[2 * (h + g)], d = -w[2 * (h + g) + 1];
            for (size_t k = k0; k < k0 + tt; k++)
            {
                double ur = x[2 * k], ui = x[2 * k + 1];
                double a = x[2 * (k + tt)], b = x[2 * (k + tt) + 1];
                double vr, vi;
                seo_cmul(a, b, c, d, &vr, &vi);
                x[2 * k]            = ur + vr;
                x[2 * k + 1]        = ui + vi;
                x[2 * (k + tt)]     = ur - vr;
                x[2 * (k + tt) + 1] = ui - vi;
            }
        }
    }
}

/* device/test/ckks_tests_common.c:136-153 */
void seo_decrypt(const seo_params *p, size_t j, const uint32_t *c0, const uint32_t *c1,
                 const uint32_t *ntt_s, uint32_t *out)
{
    for (size_t i = 0; i < p->n; i++)
        out[i] = seo_add_mod(seo_mul_mod(c1[i], ntt_s[i], p, j), c0[i], p->q[j]);
}

/* device/test/ckks_tests_common.c:59-115 */
void seo_decode(const seo_params *p, size_t j, const uint16_t *map, const uint32_t *pt,
                size_t values_len, float *out)
{
    size_t n   = p->n;
    uint32_t q = p->q[j];
    double *x  = (double *)calloc(2 * n, sizeof(double));
    for (size_t i = 0; i < n; i++)
    {
        uint32_t val = pt[i];
        double dval  = (val > q / 2) ? -(double)(q - val) : (double)val;
        x[2 * i]     = dval / p->scale;
    }
    seo_fft_inpl(x, n, p->logn);
    for (size_t i = 0; i < values_len; i++) out[i] = (float)x[2 * map[i]];
    free(x);
}

/* ------------------------------------------------------------------------------------------
 * Batched driver for the timed CPU baseline: contiguous shards, one pthread each.
 * ---------------------------------------------------------------------------------------- */
typedef struct
{
    const seo_params *p;
    const uint16_t *map;
    const float *values;
    const uint8_t *share_seeds, *seeds, *sk;
    uint32_t *c0, *c1;
    size_t lo, hi;
    int ok;
} seo_job;

static void *seo_worker(void *arg)
{
    seo_job *jb   = (seo_job *)arg;
    size_t n      = jb->p->n, np = jb->p->nprimes;
    uint32_t *s0  = (uint32_t *)malloc(np * n * sizeof(uint32_t));
    uint32_t *s1  = (uint32_t *)malloc(np * n * sizeof(uint32_t));
    jb->ok        = 1;
    for (size_t b = jb->lo; b < jb->hi; b++)
    {
        uint32_t *o0 = jb->c0 ? jb->c0 + b * np * n : s0;
        uint32_t *o1 = jb->c1 ? jb->c1 + b * np * n : s1;
        jb->ok &= seo_encrypt_sym(jb->p, jb->map, jb->values + b * (n / 2), n / 2,
                                  jb->share_seeds + 64 * b, jb->seeds + 64 * b, jb->sk, o0, o1,
                                  NULL, NULL, NULL);
    }
    free(s0);
    free(s1);
    return NULL;
}

int seo_encrypt_sym_batch(const seo_params *p, const float *values, size_t B,
                          const uint8_t *share_seeds, const uint8_t *seeds,
                          const uint8_t *sk_packed, uint32_t *c0, uint32_t *c1, int nthreads)
{
    if (nthreads < 1) nthreads = 1;
    uint16_t *map = (uint16_t *)malloc(p->n * sizeof(uint16_t));
    seo_index_map(p->n, p->logn, map);
    (void)twiddles_for(p->n, p->logn);
    pthread_t *th = (pthread_t *)malloc(nthreads * sizeof(pthread_t));
    seo_job *jobs = (seo_job *)malloc(nthreads * sizeof(seo_job));
    for (int t = 0; t < nthreads; t++)
    {
        seo_job jb = {p, map, values, share_seeds, seeds, sk_packed, c0, c1,
                      B * t / nthreads, B * (t + 1) / nthreads, 1};
        jobs[t]    = jb;
        pthread_create(&th[t], NULL, seo_worker, &jobs[t]);
    }
    int ok = 1;
    for (int t = 0; t < nthreads; t++)
    {
        pthread_join(th[t], NULL);
        ok &= jobs[t].ok;
    }
    free(th);
    free(jobs);
    free(map);
    return ok;
}

/* Public-key and encode-only counterparts (BASELINE configs 3 and 5): same sharding.  Region per
 * unit = bench_asym.c's (encode + ckks_asym_init + per-prime encrypt, pk resident) resp. encode +
 * per-prime reduce_set_pte + ntt_inpl. */
typedef struct
{
    const seo_params *p;
    const uint16_t *map;
    const float *values;
    const uint8_t *seeds;
    const uint32_t *pk0, *pk1;
    uint32_t *c0, *c1;
    size_t lo, hi;
    int mode; /* 1 asym, 2 encode + ntt */
    int ok;
} seo_job2;

static void *seo_worker2(void *arg)
{
    seo_job2 *jb  = (seo_job2 *)arg;
    size_t n      = jb->p->n, np = jb->p->nprimes;
    uint32_t *s0  = (uint32_t *)malloc(np * n * sizeof(uint32_t));
    uint32_t *s1  = (uint32_t *)malloc(np * n * sizeof(uint32_t));
    int64_t *m    = (int64_t *)malloc(n * sizeof(int64_t));
    uint32_t *rts = (uint32_t *)malloc(np * n * sizeof(uint32_t));
    for (size_t j = 0; j < np; j++) seo_ntt_roots(jb->p, j, rts + j * n);
    jb->ok = 1;
    for (size_t b = jb->lo; b < jb->hi; b++)
    {
        uint32_t *o0 = jb->c0 ? jb->c0 + b * np * n : s0;
        uint32_t *o1 = jb->c1 ? jb->c1 + b * np * n : s1;
        if (jb->mode == 1)
            jb->ok &= seo_encrypt_asym(jb->p, jb->map, jb->values + b * (n / 2), n / 2, jb->seeds + 64 * b,
                                       jb->pk0, jb->pk1, o0, o1, NULL, NULL, NULL, NULL);
        else
        {
            jb->ok &= seo_encode(jb->p, jb->values + b * (n / 2), n / 2, jb->map, m);
            for (size_t j = 0; j < np; j++)
            {
                seo_reduce_pte(jb->p, j, m, o0 + j * n);
                seo_ntt_inpl(jb->p, j, rts + j * n, o0 + j * n);
            }
        }
    }
    free(s0);
    free(s1);
    free(m);
    free(rts);
    return NULL;
}

static int seo_batch2(const seo_params *p, int mode, const float *values, size_t B, const uint8_t *seeds,
                      const uint32_t *pk0, const uint32_t *pk1, uint32_t *c0, uint32_t *c1, int nthreads)
{
    if (nthreads < 1) nthreads = 1;
    uint16_t *map = (uint16_t *)malloc(p->n * sizeof(uint16_t));
    seo_index_map(p->n, p->logn, map);
    (void)twiddles_for(p->n, p->logn);
    pthread_t *th  = (pthread_t *)malloc(nthreads * sizeof(pthread_t));
    seo_job2 *jobs = (seo_job2 *)malloc(nthreads * sizeof(seo_job2));
    for (int t = 0; t < nthreads; t++)
    {
        seo_job2 jb = {p, map, values, seeds, pk0, pk1, c0, c1, B * t / nthreads, B * (t + 1) / nthreads, mode, 1};
        jobs[t]     = jb;
        pthread_create(&th[t], NULL, seo_worker2, &jobs[t]);
    }
    int ok = 1;
    for (int t = 0; t < nthreads; t++)
    {
        pthread_join(th[t], NULL);
        ok &= jobs[t].ok;
    }
    free(th);
    free(jobs);
    free(map);
    return ok;
}

int seo_encrypt_asym_batch(const seo_params *p, const float *values, size_t B, const uint8_t *seeds,
                           const uint32_t *pk0, const uint32_t *pk1, uint32_t *c0, uint32_t *c1,
                           int nthreads)
{
    return seo_batch2(p, 1, values, B, seeds, pk0, pk1, c0, c1, nthreads);
}

int seo_encode_ntt_batch(const seo_params *p, const float *values, size_t B, uint32_t *out, int nthreads)
{
    return seo_batch2(p, 2, values, B, NULL, NULL, NULL, out, NULL, nthreads);
}

/* FNV-1a 64 (digests of callback byte streams) */
uint64_t seo_fnv1a64(const void *data, size_t len, uint64_t h)
{
    const uint8_t *b = (const uint8_t *)data;
    if (!h) h = 0xcbf29ce484222325ULL;
    for (size_t i = 0; i < len; i++)
    {
        h ^= b[i];
        h *= 0x100000001b3ULL;
    }
    return h;
}
