// se_host_tables.cpp -- parameter sets and setup-time tables (host side of the context).
//
// Restates, as flat tables:
//   default parameter sets / scale rule   /root/reference/device/lib/parameters.c:129-230
//   floor(2^64/q)                         /root/reference/device/lib/modulus.c:23-56
//   2n-th roots of unity per (n, q)       /root/reference/device/lib/ntt.c:199-291
//   index map                             /root/reference/device/lib/ckks_common.c:32-68
//   IFFT roots (host libm cos/sin)        /root/reference/device/lib/fft.c:27-45, :129
//   one-shot NTT root powers              /root/reference/device/lib/ntt.c:40-52
// The prime / ratio / root VALUES are constants of the protocol (a ciphertext produced under a
// different prime chain or root would not decrypt on the SEAL side).
#include "se_host_tables.h"

#include <math.h>
#include <string.h>

namespace seamd {

namespace {
struct PrimeRow
{
    uint32_t q;
    uint32_t psi1k, psi2k, psi4k, psi8k, psi16k;  // 0 = not tabulated for that degree
};
// 27-bit chain (n = 1024, 2048) and 30-bit chain (n >= 4096), in modulus-switching order.
const PrimeRow k27[] = {
    {134012929u, 142143u, 85250u, 7470u, 0, 0},
    {134111233u, 0, 0, 3856u, 0, 0},
    {134176769u, 0, 0, 24149u, 0, 0},
};
const PrimeRow k30[] = {
    {1053818881u, 0, 0, 503422u, 374229u, 13040u}, {1054015489u, 0, 0, 16768u, 123363u, 507u},
    {1054212097u, 0, 0, 7305u, 79941u, 1595u},     {1055260673u, 0, 0, 0, 38869u, 68507u},
    {1056178177u, 0, 0, 0, 162146u, 3073u},        {1056440321u, 0, 0, 0, 81884u, 6854u},
    {1058209793u, 0, 0, 0, 0, 44467u},             {1060175873u, 0, 0, 0, 0, 16117u},
    {1060700161u, 0, 0, 0, 0, 27607u},             {1060765697u, 0, 0, 0, 0, 222391u},
    {1061093377u, 0, 0, 0, 0, 105471u},            {1062469633u, 0, 0, 0, 0, 310222u},
    {1062535169u, 0, 0, 0, 0, 2005u},
};
uint32_t psi_of(const PrimeRow &r, size_t n)
{
    switch (n)
    {
        case 1024: return r.psi1k;
        case 2048: return r.psi2k;
        case 4096: return r.psi4k;
        case 8192: return r.psi8k;
        case 16384: return r.psi16k;
        default: return 0;
    }
}
}  // namespace

bool host_known_prime(uint32_t q)
{
    for (const PrimeRow &r : k27)
        if (r.q == q) return true;
    for (const PrimeRow &r : k30)
        if (r.q == q) return true;
    return false;
}

int host_params_init(HostParams &hp, size_t n, size_t nprimes)
{
    hp = HostParams();
    const PrimeRow *chain;
    size_t max_primes;
    double scale;
    switch (n)
    {
        case 1024: chain = k27; max_primes = 1; scale = ldexp(1.0, 20); break;
        case 2048: chain = k27; max_primes = 1; scale = ldexp(1.0, 25); break;
        case 4096: chain = k30; max_primes = 3; scale = ldexp(1.0, 25); break;
        case 8192: chain = k30; max_primes = 6; scale = ldexp(1.0, 25); break;
        case 16384: chain = k30; max_primes = 13; scale = ldexp(1.0, 25); break;
        default: return -1;
    }
    if (nprimes < 1 || nprimes > max_primes) return -2;
    hp.n       = n;
    hp.nprimes = nprimes;
    hp.scale   = scale;
    while (((size_t)1 << hp.logn) < n) hp.logn++;
    for (size_t j = 0; j < nprimes; j++)
    {
        uint32_t q = chain[j].q;
        // floor(2^64 / q) == floor((2^64 - 1) / q) because q is odd and > 1
        uint64_t ratio = ~(uint64_t)0 / q;
        hp.q[j]        = q;
        hp.cr_hi[j]    = (uint32_t)(ratio >> 32);
        hp.cr_lo[j]    = (uint32_t)ratio;
        hp.psi[j]      = psi_of(chain[j], n);
        if (!hp.psi[j]) return -3;
    }
    return 0;
}

DevParams to_dev_params(const HostParams &hp)
{
    DevParams d;
    memset(&d, 0, sizeof(d));
    d.n       = (uint32_t)hp.n;
    d.logn    = (uint32_t)hp.logn;
    d.nprimes = (uint32_t)hp.nprimes;
    for (size_t j = 0; j < hp.nprimes; j++)
    {
        d.q[j]     = hp.q[j];
        d.cr_hi[j] = hp.cr_hi[j];
        d.cr_lo[j] = hp.cr_lo[j];
        // sample.c:45-46: max_multiple = 0xFFFFFFFF - (0xFFFFFFFF mod q) - 1
        d.bound[j] = 0xFFFFFFFFu - (0xFFFFFFFFu % hp.q[j]) - 1u;
    }
    {
        uint32_t qmin = hp.q[0];
        for (size_t j = 1; j < hp.nprimes; j++) qmin = hp.q[j] < qmin ? hp.q[j] : qmin;
        d.small_bound = 2.0 * (double)qmin - 64.0;   // < 2^31 for every tabulated prime
    }
    d.n_inv = hp.scale / (double)hp.n;
    d.scale = hp.scale;
    for (size_t j = 0; j < hp.nprimes; j++)
    {
        uint32_t inv  = host_inv_mod((uint32_t)(hp.n % hp.q[j]), hp.q[j]);
        d.inv_n[j]    = inv;
        d.inv_n_sh[j] = (uint32_t)(((uint64_t)inv << 32) / hp.q[j]);
    }
    return d;
}

size_t bitrev(size_t x, size_t nbits)
{
    size_t r = 0;
    for (size_t b = 0; b < nbits; b++) r |= ((x >> b) & 1) << (nbits - 1 - b);
    return r;
}

void host_index_map(const HostParams &hp, std::vector<uint16_t> &map, std::vector<uint16_t> &inv)
{
    const size_t n = hp.n;
    map.assign(n, 0);
    inv.assign(n, 0);
    uint64_t m = 2 * (uint64_t)n, pos = 1;
    for (size_t i = 0; i < n / 2; i++)
    {
        size_t i1      = (size_t)((pos - 1) / 2);
        size_t i2      = n - 1 - i1;
        map[i]         = (uint16_t)bitrev(i1, hp.logn);
        map[i + n / 2] = (uint16_t)bitrev(i2, hp.logn);
        pos            = (pos * 3) & (m - 1);
    }
    for (size_t i = 0; i < n; i++) inv[map[i]] = (uint16_t)i;
}

void host_ifft_twiddles(const HostParams &hp, std::vector<double> &w)
{
    const size_t n = hp.n, m = n << 1;
    w.assign(2 * n, 0.0);
    for (size_t t = 0; t < n; t++)
    {
        size_t k     = bitrev(t, hp.logn) & (m - 1);
        double angle = 2 * M_PI * (double)k / (double)(m);  // left-to-right, as fft.c:29
        // fft.c:43-44 calls cos(angle) and sin(angle); gcc -O1+ (the reference's Release build, and
        // the build the golden vectors come from) fuses the pair into ONE glibc sincos() call, and
        // sincos() differs from sin()/cos() by 1 ulp at a few angles (2 of 8192 table entries at
        // n = 4096, 23 at n = 16384).  clang does not fuse, so the call is explicit here.
        double sn, cs;
        sincos(angle, &sn, &cs);
        w[2 * t]     = cs;
        w[2 * t + 1] = -sn;                                  // conjugate (fft.c:129)
    }
}

void host_ntt_root_pairs(const HostParams &hp, size_t j, std::vector<uint32_t> &rw)
{
    const size_t n   = hp.n;
    const uint64_t q = hp.q[j];
    rw.assign(2 * n, 0);
    uint64_t power = 1;
    for (size_t i = 0; i < n; i++)
    {
        size_t slot      = bitrev(i, hp.logn);
        rw[2 * slot]     = (uint32_t)power;
        rw[2 * slot + 1] = (uint32_t)((power << 32) / q);  // Shoup companion floor(w 2^32 / q)
        power            = power * hp.psi[j] % q;
    }
}

// a^-1 mod q for prime q (Fermat)
uint32_t host_inv_mod(uint32_t a, uint32_t q)
{
    uint64_t r = 1, b = a % q, e = (uint64_t)q - 2;
    while (e)
    {
        if (e & 1) r = r * b % q;
        b = b * b % q;
        e >>= 1;
    }
    return (uint32_t)r;
}

// intt.c:26-58 (one-shot form) / :160-177 (on-the-fly form): the root used by group g of the round
// with h groups is psi^-bitrev(h + g); stored at index h + g like the forward table.
void host_intt_root_pairs(const HostParams &hp, size_t j, std::vector<uint32_t> &rw)
{
    const size_t n         = hp.n;
    const uint64_t q       = hp.q[j];
    const uint64_t inv_psi = host_inv_mod(hp.psi[j], hp.q[j]);
    std::vector<uint32_t> pw(n);
    uint64_t power = 1;
    for (size_t i = 0; i < n; i++)
    {
        pw[i] = (uint32_t)power;
        power = power * inv_psi % q;
    }
    rw.assign(2 * n, 0);
    for (size_t idx = 0; idx < n; idx++)
    {
        uint64_t w      = pw[bitrev(idx, hp.logn)];
        rw[2 * idx]     = (uint32_t)w;
        rw[2 * idx + 1] = (uint32_t)((w << 32) / q);
    }
}

}  // namespace seamd
