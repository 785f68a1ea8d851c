// se_host_tables.h -- parameter sets and setup-time tables, computed on the host.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <vector>

#include "se_types.h"

namespace seamd {

struct HostParams
{
    size_t n = 0, logn = 0, nprimes = 0;
    uint32_t q[kMaxPrimes]     = {};
    uint32_t cr_hi[kMaxPrimes] = {};
    uint32_t cr_lo[kMaxPrimes] = {};
    uint32_t psi[kMaxPrimes]   = {};  // primitive 2n-th root of unity mod q
    double scale               = 0;
};

// 0 on success; negative if (n, nprimes) is not one of the reference's default parameter sets.
int host_params_init(HostParams &hp, size_t n, size_t nprimes);
DevParams to_dev_params(const HostParams &hp);
bool host_known_prime(uint32_t q);  // one of the tabulated 27-/30-bit primes (parameters.c:129-174)

size_t bitrev(size_t x, size_t nbits);
void host_index_map(const HostParams &hp, std::vector<uint16_t> &map, std::vector<uint16_t> &inv);
void host_ifft_twiddles(const HostParams &hp, std::vector<double> &w);            // [n][2]
void host_ntt_root_pairs(const HostParams &hp, size_t j, std::vector<uint32_t> &rw);  // [n][2]
void host_intt_root_pairs(const HostParams &hp, size_t j, std::vector<uint32_t> &rw); // [n][2]
uint32_t host_inv_mod(uint32_t a, uint32_t q);

}  // namespace seamd
