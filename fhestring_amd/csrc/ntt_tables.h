// Host-side exact construction of the device NTT tables and of the NTT-domain bootstrapping key.
#pragma once
#include <cstdint>
#include <vector>

namespace fhs {

struct HostNttTables {
    std::vector<double> fwd_uni;   // [2][32]
    std::vector<double> fwd_lane;  // [2][32][64]
    std::vector<double> inv_uni;   // [2][64]
    std::vector<double> inv_lane;  // [2][32][64]
    double crt_c;                  // p0^-1 mod p1, centred
};
void build_ntt_tables(HostNttTables &t);

// bsk_std: [742][2][2][2048] u64 (rounded to multiples of 2^6 inside).  out: [742][2][2][2 primes][32][64]
// doubles = forward NTT of (signed bsk / 2^6) mod p, pre-scaled by N^-1, centred, in the device's
// contiguous-layout order.  Runs on `nthreads` host threads (key loading, not the hot path).
void convert_bsk_to_ntt(const uint64_t *bsk_std, double *out, int nthreads);

}  // namespace fhs
