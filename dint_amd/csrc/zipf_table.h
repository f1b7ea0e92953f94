// zipf_table.h -- exact Zipf(theta) sampling by inverse CDF over integer thresholds.
//
// The closed-loop drivers (txn_driver.cc, fasst_client.cc on the host; k_txn.hip on the GPU) must draw the SAME key
// from the same 32-bit random word, bit for bit, so that a request stream generated on the device can be compared
// with the host's.  A closed form in floating point (pow) does not give that across libm and the device math
// library; a threshold table built once on the host and shared (uploaded) does:
//   cdf[k] = floor(2^32 * sum_{j <= k+1} j^-theta / zeta(n, theta)),  sample(x) = the smallest k with cdf[k] > x
// for a uniform 32-bit x.  Ranks are scattered over the key space with a multiplicative hash so that hot rows do
// not share buckets by construction (collisions of the scatter only merge two ranks).
#pragma once
#include <math.h>
#include <stdint.h>

#include <vector>

// The table carries a coarse index behind the thresholds (cdf[n + b] = sample(b << 16) before the scatter, b = 0 ..
// 65535, and cdf[n + 65536] = n - 1): a lookup brackets its binary search with two adjacent index words -- ~5 dependent
// loads over a 1M-row table instead of 20 (the GPU-resident clients draw a key per new transaction, a dependent chain
// per lane: 63 % of k_txn_emit's wave cycles are waits).  The bracket contains the answer, so the result is the plain
// binary search's.
#define ZIPF_IDX_BITS 16
#define ZIPF_IDX_N ((1u << ZIPF_IDX_BITS) + 1u)
struct ZipfTable {
  std::vector<uint32_t> cdf;  // n thresholds, then ZIPF_IDX_N index words
  uint64_t n = 0;
  void init(uint64_t n_, double theta) {
    n = n_ ? n_ : 1;
    cdf.assign(n + ZIPF_IDX_N, 0);
    double z = 0;
    for (uint64_t k = 1; k <= n; k++) z += pow((double)k, -theta);
    double run = 0;
    for (uint64_t k = 0; k < n; k++) {
      run += pow((double)(k + 1), -theta);
      const double c = run / z * 4294967296.0;
      cdf[k] = c >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)c;
    }
    cdf[n - 1] = 0xFFFFFFFFu;
    uint64_t k = 0;  // index: the smallest k with cdf[k] > b << (32 - ZIPF_IDX_BITS), for ascending b
    for (uint32_t b = 0; b < (1u << ZIPF_IDX_BITS); b++) {
      const uint32_t x = b << (32 - ZIPF_IDX_BITS);
      while (k < n - 1 && cdf[k] <= x) k++;
      cdf[n + b] = (uint32_t)k;
    }
    cdf[n + (1u << ZIPF_IDX_BITS)] = (uint32_t)(n - 1);
  }
};

// the lookup, shared by host and device code (cdf = table of n thresholds + the index)
#if defined(__HIPCC__)
__host__ __device__
#endif
static inline uint64_t zipf_lookup(const uint32_t *cdf, uint64_t n, uint32_t x) {
  const uint32_t b = x >> (32 - ZIPF_IDX_BITS);
  uint64_t lo = cdf[n + b], hi = cdf[n + b + 1];  // the answer is in [lo, hi]; cdf[n-1] = 2^32-1 catches x = 2^32-1 as well
  while (lo < hi) {
    const uint64_t mid = (lo + hi) >> 1;
    if (cdf[mid] > x) hi = mid; else lo = mid + 1;
  }
  return (lo * 0x9E3779B97F4A7C15ull >> 11) % n;
}
