// host_math.cc -- host-side checks of two pieces of integer arithmetic the device code relies on (tests/test_host_math.py):
//   * zipf_lookup (dint_amd/csrc/zipf_table.h): the coarse index brackets the binary search; the result must be the
//     plain binary search's for every 32-bit x -- the host and device drivers' request streams depend on it;
//   * the bin cut of the kv passes (k_kv.hip kv_cut_div): quotient and remainder by magic multiply, restated here
//     with the device's __umulhi spelled as a 64-bit product.
#include <stdint.h>

#include "../../dint_amd/csrc/zipf_table.h"

extern "C" {

// number of x among `xs` for which the indexed lookup differs from the plain binary search over the same thresholds
uint64_t zipf_mismatches(uint64_t n, double theta, const uint32_t *xs, uint64_t nx) {
  ZipfTable z;
  z.init(n, theta);
  const uint32_t *cdf = z.cdf.data();
  uint64_t bad = 0;
  for (uint64_t k = 0; k < nx; k++) {
    const uint32_t x = xs[k];
    uint64_t lo = 0, hi = z.n - 1;
    while (lo < hi) {
      const uint64_t mid = (lo + hi) >> 1;
      if (cdf[mid] > x) hi = mid; else lo = mid + 1;
    }
    const uint64_t want = (lo * 0x9E3779B97F4A7C15ull >> 11) % z.n;
    bad += zipf_lookup(cdf, z.n, x) != want;
  }
  return bad;
}

// kv_cut_div as the device computes it: q = umulhi(gk, floor(2^32 / P)), one correction step
uint64_t cut_mismatches(uint32_t P, const uint32_t *gks, uint64_t n) {
  const uint32_t magic = P > 1 ? (uint32_t)((1ull << 32) / P) : 0u;
  uint64_t bad = 0;
  for (uint64_t k = 0; k < n; k++) {
    const uint32_t gk = gks[k];
    uint32_t q, r;
    if (P <= 1) { q = gk; r = 0; }
    else {
      q = (uint32_t)(((uint64_t)gk * magic) >> 32);
      r = gk - q * P;
      if (r >= P) { r -= P; q++; }
    }
    bad += (q != gk / (P ? P : 1)) || (r != gk % (P ? P : 1)) || (P > 1 && r >= P);
  }
  return bad;
}
}
