// k_bench.hip -- random 64-byte gather/scatter microbenchmark: the measured roofline denominator
// for this engine (SURVEY.md 8d "BW_rand64").  Every lane reads one whole 64-byte sector at a
// pseudo-random 64-byte-aligned offset of a table far larger than L2 + Infinity Cache (and, with
// write_back, dirties 8 bytes of it), `iters` times with independent addresses so that many
// requests per lane are in flight.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/dint_abi.h"
#include "dint_device.h"

template <int WB>
__global__ void __launch_bounds__(256)
k_rand64(uint4 *__restrict__ tbl, uint64_t n_sectors, uint32_t iters, uint64_t seed, uint64_t *sink) {
  const uint64_t tid = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  uint4 acc = make_uint4(0, 0, 0, 0);
#pragma unroll 4
  for (uint32_t it = 0; it < iters; it++) {
    const uint64_t h = dint_hash_key(seed + tid * 0x9E3779B97F4A7C15ULL + it);
    const uint64_t s = h % n_sectors;
    uint4 *p = tbl + s * 4;
    const uint4 a = p[0], b = p[1], c = p[2], d = p[3];
    acc.x ^= a.x ^ b.y ^ c.z ^ d.w;
    acc.y += a.y + b.z + c.w + d.x;
    if (WB) ((uint2 *)p)[(h >> 40) & 7] = make_uint2(acc.x, (uint32_t)it);
  }
  if (acc.x == 0x12345678u && acc.y == 0x9abcdef0u) sink[0] = tid;  // keeps the loads alive
}

extern "C" int dint_bench_rand64(int32_t device, uint64_t bytes, uint64_t n_access, int write_back,
                                 double *out_aps, double *out_s) {
  if (!out_aps || !out_s || bytes < (1u << 20)) return DINT_EINVAL;
  if (device >= 0 && hipSetDevice(device) != hipSuccess) return DINT_ENODEV;
  uint4 *tbl = nullptr;
  uint64_t *sink = nullptr;
  if (hipMalloc((void **)&tbl, bytes) != hipSuccess) return DINT_ENOMEM;
  if (hipMalloc((void **)&sink, 8) != hipSuccess) { hipFree(tbl); return DINT_ENOMEM; }
  hipMemset(tbl, 0x5a, bytes);
  const uint64_t n_sectors = bytes / 64;
  const uint32_t threads = 256 * 2048;  // 8 blocks per CU
  uint32_t iters = (uint32_t)((n_access + threads - 1) / threads);
  if (iters == 0) iters = 1;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int rep = 0; rep < 2; rep++) {  // first run warms page tables / TLBs
    hipEventRecord(e0, 0);
    if (write_back)
      hipLaunchKernelGGL(k_rand64<1>, dim3(threads / 256), dim3(256), 0, 0, tbl, n_sectors, iters, 1234567ull + rep, sink);
    else
      hipLaunchKernelGGL(k_rand64<0>, dim3(threads / 256), dim3(256), 0, 0, tbl, n_sectors, iters, 1234567ull + rep, sink);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
  }
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  hipError_t err = hipGetLastError();
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  hipFree(tbl);
  hipFree(sink);
  if (err != hipSuccess) return DINT_EHIP;
  *out_s = ms * 1e-3;
  *out_aps = (double)threads * iters / (ms * 1e-3);
  return 0;
}
