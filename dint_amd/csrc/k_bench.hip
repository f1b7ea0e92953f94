// k_bench.hip -- memory-access microbenchmarks: the measured roofline denominators of this engine
// (SURVEY.md 8d "BW_rand64").  dint_bench_access times one access pattern over a table far larger than
// L2 + Infinity Cache: random 64-byte-aligned gathers of 8 / 16 / 64 bytes, gathers that dirty 8 bytes of
// the sector (read-modify-write), blind scatters, device-scope atomics, and the two streaming patterns the
// request / reply arrays follow.  Every lane issues `iters` independent accesses (addresses from a hash of
// the lane id, reduced with the same magic-multiply modulo the engines use -- a 64-bit `%` costs more than
// the access it feeds), four in flight per lane.  dint_bench_rand64 is the r01/r02 entry point kept for ABI
// compatibility: the 64-byte gather (or read-modify-write) at 8 blocks per CU.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/dint_abi.h"
#include "dint_device.h"

enum { BM_GATHER = 0, BM_RMW = 1, BM_SCATTER = 2, BM_ATOMIC = 3, BM_ATOMIC_RET = 4, BM_STREAM_RD = 5, BM_STREAM_WR = 6 };

template <int MODE, int WIDTH>
__global__ void __launch_bounds__(256)
k_access(uint8_t *__restrict__ tbl, dint_mod sectors, uint64_t n_vec, uint32_t iters, uint64_t seed, uint64_t *sink) {
  const uint64_t tid = (uint64_t)blockIdx.x * 256 + threadIdx.x, nthr = (uint64_t)gridDim.x * 256;
  uint32_t acc = 0;
  if (MODE == BM_STREAM_RD || MODE == BM_STREAM_WR) {  // grid-stride over 16-byte vectors: what a request array costs
    uint4 *v = (uint4 *)tbl;
#pragma unroll 4
    for (uint64_t k = tid; k < n_vec; k += nthr) {
      if (MODE == BM_STREAM_RD) { const uint4 a = v[k]; acc += a.x ^ a.w; }
      else v[k] = make_uint4((uint32_t)k, acc, 3, 4);
    }
  } else {
#pragma unroll 4
    for (uint32_t it = 0; it < iters; it++) {
      const uint64_t h = dint_hash_key(seed + tid * 0x9E3779B97F4A7C15ULL + it);
      uint8_t *p = tbl + dint_fastmod(h, sectors) * 64;
      if (MODE == BM_GATHER || MODE == BM_RMW) {
        if (WIDTH == 64) {
          const uint4 a = ((uint4 *)p)[0], b = ((uint4 *)p)[1], c = ((uint4 *)p)[2], d = ((uint4 *)p)[3];
          acc += a.x ^ b.y ^ c.z ^ d.w;
        } else if (WIDTH == 16) {
          const uint4 a = ((uint4 *)p)[(h >> 40) & 3];
          acc += a.x ^ a.w;
        } else {
          const uint2 a = ((uint2 *)p)[(h >> 40) & 7];
          acc += a.x ^ a.y;
        }
        if (MODE == BM_RMW) ((uint2 *)p)[(h >> 43) & 7] = make_uint2(acc, it);
      } else if (MODE == BM_SCATTER) {
        if (WIDTH == 64) {
          const uint4 z = make_uint4((uint32_t)h, it, 1, 2);
          ((uint4 *)p)[0] = z; ((uint4 *)p)[1] = z; ((uint4 *)p)[2] = z; ((uint4 *)p)[3] = z;
        } else if (WIDTH == 16) {
          ((uint4 *)p)[(h >> 40) & 3] = make_uint4((uint32_t)h, it, 1, 2);
        } else if (WIDTH == 8) {
          ((uint2 *)p)[(h >> 40) & 7] = make_uint2((uint32_t)h, it);
        } else {
          p[(h >> 40) & 63] = (uint8_t)it;
        }
      } else if (MODE == BM_ATOMIC) {
        __hip_atomic_fetch_add((unsigned long long *)p, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        acc += (uint32_t)atomicAdd((unsigned long long *)p, 1ull);
      }
    }
  }
  if (acc == 0x12345678u) sink[0] = tid;  // keeps the loads alive
}

template <int MODE, int WIDTH>
static void launch_access(uint32_t blocks, uint8_t *tbl, dint_mod sectors, uint64_t n_vec, uint32_t iters, uint64_t seed,
                          uint64_t *sink) {
  hipLaunchKernelGGL((k_access<MODE, WIDTH>), dim3(blocks), dim3(256), 0, 0, tbl, sectors, n_vec, iters, seed, sink);
}

extern "C" int dint_bench_access(int32_t device, uint64_t bytes, uint64_t n_access, uint32_t mode, uint32_t width,
                                 uint32_t blocks_per_cu, double *out_aps, double *out_s) {
  if (!out_aps || !out_s || bytes < (1u << 20) || mode > BM_STREAM_WR) return DINT_EINVAL;
  if (device >= 0 && hipSetDevice(device) != hipSuccess) return DINT_ENODEV;
  int dev = 0, ncu = 256;
  hipGetDevice(&dev);
  hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
  uint8_t *tbl = nullptr;
  uint64_t *sink = nullptr;
  if (hipMalloc((void **)&tbl, bytes) != hipSuccess) return DINT_ENOMEM;
  if (hipMalloc((void **)&sink, 8) != hipSuccess) { hipFree(tbl); return DINT_ENOMEM; }
  hipMemset(tbl, 0x5a, bytes);
  const dint_mod sectors = dint_make_mod(bytes / 64);
  const uint64_t n_vec = bytes / 16;
  const uint32_t blocks = (uint32_t)ncu * (blocks_per_cu ? blocks_per_cu : 8u);
  const uint64_t threads = (uint64_t)blocks * 256;
  const bool stream = mode == BM_STREAM_RD || mode == BM_STREAM_WR;
  uint32_t iters = (uint32_t)((n_access + threads - 1) / threads);
  if (iters == 0) iters = 1;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int rep = 0; rep < 2; rep++) {  // first run warms page tables / TLBs
    hipEventRecord(e0, 0);
    const uint64_t seed = 1234567ull + rep;
#define BM_CASE(M, W) launch_access<M, W>(blocks, tbl, sectors, n_vec, iters, seed, sink)
    switch (mode) {
      case BM_GATHER: if (width == 64) BM_CASE(BM_GATHER, 64); else if (width == 16) BM_CASE(BM_GATHER, 16); else BM_CASE(BM_GATHER, 8); break;
      case BM_RMW: if (width == 64) BM_CASE(BM_RMW, 64); else if (width == 16) BM_CASE(BM_RMW, 16); else BM_CASE(BM_RMW, 8); break;
      case BM_SCATTER:
        if (width == 64) BM_CASE(BM_SCATTER, 64); else if (width == 16) BM_CASE(BM_SCATTER, 16);
        else if (width == 8) BM_CASE(BM_SCATTER, 8); else BM_CASE(BM_SCATTER, 1);
        break;
      case BM_ATOMIC: BM_CASE(BM_ATOMIC, 8); break;
      case BM_ATOMIC_RET: BM_CASE(BM_ATOMIC_RET, 8); break;
      case BM_STREAM_RD: BM_CASE(BM_STREAM_RD, 16); break;
      default: BM_CASE(BM_STREAM_WR, 16); break;
    }
#undef BM_CASE
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
  *out_aps = (stream ? (double)n_vec : (double)threads * iters) / (ms * 1e-3);  // streaming: 16-byte vectors per second
  return 0;
}

extern "C" int dint_bench_rand64(int32_t device, uint64_t bytes, uint64_t n_access, int write_back,
                                 double *out_aps, double *out_s) {
  return dint_bench_access(device, bytes, n_access, write_back ? BM_RMW : BM_GATHER, 64, 8, out_aps, out_s);
}

// ---- dint_selftest: device primitives against their portable forms -------------------------------------------------
// bit 0 of the result: the wave sort network built from DPP / permlane exchanges (dint_device.h) orders 64-bit keys
// differently from the ds_bpermute network; bit 1: a single lane exchange differs.  0 = all equal.
__global__ void __launch_bounds__(256) k_selftest(uint64_t seed, uint32_t *bad) {
  const uint64_t tid = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  uint64_t w = dint_hash_key(seed + tid);
  if ((tid >> 6) & 1) w &= 0xFFFFull;                 // waves full of duplicates
  if (((tid >> 6) & 3) == 2 && (tid & 7) == 0) w = ~0ull;  // ... and of "empty lane" keys
  if (wave_sort_u64(w) != wave_sort_u64_ref(w)) atomicOr(bad, 1u);
  const uint32_t v = (uint32_t)w;
  if (lane_xor_u32<1>(v) != (uint32_t)__shfl_xor((int)v, 1, 64) || lane_xor_u32<2>(v) != (uint32_t)__shfl_xor((int)v, 2, 64) ||
      lane_xor_u32<4>(v) != (uint32_t)__shfl_xor((int)v, 4, 64) || lane_xor_u32<8>(v) != (uint32_t)__shfl_xor((int)v, 8, 64) ||
      lane_xor_u32<16>(v) != (uint32_t)__shfl_xor((int)v, 16, 64) || lane_xor_u32<32>(v) != (uint32_t)__shfl_xor((int)v, 32, 64))
    atomicOr(bad, 2u);
}
extern "C" int dint_selftest(int32_t device) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return DINT_ENODEV;
  if (device >= 0 && hipSetDevice(device) != hipSuccess) return DINT_ENODEV;
  uint32_t *d_bad = nullptr, h_bad = 0;
  if (hipMalloc((void **)&d_bad, 4) != hipSuccess) return DINT_ENOMEM;
  hipMemset(d_bad, 0, 4);
  for (uint64_t s = 1; s <= 4; s++) hipLaunchKernelGGL(k_selftest, dim3(512), dim3(256), 0, 0, s * 0x123456789ULL, d_bad);
  const hipError_t e = hipMemcpy(&h_bad, d_bad, 4, hipMemcpyDeviceToHost);
  hipFree(d_bad);
  return e == hipSuccess ? (int)h_bad : DINT_EHIP;
}
