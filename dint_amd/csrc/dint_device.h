// dint_device.h -- device-side building blocks shared by every DINT kernel (gfx950 only).
//
// Hashing follows the reference bit for bit: fasthash64 with seed 0xdeadbeef over the
// 4-byte lid (lock_fasst/udp/server.cc:81) or the 8-byte key (store/udp/kvs.h:33-35),
// lock_fasst/udp/utils.h:16-53.  Everything else here is this engine's own machinery:
// magic-multiply modulo, the batch record format, and the wave-level idx-ranking used to
// restore request order inside a bin.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define DINT_MICRO 65536u          // log append: max requests per kernel pass
#define DINT_KV_PASS 1048576u      // every other workload: max requests per kernel pass (idx fits 20 bits)
#define DINT_KV_PMAX 32768u        // ... and max bins per pass
#define DINT_KV_BINCAP 64u         // lock tables: records a bin holds in place; the rest goes to the pass's overflow area
#define DINT_KV_CMAX 2048u         // kv workloads: coarse bins per pass at most (k_kv.hip: a two-level partition)

// ---- fasthash64 ------------------------------------------------------------------------
__host__ __device__ static inline uint64_t dint_mix(uint64_t h) {
  h ^= h >> 23;
  h *= 0x2127599bf4325c37ULL;
  h ^= h >> 47;
  return h;
}
// fasthash64(&lid, 4, 0xdeadbeef): no 8-byte block, 4-byte tail  (utils.h:37-50, case 4)
__host__ __device__ static inline uint64_t dint_hash_lid(uint32_t lid) {
  const uint64_t m = 0x880355f21e6d1965ULL;
  uint64_t h = 0xdeadbeefULL ^ (4ULL * m);
  h ^= dint_mix((uint64_t)lid);
  h *= m;
  return dint_mix(h);
}
// fasthash64(&key, 8, 0xdeadbeef): one 8-byte block, no tail  (utils.h:31-35)
__host__ __device__ static inline uint64_t dint_hash_key(uint64_t key) {
  const uint64_t m = 0x880355f21e6d1965ULL;
  uint64_t h = 0xdeadbeefULL ^ (8ULL * m);
  h ^= dint_mix(key);
  h *= m;
  return dint_mix(h);
}

// ---- modulo by a run-time constant (36,000,000 / 9,000,000 / ... are not powers of 2) -----
// m = floor(2^64 / d).  q' = mulhi(h, m) is floor(h/d) or one less, so one conditional
// subtraction makes the remainder exact for every 64-bit h.
struct dint_mod {
  uint64_t d, m;
};
static inline dint_mod dint_make_mod(uint64_t d) {
  dint_mod f;
  f.d = d;
  f.m = d > 1 ? (uint64_t)((((unsigned __int128)1) << 64) / d) : 0;
  return f;
}
__host__ __device__ static inline uint64_t dint_fastmod(uint64_t h, dint_mod f) {
  if (f.d <= 1) return 0;
#ifdef __HIP_DEVICE_COMPILE__
  uint64_t q = __umul64hi(h, f.m);
#else
  uint64_t q = (uint64_t)(((unsigned __int128)h * f.m) >> 64);
#endif
  uint64_t r = h - q * f.d;
  if (r >= f.d) r -= f.d;
  return r;
}

// ---- where request i of a pass lives ------------------------------------------------------
// A pass is either one contiguous array of wire messages (request i at byte i * msg), or -- for batches that
// arrived through the multi-GPU exchange -- n_seg SEGMENTS of seg_cap message slots each, one per source rank,
// seg_stride bytes apart, of which only the first cnt[k] slots of segment k hold a request (the senders pad
// their slots to a fixed capacity so the all-to-all needs no split sizes on the host).  Request index
// i = k * seg_cap + j is then the serial position "source rank k, j-th request it sent here".
struct dint_view {
  uint32_t seg_cap;     // 0 = contiguous
  uint32_t n_seg;
  dint_mod seg;         // division by seg_cap
  uint64_t seg_stride;  // bytes between segment starts
  const uint8_t *cnt;   // u32 live count of segment k at cnt + k * cnt_stride (device memory)
  uint64_t cnt_stride;
};
static inline dint_view dint_flat_view() {
  dint_view v;
  v.seg_cap = 0; v.n_seg = 0; v.seg.d = 0; v.seg.m = 0; v.seg_stride = 0; v.cnt = nullptr; v.cnt_stride = 0;
  return v;
}
static inline dint_view dint_seg_view(uint32_t n_seg, uint32_t seg_cap, uint64_t seg_stride, const void *cnt,
                                      uint64_t cnt_stride) {
  dint_view v;
  v.seg_cap = seg_cap; v.n_seg = n_seg; v.seg = dint_make_mod(seg_cap); v.seg_stride = seg_stride;
  v.cnt = (const uint8_t *)cnt; v.cnt_stride = cnt_stride;
  return v;
}
// byte offset of request i; *live = the slot holds a request (always true for a contiguous pass)
__host__ __device__ static inline size_t dint_view_off(const dint_view &v, uint32_t i, uint32_t msg, bool *live = nullptr) {
  if (v.seg_cap == 0) {
    if (live) *live = true;
    return (size_t)i * msg;
  }
#ifdef __HIP_DEVICE_COMPILE__
  uint32_t q = (uint32_t)__umul64hi((uint64_t)i, v.seg.m);
#else
  uint32_t q = (uint32_t)(((unsigned __int128)i * v.seg.m) >> 64);
#endif
  uint32_t r = i - q * v.seg_cap;
  if (r >= v.seg_cap) { q++; r -= v.seg_cap; }
  if (live) *live = q < v.n_seg && r < *(const uint32_t *)(v.cnt + (size_t)q * v.cnt_stride);
  return (size_t)q * v.seg_stride + (size_t)r * msg;
}

// ---- wave helpers (wave = 64 lanes) ----------------------------------------------------------
__device__ static inline uint32_t lane_id() { return threadIdx.x & 63; }
__device__ static inline uint64_t lanemask_lt() {
  return (1ULL << lane_id()) - 1ULL;
}
__device__ static inline uint32_t wave_excl_scan_u32(uint32_t v, uint32_t *total) {
  uint32_t x = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    uint32_t y = __shfl_up(x, d, 64);
    if ((int)lane_id() >= d) x += y;
  }
  *total = __shfl(x, 63, 64);
  return x - v;
}

// 64 keys, one per lane, ascending (bitonic network over the wave, 21 compare-exchange steps)
__device__ static inline uint64_t wave_sort_u64(uint64_t w) {
  const uint32_t lane = lane_id();
#pragma unroll
  for (uint32_t k = 2; k <= 64; k <<= 1) {
#pragma unroll
    for (uint32_t j = k >> 1; j > 0; j >>= 1) {
      const uint32_t lo = __shfl_xor((uint32_t)w, (int)j, 64), hi = __shfl_xor((uint32_t)(w >> 32), (int)j, 64);
      const uint64_t o = ((uint64_t)hi << 32) | lo;
      const bool up = (lane & k) == 0;           // this k-block sorts ascending
      const bool low = (lane & j) == 0;          // lower lane of the pair
      w = (low == up) ? (w < o ? w : o) : (w < o ? o : w);
    }
  }
  return w;
}
