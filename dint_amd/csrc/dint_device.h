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
#define DINT_KV_BIGQ_MAX (DINT_KV_CMAX * 128u)  // ... work items of k_kv_big at most: pass / 8 hot-key pieces + 2 per big sub (k_kv.hip, kv_hot_item)

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

// ---- the value of lane (lane ^ J), J a power of two --------------------------------------------------------------
// r01-r03 exchanged with __shfl_xor, which is a ds_bpermute whatever J is: an LDS-crossbar round trip of ~50 ns, and
// the 21 steps of the sort network below depend on one another -- 2 us per sorted chunk, a tenth of a resolve
// workgroup's life.  18 of the 21 steps stay inside a row of 16 lanes, where a DPP move does the exchange at VALU speed
// (J = 1, 2: quad_perm; 8: row_ror:8; 4: row_shl:4 into banks 0 / 2 and row_shr:4 into banks 1 / 3); the three steps
// across rows use gfx950's v_permlane16_swap / v_permlane32_swap.  DINT_SORT_BPERMUTE builds the portable form
// (dint_selftest compares the two on the device).
template <uint32_t J>
__device__ __forceinline__ static uint32_t lane_xor_u32(uint32_t v) {
#ifdef DINT_SORT_BPERMUTE
  return (uint32_t)__shfl_xor((int)v, (int)J, 64);
#else
  if constexpr (J == 1) return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0xB1, 0xF, 0xF, false);        // quad_perm [1,0,3,2]
  else if constexpr (J == 2) return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x4E, 0xF, 0xF, false);   // quad_perm [2,3,0,1]
  else if constexpr (J == 4) {
    const int a = __builtin_amdgcn_update_dpp((int)v, (int)v, 0x104, 0xF, 0x5, false);                              // row_shl:4 -> lanes with bit 2 clear
    return (uint32_t)__builtin_amdgcn_update_dpp(a, (int)v, 0x114, 0xF, 0xA, false);                                // row_shr:4 -> lanes with bit 2 set
  } else if constexpr (J == 8) return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x128, 0xF, 0xF, false); // row_ror:8
  else if constexpr (J == 16) {
    const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);  // {odd rows <- even rows, even rows <- odd rows}
    return (threadIdx.x & 16) ? r[0] : r[1];
  } else {
    static_assert(J == 32, "lane_xor_u32: J = 1, 2, 4, 8, 16, 32");
    const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);  // {upper half <- lower half, lower half <- upper half}
    return (threadIdx.x & 32) ? r[0] : r[1];
  }
#endif
}
template <uint32_t K, uint32_t J>
__device__ __forceinline__ static void wave_sort_step(uint64_t &w, uint32_t lane) {
  const uint32_t lo = lane_xor_u32<J>((uint32_t)w), hi = lane_xor_u32<J>((uint32_t)(w >> 32));
  const uint64_t o = ((uint64_t)hi << 32) | lo;
  const bool up = (lane & K) == 0;           // this K-block sorts ascending
  const bool low = (lane & J) == 0;          // lower lane of the pair
  w = (low == up) ? (w < o ? w : o) : (w < o ? o : w);
  if constexpr (J > 1) wave_sort_step<K, J / 2>(w, lane);
}
template <uint32_t K>
__device__ __forceinline__ static void wave_sort_stage(uint64_t &w, uint32_t lane) {
  wave_sort_step<K, K / 2>(w, lane);
  if constexpr (K < 64) wave_sort_stage<K * 2>(w, lane);
}
// 64 keys, one per lane, ascending (bitonic network over the wave, 21 compare-exchange steps)
__device__ static inline uint64_t wave_sort_u64(uint64_t w) {
  wave_sort_stage<2>(w, lane_id());
  return w;
}
// ... the portable form (ds_bpermute exchanges), kept as the reference of dint_selftest
__device__ static inline uint64_t wave_sort_u64_ref(uint64_t w) {
  const uint32_t lane = lane_id();
#pragma unroll
  for (uint32_t k = 2; k <= 64; k <<= 1) {
#pragma unroll
    for (uint32_t j = k >> 1; j > 0; j >>= 1) {
      const uint32_t lo = __shfl_xor((uint32_t)w, (int)j, 64), hi = __shfl_xor((uint32_t)(w >> 32), (int)j, 64);
      const uint64_t o = ((uint64_t)hi << 32) | lo;
      const bool up = (lane & k) == 0;
      const bool low = (lane & j) == 0;
      w = (low == up) ? (w < o ? w : o) : (w < o ? o : w);
    }
  }
  return w;
}
