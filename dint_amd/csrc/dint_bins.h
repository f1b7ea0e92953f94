// dint_bins.h -- the pass machinery every workload's kernels share (included by k_kv.hip and k_locks.hip; everything
// here has internal linkage).  One pass = count (each request reserves a position in the bin of its group; the
// reservations of a workgroup on one bin are merged in an LDS hash) -> k_kv_scan_place (ranges of the overflow area for
// the bins of more than DINT_KV_BINCAP records, overflow records into their range; it also makes the pass's log tail
// current and clears the counters the next pass will use -- the big-bin lists and the published log counts alternate
// between passes, so nothing has to be reset behind the resolve kernel) -> resolve.  Also the
// helpers of the big-bin workgroups: an LDS / register bitonic sort of a stretch of <= KVB_NMAX 64-bit keys, and O(1)
// range queries (bits set, last / next set bit) over 4096-bit masks of the sorted stretch.
#pragma once
#include <cstdlib>

#include "dint_kernels.h"

#define KV_TB 1024u           // threads per workgroup of k_kv_count / k_kv_place (= requests per workgroup)
#define KV_NONE 0xFFFFFFFFu

__device__ static inline uint32_t block_hash_insert(uint32_t *keys, uint32_t k) {  // 2 * KV_TB slots, keys != KV_NONE
  uint32_t h = (k * 0x9E3779B1u) >> (32 - 11);
  for (;;) {
    const uint32_t old = atomicCAS(&keys[h], KV_NONE, k);
    if (old == KV_NONE || old == k) return h;
    h = (h + 1) & (2 * KV_TB - 1);
  }
}
static_assert(KV_TB == 1024, "block_hash_insert assumes 2048 slots");

#define KV_PLACE_GRID 64u
// ---- k_kv_scan_place: scan and place in ONE launch ---------------------------------------------------------------
// (r01-r03a: two launches, ~5 us of launch + dependency per pass -- a tenth of a 64k-request lock pass.)  Every one of
// the KV_PLACE_GRID workgroups runs the scan of the big-bin list itself (a few hundred entries) and stores the same
// bin_off[] words; what a workgroup reads back are its OWN stores (behind a workgroup-scope fence and barrier), so
// nothing has to cross between workgroups inside the kernel.  Workgroup 0 does what the scan kernel did on the side (the next pass's counters, the log tail, the stats).
__device__ static inline void
kv_scan_place_body(const uint32_t *__restrict__ bin_cnt, uint32_t *bin_off, const uint32_t *__restrict__ big,
                   uint32_t *__restrict__ big_next, uint32_t *__restrict__ blk_pub_next, uint32_t *tail,
                   dint_dev_stats *__restrict__ stats, const uint4 *__restrict__ ovl, uint64_t *__restrict__ ovf) {
  __shared__ uint32_t Sw[KV_TB / 64];
  const uint32_t t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const bool first = blockIdx.x == 0;
  if (first) {
    if (t < 4) big_next[t] = 0;
    blk_pub_next[t] = 0;
    if (t == 0 && tail) tail[0] = tail[1];
  }
  const uint32_t nbig = big[0], novl = big[1];
  if (nbig == 0) return;  // no bin above DINT_KV_BINCAP records: nothing was listed for placement either
  uint32_t run = 0;
  for (uint32_t lo = 0; lo < nbig; lo += KV_TB) {  // workgroup-uniform trip count; one trip unless the pass is very skewed
    const uint32_t bin = lo + t < nbig ? big[4 + lo + t] : KV_NONE;
    const uint32_t extra = bin != KV_NONE ? bin_cnt[bin] - DINT_KV_BINCAP : 0;
    uint32_t tot, x = wave_excl_scan_u32(extra, &tot);
    __syncthreads();
    if (lane == 0) Sw[wave] = tot;
    __syncthreads();
    uint32_t all = 0;
    for (uint32_t w = 0; w < KV_TB / 64; w++) {
      if (w < wave) x += Sw[w];
      all += Sw[w];
    }
    if (bin != KV_NONE) bin_off[bin] = run + x;
    run += all;
  }
  if (first && t == 0) atomicAdd(&stats->big_bin_requests, (unsigned long long)run + (unsigned long long)nbig * DINT_KV_BINCAP);
  // workgroup scope only: the waves of a workgroup share one CU, hence one L1 and one L2 -- a device-scope fence here
  // writes the L2s back across the XCDs (measured: 70 us instead of 10)
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
  __syncthreads();
  for (uint32_t i = blockIdx.x * KV_TB + t; i < novl; i += KV_PLACE_GRID * KV_TB) {
    const uint4 o = ovl[i];
    const uint32_t off = __hip_atomic_load(&bin_off[o.z], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    ovf[off + o.w - DINT_KV_BINCAP] = ((uint64_t)o.y << 32) | o.x;
  }
}
static __global__ void __launch_bounds__(KV_TB)
k_kv_scan_place(const uint32_t *__restrict__ bin_cnt, uint32_t *bin_off, const uint32_t *__restrict__ big,
                uint32_t *__restrict__ big_next, uint32_t *__restrict__ blk_pub_next, uint32_t *tail,
                dint_dev_stats *__restrict__ stats, const uint4 *__restrict__ ovl, uint64_t *__restrict__ ovf) {
  kv_scan_place_body(bin_cnt, bin_off, big, big_next, blk_pub_next, tail, stats, ovl, ovf);
}

// ---- big bins ----------------------------------------------------------------------------------------------
#define KVB_T 512u
#define KVB_W (KVB_T / 64u)
#define KVB_GRID 512u              // workgroups that walk the big-bin list (most exit at once)
#define KVB_NMAX 4096u             // requests resolved together (one stretch of a bin)
#define KVB_NW (KVB_NMAX / 64u)    // mask words of a stretch = lanes of one wave
#define KVB_NBK 2048u              // request-index buckets that cut a longer bin into stretches
#define KVB_HOT_MIN 256u           // a stretch with a key of at least this many requests (and half the stretch) takes the dominant-key path
#define KVB_HOT_MIN_LOCKS 256u     // ... the lock tables' threshold (k_locks.hip)
#define KVB_MMAX 1024u             // ... if the key has at most this many writers + lock ops
static_assert(KVB_NW == 64, "the per-word tables are built with one lane per mask word");
// the thresholds can be overridden from the environment (tuning runs); read once per process
static inline uint32_t dint_hot_min(const char *env, uint32_t dflt) {
  const char *v = getenv(env);
  return v && *v ? (uint32_t)strtoul(v, nullptr, 10) : dflt;
}
struct kvb_lead { uint32_t found_link, slot, ver0, la0, lb0; };   // found_link: found << 31 | link
struct kvb_carry { uint32_t la, lb, ver; int src; uint32_t miss; };
struct kvb_pop { uint16_t below[KVB_NW + 1]; };                  // bits set in the words before w
struct kvb_edge { int16_t last[KVB_NW], next[KVB_NW]; };         // highest set bit before word w / lowest after it, -1: none

__device__ static inline void kvb_build_pop(const uint64_t *M, kvb_pop &P) {  // one whole wave
  const uint32_t lane = lane_id();
  uint32_t tot;
  const uint32_t x = wave_excl_scan_u32((uint32_t)__popcll(M[lane]), &tot);
  P.below[lane] = (uint16_t)x;
  if (lane == 63) P.below[KVB_NW] = (uint16_t)tot;
}
__device__ static inline void kvb_build_edge(const uint64_t *M, kvb_edge &E) {  // one whole wave
  const int lane = (int)lane_id();
  const uint64_t m = M[lane];
  int hi = m ? lane * 64 + 63 - __clzll((long long)m) : -1;
  int lo = m ? lane * 64 + __ffsll((unsigned long long)m) - 1 : 0x7FFF;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int a = __shfl_up(hi, d, 64), b = __shfl_down(lo, d, 64);
    if (lane >= d) hi = max(hi, a);
    if (lane + d < 64) lo = min(lo, b);
  }
  const int a = __shfl_up(hi, 1, 64), b = __shfl_down(lo, 1, 64);
  E.last[lane] = (int16_t)(lane ? a : -1);
  E.next[lane] = (int16_t)((lane == 63 || b == 0x7FFF) ? -1 : b);
}
__device__ static inline bool kvb_bit(const uint64_t *M, uint32_t p) { return (M[p >> 6] >> (p & 63)) & 1ull; }
__device__ static inline uint32_t kvb_below(const uint64_t *M, const kvb_pop &P, uint32_t x) {  // bits set in [0, x)
  const uint32_t w = x >> 6, r = x & 63;
  return P.below[w] + (r ? (uint32_t)__popcll(M[w] & ((1ull << r) - 1ull)) : 0u);
}
__device__ static inline uint32_t kvb_popc(const uint64_t *M, const kvb_pop &P, uint32_t a, uint32_t b) {  // in [a, b)
  return a < b ? kvb_below(M, P, b) - kvb_below(M, P, a) : 0u;
}
__device__ static inline int kvb_last(const uint64_t *M, const kvb_edge &E, uint32_t a, uint32_t b) {  // highest in [a, b) or -1
  if (a >= b) return -1;
  const uint32_t w = (b - 1) >> 6, r = b & 63;
  const uint64_t m = M[w] & (r ? (1ull << r) - 1ull : ~0ull);
  const int res = m ? (int)(w * 64 + 63 - __clzll((long long)m)) : (int)E.last[w];
  return res >= (int)a ? res : -1;
}
__device__ static inline int kvb_first(const uint64_t *M, const kvb_edge &E, uint32_t a) {  // lowest at or above a, or -1
  if (a >= KVB_NMAX) return -1;
  const uint32_t w = a >> 6;
  const uint64_t m = M[w] & (~0ull << (a & 63));
  return m ? (int)(w * 64 + __ffsll((unsigned long long)m) - 1) : (int)E.next[w];
}
__device__ static inline uint32_t kvb_range_popc(const uint64_t *M, uint32_t a, uint32_t b) {  // bits set in [a, b), no table
  uint32_t cnt = 0;
  for (uint32_t w = a >> 6; w <= ((b - 1) >> 6) && a < b; w++) {
    uint64_t m = M[w];
    if (w == (a >> 6)) m &= ~0ull << (a & 63);
    if (w == ((b - 1) >> 6) && (b & 63)) m &= (1ull << (b & 63)) - 1ull;
    cnt += (uint32_t)__popcll(m);
  }
  return cnt;
}

// the value of lane (lane ^ j), j a wave-uniform run-time value below 64: ds_bpermute.  (r04 tried the DPP / permlane
// exchanges of dint_device.h here behind a switch on j: the stretch sort of 4,096 keys went from 40 to ~120 us and
// lock_fasst lost 9 % -- six-way branches inside the unrolled register loops.  The compile-time network of one wave,
// wave_sort_u64, keeps them.)
__device__ __forceinline__ static uint32_t lane_xor_rt(uint32_t v, uint32_t j) { return (uint32_t)__shfl_xor((int)v, (int)j, 64); }
__device__ __forceinline__ static uint64_t lane_xor_rt(uint64_t v, uint32_t j) {
  return ((uint64_t)lane_xor_rt((uint32_t)(v >> 32), j) << 32) | lane_xor_rt((uint32_t)v, j);
}
// Bitonic sort of KVB_T * R keys held R per thread (thread t owns positions R*t .. R*t + R - 1): the steps with a
// partner inside the thread run in registers, those inside the wave by lane exchange, and only the few with a
// partner in another wave go through LDS (the key array itself is the staging area -- every key is in a register by then --
// laid out [r][t] so the exchange is conflict-free).  KVB_SORT_INLINE (k_kv.hip's big-sub kernel): inlined -- as a function
// of its own it takes the LDS array as a generic pointer (flat instructions); the lock kernels, at their register limit,
// keep the call.
#ifdef KVB_SORT_INLINE
#define KVB_SORT_ATTR __forceinline__
#else
#define KVB_SORT_ATTR __attribute__((noinline))
#endif
template <int R, class K>
__device__ KVB_SORT_ATTR static void kvb_sort_blocked_t(K *Sk, uint32_t N) {  // N: power of two, 64 R <= N <= KVB_T R; positions >= N hold the maximum
  K *X = Sk;
  const uint32_t t = threadIdx.x;
  K v[R];
#pragma unroll
  for (int r = 0; r < R; r++) v[r] = Sk[R * t + r];
  auto cmpx = [](K &a, K &b, bool up) {  // up: a <= b afterwards
    const K lo = a < b ? a : b, hi = a < b ? b : a;
    a = up ? lo : hi; b = up ? hi : lo;
  };
  for (uint32_t k = 2; k <= N; k <<= 1) {
    for (uint32_t j = k >> 1; j >= (uint32_t)R; j >>= 1) {  // partner in thread t ^ (j / R), same register
      const uint32_t tj = j / R;
      const bool low = (t & tj) == 0;
      if (tj >= 64) {
        __syncthreads();
#pragma unroll
        for (int r = 0; r < R; r++) X[r * KVB_T + t] = v[r];
        __syncthreads();
      }
#pragma unroll
      for (int r = 0; r < R; r++) {
        const K o = tj < 64 ? lane_xor_rt(v[r], tj) : X[r * KVB_T + (t ^ tj)];
        const bool up = ((R * t + r) & k) == 0;
        v[r] = (low == up) ? (v[r] < o ? v[r] : o) : (v[r] < o ? o : v[r]);
      }
    }
#pragma unroll
    for (int jj = R / 2; jj > 0; jj >>= 1) {  // partner in the same thread
      if ((uint32_t)jj < k) {
#pragma unroll
        for (int r = 0; r < R; r++)
          if ((r & jj) == 0) cmpx(v[r], v[r | jj], ((R * t + r) & k) == 0);
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < R; r++) Sk[R * t + r] = v[r];
  __syncthreads();
}
template <int R>
__device__ __forceinline__ static void kvb_sort_blocked(uint64_t *Sk, uint32_t N = KVB_T * R) { kvb_sort_blocked_t<R, uint64_t>(Sk, N); }
template <int R>
__device__ __forceinline__ static void kvb_sort_blocked_u32(uint32_t *Sk, uint32_t N = KVB_T * R) { kvb_sort_blocked_t<R, uint32_t>(Sk, N); }

// sort Sk[0, m) ascending (m <= KVB_NMAX; the slots up to the next power of two are filled with ~0 and sort last).
// <= 512 keys: one per thread, in-wave steps by shuffle, the wide ones through LDS; more: 2 / 4 / 8 keys per thread.
__device__ static inline void kvb_sort_stretch(uint64_t *Sk, uint32_t m) {
  const uint32_t t = threadIdx.x;
  uint32_t N = 64;
  while (N < m) N <<= 1;
  for (uint32_t k = m + t; k < max(N, KVB_T); k += KVB_T) Sk[k] = ~0ull;  // empty slots sort last
  __syncthreads();
  if (N <= KVB_T) {
    kvb_sort_blocked<1>(Sk, N);  // (slots N .. KVB_T - 1 hold ~0: they stay last)
  } else if (N == 2 * KVB_T) {
    kvb_sort_blocked<2>(Sk);
  } else if (N == 4 * KVB_T) {
    kvb_sort_blocked<4>(Sk);
  } else {
    kvb_sort_blocked<8>(Sk);
  }
}
