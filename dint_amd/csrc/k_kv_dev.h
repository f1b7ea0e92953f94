// k_kv_dev.h -- device code of the store / tatp / smallbank shard servers (the kernels of k_kv.hip's passes) and the launch
// of one pass set, templated on the workload.  Included by k_kv.hip (host side: knobs, pass arguments, tables, dumps) and by
// the three instantiation units k_kv_store.hip / k_kv_tatp.hip / k_kv_smallbank.hip.  (Until r05 all of it was k_kv.hip: one
// 3,800-line unit that took two minutes to compile; the kernels themselves are unchanged by the split.)
#pragma once
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <utility>
#include <vector>

#include "../../include/dint_abi.h"
#include "dint_kv.h"
#define KVB_SORT_INLINE
#include "dint_bins.h"

// ---- wire formats ------------------------------------------------------------------------------------
template <int WL> struct Fmt;
template <> struct Fmt<DINT_WL_STORE> {  // store/udp/net.h:34-41
  static constexpr uint32_t MSG = 53, TYPE = 0, KEY = 1, VAL = 9, VER = 49, VS = 40;
  static constexpr bool HAS_TABLE = false;
  static constexpr uint32_t TABLE = 0;
};
template <> struct Fmt<DINT_WL_TATP> {  // tatp/udp/net.h:57-66
  static constexpr uint32_t MSG = 55, TYPE = 1, KEY = 3, VAL = 11, VER = 51, VS = 40;
  static constexpr bool HAS_TABLE = true;
  static constexpr uint32_t TABLE = 2;
};
template <> struct Fmt<DINT_WL_SMALLBANK> {  // smallbank/udp/net.h:41-50
  static constexpr uint32_t MSG = 23, TYPE = 1, KEY = 3, VAL = 11, VER = 19, VS = 8;
  static constexpr bool HAS_TABLE = true;
  static constexpr uint32_t TABLE = 2;
};

// request classes: 0 = unknown (reply untouched, counted), 1 = table op, 2 = log op
template <int WL>
__device__ static inline uint32_t kv_class(uint32_t type, int load_mode) {
  if (load_mode && type == DINT_KV_LOAD_OP) return 1;
  if (WL == DINT_WL_STORE) return type <= 2 ? 1 : 0;
  if (WL == DINT_WL_TATP) {
    switch (type) {
      case 0: case 1: case 2: case 12: case 13: case 18: case 19: case 22: case 23: return 1;
      case 14: case 24: return 2;
      default: return 0;
    }
  }
  return (type <= 5 || type == 17) ? 1 : (type == 6 ? 2 : 0);  // 17 = WARMUP_READ (eBPF flavour)
}

// 16-bit request descriptor carried from the scatter kernel to the resolve kernels
__device__ static inline uint32_t kv_pay(uint32_t type, uint32_t q, uint32_t kh) {
  return (type == DINT_KV_LOAD_OP ? 31u : type) | (q << 5) | (kh << 7);
}
__device__ static inline uint32_t pay_type(uint32_t p) { const uint32_t t = p & 31u; return t == 31u ? DINT_KV_LOAD_OP : t; }
__device__ static inline uint32_t pay_q(uint32_t p) { return (p >> 5) & 3u; }
__device__ static inline uint32_t pay_kh(uint32_t p) { return (p >> 7) & 511u; }
// table of a group key: group keys are allocated table by table (kv_dev::gk_base)
__device__ static inline uint32_t kv_table_of(const kv_dev *kv, uint32_t gk) {
  uint32_t t = 0;
  for (uint32_t k = 1; k < kv->n_tables; k++) t += gk >= kv->gk_base[k];
  return t;
}

__device__ static inline uint64_t ld_u64(const uint8_t *p) { uint64_t v; __builtin_memcpy(&v, p, 8); return v; }
__device__ static inline uint32_t ld_u32(const uint8_t *p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }
__device__ static inline void st_u32(uint8_t *p, uint32_t v) { __builtin_memcpy(p, &v, 4); }

// ---- memory policy of the device build: pool words are only ever touched with device-scope RMWs ------
struct kv_dev_mem {
  __device__ static inline uint32_t fetch_add(uint32_t *p, uint32_t v) { return atomicAdd(p, v); }
  __device__ static inline uint32_t load32(uint32_t *p) { return atomicAdd(p, 0u); }
  __device__ static inline void store32(uint32_t *p, uint32_t v) { atomicExch(p, v); }
  __device__ static inline unsigned long long load64(unsigned long long *p) { return atomicAdd(p, 0ull); }
  __device__ static inline bool cas64(unsigned long long *p, unsigned long long exp, unsigned long long des) {
    return atomicCAS(p, exp, des) == exp;
  }
};

// The table descriptors in LDS (every resolve / hot / big workgroup copies them first): frees of pass `seq` go to pend set
// (seq & 1) -- kv_pool_rotate (k_kv_part, the next pass but one) makes them poppable.  Call between the barrier behind the
// copy and the next barrier; nothing touches a pend list before that.
__device__ static inline void kv_dev_pend_set(kv_dev &Skv, uint32_t pno) {
  if (threadIdx.x < DINT_KV_MAX_TABLES && (pno & 1u)) Skv.tab[threadIdx.x].pend_head += KV_NLISTS;
}
// What a resolve workgroup hands to the hot-key workers of the SAME launch (k_kv_pass): the 8-byte records of a big sub, its work
// items and their "listed" tags.  The two sit on different XCDs as a rule, whose L2s are not coherent: agent-scope accesses
// (write-through / L2-bypassing on gfx950) instead of a release that would write a whole L2 back.
__device__ static inline void kv_st_agent(uint64_t *p, uint64_t v) { __hip_atomic_store((unsigned long long *)p, (unsigned long long)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ static inline uint64_t kv_ld_agent(const uint64_t *p) { return __hip_atomic_load((const unsigned long long *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ static inline void kv_st_agent(uint4 *p, const uint4 &v) {
  kv_st_agent((uint64_t *)p, ((uint64_t)v.y << 32) | v.x);
  kv_st_agent((uint64_t *)p + 1, ((uint64_t)v.w << 32) | v.z);
}
__device__ static inline uint4 kv_ld_agent(const uint4 *p) {
  const uint64_t a = kv_ld_agent((const uint64_t *)p), b = kv_ld_agent((const uint64_t *)p + 1);
  return make_uint4((uint32_t)a, (uint32_t)(a >> 32), (uint32_t)b, (uint32_t)(b >> 32));
}

// ---- batch record of the kv passes: one 64-bit word per table request ------------------------------------
//   bits 0..15            payload: type (5 bits, LOAD -> 31) | lock quadrant << 5 | 9 key-hash bits << 7
//   bits 16..15+ibits     request index inside the pass   (n <= 2^ibits)
//   bits 16+ibits..63     group key / P                   (the remainder is the bin id)
// A pass is cut into P bins, bin = group % P, P = n / 32 (dint_pick_bins_load): ANY number, so that a bin holds ~32
// records whatever n is (a power of two, r01-r03a, left 16-32).  Fuller bins were measured and do not pay.
struct kv_cut { uint32_t P, ibits, magic; };  // magic = floor(2^32 / P), 0 for P = 1
static inline kv_cut kv_make_cut(uint32_t P, uint32_t n) {
  kv_cut c;
  c.P = P;
  c.ibits = 1;
  while (c.ibits < 32 && (1ull << c.ibits) < n) c.ibits++;
  c.magic = P > 1 ? (uint32_t)((1ull << 32) / P) : 0u;
  return c;
}
__device__ static inline uint32_t kv_cut_div(uint32_t gk, const kv_cut &c, uint32_t *bin) {  // gk / P, *bin = gk % P
  if (c.P <= 1) { *bin = 0; return gk; }
  uint32_t q = __umulhi(gk, c.magic), r = gk - q * c.P;  // q is the quotient or one less
  if (r >= c.P) { r -= c.P; q++; }
  *bin = r;
  return q;
}
__device__ static inline uint32_t kv_cut_gk(uint32_t gq, uint32_t bin, const kv_cut &c) { return gq * c.P + bin; }
__device__ static inline uint64_t kv_rec(uint32_t gq, uint32_t idx, uint32_t pay, const kv_cut &c) {
  return ((uint64_t)gq << (16 + c.ibits)) | ((uint64_t)idx << 16) | (pay & 0xFFFFu);
}
__device__ static inline uint32_t kv_rec_pay(uint64_t r) { return (uint32_t)r & 0xFFFFu; }
__device__ static inline uint32_t kv_rec_idx(uint64_t r, const kv_cut &c) { return (uint32_t)(r >> 16) & (uint32_t)((1ull << c.ibits) - 1ull); }
// sort key of a record: group / P | 9 key-hash bits | idx | 7 payload bits (type, lock quadrant)
__device__ static inline uint64_t kv_sort_key(uint64_t r, const kv_cut &c) {
  const uint32_t pay = kv_rec_pay(r);
  return ((r >> (16 + c.ibits)) << (16 + c.ibits)) | ((uint64_t)((pay >> 7) & 511u) << (7 + c.ibits)) |
         ((uint64_t)kv_rec_idx(r, c) << 7) | (pay & 0x7Fu);
}

// what one request is, from its wire bytes: shared by k_kv_count and k_kv_place
struct kv_reqinfo {
  uint32_t type, table, cls;  // cls 0 = bad, 1 = table request, 2 = log request
  uint64_t key;
};
template <int WL>
__device__ static inline kv_reqinfo kv_read_request(const uint8_t *m, bool live, const kv_dev *kv, int load_mode) {
  using F = Fmt<WL>;
  kv_reqinfo r = {0, 0, 0, 0};
  if (live) {
    r.type = m[F::TYPE];
    r.table = F::HAS_TABLE ? m[F::TABLE] : 0;
    r.cls = kv_class<WL>(r.type, load_mode);
    if (r.table >= kv->n_tables) r.cls = 0;  // the reference indexes tables[] out of bounds
    if (r.cls) r.key = ld_u64(m + F::KEY);
  }
  return r;
}

// ---- one pass of one engine, as the kernels see it -------------------------------------------------------------
// A pass is a TWO-LEVEL partition (r04).  Level 1, k_kv_part: the requests are cut into C coarse
// bins, coarse = group % C with C ~ n / 512 (ANY number: kv_cut), by workgroups of KV_TB * RPT requests that count
// their records per coarse bin in LDS and reserve each run with ONE device atomic per (workgroup, coarse bin) --
// ~0.2 atomics per request where the one-level pass (bin = group % (n / 32), r01-r03) paid ~0.95 and an 8-byte
// partial-sector scatter per request.  A record is 16 bytes and carries the request's KEY, so the resolve kernel never
// gathers it from the message array again: {key, group / C | idx | type | lock quadrant | 9 key-hash bits}.  Level 2,
// k_kv_resolve: one workgroup per coarse bin splits its ~512 records by sub = (group / C) % 64 in LDS, packs
// neighbouring subs into chunks of <= 64 records and hands each chunk to a wave (kv_chunk: sorted in registers, the
// closed forms); a sub of more than 64 records -- a hot key -- is resolved by the whole workgroup afterwards
// (kv_big_bin).  Two launches per pass instead of three.
struct kv_pass_args {
  const uint8_t *req;
  uint8_t *rep;
  uint32_t n, n_tiles;
  const kv_dev *kv;
  dint_log log;
  kv_cut cut;            // cut.P = C coarse bins
  uint32_t cap;          // records a coarse bin holds in place; the rest goes to the pass's overflow list
  uint32_t lcap;         // records of a coarse bin's small subs that are resolved from LDS (<= KVR_LCAP)
  uint32_t *bin_cnt;     // [C] records per coarse bin (the resolve workgroups leave them zero)
  uint4 *kbins;          // [C][cap] records
  uint32_t *big, *big_z;     // the pass's control words (dint_kv_sets::ctl) / the set this pass's resolve stage zeroes (the pass after next's)
  uint32_t *blk_pub, *blk_pub_z;
  uint32_t *bigrdy;          // [item] listed: tagged with `seq`
  uint64_t *sbx;             // smallbank, [item][KSB_WORDS]: what the pieces of a hot row and their coordinator tell each other (kv_sb_item)
  uint32_t np_max;           // pieces of one hot key at most: KVR_NPMAX (store / tatp), DINT_KV_SB_NPMAX <= KSB_NPMAX (smallbank)
  uint32_t sb_pieces;        // smallbank with its rows in pieces (kv_sb_item)
  uint32_t pno;              // pass number & 1: which pend set the pass's frees go to, which log tail word its partition reads
  uint4 *ovl;            // overflow list: two uint4 per entry {record, {coarse bin, -, -, -}}
  uint64_t *ovf;         // 8-byte records of the big subs, one range per sub
  uint64_t *ovf2;        // ... and the same ranges again: a sub of several stretches is regrouped by stretch once
  uint4 *bigq;           // the pass's work items for k_kv_big, two uint4 each (kvq_*): big subs and hot-key pieces; big[3] = how many
  unsigned long long *hotpub;  // [item] what the pieces of one hot key tell each other (kvh_word), tagged with `seq`
  uint4 *lateq;          // what k_kv_hot leaves to k_kv_big: {bin, offset, records, 0 = in ovf / 1 = in ovf2}; big[5] = how many
  uint32_t split_min;    // a big sub of at least this many records whose requests are nearly all ONE key is cut into pieces
  uint32_t split_target; // ... of about this many requests each (<= KVB_T: one per thread of the workgroup that answers it)
  uint32_t inv_n;        // floor(2^32 / n): the piece of a request index (kv_piece_of)
  uint32_t seq;          // this pass's tag in hotpub (never 0)
  dint_dev_stats *stats;
  int load_mode, force_flags;
  uint32_t has_log;
  uint64_t *trace;       // DINT_KV_TRACE=1: 32 words per resolve workgroup (10 ns stamps of its phases), else nullptr
  dint_view V;
};
struct kv_multi_args { kv_pass_args e[DINT_KV_MULTI_MAX]; };

__device__ static inline uint64_t u4_key(const uint4 &r) { return ((uint64_t)r.y << 32) | r.x; }
__device__ static inline uint64_t u4_meta(const uint4 &r) { return ((uint64_t)r.w << 32) | r.z; }

// K 16-byte vectors per thread of a tile's messages, request array -> reply array: unconditional loads at clamped indices
// (branch-free, so the loads stay in flight together) and unconditional stores (a clamped lane rewrites vector nv-1 with
// the same bytes).  A pack expansion, not a loop over an array: private arrays that survive to the backend are promoted to
// LDS (64 KB per workgroup) or left in scratch.
template <uint32_t TB, uint32_t... Ks>
__device__ __forceinline__ static void kv_copy_tile(const uint4 *__restrict__ s, uint4 *__restrict__ d, uint32_t nv, uint32_t t,
                                                    std::integer_sequence<uint32_t, Ks...>) {
  const uint4 v[] = {s[min(t + Ks * TB, nv - 1)]...};
  ((d[min(t + Ks * TB, nv - 1)] = v[Ks]), ...);
}

// ---- k_kv_part --------------------------------------------------------------------------------------------------
// TB threads (KV_TB = 1,024 in k_kv_part; KVB_T = 512 in k_kv_hot_part, where the tiles ride beside the previous pass's hot
// keys), RPT requests per thread (request j of thread t of tile T: index T * TB * RPT + j * TB + t).  Copy the
// messages to the reply array, classify, hash, count per coarse bin in LDS, reserve the runs, store the records.  Log
// requests are finished here: the canonical 64-byte record goes to ring position tail + (#log requests below i)
// [deterministic: an exclusive scan over the tiles by decoupled look-back, not an atomic].
template <int RPT, uint32_t TB = KV_TB>
struct kv_part_lds {
  uint32_t Hc[DINT_KV_CMAX];  // records of this tile per coarse bin; then the position of the tile's first one
  uint32_t Sov[2];            // overflow records of the tile; their place in the pass's list
  uint32_t Stile;
  uint32_t Swl[TB / 64 * RPT], Swp[TB / 64];
};
template <int WL, int RPT, uint32_t TB = KV_TB>
__device__ __forceinline__ static void kv_part_body(const kv_pass_args &A, kv_part_lds<RPT, TB> &S, bool first_block) {
  using F = Fmt<WL>;
  constexpr uint32_t T = TB * RPT, NWV = TB / 64;
  auto &Hc = S.Hc; auto &Sov = S.Sov; auto &Stile = S.Stile; auto &Swl = S.Swl; auto &Swp = S.Swp;
  const uint32_t t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const uint32_t n = A.n, C = A.cut.P;
  const kv_dev *__restrict__ kv = A.kv;
  const uint8_t *__restrict__ req = A.req;
  uint8_t *rep = A.rep;
  // tiles are handed out in start order (a ticket, not blockIdx), so the tiles before mine belong to workgroups that
  // are already running: what the log-position look-back below waits for
  if (t == 0) { Stile = atomicAdd(&A.big[2], 1u); Sov[0] = 0; }
  for (uint32_t k = t; k < C; k += TB) Hc[k] = 0;
  __syncthreads();
  const uint32_t tile = Stile;
  kv_reqinfo r[RPT];
  size_t moff[RPT];
  uint64_t lm[RPT];
  uint32_t idx[RPT];
#pragma unroll
  for (int j = 0; j < RPT; j++) {
    idx[j] = tile * T + (uint32_t)j * TB + t;
    bool live;  // a segmented pass (multi-GPU exchange) has padding slots: they are no requests at all
    moff[j] = dint_view_off(A.V, idx[j] < n ? idx[j] : 0, F::MSG, &live);
    live = live && idx[j] < n;
    r[j] = kv_read_request<WL>(req + moff[j], live, kv, A.load_mode);
    lm[j] = __ballot(r[j].cls == 2);
    if (live && !r[j].cls) atomicAdd(&A.stats->bad_requests, 1ULL);
  }
  // log requests of this tile: publish the count at once (tatp / smallbank)
  if (WL != DINT_WL_STORE) {
    if (lane == 0) {
#pragma unroll
      for (int j = 0; j < RPT; j++) Swl[j * NWV + wv] = (uint32_t)__popcll(lm[j]);
    }
    __syncthreads();
    if (t == 0) {
      uint32_t c = 0;
      for (uint32_t w = 0; w < NWV * RPT; w++) c += Swl[w];
      __hip_atomic_store(&A.blk_pub[tile], 0x80000000u | c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  // copy this tile's messages to the reply array (replies are the request mutated in place; the bytes a request type
  // does not define are echoed, so the copy cannot be left to the resolve kernel without reading every message there
  // a second time).  All loads of a thread are issued before its first store.
  if (rep != req) {
    const size_t lo = (size_t)tile * T * F::MSG;
    const size_t hi = min((size_t)n * F::MSG, lo + (size_t)T * F::MSG);
    if ((((uintptr_t)req | (uintptr_t)rep) & 15) == 0) {
      constexpr uint32_t K = (T * F::MSG / 16 + TB - 1) / TB;
      const uint32_t nv = (uint32_t)((hi - lo) / 16);
      const uint4 *s = (const uint4 *)(req + lo);
      uint4 *d = (uint4 *)(rep + lo);
      if (nv) kv_copy_tile<TB>(s, d, nv, t, std::make_integer_sequence<uint32_t, K>());
      for (size_t k = lo + (size_t)nv * 16 + t; k < hi; k += TB) rep[k] = req[k];
    } else {
      for (size_t k = lo + t; k < hi; k += TB) rep[k] = req[k];
    }
  }

  uint32_t coarse[RPT], mypos[RPT];
  uint64_t meta[RPT];
#pragma unroll
  for (int j = 0; j < RPT; j++) {
    coarse[j] = KV_NONE; mypos[j] = 0; meta[j] = 0;
    if (r[j].cls == 1) {
      const uint64_t h = dint_hash_key(r[j].key);
      const uint64_t g = dint_fastmod(h, kv->mod[r[j].table]);
      uint32_t local = (uint32_t)g;
      bool mine = true;
      if (kv->shard_count > 1) {
        mine = (uint32_t)(g % kv->shard_count) == kv->shard_index;
        if (!mine && !A.load_mode) atomicAdd(&A.stats->foreign_requests, 1ULL);
        local = (uint32_t)(g / kv->shard_count);
      }
      if (mine) {
        // lock quadrant: lock_hash / hash_size, lock_hash = h % (4 * hash_size)
        const uint64_t hs = kv->mod[r[j].table].d, dq = dint_fastmod(h, kv->lockmod[r[j].table]) - g;  // 0, hs, 2hs or 3hs
        const uint32_t q = dq >= 2 * hs ? (dq >= 3 * hs ? 3u : 2u) : (dq >= hs ? 1u : 0u);
        const uint32_t gq = kv_cut_div(kv->gk_base[r[j].table] + local, A.cut, &coarse[j]);
        meta[j] = kv_rec(gq, idx[j], kv_pay(r[j].type, q, (uint32_t)(h >> 40) & 511u), A.cut);  // the table is implied by the group key
        mypos[j] = atomicAdd(&Hc[coarse[j]], 1u);
      }
    }
  }
  __syncthreads();
  for (uint32_t c = t; c < C; c += TB) {
    const uint32_t cnt = Hc[c];
    if (cnt) Hc[c] = atomicAdd(&A.bin_cnt[c], cnt);
  }
  __syncthreads();
  uint32_t orank[RPT];
  bool over[RPT];
#pragma unroll
  for (int j = 0; j < RPT; j++) {
    over[j] = false; orank[j] = 0;
    if (coarse[j] != KV_NONE) {
      const uint32_t pos = Hc[coarse[j]] + mypos[j];
      if (pos < A.cap) {
        A.kbins[(size_t)coarse[j] * A.cap + pos] = make_uint4((uint32_t)r[j].key, (uint32_t)(r[j].key >> 32), (uint32_t)meta[j], (uint32_t)(meta[j] >> 32));
      } else {  // a coarse bin that holds a hot key: one reservation in the pass's overflow list per tile
        over[j] = true;
        orank[j] = atomicAdd(&Sov[0], 1u);
      }
    }
  }
  __syncthreads();
  if (Sov[0]) {  // workgroup-uniform
    if (t == 0) Sov[1] = atomicAdd(&A.big[1], Sov[0]);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < RPT; j++)
      if (over[j]) {
        uint4 *o = A.ovl + 2 * (size_t)(Sov[1] + orank[j]);
        o[0] = make_uint4((uint32_t)r[j].key, (uint32_t)(r[j].key >> 32), (uint32_t)meta[j], (uint32_t)(meta[j] >> 32));
        o[1] = make_uint4(coarse[j], 0u, 0u, 0u);
      }
  }

  // ---- log requests: the canonical 64-byte record at ring position tail + (#log requests below i).  The tiles
  // before mine published their counts long ago (first thing they did); one count per thread, polled until it is there.
  if (WL != DINT_WL_STORE) {
    const dint_log log = A.log;
    uint32_t part = 0;
    for (uint32_t k = t; k < tile; k += TB) {  // (a pass has at most 1,024 tiles: one or two counts per thread)
      uint32_t v;
      do { v = __hip_atomic_load(&A.blk_pub[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while (!(v >> 31));
      part += v & 0x7FFFFFFFu;
    }
    uint32_t tot;
    wave_excl_scan_u32(part, &tot);
    if (lane == 0) Swp[wv] = tot;
    __syncthreads();
    uint32_t base = 0, tile_total = 0;
    for (uint32_t w = 0; w < NWV; w++) base += Swp[w];
    for (uint32_t w = 0; w < NWV * RPT; w++) tile_total += Swl[w];
    const uint32_t tw = A.pno & 1u;  // the ring position before this pass: tail[pno], after it: tail[pno ^ 1] (engine.hip, log_cur)
    if (tile == A.n_tiles - 1 && t == 0) {
      const uint32_t total = base + tile_total;
      log.tail[tw ^ 1u] = (uint32_t)(((uint64_t)log.tail[tw] + total) % log.cap);
      *(unsigned long long *)(log.tail + 2) += total;  // records ever appended (dint_log_drain)
    }
#pragma unroll
    for (int j = 0; j < RPT; j++) {
      if (r[j].cls != 2) continue;
      uint32_t pos_in_batch = base + (uint32_t)__popcll(lm[j] & lanemask_lt());
      for (uint32_t w = 0; w < (uint32_t)j * NWV + wv; w++) pos_in_batch += Swl[w];
      const uint32_t pos = (uint32_t)(((uint64_t)log.tail[tw] + pos_in_batch) % log.cap);
      uint8_t *e8 = log.ring + (size_t)pos * 64;
      const uint8_t *m = req + moff[j];
      const uint32_t ver = ld_u32(m + F::VER);
      uint8_t *rp = rep + moff[j];
      if (WL == DINT_WL_TATP && r[j].type == 24) {  // kDeleteLog: no val copy  (server_shard.cc:196-207)
        *(uint64_t *)e8 = r[j].key;
        *(uint2 *)(e8 + 48) = make_uint2(ver, 1u | (r[j].table << 8));
        rp[F::TYPE] = 27;
      } else {  // kCommitLog  (tatp server_shard.cc:182-194, smallbank server_shard.cc:175-186)
        // (scalars, not a word array: see kv_copy_tile)
        const uint8_t *v8 = m + F::VAL;
        uint4 *e4 = (uint4 *)e8;
        e4[0] = make_uint4((uint32_t)r[j].key, (uint32_t)(r[j].key >> 32), ld_u32(v8), ld_u32(v8 + 4));
        if (F::VS == 40) {
          e4[1] = make_uint4(ld_u32(v8 + 8), ld_u32(v8 + 12), ld_u32(v8 + 16), ld_u32(v8 + 20));
          e4[2] = make_uint4(ld_u32(v8 + 24), ld_u32(v8 + 28), ld_u32(v8 + 32), ld_u32(v8 + 36));
        }
        *(uint2 *)(e8 + 48) = make_uint2(ver, r[j].table << 8);
        rp[F::TYPE] = (WL == DINT_WL_TATP) ? 17 : 15;
      }
    }
  }
  // entries freed by earlier passes become reusable: the pend set THIS pass will push to (kv_pool_rotate).  LAST, by the
  // workgroup of the first block: the rotation is a chain of device-scope atomics per list (10 .. 20 us), and every tile's
  // log look-back waits for the counts of the tiles before it -- in front of tile 0's work (early r06) it delayed the whole
  // launch (k_kv_part 26 -> 39 us).  Nothing pops a free list before the next resolve stage, a launch later.
  if (first_block && t < KV_NLISTS)
    for (uint32_t k = 0; k < kv->n_tables; k++) kv_pool_rotate<kv_dev_mem>(kv->tab[k], t, (A.pno & 1u) * KV_NLISTS);
}

template <int WL, int RPT>
__global__ void __launch_bounds__(KV_TB) k_kv_part(kv_multi_args M) {
  const kv_pass_args &A = M.e[blockIdx.y];
  if (blockIdx.x >= A.n_tiles) return;
  __shared__ kv_part_lds<RPT> S;
  kv_part_body<WL, RPT>(A, S, blockIdx.x == 0);
}

// ---- optional per-wave timeline (DINT_KV_TRACE=1): lane 0 of every resolve wave stamps s_memtime at fixed
// points into trace[bin * 16 + k]; with tracing on, each stamp first drains the wave's memory queue so the
// difference of two stamps is the latency of what lies between them.  Off (nullptr) in normal runs.
__device__ static inline void kv_stamp(uint64_t *tr, uint32_t k) {
  if (tr) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    if (lane_id() == 0) tr[k] = __builtin_amdgcn_s_memrealtime();
  }
}
// device-wide constant-rate clock (100 MHz), comparable across waves: [10] = wave start, [11] = wave end
__device__ static inline void kv_stamp_real(uint64_t *tr, uint32_t k) {
  if (tr && lane_id() == 0) tr[k] = __builtin_amdgcn_s_memrealtime();
}

// The overflow-entry pool is nearly used up: a bucket run that inserts goes request by request (kv_do_request answers
// an INSERT that finds the pool full with the reject code and leaves its lock byte alone; a closed form has written
// its replies before the one physical insert at the write-back can fail -- ADVICE r02).  The margin covers the
// inserts other waves have in flight; recycled entries are ignored, which only makes the answer more careful.
#define KV_POOL_MARGIN 4096u
__device__ static inline uint32_t kv_pool_low(const kv_tab &t) {
  return __hip_atomic_load(t.pool_top, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + KV_POOL_MARGIN >= t.pool_cap;
}

// ---- one request against the table ---------------------------------------------------------------------
// Written so that the lanes of a wave, which run different request types, share their memory round trips:
//   load phase   : the bucket's inline header sector (probe keys, versions, valid bits, chain head AND the tatp lock
//                  bytes), the request's key, and the smallbank counter pair -- three independent loads, one wait;
//   decide phase : registers only -- which table action (GET / SET / INS / DEL / none), the lock transition, the
//                  reply code;
//   act phase    : kv_apply (value copy, row / header stores) and the lock-word store.
template <int WL>
__device__ static inline void kv_do_request(uint8_t *msg, uint32_t type, uint32_t table, uint32_t q, uint64_t bucket,
                                            const kv_dev *kv, dint_dev_stats *__restrict__ stats) {
  using F = Fmt<WL>;
  const kv_tab t = kv->tab[table];
  uint8_t *ie = kv_entry_ptr(t, bucket, KV_INLINE);
  // ---- load phase
  kv_hdr H;
  kv_hdr_load(H, ie);
  uint2 cnt = make_uint2(0, 0);
  if (WL == DINT_WL_SMALLBANK) cnt = *(const uint2 *)(ie + KV_SB_LOCK_OFF + 8 * q);  // {num_ex, num_sh}
  const uint64_t key = ld_u64(msg + F::KEY);
  uint8_t *val = msg + F::VAL;

  // ---- decide phase
  uint32_t act = KV_ACT_NONE, code = 0, ins_ver = 0;
  bool miss_counts = false;   // a miss of this action is an event the reference panics on
  int lock_store = -1;        // tatp: byte to store into lock byte q (-1 = none)
  bool cnt_store = false;     // smallbank: store the counter pair back
  if (type == DINT_KV_LOAD_OP) {  // bulk load: kvs_insert with the version carried in the message
    act = KV_ACT_INS;
    ins_ver = ld_u32(msg + F::VER);
  } else if (WL == DINT_WL_STORE) {
    switch (type) {
      case 0: act = KV_ACT_GET; break;   // kRead  store/udp/server.cc:77-82
      case 1: act = KV_ACT_SET; break;   // kSet   :84-89
      default: act = KV_ACT_INS; code = 8; break;  // kInsert: engine extension (kvs_insert semantics; no reference parity target, see dint_abi.h)
    }
  } else if (WL == DINT_WL_TATP) {
    const uint32_t lk = (H.lockw >> (8 * q)) & 0xFFu;
    switch (type) {
      case 0: act = KV_ACT_GET; break;                                                    // kRead  server_shard.cc:116-121
      case 1:                                                                             // kAcquireLock  :123-132
        if (lk == 0) { lock_store = 1; code = 7; }
        else code = (kv->same_key && *(const uint64_t *)(ie + KV_OWNER_OFF + 8 * q) == key) ? 28 : 8;  // lock_kern.c:289-298
        break;
      case 2: lock_store = 0; code = 9; break;                                            // kAbort  :134-138
      case 12: act = KV_ACT_SET; miss_counts = true; lock_store = 0; code = 15; break;    // kCommitPrim  :140-146
      case 18: act = KV_ACT_INS; lock_store = 0; code = 20; break;                        // kInsertPrim  :148-154
      case 22: act = KV_ACT_DEL; miss_counts = true; lock_store = 0; code = 25; break;    // kDeletePrim  :156-162
      case 13: act = KV_ACT_SET; miss_counts = true; code = 16; break;                    // kCommitBck   :164-168
      case 19: act = KV_ACT_INS; code = 21; break;                                        // kInsertBck   :170-174
      default: act = KV_ACT_DEL; miss_counts = true; code = 26; break;                    // 23 kDeleteBck  :176-180
    }
    if (lock_store >= 0 && (uint32_t)lock_store == lk) lock_store = -1;  // unchanged byte: no store
  } else {
    switch (type) {  // cnt.x = num_ex, cnt.y = num_sh   smallbank/udp/server_shard.cc:121-173
      case 0: if (cnt.x == 0) { cnt.y++; cnt_store = true; act = KV_ACT_GET; miss_counts = true; code = 7; } else code = 8; break;
      case 1: if (cnt.x == 0 && cnt.y == 0) { cnt.x++; cnt_store = true; act = KV_ACT_GET; miss_counts = true; code = 9; } else code = 10; break;
      case 2: cnt.y--; cnt_store = true; code = 11; break;
      case 3: cnt.x--; cnt_store = true; code = 12; break;
      case 4: act = KV_ACT_SET; miss_counts = true; code = 13; break;
      case 17: act = KV_ACT_GET; code = 18; break;  // WARMUP_READ (eBPF flavour): kvs_get, ack whether found or not
      default: act = KV_ACT_SET; miss_counts = true; code = 14; break;  // 5 kCommitBck
    }
  }

  // ---- act phase
  const kv_res r = kv_apply<kv_dev_mem>(t, bucket, H, act, key, val, ins_ver, blockIdx.x);
  if (act == KV_ACT_INS && !r.ok && type != DINT_KV_LOAD_OP) {
    // the overflow-entry pool is full: nothing was stored and the request is refused -- store kRejectInsert
    // (store/udp/net.h:28), tatp REJECT_COMMIT, the eBPF flavour's "send again" (tatp/ebpf/shard_kern.c:509-514,
    // taken there before the lock word is touched)
    code = WL == DINT_WL_STORE ? 9 : 11;
    lock_store = -1;
  }
  if (WL == DINT_WL_TATP && lock_store >= 0) ie[KV_LOCKB_OFF + q] = (uint8_t)lock_store;
  if (WL == DINT_WL_TATP && kv->same_key && type == 1 && code == 7) *(uint64_t *)(ie + KV_OWNER_OFF + 8 * q) = key;
  if (WL == DINT_WL_SMALLBANK && cnt_store) *(uint2 *)(ie + KV_SB_LOCK_OFF + 8 * q) = cnt;
  if (act == KV_ACT_GET && r.ok) st_u32(msg + F::VER, r.ver);
  if (act != KV_ACT_NONE && !r.ok) {
    if (act == KV_ACT_INS) atomicAdd(&stats->pool_exhausted, 1ULL);
    else if (miss_counts) atomicAdd(&stats->missing_keys, 1ULL);
  }
  if (type == DINT_KV_LOAD_OP) return;  // internal request: no reply
  if (WL == DINT_WL_STORE && type <= 1) code = r.ok ? (type == 0 ? 3 : 5) : 7;  // GRANT_READ / SET_ACK / NOT_EXIST
  if (WL == DINT_WL_TATP && type == 0) code = r.ok ? 4 : 6;                      // GRANT_READ / NOT_EXIST
  msg[F::TYPE] = (uint8_t)code;
}

// ---- one 64-chunk of a bin's requests ----------------------------------------------------------------------
// Precondition: the chunk's lanes are sorted by (bucket group, key-hash bits, idx): the requests of one bucket sit
// in adjacent lanes (a BUCKET RUN), inside it the requests of one key sit in adjacent lanes, in request order (a
// KEY SEGMENT; valid lanes first).  Runs of different buckets commute.  Inside a bucket:
//   - rows of different keys are independent (GET / SET touch one row);
//   - the lock word (tatp lock byte, smallbank counters) is shared by the keys that map to the same quadrant;
//   - INSERT / DELETE change the chain the other keys are found through.
// So a bucket run is "simple" when every request is a chain-preserving op, every key segment really holds one
// key, and at most one of its key segments carries lock ops.  Every key segment of a simple run is then
// resolved on its own and ALL of them at once:
//   1. every segment head loads the bucket's inline header (and smallbank counters) and locates its row;
//   2. every lane derives its own reply from ballots restricted to its segment's lane mask (store / tatp):
//        version seen = ver0 + #writers below in the segment, value seen = message of the last writer below,
//        lock seen    = what the last ACQUIRE (-> 1) / ABORT / COMMIT_PRIM (-> 0) below wrote, else the stored byte;
//      smallbank's shared / exclusive counters have no closed form: single requests apply their op directly,
//      longer segments are walked once each with wave-uniform registers (no memory inside the walk);
//   3. replies are written, reads copy their value from the last writer's message or from the table row;
//   4. after a fence the segment heads write the final row / version / lock word once.
// Four memory round trips per chunk however many requests collide.  Any other bucket run (inserts, deletes,
// lock ops on two keys, key-hash collisions) executes request by request in rounds, in idx order.
// Semantics per op: the same reference lines as kv_do_request.
template <int WL>
__device__ static inline bool kv_simple_op(uint32_t type) {
  if (WL == DINT_WL_STORE) return type <= 1;                                   // READ, SET
  if (WL == DINT_WL_TATP) return type <= 2 || type == 12 || type == 13;        // READ, ACQUIRE, ABORT, COMMIT_PRIM/BCK
  return type <= 5 || type == 17;                                              // every smallbank table op
}
template <int WL>
__device__ static inline bool kv_struct_op(uint32_t type) {  // inserts / deletes a row (changes the chain)
  if (WL == DINT_WL_STORE) return type == 2;
  if (WL == DINT_WL_TATP) return type == 18 || type == 19 || type == 22 || type == 23;
  return false;
}
// requests that change nothing (store / tatp READ, smallbank WARMUP_READ): in the request-by-request fallback the reads
// between two other requests of a bucket run share one round -- they only have to see what came before them
template <int WL>
__device__ static inline bool kv_pure_read(uint32_t type) { return WL == DINT_WL_SMALLBANK ? type == 17 : type == 0; }
template <int WL>
__device__ static inline bool kv_lock_op(uint32_t type) {  // touches the bucket's lock word
  if (WL == DINT_WL_STORE) return false;
  if (WL == DINT_WL_TATP) return type == 1 || type == 2 || type == 12 || type == 18 || type == 22;
  return type <= 3;
}
// requests whose reply may carry a row (val + ver) / whose message carries a value that is stored or read by others
template <int WL>
__device__ static inline bool kv_may_get(uint32_t type) {
  if (WL == DINT_WL_SMALLBANK) return type <= 1 || type == 17;  // granted ACQUIREs, WARMUP_READ
  return type == 0;                                               // READ
}
template <int WL>
__device__ static inline bool kv_carries_val(uint32_t type) {
  if (WL == DINT_WL_STORE) return type == 1 || type == 2;                                 // SET, INSERT
  if (WL == DINT_WL_TATP) return type == 12 || type == 13 || type == 18 || type == 19;    // COMMIT_*, INSERT_*
  return type == 4 || type == 5;                                                          // COMMIT_*
}
// One step of a key's row machine {exists, version, last writer} (store / tatp), in request order.  Used for key
// segments that contain an INSERT or DELETE; plain segments use the ballot closed form.  Returns the reply code of
// a row op (0 for lock-only requests, whose code comes from the lock machine).
struct kv_rowst { uint32_t exists, ver, toggles, miss, bail; int src; };
template <int WL>
__device__ static inline uint32_t kv_row_step(uint32_t op, int l, kv_rowst &st, uint32_t &get) {
  enum { GET, SET, INS, DEL, NONE } a = NONE;
  uint32_t code = 0;
  if (WL == DINT_WL_STORE) {
    if (op == 0) { a = GET; code = st.exists ? 3 : 7; }
    else if (op == 1) { a = SET; code = st.exists ? 5 : 7; }
    else { a = INS; code = 8; }
  } else {
    switch (op) {
      case 0: a = GET; code = st.exists ? 4 : 6; break;
      case 12: a = SET; code = 15; break;
      case 13: a = SET; code = 16; break;
      case 18: a = INS; code = 20; break;
      case 19: a = INS; code = 21; break;
      case 22: a = DEL; code = 25; break;
      case 23: a = DEL; code = 26; break;
      default: break;  // 1 kAcquireLock, 2 kAbort: lock word only
    }
  }
  switch (a) {
    case GET: get = st.exists; break;
    case SET: if (st.exists) { st.ver++; st.src = l; } else if (WL != DINT_WL_STORE) st.miss++; break;
    case INS: if (st.exists) st.bail = 1; else { st.exists = 1; st.ver = 0; st.src = l; st.toggles++; } break;
    case DEL: if (st.exists) { st.exists = 0; st.toggles++; } else st.miss++; break;
    default: break;
  }
  return code;
}

__device__ static inline uint64_t shfl_u64(uint64_t v, int src) {
  const uint32_t hi = (uint32_t)__shfl((uint32_t)(v >> 32), src, 64), lo = (uint32_t)__shfl((uint32_t)v, src, 64);
  return ((uint64_t)hi << 32) | lo;
}
__device__ static inline uint64_t readlane_u64(uint64_t v, int l) {
  // the builtin returns int: without the casts the low word would be sign-extended over the high one
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((uint32_t)(v >> 32), l);
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((uint32_t)v, l);
  return ((uint64_t)hi << 32) | lo;
}
// lanes [my run's head, next head): heads = ballot of run heads, le = lanes <= me, vm = ballot(valid)
__device__ static inline uint64_t run_mask(uint64_t heads, uint64_t le, uint64_t vm, int *head_lane) {
  const int hl = 63 - __clzll(heads & le);
  const uint64_t above = heads & ~le;
  const uint64_t next = above ? (above & (~above + 1ull)) : vm + 1ull;  // first invalid lane = bit nvalid = vm + 1
  *head_lane = hl;
  return (next - 1ull) & ~((1ull << hl) - 1ull);
}

template <int WL>
__device__ static inline void kv_chunk(uint8_t *rep, bool valid, uint32_t idx, uint32_t gk, uint32_t kh, uint32_t type,
                                       uint32_t table, uint32_t q, uint64_t key_in, const kv_dev *kv,
                                       dint_dev_stats *__restrict__ stats, int force_rounds, bool last_chunk, const dint_view &V,
                                       uint64_t *tr = nullptr) {
  using F = Fmt<WL>;
  const int lane = (int)lane_id();
  const uint64_t lt = lanemask_lt(), le = lt | (1ull << lane);
  uint8_t *msg = rep + dint_view_off(V, idx, F::MSG);
  const uint64_t key = valid ? key_in : 0;  // the record carries the key: no gather from the message array
  const uint64_t bucket = valid ? (uint64_t)(gk - kv->gk_base[table]) : 0;

  // ---- bucket runs and key segments
  const uint32_t gk_up = __shfl_up(gk, 1, 64), kh_up = __shfl_up(kh, 1, 64);
  const bool bhead = valid && (lane == 0 || gk_up != gk);
  const bool head = valid && (bhead || kh_up != kh);
  const uint64_t vm = __ballot(valid);
  const uint64_t bhm = __ballot(bhead), hm = __ballot(head);
  int hl = lane, bhl = lane;
  const uint64_t seg = valid ? run_mask(hm, le, vm, &hl) : 0;   // my key segment
  const uint64_t run = valid ? run_mask(bhm, le, vm, &bhl) : 0; // my bucket run

  // ---- 1. segment heads: load the bucket's inline header (and smallbank counters) -- issued before the keys
  // are compared, so the header round trip overlaps the key round trip -- then locate the row
  uint32_t found = 0, link = 0, slot = 0, ver0 = 0, la0 = 0, lb0 = 0;
  kv_tab t;
  uint8_t *ie = nullptr;
  kv_hdr H;
  if (valid) {
    t = kv->tab[table];
    ie = kv_entry_ptr(t, bucket, KV_INLINE);
  }
  if (head) {
    kv_hdr_load(H, ie);
    if (WL == DINT_WL_SMALLBANK) {
      la0 = KV_LD(uint32_t, ie + KV_SB_LOCK_OFF + 8 * q); lb0 = KV_LD(uint32_t, ie + KV_SB_LOCK_OFF + 8 * q + 4);
    }
  }
  // The value words of a request live in registers from here on (r04b: two dependent round trips and a fence less
  // per chunk).  A request that CARRIES a value (SET / COMMIT / INSERT) loads it now, together with the headers: reads
  // behind it in the same pass take it by shuffle, and the segment's last writer stores it into the row itself.  A
  // request that may READ a row loads it speculatively as soon as the header is there (below).
  constexpr uint32_t NW = F::VS / 4;
  uint32_t w[NW];
#pragma unroll
  for (uint32_t k = 0; k < NW; k++) w[k] = 0;
  if (valid && kv_carries_val<WL>(type)) {
#pragma unroll
    for (uint32_t k = 0; k < NW; k++) w[k] = ld_u32(msg + F::VAL + 4 * k);
  }
  const uint64_t hkey = shfl_u64(key, hl);
  const uint64_t m_bad = __ballot(valid && !(key == hkey && (kv_simple_op<WL>(type) || kv_struct_op<WL>(type))));
  const uint64_t m_lockop = __ballot(valid && kv_lock_op<WL>(type));
  const uint64_t m_struct = __ballot(valid && kv_struct_op<WL>(type));
  // key segments that carry lock ops, per lock quadrant: two of them on one lock word make the run non-simple
  const bool lkseg = head && (m_lockop & seg) != 0;
  bool lock_clash = false;  // several key segments of my bucket run use one lock word ...
  bool my_clash = false;    // ... the one of my quadrant
#pragma unroll
  for (uint32_t k = 0; k < 4; k++) {
    const bool cl = __popcll(__ballot(lkseg && q == k) & run) > 1;
    lock_clash |= cl;
    my_clash |= cl && q == k;
  }
  // tatp's lock byte is a last-writer-wins register (ACQUIRE leaves 1, every other lock op 0; only ACQUIRE's reply
  // depends on it), so sharing it does not serialise the run: see the lock machine below.  smallbank's counters do.
  if (WL == DINT_WL_TATP) lock_clash = false;
  const uint64_t m_stseg = __ballot(head && (m_struct & seg) != 0);  // ... that insert / delete their row
  bool simple = valid && (m_bad & run) == 0 && !lock_clash && __popcll(m_stseg & run) <= 1 && !force_rounds;
  const bool structural = (m_struct & seg) != 0;  // my key segment inserts / deletes: row machine by walk
  kv_stamp(tr, 4);
  // the row, speculatively: where the key sits in the bucket's INLINE entry (the usual case) the readers of the segment
  // load it now, while the leaders walk the chains of the buckets that overflowed (kv_locate: another round trip) --
  // whether the row really is the first match in chain order is known when they are back
  int spec = -1;
  if (head) {
#pragma unroll
    for (int i = 0; i < 4; i++)
      if (spec < 0 && kv_valid(H, (uint32_t)i) && H.key[i] == key) spec = i;
  }
  spec = __shfl(spec, hl, 64);
  const bool spec_ld = valid && spec >= 0 && kv_may_get<WL>(type);
  if (spec_ld) {
    const uint8_t *sr = ie + KV_VAL_OFF + (uint32_t)spec * F::VS;
#pragma unroll
    for (uint32_t k = 0; k < NW; k++) w[k] = KV_LD(uint32_t, sr + 4 * k);
  }
  bool leader = head && simple;
  uint32_t dupf = 0;  // a second row with my key exists (duplicate inserts of an earlier pass): no closed form for deletes
  if (leader) {
    if (WL == DINT_WL_TATP) la0 = (H.lockw >> (8 * q)) & 0xFFu;
    const kv_where w = kv_locate(t, bucket, H, key);
    found = w.found; link = w.link; slot = w.slot; ver0 = w.ver;
    if (WL != DINT_WL_SMALLBANK && structural) dupf = kv_has_dup(t, bucket, H, key, w) | kv_pool_low(t);
  }
  found = __shfl(found, hl, 64); link = __shfl(link, hl, 64); slot = __shfl(slot, hl, 64);
  ver0 = __shfl(ver0, hl, 64); la0 = __shfl(la0, hl, 64); lb0 = __shfl(lb0, hl, 64);
  kv_stamp(tr, 5);

  // ---- 2. outcome of every request of a simple segment
  uint32_t my_code = 0, my_ver = 0, my_get = 0;   // my_get: the reply carries val + ver
  int my_src = -1;                                // lane whose message holds the value this lane reads (-1: the table)
  uint32_t fin_ver = ver0, fin_la = la0, fin_lb = lb0, nmiss = 0;  // segment totals (meaningful on the head lane)
  int fin_src = -1;
  {
    const bool writer = simple && (WL == DINT_WL_STORE ? type == 1 : WL == DINT_WL_TATP ? (type == 12 || type == 13)
                                                                                       : (type == 4 || type == 5));
    const uint64_t m_wr = __ballot(writer && found) & seg;
    if (WL != DINT_WL_SMALLBANK) {
      const uint64_t wr_below = m_wr & lt;
      my_ver = ver0 + (uint32_t)__popcll(wr_below);
      my_src = wr_below ? 63 - __clzll(wr_below) : -1;
      fin_ver = ver0 + (uint32_t)__popcll(m_wr);
      fin_src = m_wr ? 63 - __clzll(m_wr) : -1;
      if (WL == DINT_WL_STORE) {
        my_code = type == 0 ? (found ? 3 : 7) : (found ? 5 : 7);
        my_get = (type == 0 && found) ? 1 : 0;
      } else {
        const uint64_t m_lk = __ballot(simple && kv_lock_op<WL>(type)) & seg;
        const uint64_t m_acq = __ballot(simple && type == 1);
        const uint64_t lk_below = m_lk & lt;
        uint32_t lock_seen = lk_below ? (uint32_t)((m_acq >> (63 - __clzll(lk_below))) & 1ull) : la0;
        if (m_lk) fin_la = (uint32_t)((m_acq >> (63 - __clzll(m_lk))) & 1ull);
        // a lock byte shared by several keys of the run: the latest lock op on it that precedes me in request
        // order, looked up among the run's lock ops (one wave-uniform step per such op; rare)
        const uint64_t cm = __ballot(simple && my_clash && kv_lock_op<WL>(type));
        if (cm) {
          int seen = -1, fin = -1;
          uint32_t seen_at = 0, fin_at = 0;
          for (uint64_t mm = cm; mm; mm &= mm - 1) {
            const int l = __ffsll((unsigned long long)mm) - 1;
            const uint32_t oi = (uint32_t)__builtin_amdgcn_readlane(idx, l), og = (uint32_t)__builtin_amdgcn_readlane(gk, l);
            const uint32_t oq = (uint32_t)__builtin_amdgcn_readlane(q, l);
            const int acq = (uint32_t)__builtin_amdgcn_readlane(type, l) == 1;
            if (og == gk && oq == q) {
              if (oi < idx && (seen < 0 || oi > seen_at)) { seen_at = oi; seen = acq; }
              if (fin < 0 || oi > fin_at) { fin_at = oi; fin = acq; }
            }
          }
          if (simple && my_clash) {
            lock_seen = seen >= 0 ? (uint32_t)seen : la0;
            if (m_lk && fin >= 0) fin_la = (uint32_t)fin;  // every segment of the group stores the same byte
          }
        }
        nmiss = found ? 0 : (uint32_t)__popcll(__ballot(writer) & seg);
        switch (type) {
          case 0: my_code = found ? 4 : 6; my_get = found; break;
          case 1: my_code = lock_seen ? 8 : 7; break;
          case 2: my_code = 9; break;
          case 12: my_code = 15; break;
          default: my_code = 16; break;  // 13 kCommitBck (a structural segment overwrites the row ops' codes below)
        }
      }
      // key segments with an INSERT / DELETE: the row machine {exists, version, last writer} is walked once per
      // segment with wave-uniform registers.  A row that is deleted and inserted again may move to another slot:
      // the write-back then deletes and re-inserts once (further pairs leave the row where the first put it).  An
      // INSERT of an existing key (a duplicate row) sends the bucket run to the rounds.
      uint32_t fin_exists = found, fin_multi = 0, my_bail = 0;
      uint64_t stm = __ballot(leader && structural);
      while (stm) {
        const int L = __ffsll((unsigned long long)stm) - 1;
        stm &= stm - 1;
        const uint64_t sm = readlane_u64(seg, L);
        kv_rowst st;
        st.exists = (uint32_t)__builtin_amdgcn_readlane(found, L);
        st.ver = (uint32_t)__builtin_amdgcn_readlane(ver0, L);
        st.toggles = 0; st.miss = 0; st.bail = (uint32_t)__builtin_amdgcn_readlane(dupf, L); st.src = -1;
        for (uint64_t m = sm; m; m &= m - 1) {
          const int l = __ffsll((unsigned long long)m) - 1;
          const uint32_t op = (uint32_t)__builtin_amdgcn_readlane(type, l);
          uint32_t get = 0;
          const uint32_t ver_seen = st.ver;
          const int src_seen = st.src;
          const uint32_t code = kv_row_step<WL>(op, l, st, get);
          if (lane == l) {
            if (code) my_code = code;  // lock-only requests keep the lock machine's code
            my_ver = ver_seen; my_src = src_seen; my_get = get;
          }
        }
        if (lane == L) {
          fin_exists = st.exists; fin_ver = st.ver; fin_src = st.src; nmiss = st.miss;
          fin_multi = st.toggles > 1;
          my_bail = st.bail;
        }
      }
      const uint64_t m_bail = __ballot(head && my_bail);
      if (m_bail & run) { simple = false; leader = false; }
      // head lane: bit 1 = the row exists after the segment, bit 2 = it was deleted and inserted on the way
      found = simple && structural ? (found | (fin_exists << 1) | (fin_multi << 2)) : found;
    } else {
      // smallbank.  cnt = {la: num_ex, lb: num_sh}
      auto sb_step = [](uint32_t op, uint32_t fnd, uint32_t &la, uint32_t &lb, uint32_t &get, uint32_t &miss,
                        bool &wr) -> uint32_t {
        switch (op) {
          case 0: if (la == 0) { lb++; get = fnd; miss += !fnd; return 7; } return 8;
          case 1: if (la == 0 && lb == 0) { la++; get = fnd; miss += !fnd; return 9; } return 10;
          case 2: lb--; return 11;
          case 3: la--; return 12;
          case 4: wr = fnd; miss += !fnd; return 13;
          case 17: get = fnd; return 18;  // WARMUP_READ -> WARMUP_READ_ACK: a plain read (smallbank/ebpf/shard_user.c:179-186)
          default: wr = fnd; miss += !fnd; return 14;  // 5 kCommitBck
        }
      };
      const bool single = simple && seg == (1ull << lane);
      if (single) {  // one request on its key: apply it directly
        bool wr = false;
        my_code = sb_step(type, found, fin_la, fin_lb, my_get, nmiss, wr);
        my_ver = ver0;
        if (wr) { fin_ver = ver0 + 1; fin_src = lane; }
      }
      uint64_t multi = __ballot(leader && !single);
      while (multi) {  // longer segments: one wave-uniform walk each, registers only
        const int L = __ffsll((unsigned long long)multi) - 1;
        multi &= multi - 1;
        const uint64_t sm = readlane_u64(seg, L);
        const uint32_t fnd = (uint32_t)__builtin_amdgcn_readlane(found, L);
        uint32_t la = (uint32_t)__builtin_amdgcn_readlane(la0, L), lb = (uint32_t)__builtin_amdgcn_readlane(lb0, L);
        uint32_t ver = (uint32_t)__builtin_amdgcn_readlane(ver0, L), miss = 0;
        int src = -1;
        for (uint64_t m = sm; m; m &= m - 1) {
          const int l = __ffsll((unsigned long long)m) - 1;
          const uint32_t op = (uint32_t)__builtin_amdgcn_readlane(type, l);
          uint32_t get = 0;
          bool wr = false;
          const uint32_t ver_seen = ver;
          const int src_seen = src;
          const uint32_t code = sb_step(op, fnd, la, lb, get, miss, wr);
          if (wr) { ver++; src = l; }
          if (lane == l) { my_code = code; my_ver = ver_seen; my_src = src_seen; my_get = get; }
        }
        if (lane == L) { fin_la = la; fin_lb = lb; fin_ver = ver; fin_src = src; nmiss = miss; }
      }
    }
  }

  // ---- 3. replies of the simple segments, all lanes in parallel
  const uint32_t fin_idx = __shfl(idx, fin_src >= 0 ? fin_src : lane, 64);
  const int seg_fin_src = __shfl(fin_src, hl, 64);          // my segment's last writer ...
  const uint32_t seg_fin_ver = __shfl(fin_ver, hl, 64);     // ... and the version it leaves
  const bool getting = simple && my_get != 0;
  if (__ballot(getting && my_src >= 0)) {  // (wave-uniform) the value of the last writer below me: out of that lane's registers
#pragma unroll
    for (uint32_t k = 0; k < NW; k++) {
      const uint32_t o = (uint32_t)__shfl((int)w[k], my_src >= 0 ? my_src : lane, 64);
      if (getting && my_src >= 0) w[k] = o;
    }
  }
  uint8_t *row = nullptr;
  if (simple) {
    row = kv_entry_ptr(t, bucket, link) + KV_VAL_OFF + slot * F::VS;  // meaningful when the row was found
    if (my_get) {
      // the row itself: the speculative load was the right one unless the key's first match lies in an overflow entry
      if (my_src < 0 && !(spec_ld && link == KV_INLINE && slot == (uint32_t)spec)) {
#pragma unroll
        for (uint32_t k = 0; k < NW; k++) w[k] = KV_LD(uint32_t, row + 4 * k);
      }
#pragma unroll
      for (uint32_t k = 0; k < NW; k++) st_u32(msg + F::VAL + 4 * k, w[k]);
      st_u32(msg + F::VER, my_ver);
    }
    msg[F::TYPE] = (uint8_t)my_code;
    // ---- 4a. the row of a plain segment: value and version of its last writer, stored by that writer (every load of
    // the chunk has been consumed by now, so nothing has to be waited for between the replies and these stores)
    if (!structural && (found & 1u) && seg_fin_src == lane) {
#pragma unroll
      for (uint32_t k = 0; k < NW; k++) KV_ST(uint32_t, row + 4 * k, w[k]);
      KV_ST(uint32_t, &kv_entry_hdr(t, bucket, link)->ver[slot], seg_fin_ver);
    }
  }
  kv_stamp(tr, 6);
  // ---- 4b. the rest of each simple segment's final state, written once by its head.  The lock word belongs to the one
  // segment of the bucket that carries lock ops (fin_la / fin_lb differ from la0 / lb0 only there).  Segments that
  // insert / delete apply their net effect to the chain here, behind a fence (rare: the branch is wave-uniform).
  if (__ballot(leader && structural)) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");  // the table reads above precede the chain operations below
  if (leader) {
    const uint32_t found0 = found & 1u, exists1 = structural ? (found >> 1) & 1u : found0;
    const bool redo = structural && found0 && exists1 && ((found >> 2) & 1u);  // deleted and inserted again
    if (found0 && exists1 && !redo) {  // the row stays where it is: value / version of the last writer
      if (structural && fin_src >= 0) {
        kv_copy_words(row, rep + dint_view_off(V, fin_idx, F::MSG) + F::VAL, F::VS);
        KV_ST(uint32_t, &kv_entry_hdr(t, bucket, link)->ver[slot], fin_ver);
      }
    } else if (found0 != exists1 || redo) {  // apply the net INSERT / DELETE (or DELETE + INSERT) to the chain once
      if (redo) {
        kv_apply<kv_dev_mem>(t, bucket, H, KV_ACT_DEL, key, nullptr, 0, blockIdx.x);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        kv_hdr_load(H, ie);
      }
      const kv_res r = kv_apply<kv_dev_mem>(t, bucket, H, exists1 ? KV_ACT_INS : KV_ACT_DEL, key,
                                            rep + dint_view_off(V, fin_idx, F::MSG) + F::VAL, fin_ver, blockIdx.x);
      if (exists1 && !r.ok) atomicAdd(&stats->pool_exhausted, 1ULL);
    }
    if (WL == DINT_WL_TATP && fin_la != la0) KV_ST(uint8_t, ie + KV_LOCKB_OFF + q, (uint8_t)fin_la);
    if (WL == DINT_WL_SMALLBANK && (fin_la != la0 || fin_lb != lb0)) {
      KV_ST(uint32_t, ie + KV_SB_LOCK_OFF + 8 * q, fin_la);
      KV_ST(uint32_t, ie + KV_SB_LOCK_OFF + 8 * q + 4, fin_lb);
    }
    if (nmiss) atomicAdd(&stats->missing_keys, (unsigned long long)nmiss);
  }
  kv_stamp(tr, 7);

  // ---- every other bucket run: request by request, in request order (the run is sorted by key first).  The w-th
  // request of the run that is not a pure read executes alone in round 2w + 1; the reads that follow it (and precede
  // the next such request) share round 2w + 2: they change nothing and only have to see what came before them.
  const bool rounds = valid && !simple;
  uint64_t rheads = __ballot(rounds && bhead);
  if (rheads) {
    const uint64_t m_rd = __ballot(valid && kv_pure_read<WL>(type));
    uint32_t pos = 0, maxlen = 0;
    while (rheads) {
      const int L = __ffsll((unsigned long long)rheads) - 1;
      rheads &= rheads - 1;
      const uint64_t rmask = readlane_u64(run, L);
      maxlen = max(maxlen, 2u * (uint32_t)__popcll(rmask & ~m_rd) + 1u);
      const bool mine = (rmask >> lane) & 1ull;
      for (uint64_t m = rmask & ~m_rd; m; m &= m - 1) {
        const int l = __ffsll((unsigned long long)m) - 1;
        const uint32_t oi = (uint32_t)__builtin_amdgcn_readlane(idx, l);
        if (mine && oi < idx) pos++;
      }
    }
    pos = 2u * pos + (kv_pure_read<WL>(type) ? 0u : 1u);
    for (uint32_t r = 0; r < maxlen; r++) {
      if (rounds && pos == r) kv_do_request<WL>(msg, type, table, q, bucket, kv, stats);
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");  // the next round must see this round's stores
    }
  }
  if (!last_chunk) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");  // ... and the next chunk this chunk's
  kv_stamp(tr, 8);
}


// ---- LDS of the big path: one struct, so that the coarse-bin split (kvr_lds) can share the buffer ---------------------
// the index-bitmap ordering of a one- or two-key stretch (kv_big_bin): bitmap words, their popcount table, the ordered copy
constexpr uint32_t KVB_BM_W = 4096, KVB_BM_OT = 1024, KVB_BM_MIN = 1024;
constexpr uint32_t KVB_BM_BYTES = KVB_BM_W * 8 + KVB_BM_W * 2 + KVB_NMAX * 8;

constexpr bool KV_HOT_BM = false;          // dominant-key path of kv_big_bin: order the key's writers + lock ops by an index bitmap (else: LDS sort + binary
                                           // search).  r05: off -- store / tatp answer their hot keys in k_kv_hot and kv_big_bin only takes what that
                                           // leaves (usually nothing), and the bitmap is 80 KB of LDS.  smallbank does not use the path.
constexpr uint32_t KV_HOT_BM_W = 8192;     // ... bitmap words: request-index spans of up to 512k
constexpr uint32_t KV_MMAX = 2048;  // dominant-key path: writers + lock ops of the key it puts in order (a tatp subscriber of 4,000 requests: ~1,100)

struct kvb_lds {
  uint64_t Sk[KVB_NMAX];           // the stretch: group / P | key-hash bits | idx | type, quadrant
  uint32_t Bcnt[KVB_NBK / 2];      // records per idx bucket, 16 bits each (a bucket spans <= 512 requests)
  uint16_t Bwin[KVB_NBK];          // stretch each idx bucket belongs to
  uint64_t Mhead[KVB_NW], Mbh[KVB_NW], Mbad[KVB_NW], Mlop[KVB_NW], Mst[KVB_NW], Mlkseg[4][KVB_NW], Mstseg[KVB_NW], Mrs[KVB_NW],
      Msimple[KVB_NW], Mwr[KVB_NW], Mlk[KVB_NW], Macq[KVB_NW];
  kvb_pop Pbad, Plop, Pst, Plkseg[4], Pstseg, Pwr, Phead;
  kvb_edge Ehead, Ebh, Ewr, Elk;
  uint64_t Mbail[KVB_W];           // tile-local
  __attribute__((aligned(8))) kvb_lead Lead[KVB_T];  // by key segment number (a stretch with more segments runs request by request)
  static constexpr uint32_t CC_ROWS = KVB_T * (sizeof(kv_rowst) > sizeof(kvb_carry) ? sizeof(kv_rowst) : sizeof(kvb_carry));
  static constexpr uint32_t CC_TABS = 3 * (KV_MMAX + 8) * 2;  // ... or the dominant-key path's three prefix tables
  __attribute__((aligned(8))) uint8_t CarryCrow[(CC_ROWS > CC_TABS ? CC_ROWS : CC_TABS) + 7 & ~7u];
  uint16_t HeadPos[KVB_T];         // sorted position of each segment's head
  uint32_t Sany, Swn;
  uint32_t Wst[KVB_T + 1], Wcur[KVB_T];  // a sub of several stretches: where each stretch's records start in the regrouped copy
  uint32_t Sred[KVB_W];
  uint32_t Hs[16];                 // dominant-key path: candidate counts, flags, the row's location
  int Hc[2][KVB_W];                // ... wave carries of its prefix tables
};

// ---- smallbank's counters over <= 64 requests of one key in request order (smallbank/udp/server_shard.cc:121-161), one wave:
// returns the granted lanes of `rem` (ACQUIREs and RELEASEs by kind as lane masks); la = num_ex, lb = num_sh, wave-uniform.
__device__ static inline uint64_t kv_sb_walk(uint64_t rem, uint64_t mAS, uint64_t mAX, uint64_t mRS, uint64_t mRX, uint32_t &la, uint32_t &lb) {
  const uint32_t lane = lane_id();
  // The counters move between two modes.  FREE (num_ex == 0): every ACQUIRE_SHARED is granted (num_sh++),
  // RELEASE_SHARED decrements, and nothing else happens until an EVENT: an ACQUIRE_EXCLUSIVE that finds
  // num_sh == 0 (granted: num_ex = 1) or a RELEASE_EXCLUSIVE (num_ex wraps to 2^32 - 1, as the reference's
  // unsigned counter does).  HELD (num_ex != 0): every ACQUIRE is rejected, RELEASE_SHARED still decrements,
  // until the num_ex-th RELEASE_EXCLUSIVE.  Between events all lanes are resolved at once (num_sh before a lane
  // = num_sh + ACQUIRE_SHAREDs - RELEASE_SHAREDs below it): a contended account changes mode rarely.
  uint64_t G = 0;
  while (rem) {
    if (la == 0) {
      const uint64_t blw = rem & lanemask_lt();
      const uint32_t lb_before = lb + (uint32_t)__popcll(blw & mAS) - (uint32_t)__popcll(blw & mRS);
      const bool me = (rem >> lane) & 1ull;
      const uint64_t ev = __ballot(me && ((((mAX >> lane) & 1ull) && lb_before == 0) || ((mRX >> lane) & 1ull)));
      const uint64_t upto = ev ? (ev & (0 - ev)) - 1ull : ~0ull;  // the lanes below the first event
      const uint64_t seg = rem & upto;
      G |= seg & mAS;
      lb += (uint32_t)__popcll(seg & mAS) - (uint32_t)__popcll(seg & mRS);
      rem &= ~upto;
      if (ev) {
        const uint64_t bit = ev & (0 - ev);
        if (bit & mAX) { G |= bit; la = 1; } else la = 0xFFFFFFFFu;
        rem &= ~bit;
      }
    } else {
      uint64_t rx = rem & mRX;
      const uint32_t nrx = (uint32_t)__popcll(rx);
      if (nrx < la) {  // held to the end of these lanes
        lb -= (uint32_t)__popcll(rem & mRS);
        la -= nrx;
        rem = 0;
      } else {
        for (uint32_t k = 1; k < la; k++) rx &= rx - 1;  // the la-th RELEASE_EXCLUSIVE (la is 1 unless the counter wrapped)
        const uint64_t bit = rx & (0 - rx), upto = bit - 1ull;
        lb -= (uint32_t)__popcll(rem & upto & mRS);
        la = 0;
        rem &= ~(upto | bit);
      }
    }
  }
  return G;
}

// ---- big subs (more than 64 records on one sub of a coarse bin: hot keys), the whole 512-thread workgroup -----------
// The same algorithm as kv_chunk, over up to KVB_NMAX requests at a time:
//   sort the bin by (bucket group, key hash, request index) in LDS; the requests of one key are then one segment of
//   the sorted order, however many there are.  Ballot masks over the WHOLE sorted stretch (heads, writers, lock ops,
//   ...) plus one small table per mask (bits set / last set bit / next set bit per 64-bit word, built by one wave
//   with a lane per word) answer "writers below me in my segment", "last writer below", "last lock op below" in
//   O(1) for every request.  The memory work then runs tile by tile (512 requests per tile, one per thread) with no
//   table dependency between tiles: a segment that crosses a tile boundary (the hot key) is located once, its
//   {row location, version, lock state} travels in LDS, and its row is written back once, by its last request.  A
//   hot key's 3000 requests therefore cost one sort and six tiles of independent loads and stores.  smallbank's
//   counters are walked wave by wave with the running state carried through LDS.  Bucket runs with several keys on
//   one lock word, a bad key or several inserts / deletes run request by request after the tiles.
// A bin of more than KVB_NMAX records is cut into stretches along request-index buckets (every request of a
// stretch precedes every request of the next one) and the stretches run one after the other.
// (always inlined: as a function of its own it takes the LDS buffer as a GENERIC pointer -- every LDS access becomes a
// flat instruction and the stretch machinery, which lives in LDS, runs at half speed: 117 -> 170 us for the r04a hot pass)
template <int WL>
__device__ __forceinline__ static void
kv_big_bin(uint8_t *rep, uint32_t n, const kv_cut cut, const kv_dev *kv, uint32_t bin, const uint64_t *__restrict__ recs,
           uint64_t *__restrict__ recs2, uint32_t c, dint_dev_stats *__restrict__ stats, int force_flags, const dint_view V, uint8_t *lds_raw,
           uint8_t *lds_bm, uint64_t *wtr) {
  using F = Fmt<WL>;
  const int force_rounds = force_flags & 1, no_hot = force_flags & 2;
  const uint32_t hot_min = (uint32_t)force_flags >> 8 ? (uint32_t)force_flags >> 8 : KVB_HOT_MIN;
  kvb_lds &LB = *(kvb_lds *)lds_raw;
  auto &Sk = LB.Sk; auto &Bcnt = LB.Bcnt; auto &Bwin = LB.Bwin;
  auto &Mhead = LB.Mhead; auto &Mbh = LB.Mbh; auto &Mbad = LB.Mbad; auto &Mlop = LB.Mlop; auto &Mst = LB.Mst; auto &Mlkseg = LB.Mlkseg;
  auto &Mstseg = LB.Mstseg; auto &Mrs = LB.Mrs; auto &Msimple = LB.Msimple; auto &Mwr = LB.Mwr; auto &Mlk = LB.Mlk; auto &Macq = LB.Macq;
  auto &Pbad = LB.Pbad; auto &Plop = LB.Plop; auto &Pst = LB.Pst; auto &Plkseg = LB.Plkseg; auto &Pstseg = LB.Pstseg; auto &Pwr = LB.Pwr;
  auto &Ehead = LB.Ehead; auto &Ebh = LB.Ebh; auto &Ewr = LB.Ewr; auto &Elk = LB.Elk;
  auto &Mbail = LB.Mbail; auto &Lead = LB.Lead; auto &CarryCrow = LB.CarryCrow; auto &HeadPos = LB.HeadPos; auto &Phead = LB.Phead;
  auto &Sany = LB.Sany; auto &Swn = LB.Swn; auto &Sred = LB.Sred; auto &Hs = LB.Hs; auto &Hc = LB.Hc;
  auto &Wst = LB.Wst; auto &Wcur = LB.Wcur;
  // smallbank walks its counters through Carry[segment], store / tatp the row machine of segments with an INSERT /
  // DELETE through Crow[segment]: never both in one instantiation, so they share one buffer
  kvb_carry *Carry = (kvb_carry *)CarryCrow;
  kv_rowst *Crow = (kv_rowst *)CarryCrow;
  // DINT_KV_TRACE: phase stamps of one stretch (the second of a sub cut into several -- the first one pays the cold misses)
  // into the workgroup's trace words: [4] stretch in, [5] gathered, [6] in order, [7] heads / op classes, [8] keys
  // checked, [9] masks, [10] rows located and lock grants walked, [11] tiles, [12] written back, [13] out (after the rounds);
  // the dominant-key path (first time round): [16] in, [17] sampled, [18] checked, [19] ordering ops sorted, [20] row located,
  // [21] answered, [22] remainder compacted; [23] a check failed, [24] ordering ops, [25] requests of the key, [26] of the stretch
#define KVB_STAMP(k) do { if (wtr && t == 0 && win == stamp_win) wtr[k] = __builtin_amdgcn_s_memrealtime(); } while (0)
  const uint32_t t = threadIdx.x, lane = t & 63, wave = t >> 6;
  // idx buckets for the stretches of a bin with more than KVB_NMAX records: 2^bs requests per bucket, <= KVB_NBK buckets
  const uint32_t nbits = n > 1 ? 32u - (uint32_t)__clz(n - 1) : 0u, bs = nbits > 11 ? nbits - 11 : 0u;
  const uint32_t wcap = KVB_NMAX - (1u << bs);  // a stretch = the buckets whose exclusive record count / wcap is equal
  const uint32_t sh_g = 16 + cut.ibits, sh_k = 7 + cut.ibits, idx_mask = (uint32_t)((1ull << cut.ibits) - 1ull);
  auto k_idx = [&](uint64_t w) -> uint32_t { return (uint32_t)(w >> 7) & idx_mask; };
  auto k_type = [&](uint64_t w) -> uint32_t { return pay_type((uint32_t)w & 0x7Fu); };
  auto k_q = [&](uint64_t w) -> uint32_t { return pay_q((uint32_t)w & 0x7Fu); };
  auto is_writer = [&](uint32_t type) -> bool {
    return WL == DINT_WL_STORE ? type == 1 : WL == DINT_WL_TATP ? (type == 12 || type == 13) : (type == 4 || type == 5);
  };
  // tatp's lock byte is a last-writer-wins register: ACQUIRE leaves 1 (granted or not), every other lock op leaves 0,
  // and only ACQUIRE's reply depends on it.  When several keys of a bucket run use one lock byte, an ACQUIRE looks
  // through the run's lock ops for the latest one on its byte that precedes it in request order.
  // Returns 1 / 0 = what that op left, -1 = there is none.
  auto lock_scan = [&](uint32_t a, uint32_t b, uint32_t qq, uint32_t before_idx) -> int {
    int kind = -1;
    uint32_t at = 0;
    for (uint32_t w = a >> 6; a < b && w <= ((b - 1) >> 6); w++) {
      uint64_t mm = Mlop[w];
      if (w == (a >> 6)) mm &= ~0ull << (a & 63);
      if (w == ((b - 1) >> 6) && (b & 63)) mm &= (1ull << (b & 63)) - 1ull;
      for (; mm; mm &= mm - 1) {
        const uint64_t o = Sk[w * 64 + (uint32_t)__ffsll((unsigned long long)mm) - 1];
        const uint32_t oi = k_idx(o);
        if (k_q(o) == qq && oi < before_idx && (kind < 0 || oi > at)) { at = oi; kind = k_type(o) == 1; }
      }
    }
    return kind;
  };
  __syncthreads();  // the LDS buffer is free (the chunk path, or the previous big sub, is done with it)
  auto rec_at = [&](uint32_t k) -> uint64_t { return recs[k]; };

  uint32_t nwin = 1;
  bool regrouped = false;
  if (c > KVB_NMAX) {
    for (uint32_t w = t; w < KVB_NBK / 2; w += KVB_T) Bcnt[w] = 0;
    __syncthreads();
#pragma unroll 4
    for (uint32_t k = t; k < c; k += KVB_T) {
      const uint32_t b = kv_rec_idx(rec_at(k), cut) >> bs;
      atomicAdd(&Bcnt[b >> 1], 1u << (16 * (b & 1)));
    }
    __syncthreads();
    uint32_t cw[2], run = 0;  // thread t owns buckets 4t .. 4t+3
#pragma unroll
    for (uint32_t j = 0; j < 2; j++) {
      cw[j] = Bcnt[2 * t + j];
      run += (cw[j] & 0xFFFF) + (cw[j] >> 16);
    }
    uint32_t tot, base = wave_excl_scan_u32(run, &tot);
    if (lane == 0) Sred[wave] = tot;
    __syncthreads();
    for (uint32_t w = 0; w < wave; w++) base += Sred[w];
    nwin = (c - 1) / wcap + 1;  // upper bound: the last one may be empty
    // a sub of three stretches or more is regrouped by stretch ONCE (recs2): each stretch then reads its own ~4,000
    // records instead of looking through all of them (r03: 13 us per stretch of smallbank's 35,000-request account)
    regrouped = recs2 != nullptr && nwin >= 3 && nwin <= KVB_T;
    if (regrouped) {
      for (uint32_t w = t; w <= KVB_T; w += KVB_T) Wst[w] = c;
      Wst[t] = c; Wcur[t] = 0;
    }
    __syncthreads();
    uint32_t prevcnt = t ? Bcnt[2 * t - 1] >> 16 : 0u;
#pragma unroll
    for (uint32_t j = 0; j < 4; j++) {
      const uint32_t wb = base / wcap, cntb = (cw[j >> 1] >> (16 * (j & 1))) & 0xFFFF;
      Bwin[4 * t + j] = (uint16_t)wb;
      if (regrouped && (4 * t + j == 0 || (base - prevcnt) / wcap != wb)) Wst[wb] = base;  // the stretch's first bucket
      prevcnt = cntb;
      base += cntb;
    }
    __syncthreads();
    if (regrouped) {
      for (uint32_t k0 = 0; k0 < c; k0 += 4 * KVB_T) {
        uint64_t r4[4];
#pragma unroll
        for (uint32_t j = 0; j < 4; j++) {
          const uint32_t k = k0 + j * KVB_T + t;
          r4[j] = k < c ? rec_at(k) : 0;
        }
#pragma unroll
        for (uint32_t j = 0; j < 4; j++) {
          const uint32_t k = k0 + j * KVB_T + t;
          const uint32_t w = k < c ? Bwin[kv_rec_idx(r4[j], cut) >> bs] : 0xFFFFFFFFu;
          for (uint64_t todo = __ballot(k < c); todo;) {  // one LDS atomic per wave and stretch (records arrive nearly in request order)
            const int l = __ffsll((unsigned long long)todo) - 1;
            const uint32_t w0 = (uint32_t)__builtin_amdgcn_readlane((int)w, l);
            const uint64_t mm = __ballot(w == w0);
            uint32_t at = 0;
            if ((int)lane == l) at = atomicAdd(&Wcur[w0], (uint32_t)__popcll(mm));
            at = (uint32_t)__builtin_amdgcn_readlane((int)at, l);
            if (w == w0) recs2[Wst[w0] + at + (uint32_t)__popcll(mm & lanemask_lt())] = r4[j];
            todo &= ~mm;
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    }
  }
  __syncthreads();

  const uint32_t stamp_win = nwin > 1 ? 1u : 0u;
  bool pf = false;      // the next stretch's records are already in LDS (smallbank: the ordered-copy buffer, free by then)
  uint32_t pf_m = 0;
  for (uint32_t win = 0; win < nwin; win++) {
    if (t == 0) Swn = 0;
    __syncthreads();
    KVB_STAMP(4);
    // ---- gather the stretch (any order) as sort keys
    auto sort_key = [&](uint64_t r) -> uint64_t {
      return kv_sort_key(r, cut);
    };
    if (c <= KVB_NMAX) {  // the whole bin
      for (uint32_t k = t; k < c; k += KVB_T) Sk[k] = sort_key(rec_at(k));
      if (t == 0) Swn = c;
    } else if (regrouped && pf) {
      if (wtr && t == 0 && win == stamp_win) wtr[27] = 1;
#pragma unroll
      for (uint32_t j = 0; j < KVB_NMAX / KVB_T; j++)
        if (j * KVB_T + t < pf_m) Sk[j * KVB_T + t] = ((const uint64_t *)(lds_bm + KVB_BM_W * 10))[j * KVB_T + t];
      if (t == 0) Swn = pf_m;
      pf = false;
    } else if (regrouped) {
      const uint32_t a = Wst[win], b = max(a, Wst[win + 1]);  // (an empty stretch between two others keeps the end mark c)
      uint64_t r8[KVB_NMAX / KVB_T];  // (all loads first: a loop of load / LDS store pairs waits for every load on its own)
#pragma unroll
      for (uint32_t j = 0; j < KVB_NMAX / KVB_T; j++) r8[j] = a + j * KVB_T + t < b ? recs2[a + j * KVB_T + t] : 0;
#pragma unroll
      for (uint32_t j = 0; j < KVB_NMAX / KVB_T; j++)
        if (a + j * KVB_T + t < b) Sk[j * KVB_T + t] = sort_key(r8[j]);
      if (t == 0) Swn = b - a;
    } else {
      for (uint32_t k0 = 0; k0 < c; k0 += 4 * KVB_T) {  // four records per thread in flight; one slot reservation per wave and step
        uint64_t r4[4];
#pragma unroll
        for (uint32_t j = 0; j < 4; j++) {
          const uint32_t k = k0 + j * KVB_T + t;
          r4[j] = k < c ? rec_at(k) : 0;
        }
#pragma unroll
        for (uint32_t j = 0; j < 4; j++) {
          const uint32_t k = k0 + j * KVB_T + t;
          const uint64_t r = r4[j];
          const bool in = k < c && Bwin[kv_rec_idx(r, cut) >> bs] == win;
          const uint64_t im = __ballot(in);
          uint32_t base = 0;
          if (lane == 0 && im) base = atomicAdd(&Swn, (uint32_t)__popcll(im));
          base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
          if (in) Sk[base + (uint32_t)__popcll(im & lanemask_lt())] = sort_key(r);
        }
      }
    }
    KVB_STAMP(31);
    if (wtr && t == 0 && win == stamp_win) { wtr[28] = regrouped; wtr[29] = nwin; }
    __syncthreads();
    uint32_t m = Swn;
    __syncthreads();  // Swn is reset at the top of the next stretch
    KVB_STAMP(5);
    if (m == 0) continue;  // workgroup-uniform
    bool tiny = m <= 64;  // what one wave resolves in registers (kv_chunk): below, also for what a dominant key leaves behind
    if (!tiny) {

    // ---- the stretch's DOMINANT KEY (a hot row: most of a big bin is one key) is answered without sorting the stretch.
    // Only the requests that change what a later request sees -- writers and lock ops, a few hundred of the thousands
    // -- are put in request order (the list M: one LDS sort of <= 1024 words); every request of the key then finds,
    // by binary search on its index, how many of them precede it: version = v0 + writers before me, value = message
    // of the last writer before me, lock = what the last lock op before me left.  The closed forms are those of
    // kv_chunk; anything they do not cover (another key of the same bucket on the same lock byte, inserts / deletes,
    // a key-hash collision, > 1024 ordering ops) leaves the whole stretch to the general path below.
    // (repeated while what is left still has a dominant key: a sub with TWO hot keys -- tatp's hottest subscriber and a
    // neighbour -- used to send the second one, ~1,000 requests, through the whole stretch machinery: 43 us behind the 50)
    bool again = WL != DINT_WL_SMALLBANK && !force_rounds && !no_hot;
    for (uint32_t hot_pass = 0; again && !tiny && m >= hot_min && hot_pass < 4; hot_pass++) {
      again = false;
      if (hot_pass == 0) KVB_STAMP(16);
      uint32_t *Mk = (uint32_t *)Lead;                 // [1024] idx << 12 | position in Sk, ascending
      uint16_t *Mwc = (uint16_t *)Carry;                // [j] writers among the first j ops of M
      int16_t *Mlw = (int16_t *)(Mwc + KV_MMAX + 8);   // [j] last writer among the first j (index into Mk), -1: none
      int16_t *Mll = Mlw + KV_MMAX + 8;                // [j] last lock op among the first j
      // 1. the most frequent (bucket group, key hash) among eight samples
      uint64_t cand[8];
      uint32_t cc[8];
#pragma unroll
      for (uint32_t k = 0; k < 8; k++) { cand[k] = Sk[(uint32_t)(((uint64_t)m * k) >> 3)] >> sh_k; cc[k] = 0; }
      if (t < 16) Hs[t] = 0;
      __syncthreads();
      for (uint32_t p = t; p < m; p += KVB_T) {
        const uint64_t pf = Sk[p] >> sh_k;
#pragma unroll
        for (uint32_t k = 0; k < 8; k++) cc[k] += pf == cand[k];
      }
#pragma unroll
      for (uint32_t k = 0; k < 8; k++) {
        uint32_t v = cc[k];
        for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
        if (lane == 0 && v) atomicAdd(&Hs[k], v);
      }
      __syncthreads();
      uint32_t best = 0;
#pragma unroll
      for (uint32_t k = 1; k < 8; k++) best = Hs[k] > Hs[best] ? k : best;
      const uint32_t hot_n = Hs[best];
      const uint64_t hpf = cand[best];
      const uint32_t hsp = (uint32_t)(((uint64_t)m * best) >> 3);  // a position that holds the hot key
      __syncthreads();
      if (hot_pass == 0) KVB_STAMP(17);
      if (hot_n >= hot_min && 2 * hot_n >= m) {  // workgroup-uniform
        // 2. everything the closed form needs to hold, checked before anything is written
        const uint64_t hcur = Sk[hsp];
        const uint32_t hq = k_q(hcur);
        const uint64_t hkey = ld_u64(rep + dint_view_off(V, k_idx(hcur), F::MSG) + F::KEY);
        if (t < 16) Hs[t] = t == 3 ? 0xFFFFFFFFu : 0u;  // [0] bad, [1] ordering ops, [3] / [4] their lowest / highest request index
        __syncthreads();
        uint32_t bad = 0, nord = 0, olo = 0xFFFFFFFFu, ohi = 0;
#pragma unroll 8
        for (uint32_t p = t; p < m; p += KVB_T) {  // (unrolled: the key gathers of a thread's <= 8 records in flight together)
          const uint64_t cur = Sk[p];
          const uint32_t type = k_type(cur);
          if ((cur >> sh_k) == hpf) {
            bad |= !kv_simple_op<WL>(type);
            bad |= ld_u64(rep + dint_view_off(V, k_idx(cur), F::MSG) + F::KEY) != hkey;  // 9 hash bits can collide
            if (is_writer(type) || kv_lock_op<WL>(type)) { nord++; olo = min(olo, k_idx(cur)); ohi = max(ohi, k_idx(cur)); }
          } else if ((cur >> sh_g) == (hpf >> 9)) {  // another key of the hot bucket: must not touch my lock byte or the chain
            bad |= kv_struct_op<WL>(type) || (kv_lock_op<WL>(type) && k_q(cur) == hq);
          }
        }
        for (int d = 32; d > 0; d >>= 1) {
          nord += __shfl_xor(nord, d, 64);
          olo = min(olo, (uint32_t)__shfl_xor(olo, d, 64)); ohi = max(ohi, (uint32_t)__shfl_xor(ohi, d, 64));
        }
        if (lane == 0 && nord) { atomicAdd(&Hs[1], nord); atomicMin(&Hs[3], olo); atomicMax(&Hs[4], ohi); }
        if (bad) Hs[0] = 1;
        __syncthreads();
        const uint32_t nM = Hs[1];
        olo = Hs[3];
        const uint32_t ospan = nM ? Hs[4] - olo + 1 : 0;
        const bool hot_ok = !Hs[0] && nM <= KV_MMAX;
        __syncthreads();
        if (hot_pass == 0) KVB_STAMP(18);
        if (wtr && t == 0 && win == stamp_win && hot_pass == 0) { wtr[23] = Hs[0]; wtr[24] = nM; wtr[25] = hot_n; wtr[26] = m; }
        if (hot_ok) {
          // 3. M = the key's writers and lock ops, sorted by request index.  Request indices are distinct, so when their span
          // fits the index bitmap (a pass of up to 512k requests) an op's place in M is the number of set bits below its
          // own -- and every request of the key later finds "ops before me" the same way, not by binary search.
          // (r03: compaction + LDS sort of M, 15 us for the 1,075 ops of tatp's hottest subscriber; 11-step searches.)
          const bool by_bitmap = KV_HOT_BM && lds_bm != nullptr && ospan <= KV_HOT_BM_W * 64;  // workgroup-uniform
          uint64_t *Bm = (uint64_t *)lds_bm;                  // [KV_HOT_BM_W]
          uint16_t *Wp = (uint16_t *)(Bm + KV_HOT_BM_W);       // [KV_HOT_BM_W] bits set below each word
          auto ops_below = [&](uint32_t idx) -> uint32_t {     // ordering ops with a smaller request index
            if (idx <= olo) return 0u;
            const uint32_t b = idx - olo;
            if (b >= ospan) return nM;
            return Wp[b >> 6] + (uint32_t)__popcll(Bm[b >> 6] & ((1ull << (b & 63)) - 1ull));
          };
          if (by_bitmap) {
            const uint32_t nw = (ospan + 63) >> 6;
            for (uint32_t w = t; w < nw; w += KVB_T) Bm[w] = 0;
            __syncthreads();
            for (uint32_t p = t; p < m; p += KVB_T) {
              const uint64_t cur = Sk[p];
              const uint32_t type = k_type(cur);
              if ((cur >> sh_k) == hpf && (is_writer(type) || kv_lock_op<WL>(type))) {
                const uint32_t b = k_idx(cur) - olo;
                atomicOr((unsigned long long *)&Bm[b >> 6], 1ull << (b & 63));
              }
            }
            __syncthreads();
            {  // thread t owns words PW t .. PW t + PW - 1
              constexpr uint32_t PW = KV_HOT_BM_W / KVB_T;
              uint32_t run = 0;
#pragma unroll 4
              for (uint32_t j = 0; j < PW; j++) run += PW * t + j < nw ? (uint32_t)__popcll(Bm[PW * t + j]) : 0u;
              uint32_t tot, base = wave_excl_scan_u32(run, &tot);
              if (lane == 0) Sred[wave] = tot;
              __syncthreads();
              for (uint32_t w = 0; w < wave; w++) base += Sred[w];
#pragma unroll 4
              for (uint32_t j = 0; j < PW; j++) {  // (the words are read again: sixteen counts per thread do not fit the register file here)
                const uint32_t w = PW * t + j;
                if (w < nw) { Wp[w] = (uint16_t)base; base += (uint32_t)__popcll(Bm[w]); }
              }
            }
            __syncthreads();
            for (uint32_t p = t; p < m; p += KVB_T) {
              const uint64_t cur = Sk[p];
              const uint32_t type = k_type(cur);
              if ((cur >> sh_k) == hpf && (is_writer(type) || kv_lock_op<WL>(type))) Mk[ops_below(k_idx(cur))] = (k_idx(cur) << 12) | p;
            }
            __syncthreads();
          } else {
          if (t == 0) Hs[2] = 0;
          __syncthreads();
          for (uint32_t p0 = 0; p0 < m; p0 += KVB_T) {
            const uint32_t p = p0 + t;
            const uint64_t cur = p < m ? Sk[p] : 0;
            const uint32_t type = k_type(cur);
            const bool in = p < m && (cur >> sh_k) == hpf && (is_writer(type) || kv_lock_op<WL>(type));
            const uint64_t im = __ballot(in);
            uint32_t base = 0;
            if (lane == 0 && im) base = atomicAdd(&Hs[2], (uint32_t)__popcll(im));
            base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
            if (in) Mk[base + (uint32_t)__popcll(im & lanemask_lt())] = (k_idx(cur) << 12) | p;
          }
          // (sorting M with kvb_sort_stretch -- registers / shuffles instead of an LDS step per barrier -- was measured:
          // the list sort itself is faster, the bench 1.3 % slower on the same box; the kernel is code-size sensitive)
          // (r02-r03 sorted M with one LDS compare-exchange step per barrier -- 55 barriers for 1,024 words, 13 of the hot
          // key's 54 us -- because the register / shuffle sort made the shared resolve kernel 1.3 % slower by its code size;
          // the big path is a kernel of its own now)
          __syncthreads();
          for (uint32_t k = nM + t; k < KV_MMAX; k += KVB_T) Mk[k] = 0xFFFFFFFFu;  // empty slots sort last
          __syncthreads();
          {
            uint32_t N2 = 64;
            while (N2 < nM) N2 <<= 1;
            if (nM > 1) kvb_sort_blocked_u32<4>(Mk, max(N2, 256u));  // (the fallback of passes beyond 512k requests: one instantiation)
          }
          }
          if (hot_pass == 0) KVB_STAMP(19);
          // prefix tables over M (nM + 1 rows: what precedes op j; row nM = the totals): thread t owns ops PER t .. PER t + PER - 1
          {
            constexpr uint32_t PER = KV_MMAX / KVB_T;
            bool wv[PER], lv[PER];
            uint32_t nwr = 0;
            int iw = -1, il = -1;  // last writer / lock op among my ops
#pragma unroll
            for (uint32_t r = 0; r < PER; r++) {
              const uint32_t j = PER * t + r;
              const uint32_t ty = j < nM ? k_type(Sk[Mk[j] & 4095u]) : 0xFFu;
              wv[r] = j < nM && is_writer(ty);
              lv[r] = j < nM && kv_lock_op<WL>(ty);
              nwr += wv[r];
              if (wv[r]) iw = (int)j;
              if (lv[r]) il = (int)j;
            }
            uint32_t wt, wx = wave_excl_scan_u32(nwr, &wt);
            for (int d = 1; d < 64; d <<= 1) {  // inclusive running maxima over the wave
              const int a2 = __shfl_up(iw, d, 64), b2 = __shfl_up(il, d, 64);
              if ((int)lane >= d) { iw = max(iw, a2); il = max(il, b2); }
            }
            if (lane == 63) { Sred[wave] = wt; Hc[0][wave] = iw; Hc[1][wave] = il; }
            int ew = __shfl_up(iw, 1, 64), el = __shfl_up(il, 1, 64);  // exclusive: what precedes my ops inside the wave
            if (lane == 0) { ew = -1; el = -1; }
            __syncthreads();
            for (uint32_t w = 0; w < wave; w++) { wx += Sred[w]; ew = max(ew, Hc[0][w]); el = max(el, Hc[1][w]); }
#pragma unroll
            for (uint32_t r = 0; r < PER; r++) {
              const uint32_t j = PER * t + r;
              if (j <= nM) { Mwc[j] = (uint16_t)wx; Mlw[j] = (int16_t)ew; Mll[j] = (int16_t)el; }
              wx += wv[r];
              if (wv[r]) ew = (int)j;
              if (lv[r]) el = (int)j;
            }
            if (PER * t + PER == nM) { Mwc[nM] = (uint16_t)wx; Mlw[nM] = (int16_t)ew; Mll[nM] = (int16_t)el; }  // my ops end M: the totals
          }
          __syncthreads();
          // 4. the row: one thread probes the bucket
          if (t == 0) {
            const uint32_t gk = kv_cut_gk((uint32_t)(hcur >> sh_g), bin, cut), table = kv_table_of(kv, gk);
            const uint64_t bucket = (uint64_t)(gk - kv->gk_base[table]);
            const kv_tab tb = kv->tab[table];
            const uint8_t *ie = kv_entry_ptr(tb, bucket, KV_INLINE);
            kv_hdr H;
            kv_hdr_load(H, ie);
            const kv_where wh = kv_locate(tb, bucket, H, hkey);
            Hs[8] = wh.found; Hs[9] = wh.link; Hs[10] = wh.slot; Hs[11] = wh.ver;
            Hs[12] = WL == DINT_WL_TATP ? (H.lockw >> (8 * hq)) & 0xFFu : 0;
          }
          __syncthreads();
          if (hot_pass == 0) KVB_STAMP(20);
          const uint32_t found = Hs[8], link = Hs[9], slot = Hs[10], ver0 = Hs[11], la0 = Hs[12];
          const uint32_t hgk = kv_cut_gk((uint32_t)(hcur >> sh_g), bin, cut), htable = kv_table_of(kv, hgk);
          const uint64_t hbucket = (uint64_t)(hgk - kv->gk_base[htable]);
          const kv_tab htb = kv->tab[htable];
          uint8_t *hrow = kv_entry_ptr(htb, hbucket, link) + KV_VAL_OFF + slot * F::VS;  // meaningful when found
          // 5. every request of the key: no table dependency between them.  A thread's <= 8 requests are answered in two
          // sweeps -- outcomes and value loads of all of them, then the stores -- so the loads ride on one round trip
          // (as a loop of load / store pairs every request waited for its own: 10 of the hot key's 54 us)
          constexpr uint32_t NP = 4, NWD = F::VS / 4;  // (four at a time: eight value buffers do not fit the register file)
          static_assert(KVB_NMAX / KVB_T == 2 * NP, "two sweeps of NP requests per thread");
          for (uint32_t j0 = 0; j0 < 2 * NP && j0 * KVB_T < m; j0 += NP) {
          uint32_t a_code[NP], a_ver[NP], a_idx[NP], a_w[NP][NWD];
          bool a_on[NP], a_get[NP];
#pragma unroll
          for (uint32_t j = 0; j < NP; j++) {
            const uint32_t p = t + (j0 + j) * KVB_T;
            const uint64_t cur = p < m ? Sk[p] : 0;
            a_on[j] = p < m && (cur >> sh_k) == hpf;
            a_get[j] = false; a_code[j] = 0; a_ver[j] = 0; a_idx[j] = 0;
            if (!a_on[j]) continue;
            const uint32_t type = k_type(cur), idx = k_idx(cur);
            uint32_t lo = 0, hi = nM;  // ops of M with a smaller request index
            if (by_bitmap) {
              lo = ops_below(idx);
            } else {
              const uint32_t key32 = idx << 12;
              while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                if (Mk[mid] < key32) lo = mid + 1; else hi = mid;
              }
            }
            const int lw = found ? (int)Mlw[lo] : -1, ll = (int)Mll[lo];
            uint32_t code, get = 0;
            if (WL == DINT_WL_STORE) {
              code = type == 0 ? (found ? 3 : 7) : (found ? 5 : 7);
              get = type == 0 && found;
            } else {
              const uint32_t lock_seen = ll >= 0 ? (uint32_t)(k_type(Sk[Mk[ll] & 4095u]) == 1) : la0;
              switch (type) {
                case 0: code = found ? 4 : 6; get = found; break;
                case 1: code = lock_seen ? 8 : 7; break;
                case 2: code = 9; break;
                case 12: code = 15; break;
                default: code = 16; break;  // 13 kCommitBck
              }
            }
            a_code[j] = code; a_ver[j] = ver0 + (found ? Mwc[lo] : 0u); a_idx[j] = idx; a_get[j] = get != 0;
            if (get) {
              const uint8_t *from = lw >= 0 ? rep + dint_view_off(V, k_idx(Sk[Mk[lw] & 4095u]), F::MSG) + F::VAL : hrow;
#pragma unroll
              for (uint32_t k = 0; k < NWD; k++) a_w[j][k] = ld_u32(from + 4 * k);
            }
          }
#pragma unroll
          for (uint32_t j = 0; j < NP; j++) {
            if (!a_on[j]) continue;
            uint8_t *msg = rep + dint_view_off(V, a_idx[j], F::MSG);
            if (a_get[j]) {
#pragma unroll
              for (uint32_t k = 0; k < NWD; k++) st_u32(msg + F::VAL + 4 * k, a_w[j][k]);
              st_u32(msg + F::VER, a_ver[j]);
            }
            msg[F::TYPE] = (uint8_t)a_code[j];
          }
          }
          __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
          __syncthreads();  // every read of the row precedes its write-back
          if (hot_pass == 0) KVB_STAMP(21);
          // 6. final state, written once
          if (t == 0) {
            const uint32_t nw = Mwc[nM];
            const int lw = (int)Mlw[nM], ll = (int)Mll[nM];
            if (found && lw >= 0) {
              kv_copy_words(hrow, rep + dint_view_off(V, k_idx(Sk[Mk[lw] & 4095u]), F::MSG) + F::VAL, F::VS);
              kv_entry_hdr(htb, hbucket, link)->ver[slot] = ver0 + nw;
            }
            if (WL == DINT_WL_TATP && ll >= 0) {
              const uint32_t fin = (uint32_t)(k_type(Sk[Mk[ll] & 4095u]) == 1);
              if (fin != la0) kv_entry_ptr(htb, hbucket, KV_INLINE)[KV_LOCKB_OFF + hq] = (uint8_t)fin;
            }
            if (WL == DINT_WL_TATP && !found && nw) atomicAdd(&stats->missing_keys, (unsigned long long)nw);
          }
          // 7. what is left of the stretch moves to the front (destinations never overtake unread sources)
          if (t == 0) Hs[2] = 0;
          __syncthreads();
          for (uint32_t p0 = 0; p0 < m; p0 += KVB_T) {
            const uint32_t p = p0 + t;
            const uint64_t cur = p < m ? Sk[p] : 0;
            const bool keep = p < m && (cur >> sh_k) != hpf;
            const uint64_t km = __ballot(keep);
            __syncthreads();  // the round's sources are read
            uint32_t base = 0;
            if (lane == 0 && km) base = atomicAdd(&Hs[2], (uint32_t)__popcll(km));
            base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
            if (keep) Sk[base + (uint32_t)__popcll(km & lanemask_lt())] = cur;
            __syncthreads();
          }
          m = Hs[2];
          __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
          __syncthreads();
          tiny = m <= 64;
          again = true;
          if (hot_pass == 0) KVB_STAMP(22);
        }
      }
    }
    }
    if (m == 0) continue;  // workgroup-uniform: the dominant keys were all of it
    if (tiny) {
      // <= 64 records left (a big sub is its hot key and a handful of neighbours; the all-big fallback of a crowded coarse bin
      // hands over subs of any size): one wave, as a chunk of the resolve kernel -- not ~25 barriers of stretch machinery
      if (wave == 0) {
        const bool valid = lane < m;
        uint64_t wv = valid ? Sk[lane] : ~0ull;  // group / P | key-hash bits | idx | type, quadrant: the chunk's sort word
        wv = wave_sort_u64(wv);
        const uint32_t idx = valid ? k_idx(wv) : 0, gk = kv_cut_gk((uint32_t)(wv >> sh_g), bin, cut);
        const uint64_t key = valid ? ld_u64(rep + dint_view_off(V, idx, F::MSG) + F::KEY) : 0;
        kv_chunk<WL>(rep, valid, idx, gk, (uint32_t)(wv >> sh_k) & 511u, k_type(wv), valid ? kv_table_of(kv, gk) : 0, k_q(wv), key, kv,
                     stats, force_rounds, true, V);
      }
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
      __syncthreads();
      continue;
    }
    // ---- a stretch that is (nearly) one or two keys -- smallbank's hot account, its savings and its checking row -- is put
    // in order WITHOUT the LDS sort (40 us for 4,096 words, five stretches a pass: the largest part of the hot pass in
    // r03).  Request indices are distinct, so the sorted position of a record among those of its key is the number of
    // set bits below its own in a bitmap over the stretch's index span: one LDS atomic to set the bit, one popcount
    // table, one popcount.  The few records of other keys are sorted on their own (<= 1,024) and the runs spliced by
    // comparing key prefixes.  Falls through to the sort when the span or the other keys do not fit.
    bool ordered = false;
    if (WL == DINT_WL_SMALLBANK && lds_bm != nullptr && !(force_flags & 4) && m >= KVB_BM_MIN) {  // (store / tatp: the dominant-key path above)
      uint64_t *Bm = (uint64_t *)lds_bm;              // [KVB_BM_W] bit (idx - lo) of the class in hand
      uint16_t *Wp = (uint16_t *)(Bm + KVB_BM_W);      // [KVB_BM_W] bits set below each word
      uint64_t *Sk2 = (uint64_t *)(Wp + KVB_BM_W);     // [KVB_NMAX] the stretch in order
      uint64_t *Ot = (uint64_t *)Lead;                 // [KVB_BM_OT] records of other keys
      static_assert(sizeof(LB.Lead) >= KVB_BM_OT * 8, "the other keys' list lives where the segment leaders will");
      uint64_t cand[8];
      uint32_t cc[8];
#pragma unroll
      for (uint32_t k = 0; k < 8; k++) { cand[k] = Sk[(uint32_t)(((uint64_t)m * k) >> 3)] >> sh_k; cc[k] = 0; }
      if (t < 16) Hs[t] = 0;
      __syncthreads();
      for (uint32_t p = t; p < m; p += KVB_T) {
        const uint64_t pfx = Sk[p] >> sh_k;
#pragma unroll
        for (uint32_t k = 0; k < 8; k++) cc[k] += pfx == cand[k];
      }
#pragma unroll
      for (uint32_t k = 0; k < 8; k++) {
        uint32_t v = cc[k];
        for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
        if (lane == 0 && v) atomicAdd(&Hs[k], v);
      }
      __syncthreads();
      uint32_t b0 = 0, b1 = 8;
#pragma unroll
      for (uint32_t k = 1; k < 8; k++) b0 = Hs[k] > Hs[b0] ? k : b0;
#pragma unroll
      for (uint32_t k = 0; k < 8; k++)
        if (cand[k] != cand[b0] && (b1 == 8 || Hs[k] > Hs[b1])) b1 = k;
      const uint64_t pf0 = cand[b0];
      const uint32_t n0 = Hs[b0];
      uint32_t n1 = b1 < 8 ? Hs[b1] : 0;
      if (8 * n1 < m) n1 = 0;  // a second class only when it is worth a bitmap of its own
      const uint64_t pf1 = n1 ? cand[b1] : ~0ull;  // (no prefix equals ~0: sh_k > 0)
      const uint32_t m_o = m - n0 - n1;
      __syncthreads();
      if (m_o <= KVB_BM_OT) {  // workgroup-uniform
        if (t < 16) Hs[t] = (t == 0 || t == 2) ? 0xFFFFFFFFu : 0u;  // lo0 hi0 lo1 hi1 | others placed, below pf0, below pf1
        for (uint32_t k = t; k < KVB_BM_OT; k += KVB_T) Ot[k] = ~0ull;
        __syncthreads();
        uint32_t lo0 = 0xFFFFFFFFu, hi0 = 0, lo1 = 0xFFFFFFFFu, hi1 = 0, l0 = 0, l1 = 0;
        for (uint32_t p0 = 0; p0 < m; p0 += KVB_T) {
          const uint32_t p = p0 + t;
          const uint64_t cur = p < m ? Sk[p] : 0;
          const uint64_t pfx = cur >> sh_k;
          const uint32_t idx = k_idx(cur);
          const bool c0 = p < m && pfx == pf0, c1 = p < m && pfx == pf1, oth = p < m && !c0 && !c1;
          if (c0) { lo0 = min(lo0, idx); hi0 = max(hi0, idx); }
          if (c1) { lo1 = min(lo1, idx); hi1 = max(hi1, idx); }
          const uint64_t om = __ballot(oth);
          uint32_t base = 0;
          if (lane == 0 && om) base = atomicAdd(&Hs[4], (uint32_t)__popcll(om));
          base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
          if (oth) {
            Ot[base + (uint32_t)__popcll(om & lanemask_lt())] = cur;
            l0 += pfx < pf0;
            l1 += pfx < pf1;
          }
        }
        for (int d = 32; d > 0; d >>= 1) {
          lo0 = min(lo0, (uint32_t)__shfl_xor(lo0, d, 64)); hi0 = max(hi0, (uint32_t)__shfl_xor(hi0, d, 64));
          lo1 = min(lo1, (uint32_t)__shfl_xor(lo1, d, 64)); hi1 = max(hi1, (uint32_t)__shfl_xor(hi1, d, 64));
          l0 += __shfl_xor(l0, d, 64); l1 += __shfl_xor(l1, d, 64);
        }
        if (lane == 0) {
          atomicMin(&Hs[0], lo0); atomicMax(&Hs[1], hi0); atomicMin(&Hs[2], lo1); atomicMax(&Hs[3], hi1);
          if (l0) atomicAdd(&Hs[5], l0);
          if (l1) atomicAdd(&Hs[6], l1);
        }
        __syncthreads();
        const uint32_t glo0 = Hs[0], glo1 = Hs[2];
        const uint32_t span0 = n0 ? Hs[1] - glo0 + 1 : 0, span1 = n1 ? Hs[3] - glo1 + 1 : 0;
        const uint32_t less0 = Hs[5], less1 = Hs[6];
        __syncthreads();
        if (span0 <= KVB_BM_W * 64 && span1 <= KVB_BM_W * 64) {  // workgroup-uniform
          if (m_o > KVB_T) kvb_sort_blocked<2>(Ot, KVB_BM_OT);
          else if (m_o > 64) kvb_sort_blocked<1>(Ot, KVB_T);
          else if (m_o > 1) {  // a handful of neighbours of the hot account: one wave, in registers
            if (wave == 0) Ot[lane] = wave_sort_u64(Ot[lane]);
            __syncthreads();
          }
#pragma unroll 1
          for (uint32_t cls = 0; cls < 2; cls++) {
            const uint32_t nk = cls ? n1 : n0;
            if (nk == 0) continue;
            const uint64_t pfc = cls ? pf1 : pf0;
            const uint32_t glo = cls ? glo1 : glo0, nw = ((cls ? span1 : span0) + 63) >> 6;
            // where the class starts: the other class and the other keys that sort below it
            const uint32_t cbase = (cls ? (pf0 < pf1 ? n0 : 0u) : (pf1 < pf0 ? n1 : 0u)) + (cls ? less1 : less0);
            for (uint32_t w = t; w < nw; w += KVB_T) Bm[w] = 0;
            __syncthreads();
            for (uint32_t p = t; p < m; p += KVB_T) {
              const uint64_t cur = Sk[p];
              if ((cur >> sh_k) == pfc) {
                const uint32_t b = k_idx(cur) - glo;
                atomicOr((unsigned long long *)&Bm[b >> 6], 1ull << (b & 63));
              }
            }
            __syncthreads();
            {  // bits below each word: thread t owns words 8t .. 8t + 7
              uint32_t pc[8], run = 0;
#pragma unroll
              for (uint32_t j = 0; j < 8; j++) {
                const uint32_t w = 8 * t + j;
                pc[j] = w < nw ? (uint32_t)__popcll(Bm[w]) : 0u;
                run += pc[j];
              }
              uint32_t tot, base = wave_excl_scan_u32(run, &tot);
              if (lane == 0) Sred[wave] = tot;
              __syncthreads();
              for (uint32_t w = 0; w < wave; w++) base += Sred[w];
#pragma unroll
              for (uint32_t j = 0; j < 8; j++) {
                const uint32_t w = 8 * t + j;
                if (w < nw) Wp[w] = (uint16_t)base;
                base += pc[j];
              }
            }
            __syncthreads();
            for (uint32_t p = t; p < m; p += KVB_T) {
              const uint64_t cur = Sk[p];
              if ((cur >> sh_k) == pfc) {
                const uint32_t b = k_idx(cur) - glo;
                Sk2[cbase + Wp[b >> 6] + (uint32_t)__popcll(Bm[b >> 6] & ((1ull << (b & 63)) - 1ull))] = cur;
              }
            }
            __syncthreads();
          }
          for (uint32_t j = t; j < m_o; j += KVB_T) {
            const uint64_t o = Ot[j];
            const uint64_t pfx = o >> sh_k;
            Sk2[j + (pf0 < pfx ? n0 : 0u) + (pf1 < pfx ? n1 : 0u)] = o;
          }
          __syncthreads();
          uint32_t N = 64;
          while (N < m) N <<= 1;
          for (uint32_t k = t; k < max(N, KVB_T); k += KVB_T) Sk[k] = k < m ? Sk2[k] : ~0ull;  // (as the sort leaves it)
          __syncthreads();
          ordered = true;
        }
      }
    }
    if (!ordered) kvb_sort_stretch(Sk, m);
    // The replies of a stretch are thousands of scattered byte and word stores; a wave's next LOAD returns only after its
    // older stores have (one counter for both on gfx9): the first global load of the following stretch used to wait ~13 us
    // for them.  The next stretch's records are therefore fetched HERE, before this stretch's stores are issued, into the
    // ordered-copy buffer (free from now on); the stores then drain under the next stretch's ordering phase, which works
    // in LDS only.
    if (WL == DINT_WL_SMALLBANK && lds_bm != nullptr && regrouped && win + 1 < nwin) {
      uint64_t *Nx = (uint64_t *)(lds_bm + KVB_BM_W * 10);
      const uint32_t a2 = Wst[win + 1], b2 = max(a2, Wst[min(win + 2, KVB_T)]);
      uint64_t r8[KVB_NMAX / KVB_T];
#pragma unroll
      for (uint32_t j = 0; j < KVB_NMAX / KVB_T; j++) r8[j] = a2 + j * KVB_T + t < b2 ? recs2[a2 + j * KVB_T + t] : 0;
#pragma unroll
      for (uint32_t j = 0; j < KVB_NMAX / KVB_T; j++)
        if (a2 + j * KVB_T + t < b2) Nx[j * KVB_T + t] = kv_sort_key(r8[j], cut);
      pf = true;
      pf_m = b2 - a2;
    }
    const uint32_t ntile = (m + KVB_T - 1) / KVB_T;
    KVB_STAMP(6);

    // ---- pass A: segment heads (key hash changes), bucket-run heads (group changes), op classes
    for (uint32_t j = 0; j < ntile; j++) {
      const uint32_t p = j * KVB_T + t;
      const uint64_t cur = Sk[p], prev = p ? Sk[p - 1] : ~0ull;
      const bool valid = p < m;
      const uint32_t type = k_type(cur);
      const uint64_t m1 = __ballot(valid && (p == 0 || (cur >> sh_k) != (prev >> sh_k)));
      const uint64_t m2 = __ballot(valid && (p == 0 || (cur >> sh_g) != (prev >> sh_g)));
      const uint64_t m3 = __ballot(valid && kv_lock_op<WL>(type));
      const uint64_t m4 = __ballot(valid && kv_struct_op<WL>(type));
      if (lane == 0) { Mhead[p >> 6] = m1; Mbh[p >> 6] = m2; Mlop[p >> 6] = m3; Mst[p >> 6] = m4; }
    }
    for (uint32_t w = ntile * KVB_W + t; w < KVB_NW; w += KVB_T) {
      Mhead[w] = 0; Mbh[w] = 0; Mlop[w] = 0; Mst[w] = 0; Mbad[w] = 0; Msimple[w] = 0; Mwr[w] = 0; Mlk[w] = 0; Macq[w] = 0;
    }
    for (uint32_t w = t; w < KVB_NW; w += KVB_T) {  // set bit by bit below
      Mstseg[w] = 0; Mrs[w] = 0; Mlkseg[0][w] = 0; Mlkseg[1][w] = 0; Mlkseg[2][w] = 0; Mlkseg[3][w] = 0;
    }
    __syncthreads();
    if (wave == 0) kvb_build_edge(Mhead, Ehead);
    if (wave == 1) kvb_build_edge(Mbh, Ebh);
    if (wave == 2) kvb_build_pop(Mlop, Plop);
    if (wave == 3) kvb_build_pop(Mst, Pst);
    if (wave == 4) kvb_build_pop(Mhead, Phead);
    __syncthreads();
    const uint32_t nseg = Phead.below[KVB_NW];
    KVB_STAMP(7);
    // ---- pass B: a segment must be one key (9 hash bits can collide) and carry only ops the closed form knows;
    // list the segment heads
    // (r05 also tried the keys of all eight tiles in flight before the checks: the sixteen registers more spill, 8.8 -> 11.2 us)
    for (uint32_t j = 0; j < ntile; j++) {
      const uint32_t p = j * KVB_T + t;
      const uint64_t cur = Sk[p];
      const bool valid = p < m;
      const uint32_t type = k_type(cur);
      const uint32_t sn = valid ? kvb_below(Mhead, Phead, p + 1) - 1 : 0;  // my segment's number
      const bool head = valid && kvb_bit(Mhead, p);
      if (head && sn < KVB_T) HeadPos[sn] = (uint16_t)p;
      const uint32_t seg_a = valid ? (uint32_t)kvb_last(Mhead, Ehead, 0, p + 1) : 0;
      const uint64_t key = valid ? ld_u64(rep + dint_view_off(V, k_idx(cur), F::MSG) + F::KEY) : 0;
      const uint64_t hkey = valid ? ld_u64(rep + dint_view_off(V, k_idx(Sk[seg_a]), F::MSG) + F::KEY) : 0;
      const uint64_t bm = __ballot(valid && !(key == hkey && (kv_simple_op<WL>(type) || kv_struct_op<WL>(type))));
      if (lane == 0) Mbad[p >> 6] = bm;
    }
    __syncthreads();
    if (wave == 0) kvb_build_pop(Mbad, Pbad);
    KVB_STAMP(8);
    // ---- pass C, one thread per key segment: segments that carry lock ops, per lock quadrant (smallbank: two on one
    // lock word make the bucket run non-simple); segments that insert / delete
    const bool closed = !force_rounds && nseg <= KVB_T;  // else: every request of the stretch runs on its own, in rounds
    const bool mine = closed && t < nseg;
    uint32_t ha = 0, hb = 0;  // my segment = sorted positions [ha, hb)
    if (mine) {
      ha = HeadPos[t];
      hb = t + 1 < nseg ? HeadPos[t + 1] : m;
      if (kvb_popc(Mlop, Plop, ha, hb) != 0) atomicOr((unsigned long long *)&Mlkseg[k_q(Sk[ha])][ha >> 6], 1ull << (ha & 63));
      if (kvb_popc(Mst, Pst, ha, hb) != 0) atomicOr((unsigned long long *)&Mstseg[ha >> 6], 1ull << (ha & 63));
    }
    __syncthreads();
    if (wave < 4) kvb_build_pop(Mlkseg[wave], Plkseg[wave]);
    if (wave == 4) kvb_build_pop(Mstseg, Pstseg);
    __syncthreads();
    // ---- pass D, one thread per bucket run: is it simple?
    if (mine && kvb_bit(Mbh, ha)) {
      const int bx = kvb_first(Mbh, Ebh, ha + 1);
      const uint32_t bk_b = bx >= 0 ? (uint32_t)bx : m;
      bool clash = false;
#pragma unroll
      for (uint32_t k = 0; k < 4; k++) clash |= WL != DINT_WL_TATP && kvb_popc(Mlkseg[k], Plkseg[k], ha, bk_b) > 1;
      const uint32_t nst = kvb_popc(Mstseg, Pstseg, ha, bk_b);
      const bool spans = ha / KVB_T != (bk_b - 1) / KVB_T;  // a run with an insert / delete stays inside one tile
      if (kvb_popc(Mbad, Pbad, ha, bk_b) == 0 && !clash && nst <= 1 && !(nst && spans))
        atomicOr((unsigned long long *)&Mrs[ha >> 6], 1ull << (ha & 63));
    }
    __syncthreads();
    // ---- ... and every request: in a simple run?  the writers and lock ops of the simple runs
    for (uint32_t j = 0; j < ntile; j++) {
      const uint32_t p = j * KVB_T + t;
      const bool valid = p < m;
      const uint32_t type = k_type(Sk[p]);
      const bool simple = valid && closed && kvb_bit(Mrs, (uint32_t)kvb_last(Mbh, Ebh, 0, p + 1));
      const uint64_t m0 = __ballot(simple);
      const uint64_t m1 = __ballot(simple && is_writer(type));
      const uint64_t m2 = __ballot(simple && WL == DINT_WL_TATP && kv_lock_op<WL>(type));
      const uint64_t m3 = __ballot(simple && WL == DINT_WL_TATP && type == 1);
      if (lane == 0) { Msimple[p >> 6] = m0; Mwr[p >> 6] = m1; Mlk[p >> 6] = m2; Macq[p >> 6] = m3; }
      if (WL == DINT_WL_SMALLBANK) {  // the lock ops by kind, for the grant walk below (pass D is done with Mlkseg)
        const uint64_t k0 = __ballot(valid && type == 0), k1 = __ballot(valid && type == 1);
        const uint64_t k2 = __ballot(valid && type == 2), k3 = __ballot(valid && type == 3);
        if (lane == 0) { Mlkseg[0][p >> 6] = k0; Mlkseg[1][p >> 6] = k1; Mlkseg[2][p >> 6] = k2; Mlkseg[3][p >> 6] = k3; }
      }
    }
    __syncthreads();
    if (wave == 0) kvb_build_pop(Mwr, Pwr);
    if (wave == 1) kvb_build_edge(Mwr, Ewr);
    if (wave == 2) kvb_build_edge(Mlk, Elk);
    __syncthreads();
    KVB_STAMP(9);

    // ---- leaders: one thread per simple key segment loads the bucket's inline header (+ smallbank counters) and
    // locates the row; every segment of the stretch at once
    if (t == 0) Sany = 0;
    __syncthreads();
    if (mine) {
      const uint32_t a = ha;
      if (kvb_bit(Msimple, a)) {
        const uint64_t cur = Sk[a];
        const uint32_t gk = kv_cut_gk((uint32_t)(cur >> sh_g), bin, cut), table = kv_table_of(kv, gk), q = k_q(cur);
        const uint64_t bucket = (uint64_t)(gk - kv->gk_base[table]);
        const kv_tab tb = kv->tab[table];
        const uint8_t *ie = kv_entry_ptr(tb, bucket, KV_INLINE);
        kv_hdr H;
        kv_hdr_load(H, ie);
        const uint64_t key = ld_u64(rep + dint_view_off(V, k_idx(cur), F::MSG) + F::KEY);
        uint32_t la0 = 0, lb0 = 0;
        if (WL == DINT_WL_SMALLBANK) {
          const uint2 cc = *(const uint2 *)(ie + KV_SB_LOCK_OFF + 8 * q);
          la0 = cc.x; lb0 = cc.y;
        }
        if (WL == DINT_WL_TATP) la0 = (H.lockw >> (8 * q)) & 0xFFu;
        const kv_where wh = kv_locate(tb, bucket, H, key);
        Lead[t].found_link = (wh.found << 31) | wh.link;
        Lead[t].slot = wh.slot; Lead[t].ver0 = wh.ver; Lead[t].la0 = la0; Lead[t].lb0 = lb0;
        if (WL == DINT_WL_SMALLBANK) { Carry[t].la = la0; Carry[t].lb = lb0; Carry[t].ver = wh.ver; Carry[t].src = -1; Carry[t].miss = 0; }
        if (WL != DINT_WL_SMALLBANK) {
          if (kvb_bit(Mstseg, a)) {  // the segment inserts / deletes
            Crow[t].exists = wh.found; Crow[t].ver = wh.ver; Crow[t].toggles = 0; Crow[t].miss = 0; Crow[t].src = -1;
            Crow[t].bail = kv_has_dup(tb, bucket, H, key, wh) | kv_pool_low(tb);  // duplicate rows of this key, or hardly an overflow entry left: request by request
            Sany = 1;
          }
        }
      }
    }
    __syncthreads();

    // ---- smallbank: which ACQUIREs are granted (smallbank/udp/server_shard.cc:121-147).  The counters have no closed
    // form, but only the grants are inherently serial -- versions and values do not depend on the counters (a COMMIT
    // writes whatever the locks say, :163-173) and come from the write mask like store / tatp.  A key segment is walked
    // 64 requests at a time with the op kinds as ballot masks and the counters in scalar registers (sb_walk: one
    // iteration per mode change, not per request).  First every 64-chunk walks the segments that START in it, all
    // chunks at once; then the segments that run on past their first chunk (at most one per chunk boundary) are taken
    // to their end, one wave each, the op kinds of chunk c read out of lane c's registers.  (r02 walked the stretch
    // wave after wave, tile after tile: 64 barrier-separated steps, 38 us of a hot account's 96 us stretch; this is
    // ~20 us, what is left is the dependent chain of one wave over the ~50 chunks of the hot account.)  The grants land
    // in Mlk (unused by smallbank otherwise), the counters a segment leaves in Carry[segment].
    if (WL == DINT_WL_SMALLBANK) {
      // returns the granted lanes of `rem` (ACQUIREs and RELEASEs of one key in request order); la = num_ex, lb = num_sh
      auto sb_walk = [&](uint64_t rem, uint64_t mAS, uint64_t mAX, uint64_t mRS, uint64_t mRX, uint32_t &la, uint32_t &lb) -> uint64_t {
        return kv_sb_walk(rem, mAS, mAX, mRS, mRX, la, lb);
      };
      for (uint32_t c = wave; c * 64 < m; c += KVB_W) {  // the segments that start in chunk c, as far as the chunk goes
        const uint32_t p = c * 64 + lane;
        const bool simple = p < m && kvb_bit(Msimple, p);
        const uint32_t seg_a = simple ? (uint32_t)kvb_last(Mhead, Ehead, 0, p + 1) : 0;
        const uint32_t si = simple ? kvb_below(Mhead, Phead, p + 1) - 1 : 0;
        const bool here = simple && seg_a >= c * 64;
        uint64_t Gw = 0, todo = __ballot(here);
        const uint64_t mAS = Mlkseg[0][c] & todo, mAX = Mlkseg[1][c] & todo, mRS = Mlkseg[2][c] & todo, mRX = Mlkseg[3][c] & todo;
        while (todo) {
          const int l0 = __ffsll((unsigned long long)todo) - 1;
          const uint32_t a = (uint32_t)__builtin_amdgcn_readlane(seg_a, l0), sa = (uint32_t)__builtin_amdgcn_readlane(si, l0);
          const uint64_t mem = __ballot(here && seg_a == a);
          todo &= ~mem;
          uint32_t la = (uint32_t)__builtin_amdgcn_readfirstlane((int)Carry[sa].la), lb = (uint32_t)__builtin_amdgcn_readfirstlane((int)Carry[sa].lb);
          Gw |= sb_walk(mem & (mAS | mAX | mRS | mRX), mAS, mAX, mRS, mRX, la, lb);
          if ((int)lane == l0) { Carry[sa].la = la; Carry[sa].lb = lb; }
        }
        if (lane == 0) Mlk[c] = Gw;
      }
      __syncthreads();
      uint64_t cross;  // chunk boundaries a segment crosses for the first time: lane c looks at boundary 64 c
      {
        const uint32_t b0 = lane * 64;
        bool x = lane >= 1 && b0 < m && !kvb_bit(Mhead, b0) && kvb_bit(Msimple, b0);
        if (x) x = kvb_last(Mhead, Ehead, 0, b0) >= (int)b0 - 64;  // else it crossed an earlier boundary first: taken there
        cross = __ballot(x);
      }
      const uint64_t vAS = Mlkseg[0][lane], vAX = Mlkseg[1][lane], vRS = Mlkseg[2][lane], vRX = Mlkseg[3][lane];
      for (uint32_t seen = 0; cross; cross &= cross - 1, seen++) {
        if ((seen & (KVB_W - 1)) != wave) continue;
        const uint32_t b0 = 64u * ((uint32_t)__ffsll((unsigned long long)cross) - 1);
        const uint32_t a = (uint32_t)kvb_last(Mhead, Ehead, 0, b0);
        const uint32_t sa = kvb_below(Mhead, Phead, a + 1) - 1;
        const int nx = kvb_first(Mhead, Ehead, b0);
        const uint32_t hb2 = nx >= 0 ? (uint32_t)nx : m;
        uint32_t la = (uint32_t)__builtin_amdgcn_readfirstlane((int)Carry[sa].la), lb = (uint32_t)__builtin_amdgcn_readfirstlane((int)Carry[sa].lb);
        // Lane c holds the op kinds of chunk c.  Most chunks of a contended account cannot change the counters' mode whatever
        // order their requests come in (as lock_2pl's groups, k_locks.hip): HELD and no RELEASE_EXCLUSIVE in the chunk --
        // every ACQUIRE rejected, num_sh -= RELEASE_SHAREDs; FREE, no RELEASE_EXCLUSIVE and no ACQUIRE_EXCLUSIVE that could
        // find num_sh == 0 -- every ACQUIRE_SHARED granted.  So all chunks of the segment are checked AT ONCE under the
        // assumption that the mode holds (num_sh at a chunk's entry = a prefix sum over the chunks before it); up to the first
        // chunk that is not inert the assumption was right, that chunk is walked (sb_walk), and the rest is checked again from
        // the state it leaves.  (r02 - r05 walked the ~60 chunks of a hot account's stretch one after the other: ~10 us of the
        // stretch's 57.)
        {
          const uint32_t c0 = b0 >> 6, c1 = (hb2 - 1) >> 6;
          const bool mine_c = lane >= c0 && lane <= c1;
          const uint32_t tail = hb2 - 64 * c1;  // requests of the segment in its last chunk: 1 .. 64
          const uint64_t in = !mine_c ? 0ull : (lane == c1 && tail < 64 ? (1ull << tail) - 1ull : ~0ull);
          const uint64_t cAS = vAS & in, cAX = vAX & in, cRS = vRS & in, cRX = vRX & in;
          const uint32_t nas = (uint32_t)__popcll(cAS), nrs = (uint32_t)__popcll(cRS);
          uint64_t pend = __ballot((cAS | cAX | cRS | cRX) != 0);
          while (pend) {
            const bool me = (pend >> lane) & 1ull;
            const uint32_t dl = me ? (la != 0 ? 0u - nrs : nas - nrs) : 0u;  // what my chunk adds to num_sh if it is inert
            uint32_t tot, pre = wave_excl_scan_u32(dl, &tot);
            const uint32_t lb_in = lb + pre;
            const bool inert = la != 0 ? cRX == 0 : (cRX == 0 && (cAX == 0 || (lb_in > nrs && lb_in <= 0xFFFFFFFFu - nas)));
            const uint64_t stop = __ballot(me && !inert);
            const uint64_t ok = stop ? pend & ((stop & (0 - stop)) - 1ull) : pend;  // the chunks before the first one that is not
            if (((ok >> lane) & 1ull) && la == 0 && cAS) atomicOr((unsigned long long *)&Mlk[lane], (unsigned long long)cAS);
            if (!stop) { lb += tot; break; }
            const int f = __ffsll((unsigned long long)stop) - 1;
            lb += (uint32_t)__builtin_amdgcn_readlane((int)pre, f);  // (the chunks before f)
            const uint64_t mAS = readlane_u64(cAS, f), mAX = readlane_u64(cAX, f), mRS = readlane_u64(cRS, f), mRX = readlane_u64(cRX, f);
            const uint64_t G = sb_walk(mAS | mAX | mRS | mRX, mAS, mAX, mRS, mRX, la, lb);
            if (lane == 0 && G) atomicOr((unsigned long long *)&Mlk[f], (unsigned long long)G);
            pend &= ~(ok | (1ull << f));
          }
        }
        if (lane == 0) { Carry[sa].la = la; Carry[sa].lb = lb; }
      }
      __syncthreads();
    }

    // ---- tiles: outcomes and replies of the simple segments, 512 requests at a time.  Nothing a tile reads from
    // the table is written before the last tile is done, so the tiles' loads and stores stream back to back.
    const bool walks = WL != DINT_WL_SMALLBANK && Sany;  // workgroup-uniform
    KVB_STAMP(10);
    struct kvb_out { uint8_t *msg; const uint8_t *from; uint32_t ver, code; bool simple, get; };
    auto outcome = [&](uint32_t j, kvb_out &o) {  // tile j: what each request answers, and where a read finds its value
      const uint32_t lo = j * KVB_T, hi = min(lo + KVB_T, m), p = lo + t;
      const bool valid = p < m;
      const uint64_t cur = Sk[p];
      const uint32_t gk = valid ? kv_cut_gk((uint32_t)(cur >> sh_g), bin, cut) : 0, idx = valid ? k_idx(cur) : 0;
      const uint32_t type = k_type(cur), table = valid ? kv_table_of(kv, gk) : 0;
      const uint64_t bucket = valid ? (uint64_t)(gk - kv->gk_base[table]) : 0;
      uint8_t *msg = rep + dint_view_off(V, idx, F::MSG);
      bool simple = valid && kvb_bit(Msimple, p);
      const uint32_t si = simple ? kvb_below(Mhead, Phead, p + 1) - 1 : 0;  // my segment's number = its Lead / Carry slot
      const uint32_t seg_a = simple ? HeadPos[si] : 0;                       // my key segment starts here
      uint32_t found = 0, link = 0, slot = 0, ver0 = 0, la0 = 0;
      if (simple) {
        const kvb_lead L = Lead[si];
        found = L.found_link >> 31; link = L.found_link & 0x7FFFFFFFu; slot = L.slot; ver0 = L.ver0; la0 = L.la0;
      }
      uint32_t my_code = 0, my_ver = 0, my_get = 0;
      int my_src = -1;  // SORTED position of the request whose message holds the value this one reads
      if (WL != DINT_WL_SMALLBANK) {
        if (simple) {
          my_ver = ver0 + (found ? kvb_popc(Mwr, Pwr, seg_a, p) : 0);
          my_src = found ? kvb_last(Mwr, Ewr, seg_a, p) : -1;
          if (WL == DINT_WL_STORE) {
            my_code = type == 0 ? (found ? 3 : 7) : (found ? 5 : 7);
            my_get = (type == 0 && found) ? 1 : 0;
          } else {
            uint32_t lock_seen = la0;
            if (type == 1) {
              const uint32_t q = k_q(cur);
              const uint32_t bk_a = (uint32_t)kvb_last(Mbh, Ebh, 0, p + 1);
              const int bx = kvb_first(Mbh, Ebh, p + 1);
              const uint32_t bk_b = bx >= 0 ? (uint32_t)bx : m;
              int lk;
              if (kvb_popc(Mlkseg[q], Plkseg[q], bk_a, bk_b) > 1) {  // other keys of the bucket use my lock byte
                lk = lock_scan(bk_a, bk_b, q, idx);
              } else {
                lk = kvb_last(Mlk, Elk, seg_a, p);
                if (lk >= 0) lk = (int)kvb_bit(Macq, (uint32_t)lk);
              }
              if (lk >= 0) lock_seen = (uint32_t)lk;
            }
            switch (type) {
              case 0: my_code = found ? 4 : 6; my_get = found; break;
              case 1: my_code = lock_seen ? 8 : 7; break;
              case 2: my_code = 9; break;
              case 12: my_code = 15; break;
              default: my_code = 16; break;  // 13 kCommitBck
            }
          }
        }
        // key segments with an INSERT / DELETE (their bucket run lies inside the tile): the row machine is walked
        // in sorted order, wave after wave, with the state carried through Crow[segment] (as kv_chunk; rare, so
        // the whole step is skipped when the stretch has no such segment)
        if (walks) {
          const bool structural = simple && kvb_bit(Mstseg, seg_a);
          if (t < KVB_W) Mbail[t] = 0;
          __syncthreads();
          for (uint32_t wv = 0; wv < KVB_W; wv++) {
            if (wave == wv) {
              uint64_t todo = __ballot(structural);
              while (todo) {
                const int l0 = __ffsll((unsigned long long)todo) - 1;
                const uint32_t a = (uint32_t)__builtin_amdgcn_readlane(seg_a, l0);
                const uint32_t sa = (uint32_t)__builtin_amdgcn_readlane(si, l0);
                const uint64_t mem = __ballot(structural && seg_a == a);
                todo &= ~mem;
                kv_rowst st = Crow[sa];
                for (uint64_t mm = mem; mm; mm &= mm - 1) {
                  const int l = __ffsll((unsigned long long)mm) - 1;
                  const uint32_t op = (uint32_t)__builtin_amdgcn_readlane(type, l);
                  uint32_t get = 0;
                  const uint32_t ver_seen = st.ver;
                  const int src_seen = st.src;
                  const uint32_t code = kv_row_step<WL>(op, (int)(lo + wv * 64 + l), st, get);
                  if ((int)lane == l) {
                    if (code) my_code = code;
                    my_ver = ver_seen; my_src = src_seen; my_get = get;
                  }
                }
                if ((int)lane == l0) Crow[sa] = st;
              }
            }
            __syncthreads();
          }
          // a segment whose walk bailed out (insert of an existing row) sends its bucket run to the
          // request-by-request path
          const bool bhead = valid && kvb_bit(Mhead, p) && structural;
          uint32_t my_bail = 0;
          if (bhead) { const kv_rowst st = Crow[si]; my_bail = st.bail; }
          const uint64_t bm = __ballot(my_bail != 0);
          if (lane == 0) Mbail[wave] = bm;
          __syncthreads();
          if (simple && (Mbail[0] | Mbail[1] | Mbail[2] | Mbail[3] | Mbail[4] | Mbail[5] | Mbail[6] | Mbail[7])) {
            const uint32_t bk_a = (uint32_t)kvb_last(Mbh, Ebh, 0, p + 1);
            const int bx = kvb_first(Mbh, Ebh, p + 1);
            const uint32_t bk_b = bx >= 0 ? (uint32_t)bx : m;
            if (bk_a >= lo && bk_b <= hi && kvb_range_popc(Mbail, bk_a - lo, bk_b - lo) != 0) simple = false;
          }
          const uint64_t sm = __ballot(simple);
          if (lane == 0) Msimple[p >> 6] = sm;
        }
      } else {
        // smallbank: the grants were settled before the tiles (Mlk); version seen = version + writes below me, value
        // seen = message of the last write below me, both straight from the write mask
        if (simple) {
          const bool granted = kvb_bit(Mlk, p);
          my_ver = ver0 + (found ? kvb_popc(Mwr, Pwr, seg_a, p) : 0);
          my_src = found ? kvb_last(Mwr, Ewr, seg_a, p) : -1;
          switch (type) {  // smallbank/udp/server_shard.cc:121-173
            case 0: my_code = granted ? 7 : 8; my_get = granted ? found : 0; break;
            case 1: my_code = granted ? 9 : 10; my_get = granted ? found : 0; break;
            case 2: my_code = 11; break;
            case 3: my_code = 12; break;
            case 4: my_code = 13; break;
            case 17: my_code = 18; my_get = found; break;  // WARMUP_READ
            default: my_code = 14; break;                   // 5 kCommitBck
          }
        }
      }
      o.msg = msg; o.simple = simple; o.get = simple && my_get != 0; o.code = my_code; o.ver = my_ver; o.from = nullptr;
      if (o.get) {
        const kv_tab tb = kv->tab[table];
        o.from = my_src >= 0 ? rep + dint_view_off(V, k_idx(Sk[my_src]), F::MSG) + F::VAL
                             : kv_entry_ptr(tb, bucket, link) + KV_VAL_OFF + slot * F::VS;
      }
    };
    auto fetch = [&](const kvb_out &o, uint32_t (&w)[10]) {
      if (o.get) {
#pragma unroll
        for (uint32_t k = 0; k < F::VS / 4; k++) w[k] = ld_u32(o.from + 4 * k);
      }
    };
    auto reply = [&](const kvb_out &o, const uint32_t (&w)[10]) {
      if (o.simple) {
        if (o.get) {
#pragma unroll
          for (uint32_t k = 0; k < F::VS / 4; k++) st_u32(o.msg + F::VAL + 4 * k, w[k]);
          st_u32(o.msg + F::VER, o.ver);
        }
        o.msg[F::TYPE] = (uint8_t)o.code;
      }
    };
    for (uint32_t j = 0; j < ntile; j++) {
      kvb_out o;
      uint32_t w[10];
      outcome(j, o);
      fetch(o, w);
      reply(o, w);
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __syncthreads();  // every table read of the stretch precedes the write-backs
    KVB_STAMP(11);

    // ---- write-back: one thread per simple key segment
    if (mine) {
      const uint32_t a = ha;
      if (kvb_bit(Msimple, a)) {
        const uint64_t cur = Sk[a];
        const uint32_t gk = kv_cut_gk((uint32_t)(cur >> sh_g), bin, cut), table = kv_table_of(kv, gk), q = k_q(cur);
        const uint64_t bucket = (uint64_t)(gk - kv->gk_base[table]);
        const kv_tab tb = kv->tab[table];
        uint8_t *ie = kv_entry_ptr(tb, bucket, KV_INLINE);
        const uint32_t seg_b = hb;
        const kvb_lead L = Lead[t];
        const uint32_t found0 = L.found_link >> 31, link = L.found_link & 0x7FFFFFFFu, slot = L.slot, la0 = L.la0, lb0 = L.lb0;
        uint32_t exists1 = found0, fin_ver = L.ver0, fin_la = la0, fin_lb = lb0, nmiss = 0;
        int fin_src = -1;
        bool redo = false;  // the row was deleted and inserted again: it may have moved
        if (WL == DINT_WL_SMALLBANK) {
          const uint32_t nw = kvb_popc(Mwr, Pwr, a, seg_b);
          fin_la = Carry[t].la; fin_lb = Carry[t].lb;
          fin_ver = L.ver0 + (found0 ? nw : 0);
          fin_src = found0 ? kvb_last(Mwr, Ewr, a, seg_b) : -1;
          nmiss = found0 ? 0 : nw + kvb_range_popc(Mlk, a, seg_b);  // every grant and every commit misses the row
        } else if (kvb_bit(Mstseg, a)) {
          const kv_rowst st = Crow[t];
          exists1 = st.exists; fin_ver = st.ver; fin_src = st.src; nmiss = st.miss;
          redo = found0 && exists1 && st.toggles > 1;
        } else {
          const uint32_t nw = kvb_popc(Mwr, Pwr, a, seg_b);
          fin_ver = L.ver0 + (found0 ? nw : 0);
          fin_src = found0 ? kvb_last(Mwr, Ewr, a, seg_b) : -1;
          nmiss = (found0 || WL == DINT_WL_STORE) ? 0 : nw;
        }
        if (WL == DINT_WL_TATP && kvb_popc(Mlop, Plop, a, seg_b) != 0) {  // what the last lock op on my lock byte leaves
          const uint32_t bk_a = (uint32_t)kvb_last(Mbh, Ebh, 0, a + 1);
          const int bx = kvb_first(Mbh, Ebh, a + 1);
          const uint32_t bk_b = bx >= 0 ? (uint32_t)bx : m;
          int lk;
          if (kvb_popc(Mlkseg[q], Plkseg[q], bk_a, bk_b) > 1) {  // shared with other keys of the bucket: every one of
            lk = lock_scan(bk_a, bk_b, q, 0xFFFFFFFFu);          // their segments stores the same byte
          } else {
            lk = kvb_last(Mlk, Elk, a, seg_b);
            if (lk >= 0) lk = (int)kvb_bit(Macq, (uint32_t)lk);
          }
          if (lk >= 0) fin_la = (uint32_t)lk;
        }
        const uint8_t *fin_val = fin_src >= 0 ? rep + dint_view_off(V, k_idx(Sk[fin_src]), F::MSG) + F::VAL : nullptr;
        if (found0 && exists1 && !redo) {  // the row stays where it is: value / version of the last writer
          if (fin_src >= 0) {
            kv_copy_words(kv_entry_ptr(tb, bucket, link) + KV_VAL_OFF + slot * F::VS, fin_val, F::VS);
            kv_entry_hdr(tb, bucket, link)->ver[slot] = fin_ver;
          }
        } else if (found0 != exists1 || redo) {  // apply the net INSERT / DELETE (or DELETE + INSERT) to the chain once
          kv_hdr H;
          kv_hdr_load(H, ie);
          const uint64_t key = ld_u64(rep + dint_view_off(V, k_idx(cur), F::MSG) + F::KEY);
          if (redo) {
            kv_apply<kv_dev_mem>(tb, bucket, H, KV_ACT_DEL, key, nullptr, 0, blockIdx.x);
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            kv_hdr_load(H, ie);
          }
          const kv_res r = kv_apply<kv_dev_mem>(tb, bucket, H, exists1 ? KV_ACT_INS : KV_ACT_DEL, key, (uint8_t *)fin_val,
                                                fin_ver, blockIdx.x);
          if (exists1 && !r.ok) atomicAdd(&stats->pool_exhausted, 1ULL);
        }
        if (WL == DINT_WL_TATP && fin_la != la0) ie[KV_LOCKB_OFF + q] = (uint8_t)fin_la;
        if (WL == DINT_WL_SMALLBANK && (fin_la != la0 || fin_lb != lb0)) *(uint2 *)(ie + KV_SB_LOCK_OFF + 8 * q) = make_uint2(fin_la, fin_lb);
        if (nmiss) atomicAdd(&stats->missing_keys, (unsigned long long)nmiss);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __syncthreads();

    // ---- every other bucket run: request by request.  What has to stay in request order inside a run is less than
    // everything:
    //   W  a request that changes rows (SET / INSERT / DELETE, smallbank's lock ops): after everything before it;
    //   R  a pure read: after the W requests before it -- the reads between two W requests share one round;
    //   A  tatp ACQUIRE_LOCK, B tatp ABORT: they touch the lock byte only, reads and they do not see each other.  The
    //      first ACQUIRE after a W or an ABORT (per lock byte) finds out whether the byte is free; the ACQUIREs that
    //      follow it find the byte taken whatever it answered, so they share the next round.
    // With w = W requests before a request, j = ABORTs before it behind its last W, nB = ABORTs of the run, the round
    // of a request is  w * (3 nB + 3) + {R: 0, first A: 3j, other A: 3j + 1, B: 3j + 2, W: 3 nB + 2};  empty rounds are
    // skipped through a bitmap.  A run of ONE key segment lies in request order, so a wave numbers it 64 requests at
    // a time from ballots and a carry.  A run of several segments (sorted by key first), or one that would need more
    // than KVB_RMAX rounds, numbers its requests 2v + 1 (not a read) / 2v (a read), v = requests before it that are
    // not reads.
    // A hot call-forwarding row that is deleted and inserted again and again is such a run (kvs_insert never checks
    // for an existing row, so the reference's population holds duplicate rows of some keys: no closed form covers
    // those): a hundred READs and dozens of refused ACQUIREs around one DELETE went one request per round, ~1.6 us
    // each -- the p99 of r02's epoch latency.
    enum : uint32_t { KVB_RMAX = 32768u };
    uint16_t *Rpos = (uint16_t *)Lead;  // by sorted position; Lead and CarryCrow are free now
    uint32_t *Rbits = (uint32_t *)CarryCrow;
    static_assert(sizeof(Lead) >= 2 * KVB_NMAX && sizeof(CarryCrow) >= KVB_RMAX / 8, "round numbering scratch");
    auto r_class = [&](uint32_t type) -> uint32_t {  // 0 R, 1 A, 2 B, 3 W
      if (kv_pure_read<WL>(type)) return 0u;
      if (WL == DINT_WL_TATP && type == 1) return 1u;
      if (WL == DINT_WL_TATP && type == 2) return 2u;
      return 3u;
    };
    KVB_STAMP(12);
    bool ns = false;  // anything left for the rounds? (a hot account's stretch: nothing)
    for (uint32_t j = 0; j < ntile; j++) ns |= j * KVB_T + t < m && !kvb_bit(Msimple, j * KVB_T + t);
    uint32_t nrounds = 0;
    if (__syncthreads_or((int)ns)) {
    for (uint32_t k = t; k < KVB_RMAX / 32; k += KVB_T) Rbits[k] = 0;
    uint32_t mylen = 0, tot;
    for (uint32_t j = 0; j < ntile; j++) {  // runs of several key segments: every request counts its predecessors
      const uint32_t p = j * KVB_T + t;
      uint32_t pos = 0xFFFFu;
      if (p < m && !kvb_bit(Msimple, p)) {
        const uint32_t bk_a = (uint32_t)kvb_last(Mbh, Ebh, 0, p + 1);
        const int bx = kvb_first(Mbh, Ebh, p + 1);
        const uint32_t bk_b = bx >= 0 ? (uint32_t)bx : m, myidx = k_idx(Sk[p]);
        if (kvb_popc(Mhead, Phead, bk_a, bk_b) == 1) {
          pos = 0xFFFEu;  // numbered by a wave below
        } else {
          uint32_t v = 0, nv = 0;
          for (uint32_t k = bk_a; k < bk_b; k++) {
            const bool other = !kv_pure_read<WL>(k_type(Sk[k]));
            nv += other;
            v += other && k_idx(Sk[k]) < myidx;
          }
          pos = 2u * v + (kv_pure_read<WL>(k_type(Sk[p])) ? 0u : 1u);
          if (p == bk_a) mylen = max(mylen, 2u * nv + 1u);
        }
      }
      if (p < m) Rpos[p] = (uint16_t)pos;
    }
    __syncthreads();
    {  // runs of one key segment: the waves take them in turn
      const uint64_t below = (1ull << lane) - 1ull;
      auto top = [](uint64_t x) -> int { return x ? 63 - __clzll((long long)x) : -1; };        // highest set bit
      auto above = [](int bit) -> uint64_t { return bit < 0 ? ~0ull : bit >= 63 ? 0ull : ~0ull << (bit + 1); };
      uint32_t seen = 0;
      for (uint32_t wd = 0; wd * 64 < m; wd++) {
        for (uint64_t hb = Mbh[wd] & ~Msimple[wd]; hb; hb &= hb - 1) {
          const uint32_t ra = wd * 64 + (uint32_t)__ffsll((unsigned long long)hb) - 1;
          if ((seen++ & (KVB_W - 1)) != wave || Rpos[ra] != 0xFFFEu) continue;
          const int bx = kvb_first(Mbh, Ebh, ra + 1);
          const uint32_t rb = bx >= 0 ? (uint32_t)bx : m;
          uint32_t nW = 0, nB = 0, nV = 0;
          for (uint32_t c0 = ra; c0 < rb; c0 += 64) {
            const uint32_t p = c0 + lane, cl = p < rb ? r_class(k_type(Sk[p])) : 0u;
            nW += (uint32_t)__popcll(__ballot(cl == 3));
            nB += (uint32_t)__popcll(__ballot(cl == 2));
            nV += (uint32_t)__popcll(__ballot(cl != 0));
          }
          const uint32_t stride = 3u * nB + 3u;
          const bool fancy = (nW + 1u) * stride <= KVB_RMAX;
          if (lane == 0) mylen = max(mylen, fancy ? (nW + 1u) * stride : 2u * nV + 1u);
          uint32_t wc = 0, jc = 0, af = 0, vc = 0;  // carries: W so far, ABORTs behind the last W, lock bytes with an ACQUIRE behind the last W / ABORT, not-reads so far
          for (uint32_t c0 = ra; c0 < rb; c0 += 64) {
            const uint32_t p = c0 + lane;
            const bool valid = p < rb;
            const uint64_t cur = valid ? Sk[p] : 0;
            const uint32_t cl = valid ? r_class(k_type(cur)) : 0u, q = k_q(cur);
            const uint64_t mW = __ballot(cl == 3), mB = __ballot(cl == 2), mV = __ballot(cl != 0);
            uint64_t mA[4];
#pragma unroll
            for (uint32_t i = 0; i < 4; i++) mA[i] = __ballot(cl == 1 && q == i);
            uint32_t pos;
            if (fancy) {
              const int lw = top(mW & below), lwb = top((mW | mB) & below);
              const uint32_t w = wc + (uint32_t)__popcll(mW & below);
              const uint32_t jj = (lw < 0 ? jc : 0u) + (uint32_t)__popcll(mB & below & above(lw));
              const uint64_t mine = q == 0 ? mA[0] : q == 1 ? mA[1] : q == 2 ? mA[2] : mA[3];
              const bool follows = (mine & below & above(lwb)) != 0 || (lwb < 0 && ((af >> q) & 1u));
              pos = w * stride + (cl == 0 ? 0u : cl == 1 ? 3u * jj + (follows ? 1u : 0u) : cl == 2 ? 3u * jj + 2u : stride - 1u);
            } else {
              pos = 2u * (vc + (uint32_t)__popcll(mV & below)) + (cl ? 1u : 0u);
            }
            if (valid) Rpos[p] = (uint16_t)pos;
            const int tw = top(mW), twb = top(mW | mB);
            wc += (uint32_t)__popcll(mW);
            jc = (tw < 0 ? jc : 0u) + (uint32_t)__popcll(mB & above(tw));
            if (twb >= 0) af = 0;
#pragma unroll
            for (uint32_t i = 0; i < 4; i++) af |= (mA[i] & above(twb)) ? 1u << i : 0u;
            vc += (uint32_t)__popcll(mV);
          }
        }
      }
    }
    __syncthreads();
    for (uint32_t j = 0; j < ntile; j++) {
      const uint32_t p = j * KVB_T + t;
      const uint32_t pos = p < m ? Rpos[p] : 0xFFFFu;
      if (pos != 0xFFFFu) atomicOr(&Rbits[pos >> 5], 1u << (pos & 31));
    }
    {  // rounds of the longest run (block max)
      uint32_t mx = mylen;
      for (int d = 32; d > 0; d >>= 1) mx = max(mx, (uint32_t)__shfl_xor(mx, d, 64));
      if (lane == 0) Sred[wave] = mx;
      __syncthreads();
      tot = 0;
      for (uint32_t w = 0; w < KVB_W; w++) tot = max(tot, Sred[w]);
    }
    for (uint32_t rw = 0; rw * 32 < tot; rw++) {
      for (uint32_t bits = Rbits[rw]; bits; bits &= bits - 1) {  // the same word for every thread
        const uint32_t r = rw * 32 + (uint32_t)__ffs((int)bits) - 1;
        nrounds++;
        for (uint32_t j = 0; j < ntile; j++) {
          const uint32_t p = j * KVB_T + t;
          if (p < m && Rpos[p] == r) {
            const uint64_t cur = Sk[p];
            const uint32_t gk = kv_cut_gk((uint32_t)(cur >> sh_g), bin, cut), table = kv_table_of(kv, gk);
            kv_do_request<WL>(rep + dint_view_off(V, k_idx(cur), F::MSG), k_type(cur), table, k_q(cur),
                              (uint64_t)(gk - kv->gk_base[table]), kv, stats);
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        __syncthreads();
      }
    }
    }
    if (wtr && t == 0) { wtr[14] += nrounds; wtr[15] += 1; }  // rounds of the request-by-request fallback; stretches
    KVB_STAMP(13);
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __syncthreads();  // the next stretch sees this stretch's stores; LDS arrays are free again
  }
#undef KVB_STAMP
}

// ---- k_kv_resolve: one workgroup per coarse bin -------------------------------------------------------------------
// LDS of the workgroup: the records of the bin's small subs, split by sub (KVR_LCAP * 16 B) -- or, afterwards, the big
// path's stretch machinery (kvb_lds); the two never live at the same time and share one buffer.
#define KVR_LCAP 1024u   // records of a coarse bin's small subs that fit the LDS split (the average bin holds ~512)
#define KVR_F 64u        // subs per coarse bin = lanes of the wave that lays them out
#define KVR_NPMAX 64u     // pieces of one hot key at most (a key of up to ~24,000 requests per pass -- the hottest subscriber of a
                          // 2M-client tatp pass has 16,000; beyond: the late list.  r05 / early r06: 32)
struct kvr_lds {
  uint4 rec[KVR_LCAP];         // records of the small subs, sub after sub
  uint32_t hist[KVR_F];        // records per sub
  uint32_t cur[KVR_F];         // ... placed so far
  uint32_t off[KVR_F + 1];     // start of each small sub in rec[]
  uint32_t bigoff[KVR_F];      // start of each big sub's 8-byte records in ovf[], KV_NONE for a small sub
  uint2 chs[KVR_F];            // chunks: [first, last) record in rec[]
  uint32_t nch;
  // hot keys (kv_hot_item): per sub, the key of one of its records -- of a sub that is one hot key, almost surely that key --
  // and that record in the big path's form (bucket group, key-hash bits, lock quadrant)
  uint32_t cflag[KVR_F];
  uint64_t ckey[KVR_F], hrec[KVR_F];
};
// which piece a request index belongs to: monotone in idx (a piece is a range of request indices), ~n / np indices each
__device__ static inline uint32_t kv_piece_of(uint32_t idx, uint32_t np, uint32_t inv_n) {
  const uint32_t p = (uint32_t)(((uint64_t)idx * np * inv_n) >> 32);
  return p < np ? p : np - 1;
}
// work items of k_kv_big (bigq): KVQ_W uint4 each
//   [0] = {bin = coarse bin + C * sub, offset of the sub in ovf, records of the sub, kind | piece << 2 | pieces << 8}
//   [1] = kind 0: unused.  Else {hot key lo, hi, item index of the sub's first item, -}
//   [2] = kind 0: unused.  Else {one record of the hot key (big path's form) lo, hi, -, -}
#define KVQ_W 3u
enum : uint32_t { KVQ_SUB = 0, KVQ_PIECE = 1, KVQ_REM = 2, KVQ_SOLO = 3 };

// One wave lists the big subs of a coarse bin as work items (k_kv_hot / the workers of k_kv_pass / k_kv_big).
// A sub of at least split_min records is listed as `np` hot-key PIECES (ranges of the request index) + its REMAINDER (the other
// keys), side by side -- or, when one piece is enough, as one SOLO item (kv_hot_item); every item names the whole sub.  Entries
// and "listed" tags are agent-scope stores, entries first: a worker that sees item i's tag sees the item and its records.
// Last: this bin counts as listed (big[6]) -- when all C have, big[3] is final.
__device__ static inline void kv_list_items(const kv_pass_args &A, uint32_t b, const kvr_lds &L, const uint2 *Sbig) {
  const uint32_t t = threadIdx.x & 63u;
  const uint2 bs = Sbig[t];
  // A piece is a range of the request INDEX, and the requests of a segmented pass (the closed loop's batches, the exchange's
  // slots) fill only the front of each segment: a range that lies in a filled stretch holds n / live times the average.  The
  // pieces are sized for that -- r05 / early r06 sized them for the average, every hot key of a closed-loop pass (slots 80 %
  // full) had a piece of more than KVB_T requests, said so, and went the slow way: k_kv_late 240 us per epoch.
  uint32_t target = A.split_target;
  if (A.V.seg_cap) {
    uint32_t live = t < A.V.n_seg ? min(A.V.seg_cap, *(const uint32_t *)(A.V.cnt + (size_t)t * A.V.cnt_stride)) : 0u, tot;
    (void)wave_excl_scan_u32(live, &tot);
    if (tot < A.n) target = max(8u, (uint32_t)(((uint64_t)target * max(tot, 1u)) / A.n));
  }
  const uint32_t np0 = (bs.y + target - 1) / target;
  // (more than the pieces can hold: k_kv_late's / kv_big_bin's, via the late list; smallbank (np_max = KSB_NPMAX): pieces or nothing)
  // (smallbank, one piece: a SOLO item of kv_sb_item -- its closed form covers a second key on the row's counter pair)
  const bool hot = bs.y >= A.split_min && np0 <= A.np_max;
  const uint32_t np = hot ? np0 : 0u;
  const uint32_t nent = bs.y ? (np > 1 ? np + 1 : 1u) : 0u;
  uint32_t tot, at = wave_excl_scan_u32(nent, &tot);
  uint32_t base = 0;
  if (t == 0 && tot) base = atomicAdd(&A.big[3], tot);
  base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
  at += base;
  const uint32_t bin = b + A.cut.P * t;
  if (nent && !hot) {
    kv_st_agent(A.bigq + KVQ_W * (size_t)at, make_uint4(bin, bs.x, bs.y, KVQ_SUB));
  } else if (nent) {
    uint64_t hk = L.ckey[t], hr = L.hrec[t];
    if (A.sb_pieces) {
      // smallbank: WHICH row the pieces are cut around decides whether the sub is answered in closed form.  "Whichever record came
      // first" is the hot row's nine times out of ten in a sub of thousands; in a sub of 200 .. 1,200 records (a warm row among cold
      // ones) it was a cold row's often enough that the warm row and a neighbour on its counter pair went to kv_big_bin's rounds
      // in the remainder: 0.2 .. 1.5 ms, the passes behind the p99 (r06b trace).  Eight records from across the sub vote.
      using F = Fmt<DINT_WL_SMALLBANK>;
      const uint32_t sh_g = 16 + A.cut.ibits;
      uint64_t smp[8];
#pragma unroll
      for (uint32_t j = 0; j < 8; j++) smp[j] = kv_ld_agent(&A.ovf[bs.x + (uint32_t)(((uint64_t)(2 * j + 1) * bs.y) >> 4)]);
      uint32_t best = 0, bj = 0;
#pragma unroll
      for (uint32_t j = 0; j < 8; j++) {
        uint32_t c = 0;
#pragma unroll
        for (uint32_t i = 0; i < 8; i++) c += (smp[i] >> sh_g) == (smp[j] >> sh_g) && pay_q(kv_rec_pay(smp[i])) == pay_q(kv_rec_pay(smp[j]));
        if (c > best) { best = c; bj = j; }
      }
      hr = smp[0];
#pragma unroll
      for (uint32_t j = 1; j < 8; j++) if (j == bj) hr = smp[j];
      const uint32_t hidx = (uint32_t)(hr >> 16) & (uint32_t)((1ull << A.cut.ibits) - 1ull);
      hk = ld_u64(A.rep + dint_view_off(A.V, hidx, F::MSG) + F::KEY);
    }
    for (uint32_t p = 0; p < nent; p++) {
      const uint32_t kind = np == 1 ? KVQ_SOLO : (p < np ? KVQ_PIECE : KVQ_REM);
      uint4 *q = A.bigq + KVQ_W * (size_t)(at + p);
      kv_st_agent(q, make_uint4(bin, bs.x, bs.y, kind | (p << 2) | (np << 10)));
      kv_st_agent(q + 1, make_uint4((uint32_t)hk, (uint32_t)(hk >> 32), at, 0u));
      kv_st_agent(q + 2, make_uint4((uint32_t)hr, (uint32_t)(hr >> 32), 0u, 0u));
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  for (uint32_t p = 0; p < nent; p++)
    if (at + p < DINT_KV_BIGQ_MAX) __hip_atomic_store(&A.bigrdy[at + p], A.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (t == 0) atomicAdd(&A.big[6], 1u);
}

template <int WL, uint32_t NT>
__device__ static inline void kv_coarse_bin(const kv_pass_args &A, const kv_dev *kv, uint32_t coarse, uint8_t *lds_raw,
                                            uint2 *Sbig /* [KVR_F] {offset in ovf, records} of the bin's big subs */,
                                            uint32_t cnt, const uint4 &r0, const uint4 &r1, uint64_t *tr) {
  kvr_lds &L = *(kvr_lds *)lds_raw;
  const uint32_t t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const kv_cut &cut = A.cut;
  const uint32_t C = cut.P, cap = A.cap, sh = 16 + cut.ibits;
  const uint32_t idx_mask = (uint32_t)((1ull << cut.ibits) - 1ull);
  const uint4 *__restrict__ recs = A.kbins + (size_t)coarse * cap;
  if (cnt == 0) {  // workgroup-uniform: nothing to resolve, nothing to list
    if (t == 0) atomicAdd(&A.big[6], 1u);
    return;
  }
  const uint32_t n_in = min(cnt, cap);
  const uint32_t novl = cnt > cap ? A.big[1] : 0u;  // my records beyond `cap` are somewhere in the pass's overflow list
  // (the first two records of every thread -- r0, r1, loaded by the kernel together with the counter -- stay in
  // registers between the two phases: the usual bin is read once)
  if (t == 0) A.bin_cnt[coarse] = 0;  // leave the counters clean for the next pass (every thread read it before the kernel's barrier)
  auto for_each_record = [&](auto &&f) {
    if (t < n_in) f(r0);
    if (t + NT < n_in) f(r1);
    // (a hot key's bin: thousands of records in place -- eight loads per thread in flight; one at a time was 31 dependent round
    // trips, 50 us, for smallbank's 16,000-record bin in each of the two phases, r05 trace)
    for (uint32_t k0 = 2 * NT; k0 < n_in; k0 += 8 * NT) {
      uint4 q4[8];
#pragma unroll
      for (uint32_t j = 0; j < 8; j++) {
        const uint32_t k = k0 + j * NT + t;
        q4[j] = k < n_in ? recs[k] : make_uint4(0, 0, 0, 0);
      }
#pragma unroll
      for (uint32_t j = 0; j < 8; j++)
        if (k0 + j * NT + t < n_in) f(q4[j]);
    }
    // (the pass's overflow list, all of it, by every bin that has records there -- smallbank at Zipf 0.99: 40,000 entries, a
    // dozen such bins: eight loads per thread in flight, not one -- r05 trace: 2 x 50 us of a 120 us workgroup were this loop)
    for (uint32_t k0 = 0; k0 < novl; k0 += 8 * NT) {
      uint32_t b8[8];
#pragma unroll
      for (uint32_t j = 0; j < 8; j++) {
        const uint32_t k = k0 + j * NT + t;
        b8[j] = k < novl ? A.ovl[2 * (size_t)k + 1].x : KV_NONE;
      }
#pragma unroll
      for (uint32_t j = 0; j < 8; j++)
        if (b8[j] == coarse) f(A.ovl[2 * (size_t)(k0 + j * NT + t)]);
    }
  };
  // ---- phase A: records per sub; and per sub the key of whichever record comes first (a sub of hundreds of records is one
  // hot key's: k_kv_big cuts it into pieces around that key, kv_hot_item)
  auto big_rec = [&](uint64_t m) -> uint64_t { return ((m >> (sh + 6)) << sh) | (m & ((1ull << sh) - 1ull)); };  // group / (64 C) | idx | payload
  for_each_record([&](const uint4 &r) {
    const uint64_t m = u4_meta(r);
    const uint32_t sub = (uint32_t)(m >> sh) & (KVR_F - 1);
    // (a hot key's bin is one sub: the lanes that share the first active lane's sub count themselves with ONE LDS atomic)
    const uint32_t sub0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)sub);
    const uint64_t same = __ballot(sub == sub0);
    if (sub != sub0) atomicAdd(&L.hist[sub], 1u);
    else if (lane == (uint32_t)__ffsll((unsigned long long)same) - 1) atomicAdd(&L.hist[sub], (uint32_t)__popcll(same));
    if (L.cflag[sub] == 0 && atomicCAS(&L.cflag[sub], 0u, 1u) == 0u) { L.ckey[sub] = u4_key(r); L.hrec[sub] = big_rec(m); }
  });
  __syncthreads();
  if (tr && t == 0) tr[2] = __builtin_amdgcn_s_memrealtime();
  // ---- layout (one wave, a lane per sub): small subs get a range of rec[], big subs a range of ovf[]; neighbouring
  // small subs are packed into chunks of <= 64 records, greedily
  if (wave == 0) {
    const uint32_t h = L.hist[lane];
    bool big = h > 64;
    uint32_t stot, sc = big ? 0u : h, soff = wave_excl_scan_u32(sc, &stot);
    if (stot > A.lcap) {  // more small-sub records than the LDS split holds (wave-uniform): every sub takes the big path
      big = h > 0; sc = 0; soff = 0; stot = 0;
    }
    uint32_t btot = 0, boff = 0, gbase = 0;
    if (__ballot(big)) {  // (wave-uniform; the usual bin has no big sub)
      boff = wave_excl_scan_u32(big ? h : 0u, &btot);
      if (lane == 0) {
        gbase = atomicAdd(&A.big[0], btot);
        atomicAdd(&A.stats->big_bin_requests, (unsigned long long)btot);
      }
      gbase = (uint32_t)__builtin_amdgcn_readfirstlane((int)gbase);
    }
    L.off[lane] = soff;
    if (lane == 63) L.off[KVR_F] = stot;
    L.bigoff[lane] = big ? gbase + boff : KV_NONE;
    Sbig[lane] = make_uint2(big ? gbase + boff : 0u, big ? h : 0u);
    // next[l] = the sub after the chunk that starts at sub l: the largest e in (l, 64] with off[e] - off[l] <= 64
    // (a small sub alone always fits).  Binary search over the lanes' registers, the same trip count for every lane.
    uint32_t lo = lane + 1, hi = KVR_F;
#pragma unroll
    for (int it = 0; it < 7; it++) {
      const uint32_t mid = min((lo + hi + 1) >> 1, KVR_F);
      const uint32_t v = (uint32_t)__shfl((int)soff, (int)(mid & (KVR_F - 1)), 64);
      const uint32_t om = mid == KVR_F ? stot : v;
      if (lo < hi) {
        if (om - soff <= 64u) lo = mid; else hi = mid - 1;
      }
    }
    uint32_t s = 0, nch = 0;
    while (s < KVR_F) {  // wave-uniform walk over the chunk starts
      const uint32_t e = (uint32_t)__builtin_amdgcn_readlane((int)lo, (int)s);
      const uint32_t os = (uint32_t)__builtin_amdgcn_readlane((int)soff, (int)s);
      const uint32_t oe = e >= KVR_F ? stot : (uint32_t)__builtin_amdgcn_readlane((int)soff, (int)e);
      if (oe > os) {
        if (lane == 0) L.chs[nch] = make_uint2(os, oe);
        nch++;
      }
      s = e;
    }
    if (lane == 0) L.nch = nch;
    if (tr && lane == 0) { tr[14] = cnt; tr[15] = btot; tr[16] = nch; }
  }
  __syncthreads();
  if (tr && t == 0) tr[3] = __builtin_amdgcn_s_memrealtime();
  // ---- phase B: every record to its sub's range
  for_each_record([&](const uint4 &r) {
    const uint64_t m = u4_meta(r);
    const uint32_t sub = (uint32_t)(m >> sh) & (KVR_F - 1);
    const uint32_t sub0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)sub);
    const uint64_t same = __ballot(sub == sub0);
    const int lead = __ffsll((unsigned long long)same) - 1;
    uint32_t pos = 0;
    if (sub != sub0) pos = atomicAdd(&L.cur[sub], 1u);
    else if ((int)lane == lead) pos = atomicAdd(&L.cur[sub], (uint32_t)__popcll(same));
    if (sub == sub0) pos = (uint32_t)__builtin_amdgcn_readlane((int)pos, lead) + (uint32_t)__popcll(same & lanemask_lt());
    const uint32_t bo = L.bigoff[sub];
    if (bo == KV_NONE) L.rec[L.off[sub] + pos] = r;
    else kv_st_agent(&A.ovf[bo + pos], big_rec(m));  // the big path's record (read by another workgroup of this launch: kv_st_agent)
  });
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (every record is where a worker will look for it before the items are listed)
  __syncthreads();
  if (tr && t == 0) tr[4] = __builtin_amdgcn_s_memrealtime();
  // ---- the bin's big subs become work items NOW (r06), before the chunks: the hot-key workers of this launch take them while
  // this workgroup resolves its small subs.  The last wave lists (wave 0 starts the first chunk).
  if (wave == NT / 64 - 1) kv_list_items(A, coarse, L, Sbig);
  // ---- the chunks, one wave each: sort by (group / C, key hash, idx) in registers -- groups commute, so any order that
  // keeps each group's requests in idx order is serial-equivalent, and after the sort the requests of a group sit in
  // adjacent lanes.  The sort word carries the lane the record came from; key and payload follow by one shuffle each.
  const uint32_t nch = L.nch;
  for (uint32_t ch = wave; ch < nch; ch += NT / 64) {
    const uint2 ab = L.chs[ch];
    const uint32_t c = ab.y - ab.x;
    const bool has = lane < c;
    const uint4 r = has ? L.rec[ab.x + lane] : make_uint4(0, 0, 0, 0);
    const uint64_t key0 = u4_key(r), m = u4_meta(r);
    const uint32_t pay0 = kv_rec_pay(m);
    uint64_t w = ~0ull;  // empty lanes sort last
    if (has) w = ((m >> sh) << (15 + cut.ibits)) | ((uint64_t)pay_kh(pay0) << (6 + cut.ibits)) | ((uint64_t)((uint32_t)(m >> 16) & idx_mask) << 6) | lane;
    w = wave_sort_u64(w);
    uint64_t *ctr = (tr && ch == 0) ? tr + 2 : nullptr;  // the first chunk's kv_chunk stamps 4 .. 8 land in tr[6 .. 10]
    if (ctr && lane == 0) tr[5] = __builtin_amdgcn_s_memrealtime();
    const bool valid = lane < c;
    const int src = (int)((uint32_t)w & 63u);
    const uint64_t key = shfl_u64(key0, src);
    const uint32_t pay = (uint32_t)__shfl((int)pay0, src, 64);
    const uint32_t gk = kv_cut_gk((uint32_t)(w >> (15 + cut.ibits)), coarse, cut), kh = (uint32_t)(w >> (6 + cut.ibits)) & 511u;
    const uint32_t idx = (uint32_t)(w >> 6) & idx_mask;
    kv_chunk<WL>(A.rep, valid, valid ? idx : 0, gk, kh, pay_type(pay), valid ? kv_table_of(kv, gk) : 0, pay_q(pay), key, kv, A.stats,
                 A.force_flags & 1, true, A.V, ctr);
  }
}

// The big subs (hot keys) are not resolved here but listed for k_kv_big, a launch of its own behind this one: the chunk
// workgroups then carry 19 KB of LDS and ~115 VGPRs without scratch (two workgroups = 16 waves per CU), and the big path
// -- compiled on its own at 256 VGPRs -- spills 3 .. 40 dwords per lane instead of r03's 62 in the shared kernel
// (profiles/r05_kernel_resources.txt).  Measured against
// (a) resolving them in place, behind the bin's chunks, and (c) handing the hot ones to worker workgroups at the end of
// THIS launch, which take them as they are listed (agent-scope stores, a ticket per worker: parity-green, no hang) -- the
// overlap that costs the big path its registers again (one kernel, 128 VGPRs, ~100 spilled): TATP 1,750 against 2,170
// Mtxn/s, store 2,370 against 3,290.  NOTEBOOK.md section 1.
template <int WL>
__device__ __forceinline__ static void kv_resolve_role(const kv_pass_args &A, uint32_t b, kv_dev &Skv, uint8_t *Lraw, uint2 *Sbig) {
  constexpr uint32_t NT = KVB_T;
  const uint32_t t = threadIdx.x;
  uint64_t *tr = A.trace ? A.trace + 32 * (size_t)b : nullptr;  // per engine (its own trace buffer), by ITS bin: < 2048 * 32, where k_kv_big's rows start
  if (tr && t == 0) tr[0] = __builtin_amdgcn_s_memrealtime();
  // everything the workgroup needs from memory before its LDS phases, in flight together: the table descriptors, the
  // bin's record count and -- without waiting for the count: the bin's region always exists -- its first 2 x NT records
  // (the counter is loaded LAST: the compiler makes it a scalar at once -- a wait -- and the loads issued before it ride
  // on the same round trip)
  static_assert(sizeof(kv_dev) / 4 <= NT, "one word of the table descriptors per thread");
  const uint4 *__restrict__ recs = A.kbins + (size_t)b * A.cap;
  const uint4 r0 = t < A.cap ? recs[t] : make_uint4(0, 0, 0, 0);
  const uint4 r1 = t + NT < A.cap ? recs[t + NT] : make_uint4(0, 0, 0, 0);
  const uint32_t kvw = t < sizeof(kv_dev) / 4 ? ((const uint32_t *)A.kv)[t] : 0u;
  const uint32_t cnt = A.bin_cnt[b];
  if (t < sizeof(kv_dev) / 4) ((uint32_t *)&Skv)[t] = kvw;
  if (t < KVR_F) {
    Sbig[t] = make_uint2(0u, 0u);
    kvr_lds &L = *(kvr_lds *)Lraw;
    L.hist[t] = 0; L.cur[t] = 0; L.cflag[t] = 0;
  }
  if (b == 0) {  // the control words of the pass AFTER NEXT: the next pass's partition may be running beside this workgroup (k_kv_pass)
    if (t < 16) A.big_z[t] = 0;
    for (uint32_t k = t; k < 1024; k += NT) A.blk_pub_z[k] = 0;
  }
  __syncthreads();
  kv_dev_pend_set(Skv, A.pno);  // (the LDS phases of kv_coarse_bin put barriers between this and the first table access)
  if (tr && t == 0) tr[1] = __builtin_amdgcn_s_memrealtime();
  // (r04b also prefetched every record's bucket header here, ~6 us of LDS work ahead of the chunk that needs it: the
  // header round trip of the chunk fell from 2.4 to 1.6 us and the bench lost 3 % -- the prefetch is one more transaction
  // per request on a memory system that is the bottleneck once three engines run side by side.  Removed.)
  kv_coarse_bin<WL, NT>(A, &Skv, b, Lraw, Sbig, cnt, r0, r1, tr);
  if (tr && t == 0) { tr[11] = tr[12] = __builtin_amdgcn_s_memrealtime(); tr[13] = b; }
}
template <int WL>
__global__ void __launch_bounds__(KVB_T, 4) k_kv_resolve(kv_multi_args M, uint32_t n_eng) {
  __shared__ kv_dev Skv;  // table descriptors: per-lane lookups by table id become LDS reads
  __shared__ __attribute__((aligned(16))) uint8_t Lraw[sizeof(kvr_lds)];
  __shared__ uint2 Sbig[KVR_F];
  uint32_t e = 0, b = blockIdx.x;
  while (e + 1 < n_eng && b >= M.e[e].cut.P) { b -= M.e[e].cut.P; e++; }
  kv_resolve_role<WL>(M.e[e], b, Skv, Lraw, Sbig);
}

// ---- hot keys: one key's hundreds or thousands of requests, in closed form, several workgroups at once ---------------------
// A big sub is (nearly) one key -- tatp's hot subscribers at Zipf-0.8: 100 .. 4,000 requests of a 240k-request pass each, a
// hundred such subs per pass -- and kv_big_bin's stretch machinery took 20 .. 50 us for it on ONE workgroup while the pass
// waited ("several workgroups per hot key"; r05 measured 42 % of an epoch behind k_kv_big).  k_kv_resolve
// names a candidate key per sub (a record's, compared in full here) and lists a sub of `split_min` records or more as `np`
// PIECES -- piece j = the key's requests whose index falls into the j-th of np ranges (kv_piece_of), <= KVB_T of them, one
// per thread -- and a REMAINDER (every record of another (bucket group, key hash)); np = 1: one SOLO item for both.  The
// closed forms of kv_chunk hold across pieces because what a request sees of its predecessors is tiny:
//     version seen   = v0 + writers before me        value seen = message of the last writer before me (else the row)
//     lock byte seen = what the last lock op before me left (ACQUIRE: 1, ABORT / COMMIT_PRIM: 0; else the stored byte)
// so a piece needs from the pieces before it only {writers, index of the last writer, what the last lock op left}: one 64-bit
// word per item (kvh_word), published with an agent-scope store once the piece has its requests in order and read by all its
// siblings (items are handed out by ticket and siblings sit side by side in the list: whoever waits, waits for workgroups
// that are running or that draw the very next tickets).  Every piece reads the bucket header and the row BEFORE it publishes,
// and stores to the table only after it has seen every sibling's word: no write-back overtakes a sibling's read.
// ALL OR NOTHING: a piece that finds an op outside the closed form (INSERT / DELETE of the row), more than KVB_T requests, or a
// different key behind the candidate's hash bits -- or a remainder in which another key of the hot BUCKET restructures the
// chain or uses the hot key's lock byte -- says so in its word; then item 0 resolves the whole sub the old way (kv_big_bin
// over the sub's records, which nobody has touched) and its siblings do nothing.  Otherwise every piece answers its requests
// at once, the piece with the pass's last writer stores row and version, the one with the last lock op the lock byte, and the
// remainder goes through kv_big_bin beside them (its records compacted into ovf2).
// Semantics per op: tatp/udp/server_shard.cc:116-168, store/udp/server.cc:75-97 (as kv_do_request).
//   kvh_word: [1:0] last lock op {left the byte set, there is one} | [21:2] request index of the last writer | [22] there is one |
//             [32:23] writers | [33] ok | [63:34] the pass's tag
__device__ static inline unsigned long long kvh_word(uint32_t seq, bool ok, uint32_t nwr, int lw_idx, int ll_acq) {
  return ((unsigned long long)(seq & 0x3FFFFFFFu) << 34) | ((unsigned long long)(ok ? 1u : 0u) << 33) | ((unsigned long long)nwr << 23) |
         (lw_idx >= 0 ? (1ull << 22) | ((unsigned long long)(uint32_t)lw_idx << 2) : 0ull) | (ll_acq >= 0 ? 2ull | (unsigned long long)(ll_acq & 1) : 0ull);
}
struct kvh_lds {
  uint32_t key[KVB_T];            // idx << 9 | slot of the record: sorted = the piece in request order
  uint32_t idx[KVB_T];            // request index at each sorted position
  uint8_t typ[KVB_T];             // request type by slot
  uint64_t rem[KVB_T];            // the other keys' records (a remainder / solo item), then their sort words
  uint64_t all[KVB_T];            // a solo item: every record of the sub (the candidate key may be no hot key at all)
  uint16_t cs[KVB_T + 2];         // chunks of the sorted remainder: first record of each, then the end
  uint64_t Bw[KVB_W], Bl[KVB_W], Ba[KVB_W];  // per wave of sorted positions: writers, lock ops, ACQUIREs; group heads of the remainder
  unsigned long long pub[KVR_NPMAX + 1];     // the siblings' words
  uint32_t bad, timeout, nhot, nrem, nall, nsame, nbr, nch, too_big;
  uint32_t found, link, slot, ver0, la0, table;
  uint32_t rowv[10];              // the row's value before the pass
  // a solo item: the row machine of a key whose requests insert / delete the row (state after each row-changing request)
  uint32_t best;                  // longest run of one (bucket group, key hash): length << 16 | first sorted position
  unsigned long long hkey;
  uint32_t ev_ver[KVB_T];
  int16_t ev_src[KVB_T];
  uint8_t ev_ex[KVB_T];
  uint32_t fin_ex, fin_ver, fin_multi, fin_miss, dupf;
  int fin_src;
  // group phases (kv_group_phases): a bucket group request by request where it matters
  uint8_t qq[KVB_T];              // lock quadrant by slot
  uint64_t Lq[4][KVB_W], Aq[4][KVB_W];  // per quadrant and wave of idx-sorted positions: lock ops, ACQUIREs
  uint32_t lockw, bestg, gbad;
};
// m <= KVB_T records of OTHER keys in R[] (the big path's 8-byte form), whole workgroup: sorted by (bucket group, key hash,
// idx), cut into chunks of <= 64 at bucket-group boundaries, every chunk resolved by a wave as k_kv_resolve would (kv_chunk:
// all closed forms, rounds where they do not apply) -- 4 memory round trips instead of kv_big_bin's stretch machinery (30 .. 40
// us for a hundred records, r05).  false (nothing touched): a bucket group of more than 64 records -- kv_big_bin's job.
template <int WL>
__device__ __forceinline__ static bool kv_rem_chunks(uint8_t *rep, const kv_cut cut2, const kv_dev *kv, uint32_t bin, uint64_t *R, uint32_t m,
                                                     kvh_lds &H, dint_dev_stats *__restrict__ stats, int force_rounds, const dint_view V,
                                                     bool sorted = false /* R[] holds sort words in order already */) {
  using F = Fmt<WL>;
  const uint32_t t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const uint32_t sh_g = 16 + cut2.ibits, sh_k = 7 + cut2.ibits, idx_mask = (uint32_t)((1ull << cut2.ibits) - 1ull);
  __syncthreads();
  if (!sorted) {
    const uint64_t mine = t < m ? kv_sort_key(R[t], cut2) : ~0ull;
    __syncthreads();
    R[t] = mine;
    __syncthreads();
    uint32_t N = 64;
    while (N < m) N <<= 1;
    kvb_sort_blocked<1>(R, N);  // (slots N .. KVB_T - 1 hold ~0: they stay last)
  }
  const uint64_t w = R[t];
  const bool head = t < m && (t == 0 || (R[t - 1] >> sh_g) != (w >> sh_g));
  const uint64_t hm = __ballot(head);
  if (lane == 0) H.Bw[wave] = hm;
  __syncthreads();
  if (t == 0) {  // greedy chunks: cut at a group's first record whenever the next group no longer fits
    uint32_t nch = 0, start = 0, prev = 0, big = 0;
    for (uint32_t wv = 0; wv < KVB_W; wv++)
      for (uint64_t mm = H.Bw[wv]; mm; mm &= mm - 1) {
        const uint32_t hp = wv * 64 + (uint32_t)__ffsll((unsigned long long)mm) - 1;
        if (hp == 0) continue;
        if (hp - start > 64) { H.cs[nch++] = (uint16_t)start; start = prev; big |= hp - start > 64; }
        prev = hp;
      }
    if (m - start > 64) { H.cs[nch++] = (uint16_t)start; start = prev; big |= m - start > 64; }
    H.cs[nch++] = (uint16_t)start;
    H.cs[nch] = (uint16_t)m;
    H.nch = nch; H.too_big = big;
  }
  __syncthreads();
  if (H.too_big) return false;
  const uint32_t nch = H.nch;
  for (uint32_t ch = wave; ch < nch; ch += KVB_W) {
    const uint32_t a = H.cs[ch], c = H.cs[ch + 1] - a;
    const bool valid = lane < c;
    const uint64_t wv = valid ? R[a + lane] : ~0ull;
    const uint32_t idx = valid ? (uint32_t)(wv >> 7) & idx_mask : 0, gk = kv_cut_gk((uint32_t)(wv >> sh_g), bin, cut2);
    const uint32_t pay = (uint32_t)wv & 0x7Fu;
    const uint64_t key = valid ? ld_u64(rep + dint_view_off(V, idx, F::MSG) + F::KEY) : 0;
    kv_chunk<WL>(rep, valid, idx, gk, (uint32_t)(wv >> sh_k) & 511u, pay_type(pay), valid ? kv_table_of(kv, gk) : 0, pay_q(pay), key, kv, stats,
                 force_rounds, ch + KVB_W >= nch, V);
  }
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
  __syncthreads();
  return true;
}
// ---- one BUCKET GROUP request by request where it matters, whole workgroup (kv_solo_item) ---------------------------------
// For a bucket the closed forms do not cover -- a duplicate row in the table (the reference's population makes them), two keys on
// one lock byte, a neighbour that restructures the chain, two keys behind one key hash, an INSERT of an existing row -- and that
// holds more requests than a chunk.  What is serial about a bucket is little: its ROW-CHANGING requests (SET / INSERT / DELETE,
// a handful per pass), and per lock byte which lock op came last.  So: the group's requests in request order (one sort of <= 512
// words); the lock bytes in closed form per quadrant, across keys (ACQUIRE sees what the last lock op on its byte left); the
// rows in PHASES -- the READs behind e row-changing requests all at once, each with the reference's own walk of the chain
// (kv_apply: first match, duplicates and all), then the e-th row-changing request applied to the chain by its own thread.  A
// phase is two memory round trips; kv_big_bin's stretch machinery with its rounds took 36 us for such a group of 116 (r05).
// R[Gs, Ge) = the group's sort words; false (nothing touched): more than KVG_MAXW row-changing requests, an INSERT with the
// overflow pool nearly empty (it may be refused, and then its lock byte stays: kv_do_request), or a request type the servers
// do not know -- kv_big_bin's.
#define KVG_MAXW 24u
template <int WL>
__device__ __forceinline__ static bool kv_group_phases(uint8_t *rep, const kv_cut cut2, const kv_dev *kv, uint32_t bin, const uint64_t *R, uint32_t Gs,
                                                       uint32_t Ge, kvh_lds &H, dint_dev_stats *__restrict__ stats, const dint_view V) {
  using F = Fmt<WL>;
  const uint32_t t = threadIdx.x, lane = t & 63, wave = t >> 6, n = Ge - Gs;
  const uint32_t sh_g = 16 + cut2.ibits, idx_mask = (uint32_t)((1ull << cut2.ibits) - 1ull);
  auto is_set = [](uint32_t type) -> bool { return WL == DINT_WL_STORE ? type == 1 : (type == 12 || type == 13); };
  auto is_ins = [](uint32_t type) -> bool { return WL == DINT_WL_STORE ? type == 2 : (type == 18 || type == 19); };
  auto is_del = [](uint32_t type) -> bool { return WL == DINT_WL_TATP && (type == 22 || type == 23); };
  const uint32_t gk = kv_cut_gk((uint32_t)(R[Gs] >> sh_g), bin, cut2), table = kv_table_of(kv, gk);
  const kv_tab tb = kv->tab[table];
  const uint64_t bucket = (uint64_t)(gk - kv->gk_base[table]);
  uint8_t *ie = kv_entry_ptr(tb, bucket, KV_INLINE);
  __syncthreads();
  {
    const uint64_t w = t < n ? R[Gs + t] : 0;
    const uint32_t idx = (uint32_t)(w >> 7) & idx_mask, pay = (uint32_t)w & 0x7Fu;
    H.key[t] = t < n ? (idx << 9) | t : 0xFFFFFFFFu;
    H.typ[t] = (uint8_t)(pay_type(pay) & 0xFFu);
    H.qq[t] = (uint8_t)pay_q(pay);
    H.rem[t] = t < n ? ld_u64(rep + dint_view_off(V, idx, F::MSG) + F::KEY) : 0;  // every request with its own full key
    if (t == 0) { H.gbad = 0; H.lockw = KV_LD(uint32_t, ie + KV_LOCKB_OFF); }
  }
  __syncthreads();
  kvb_sort_blocked_u32<1>(H.key, KVB_T);
  const uint32_t sk = H.key[t];
  const bool v = sk != 0xFFFFFFFFu;
  const uint32_t slot = sk & 511u, my_idx = sk >> 9, my_type = v ? H.typ[slot] : 0xFFu, my_q = v ? H.qq[slot] : 0u;
  const uint64_t my_key = v ? H.rem[slot] : 0;
  const bool isW = v && (is_set(my_type) || is_ins(my_type) || is_del(my_type)), isR = v && my_type == 0;
  const bool lk = v && kv_lock_op<WL>(my_type), aq = v && WL == DINT_WL_TATP && my_type == 1;
  if (v && !(kv_simple_op<WL>(my_type) || kv_struct_op<WL>(my_type))) H.gbad = 1;
  if (v && is_ins(my_type) && kv_pool_low(tb)) H.gbad = 1;
  const uint64_t bw = __ballot(isW);
  if (lane == 0) H.Bw[wave] = bw;
  if (WL == DINT_WL_TATP) {
#pragma unroll
    for (uint32_t q = 0; q < 4; q++) {
      const uint64_t ml = __ballot(lk && my_q == q), ma = __ballot(aq && my_q == q);
      if (lane == 0) { H.Lq[q][wave] = ml; H.Aq[q][wave] = ma; }
    }
  }
  __syncthreads();
  uint32_t ev_below = 0, nW = 0;
#pragma unroll
  for (uint32_t wv = 0; wv < KVB_W; wv++) {
    const uint64_t m = H.Bw[wv];
    nW += (uint32_t)__popcll(m);
    ev_below += (uint32_t)__popcll(wv < wave ? m : (wv == wave ? m & lanemask_lt() : 0ull));
  }
  if (H.gbad || nW > KVG_MAXW) return false;
  uint8_t *msg = rep + dint_view_off(V, my_idx, F::MSG);
  for (uint32_t e = 0; e <= nW; e++) {
    if (isR && ev_below == e) {
      kv_hdr Hh;
      kv_hdr_load(Hh, ie);
      const kv_res r = kv_apply<kv_dev_mem>(tb, bucket, Hh, KV_ACT_GET, my_key, msg + F::VAL, 0, blockIdx.x);
      if (r.ok) st_u32(msg + F::VER, r.ver);
      msg[F::TYPE] = (uint8_t)(WL == DINT_WL_STORE ? (r.ok ? 3 : 7) : (r.ok ? 4 : 6));
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __syncthreads();
    if (isW && ev_below == e) {  // (one thread: the e-th row-changing request of the group)
      kv_hdr Hh;
      kv_hdr_load(Hh, ie);
      const uint32_t act = is_set(my_type) ? KV_ACT_SET : is_ins(my_type) ? KV_ACT_INS : KV_ACT_DEL;
      const kv_res r = kv_apply<kv_dev_mem>(tb, bucket, Hh, act, my_key, msg + F::VAL, 0, blockIdx.x);
      uint32_t code;
      if (WL == DINT_WL_STORE) code = my_type == 1 ? (r.ok ? 5 : 7) : 8;
      else code = my_type == 12 ? 15 : my_type == 13 ? 16 : my_type == 18 ? 20 : my_type == 19 ? 21 : my_type == 22 ? 25 : 26;
      if (!r.ok) {
        if (act == KV_ACT_INS) atomicAdd(&stats->pool_exhausted, 1ULL);
        else if (WL != DINT_WL_STORE) atomicAdd(&stats->missing_keys, 1ULL);
      }
      msg[F::TYPE] = (uint8_t)code;
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __syncthreads();
  }
  if (WL == DINT_WL_TATP) {
    // the lock bytes, per quadrant across the group's keys: a last-writer-wins register (ACQUIRE leaves 1, every other lock op 0)
    const uint32_t lockw = H.lockw;
    if (v && (my_type == 1 || my_type == 2)) {
      int last = -1;
#pragma unroll
      for (uint32_t wv = 0; wv < KVB_W; wv++) {
        const uint64_t m = H.Lq[my_q][wv], mb = wv < wave ? m : (wv == wave ? m & lanemask_lt() : 0ull);
        if (mb) last = (int)(wv * 64 + 63 - __clzll((long long)mb));
      }
      const uint32_t seen = last >= 0 ? (uint32_t)((H.Aq[my_q][last >> 6] >> (last & 63)) & 1ull) : (lockw >> (8 * my_q)) & 0xFFu;
      msg[F::TYPE] = (uint8_t)(my_type == 1 ? (seen ? 8 : 7) : 9);
    }
    if (t < 4) {
      int last = -1;
      for (uint32_t wv = 0; wv < KVB_W; wv++)
        if (H.Lq[t][wv]) last = (int)(wv * 64 + 63 - __clzll((long long)H.Lq[t][wv]));
      if (last >= 0) {
        const uint32_t fin = (uint32_t)((H.Aq[t][last >> 6] >> (last & 63)) & 1ull);
        if (fin != ((lockw >> (8 * t)) & 0xFFu)) KV_ST(uint8_t, ie + KV_LOCKB_OFF + t, (uint8_t)fin);
      }
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
  __syncthreads();
  return true;
}
// ---- a SOLO item: a big sub of at most split_target (<= KVB_T) records, one workgroup, nobody to wait for -------------------
// The sub in LDS, sorted once by (bucket group, key hash, idx): what the chunk path needs anyway.  No run of one key longer
// than 64: chunks (kv_rem_chunks) and done -- the sub was a handful of keys, or a crowded coarse bin's all-big fallback.
// Else the LONGEST run is the hot key (found here, not guessed: its requests are contiguous and already in request order):
// answered in closed form by one thread per request -- the forms of the pieces with nothing before them, plus the ROW MACHINE
// of kv_chunk for a key whose requests insert / delete its row (tatp's hot CALL_FORWARDING rows: ~100 READs, a dozen ACQUIREs
// and one INSERT or DELETE per pass; they used to fall back to kv_big_bin's rounds: 36 .. 47 us, r05): one thread walks the
// row-changing requests {SET, INSERT, DELETE} in order and leaves {exists, version, value source} after each in LDS, every
// request reads the state behind the last one before it, and the net effect reaches the chain once (kv_chunk step 4b).
// The rest of the sub goes through the chunks afterwards.  Not in closed form (nothing touched): another key of the hot bucket
// that restructures the chain or shares the lock byte, a second key behind the hash bits, an INSERT of an existing row
// (duplicate rows), a duplicate row in the table, a nearly empty overflow pool -- kv_big_bin over the whole sub.
template <int WL>
__device__ __forceinline__ static int kv_solo_item(uint8_t *rep, const kv_cut cut2, const kv_dev *kv, const uint4 d,
                                                   const uint64_t *__restrict__ ovf, uint64_t *__restrict__ ovf2,
                                                   dint_dev_stats *__restrict__ stats, int force_rounds, const dint_view V, uint8_t *lds_raw,
                                                   uint32_t *src, uint32_t *off, uint32_t *cnt, uint64_t *ttr,
                                                   bool preloaded = false /* k_kv_late: the d.z sort words are in H.all already (any order,
                                                                             ~0 behind them), ovf2 is nullptr: a remainder stays in H.rem */) {
  using F = Fmt<WL>;
  kvh_lds &H = *(kvh_lds *)lds_raw;
  const uint32_t t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const uint32_t h = d.z;
  const uint32_t sh_g = 16 + cut2.ibits, sh_k = 7 + cut2.ibits, idx_mask = (uint32_t)((1ull << cut2.ibits) - 1ull);
  *src = 0; *off = d.y; *cnt = h;  // (the way out when something is not in closed form)
  if (h > KVB_T) return 1;
  auto is_set = [](uint32_t type) -> bool { return WL == DINT_WL_STORE ? type == 1 : (type == 12 || type == 13); };
  auto is_ins = [](uint32_t type) -> bool { return WL == DINT_WL_STORE ? type == 2 : (type == 18 || type == 19); };
  auto is_del = [](uint32_t type) -> bool { return WL == DINT_WL_TATP && (type == 22 || type == 23); };
  __syncthreads();  // the LDS buffer is free
  if (!preloaded) H.all[t] = t < h ? kv_sort_key(kv_ld_agent(&ovf[d.y + t]), cut2) : ~0ull;
  if (t == 0) { H.bad = 0; H.best = 0; H.bestg = 0; H.dupf = 0; }
  __syncthreads();
  {
    uint32_t N = 64;
    while (N < h) N <<= 1;
    kvb_sort_blocked<1>(H.all, N);
  }
  const uint64_t w = H.all[t];
  const bool valid = t < h;
  const uint64_t prevw = t ? H.all[t - 1] : 0;
  const bool head_k = valid && (t == 0 || (prevw >> sh_k) != (w >> sh_k));  // (bucket group, key hash) changes
  const bool head_g = valid && (t == 0 || (prevw >> sh_g) != (w >> sh_g));  // the bucket group changes
  const uint64_t mk = __ballot(head_k), mg = __ballot(head_g);
  if (lane == 0) { H.Bw[wave] = mk; H.Bl[wave] = mg; }
  __syncthreads();
  auto next_head = [&](const uint64_t *M) -> uint32_t {  // the first head above me, or h
    uint32_t nxt = h;
    for (uint32_t wv = wave; wv < KVB_W && nxt == h; wv++) {
      uint64_t mm = M[wv];
      if (wv == wave) mm &= ~((2ull << lane) - 1ull);
      if (mm) nxt = wv * 64 + (uint32_t)__ffsll((unsigned long long)mm) - 1;
    }
    return nxt;
  };
  if (head_k) atomicMax(&H.best, ((next_head(H.Bw) - t) << 16) | t);   // the longest run of one key ...
  if (head_g) atomicMax(&H.bestg, ((next_head(H.Bl) - t) << 16) | t);  // ... of one bucket group
  __syncthreads();
  const uint32_t L = H.best >> 16, P = H.best & 0xFFFFu;
  // the bucket group of the longest key run (a hot key), else the longest group
  uint32_t Gs = H.bestg & 0xFFFFu, Ge = Gs + (H.bestg >> 16);
  if (L > 64) {
    Gs = 0;
    for (uint32_t wv = 0; wv < KVB_W; wv++) {  // the last group head at or below P
      uint64_t mm = H.Bl[wv];
      if (wv * 64 > P) break;
      if (wv == (P >> 6)) mm &= (2ull << (P & 63)) - 1ull;
      if (mm) Gs = wv * 64 + 63 - (uint32_t)__clzll((long long)mm);
    }
    Ge = h;
    for (uint32_t wv = P >> 6; wv < KVB_W && Ge == h; wv++) {  // the first one above it
      uint64_t mm = H.Bl[wv];
      if (wv == (P >> 6)) mm &= ~((2ull << (P & 63)) - 1ull);
      if (mm) Ge = wv * 64 + (uint32_t)__ffsll((unsigned long long)mm) - 1;
    }
  }
  // the records outside [lo, lo + len), still sorted, through the chunks (kv_big_bin for them when a second bucket group is too long)
  auto rest = [&](uint32_t lo, uint32_t len) -> int {
    const uint32_t c_rem = h - len;
    if (c_rem == 0) return 0;
    __syncthreads();
    if (valid && (t < lo || t >= lo + len)) H.rem[t < lo ? t : t - len] = w;
    if (t >= c_rem) H.rem[t] = ~0ull;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");  // the hot bucket's write-backs above, then (perhaps) its other rows
    if (kv_rem_chunks<WL>(rep, cut2, kv, d.x, H.rem, c_rem, H, stats, force_rounds, V, true)) return 0;
    if (ovf2 && t < c_rem) {  // back into the big path's record form, for kv_big_bin
      const uint64_t sw = H.rem[t];
      ovf2[d.y + t] = ((sw >> sh_g) << sh_g) | ((uint64_t)((uint32_t)(sw >> 7) & idx_mask) << 16) | ((uint32_t)sw & 0x7Fu) | (((uint32_t)(sw >> sh_k) & 511u) << 7);
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    *src = 1; *off = d.y; *cnt = c_rem;
    return 1;
  };
  // a bucket group the closed forms do not cover: request by request where it matters; the rest through the chunks
  auto phases = [&]() -> int {
    if (!kv_group_phases<WL>(rep, cut2, kv, d.x, H.all, Gs, Ge, H, stats, V)) return 1;  // (nothing touched: kv_big_bin, the whole sub)
    return rest(Gs, Ge - Gs);
  };
  if (L <= 64) {  // no hot key here: chunks -- unless one bucket group is too long for a chunk
    if (ttr && t == 0) ttr[30] = (uint64_t)(d.w & 0xFFFFu) | (1ull << 17) | ((uint64_t)L << 20) | ((uint64_t)h << 32);
    if (Ge - Gs <= 64) return kv_rem_chunks<WL>(rep, cut2, kv, d.x, H.all, h, H, stats, force_rounds, V, true) ? 0 : 1;
    return phases();
  }
  // ---- the hot run [P, P + L)
  const uint64_t hw = H.all[P];
  const uint32_t hq = pay_q((uint32_t)hw & 0x7Fu);
  const bool v = valid && t >= P && t < P + L;
  const uint32_t my_idx = (uint32_t)(w >> 7) & idx_mask, my_type = valid ? pay_type((uint32_t)w & 0x7Fu) : 0xFFu;
  // another key of the hot bucket must not restructure the chain or use the hot key's lock byte
  if (valid && !v && (w >> sh_g) == (hw >> sh_g) && (kv_struct_op<WL>(my_type) || (kv_lock_op<WL>(my_type) && pay_q((uint32_t)w & 0x7Fu) == hq))) atomicOr(&H.bad, kv_struct_op<WL>(my_type) ? 1u : 2u);
  const uint64_t my_key = v ? ld_u64(rep + dint_view_off(V, my_idx, F::MSG) + F::KEY) : 0;
  if (t == P) H.hkey = my_key;
  const bool setop = v && is_set(my_type), insop = v && is_ins(my_type), delop = v && is_del(my_type);
  const bool lk = v && kv_lock_op<WL>(my_type), aq = v && WL == DINT_WL_TATP && my_type == 1;
  if (v && !(kv_simple_op<WL>(my_type) || kv_struct_op<WL>(my_type))) atomicOr(&H.bad, 4u);
  const uint64_t bev = __ballot(setop || insop || delop), bl = __ballot(lk), ba = __ballot(aq), bst = __ballot(insop || delop);
  if (lane == 0) { H.Bw[wave] = bev; H.Bl[wave] = bl; H.Ba[wave] = ba; }
  if (bst && lane == 0) H.dupf = 1;  // (reused below as "the run inserts / deletes"; the real duplicate check is the locator's)
  H.idx[t] = my_idx;
  H.typ[t] = (uint8_t)(my_type & 0xFFu);
  __syncthreads();
  const uint64_t hkey = H.hkey;
  if (v && my_key != hkey) atomicOr(&H.bad, 8u);  // 9 hash bits can collide
  const bool structural = H.dupf != 0;
  __syncthreads();
  // ---- thread P: the row (header, location, value), then the row machine over the run's row-changing requests, in order
  kv_hdr Hd;
  kv_tab tb;
  uint64_t bucket = 0;
  uint32_t table = 0;
  {
    const uint32_t gk = kv_cut_gk((uint32_t)(hw >> sh_g), d.x, cut2);
    table = kv_table_of(kv, gk);
    tb = kv->tab[table];
    bucket = (uint64_t)(gk - kv->gk_base[table]);
  }
  if (t == P) {
    kv_hdr_load(Hd, kv_entry_ptr(tb, bucket, KV_INLINE));
    const kv_where wh = kv_locate(tb, bucket, Hd, hkey);
    H.found = wh.found; H.link = wh.link; H.slot = wh.slot; H.ver0 = wh.ver;
    H.la0 = WL == DINT_WL_TATP ? (Hd.lockw >> (8 * hq)) & 0xFFu : 0u;
    if (wh.found) {
      const uint8_t *rv = kv_entry_ptr(tb, bucket, wh.link) + KV_VAL_OFF + wh.slot * F::VS;
#pragma unroll
      for (uint32_t k = 0; k < F::VS / 4; k++) H.rowv[k] = KV_LD(uint32_t, rv + 4 * k);
    }
    uint32_t ex = wh.found, ver = wh.ver, toggles = 0, miss = 0, e = 0;
    int sr = -1;
    const uint32_t plow = structural ? kv_pool_low(tb) : 0u;
    uint32_t bail = plow | ((structural && kv_has_dup(tb, bucket, Hd, hkey, wh)) ? 1u : 0u);
    for (uint32_t wv = 0; wv < KVB_W; wv++)
      for (uint64_t mm = H.Bw[wv]; mm; mm &= mm - 1) {
        const uint32_t p = wv * 64 + (uint32_t)__ffsll((unsigned long long)mm) - 1;
        const uint32_t ty = H.typ[p];
        if (!bail) {
          if (is_set(ty)) { if (ex) { ver++; sr = (int)p; } else if (WL != DINT_WL_STORE) miss++; }
          else if (is_ins(ty)) { if (ex) bail = 1; else { ex = 1; ver = 0; sr = (int)p; toggles++; } }
          else { if (ex) { ex = 0; toggles++; } else miss++; }
          H.ev_ex[e] = (uint8_t)ex; H.ev_ver[e] = ver; H.ev_src[e] = (int16_t)sr;
        }
        e++;
      }
    // (no closed form for the row -- a duplicate row in the table, which the reference's population makes; an INSERT of an
    // existing row; the pool nearly empty --: the bucket group goes through kv_group_phases)
    if (bail) atomicOr(&H.bad, 16u);
    H.fin_ex = ex; H.fin_ver = ver; H.fin_src = sr; H.fin_multi = toggles > 1; H.fin_miss = miss;
  }
  __syncthreads();
  if (ttr && t == 0) ttr[30] = (uint64_t)(d.w & 0xFFFFu) | ((uint64_t)(H.bad ? 0u : 1u) << 16) | ((uint64_t)(structural ? 1u : 0u) << 18) | ((uint64_t)(L & 0xFFFu) << 20) | ((uint64_t)(H.bad & 31u) << 56) | ((uint64_t)(h - L) << 32);
  if (H.bad) return phases();  // (nothing has been touched)
  const uint32_t found = H.found, ver0 = H.ver0, la0 = H.la0;
  // ---- every request of the run: the state behind the last row-changing request before it, the last lock op before it
  uint32_t ev_below = 0;
  int ll_below = -1, ll_tot = -1;
#pragma unroll
  for (uint32_t wv = 0; wv < KVB_W; wv++) {
    const uint64_t me = H.Bw[wv], ml = H.Bl[wv];
    const uint64_t meb = wv < wave ? me : (wv == wave ? me & lanemask_lt() : 0ull), mlb = wv < wave ? ml : (wv == wave ? ml & lanemask_lt() : 0ull);
    ev_below += (uint32_t)__popcll(meb);
    if (mlb) ll_below = (int)(wv * 64 + 63 - __clzll((long long)mlb));
    if (ml) ll_tot = (int)(wv * 64 + 63 - __clzll((long long)ml));
  }
  auto acq_at = [&](int p) -> int { return (int)((H.Ba[p >> 6] >> (p & 63)) & 1ull); };
  if (v) {
    const uint32_t ex_b = ev_below ? H.ev_ex[ev_below - 1] : found, ver_b = ev_below ? H.ev_ver[ev_below - 1] : ver0;
    const int src_b = ev_below ? (int)H.ev_src[ev_below - 1] : -1;
    uint8_t *msg = rep + dint_view_off(V, my_idx, F::MSG);
    uint32_t code;
    bool get = false;
    if (WL == DINT_WL_STORE) {
      code = my_type == 0 ? (ex_b ? 3 : 7) : my_type == 1 ? (ex_b ? 5 : 7) : 8;
      get = my_type == 0 && ex_b;
    } else {
      const int seen = ll_below >= 0 ? acq_at(ll_below) : (int)la0;
      switch (my_type) {
        case 0: code = ex_b ? 4 : 6; get = ex_b != 0; break;
        case 1: code = seen ? 8 : 7; break;
        case 2: code = 9; break;
        case 12: code = 15; break;
        case 13: code = 16; break;
        case 18: code = 20; break;
        case 19: code = 21; break;
        case 22: code = 25; break;
        default: code = 26; break;  // 23 kDeleteBck
      }
    }
    if (get) {
      if (src_b >= 0) {
        kv_copy_words(msg + F::VAL, rep + dint_view_off(V, H.idx[src_b], F::MSG) + F::VAL, F::VS);
      } else {
#pragma unroll
        for (uint32_t k = 0; k < F::VS / 4; k++) st_u32(msg + F::VAL + 4 * k, H.rowv[k]);
      }
      st_u32(msg + F::VER, ver_b);
    }
    msg[F::TYPE] = (uint8_t)code;
  }
  // ---- the run's net effect on the table, once (kv_chunk steps 4a / 4b), by thread P, which holds the header
  if (t == P) {
    const uint32_t exists1 = H.fin_ex, fin_ver = H.fin_ver;
    const int fin_src = H.fin_src;
    const bool redo = found && exists1 && H.fin_multi;  // deleted and inserted again: the row may move
    uint8_t *ie = kv_entry_ptr(tb, bucket, KV_INLINE);
    const uint8_t *fval = fin_src >= 0 ? rep + dint_view_off(V, H.idx[fin_src], F::MSG) + F::VAL : nullptr;
    if (found && exists1 && !redo) {
      if (fin_src >= 0) {
        kv_copy_words(kv_entry_ptr(tb, bucket, H.link) + KV_VAL_OFF + H.slot * F::VS, fval, F::VS);
        KV_ST(uint32_t, &kv_entry_hdr(tb, bucket, H.link)->ver[H.slot], fin_ver);
      }
    } else if (found != exists1 || redo) {
      if (redo) {
        kv_apply<kv_dev_mem>(tb, bucket, Hd, KV_ACT_DEL, hkey, nullptr, 0, blockIdx.x);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        kv_hdr_load(Hd, ie);
      }
      const kv_res r = kv_apply<kv_dev_mem>(tb, bucket, Hd, exists1 ? KV_ACT_INS : KV_ACT_DEL, hkey, (uint8_t *)fval, fin_ver, blockIdx.x);
      if (exists1 && !r.ok) atomicAdd(&stats->pool_exhausted, 1ULL);
    }
    if (WL == DINT_WL_TATP && ll_tot >= 0) {
      const uint32_t la_fin = (uint32_t)acq_at(ll_tot);
      if (la_fin != la0) KV_ST(uint8_t, ie + KV_LOCKB_OFF + hq, (uint8_t)la_fin);
    }
    if (H.fin_miss) atomicAdd(&stats->missing_keys, (unsigned long long)H.fin_miss);
  }
  // ---- the rest of the sub (sorted already): chunks
  return rest(P, L);
}
// returns 0: the item is done (or not this workgroup's to do); 1: run kv_big_bin over recs[*src][*off, *off + *cnt), src 0 = ovf, 1 = ovf2
template <int WL>
__device__ __forceinline__ static int kv_hot_item(uint8_t *rep, const kv_cut cut2, const kv_dev *kv, const uint4 d, const uint4 x, const uint4 y,
                                                  const uint64_t *__restrict__ ovf, uint64_t *__restrict__ ovf2, unsigned long long *hotpub,
                                                  uint32_t seq, uint32_t inv_n, dint_dev_stats *__restrict__ stats, int force_rounds, const dint_view V,
                                                  uint8_t *lds_raw, uint32_t *src, uint32_t *off, uint32_t *cnt, uint64_t *ttr) {
  using F = Fmt<WL>;
  kvh_lds &H = *(kvh_lds *)lds_raw;
  const uint32_t t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const uint32_t kind = d.w & 3u, j = (d.w >> 2) & 255u, np = (d.w >> 10) & 255u;
  const uint32_t first = x.z, h = d.z;
  const uint32_t sh_g = 16 + cut2.ibits, idx_mask = (uint32_t)((1ull << cut2.ibits) - 1ull);
  const uint64_t hkey = ((uint64_t)x.y << 32) | x.x, hrec = ((uint64_t)y.y << 32) | y.x;
  const uint32_t hq = pay_q(kv_rec_pay(hrec)), hkh = pay_kh(kv_rec_pay(hrec));
  const bool solo = kind == KVQ_SOLO;
  bool do_piece = kind != KVQ_REM;
  const bool do_rem = kind != KVQ_PIECE;
  unsigned long long *pub = hotpub + first;
  auto is_writer = [](uint32_t type) -> bool { return WL == DINT_WL_STORE ? type == 1 : (type == 12 || type == 13); };
  __syncthreads();  // the LDS buffer is free (the previous item is done with it)
  H.key[t] = 0xFFFFFFFFu;
  if (t == 0) { H.bad = 0; H.timeout = 0; H.nhot = 0; H.nrem = 0; H.nall = 0; H.nsame = 0; H.nbr = 0; }
  __syncthreads();

  // ---- one pass over the sub's records.  The key's requests of my index range are mine (a piece); every record of another
  // (bucket group, key hash) is the remainder's.  Every item of the sub sees every record, so all of them know alike whether
  // the hot BUCKET holds requests for another key (`nbr`): only then does the remainder take part in the all-or-nothing vote.
  for (uint32_t k0 = 0; k0 < h; k0 += KVB_T) {
    const uint32_t k = k0 + t;
    const uint64_t r = k < h ? kv_ld_agent(&ovf[d.y + k]) : 0;
    const uint32_t pay = kv_rec_pay(r), ridx = (uint32_t)(r >> 16) & idx_mask, type = pay_type(pay);
    const bool same_g = k < h && (r >> sh_g) == (hrec >> sh_g), same = same_g && pay_kh(pay) == hkh;
    if (same_g && !same) {
      H.nbr = 1;
      // another key of the hot bucket must not restructure the chain or use the hot key's lock byte
      if (kv_struct_op<WL>(type) || (kv_lock_op<WL>(type) && pay_q(pay) == hq)) H.bad = 1;
    }
    if (do_piece) {
      const bool mine = same && (np == 1 || kv_piece_of(ridx, np, inv_n) == j);
      // behind the candidate's hash bits there may be another key (9 bits): the closed form is not for it
      const bool really = mine && ld_u64(rep + dint_view_off(V, ridx, F::MSG) + F::KEY) == hkey;
      if (mine && !really) H.bad = 1;
      const uint64_t mm = __ballot(really), ms = __ballot(same);
      uint32_t at = 0;
      if (lane == 0 && mm) at = atomicAdd(&H.nhot, (uint32_t)__popcll(mm));
      if (lane == 0 && ms) atomicAdd(&H.nsame, (uint32_t)__popcll(ms));
      at = (uint32_t)__builtin_amdgcn_readfirstlane((int)at) + (uint32_t)__popcll(mm & lanemask_lt());
      if (really && at < KVB_T) { H.key[at] = (ridx << 9) | at; H.typ[at] = (uint8_t)(type & 0xFFu); }
    }
    if (do_rem) {
      const bool other = k < h && !same;
      const uint64_t mm = __ballot(other);
      uint32_t at = 0;
      if (lane == 0 && mm) at = atomicAdd(&H.nrem, (uint32_t)__popcll(mm));
      at = (uint32_t)__builtin_amdgcn_readfirstlane((int)at) + (uint32_t)__popcll(mm & lanemask_lt());
      if (other) {  // compacted: in LDS for the chunk path, in ovf2 for kv_big_bin (ovf itself stays whole: the old way needs it)
        ovf2[d.y + at] = r;
        if (at < KVB_T) H.rem[at] = r;
      }
    }
    if (solo && k < h && k < KVB_T) H.all[k] = r;  // (a solo sub has at most split_target <= KVB_T records)
  }
  __syncthreads();
  const uint32_t c_rem = do_rem ? H.nrem : 0u, nbr = H.nbr;
  if (solo && H.nhot <= 64 && h <= KVB_T) {
    // the candidate key is no hot key (a sub of several keys, or a crowded coarse bin's all-big fallback): no bucket group of
    // more than 64 requests in sight -- the whole sub through the chunk path, nothing through the closed form
    if (ttr && t == 0) ttr[30] = (uint64_t)(d.w & 0xFFFFu) | (1ull << 17) | ((uint64_t)H.nhot << 20) | ((uint64_t)h << 32);
    if (kv_rem_chunks<WL>(rep, cut2, kv, d.x, H.all, h, H, stats, force_rounds, V)) return 0;
    *src = 0; *off = d.y; *cnt = h;
    return 1;
  }
  const uint32_t c = do_piece ? H.nhot : 0u;
  if (t == 0 && c > KVB_T) H.bad = 1;  // more of the key's requests in this range than a workgroup has threads
  const uint32_t nsib = np > 1 ? np + (nbr ? 1u : 0u) : 1u;
  const bool voting = do_piece || (nbr && np > 1);  // a remainder beside a bucket the hot key has to itself is nobody's business
  // ---- the bucket's header now (the round trip overlaps the sort), the row behind the sort
  kv_hdr Hd;
  kv_tab tb0;
  uint64_t bucket0 = 0;
  uint32_t table0 = 0;
  const bool locator = t == 0 && c != 0;  // (an empty piece answers nothing and stores nothing: it only says so)
  if (locator) {
    const uint32_t gk = kv_cut_gk((uint32_t)(hrec >> sh_g), d.x, cut2);
    table0 = kv_table_of(kv, gk);
    tb0 = kv->tab[table0];
    bucket0 = (uint64_t)(gk - kv->gk_base[table0]);
    kv_hdr_load(Hd, kv_entry_ptr(tb0, bucket0, KV_INLINE));
  }
  __syncthreads();
  if (do_piece) kvb_sort_blocked_u32<1>(H.key, KVB_T);
  if (locator) {
    const kv_where w = kv_locate(tb0, bucket0, Hd, hkey);
    H.found = w.found; H.link = w.link; H.slot = w.slot; H.ver0 = w.ver; H.table = table0;
    H.la0 = WL == DINT_WL_TATP ? (Hd.lockw >> (8 * hq)) & 0xFFu : 0u;
    if (w.found) {
      const uint8_t *rv = kv_entry_ptr(tb0, bucket0, w.link) + KV_VAL_OFF + w.slot * F::VS;
#pragma unroll
      for (uint32_t k = 0; k < F::VS / 4; k++) H.rowv[k] = KV_LD(uint32_t, rv + 4 * k);
    }
  }
  const uint32_t sk = H.key[t];
  const bool v = do_piece && sk != 0xFFFFFFFFu;
  const uint32_t my_idx = sk >> 9, my_type = v ? H.typ[sk & 511u] : 0xFFu;
  if (v && !kv_simple_op<WL>(my_type)) H.bad = 1;
  const bool wr = v && is_writer(my_type), lk = v && kv_lock_op<WL>(my_type), aq = v && WL == DINT_WL_TATP && my_type == 1;
  const uint64_t bw = __ballot(wr), bl = __ballot(lk), ba = __ballot(aq);
  if (lane == 0) { H.Bw[wave] = bw; H.Bl[wave] = bl; H.Ba[wave] = ba; }
  H.idx[t] = my_idx;
  __syncthreads();
  // what precedes me inside the piece, and the piece's totals (every thread: 8 words each)
  uint32_t wr_below = 0, wr_tot = 0;
  int lw_below = -1, lw_tot = -1, ll_below = -1, ll_tot = -1;  // sorted positions
#pragma unroll
  for (uint32_t w = 0; w < KVB_W; w++) {
    const uint64_t mw = H.Bw[w], ml = H.Bl[w];
    const uint64_t mwb = w < wave ? mw : (w == wave ? mw & lanemask_lt() : 0ull), mlb = w < wave ? ml : (w == wave ? ml & lanemask_lt() : 0ull);
    wr_below += (uint32_t)__popcll(mwb); wr_tot += (uint32_t)__popcll(mw);
    if (mwb) lw_below = (int)(w * 64 + 63 - __clzll((long long)mwb));
    if (mw) lw_tot = (int)(w * 64 + 63 - __clzll((long long)mw));
    if (mlb) ll_below = (int)(w * 64 + 63 - __clzll((long long)mlb));
    if (ml) ll_tot = (int)(w * 64 + 63 - __clzll((long long)ml));
  }
  auto acq_at = [&](int p) -> int { return (int)((H.Ba[p >> 6] >> (p & 63)) & 1ull); };
  const unsigned long long mine_w = kvh_word(seq, !H.bad, wr_tot, lw_tot >= 0 ? (int)H.idx[lw_tot] : -1, ll_tot >= 0 ? acq_at(ll_tot) : -1);
  bool all_ok = !H.bad;
  if (nsib > 1 && voting) {  // tell the siblings, hear from them
    if (t == 0) __hip_atomic_store(&pub[j], mine_w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (t < nsib) {
      unsigned long long w = 0;
      uint32_t spins = 0;
      for (;;) {
        w = __hip_atomic_load(&pub[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((uint32_t)(w >> 34) == (seq & 0x3FFFFFFFu)) break;
        // (siblings are running or about to be started -- see above; a bound nevertheless: a hung GPU is worse than a trap)
        if (++spins > (1u << 22)) { H.timeout = 1; break; }
        __builtin_amdgcn_s_sleep(2);
      }
      H.pub[t] = w;
    }
    __syncthreads();
    if (H.timeout) __builtin_trap();
    for (uint32_t k = 0; k < nsib; k++) all_ok = all_ok && ((H.pub[k] >> 33) & 1ull);
  } else {
    if (t == 0) H.pub[0] = mine_w;
    __syncthreads();
  }
  // (DINT_KV_TRACE: kind, piece, pieces | in closed form << 16 | the key's requests here << 20 | the remainder's << 32)
  if (ttr && t == 0) ttr[30] = (uint64_t)(d.w & 0xFFFFu) | ((uint64_t)(all_ok ? 1u : 0u) << 16) | ((uint64_t)(c & 0xFFFu) << 20) | ((uint64_t)c_rem << 32);
  if (!all_ok && voting) {
    // not in closed form: the old way, by the sub's first item.  The hot bucket has neighbours: the whole sub (its remainder
    // waited for the vote and does nothing).  It has none: the hot key's records only, compacted behind the remainder's in
    // ovf2 -- the remainder went its own way long ago.
    if (j != 0 || !do_piece) return 0;
    if (nbr || solo) { *src = 0; *off = d.y; *cnt = h; return 1; }
    const uint32_t nsame = H.nsame;
    __syncthreads();
    if (t == 0) H.nhot = 0;
    __syncthreads();
    for (uint32_t k0 = 0; k0 < h; k0 += KVB_T) {
      const uint32_t k = k0 + t;
      const uint64_t r = k < h ? kv_ld_agent(&ovf[d.y + k]) : 0;
      const bool same = k < h && (r >> sh_g) == (hrec >> sh_g) && pay_kh(kv_rec_pay(r)) == hkh;
      const uint64_t mm = __ballot(same);
      uint32_t at = 0;
      if (lane == 0 && mm) at = atomicAdd(&H.nhot, (uint32_t)__popcll(mm));
      at = (uint32_t)__builtin_amdgcn_readfirstlane((int)at) + (uint32_t)__popcll(mm & lanemask_lt());
      if (same) ovf2[d.y + (h - nsame) + at] = r;
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    *src = 1; *off = d.y + (h - nsame); *cnt = nsame;
    return 1;
  }
  if (c != 0) {
    // ---- what the pieces before mine leave me, and who stores the row / the lock byte at the end
    uint32_t nw_before = 0, nw_all = 0;
    int lw_before = -1, la_before = -1, jw = -1, jl = -1, la_fin = -1;
    for (uint32_t k = 0; k < np; k++) {
      const unsigned long long w = H.pub[k];
      const uint32_t nwk = (uint32_t)(w >> 23) & 1023u;
      if (k < j) {
        nw_before += nwk;
        if ((w >> 22) & 1ull) lw_before = (int)((w >> 2) & 0xFFFFFu);
        if (w & 2ull) la_before = (int)(w & 1ull);
      }
      nw_all += nwk;
      if (nwk) jw = (int)k;
      if (w & 2ull) { jl = (int)k; la_fin = (int)(w & 1ull); }
    }
    const uint32_t found = H.found, ver0 = H.ver0, la0 = H.la0;
    const kv_tab tb = kv->tab[H.table];
    const uint32_t gk = kv_cut_gk((uint32_t)(hrec >> sh_g), d.x, cut2);
    const uint64_t bucket = (uint64_t)(gk - kv->gk_base[H.table]);
    uint8_t *row = kv_entry_ptr(tb, bucket, H.link) + KV_VAL_OFF + H.slot * F::VS;  // meaningful when found (written, never read, here)
    if (v) {
      uint8_t *msg = rep + dint_view_off(V, my_idx, F::MSG);
      uint32_t code;
      bool get = false;
      if (WL == DINT_WL_STORE) {
        code = my_type == 0 ? (found ? 3 : 7) : (found ? 5 : 7);
        get = my_type == 0 && found;
      } else {
        const int seen = ll_below >= 0 ? acq_at(ll_below) : (la_before >= 0 ? la_before : (int)la0);
        switch (my_type) {
          case 0: code = found ? 4 : 6; get = found != 0; break;
          case 1: code = seen ? 8 : 7; break;
          case 2: code = 9; break;
          case 12: code = 15; break;
          default: code = 16; break;  // 13 kCommitBck
        }
      }
      if (get) {  // the row as of my position: the last writer before me (its message still holds the value), else the row as
        const int widx = lw_below >= 0 ? (int)H.idx[lw_below] : lw_before;  // it was before the pass (H.rowv)
        if (widx >= 0) {
          kv_copy_words(msg + F::VAL, rep + dint_view_off(V, (uint32_t)widx, F::MSG) + F::VAL, F::VS);
        } else {
#pragma unroll
          for (uint32_t k = 0; k < F::VS / 4; k++) st_u32(msg + F::VAL + 4 * k, H.rowv[k]);
        }
        st_u32(msg + F::VER, ver0 + nw_before + wr_below);
      }
      msg[F::TYPE] = (uint8_t)code;
    }
    // ---- the row and the lock byte, once: by the piece that holds the pass's last writer / last lock op
    if (found && nw_all && (int)j == jw && (int)t == lw_tot) {
      kv_copy_words(row, rep + dint_view_off(V, my_idx, F::MSG) + F::VAL, F::VS);
      KV_ST(uint32_t, &kv_entry_hdr(tb, bucket, H.link)->ver[H.slot], ver0 + nw_all);
    }
    if (WL == DINT_WL_TATP && t == 0) {
      if ((int)j == jl && la_fin >= 0 && (uint32_t)la_fin != la0) KV_ST(uint8_t, kv_entry_ptr(tb, bucket, KV_INLINE) + KV_LOCKB_OFF + hq, (uint8_t)la_fin);
      if (!found && wr_tot) atomicAdd(&stats->missing_keys, (unsigned long long)wr_tot);  // tatp/udp/kvs.h:91 (the reference panics)
    }
  }
  if (c_rem == 0) return 0;
  // ---- the remainder: other buckets, or other rows of the hot bucket -- beside the pieces
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");  // (a solo item: the hot row's write-back above, then its bucket's other rows)
  // (r06 tried the remainder as a SOLO item here -- a second hot key of the sub answered in closed form instead of going to the
  // late list: late items 50 -> 30 per 600 passes, the kernel 72 -> 224 bytes of scratch per lane, the bench +0.7 % (tatp) / -2 %
  // (store).  Not kept: k_kv_late does the same off this kernel's register budget.)
  if (c_rem <= KVB_T && kv_rem_chunks<WL>(rep, cut2, kv, d.x, H.rem, c_rem, H, stats, force_rounds, V)) return 0;
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");  // ... this workgroup's own ovf2 stores before kv_big_bin's loads
  *src = 1; *off = d.y; *cnt = c_rem;
  return 1;
}

// the pass's big subs, KVB_GRID workgroups per engine taking them in turn (the longest job of a pass: a hot key)
// ---- the LIGHT general path for what k_kv_hot leaves (r06: k_kv_late) -----------------------------------------------------------
// r05 handed a late item to k_kv_big behind k_kv_hot: kv_big_bin's workgroup (256 VGPRs x 8 waves, 80 KB of LDS) needs an EMPTY
// compute unit, and beside the other shard servers' kernels even the EMPTY launch of the usual pass waited 14 .. 19 us for one --
// a seventh of the pass's chain for nothing.  A late item is rare (none in the bench streams) and small, so it gets a general
// path out of the pieces k_kv_hot is made of, at k_kv_hot's footprint (128 VGPRs, ~30 KB of LDS: the kernel starts beside
// anything): the item's records in STRETCHES of <= KVB_T along the request index (every request of a stretch precedes every
// request of the next; one stretch when the item is small), a stretch sorted by (bucket group, key hash, idx), and then, until
// nothing is left: no bucket group longer than a chunk -> kv_rem_chunks (kv_chunk: every closed form, rounds where none applies);
// else the longest group through kv_group_phases (its requests in request order, lock bytes per quadrant, rows in phases) or --
// more than KVG_MAXW row-changing requests, an op outside the servers' set, the overflow pool nearly empty -- request by
// request through kv_do_request, the reference-shaped implementation every closed form is tested against
// (tatp/udp/server_shard.cc:116-207, store/udp/server.cc:75-97).  Slower than kv_big_bin per record (a 3,000-request hot row
// with a DELETE in it: ~150 us instead of ~50), correct for anything, and 0 us when there is nothing to do.
template <int WL>
__device__ __forceinline__ static void kv_serial_group(uint8_t *rep, const kv_cut cut2, const kv_dev *kv, uint32_t bin, const uint64_t *R, uint32_t Gs,
                                                       uint32_t Ge, kvh_lds &H, dint_dev_stats *__restrict__ stats, const dint_view V) {
  using F = Fmt<WL>;
  const uint32_t t = threadIdx.x, n = Ge - Gs;
  const uint32_t sh_g = 16 + cut2.ibits, idx_mask = (uint32_t)((1ull << cut2.ibits) - 1ull);
  __syncthreads();
  {
    const uint64_t w = t < n ? R[Gs + t] : 0;
    const uint32_t idx = (uint32_t)(w >> 7) & idx_mask, pay = (uint32_t)w & 0x7Fu;
    H.key[t] = t < n ? (idx << 9) | t : 0xFFFFFFFFu;
    H.typ[t] = (uint8_t)(pay_type(pay) & 0xFFu);
    H.qq[t] = (uint8_t)pay_q(pay);
  }
  __syncthreads();
  kvb_sort_blocked_u32<1>(H.key, KVB_T);
  if (t == 0) {
    const uint32_t gk = kv_cut_gk((uint32_t)(R[Gs] >> sh_g), bin, cut2), table = kv_table_of(kv, gk);
    const uint64_t bucket = (uint64_t)(gk - kv->gk_base[table]);
    for (uint32_t p = 0; p < n; p++) {
      const uint32_t sk = H.key[p], slot = sk & 511u;
      kv_do_request<WL>(rep + dint_view_off(V, sk >> 9, F::MSG), H.typ[slot], table, H.qq[slot], bucket, kv, stats);
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");  // the next request of the bucket sees this one
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
  __syncthreads();
}
// h <= KVB_T sort words in H.all[0, h), ~0 behind them, in any order
template <int WL>
__device__ __forceinline__ static void kv_late_stretch(uint8_t *rep, const kv_cut cut2, const kv_dev *kv, uint32_t bin, uint32_t h, kvh_lds &H,
                                                       dint_dev_stats *__restrict__ stats, int force_rounds, const dint_view V) {
  const uint32_t t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const uint32_t sh_g = 16 + cut2.ibits;
  // The closed forms first, as long as they make progress: what lands here is usually a SECOND hot key of a sub (the remainder
  // beside another key's pieces: two hot subscribers in one of the pass's ~30,000 subs -- one pass in ten of the bench stream)
  // or a hot key one of whose pieces said no.  kv_solo_item sorts the stretch, answers its longest run of one key in closed
  // form (or its longest bucket group in phases), the rest through the chunks -- and leaves in H.rem what the chunks refuse
  // (another long group): that is the next round's stretch.
  for (;;) {  // (workgroup-uniform; every round answers at least 65 requests)
    uint32_t src = 0, off = 0, cnt = h;
    const int run = kv_solo_item<WL>(rep, cut2, kv, make_uint4(bin, 0u, h, KVQ_SOLO), nullptr, nullptr, stats, force_rounds, V, (uint8_t *)&H, &src, &off,
                                     &cnt, nullptr, true);
    if (!run) return;
    if (!(src == 1 && cnt < h)) break;  // nothing touched: H.all holds the stretch, sorted
    __syncthreads();
    const uint64_t w = t < cnt ? H.rem[t] : ~0ull;
    __syncthreads();
    H.all[t] = w;
    h = cnt;
    __syncthreads();
  }
  uint32_t m = h;
  for (;;) {  // (workgroup-uniform: at most KVB_T / 65 groups are longer than a chunk)
    __syncthreads();
    const uint64_t w = H.all[t];
    const bool head_g = t < m && (t == 0 || (H.all[t - 1] >> sh_g) != (w >> sh_g));
    const uint64_t mg = __ballot(head_g);
    if (t == 0) H.bestg = 0;
    if (lane == 0) H.Bl[wave] = mg;
    __syncthreads();
    if (head_g) {  // the first group head above me, or m
      uint32_t nxt = m;
      for (uint32_t wv = wave; wv < KVB_W && nxt == m; wv++) {
        uint64_t mm = H.Bl[wv];
        if (wv == wave) mm &= ~((2ull << lane) - 1ull);
        if (mm) nxt = wv * 64 + (uint32_t)__ffsll((unsigned long long)mm) - 1;
      }
      atomicMax(&H.bestg, ((nxt - t) << 16) | t);
    }
    __syncthreads();
    const uint32_t Gs = H.bestg & 0xFFFFu, len = H.bestg >> 16;
    if (len <= 64) {  // chunks for everything that is left
      kv_rem_chunks<WL>(rep, cut2, kv, bin, H.all, m, H, stats, force_rounds, V, true);
      return;
    }
    if (force_rounds || !kv_group_phases<WL>(rep, cut2, kv, bin, H.all, Gs, Gs + len, H, stats, V))
      kv_serial_group<WL>(rep, cut2, kv, bin, H.all, Gs, Gs + len, H, stats, V);  // (kv_do_request: 550 of k_kv_late's 760 bytes of scratch per lane
                                                                                   // -- a build without it ran the bench no faster, r06)
    __syncthreads();
    const bool mv = t >= Gs + len && t < m;  // the group leaves the array
    const uint64_t mw = mv ? H.all[t] : 0;
    __syncthreads();
    if (mv) H.all[t - len] = mw;
    if (t >= m - len && t < m) H.all[t] = ~0ull;
    m -= len;
    if (m == 0) return;
  }
}
#define KVL_RANGES 4096u  // index ranges of a long late item: <= 256 request indices each, hence <= 256 records each
struct kvl_lds {
  uint32_t pre[KVL_RANGES];  // records per index range, then their exclusive prefix sums
  uint32_t wsum[KVB_W];
  uint32_t cnt;
};
template <int WL>
__device__ __forceinline__ static void kv_late_item(const kv_pass_args &A, const kv_cut cut2, const kv_dev *kv, const uint4 d, kvh_lds &H, kvl_lds &L) {
  const uint32_t t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const uint64_t *__restrict__ recs = (d.w ? (const uint64_t *)A.ovf2 : A.ovf) + d.y;
  const uint32_t cnt = d.z, bin = d.x;
  if (cnt == 0) return;
  if (cnt <= KVB_T) {
    __syncthreads();
    H.all[t] = t < cnt ? kv_sort_key(recs[t], cut2) : ~0ull;
    kv_late_stretch<WL>(A.rep, cut2, kv, bin, cnt, H, A.stats, A.force_flags & 1, A.V);
    return;
  }
  // Stretches along the request index (every request of a stretch precedes every request of the next): a range of `rw` <= 256
  // indices holds at most 256 records (indices are distinct); stretch of a range = (records in the ranges before it) / 256, so
  // a stretch is consecutive ranges that START inside one window of 256 records: at most 256 + 256 records.
  const uint32_t rw = (A.n + KVL_RANGES - 1) / KVL_RANGES;
  __syncthreads();
  for (uint32_t k = t; k < KVL_RANGES; k += KVB_T) L.pre[k] = 0;
  __syncthreads();
  for (uint32_t k = t; k < cnt; k += KVB_T) atomicAdd(&L.pre[min(KVL_RANGES - 1, kv_rec_idx(recs[k], cut2) / rw)], 1u);
  __syncthreads();
  {  // exclusive prefix sums in place: KVL_RANGES / KVB_T consecutive ranges per thread, a wave scan, the waves' totals
    constexpr uint32_t PT = KVL_RANGES / KVB_T;
    uint32_t v[PT], sum = 0;
#pragma unroll
    for (uint32_t k = 0; k < PT; k++) { v[k] = L.pre[t * PT + k]; sum += v[k]; }
    uint32_t tot, base = wave_excl_scan_u32(sum, &tot);
    if (lane == 0) L.wsum[wave] = tot;
    __syncthreads();
    for (uint32_t w = 0; w < wave; w++) base += L.wsum[w];
#pragma unroll
    for (uint32_t k = 0; k < PT; k++) { L.pre[t * PT + k] = base; base += v[k]; }
  }
  __syncthreads();
  const uint32_t ns = (cnt + 255u) >> 8;
  for (uint32_t s = 0; s < ns; s++) {
    __syncthreads();
    if (t == 0) L.cnt = 0;
    H.all[t] = ~0ull;
    __syncthreads();
    for (uint32_t k = t; k < cnt; k += KVB_T) {
      const uint64_t r = recs[k];
      if ((L.pre[min(KVL_RANGES - 1, kv_rec_idx(r, cut2) / rw)] >> 8) == s) H.all[atomicAdd(&L.cnt, 1u)] = kv_sort_key(r, cut2);
    }
    __syncthreads();
    const uint32_t h = L.cnt;
    if (h) kv_late_stretch<WL>(A.rep, cut2, kv, bin, h, H, A.stats, A.force_flags & 1, A.V);
  }
}
#define KVL_GRID 4u
template <int WL, int WPS /* waves per SIMD: 4 = k_kv_hot's footprint (128 VGPRs, spills), 2 = 256 VGPRs (needs an empty CU) */>
__global__ void __launch_bounds__(KVB_T, WPS) k_kv_late(kv_multi_args M, uint32_t dry /* timing experiments: read the count and leave */) {
  __shared__ kv_dev Skv;
  __shared__ __attribute__((aligned(16))) uint8_t Lraw[sizeof(kvh_lds)];
  __shared__ kvl_lds Ll;
  const kv_pass_args &A = M.e[blockIdx.y];
  const uint32_t nq = A.big[5];
  if (blockIdx.x >= nq || dry) return;  // (almost always: nothing was left)
  const uint32_t t = threadIdx.x;
  if (t < sizeof(kv_dev) / 4) ((uint32_t *)&Skv)[t] = ((const uint32_t *)A.kv)[t];
  __syncthreads();
  kv_dev_pend_set(Skv, A.pno);
  __syncthreads();
  kv_cut cut2 = A.cut;
  cut2.P = A.cut.P * KVR_F;
  for (uint32_t i = blockIdx.x; i < nq; i += gridDim.x) kv_late_item<WL>(A, cut2, &Skv, A.lateq[i], *(kvh_lds *)Lraw, Ll);
}

// ---- smallbank: a hot account's row in PIECES, several workgroups at once (r06; VERDICT r03 .. r05) ----------------------------------
// kv_big_bin takes a hot account -- 16,000 .. 35,000 requests of a 360k-request pass at Zipf 0.99 -- in stretches of 4,096, one
// after the other on ONE workgroup: 4 .. 9 x 45 us while 255 compute units wait.  Of a stretch only the grant decisions are
// serial (the shared / exclusive counters: FREE until an ACQUIRE_EXCLUSIVE finds num_sh == 0, HELD until the num_ex-th
// RELEASE_EXCLUSIVE -- kv_sb_walk), ~5 us of the 45; collecting, ordering, key checks and the replies are not.  So, as store /
// tatp's hot keys (kv_hot_item): the resolve workgroup lists the sub as np PIECES (ranges of the request index, <= KVB_T requests
// of the hot key each) + a REMAINDER (the sub's other keys), and
//   every piece   picks its requests out of the sub's records, orders them by index, checks their keys, reads header and row, and
//                 publishes its op kinds as lane masks per 64-request chunk {ACQ_SHARED, ACQ_EXCL, REL_SHARED, REL_EXCL} (32 words)
//                 + one word {ok, writers, index of the last writer};
//   piece 0       -- the COORDINATOR -- waits for all np words, gathers the masks (np x 8 chunks, one per lane, 64 at a time) and
//                 walks them with the counters in registers: all chunks of a round checked AT ONCE for whether they can change the
//                 counters' mode, only those that can are walked (the scan kv_big_bin runs inside one stretch, lock_2pl's groups);
//                 it publishes every chunk's granted lanes, stores the counters, and says "go";
//   every piece   answers its requests: grants from the coordinator's masks, version = v0 + writers before me, value = message of
//                 the last writer before me (else the row) -- the writers before a piece come from the pieces' words;
//                 the piece with the pass's last COMMIT stores row and version.
// The serial part of a 35,000-request account is one wave's pass over ~560 chunk masks instead of nine stretches.
// ANOTHER KEY ON THE ROW'S COUNTER PAIR (a cold account whose row hashes to the same bucket and lock quadrant -- the lock table is
// 2/3 full -- with a request or two in this pass): kv_big_bin then runs the hot row's thousands of requests in ROUNDS, 1 .. 4 ms,
// the passes behind r05's p99.  Here a piece takes every request ON THE PAIR of its index range: the foreign ones' lock ops sit in
// the masks like the hot row's (the counters see them in request order), their row ops (the GET of a granted ACQUIRE, a COMMIT's
// SET) are few and are done by the coordinator, one after the other in request order, before anybody else touches the table.
// ALL OR NOTHING as kv_hot_item: a piece with more than KVB_T requests or more than KSB_FMAX foreign ones, an unknown op -- then
// piece 0 takes the whole sub the old way (kv_big_bin) and its siblings do nothing.
// Semantics per op: smallbank/udp/server_shard.cc:121-173 (as kv_do_request).  Nobody stores to the table before the
// coordinator's word is out, i.e. before every piece has read header and row.
#define KSB_NPMAX 128u   // pieces of one hot row at most (~49,000 requests; beyond: kv_big_bin)
#define KSB_WORDS DINT_KV_SBX_WORDS  // words per item in sbx: [4 c + k] op kind k of chunk c (k: AS, AX, RS, RX), [32 + c] the granted lanes of chunk c,
                                    // [40 + f] the piece's f-th FOREIGN request (another key on the row's counter pair): valid << 63 | type << 48 | position << 32 | idx
#define KSB_FMAX 8u      // foreign requests per piece at most (a cold account that shares the hot row's lock slot: a handful per pass)
struct kvs_lds {
  uint32_t key[KVB_T];            // idx << 9 | slot: sorted = the piece in request order
  uint32_t idx[KVB_T];            // request index at each sorted position
  uint8_t typ[KVB_T];             // request type by slot
  uint64_t Mk[4][KVB_W], Mw[KVB_W], G[KVB_W];
  unsigned long long pub[KSB_NPMAX + 1];
  uint64_t fl[KSB_FMAX];              // a piece's own foreign requests
  uint32_t bad, timeout, nhot, nrem, found, link, slot, ver0, table, la, lb, allok, nf;
  uint32_t rowv[2];
};
template <int WL>
__device__ __forceinline__ static int kv_sb_item(uint8_t *rep, const kv_cut cut2, const kv_dev *kv, const uint4 d, const uint4 x, const uint4 y,
                                                 const uint64_t *__restrict__ ovf, uint64_t *__restrict__ ovf2, unsigned long long *hotpub, uint64_t *sbx,
                                                 uint32_t seq, uint32_t inv_n, dint_dev_stats *__restrict__ stats, const dint_view V, uint8_t *lds_raw,
                                                 uint32_t *src, uint32_t *off, uint32_t *cnt) {
  using F = Fmt<WL>;
  static_assert(sizeof(kvs_lds) <= sizeof(kvb_lds) && sizeof(kvs_lds) <= 16384, "the pieces' LDS lives in kv_big_bin's buffer (k_kv_big) or a worker's (k_kv_pass: four workgroups per CU)");
  kvs_lds &H = *(kvs_lds *)lds_raw;
  const uint32_t t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const uint32_t kind = d.w & 3u, j = (d.w >> 2) & 255u, np = (d.w >> 10) & 255u;
  const uint32_t first = x.z, h = d.z;
  const uint32_t sh_g = 16 + cut2.ibits, idx_mask = (uint32_t)((1ull << cut2.ibits) - 1ull);
  const uint64_t hkey = ((uint64_t)x.y << 32) | x.x, hrec = ((uint64_t)y.y << 32) | y.x;
  const uint32_t hq = pay_q(kv_rec_pay(hrec)), hkh = pay_kh(kv_rec_pay(hrec));
  // (SOLO: the sub's only piece answers the row's counter pair in closed form and hands the sub's other keys to kv_big_bin itself --
  // nobody else reads its words, which live apart from the item words of the neighbours)
  const bool solo = kind == KVQ_SOLO, do_piece = kind != KVQ_REM;
  unsigned long long *pub = solo ? hotpub + DINT_KV_BIGQ_MAX + 2 * (size_t)first : hotpub + first;  // [0, np): the pieces' words; [np]: the coordinator's
  *src = 0; *off = d.y; *cnt = h;
  auto spin_for = [&](unsigned long long *p) -> unsigned long long {  // a word of this pass (siblings hold tickets or draw the next ones)
    unsigned long long w = 0;
    for (uint32_t spins = 0;; spins++) {
      w = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if ((uint32_t)(w >> 34) == (seq & 0x3FFFFFFFu)) break;
      if (spins > (1u << 22)) { H.timeout = 1; break; }
      __builtin_amdgcn_s_sleep(2);
    }
    return w;
  };
  __syncthreads();  // the LDS buffer is free
  H.key[t] = 0xFFFFFFFFu;
  if (t == 0) { H.bad = 0; H.timeout = 0; H.nhot = 0; H.nrem = 0; H.allok = 0; H.nf = 0; }
  if (t < KSB_FMAX) H.fl[t] = 0;
  __syncthreads();
  // ---- one pass over the sub's records: the requests on the hot row's COUNTER PAIR of my index range are mine; the others are the remainder's
  for (uint32_t k0 = 0; k0 < h; k0 += KVB_T) {
    const uint32_t k = k0 + t;
    const uint64_t r = k < h ? kv_ld_agent(&ovf[d.y + k]) : 0;
    const uint32_t pay = kv_rec_pay(r), ridx = (uint32_t)(r >> 16) & idx_mask, type = pay_type(pay);
    const bool same = k < h && (r >> sh_g) == (hrec >> sh_g) && pay_q(pay) == hq;  // on the pair: the hot row, or another key of its bucket and quadrant
    (void)type; (void)hkh;
    if (do_piece) {
      const bool mine = same && kv_piece_of(ridx, np, inv_n) == j;
      const uint64_t mm = __ballot(mine);
      uint32_t at = 0;
      if (lane == 0 && mm) at = atomicAdd(&H.nhot, (uint32_t)__popcll(mm));
      at = (uint32_t)__builtin_amdgcn_readfirstlane((int)at) + (uint32_t)__popcll(mm & lanemask_lt());
      if (mine && at < KVB_T) { H.key[at] = (ridx << 9) | at; H.typ[at] = (uint8_t)(type & 0xFFu); }
    }
    if (!do_piece || solo) {
      const bool other = k < h && !same;
      const uint64_t mm = __ballot(other);
      uint32_t at = 0;
      if (lane == 0 && mm) at = atomicAdd(&H.nrem, (uint32_t)__popcll(mm));
      at = (uint32_t)__builtin_amdgcn_readfirstlane((int)at) + (uint32_t)__popcll(mm & lanemask_lt());
      if (other) ovf2[d.y + at] = r;  // compacted, for kv_big_bin
    }
  }
  __syncthreads();
  if (!do_piece) {  // the REMAINDER: once the pieces are in closed form, the sub's other keys the old way -- else nothing (piece 0 takes the sub)
    if (t == 0) H.pub[0] = spin_for(&pub[np]);
    __syncthreads();
    if (H.timeout) __builtin_trap();
    if (!((H.pub[0] >> 33) & 1ull) || H.nrem == 0) return 0;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    *src = 1; *off = d.y; *cnt = H.nrem;
    return 1;
  }
  const uint32_t c = H.nhot;
  if (t == 0 && c > KVB_T) { H.bad = 1; atomicAdd(&stats->late_items[0], 1ULL); }  // (why a hot row was not answered in pieces: stats, late_items)
  // ---- the row (every piece: the value a read may need) and, piece 0, the counters -- BEFORE anything is published
  kv_tab tb;
  uint64_t bucket = 0;
  uint32_t table = 0;
  {
    const uint32_t gk = kv_cut_gk((uint32_t)(hrec >> sh_g), d.x, cut2);
    table = kv_table_of(kv, gk);
    tb = kv->tab[table];
    bucket = (uint64_t)(gk - kv->gk_base[table]);
  }
  uint8_t *ie = kv_entry_ptr(tb, bucket, KV_INLINE);
  if (t == 0) {
    kv_hdr Hd;
    kv_hdr_load(Hd, ie);
    const kv_where w = kv_locate(tb, bucket, Hd, hkey);
    H.found = w.found; H.link = w.link; H.slot = w.slot; H.ver0 = w.ver; H.table = table;
    if (w.found) {
      const uint8_t *rv = kv_entry_ptr(tb, bucket, w.link) + KV_VAL_OFF + w.slot * F::VS;
      H.rowv[0] = KV_LD(uint32_t, rv); H.rowv[1] = KV_LD(uint32_t, rv + 4);
    }
    const uint2 cc = *(const uint2 *)(ie + KV_SB_LOCK_OFF + 8 * hq);
    H.la = cc.x; H.lb = cc.y;
  }
  __syncthreads();
  kvb_sort_blocked_u32<1>(H.key, KVB_T);
  const uint32_t sk = H.key[t];
  const bool v = sk != 0xFFFFFFFFu;
  const uint32_t my_idx = sk >> 9, my_type = v ? H.typ[sk & 511u] : 0xFFu;
  uint8_t *msg = rep + dint_view_off(V, v ? my_idx : 0u, F::MSG);
  const bool foreign = v && ld_u64(msg + F::KEY) != hkey;  // another key on the pair: its lock ops count, its row ops are the coordinator's
  if (v && !(my_type <= 5u || my_type == 17u)) { H.bad = 1; atomicAdd(&stats->late_items[2], 1ULL); }
  const bool wr = v && !foreign && (my_type == 4u || my_type == 5u);
  {
    const uint64_t b0 = __ballot(v && my_type == 0u), b1 = __ballot(v && my_type == 1u), b2 = __ballot(v && my_type == 2u), b3 = __ballot(v && my_type == 3u);
    const uint64_t bw = __ballot(wr), bf = __ballot(foreign);
    if (lane == 0) { H.Mk[0][wave] = b0; H.Mk[1][wave] = b1; H.Mk[2][wave] = b2; H.Mk[3][wave] = b3; H.Mw[wave] = bw; H.G[wave] = bf; }
  }
  H.idx[t] = my_idx;
  __syncthreads();
  if (foreign) {  // the piece's foreign requests in request order (the piece is sorted by index): slot = foreign requests before me
    uint32_t f = (uint32_t)__popcll(H.G[wave] & lanemask_lt());
    for (uint32_t w = 0; w < wave; w++) f += (uint32_t)__popcll(H.G[w]);
    if (f < KSB_FMAX) H.fl[f] = (1ull << 63) | ((uint64_t)my_type << 48) | ((uint64_t)t << 32) | my_idx;
    else { H.bad = 1; if (f == KSB_FMAX) atomicAdd(&stats->late_items[1], 1ULL); }
  }
  __syncthreads();
  uint32_t wr_below = 0, wr_tot = 0;
  int lw_below = -1, lw_tot = -1;
#pragma unroll
  for (uint32_t w = 0; w < KVB_W; w++) {
    const uint64_t mw = H.Mw[w], mwb = w < wave ? mw : (w == wave ? mw & lanemask_lt() : 0ull);
    wr_below += (uint32_t)__popcll(mwb); wr_tot += (uint32_t)__popcll(mw);
    if (mwb) lw_below = (int)(w * 64 + 63 - __clzll((long long)mwb));
    if (mw) lw_tot = (int)(w * 64 + 63 - __clzll((long long)mw));
  }
  // ---- publish: the op masks and the foreign requests, then the word
  if (t < 4 * KVB_W) kv_st_agent(&sbx[(size_t)(first + j) * KSB_WORDS + t], H.Mk[t & 3u][t >> 2]);
  if (t < KSB_FMAX) kv_st_agent(&sbx[(size_t)(first + j) * KSB_WORDS + 5 * KVB_W + t], H.fl[t]);
  if (wave == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (t == 0) __hip_atomic_store(&pub[j], kvh_word(seq, !H.bad, wr_tot, lw_tot >= 0 ? (int)H.idx[lw_tot] : -1, -1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  // ---- piece 0, the coordinator
  if (j == 0) {
    if (t < np) H.pub[t] = spin_for(&pub[t]);
    __syncthreads();
    if (H.timeout) __builtin_trap();
    bool all_ok = true;
    for (uint32_t k = 0; k < np; k++) all_ok = all_ok && ((H.pub[k] >> 33) & 1ull);
    const uint32_t nchunk = np * KVB_W;
    if (all_ok) {
      if (wave == 0) {  // lane l holds chunk base + l; all 64 checked at once under the assumption that the mode holds (kv_big_bin)
        uint32_t la = H.la, lb = H.lb;
        for (uint32_t base = 0; base < nchunk; base += 64) {
          const uint32_t ci = base + lane;
          const bool inr = ci < nchunk;
          // (the pieces' masks straight from where they were published: no copy in LDS -- 40 KB that kept kv_sb_item out of the light kernels)
          uint64_t *mine_w = &sbx[(size_t)(first + ci / KVB_W) * KSB_WORDS];
          const uint64_t *mw = mine_w + 4 * (ci % KVB_W);
          const uint64_t cAS = inr ? kv_ld_agent(mw) : 0ull, cAX = inr ? kv_ld_agent(mw + 1) : 0ull, cRS = inr ? kv_ld_agent(mw + 2) : 0ull, cRX = inr ? kv_ld_agent(mw + 3) : 0ull;
          const uint32_t nas = (uint32_t)__popcll(cAS), nrs = (uint32_t)__popcll(cRS);
          uint64_t g = 0, pend = __ballot((cAS | cAX | cRS | cRX) != 0);
          while (pend) {
            const bool me = (pend >> lane) & 1ull;
            const uint32_t dl = me ? (la != 0 ? 0u - nrs : nas - nrs) : 0u;  // what my chunk adds to num_sh if it is inert
            uint32_t tot, pre = wave_excl_scan_u32(dl, &tot);
            const uint32_t lb_in = lb + pre;
            const bool inert = la != 0 ? cRX == 0 : (cRX == 0 && (cAX == 0 || (lb_in > nrs && lb_in <= 0xFFFFFFFFu - nas)));
            const uint64_t stop = __ballot(me && !inert);
            const uint64_t ok = stop ? pend & ((stop & (0 - stop)) - 1ull) : pend;  // the chunks before the first one that is not
            if (((ok >> lane) & 1ull) && la == 0) g |= cAS;
            if (!stop) { lb += tot; break; }
            const int f = __ffsll((unsigned long long)stop) - 1;
            lb += (uint32_t)__builtin_amdgcn_readlane((int)pre, f);  // (the chunks before f)
            const uint64_t mAS = readlane_u64(cAS, f), mAX = readlane_u64(cAX, f), mRS = readlane_u64(cRS, f), mRX = readlane_u64(cRX, f);
            const uint64_t G = kv_sb_walk(mAS | mAX | mRS | mRX, mAS, mAX, mRS, mRX, la, lb);
            if ((int)lane == f) g |= G;
            pend &= ~(ok | (1ull << f));
          }
          if (inr) kv_st_agent(mine_w + 4 * KVB_W + (ci % KVB_W), g);
        }
        if (lane == 0 && (la != H.la || lb != H.lb)) *(uint2 *)(ie + KV_SB_LOCK_OFF + 8 * hq) = make_uint2(la, lb);  // (every piece has read the row: their words are in)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the grants are out: the foreign requests below read them back)
      }
      // the foreign requests of the whole sub, in request order -- piece after piece, and every piece lists its own in order --, one
      // lane: grants from the walk, row ops with the reference's own chain walk (kv_apply); nobody else has stored to the table yet
      if (wave == 0) {
        for (uint32_t base = 0; base < np * KSB_FMAX; base += 64) {
          const uint32_t fw = base + lane;
          const uint64_t fe_l = fw < np * KSB_FMAX ? kv_ld_agent(&sbx[(size_t)(first + fw / KSB_FMAX) * KSB_WORDS + 5 * KVB_W + (fw % KSB_FMAX)]) : 0ull;
          uint64_t todo = __ballot((fe_l >> 63) != 0);
          while (todo) {
            const int fl = __ffsll((unsigned long long)todo) - 1;
            todo &= todo - 1;
            const uint64_t fe = readlane_u64(fe_l, fl);
            if (lane != 0) continue;
            const uint32_t ftype = (uint32_t)(fe >> 48) & 0xFFu, fpos = (uint32_t)(fe >> 32) & 511u, fpiece = (base + (uint32_t)fl) / KSB_FMAX;
            uint8_t *fm = rep + dint_view_off(V, (uint32_t)(fe & 0xFFFFFu), F::MSG);
            const uint64_t fkey = ld_u64(fm + F::KEY);
            const bool granted = (kv_ld_agent(&sbx[(size_t)(first + fpiece) * KSB_WORDS + 4 * KVB_W + (fpos >> 6)]) >> (fpos & 63u)) & 1ull;
            uint32_t code, act = KV_ACT_NONE;
            bool counts = false;
            switch (ftype) {
              case 0: code = granted ? 7 : 8; if (granted) { act = KV_ACT_GET; counts = true; } break;
              case 1: code = granted ? 9 : 10; if (granted) { act = KV_ACT_GET; counts = true; } break;
              case 2: code = 11; break;
              case 3: code = 12; break;
              case 4: code = 13; act = KV_ACT_SET; counts = true; break;
              case 5: code = 14; act = KV_ACT_SET; counts = true; break;
              default: code = 18; act = KV_ACT_GET; break;  // 17 WARMUP_READ
            }
            if (act != KV_ACT_NONE) {
              kv_hdr Hf;
              kv_hdr_load(Hf, ie);
              const kv_res fr = kv_apply<kv_dev_mem>(tb, bucket, Hf, act, fkey, fm + F::VAL, 0, blockIdx.x);
              if (act == KV_ACT_GET && fr.ok) st_u32(fm + F::VER, fr.ver);
              if (!fr.ok && counts) atomicAdd(&stats->missing_keys, 1ULL);
              __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");  // (the next foreign request may read what this one wrote)
            }
            fm[F::TYPE] = (uint8_t)code;
          }
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    if (t == 0) __hip_atomic_store(&pub[np], kvh_word(seq, all_ok, 0, -1, -1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  // ---- every piece: the coordinator's word, my grants, the siblings' words
  if (t == 0) H.pub[KSB_NPMAX] = spin_for(&pub[np]);
  __syncthreads();
  if (H.timeout) __builtin_trap();
  if (!((H.pub[KSB_NPMAX] >> 33) & 1ull)) {  // not in closed form: piece 0 takes the whole sub (nobody has touched it), the others nothing
    if (j != 0) return 0;
    if (t == 0) atomicAdd(&stats->late_requests, (unsigned long long)h);
    *src = 0; *off = d.y; *cnt = h;
    return 1;
  }
  const uint32_t solo_rem = solo ? H.nrem : 0u;  // (SOLO: the sub's other keys, compacted in ovf2, are mine too -- behind the row's)
  auto done = [&]() -> int {
    if (!solo_rem) return 0;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    *src = 1; *off = d.y; *cnt = solo_rem;
    return 1;
  };
  if (t < KVB_W) H.G[t] = kv_ld_agent(&sbx[(size_t)(first + j) * KSB_WORDS + 4 * KVB_W + t]);
  if (t < np) H.pub[t] = __hip_atomic_load(&pub[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  if (c == 0) return done();
  uint32_t nw_before = 0, nw_all = 0;
  int lw_before = -1, jw = -1;
  for (uint32_t k = 0; k < np; k++) {
    const unsigned long long w = H.pub[k];
    const uint32_t nwk = (uint32_t)(w >> 23) & 1023u;
    if (k < j) {
      nw_before += nwk;
      if ((w >> 22) & 1ull) lw_before = (int)((w >> 2) & 0xFFFFFu);
    }
    nw_all += nwk;
    if (nwk) jw = (int)k;
  }
  const uint32_t found = H.found, ver0 = H.ver0;
  uint32_t miss = 0;
  if (v && !foreign) {  // (a foreign request has its answer from the coordinator)
    const bool granted = (H.G[wave] >> lane) & 1ull;
    uint32_t code;
    bool get = false;
    switch (my_type) {
      case 0: code = granted ? 7 : 8; get = granted; break;
      case 1: code = granted ? 9 : 10; get = granted; break;
      case 2: code = 11; break;
      case 3: code = 12; break;
      case 4: code = 13; miss = !found; break;
      case 5: code = 14; miss = !found; break;
      default: code = 18; get = true; break;  // 17 WARMUP_READ
    }
    if (get && found) {  // the row as of my position: the last COMMIT before me (its message holds the value), else the row before the pass
      const int widx = lw_below >= 0 ? (int)H.idx[lw_below] : lw_before;
      if (widx >= 0) {
        const uint8_t *fm = rep + dint_view_off(V, (uint32_t)widx, F::MSG) + F::VAL;
        const uint32_t a0 = ld_u32(fm), a1 = ld_u32(fm + 4);
        st_u32(msg + F::VAL, a0); st_u32(msg + F::VAL + 4, a1);
      } else {
        st_u32(msg + F::VAL, H.rowv[0]); st_u32(msg + F::VAL + 4, H.rowv[1]);
      }
      st_u32(msg + F::VER, ver0 + nw_before + wr_below);
    } else if (get && my_type != 17u) {
      miss = 1;  // a granted ACQUIRE reads a row that is not there (the reference panics: smallbank/udp/kvs.h; counted)
    }
    msg[F::TYPE] = (uint8_t)code;
  }
  {
    const uint64_t mm = __ballot(miss != 0);
    if (lane == 0 && mm) atomicAdd(&stats->missing_keys, (unsigned long long)__popcll(mm));
  }
  // ---- row and version, once: by the piece that holds the pass's last COMMIT
  if (found && nw_all && (int)j == jw && (int)t == lw_tot) {
    uint8_t *row = kv_entry_ptr(tb, bucket, H.link) + KV_VAL_OFF + H.slot * F::VS;
    KV_ST(uint32_t, row, ld_u32(msg + F::VAL));
    KV_ST(uint32_t, row + 4, ld_u32(msg + F::VAL + 4));
    KV_ST(uint32_t, &kv_entry_hdr(tb, bucket, H.link)->ver[H.slot], ver0 + nw_all);
  }
  return done();
}

// One workgroup per CU: its 8 waves are 2 per SIMD (the second launch bound is waves per SIMD, not workgroups per CU) at
// the full 256 VGPRs, and its LDS (static_assert below) leaves no room for a second one.
static_assert(sizeof(kvb_lds) + KVB_BM_BYTES + sizeof(kv_dev) <= 160 * 1024, "k_kv_big must fit the 160 KB of LDS of a gfx950 CU");
static_assert(sizeof(kvb_lds) + (KV_HOT_BM ? KV_HOT_BM_W * 10 : 16) + sizeof(kv_dev) <= 160 * 1024, "k_kv_big must fit the 160 KB of LDS of a gfx950 CU");
// ---- k_kv_hot: the work items of a store / tatp pass that are hot keys in closed form -- pieces, remainders, solo subs.  A
// kernel of its own (r05) because it is LIGHT: the LDS of a k_kv_resolve workgroup and half the registers of k_kv_big, whose
// kv_big_bin (256 VGPRs, ~155 KB of LDS) needs an EMPTY compute unit -- with three shard servers side by side on the GPU
// its workgroups waited for the other servers' resolve workgroups to drain (k_kv_big: 36 us alone, 47 us in company, for
// items of 13 .. 25 us).  What the closed forms and the group phases do not cover is left to k_kv_big in `lateq`.
// A WORKER: takes the pass's work items by ticket, in list order, as they are listed -- by the resolve workgroups of the SAME
// launch (k_kv_pass, r06) or of the launch before (k_kv_hot, k_kv_hot_part).  Ticket i waits for item i's tag; it gives up when
// every coarse bin has listed (big[6] == C) and there are no more than i items.  Workers never hold anything a resolve
// workgroup waits for, and a launch places every resolve workgroup before its first worker (block order), so whoever a worker
// waits for is running; the siblings of a hot key's piece are listed together and sit side by side in the list, so whoever a
// piece waits for holds a ticket or draws the very next ones (kv_hot_item).
#define KVW_GRID 96u  // workers per engine of k_kv_pass by default (DINT_KV_WORKERS; a worker takes as many items as it gets)
template <int WL>
__device__ __forceinline__ static void kv_hot_role(const kv_pass_args &A, kv_dev &Skv, uint8_t *Lraw, uint32_t &Stk, uint32_t bx, uint32_t by) {
  const uint32_t t = threadIdx.x;
  if (t < sizeof(kv_dev) / 4) ((uint32_t *)&Skv)[t] = ((const uint32_t *)A.kv)[t];
  __syncthreads();
  kv_dev_pend_set(Skv, A.pno);  // (the ticket loop below starts with a barrier)
  kv_cut cut2 = A.cut;
  cut2.P = A.cut.P * KVR_F;
  uint64_t *tr = A.trace ? A.trace + 2048 * 32 + 32 * (size_t)(by * KVB_GRID + bx) : nullptr;
  for (bool first = true;; first = false) {
    __syncthreads();
    if (t == 0) {
      const uint32_t i = atomicAdd(&A.big[4], 1u);
      uint32_t got = KV_NONE;
      for (uint32_t spins = 0; i < DINT_KV_BIGQ_MAX; spins++) {
        if (__hip_atomic_load(&A.bigrdy[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == A.seq) { got = i; break; }
        if (__hip_atomic_load(&A.big[6], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= A.cut.P &&
            i >= __hip_atomic_load(&A.big[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;  // every bin has listed: there is no item i
        if (spins > (1u << 22)) __builtin_trap();  // (a hung GPU is worse than a trap)
        __builtin_amdgcn_s_sleep(4);
      }
      Stk = got;
    }
    __syncthreads();
    const uint32_t i = Stk;
    if (i == KV_NONE) break;
    const uint4 d = kv_ld_agent(A.bigq + KVQ_W * (size_t)i);
    uint32_t src = 0, off = d.y, cnt = d.z;
    uint64_t *ttr = first ? tr : nullptr;
    if (ttr && t == 0) { ttr[0] = __builtin_amdgcn_s_memrealtime(); ttr[2] = d.z; ttr[3] = d.x; ttr[30] = d.w; }
    int run = 1;
    if constexpr (WL == DINT_WL_SMALLBANK) {
      // smallbank (r06b): the pieces of a sub's row, its coordinator and the foreign requests -- kv_sb_item -- here, beside the
      // resolve workgroups; what is left of the sub (the other keys: kv_big_bin, 256 VGPRs and an empty compute unit) goes to the
      // list of the k_kv_big launch behind this kernel.  (That is the normal case, not a late one: the stats are kv_sb_item's.)
      if ((d.w & 3u) != KVQ_SUB)
        run = kv_sb_item<WL>(A.rep, cut2, &Skv, d, kv_ld_agent(A.bigq + KVQ_W * (size_t)i + 1), kv_ld_agent(A.bigq + KVQ_W * (size_t)i + 2), A.ovf, A.ovf2,
                             A.hotpub, A.sbx, A.seq, A.inv_n, A.stats, A.V, Lraw, &src, &off, &cnt);
      if (run && src == 1 && cnt <= KVB_T) {
        // a remainder of at most a workgroup's worth (a SOLO sub's other keys: cold rows, a few requests each), compacted in ovf2 by this
        // workgroup: the light way -- chunks of bucket groups through kv_chunk, as the resolve workgroups do -- instead of an
        // empty compute unit for kv_big_bin.  (false: a bucket group longer than a chunk -- kv_big_bin's)
        kvh_lds &Hh = *(kvh_lds *)Lraw;
        __syncthreads();
        Hh.rem[t] = t < cnt ? A.ovf2[off + t] : 0ull;
        if (kv_rem_chunks<WL>(A.rep, cut2, &Skv, d.x, Hh.rem, cnt, Hh, A.stats, A.force_flags & 1, A.V)) run = 0;
      }
      if (run && t == 0) A.lateq[atomicAdd(&A.big[5], 1u)] = make_uint4(d.x, off, cnt, src);
    } else {
    if ((d.w & 3u) == KVQ_SOLO)
      run = kv_solo_item<WL>(A.rep, cut2, &Skv, d, A.ovf, A.ovf2, A.stats, A.force_flags & 1, A.V, Lraw, &src, &off, &cnt, ttr);
    else if ((d.w & 3u) != KVQ_SUB)
      run = kv_hot_item<WL>(A.rep, cut2, &Skv, d, kv_ld_agent(A.bigq + KVQ_W * (size_t)i + 1), kv_ld_agent(A.bigq + KVQ_W * (size_t)i + 2), A.ovf, A.ovf2,
                            A.hotpub, A.seq, A.inv_n, A.stats, A.force_flags & 1, A.V, Lraw, &src, &off, &cnt, ttr);
    if (run && t == 0) {  // not in closed form: k_kv_late's (or k_kv_big's), behind this kernel
      A.lateq[atomicAdd(&A.big[5], 1u)] = make_uint4(d.x, off, cnt, src);
      atomicAdd(&A.stats->late_requests, (unsigned long long)cnt);
      atomicAdd(&A.stats->late_items[(d.w & 3u) == KVQ_SUB ? 0 : (d.w & 3u) == KVQ_SOLO ? 1 : 2], 1ULL);
    }
    }
    if (ttr && t == 0) ttr[1] = __builtin_amdgcn_s_memrealtime();
  }
}
template <int WL>
__global__ void __launch_bounds__(KVB_T, 4) k_kv_hot(kv_multi_args M) {
  __shared__ kv_dev Skv;
  __shared__ __attribute__((aligned(16))) uint8_t Lraw[sizeof(kvh_lds)];
  __shared__ uint32_t Stk;
  const kv_pass_args &A = M.e[blockIdx.y];
  if (blockIdx.x >= A.big[3]) return;  // (behind the resolve kernel: the list is final)
  kv_hot_role<WL>(A, Skv, Lraw, Stk, blockIdx.x, blockIdx.y);
}

// ---- k_kv_hot_part (r06): the hot keys of pass k AND the partition of pass k + 1 in one launch.  The partition touches no
// table (it reads the next batch, writes its records into the coarse bins of the OTHER set, its log records behind the tail
// the previous partition left, and the control words of the next set), the hot keys are ~150 work items of 13 .. 25 us on a
// mostly idle GPU: side by side the pair costs what the longer one costs.  One engine per launch; workgroups 0 ..
// KVB_GRID-1 take the hot items (dispatched first: they are the long pole), the others a tile each of KVB_T x RPT
// requests.  The pool rotation of pass k + 1 runs here beside pass k's frees and pops: two pend sets, kv_pool_rotate.
// (DINT_KV_NO_FUSE=1: since k_kv_pass the default pass is ONE launch -- resolve, hot keys and the next partition together.)
template <int WL, int RPT>
__global__ void __launch_bounds__(KVB_T, 4) k_kv_hot_part(kv_pass_args H, kv_pass_args P) {
  constexpr size_t LB = sizeof(kvh_lds) > sizeof(kv_part_lds<RPT, KVB_T>) ? sizeof(kvh_lds) : sizeof(kv_part_lds<RPT, KVB_T>);
  __shared__ kv_dev Skv;
  __shared__ __attribute__((aligned(16))) uint8_t Lraw[LB];
  __shared__ uint32_t Stk;
  if (blockIdx.x < KVB_GRID) {
    if (blockIdx.x < H.big[3]) kv_hot_role<WL>(H, Skv, Lraw, Stk, blockIdx.x, 0);
    return;
  }
  const uint32_t b = blockIdx.x - KVB_GRID;
  if (b >= P.n_tiles) return;
  kv_part_body<WL, RPT, KVB_T>(P, *(kv_part_lds<RPT, KVB_T> *)Lraw, b == 0);
}

// ---- k_kv_pass (r06): ONE launch per pass -- the resolve stage of pass k, its hot keys, and the partition of pass k + 1.
//   blocks [0, sum C)                 one coarse bin each (kv_resolve_role): lists its big subs as work items BEFORE its chunks
//   blocks [sum C, + KVW_GRID * n_eng) workers (kv_hot_role): take the items as they are listed, while the chunks run
//   blocks behind them                (one engine, look-ahead) a tile each of the NEXT pass's partition (kv_part_body)
// r05's pass was four launches (part, resolve, hot, big) and a launch boundary between the engines' kernels costs its stream
// 5 .. 15 us; the hot keys (~150 items of 13 .. 25 us) ran on an idle GPU behind the resolve kernel, the partition (bandwidth)
// behind them.  Here the three overlap: the pass's chain is the resolve workgroups plus the tail of the items listed last.
// The hand-over resolve -> worker crosses XCDs: agent-scope stores and loads for records, items and tags (kv_st_agent), no L2
// write-back.  The next partition beside this resolve needs its own coarse bins and control words: dint_kv_sets.
// RPT = 0: no partition role (plain dint_submit_device calls, launch sets of several engines).
template <int WL, int RPT>
__global__ void __launch_bounds__(KVB_T, 4) k_kv_pass(kv_multi_args M, uint32_t n_eng, uint32_t sum_c, kv_multi_args PM /* the engines' NEXT passes */,
                                                       uint32_t max_tiles /* tiles of the longest of them (0: none announced) */,
                                                       uint32_t n_work /* workers per engine */,
                                                       uint32_t part_first /* the partition's tiles are placed before the workers */) {
  constexpr size_t LW = sizeof(kvs_lds) > sizeof(kvh_lds) ? sizeof(kvs_lds) : sizeof(kvh_lds);  // a worker's (smallbank: kv_sb_item's, then kv_rem_chunks')
  constexpr size_t L1 = LW > sizeof(kvr_lds) ? LW : sizeof(kvr_lds);
  constexpr size_t LB = L1 > sizeof(kv_part_lds<RPT ? RPT : 1, KVB_T>) ? L1 : sizeof(kv_part_lds<RPT ? RPT : 1, KVB_T>);
  __shared__ kv_dev Skv;
  __shared__ __attribute__((aligned(16))) uint8_t Lraw[LB];
  __shared__ uint2 Sbig[KVR_F];
  __shared__ uint32_t Stk;
  uint32_t b = blockIdx.x;
  if (b < sum_c) {
    uint32_t e = 0;
    while (e + 1 < n_eng && b >= M.e[e].cut.P) { b -= M.e[e].cut.P; e++; }
    kv_resolve_role<WL>(M.e[e], b, Skv, Lraw, Sbig);
    return;
  }
  b -= sum_c;
  const uint32_t nw = n_work * n_eng, nt = RPT != 0 ? max_tiles * n_eng : 0u;
  const bool worker = part_first ? b >= nt : b < nw;
  if (worker) {
    if (part_first) b -= nt;
    if (b >= nw) return;
    const uint32_t e = b / n_work;
    kv_hot_role<WL>(M.e[e], Skv, Lraw, Stk, b - e * n_work, e);
    return;
  }
  if constexpr (RPT != 0) {
    if (!part_first) b -= nw;
    if (b >= nt) return;
    const uint32_t e = b / max_tiles, slot = b - e * max_tiles;  // (tiles draw tickets: only how many blocks an engine gets matters)
    const kv_pass_args &P = PM.e[e];
    if (slot >= P.n_tiles) return;
    kv_part_body<WL, RPT, KVB_T>(P, *(kv_part_lds<RPT, KVB_T> *)Lraw, slot == 0);
  }
}

// (r05 tried 128 VGPRs for store / tatp, where the kernel is usually empty, so that a launch need not wait for empty compute
// units: kv_big_bin then spills 640 bytes per lane, and a dispatch with that much scratch stalls its queue while the runtime
// resizes it -- k_kv_hot + k_kv_big went from 40 to 62 us.  A SMALL GRID instead: behind k_kv_hot only 8 workgroups per engine
// have to find an empty compute unit.)
// (256 VGPRs x 8 waves: one workgroup per compute unit whatever the LDS says; the stretch machinery still spills a few dozen
// registers -- profiles/r05_kernel_resources.txt has the numbers of the build)
template <int WL>
__global__ void __launch_bounds__(KVB_T, 1) k_kv_big(kv_multi_args M, uint32_t from_late) {
  static_assert(sizeof(kv_dev) + sizeof(kvb_lds) + (WL == DINT_WL_SMALLBANK ? KVB_BM_BYTES : 16) + 64 <= 160 * 1024, "k_kv_big's LDS must fit a gfx950 compute unit");
  __shared__ kv_dev Skv;
  __shared__ __attribute__((aligned(16))) uint8_t Lraw[sizeof(kvb_lds)];
  __shared__ __attribute__((aligned(16))) uint8_t Lbm[WL == DINT_WL_SMALLBANK ? KVB_BM_BYTES : KV_HOT_BM ? KV_HOT_BM_W * 10 : 16];  // (one workgroup per CU either way: 8 waves of 256 VGPRs)
  __shared__ uint32_t Stk;
  const kv_pass_args &A = M.e[blockIdx.y];
  const uint32_t nq = from_late ? A.big[5] : A.big[3];
  if (blockIdx.x >= nq) return;
  const uint32_t t = threadIdx.x;
  if (t < sizeof(kv_dev) / 4) ((uint32_t *)&Skv)[t] = ((const uint32_t *)A.kv)[t];
  __syncthreads();
  kv_dev_pend_set(Skv, A.pno);
  __syncthreads();
  kv_cut cut2 = A.cut;
  cut2.P = A.cut.P * KVR_F;
  if (from_late) {  // behind k_kv_hot: what it left (usually nothing)
    for (uint32_t i = blockIdx.x; i < nq; i += gridDim.x) {
      const uint4 d = A.lateq[i];
      const uint64_t *recs = (d.w ? (const uint64_t *)A.ovf2 : A.ovf) + d.y;
      kv_big_bin<WL>(A.rep, A.n, cut2, &Skv, d.x, recs, d.w || !A.ovf2 ? nullptr : A.ovf2 + d.y, d.z, A.stats, A.force_flags, A.V, Lraw, Lbm, nullptr);
    }
    return;
  }
  // DINT_KV_TRACE: 32 words per workgroup -- {in, out, records, sub} of the first sub it takes, then kv_big_bin's stamps
  uint64_t *tr = A.trace ? A.trace + 2048 * 32 + 32 * (size_t)(blockIdx.y * KVB_GRID + blockIdx.x) : nullptr;
  // work items by ticket, in list order: the pieces of a hot key sit side by side in the list, so a piece that waits for its
  // siblings waits for workgroups that hold a ticket already or will draw the next ones (kv_hot_item)
  for (bool first = true;; first = false) {
    __syncthreads();
    if (t == 0) Stk = atomicAdd(&A.big[4], 1u);
    __syncthreads();
    const uint32_t i = Stk;
    if (i >= nq) break;
    const uint4 d = A.bigq[KVQ_W * (size_t)i];
    uint32_t src = 0, off = d.y, cnt = d.z;
    uint64_t *ttr = first ? tr : nullptr;
    if (ttr && t == 0) { ttr[0] = __builtin_amdgcn_s_memrealtime(); ttr[2] = d.z; ttr[3] = d.x; ttr[30] = d.w; }
    const uint64_t t_in = tr ? __builtin_amdgcn_s_memrealtime() : 0ull;  // [31]: the LONGEST item of the workgroup: ticks << 40 | records << 20 | went on to kv_big_bin << 18 | kind word
    int run = 1;
    if ((d.w & 3u) == KVQ_SOLO && WL != DINT_WL_SMALLBANK)
      run = kv_solo_item<WL>(A.rep, cut2, &Skv, d, A.ovf, A.ovf2, A.stats, A.force_flags & 1, A.V, Lraw, &src, &off, &cnt, ttr);
    else if ((d.w & 3u) != KVQ_SUB && WL == DINT_WL_SMALLBANK)
      run = kv_sb_item<WL>(A.rep, cut2, &Skv, d, A.bigq[KVQ_W * (size_t)i + 1], A.bigq[KVQ_W * (size_t)i + 2], A.ovf, A.ovf2, A.hotpub, A.sbx, A.seq,
                           A.inv_n, A.stats, A.V, Lraw, &src, &off, &cnt);
    else if ((d.w & 3u) != KVQ_SUB)
      run = kv_hot_item<WL>(A.rep, cut2, &Skv, d, A.bigq[KVQ_W * (size_t)i + 1], A.bigq[KVQ_W * (size_t)i + 2], A.ovf, A.ovf2, A.hotpub, A.seq,
                            A.inv_n, A.stats, A.force_flags & 1, A.V, Lraw, &src, &off, &cnt, ttr);
    if (run) {
      // (a remainder lives in ovf2: no second copy of the scratch to regroup it by stretch -- it is small)
      const uint64_t *recs = (src ? (const uint64_t *)A.ovf2 : A.ovf) + off;
      kv_big_bin<WL>(A.rep, A.n, cut2, &Skv, d.x, recs, src || !A.ovf2 ? nullptr : A.ovf2 + off, cnt, A.stats, A.force_flags, A.V, Lraw, Lbm, ttr);
    }
    if (ttr && t == 0) ttr[1] = __builtin_amdgcn_s_memrealtime();
    if (tr && t == 0) {
      const uint64_t dt = __builtin_amdgcn_s_memrealtime() - t_in;
      if (dt > (tr[31] >> 40)) tr[31] = (dt << 40) | ((uint64_t)(d.z & 0xFFFFFu) << 20) | (d.w & 0x3FFFFu) | (run ? 1ull << 18 : 0ull);
    }
  }
}

// Workgroups of the kernel that answers a hot key's PIECES that the device can hold at once (the pieces of a key wait for each
// other: kv_hot_item, kv_sb_item).  dint_kv_create switches the pieces off where that is fewer than twice the siblings a key can
// have -- a CU mask, a partitioned or shared GPU: the hot keys then go through kv_big_bin, one workgroup each.
template <int WL>
int kv_piece_residency(int device) {
  int per_cu = 0, cus = 0;
  hipError_t e1 = WL == DINT_WL_SMALLBANK
                      ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_kv_big<WL>, (int)KVB_T, 0)
                      : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_kv_pass<WL, 0>, (int)KVB_T, 0);
  hipError_t e2 = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device);
  if (e1 != hipSuccess || e2 != hipSuccess) { (void)hipGetLastError(); return 0; }
  return per_cu * cus;
}

// ---- launch of one pass set (all engines of `M` side by side) -------------------------------------------------------------------
// `next` (one engine only): the partition of the engine's next pass, launched with this pass's hot keys (k_kv_hot_part)
template <int WL>
void launch_kv_passes(kv_multi_args &M, uint32_t n_eng, uint32_t rpt, hipStream_t st, hipEvent_t *ev, const dint_kv_knobs &K,
                      bool part_done, const kv_multi_args *next /* every engine's next pass, or nullptr */, uint32_t next_rpt) {
  uint32_t max_tiles = 0, sum_c = 0;
  for (uint32_t k = 0; k < n_eng; k++) {
    max_tiles = std::max(max_tiles, M.e[k].n_tiles);
    sum_c += M.e[k].cut.P;
  }
  if (ev) hipEventRecord(ev[0], st);
  if (part_done) {}  // (ran inside the previous pass's k_kv_hot_part)
  else if (rpt == 1) hipLaunchKernelGGL((k_kv_part<WL, 1>), dim3(max_tiles, n_eng), dim3(KV_TB), 0, st, M);
  else if (rpt == 2) hipLaunchKernelGGL((k_kv_part<WL, 2>), dim3(max_tiles, n_eng), dim3(KV_TB), 0, st, M);
  else hipLaunchKernelGGL((k_kv_part<WL, 4>), dim3(max_tiles, n_eng), dim3(KV_TB), 0, st, M);
  if (ev) hipEventRecord(ev[1], st);
  // the hot keys.  store / tatp: workers in the resolve launch (k_kv_pass; DINT_KV_NO_FUSE: k_kv_hot / k_kv_hot_part behind
  // k_kv_resolve, r05 / early r06), then k_kv_late for what the closed forms left -- usually nothing;
  // smallbank: workers in the resolve launch too (kv_sb_item), then k_kv_big for the remainders; DINT_KV_NO_SPLIT / DINT_KV_ONE_BIG_KERNEL: k_kv_big alone
  const bool hot = WL != DINT_WL_SMALLBANK && M.e[0].split_min != 0xFFFFFFFFu && !K.one_big_kernel;  // (store / tatp: k_kv_hot / k_kv_late exist)
  bool fused = false, one = false;
  uint32_t sb_workers = 0;
  if constexpr (WL != DINT_WL_SMALLBANK) {
    if (hot && !K.no_fuse) {
      uint32_t nmax = 0;
      for (uint32_t k = 0; next && k < n_eng; k++) nmax = std::max(nmax, next->e[k].n_tiles);
      const dim3 g(sum_c + K.workers * n_eng + nmax * n_eng);
      if (!next) hipLaunchKernelGGL((k_kv_pass<WL, 0>), g, dim3(KVB_T), 0, st, M, n_eng, sum_c, M, 0u, K.workers, K.part_first);
      else if (next_rpt == 2) hipLaunchKernelGGL((k_kv_pass<WL, 2>), g, dim3(KVB_T), 0, st, M, n_eng, sum_c, *next, nmax, K.workers, K.part_first);
      else hipLaunchKernelGGL((k_kv_pass<WL, 4>), g, dim3(KVB_T), 0, st, M, n_eng, sum_c, *next, nmax, K.workers, K.part_first);
      one = fused = true;
    }
  } else {
    // smallbank with the next batch announced: its partition beside this pass's resolve workgroups (k_kv_pass without workers --
    // the big subs and the pieces of the hot accounts are k_kv_big's, behind this launch)
    // (r06b: with workers -- the pieces of the subs' rows beside the resolve workgroups, the remainders in the k_kv_big launch behind;
    // DINT_KV_SB_WORKERS=0: r06a's k_kv_pass without workers, every item in k_kv_big)
    sb_workers = !K.no_fuse && !K.one_big_kernel && M.e[0].sb_pieces ? K.sb_workers : 0u;
    if ((next || sb_workers) && !K.no_fuse) {
      uint32_t nmax = 0;
      for (uint32_t k = 0; next && k < n_eng; k++) nmax = std::max(nmax, next->e[k].n_tiles);
      const dim3 g(sum_c + sb_workers * n_eng + nmax * n_eng);
      if (!next) hipLaunchKernelGGL((k_kv_pass<WL, 0>), g, dim3(KVB_T), 0, st, M, n_eng, sum_c, M, 0u, sb_workers, 0u);
      else if (next_rpt == 2) hipLaunchKernelGGL((k_kv_pass<WL, 2>), g, dim3(KVB_T), 0, st, M, n_eng, sum_c, *next, nmax, sb_workers, K.part_first);
      else hipLaunchKernelGGL((k_kv_pass<WL, 4>), g, dim3(KVB_T), 0, st, M, n_eng, sum_c, *next, nmax, sb_workers, K.part_first);
      one = true;
    }
  }
  if (!one) hipLaunchKernelGGL((k_kv_resolve<WL>), dim3(sum_c), dim3(KVB_T), 0, st, M, n_eng);
  const bool ev3 = ev && one && WL != DINT_WL_SMALLBANK;  // (store / tatp, one launch: three timed intervals -- part, pass, late -- not four)
  if (ev) hipEventRecord(ev[2], st);
  if constexpr (WL != DINT_WL_SMALLBANK) {
    if (hot && next && !one && n_eng == 1) {
      const dim3 g(KVB_GRID + next->e[0].n_tiles);
      if (next_rpt == 2) hipLaunchKernelGGL((k_kv_hot_part<WL, 2>), g, dim3(KVB_T), 0, st, M.e[0], next->e[0]);
      else hipLaunchKernelGGL((k_kv_hot_part<WL, 4>), g, dim3(KVB_T), 0, st, M.e[0], next->e[0]);
      fused = true;
    }
  }
  if (hot && !fused) hipLaunchKernelGGL((k_kv_hot<WL>), dim3(KVB_GRID, n_eng), dim3(KVB_T), 0, st, M);
  if (ev && !ev3) hipEventRecord(ev[3], st);
  // (behind k_kv_hot the launch is almost always empty, and k_kv_big's workgroups -- 256 VGPRs x 8 waves -- each wait for a compute
  // unit with nothing on it: 14 .. 19 us per pass beside the other servers' kernels, whether 8 of them or one.  k_kv_late has
  // k_kv_hot's footprint and starts beside anything; DINT_KV_LATE_BIG=1 keeps r05's launch for A/B runs and cross-checks)
  if constexpr (WL != DINT_WL_SMALLBANK) {
    if (hot && K.exp_no_late == 1) {}  // (DINT_EXP_NO_LATE=1: timing experiments on streams that leave nothing -- what the third launch costs)
    else if (hot && !K.late_big) {
      for (uint32_t r = 0; r < std::max(1u, K.exp_no_late >> 1); r++)  // (DINT_EXP_NO_LATE=2k: k launches -- the second finds its list done)
        if (K.late_fat) hipLaunchKernelGGL((k_kv_late<WL, 2>), dim3(KVL_GRID, n_eng), dim3(KVB_T), 0, st, M, r);
        else hipLaunchKernelGGL((k_kv_late<WL, 4>), dim3(KVL_GRID, n_eng), dim3(KVB_T), 0, st, M, r);
    }
    else hipLaunchKernelGGL((k_kv_big<WL>), dim3(hot ? K.late_grid : KVB_GRID, n_eng), dim3(KVB_T), 0, st, M, hot ? 1u : 0u);
  } else {
    // (behind the workers what is left is a dozen remainders: a SMALL grid -- every workgroup of this kernel waits for an empty compute unit)
    hipLaunchKernelGGL((k_kv_big<WL>), dim3(sb_workers ? K.sb_late_grid : KVB_GRID, n_eng), dim3(KVB_T), 0, st, M, sb_workers ? 1u : 0u);
  }
  if (ev) hipEventRecord(ev3 ? ev[3] : ev[4], st);
}

