// k_kv.hip -- store / tatp / smallbank shard servers on gfx950.
//
// Reference semantics (serial, one message at a time):
//   store/udp/server.cc:75-97            READ / SET on one kvs  (INSERT as store/ebpf/store_kern.c:226-297)
//   tatp/udp/server_shard.cc:113-210     13 request types over 5 kvs tables + txn_locks[5][] + log ring
//   smallbank/udp/server_shard.cc:107-189  9 request types over 2 tables + num_ex/num_sh counters + log ring
// Every request touches exactly one bucket of one table (lock_hash % hash_size == kvs bucket), or only
// the log ring.  Requests on different buckets commute; requests on one bucket apply in request order.
//
// One pass (n <= 65,536 requests) = up to three kernels:
//   k_kv_prepass (tatp / smallbank): per-block count of log requests, publishes the ring tail.
//   k_kv_scatter : one thread per request -- copy the message to the reply array, classify, hash, and
//                  either append the canonical 64-byte log record at ring position
//                  tail + (#log requests below i)   [deterministic: an exclusive scan, not an atomic], or
//                  append a {bucket group, idx, type, table|quadrant} record to bin = group & (P-1).
//   k_kv_resolve : one wave per bin.  A bin of <= 64 records (the common case: ~32 per bin) is sorted by
//                  (bucket group, idx) in registers and handled as one chunk.  Larger bins (hot keys) restore
//                  request order with a bitmap rank over idx, group each window of 512 records by bucket in
//                  an LDS hash and walk it 64 at a time.  Inside a chunk, lanes whose bucket is unique run at
//                  once; several requests on ONE key are resolved in closed form (ballots: version = v0 +
//                  #writers below, value = message of the last writer below, lock = last lock op below); any
//                  other same-bucket group runs in rounds (k-th request in round k, workgroup fence between
//                  rounds), so every request sees the table exactly as the serial reference would.  The table
//                  is the HBM layout of dint_kv_core.h: one 64-byte header sector answers the probe.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/dint_abi.h"
#include "dint_kv.h"

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

dint_kv_fmt dint_kv_format(uint32_t workload) {
  switch (workload) {
    case DINT_WL_STORE: return {53, 0, 0xFFFFFFFFu, 1, 9, 49, 40};
    case DINT_WL_TATP: return {55, 1, 2, 3, 11, 51, 40};
    default: return {23, 1, 2, 3, 11, 19, 8};
  }
}

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
  return type <= 5 ? 1 : (type == 6 ? 2 : 0);
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

// ---- k_kv_prepass ------------------------------------------------------------------------------------
template <int WL>
__global__ void __launch_bounds__(256)
k_kv_prepass(const uint8_t *__restrict__ req, uint32_t n, const kv_dev *__restrict__ kv, uint32_t *__restrict__ blk_cnt,
             uint32_t *tail) {
  using F = Fmt<WL>;
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (blockIdx.x == 0 && threadIdx.x == 0) tail[0] = tail[1];
  bool is_log = false;
  if (i < n) {
    const uint8_t *m = req + (size_t)i * F::MSG;
    is_log = kv_class<WL>(m[F::TYPE], 0) == 2 && m[F::TABLE] < kv->n_tables;
  }
  const uint32_t cnt = __syncthreads_count(is_log);
  if (threadIdx.x == 0) blk_cnt[blockIdx.x] = cnt;
}

// ---- k_kv_scatter ------------------------------------------------------------------------------------
template <int WL>
__global__ void __launch_bounds__(256)
k_kv_scatter(const uint8_t *__restrict__ req, uint8_t *rep, uint32_t n, const kv_dev *__restrict__ kv, dint_log log,
             const uint32_t *__restrict__ blk_cnt, uint32_t pmask, uint32_t *__restrict__ bin_cnt,
             uint64_t *__restrict__ bins, dint_dev_stats *__restrict__ stats, int load_mode) {
  using F = Fmt<WL>;
  __shared__ uint32_t red[4];
  __shared__ uint32_t wcnt[4];
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;

  if (blockIdx.x == 0 && threadIdx.x < KV_NLISTS)  // entries freed by earlier passes become reusable
    for (uint32_t t = 0; t < kv->n_tables; t++) kv_pool_rotate<kv_dev_mem>(kv->tab[t], threadIdx.x);

  // copy this block's messages to the reply array (replies are the request mutated in place).  All loads of a
  // thread are issued before its first store: one memory round trip for the 256 * MSG <= 16 KiB of the block.
  if (rep != req) {
    const size_t lo = (size_t)blockIdx.x * 256u * F::MSG;
    const size_t hi = min((size_t)n * F::MSG, lo + (size_t)256u * F::MSG);
    if ((((uintptr_t)req | (uintptr_t)rep) & 15) == 0) {
      const uint32_t nv = (uint32_t)((hi - lo) / 16);  // <= 880 vectors
      const uint4 *s = (const uint4 *)(req + lo);
      uint4 *d = (uint4 *)(rep + lo);
      // unconditional loads at clamped indices (branch-free, so the four loads stay in flight together; nv >= 1)
      const uint32_t a0 = min(threadIdx.x, nv - 1), a1 = min(threadIdx.x + 256u, nv - 1);
      const uint32_t a2 = min(threadIdx.x + 512u, nv - 1), a3 = min(threadIdx.x + 768u, nv - 1);
      const uint4 v0 = s[a0], v1 = s[a1], v2 = s[a2], v3 = s[a3];
      // ... and unconditional stores: a clamped lane rewrites vector nv-1 with the same bytes
      d[a0] = v0; d[a1] = v1; d[a2] = v2; d[a3] = v3;
      for (size_t k = lo + (size_t)nv * 16 + threadIdx.x; k < hi; k += 256) rep[k] = req[k];
    } else {
      for (size_t k = lo + threadIdx.x; k < hi; k += 256) rep[k] = req[k];
    }
  }

  uint32_t type = 0, table = 0, cls = 0;
  uint64_t key = 0;
  const uint8_t *m = req + (size_t)i * F::MSG;
  if (i < n) {
    type = m[F::TYPE];
    table = F::HAS_TABLE ? m[F::TABLE] : 0;
    cls = kv_class<WL>(type, load_mode);
    if (table >= kv->n_tables) cls = 0;  // the reference indexes tables[] out of bounds
    if (cls) key = ld_u64(m + F::KEY);
    else atomicAdd(&stats->bad_requests, 1ULL);
  }

  // ---- log requests: ring position = tail + exclusive count of log requests below i ----
  if (WL != DINT_WL_STORE) {
    uint32_t part = (threadIdx.x < blockIdx.x) ? blk_cnt[threadIdx.x] : 0, tot;
    wave_excl_scan_u32(part, &tot);
    if (lane == 0) red[wv] = tot;
    const uint64_t lm = __ballot(cls == 2);
    if (lane == 0) wcnt[wv] = (uint32_t)__popcll(lm);
    __syncthreads();
    uint32_t base = red[0] + red[1] + red[2] + red[3];
    for (uint32_t w = 0; w < wv; w++) base += wcnt[w];
    const uint32_t pos_in_batch = base + (uint32_t)__popcll(lm & lanemask_lt());
    if (cls == 2) {
      const uint32_t pos = (uint32_t)(((uint64_t)log.tail[0] + pos_in_batch) % log.cap);
      uint8_t *e = log.ring + (size_t)pos * 64;
      const uint32_t ver = ld_u32(m + F::VER);
      uint8_t *r = rep + (size_t)i * F::MSG;
      if (WL == DINT_WL_TATP && type == 24) {  // kDeleteLog: no val copy  (server_shard.cc:196-207)
        *(uint64_t *)e = key;
        *(uint2 *)(e + 48) = make_uint2(ver, 1u | (table << 8));
        r[F::TYPE] = 27;
      } else {  // kCommitLog  (tatp server_shard.cc:182-194, smallbank server_shard.cc:175-186)
        uint32_t w[16];
        __builtin_memcpy(&w[0], &key, 8);
#pragma unroll
        for (uint32_t k = 0; k < F::VS / 4; k++) w[2 + k] = ld_u32(m + F::VAL + 4 * k);
        uint4 *e4 = (uint4 *)e;
        e4[0] = make_uint4(w[0], w[1], w[2], w[3]);
        if (F::VS == 40) {
          e4[1] = make_uint4(w[4], w[5], w[6], w[7]);
          e4[2] = make_uint4(w[8], w[9], w[10], w[11]);
        }
        *(uint2 *)(e + 48) = make_uint2(ver, table << 8);
        r[F::TYPE] = (WL == DINT_WL_TATP) ? 17 : 15;
      }
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 255) {
      const uint32_t total = pos_in_batch + (cls == 2 ? 1u : 0u);
      log.tail[1] = (uint32_t)(((uint64_t)log.tail[0] + total) % log.cap);
    }
  }

  // ---- table requests: record -> bin ----
  if (cls == 1) {
    const uint64_t h = dint_hash_key(key);
    const uint64_t g = dint_fastmod(h, kv->mod[table]);
    uint32_t local = (uint32_t)g;
    if (kv->shard_count > 1) {
      if ((uint32_t)(g % kv->shard_count) != kv->shard_index) {
        if (!load_mode) atomicAdd(&stats->foreign_requests, 1ULL);
        return;
      }
      local = (uint32_t)(g / kv->shard_count);
    }
    // lock quadrant: lock_hash / hash_size, lock_hash = h % (4 * hash_size)
    const uint64_t hs = kv->mod[table].d, dq = dint_fastmod(h, kv->lockmod[table]) - g;  // 0, hs, 2hs or 3hs
    const uint32_t q = dq >= 2 * hs ? (dq >= 3 * hs ? 3u : 2u) : (dq >= hs ? 1u : 0u);
    const uint32_t gk = kv->gk_base[table] + local;
    const uint32_t bin = gk & pmask;
    const uint32_t pos = atomicAdd(&bin_cnt[bin], 1u);
    bins[(size_t)bin * DINT_MICRO + pos] = dint_rec(gk, i, type, table | (q << 4));
  }
}

// ---- optional per-wave timeline (DINT_KV_TRACE=1): lane 0 of every resolve wave stamps s_memtime at fixed
// points into trace[bin * 16 + k]; with tracing on, each stamp first drains the wave's memory queue so the
// difference of two stamps is the latency of what lies between them.  Off (nullptr) in normal runs.
__device__ static inline void kv_stamp(uint64_t *tr, uint32_t k) {
  if (tr) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    if (lane_id() == 0) tr[k] = __builtin_amdgcn_s_memtime();
  }
}
// device-wide constant-rate clock (100 MHz), comparable across waves: [10] = wave start, [11] = wave end
__device__ static inline void kv_stamp_real(uint64_t *tr, uint32_t k) {
  if (tr && lane_id() == 0) tr[k] = __builtin_amdgcn_s_memrealtime();
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
                                            const kv_dev *kv, dint_dev_stats *__restrict__ stats, uint64_t *tr = nullptr) {
  using F = Fmt<WL>;
  const kv_tab t = kv->tab[table];
  uint8_t *ie = kv_entry_ptr(t, bucket, KV_INLINE);
  // ---- load phase
  kv_hdr H;
  kv_hdr_copy(H, *(const kv_hdr *)ie);
  uint2 cnt = make_uint2(0, 0);
  if (WL == DINT_WL_SMALLBANK) cnt = *(const uint2 *)(ie + KV_SB_LOCK_OFF + 8 * q);  // {num_ex, num_sh}
  const uint64_t key = ld_u64(msg + F::KEY);
  uint8_t *val = msg + F::VAL;
  if (tr) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); tr[12] = __builtin_amdgcn_s_memtime(); tr[14] = type; }

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
      default: act = KV_ACT_INS; code = 8; break;  // kInsert (eBPF store)
    }
  } else if (WL == DINT_WL_TATP) {
    const uint32_t lk = (H.lockw >> (8 * q)) & 0xFFu;
    switch (type) {
      case 0: act = KV_ACT_GET; break;                                                    // kRead  server_shard.cc:116-121
      case 1: if (lk == 0) { lock_store = 1; code = 7; } else code = 8; break;           // kAcquireLock  :123-132
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
      default: act = KV_ACT_SET; miss_counts = true; code = 14; break;  // 5 kCommitBck
    }
  }

  // ---- act phase
  const kv_res r = kv_apply<kv_dev_mem>(t, bucket, H, act, key, val, ins_ver, blockIdx.x);
  if (tr) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); tr[13] = __builtin_amdgcn_s_memtime(); }
  if (WL == DINT_WL_TATP && lock_store >= 0) ie[KV_LOCKB_OFF + q] = (uint8_t)lock_store;
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
// Precondition: the chunk's lanes are sorted by (bucket group, idx), so the requests of one bucket sit in
// adjacent lanes, in request order (valid lanes first).  Such a run is a SEGMENT; segments commute.
//
// A segment is "simple" when all its requests address ONE key with ops that never change the chain (no INSERT /
// DELETE): READ, SET / COMMIT_*, lock ops.  Its serial outcome then depends on a few words of state -- row found?,
// version, last writer, lock word -- and ALL simple segments of the chunk (single requests included) are
// resolved together:
//   1. every segment head loads its bucket's inline header (and smallbank counters) and locates the row;
//   2. every lane derives its own reply from ballots restricted to its segment's lane mask (store / tatp):
//        version seen = ver0 + #writers below in the segment, value seen = message of the last writer below,
//        lock seen    = what the last ACQUIRE (-> 1) / ABORT / COMMIT_PRIM (-> 0) below wrote, else the stored byte;
//      smallbank's shared / exclusive counters have no closed form: single requests apply their op directly,
//      longer segments are walked once each with wave-uniform registers (no memory inside the walk);
//   3. replies are written, reads copy their value from the last writer's message or from the table row;
//   4. after a fence the segment heads write the final row / version / lock word once.
// Four memory round trips per chunk however many requests collide.  Any other segment (several keys of one
// bucket, inserts, deletes) runs in rounds: its k-th request executes in round k through kv_do_request.
// Semantics per op: the same reference lines as kv_do_request.
template <int WL>
__device__ static inline bool kv_simple_op(uint32_t type) {
  if (WL == DINT_WL_STORE) return type <= 1;                                   // READ, SET
  if (WL == DINT_WL_TATP) return type <= 2 || type == 12 || type == 13;        // READ, ACQUIRE, ABORT, COMMIT_PRIM/BCK
  return type <= 5;                                                            // every smallbank table op
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

template <int WL>
__device__ static inline void kv_chunk(uint8_t *rep, bool valid, uint32_t idx, uint32_t gk, uint32_t type, uint32_t table,
                                       uint32_t q, const kv_dev *kv, dint_dev_stats *__restrict__ stats, int force_rounds,
                                       bool last_chunk, uint64_t *tr = nullptr) {
  using F = Fmt<WL>;
  const int lane = (int)lane_id();
  const uint64_t lt = lanemask_lt(), le = lt | (1ull << lane);
  uint8_t *msg = rep + (size_t)idx * F::MSG;
  const uint64_t key = valid ? ld_u64(msg + F::KEY) : 0;
  const uint64_t bucket = valid ? (uint64_t)(gk - kv->gk_base[table]) : 0;

  // ---- segments
  const uint32_t gk_up = __shfl_up(gk, 1, 64);
  const bool head = valid && (lane == 0 || gk_up != gk);
  const uint64_t hm = __ballot(head);
  const uint64_t vm = __ballot(valid);
  const int hl = valid ? 63 - __clzll(hm & le) : lane;                       // my segment's head lane
  // lanes of my segment = [hl, next head) ; valid lanes are lanes 0..nvalid-1, so without a head above me the
  // segment ends at the first invalid lane: bit nvalid = vm + 1 (0 when all 64 lanes are valid -> mask of all ones)
  const uint64_t above = hm & ~le;
  const uint64_t next = above ? (above & (~above + 1ull)) : vm + 1ull;
  const uint64_t seg = valid ? ((next - 1ull) & ~((1ull << hl) - 1ull)) : 0;
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
    kv_hdr_copy(H, *(const kv_hdr *)ie);
    if (WL == DINT_WL_SMALLBANK) {
      const uint2 c = *(const uint2 *)(ie + KV_SB_LOCK_OFF + 8 * q);
      la0 = c.x; lb0 = c.y;
    }
  }
  const uint64_t hkey = shfl_u64(key, hl);
  const uint64_t m_bad = __ballot(valid && !(key == hkey && kv_simple_op<WL>(type)));
  const bool simple = valid && (m_bad & seg) == 0 && !force_rounds;
  kv_stamp(tr, 4);
  const bool leader = head && simple;
  if (leader) {
    if (WL == DINT_WL_TATP) la0 = (H.lockw >> (8 * q)) & 0xFFu;
    const kv_where w = kv_locate(t, bucket, H, key);
    found = w.found; link = w.link; slot = w.slot; ver0 = w.ver;
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
        const uint64_t m_lk = __ballot(simple && (type == 1 || type == 2 || type == 12)) & seg;
        const uint64_t m_acq = __ballot(simple && type == 1);
        const uint64_t lk_below = m_lk & lt;
        const uint32_t lock_seen = lk_below ? (uint32_t)((m_acq >> (63 - __clzll(lk_below))) & 1ull) : la0;
        if (m_lk) fin_la = (uint32_t)((m_acq >> (63 - __clzll(m_lk))) & 1ull);
        nmiss = found ? 0 : (uint32_t)__popcll(__ballot(writer) & seg);
        switch (type) {
          case 0: my_code = found ? 4 : 6; my_get = found; break;
          case 1: my_code = lock_seen ? 8 : 7; break;
          case 2: my_code = 9; break;
          case 12: my_code = 15; break;
          default: my_code = 16; break;  // 13 kCommitBck
        }
      }
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
          default: wr = fnd; miss += !fnd; return 14;  // 5 kCommitBck
        }
      };
      const bool single = simple && seg == (1ull << lane);
      if (single) {  // one request on its bucket: apply it directly
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
        const uint32_t fnd = __builtin_amdgcn_readlane(found, L);
        uint32_t la = __builtin_amdgcn_readlane(la0, L), lb = __builtin_amdgcn_readlane(lb0, L);
        uint32_t ver = __builtin_amdgcn_readlane(ver0, L), miss = 0;
        int src = -1;
        for (uint64_t m = sm; m; m &= m - 1) {
          const int l = __ffsll((unsigned long long)m) - 1;
          const uint32_t op = __builtin_amdgcn_readlane(type, l);
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
  const uint32_t src_idx = __shfl(idx, my_src >= 0 ? my_src : lane, 64);
  const uint32_t fin_idx = __shfl(idx, fin_src >= 0 ? fin_src : lane, 64);
  uint8_t *row = nullptr;
  if (simple) {
    row = kv_entry_ptr(t, bucket, link) + KV_VAL_OFF + slot * F::VS;
    if (my_get) {
      const uint8_t *from = my_src >= 0 ? rep + (size_t)src_idx * F::MSG + F::VAL : row;
      kv_copy_words(msg + F::VAL, from, F::VS);
      st_u32(msg + F::VER, my_ver);
    }
    msg[F::TYPE] = (uint8_t)my_code;
  }
  kv_stamp(tr, 6);
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");  // the table reads above precede the write-backs below
  // ---- 4. final state of each simple segment, written once by its head
  if (leader) {
    if (fin_src >= 0) {
      kv_copy_words(row, rep + (size_t)fin_idx * F::MSG + F::VAL, F::VS);
      kv_entry_hdr(t, bucket, link)->ver[slot] = fin_ver;
    }
    if (WL == DINT_WL_TATP && fin_la != la0) ie[KV_LOCKB_OFF + q] = (uint8_t)fin_la;
    if (WL == DINT_WL_SMALLBANK && (fin_la != la0 || fin_lb != lb0)) *(uint2 *)(ie + KV_SB_LOCK_OFF + 8 * q) = make_uint2(fin_la, fin_lb);
    if (nmiss) atomicAdd(&stats->missing_keys, (unsigned long long)nmiss);
  }
  kv_stamp(tr, 7);

  // ---- everything else: rounds
  const bool rounds = valid && !simple;
  const uint32_t pos = (uint32_t)(lane - hl);
  const uint64_t rm = __ballot(rounds);
  uint32_t maxpos = 0;
  if (rm) {
    // longest non-simple segment
    uint64_t heads = rm & hm;
    while (heads) {
      const int L = __ffsll((unsigned long long)heads) - 1;
      heads &= heads - 1;
      maxpos = max(maxpos, (uint32_t)__popcll(readlane_u64(seg, L)) - 1u);
    }
    for (uint32_t r = 0; r <= maxpos; r++) {
      if (rounds && pos == r) kv_do_request<WL>(msg, type, table, q, bucket, kv, stats, r == 0 ? tr : nullptr);
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");  // the next round must see this round's stores
    }
  }
  if (!last_chunk) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");  // ... and the next chunk this chunk's
  kv_stamp(tr, 8);
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

// ---- k_kv_resolve ----------------------------------------------------------------------------------------
template <int WL>
__global__ void __launch_bounds__(64)
k_kv_resolve(uint8_t *rep, uint32_t n, const kv_dev *__restrict__ kv_g, uint32_t *__restrict__ bin_cnt,
             const uint64_t *__restrict__ bins, dint_dev_stats *__restrict__ stats, int kv_force_rounds,
             uint64_t *trace) {
  __shared__ dint_rank_lds R;
  __shared__ uint32_t Srec[DINT_WCAP];  // idx | hash entry << 16, in request order
  __shared__ uint16_t Sop[DINT_WCAP];   // type | table << 8 | quadrant << 12
  __shared__ uint32_t Hk[DINT_HSIZE];   // bucket group of each hash entry
  __shared__ kv_dev Skv;                // table descriptors: per-lane lookups by table id become LDS reads
  const uint32_t bin = blockIdx.x, lane = threadIdx.x;
  uint64_t *tr = trace ? trace + (size_t)bin * 16 : nullptr;
  kv_stamp_real(tr, 10);
  kv_stamp(tr, 0);
  const uint64_t *recs = bins + (size_t)bin * DINT_MICRO;
  const uint64_t r0 = recs[lane];  // speculative (the bin region always exists): overlaps the counter load
  const uint32_t c = bin_cnt[bin];
  if (tr && lane == 0) tr[15] = c;
  if (c == 0) return;
  if (lane == 0) bin_cnt[bin] = 0;  // leave the counters clean for the next pass
  for (uint32_t k = lane; k < sizeof(kv_dev) / 4; k += 64) ((uint32_t *)&Skv)[k] = ((const uint32_t *)kv_g)[k];
  __syncthreads();
  const kv_dev *kv = &Skv;
  kv_stamp(tr, 1);

  if (c <= 64) {
    // The common case: the whole bin is one chunk.  Sort the records by (bucket group, idx) in registers:
    // groups commute, so any order that keeps each group's requests in idx order is serial-equivalent, and
    // after the sort the requests of a group sit in adjacent lanes -- no LDS, no rank bitmap, no hash.
    uint64_t w = ~0ull;  // empty lanes sort last
    if (lane < c) {
      const uint64_t r = r0;
      w = ((uint64_t)rec_gk(r) << 32) | ((uint64_t)rec_idx(r) << 16) | ((uint64_t)rec_op(r) << 8) | rec_aux(r);
    }
    kv_stamp(tr, 2);
    w = wave_sort_u64(w);
    kv_stamp(tr, 3);
    const bool valid = lane < c;
    const uint32_t gk = (uint32_t)(w >> 32), idx = (uint32_t)(w >> 16) & 0xFFFF;
    const uint32_t type = (uint32_t)(w >> 8) & 0xFF, aux = (uint32_t)w & 0xFF;
    kv_chunk<WL>(rep, valid, idx, gk, type, aux & 15u, aux >> 4, kv, stats, kv_force_rounds, true, tr);
    kv_stamp(tr, 9);
    kv_stamp_real(tr, 11);
    return;
  }

  rank_build(R, recs, c, n);
  for (uint32_t lo = 0; lo < c; lo += DINT_WCAP) {
    const uint32_t wn = min(DINT_WCAP, c - lo);
    for (uint32_t h = lane; h < DINT_HSIZE; h += 64) Hk[h] = DINT_EMPTY;
    __syncthreads();
    for (uint32_t k = lane; k < c; k += 64) {
      const uint64_t r = recs[k];
      const uint32_t rk = rank_of(R, rec_idx(r), n) - lo;
      if (rk < wn) {
        bool nw;
        const uint32_t e = lds_hash_insert(Hk, rec_gk(r), &nw);
        Srec[rk] = rec_idx(r) | (e << 16);
        const uint32_t aux = rec_aux(r);
        Sop[rk] = (uint16_t)(rec_op(r) | ((aux & 15u) << 8) | ((aux >> 4) << 12));
      }
    }
    __syncthreads();

    for (uint32_t ch = 0; ch < wn; ch += 64) {
      // 64 consecutive requests of the window; sorted by (hash entry, idx) so that each bucket's requests are
      // adjacent and in request order, as kv_chunk expects
      const uint32_t j = ch + lane;
      uint64_t w = ~0ull;
      if (j < wn) {
        const uint32_t sr = Srec[j];
        w = ((uint64_t)(sr >> 16) << 32) | ((uint64_t)(sr & 0xFFFF) << 16) | Sop[j];
      }
      w = wave_sort_u64(w);
      const bool valid = w != ~0ull;
      const uint32_t e = (uint32_t)(w >> 32) & (DINT_HSIZE - 1), idx = (uint32_t)(w >> 16) & 0xFFFF, so = (uint32_t)w & 0xFFFF;
      kv_chunk<WL>(rep, valid, idx, valid ? Hk[e] : 0xFFFFFFFFu, so & 0xFF, (so >> 8) & 15u, so >> 12, kv, stats,
                   kv_force_rounds, false);
    }
    __syncthreads();
  }
  kv_stamp_real(tr, 11);
}

// ---- launch -------------------------------------------------------------------------------------------
template <int WL>
static void launch_kv(const void *d_req, void *d_rep, uint32_t n, const dint_kv &kv, dint_log log, dint_scratch s,
                      int load_mode, hipStream_t st, hipEvent_t *ev) {
  const uint32_t P = dint_pick_bins(n);
  const uint32_t nb = (n + 255) / 256;
  if (ev) hipEventRecord(ev[0], st);
  if (WL != DINT_WL_STORE)
    hipLaunchKernelGGL((k_kv_prepass<WL>), dim3(nb), dim3(256), 0, st, (const uint8_t *)d_req, n, kv.d_dev, s.blk_cnt,
                       log.tail);
  hipLaunchKernelGGL((k_kv_scatter<WL>), dim3(nb), dim3(256), 0, st, (const uint8_t *)d_req, (uint8_t *)d_rep, n,
                     kv.d_dev, log, (const uint32_t *)s.blk_cnt, P - 1, s.bin_cnt, s.bins, s.stats, load_mode);
  if (ev) hipEventRecord(ev[1], st);
  hipLaunchKernelGGL((k_kv_resolve<WL>), dim3(P), dim3(64), 0, st, (uint8_t *)d_rep, n, kv.d_dev, s.bin_cnt,
                     (const uint64_t *)s.bins, s.stats, kv.force_rounds, kv.d_trace);
  if (ev) hipEventRecord(ev[2], st);
}

void dint_launch_kv(const void *d_req, void *d_rep, uint32_t n, const dint_kv &kv, dint_log log, dint_scratch s,
                    int load_mode, hipStream_t st, hipEvent_t *ev) {
  if (n == 0) return;
  switch (kv.workload) {
    case DINT_WL_STORE: launch_kv<DINT_WL_STORE>(d_req, d_rep, n, kv, log, s, load_mode, st, ev); break;
    case DINT_WL_TATP: launch_kv<DINT_WL_TATP>(d_req, d_rep, n, kv, log, s, load_mode, st, ev); break;
    default: launch_kv<DINT_WL_SMALLBANK>(d_req, d_rep, n, kv, log, s, load_mode, st, ev); break;
  }
}

// ---- home shard of each request (multi-GPU routing): global bucket % shard_count -------------------------
__global__ void __launch_bounds__(256)
k_home_kv(const uint8_t *__restrict__ req, uint32_t n, const kv_dev *__restrict__ kv, dint_kv_fmt f,
          uint8_t *__restrict__ home) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  const uint8_t *m = req + (size_t)i * f.msg;
  const uint32_t table = f.table == 0xFFFFFFFFu ? 0 : m[f.table];
  if (table >= kv->n_tables) { home[i] = 0xFF; return; }
  const uint64_t g = dint_fastmod(dint_hash_key(ld_u64(m + f.key)), kv->mod[table]);
  home[i] = (uint8_t)(g % (kv->shard_count ? kv->shard_count : 1));
}
void dint_launch_home_kv(const void *d_req, uint32_t n, const dint_kv &kv, uint8_t *d_home, hipStream_t st) {
  if (n == 0) return;
  hipLaunchKernelGGL(k_home_kv, dim3((n + 255) / 256), dim3(256), 0, st, (const uint8_t *)d_req, n, kv.d_dev,
                     dint_kv_format(kv.workload), d_home);
}

// ---- table management (host) ------------------------------------------------------------------------------
int dint_kv_create(dint_kv *kv, uint32_t workload, uint64_t n_rows, dint_shard shard) {
  *kv = dint_kv();
  kv->workload = workload;
  uint64_t hs[DINT_KV_MAX_TABLES] = {0, 0, 0, 0, 0};
  if (workload == DINT_WL_STORE) {
    const uint64_t n = n_rows ? n_rows : 2000000ull;  // store/udp/tatp.h:10
    kv->n_tables = 1;
    kv->val_size = 40;
    hs[0] = n * 18 / 4;  // store/udp/server.cc:112-114: 12 rows per subscriber * 3/2 / 4 slots
  } else if (workload == DINT_WL_TATP) {
    const uint64_t n = n_rows ? n_rows : 7000000ull;  // tatp/udp/tatp.h:28
    kv->n_tables = 5;
    kv->val_size = 40;
    hs[0] = hs[1] = n * 3 / 2 / 4;   // tatp/udp/server_shard.cc:75-76
    hs[2] = hs[3] = n * 15 / 4 / 4;  // :77-78
    hs[4] = n * 45 / 8 / 4;          // :79
  } else {
    const uint64_t n = n_rows ? n_rows : 24000000ull;  // smallbank/udp/smallbank.h:17
    kv->n_tables = 2;
    kv->val_size = 8;
    hs[0] = hs[1] = n * 3 / 2 / 4;  // smallbank/udp/server_shard.cc:75-76
  }
  const uint32_t stride = kv->val_size == 40 ? 256u : 128u;
  const uint32_t count = shard.count ? shard.count : 1;
  kv->h.n_tables = kv->n_tables;
  kv->h.shard_index = shard.index;
  kv->h.shard_count = count;
  if (hipMalloc((void **)&kv->d_ctl, DINT_KV_CTL_BYTES * DINT_KV_MAX_TABLES) != hipSuccess) return DINT_ENOMEM;
  hipMemset(kv->d_ctl, 0, DINT_KV_CTL_BYTES * DINT_KV_MAX_TABLES);
  uint64_t gk = 0;
  for (uint32_t t = 0; t < kv->n_tables; t++) {
    if (hs[t] == 0) hs[t] = 1;
    kv->hash_size[t] = hs[t];
    kv_tab &tb = kv->h.tab[t];
    tb.n_local = (hs[t] + count - 1) / count;
    const uint64_t pool = tb.n_local / 4 + 4096;  // expected overflow at the reference's 2.67 rows/bucket: 0.14/bucket
    if (pool > 0xFFFFFFF0ull || gk + tb.n_local > 0xFFFFFFF0ull) return DINT_EINVAL;
    tb.pool_cap = (uint32_t)pool;
    tb.stride = stride;
    tb.val_size = kv->val_size;
    kv->entry_bytes[t] = (size_t)(tb.n_local + tb.pool_cap) * stride;
    if (hipMalloc((void **)&tb.entries, kv->entry_bytes[t]) != hipSuccess) return DINT_ENOMEM;
    if (hipMemset(tb.entries, 0, kv->entry_bytes[t]) != hipSuccess) return DINT_EHIP;
    if (hipMalloc((void **)&tb.pool_next, (size_t)tb.pool_cap * 4) != hipSuccess) return DINT_ENOMEM;
    hipMemset(tb.pool_next, 0, (size_t)tb.pool_cap * 4);
    uint8_t *ctl = kv->d_ctl + DINT_KV_CTL_BYTES * t;
    tb.pool_top = (uint32_t *)ctl;
    tb.free_head = (unsigned long long *)(ctl + 64);
    tb.pend_head = (unsigned long long *)(ctl + 64 + 8 * KV_NLISTS);
    kv->h.mod[t] = dint_make_mod(hs[t]);
    kv->h.lockmod[t] = dint_make_mod(4 * hs[t]);
    kv->h.gk_base[t] = (uint32_t)gk;
    gk += tb.n_local;
  }
  if (getenv("DINT_KV_TRACE")) {
    if (hipMalloc((void **)&kv->d_trace, (size_t)DINT_PMAX * 16 * 8) != hipSuccess) return DINT_ENOMEM;
    hipMemset(kv->d_trace, 0, (size_t)DINT_PMAX * 16 * 8);
  }
  if (hipMalloc((void **)&kv->d_dev, sizeof(kv_dev)) != hipSuccess) return DINT_ENOMEM;
  if (hipMemcpy(kv->d_dev, &kv->h, sizeof(kv_dev), hipMemcpyHostToDevice) != hipSuccess) return DINT_EHIP;
  return 0;
}

void dint_kv_destroy(dint_kv *kv) {
  for (uint32_t t = 0; t < DINT_KV_MAX_TABLES; t++) {
    if (kv->h.tab[t].entries) hipFree(kv->h.tab[t].entries);
    if (kv->h.tab[t].pool_next) hipFree(kv->h.tab[t].pool_next);
  }
  if (kv->d_ctl) hipFree(kv->d_ctl);
  if (kv->d_dev) hipFree(kv->d_dev);
  if (kv->d_trace) hipFree(kv->d_trace);
  *kv = dint_kv();
}

std::vector<std::pair<void *, size_t>> dint_kv_regions(dint_kv *kv) {
  std::vector<std::pair<void *, size_t>> r;
  for (uint32_t t = 0; t < kv->n_tables; t++) {
    r.push_back({kv->h.tab[t].entries, kv->entry_bytes[t]});
    r.push_back({kv->h.tab[t].pool_next, (size_t)kv->h.tab[t].pool_cap * 4});
  }
  if (kv->d_ctl) r.push_back({kv->d_ctl, (size_t)DINT_KV_CTL_BYTES * DINT_KV_MAX_TABLES});
  return r;
}

// ---- dumps (parity tooling; not on the hot path) -------------------------------------------------------------
// rows of a bucket in chain order
__device__ static inline uint32_t bucket_rows(const kv_tab &t, uint64_t b, uint64_t *keys, uint32_t *vers, uint8_t *vals) {
  uint32_t cur = kv_entry_hdr(t, b, KV_INLINE)->head, nrow = 0;
  for (uint32_t steps = 0; cur != KV_NULL && steps < KV_MAX_CHAIN; steps++) {
    const uint8_t *e = kv_entry_ptr(t, b, cur);
    const kv_hdr *h = (const kv_hdr *)e;
    for (uint32_t i = 0; i < 4; i++)
      if (kv_valid(*h, i)) {
        if (keys) {
          keys[nrow] = h->key[i];
          vers[nrow] = h->ver[i];
          for (uint32_t o = 0; o < t.val_size; o++) vals[(size_t)nrow * t.val_size + o] = e[KV_VAL_OFF + i * t.val_size + o];
        }
        nrow++;
      }
    cur = h->next;
  }
  return nrow;
}
// pass 1: rows per 256-bucket block
__global__ void __launch_bounds__(256) k_kv_dump_count(kv_tab t, uint32_t *__restrict__ blk_rows) {
  __shared__ uint32_t red[4];
  const uint64_t b = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  uint32_t nrow = b < t.n_local ? bucket_rows(t, b, nullptr, nullptr, nullptr) : 0, tot;
  wave_excl_scan_u32(nrow, &tot);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = tot;
  __syncthreads();
  if (threadIdx.x == 0) blk_rows[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
// pass 2: write rows at blk_off[block] + in-block exclusive offset
__global__ void __launch_bounds__(256) k_kv_dump_write(kv_tab t, const uint64_t *__restrict__ blk_off, uint64_t *keys,
                                                       uint32_t *vers, uint8_t *vals, uint64_t cap) {
  __shared__ uint32_t red[4];
  const uint64_t b = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  const uint32_t wv = threadIdx.x >> 6;
  const uint32_t nrow = b < t.n_local ? bucket_rows(t, b, nullptr, nullptr, nullptr) : 0;
  uint32_t tot;
  uint32_t off = wave_excl_scan_u32(nrow, &tot);
  if ((threadIdx.x & 63) == 0) red[wv] = tot;
  __syncthreads();
  for (uint32_t w = 0; w < wv; w++) off += red[w];
  const uint64_t at = blk_off[blockIdx.x] + off;
  if (nrow && at + nrow <= cap) bucket_rows(t, b, keys + at, vers + at, vals + at * t.val_size);
}

int64_t dint_kv_dump_rows(dint_kv *kv, uint32_t table, uint64_t *keys, uint32_t *vers, uint8_t *vals, uint64_t cap) {
  if (table >= kv->n_tables) return DINT_EINVAL;
  const kv_tab t = kv->h.tab[table];
  const uint32_t nb = (uint32_t)((t.n_local + 255) / 256);
  uint32_t *d_cnt = nullptr;
  if (hipMalloc((void **)&d_cnt, (size_t)nb * 4) != hipSuccess) return DINT_ENOMEM;
  hipLaunchKernelGGL(k_kv_dump_count, dim3(nb), dim3(256), 0, 0, t, d_cnt);
  std::vector<uint32_t> cnt(nb);
  hipMemcpy(cnt.data(), d_cnt, (size_t)nb * 4, hipMemcpyDeviceToHost);
  hipFree(d_cnt);
  std::vector<uint64_t> off(nb);
  uint64_t total = 0;
  for (uint32_t i = 0; i < nb; i++) { off[i] = total; total += cnt[i]; }
  if (!keys || !vers || !vals || cap == 0 || total == 0) return (int64_t)total;
  const uint64_t m = std::min<uint64_t>(cap, total);
  uint64_t *d_off = nullptr, *d_keys = nullptr;
  uint32_t *d_vers = nullptr;
  uint8_t *d_vals = nullptr;
  int64_t rc = (int64_t)total;
  if (hipMalloc((void **)&d_off, (size_t)nb * 8) != hipSuccess || hipMalloc((void **)&d_keys, m * 8) != hipSuccess ||
      hipMalloc((void **)&d_vers, m * 4) != hipSuccess || hipMalloc((void **)&d_vals, m * t.val_size) != hipSuccess) {
    rc = DINT_ENOMEM;
  } else {
    hipMemcpy(d_off, off.data(), (size_t)nb * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_kv_dump_write, dim3(nb), dim3(256), 0, 0, t, (const uint64_t *)d_off, d_keys, d_vers, d_vals, m);
    hipMemcpy(keys, d_keys, m * 8, hipMemcpyDeviceToHost);
    hipMemcpy(vers, d_vers, m * 4, hipMemcpyDeviceToHost);
    if (hipMemcpy(vals, d_vals, m * t.val_size, hipMemcpyDeviceToHost) != hipSuccess) rc = DINT_EHIP;
  }
  hipFree(d_off); hipFree(d_keys); hipFree(d_vers); hipFree(d_vals);
  return rc;
}

// lock words -> a[q * n_local + local], b[...]   tatp: a = txn lock; smallbank: a = num_ex, b = num_sh
__global__ void __launch_bounds__(256) k_kv_read_locks(kv_tab t, int is_sb, uint32_t *a, uint32_t *b, uint64_t cap) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= t.n_local) return;
  const uint8_t *e = kv_entry_ptr(t, i, KV_INLINE);
  for (uint32_t q = 0; q < 4; q++) {
    const uint64_t at = (uint64_t)q * t.n_local + i;
    if (at >= cap) continue;
    if (is_sb) {
      const uint32_t *c = (const uint32_t *)(e + KV_SB_LOCK_OFF) + 2 * q;
      a[at] = c[0];
      if (b) b[at] = c[1];
    } else {
      a[at] = e[KV_LOCKB_OFF + q];
      if (b) b[at] = 0;
    }
  }
}

int64_t dint_kv_read_locks(dint_kv *kv, uint32_t table, uint32_t *a, uint32_t *b, uint64_t cap) {
  if (table >= kv->n_tables || kv->workload == DINT_WL_STORE) return DINT_EINVAL;
  const kv_tab t = kv->h.tab[table];
  const uint64_t total = 4 * t.n_local;
  if (!a || cap == 0) return (int64_t)total;
  const uint64_t m = std::min<uint64_t>(cap, total);
  uint32_t *d_a = nullptr, *d_b = nullptr;
  if (hipMalloc((void **)&d_a, m * 4) != hipSuccess || hipMalloc((void **)&d_b, m * 4) != hipSuccess) {
    hipFree(d_a);
    return DINT_ENOMEM;
  }
  hipLaunchKernelGGL(k_kv_read_locks, dim3((uint32_t)((t.n_local + 255) / 256)), dim3(256), 0, 0, t,
                     kv->workload == DINT_WL_SMALLBANK ? 1 : 0, d_a, d_b, m);
  hipMemcpy(a, d_a, m * 4, hipMemcpyDeviceToHost);
  if (b) hipMemcpy(b, d_b, m * 4, hipMemcpyDeviceToHost);
  hipFree(d_a);
  hipFree(d_b);
  return (int64_t)total;
}
