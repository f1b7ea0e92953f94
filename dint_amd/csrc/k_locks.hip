// k_locks.hip -- lock_fasst and lock_2pl on gfx950, on the pass structure of the kv workloads (dint_bins.h).
//
// Reference semantics (serial, one message at a time):
//   lock_fasst/udp/server.cc:78-119   READ / ACQUIRE_LOCK / ABORT / COMMIT on locks[], ver_table[]
//   lock_2pl/udp/server.cc:70-122     ACQUIRE shared|exclusive / RELEASE on num_ex[], num_sh[]
// Both index a direct-mapped table with slot = fasthash64(&lid,4,0xdeadbeef) % n_slots; lids that
// collide modulo n_slots share one lock word -- that aliasing is part of the semantics.
//
// GPU formulation.  Ops on different slots commute, ops on one slot must apply in request order.  One pass
// (n <= 2^20 requests):
//   k_lock_count       : one thread per request: decode, hash, slot, reserve a position in bin = (slot >> 4) & (P-1)
//                        (P ~ n / 32; the 16 slots of a 128-byte line share a bin) -- the reservations of a workgroup on
//                        one bin are merged in an LDS hash, so a hot slot costs one device atomic per workgroup -- and
//                        store the 64-bit record {slot, idx, op} in place (positions < 64) or on the overflow list;
//                        copy the request bytes to the reply array.
//                        (passes of <= 65,536 requests: a big bin's records beyond 64 go straight to a region of the bin's
//                        own, named by whoever took the bin past 64 -- no overflow list, no k_kv_scan_place: two launches;
//                        the copy also carries every request's DEFAULT reply code, so the resolve kernel stores only grants)
//   k_kv_scan_place    : larger passes: ranges of the overflow area for the bins of more than 64 records, records placed
//   k_lock_resolve     : every bin of the pass in one launch.  One wave per bin of <= 64 records: sort by (slot, idx) in
//                        registers -- slots commute, so any order that keeps each slot's requests in request order is
//                        serial-equivalent -- fetch every slot's 8-byte word once, resolve all slots of the chunk at
//                        once (lock_fasst: closed form with ballots; lock_2pl: the counters are walked per slot with
//                        wave-uniform registers), write each changed word back once.  No LDS, no global atomics.
//                        One 512-thread workgroup per bigger bin (a hot slot; workgroups 0 .. 511 walk the list): the bin
//                        is sorted in LDS a stretch of <= 4096 records at a time; slots whose requests sit inside one
//                        64-record chunk are resolved as above, all chunks in parallel; a slot whose requests cross
//                        chunks is walked by one wave with the slot's word in registers -- O(requests), where r01
//                        re-ranked the bin once per 512-record window (O(c^2 / 512)) and kept 1 GB of worst-case scratch.
// The table is an array of uint2 in HBM: fasst {lock, ver}, 2pl {num_ex, num_sh}.
#include "dint_bins.h"

struct __attribute__((packed)) fasst_msg {  // lock_fasst/udp/net.h:23-29
  uint8_t type;
  uint32_t lid;
  uint32_t ver;
};
struct __attribute__((packed)) tpl_msg {  // lock_2pl/udp/net.h:25-31
  uint8_t action;
  uint32_t lid;
  uint8_t type;
};

// batch record: slot (32 bits) << 23 | request index (20 bits) << 3 | op (3 bits).  Sorting the records as integers
// groups them by slot, request order inside a slot.
__device__ static inline uint64_t lk_rec(uint32_t slot, uint32_t idx, uint32_t op) { return ((uint64_t)slot << 23) | ((uint64_t)idx << 3) | op; }
__device__ static inline uint32_t lk_slot(uint64_t r) { return (uint32_t)(r >> 23); }
__device__ static inline uint32_t lk_idx(uint64_t r) { return (uint32_t)(r >> 3) & 0xFFFFFu; }
__device__ static inline uint32_t lk_op(uint64_t r) { return (uint32_t)r & 7u; }
#define LK_DIRECT_NMAX 65536u  // passes of at most this many requests put a big bin's records beyond 64 straight into its own region (a bin holds at most the pass)
// exclusive prefix sum over the 64 lanes at VALU speed: four row_shr steps inside the rows of 16, row_bcast:15 / :31 across
// them (the six dependent ds_bpermute round trips of wave_excl_scan_u32 are 0.4 us -- the mode walk of lock_2pl's dominant
// slot does one scan per 64 groups)
__device__ static inline uint32_t lk_excl_scan_u32(uint32_t x, uint32_t *total) {
  int v = (int)x;
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, false);  // row_shr:1 (lanes without a source keep `old` = 0)
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, false);  // row_shr:2
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, false);  // row_shr:4
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, false);  // row_shr:8
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);  // row_bcast:15 -> rows 1, 3
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);  // row_bcast:31 -> rows 2, 3
  *total = (uint32_t)__builtin_amdgcn_readlane(v, 63);
  return (uint32_t)v - x;
}
// lane `src`'s 64-bit value on every lane (src wave-uniform)
__device__ static inline uint64_t lk_readlane_u64(uint64_t v, int src) {
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, src), hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), src);
  return ((uint64_t)hi << 32) | lo;
}

// ------------------------------------------------------------------------------------------
// (TB threads = TB requests per workgroup: KV_TB in k_lock_count, KVB_T in k_lock_pass, where the count of the NEXT batch rides
// beside this batch's resolve workgroups; `blk` = which slice of the batch)
template <uint32_t TB>
__device__ static inline uint32_t lk_hash_insert(uint32_t *keys, uint32_t k) {  // 2 * TB slots, keys != KV_NONE
  uint32_t h = ((k * 0x9E3779B1u) >> 8) & (2 * TB - 1);
  for (;;) {
    const uint32_t old = atomicCAS(&keys[h], KV_NONE, k);
    if (old == KV_NONE || old == k) return h;
    h = (h + 1) & (2 * TB - 1);
  }
}
struct lk_count_args {
  const uint8_t *req;
  uint8_t *rep;
  uint32_t n;
  dint_mod slots;
  dint_shard shard;
  uint32_t pbits;
  uint32_t *bin_cnt;
  uint64_t *bins;
  uint32_t *big;
  uint4 *ovl;
  dint_dev_stats *stats;
  dint_view V;
  uint64_t *bigrec;    // DIRECT big bins (passes of <= LK_DIRECT_NMAX requests), else nullptr
  uint32_t *slot_of;
};
template <int WL, uint32_t TB>  // WL: 0 = lock_fasst, 1 = lock_2pl
__device__ __forceinline__ static void lk_count_body(const lk_count_args &A, uint32_t blk) {
  constexpr uint32_t MSG = WL == 0 ? sizeof(fasst_msg) : sizeof(tpl_msg);
  const uint8_t *__restrict__ req = A.req;
  uint8_t *rep = A.rep;
  const uint32_t n = A.n, pbits = A.pbits;
  const dint_mod slots = A.slots;
  const dint_shard shard = A.shard;
  uint32_t *__restrict__ bin_cnt = A.bin_cnt, *__restrict__ big = A.big, *slot_of = A.slot_of;
  uint64_t *__restrict__ bins = A.bins, *__restrict__ bigrec = A.bigrec;
  uint4 *__restrict__ ovl = A.ovl;
  dint_dev_stats *__restrict__ stats = A.stats;
  const dint_view V = A.V;
  __shared__ uint32_t Hb[2 * TB];  // bins this workgroup appends to
  __shared__ uint32_t Hc[2 * TB];  // ... how many records each; then the position of the workgroup's first one
  __shared__ uint32_t Sov[2];
  const uint32_t t = threadIdx.x, i = blk * TB + t;
  Hb[t] = KV_NONE; Hb[t + TB] = KV_NONE;
  Hc[t] = 0; Hc[t + TB] = 0;
  if (t == 0) Sov[0] = 0;
  __syncthreads();
  bool live;
  const size_t off = dint_view_off(V, i < n ? i : 0, MSG, &live);
  live = live && i < n;
  // replies are the request mutated in place: copy this slice (contiguous passes with separate arrays only)
  if (rep != req) {
    const size_t lo = (size_t)blk * TB * MSG, hi = min((size_t)n * MSG, lo + (size_t)TB * MSG);
    if ((((uintptr_t)req | (uintptr_t)rep) & 15) == 0) {
      const uint32_t nv = (uint32_t)((hi - lo) / 16);  // <= 9 / 16 of TB vectors
      if (t < nv) ((uint4 *)(rep + lo))[t] = ((const uint4 *)(req + lo))[t];
      for (size_t k = lo + (size_t)nv * 16 + t; k < hi; k += TB) rep[k] = req[k];
    } else {
      for (size_t k = lo + t; k < hi; k += TB) rep[k] = req[k];
    }
  }
  uint32_t lid = 0, op = 0;
  bool ok = false;
  if (live) {
    if (WL == 0) {
      const fasst_msg m = *(const fasst_msg *)(req + off);
      lid = m.lid;
      op = m.type;  // 0 READ, 1 ACQUIRE_LOCK, 2 ABORT, 3 COMMIT
      ok = op <= 3;
    } else {
      const tpl_msg m = *(const tpl_msg *)(req + off);
      lid = m.lid;
      if (m.action == 0) {  // ACQUIRE: op 0 shared, 1 exclusive; any other lock type panics in the reference
        ok = m.type <= 1;
        op = m.type;
      } else if (m.action == 1) {  // RELEASE: op 2 shared, 3 exclusive, 4 = unknown type (ack, no change)
        ok = true;
        op = 2u + (m.type <= 1 ? m.type : 2u);
      }
    }
    if (!ok) atomicAdd(&stats->bad_requests, 1ULL);
  }
  uint32_t bin = KV_NONE, local = 0;
  if (ok) {
    const uint64_t g = dint_fastmod(dint_hash_lid(lid), slots);
    local = (uint32_t)g;
    bool mine = true;
    if (shard.count > 1) {
      mine = (uint32_t)(g % shard.count) == shard.index;
      if (!mine) atomicAdd(&stats->foreign_requests, 1ULL);
      local = (uint32_t)(g / shard.count);
    }
    if (mine) bin = (local >> 4) & ((1u << pbits) - 1u);
  }
  uint32_t e = 0, mypos = 0;
  if (bin != KV_NONE) {
    e = lk_hash_insert<TB>(Hb, bin);
    mypos = atomicAdd(&Hc[e], 1u);
  }
  __syncthreads();
#pragma unroll
  for (uint32_t k = 0; k < 2; k++) {
    const uint32_t sl = t + k * TB;
    if (Hb[sl] != KV_NONE) {
      const uint32_t cnt = Hc[sl], base = atomicAdd(&bin_cnt[Hb[sl]], cnt);
      Hc[sl] = base;
      if (base <= DINT_KV_BINCAP && base + cnt > DINT_KV_BINCAP) {  // this run takes the bin past its in-place region: list it
        const uint32_t k_big = atomicAdd(&big[0], 1u);
        big[4 + k_big] = Hb[sl];
        // direct: the bin's position in the list names its region of `bigrec`; published for the workgroups whose runs
        // landed behind this one (they hold a reservation that only exists because this thread has made its own)
        if (bigrec) __hip_atomic_store(&slot_of[Hb[sl]], k_big, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
  __syncthreads();
  if (bin != KV_NONE) mypos += Hc[e];
  const uint64_t rec = lk_rec(local, i, op);
  const bool over = bin != KV_NONE && mypos >= DINT_KV_BINCAP;
  if (bin != KV_NONE && !over) bins[(size_t)bin * DINT_KV_BINCAP + mypos] = rec;
  if (bigrec) {
    // A record beyond the 64 in place goes STRAIGHT to its bin's region -- no overflow list, no k_kv_scan_place between this
    // kernel and the resolve kernel (r01-r05: 7 us of a 37 us pass).  The region is named by whoever took the bin past 64: a
    // thread of this kernel that has done its atomic already and publishes right behind it, before any barrier -- the wait
    // below is a few hundred nanoseconds, and bounded: it traps instead of hanging should that ever not hold.
    if (over) {
      uint32_t sl = KV_NONE;
      for (uint32_t spin = 0; spin < (1u << 22); spin++) {
        sl = __hip_atomic_load(&slot_of[bin], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (sl != KV_NONE) break;
        __builtin_amdgcn_s_sleep(1);
      }
      if (sl == KV_NONE) __builtin_trap();
      bigrec[(size_t)sl * LK_DIRECT_NMAX + (mypos - DINT_KV_BINCAP)] = rec;
    }
  } else {
  uint32_t orank = 0;
  if (over) orank = atomicAdd(&Sov[0], 1u);
  __syncthreads();
  if (Sov[0]) {  // workgroup-uniform
    if (t == 0) Sov[1] = atomicAdd(&big[1], Sov[0]);
    __syncthreads();
    if (over) ovl[Sov[1] + orank] = make_uint4((uint32_t)rec, (uint32_t)(rec >> 32), bin, mypos);
  }
  }
  // The reply code a request gets UNLESS the table grants it something (Ops::write_reply): every RELEASE / ABORT / COMMIT is
  // acked whatever the table holds, an ACQUIRE is rejected unless granted.  Written here, beside the copy of the request
  // bytes -- neighbouring threads, neighbouring messages --, so that the resolve kernel stores only the grants: a lock that
  // a fifth of the closed-loop workers keep retrying is 12,800 REJECTs per 64k batch, and its workgroup waited for
  // 12,800 scattered byte stores between the rounds of its record loads (r05: 14 of its 30 us).
  // (after the barriers above: the vector copy of this slice is complete)
  if (bin != KV_NONE) rep[off] = (uint8_t)(WL == 0 ? (op == 0 ? 4u : op == 1 ? 6u : op == 2 ? 7u : 8u) : (op <= 1 ? 3u : 5u));
}
template <int WL>
__global__ void __launch_bounds__(KV_TB)
k_lock_count(lk_count_args A) {
  lk_count_body<WL, KV_TB>(A, blockIdx.x);
}

// ------------------------------------------------------------------------------------------
struct FasstOps {
  static constexpr bool CLOSED = true;  // a slot's requests have a closed form: any number of them resolves in parallel
  // request at sorted position p of a slot segment: `lock_before` = what the last lock-writing op below it left (ACQUIRE
  // leaves 1 whether granted or not, ABORT / COMMIT leave 0), else the stored lock; ver_before = ver0 + COMMITs below
  __device__ static void closed(uint32_t op, uint32_t lock_before, uint32_t ver_before, uint32_t &code, uint32_t &rv) {
    switch (op) {
      case 0: code = 4; rv = ver_before; break;
      case 1: code = lock_before ? 6 : 5; break;
      case 2: code = 7; break;
      default: code = 8; break;
    }
  }
  // one op on one slot word {x = lock, y = ver}: lock_fasst/udp/server.cc:85-114
  __device__ static uint32_t apply(uint32_t op, uint2 &st, uint32_t &rv, bool &wr) {
    switch (op) {
      case 0: rv = st.y; return 4;                                 // READ -> GRANT_READ, ver
      case 1: if (st.x == 0) { st.x = 1; wr = true; return 5; }    // ACQUIRE: CAS 0->1 GRANT_LOCK
              return 6;                                            //          else REJECT_LOCK
      case 2: wr = st.x != 0; st.x = 0; return 7;                  // ABORT: CAS 1->0, ABORT_ACK
      default: st.y++; st.x = 0; wr = true; return 8;              // COMMIT: ver++, unlock, COMMIT_ACK
    }
  }
  // the same closed form over a sorted chunk: `seg` = lane mask of this lane's slot (adjacent lanes, request
  // order); every slot of the chunk is resolved at once.  st0 = the slot's word (same on all lanes of a segment)
  __device__ static void resolve_sorted(bool valid, uint64_t seg, uint32_t op, uint2 st0, uint32_t &code, uint32_t &rv,
                                        uint2 &fin, bool &dirty) {
    const uint64_t lt = lanemask_lt();
    const uint64_t m_set = __ballot(valid && op != 0) & seg;
    const uint64_t m_acq = __ballot(valid && op == 1);
    const uint64_t m_com = __ballot(valid && op == 3) & seg;
    const uint64_t prev = m_set & lt;
    const uint32_t lock_before = prev ? (uint32_t)((m_acq >> (63 - __clzll(prev))) & 1ULL) : st0.x;
    const uint32_t ver_before = st0.y + (uint32_t)__popcll(m_com & lt);
    switch (op) {
      case 0: code = 4; rv = ver_before; break;
      case 1: code = lock_before ? 6 : 5; break;
      case 2: code = 7; break;
      default: code = 8; break;
    }
    fin.x = m_set ? (uint32_t)((m_acq >> (63 - __clzll(m_set))) & 1ULL) : st0.x;
    fin.y = st0.y + (uint32_t)__popcll(m_com);
    dirty = fin.x != st0.x || fin.y != st0.y;
  }
  __device__ static void walk64(bool, uint32_t, uint2 &, uint32_t &) {}  // (never called: CLOSED)
  __device__ static uint64_t walk64_mask(bool, uint32_t, uint2 &) { return 0; }
  __device__ static void write_reply(uint8_t *rep, const dint_view &V, uint32_t idx, uint32_t op, uint32_t code, uint32_t rv) {
    fasst_msg *m = (fasst_msg *)(rep + dint_view_off(V, idx, sizeof(fasst_msg)));
    if (op == 0) m->ver = rv;  // ver is echoed on every non-READ reply
    else if (code == 5) m->type = 5;  // GRANT_LOCK; every other code is what k_lock_count wrote already
  }
};

struct TplOps {
  static constexpr bool CLOSED = false;  // counters: a slot's requests are walked in order
  __device__ static void closed(uint32_t, uint32_t, uint32_t, uint32_t &, uint32_t &) {}
  // {x = num_ex, y = num_sh}: lock_2pl/udp/server.cc:83-121 (the per-slot spin lock is never
  // contended in a serial replay, so RETRY never occurs)
  __device__ static uint32_t apply(uint32_t op, uint2 &st, uint32_t &rv, bool &wr) {
    (void)rv;
    switch (op) {
      case 0: if (st.x == 0) { st.y++; wr = true; return 2; } return 3;               // shared
      case 1: if (st.x == 0 && st.y == 0) { st.x++; wr = true; return 2; } return 3;  // exclusive
      case 2: st.y--; wr = true; return 5;  // release shared (unsigned wrap if unmatched, as the reference)
      case 3: st.x--; wr = true; return 5;  // release exclusive
      default: return 5;                    // release with unknown lock type: ack only
    }
  }
  // sorted chunk: single requests apply their op directly; longer segments are walked once each with
  // wave-uniform registers (the counters have no closed form)
  __device__ static void resolve_sorted(bool valid, uint64_t seg, uint32_t op, uint2 st0, uint32_t &code, uint32_t &rv,
                                        uint2 &fin, bool &dirty) {
    const uint32_t lane = lane_id();
    fin = st0;
    dirty = false;
    const bool single = valid && seg == (1ull << lane);
    if (single) code = apply(op, fin, rv, dirty);
    const bool head = valid && (seg & lanemask_lt()) == 0;
    uint64_t multi = __ballot(head && !single);
    while (multi) {
      const int L = __ffsll((unsigned long long)multi) - 1;
      multi &= multi - 1;
      const uint32_t shi = (uint32_t)__builtin_amdgcn_readlane((uint32_t)(seg >> 32), L);
      const uint32_t slo = (uint32_t)__builtin_amdgcn_readlane((uint32_t)seg, L);
      uint2 st;
      st.x = (uint32_t)__builtin_amdgcn_readlane(st0.x, L);
      st.y = (uint32_t)__builtin_amdgcn_readlane(st0.y, L);
      bool wr = false;
      for (uint64_t m = ((uint64_t)shi << 32) | slo; m; m &= m - 1) {
        const int l = __ffsll((unsigned long long)m) - 1;
        const uint32_t lop = __builtin_amdgcn_readlane(op, l);
        uint32_t lrv = 0;
        bool lwr = false;
        const uint32_t lcode = apply(lop, st, lrv, lwr);
        wr |= lwr;
        if ((int)lane == l) { code = lcode; rv = lrv; }
      }
      if ((int)lane == L) { fin = st; dirty = wr; }
    }
  }
  // <= 64 consecutive requests of ONE slot, one per lane in request order (a whole wave calls this), the slot's counters
  // in wave-uniform registers.  Which ACQUIREs are granted is the only thing that depends on the counters, and they move
  // between two modes: FREE (num_ex == 0: every shared ACQUIRE is granted, until an exclusive ACQUIRE finds num_sh == 0
  // -- granted, num_ex = 1 -- or a RELEASE of an exclusive lock nobody holds wraps num_ex) and HELD (num_ex != 0: every
  // ACQUIRE is rejected until the num_ex-th exclusive RELEASE).  Between two mode changes all lanes are resolved at once
  // from ballot masks (num_sh before a lane = num_sh + shared acquires - shared releases below it), so the loop runs
  // once per mode change, not once per request: a lock that hundreds of workers hammer changes mode rarely.
  // (the same walk as smallbank's counters in k_kv.hip; lock_2pl/udp/server.cc:83-121)
  __device__ static void walk64(bool valid, uint32_t op, uint2 &st, uint32_t &code) {
    const uint64_t G = walk64_mask(valid, op, st);
    code = op <= 1 ? (((G >> lane_id()) & 1ull) ? 2u : 3u) : 5u;
  }
  // ... returning the lanes whose ACQUIRE is granted
  __device__ static uint64_t walk64_mask(bool valid, uint32_t op, uint2 &st) {
    const uint32_t lane = lane_id();
    const uint64_t mAS = __ballot(valid && op == 0), mAX = __ballot(valid && op == 1);
    const uint64_t mRS = __ballot(valid && op == 2), mRX = __ballot(valid && op == 3);
    uint32_t la = (uint32_t)__builtin_amdgcn_readfirstlane((int)st.x), lb = (uint32_t)__builtin_amdgcn_readfirstlane((int)st.y);
    uint64_t G = 0, rem = mAS | mAX | mRS | mRX;
    while (rem) {
      if (la == 0) {
        const uint64_t blw = rem & lanemask_lt();
        const uint32_t lb_before = lb + (uint32_t)__popcll(blw & mAS) - (uint32_t)__popcll(blw & mRS);
        const bool me = (rem >> lane) & 1ull;
        const uint64_t ev = __ballot(me && ((op == 1 && lb_before == 0) || op == 3));
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
          for (uint32_t k = 1; k < la; k++) rx &= rx - 1;  // the la-th exclusive RELEASE (la is 1 unless the counter wrapped)
          const uint64_t bit = rx & (0 - rx), upto = bit - 1ull;
          lb -= (uint32_t)__popcll(rem & upto & mRS);
          la = 0;
          rem &= ~(upto | bit);
        }
      }
    }
    st.x = la; st.y = lb;
    return G;
  }
  __device__ static void write_reply(uint8_t *rep, const dint_view &V, uint32_t idx, uint32_t op, uint32_t code, uint32_t rv) {
    (void)op; (void)rv;
    if (code == 2) ((tpl_msg *)(rep + dint_view_off(V, idx, sizeof(tpl_msg))))->action = 2;  // GRANT_LOCK; REJECT / RELEASE_ACK: k_lock_count's
  }
};

// one sorted chunk of <= 64 records (ascending, invalid lanes last): every slot whose requests all sit in the chunk.
// `take` = this lane's slot segment is handled here (false: its segment crosses into another chunk -- see the big bins)
template <class Ops>
__device__ static inline void lk_chunk(uint8_t *rep, const dint_view &V, uint2 *__restrict__ table, uint64_t w, bool valid,
                                       bool first_is_head, bool last_is_tail) {
  const uint32_t lane = lane_id();
  const uint32_t slot = lk_slot(w), idx = lk_idx(w), op = lk_op(w);
  const uint32_t up = __shfl_up(slot, 1, 64);
  const bool head = valid && (lane == 0 || up != slot);
  const uint64_t hm = __ballot(head), vm = __ballot(valid);
  const uint64_t lt = lanemask_lt(), le = lt | (1ull << lane);
  const int hl = valid ? 63 - __clzll(hm & le) : (int)lane;
  const uint64_t above = hm & ~le;
  const uint64_t next = above ? (above & (~above + 1ull)) : vm + 1ull;
  uint64_t seg = valid ? ((next - 1ull) & ~((1ull << hl) - 1ull)) : 0;
  // a segment that starts before the chunk or goes on after it is not mine
  const int nvalid = __popcll(vm);
  bool take = valid;
  if (!first_is_head && hl == 0) take = false;
  if (!last_is_tail && valid && ((seg >> (nvalid - 1)) & 1ull)) take = false;
  if (!take) seg = 0;
  const bool thead = head && take;
  uint2 st0 = make_uint2(0, 0);
  if (thead) st0 = table[slot];
  st0.x = __shfl(st0.x, hl, 64);
  st0.y = __shfl(st0.y, hl, 64);
  uint32_t code = 0, rv = 0;
  uint2 fin = st0;
  bool dirty = false;
  Ops::resolve_sorted(take, seg, op, st0, code, rv, fin, dirty);
  if (take) Ops::write_reply(rep, V, idx, op, code, rv);
  if (thead && dirty) table[slot] = fin;
}

template <class Ops>
__device__ static inline void
lk_small_bin(uint8_t *rep, uint32_t pbits, uint2 *__restrict__ table, uint32_t *__restrict__ bin_cnt,
             const uint64_t *__restrict__ bins, const dint_view &V, uint32_t bin) {
  const uint32_t lane = threadIdx.x & 63;
  if (bin >= (1u << pbits)) return;
  const uint64_t r0 = bins[(size_t)bin * DINT_KV_BINCAP + lane];  // speculative (the bin region always exists): overlaps the counter load
  const uint32_t c = bin_cnt[bin];
  if (c == 0 || c > DINT_KV_BINCAP) return;  // larger bins are on the big-bin list
  if (lane == 0) bin_cnt[bin] = 0;  // leave the counters clean for the next pass
  const uint64_t w = wave_sort_u64(lane < c ? r0 : ~0ull);
  lk_chunk<Ops>(rep, V, table, w, lane < c, true, true);
}

#define TPL_HOT_NMAX 65536u   // lock_2pl dominant-slot path: passes of at most this many requests (an 8 KB index bitmap)
static_assert(TPL_HOT_NMAX / 32 * 4 == KVB_NMAX / 4 * 8, "an index bitmap of the dominant-slot path is a quarter of the stretch buffer");

// ---- big bins: one 512-thread workgroup each ---------------------------------------------------------------------
template <class Ops>
__device__ static inline void
lk_big_bins(uint8_t *rep, uint32_t n, uint2 *__restrict__ table, uint32_t *__restrict__ bin_cnt,
            const uint64_t *__restrict__ bins, const uint32_t *__restrict__ big, uint32_t *bin_off,
            const uint64_t *__restrict__ ovf, uint32_t hot_min_, uint64_t *trace, const dint_view &V, const uint32_t vb,
            const uint32_t n_walk,  // workgroup vb of the n_walk that walk the big-bin list
            const uint64_t *__restrict__ bigrec, dint_dev_stats *__restrict__ stats) {  // direct big bins (else nullptr): bin bi of the list owns region bi
  const uint32_t hot_min = hot_min_ & 0x7FFFFFFFu;
  const bool misguess = hot_min_ >> 31;  // DINT_LOCK_MISGUESS (tests): the bitmaps are filled for nobody first, then for the dominant slot
  const uint32_t bin_first = big[4 + vb];  // speculative: in flight together with the list length
  const uint32_t nbig = big[0];
  if (vb >= nbig) return;
  // tracing (DINT_KV_TRACE=1): phase stamps of this workgroup's first bin, 10 ns ticks
  unsigned long long *tw = trace ? (unsigned long long *)trace + (size_t)DINT_KV_PMAX * 16 + 16 * vb : nullptr;
#define LK_STAMP(k) do { if (tw && threadIdx.x == 0 && bi == vb && win == 0) tw[k] = __builtin_amdgcn_s_memrealtime(); } while (0)
  __shared__ uint64_t Sk[KVB_NMAX];
  __shared__ uint32_t Bcnt[KVB_NBK / 2];
  __shared__ uint16_t Bwin[KVB_NBK];
  __shared__ uint64_t Mhead[KVB_NW];
  __shared__ kvb_edge Ehead;
  __shared__ uint64_t Mset[KVB_NW], Macq[KVB_NW], Mcom[KVB_NW];  // closed form: lock-writing ops, ACQUIREs, COMMITs
  __shared__ kvb_edge Eset;
  __shared__ kvb_pop Pcom;
  __shared__ uint32_t Xs[KVB_NW][2];  // slots whose requests cross 64-record chunks: [a, b) in the sorted stretch
  __shared__ uint32_t Swn, Snx, Sred[KVB_W], Srest;
  __shared__ uint32_t Hs[16];                  // dominant-slot path: counters
  __shared__ uint64_t Mk[KVB_MMAX];            // ... its lock-writing ops, idx << 12 | position, ascending
  __shared__ uint16_t Mcc[KVB_MMAX + 8];       // ... COMMITs among the first j of them
  __shared__ uint16_t Pw[TPL_HOT_NMAX / 32 + 1];  // dominant slot (lock_fasst): COMMITs in the index words below w
  const uint32_t t = threadIdx.x, lane = t & 63, wave = t >> 6;
  // request-index buckets that cut a bin of more than KVB_NMAX records into stretches (every request of a stretch
  // precedes every request of the next one)
  const uint32_t nbits = n > 1 ? 32u - (uint32_t)__clz(n - 1) : 0u, bs = nbits > 11 ? nbits - 11 : 0u;
  const uint32_t wcap = KVB_NMAX - (1u << bs);
  for (uint32_t bi = vb; bi < nbig; bi += n_walk) {
    const uint32_t bin = bi == vb ? bin_first : big[4 + bi];
    __syncthreads();
    const uint64_t *recs_lo = bins + (size_t)bin * DINT_KV_BINCAP;
    // eight of the bin's first 64 records (they exist whatever the count is: on the same round trip as the count) name the
    // candidates for its dominant slot
    uint32_t cand[8];
#pragma unroll
    for (uint32_t k = 0; k < 8; k++) cand[k] = lk_slot(recs_lo[8 * k + 3]);
    const uint32_t c = bin_cnt[bin];
    if (tw && threadIdx.x == 0 && bi == vb) { tw[0] = __builtin_amdgcn_s_memrealtime(); tw[8] = c; tw[9] = nbig; }
    const uint64_t *recs_hi = (bigrec ? bigrec + (size_t)bi * LK_DIRECT_NMAX : ovf + bin_off[bin]) - DINT_KV_BINCAP;
    auto rec_at = [&](uint32_t k) -> uint64_t { return k < DINT_KV_BINCAP ? recs_lo[k] : recs_hi[k]; };
    // ---- the bin's DOMINANT SLOT in a pass of <= 65,536 requests (a lid that hundreds of closed-loop workers keep
    // retrying: most of a big bin is one slot, up to a sixth of a 64k batch) is resolved WITHOUT a sort: every request
    // owns one bit of an index bitmap, so "in request order" is "in bit order".
    //   lock_2pl  : four bitmaps -- ACQUIREs and RELEASEs, shared and exclusive; the requests of one 64-bit word are a group,
    //               and one wave goes through the groups 64 at a time (below; TplOps::walk64_mask for the few that change
    //               the counters' mode: once per mode change, not once per request).
    //   lock_fasst: three bitmaps -- lock-writing ops, ACQUIREs, COMMITs -- answer every request in O(1): lock seen =
    //               was the last lock-writing op below me an ACQUIRE, version seen = ver0 + COMMITs below me; no limit
    //               on the number of ordering ops (the stretch-level path below sorts at most 1024 of them).
    // The rest of the bin takes the general path below.  All passes over the bin's records keep 8 loads per thread
    // in flight (one memory round trip per 4096 records instead of one per 512), and there is one of them for lock_2pl, two
    // for lock_fasst (r03 / r04: four): the pass that counts the candidates fills the bitmaps for the likeliest one (the
    // majority of the samples) and collects the bin's other records (<= 1024 of them) in LDS on the way; lock_2pl's replies
    // are the grant bits that are left, lock_fasst reads the records once more to write them.
    uint32_t hslot_done = KV_NONE, c_rest = c;
    bool rest_lds = false;
#define LK_FOR_RECORDS(...)                                                       \
    for (uint32_t k0_ = 0; k0_ < c; k0_ += 8 * KVB_T) {                           \
      uint64_t r8_[8];                                                            \
      _Pragma("unroll") for (uint32_t j_ = 0; j_ < 8; j_++) {                     \
        const uint32_t k_ = k0_ + j_ * KVB_T + t;                                 \
        r8_[j_] = k_ < c ? rec_at(k_) : ~0ull;                                    \
      }                                                                           \
      _Pragma("unroll") for (uint32_t j_ = 0; j_ < 8; j_++) {                     \
        const uint64_t r = r8_[j_];                                               \
        if (k0_ + j_ * KVB_T + t < c) { __VA_ARGS__ }                                 \
      }                                                                           \
    }
    if (c >= hot_min && n <= TPL_HOT_NMAX) {
      uint32_t *Bm = (uint32_t *)Mk;              // [n / 32] lock_2pl: the slot's shared ACQUIREs, then the grants; lock_fasst: its lock-writing ops
      uint32_t *Bax = (uint32_t *)Sk, *Brs = Bax + TPL_HOT_NMAX / 32, *Brx = Brs + TPL_HOT_NMAX / 32;  // lock_2pl: exclusive ACQUIREs, RELEASEs shared / exclusive
      uint32_t *Bacq = (uint32_t *)Sk, *Bcom = Bacq + TPL_HOT_NMAX / 32, *Lw = Bcom + TPL_HOT_NMAX / 32;  // lock_fasst
      uint32_t cc[8], guess = cand[0], gv = 0;
#pragma unroll
      for (uint32_t j = 0; j < 8; j++) {
        uint32_t v = 0;
#pragma unroll
        for (uint32_t k = 0; k < 8; k++) v += cand[k] == cand[j];
        if (v > gv) { gv = v; guess = cand[j]; }
        cc[j] = 0;
      }
      if (misguess) guess = KV_NONE;
      if (t < 16) Hs[t] = 0;
      if (t == 0) Srest = 0;
      for (uint32_t w = t; w < TPL_HOT_NMAX / 32; w += KVB_T) {
        Bm[w] = 0;
        if (Ops::CLOSED) { Bacq[w] = 0; Bcom[w] = 0; }
        else { Bax[w] = 0; Brs[w] = 0; Brx[w] = 0; }
      }
      __syncthreads();
#define LK_FILL_BITMAPS()                                                                                          \
      do {                                                                                                         \
        const uint32_t idx = lk_idx(r), op = lk_op(r), w = idx >> 5, bit = 1u << (idx & 31u);                      \
        if (!Ops::CLOSED) {                                                                                        \
          /* (4: a RELEASE of an unknown lock type changes nothing) */                                             \
          if (op < 4) atomicOr(op == 0 ? &Bm[w] : op == 1 ? &Bax[w] : op == 2 ? &Brs[w] : &Brx[w], bit);          \
        } else if (op != 0) {                                                                                      \
          atomicOr(&Bm[w], bit);                                                                                   \
          if (op == 1) atomicOr(&Bacq[w], bit);                                                                    \
          if (op == 3) atomicOr(&Bcom[w], bit);                                                                    \
        }                                                                                                          \
      } while (0)
      uint64_t *Rest = Sk + 3 * (KVB_NMAX / 4);  // (the bitmaps take three quarters of Sk)
#define LK_COLLECT_REST() do { const uint32_t k_ = atomicAdd(&Srest, 1u); if (k_ < KVB_NMAX / 4) Rest[k_] = r; } while (0)
      uint2 st_guess = make_uint2(0, 0);
      if (guess != KV_NONE) st_guess = table[guess];  // (on its way while the records are read; nobody else writes a slot of this bin)
      LK_FOR_RECORDS({
        const uint32_t sl = lk_slot(r);
        _Pragma("unroll") for (uint32_t j = 0; j < 8; j++) cc[j] += sl == cand[j];
        if (sl == guess) LK_FILL_BITMAPS();
        else LK_COLLECT_REST();
      })
#pragma unroll
      for (uint32_t j = 0; j < 8; j++) {
        uint32_t v = cc[j];
        for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
        if (lane == 0 && v) atomicAdd(&Hs[j], v);
      }
      __syncthreads();
      uint32_t best = 0;
#pragma unroll
      for (uint32_t j = 1; j < 8; j++) best = Hs[j] > Hs[best] ? j : best;
      const uint32_t hot_n = Hs[best], hslot = cand[best];
      __syncthreads();
      if (tw && t == 0 && bi == vb) { tw[10] = __builtin_amdgcn_s_memrealtime(); tw[13] = hot_n; }
      if (hot_n >= hot_min && 2 * hot_n >= c) {  // workgroup-uniform
        if (hslot != guess) {  // (the samples misled: the bitmaps again, for the slot that is)
          for (uint32_t w = t; w < TPL_HOT_NMAX / 32; w += KVB_T) {
            Bm[w] = 0;
            if (Ops::CLOSED) { Bacq[w] = 0; Bcom[w] = 0; }
            else { Bax[w] = 0; Brs[w] = 0; Brx[w] = 0; }
          }
          if (t == 0) Srest = 0;
          __syncthreads();
          LK_FOR_RECORDS({
            if (lk_slot(r) == hslot) LK_FILL_BITMAPS();
            else LK_COLLECT_REST();
          })
          __syncthreads();
        }
        rest_lds = c - hot_n <= KVB_NMAX / 4;  // (= Srest: the bin's other records are in LDS)
        if (Ops::CLOSED) {  // Pw[w] = COMMITs in the words before w; thread t owns words 4t .. 4t + 3.
           // Lw[w] = index of the last lock-writing op in the words before w, ~0u = none
          const uint32_t *src = Bcom;
          uint32_t pc[4], run = 0, last = ~0u;
#pragma unroll
          for (uint32_t j = 0; j < 4; j++) {
            pc[j] = (uint32_t)__popc(src[4 * t + j]);
            run += pc[j];
            if (Ops::CLOSED && Bm[4 * t + j]) last = (4 * t + j) * 32 + 31 - (uint32_t)__clz(Bm[4 * t + j]);
          }
          uint32_t tot, base = wave_excl_scan_u32(run, &tot);
          int lm = (int)last;  // running maximum of "last" over the threads below me (~0u = -1 sorts lowest)
          for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(lm, d, 64);
            if ((int)lane >= d) lm = max(lm, o);
          }
          if (lane == 63) { Sred[wave] = tot; Hs[8 + wave] = (uint32_t)lm; }
          int ex = __shfl_up(lm, 1, 64);  // exclusive: the threads strictly below me in my wave
          if (lane == 0) ex = -1;
          __syncthreads();
          for (uint32_t w = 0; w < wave; w++) { base += Sred[w]; ex = max(ex, (int)Hs[8 + w]); }
#pragma unroll
          for (uint32_t j = 0; j < 4; j++) {
            Pw[4 * t + j] = (uint16_t)base;
            base += pc[j];
            if (Ops::CLOSED) {
              Lw[4 * t + j] = (uint32_t)ex;
              if (Bm[4 * t + j]) ex = (int)((4 * t + j) * 32 + 31 - (uint32_t)__clz(Bm[4 * t + j]));
            }
          }
          if (t == KVB_T - 1) { Hs[6] = base; Hs[7] = (uint32_t)ex; }  // totals: COMMITs (a 32-bit word: 65,536 COMMITs on one slot do not fit Pw's 16 bits) / last op of all
        }
        __syncthreads();
        uint2 st = st_guess;
        if (hslot != guess) st = table[hslot];  // workgroup-uniform address
        const uint2 st_in = st;
        if (tw && t == 0 && bi == vb) tw[11] = __builtin_amdgcn_s_memrealtime();
        if (Ops::CLOSED) {
          __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
          __syncthreads();  // everyone holds the slot's word before anyone can write it
          LK_FOR_RECORDS({
            if (lk_slot(r) == hslot) {
              const uint32_t idx = lk_idx(r), op = lk_op(r), w = idx >> 5, below = (1u << (idx & 31u)) - 1u;
              const uint32_t mw = Bm[w] & below;
              const uint32_t prev = mw ? w * 32 + 31 - (uint32_t)__clz(mw) : Lw[w];  // last lock-writing op below me
              const uint32_t lock_before = prev != ~0u ? (Bacq[prev >> 5] >> (prev & 31u)) & 1u : st.x;
              uint32_t code = 0, rv = 0;
              Ops::closed(op, lock_before, st.y + Pw[w] + (uint32_t)__popc(Bcom[w] & below), code, rv);
              Ops::write_reply(rep, V, idx, op, code, rv);
            }
          })
          if (t == 0) {
            const uint32_t lastop = Hs[7];
            if (lastop != ~0u) st.x = (Bacq[lastop >> 5] >> (lastop & 31u)) & 1u;
            st.y += Hs[6];
          }
        } else {
          // lock_2pl: the slot's requests as FOUR index bitmaps, one per op class.  A GROUP = the requests whose index falls
          // into one 64-bit word: consecutive requests of the slot, whatever their number.  Most groups cannot change the
          // counters' mode, whatever order their requests come in:
          //   HELD (num_ex != 0), no exclusive RELEASE in the group: every ACQUIRE is rejected, num_sh -= shared RELEASEs;
          //   FREE (num_ex == 0), no exclusive RELEASE, and no exclusive ACQUIRE that could find num_sh == 0 (num_sh
          //   stays above 0 even if all the group's shared RELEASEs came first): every shared ACQUIRE is granted,
          //   every exclusive one rejected, num_sh += shared ACQUIREs - shared RELEASEs.
          // One wave takes 64 groups at a time: ASSUMING the mode holds, num_sh at every group's entry is a prefix sum
          // over the groups before it, and every lane checks its own group; up to the first group that is not inert the
          // assumption was right -- that group is walked (walk64_mask: its word IS the lane mask), and the rest of the 64
          // is checked again from the state it leaves.  Steps = words / 64 + groups walked, where r03 / r04 ranked the
          // slot's requests into windows of 8,192 and went through their groups of 64 one by one (55 us for a slot of
          // 12,800 requests: a lid that a fifth of the closed-loop workers keep retrying; r05 trace).  The grant bits
          // replace the shared-ACQUIRE bitmap; the replies are written from them.
          uint64_t *G64 = (uint64_t *)Bm;
          const uint64_t *AX64 = (const uint64_t *)Bax, *RS64 = (const uint64_t *)Brs, *RX64 = (const uint64_t *)Brx;
          if (wave == 0) {
            uint32_t la = (uint32_t)__builtin_amdgcn_readfirstlane((int)st.x), lb = (uint32_t)__builtin_amdgcn_readfirstlane((int)st.y);
            const uint32_t ng = (n + 63) / 64;  // <= TPL_HOT_NMAX / 64
            for (uint32_t g0 = 0; g0 < ng; g0 += 64) {
              const uint32_t g = g0 + lane;
              const bool in = g < ng;
              const uint64_t as = in ? G64[g] : 0, ax = in ? AX64[g] : 0, rs = in ? RS64[g] : 0, rx = in ? RX64[g] : 0;
              const uint32_t nas = (uint32_t)__popcll(as), nrs = (uint32_t)__popcll(rs);
              uint64_t pend = __ballot(in && (as | ax | rs | rx) != 0);  // groups without a request of the slot: nothing to do
              while (pend) {
                const bool mine = (pend >> lane) & 1ull;
                const uint32_t dl = mine ? (la != 0 ? 0u - nrs : nas - nrs) : 0u;  // what my group adds to num_sh if it is inert
                uint32_t tot, pre = lk_excl_scan_u32(dl, &tot);
                const uint32_t lb_in = lb + pre;
                const bool inert = la != 0 ? rx == 0 : (rx == 0 && (ax == 0 || (lb_in > nrs && lb_in <= 0xFFFFFFFFu - nas)));
                const uint64_t stop = __ballot(mine && !inert);
                const uint64_t ok = stop ? pend & ((stop & (0 - stop)) - 1ull) : pend;  // the groups before the first one that is not
                if ((ok >> lane) & 1ull) G64[g] = la == 0 ? as : 0ull;
                if (!stop) { lb += tot; break; }
                const int f = __ffsll((unsigned long long)stop) - 1;
                lb += (uint32_t)__builtin_amdgcn_readlane((int)pre, f);  // (the groups before f)
                // group f, request by request as far as the mode changes: lane l = request index 64 (g0 + f) + l
                const uint64_t fas = lk_readlane_u64(as, f), fax = lk_readlane_u64(ax, f), frs = lk_readlane_u64(rs, f), frx = lk_readlane_u64(rx, f);
                const bool v = ((fas | fax | frs | frx) >> lane) & 1ull;
                const uint32_t op = (uint32_t)((fax >> lane) & 1ull) + 2u * (uint32_t)((frs >> lane) & 1ull) + 3u * (uint32_t)((frx >> lane) & 1ull);
                uint2 sw = make_uint2(la, lb);
                const uint64_t Gf = Ops::walk64_mask(v, op, sw);
                la = sw.x; lb = sw.y;
                if ((int)lane == f) G64[g] = Gf;
                pend &= ~(ok | (1ull << f));
              }
            }
            st.x = la; st.y = lb;
          }
          __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
          __syncthreads();
          // the replies: a GRANT for every bit that is left (every other request of the slot has its reply: k_lock_count's)
          for (uint32_t w = t; w < (n + 31) / 32; w += KVB_T)
            for (uint32_t m = Bm[w]; m; m &= m - 1) Ops::write_reply(rep, V, w * 32 + (uint32_t)__ffs((int)m) - 1, 0, 2, 0);
        }
        if (t == 0 && (st.x != st_in.x || st.y != st_in.y)) table[hslot] = st;
        if (tw && t == 0 && bi == vb) tw[12] = __builtin_amdgcn_s_memrealtime();
        hslot_done = hslot;
        c_rest = c - hot_n;
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        __syncthreads();
      }
    }
    uint32_t nwin = 1;
    if (c_rest > KVB_NMAX) {
      for (uint32_t w = t; w < KVB_NBK / 2; w += KVB_T) Bcnt[w] = 0;
      __syncthreads();
      LK_FOR_RECORDS({
        if (lk_slot(r) != hslot_done) {
          const uint32_t b = lk_idx(r) >> bs;
          atomicAdd(&Bcnt[b >> 1], 1u << (16 * (b & 1)));
        }
      })
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
#pragma unroll
      for (uint32_t j = 0; j < 4; j++) {
        Bwin[4 * t + j] = (uint16_t)(base / wcap);
        base += (cw[j >> 1] >> (16 * (j & 1))) & 0xFFFF;
      }
      nwin = (c_rest - 1) / wcap + 1;
    }
    __syncthreads();
    if (t == 0) {
      bin_cnt[bin] = 0;  // every thread has read c
      bin_off[bin] = KV_NONE;  // ... and where the bin's records beyond 64 were: a direct pass's k_lock_count names a region anew
      if (bigrec) atomicAdd(&stats->big_bin_requests, (unsigned long long)c);  // (k_kv_scan_place's job otherwise)
    }
    for (uint32_t win = 0; win < nwin; win++) {
      if (t == 0) { Swn = 0; Snx = 0; }
      __syncthreads();
      if (rest_lds) {  // the dominant slot's reply pass has collected the bin's other records
        for (uint32_t k = t; k < c_rest; k += KVB_T) Sk[k] = Sk[3 * (KVB_NMAX / 4) + k];
        if (t == 0) Swn = c_rest;
      } else if (c_rest <= KVB_NMAX && hslot_done == KV_NONE) {
        for (uint32_t k = t; k < c; k += KVB_T) Sk[k] = rec_at(k);
        if (t == 0) Swn = c;
      } else {
        for (uint32_t k0 = 0; k0 < c; k0 += 8 * KVB_T) {  // (8 loads per thread in flight, as every pass over the bin's records)
          uint64_t r8[8];
#pragma unroll
          for (uint32_t j = 0; j < 8; j++) {
            const uint32_t k = k0 + j * KVB_T + t;
            r8[j] = k < c ? rec_at(k) : 0;
          }
#pragma unroll
          for (uint32_t j = 0; j < 8; j++) {
            const uint32_t k = k0 + j * KVB_T + t;
            const uint64_t r = r8[j];
            const bool in = k < c && lk_slot(r) != hslot_done && (c_rest <= KVB_NMAX || Bwin[lk_idx(r) >> bs] == win);
            const uint64_t im = __ballot(in);
            uint32_t base = 0;
            if (lane == 0 && im) base = atomicAdd(&Swn, (uint32_t)__popcll(im));
            base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
            if (in) Sk[base + (uint32_t)__popcll(im & lanemask_lt())] = r;
          }
        }
      }
      __syncthreads();
      uint32_t m = Swn;
      if (m == 0) continue;  // workgroup-uniform
      LK_STAMP(1);
      if (m <= 64) {  // (what a dominant slot leaves of its bin, mostly: one wave, as a small bin)
        if (wave == 0) lk_chunk<Ops>(rep, V, table, wave_sort_u64(lane < m ? Sk[lane] : ~0ull), lane < m, true, true);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        __syncthreads();
        LK_STAMP(5);
        continue;
      }
      // ---- the stretch's DOMINANT SLOT (lock_fasst; a hot lid: most of a big bin is one slot) without sorting the
      // stretch: only its lock-writing ops (ACQUIRE / ABORT / COMMIT, a minority) are put in request order -- one LDS sort
      // of <= 1024 words -- and every request of the slot finds by binary search on its index how many precede it:
      // lock seen = what the last of them left, version seen = ver0 + the COMMITs among them.
      if (Ops::CLOSED && m >= hot_min) {
        uint64_t cand[8];
        uint32_t cc[8];
#pragma unroll
        for (uint32_t k = 0; k < 8; k++) { cand[k] = Sk[(uint32_t)(((uint64_t)m * k) >> 3)] >> 23; cc[k] = 0; }
        if (t < 16) Hs[t] = 0;
        __syncthreads();
        for (uint32_t p = t; p < m; p += KVB_T) {
          const uint64_t pf = Sk[p] >> 23;
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
        const uint64_t hslot = cand[best];
        __syncthreads();
        if (hot_n >= hot_min && 2 * hot_n >= m) {  // workgroup-uniform
          if (t < 16) Hs[t] = 0;  // [1] lock-writing ops of the slot, [2] append cursor
          __syncthreads();
          uint32_t nord = 0;
          for (uint32_t p = t; p < m; p += KVB_T) nord += (Sk[p] >> 23) == hslot && lk_op(Sk[p]) != 0;
          for (int d = 32; d > 0; d >>= 1) nord += __shfl_xor(nord, d, 64);
          if (lane == 0 && nord) atomicAdd(&Hs[1], nord);
          __syncthreads();
          const uint32_t nM = Hs[1];
          if (nM <= KVB_MMAX) {  // workgroup-uniform
            for (uint32_t p0 = 0; p0 < m; p0 += KVB_T) {
              const uint32_t p = p0 + t;
              const uint64_t cur = p < m ? Sk[p] : 0;
              const bool in = p < m && (cur >> 23) == hslot && lk_op(cur) != 0;
              const uint64_t im = __ballot(in);
              uint32_t base = 0;
              if (lane == 0 && im) base = atomicAdd(&Hs[2], (uint32_t)__popcll(im));
              base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
              if (in) Mk[base + (uint32_t)__popcll(im & lanemask_lt())] = ((uint64_t)lk_idx(cur) << 12) | p;
            }
            __syncthreads();
            kvb_sort_stretch(Mk, nM);  // in registers / by shuffle inside a wave, the wide steps through LDS
            // COMMITs among the first j ops of M (nM + 1 rows): thread t owns ops 2t, 2t + 1
            {
              const uint32_t j0 = 2 * t, j1 = 2 * t + 1;
              const bool c0 = j0 < nM && lk_op(Sk[Mk[j0] & 4095u]) == 3, c1 = j1 < nM && lk_op(Sk[Mk[j1] & 4095u]) == 3;
              uint32_t wt, wx = wave_excl_scan_u32((uint32_t)c0 + (uint32_t)c1, &wt);
              if (lane == 63) Sred[wave] = wt;
              __syncthreads();
              for (uint32_t w = 0; w < wave; w++) wx += Sred[w];
              if (j0 <= nM) Mcc[j0] = (uint16_t)wx;
              if (j1 <= nM) Mcc[j1] = (uint16_t)(wx + c0);
              if (j1 + 1 == nM) Mcc[nM] = (uint16_t)(wx + c0 + c1);
            }
            const uint2 st0 = table[(uint32_t)hslot];  // one word for the whole slot (workgroup-uniform address)
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            __syncthreads();  // the tables are built and everyone holds the slot's word
            for (uint32_t p = t; p < m; p += KVB_T) {
              const uint64_t cur = Sk[p];
              if ((cur >> 23) != hslot) continue;
              const uint32_t op = lk_op(cur), key32 = lk_idx(cur) << 12;
              uint32_t lo = 0, hi = nM;  // lock-writing ops of the slot with a smaller request index
              while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                if (Mk[mid] < key32) lo = mid + 1; else hi = mid;
              }
              const uint32_t lock_before = lo ? (uint32_t)(lk_op(Sk[Mk[lo - 1] & 4095u]) == 1) : st0.x;
              uint32_t code = 0, rv = 0;
              Ops::closed(op, lock_before, st0.y + Mcc[lo], code, rv);
              Ops::write_reply(rep, V, lk_idx(cur), op, code, rv);
            }
            if (t == 0 && nM) {
              uint2 fin;
              fin.x = (uint32_t)(lk_op(Sk[Mk[nM - 1] & 4095u]) == 1);
              fin.y = st0.y + Mcc[nM];
              if (fin.x != st0.x || fin.y != st0.y) table[(uint32_t)hslot] = fin;
            }
            // what is left of the stretch moves to the front (destinations never overtake unread sources)
            __syncthreads();
            if (t == 0) Hs[2] = 0;
            __syncthreads();
            for (uint32_t p0 = 0; p0 < m; p0 += KVB_T) {
              const uint32_t p = p0 + t;
              const uint64_t cur = p < m ? Sk[p] : 0;
              const bool keep = p < m && (cur >> 23) != hslot;
              const uint64_t km = __ballot(keep);
              __syncthreads();
              uint32_t base = 0;
              if (lane == 0 && km) base = atomicAdd(&Hs[2], (uint32_t)__popcll(km));
              base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
              if (keep) Sk[base + (uint32_t)__popcll(km & lanemask_lt())] = cur;
              __syncthreads();
            }
            m = Hs[2];
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            __syncthreads();
            if (m == 0) continue;  // workgroup-uniform
          }
        }
      }
      LK_STAMP(2);
      kvb_sort_stretch(Sk, m);
      LK_STAMP(3);
      const uint32_t ntile = (m + KVB_T - 1) / KVB_T;
      // slot heads of the sorted stretch
      for (uint32_t j = 0; j < ntile; j++) {
        const uint32_t p = j * KVB_T + t;
        const bool valid = p < m;
        const uint64_t hm = __ballot(valid && (p == 0 || lk_slot(Sk[p]) != lk_slot(Sk[p - 1])));
        if (lane == 0) Mhead[p >> 6] = hm;
      }
      for (uint32_t w = ntile * KVB_W + t; w < KVB_NW; w += KVB_T) Mhead[w] = 0;
      __syncthreads();
      if (wave == 0) kvb_build_edge(Mhead, Ehead);
      __syncthreads();
      // chunks of 64: the slots inside one chunk, all chunks in parallel; list the slots that cross chunks
      for (uint32_t j = 0; j < ntile; j++) {
        const uint32_t p = j * KVB_T + t, p0 = p & ~63u;
        const bool valid = p < m;
        if (p0 < m) {  // wave-uniform
          const uint32_t pl = min(p0 + 63u, m - 1);  // last valid position of my chunk
          const bool first_is_head = kvb_bit(Mhead, p0), last_is_tail = pl + 1 == m || kvb_bit(Mhead, pl + 1);
          lk_chunk<Ops>(rep, V, table, valid ? Sk[p] : ~0ull, valid, first_is_head, last_is_tail);
          if (valid && kvb_bit(Mhead, p)) {
            const int nx = kvb_first(Mhead, Ehead, p + 1);
            const uint32_t b = nx >= 0 ? (uint32_t)nx : m;
            if ((p >> 6) != ((b - 1) >> 6)) {
              const uint32_t k = atomicAdd(&Snx, 1u);
              Xs[k][0] = p; Xs[k][1] = b;
            }
          }
        }
      }
      __syncthreads();
      LK_STAMP(4);
      if (Ops::CLOSED) {
        // the slots that cross chunks, every request in parallel: masks over the whole sorted stretch + O(1) range tables
        // answer "last lock-writing op below me in my slot" and "COMMITs below me in my slot"
        if (Snx) {  // workgroup-uniform
          for (uint32_t j = 0; j < ntile; j++) {
            const uint32_t p = j * KVB_T + t;
            const bool valid = p < m;
            const uint32_t op = valid ? lk_op(Sk[p]) : 0;
            const uint64_t m1 = __ballot(valid && op != 0), m2 = __ballot(valid && op == 1), m3 = __ballot(valid && op == 3);
            if (lane == 0) { Mset[p >> 6] = m1; Macq[p >> 6] = m2; Mcom[p >> 6] = m3; }
          }
          for (uint32_t w = ntile * KVB_W + t; w < KVB_NW; w += KVB_T) { Mset[w] = 0; Macq[w] = 0; Mcom[w] = 0; }
          __syncthreads();
          if (wave == 0) kvb_build_edge(Mset, Eset);
          if (wave == 1) kvb_build_pop(Mcom, Pcom);
          __syncthreads();
          uint2 st[KVB_NMAX / KVB_T];
          uint32_t sa[KVB_NMAX / KVB_T], sb[KVB_NMAX / KVB_T];
#pragma unroll
          for (uint32_t j = 0; j < KVB_NMAX / KVB_T; j++) {
            const uint32_t p = j * KVB_T + t;
            sa[j] = sb[j] = 0;
            st[j] = make_uint2(0, 0);
            if (p < m) {
              const uint32_t a = (uint32_t)kvb_last(Mhead, Ehead, 0, p + 1);
              const int nx = kvb_first(Mhead, Ehead, p + 1);
              const uint32_t b = nx >= 0 ? (uint32_t)nx : m;
              if ((a >> 6) != ((b - 1) >> 6)) {  // my slot crosses chunks
                sa[j] = a; sb[j] = b;
                st[j] = table[lk_slot(Sk[p])];
              }
            }
          }
          __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
          __syncthreads();  // every request has read its slot's word: the write-backs below cannot be seen by them
#pragma unroll
          for (uint32_t j = 0; j < KVB_NMAX / KVB_T; j++) {
            const uint32_t p = j * KVB_T + t, a = sa[j], b = sb[j];
            if (b == 0) continue;
            const uint64_t w = Sk[p];
            const uint32_t op = lk_op(w);
            const int prev = kvb_last(Mset, Eset, a, p);
            const uint32_t lock_before = prev >= 0 ? (uint32_t)kvb_bit(Macq, (uint32_t)prev) : st[j].x;
            uint32_t code = 0, rv = 0;
            Ops::closed(op, lock_before, st[j].y + kvb_popc(Mcom, Pcom, a, p), code, rv);
            Ops::write_reply(rep, V, lk_idx(w), op, code, rv);
            if (p + 1 == b) {  // the slot's last request writes the word back
              const int last = kvb_last(Mset, Eset, a, b);
              uint2 fin;
              fin.x = last >= 0 ? (uint32_t)kvb_bit(Macq, (uint32_t)last) : st[j].x;
              fin.y = st[j].y + kvb_popc(Mcom, Pcom, a, b);
              if (fin.x != st[j].x || fin.y != st[j].y) table[lk_slot(w)] = fin;
            }
          }
        }
      } else {
      // the slots that cross chunks: one wave each, the slot's word in registers, 64 requests at a time
      for (uint32_t k = wave; k < Snx; k += KVB_W) {
        const uint32_t a = Xs[k][0], b = Xs[k][1];
        const uint32_t slot = lk_slot(Sk[a]);
        uint2 st = table[slot];  // wave-uniform address
        st.x = (uint32_t)__builtin_amdgcn_readfirstlane((int)st.x);
        st.y = (uint32_t)__builtin_amdgcn_readfirstlane((int)st.y);
        const uint2 st_in = st;
        for (uint32_t base = a; base < b; base += 64) {  // request order = lane order; 64 requests per step (Ops::walk64)
          const uint32_t p = base + lane;
          const bool valid = p < b;
          const uint64_t w = valid ? Sk[p] : 0;
          const uint32_t op = lk_op(w);
          uint32_t code = 0;
          Ops::walk64(valid, op, st, code);
          if (valid) Ops::write_reply(rep, V, lk_idx(w), op, code, 0);
        }
        if (lane == 0 && (st.x != st_in.x || st.y != st_in.y)) table[slot] = st;
      }
      }
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
      __syncthreads();  // the next stretch sees this stretch's stores; LDS is free again
      LK_STAMP(5);
    }
  }
}

// ---- k_lock_resolve: every bin of the pass, one launch ----------------------------------------------------------------
// Workgroups 0 .. KVB_GRID-1 walk the big-bin list (most exit at once), each wave of the others resolves one bin of <= 64
// records.  The two kinds own disjoint bins, hence disjoint slots, so the hot slots are resolved beside the bulk of the
// pass -- as the kv passes always did.  (r01-r03a: two launches, k_lock_resolve_big then the 256-thread
// k_lock_resolve: for a 64k-request pass one launch gap of ~5 us and the small bins' 4 us behind the hot slot's 23.)
template <class Ops>
__global__ void __launch_bounds__(KVB_T)
k_lock_resolve(uint8_t *rep, uint32_t n, uint32_t pbits, uint2 *__restrict__ table, uint32_t *__restrict__ bin_cnt,
               const uint64_t *__restrict__ bins, const uint32_t *__restrict__ big, uint32_t *bin_off,
               const uint64_t *__restrict__ ovf, uint32_t hot_min, uint64_t *trace, dint_view V,
               const uint64_t *__restrict__ bigrec, uint32_t *big_next, uint32_t *blk_pub_next, dint_dev_stats *stats) {
  if (bigrec && blockIdx.x == KVB_GRID) {  // direct passes have no k_kv_scan_place: the next pass's counters, here
    if (threadIdx.x < 4) big_next[threadIdx.x] = 0;
    for (uint32_t k = threadIdx.x; k < 1024; k += KVB_T) blk_pub_next[k] = 0;
  }
  if (blockIdx.x < KVB_GRID) lk_big_bins<Ops>(rep, n, table, bin_cnt, bins, big, bin_off, ovf, hot_min, trace, V, blockIdx.x, KVB_GRID, bigrec, stats);
  else lk_small_bin<Ops>(rep, pbits, table, bin_cnt, bins, V, (blockIdx.x - KVB_GRID) * KVB_W + (threadIdx.x >> 6));
}

// ---- k_lock_pass (r06): the resolve stage of pass k AND the count stage of pass k + 1 in one launch ----------------------------
// A lock pass of 65,536 requests is launch-bound: count 11 us + resolve 20 us (rocprofv3) and a launch boundary between them,
// 33 us per pass for 1.8 MB of traffic.  The count stage touches no table -- it reads the NEXT batch and fills that batch's own
// scratch set (bins, counters, big-bin list, direct regions; engine.hip keeps two) -- so with the next batch announced
// (dint_submit_device_ahead, or the next 65,536 requests of one long dint_submit_device) it rides beside this batch's resolve
// workgroups, which are placed first: a pass is one launch, as long as its hot slot's workgroup.  Direct passes only (<= 65,536).
template <int WL, class Ops>
__global__ void __launch_bounds__(KVB_T)
k_lock_pass(uint8_t *rep, uint32_t n, uint32_t pbits, uint2 *__restrict__ table, uint32_t *__restrict__ bin_cnt,
            const uint64_t *__restrict__ bins, const uint32_t *__restrict__ big, uint32_t *bin_off,
            const uint64_t *__restrict__ ovf, uint32_t hot_min, uint64_t *trace, dint_view V,
            const uint64_t *__restrict__ bigrec, uint32_t *big_next, uint32_t *blk_pub_next, dint_dev_stats *stats,
            uint32_t n_resolve, lk_count_args C) {
  if (blockIdx.x >= n_resolve) {
    lk_count_body<WL, KVB_T>(C, blockIdx.x - n_resolve);
    return;
  }
  if (blockIdx.x == KVB_GRID) {
    if (threadIdx.x < 4) big_next[threadIdx.x] = 0;
    for (uint32_t k = threadIdx.x; k < 1024; k += KVB_T) blk_pub_next[k] = 0;
  }
  if (blockIdx.x < KVB_GRID) lk_big_bins<Ops>(rep, n, table, bin_cnt, bins, big, bin_off, ovf, hot_min, trace, V, blockIdx.x, KVB_GRID, bigrec, stats);
  else lk_small_bin<Ops>(rep, pbits, table, bin_cnt, bins, V, (blockIdx.x - KVB_GRID) * KVB_W + (threadIdx.x >> 6));
}

// ------------------------------------------------------------------------------------------
// stages: 1 = count + scan / place, 2 = resolve, 3 = both (one stream).  The two halves of a pass touch disjoint state but
// for the pass scratch `s`: the first reads the requests and writes records, counters and the replies' default bytes; the
// second reads the records and owns the table (dint_launch_lock_stage: the engine may run stage 1 of pass k + 1 beside stage 2
// of pass k, on scratch sets of their own)
template <int WL, class Ops>
static void launch_locks(const void *d_req, void *d_rep, uint32_t n, uint2 *table, dint_mod slots, dint_shard shard,
                         dint_scratch s, hipStream_t st, hipEvent_t *ev, const dint_view &view, int stages = 3) {
  if (n == 0) return;
  const uint32_t P = dint_pick_bins_kv(n);
  uint32_t pbits = 0;
  while ((1u << pbits) < P) pbits++;
  // DIRECT big bins: a pass of at most LK_DIRECT_NMAX requests on an engine that holds the regions (s.kbins: 1,024 regions of
  // LK_DIRECT_NMAX records -- a lock engine's `kbins` is this, engine.hip; s.bin_off holds the regions' names, KV_NONE between
  // passes) is TWO launches: k_lock_count stores a big bin's records beyond 64 straight into its region, the resolve kernel
  // does what is left of k_kv_scan_place's work.  DINT_LOCK_NO_DIRECT=1: r01-r05's three launches (tests run both).
  uint64_t *bigrec = s.kbins && n <= LK_DIRECT_NMAX && !getenv("DINT_LOCK_NO_DIRECT") ? (uint64_t *)s.kbins : nullptr;
  if (stages & 1) {
    if (ev) hipEventRecord(ev[0], st);
    const lk_count_args C = {(const uint8_t *)d_req, (uint8_t *)d_rep, n, slots, shard, pbits, s.bin_cnt, s.bins, s.big, s.ovl, s.stats, view, bigrec, s.bin_off};
    hipLaunchKernelGGL((k_lock_count<WL>), dim3((n + KV_TB - 1) / KV_TB), dim3(KV_TB), 0, st, C);
    if (ev) hipEventRecord(ev[1], st);
    if (!bigrec)
      hipLaunchKernelGGL(k_kv_scan_place, dim3(KV_PLACE_GRID), dim3(KV_TB), 0, st, (const uint32_t *)s.bin_cnt, s.bin_off,
                         (const uint32_t *)s.big, s.big_next, s.blk_pub_next, (uint32_t *)nullptr, s.stats, (const uint4 *)s.ovl, s.ovf);
  }
  if (stages & 2) {
    if (ev) hipEventRecord(ev[2], st);
    hipLaunchKernelGGL((k_lock_resolve<Ops>), dim3(KVB_GRID + (P + KVB_W - 1) / KVB_W), dim3(KVB_T), 0, st, (uint8_t *)d_rep, n, pbits,
                       table, s.bin_cnt, (const uint64_t *)s.bins, (const uint32_t *)s.big, s.bin_off,
                       (const uint64_t *)s.ovf, dint_hot_min("DINT_LOCK_HOT_MIN", KVB_HOT_MIN_LOCKS) | (getenv("DINT_LOCK_MISGUESS") ? 0x80000000u : 0u),  // (tests: the samples name no slot)
                       s.lock_trace, view, (const uint64_t *)bigrec, s.big_next, s.blk_pub_next, s.stats);
    if (ev) hipEventRecord(ev[3], st);
  }
}

// the resolve stage of one pass (scratch set `s`: its count stage has run) and the count stage of the next (`sn`, `next_*`) in ONE
// launch; both passes direct (n, next_n <= LK_DIRECT_NMAX, regions allocated).  false: not possible -- the caller launches the stages
template <int WL, class Ops>
static bool launch_lock_pass_fused(void *d_rep, uint32_t n, uint2 *table, dint_mod slots, dint_shard shard, dint_scratch s, const dint_view &view,
                                   const void *next_req, void *next_rep, uint32_t next_n, dint_scratch sn, const dint_view &next_view, hipStream_t st) {
  if (!n || !next_n || n > LK_DIRECT_NMAX || next_n > LK_DIRECT_NMAX || !s.kbins || !sn.kbins || getenv("DINT_LOCK_NO_DIRECT") || getenv("DINT_LOCK_NO_FUSE")) return false;
  const uint32_t P = dint_pick_bins_kv(n), Pn = dint_pick_bins_kv(next_n);
  uint32_t pbits = 0, pbits_n = 0;
  while ((1u << pbits) < P) pbits++;
  while ((1u << pbits_n) < Pn) pbits_n++;
  const lk_count_args C = {(const uint8_t *)next_req, (uint8_t *)next_rep, next_n, slots, shard, pbits_n, sn.bin_cnt, sn.bins, sn.big, sn.ovl, sn.stats,
                           next_view, (uint64_t *)sn.kbins, sn.bin_off};
  const uint32_t n_resolve = KVB_GRID + (P + KVB_W - 1) / KVB_W;
  hipLaunchKernelGGL((k_lock_pass<WL, Ops>), dim3(n_resolve + (next_n + KVB_T - 1) / KVB_T), dim3(KVB_T), 0, st, (uint8_t *)d_rep, n, pbits, table, s.bin_cnt,
                     (const uint64_t *)s.bins, (const uint32_t *)s.big, s.bin_off, (const uint64_t *)s.ovf,
                     dint_hot_min("DINT_LOCK_HOT_MIN", KVB_HOT_MIN_LOCKS) | (getenv("DINT_LOCK_MISGUESS") ? 0x80000000u : 0u), s.lock_trace, view,
                     (const uint64_t *)s.kbins, s.big_next, s.blk_pub_next, s.stats, n_resolve, C);
  return true;
}
bool dint_launch_lock_fused(uint32_t workload, void *d_rep, uint32_t n, uint2 *table, dint_mod slots, dint_shard shard, dint_scratch s, const dint_view &view,
                            const void *next_req, void *next_rep, uint32_t next_n, dint_scratch sn, const dint_view &next_view, hipStream_t st) {
  if (workload == 0) return launch_lock_pass_fused<0, FasstOps>(d_rep, n, table, slots, shard, s, view, next_req, next_rep, next_n, sn, next_view, st);
  return launch_lock_pass_fused<1, TplOps>(d_rep, n, table, slots, shard, s, view, next_req, next_rep, next_n, sn, next_view, st);
}

// one half of a lock pass on `st` (declared in engine.hip: DINT_FLAG_INPUTS_READY)
void dint_launch_lock_stage(uint32_t workload, int stage, const void *d_req, void *d_rep, uint32_t n, uint2 *table, dint_mod slots,
                            dint_shard shard, dint_scratch s, hipStream_t st, const dint_view &view) {
  if (workload == 0 /* DINT_WL_FASST */) launch_locks<0, FasstOps>(d_req, d_rep, n, table, slots, shard, s, st, nullptr, view, stage);
  else launch_locks<1, TplOps>(d_req, d_rep, n, table, slots, shard, s, st, nullptr, view, stage);
}

void dint_launch_fasst(const void *d_req, void *d_rep, uint32_t n, uint2 *table, dint_mod slots, dint_shard shard,
                       dint_scratch s, hipStream_t st, hipEvent_t *ev, const dint_view &view) {
  launch_locks<0, FasstOps>(d_req, d_rep, n, table, slots, shard, s, st, ev, view);
}
void dint_launch_2pl(const void *d_req, void *d_rep, uint32_t n, uint2 *table, dint_mod slots, dint_shard shard,
                     dint_scratch s, hipStream_t st, hipEvent_t *ev, const dint_view &view) {
  launch_locks<1, TplOps>(d_req, d_rep, n, table, slots, shard, s, st, ev, view);
}

// home shard of each lock request (multi-GPU routing): global slot % shard_count
__global__ void __launch_bounds__(256)
k_home_lid(const uint8_t *__restrict__ req, uint32_t msg_size, uint32_t n, dint_mod slots, uint32_t shard_count,
           uint8_t *__restrict__ home) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  uint32_t lid;
  __builtin_memcpy(&lid, req + (size_t)i * msg_size + 1, 4);  // lid sits at byte 1 in both structs
  home[i] = (uint8_t)(dint_fastmod(dint_hash_lid(lid), slots) % shard_count);
}
void dint_launch_home_lid(const void *d_req, uint32_t msg_size, uint32_t n, dint_mod slots, uint32_t shard_count,
                          uint8_t *d_home, hipStream_t st) {
  if (n == 0) return;
  hipLaunchKernelGGL(k_home_lid, dim3((n + 255) / 256), dim3(256), 0, st, (const uint8_t *)d_req, msg_size, n,
                     slots, shard_count, d_home);
}
