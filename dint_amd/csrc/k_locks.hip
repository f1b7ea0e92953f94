// k_locks.hip -- lock_fasst and lock_2pl on gfx950: one pass = two kernels.
//
// Reference semantics (serial, one message at a time):
//   lock_fasst/udp/server.cc:78-119   READ / ACQUIRE_LOCK / ABORT / COMMIT on locks[], ver_table[]
//   lock_2pl/udp/server.cc:70-122     ACQUIRE shared|exclusive / RELEASE on num_ex[], num_sh[]
// Both index a direct-mapped table with slot = fasthash64(&lid,4,0xdeadbeef) % n_slots; lids that
// collide modulo n_slots share one lock word -- that aliasing is part of the semantics.
//
// GPU formulation.  Ops on different slots commute, ops on one slot must apply in request order.
//   kernel A  k_lock_scatter : one thread per request: decode, hash, slot, append a 64-bit record
//             {slot, idx, op} to bin = (slot >> 4) & (P-1) (atomic reservation; order inside a bin is
//             arbitrary), and copy the request bytes to the reply array.
//   kernel B  k_lock_resolve : one wave per bin: restore request order with a bitmap rank over idx,
//             group the window's records by slot in an LDS hash, fetch every distinct slot's 8-byte
//             {a,b} word from HBM once (all loads in flight together), then walk the records in
//             request order 64 at a time: lanes whose slot is unique in their chunk apply their op
//             directly on the LDS copy; slots hit by several lanes of a chunk are resolved in lane
//             (= request) order -- lock_fasst in closed form with ballots, lock_2pl by a short
//             wave-uniform loop.  Dirty words are written back once.  No global atomics, no locks.
// The table is an array of uint2 in HBM: fasst {lock, ver}, 2pl {num_ex, num_sh}; (slot >> 4) keeps the
// 16 slots of a 128-byte line in one bin, hence in one wave and one XCD's L2.
#include "dint_kernels.h"

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

// ------------------------------------------------------------------------------------------
template <int WL>  // 0 = lock_fasst, 1 = lock_2pl
__global__ void __launch_bounds__(256)
k_lock_scatter(const uint8_t *__restrict__ req, uint8_t *rep, uint32_t n, dint_mod slots, dint_shard shard,
               uint32_t pmask, uint32_t *__restrict__ bin_cnt, uint64_t *__restrict__ bins,
               dint_dev_stats *__restrict__ stats, dint_view V) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  bool live;
  const size_t off = dint_view_off(V, i, WL == 0 ? sizeof(fasst_msg) : sizeof(tpl_msg), &live);
  if (!live) return;  // padding slot of a segmented pass
  uint32_t lid, op;
  bool ok;
  if (WL == 0) {
    fasst_msg m = *(const fasst_msg *)(req + off);
    if (rep != req) *(fasst_msg *)(rep + off) = m;
    lid = m.lid;
    op = m.type;  // 0 READ, 1 ACQUIRE_LOCK, 2 ABORT, 3 COMMIT
    ok = op <= 3;
  } else {
    tpl_msg m = *(const tpl_msg *)(req + off);
    if (rep != req) *(tpl_msg *)(rep + off) = m;
    lid = m.lid;
    if (m.action == 0) {  // ACQUIRE: op 0 shared, 1 exclusive; any other lock type panics in the reference
      ok = m.type <= 1;
      op = m.type;
    } else if (m.action == 1) {  // RELEASE: op 2 shared, 3 exclusive, 4 = unknown type (ack, no change)
      ok = true;
      op = 2u + (m.type <= 1 ? m.type : 2u);
    } else {
      ok = false;
      op = 0;
    }
  }
  if (!ok) {
    atomicAdd(&stats->bad_requests, 1ULL);
    return;
  }
  uint64_t g = dint_fastmod(dint_hash_lid(lid), slots);
  uint32_t local = (uint32_t)g;
  if (shard.count > 1) {
    if ((uint32_t)(g % shard.count) != shard.index) {
      atomicAdd(&stats->foreign_requests, 1ULL);
      return;
    }
    local = (uint32_t)(g / shard.count);
  }
  const uint32_t bin = (local >> 4) & pmask;
  const uint32_t pos = atomicAdd(&bin_cnt[bin], 1u);
  bins[(size_t)bin * DINT_MICRO + pos] = dint_rec(local, i, op, 0);
}

// ------------------------------------------------------------------------------------------
struct FasstOps {
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
  // all lanes of `same` target one slot; resolve them in lane order in closed form:
  //   lock before lane l = value written by the last lock-writing op (ACQUIRE sets 1 whether granted
  //   or not; ABORT/COMMIT set 0) below l, else the initial lock;  ver before l = ver0 + #COMMITs below l.
  __device__ static void resolve_group(uint64_t same, bool mine, uint32_t op, uint2 *Hst, uint32_t *Hfl,
                                       uint32_t se, bool is_leader, uint32_t &code, uint32_t &rv) {
    const uint2 st0 = Hst[se];
    const uint64_t m_set = __ballot(mine && op != 0);
    const uint64_t m_acq = __ballot(mine && op == 1);
    const uint64_t m_com = __ballot(mine && op == 3);
    (void)same;
    if (mine) {
      const uint64_t lt = lanemask_lt();
      const uint64_t prev = m_set & lt;
      const uint32_t lock_before = prev ? (uint32_t)((m_acq >> (63 - __clzll(prev))) & 1ULL) : st0.x;
      const uint32_t ver_before = st0.y + (uint32_t)__popcll(m_com & lt);
      switch (op) {
        case 0: code = 4; rv = ver_before; break;
        case 1: code = lock_before ? 6 : 5; break;
        case 2: code = 7; break;
        default: code = 8; break;
      }
    }
    if (is_leader && m_set) {
      uint2 fin;
      fin.x = (uint32_t)((m_acq >> (63 - __clzll(m_set))) & 1ULL);
      fin.y = st0.y + (uint32_t)__popcll(m_com);
      Hst[se] = fin;
      atomicOr(&Hfl[se], 0x80000000u);
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
  __device__ static void write_reply(uint8_t *rep, const dint_view &V, uint32_t idx, uint32_t op, uint32_t code, uint32_t rv) {
    fasst_msg *m = (fasst_msg *)(rep + dint_view_off(V, idx, sizeof(fasst_msg)));
    m->type = (uint8_t)code;
    if (op == 0) m->ver = rv;  // ver is echoed on every non-READ reply
  }
};

struct TplOps {
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
  // counters have no closed form: apply the group's ops one by one in lane order (wave-uniform loop)
  __device__ static void resolve_group(uint64_t same, bool mine, uint32_t op, uint2 *Hst, uint32_t *Hfl,
                                       uint32_t se, bool is_leader, uint32_t &code, uint32_t &rv) {
    uint2 st = Hst[se];
    bool wr = false;
    const uint32_t lane = lane_id();
    (void)mine;
    for (uint64_t m = same; m; m &= m - 1) {
      const int l = __ffsll((unsigned long long)m) - 1;
      const uint32_t lop = __builtin_amdgcn_readlane(op, l);
      uint32_t lrv = 0;
      bool lwr = false;
      const uint32_t lcode = apply(lop, st, lrv, lwr);
      wr |= lwr;
      if ((int)lane == l) { code = lcode; rv = lrv; }
    }
    if (is_leader && wr) {
      Hst[se] = st;
      atomicOr(&Hfl[se], 0x80000000u);
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
  __device__ static void write_reply(uint8_t *rep, const dint_view &V, uint32_t idx, uint32_t op, uint32_t code, uint32_t rv) {
    (void)op; (void)rv;
    ((tpl_msg *)(rep + dint_view_off(V, idx, sizeof(tpl_msg))))->action = (uint8_t)code;
  }
};

template <class Ops>
__global__ void __launch_bounds__(64)
k_lock_resolve(uint8_t *rep, uint32_t n, uint2 *__restrict__ table, uint32_t *__restrict__ bin_cnt,
               const uint64_t *__restrict__ bins, dint_view V) {
  __shared__ dint_rank_lds R;
  __shared__ uint32_t Srec[DINT_WCAP];  // idx | entry << 16 | op << 26, in request order
  __shared__ uint32_t Hk[DINT_HSIZE];   // slot of each hash entry
  __shared__ uint2 Hst[DINT_HSIZE];     // its table word
  __shared__ uint32_t Hfl[DINT_HSIZE];  // low 16 bits: lanes of the current chunk on it; bit 31: dirty
  const uint32_t bin = blockIdx.x, lane = threadIdx.x;
  const uint64_t *recs = bins + (size_t)bin * DINT_MICRO;
  const uint64_t r0 = recs[lane];  // speculative (the bin region always exists): overlaps the counter load
  const uint32_t c = bin_cnt[bin];
  if (c == 0) return;
  if (c <= 64) {
    // The common case (~32 records per bin): sort the records by (slot, idx) in registers.  Slots commute, so any
    // order that keeps each slot's requests in idx order is serial-equivalent; after the sort they sit in adjacent
    // lanes.  Every slot's word is fetched once by its first lane, all slots are resolved at once, and each
    // changed word is written back once.  No LDS.
    if (lane == 0) bin_cnt[bin] = 0;
    uint64_t w = ~0ull;
    if (lane < c) w = ((uint64_t)rec_gk(r0) << 32) | ((uint64_t)rec_idx(r0) << 16) | rec_op(r0);
    w = wave_sort_u64(w);
    const bool valid = lane < c;
    const uint32_t slot = (uint32_t)(w >> 32), idx = (uint32_t)(w >> 16) & 0xFFFF, op = (uint32_t)w & 0xFF;
    const uint32_t up = __shfl_up(slot, 1, 64);
    const bool head = valid && (lane == 0 || up != slot);
    const uint64_t hm = __ballot(head), vm = __ballot(valid);
    const uint64_t lt = lanemask_lt(), le = lt | (1ull << lane);
    const int hl = valid ? 63 - __clzll(hm & le) : (int)lane;
    const uint64_t above = hm & ~le;
    const uint64_t next = above ? (above & (~above + 1ull)) : vm + 1ull;
    const uint64_t seg = valid ? ((next - 1ull) & ~((1ull << hl) - 1ull)) : 0;
    uint2 st0 = make_uint2(0, 0);
    if (head) st0 = table[slot];
    st0.x = __shfl(st0.x, hl, 64);
    st0.y = __shfl(st0.y, hl, 64);
    uint32_t code = 0, rv = 0;
    uint2 fin = st0;
    bool dirty = false;
    Ops::resolve_sorted(valid, seg, op, st0, code, rv, fin, dirty);
    if (valid) Ops::write_reply(rep, V, idx, op, code, rv);
    if (head && dirty) table[slot] = fin;
    return;
  }
  rank_build(R, recs, c, n);

  for (uint32_t lo = 0; lo < c; lo += DINT_WCAP) {
    const uint32_t wn = min(DINT_WCAP, c - lo);
    for (uint32_t h = lane; h < DINT_HSIZE; h += 64) { Hk[h] = DINT_EMPTY; Hfl[h] = 0; }
    __syncthreads();
    // gather this window's records in request order and group them by slot
    for (uint32_t k = lane; k < c; k += 64) {
      const uint64_t r = recs[k];
      const uint32_t rk = rank_of(R, rec_idx(r), n) - lo;
      if (rk < wn) {
        bool nw;
        const uint32_t e = lds_hash_insert(Hk, rec_gk(r), &nw);
        Srec[rk] = rec_idx(r) | (e << 16) | (rec_op(r) << 26);
      }
    }
    __syncthreads();
    // one HBM read per distinct slot, all in flight together
    uint2 v[DINT_HSIZE / 64];
#pragma unroll
    for (uint32_t j = 0; j < DINT_HSIZE / 64; j++) {
      const uint32_t k = Hk[lane + 64 * j];
      v[j] = make_uint2(0, 0);
      if (k != DINT_EMPTY) v[j] = table[k];
    }
#pragma unroll
    for (uint32_t j = 0; j < DINT_HSIZE / 64; j++) Hst[lane + 64 * j] = v[j];
    __syncthreads();

    for (uint32_t ch = 0; ch < wn; ch += 64) {
      const uint32_t j = ch + lane;
      const bool valid = j < wn;
      const uint32_t sr = valid ? Srec[j] : 0;
      const uint32_t idx = sr & 0xFFFF, e = (sr >> 16) & (DINT_HSIZE - 1), op = sr >> 26;
      if (valid) atomicAdd(&Hfl[e], 1u);
      __syncthreads();
      const uint32_t cnt = valid ? (Hfl[e] & 0xFFFF) : 0;
      uint32_t code = 0, rv = 0;
      if (valid && cnt == 1) {  // the only request of this chunk on its slot
        uint2 st = Hst[e];
        bool wr = false;
        code = Ops::apply(op, st, rv, wr);
        if (wr) { Hst[e] = st; atomicOr(&Hfl[e], 0x80000000u); }
      }
      uint64_t conf = __ballot(valid && cnt > 1);
      while (conf) {  // one iteration per slot shared by several lanes of the chunk
        const int leader = __ffsll((unsigned long long)conf) - 1;
        const uint32_t se = __builtin_amdgcn_readlane(e, leader);
        const bool mine = valid && e == se;
        const uint64_t same = __ballot(mine);
        Ops::resolve_group(same, mine, op, Hst, Hfl, se, (int)lane == leader, code, rv);
        conf &= ~same;
      }
      __syncthreads();
      if (valid) {
        atomicAnd(&Hfl[e], 0x80000000u);
        Ops::write_reply(rep, V, idx, op, code, rv);
      }
      __syncthreads();
    }
    // write back the words that changed
#pragma unroll
    for (uint32_t j = 0; j < DINT_HSIZE / 64; j++) {
      const uint32_t h = lane + 64 * j;
      const uint32_t k = Hk[h];
      if (k != DINT_EMPTY && (Hfl[h] >> 31)) table[k] = Hst[h];
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");  // next window may re-read these words
    __syncthreads();
  }
  if (lane == 0) bin_cnt[bin] = 0;  // leave the counters clean for the next pass
}

// ------------------------------------------------------------------------------------------
template <int WL, class Ops>
static void launch_locks(const void *d_req, void *d_rep, uint32_t n, uint2 *table, dint_mod slots, dint_shard shard,
                         dint_scratch s, hipStream_t st, hipEvent_t *ev, const dint_view &view) {
  if (n == 0) return;
  const uint32_t P = dint_pick_bins(n);
  if (ev) hipEventRecord(ev[0], st);
  hipLaunchKernelGGL((k_lock_scatter<WL>), dim3((n + 255) / 256), dim3(256), 0, st, (const uint8_t *)d_req,
                     (uint8_t *)d_rep, n, slots, shard, P - 1, s.bin_cnt, s.bins, s.stats, view);
  if (ev) hipEventRecord(ev[1], st);
  hipLaunchKernelGGL((k_lock_resolve<Ops>), dim3(P), dim3(64), 0, st, (uint8_t *)d_rep, n, table, s.bin_cnt,
                     (const uint64_t *)s.bins, view);
  if (ev) hipEventRecord(ev[2], st);
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
