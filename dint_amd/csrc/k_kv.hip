// k_kv.hip -- store / tatp / smallbank shard servers on gfx950.
//
// Reference semantics (serial, one message at a time):
//   store/udp/server.cc:75-97            READ / SET on one kvs  (INSERT as store/ebpf/store_kern.c:226-297)
//   tatp/udp/server_shard.cc:113-210     13 request types over 5 kvs tables + txn_locks[5][] + log ring
//   smallbank/udp/server_shard.cc:107-189  9 request types over 2 tables + num_ex/num_sh counters + log ring
// Every request touches exactly one bucket of one table (lock_hash % hash_size == kvs bucket), or only
// the log ring.  Requests on different buckets commute; requests on one bucket apply in request order.
//
// One pass (n <= 2^20 requests) = four kernels (k_kv_part, k_kv_resolve, k_kv_hot, k_kv_big), a TWO-LEVEL partition; the passes
// of several engines share one set of launches (grid.y / grid ranges = engine, dint_launch_kv_multi):
//   k_kv_part    : KV_TB * RPT requests per workgroup -- copy the messages to the reply array, classify, hash, count the
//                  records per COARSE bin (coarse = group % C, C ~ n / 512, any number: kv_cut) in LDS, reserve each run
//                  with one device atomic per (workgroup, coarse bin) and store the 16-byte records {key, group / C |
//                  idx | type | lock quadrant | 9 key-hash bits}; a coarse bin holds `cap` records in place (64 x the mean
//                  load: a hot key's bin fits; what does not goes to the pass's overflow list).  Log requests are finished
//                  here: the canonical 64-byte record goes to ring position tail + (#log requests below i)
//                  [deterministic: an exclusive scan, not an atomic].
//   k_kv_resolve : one 512-thread workgroup per coarse bin.  It splits the bin's records by sub = (group / C) % 64 in
//                  LDS and packs neighbouring subs into chunks of <= 64 records; every chunk is one wave: sorted by
//                  (bucket group, key hash, idx) in registers and resolved at once (kv_chunk).  A sub of more than 64
//                  records (a hot key) is listed as work items: its records move to the big path's 8-byte form.
//   k_kv_hot     : store / tatp: the items that have a closed form -- a hot key alone in its sub (a SOLO item), or cut by
//                  request-index range into PIECES that several workgroups answer at once, publishing to each other what
//                  a piece needs of its predecessors, plus the sub's REMAINDER (kv_solo_item, kv_hot_item, kv_rem_chunks,
//                  kv_group_phases).  128 VGPRs, 19 KB of LDS: it runs beside anything.  What it cannot answer it lists
//                  for k_kv_big.
//   k_kv_big     : one 512-thread workgroup per listed sub (smallbank: every big sub): sorted in LDS -- or ordered by index
//                  bitmaps --, ballot masks over the whole sorted stretch with O(1) range tables, then leaders /
//                  512-request tiles / write-back (kv_big_bin).
//                  Inside a chunk or stretch, several requests on ONE key are resolved in closed form
//                  (version = v0 + #writers below, value = message of the last writer below, lock = last lock op
//                  below; smallbank's counters: a walk per mode change, and whole chunks that cannot change the mode
//                  skipped at once); what the closed forms do not cover runs in rounds (k-th request of a bucket run
//                  in round k, workgroup fence between rounds), so every request sees the table exactly as the serial
//                  reference would.  The table is the HBM layout of dint_kv_core.h: one 64-byte header sector answers
//                  the probe.
// (r01-r03: one level, bin = group % (n / 32), 8-byte records, three launches: one returning device atomic and one
// partial-sector scatter per request in the count kernel, a key gather per request and half-empty waves in the resolve
// kernel.  NOTEBOOK.md section 1.)
#include "k_kv_dev.h"

dint_kv_fmt dint_kv_format(uint32_t workload) {
  switch (workload) {
    case DINT_WL_STORE: return {53, 0, 0xFFFFFFFFu, 1, 9, 49, 40};
    case DINT_WL_TATP: return {55, 1, 2, 3, 11, 51, 40};
    default: return {23, 1, 2, 3, 11, 19, 8};
  }
}


// ---- launch -------------------------------------------------------------------------------------------
// coarse bins of a pass: ~`load` records each (512: one chunk per wave of the resolve workgroup), any number
static inline uint32_t kv_pick_coarse(uint32_t n, uint32_t load) {
  uint64_t c = ((uint64_t)n + load - 1) / load;
  if (c < 1) c = 1;
  if (c > DINT_KV_CMAX) c = DINT_KV_CMAX;
  return (uint32_t)c;
}
static uint32_t kv_env(const char *name, uint32_t dflt) {
  const char *v = getenv(name);
  return v && *v ? (uint32_t)strtoul(v, nullptr, 10) : dflt;
}
// the knobs of an engine's kv passes: read once, by dint_kv_create
static dint_kv_knobs kv_read_knobs() {
  dint_kv_knobs k;
  k.coarse_load = std::min(8192u, std::max(64u, kv_env("DINT_KV_COARSE_LOAD", KVB_T)));
  k.cap = kv_env("DINT_KV_CAP", 0);
  k.lcap = std::min(KVR_LCAP, kv_env("DINT_KV_LCAP", KVR_LCAP));
  k.cap_mult = dint_kv_cap_mult();
  k.rpt = kv_env("DINT_KV_RPT", 0);
  k.no_bm = kv_env("DINT_KV_NO_BM", 0);
  k.hot_min = kv_env("DINT_KV_HOT_MIN", 0);
  k.no_split = kv_env("DINT_KV_NO_SPLIT", 0);
  k.split_min = std::max(65u, kv_env("DINT_KV_SPLIT_MIN", 65u));
  k.split_target = std::min(448u, std::max(8u, kv_env("DINT_KV_SPLIT_TARGET", 384u)));
  k.one_big_kernel = kv_env("DINT_KV_ONE_BIG_KERNEL", 0);
  k.no_ahead = kv_env("DINT_KV_NO_AHEAD", 0);
  k.no_fuse = kv_env("DINT_KV_NO_FUSE", 0);
  k.sb_split_min = kv_env("DINT_KV_SB_SPLIT_MIN", 65u);
  k.sb_late_grid = std::max(1u, std::min(KVB_GRID, kv_env("DINT_KV_SB_LATE_GRID", 32u)));
  k.sb_npmax = std::max(2u, std::min(KSB_NPMAX, kv_env("DINT_KV_SB_NPMAX", 128u)));
  k.sb_workers = kv_env("DINT_KV_SB_WORKERS", k.sb_npmax + 2u);
  if (k.sb_workers) k.sb_workers = std::max(k.sb_workers, k.sb_npmax + 2u);  // (a row's pieces wait for each other, its remainder for them)
  // (never fewer workers than a hot key has pieces + a remainder, + 1: the pieces wait for each other.  An idle worker holds half
  // a compute unit that another shard server's resolve workgroup is waiting for: 320 / 192 / 96 / 48 workers per engine gave
  // 2,320 / 2,650 .. 2,840 / 2,980 .. 3,030 / 2,820 Mtxn/s on the tatp bench stream -- ~160 items per pass)
  k.workers = std::min(1024u, std::max(KVR_NPMAX + 2u, kv_env("DINT_KV_WORKERS", KVW_GRID)));  // (a hot key's pieces + remainder wait for each other: never fewer workers)
  k.part_first = kv_env("DINT_KV_PART_FIRST", 0) & 1u;
  k.late_grid = std::min(KVB_GRID, std::max(1u, kv_env("DINT_KV_LATE_GRID", 8u)));
  k.late_big = kv_env("DINT_KV_LATE_BIG", 0);
  k.exp_no_late = kv_env("DINT_EXP_NO_LATE", 0);
  k.late_fat = kv_env("DINT_KV_LATE_FAT", 0);
  return k;
}

// hot keys in pieces / closed forms on their own light kernel (k_kv_hot): store / tatp; never with the closed forms switched
// off (DINT_FLAG_KV_ROUNDS, DINT_FLAG_LOCK_SAME_KEY, DINT_FLAG_KV_NO_HOT) or for the internal LOAD requests
static inline bool kv_uses_hot(const dint_kv &kv, int load_mode) {
  return kv.workload != DINT_WL_SMALLBANK && !(kv.force_rounds & 3) && !load_mode && !kv.knobs.no_split;
}
// a pass of this engine is k_kv_pass (+ k_kv_late): the engine's kernel timer then has three intervals (part, pass, late)
bool dint_kv_one_launch(const dint_kv &kv, int load_mode) {
  return kv_uses_hot(kv, load_mode) && !kv.knobs.one_big_kernel && !kv.knobs.no_fuse;
}
bool dint_kv_ahead_ok(const dint_kv &kv, int load_mode) {
  if (kv.workload == DINT_WL_SMALLBANK) return !load_mode && !kv.knobs.no_ahead && !kv.knobs.no_fuse;  // (the partition beside the resolve stage)
  return kv_uses_hot(kv, load_mode) && !kv.knobs.one_big_kernel && !kv.knobs.no_ahead;
}

static void kv_fill_pass(kv_pass_args &A, const void *d_req, void *d_rep, uint32_t n, const dint_kv &kv, dint_log log,
                         const dint_scratch &s, int load_mode, const dint_view &view, uint32_t tile) {
  // tuning / test knobs (dint_kv_knobs, read when the engine was created): DINT_KV_COARSE_LOAD = records per coarse bin,
  // DINT_KV_CAP = records a coarse bin holds in place (small values exercise the overflow list), DINT_KV_LCAP = records of a
  // bin's small subs resolved from LDS (small values exercise the all-big fallback)
  const dint_kv_knobs &K = kv.knobs;
  const uint32_t C = kv_pick_coarse(n, K.coarse_load);
  const uint32_t mean = (n + C - 1) / C;
  uint32_t cap = K.cap_mult * mean + 64;  // (engine.hip sizes kbins for it)
  cap = std::min(cap, (uint32_t)(s.kbins_slots / C));
  cap = std::max(1u, K.cap ? std::min(cap, K.cap) : cap);
  A.req = (const uint8_t *)d_req; A.rep = (uint8_t *)d_rep; A.n = n;
  A.n_tiles = (n + tile - 1) / tile;
  A.kv = kv.d_dev; A.log = log; A.cut = kv_make_cut(C, n); A.cap = cap;
  A.lcap = K.lcap;
  // the sets of this pass (dint_kv_sets): what its partition writes by pass number & 1, its control words by pass number % 3;
  // its resolve stage zeroes the control words of the pass after next
  const uint64_t pn = s.kvs.pass_no;
  A.pno = (uint32_t)(pn & 1u);
  A.bin_cnt = s.kvs.bin_cnt[pn & 1]; A.kbins = s.kvs.kbins[pn & 1]; A.ovl = s.kvs.ovl[pn & 1];
  A.big = s.kvs.ctl[pn % 3]; A.big_z = s.kvs.ctl[(pn + 2) % 3];
  A.blk_pub = s.kvs.pub[pn % 3]; A.blk_pub_z = s.kvs.pub[(pn + 2) % 3];
  A.bigrdy = s.kvs.bigrdy;
  A.ovf = s.ovf; A.ovf2 = s.ovf2; A.stats = s.stats;
  A.bigq = s.bigq;
  A.hotpub = s.hotpub;
  A.lateq = s.lateq;
  A.seq = s.pass_seq ? s.pass_seq : 1u;
  A.inv_n = n > 1 ? (uint32_t)((1ull << 32) / n) : 0xFFFFFFFFu;
  A.load_mode = load_mode;
  A.force_flags = kv.force_rounds | (K.no_bm ? 4 : 0) | (int)(K.hot_min << 8);
  // hot keys in pieces (kv_hot_item): DINT_KV_SPLIT_MIN / _TARGET: the smallest sub that takes the path (default: every big
  // sub) / requests per piece (tests run small values); DINT_KV_NO_SPLIT=1: r04's kv_big_bin for every big sub, one
  // workgroup per hot key
  A.split_min = kv_uses_hot(kv, load_mode) ? K.split_min : 0xFFFFFFFFu;
  A.split_target = K.split_target;
  A.np_max = KVR_NPMAX;
  A.sb_pieces = 0;
  A.sbx = s.kvs.sbx;
  // smallbank (r06): a hot account's row of at least DINT_KV_SB_SPLIT_MIN requests in pieces, several workgroups of k_kv_big at
  // once (kv_sb_item); never with the closed forms switched off or for LOAD requests
  if (kv.workload == DINT_WL_SMALLBANK && K.sb_split_min && s.kvs.sbx && !(kv.force_rounds & 3) && !load_mode && !K.no_split) {
    A.split_min = std::max(8u, K.sb_split_min);
    A.np_max = K.sb_npmax;
    A.sb_pieces = 1;
  }
  A.has_log = kv.workload != DINT_WL_STORE;
  A.trace = kv.d_trace;
  A.V = view;
}

// (launch_kv_passes<WL>: k_kv_dev.h -- instantiated by k_kv_store.hip / k_kv_tatp.hip / k_kv_smallbank.hip, one workload each,
// so that the three compile side by side)
extern template int kv_piece_residency<DINT_WL_STORE>(int);
extern template int kv_piece_residency<DINT_WL_TATP>(int);
extern template int kv_piece_residency<DINT_WL_SMALLBANK>(int);
extern template void launch_kv_passes<DINT_WL_STORE>(kv_multi_args &, uint32_t, uint32_t, hipStream_t, hipEvent_t *, const dint_kv_knobs &, bool, const kv_multi_args *, uint32_t);
extern template void launch_kv_passes<DINT_WL_TATP>(kv_multi_args &, uint32_t, uint32_t, hipStream_t, hipEvent_t *, const dint_kv_knobs &, bool, const kv_multi_args *, uint32_t);
extern template void launch_kv_passes<DINT_WL_SMALLBANK>(kv_multi_args &, uint32_t, uint32_t, hipStream_t, hipEvent_t *, const dint_kv_knobs &, bool, const kv_multi_args *, uint32_t);

// requests per thread of k_kv_part: longer tiles = fewer, longer runs per (tile, coarse bin), but a pass must still
// fill the GPU's 256 CUs
static inline uint32_t kv_pick_rpt(uint32_t n_total, const dint_kv_knobs &K) {
  if (K.rpt == 1 || K.rpt == 2 || K.rpt == 4) return K.rpt;
  return n_total >= 1024u * 1024u ? 4u : n_total >= 256u * 1024u ? 2u : 1u;
}
// ... of the tiles of k_kv_hot_part (KVB_T threads): tiles of 1,024 requests (at most 1,024 tiles per pass: blk_pub) or 2,048
static inline uint32_t kv_pick_rpt_ahead(uint32_t n, const dint_kv_knobs &K) {
  if (K.rpt == 4 || (K.rpt == 0 && n >= 256u * 1024u)) return 4u;
  return 2u;
}

static void launch_kv_dispatch(uint32_t workload, kv_multi_args &M, uint32_t n_eng, uint32_t rpt, hipStream_t st, hipEvent_t *ev,
                               const dint_kv_knobs &K, bool part_done = false, const kv_multi_args *next = nullptr, uint32_t next_rpt = 0) {
  switch (workload) {
    case DINT_WL_STORE: launch_kv_passes<DINT_WL_STORE>(M, n_eng, rpt, st, ev, K, part_done, next, next_rpt); break;
    case DINT_WL_TATP: launch_kv_passes<DINT_WL_TATP>(M, n_eng, rpt, st, ev, K, part_done, next, next_rpt); break;
    default: launch_kv_passes<DINT_WL_SMALLBANK>(M, n_eng, rpt, st, ev, K, part_done, next, next_rpt); break;
  }
}

void dint_launch_kv(const void *d_req, void *d_rep, uint32_t n, const dint_kv &kv, dint_log log, dint_scratch s,
                    int load_mode, hipStream_t st, hipEvent_t *ev, const dint_view &view, bool part_done, const dint_kv_ahead *next) {
  if (n == 0) return;
  kv_multi_args M;
  memset(&M, 0, sizeof M);
  // (a pass whose partition ran ahead was cut into k_kv_hot_part's tiles: n_tiles must say so -- the last tile writes the log tail)
  const uint32_t rpt = kv_pick_rpt(n, kv.knobs), rpt_a = kv_pick_rpt_ahead(n, kv.knobs);
  kv_fill_pass(M.e[0], d_req, d_rep, n, kv, log, s, load_mode, view, part_done ? KVB_T * rpt_a : KV_TB * rpt);
  kv_multi_args N;
  uint32_t next_rpt = 0;
  if (next && next->n && dint_kv_ahead_ok(kv, load_mode)) {
    // the next pass's view of the scratch: its own sets (by pass number), the next tag
    dint_scratch sn = s;
    sn.kvs.pass_no = s.kvs.pass_no + 1;
    sn.pass_seq = s.pass_seq + 1 >= 0x3FFFFFFFu ? 1u : s.pass_seq + 1;
    next_rpt = kv_pick_rpt_ahead(next->n, kv.knobs);
    memset(&N, 0, sizeof N);
    kv_fill_pass(N.e[0], next->d_req, next->d_rep, next->n, kv, log, sn, load_mode, next->view, KVB_T * next_rpt);
  } else {
    next = nullptr;
  }
  launch_kv_dispatch(kv.workload, M, 1, rpt, st, ev, kv.knobs, part_done, next ? &N : nullptr, next_rpt);
}

// one pass of each of n_eng (<= DINT_KV_MULTI_MAX) engines of ONE kv workload, all on stream st: a closed-loop epoch
// (and a step of the multi-GPU exchange) hands every shard server of the GPU one batch at the same moment; with the
// engines' kernels side by side in one grid the epoch is one stream, no fork and join over engine streams.  The engines
// stay independent (own tables, own scratch, own log).
// r06: `next` (or nullptr) = every engine's NEXT pass, announced: their partitions ride in this launch set's k_kv_pass, and that
// next call comes with part_done = true.
void dint_launch_kv_multi(const dint_kv_pass *p, uint32_t n_eng, hipStream_t st, bool part_done, const dint_kv_pass *next) {
  if (n_eng == 0) return;
  kv_multi_args M, N;
  memset(&M, 0, sizeof M);
  uint32_t n_total = 0;
  for (uint32_t k = 0; k < n_eng; k++) n_total += p[k].n;
  const uint32_t rpt = kv_pick_rpt(n_total, p[0].kv->knobs);
  for (uint32_t k = 0; k < n_eng; k++) kv_fill_pass(M.e[k], p[k].d_req, p[k].d_rep, p[k].n, *p[k].kv, p[k].log, p[k].s, 0, p[k].view, KV_TB * rpt);
  uint32_t next_rpt = 0;
  if (next) {
    uint32_t nmax = 0;
    for (uint32_t k = 0; k < n_eng; k++) nmax = std::max(nmax, next[k].n);
    next_rpt = kv_pick_rpt_ahead(nmax, p[0].kv->knobs);
    memset(&N, 0, sizeof N);
    for (uint32_t k = 0; k < n_eng; k++) {
      dint_scratch sn = next[k].s;  // (the caller's copy of the engine's scratch: the next pass's number and tag)
      sn.kvs.pass_no = p[k].s.kvs.pass_no + 1;
      sn.pass_seq = p[k].s.pass_seq + 1 >= 0x3FFFFFFFu ? 1u : p[k].s.pass_seq + 1;
      kv_fill_pass(N.e[k], next[k].d_req, next[k].d_rep, next[k].n, *next[k].kv, next[k].log, sn, 0, next[k].view, KVB_T * next_rpt);
    }
  }
  launch_kv_dispatch(p[0].kv->workload, M, n_eng, rpt, st, nullptr, p[0].kv->knobs, part_done, next ? &N : nullptr, next_rpt);
}
// can the engines of a launch set take an announcement?  (all of one workload: the first decides)
bool dint_kv_multi_ahead_ok(const dint_kv &kv) { return dint_kv_ahead_ok(kv, 0) && !kv.knobs.no_fuse; }

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
int dint_kv_create(dint_kv *kv, uint32_t workload, uint64_t n_rows, dint_shard shard, uint32_t pool_entries,
                   uint32_t flags) {
  *kv = dint_kv();
  kv->workload = workload;
  kv->h.same_key = (workload == DINT_WL_TATP && (flags & DINT_FLAG_LOCK_SAME_KEY)) ? 1 : 0;
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
    // expected overflow at the reference's 2.67 rows/bucket: 0.14 entries per bucket
    const uint64_t pool = pool_entries ? pool_entries : tb.n_local / 4 + 4096;
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
    tb.pend_head = (unsigned long long *)(ctl + 64 + 8 * KV_NLISTS);  // two sets of KV_NLISTS (kv_dev_pend_set)
    kv->h.mod[t] = dint_make_mod(hs[t]);
    kv->h.lockmod[t] = dint_make_mod(4 * hs[t]);
    kv->h.gk_base[t] = (uint32_t)gk;
    gk += tb.n_local;
  }
  kv->knobs = kv_read_knobs();
  {
    // the pieces of a hot key wait for each other: only where all of a key's siblings (and as many again) fit the device at once
    int dev = 0;
    (void)hipGetDevice(&dev);
    const int res = workload == DINT_WL_STORE ? kv_piece_residency<DINT_WL_STORE>(dev)
                    : workload == DINT_WL_TATP ? kv_piece_residency<DINT_WL_TATP>(dev) : kv_piece_residency<DINT_WL_SMALLBANK>(dev);
    kv->knobs.residency = (uint32_t)std::max(res, 0);
    if (workload == DINT_WL_SMALLBANK) {
      if (res < (int)(KSB_NPMAX + 2u + 32u)) kv->knobs.sb_split_min = 0;  // (k_kv_big: one workgroup per compute unit, 256 on an MI355X)
    } else {
      if (res < 2 * (int)(KVR_NPMAX + 2u)) kv->knobs.no_split = 1;
      kv->knobs.workers = std::max(KVR_NPMAX + 2u, std::min(kv->knobs.workers, (uint32_t)std::max(res, 0) / 2u));
    }
  }
  if (getenv("DINT_KV_TRACE")) {
    if (hipMalloc((void **)&kv->d_trace, (size_t)DINT_KV_TRACE_WORDS * 8) != hipSuccess) return DINT_ENOMEM;
    hipMemset(kv->d_trace, 0, (size_t)DINT_KV_TRACE_WORDS * 8);
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
