// dint_kv.h -- HBM key-value tables of the store / tatp / smallbank workloads (k_kv.hip).
#pragma once
#include <cstdlib>
#include <utility>
#include <vector>

#include "dint_kernels.h"
#include "dint_kv_core.h"

#define DINT_KV_MAX_TABLES 5
#define DINT_KV_CTL_BYTES (64 + 24 * KV_NLISTS)  // per table: pool_top, free_head[], pend_head[2][] (two sets: dint_kv_core.h, kv_pool_rotate)
#define DINT_KV_TRACE_WORDS ((size_t)DINT_KV_PMAX * 16 + 16 * 8192)  // per bin 16 words, then per workgroup 16
#define DINT_KV_LOAD_OP 0xF0u  // internal request type: insert a row with the version carried in msg.ver

// everything the kernels need about the tables; lives in device memory (d_dev) and in a host mirror
struct kv_dev {
  kv_tab tab[DINT_KV_MAX_TABLES];
  dint_mod mod[DINT_KV_MAX_TABLES];      // % hash_size (global bucket)
  dint_mod lockmod[DINT_KV_MAX_TABLES];  // % (4 * hash_size)  (tatp/udp/tatp.h:12-14)
  uint32_t gk_base[DINT_KV_MAX_TABLES];  // group key of local bucket 0 of each table
  uint32_t n_tables;
  uint32_t shard_index, shard_count;
  uint32_t same_key;  // tatp, DINT_FLAG_LOCK_SAME_KEY: lock slots remember their owner's key (tatp/ebpf/lock_kern.c)
};

// Tuning / test knobs of the kv passes, read from the environment ONCE, when the engine is created (r01-r05 read them with
// getenv at every launch -- ten per pass; VERDICT / ADVICE r05).  Tests set the variable before they create their engine.
struct dint_kv_knobs {
  uint32_t coarse_load = 512;   // DINT_KV_COARSE_LOAD: records per coarse bin
  uint32_t cap = 0;             // DINT_KV_CAP: records a coarse bin holds in place (0 = cap_mult x the mean load + 64)
  uint32_t lcap = 1024;         // DINT_KV_LCAP: records of a bin's small subs resolved from LDS
  uint32_t cap_mult = 64;       // DINT_KV_CAP_MULT
  uint32_t rpt = 0;             // DINT_KV_RPT: requests per thread of k_kv_part (0 = by pass size)
  uint32_t no_bm = 0;           // DINT_KV_NO_BM: kv_big_bin sorts where it would rank by index bitmap
  uint32_t hot_min = 0;         // DINT_KV_HOT_MIN
  uint32_t no_split = 0;        // DINT_KV_NO_SPLIT: r04's kv_big_bin for every big sub
  uint32_t split_min = 65;      // DINT_KV_SPLIT_MIN
  uint32_t split_target = 384;  // DINT_KV_SPLIT_TARGET
  uint32_t one_big_kernel = 0;  // DINT_KV_ONE_BIG_KERNEL: no k_kv_hot launch
  uint32_t workers = 96;        // DINT_KV_WORKERS: hot-key workers per engine of k_kv_pass
  uint32_t part_first = 0;      // DINT_KV_PART_FIRST: k_kv_pass places the next partition's tiles before (1) / behind (0) the workers
  uint32_t sb_split_min = 65;   // DINT_KV_SB_SPLIT_MIN: smallbank, the smallest big sub whose row is answered in pieces (65: every big sub; 0: never -- r05's kv_big_bin)
  uint32_t sb_npmax = 128;      // DINT_KV_SB_NPMAX: smallbank, pieces of one row at most (<= KSB_NPMAX; x 384 requests: 49,000 -- a bigger row goes the old way)
  uint32_t sb_workers = 130;    // DINT_KV_SB_WORKERS: smallbank, workers per engine of k_kv_pass (>= KSB_NPMAX + 2: a row's pieces wait for each other; 0: every item in k_kv_big)
  uint32_t sb_late_grid = 32;   // DINT_KV_SB_LATE_GRID: smallbank, workgroups per engine of the k_kv_big launch behind the workers
  uint32_t residency = 0;       // workgroups of the pieces' kernel the device holds at once (kv_piece_residency)
  uint32_t no_fuse = 0;         // DINT_KV_NO_FUSE: k_kv_resolve and k_kv_hot / k_kv_hot_part as launches of their own, not k_kv_pass
  uint32_t no_ahead = 0;        // DINT_KV_NO_AHEAD: never run a pass's partition beside the previous pass's hot keys
  uint32_t late_grid = 8;       // DINT_KV_LATE_GRID: workgroups of the k_kv_big launch behind k_kv_hot (DINT_KV_LATE_BIG)
  uint32_t exp_no_late = 0;     // DINT_EXP_NO_LATE: experiments only -- no launch behind k_kv_hot at all (late items stay unanswered)
  uint32_t late_fat = 0;        // DINT_KV_LATE_FAT: k_kv_late at 256 VGPRs (no spills, but its workgroups want an empty compute unit)
  uint32_t late_big = 0;        // DINT_KV_LATE_BIG: what k_kv_hot leaves goes to k_kv_big (r05) instead of k_kv_late
};

struct dint_kv {
  uint32_t workload = 0;
  uint32_t n_tables = 0;
  uint32_t val_size = 0;
  int force_rounds = 0;  // DINT_FLAG_KV_ROUNDS: never use the closed-form same-key path (A/B and parity testing)
  uint64_t hash_size[DINT_KV_MAX_TABLES] = {0, 0, 0, 0, 0};  // global bucket counts
  kv_dev h{};                // host mirror (device pointers inside)
  kv_dev *d_dev = nullptr;   // device copy
  uint8_t *d_ctl = nullptr;  // pool_top / free_head / pend_head words of all tables
  uint64_t *d_trace = nullptr;  // DINT_KV_TRACE=1: [DINT_PMAX][16] per-wave s_memtime stamps of the last resolve launch
  size_t entry_bytes[DINT_KV_MAX_TABLES] = {0, 0, 0, 0, 0};
  dint_kv_knobs knobs;
};

// records a coarse bin of a kv pass holds IN PLACE, in units of the mean load of a bin (engine.hip sizes the scratch for it,
// k_kv.hip's kv_fill_pass sets the pass's `cap`); DINT_KV_CAP_MULT overrides (2 = r04 / r05: a hot key's bin overflows)
static inline uint32_t dint_kv_cap_mult() {
  const char *v = getenv("DINT_KV_CAP_MULT");
  const unsigned long m = v && *v ? strtoul(v, nullptr, 10) : 64ul;
  return (uint32_t)(m < 2 ? 2 : m > 256 ? 256 : m);
}
int dint_kv_create(dint_kv *kv, uint32_t workload, uint64_t n_rows, dint_shard shard, uint32_t pool_entries = 0,
                   uint32_t flags = 0);
void dint_kv_destroy(dint_kv *kv);
std::vector<std::pair<void *, size_t>> dint_kv_regions(dint_kv *kv);
// valid rows of `table` in bucket order, chain order inside a bucket; returns the row count
int64_t dint_kv_dump_rows(dint_kv *kv, uint32_t table, uint64_t *keys, uint32_t *vers, uint8_t *vals, uint64_t cap);
// lock words, index = q * n_local + local bucket  (== lock_hash when unsharded)
int64_t dint_kv_read_locks(dint_kv *kv, uint32_t table, uint32_t *a, uint32_t *b, uint64_t cap);
// one pass (n <= DINT_MICRO and, when a log is attached, n <= log.cap).  load_mode: accept DINT_KV_LOAD_OP
// rows and ignore rows of other shards silently.
// `view`: where request i lives (contiguous array, or the segments of a multi-GPU exchange buffer)
// Look-ahead (r06; store / tatp): `next` = the batch of the engine's NEXT pass, complete in device memory in the order of
// `st`: its partition stage (k_kv_part: no table access) then runs in ONE launch with this pass's hot keys (k_kv_hot_part),
// and that next pass is launched with part_done = true.  dint_kv_ahead_ok says whether a pass of this engine can take one.
struct dint_kv_ahead {
  const void *d_req;
  void *d_rep;
  uint32_t n;
  dint_view view;
};
bool dint_kv_ahead_ok(const dint_kv &kv, int load_mode);
bool dint_kv_one_launch(const dint_kv &kv, int load_mode);
void dint_launch_kv(const void *d_req, void *d_rep, uint32_t n, const dint_kv &kv, dint_log log, dint_scratch s,
                    int load_mode, hipStream_t st, hipEvent_t *ev, const dint_view &view = dint_flat_view(),
                    bool part_done = false, const dint_kv_ahead *next = nullptr);
// the passes of several engines of one kv workload in one launch set (grid.y = engine), all on one stream: what a
// closed-loop epoch or an exchange step hands the GPU's shard servers at the same moment (n > 0 for every engine)
#define DINT_KV_MULTI_MAX 4u
struct dint_kv_pass {
  const void *d_req;
  void *d_rep;
  uint32_t n;
  const dint_kv *kv;
  dint_log log;
  dint_scratch s;
  dint_view view;
};
void dint_launch_kv_multi(const dint_kv_pass *p, uint32_t n_eng, hipStream_t st, bool part_done = false, const dint_kv_pass *next = nullptr);
bool dint_kv_multi_ahead_ok(const dint_kv &kv);
void dint_launch_home_kv(const void *d_req, uint32_t n, const dint_kv &kv, uint8_t *d_home, hipStream_t st);
// wire message size / field offsets of a kv workload
struct dint_kv_fmt {
  uint32_t msg, type, table, key, val, ver, val_size;  // table == 0xFFFFFFFF: no table field
};
dint_kv_fmt dint_kv_format(uint32_t workload);

// ---- multi-GPU routing (k_route.hip) --------------------------------------------------------------------------
#define DINT_ROUTE_MAXW 64u        // ranks a batch can be routed to
#define DINT_ROUTE_MAXN 1048576u   // requests per dint_route_pack call
#define DINT_ROUTE_BLK_WORDS (1024u * DINT_ROUTE_MAXW + 64u)  // one copy of the pack kernel's scratch
struct dint_route_scratch {
  uint32_t *blk;       // [1024 tiles][world] requests per tile and destination, then {ticket, words published}: all zero between calls
  uint32_t *blk_next;  // the copy the NEXT call uses (this call zeroes it; the engine swaps the two after every call)
};
// stable partition of n contiguous requests by home rank into `shard.count` slots of `cap` messages, slot w at
// d_send + w * stride, its live count (u32) at d_cnt + w * cnt_stride; d_slot[i] = home * cap + position (or ~0u)
#define DINT_ROUTE_MAXS 4u         // batches (logical servers of one rank) routed by one launch set
struct dint_route_job {            // one batch and the engine whose hash / modulus routes it
  uint32_t workload, msg;
  dint_mod slots;
  const dint_kv *kv;
  dint_shard shard;
  const void *d_req;
  void *d_rep;                     // unpack only
  uint32_t n, cap;
  const uint32_t *d_n;             // live request count on the device (n is then its upper bound), or nullptr
  void *d_send, *d_cnt;            // this batch's slot / live-count word of peer 0
  uint64_t cnt_stride;
  uint32_t *d_slot;
  dint_route_scratch rs;
  dint_dev_stats *stats;
};
void dint_launch_route_pack(const dint_route_job *jobs, uint32_t n_jobs, uint64_t stride, hipStream_t st);
void dint_launch_route_unpack(const dint_route_job *jobs, uint32_t n_jobs, uint64_t stride, hipStream_t st);
