// dint_kernels.h -- host-callable launchers of the HIP kernels (one .hip file per family).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dint_device.h"

// device-resident counters mirrored into dint_stats
struct dint_dev_stats {
  unsigned long long bad_requests;
  unsigned long long missing_keys;
  unsigned long long foreign_requests;
  unsigned long long pool_exhausted;
  unsigned long long route_overflow;  // requests dropped by dint_route_pack: a destination slot was full
  unsigned long long big_bin_requests;  // requests resolved by the big-bin workgroups (bins of > 64 records)
  unsigned long long late_requests;     // store / tatp: requests k_kv_hot's closed forms left to the general path (k_kv_late / k_kv_big)
  unsigned long long late_items[3];     // ... the work items they came from, by kind: a sub listed as it was / a solo item / the pieces of a hot key
};

// kv passes (r06): a pass's partition stage may run in the SAME launch as the previous pass's resolve stage (k_kv_pass), so
// what the partition writes exists twice (pass number & 1), and the control words -- zeroed one pass before they are used, by
// the resolve stage -- three times (pass number % 3).  What only the resolve / hot / late stages of one pass touch (ovf, ovf2,
// bigq, hotpub, lateq, bigrdy) exists once: those stages of two passes never overlap.
struct dint_kv_sets {
  uint32_t *ctl[3] = {nullptr, nullptr, nullptr};   // [16] {[0] records handed to the big-sub path, [1] overflow-list entries, [2] tiles handed
                                                    // out, [3] work items listed, [4] item tickets, [5] late items, [6] coarse bins that have listed}
  uint32_t *pub[3] = {nullptr, nullptr, nullptr};   // [1024] log requests per tile of the partition, bit 31 = published
  uint32_t *bin_cnt[2] = {nullptr, nullptr};        // [DINT_KV_CMAX] records per coarse bin
  uint4 *kbins[2] = {nullptr, nullptr};             // [C][cap] the coarse bins
  uint4 *ovl[2] = {nullptr, nullptr};               // the pass's overflow list
  uint32_t *bigrdy = nullptr;                       // [DINT_KV_BIGQ_MAX] work item i is listed: the pass's tag (pass_seq)
  uint64_t *sbx = nullptr;                          // smallbank: [DINT_KV_SBX_ITEMS][40] the pieces' op masks and grants (k_kv_dev.h, kv_sb_item)
  uint64_t pass_no = 0;                             // passes launched so far (host side)
};

#define DINT_KV_HOTPUB_WORDS (3u * DINT_KV_BIGQ_MAX)  // hotpub: [i] item i's word; [BIGQ_MAX + 2 i ..] the two words of a smallbank item that is a sub's only piece
#define DINT_KV_SBX_WORDS 48u  // smallbank: words per work item in dint_kv_sets::sbx (k_kv_dev.h, kv_sb_item)

// scratch shared by every workload: bins of batch records
struct dint_scratch {
  uint32_t *bin_cnt;   // [DINT_KV_PMAX]   zero between passes (the resolve kernels re-zero their own)
  uint64_t *bins;      // [DINT_KV_PMAX][DINT_KV_BINCAP]
  dint_dev_stats *stats;
  uint32_t *blk_cnt;   // per-block counts of the log scan [256]  (log_server)
  // ---- every workload but log_server (dint_bins.h) ----
  uint32_t *blk_pub;   // [1024] log requests per 1024-request slice of the pass, bit 31 = published
  uint32_t *blk_pub_next;  // ... of the next pass (two arrays, used alternately)
  uint32_t *big;       // [4 + DINT_KV_PMAX]: big[0] = number of bins with more than DINT_KV_BINCAP records in this
                       // pass, big[1] = number of overflow records, big[2] = slices handed out, then the bin ids
  uint32_t *big_next;  // the list of the next pass (two lists, used alternately)
  uint32_t *bin_off;   // [DINT_KV_PMAX] start of a big bin's records DINT_KV_BINCAP.. in `ovf`
  uint4 *ovl;          // [pass_max] overflow records as counted: {record lo, record hi, bin, position in bin}
  uint64_t *ovf;       // [pass_max] overflow records grouped by bin
  uint64_t *ovf2;      // [pass_max] kv: a big sub's records again, grouped by stretch (kv_big_bin's one-time partition)
  // ---- kv workloads (k_kv.hip): the coarse bins of the two-level partition; their overflow list is `ovl` with two
  // uint4 per entry, the 8-byte records of the big subs go to `ovf`
  uint4 *kbins = nullptr;          // [C][cap] 16-byte records {key, group / C | idx | payload}
  uint64_t kbins_slots = 0;        // records `kbins` holds: C * cap never exceeds it
  uint4 *bigq = nullptr;           // [DINT_KV_BIGQ_MAX][3] the pass's big subs and hot-key pieces (work items of k_kv_big)
  unsigned long long *hotpub = nullptr;  // [DINT_KV_HOTPUB_WORDS] what the pieces of a hot key tell each other (tagged with pass_seq)
  uint4 *lateq = nullptr;          // [DINT_KV_BIGQ_MAX] what k_kv_hot leaves to k_kv_big {bin, offset, records, 0: in ovf / 1: in ovf2}
  uint32_t pass_seq = 0;           // host side: passes launched so far (never 0 in a launch)
  dint_kv_sets kvs;                // kv workloads: the sets by pass number (the fields above are their set 0 / unused)
  uint64_t *lock_trace = nullptr;  // DINT_KV_TRACE=1 on a lock engine: per big-bin workgroup 16 s_memrealtime stamps
                                   // of its first bin, at word DINT_KV_PMAX * 16 + 16 * workgroup (dint_kv_trace_read)
};

struct dint_shard {
  uint32_t index, count;  // count >= 1
};

// lock tables: ~32 records per bin on average, so that almost every bin fits one 64-lane chunk (one wave resolves one
// bin); power of two, <= DINT_KV_PMAX
static inline uint32_t dint_pick_bins_kv(uint32_t n) {
  uint32_t p = 1;
  while (p < DINT_KV_PMAX && p * 32u < n) p <<= 1;
  return p;
}
// kv passes: any number of bins (bin = group % P), `load` records per bin on average.  32 by default: measured on the
// TATP bench stream (tools/exp_binload.sh, profiles/r03_experiments.md section 5) 26 / 34 / 40 / 46 / 52 records per bin give
// 2,002 / 2,066 / 2,043 / 1,902 / 1,730 Mtxn/s -- fuller waves do not pay (the pass is bound by its memory transactions
// per request, not by instructions per wave) and the Poisson tail above 64 records goes to the big-bin workgroups.
// What the free choice of P buys is a load that does not depend on n: a power of two left 16 records per bin for a pass
// just above 32 * 2^k requests.  DINT_KV_BIN_LOAD overrides it for tuning runs.
static inline uint32_t dint_pick_bins_load(uint32_t n, uint32_t load) {
  if (load < 8) load = 8;
  if (load > 56) load = 56;
  uint64_t p = ((uint64_t)n + load - 1) / load;
  if (p < 1) p = 1;
  if (p > DINT_KV_PMAX) p = DINT_KV_PMAX;
  return (uint32_t)p;
}

// ---- lock tables (lock_fasst, lock_2pl): k_locks.hip ----------------------------------------
// table entry: uint2 {a, b} = fasst {lock, ver} / 2pl {num_ex, num_sh}
void dint_launch_fasst(const void *d_req, void *d_rep, uint32_t n, uint2 *table, dint_mod slots,
                       dint_shard shard, dint_scratch s, hipStream_t st, hipEvent_t *ev,
                       const dint_view &view = dint_flat_view());
void dint_launch_2pl(const void *d_req, void *d_rep, uint32_t n, uint2 *table, dint_mod slots,
                     dint_shard shard, dint_scratch s, hipStream_t st, hipEvent_t *ev,
                     const dint_view &view = dint_flat_view());
void dint_launch_home_lid(const void *d_req, uint32_t msg_size, uint32_t n, dint_mod slots, uint32_t shard_count,
                          uint8_t *d_home, hipStream_t st);

// ---- log append: k_log.hip ---------------------------------------------------------------------
struct dint_log {
  uint8_t *ring;        // [cap][64] canonical records
  uint32_t *tail;       // device words {cur, next, appended lo, appended hi}: ring position before / after the pass in
                        // flight, and the number of records ever appended
  uint32_t cap;
};
void dint_launch_log(const void *d_req, void *d_rep, uint32_t n, dint_log log, dint_scratch s, hipStream_t st,
                     hipEvent_t *ev);
