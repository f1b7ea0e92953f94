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
};

// scratch shared by every workload: bins of batch records
struct dint_scratch {
  uint32_t *bin_cnt;   // [DINT_PMAX]   zero between passes (the resolve kernel re-zeroes its own)
  uint64_t *bins;      // [DINT_PMAX][DINT_MICRO]
  dint_dev_stats *stats;
  uint32_t *blk_cnt;   // [256] per-block counts of the log scan
};

struct dint_shard {
  uint32_t index, count;  // count >= 1
};

static inline uint32_t dint_pick_bins(uint32_t n) {
  // ~32 records per bin on average, so that almost every bin fits one 64-lane chunk (one wave resolves one
  // bin); power of two, <= DINT_PMAX
  uint32_t p = 1;
  while (p < DINT_PMAX && p * 32u < n) p <<= 1;
  return p;
}

// ---- lock tables (lock_fasst, lock_2pl): k_locks.hip ----------------------------------------
// table entry: uint2 {a, b} = fasst {lock, ver} / 2pl {num_ex, num_sh}
void dint_launch_fasst(const void *d_req, void *d_rep, uint32_t n, uint2 *table, dint_mod slots,
                       dint_shard shard, dint_scratch s, hipStream_t st, hipEvent_t *ev);
void dint_launch_2pl(const void *d_req, void *d_rep, uint32_t n, uint2 *table, dint_mod slots,
                     dint_shard shard, dint_scratch s, hipStream_t st, hipEvent_t *ev);
void dint_launch_home_lid(const void *d_req, uint32_t msg_size, uint32_t n, dint_mod slots, uint32_t shard_count,
                          uint8_t *d_home, hipStream_t st);

// ---- log append: k_log.hip ---------------------------------------------------------------------
struct dint_log {
  uint8_t *ring;        // [cap][64] canonical records
  uint32_t *tail;       // device word
  uint32_t cap;
};
void dint_launch_log(const void *d_req, void *d_rep, uint32_t n, dint_log log, dint_scratch s, hipStream_t st,
                     hipEvent_t *ev);
