// dint_kv.h -- HBM key-value tables of the store / tatp / smallbank workloads (k_kv.hip).
#pragma once
#include <utility>
#include <vector>

#include "dint_kernels.h"

struct dint_kv_table {
  uint64_t hash_size = 0;     // global bucket count (before sharding)
  uint64_t n_local = 0;       // buckets stored on this shard
  dint_mod mod{};             // % hash_size
  dint_mod lock_mod{};        // % (4 * hash_size)   (tatp/udp/tatp.h:12-14)
  uint8_t *entries = nullptr; // n_local inline entries + overflow pool
  uint32_t gk_base = 0;       // first group key of this table
};

struct dint_kv {
  uint32_t workload = 0;
  uint32_t n_tables = 0;
  uint32_t val_size = 0;
  dint_kv_table tab[5];
};

int dint_kv_create(dint_kv *kv, uint32_t workload, uint64_t n_rows, dint_shard shard);
void dint_kv_destroy(dint_kv *kv);
void dint_kv_reset(dint_kv *kv);
std::vector<std::pair<void *, size_t>> dint_kv_regions(dint_kv *kv);
int dint_kv_load_rows(dint_kv *kv, uint32_t table, const uint64_t *keys, const uint32_t *vers, const uint8_t *vals,
                      uint64_t n, dint_scratch s, hipStream_t st);
int dint_kv_populate(dint_kv *kv, uint32_t workload, uint64_t populate_n, dint_scratch s, hipStream_t st);
int64_t dint_kv_dump_rows(dint_kv *kv, uint32_t table, uint64_t *keys, uint32_t *vers, uint8_t *vals, uint64_t cap);
int64_t dint_kv_read_locks(dint_kv *kv, uint32_t table, uint32_t *a, uint32_t *b, uint64_t cap);
void dint_launch_kv(const void *d_req, void *d_rep, uint32_t n, dint_kv kv, dint_log log, dint_shard shard,
                    dint_scratch s, hipStream_t st, hipEvent_t *ev);
void dint_launch_home_kv(const void *d_req, uint32_t n, dint_kv kv, uint32_t shard_count, uint8_t *d_home,
                         hipStream_t st);
