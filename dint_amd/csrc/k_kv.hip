// k_kv.hip -- placeholder until the KV kernels land (store / tatp / smallbank).
#include "dint_kv.h"
#include "../../include/dint_abi.h"
int dint_kv_create(dint_kv *, uint32_t, uint64_t, dint_shard) { return DINT_EINVAL; }
void dint_kv_destroy(dint_kv *) {}
void dint_kv_reset(dint_kv *) {}
std::vector<std::pair<void *, size_t>> dint_kv_regions(dint_kv *) { return {}; }
int dint_kv_load_rows(dint_kv *, uint32_t, const uint64_t *, const uint32_t *, const uint8_t *, uint64_t, dint_scratch, hipStream_t) { return DINT_ESTATE; }
int dint_kv_populate(dint_kv *, uint32_t, uint64_t, dint_scratch, hipStream_t) { return DINT_ESTATE; }
int64_t dint_kv_dump_rows(dint_kv *, uint32_t, uint64_t *, uint32_t *, uint8_t *, uint64_t) { return DINT_ESTATE; }
int64_t dint_kv_read_locks(dint_kv *, uint32_t, uint32_t *, uint32_t *, uint64_t) { return DINT_ESTATE; }
void dint_launch_kv(const void *, void *, uint32_t, dint_kv, dint_log, dint_shard, dint_scratch, hipStream_t, hipEvent_t *) {}
void dint_launch_home_kv(const void *, uint32_t, dint_kv, uint32_t, uint8_t *, hipStream_t) {}
