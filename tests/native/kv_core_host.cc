// Host build of the HBM table layout (dint_amd/csrc/dint_kv_core.h) for the CPU unit tests:
// the same find/get/set/insert/delete code the HIP kernels run, over host memory, single thread.
// TEST TOOLING ONLY -- built by tests/test_kv_core_host.py with g++.
#include <stdlib.h>

#include <vector>

#include "../../dint_amd/csrc/dint_kv_core.h"

struct kvh {
  kv_tab t;
  uint32_t pool_top;
  unsigned long long free_head[KV_NLISTS], pend_head[KV_NLISTS];
};

extern "C" {

kvh *kvh_create(uint64_t n_buckets, uint32_t pool_cap, uint32_t val_size) {
  kvh *h = (kvh *)calloc(1, sizeof(kvh));
  h->t.n_local = n_buckets;
  h->t.pool_cap = pool_cap;
  h->t.stride = val_size == 40 ? 256 : 128;
  h->t.val_size = val_size;
  h->t.entries = (uint8_t *)calloc(n_buckets + pool_cap, h->t.stride);
  h->t.pool_next = (uint32_t *)calloc(pool_cap ? pool_cap : 1, 4);
  h->t.pool_top = &h->pool_top;
  h->t.free_head = h->free_head;
  h->t.pend_head = h->pend_head;
  return h;
}
void kvh_destroy(kvh *h) {
  free(h->t.entries);
  free(h->t.pool_next);
  free(h);
}
int kvh_get(kvh *h, uint64_t bucket, uint64_t key, uint8_t *val, uint32_t *ver) { return kv_get(h->t, bucket, key, val, ver) ? 0 : 1; }
int kvh_set(kvh *h, uint64_t bucket, uint64_t key, const uint8_t *val) { return kv_set(h->t, bucket, key, val) ? 0 : 1; }
int kvh_insert(kvh *h, uint64_t bucket, uint64_t key, const uint8_t *val, uint32_t ver) {
  return kv_insert<kv_host_mem>(h->t, bucket, key, val, ver) ? 0 : 1;
}
int kvh_delete(kvh *h, uint64_t bucket, uint64_t key) { return kv_delete<kv_host_mem>(h->t, bucket, key) ? 0 : 1; }
void kvh_rotate(kvh *h) { for (uint32_t l = 0; l < KV_NLISTS; l++) kv_pool_rotate<kv_host_mem>(h->t, l); }
uint32_t kvh_pool_top(kvh *h) { return h->pool_top; }
// lock words share the inline entry with the rows: poke them to prove row ops never clobber them
void kvh_set_lock_bytes(kvh *h, uint64_t bucket, uint32_t v) {
  kv_entry_hdr(h->t, bucket, KV_INLINE)->lockw = v;
}
uint32_t kvh_get_lock_bytes(kvh *h, uint64_t bucket) {
  uint32_t v;
  v = kv_entry_hdr(h->t, bucket, KV_INLINE)->lockw;
  return v;
}
// valid rows in bucket order, chain order inside a bucket
uint64_t kvh_dump(kvh *h, uint64_t *keys, uint32_t *vers, uint8_t *vals, uint64_t cap) {
  uint64_t n = 0;
  for (uint64_t b = 0; b < h->t.n_local; b++) {
    uint32_t cur = kv_entry_hdr(h->t, b, KV_INLINE)->head;
    while (cur) {
      const uint8_t *e = kv_entry_ptr(h->t, b, cur);
      const kv_hdr *hd = (const kv_hdr *)e;
      for (int i = 0; i < 4; i++)
        if (kv_valid(*hd, i)) {
          if (n < cap) {
            keys[n] = hd->key[i];
            vers[n] = hd->ver[i];
            memcpy(vals + n * h->t.val_size, e + KV_VAL_OFF + i * h->t.val_size, h->t.val_size);
          }
          n++;
        }
      cur = hd->next;
    }
  }
  return n;
}
}
