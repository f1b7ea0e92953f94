// dint_kv_core.h -- the HBM key-value table of the store / tatp / smallbank workloads: layout and
// the five single-key operations (find / get / set / insert / delete).
//
// Reference semantics (one bucket = a chain of 4-slot entries, newest entry first):
//   store/udp/kvs.h:37-136, tatp/udp/kvs.h:55-153, smallbank/udp/kvs.h:51-150
//     get    : first slot in chain order with key == k && valid
//     set    : same slot; copy val, ver++
//     insert : first invalid slot in chain order, else a NEW entry prepended to the chain; ver = 0
//     delete : clear valid; an entry whose 4 slots are all invalid is unlinked and freed
// The chain order is reproduced exactly (also for duplicate keys), but the memory layout is ours:
//
//   entries[]  = n_local INLINE entries (one per local bucket, at index = local bucket)
//              + pool_cap OVERFLOW entries (bump allocated, recycled through a deferred free list)
//   entry      = 64-byte header sector {key[4], ver[4], valid[4], next, head, lockb[4]}
//              + 4 values (val_size bytes each) + (smallbank) 4 x {num_ex, num_sh}
//     stride 256 B for 40-byte values (store / tatp), 128 B for 8-byte values (smallbank).
//   A link is 0 = end of chain, 1 = the bucket's own inline entry, k >= 2 = pool entry k-2.
//   The chain head of a bucket lives in its inline header (`head`), so a lookup in a bucket that
//   never overflowed costs exactly one 64-byte sector for the probe (+ the value sector(s) on a hit);
//   the reference needs two dependent misses (bucket_heads[] -> entry).  The inline entry is an
//   ordinary chain node: it is used for the first entry the reference would `new`, and again whenever
//   it is not linked at the time the reference would allocate.
//   The per-bucket lock words of the shard servers live in the same inline entry:
//     tatp      txn_locks[table][lock_hash]          -> hdr.lockb[q]     (q = lock_hash / hash_size, 0..3)
//     smallbank num_ex/num_sh[table][lock_hash]      -> u32 pair at byte 96 + 8 q
//   because lock_hash % hash_size == bucket (tatp/udp/tatp.h:12-14 vs kvs.h:51-53).
//
// The functions are templated on a memory policy so the same code runs inside the HIP kernels
// (device policy, k_kv.hip) and in a host build used only by the CPU unit tests of this layout.
#pragma once
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define KV_HD __host__ __device__
#else
#define KV_HD
#endif

#define KV_NULL 0u
#define KV_INLINE 1u
#define KV_MAX_CHAIN 4096u  // walk bound: a corrupt chain must never hang a GPU

struct kv_hdr {
  uint64_t key[4];
  uint32_t ver[4];
  uint8_t valid[4];
  uint32_t next;     // successor of this entry in its chain
  uint32_t head;     // inline entries only: first entry of the bucket's chain
  uint8_t lockb[4];  // inline entries only: tatp txn lock per quadrant
};
static_assert(sizeof(kv_hdr) == 64, "header must be one 64-byte sector");

#define KV_VAL_OFF 64u
#define KV_SB_LOCK_OFF 96u  // smallbank: 4 x {u32 num_ex, u32 num_sh}

// device view of one table (plain pointers, passed to kernels by value)
struct kv_tab {
  uint8_t *entries;
  uint64_t n_local;      // inline entries (= local buckets)
  uint32_t pool_cap;     // overflow entries
  uint32_t stride;       // bytes per entry
  uint32_t val_size;
  uint32_t *pool_top;    // bump allocator (entries handed out so far)
  uint32_t *pool_next;   // [pool_cap] free-list links, touched with atomics only
  // {tag:32, link:32}; frees go to `pend`; `pend` becomes poppable at the next pass boundary, when the
  // freeing workgroup's dirty lines are known to have left its XCD's L2 (see kv_pool_rotate)
  unsigned long long *free_head;
  unsigned long long *pend_head;
};

KV_HD static inline uint8_t *kv_entry_ptr(const kv_tab &t, uint64_t bucket, uint32_t link) {
  const uint64_t e = (link == KV_INLINE) ? bucket : t.n_local + (uint64_t)(link - 2u);
  return t.entries + e * (uint64_t)t.stride;
}
KV_HD static inline kv_hdr *kv_entry_hdr(const kv_tab &t, uint64_t bucket, uint32_t link) {
  return (kv_hdr *)kv_entry_ptr(t, bucket, link);
}

struct kv_loc {
  uint32_t link;  // entry holding the key
  uint32_t slot;  // 0..3
  uint32_t prev;  // predecessor entry in the chain (KV_NULL = the entry is the chain head)
};

// ---- memory policy of the host build (single thread, plain memory) ---------------------------------
struct kv_host_mem {
  static inline uint32_t fetch_add(uint32_t *p, uint32_t v) { uint32_t o = *p; *p += v; return o; }
  static inline uint32_t load32(uint32_t *p) { return *p; }
  static inline void store32(uint32_t *p, uint32_t v) { *p = v; }
  static inline unsigned long long load64(unsigned long long *p) { return *p; }
  static inline bool cas64(unsigned long long *p, unsigned long long exp, unsigned long long des) {
    if (*p != exp) return false;
    *p = des;
    return true;
  }
};

// ---- overflow-entry pool -------------------------------------------------------------------------------
template <class M>
KV_HD static inline uint32_t kv_pool_alloc(const kv_tab &t) {
  for (uint32_t spin = 0; spin < 1024; spin++) {  // recycled entries first
    const unsigned long long old = M::load64(t.free_head);
    const uint32_t link = (uint32_t)old;
    if (link == KV_NULL) break;
    const uint32_t nxt = M::load32(&t.pool_next[link - 2u]);
    if (M::cas64(t.free_head, old, ((old >> 32) + 1ull) << 32 | nxt)) return link;
  }
  const uint32_t p = M::fetch_add(t.pool_top, 1u);
  if (p >= t.pool_cap) {
    M::store32(t.pool_top, t.pool_cap);  // keep the counter from wrapping after ~4G failed inserts
    return KV_NULL;
  }
  return p + 2u;
}
template <class M>
KV_HD static inline void kv_pool_free(const kv_tab &t, uint32_t link) {
  for (uint32_t spin = 0; spin < 65536; spin++) {
    const unsigned long long old = M::load64(t.pend_head);
    M::store32(&t.pool_next[link - 2u], (uint32_t)old);
    if (M::cas64(t.pend_head, old, ((old >> 32) + 1ull) << 32 | link)) return;
  }
  // give up: the entry leaks (bounded spin, never reached in practice)
}
// pass boundary (single thread, no pass in flight): entries freed during earlier passes become poppable
template <class M>
KV_HD static inline void kv_pool_rotate(const kv_tab &t) {
  const unsigned long long f = M::load64(t.free_head), p = M::load64(t.pend_head);
  if ((uint32_t)f == KV_NULL && (uint32_t)p != KV_NULL) {
    *t.free_head = ((f >> 32) + 1ull) << 32 | (uint32_t)p;
    *t.pend_head = ((p >> 32) + 1ull) << 32;
  }
}

// ---- lookups ---------------------------------------------------------------------------------------------
// kvs_get / kvs_set / kvs_delete all start with the same walk (kvs.h:59-70)
KV_HD static inline bool kv_find(const kv_tab &t, uint64_t bucket, uint64_t key, kv_loc *loc) {
  uint32_t cur = kv_entry_hdr(t, bucket, KV_INLINE)->head, prev = KV_NULL;
  for (uint32_t steps = 0; cur != KV_NULL && steps < KV_MAX_CHAIN; steps++) {
    const kv_hdr *h = kv_entry_hdr(t, bucket, cur);
#pragma unroll
    for (uint32_t i = 0; i < 4; i++)
      if (h->key[i] == key && h->valid[i]) {
        loc->link = cur;
        loc->slot = i;
        loc->prev = prev;
        return true;
      }
    prev = cur;
    cur = h->next;
  }
  return false;
}

KV_HD static inline void kv_copy_words(uint8_t *dst, const uint8_t *src, uint32_t bytes) {
  // entry values are 4-byte aligned; message values are not (packed wire structs) -> memcpy per word
  for (uint32_t o = 0; o < bytes; o += 4) {
    uint32_t w;
    __builtin_memcpy(&w, src + o, 4);
    __builtin_memcpy(dst + o, &w, 4);
  }
}

// kvs_get (kvs.h:55-73): true = found; val_out may be unaligned
KV_HD static inline bool kv_get(const kv_tab &t, uint64_t bucket, uint64_t key, uint8_t *val_out, uint32_t *ver_out) {
  kv_loc l;
  if (!kv_find(t, bucket, key, &l)) return false;
  const uint8_t *e = kv_entry_ptr(t, bucket, l.link);
  kv_copy_words(val_out, e + KV_VAL_OFF + l.slot * t.val_size, t.val_size);
  *ver_out = ((const kv_hdr *)e)->ver[l.slot];
  return true;
}

// kvs_set (kvs.h:75-92): true = found (the reference panics otherwise)
KV_HD static inline bool kv_set(const kv_tab &t, uint64_t bucket, uint64_t key, const uint8_t *val) {
  kv_loc l;
  if (!kv_find(t, bucket, key, &l)) return false;
  uint8_t *e = kv_entry_ptr(t, bucket, l.link);
  kv_copy_words(e + KV_VAL_OFF + l.slot * t.val_size, val, t.val_size);
  ((kv_hdr *)e)->ver[l.slot]++;
  return true;
}

// kvs_insert (kvs.h:94-121) with an explicit initial version (0 for the wire INSERT ops).
// false = the overflow pool is exhausted and the row was dropped.
template <class M>
KV_HD static inline bool kv_insert(const kv_tab &t, uint64_t bucket, uint64_t key, const uint8_t *val, uint32_t ver) {
  kv_hdr *ih = kv_entry_hdr(t, bucket, KV_INLINE);
  const uint32_t head = ih->head;
  uint32_t cur = head;
  bool inline_linked = false;
  for (uint32_t steps = 0; cur != KV_NULL && steps < KV_MAX_CHAIN; steps++) {
    uint8_t *e = kv_entry_ptr(t, bucket, cur);
    kv_hdr *h = (kv_hdr *)e;
    if (cur == KV_INLINE) inline_linked = true;
#pragma unroll
    for (uint32_t i = 0; i < 4; i++)
      if (!h->valid[i]) {
        h->key[i] = key;
        kv_copy_words(e + KV_VAL_OFF + i * t.val_size, val, t.val_size);
        h->ver[i] = ver;
        h->valid[i] = 1;
        return true;
      }
    cur = h->next;
  }
  // every slot of the chain is taken: new entry, prepended (kvs.h:112-119)
  const uint32_t nl = inline_linked ? kv_pool_alloc<M>(t) : KV_INLINE;
  if (nl == KV_NULL) return false;
  uint8_t *e = kv_entry_ptr(t, bucket, nl);
  kv_hdr *h = (kv_hdr *)e;
  h->key[0] = key;
  h->key[1] = h->key[2] = h->key[3] = 0;
  h->ver[0] = ver;
  h->ver[1] = h->ver[2] = h->ver[3] = 0;
  h->valid[0] = 1;
  h->valid[1] = h->valid[2] = h->valid[3] = 0;
  kv_copy_words(e + KV_VAL_OFF, val, t.val_size);
  h->next = head;
  ih->head = nl;  // for nl == KV_INLINE this is the same header: head and lock words are preserved
  return true;
}

// kvs_delete (kvs.h:123-153): true = found (the reference panics otherwise)
template <class M>
KV_HD static inline bool kv_delete(const kv_tab &t, uint64_t bucket, uint64_t key) {
  kv_loc l;
  if (!kv_find(t, bucket, key, &l)) return false;
  kv_hdr *h = kv_entry_hdr(t, bucket, l.link);
  h->valid[l.slot] = 0;
  if (!(h->valid[0] | h->valid[1] | h->valid[2] | h->valid[3])) {  // unlink and free the empty entry
    const uint32_t nxt = h->next;
    if (l.prev == KV_NULL) kv_entry_hdr(t, bucket, KV_INLINE)->head = nxt;
    else kv_entry_hdr(t, bucket, l.prev)->next = nxt;
    if (l.link != KV_INLINE) kv_pool_free<M>(t, l.link);
  }
  return true;
}
