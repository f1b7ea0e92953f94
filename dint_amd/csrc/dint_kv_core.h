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
//   entry      = 64-byte header sector {key[4], ver[4], validw, next, head, lockw}
//              + 4 values (val_size bytes each) + (smallbank) 4 x {num_ex, num_sh}
//     stride 256 B for 40-byte values (store / tatp), 128 B for 8-byte values (smallbank).
//   A link is 0 = end of chain, 1 = the bucket's own inline entry, k >= 2 = pool entry k-2.
//   The chain head of a bucket lives in its inline header (`head`), so a lookup in a bucket that
//   never overflowed costs exactly one 64-byte sector for the probe (+ the value sector(s) on a hit);
//   the reference needs two dependent misses (bucket_heads[] -> entry).  The inline entry is an
//   ordinary chain node: it is used for the first entry the reference would `new`, and again whenever
//   it is not linked at the time the reference would allocate.
//   The per-bucket lock words of the shard servers live in the same inline entry:
//     tatp      txn_locks[table][lock_hash]          -> byte q of hdr.lockw (q = lock_hash / hash_size, 0..3)
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
#define KV_NLISTS 64u        // free lists per table

struct kv_hdr {
  uint64_t key[4];
  uint32_t ver[4];
  uint32_t validw;   // valid flags of the 4 slots: slot i in bits 8i..8i+7 (byte i in memory), 0 or 1
  uint32_t next;     // successor of this entry in its chain
  uint32_t head;     // inline entries only: first entry of the bucket's chain
  uint32_t lockw;    // inline entries only: tatp txn lock bytes, quadrant q in bits 8q..8q+7 (byte q in memory)
};
static_assert(sizeof(kv_hdr) == 64, "header must be one 64-byte sector");

#define KV_VAL_OFF 64u
#define KV_VALID_OFF 48u     // byte offset of validw in a header
#define KV_LOCKB_OFF 60u     // byte offset of lockw in the inline header
#define KV_SB_LOCK_OFF 96u  // smallbank: 4 x {u32 num_ex, u32 num_sh}
#define KV_OWNER_OFF 224u   // tatp (256-byte entries), DINT_FLAG_LOCK_SAME_KEY: 4 x u64 key the quadrant's lock was granted to

// device view of one table (plain pointers, passed to kernels by value)
struct kv_tab {
  uint8_t *entries;
  uint64_t n_local;      // inline entries (= local buckets)
  uint32_t pool_cap;     // overflow entries
  uint32_t stride;       // bytes per entry
  uint32_t val_size;
  uint32_t *pool_top;    // bump allocator (entries handed out so far)
  uint32_t *pool_next;   // [pool_cap] free-list links, touched with atomics only
  // KV_NLISTS free lists, each {tag:32, link:32}; a workgroup uses list (its bin & (KV_NLISTS-1)), so concurrent
  // inserts / deletes rarely CAS the same word.  Frees go to `pend`; `pend` becomes poppable at the next pass
  // boundary, when the freeing workgroup's dirty lines are known to have left its XCD's L2 (see kv_pool_rotate)
  unsigned long long *free_head;  // [KV_NLISTS]
  unsigned long long *pend_head;  // [KV_NLISTS]
};

// Device build: the table descriptors reach the kernels through memory (a kv_dev in HBM, copied to LDS), so the compiler
// sees GENERIC pointers and would access the tables with flat instructions -- which count against the LDS counter as well
// and so serialise with the shuffles and LDS accesses around them.  The tables (and the messages) live in HBM; the hot
// accessors say so with address-space-1 pointers (a cast to address space 1 and back is folded away: the ACCESS has to go
// through the qualified type).  Host build (the CPU unit tests of this layout): plain pointers.
#if defined(__HIP_DEVICE_COMPILE__)
#define KV_G(T) __attribute__((address_space(1))) T
#else
#define KV_G(T) T
#endif
#define KV_LD(T, p) (*(const KV_G(T) *)(p))
#define KV_ST(T, p, v) (*(KV_G(T) *)(p) = (v))
KV_HD static inline uint8_t *kv_entry_ptr(const kv_tab &t, uint64_t bucket, uint32_t link) {
  const uint64_t e = (link == KV_INLINE) ? bucket : t.n_local + (uint64_t)(link - 2u);
  return t.entries + e * (uint64_t)t.stride;
}
KV_HD static inline kv_hdr *kv_entry_hdr(const kv_tab &t, uint64_t bucket, uint32_t link) {
  return (kv_hdr *)kv_entry_ptr(t, bucket, link);
}
// field-wise copy: keeps headers in registers (an aggregate copy is lowered through stack memory on the GPU)
KV_HD static inline void kv_hdr_copy(kv_hdr &d, const kv_hdr &s) {
  d.key[0] = s.key[0]; d.key[1] = s.key[1]; d.key[2] = s.key[2]; d.key[3] = s.key[3];
  d.ver[0] = s.ver[0]; d.ver[1] = s.ver[1]; d.ver[2] = s.ver[2]; d.ver[3] = s.ver[3];
  d.validw = s.validw; d.next = s.next; d.head = s.head; d.lockw = s.lockw;
}
// a header from table memory (HBM) into registers
KV_HD static inline void kv_hdr_load(kv_hdr &d, const void *p) {
  const KV_G(kv_hdr) *s = (const KV_G(kv_hdr) *)p;
  d.key[0] = s->key[0]; d.key[1] = s->key[1]; d.key[2] = s->key[2]; d.key[3] = s->key[3];
  d.ver[0] = s->ver[0]; d.ver[1] = s->ver[1]; d.ver[2] = s->ver[2]; d.ver[3] = s->ver[3];
  d.validw = s->validw; d.next = s->next; d.head = s->head; d.lockw = s->lockw;
}
KV_HD static inline bool kv_valid(const kv_hdr &h, uint32_t slot) { return (h.validw >> (8 * slot)) & 0xFFu; }


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
KV_HD static inline uint32_t kv_pool_alloc(const kv_tab &t, uint32_t lst) {
  unsigned long long *fh = t.free_head + (lst & (KV_NLISTS - 1));
  for (uint32_t spin = 0; spin < 1024; spin++) {  // recycled entries first
    const unsigned long long old = M::load64(fh);
    const uint32_t link = (uint32_t)old;
    if (link == KV_NULL) break;
    const uint32_t nxt = M::load32(&t.pool_next[link - 2u]);
    if (M::cas64(fh, old, ((old >> 32) + 1ull) << 32 | nxt)) return link;
  }
  const uint32_t p = M::fetch_add(t.pool_top, 1u);
  if (p >= t.pool_cap) {
    M::store32(t.pool_top, t.pool_cap);  // keep the counter from wrapping after ~4G failed inserts
    return KV_NULL;
  }
  return p + 2u;
}
template <class M>
KV_HD static inline void kv_pool_free(const kv_tab &t, uint32_t link, uint32_t lst) {
  unsigned long long *ph = t.pend_head + (lst & (KV_NLISTS - 1));
  for (uint32_t spin = 0; spin < 65536; spin++) {
    const unsigned long long old = M::load64(ph);
    M::store32(&t.pool_next[link - 2u], (uint32_t)old);
    if (M::cas64(ph, old, ((old >> 32) + 1ull) << 32 | link)) return;
  }
  // give up: the entry leaks (bounded spin, never reached in practice)
}
// Pass boundary (one thread per list): entries freed during EARLIER passes become poppable.  The device keeps TWO sets of
// pend lists and a pass pushes its frees to set (pass number & 1) (k_kv.hip: kv_dev_to_lds), so the set a pass is about to
// push to -- `ph_off` = that set's offset in lists -- holds frees of the pass before the previous one and older: nobody
// pushes to it while it is rotated, also when this pass's partition kernel runs beside the previous pass's hot-key kernels
// (r06: k_kv_hot_part).  A free list is only ever rotated INTO when it is empty, through a CAS: a concurrent pop of an empty
// list fails and bump-allocates, so pops of the previous pass beside the rotation are safe.  (Host build: one set, offset 0.)
template <class M>
KV_HD static inline void kv_pool_rotate(const kv_tab &t, uint32_t lst, uint32_t ph_off = 0) {
  unsigned long long *fh = t.free_head + lst, *ph = t.pend_head + ph_off + lst;
  const unsigned long long f = M::load64(fh), p = M::load64(ph);
  if ((uint32_t)f == KV_NULL && (uint32_t)p != KV_NULL) {
    if (M::cas64(ph, p, ((p >> 32) + 1ull) << 32)) {  // (nobody pushes to this set now; the CAS only keeps the tag discipline)
      if (!M::cas64(fh, f, ((f >> 32) + 1ull) << 32 | (uint32_t)p)) {
        // the free list changed under us (cannot happen: only a rotation writes an empty list) -- put the chain back
        M::cas64(ph, ((p >> 32) + 1ull) << 32, ((p >> 32) + 2ull) << 32 | (uint32_t)p);
      }
    }
  }
}

// ---- the one table operation -------------------------------------------------------------------------------
// Every request does at most one of: GET (kvs_get), SET (kvs_set), INS (kvs_insert), DEL (kvs_delete).  They all
// start with the same chain walk (kvs.h:59-70 / 97-110 / 127-150), so they are one function: a single walk that
// records the first matching slot and the first invalid slot, then a short action.  On the GPU this matters:
// the lanes of a wave run different request types, and one shared load phase (the caller passes the bucket's
// inline header H, already loaded) costs one memory round trip instead of one per request type.
enum : uint32_t { KV_ACT_NONE = 0, KV_ACT_GET = 1, KV_ACT_SET = 2, KV_ACT_INS = 3, KV_ACT_DEL = 4 };

struct kv_res {
  bool ok;       // GET / SET / DEL: the key was found.  INS: the row was stored (false = pool exhausted)
  uint32_t ver;  // GET: the row's version
};

// Copy one value (40 or 8 bytes).  All loads are issued before the first store: source and destination may
// alias as far as the compiler knows, and a load/store/load/store chain would cost one memory round trip per
// word on the GPU.  Entry values are 4-byte aligned; message values are not (packed wire structs) -> memcpy.
KV_HD static inline void kv_copy_words(uint8_t *dst, const uint8_t *src, uint32_t bytes) {
  // (packed / may_alias word types: message values are unaligned; in the device build both sides are HBM, see KV_G)
  typedef uint32_t __attribute__((aligned(1), may_alias)) kv_w32;
  const KV_G(kv_w32) *s = (const KV_G(kv_w32) *)src;
  KV_G(kv_w32) *d = (KV_G(kv_w32) *)dst;
  uint32_t w[10];
  if (bytes == 40) {
#pragma unroll
    for (uint32_t k = 0; k < 10; k++) w[k] = s[k];
#pragma unroll
    for (uint32_t k = 0; k < 10; k++) d[k] = w[k];
  } else {
#pragma unroll
    for (uint32_t k = 0; k < 2; k++) w[k] = s[k];
#pragma unroll
    for (uint32_t k = 0; k < 2; k++) d[k] = w[k];
  }
}

// `H` = a copy of the bucket's inline header as it is in memory now.  val: output of GET, input of SET / INS
// (may be unaligned).  ins_ver: version of an inserted row (0 for the wire INSERT ops).
template <class M>
KV_HD static inline kv_res kv_apply(const kv_tab &t, uint64_t bucket, const kv_hdr &H, uint32_t act, uint64_t key,
                                    uint8_t *val, uint32_t ins_ver, uint32_t lst = 0) {
  kv_res res = {false, 0};
  if (act == KV_ACT_NONE) return res;
  const bool want_match = act != KV_ACT_INS;
  bool matched = false, have_free = false, inline_linked = false;
  uint32_t m_link = 0, m_slot = 0, m_prev = 0, m_next = 0, m_ver = 0, m_valid = 0;  // m_valid: valid[0..3] packed
  uint32_t f_link = 0, f_slot = 0;
  uint32_t cur = H.head, prev = KV_NULL;
  for (uint32_t steps = 0; cur != KV_NULL && steps < KV_MAX_CHAIN; steps++) {
    // the inline entry's header is already in registers (it does not change between the caller's load and here)
    kv_hdr h;
    if (cur == KV_INLINE) { kv_hdr_copy(h, H); inline_linked = true; }
    else kv_hdr_load(h, kv_entry_hdr(t, bucket, cur));
#pragma unroll
    for (uint32_t i = 0; i < 4; i++) {
      if (want_match && !matched && kv_valid(h, i) && h.key[i] == key) {
        matched = true; m_link = cur; m_slot = i; m_prev = prev; m_next = h.next; m_ver = h.ver[i];
        m_valid = h.validw;
      }
      if (!want_match && !have_free && !kv_valid(h, i)) { have_free = true; f_link = cur; f_slot = i; }
    }
    if (matched || have_free) break;
    prev = cur;
    cur = h.next;
  }
  switch (act) {
    case KV_ACT_GET:  // kvs.h:55-73
      if (matched) {
        kv_copy_words(val, kv_entry_ptr(t, bucket, m_link) + KV_VAL_OFF + m_slot * t.val_size, t.val_size);
        res.ok = true;
        res.ver = m_ver;
      }
      break;
    case KV_ACT_SET:  // kvs.h:75-92
      if (matched) {
        uint8_t *e = kv_entry_ptr(t, bucket, m_link);
        kv_copy_words(e + KV_VAL_OFF + m_slot * t.val_size, val, t.val_size);
        ((kv_hdr *)e)->ver[m_slot] = m_ver + 1;
        res.ok = true;
      }
      break;
    case KV_ACT_INS: {  // kvs.h:94-121
      uint32_t link = f_link, slot = f_slot;
      if (!have_free) {  // every slot of the chain is taken: a new entry, prepended (kvs.h:112-119)
        link = inline_linked ? kv_pool_alloc<M>(t, lst) : KV_INLINE;
        if (link == KV_NULL) break;
        slot = 0;
      }
      uint8_t *e = kv_entry_ptr(t, bucket, link);
      kv_hdr *h = (kv_hdr *)e;
      h->key[slot] = key;
      h->ver[slot] = ins_ver;
      kv_copy_words(e + KV_VAL_OFF + slot * t.val_size, val, t.val_size);
      if (have_free) {
        e[KV_VALID_OFF + slot] = 1;
      } else {
        h->key[1] = h->key[2] = h->key[3] = 0;
        h->ver[1] = h->ver[2] = h->ver[3] = 0;
        h->validw = 1;  // slot 0 valid, 1..3 invalid
        h->next = H.head;
        kv_entry_hdr(t, bucket, KV_INLINE)->head = link;  // for link == KV_INLINE the same header: lock words stay
      }
      res.ok = true;
      break;
    }
    default:  // KV_ACT_DEL  kvs.h:123-153
      if (matched) {
        kv_hdr *h = kv_entry_hdr(t, bucket, m_link);
        ((uint8_t *)h)[KV_VALID_OFF + m_slot] = 0;
        if ((m_valid & ~(0xFFu << (8 * m_slot))) == 0) {  // the entry is empty now: unlink it and free it
          if (m_prev == KV_NULL) kv_entry_hdr(t, bucket, KV_INLINE)->head = m_next;
          else kv_entry_hdr(t, bucket, m_prev)->next = m_next;
          if (m_link != KV_INLINE) kv_pool_free<M>(t, m_link, lst);
        }
        res.ok = true;
      }
      break;
  }
  return res;
}

// ---- the reference's four functions, as thin wrappers (host tests, dumps, non-hot paths) -----------------------
KV_HD static inline bool kv_get(const kv_tab &t, uint64_t bucket, uint64_t key, uint8_t *val_out, uint32_t *ver_out) {
  const kv_hdr H = *kv_entry_hdr(t, bucket, KV_INLINE);
  const kv_res r = kv_apply<kv_host_mem>(t, bucket, H, KV_ACT_GET, key, val_out, 0);  // GET never touches the pool
  if (r.ok) *ver_out = r.ver;
  return r.ok;
}
KV_HD static inline bool kv_set(const kv_tab &t, uint64_t bucket, uint64_t key, const uint8_t *val) {
  const kv_hdr H = *kv_entry_hdr(t, bucket, KV_INLINE);
  return kv_apply<kv_host_mem>(t, bucket, H, KV_ACT_SET, key, (uint8_t *)val, 0).ok;
}
template <class M>
KV_HD static inline bool kv_insert(const kv_tab &t, uint64_t bucket, uint64_t key, const uint8_t *val, uint32_t ver) {
  const kv_hdr H = *kv_entry_hdr(t, bucket, KV_INLINE);
  return kv_apply<M>(t, bucket, H, KV_ACT_INS, key, (uint8_t *)val, ver).ok;
}
template <class M>
KV_HD static inline bool kv_delete(const kv_tab &t, uint64_t bucket, uint64_t key) {
  const kv_hdr H = *kv_entry_hdr(t, bucket, KV_INLINE);
  return kv_apply<M>(t, bucket, H, KV_ACT_DEL, key, nullptr, 0).ok;
}
// where a key lives (closed-form group resolution, k_kv.hip)
struct kv_loc {
  uint32_t link;  // entry holding the key
  uint32_t slot;  // 0..3
};
struct kv_where {
  uint32_t found, link, slot, ver;
};
// the lookup walk alone, from the preloaded inline header H
KV_HD static inline kv_where kv_locate(const kv_tab &t, uint64_t bucket, const kv_hdr &H, uint64_t key) {
  kv_where w = {0, 0, 0, 0};
  uint32_t cur = H.head;
  for (uint32_t steps = 0; cur != KV_NULL && steps < KV_MAX_CHAIN; steps++) {
    kv_hdr h;
    if (cur == KV_INLINE) kv_hdr_copy(h, H);
    else kv_hdr_load(h, kv_entry_hdr(t, bucket, cur));
#pragma unroll
    for (uint32_t i = 0; i < 4; i++)
      if (!w.found && kv_valid(h, i) && h.key[i] == key) { w.found = 1; w.link = cur; w.slot = i; w.ver = h.ver[i]; }
    if (w.found) break;
    cur = h.next;
  }
  return w;
}
// Is there a SECOND valid row with this key after the one kv_locate found (w)?  The reference's kvs_insert does not
// check for the key (kvs.h:94-121), so a trace that inserts an existing key leaves duplicate rows; a later delete then
// removes only the first one and reads see the next.  The same-key closed forms assume one row per key: a key
// segment that inserts / deletes asks this first and goes request by request when the answer is yes.
KV_HD static inline bool kv_has_dup(const kv_tab &t, uint64_t bucket, const kv_hdr &H, uint64_t key, const kv_where &w) {
  if (!w.found) return false;
  uint32_t cur = w.link;
  bool first = true;
  for (uint32_t steps = 0; cur != KV_NULL && steps < KV_MAX_CHAIN; steps++) {
    kv_hdr h;
    if (cur == KV_INLINE) kv_hdr_copy(h, H);
    else kv_hdr_load(h, kv_entry_hdr(t, bucket, cur));
#pragma unroll
    for (uint32_t i = 0; i < 4; i++)
      if ((!first || i > w.slot) && kv_valid(h, i) && h.key[i] == key) return true;
    first = false;
    cur = h.next;
  }
  return false;
}
KV_HD static inline bool kv_find(const kv_tab &t, uint64_t bucket, uint64_t key, kv_loc *loc) {
  uint32_t cur = kv_entry_hdr(t, bucket, KV_INLINE)->head;
  for (uint32_t steps = 0; cur != KV_NULL && steps < KV_MAX_CHAIN; steps++) {
    const kv_hdr *h = kv_entry_hdr(t, bucket, cur);
#pragma unroll
    for (uint32_t i = 0; i < 4; i++)
      if (h->key[i] == key && kv_valid(*h, i)) {
        loc->link = cur;
        loc->slot = i;
        return true;
      }
    cur = h->next;
  }
  return false;
}
