/*
 * dint_oracle.c -- CPU restatement of the DINT per-packet server state machines.
 *
 * TEST INFRASTRUCTURE ONLY (see dint_oracle.h).  Serial, single-thread semantics
 * of the reference `udp/` servers; table sizes are run-time parameters so the
 * same code checks the reference sizes (36M slots, 7M subscribers, 24M accounts)
 * and the BASELINE.json sizes (1M slots, 1M subscribers, ...).
 *
 * Deviations from the reference, all documented in SURVEY.md 8 "parity target":
 *   - unknown packet types: the reference panics (exit 1); here the message is
 *     left untouched and counted in the return value of *_replay;
 *   - tatp kvs_set / kvs_delete / smallbank kvs_get on a missing key: the
 *     reference panics (tatp/udp/kvs.h:91,152; smallbank/udp/kvs.h:67); here the
 *     ack is still produced, the table is untouched, and the event is counted;
 *   - populate: value structs are zero-initialised before the assigned fields are
 *     written (the reference copies partially initialised stack structs,
 *     tatp/udp/tatp.h:291-307).
 *
 * PARITY PINNED.  The six udp/ servers: the .npz fixtures, long_traces.json and fasst_24m.json under tests/golden were
 * recorded from the UNMODIFIED reference servers (oracle/Makefile ref).  The eBPF flavour -- STORE INSERT
 * (store/ebpf/store_kern.c:226-297), tatp REJECT_LOCK_SAME_KEY (orc_tatp_same_key_mode, tatp/ebpf/lock_kern.c:289-298),
 * smallbank WARMUP_READ (smallbank/ebpf/shard_user.c:179-186), the signed counters of the eBPF lock tables: the
 * UNMODIFIED *_kern.c / *_user.c compiled against a stub of the BPF helpers and run by oracle/ref_harness/ebpf/emu_main.c
 * (oracle/Makefile ref_ebpf; fixtures tests/golden/ebpf_*.npz).  The transaction handlers: the UNMODIFIED
 * caladan/client_udp_shard.cc against a synchronous stand-in for the runtime (oracle/Makefile ref_client;
 * tests/golden/clients.npz).
 */
#include "dint_oracle.h"

#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------- */
/* fasthash64: lock_fasst/udp/utils.h:16-53 (identical copy in every utils.h) */
static inline uint64_t fh_mix(uint64_t h) {
  h ^= h >> 23;
  h *= 0x2127599bf4325c37ULL;
  h ^= h >> 47;
  return h;
}

uint64_t orc_fasthash64(const void *buf, uint64_t len, uint64_t seed) {
  const uint64_t m = 0x880355f21e6d1965ULL;
  const unsigned char *p = (const unsigned char *)buf;
  uint64_t h = seed ^ (len * m);
  uint64_t nblk = len / 8;
  for (uint64_t i = 0; i < nblk; i++) {
    uint64_t v;
    memcpy(&v, p + 8 * i, 8);
    h ^= fh_mix(v);
    h *= m;
  }
  p += 8 * nblk;
  uint64_t rem = len & 7;
  if (rem) {
    uint64_t v = 0;
    for (uint64_t i = 0; i < rem; i++) v ^= (uint64_t)p[i] << (8 * i);
    h ^= fh_mix(v);
    h *= m;
  }
  return fh_mix(h);
}

/* fastrand LCG: tatp/udp/tatp.h:32-35 */
uint32_t orc_fastrand(uint64_t *seed) {
  *seed = *seed * 1103515245ULL + 12345ULL;
  return (uint32_t)(*seed >> 32);
}

static inline uint64_t hash_lid(uint32_t lid) { return orc_fasthash64(&lid, 4, 0xdeadbeef); }
static inline uint64_t hash_key(uint64_t key) { return orc_fasthash64(&key, 8, 0xdeadbeef); }

static inline uint32_t ld32(const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return v; }
static inline uint64_t ld64(const uint8_t *p) { uint64_t v; memcpy(&v, p, 8); return v; }
static inline void st32(uint8_t *p, uint32_t v) { memcpy(p, &v, 4); }

/* ------------------------------------------------------------------------- */
/* lock_fasst: lock_fasst/udp/server.cc:78-119                                */
struct orc_fasst { uint32_t n; uint32_t *locks; uint32_t *vers; };

orc_fasst *orc_fasst_create(uint32_t nslots) {
  orc_fasst *s = (orc_fasst *)calloc(1, sizeof(*s));
  s->n = nslots;
  s->locks = (uint32_t *)calloc(nslots, 4);
  s->vers = (uint32_t *)calloc(nslots, 4);
  return s;
}
void orc_fasst_destroy(orc_fasst *s) { if (s) { free(s->locks); free(s->vers); free(s); } }
uint32_t *orc_fasst_locks(orc_fasst *s) { return s->locks; }
uint32_t *orc_fasst_vers(orc_fasst *s) { return s->vers; }

uint64_t orc_fasst_replay(orc_fasst *s, void *msgs, size_t n) {
  uint8_t *m = (uint8_t *)msgs;
  uint64_t bad = 0;
  for (size_t i = 0; i < n; i++, m += 9) {
    uint32_t lid = ld32(m + 1);
    uint32_t slot = (uint32_t)(hash_lid(lid) % (uint64_t)s->n); /* server.cc:81-82 */
    switch (m[0]) {
      case 0: /* kRead -> kGrantRead, ver = ver_table[slot]   server.cc:86-90 */
        m[0] = 4;
        st32(m + 5, s->vers[slot]);
        break;
      case 1: /* kAcquireLock: CAS 0->1                        server.cc:92-101 */
        if (s->locks[slot] == 0) { s->locks[slot] = 1; m[0] = 5; }
        else m[0] = 6;
        break;
      case 2: /* kAbort: CAS 1->0                              server.cc:103-107 */
        s->locks[slot] = 0;
        m[0] = 7;
        break;
      case 3: /* kCommit: ver++, CAS 1->0 (no ownership check) server.cc:109-114 */
        s->vers[slot]++;
        s->locks[slot] = 0;
        m[0] = 8;
        break;
      default:
        bad++;
    }
  }
  return bad;
}

/* ------------------------------------------------------------------------- */
/* lock_2pl: lock_2pl/udp/server.cc:70-122 (serial: spin lock never contended) */
struct orc_2pl { uint32_t n; uint32_t *num_ex; uint32_t *num_sh; };

orc_2pl *orc_2pl_create(uint32_t nslots) {
  orc_2pl *s = (orc_2pl *)calloc(1, sizeof(*s));
  s->n = nslots;
  s->num_ex = (uint32_t *)calloc(nslots, 4);
  s->num_sh = (uint32_t *)calloc(nslots, 4);
  return s;
}
void orc_2pl_destroy(orc_2pl *s) { if (s) { free(s->num_ex); free(s->num_sh); free(s); } }
uint32_t *orc_2pl_num_ex(orc_2pl *s) { return s->num_ex; }
uint32_t *orc_2pl_num_sh(orc_2pl *s) { return s->num_sh; }

uint64_t orc_2pl_replay(orc_2pl *s, void *msgs, size_t n) {
  uint8_t *m = (uint8_t *)msgs;
  uint64_t bad = 0;
  for (size_t i = 0; i < n; i++, m += 6) {
    uint32_t lid = ld32(m + 1);
    uint32_t slot = (uint32_t)(hash_lid(lid) % (uint64_t)s->n);
    uint8_t action = m[0], type = m[5];
    if (action == 0) { /* kAcquireLock  server.cc:83-112 */
      if (type == 0) { /* shared */
        if (s->num_ex[slot] == 0) { s->num_sh[slot]++; m[0] = 2; }
        else m[0] = 3;
      } else if (type == 1) { /* exclusive */
        if (s->num_ex[slot] == 0 && s->num_sh[slot] == 0) { s->num_ex[slot]++; m[0] = 2; }
        else m[0] = 3;
      } else bad++;
    } else if (action == 1) { /* kReleaseLock  server.cc:114-121 */
      if (type == 0) s->num_sh[slot]--;
      else if (type == 1) s->num_ex[slot]--;
      m[0] = 5; /* note: an unknown lock type still gets kReleaseAck in the reference */
    } else bad++;
  }
  return bad;
}

/* ------------------------------------------------------------------------- */
/* log ring.  Canonical record (64 B):                                        */
/*   {u64 key; u8 val[40]; u32 ver; u8 is_del; u8 table; u8 pad[10]}          */
typedef struct { uint32_t cap; uint32_t tail; uint8_t *ring; } logring;

static void logring_init(logring *l, uint32_t cap) {
  l->cap = cap; l->tail = 0;
  l->ring = (uint8_t *)calloc((size_t)cap, 64);
}
static inline void logring_append(logring *l, uint64_t key, const uint8_t *val,
                                  uint32_t val_size, uint32_t ver, uint8_t is_del,
                                  uint8_t table) {
  uint8_t *e = l->ring + (size_t)l->tail * 64;
  memcpy(e, &key, 8);
  if (val) memcpy(e + 8, val, val_size); /* DELETE_LOG skips the val copy */
  memcpy(e + 48, &ver, 4);
  e[52] = is_del;
  e[53] = table;
  l->tail = (l->tail + 1) % l->cap;
}

/* log_server: log_server/udp/server.cc:73-88 */
struct orc_log { logring l; };
orc_log *orc_log_create(uint32_t ring_entries) {
  orc_log *s = (orc_log *)calloc(1, sizeof(*s));
  logring_init(&s->l, ring_entries);
  return s;
}
void orc_log_destroy(orc_log *s) { if (s) { free(s->l.ring); free(s); } }
uint8_t *orc_log_ring(orc_log *s) { return s->l.ring; }
uint32_t orc_log_tail(orc_log *s) { return s->l.tail; }

uint64_t orc_log_replay(orc_log *s, void *msgs, size_t n) {
  uint8_t *m = (uint8_t *)msgs;
  uint64_t bad = 0;
  for (size_t i = 0; i < n; i++, m += 53) {
    if (m[0] != 0) { bad++; continue; } /* reference: panic("unknown operation") */
    logring_append(&s->l, ld64(m + 1), m + 9, 40, ld32(m + 49), 0, 0);
    m[0] = 1; /* kAck */
  }
  return bad;
}

/* ------------------------------------------------------------------------- */
/* chained 4-way kvs: store/udp/kvs.h:13-136, tatp/udp/kvs.h:31-153           */
typedef struct orc_ent {
  uint64_t key[4];
  uint32_t ver[4];
  uint8_t valid[4];
  struct orc_ent *next;
  uint8_t val[]; /* 4 * val_size */
} orc_ent;

struct orc_kvs {
  uint32_t hash_size, val_size;
  orc_ent **heads;
  orc_ent *free_list;
  uint8_t *arena; size_t arena_left;
  void **arena_blocks; size_t n_blocks, cap_blocks;
  uint64_t count;
};

static size_t ent_bytes(const orc_kvs *t) {
  size_t b = sizeof(orc_ent) + 4u * t->val_size;
  return (b + 15) & ~(size_t)15;
}
static orc_ent *ent_alloc(orc_kvs *t) {
  if (t->free_list) { orc_ent *e = t->free_list; t->free_list = e->next; return e; }
  size_t b = ent_bytes(t);
  if (t->arena_left < b) {
    size_t blk = (size_t)1 << 24;
    if (t->n_blocks == t->cap_blocks) {
      t->cap_blocks = t->cap_blocks ? 2 * t->cap_blocks : 64;
      t->arena_blocks = (void **)realloc(t->arena_blocks, t->cap_blocks * sizeof(void *));
    }
    t->arena = (uint8_t *)malloc(blk);
    t->arena_blocks[t->n_blocks++] = t->arena;
    t->arena_left = blk;
  }
  orc_ent *e = (orc_ent *)t->arena;
  t->arena += b; t->arena_left -= b;
  return e;
}
static void ent_free(orc_kvs *t, orc_ent *e) { e->next = t->free_list; t->free_list = e; }

orc_kvs *orc_kvs_create(uint32_t hash_size, uint32_t val_size) {
  orc_kvs *t = (orc_kvs *)calloc(1, sizeof(*t));
  t->hash_size = hash_size; t->val_size = val_size;
  t->heads = (orc_ent **)calloc(hash_size, sizeof(orc_ent *));
  return t;
}
void orc_kvs_destroy(orc_kvs *t) {
  if (!t) return;
  for (size_t i = 0; i < t->n_blocks; i++) free(t->arena_blocks[i]);
  free(t->arena_blocks); free(t->heads); free(t);
}
uint64_t orc_kvs_count(orc_kvs *t) { return t->count; }

static inline uint32_t kvs_bucket(const orc_kvs *t, uint64_t key) { /* kvs.h:33-35 */
  return (uint32_t)(hash_key(key) % (uint64_t)t->hash_size);
}

int orc_kvs_get(orc_kvs *t, uint64_t key, uint8_t *val, uint32_t *ver) { /* kvs.h:37-55 */
  for (orc_ent *e = t->heads[kvs_bucket(t, key)]; e; e = e->next)
    for (int i = 0; i < 4; i++)
      if (e->key[i] == key && e->valid[i]) {
        memcpy(val, e->val + (size_t)i * t->val_size, t->val_size);
        *ver = e->ver[i];
        return 0;
      }
  return 1;
}

int orc_kvs_set(orc_kvs *t, uint64_t key, const uint8_t *val) { /* kvs.h:57-75 */
  for (orc_ent *e = t->heads[kvs_bucket(t, key)]; e; e = e->next)
    for (int i = 0; i < 4; i++)
      if (e->key[i] == key && e->valid[i]) {
        memcpy(e->val + (size_t)i * t->val_size, val, t->val_size);
        e->ver[i]++;
        return 0;
      }
  return 1;
}

static void kvs_insert_ver(orc_kvs *t, uint64_t key, const uint8_t *val, uint32_t ver) {
  uint32_t b = kvs_bucket(t, key);
  t->count++;
  for (orc_ent *e = t->heads[b]; e; e = e->next) /* first invalid slot in chain order */
    for (int i = 0; i < 4; i++)
      if (!e->valid[i]) {
        e->key[i] = key;
        memcpy(e->val + (size_t)i * t->val_size, val, t->val_size);
        e->ver[i] = ver;
        e->valid[i] = 1;
        return;
      }
  orc_ent *e = ent_alloc(t); /* else prepend a new entry: kvs.h:95-103 */
  memset(e, 0, ent_bytes(t));
  e->key[0] = key;
  memcpy(e->val, val, t->val_size);
  e->ver[0] = ver;
  e->valid[0] = 1;
  e->next = t->heads[b];
  t->heads[b] = e;
}

void orc_kvs_insert(orc_kvs *t, uint64_t key, const uint8_t *val) { /* kvs.h:77-104 */
  kvs_insert_ver(t, key, val, 0);
}

int orc_kvs_delete(orc_kvs *t, uint64_t key) { /* kvs.h:106-136 */
  uint32_t b = kvs_bucket(t, key);
  orc_ent *prev = NULL;
  for (orc_ent *e = t->heads[b]; e; prev = e, e = e->next)
    for (int i = 0; i < 4; i++)
      if (e->key[i] == key && e->valid[i]) {
        e->valid[i] = 0;
        t->count--;
        if (!e->valid[0] && !e->valid[1] && !e->valid[2] && !e->valid[3]) {
          if (prev) prev->next = e->next; else t->heads[b] = e->next;
          ent_free(t, e);
        }
        return 0;
      }
  return 1; /* reference: panic("kvs_delete: key not found") */
}

uint64_t orc_kvs_dump(orc_kvs *t, uint64_t *keys, uint32_t *vers, uint8_t *vals, uint64_t cap) {
  uint64_t n = 0;
  for (uint32_t b = 0; b < t->hash_size; b++)
    for (orc_ent *e = t->heads[b]; e; e = e->next)
      for (int i = 0; i < 4; i++)
        if (e->valid[i]) {
          if (n < cap) {
            keys[n] = e->key[i];
            vers[n] = e->ver[i];
            memcpy(vals + n * t->val_size, e->val + (size_t)i * t->val_size, t->val_size);
          }
          n++;
        }
  return n;
}

void orc_kvs_load(orc_kvs *t, const uint64_t *keys, const uint32_t *vers,
                  const uint8_t *vals, uint64_t n) {
  for (uint64_t i = 0; i < n; i++)
    kvs_insert_ver(t, keys[i], vals + i * t->val_size, vers ? vers[i] : 0);
}

/* ------------------------------------------------------------------------- */
/* store: store/udp/server.cc:75-97, populate store/udp/tatp.h:44-66          */
struct orc_store { orc_kvs *t; };

orc_store *orc_store_create(uint32_t hash_size, uint32_t populate_n) {
  orc_store *s = (orc_store *)calloc(1, sizeof(*s));
  s->t = orc_kvs_create(hash_size, 40);
  {
    uint64_t seed = 0xdeadbeef;
    for (uint32_t s_id = 0; s_id < populate_n; s_id++)
      for (uint32_t sf = 1; sf <= 4; sf++)
        for (uint32_t st = 0; st <= 16; st += 8) {
          uint64_t key = (uint64_t)s_id | ((uint64_t)sf << 32) | ((uint64_t)st << 40);
          uint8_t val[40] = {0};
          val[0] = (uint8_t)((orc_fastrand(&seed) % 24) + 1);
          val[1] = 0x5a;
          orc_kvs_insert(s->t, key, val);
        }
  }
  return s;
}
void orc_store_destroy(orc_store *s) { if (s) { orc_kvs_destroy(s->t); free(s); } }
orc_kvs *orc_store_table(orc_store *s) { return s->t; }

uint64_t orc_store_replay(orc_store *s, void *msgs, size_t n) {
  uint8_t *m = (uint8_t *)msgs;
  uint64_t bad = 0;
  for (size_t i = 0; i < n; i++, m += 53) {
    uint64_t key = ld64(m + 1);
    if (m[0] == 0) { /* kRead  server.cc:77-82 */
      uint32_t ver;
      if (orc_kvs_get(s->t, key, m + 9, &ver) == 0) { st32(m + 49, ver); m[0] = 3; }
      else m[0] = 7;
    } else if (m[0] == 1) { /* kSet  server.cc:84-89 */
      m[0] = (orc_kvs_set(s->t, key, m + 9) == 0) ? 5 : 7;
    } else if (m[0] == 2) { /* kInsert: udp panics (server.cc:94-95); eBPF store
                               inserts and acks (store/ebpf/store_kern.c:226-297) */
      orc_kvs_insert(s->t, key, m + 9);
      m[0] = 8;
    } else bad++;
  }
  return bad;
}

/* ------------------------------------------------------------------------- */
/* tatp: tatp/udp/server_shard.cc:113-210, populate tatp/udp/tatp.h:283-412   */
struct orc_tatp {
  orc_kvs *t[5];
  uint32_t hs[5];
  uint8_t *locks[5];
  uint64_t *owner[5]; /* eBPF lock_kern.c flavour only: key of the last granted lock (struct txn_lock, lock_kern.c:12-15) */
  logring l;
};

static uint64_t sid_to_sub_nbr(uint32_t s_id) { /* tatp.h:132-144, map_1000 :18-26 */
  uint64_t r = 0;
  for (int g = 0; g < 3; g++) {
    uint32_t i = s_id % 1000; s_id /= 1000;
    uint64_t m = ((uint64_t)((i / 100) % 10) << 8) | ((uint64_t)((i / 10) % 10) << 4) | (i % 10);
    r |= m << (12 * g);
  }
  return r;
}

/* tatp.h:254-281: returns count, values in out[] in selection order */
static int select_between(uint64_t *seed, int nvals, int N, int M, uint8_t *out) {
  int used[32] = {0};
  int to_select = (int)(orc_fastrand(seed) % (uint32_t)(M - N + 1)) + N;
  int cnt = 0;
  while (cnt < to_select) {
    int idx = (int)(orc_fastrand(seed) % (uint32_t)nvals);
    uint8_t v = (uint8_t)(idx + 1); /* values = {1,2,3,4} */
    if (used[v]) continue;
    used[v] = 1;
    out[cnt++] = v;
  }
  return cnt;
}

orc_tatp *orc_tatp_create(uint32_t n_sub, uint32_t log_entries, uint32_t populate_n) {
  orc_tatp *s = (orc_tatp *)calloc(1, sizeof(*s));
  uint64_t n = n_sub;
  s->hs[0] = s->hs[1] = (uint32_t)(n * 3 / 2 / 4);   /* server_shard.cc:75-76 */
  s->hs[2] = s->hs[3] = (uint32_t)(n * 15 / 4 / 4);  /* :77-78 */
  s->hs[4] = (uint32_t)(n * 45 / 8 / 4);             /* :79 */
  for (int i = 0; i < 5; i++) {
    if (s->hs[i] == 0) s->hs[i] = 1;
    s->t[i] = orc_kvs_create(s->hs[i], 40);
    s->locks[i] = (uint8_t *)calloc((size_t)4 * s->hs[i], 1);
  }
  logring_init(&s->l, log_entries);
  if (!populate_n) return s;
  /* rows are generated in s_id order from per-table seeded streams, so the rows of the
   * first populate_n subscribers do not depend on n_sub (which only sizes the tables) */
  n_sub = populate_n;

  uint64_t seed = 0xdeadbeef; /* subscriber: tatp.h:283-309 */
  for (uint32_t s_id = 0; s_id < n_sub; s_id++) {
    uint8_t v[40] = {0};
    uint64_t nbr = sid_to_sub_nbr(s_id);
    memcpy(v, &nbr, 8);
    for (int i = 0; i < 5; i++) v[15 + i] = (uint8_t)orc_fastrand(&seed);
    for (int i = 0; i < 10; i++) v[20 + i] = (uint8_t)orc_fastrand(&seed);
    uint16_t bits = (uint16_t)orc_fastrand(&seed);
    memcpy(v + 30, &bits, 2);
    uint32_t msc = 97; memcpy(v + 32, &msc, 4);
    uint32_t vlr = orc_fastrand(&seed); memcpy(v + 36, &vlr, 4);
    orc_kvs_insert(s->t[0], (uint64_t)s_id, v);
  }
  for (uint32_t s_id = 0; s_id < n_sub; s_id++) { /* second subscriber: tatp.h:312-327 */
    uint8_t v[40] = {0};
    memcpy(v, &s_id, 4);
    v[4] = 98;
    orc_kvs_insert(s->t[1], sid_to_sub_nbr(s_id), v);
  }
  seed = 0xdeadbeef; /* access info: tatp.h:330-354 */
  for (uint32_t s_id = 0; s_id < n_sub; s_id++) {
    uint8_t sel[4];
    int c = select_between(&seed, 4, 1, 4, sel);
    for (int k = 0; k < c; k++) {
      uint8_t v[40] = {0};
      v[0] = 99;
      orc_kvs_insert(s->t[2], (uint64_t)s_id | ((uint64_t)sel[k] << 32), v);
    }
  }
  seed = 0xdeadbeef; /* special facility + call forwarding: tatp.h:357-412 */
  for (uint32_t s_id = 0; s_id < n_sub; s_id++) {
    uint8_t sel[4];
    int c = select_between(&seed, 4, 1, 4, sel);
    for (int k = 0; k < c; k++) {
      uint8_t v[40] = {0};
      v[3] = 100;
      v[0] = (orc_fastrand(&seed) % 100 < 85) ? 1 : 0;
      orc_kvs_insert(s->t[3], (uint64_t)s_id | ((uint64_t)sel[k] << 32), v);
      for (uint32_t st = 0; st <= 16; st += 8) {
        if (orc_fastrand(&seed) % 2 == 0) continue;
        uint8_t w[40] = {0};
        w[1] = 101;
        w[0] = (uint8_t)((orc_fastrand(&seed) % 24) + 1);
        orc_kvs_insert(s->t[4],
                       (uint64_t)s_id | ((uint64_t)sel[k] << 32) | ((uint64_t)st << 40), w);
      }
    }
  }
  return s;
}

void orc_tatp_destroy(orc_tatp *s) {
  if (!s) return;
  for (int i = 0; i < 5; i++) { orc_kvs_destroy(s->t[i]); free(s->locks[i]); free(s->owner[i]); }
  free(s->l.ring); free(s);
}
orc_kvs *orc_tatp_table(orc_tatp *s, int table) { return s->t[table]; }
uint32_t orc_tatp_hash_size(orc_tatp *s, int table) { return s->hs[table]; }
uint8_t *orc_tatp_locks(orc_tatp *s, int table) { return s->locks[table]; }
void orc_tatp_same_key_mode(orc_tatp *s) {
  for (int i = 0; i < 5; i++)
    if (!s->owner[i]) s->owner[i] = (uint64_t *)calloc((size_t)4 * s->hs[i], 8);
}
uint8_t *orc_tatp_log_ring(orc_tatp *s) { return s->l.ring; }
uint32_t orc_tatp_log_tail(orc_tatp *s) { return s->l.tail; }

uint64_t orc_tatp_replay(orc_tatp *s, void *msgs, size_t n) {
  uint8_t *m = (uint8_t *)msgs;
  uint64_t bad = 0;
  for (size_t i = 0; i < n; i++, m += 55) {
    uint8_t type = m[1], tb = m[2];
    uint64_t key = ld64(m + 3);
    uint8_t *val = m + 11;
    if (tb >= 5) { bad++; continue; } /* reference: out-of-bounds table index (UB) */
    orc_kvs *t = s->t[tb];
    /* lock_hash: tatp.h:12-14 */
    uint8_t *lk = &s->locks[tb][hash_key(key) % ((uint64_t)4 * s->hs[tb])];
    switch (type) {
      case 0: { /* kRead  server_shard.cc:116-121 */
        uint32_t ver;
        if (orc_kvs_get(t, key, val, &ver) == 0) { st32(m + 51, ver); m[1] = 4; }
        else m[1] = 6;
        break;
      }
      case 1: /* kAcquireLock  :123-132;  eBPF ablation build tatp/ebpf/lock_kern.c:289-298: the slot remembers the
                 key it was granted to, and a rejected request for that same key is told so (REJECT_LOCK_SAME_KEY = 28) */
        if (*lk == 0) {
          *lk = 1; m[1] = 7;
          if (s->owner[tb]) s->owner[tb][lk - s->locks[tb]] = key;
        } else {
          m[1] = (s->owner[tb] && s->owner[tb][lk - s->locks[tb]] == key) ? 28 : 8;
        }
        break;
      case 2: /* kAbort  :134-138 */
        *lk = 0; m[1] = 9;
        break;
      case 12: /* kCommitPrim: set + unlock  :140-146 */
        if (orc_kvs_set(t, key, val)) bad++;
        *lk = 0; m[1] = 15;
        break;
      case 18: /* kInsertPrim  :148-154 */
        orc_kvs_insert(t, key, val);
        *lk = 0; m[1] = 20;
        break;
      case 22: /* kDeletePrim  :156-162 */
        if (orc_kvs_delete(t, key)) bad++;
        *lk = 0; m[1] = 25;
        break;
      case 13: /* kCommitBck  :164-168 */
        if (orc_kvs_set(t, key, val)) bad++;
        m[1] = 16;
        break;
      case 19: /* kInsertBck  :170-174 */
        orc_kvs_insert(t, key, val);
        m[1] = 21;
        break;
      case 23: /* kDeleteBck  :176-180 */
        if (orc_kvs_delete(t, key)) bad++;
        m[1] = 26;
        break;
      case 14: /* kCommitLog  :182-194 */
        logring_append(&s->l, key, val, 40, ld32(m + 51), 0, tb);
        m[1] = 17;
        break;
      case 24: /* kDeleteLog (no val copy)  :196-207 */
        logring_append(&s->l, key, NULL, 0, ld32(m + 51), 1, tb);
        m[1] = 27;
        break;
      default:
        bad++;
    }
  }
  return bad;
}

/* ------------------------------------------------------------------------- */
/* smallbank: smallbank/udp/server_shard.cc:107-189, populate smallbank.h:105-127 */
struct orc_sb {
  orc_kvs *t[2];
  uint32_t hs[2];
  uint32_t *num_ex[2], *num_sh[2];
  logring l;
};

orc_sb *orc_sb_create(uint32_t n_acct, uint32_t log_entries, uint32_t populate_n) {
  orc_sb *s = (orc_sb *)calloc(1, sizeof(*s));
  for (int i = 0; i < 2; i++) {
    s->hs[i] = (uint32_t)((uint64_t)n_acct * 3 / 2 / 4); /* server_shard.cc:75-76 */
    if (s->hs[i] == 0) s->hs[i] = 1;
    s->t[i] = orc_kvs_create(s->hs[i], 8);
    s->num_ex[i] = (uint32_t *)calloc((size_t)4 * s->hs[i], 4);
    s->num_sh[i] = (uint32_t *)calloc((size_t)4 * s->hs[i], 4);
  }
  logring_init(&s->l, log_entries);
  {
    float bal = 1000000000ull;
    for (uint32_t a = 0; a < populate_n; a++) {
      uint8_t v[8];
      uint32_t magic = 97;
      memcpy(v, &magic, 4); memcpy(v + 4, &bal, 4);
      orc_kvs_insert(s->t[0], a, v);
      magic = 98;
      memcpy(v, &magic, 4);
      orc_kvs_insert(s->t[1], a, v);
    }
  }
  return s;
}
void orc_sb_destroy(orc_sb *s) {
  if (!s) return;
  for (int i = 0; i < 2; i++) { orc_kvs_destroy(s->t[i]); free(s->num_ex[i]); free(s->num_sh[i]); }
  free(s->l.ring); free(s);
}
orc_kvs *orc_sb_table(orc_sb *s, int table) { return s->t[table]; }
uint32_t orc_sb_hash_size(orc_sb *s, int table) { return s->hs[table]; }
uint32_t *orc_sb_num_ex(orc_sb *s, int table) { return s->num_ex[table]; }
uint32_t *orc_sb_num_sh(orc_sb *s, int table) { return s->num_sh[table]; }
uint8_t *orc_sb_log_ring(orc_sb *s) { return s->l.ring; }
uint32_t orc_sb_log_tail(orc_sb *s) { return s->l.tail; }

uint64_t orc_sb_replay(orc_sb *s, void *msgs, size_t n) {
  uint8_t *m = (uint8_t *)msgs;
  uint64_t bad = 0;
  for (size_t i = 0; i < n; i++, m += 23) {
    uint8_t type = m[1], tb = m[2];
    uint64_t key = ld64(m + 3);
    uint8_t *val = m + 11;
    if (tb >= 2) { bad++; continue; }
    orc_kvs *t = s->t[tb];
    uint64_t lh = hash_key(key) % ((uint64_t)4 * s->hs[tb]); /* smallbank.h:12-14 */
    uint32_t *ex = &s->num_ex[tb][lh], *sh = &s->num_sh[tb][lh];
    uint32_t ver;
    switch (type) {
      case 0: /* kAcquireShared  server_shard.cc:121-133 */
        if (*ex == 0) {
          (*sh)++;
          if (orc_kvs_get(t, key, val, &ver) == 0) st32(m + 19, ver); else bad++;
          m[1] = 7;
        } else m[1] = 8;
        break;
      case 1: /* kAcquireExclusive  :135-147 */
        if (*ex == 0 && *sh == 0) {
          (*ex)++;
          if (orc_kvs_get(t, key, val, &ver) == 0) st32(m + 19, ver); else bad++;
          m[1] = 9;
        } else m[1] = 10;
        break;
      case 2: (*sh)--; m[1] = 11; break; /* kReleaseShared  :149-154 */
      case 3: (*ex)--; m[1] = 12; break; /* kReleaseExclusive  :156-161 */
      case 4: /* kCommitPrim  :163-167 */
        if (orc_kvs_set(t, key, val)) bad++;
        m[1] = 13;
        break;
      case 5: /* kCommitBck  :169-173 */
        if (orc_kvs_set(t, key, val)) bad++;
        m[1] = 14;
        break;
      case 6: /* kCommitLog  :175-186 */
        logring_append(&s->l, key, val, 8, ld32(m + 19), 0, tb);
        m[1] = 15;
        break;
      case 17: /* WARMUP_READ (eBPF flavour only; the udp server has no handler): a plain kvs_get answered
                  WARMUP_READ_ACK by the user-space store, smallbank/ebpf/shard_user.c:179-186 -- the return value of
                  kvs_get is ignored there, so a missing key echoes val / ver and still gets the ack */
        if (orc_kvs_get(t, key, val, &ver) == 0) st32(m + 19, ver);
        m[1] = 18;
        break;
      default:
        bad++;
    }
  }
  return bad;
}
