/*
 * dint_oracle.h -- CPU restatement of the DINT per-packet server state machines.
 *
 * TEST INFRASTRUCTURE ONLY.  This is the parity oracle: a plain-C, single-thread,
 * serial restatement of the reference `udp/` servers with run-time table sizes.
 * Nothing in the product path (dint_amd/, include/, bench.py's GPU leg) may link,
 * import or call it; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg do, and only as the checker.
 *
 * Every function cites the reference file:line (relative to /root/reference) it
 * follows.  The restatement is pinned against the unmodified reference servers
 * compiled under oracle/_ref (see oracle/Makefile, oracle/ref_harness/) and the
 * known-answer vectors of SURVEY.md 8(c); tests/test_oracle_*.py hold the pins.
 *
 * Wire structs are the reference's `#pragma pack(1)` structs, byte for byte:
 *   fasst  9 B  {u8 type; u32 lid; u32 ver}              lock_fasst/udp/net.h:23-29
 *   2pl    6 B  {u8 action; u32 lid; u8 type}            lock_2pl/udp/net.h:25-31
 *   log   53 B  {u8 type; u64 key; u8 val[40]; u32 ver}  log_server/udp/net.h:23-30
 *   store 53 B  (same layout)                            store/udp/net.h:34-41
 *   tatp  55 B  {u8 ord,type,table; u64 key; u8 val[40]; u32 ver}  tatp/udp/net.h:57-66
 *   sb    23 B  {u8 ord,type,table; u64 key; u8 val[8];  u32 ver}  smallbank/udp/net.h:41-50
 */
#ifndef DINT_ORACLE_H
#define DINT_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- hashing / PRNG ------------------------------------------------------ */
uint64_t orc_fasthash64(const void *buf, uint64_t len, uint64_t seed);
uint32_t orc_fastrand(uint64_t *seed);

/* ---- lock_fasst ---------------------------------------------------------- */
typedef struct orc_fasst orc_fasst;
orc_fasst *orc_fasst_create(uint32_t nslots);
void orc_fasst_destroy(orc_fasst *s);
/* replay n packed 9-byte messages in place; returns number of unknown-type msgs */
uint64_t orc_fasst_replay(orc_fasst *s, void *msgs, size_t n);
uint32_t *orc_fasst_locks(orc_fasst *s);
uint32_t *orc_fasst_vers(orc_fasst *s);

/* ---- lock_2pl ------------------------------------------------------------ */
typedef struct orc_2pl orc_2pl;
orc_2pl *orc_2pl_create(uint32_t nslots);
void orc_2pl_destroy(orc_2pl *s);
uint64_t orc_2pl_replay(orc_2pl *s, void *msgs, size_t n);
uint32_t *orc_2pl_num_ex(orc_2pl *s);
uint32_t *orc_2pl_num_sh(orc_2pl *s);

/* ---- log_server ---------------------------------------------------------- */
typedef struct orc_log orc_log;
orc_log *orc_log_create(uint32_t ring_entries);
void orc_log_destroy(orc_log *s);
uint64_t orc_log_replay(orc_log *s, void *msgs, size_t n);
/* canonical 64-byte log record, common to all workloads:
 *   {u64 key; u8 val[40]; u32 ver; u8 is_del; u8 table; u8 pad[10]}        */
uint8_t *orc_log_ring(orc_log *s);
uint32_t orc_log_tail(orc_log *s);

/* ---- generic chained 4-way KV (store / tatp / smallbank tables) ---------- */
typedef struct orc_kvs orc_kvs;
orc_kvs *orc_kvs_create(uint32_t hash_size, uint32_t val_size);
void orc_kvs_destroy(orc_kvs *t);
int orc_kvs_get(orc_kvs *t, uint64_t key, uint8_t *val, uint32_t *ver);
int orc_kvs_set(orc_kvs *t, uint64_t key, const uint8_t *val);
void orc_kvs_insert(orc_kvs *t, uint64_t key, const uint8_t *val);
int orc_kvs_delete(orc_kvs *t, uint64_t key);
uint64_t orc_kvs_count(orc_kvs *t);
/* dump valid rows in chain order per bucket: keys[i], vers[i], vals[i*val_size..] */
uint64_t orc_kvs_dump(orc_kvs *t, uint64_t *keys, uint32_t *vers, uint8_t *vals,
                      uint64_t cap);
/* bulk insert rows (chain-order semantics = kvs_insert applied in row order) */
void orc_kvs_load(orc_kvs *t, const uint64_t *keys, const uint32_t *vers,
                  const uint8_t *vals, uint64_t n);

/* ---- store --------------------------------------------------------------- */
typedef struct orc_store orc_store;
/* hash_size buckets; rows of the first populate_n subscribers per store/udp/tatp.h:44-66 */
orc_store *orc_store_create(uint32_t hash_size, uint32_t populate_n);
void orc_store_destroy(orc_store *s);
uint64_t orc_store_replay(orc_store *s, void *msgs, size_t n);
orc_kvs *orc_store_table(orc_store *s);

/* ---- tatp shard server --------------------------------------------------- */
typedef struct orc_tatp orc_tatp;
/* n_sub sizes the tables (bucket counts per tatp/udp/server_shard.cc:75-79); the rows of
 * the first populate_n subscribers are generated per tatp/udp/tatp.h:283-412 with
 * zero-initialised value structs (populate_n == n_sub is the reference's population) */
orc_tatp *orc_tatp_create(uint32_t n_sub, uint32_t log_entries, uint32_t populate_n);
void orc_tatp_destroy(orc_tatp *s);
uint64_t orc_tatp_replay(orc_tatp *s, void *msgs, size_t n);
orc_kvs *orc_tatp_table(orc_tatp *s, int table);
uint32_t orc_tatp_hash_size(orc_tatp *s, int table);
uint8_t *orc_tatp_locks(orc_tatp *s, int table); /* 4*hash_size bytes */
/* switch to the semantics of the eBPF ablation build tatp/ebpf/lock_kern.c (REJECT_LOCK_SAME_KEY) */
void orc_tatp_same_key_mode(orc_tatp *s);
uint8_t *orc_tatp_log_ring(orc_tatp *s);
uint32_t orc_tatp_log_tail(orc_tatp *s);

/* ---- smallbank shard server ---------------------------------------------- */
typedef struct orc_sb orc_sb;
orc_sb *orc_sb_create(uint32_t n_acct, uint32_t log_entries, uint32_t populate_n);
void orc_sb_destroy(orc_sb *s);
uint64_t orc_sb_replay(orc_sb *s, void *msgs, size_t n);
orc_kvs *orc_sb_table(orc_sb *s, int table);
uint32_t orc_sb_hash_size(orc_sb *s, int table);
uint32_t *orc_sb_num_ex(orc_sb *s, int table); /* 4*hash_size words */
uint32_t *orc_sb_num_sh(orc_sb *s, int table);
uint8_t *orc_sb_log_ring(orc_sb *s);
uint32_t orc_sb_log_tail(orc_sb *s);

#ifdef __cplusplus
}
#endif
#endif
