/*
 * dint_driver.h -- C ABI of the closed-loop transaction drivers (the CALLER of the hot path).
 *
 * The reference's load generators are Caladan programs, one uthread per client, each running
 * complete OCC / 2PL transactions against three replicated shard servers
 * (tatp/caladan/client_udp_shard.cc:177-1185, smallbank/caladan/client_udp_shard.cc:169-1240).
 * They cannot be built here (DPDK / rdma-core / SPDK submodules are not vendored), and no NIC can
 * offer the > 100 M requests/s one MI355X absorbs, so the same transaction state machines are
 * restated here as a deterministic, epoch-synchronous driver:
 *
 *   - W virtual clients, client g seeded 0xdeadbeef + g exactly as ClientLoop does (:1122);
 *   - a transaction is a sequence of PHASES; the messages of one phase are sent together (the
 *     reference sends them from one uthread per shard and joins) and all replies are awaited;
 *   - one EPOCH = every client emits the messages of its current phase; the messages addressed to
 *     shard s (s = key % 3 for reads / locks / primary ops, (s+1)%3 and (s+2)%3 for backups, all three
 *     for logs -- client_udp_shard.cc:187,493-531) form batch s, ordered by client id then send order;
 *     the three shard servers answer; every client consumes its replies and moves on.
 *
 * The driver owns no sockets and no GPU state: the caller carries the batches to the servers
 * (dint_submit of three engines, or the CPU oracle in tests) and hands the replies back.
 */
#ifndef DINT_DRIVER_H
#define DINT_DRIVER_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DINT_N_SHARDS 3 /* the reference's deployment: 3 servers, 3-way replication */

typedef struct dint_driver dint_driver_t;

typedef struct dint_driver_config {
  uint32_t workload;      /* DINT_WL_TATP or DINT_WL_SMALLBANK */
  uint32_t n_clients;     /* W */
  uint64_t n_rows;        /* subscribers (tatp) / accounts (smallbank) the keys are drawn from */
  uint32_t first_client;  /* global id of client 0 (seed = 0xdeadbeef + first_client + i) */
  /* key distribution: 0 = the reference's own (tatp_nurand, tatp/caladan/tatp.h:40-43; smallbank hot/cold
   * picker, smallbank.h:30-50); 1 = Zipf(theta) over the rows (BASELINE.json's stress distribution) */
  uint32_t key_dist;
  double zipf_theta;
  uint32_t reserved[8];
} dint_driver_config;

typedef struct dint_driver_stats {
  uint64_t txns;          /* finished transactions (committed or not) = the reference's "throughput" count */
  uint64_t committed;     /* = "goodput" */
  uint64_t messages;      /* requests emitted */
  uint64_t by_type[8];    /* finished, per transaction type */
  uint64_t committed_by_type[8];
  uint64_t epochs;
} dint_driver_stats;

int dint_driver_create(const dint_driver_config *cfg, dint_driver_t **out);
void dint_driver_destroy(dint_driver_t *d);
/* wire message size of the driver's workload (55 / 23) */
int dint_driver_msg_size(const dint_driver_t *d);
/* Emit the next epoch.  counts[s] receives the number of messages for shard s; the messages stay in
 * driver-owned buffers returned by dint_driver_batch.  Must alternate with dint_driver_consume. */
int dint_driver_next(dint_driver_t *d, uint32_t counts[DINT_N_SHARDS]);
/* request batch of shard s for the current epoch (valid until the next dint_driver_next) */
const void *dint_driver_batch(dint_driver_t *d, uint32_t shard);
/* hand back the replies (same order, same count as the batches) and advance every client */
int dint_driver_consume(dint_driver_t *d, const void *const replies[DINT_N_SHARDS]);
int dint_driver_get_stats(const dint_driver_t *d, dint_driver_stats *out);

/* ---- the same drivers, resident on the GPU -----------------------------------------------
 * Client state lives in HBM; dint_gdriver_next launches a kernel in which every client runs one phase and emits its
 * messages into three driver-owned device arrays -- at exactly the positions, hence with exactly the bytes, of the
 * host driver (tests compare the two streams) -- and writes the three batch sizes to device memory;
 * the shard servers answer IN PLACE (dint_submit_segments(engine_s, dint_gdriver_batch(d, s), 1, cap, cap * msg,
 * counts + s, 0, stream): one segment whose live count the engine reads on the device); dint_gdriver_consume
 * makes every client pick up its replies -- in a kernel of its own, or, when it is issued on the stream of the last
 * dint_gdriver_next (the closed loop), fused into the next emit kernel: the epoch's batches alternate between two buffer
 * sets, so ask dint_gdriver_batch again after every dint_gdriver_next, and keep that stream alive until then.
 * Nothing crosses PCIe.  `cap_per_shard` = slots of each batch array; messages beyond it are dropped and counted
 * (overflow: size it ~1.3x the expected batch). */
typedef struct dint_gdriver dint_gdriver_t;
int dint_gdriver_create(const dint_driver_config *cfg, int32_t device, uint32_t cap_per_shard, dint_gdriver_t **out);
void dint_gdriver_destroy(dint_gdriver_t *g);
int dint_gdriver_next(dint_gdriver_t *g, void *stream);     /* must alternate with dint_gdriver_consume */
int dint_gdriver_consume(dint_gdriver_t *g, void *stream);
void *dint_gdriver_batch(dint_gdriver_t *g, uint32_t shard);  /* device pointer: the CURRENT epoch's cap_per_shard message slots */
const void *dint_gdriver_counts(dint_gdriver_t *g);           /* device pointer: uint32_t[3] live messages per shard */
uint32_t dint_gdriver_cap(const dint_gdriver_t *g);
/* tests / debugging: synchronise and copy the current batch of `shard` to the host; returns its message count */
int64_t dint_gdriver_read_batch(dint_gdriver_t *g, uint32_t shard, void *host, uint64_t cap_msgs);
int dint_gdriver_get_stats(dint_gdriver_t *g, dint_driver_stats *out, uint64_t *overflow);  /* synchronises the device */

/* ---- lock_fasst load generator ---------------------------------------------------------------------------
 * lock_fasst/caladan/client.cc:183-280 (ClientLoop) over transactions shaped like lock_fasst/caladan/trace_init.sh
 * :6-27, W workers in lock step, one outstanding request each.  dint_fasst_client_next returns the epoch's W 9-byte
 * requests (worker order), dint_fasst_client_consume takes the W replies.  Epochs laid end to end = the request trace
 * (the "24M-op trace": 4096 workers, 24,000,000 keys, read_pct 80, Zipf 0.8 or uniform). */
typedef struct dint_fasst_client dint_fasst_client_t;
typedef struct dint_fasst_client_config {
  uint32_t n_workers;     /* W virtual workers */
  uint32_t first_worker;  /* seed of worker i = 0xdeadbeef + first_worker + i */
  uint32_t key_space;     /* lids are drawn from [0, key_space) */
  uint32_t read_pct;      /* a key of a transaction is read-only with this probability (80) */
  uint32_t key_dist;      /* 0 = uniform (the reference's traces), 1 = Zipf(zipf_theta) */
  uint32_t reserved0;
  double zipf_theta;
  uint32_t reserved[8];
} dint_fasst_client_config;
typedef struct dint_fasst_client_stats {
  uint64_t requests, epochs;
  uint64_t committed;        /* transactions that reached COMMIT (or finished read-only) */
  uint64_t rejects;          /* REJECT_LOCK replies: abort what was locked, restart the transaction */
  uint64_t rollbacks;        /* validation failures: abort every write key, restart */
  uint64_t protocol_errors;  /* a reply the reference client would assert / panic on */
  uint64_t reserved[2];
} dint_fasst_client_stats;
int dint_fasst_client_create(const dint_fasst_client_config *cfg, dint_fasst_client_t **out);
void dint_fasst_client_destroy(dint_fasst_client_t *c);
const void *dint_fasst_client_next(dint_fasst_client_t *c);
int dint_fasst_client_consume(dint_fasst_client_t *c, const void *replies);
int dint_fasst_client_get_stats(const dint_fasst_client_t *c, dint_fasst_client_stats *out);
/* the transaction `worker` is running: keys[0 .. *n_keys) = its sorted read set, wkeys[0 .. *n_wkeys) = its write set (room
 * for 10 each) -- one transaction of lock_fasst/caladan/trace_init.sh's trace files */
int dint_fasst_client_peek(const dint_fasst_client_t *c, uint32_t worker, uint32_t *keys, uint32_t *n_keys, uint32_t *wkeys,
                           uint32_t *n_wkeys);

#ifdef __cplusplus
}
#endif
#endif /* DINT_DRIVER_H */
