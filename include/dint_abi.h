/*
 * dint_abi.h -- C ABI of the MI355X batched transaction-certification engine.
 *
 * This is the drop-in boundary for ONE path of DINT: the per-packet server state
 * machine (lock_fasst / lock_2pl lock tables, store KV, log append, TATP and
 * SmallBank shard servers).  The reference has no function API for this path --
 * its boundary is the UDP wire protocol (one packed request struct per datagram,
 * the reply is the same struct mutated in place) -- so the ABI is the *batched form
 * of the reference's handler loop*:
 *
 *   reference (one message at a time)                      this ABI (N messages)
 *   -----------------------------------------------------  ----------------------------
 *   lock_fasst/udp/server.cc:78-119   server_loop body     dint_submit(workload FASST)
 *   lock_2pl/udp/server.cc:70-122     server_loop body     dint_submit(workload 2PL)
 *   log_server/udp/server.cc:73-88    server_handler body  dint_submit(workload LOG)
 *   store/udp/server.cc:72-98         server_handler body  dint_submit(workload STORE)
 *   tatp/udp/server_shard.cc:113-210  server_handler body  dint_submit(workload TATP)
 *   smallbank/udp/server_shard.cc:107-189                  dint_submit(workload SMALLBANK)
 *   (eBPF flavour: lock_fasst/ebpf/ls_kern.c:32-100 ls_xdp_main and siblings)
 *
 * Contract (SURVEY.md 8 "parity target"): for a request array R[0..n) the reply
 * array and the final table/log state equal those of the reference udp/ server
 * processing R in index order on one thread.  reqs/replies are arrays of the exact
 * `#pragma pack(1)` wire structs (9 / 6 / 53 / 53 / 55 / 23 bytes, see dint_msg_size);
 * replies[i] answers reqs[i]; every byte the reference does not overwrite is echoed.
 * Per-request errors never abort a batch: an unknown type/table leaves the reply
 * equal to the request and is counted in dint_stats (the reference would panic()).
 *
 * No torch/HIP types appear in any signature: device pointers and the stream are
 * passed as void*.  All functions return 0 or a negative DINT_E* code.
 */
#ifndef DINT_ABI_H
#define DINT_ABI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DINT_ABI_VERSION 4 /* v4 (round 6): dint_submit_device_ahead, dint_stats.late_requests */

/* dint_config.flags */
#define DINT_FLAG_KV_ROUNDS 1u /* kv workloads: resolve same-key conflicts request by request instead of in
                                  closed form (slow; identical results -- used for A/B runs and parity tests) */

#define DINT_FLAG_COPY_STREAMS 2u /* host path: dedicated H2D / D2H streams, so that one engine overlaps the copies of
                                     neighbouring passes with its kernels.  Off by default: a process that runs several
                                     engines has them overlap each other, and every extra stream beyond HIP's hardware
                                     queues (GPU_MAX_HW_QUEUES, 4) can serialise two engines behind one queue */

#define DINT_FLAG_LOCK_SAME_KEY 4u /* tatp: the eBPF ablation build tatp/ebpf/lock_kern.c -- a lock slot remembers the key it
                                     was granted to and a rejected ACQUIRE_LOCK for that same key is answered
                                     REJECT_LOCK_SAME_KEY (28) instead of REJECT_LOCK (:289-298; counted by
                                     tatp/caladan/client_lock.cc:762-771 to tell true conflicts from slot aliasing).
                                     Diagnostic: requests are resolved one by one (as DINT_FLAG_KV_ROUNDS) */

#define DINT_FLAG_KV_NO_HOT 8u /* kv workloads: big bins never take the dominant-key path (A/B runs; identical results) */

#define DINT_FLAG_INPUTS_READY 16u /* lock tables, dint_submit_device only.  The caller promises that (1) the request bytes of a
                                      batch are in device memory when the call is made -- not produced by work still
                                      pending on `stream` --, and (2) the request / reply buffers of a call do not overlap
                                      the buffers of EARLIER calls whose replies have not been consumed yet.  The engine
                                      then runs the first half of a pass (classify, hash, bin the requests: it touches
                                      no table) on a stream of its own beside the second half of the previous pass;
                                      replies still complete in the order of the calls on `stream`, and the table is
                                      still updated batch after batch.  A receive ring that is filled by a copy engine
                                      and handed over when complete -- what the reference's recv loop does with its
                                      socket buffer (lock_fasst/udp/net.h:33-48) -- satisfies both.  Without the flag a
                                      call is fully stream-ordered (the default).  Memory: the first piped call allocates two
                                      more sets of pass scratch, ~0.5 GB of address space each (touched as filled).  The kv
                                      workloads' form of the promise is per call: dint_submit_device_ahead */

/* workloads (dint_config.workload) */
enum {
  DINT_WL_FASST = 0,     /* lock_fasst: 9-byte {u8 type; u32 lid; u32 ver}            net.h:23-29 */
  DINT_WL_2PL = 1,       /* lock_2pl:   6-byte {u8 action; u32 lid; u8 type}          net.h:25-31 */
  DINT_WL_LOG = 2,       /* log_server: 53-byte {u8 type; u64 key; u8 val[40]; u32 ver}           */
  DINT_WL_STORE = 3,     /* store:      53-byte, same layout.  READ / SET are store/udp/server.cc:75-97; INSERT
                            (type 2 -> 8) is an ENGINE EXTENSION with kvs_insert semantics and no reference parity
                            target: store/udp panics on it and store/ebpf's type 2 is a cache fill */
  DINT_WL_TATP = 4,      /* tatp:       55-byte {u8 ord,type,table; u64 key; u8 val[40]; u32 ver} */
  DINT_WL_SMALLBANK = 5, /* smallbank:  23-byte {u8 ord,type,table; u64 key; u8 val[8]; u32 ver};
                            the udp server's 7 request types + WARMUP_READ (17 -> 18) of the eBPF flavour */
  DINT_WL_COUNT = 6
};

/* error codes */
enum {
  DINT_OK = 0,
  DINT_EINVAL = -1,   /* bad argument / unsupported configuration */
  DINT_ENOMEM = -2,   /* host or device allocation failed */
  DINT_EHIP = -3,     /* a HIP runtime call failed (see dint_last_error) */
  DINT_ENODEV = -4,   /* no usable gfx950 device */
  DINT_ESTATE = -5    /* call not valid for this engine's workload */
};

/* The largest request count one kernel pass handles: dint_submit splits larger
 * arrays into consecutive passes (correct by the serial-order contract).  DINT_KV_PASS_MAX for every
 * workload (see max_pass), never more than the log ring holds (log_server, tatp, smallbank); DINT_MICRO_BATCH was the
 * log_server limit until r03 and remains as the batch size of BASELINE's micro configs. */
#define DINT_MICRO_BATCH 65536u
#define DINT_KV_PASS_MAX 1048576u

typedef struct dint_config {
  uint32_t abi_version;  /* DINT_ABI_VERSION */
  uint32_t workload;     /* DINT_WL_* */
  int32_t device;        /* HIP device ordinal; -1 = current device */
  uint32_t flags;        /* DINT_FLAG_* */
  /* FASST / 2PL: number of lock slots; slot = fasthash64(lid,4,0xdeadbeef) % n_slots.
   * 0 = the reference's 36,000,000 (lock_fasst/udp/utils.h:12). */
  uint64_t n_slots;
  /* STORE: subscribers (buckets = n_rows*18/4, store/udp/server.cc:112-114; 0 = 2,000,000)
   * TATP : subscribers (bucket counts per tatp/udp/server_shard.cc:75-79; 0 = 7,000,000)
   * SMALLBANK: accounts (buckets = n_rows*3/2/4, smallbank/udp/server_shard.cc:75-76; 0 = 24,000,000) */
  uint64_t n_rows;
  /* log ring entries (LOG/TATP/SMALLBANK); 0 = the reference's 1,000,000 */
  uint32_t log_entries;
  /* multi-GPU hash sharding: this engine owns global slots/buckets g
   * with g % shard_count == shard_index and stores them at g / shard_count.  Requests
   * handed to dint_submit must already be routed to their home shard (dint_home_shard).
   * shard_count 0 or 1 = unsharded. */
  uint32_t shard_index;
  uint32_t shard_count;
  /* requests per kernel pass; longer submissions run as several passes.  0 = the engine's default: 1,048,576 (LOG / TATP /
   * SMALLBANK never more than log_entries), FASST / 2PL 65,536 (BASELINE's batch size: the lock kernels' hot-slot path
   * covers passes up to that; a longer pass is legal -- ask for it here -- and slower per request on skewed streams). */
  uint32_t max_pass;
  /* STORE / TATP / SMALLBANK: overflow entries per table (a bucket whose 4 inline slots are taken chains 4-slot
   * entries from this pool; the reference `new`s them without bound, store/udp/kvs.h:95-102).  0 = local buckets / 4
   * + 4096 (twice what the reference population needs).  A full pool refuses INSERTs: see dint_stats.pool_exhausted. */
  uint32_t pool_entries;
  uint32_t reserved[3];
} dint_config;

typedef struct dint_stats {
  uint64_t batches;        /* micro-batches processed */
  uint64_t requests;       /* requests processed */
  uint64_t bad_requests;   /* unknown type / table: reply left equal to request */
  uint64_t missing_keys;   /* SET/COMMIT/DELETE (tatp) or lock+read (smallbank) on a missing key:
                              the reference panics (tatp/udp/kvs.h:91,152); ack still sent */
  uint64_t foreign_requests; /* request whose home shard is not this engine: left untouched */
  uint64_t pool_exhausted; /* INSERTs that found the overflow-entry pool full: nothing is stored; a request resolved on
                              its own is answered REJECT_INSERT (store, 9) / REJECT_COMMIT (tatp, 11 -- the eBPF
                              flavour's "refused, send again", tatp/ebpf/shard_kern.c:509-514), one folded into a
                              same-key closed form keeps its ack; dint_wait / dint_submit return DINT_ENOMEM */
  uint64_t route_overflow; /* requests dint_route_pack could not place (destination slot full): answered by
                              dint_route_unpack with the back-pressure reply of dint_refuse ("not now, send again") */
  uint64_t big_bin_requests; /* kv workloads: requests that were resolved by the big-bin kernel (hot keys) */
  uint64_t late_requests;  /* store / tatp (ABI v4): requests of hot subs that no closed form of k_kv_hot covered and the general
                              path answered (k_kv_late) -- a diagnostic: they are the slow ones.  smallbank: requests of subs whose
                              row the pieces refused (kv_sb_item: more than 512 requests or more than 8 foreign ones in a piece,
                              an unknown op) and one workgroup answered the old way */
  uint64_t reserved[3];    /* [0..2] (v4): the late work items by kind -- store / tatp {sub as listed, solo, pieces}; smallbank: why a
                              row was refused {a piece over 512 requests, over 8 foreign requests, an unknown op} */
} dint_stats;

typedef struct dint_engine dint_engine_t;

/* ---- lifecycle ---------------------------------------------------------- */
int dint_engine_create(const dint_config *cfg, dint_engine_t **out);
void dint_engine_destroy(dint_engine_t *e);
/* bytes of one wire message of `workload` (9/6/53/53/55/23), or DINT_EINVAL */
int dint_msg_size(uint32_t workload);
/* human-readable text of the last failure on this thread */
const char *dint_last_error(void);

/* ---- the hot path -------------------------------------------------------- */
/* Host buffers: copies reqs to the GPU, runs the batch, copies replies back;
 * returns when replies are complete.  reqs == replies (in place) is allowed.
 * = dint_submit_async + dint_wait. */
int dint_submit(dint_engine_t *e, const void *reqs, uint32_t n, void *replies);
/* Pipelined form: enqueue and return at once; *ticket identifies the submission.  The array is cut
 * into passes of max_pass requests; pass k+1's host-to-device copy and pass k-1's device-to-host copy overlap pass
 * k's kernels (DINT_FLAG_COPY_STREAMS: three HIP streams; three staging slots), and successive submissions pipeline the same way -- as long
 * as reqs / replies are page-locked (dint_alloc_pinned, or the caller's own hipHostMalloc / hipHostRegister
 * memory).  Pageable buffers (plain malloc memory, as the reference's stack `message` of lock_fasst/udp/net.h:33-48) work
 * too: the engine stages them through page-locked buffers of its own -- one host memcpy on submission, and the replies
 * reach the caller's buffer inside dint_wait -- so the HIP runtime is never handed pageable memory.  Buffers must stay
 * untouched until dint_wait(ticket) returns; replies are valid only then.  Submissions are applied in call order (one
 * serial history per engine). */
typedef uint64_t dint_ticket;
int dint_submit_async(dint_engine_t *e, const void *reqs, uint32_t n, void *replies, dint_ticket *ticket);
/* replies of `ticket` (and of every earlier ticket) are complete.  DINT_ENOMEM if INSERTs were refused since the
 * last check (dint_stats.pool_exhausted) -- the replies are valid, rows were not stored. */
int dint_wait(dint_engine_t *e, dint_ticket ticket);
int dint_alloc_pinned(size_t bytes, void **out);
void dint_free_pinned(void *p);
/* Device buffers (HBM-resident, e.g. torch tensors): enqueues the batch on `stream`
 * (a hipStream_t, NULL = the engine's own stream) and returns immediately; in-place
 * allowed.  Buffers must stay valid until the stream reaches the end of the batch.
 * Stream rule: an engine has ONE set of batch scratch, so its passes run one after the other.  When consecutive
 * calls (submit, load, populate, route) name different streams the engine inserts the event wait itself; work the
 * CALLER enqueues on other streams (producing d_reqs, consuming d_replies) is the caller's to order --
 * dint_stream_wait / dint_stream_signal do that for the engine's own stream. */
int dint_submit_device(dint_engine_t *e, const void *d_reqs, uint32_t n, void *d_replies, void *stream);
/* The same with a LOOK-AHEAD (round 6; store / tatp / lock_fasst / lock_2pl -- smallbank and log_server ignore the announcement): (d_next_reqs, next_n,
 * d_next_replies) is the batch of the engine's NEXT dint_submit_device[_ahead] call.  The caller promises that (1) that call
 * will be made, with exactly these pointers and this count, before anything else is submitted to the engine, (2) the
 * batch's request bytes are complete in device memory in the order of `stream` -- produced by work enqueued on `stream`
 * before THIS call, or simply there: a receive ring the NIC has filled, as the reference's recv loop finds a whole
 * `message` in its socket buffer before the switch runs (tatp/udp/server_shard.cc:110-114) --, and (3) its buffers do not
 * overlap this call's.  The engine then runs the next batch's table-free first stage (classify, hash, partition into the
 * pass's coarse bins, and the log appends of its COMMIT_LOG / DELETE_LOG requests) in the SAME kernel launch as this batch's
 * hot keys (k_kv_hot_part) and starts the next call at its resolve kernel: a pass's chain is two launches instead of four.
 * Replies, table state and log ring are exactly those of two plain calls; between the two calls dint_read_log /
 * dint_log_drain already show the announced batch's log records.  A different next submission (or dint_snapshot) fails
 * with DINT_ESTATE after the engine has cleaned its scratch -- the announced batch's log records stay where they are, and
 * its reply buffer (= its request buffer, when in place) holds what the first stage wrote: undefined for the caller;
 * dint_reset / dint_restore drop the announcement silently.  lock_fasst / lock_2pl (batches of at most 65,536): the count
 * stage of the next batch rides in this batch's resolve launch the same way (k_lock_pass).  A dint_submit_device of more requests than one pass takes looks
 * ahead from pass to pass by itself (stream order already has the whole array complete). */
int dint_submit_device_ahead(dint_engine_t *e, const void *d_reqs, uint32_t n, void *d_replies, const void *d_next_reqs,
                             uint32_t next_n, void *d_next_replies, void *stream);
/* wait for everything enqueued on the engine's own streams; replies of dint_submit_async calls from pageable memory that are
 * still staged in the engine's page-locked buffers reach the caller's buffers here too (as in dint_wait) */
int dint_sync(dint_engine_t *e);
/* the engine's own stream (a hipStream_t) */
void *dint_engine_stream(dint_engine_t *e);
/* requests one kernel pass takes (dint_config.max_pass after clamping) */
uint32_t dint_max_pass(dint_engine_t *e);
/* the engine's own stream waits for everything enqueued so far on other_stream / other_stream waits for the engine */
int dint_stream_wait(dint_engine_t *e, void *other_stream);
int dint_stream_signal(dint_engine_t *e, void *other_stream);

/* Back-pressure: the reply the reference's eBPF servers give when they cannot take a request right
 * now -- REJECT_READ / REJECT_LOCK / REJECT_COMMIT (tatp), RETRY (smallbank, lock_2pl), kReject* (store) -- which every
 * client answers by sending the request again (e.g. tatp/caladan/client_ebpf_shard.cc:434-442).  A front end that must
 * shed load (the UDP shim with its submission queue full) fills replies with this instead of queueing; no GPU work,
 * no state change.  Request types that the eBPF servers never refuse are left as they are (= not answered). */
int dint_refuse(uint32_t workload, const void *reqs, uint32_t n, void *replies);

/* ---- population / state (parity + checkpointing) -------------------------- */
/* KV workloads: bulk-insert rows in order with kvs_insert semantics (ver given, or 0 if
 * vers == NULL).  val_size is 40 (store/tatp) or 8 (smallbank).  Host pointers. */
int dint_load_rows(dint_engine_t *e, uint32_t table, const uint64_t *keys, const uint32_t *vers,
                   const void *vals, uint64_t n);
/* generate and load the reference's initial population for the first `populate_n`
 * subscribers/accounts (store/udp/tatp.h:44-66, tatp/udp/tatp.h:283-412,
 * smallbank/udp/smallbank.h:105-127); value structs are zero-initialised first. */
int dint_populate(dint_engine_t *e, uint64_t populate_n);
/* number of buckets of `table` (global, before sharding) */
int64_t dint_hash_size(dint_engine_t *e, uint32_t table);
/* dump all valid rows of `table` (bucket order, chain order inside a bucket);
 * returns the row count (also when it exceeds cap; only cap rows are written). */
int64_t dint_dump_rows(dint_engine_t *e, uint32_t table, uint64_t *keys, uint32_t *vers, void *vals,
                       uint64_t cap);
/* lock-word state, one u32 per LOCAL lock slot:
 *   FASST: a=lock (0/1), b=version        2PL/SMALLBANK: a=num_ex, b=num_sh
 *   TATP : a=txn lock (0/1), b=0 (b may be NULL)
 * FASST/2PL: n = ceil(n_slots / shard_count) words, index = global slot / shard_count.
 * TATP/SMALLBANK (`table` selects the table): n = 4 * local buckets, index = q * n_local + local bucket with
 * q = lock_hash / hash_size -- i.e. exactly lock_hash (tatp/udp/tatp.h:12-14) when unsharded.
 * Returns n (also when cap is smaller; only cap words are written). */
int64_t dint_read_locks(dint_engine_t *e, uint32_t table, uint32_t *a, uint32_t *b, uint64_t cap);
/* log ring: copies up to cap canonical 64-byte records
 * {u64 key; u8 val[40]; u32 ver; u8 is_del; u8 table; u8 pad[10]} and returns the tail index */
int64_t dint_read_log(dint_engine_t *e, void *records, uint64_t cap);
/* Log drain (the reference writes its logs and never reads them, tatp/udp/server_shard.cc:182-207): copies
 * the records appended since the previous call, oldest first, up to cap; returns how many.  *lost (may be NULL) = records
 * the ring overwrote before they were drained (drain at least once per log_entries appended).  The stream position
 * survives ring wrap-around; dint_reset rewinds it.  dint_amd/recovery.py replays a drained log into a replica. */
int64_t dint_log_drain(dint_engine_t *e, void *records, uint64_t cap, uint64_t *lost);
int dint_get_stats(dint_engine_t *e, dint_stats *out);
/* reset tables, locks, log and stats to the freshly-created (unpopulated) state */
int dint_reset(dint_engine_t *e);
/* snapshot / restore the whole engine state in HBM (used to replay a recorded trace) */
int dint_snapshot(dint_engine_t *e);
int dint_restore(dint_engine_t *e);

/* ---- multi-GPU routing -------------------------------------- */
/* One step on rank r of G (every engine created with shard_index = r, shard_count = G):
 *   dint_route_pack      stable partition of the ingested batch by home rank into G fixed-capacity slots
 *   all-to-all           of the slots (RCCL; equal split sizes, so no host round trip)
 *   dint_submit_segments the home engine answers what it received, in (source rank, index) order, in place
 *   all-to-all           back
 *   dint_route_unpack    replies to their original positions
 * A slot = seg_cap messages at d_send + w * seg_stride, with its live count (u32) at d_cnt + w * cnt_stride (the
 * caller decides where the header lives, e.g. in front of the slot so that it travels with it).  d_slot[n] (u32)
 * remembers where each request went.  Requests beyond a slot's capacity are not sent: dint_route_unpack answers them
 * with the back-pressure reply of dint_refuse, which every client answers by sending the request again
 * (dint_stats.route_overflow counts them); n <= 1,048,576 and G <= 64 per call. */
int dint_route_pack(dint_engine_t *e, const void *d_reqs, uint32_t n, void *d_send, uint32_t seg_cap,
                    uint64_t seg_stride, void *d_cnt, uint64_t cnt_stride, uint32_t *d_slot, void *stream);
int dint_route_unpack(dint_engine_t *e, const void *d_back, uint32_t seg_cap, uint64_t seg_stride,
                      const uint32_t *d_slot, const void *d_reqs, uint32_t n, void *d_replies, void *stream);
/* The same for several batches at once -- the S logical servers of a rank (3 for tatp / smallbank), each routed with
 * its own engine's hash and modulus -- in ONE set of kernel launches (grid.y = batch): a step then costs 3 + 1 launches
 * on the exchange stream whatever S is.  d_slots = this batch's slot of peer 0 (peer w at + w * seg_stride) in the send
 * buffer (pack) or in the returned buffer (unpack); d_cnt = its live-count word of peer 0 (peer w at + w *
 * cnt_stride).  At most 4 batches per call, all on one device; callers pass the engines in the same order. */
typedef struct dint_route_item {
  dint_engine_t *engine;
  const void *d_reqs;
  uint32_t n;
  uint32_t seg_cap;
  void *d_slots;
  void *d_cnt;      /* pack only */
  uint32_t *d_slot; /* [n]: written by pack, read by unpack */
  void *d_replies;  /* unpack only */
  const uint32_t *d_n; /* ABI v3: the batch's live request count in DEVICE memory (a batch a kernel produced -- the
                          GPU-resident clients -- whose size the host never sees), n is then its upper bound; NULL: n
                          is the count */
} dint_route_item;
int dint_route_pack_multi(const dint_route_item *items, uint32_t n_items, uint64_t seg_stride, uint64_t cnt_stride,
                          void *stream);
int dint_route_unpack_multi(const dint_route_item *items, uint32_t n_items, uint64_t seg_stride, void *stream);
/* n_seg segments of seg_cap message slots, segment k at d_base + k * seg_stride holding *(u32 *)(d_cnt + k *
 * cnt_stride) requests: processed in place as ONE serial history, segment by segment (kernel passes take whole
 * segments: seg_cap <= max_pass). */
int dint_submit_segments(dint_engine_t *e, void *d_base, uint32_t n_seg, uint32_t seg_cap, uint64_t seg_stride,
                         const void *d_cnt, uint64_t cnt_stride, void *stream);
/* (round 3; the config layout and DINT_ABI_VERSION are unchanged) the segments of SEVERAL engines (the shard servers of one GPU: same workload -- store, tatp or smallbank --
 * same device) answered by ONE set of kernel launches on ONE stream, the engines' kernels side by side in one grid.  A
 * closed-loop epoch or an exchange step hands every server its batch at the same moment; with one stream per engine that
 * is a fork and a join across streams per step, with this call it is three launches on the caller's stream (NULL: the
 * first engine's).  Every engine appears at most once in `items` (DINT_EINVAL otherwise).  Each engine's history is what dint_submit_segments would have produced.  Batches that do not fit one
 * kernel pass (n_seg * seg_cap > max_pass), lock / log engines and more than 4 items fall back to one
 * dint_submit_segments per item on that stream. */
typedef struct dint_segments_item {
  dint_engine_t *engine;
  void *d_base;
  uint32_t n_seg, seg_cap;
  uint64_t seg_stride;
  const void *d_cnt;
  uint64_t cnt_stride;
} dint_segments_item;
int dint_submit_segments_multi(const dint_segments_item *items, uint32_t n_items, void *stream);
/* ... with a LOOK-AHEAD (round 6, as dint_submit_device_ahead): next[k] = the segments engine k will be handed by the NEXT call of
 * this kind -- the same engines in the same order, the same geometry (n_seg, seg_cap, strides), other buffers, complete in
 * device memory in the order of `stream` (the exchange has delivered them).  The engines' partitions of that step ride in this
 * step's launch set.  next == NULL, or an announcement that does not fit (other engines, another geometry, a workload without
 * the one-launch pass): plain dint_submit_segments_multi.  The announced step MUST be the engines' next submission. */
int dint_submit_segments_multi_ahead(const dint_segments_item *items, uint32_t n_items, const dint_segments_item *next, void *stream);
/* d_home[i] = home shard (0..shard_count-1) of d_reqs[i], computed on the GPU with the
 * same hash/modulus the engine uses; 0xFF for requests that have no home (bad table). */
int dint_home_shard(dint_engine_t *e, const void *d_reqs, uint32_t n, uint8_t *d_home, void *stream);

/* ---- measurement helpers -------------------------------------------------- */
/* Random 64-byte gather microbenchmark over `bytes` of HBM (roofline denominator,
 * SURVEY.md 8d): returns accesses per second via *out_aps, elapsed seconds via *out_s. */
int dint_bench_rand64(int32_t device, uint64_t bytes, uint64_t n_access, int write_back,
                      double *out_aps, double *out_s);
/* The general form: one access pattern over `bytes` of HBM, `blocks_per_cu` 256-thread workgroups per CU (0 = 8).
 * mode 0 = random gather of `width` (8 / 16 / 64) bytes from a 64-byte-aligned sector; 1 = the same + an 8-byte store
 * into the sector (read-modify-write); 2 = blind scatter of `width` (1 / 8 / 16 / 64) bytes; 3 = device-scope 64-bit
 * atomic add without / 4 = with a returned value; 5 / 6 = streaming read / write of the whole table in 16-byte
 * vectors (*out_aps = vectors per second).  Indices come from the engines' own hash + magic-multiply modulo. */
int dint_bench_access(int32_t device, uint64_t bytes, uint64_t n_access, uint32_t mode, uint32_t width,
                      uint32_t blocks_per_cu, double *out_aps, double *out_s);

/* Device primitives against their portable forms, on `device` (-1 = current): 0 = equal; bit 0 = the wave sort network
 * built from DPP / permlane exchanges differs from the ds_bpermute network, bit 1 = a single lane exchange differs;
 * negative = error.  (tests/test_gpu_locks.py; a diagnostic, not a hot-path call.) */
int dint_selftest(int32_t device);
/* Per-kernel launch time of the most recent dint_submit_device micro-batch sequence, measured
 * with HIP events on the stream the kernels ran on.  Call dint_timing_enable(e,1) first. */
int dint_timing_enable(dint_engine_t *e, int on);
/* returns number of kernels written; names[i] points to a static string */
int dint_timing_read(dint_engine_t *e, const char **names, double *avg_us, uint64_t *launches, int cap);

/* Debug (kv workloads, engine created with DINT_KV_TRACE=1 in the environment): per-wave timeline of the most
 * recent resolve launch, 16 u64 per bin: [0..9] s_memtime stamps (see k_kv.hip kv_stamp), [15] records in the bin.
 * Copies min(cap, 2048 * 16) words; returns the number of bins traced or DINT_ESTATE when tracing is off. */
int dint_kv_trace_read(dint_engine_t *e, uint64_t *out, uint64_t cap);

#ifdef __cplusplus
}
#endif
#endif /* DINT_ABI_H */
