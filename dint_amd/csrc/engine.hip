// engine.hip -- host side of the C ABI (include/dint_abi.h): owns the HBM tables, the batch
// scratch, the streams, and turns dint_submit* into kernel launches.  Product path: there is no
// CPU fallback -- every entry point fails with DINT_ENODEV / DINT_EHIP when no gfx950 device works.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/dint_abi.h"
#include "dint_kernels.h"
#include "dint_kv.h"
#include "dint_populate.h"

// k_locks.hip: one half of a lock pass (stage 1 = count + scan / place, 2 = resolve) on `st`
void dint_launch_lock_stage(uint32_t workload, int stage, const void *d_req, void *d_rep, uint32_t n, uint2 *table, dint_mod slots,
                            dint_shard shard, dint_scratch s, hipStream_t st, const dint_view &view);
// k_locks.hip: the resolve stage of a pass (set `s`) and the count stage of the next (set `sn`) in one launch; false = not possible
bool dint_launch_lock_fused(uint32_t workload, void *d_rep, uint32_t n, uint2 *table, dint_mod slots, dint_shard shard, dint_scratch s,
                            const dint_view &view, const void *next_req, void *next_rep, uint32_t next_n, dint_scratch sn,
                            const dint_view &next_view, hipStream_t st);

namespace {

thread_local std::string g_err;

int fail(int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

#define HIP_TRY(expr)                                                                      \
  do {                                                                                     \
    hipError_t _e = (expr);                                                                \
    if (_e != hipSuccess) return fail(DINT_EHIP, "%s: %s", #expr, hipGetErrorString(_e));  \
  } while (0)

const int kMsgSize[DINT_WL_COUNT] = {9, 6, 53, 53, 55, 23};

// kernel timing with HIP events on the stream the kernels run on
struct KernelTimer {
  static const int kMaxLaunch = 4096;
  bool on = false;
  int n_kernels = 0;             // kernels per pass (events per pass = n_kernels + 1)
  const char *names[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  std::vector<hipEvent_t> ev;    // (n_kernels + 1) per recorded pass
  int passes = 0;
};

}  // namespace

struct dint_engine {
  dint_config cfg;
  uint32_t pass_max = DINT_MICRO;  // requests per kernel pass (<= the log ring size when a log is attached)
  int device = 0;
  hipStream_t stream = nullptr;
  std::mutex mu;  // dint_submit* is serialised per engine: submission order = serial order
  uint32_t msg_size = 0;

  // batch scratch
  dint_scratch scratch{};
  dint_route_scratch route{};                              // dint_route_pack (allocated on first use)
  uint8_t *h_pinned = nullptr;                             // pinned buffer of dint_load_rows / dint_populate

  // host path: kNSlot staging slots {device request / reply buffers of one pass}; chunk k of a host submission
  // uses slot k % kNSlot, so the H2D copy of chunk k+1 and the D2H copy of chunk k-1 overlap the kernels of
  // chunk k (three streams, ordered by events).  Tickets are chunk sequence numbers.
  static const int kNSlot = 3;
  struct Slot {
    uint8_t *d_req = nullptr, *d_rep = nullptr;
    hipEvent_t h2d = nullptr, comp = nullptr, done = nullptr;
    uint64_t seq = 0;  // sequence number of the chunk that used the slot last (0 = never)
    // Pageable callers (a plain malloc / numpy buffer): hipMemcpyAsync never sees their memory.  The chunk is staged through
    // these page-locked buffers of the engine -- a host memcpy in on submission, a host memcpy out once the chunk has left
    // the GPU (at dint_wait, or when the slot is taken again).  r04: under rocprofv3 a pageable D2H / H2D of the HIP runtime
    // left the tail of a 4 KB page of one reply batch holding bytes of another (NOTEBOOK.md); the boundary contract
    // (lock_fasst/udp/net.h:33-48: the caller owns plain buffers) must not depend on the runtime's pageable path.
    uint8_t *h_req = nullptr, *h_rep = nullptr;
    uint8_t *deliver_to = nullptr;  // replies of the slot's chunk still to be copied from h_rep to the caller
    size_t deliver_bytes = 0;
  } slot[kNSlot];
  hipStream_t s_h2d = nullptr, s_d2h = nullptr;
  uint64_t next_seq = 1;
  uint64_t pool_seen = 0;  // pool_exhausted at the last host-path check
  unsigned long long *h_pool = nullptr;  // pinned: pool_exhausted counter read back with each host chunk

  // stream ordering: passes share one scratch set, so a pass enqueued on another stream than the previous one
  // first waits for it (dint_submit_device with a caller stream)
  hipStream_t last_stream = nullptr;
  hipEvent_t ev_order = nullptr;
  // ... the routing kernels have scratch of their own (route.blk) and touch neither tables nor pass
  // scratch: they are ordered among themselves only, so a pack for the next step overlaps the pass of this one
  hipStream_t route_last_stream = nullptr;
  hipEvent_t ev_route_order = nullptr;
  hipEvent_t ev_wait = nullptr, ev_signal = nullptr;  // dint_stream_wait / dint_stream_signal (re-recorded every call)

  // lock tables with DINT_FLAG_INPUTS_READY: the first half of pass k + 1 (k_lock_count, k_kv_scan_place: no table access) on
  // `helper` beside the second half of pass k; three scratch sets used in turn ([0] = `scratch`), an event pair per set
  struct LockPipe {
    static const int kSets = 3;
    bool ready = false;
    hipStream_t helper = nullptr;
    dint_scratch set[kSets];
    hipEvent_t counted[kSets] = {}, freed[kSets] = {}, ev_in = nullptr;
    bool used[kSets] = {};
    uint64_t seq = 0;
    bool after_serial = true;  // the previous pass ran on one stream (or there was none): the helper first waits for the caller's stream
  } lp;

  // lock tables (fasst / 2pl)
  uint2 *d_lock_tbl = nullptr;
  uint64_t n_slots = 0, n_local_slots = 0;
  dint_mod slots_mod{};

  // log
  dint_log log{};
  uint64_t log_drained = 0;  // records handed out (or given up as lost) by dint_log_drain
  uint64_t snap_log_drained = 0;  // ... when the snapshot was taken (dint_restore puts it back)

  // kv workloads (store / tatp / smallbank)
  dint_kv kv{};
  // look-ahead (r06, store / tatp): the batch whose partition stage ran inside the previous pass's k_kv_hot_part -- the
  // engine's next pass MUST be this one (its log records are in the ring, its records in the coarse bins)
  struct Ahead {
    bool valid = false;
    const void *req = nullptr;
    void *rep = nullptr;
    uint32_t n = 0;
    int set = 0;  // lock tables: the scratch set (0 = `scratch`, 1 = lp.set[1]) the announced batch's count stage filled
    uint32_t seg_cap = 0, n_seg = 0;  // a segmented batch (dint_submit_segments_multi_ahead): its geometry
    const void *cnt = nullptr;
  } ahead;

  // snapshot
  std::vector<std::pair<void *, size_t>> regions;  // device regions making up the engine state
  std::vector<void *> snap;

  dint_shard shard{0, 1};
  uint64_t batches = 0, requests = 0;
  KernelTimer timer;
};

namespace {

int dev_alloc(void **p, size_t bytes, bool zero = true) {
  hipError_t e = hipMalloc(p, bytes ? bytes : 1);
  if (e != hipSuccess) return fail(DINT_ENOMEM, "hipMalloc(%zu): %s", bytes, hipGetErrorString(e));
  if (zero) {
    e = hipMemset(*p, 0, bytes);
    if (e != hipSuccess) return fail(DINT_EHIP, "hipMemset: %s", hipGetErrorString(e));
  }
  return 0;
}

void add_region(dint_engine *e, void *p, size_t bytes) { e->regions.push_back({p, bytes}); }

// Every enqueue on behalf of the engine goes through here: work on `st` is ordered after whatever the engine
// enqueued last on another stream.
int order_stream(dint_engine *e, hipStream_t st) {
  if (e->last_stream && e->last_stream != st) {
    // the engine's own stream outlives every call: its position is recorded now.  A CALLER's stream is never touched
    // after the call that used it (the caller may have destroyed it by now -- ADVICE r02): mark_stream recorded
    // ev_order on it at the end of that call.
    if (e->last_stream == e->stream) HIP_TRY(hipEventRecord(e->ev_order, e->stream));
    HIP_TRY(hipStreamWaitEvent(st, e->ev_order, 0));
  }
  e->last_stream = st;
  return 0;
}
int mark_stream(dint_engine *e, hipStream_t st) {  // end of a call that enqueued the engine's work on `st`
  if (st != e->stream) HIP_TRY(hipEventRecord(e->ev_order, st));
  return 0;
}

// The routing calls are ordered the same way, in a domain of their own: a caller's stream is never touched after the
// call that used it (mark_route_stream records ev_route_order on it before that call returns -- ADVICE r03).
int order_route_stream(dint_engine *e, hipStream_t st) {
  if (!e->ev_route_order) HIP_TRY(hipEventCreateWithFlags(&e->ev_route_order, hipEventDisableTiming));
  if (e->route_last_stream && e->route_last_stream != st) {
    if (e->route_last_stream == e->stream) HIP_TRY(hipEventRecord(e->ev_route_order, e->stream));
    HIP_TRY(hipStreamWaitEvent(st, e->ev_route_order, 0));
  }
  e->route_last_stream = st;
  return 0;
}
int mark_route_stream(dint_engine *e, hipStream_t st) {  // end of a routing call that enqueued on `st`
  if (st != e->stream && e->route_last_stream == st) HIP_TRY(hipEventRecord(e->ev_route_order, st));
  return 0;
}

int slot_alloc(dint_engine *e, int k) {
  dint_engine::Slot &sl = e->slot[k];
  if (sl.d_req) return 0;
  const size_t bytes = (size_t)e->pass_max * e->msg_size + 64;
  int rc = dev_alloc((void **)&sl.d_req, bytes, false);
  if (!rc) rc = dev_alloc((void **)&sl.d_rep, bytes, false);
  if (rc) return rc;
  HIP_TRY(hipEventCreateWithFlags(&sl.h2d, hipEventDisableTiming));
  HIP_TRY(hipEventCreateWithFlags(&sl.comp, hipEventDisableTiming));
  HIP_TRY(hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
  return 0;
}

hipEvent_t *timer_events(dint_engine *e, int n_kernels, const char *const *names) {
  KernelTimer &t = e->timer;
  if (!t.on || t.passes >= KernelTimer::kMaxLaunch) return nullptr;
  t.n_kernels = n_kernels;
  for (int i = 0; i < n_kernels; i++) t.names[i] = names[i];
  size_t need = (size_t)(t.passes + 1) * (n_kernels + 1);
  while (t.ev.size() < need) {
    hipEvent_t ev;
    if (hipEventCreate(&ev) != hipSuccess) return nullptr;
    t.ev.push_back(ev);
  }
  hipEvent_t *p = &t.ev[(size_t)t.passes * (n_kernels + 1)];
  t.passes++;
  return p;
}

// kv passes: the tag of what the pieces of a hot key publish to each other in hotpub (k_kv.hip, kvh_word: 30 bits, never 0).
// When the counter wraps, the words are cleared on the pass's stream first, so a word left by the pass that carried the
// same tag 2^30 passes ago cannot be taken for this pass's.
int next_pass_seq(dint_engine *e, hipStream_t st) {
  e->scratch.kvs.pass_no++;  // (which set of the pass scratch a pass uses: dint_kv_sets)
  if (++e->scratch.pass_seq >= 0x3FFFFFFFu) {
    if (e->scratch.hotpub) HIP_TRY(hipMemsetAsync(e->scratch.hotpub, 0, (size_t)DINT_KV_HOTPUB_WORDS * sizeof(unsigned long long), st));
    if (e->scratch.kvs.bigrdy) HIP_TRY(hipMemsetAsync(e->scratch.kvs.bigrdy, 0, (size_t)DINT_KV_BIGQ_MAX * sizeof(uint32_t), st));
    e->scratch.pass_seq = 1;
  }
  return 0;
}
// kv engines: the log's ring position after the last pass lives in tail[(passes + 1) & 1] -- the partition of pass p reads
// tail[p & 1] and writes tail[(p + 1) & 1], because it may run beside the resolve stage of pass p - 1 (k_kv_pass), which used to
// copy one word onto the other.  dint_snapshot / dint_restore / dint_reset leave both words equal.
int log_cur(const dint_engine *e) { return e->kv.n_tables ? (int)((e->scratch.kvs.pass_no + 1) & 1) : 1; }

// one pass (n <= pass_max) on device buffers
// scratch sets 1 .. of the lock pipe: what k_lock_count / k_kv_scan_place / k_lock_resolve share within one pass
int lock_pipe_init(dint_engine *e) {
  dint_engine::LockPipe &lp = e->lp;
  if (lp.ready) return 0;
  // (a call after a failed one -- an allocation that did not fit -- must not create streams, events or buffers a second time)
  if (!lp.helper) HIP_TRY(hipStreamCreateWithFlags(&lp.helper, hipStreamNonBlocking));
  if (!lp.ev_in) HIP_TRY(hipEventCreateWithFlags(&lp.ev_in, hipEventDisableTiming));
  for (int k = 0; k < dint_engine::LockPipe::kSets; k++) {
    if (!lp.counted[k]) HIP_TRY(hipEventCreateWithFlags(&lp.counted[k], hipEventDisableTiming));
    if (!lp.freed[k]) HIP_TRY(hipEventCreateWithFlags(&lp.freed[k], hipEventDisableTiming));
    if (k == 0) continue;  // set 0 is e->scratch itself
    dint_scratch &s = lp.set[k];
    if (s.stats) {  // a set an earlier, failed call began: what it got is given back first
      const dint_scratch &o = e->scratch;
      if (s.bin_cnt != o.bin_cnt) hipFree(s.bin_cnt);
      if (s.bins != o.bins) hipFree(s.bins);
      if (s.blk_pub != o.blk_pub && s.blk_pub != o.blk_pub_next && s.blk_pub_next != o.blk_pub) hipFree(std::min(s.blk_pub, s.blk_pub_next));
      if (s.big != o.big && s.big != o.big_next && s.big_next != o.big) hipFree(std::min(s.big, s.big_next));
      if (s.bin_off != o.bin_off) hipFree(s.bin_off);
      if (s.ovl != o.ovl) hipFree(s.ovl);
      if (s.ovf != o.ovf) hipFree(s.ovf);
      if (s.kbins != o.kbins) hipFree(s.kbins);
    }
    s = e->scratch;  // (stats, lock_trace: shared)
    int rc = dev_alloc((void **)&s.bin_cnt, DINT_KV_PMAX * sizeof(uint32_t));
    if (!rc) rc = dev_alloc((void **)&s.bins, (size_t)DINT_KV_PMAX * DINT_KV_BINCAP * sizeof(uint64_t), false);
    if (!rc) rc = dev_alloc((void **)&s.blk_pub, 2 * 1024 * sizeof(uint32_t));
    if (!rc) rc = dev_alloc((void **)&s.big, 2 * (4 + DINT_KV_PMAX) * sizeof(uint32_t));
    if (!rc) rc = dev_alloc((void **)&s.bin_off, DINT_KV_PMAX * sizeof(uint32_t));
    if (!rc && hipMemset(s.bin_off, 0xFF, DINT_KV_PMAX * sizeof(uint32_t)) != hipSuccess) rc = fail(DINT_EHIP, "hipMemset");
    if (!rc && e->scratch.kbins) rc = dev_alloc((void **)&s.kbins, (size_t)e->scratch.kbins_slots * sizeof(uint64_t), false);
    if (!rc) rc = dev_alloc((void **)&s.ovl, (size_t)e->pass_max * sizeof(uint4), false);
    if (!rc) rc = dev_alloc((void **)&s.ovf, (size_t)e->pass_max * sizeof(uint64_t), false);
    if (rc) return rc;
    s.blk_pub_next = s.blk_pub + 1024;
    s.big_next = s.big + (4 + DINT_KV_PMAX);
  }
  // The zero fills above ran on the NULL stream, which the helper stream and the engine's own (non-blocking) do not wait for: the
  // first count stage on a new set could find its counters and region names not yet cleared.  Never seen in a plain run (the fills
  // take microseconds); under `rocprofv3 --pmc`, which serializes every dispatch, it was `k_lock_count`'s trap and wrong replies of
  // the DINT_FLAG_INPUTS_READY engine (r06, NOTEBOOK.md).
  HIP_TRY(hipDeviceSynchronize());
  lp.ready = true;
  return 0;
}
void lock_pipe_destroy(dint_engine *e) {
  dint_engine::LockPipe &lp = e->lp;
  if (lp.helper) { hipStreamSynchronize(lp.helper); hipStreamDestroy(lp.helper); }
  if (lp.ev_in) hipEventDestroy(lp.ev_in);
  for (int k = 0; k < dint_engine::LockPipe::kSets; k++) {
    if (lp.counted[k]) hipEventDestroy(lp.counted[k]);
    if (lp.freed[k]) hipEventDestroy(lp.freed[k]);
    if (k == 0) continue;
    // (whatever lock_pipe_init got before a failed allocation is freed too: a set's pointers start as copies of the engine's own
    // and are replaced one by one -- only the replaced ones are this set's)
    dint_scratch &s = lp.set[k];
    const dint_scratch &o = e->scratch;
    if (s.bin_cnt != o.bin_cnt) hipFree(s.bin_cnt);
    if (s.bins != o.bins) hipFree(s.bins);
    if (s.blk_pub != o.blk_pub && s.blk_pub != o.blk_pub_next && s.blk_pub_next != o.blk_pub) hipFree(std::min(s.blk_pub, s.blk_pub_next));
    if (s.big != o.big && s.big != o.big_next && s.big_next != o.big) hipFree(std::min(s.big, s.big_next));
    if (s.bin_off != o.bin_off) hipFree(s.bin_off);
    if (s.ovl != o.ovl) hipFree(s.ovl);
    if (s.ovf != o.ovf) hipFree(s.ovf);
    if (s.kbins != o.kbins) hipFree(s.kbins);
  }
}
// a lock pass in two halves (DINT_FLAG_INPUTS_READY, dint_submit_device): count + scan / place on the helper stream as soon
// as the scratch set is free, resolve on the caller's stream behind it -- and behind the previous pass's resolve, which owns
// the table.  `st` has been ordered behind the previous pass's stream already (order_stream).
int run_lock_pass_piped(dint_engine *e, const void *d_req, uint32_t n, void *d_rep, hipStream_t st, const dint_view &view) {
  if (int rc = lock_pipe_init(e)) return rc;
  dint_engine::LockPipe &lp = e->lp;
  const int b = (int)(lp.seq++ % dint_engine::LockPipe::kSets);
  dint_scratch &s = b == 0 ? e->scratch : lp.set[b];
  if (lp.after_serial) {  // whatever ran on one stream before (a pass with timing on, a restore, the first call): behind it
    HIP_TRY(hipEventRecord(lp.ev_in, st));
    HIP_TRY(hipStreamWaitEvent(lp.helper, lp.ev_in, 0));
    lp.after_serial = false;
  }
  if (lp.used[b]) HIP_TRY(hipStreamWaitEvent(lp.helper, lp.freed[b], 0));  // the resolve that read this set last
  dint_launch_lock_stage(e->cfg.workload, 1, d_req, d_rep, n, e->d_lock_tbl, e->slots_mod, e->shard, s, lp.helper, view);
  HIP_TRY(hipEventRecord(lp.counted[b], lp.helper));
  HIP_TRY(hipStreamWaitEvent(st, lp.counted[b], 0));
  dint_launch_lock_stage(e->cfg.workload, 2, d_req, d_rep, n, e->d_lock_tbl, e->slots_mod, e->shard, s, st, view);
  HIP_TRY(hipEventRecord(lp.freed[b], st));
  lp.used[b] = true;
  std::swap(s.big, s.big_next);  // the big-bin lists alternate between the passes of a set
  std::swap(s.blk_pub, s.blk_pub_next);
  return 0;
}

// An announced batch (dint_submit_device_ahead) that will not be submitted after all: its partition stage has filled the
// coarse bins and the control words of the next pass -- drain the GPU and leave the pass scratch as between passes.  What
// the stage wrote outside the scratch stays: the announced batch's log records (the ring is about to be restored or reset
// by the callers that cancel silently; the others report DINT_ESTATE).
int ahead_cancel(dint_engine *e) {
  if (!e->ahead.valid) return 0;
  e->ahead.valid = false;
  HIP_TRY(hipDeviceSynchronize());
  if (!e->kv.n_tables) {  // a lock engine: the counters, the big-bin list and the region names of the set the count stage filled
    dint_scratch &ls = e->ahead.set == 0 ? e->scratch : e->lp.set[e->ahead.set];
    HIP_TRY(hipMemset(ls.bin_cnt, 0, DINT_KV_PMAX * sizeof(uint32_t)));
    HIP_TRY(hipMemset(std::min(ls.big, ls.big_next), 0, 2 * (4 + DINT_KV_PMAX) * sizeof(uint32_t)));
    HIP_TRY(hipMemset(ls.bin_off, 0xFF, DINT_KV_PMAX * sizeof(uint32_t)));
    HIP_TRY(hipDeviceSynchronize());  // (the fills ran on the null stream: the engine's streams do not wait for it)
    return 0;
  }
  dint_scratch &s = e->scratch;
  for (int k = 0; k < 2; k++) HIP_TRY(hipMemset(s.kvs.bin_cnt[k], 0, DINT_KV_PMAX * sizeof(uint32_t)));
  HIP_TRY(hipMemset(s.kvs.ctl[0], 0, 3 * 16 * sizeof(uint32_t)));
  HIP_TRY(hipMemset(s.kvs.pub[0], 0, 3 * 1024 * sizeof(uint32_t)));
  if (e->log.tail) {  // the tail the cancelled partition published is not the current one
    uint32_t t[4];
    HIP_TRY(hipMemcpy(t, e->log.tail, sizeof t, hipMemcpyDeviceToHost));
    const int c = log_cur(e);  // (the cancelled pass was never counted: `c` is the word it read, c ^ 1 the one it wrote)
    uint64_t appended = (uint64_t)t[2] | ((uint64_t)t[3] << 32);
    appended -= std::min<uint64_t>(appended, (t[c ^ 1] + e->log.cap - t[c]) % e->log.cap);
    t[c ^ 1] = t[c]; t[2] = (uint32_t)appended; t[3] = (uint32_t)(appended >> 32);
    HIP_TRY(hipMemcpy(e->log.tail, t, sizeof t, hipMemcpyHostToDevice));
  }
  HIP_TRY(hipDeviceSynchronize());  // (the fills ran on the null stream: the engine's streams do not wait for it)
  return 0;
}

int run_pass(dint_engine *e, const void *d_req, uint32_t n, void *d_rep, hipStream_t st, int load_mode = 0,
             const dint_view &view = dint_flat_view(), bool inputs_ready = false, const dint_kv_ahead *next = nullptr) {
  bool part_done = false;
  if (e->ahead.valid) {  // the pass that was announced, and nothing else
    if (e->ahead.req != d_req || e->ahead.rep != d_rep || e->ahead.n != n || view.seg_cap != e->ahead.seg_cap || load_mode) {
      if (int rc = ahead_cancel(e)) return rc;
      return fail(DINT_ESTATE, "the batch announced by dint_submit_device_ahead (%u requests at %p) must be the engine's next "
                               "submission; its log records are appended already", e->ahead.n, e->ahead.req);
    }
    e->ahead.valid = false;
    part_done = true;
  }
  if (int rc = order_stream(e, st)) return rc;
  const bool piped = inputs_ready && n && !e->timer.on && (e->cfg.workload == DINT_WL_FASST || e->cfg.workload == DINT_WL_2PL);
  if (!piped) e->lp.after_serial = true;
  static const char *const lock_names[] = {"k_lock_count", "k_kv_scan_place", "k_lock_resolve"};
  static const char *const log_names[] = {"k_log_append"};
  static const char *const kv_names[] = {"k_kv_part", "k_kv_resolve", "k_kv_hot", "k_kv_big"};
  static const char *const kv_names_one[] = {"k_kv_part", "k_kv_pass", "k_kv_late"};  // (store / tatp, r06: one launch per pass + the late list's)
  switch (piped ? DINT_WL_COUNT : e->cfg.workload) {
    case DINT_WL_COUNT:
      if (int rc = run_lock_pass_piped(e, d_req, n, d_rep, st, view)) return rc;
      break;
    case DINT_WL_FASST:
    case DINT_WL_2PL: {
      // r06: with the next batch announced (or the next slice of one long submission), this pass's resolve stage and the next
      // pass's count stage are ONE launch (k_lock_pass) -- two scratch sets used in turn, `scratch` and lp.set[1]
      const int b = part_done ? e->ahead.set : 0;
      if (next && (!n || !next->n || view.seg_cap || next->view.seg_cap || n > DINT_MICRO_BATCH || next->n > DINT_MICRO_BATCH ||
                   !e->scratch.kbins || getenv("DINT_LOCK_NO_FUSE")))
        next = nullptr;
      if ((next || b) && !e->lp.ready)
        if (int rc = lock_pipe_init(e)) return rc;
      dint_scratch &s = b == 0 ? e->scratch : e->lp.set[b];
      hipEvent_t *ev = timer_events(e, 3, lock_names);
      if (!part_done && !next) {
        if (e->cfg.workload == DINT_WL_FASST) dint_launch_fasst(d_req, d_rep, n, e->d_lock_tbl, e->slots_mod, e->shard, s, st, ev, view);
        else dint_launch_2pl(d_req, d_rep, n, e->d_lock_tbl, e->slots_mod, e->shard, s, st, ev, view);
      } else {
        if (ev) HIP_TRY(hipEventRecord(ev[0], st));
        if (!part_done) dint_launch_lock_stage(e->cfg.workload, 1, d_req, d_rep, n, e->d_lock_tbl, e->slots_mod, e->shard, s, st, view);
        if (ev) { HIP_TRY(hipEventRecord(ev[1], st)); HIP_TRY(hipEventRecord(ev[2], st)); }
        dint_scratch &sn = b == 0 ? e->lp.set[1] : e->scratch;
        if (!next || !dint_launch_lock_fused(e->cfg.workload, d_rep, n, e->d_lock_tbl, e->slots_mod, e->shard, s, view, next->d_req, next->d_rep,
                                             next->n, sn, next->view, st)) {
          dint_launch_lock_stage(e->cfg.workload, 2, d_req, d_rep, n, e->d_lock_tbl, e->slots_mod, e->shard, s, st, view);
          next = nullptr;
        }
        if (ev) HIP_TRY(hipEventRecord(ev[3], st));
      }
      std::swap(s.big, s.big_next);  // the big-bin lists alternate between the passes of a set
      std::swap(s.blk_pub, s.blk_pub_next);
      if (next) {
        e->ahead.valid = true;
        e->ahead.req = next->d_req; e->ahead.rep = next->d_rep; e->ahead.n = next->n;
        e->ahead.set = b ^ 1;
        e->ahead.seg_cap = 0; e->ahead.n_seg = 0; e->ahead.cnt = nullptr;
      }
      break;
    }
    case DINT_WL_LOG:
      if (view.seg_cap) return fail(DINT_ESTATE, "the log workload is not sharded by key");
      dint_launch_log(d_req, d_rep, n, e->log, e->scratch, st, timer_events(e, 1, log_names));
      std::swap(e->scratch.blk_pub, e->scratch.blk_pub_next);  // the tile counts alternate between passes
      break;
    case DINT_WL_STORE:
    case DINT_WL_TATP:
    case DINT_WL_SMALLBANK:
      if (int rc = next_pass_seq(e, st)) return rc;  // tags what the pieces of a hot key publish in this pass
      if (next && (!n || !next->n || view.seg_cap || next->view.seg_cap || !dint_kv_ahead_ok(e->kv, load_mode))) next = nullptr;
      dint_launch_kv(d_req, d_rep, n, e->kv, e->log, e->scratch, load_mode, st,
                     dint_kv_one_launch(e->kv, load_mode) ? timer_events(e, 3, kv_names_one) : timer_events(e, 4, kv_names), view, part_done, next);
      if (next) {
        e->ahead.valid = true;
        e->ahead.req = next->d_req; e->ahead.rep = next->d_rep; e->ahead.n = next->n;
        e->ahead.seg_cap = 0; e->ahead.n_seg = 0; e->ahead.cnt = nullptr;
      }
      break;
    default:
      return fail(DINT_EINVAL, "bad workload");
  }
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) return fail(DINT_EHIP, "kernel launch: %s", hipGetErrorString(err));
  if (int rc = mark_stream(e, st)) return rc;
  e->batches++;
  e->requests += n;
  return 0;
}

// bulk load (kvs_insert semantics, explicit version) = passes of internal LOAD requests through the
// same scatter/resolve kernels, so rows land in request order exactly like wire INSERTs
int load_rows_locked(dint_engine *e, uint32_t table, const uint64_t *keys, const uint32_t *vers, const uint8_t *vals,
                     uint64_t n) {
  const dint_kv_fmt f = dint_kv_format(e->cfg.workload);
  for (uint64_t off = 0; off < n; off += e->pass_max) {
    const uint32_t m = (uint32_t)std::min<uint64_t>(e->pass_max, n - off);
    const size_t bytes = (size_t)m * f.msg;
    memset(e->h_pinned, 0, bytes);
    for (uint32_t i = 0; i < m; i++) {
      uint8_t *msg = e->h_pinned + (size_t)i * f.msg;
      msg[f.type] = (uint8_t)DINT_KV_LOAD_OP;
      if (f.table != 0xFFFFFFFFu) msg[f.table] = (uint8_t)table;
      memcpy(msg + f.key, &keys[off + i], 8);
      memcpy(msg + f.val, vals + (off + i) * f.val_size, f.val_size);
      const uint32_t ver = vers ? vers[off + i] : 0;
      memcpy(msg + f.ver, &ver, 4);
    }
    if (int rc = order_stream(e, e->stream)) return rc;
    HIP_TRY(hipMemcpyAsync(e->slot[0].d_req, e->h_pinned, bytes, hipMemcpyHostToDevice, e->stream));
    if (int rc = run_pass(e, e->slot[0].d_req, m, e->slot[0].d_req, e->stream, 1)) return rc;
    e->batches--;  // population passes are not request batches
    e->requests -= m;
    HIP_TRY(hipStreamSynchronize(e->stream));
  }
  return 0;
}

}  // namespace

extern "C" {

const char *dint_last_error(void) { return g_err.c_str(); }

int dint_msg_size(uint32_t workload) { return workload < DINT_WL_COUNT ? kMsgSize[workload] : DINT_EINVAL; }

int dint_engine_create(const dint_config *cfg, dint_engine_t **out) {
  if (!cfg || !out) return fail(DINT_EINVAL, "null argument");
  if (cfg->abi_version != DINT_ABI_VERSION) return fail(DINT_EINVAL, "abi_version %u != %u", cfg->abi_version, DINT_ABI_VERSION);
  if (cfg->workload >= DINT_WL_COUNT) return fail(DINT_EINVAL, "unknown workload %u", cfg->workload);
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(DINT_ENODEV, "no HIP device");
  int dev = cfg->device;
  if (dev < 0) HIP_TRY(hipGetDevice(&dev));
  if (dev >= ndev) return fail(DINT_ENODEV, "device %d of %d", dev, ndev);
  HIP_TRY(hipSetDevice(dev));
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, dev));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(DINT_ENODEV, "device %d is %s, this library is built for gfx950 only", dev, prop.gcnArchName);

  dint_engine *e = new dint_engine();
  e->cfg = *cfg;
  e->device = dev;
  e->msg_size = kMsgSize[cfg->workload];
  e->shard.count = cfg->shard_count ? cfg->shard_count : 1;
  e->shard.index = cfg->shard_index;
  int rc = 0;
#define TRY(x) do { rc = (x); if (rc) { dint_engine_destroy(e); return rc; } } while (0)
  if (e->shard.index >= e->shard.count || e->shard.count > 255) {
    delete e;
    return fail(DINT_EINVAL, "shard %u of %u", cfg->shard_index, cfg->shard_count);
  }
  // (the copy streams of the host path exist only with DINT_FLAG_COPY_STREAMS, and are created on first use)
  // (a high-priority engine stream beside the wide routing kernels of a sharded step was measured: --force-exchange 1,310 ->
  // 804 Mtxn/s; DINT_ENGINE_PRIORITY keeps the knob)
  const char *pe = getenv("DINT_ENGINE_PRIORITY");
  const int prio = pe ? atoi(pe) : 0;
  if (hipStreamCreateWithPriority(&e->stream, hipStreamNonBlocking, prio) != hipSuccess ||
      hipEventCreateWithFlags(&e->ev_order, hipEventDisableTiming) != hipSuccess) {
    dint_engine_destroy(e);
    return fail(DINT_EHIP, "hipStreamCreate");
  }
  const uint32_t wl = cfg->workload;
  const bool is_kv = wl == DINT_WL_STORE || wl == DINT_WL_TATP || wl == DINT_WL_SMALLBANK;
  // requests per kernel pass: the request index must fit the batch record (20 bits)
  // The lock tables default to BASELINE's batch size: their dominant-slot path (index bitmaps, k_locks.hip) covers passes of
  // up to 65,536 requests, and a longer pass runs 3x (lock_fasst) to 16x (lock_2pl) slower PER REQUEST on a Zipf stream (r05
  // bench, pass_1m) -- so a dint_submit of a million requests is sixteen passes unless the caller asks for longer ones.
  e->pass_max = (wl == DINT_WL_FASST || wl == DINT_WL_2PL) ? DINT_MICRO_BATCH : DINT_KV_PASS;
  if (cfg->max_pass) e->pass_max = std::min<uint32_t>(DINT_KV_PASS, std::max<uint32_t>(cfg->max_pass, 64u));
  if (wl == DINT_WL_LOG || wl == DINT_WL_TATP || wl == DINT_WL_SMALLBANK) {
    // a pass never laps the log ring, so a DELETE_LOG record keeps the val bytes of the record it overwrites
    // exactly as in the serial reference
    const uint32_t cap = cfg->log_entries ? cfg->log_entries : 1000000u;
    e->pass_max = std::min<uint32_t>(e->pass_max, cap);
  }
  TRY(dev_alloc((void **)&e->scratch.stats, sizeof(dint_dev_stats)));
  add_region(e, e->scratch.stats, sizeof(dint_dev_stats));
  if (wl != DINT_WL_LOG) {  // the bins of one pass: 64 records in place per bin + the pass's overflow area
    TRY(dev_alloc((void **)&e->scratch.bin_cnt, DINT_KV_PMAX * sizeof(uint32_t)));
    if (is_kv) {
      // coarse bins of 16-byte records, DINT_KV_CAP_MULT (64) times the mean load in place each: C * cap = C * (64 ceil(n / C)
      // + 64) <= 64 n + 128 C slots -- 1 GB for passes of 2^20 requests, of which a pass touches what it fills.  (r04 / r05
      // kept 2 x the mean and sent a hot key's excess to the pass's overflow list, which every bin that has records there
      // reads from end to end, twice: smallbank at Zipf 0.99 -- 100,000 entries, a dozen such bins -- spent 100 us of its
      // 135 us resolve kernel there.  288 GB of HBM are there to be used.)
      e->scratch.kbins_slots = (uint64_t)dint_kv_cap_mult() * e->pass_max + 128ull * DINT_KV_CMAX;
      TRY(dev_alloc((void **)&e->scratch.kbins, (size_t)e->scratch.kbins_slots * sizeof(uint4), false));
      TRY(dev_alloc((void **)&e->scratch.bigq, (size_t)DINT_KV_BIGQ_MAX * 3 * sizeof(uint4), false));  // KVQ_W uint4 per work item (k_kv.hip)
      TRY(dev_alloc((void **)&e->scratch.hotpub, (size_t)DINT_KV_HOTPUB_WORDS * sizeof(unsigned long long)));
      TRY(dev_alloc((void **)&e->scratch.lateq, (size_t)DINT_KV_BIGQ_MAX * sizeof(uint4), false));
      // the second set of what the partition stage writes (dint_kv_sets): 288 GB of HBM are there to be used
      TRY(dev_alloc((void **)&e->scratch.kvs.bin_cnt[1], DINT_KV_PMAX * sizeof(uint32_t)));
      TRY(dev_alloc((void **)&e->scratch.kvs.kbins[1], (size_t)e->scratch.kbins_slots * sizeof(uint4), false));
      TRY(dev_alloc((void **)&e->scratch.kvs.ovl[1], (size_t)e->pass_max * sizeof(uint4) * 2, false));
      TRY(dev_alloc((void **)&e->scratch.kvs.bigrdy, (size_t)DINT_KV_BIGQ_MAX * sizeof(uint32_t)));
      if (wl == DINT_WL_SMALLBANK)  // what the pieces of a hot account's row tell each other: 40 words per work item (kv_sb_item)
        TRY(dev_alloc((void **)&e->scratch.kvs.sbx, (size_t)DINT_KV_BIGQ_MAX * DINT_KV_SBX_WORDS * sizeof(uint64_t), false));
    } else {
      TRY(dev_alloc((void **)&e->scratch.bins, (size_t)DINT_KV_PMAX * DINT_KV_BINCAP * sizeof(uint64_t), false));
      // lock tables, passes of <= 65,536 requests (k_locks.hip, LK_DIRECT_NMAX): a big bin's records beyond the 64 in place go
      // straight to a region of its own -- 1,024 regions (a pass of 65,536 has at most that many bins of more than 64) of
      // 65,536 records (a bin holds at most the pass): 512 MB of address space, touched as filled.  DINT_LOCK_NO_DIRECT at
      // creation: not allocated (r01-r05's overflow list + k_kv_scan_place for every pass).
      if (!getenv("DINT_LOCK_NO_DIRECT")) {
        e->scratch.kbins_slots = 1024ull * 65536ull;
        TRY(dev_alloc((void **)&e->scratch.kbins, (size_t)e->scratch.kbins_slots * sizeof(uint64_t), false));
      }
    }
    TRY(dev_alloc((void **)&e->scratch.blk_pub, 3 * 1024 * sizeof(uint32_t)));
    e->scratch.blk_pub_next = e->scratch.blk_pub + 1024;
    TRY(dev_alloc((void **)&e->scratch.big, 2 * (4 + DINT_KV_PMAX) * sizeof(uint32_t)));
    e->scratch.big_next = e->scratch.big + (4 + DINT_KV_PMAX);
    if (is_kv)  // (the three sets of control words and tile counts live in the allocations the lock tables use as two)
      for (int k = 0; k < 3; k++) { e->scratch.kvs.ctl[k] = e->scratch.big + 16 * k; e->scratch.kvs.pub[k] = e->scratch.blk_pub + 1024 * k; }
    TRY(dev_alloc((void **)&e->scratch.bin_off, DINT_KV_PMAX * sizeof(uint32_t)));
    TRY(hipMemset(e->scratch.bin_off, 0xFF, DINT_KV_PMAX * sizeof(uint32_t)) == hipSuccess ? 0 : fail(DINT_EHIP, "hipMemset"));  // (lock tables, direct big bins: "no region named" between passes)
    TRY(dev_alloc((void **)&e->scratch.ovl, (size_t)e->pass_max * sizeof(uint4) * (is_kv ? 2 : 1), false));
    TRY(dev_alloc((void **)&e->scratch.ovf, (size_t)e->pass_max * sizeof(uint64_t), false));
    if (is_kv) {
      TRY(dev_alloc((void **)&e->scratch.ovf2, (size_t)e->pass_max * sizeof(uint64_t), false));
      e->scratch.kvs.bin_cnt[0] = e->scratch.bin_cnt; e->scratch.kvs.kbins[0] = e->scratch.kbins; e->scratch.kvs.ovl[0] = e->scratch.ovl;
    }
    if ((e->cfg.workload == DINT_WL_FASST || e->cfg.workload == DINT_WL_2PL) && getenv("DINT_KV_TRACE")) {
      TRY(dev_alloc((void **)&e->kv.d_trace, (size_t)DINT_KV_TRACE_WORDS * 8, true));
      e->scratch.lock_trace = e->kv.d_trace;
    }
  } else {  // the log append has no bins: per-block counts of its scan only
    TRY(dev_alloc((void **)&e->scratch.blk_pub, 2 * (1024 + 16) * sizeof(uint32_t)));  // per tile: valid requests; + ticket, finished
    e->scratch.blk_pub_next = e->scratch.blk_pub + (1024 + 16);
  }
  TRY(slot_alloc(e, 0));
  if (hipHostMalloc((void **)&e->h_pinned, (size_t)e->pass_max * e->msg_size + 64, hipHostMallocDefault) != hipSuccess ||
      hipHostMalloc((void **)&e->h_pool, 64, hipHostMallocDefault) != hipSuccess) {
    dint_engine_destroy(e);
    return fail(DINT_ENOMEM, "hipHostMalloc");
  }
  *e->h_pool = 0;

  if (wl == DINT_WL_FASST || wl == DINT_WL_2PL) {
    e->n_slots = cfg->n_slots ? cfg->n_slots : 36000000ull;  // lock_fasst/udp/utils.h:12
    if (e->n_slots > 0xFFFFFFF0ull) { dint_engine_destroy(e); return fail(DINT_EINVAL, "n_slots too large"); }
    e->slots_mod = dint_make_mod(e->n_slots);
    e->n_local_slots = (e->n_slots + e->shard.count - 1) / e->shard.count;
    TRY(dev_alloc((void **)&e->d_lock_tbl, e->n_local_slots * sizeof(uint2)));
    add_region(e, e->d_lock_tbl, e->n_local_slots * sizeof(uint2));
  }
  if (wl == DINT_WL_LOG || wl == DINT_WL_TATP || wl == DINT_WL_SMALLBANK) {
    e->log.cap = cfg->log_entries ? cfg->log_entries : 1000000u;  // log_server/udp/utils.h:16
    TRY(dev_alloc((void **)&e->log.ring, (size_t)e->log.cap * 64));
    TRY(dev_alloc((void **)&e->log.tail, 4 * sizeof(uint32_t)));
    add_region(e, e->log.ring, (size_t)e->log.cap * 64);
    add_region(e, e->log.tail, 4 * sizeof(uint32_t));
  }
  if (is_kv) {
    rc = dint_kv_create(&e->kv, wl, cfg->n_rows, e->shard, cfg->pool_entries, cfg->flags);
    if (rc) { dint_engine_destroy(e); return fail(rc, "kv table allocation failed (%s)", g_err.c_str()); }
    // the owner-key comparison of DINT_FLAG_LOCK_SAME_KEY exists on the request-by-request path only
    e->kv.force_rounds = ((cfg->flags & (DINT_FLAG_KV_ROUNDS | DINT_FLAG_LOCK_SAME_KEY)) ? 1 : 0) |
                         ((cfg->flags & DINT_FLAG_KV_NO_HOT) ? 2 : 0);
    for (auto &r : dint_kv_regions(&e->kv)) add_region(e, r.first, r.second);
  }
  if (hipDeviceSynchronize() != hipSuccess) {
    dint_engine_destroy(e);
    return fail(DINT_EHIP, "hipDeviceSynchronize after engine set-up");
  }
#undef TRY
  *out = e;
  return 0;
}

void dint_engine_destroy(dint_engine_t *e) {
  if (!e) return;
  hipSetDevice(e->device);
  hipDeviceSynchronize();
  for (auto &sl : e->slot)  // (everything has left the GPU: undelivered replies of pageable callers are not dropped)
    if (sl.deliver_to && sl.h_rep) { memcpy(sl.deliver_to, sl.h_rep, sl.deliver_bytes); sl.deliver_to = nullptr; }
  for (hipEvent_t ev : e->timer.ev) hipEventDestroy(ev);
  for (void *p : e->snap) hipFree(p);
  hipFree(e->scratch.bin_cnt);
  hipFree(e->scratch.bins);
  hipFree(e->scratch.kbins);
  hipFree(e->scratch.bigq);
  hipFree(e->scratch.hotpub);
  hipFree(e->scratch.lateq);
  hipFree(e->scratch.kvs.bin_cnt[1]);
  hipFree(e->scratch.kvs.kbins[1]);
  hipFree(e->scratch.kvs.ovl[1]);
  hipFree(e->scratch.kvs.bigrdy);
  hipFree(e->scratch.kvs.sbx);
  hipFree(e->scratch.stats);
  hipFree(e->scratch.blk_cnt);
  hipFree(std::min(e->scratch.blk_pub, e->scratch.blk_pub_next));
  hipFree(std::min(e->scratch.big, e->scratch.big_next));
  hipFree(e->scratch.bin_off);
  hipFree(e->scratch.ovl);
  hipFree(e->scratch.ovf);
  hipFree(e->scratch.ovf2);
  if (e->scratch.lock_trace) { hipFree(e->scratch.lock_trace); e->kv.d_trace = nullptr; }
  for (auto &sl : e->slot) {
    hipFree(sl.d_req);
    hipFree(sl.d_rep);
    if (sl.h_req) hipHostFree(sl.h_req);
    if (sl.h_rep) hipHostFree(sl.h_rep);
    if (sl.h2d) hipEventDestroy(sl.h2d);
    if (sl.comp) hipEventDestroy(sl.comp);
    if (sl.done) hipEventDestroy(sl.done);
  }
  hipFree(std::min(e->route.blk, e->route.blk_next));
  if (e->h_pinned) hipHostFree(e->h_pinned);
  if (e->h_pool) hipHostFree(e->h_pool);
  if (e->ev_order) hipEventDestroy(e->ev_order);
  if (e->ev_route_order) hipEventDestroy(e->ev_route_order);
  if (e->ev_wait) hipEventDestroy(e->ev_wait);
  if (e->ev_signal) hipEventDestroy(e->ev_signal);
  if (e->s_h2d && e->s_h2d != e->stream) hipStreamDestroy(e->s_h2d);
  if (e->s_d2h && e->s_d2h != e->stream) hipStreamDestroy(e->s_d2h);
  lock_pipe_destroy(e);
  hipFree(e->d_lock_tbl);
  hipFree(e->log.ring);
  hipFree(e->log.tail);
  dint_kv_destroy(&e->kv);
  if (e->stream) hipStreamDestroy(e->stream);
  delete e;
}

namespace {
int submit_device_locked(dint_engine *e, const void *d_reqs, uint32_t n, void *d_replies, const void *d_next_reqs, uint32_t next_n,
                         void *d_next_replies, void *stream) {
  HIP_TRY(hipSetDevice(e->device));
  hipStream_t st = stream ? (hipStream_t)stream : e->stream;
  const uint8_t *rq = (const uint8_t *)d_reqs;
  uint8_t *rp = (uint8_t *)d_replies;
  if (n == 0 && e->ahead.valid) return run_pass(e, d_reqs, 0, d_replies, st);  // (reports the broken announcement)
  for (uint32_t off = 0; off < n; off += e->pass_max) {
    uint32_t m = std::min<uint32_t>(e->pass_max, n - off);
    // what the engine's next pass will be: the next slice of this array (stream order has it complete: the whole array
    // precedes this call) -- or the first pass of the batch the caller announced
    dint_kv_ahead nx = {nullptr, nullptr, 0, dint_flat_view()};
    if (off + m < n) {
      nx.d_req = rq + (size_t)(off + m) * e->msg_size; nx.d_rep = rp + (size_t)(off + m) * e->msg_size;
      nx.n = std::min<uint32_t>(e->pass_max, n - off - m);
    } else if (next_n) {
      nx.d_req = d_next_reqs; nx.d_rep = d_next_replies; nx.n = std::min<uint32_t>(e->pass_max, next_n);
    }
    int rc = run_pass(e, rq + (size_t)off * e->msg_size, m, rp + (size_t)off * e->msg_size, st, 0, dint_flat_view(),
                      (e->cfg.flags & DINT_FLAG_INPUTS_READY) != 0,
                      (e->kv.n_tables || e->cfg.workload == DINT_WL_FASST || e->cfg.workload == DINT_WL_2PL) && nx.n ? &nx : nullptr);
    if (rc) return rc;
  }
  return 0;
}
}  // namespace

int dint_submit_device(dint_engine_t *e, const void *d_reqs, uint32_t n, void *d_replies, void *stream) {
  if (!e || (n && (!d_reqs || !d_replies))) return fail(DINT_EINVAL, "null argument");
  std::lock_guard<std::mutex> lk(e->mu);
  return submit_device_locked(e, d_reqs, n, d_replies, nullptr, 0, nullptr, stream);
}

int dint_submit_device_ahead(dint_engine_t *e, const void *d_reqs, uint32_t n, void *d_replies, const void *d_next_reqs,
                             uint32_t next_n, void *d_next_replies, void *stream) {
  if (!e || (n && (!d_reqs || !d_replies)) || (next_n && (!d_next_reqs || !d_next_replies))) return fail(DINT_EINVAL, "null argument");
  std::lock_guard<std::mutex> lk(e->mu);
  return submit_device_locked(e, d_reqs, n, d_replies, d_next_reqs, next_n, d_next_replies, stream);
}

int dint_submit_segments(dint_engine_t *e, void *d_base, uint32_t n_seg, uint32_t seg_cap, uint64_t seg_stride,
                         const void *d_cnt, uint64_t cnt_stride, void *stream) {
  if (!e || (n_seg && (!d_base || !d_cnt))) return fail(DINT_EINVAL, "null argument");
  if (seg_cap < 2 || seg_cap > e->pass_max) return fail(DINT_EINVAL, "seg_cap %u outside [2, %u]", seg_cap, e->pass_max);
  if (seg_stride < (uint64_t)seg_cap * e->msg_size) return fail(DINT_EINVAL, "seg_stride smaller than a segment");
  std::lock_guard<std::mutex> lk(e->mu);
  HIP_TRY(hipSetDevice(e->device));
  hipStream_t st = stream ? (hipStream_t)stream : e->stream;
  const uint32_t per_pass = e->pass_max / seg_cap;  // whole segments per kernel pass
  for (uint32_t k0 = 0; k0 < n_seg; k0 += per_pass) {
    const uint32_t ns = std::min(per_pass, n_seg - k0);
    uint8_t *base = (uint8_t *)d_base + (size_t)k0 * seg_stride;
    const dint_view v = dint_seg_view(ns, seg_cap, seg_stride, (const uint8_t *)d_cnt + (size_t)k0 * cnt_stride, cnt_stride);
    int rc = run_pass(e, base, ns * seg_cap, base, st, 0, v);
    if (rc) return rc;
  }
  return 0;
}

int dint_submit_segments_multi(const dint_segments_item *items, uint32_t n_items, void *stream) {
  return dint_submit_segments_multi_ahead(items, n_items, nullptr, stream);
}

int dint_submit_segments_multi_ahead(const dint_segments_item *items, uint32_t n_items, const dint_segments_item *next, void *stream) {
  if (!items || n_items == 0) return fail(DINT_EINVAL, "null argument");
  for (uint32_t k = 0; k < n_items; k++) {
    if (!items[k].engine) return fail(DINT_EINVAL, "null engine");
    for (uint32_t j = 0; j < k; j++)
      if (items[j].engine == items[k].engine) return fail(DINT_EINVAL, "an engine appears twice");
  }
  hipStream_t st = stream ? (hipStream_t)stream : items[0].engine->stream;
  bool one_set = n_items <= DINT_KV_MULTI_MAX;
  for (uint32_t k = 0; k < n_items && one_set; k++) {
    const dint_engine *e = items[k].engine;
    const uint32_t wl = e->cfg.workload;
    one_set = (wl == DINT_WL_STORE || wl == DINT_WL_TATP || wl == DINT_WL_SMALLBANK) && wl == items[0].engine->cfg.workload &&
              e->device == items[0].engine->device && items[k].n_seg > 0 && items[k].seg_cap >= 2 &&
              (uint64_t)items[k].n_seg * items[k].seg_cap <= e->pass_max;
  }
  if (!one_set) {
    for (uint32_t k = 0; k < n_items; k++)
      if (int rc = dint_submit_segments(items[k].engine, items[k].d_base, items[k].n_seg, items[k].seg_cap, items[k].seg_stride,
                                        items[k].d_cnt, items[k].cnt_stride, st))
        return rc;
    return 0;  // (an announcement is dropped: nothing ran ahead)
  }
  // the announced next step: the same engines, the same geometry, other buffers -- else it is ignored
  for (uint32_t k = 0; next && k < n_items; k++)
    if (next[k].engine != items[k].engine || !next[k].d_base || !next[k].d_cnt || next[k].n_seg != items[k].n_seg ||
        next[k].seg_cap != items[k].seg_cap || next[k].seg_stride != items[k].seg_stride || next[k].cnt_stride != items[k].cnt_stride)
      next = nullptr;
  if (next && !dint_kv_multi_ahead_ok(items[0].engine->kv)) next = nullptr;
  // every argument is checked before an engine is locked or a stream ordered: an EINVAL on item k must not leave the
  // engines of items 0 .. k-1 with a new last_stream and a stale ordering event
  for (uint32_t k = 0; k < n_items; k++) {
    const dint_segments_item &it = items[k];
    if (!it.d_base || !it.d_cnt) return fail(DINT_EINVAL, "null argument");
    if (it.seg_stride < (uint64_t)it.seg_cap * it.engine->msg_size) return fail(DINT_EINVAL, "seg_stride smaller than a segment");
  }
  std::vector<dint_engine *> es;
  for (uint32_t k = 0; k < n_items; k++) es.push_back(items[k].engine);
  std::sort(es.begin(), es.end());
  std::vector<std::unique_lock<std::mutex>> locks;
  for (dint_engine *e : es) locks.emplace_back(e->mu);  // address order: two calls that share engines cannot deadlock
  HIP_TRY(hipSetDevice(items[0].engine->device));
  dint_kv_pass pass[DINT_KV_MULTI_MAX], npass[DINT_KV_MULTI_MAX];
  // what ran ahead: all of the engines' partitions (the previous call of this kind announced exactly this step), or none
  uint32_t n_done = 0;
  for (uint32_t k = 0; k < n_items; k++) {
    const dint_engine::Ahead &a = items[k].engine->ahead;
    if (a.valid && a.req == items[k].d_base && a.n == items[k].n_seg * items[k].seg_cap && a.seg_cap == items[k].seg_cap &&
        a.n_seg == items[k].n_seg && a.cnt == items[k].d_cnt)
      n_done++;
    else if (a.valid)
      n_done = 0xFFFFu;
  }
  if (n_done != 0 && n_done != n_items) {
    for (uint32_t k = 0; k < n_items; k++)
      if (int rc = ahead_cancel(items[k].engine)) return rc;
    return fail(DINT_ESTATE, "a batch announced by dint_submit_device_ahead / dint_submit_segments_multi_ahead must be the engine's next submission");
  }
  const bool part_done = n_done == n_items;
  for (uint32_t k = 0; k < n_items; k++) items[k].engine->ahead.valid = false;
  for (uint32_t k = 0; k < n_items; k++) {
    const dint_segments_item &it = items[k];
    dint_engine *e = it.engine;
    if (int rc = order_stream(e, st)) return rc;
    if (int rc = next_pass_seq(e, st)) return rc;
    pass[k].d_req = it.d_base; pass[k].d_rep = it.d_base; pass[k].n = it.n_seg * it.seg_cap;
    pass[k].kv = &e->kv; pass[k].log = e->log; pass[k].s = e->scratch;
    pass[k].view = dint_seg_view(it.n_seg, it.seg_cap, it.seg_stride, it.d_cnt, it.cnt_stride);
    if (next) {
      npass[k] = pass[k];
      npass[k].d_req = next[k].d_base; npass[k].d_rep = next[k].d_base;
      npass[k].view = dint_seg_view(it.n_seg, it.seg_cap, it.seg_stride, next[k].d_cnt, it.cnt_stride);
    }
  }
  dint_launch_kv_multi(pass, n_items, st, part_done, next ? npass : nullptr);
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) return fail(DINT_EHIP, "kernel launch: %s", hipGetErrorString(err));
  for (uint32_t k = 0; k < n_items; k++) {
    dint_engine *e = items[k].engine;
    if (int rc = mark_stream(e, st)) return rc;
    e->batches++;
    e->requests += pass[k].n;
    if (next) {
      e->ahead.valid = true;
      e->ahead.req = next[k].d_base; e->ahead.rep = next[k].d_base; e->ahead.n = pass[k].n;
      e->ahead.seg_cap = items[k].seg_cap; e->ahead.n_seg = items[k].n_seg; e->ahead.cnt = next[k].d_cnt;
    }
  }
  return 0;
}

// ---- host buffers: pipelined H2D / kernels / D2H ---------------------------------------------------------------
namespace {
// enqueue chunk [off, off + m) of a host submission; returns its sequence number in *seq
// is [p, p + bytes) page-locked memory the HIP runtime knows (dint_alloc_pinned, hipHostMalloc, hipHostRegister)?
bool host_range_pinned(const void *p, size_t bytes) {
  if (bytes == 0) return true;
  static const bool no_bounce = getenv("DINT_NO_BOUNCE") != nullptr;  // diagnostic (tools/stress_pageable.py): r04's direct copies
  if (no_bounce) return true;
  const uint8_t *ends[2] = {(const uint8_t *)p, (const uint8_t *)p + bytes - 1};
  for (const uint8_t *q : ends) {
    hipPointerAttribute_t a;
    memset(&a, 0, sizeof a);
    if (hipPointerGetAttributes(&a, q) != hipSuccess) {
      (void)hipGetLastError();  // an unregistered pointer is an error to older runtimes: not ours
      return false;
    }
    if (a.type != hipMemoryTypeHost) return false;  // hipMemoryTypeUnregistered (plain memory), or not host memory at all
  }
  return true;
}
// a DEVICE pointer handed to the host path (dint_submit / dint_submit_async): the staging memcpy would read it on the CPU
bool is_device_ptr(const void *p) {
  if (!p) return false;
  hipPointerAttribute_t a;
  memset(&a, 0, sizeof a);
  if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
  return a.type == hipMemoryTypeDevice;
}
// copy a delivered chunk's replies out of the slot's bounce buffer (the chunk has left the GPU: sl.done was waited for)
void slot_deliver(dint_engine::Slot &sl) {
  if (sl.deliver_to) memcpy(sl.deliver_to, sl.h_rep, sl.deliver_bytes);
  sl.deliver_to = nullptr;
  sl.deliver_bytes = 0;
}
int slot_bounce_alloc(dint_engine *e, dint_engine::Slot &sl, bool req, bool rep) {
  const size_t bytes = (size_t)e->pass_max * e->msg_size + 64;
  if (req && !sl.h_req && hipHostMalloc((void **)&sl.h_req, bytes, hipHostMallocDefault) != hipSuccess)
    return fail(DINT_ENOMEM, "hipHostMalloc(%zu) for the staging of a pageable request buffer", bytes);
  if (rep && !sl.h_rep && hipHostMalloc((void **)&sl.h_rep, bytes, hipHostMallocDefault) != hipSuccess)
    return fail(DINT_ENOMEM, "hipHostMalloc(%zu) for the staging of a pageable reply buffer", bytes);
  return 0;
}
int enqueue_chunk(dint_engine *e, const uint8_t *rq, uint8_t *rp, uint32_t m, uint64_t *seq, bool rq_pinned, bool rp_pinned) {
  const uint64_t sq = e->next_seq;
  const int k = (int)(sq % dint_engine::kNSlot);
  if (int rc = slot_alloc(e, k)) return rc;
  if (!e->s_h2d) {
    if (!(e->cfg.flags & DINT_FLAG_COPY_STREAMS)) {
      // default: copies on the engine's own stream.  Measured on MI355X (same-box A/B, r02; NOTEBOOK.md): with three engines
      // in one process, two more streams per engine push the process past HIP's 4 hardware queues and -- depending on
      // which queues end up shared -- serialise the engines' kernel chains (TATP replay 1.2 instead of 1.7 G txn/s).
      // Several engines overlap each other's copies anyway; one engine alone wants DINT_FLAG_COPY_STREAMS.
      e->s_h2d = e->s_d2h = e->stream;
    } else {
      HIP_TRY(hipStreamCreateWithFlags(&e->s_h2d, hipStreamNonBlocking));
      HIP_TRY(hipStreamCreateWithFlags(&e->s_d2h, hipStreamNonBlocking));
    }
  }
  dint_engine::Slot &sl = e->slot[k];
  if (sl.seq) HIP_TRY(hipEventSynchronize(sl.done));  // the slot's previous chunk has left the GPU
  slot_deliver(sl);                                    // ... and, for a pageable caller, reaches its reply buffer now
  const size_t bytes = (size_t)m * e->msg_size;
  if (int rc = slot_bounce_alloc(e, sl, !rq_pinned, !rp_pinned)) return rc;
  if (!rq_pinned) {
    memcpy(sl.h_req, rq, bytes);
    rq = sl.h_req;
  }
  HIP_TRY(hipMemcpyAsync(sl.d_req, rq, bytes, hipMemcpyHostToDevice, e->s_h2d));
  HIP_TRY(hipEventRecord(sl.h2d, e->s_h2d));
  HIP_TRY(hipStreamWaitEvent(e->stream, sl.h2d, 0));
  if (int rc = run_pass(e, sl.d_req, m, sl.d_rep, e->stream)) return rc;
  HIP_TRY(hipEventRecord(sl.comp, e->stream));
  HIP_TRY(hipStreamWaitEvent(e->s_d2h, sl.comp, 0));
  if (!rp_pinned) {
    sl.deliver_to = rp;
    sl.deliver_bytes = bytes;
    rp = sl.h_rep;
  }
  HIP_TRY(hipMemcpyAsync(rp, sl.d_rep, bytes, hipMemcpyDeviceToHost, e->s_d2h));
  if (e->kv.n_tables)  // the overflow-pool counter travels with the replies (see dint_wait)
    HIP_TRY(hipMemcpyAsync(e->h_pool, &e->scratch.stats->pool_exhausted, sizeof(unsigned long long), hipMemcpyDeviceToHost, e->s_d2h));
  HIP_TRY(hipEventRecord(sl.done, e->s_d2h));
  sl.seq = sq;
  e->next_seq++;
  *seq = sq;
  return 0;
}
int wait_seq(dint_engine *e, uint64_t seq) {
  if (seq == 0 || seq >= e->next_seq) return fail(DINT_EINVAL, "unknown ticket");
  dint_engine::Slot &sl = e->slot[seq % dint_engine::kNSlot];
  if (sl.seq == seq) HIP_TRY(hipEventSynchronize(sl.done));  // else: the slot was reused, so the chunk finished long ago
  // replies of `seq` and of every earlier chunk are complete: the ones that went through a bounce buffer reach the
  // caller's memory here (chunks are enqueued, and leave the GPU, in sequence order)
  for (auto &o : e->slot)
    if (o.deliver_to && o.seq <= seq) {
      if (o.seq != seq) HIP_TRY(hipEventSynchronize(o.done));
      slot_deliver(o);
    }
  if (*e->h_pool > e->pool_seen) {
    const unsigned long long lost = *e->h_pool - e->pool_seen;
    e->pool_seen = *e->h_pool;
    return fail(DINT_ENOMEM, "%llu INSERTs found the overflow-entry pool full (dint_config.pool_entries); "
                             "they were answered with a reject code and stored nothing", lost);
  }
  return 0;
}
}  // namespace

int dint_submit_async(dint_engine_t *e, const void *reqs, uint32_t n, void *replies, dint_ticket *ticket) {
  if (!e || !ticket || (n && (!reqs || !replies))) return fail(DINT_EINVAL, "null argument");
  std::lock_guard<std::mutex> lk(e->mu);
  HIP_TRY(hipSetDevice(e->device));
  const uint8_t *rq = (const uint8_t *)reqs;
  uint8_t *rp = (uint8_t *)replies;
  uint64_t seq = e->next_seq - 1;  // n == 0: the ticket of whatever was submitted last
  if (n && (is_device_ptr(rq) || is_device_ptr(rp)))
    return fail(DINT_EINVAL, "dint_submit / dint_submit_async take HOST buffers; device memory goes to dint_submit_device");
  const bool rq_pinned = host_range_pinned(rq, (size_t)n * e->msg_size), rp_pinned = host_range_pinned(rp, (size_t)n * e->msg_size);
  for (uint32_t off = 0; off < n; off += e->pass_max) {
    const uint32_t m = std::min<uint32_t>(e->pass_max, n - off);
    if (int rc = enqueue_chunk(e, rq + (size_t)off * e->msg_size, rp + (size_t)off * e->msg_size, m, &seq, rq_pinned, rp_pinned)) return rc;
  }
  *ticket = seq;
  return 0;
}

int dint_wait(dint_engine_t *e, dint_ticket ticket) {
  if (!e) return fail(DINT_EINVAL, "null engine");
  std::lock_guard<std::mutex> lk(e->mu);
  HIP_TRY(hipSetDevice(e->device));
  if (ticket == 0) return 0;  // nothing was ever submitted
  return wait_seq(e, ticket);
}

int dint_submit(dint_engine_t *e, const void *reqs, uint32_t n, void *replies) {
  dint_ticket t = 0;
  if (int rc = dint_submit_async(e, reqs, n, replies, &t)) return rc;
  return n ? dint_wait(e, t) : 0;
}

int dint_alloc_pinned(size_t bytes, void **out) {
  if (!out) return fail(DINT_EINVAL, "null argument");
  if (hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) return fail(DINT_ENOMEM, "hipHostMalloc(%zu)", bytes);
  return 0;
}
void dint_free_pinned(void *p) {
  if (p) hipHostFree(p);
}

// ---- multi-GPU routing ---------------------------------------------------------------------------------------
namespace {
// validate one batch of a routing call, make sure the engine's routing scratch exists, order the stream, fill the job
int route_job(const dint_route_item &it, bool pack, uint64_t cnt_stride, hipStream_t &st, void *stream, dint_route_job *job) {
  dint_engine *e = it.engine;
  if (!e) return fail(DINT_EINVAL, "null engine");
  if (pack && (!it.d_slots || !it.d_cnt || (it.n && (!it.d_reqs || !it.d_slot)))) return fail(DINT_EINVAL, "null argument");
  if (!pack && it.n && (!it.d_slots || !it.d_slot || !it.d_reqs || !it.d_replies)) return fail(DINT_EINVAL, "null argument");
  if (e->cfg.workload == DINT_WL_LOG) return fail(DINT_ESTATE, "workload is not sharded by key");
  if (it.n > DINT_ROUTE_MAXN) return fail(DINT_EINVAL, "at most %u requests per routed batch", DINT_ROUTE_MAXN);
  if (e->shard.count > DINT_ROUTE_MAXW) return fail(DINT_EINVAL, "at most %u shards can be routed to", DINT_ROUTE_MAXW);
  if (it.seg_cap == 0 || (uint64_t)it.seg_cap * e->shard.count > 0xFFFFFFF0ull) return fail(DINT_EINVAL, "bad seg_cap");
  HIP_TRY(hipSetDevice(e->device));
  if (!st) st = stream ? (hipStream_t)stream : e->stream;
  if (pack) {
    if (!e->route.blk) {
      if (int rc = dev_alloc((void **)&e->route.blk, (size_t)2 * DINT_ROUTE_BLK_WORDS * 4)) return rc;
      e->route.blk_next = e->route.blk + DINT_ROUTE_BLK_WORDS;
      HIP_TRY(hipDeviceSynchronize());  // the zero-fill ran on the null stream, which a caller's non-blocking stream does not wait for
    }
    if (int rc = order_route_stream(e, st)) return rc;  // the routing scratch is per engine, and only the routing kernels use it
  }
  job->workload = e->cfg.workload;
  job->msg = e->msg_size;
  job->slots = e->slots_mod;
  job->kv = &e->kv;
  job->shard = e->shard;
  job->d_req = it.d_reqs;
  job->d_rep = it.d_replies;
  job->n = it.n;
  job->d_n = it.d_n;
  job->cap = it.seg_cap;
  job->d_send = it.d_slots;
  job->d_cnt = it.d_cnt;
  job->cnt_stride = cnt_stride;
  job->d_slot = it.d_slot;
  job->rs = e->route;
  job->stats = e->scratch.stats;
  return 0;
}
// every engine of a routing call, each once, locked in address order (two calls that share engines cannot deadlock)
int lock_engines(const dint_route_item *items, uint32_t n_items, std::vector<std::unique_lock<std::mutex>> &locks) {
  std::vector<dint_engine *> es;
  for (uint32_t k = 0; k < n_items; k++) {
    if (!items[k].engine) return fail(DINT_EINVAL, "null engine");
    if (items[k].engine->device != items[0].engine->device) return fail(DINT_EINVAL, "the batches of one call share a device");
    es.push_back(items[k].engine);
  }
  std::sort(es.begin(), es.end());
  es.erase(std::unique(es.begin(), es.end()), es.end());
  for (dint_engine *e : es) locks.emplace_back(e->mu);
  return 0;
}
}  // namespace

int dint_route_pack_multi(const dint_route_item *items, uint32_t n_items, uint64_t seg_stride, uint64_t cnt_stride,
                          void *stream) {
  if (!items || n_items == 0 || n_items > DINT_ROUTE_MAXS) return fail(DINT_EINVAL, "1 .. %u batches per call", DINT_ROUTE_MAXS);
  dint_route_job jobs[DINT_ROUTE_MAXS];
  hipStream_t st = nullptr;
  std::vector<std::unique_lock<std::mutex>> locks;
  if (int rc = lock_engines(items, n_items, locks)) return rc;  // held across the launch: two threads packing with
  if (locks.size() != n_items) return fail(DINT_EINVAL, "an engine routes one batch per call (its routing scratch is one batch's)");
  for (uint32_t k = 0; k < n_items; k++)                        // one engine launch in the order order_route_stream recorded
    if (int rc = route_job(items[k], true, cnt_stride, st, stream, &jobs[k])) return rc;
  dint_launch_route_pack(jobs, n_items, seg_stride, st);
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) {
    // some of the per-format launches may be queued already: their tickets and tile words would be left dirty in a
    // copy the engines still point at.  Drain the stream and hand every engine two clean copies again.
    hipStreamSynchronize(st);
    for (uint32_t k = 0; k < n_items; k++) {
      dint_engine *e = items[k].engine;
      hipMemset(std::min(e->route.blk, e->route.blk_next), 0, (size_t)2 * DINT_ROUTE_BLK_WORDS * 4);
      mark_route_stream(e, st);
    }
    hipDeviceSynchronize();
    return fail(DINT_EHIP, "kernel launch: %s", hipGetErrorString(err));
  }
  for (uint32_t k = 0; k < n_items; k++) {
    dint_engine *e = items[k].engine;
    std::swap(e->route.blk, e->route.blk_next);  // the launch left the other copy of the scratch zeroed for the next call
    if (int rc = mark_route_stream(e, st)) return rc;
  }
  return 0;
}

int dint_route_unpack_multi(const dint_route_item *items, uint32_t n_items, uint64_t seg_stride, void *stream) {
  if (!items || n_items == 0 || n_items > DINT_ROUTE_MAXS) return fail(DINT_EINVAL, "1 .. %u batches per call", DINT_ROUTE_MAXS);
  dint_route_job jobs[DINT_ROUTE_MAXS];
  hipStream_t st = nullptr;
  std::vector<std::unique_lock<std::mutex>> locks;
  if (int rc = lock_engines(items, n_items, locks)) return rc;
  for (uint32_t k = 0; k < n_items; k++)
    if (int rc = route_job(items[k], false, 0, st, stream, &jobs[k])) return rc;
  dint_launch_route_unpack(jobs, n_items, seg_stride, st);
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) return fail(DINT_EHIP, "kernel launch: %s", hipGetErrorString(err));
  return 0;
}

int dint_route_pack(dint_engine_t *e, const void *d_reqs, uint32_t n, void *d_send, uint32_t seg_cap,
                    uint64_t seg_stride, void *d_cnt, uint64_t cnt_stride, uint32_t *d_slot, void *stream) {
  dint_route_item it = {e, d_reqs, n, seg_cap, d_send, d_cnt, d_slot, nullptr, nullptr};
  return dint_route_pack_multi(&it, 1, seg_stride, cnt_stride, stream);
}

int dint_route_unpack(dint_engine_t *e, const void *d_back, uint32_t seg_cap, uint64_t seg_stride,
                      const uint32_t *d_slot, const void *d_reqs, uint32_t n, void *d_replies, void *stream) {
  dint_route_item it = {e, d_reqs, n, seg_cap, const_cast<void *>(d_back), nullptr, const_cast<uint32_t *>(d_slot), d_replies, nullptr};
  return dint_route_unpack_multi(&it, 1, seg_stride, stream);
}

void *dint_engine_stream(dint_engine_t *e) { return e ? (void *)e->stream : nullptr; }
uint32_t dint_max_pass(dint_engine_t *e) { return e ? e->pass_max : 0; }

int dint_stream_wait(dint_engine_t *e, void *other_stream) {
  if (!e) return fail(DINT_EINVAL, "null engine");
  std::lock_guard<std::mutex> lk(e->mu);
  HIP_TRY(hipSetDevice(e->device));
  // one event per engine and direction: a wait captures the state the event had when the wait was enqueued, so
  // recording it again for the next call does not disturb waits that are still pending
  if (!e->ev_wait) HIP_TRY(hipEventCreateWithFlags(&e->ev_wait, hipEventDisableTiming));
  HIP_TRY(hipEventRecord(e->ev_wait, (hipStream_t)other_stream));
  HIP_TRY(hipStreamWaitEvent(e->stream, e->ev_wait, 0));
  return 0;
}

int dint_stream_signal(dint_engine_t *e, void *other_stream) {
  if (!e) return fail(DINT_EINVAL, "null engine");
  std::lock_guard<std::mutex> lk(e->mu);
  HIP_TRY(hipSetDevice(e->device));
  if (!e->ev_signal) HIP_TRY(hipEventCreateWithFlags(&e->ev_signal, hipEventDisableTiming));
  HIP_TRY(hipEventRecord(e->ev_signal, e->stream));
  HIP_TRY(hipStreamWaitEvent((hipStream_t)other_stream, e->ev_signal, 0));
  return 0;
}

int dint_sync(dint_engine_t *e) {
  if (!e) return fail(DINT_EINVAL, "null engine");
  hipStream_t s0, s1, s2;
  {
    std::lock_guard<std::mutex> lk(e->mu);  // the copy streams are created lazily by the first host submission
    s0 = e->stream; s1 = e->s_h2d; s2 = e->s_d2h;
  }
  HIP_TRY(hipSetDevice(e->device));
  HIP_TRY(hipStreamSynchronize(s0));
  if (s1 && s1 != s0) HIP_TRY(hipStreamSynchronize(s1));
  if (s2 && s2 != s0) HIP_TRY(hipStreamSynchronize(s2));
  // replies of pageable callers that still sit in a slot's page-locked buffer reach the caller's memory here as well (ADVICE r05:
  // a dint_submit_async + dint_sync caller never calls dint_wait)
  std::lock_guard<std::mutex> lk(e->mu);
  for (auto &sl : e->slot)
    if (sl.deliver_to) {
      HIP_TRY(hipEventSynchronize(sl.done));
      slot_deliver(sl);
    }
  return 0;
}

int64_t dint_read_locks(dint_engine_t *e, uint32_t table, uint32_t *a, uint32_t *b, uint64_t cap) {
  if (!e) return fail(DINT_EINVAL, "null engine");
  std::lock_guard<std::mutex> lk(e->mu);
  HIP_TRY(hipSetDevice(e->device));
  HIP_TRY(hipDeviceSynchronize());
  if (e->cfg.workload == DINT_WL_FASST || e->cfg.workload == DINT_WL_2PL) {
    uint64_t n = std::min<uint64_t>(cap, e->n_local_slots);
    std::vector<uint2> h(n);
    HIP_TRY(hipMemcpy(h.data(), e->d_lock_tbl, n * sizeof(uint2), hipMemcpyDeviceToHost));
    for (uint64_t i = 0; i < n; i++) {
      if (a) a[i] = h[i].x;
      if (b) b[i] = h[i].y;
    }
    return (int64_t)e->n_local_slots;
  }
  if (e->cfg.workload == DINT_WL_TATP || e->cfg.workload == DINT_WL_SMALLBANK) {
    int64_t r = dint_kv_read_locks(&e->kv, table, a, b, cap);
    if (r < 0) return fail((int)r, "bad table %u", table);
    return r;
  }
  return fail(DINT_ESTATE, "workload has no lock table");
}

int64_t dint_read_log(dint_engine_t *e, void *records, uint64_t cap) {
  if (!e) return fail(DINT_EINVAL, "null engine");
  if (!e->log.ring) return fail(DINT_ESTATE, "workload has no log");
  std::lock_guard<std::mutex> lk(e->mu);
  HIP_TRY(hipSetDevice(e->device));
  HIP_TRY(hipDeviceSynchronize());
  uint64_t n = std::min<uint64_t>(cap, e->log.cap);
  if (records && n) HIP_TRY(hipMemcpy(records, e->log.ring, n * 64, hipMemcpyDeviceToHost));
  uint32_t t[2];
  HIP_TRY(hipMemcpy(t, e->log.tail, sizeof t, hipMemcpyDeviceToHost));
  return (int64_t)t[log_cur(e)];
}

int64_t dint_log_drain(dint_engine_t *e, void *records, uint64_t cap, uint64_t *lost) {
  if (!e || (cap && !records)) return fail(DINT_EINVAL, "null argument");
  if (!e->log.ring) return fail(DINT_ESTATE, "workload has no log");
  std::lock_guard<std::mutex> lk(e->mu);
  HIP_TRY(hipSetDevice(e->device));
  HIP_TRY(hipDeviceSynchronize());
  uint32_t t[4];
  HIP_TRY(hipMemcpy(t, e->log.tail, sizeof t, hipMemcpyDeviceToHost));
  const uint64_t total = (uint64_t)t[2] | ((uint64_t)t[3] << 32);
  if (e->log_drained > total) e->log_drained = total;  // the engine was reset / restored to an earlier state
  uint64_t pending = total - e->log_drained, gone = 0;
  if (pending > e->log.cap) {  // the ring lapped the reader: the oldest records are overwritten
    gone = pending - e->log.cap;
    pending = e->log.cap;
  }
  if (lost) *lost = gone;
  e->log_drained += gone;
  const uint64_t n = std::min<uint64_t>(pending, cap);
  uint64_t pos = e->log_drained % e->log.cap;  // record k of the stream lives at ring slot k % cap
  uint8_t *out = (uint8_t *)records;
  for (uint64_t done = 0; done < n;) {
    const uint64_t run = std::min<uint64_t>(n - done, e->log.cap - pos);
    HIP_TRY(hipMemcpy(out + done * 64, e->log.ring + pos * 64, run * 64, hipMemcpyDeviceToHost));
    done += run;
    pos = (pos + run) % e->log.cap;
  }
  e->log_drained += n;
  return (int64_t)n;
}

int dint_refuse(uint32_t workload, const void *reqs, uint32_t n, void *replies) {
  if (workload >= DINT_WL_COUNT || (n && (!reqs || !replies))) return fail(DINT_EINVAL, "bad argument");
  const uint32_t msg = kMsgSize[workload];
  const uint8_t *rq = (const uint8_t *)reqs;
  uint8_t *rp = (uint8_t *)replies;
  if (rp != rq) memcpy(rp, rq, (size_t)n * msg);
  for (uint32_t i = 0; i < n; i++) {
    uint8_t *m = rp + (size_t)i * msg;
    switch (workload) {
      case DINT_WL_2PL:  // RETRY, lock_2pl/ebpf/ls_kern.c:59-64
        m[0] = 4;
        break;
      case DINT_WL_STORE:  // kRejectRead / kRejectSet / kRejectInsert, store/ebpf/store_kern.c:57-62
        if (m[0] <= 2) m[0] = (uint8_t)(4 + 2 * m[0] + (m[0] == 2));  // 0 -> 4, 1 -> 6, 2 -> 9
        break;
      case DINT_WL_TATP:  // REJECT_READ / REJECT_LOCK / REJECT_COMMIT, tatp/ebpf/shard_kern.c:173-178,289-293,371-376
        if (m[1] == 0) m[1] = 5;
        else if (m[1] == 1) m[1] = 8;
        else if (m[1] == 12 || m[1] == 13 || m[1] == 18 || m[1] == 19 || m[1] == 22 || m[1] == 23) m[1] = 11;
        break;  // ABORT and the log appends are never refused by the eBPF server either
      case DINT_WL_SMALLBANK:  // RETRY, smallbank/ebpf/shard_kern.c:96-110,603-608
        if (m[1] <= 5 || m[1] == 17) m[1] = 16;
        break;
      default:  // lock_fasst: REJECT_LOCK is the only "not now" a client understands; the log server has none
        if (workload == DINT_WL_FASST && m[0] == 1) m[0] = 6;
        break;
    }
  }
  return 0;
}

int dint_get_stats(dint_engine_t *e, dint_stats *out) {
  if (!e || !out) return fail(DINT_EINVAL, "null argument");
  std::lock_guard<std::mutex> lk(e->mu);
  HIP_TRY(hipSetDevice(e->device));
  HIP_TRY(hipDeviceSynchronize());
  dint_dev_stats d;
  HIP_TRY(hipMemcpy(&d, e->scratch.stats, sizeof d, hipMemcpyDeviceToHost));
  memset(out, 0, sizeof *out);
  out->batches = e->batches;
  out->requests = e->requests;
  out->bad_requests = d.bad_requests;
  out->missing_keys = d.missing_keys;
  out->foreign_requests = d.foreign_requests;
  out->pool_exhausted = d.pool_exhausted;
  out->route_overflow = d.route_overflow;
  out->big_bin_requests = d.big_bin_requests;
  out->late_requests = d.late_requests;
  for (int k = 0; k < 3; k++) out->reserved[k] = d.late_items[k];  // (diagnostic: late work items by kind -- sub / solo / pieces)
  return 0;
}

int dint_reset(dint_engine_t *e) {
  if (!e) return fail(DINT_EINVAL, "null engine");
  std::lock_guard<std::mutex> lk(e->mu);
  HIP_TRY(hipSetDevice(e->device));
  HIP_TRY(hipDeviceSynchronize());
  if (int rc = ahead_cancel(e)) return rc;
  for (auto &r : e->regions) HIP_TRY(hipMemset(r.first, 0, r.second));
  e->batches = e->requests = 0;
  e->log_drained = 0;
  e->pool_seen = 0;
  *e->h_pool = 0;
  HIP_TRY(hipDeviceSynchronize());
  return 0;
}

int dint_snapshot(dint_engine_t *e) {
  if (!e) return fail(DINT_EINVAL, "null engine");
  std::lock_guard<std::mutex> lk(e->mu);
  HIP_TRY(hipSetDevice(e->device));
  HIP_TRY(hipDeviceSynchronize());
  if (e->ahead.valid) return fail(DINT_ESTATE, "a batch announced by dint_submit_device_ahead is pending (its log records are appended)");
  if (e->kv.n_tables && e->log.tail) {  // both tail words say where the ring stands: whatever the parity of the passes after a restore
    uint32_t t[2];
    HIP_TRY(hipMemcpy(t, e->log.tail, sizeof t, hipMemcpyDeviceToHost));
    t[0] = t[1] = t[log_cur(e)];
    HIP_TRY(hipMemcpy(e->log.tail, t, sizeof t, hipMemcpyHostToDevice));
  }
  if (e->snap.empty()) {
    for (auto &r : e->regions) {
      void *p = nullptr;
      int rc = dev_alloc(&p, r.second, false);
      if (rc) return rc;
      e->snap.push_back(p);
    }
  }
  for (size_t i = 0; i < e->regions.size(); i++)
    HIP_TRY(hipMemcpy(e->snap[i], e->regions[i].first, e->regions[i].second, hipMemcpyDeviceToDevice));
  HIP_TRY(hipDeviceSynchronize());
  e->snap_log_drained = e->log_drained;
  return 0;
}

int dint_restore(dint_engine_t *e) {
  if (!e) return fail(DINT_EINVAL, "null engine");
  std::lock_guard<std::mutex> lk(e->mu);
  if (e->snap.empty()) return fail(DINT_ESTATE, "no snapshot taken");
  HIP_TRY(hipSetDevice(e->device));
  HIP_TRY(hipDeviceSynchronize());
  if (int rc = ahead_cancel(e)) return rc;
  for (size_t i = 0; i < e->regions.size(); i++)
    HIP_TRY(hipMemcpy(e->regions[i].first, e->snap[i], e->regions[i].second, hipMemcpyDeviceToDevice));
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(e->h_pool, &e->scratch.stats->pool_exhausted, sizeof(unsigned long long), hipMemcpyDeviceToHost));
  e->pool_seen = *e->h_pool;
  e->log_drained = e->snap_log_drained;  // the drain cursor belongs to the log's history
  return 0;
}

int dint_load_rows(dint_engine_t *e, uint32_t table, const uint64_t *keys, const uint32_t *vers, const void *vals,
                   uint64_t n) {
  if (!e || (n && (!keys || !vals))) return fail(DINT_EINVAL, "null argument");
  if (!e->kv.n_tables) return fail(DINT_ESTATE, "workload has no kv table");
  if (table >= e->kv.n_tables) return fail(DINT_EINVAL, "bad table %u", table);
  std::lock_guard<std::mutex> lk(e->mu);
  HIP_TRY(hipSetDevice(e->device));
  return load_rows_locked(e, table, keys, vers, (const uint8_t *)vals, n);
}

int dint_populate(dint_engine_t *e, uint64_t populate_n) {
  if (!e) return fail(DINT_EINVAL, "null engine");
  if (!e->kv.n_tables) return fail(DINT_ESTATE, "workload has no kv table");
  std::lock_guard<std::mutex> lk(e->mu);
  HIP_TRY(hipSetDevice(e->device));
  return dint_pop::generate(e->cfg.workload, populate_n,
                            [e](uint32_t table, const uint64_t *keys, const uint8_t *vals, uint64_t n) {
                              return load_rows_locked(e, table, keys, nullptr, vals, n);
                            });
}

int64_t dint_hash_size(dint_engine_t *e, uint32_t table) {
  if (!e) return fail(DINT_EINVAL, "null engine");
  if (table >= e->kv.n_tables) return fail(DINT_EINVAL, "bad table %u", table);
  return (int64_t)e->kv.hash_size[table];
}

int64_t dint_dump_rows(dint_engine_t *e, uint32_t table, uint64_t *keys, uint32_t *vers, void *vals, uint64_t cap) {
  if (!e) return fail(DINT_EINVAL, "null engine");
  if (table >= e->kv.n_tables) return fail(DINT_EINVAL, "bad table %u", table);
  std::lock_guard<std::mutex> lk(e->mu);
  HIP_TRY(hipSetDevice(e->device));
  HIP_TRY(hipDeviceSynchronize());
  return dint_kv_dump_rows(&e->kv, table, keys, vers, (uint8_t *)vals, cap);
}

int dint_home_shard(dint_engine_t *e, const void *d_reqs, uint32_t n, uint8_t *d_home, void *stream) {
  if (!e || (n && (!d_reqs || !d_home))) return fail(DINT_EINVAL, "null argument");
  std::lock_guard<std::mutex> lk(e->mu);
  HIP_TRY(hipSetDevice(e->device));
  hipStream_t st = stream ? (hipStream_t)stream : e->stream;
  switch (e->cfg.workload) {
    case DINT_WL_FASST:
    case DINT_WL_2PL:
      dint_launch_home_lid(d_reqs, e->msg_size, n, e->slots_mod, e->shard.count, d_home, st);
      break;
    case DINT_WL_STORE:
    case DINT_WL_TATP:
    case DINT_WL_SMALLBANK:
      dint_launch_home_kv(d_reqs, n, e->kv, d_home, st);
      break;
    default:
      return fail(DINT_ESTATE, "workload is not sharded by key");
  }
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) return fail(DINT_EHIP, "kernel launch: %s", hipGetErrorString(err));
  return 0;
}

int dint_kv_trace_read(dint_engine_t *e, uint64_t *out, uint64_t cap) {
  if (!e || !out) return fail(DINT_EINVAL, "null argument");
  if (!e->kv.d_trace) return fail(DINT_ESTATE, "tracing is off (set DINT_KV_TRACE=1 before creating the engine)");
  std::lock_guard<std::mutex> lk(e->mu);
  HIP_TRY(hipSetDevice(e->device));
  HIP_TRY(hipDeviceSynchronize());
  const uint64_t words = std::min<uint64_t>(cap, (uint64_t)DINT_KV_TRACE_WORDS);
  HIP_TRY(hipMemcpy(out, e->kv.d_trace, words * 8, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemset(e->kv.d_trace, 0, DINT_KV_TRACE_WORDS * 8));  // the next read sees one launch only
  HIP_TRY(hipDeviceSynchronize());  // (a fill on the null stream: the engine's streams do not wait for it)
  return (int)DINT_KV_PMAX;
}

int dint_timing_enable(dint_engine_t *e, int on) {
  if (!e) return fail(DINT_EINVAL, "null engine");
  std::lock_guard<std::mutex> lk(e->mu);
  e->timer.on = on != 0;
  e->timer.passes = 0;
  return 0;
}

int dint_timing_read(dint_engine_t *e, const char **names, double *avg_us, uint64_t *launches, int cap) {
  if (!e) return fail(DINT_EINVAL, "null engine");
  std::lock_guard<std::mutex> lk(e->mu);
  HIP_TRY(hipSetDevice(e->device));
  HIP_TRY(hipDeviceSynchronize());
  KernelTimer &t = e->timer;
  int nk = std::min(t.n_kernels, cap);
  for (int k = 0; k < nk; k++) {
    double sum = 0;
    for (int p = 0; p < t.passes; p++) {
      float ms = 0;
      hipEvent_t *ev = &t.ev[(size_t)p * (t.n_kernels + 1)];
      if (hipEventElapsedTime(&ms, ev[k], ev[k + 1]) == hipSuccess) sum += ms;
    }
    names[k] = t.names[k];
    avg_us[k] = t.passes ? sum * 1000.0 / t.passes : 0.0;
    launches[k] = (uint64_t)t.passes;
  }
  t.passes = 0;
  return nk;
}

}  // extern "C"
