// k_txn.hip -- the GPU-resident closed-loop transaction driver (include/dint_driver.h dint_gdriver_*).
//
// The callers of the hot path -- the reference's TATP / SmallBank clients, tatp/caladan/client_udp_shard.cc:177-1185,
// smallbank/caladan/client_udp_shard.cc:169-1240 -- as device code: the SAME state machines the host driver runs
// (txn_clients.h is compiled for both), one lane per client, client state resident in HBM.  No NIC can offer the
// > 10^9 requests/s one MI355X absorbs, and a host-side generator caps the loop at PCIe speed; with the clients on
// the GPU the closed loop {emit -> three shard servers -> consume} never leaves the device.
//
//   k_txn_emit    : every client runs one phase (finishing / starting transactions on the way) and emits its <= 9
//                   messages into the three per-shard request arrays at EXACTLY the positions the host driver uses
//                   -- batch s is ordered by client id, then send order -- so the request stream is bit-identical
//                   to the host driver's: position = messages of smaller client ids to that shard (a workgroup scan
//                   + a decoupled look-back over the workgroups before mine, which are handed out by ticket and
//                   publish their totals first thing, their inclusive prefixes as soon as they know them) + msg.ord.
//                   The last workgroup writes the three batch sizes for the engines (dint_submit_segments* read them
//                   on the device: no host round trip).
//   k_txn_consume : every client takes what it needs of the replies it waits for out of the (in place) reply arrays:
//                   type / version / first value byte into its header, the whole reply where the transaction keeps the
//                   row (tx_consume_one).  When consume and the next emit are issued on the same stream (the closed
//                   loop does) this is FUSED into k_txn_emit instead: the epoch's batches alternate between two buffer
//                   sets, so a client reads the replies of epoch k from one set while the requests of epoch k+1 are
//                   written into the other -- one kernel and one header load per epoch less.
#include <hip/hip_runtime.h>

#include <new>
#include <vector>

#include "../../include/dint_abi.h"
#include "../../include/dint_driver.h"
#include "dint_device.h"
#include "txn_clients.h"

#define TXG_TB 512u   // clients per workgroup (256: 408 us per closed-loop epoch, 512: 392, 1024: 419 -- fewer tickets and look-back entries against fewer resident workgroups)
#define TXG_AGG 0x40000000u  // look-back word: the workgroup's own message count ...
#define TXG_PFX 0x80000000u  // ... or the count of all workgroups up to and including it
#define TXG_VAL 0x3FFFFFFFu

struct txg_stats {
  unsigned long long txns, committed, messages, by_type[8], committed_by_type[8], overflow;
};
#define TXG_NSTAT (sizeof(txg_stats) / 8)

void dint_driver_params(const dint_driver_config &c, TxParams *P, ZipfTable *zipf);  // txn_driver.cc

// The client headers live in HBM as dword COLUMNS (word j of client i at cols[j * n_clients + i]): the lanes of a wave
// are consecutive clients, so every load / store instruction of a header is one contiguous 256-byte run -- whole
// sectors, two cache lines per instruction.  (As an array of 128-byte structs every 16-byte piece of an instruction
// went to a line of its own: 64 lines per instruction, and partial-sector writes on the way back.)
// The words of `m` (where the client's working messages live: set by the kernel on every load) do not travel.
template <class C> using c_m_t = decltype(C::m);
template <class C>
__device__ static inline bool txg_is_m(uint32_t j) { return j >= offsetof(C, m) / 4 && j < (offsetof(C, m) + sizeof(c_m_t<C>)) / 4; }
template <class C>
__device__ static inline void txg_load_client(C &c, const uint32_t *cols, uint32_t n_clients, uint32_t i) {
  static_assert(sizeof(C) % 4 == 0 && offsetof(C, m) % 4 == 0, "client header in dwords");
  uint32_t w[sizeof(C) / 4];
#pragma unroll
  for (uint32_t j = 0; j < sizeof(C) / 4; j++) w[j] = txg_is_m<C>(j) ? 0u : cols[(size_t)j * n_clients + i];
  __builtin_memcpy(&c, w, sizeof(C));
}
template <class C>
__device__ static inline void txg_store_client(const C &c, uint32_t *cols, uint32_t n_clients, uint32_t i) {
  uint32_t w[sizeof(C) / 4];
  __builtin_memcpy(w, &c, sizeof(C));
#pragma unroll
  for (uint32_t j = 0; j < sizeof(C) / 4; j++)
    if (!txg_is_m<C>(j)) cols[(size_t)j * n_clients + i] = w[j];
}

// W = waves per SIMD the kernel is compiled for (the register budget: 3 -> 168, 4 -> 128, 5 -> 96 VGPRs; DINT_TXN_WAVES)
template <class T, int W>
__global__ void __launch_bounds__(TXG_TB, W)
k_txn_emit(uint32_t *cl, uint8_t *store, uint32_t n_clients, TxParams P, uint8_t *out0, uint8_t *out1,
           uint8_t *out2, const uint8_t *rep0, const uint8_t *rep1, const uint8_t *rep2, uint32_t cap, uint32_t *pub,
           uint32_t *ticket, uint32_t *pub_other, uint32_t *ticket_other, uint32_t *counts, txg_stats *st, uint32_t dbg) {
  typedef typename T::Msg Msg;
  __shared__ uint32_t Stile, Sw[3][TXG_TB / 64];
  __shared__ unsigned long long Sst[TXG_NSTAT];
  const uint32_t t = threadIdx.x, lane = t & 63, wave = t >> 6;
  if (t == 0) Stile = atomicAdd(ticket, 1u);
  if (t < TXG_NSTAT) Sst[t] = 0;
  __syncthreads();
  const uint32_t tile = Stile, i = tile * TXG_TB + t, ntiles = gridDim.x;
  const bool valid = i < n_clients;
  // the look-back words and the ticket alternate between two sets: leave the other set clean for the next epoch (the
  // kernel that used it has finished: same stream)
  if (t < 4) pub_other[tile * 4 + t] = 0;
  if (tile == 0 && t == 0) *ticket_other = 0;

  // the client's header travels through registers: one 128 / 112-byte load here, one store at the end.  It carries
  // what the phase logic reads of the replies (tatp); the working messages stay in memory and are touched only where a
  // transaction keeps a whole row.  The phase's message queue lives in LDS (dynamically indexed byte arrays: left to
  // the compiler it went to scratch).
  __shared__ typename T::Out So[TXG_TB];
  typename T::Out &o = So[t];
  o.clear();
  typename T::Client c;
  if (valid) {
    txg_load_client(c, cl, n_clients, i);
    c.m.base = store + (size_t)i * TX_DEV_MSG_STRIDE;  // message k of every client is one array of sectors (TxMsgs)
    c.m.stride = (uint64_t)n_clients * TX_DEV_MSG_STRIDE;
    if (rep0) {  // fused consume: the replies of the previous epoch (the other buffer set), then the phase that reads them
      // The three fields of every awaited reply first, ALL loads in flight together (unconditional, from slot 0 of
      // the batch where there is nothing to read: one round trip instead of one per reply), then the summaries; whole
      // rows, where a transaction keeps them, behind that.
      const uint8_t n = c.n_out;
      const Msg *rp[T::MAXOUT];
      uint8_t r_ty[T::MAXOUT], r_v0[T::MAXOUT];
      uint32_t r_ver[T::MAXOUT];
#pragma unroll
      for (uint8_t k = 0; k < T::MAXOUT; k++) {  // unrolled: the header's arrays stay in registers
        const bool use = k < n && c.out_pos[k] < cap && c.out_dst[k] != TX_NO_DST;
        const uint8_t sh = use ? c.out_shard[k] : 0;
        const uint8_t *rb = sh == 0 ? rep0 : sh == 1 ? rep1 : rep2;
        rp[k] = (const Msg *)(rb + (size_t)(use ? c.out_pos[k] : 0u) * sizeof(Msg));
        r_ty[k] = rp[k]->type; r_v0[k] = rp[k]->val[0]; r_ver[k] = rp[k]->ver;
      }
#pragma unroll
      for (uint8_t k = 0; k < T::MAXOUT; k++) {
        const uint8_t d = c.out_dst[k];
        if (k < n && c.out_pos[k] < cap && d != TX_NO_DST) {
          c.note_reply((uint8_t)(d & TX_DST_MASK), r_ty[k], r_v0[k], r_ver[k]);
          if (d & TX_FULL) c.m[d & TX_DST_MASK] = *rp[k];
        }
      }
    }
    if (!(dbg & 2)) T::run(c, P, o);
  }
  uint32_t nmsg[3] = {0, 0, 0};
  for (uint8_t k = 0; k < o.n; k++) nmsg[o.shard[k]]++;

  // ---- my messages' positions: workgroup scan per shard ...
  uint32_t x[3], tot[3];
#pragma unroll
  for (int s = 0; s < 3; s++) {
    uint32_t wt;
    x[s] = wave_excl_scan_u32(nmsg[s], &wt);
    if (lane == 0) Sw[s][wave] = wt;
  }
  __syncthreads();
#pragma unroll
  for (int s = 0; s < 3; s++) {
    tot[s] = 0;
    for (uint32_t w = 0; w < TXG_TB / 64; w++) {
      if (w < wave) x[s] += Sw[s][w];
      tot[s] += Sw[s][w];
    }
  }
  // ---- ... + the messages of the workgroups before mine: decoupled look-back.  A workgroup publishes its own totals
  // first thing (TXG_AGG), then walks back over the workgroups before it until, per shard, it meets one that already
  // knows its inclusive prefix (TXG_PFX), and publishes its own.  Workgroups are handed out by ticket, so the ones a
  // look-back waits for are running.  The resident workgroups (hundreds) finish their phases at about the same time,
  // so a walk is hundreds of entries deep: all TXG_TB threads look at once (one entry each, its three words loaded
  // together), TXG_TB entries per step.  (r03a: one wave, one shard at a time, 64 entries per step -- 41 of the
  // kernel's 204 us, measured with DINT_TXN_DBG=1.  r02 summed ALL earlier totals in every workgroup.)
  __shared__ uint32_t Lsum[3][TXG_TB / 64], Lhit[3][TXG_TB / 64];
  if (t < 3) __hip_atomic_store(&pub[tile * 4 + t], TXG_AGG | tot[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  uint32_t base[3] = {0, 0, 0};
  uint32_t open = (tile == 0 || (dbg & 1)) ? 0u : 7u;  // shards whose prefix is not known yet (workgroup-uniform)
  for (int k0 = (int)tile - 1; open; k0 -= (int)TXG_TB) {
    const int k = k0 - (int)t;  // thread 0 looks at the nearest workgroup
    uint32_t v[3] = {TXG_PFX, TXG_PFX, TXG_PFX};  // before the first workgroup: prefix 0
    if (k >= 0) {
      do {
#pragma unroll
        for (int s = 0; s < 3; s++) v[s] = __hip_atomic_load(&pub[k * 4 + s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } while (!(v[0] >> 30) || !(v[1] >> 30) || !(v[2] >> 30));
    }
#pragma unroll
    for (int s = 0; s < 3; s++) {
      const uint64_t mp = __ballot((v[s] & TXG_PFX) != 0);  // lane 0 is the nearest of this wave's 64
      const int stop = mp ? __ffsll((unsigned long long)mp) - 1 : 64;
      uint32_t add = (int)lane <= stop ? v[s] & TXG_VAL : 0u;
      for (int d = 32; d > 0; d >>= 1) add += __shfl_xor(add, d, 64);
      if (lane == 0) { Lsum[s][wave] = add; Lhit[s][wave] = mp != 0; }
    }
    __syncthreads();
    uint32_t still = open;
#pragma unroll
    for (int s = 0; s < 3; s++) {
      if (!((open >> s) & 1u)) continue;
      for (uint32_t w = 0; w < TXG_TB / 64; w++) {  // wave 0 holds the nearest entries
        base[s] += Lsum[s][w];
        if (Lhit[s][w]) { still &= ~(1u << s); break; }
      }
    }
    open = still;
    __syncthreads();  // Lsum / Lhit are rewritten by the next step
  }
  if (t < 3) {
    const uint32_t b = t == 0 ? base[0] : t == 1 ? base[1] : base[2];
    const uint32_t mine = t == 0 ? tot[0] : t == 1 ? tot[1] : tot[2];
    __hip_atomic_store(&pub[tile * 4 + t], TXG_PFX | (b + mine), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tile == ntiles - 1) counts[t] = min(b + mine, cap);  // batch sizes of this epoch (read by the engines)
  }

  // ---- emit
  uint32_t lost = 0;
#pragma unroll
  for (uint8_t k = 0; k < T::MAXOUT; k++) {
    if (k < o.n) {
      const uint32_t s = o.shard[k], pos = (s == 0 ? base[0] + x[0] : s == 1 ? base[1] + x[1] : base[2] + x[2]) + o.ord[k];
      c.out_pos[k] = pos;
      const Msg m = o.materialize(c, k);
      if (pos < cap) {
        *(Msg *)((s == 0 ? out0 : s == 1 ? out1 : out2) + (size_t)pos * sizeof(Msg)) = m;
      } else {  // no room in the batch: nothing is sent, nothing will come back -- the client sees its request unanswered
        lost++;
        if (o.is_new(k)) { Msg q = m; q.ord = 0; tx_consume_one(c, o.dst[k], &q); }
      }
    }
  }
  if (valid) txg_store_client(c, cl, n_clients, i);
  // ---- statistics: LDS first, then one device atomic per counter and workgroup
  if (valid) {
    atomicAdd(&Sst[offsetof(txg_stats, messages) / 8], (unsigned long long)o.n);
    if (lost) atomicAdd(&Sst[offsetof(txg_stats, overflow) / 8], (unsigned long long)lost);
    for (uint8_t k = 0; k < o.n_fin && k < 2; k++) {
      atomicAdd(&Sst[offsetof(txg_stats, txns) / 8], 1ull);
      atomicAdd(&Sst[offsetof(txg_stats, by_type) / 8 + o.fin_txn[k]], 1ull);
      if (o.fin_ok[k]) {
        atomicAdd(&Sst[offsetof(txg_stats, committed) / 8], 1ull);
        atomicAdd(&Sst[offsetof(txg_stats, committed_by_type) / 8 + o.fin_txn[k]], 1ull);
      }
    }
  }
  __syncthreads();
  if (t < TXG_NSTAT && Sst[t]) atomicAdd((unsigned long long *)st + t, Sst[t]);
}

template <class T>
__global__ void __launch_bounds__(TXG_TB)
k_txn_consume(uint32_t *cl, uint8_t *store, uint32_t n_clients, const uint8_t *rep0,
              const uint8_t *rep1, const uint8_t *rep2, uint32_t cap) {
  typedef typename T::Msg Msg;
  const uint32_t i = blockIdx.x * TXG_TB + threadIdx.x;
  if (i >= n_clients) return;
  typename T::Client c;
  txg_load_client(c, cl, n_clients, i);
  c.m.base = store + (size_t)i * TX_DEV_MSG_STRIDE;
  c.m.stride = (uint64_t)n_clients * TX_DEV_MSG_STRIDE;
  const uint8_t n = c.n_out;
#pragma unroll
  for (uint8_t k = 0; k < T::MAXOUT; k++) {
    if (k < n && c.out_pos[k] < cap) {
      const uint8_t sh = c.out_shard[k];
      const uint8_t *rb = sh == 0 ? rep0 : sh == 1 ? rep1 : rep2;
      tx_consume_one(c, c.out_dst[k], (const Msg *)(rb + (size_t)c.out_pos[k] * sizeof(Msg)));
    }
  }
  txg_store_client(c, cl, n_clients, i);  // the reply summaries live in the header
}

// ------------------------------------------------------------------------------------------------- host side
struct dint_gdriver {
  dint_driver_config cfg{};
  int device = 0;
  uint32_t cap = 0, ntiles = 0, msg = 0;
  bool awaiting = false;
  bool fuse = true;            // DINT_TXN_FUSE=0: consume always in its own kernel
  int waves = 4;               // DINT_TXN_WAVES: register budget variant of k_txn_emit (3, 4 or 5 waves per SIMD)
  bool pending = false;        // a consume deferred into the next emit ...
  hipStream_t pending_stream = nullptr;  // ... which was issued on this stream
  hipStream_t next_stream = nullptr;     // stream of the last dint_gdriver_next
  hipEvent_t ev_pending = nullptr;
  uint32_t cur = 0;            // buffer set of the current epoch (the sets alternate)
  uint64_t epochs = 0;
  void *d_clients = nullptr, *d_store = nullptr;
  uint8_t *d_batch[2][DINT_N_SHARDS] = {{nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}};
  uint32_t *d_counts = nullptr, *d_pub = nullptr, *d_ticket = nullptr, *d_zipf = nullptr;
  txg_stats *d_stats = nullptr;
  TxParams P{};
  uint32_t dbg = 0;  // DINT_TXN_DBG experiments: 1 = no look-back, 2 = no client logic
};

namespace {
template <class T>
int upload_clients(dint_gdriver *g) {
  typedef typename T::Client Client;
  const size_t n = g->cfg.n_clients, nw = sizeof(Client) / 4;
  std::vector<uint32_t> h(n * nw);  // dword columns: word j of client i at h[j * n + i] (txg_load_client)
  for (size_t i = 0; i < n; i++) {
    Client c;
    memset(&c, 0, sizeof(Client));
    c.rng.s = 0xdeadbeefull + g->cfg.first_client + i;  // ClientLoop :1122
    uint32_t w[sizeof(Client) / 4];
    memcpy(w, &c, sizeof(Client));
    for (size_t j = 0; j < nw; j++) h[j * n + i] = w[j];
  }
  const size_t bytes = h.size() * 4, sbytes = n * T::NMSG * (size_t)TX_DEV_MSG_STRIDE;
  if (hipMalloc(&g->d_clients, bytes) != hipSuccess || hipMalloc(&g->d_store, sbytes) != hipSuccess) return DINT_ENOMEM;
  if (hipMemcpy(g->d_clients, h.data(), bytes, hipMemcpyHostToDevice) != hipSuccess) return DINT_EHIP;
  if (hipMemset(g->d_store, 0, sbytes) != hipSuccess) return DINT_EHIP;
  return 0;
}
}  // namespace

namespace {
template <class T>
void launch_consume(dint_gdriver *g, hipStream_t st, uint32_t set) {
  hipLaunchKernelGGL((k_txn_consume<T>), dim3(g->ntiles), dim3(TXG_TB), 0, st, (uint32_t *)g->d_clients,
                     (uint8_t *)g->d_store, g->cfg.n_clients, g->d_batch[set][0], g->d_batch[set][1], g->d_batch[set][2], g->cap);
}
template <class T>
void launch_emit(dint_gdriver *g, hipStream_t st, bool fused) {
  const uint32_t b = g->cur, o = b ^ 1u;  // requests go into set b; the replies of the previous epoch sit in set o
#define TXG_LAUNCH(W)                                                                                                      \
  hipLaunchKernelGGL((k_txn_emit<T, W>), dim3(g->ntiles), dim3(TXG_TB), 0, st, (uint32_t *)g->d_clients,                     \
                     (uint8_t *)g->d_store, g->cfg.n_clients, g->P, g->d_batch[b][0], g->d_batch[b][1], g->d_batch[b][2],    \
                     fused ? g->d_batch[o][0] : nullptr, fused ? g->d_batch[o][1] : nullptr, fused ? g->d_batch[o][2] : nullptr, \
                     g->cap, g->d_pub + (size_t)b * g->ntiles * 4, g->d_ticket + b, g->d_pub + (size_t)o * g->ntiles * 4,     \
                     g->d_ticket + o, g->d_counts, g->d_stats, g->dbg)
  if (g->waves == 3) TXG_LAUNCH(3);
  else if (g->waves == 5) TXG_LAUNCH(5);
  else TXG_LAUNCH(4);
#undef TXG_LAUNCH
}
}  // namespace

extern "C" {

int dint_gdriver_create(const dint_driver_config *cfg, int32_t device, uint32_t cap_per_shard, dint_gdriver_t **out) {
  if (!cfg || !out || cfg->n_clients == 0 || cfg->n_rows == 0 || cap_per_shard < 2) return DINT_EINVAL;
  if (cfg->workload != DINT_WL_TATP && cfg->workload != DINT_WL_SMALLBANK) return DINT_EINVAL;
  if (cfg->key_dist > 1 || (cfg->key_dist == 1 && !(cfg->zipf_theta > 0 && cfg->zipf_theta < 1))) return DINT_EINVAL;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return DINT_ENODEV;
  if (device < 0 && hipGetDevice(&device) != hipSuccess) return DINT_ENODEV;
  if (device >= ndev || hipSetDevice(device) != hipSuccess) return DINT_ENODEV;
  dint_gdriver *g = new (std::nothrow) dint_gdriver();
  if (!g) return DINT_ENOMEM;
  g->cfg = *cfg;
  g->device = device;
  g->cap = cap_per_shard;
  g->msg = cfg->workload == DINT_WL_TATP ? 55 : 23;
  g->ntiles = (cfg->n_clients + TXG_TB - 1) / TXG_TB;
  if (getenv("DINT_TXN_DBG")) g->dbg = (uint32_t)atoi(getenv("DINT_TXN_DBG"));
  if (getenv("DINT_TXN_FUSE")) g->fuse = atoi(getenv("DINT_TXN_FUSE")) != 0;
  if (getenv("DINT_TXN_WAVES")) g->waves = atoi(getenv("DINT_TXN_WAVES"));
  int rc = 0;
  try {
    ZipfTable zipf;
    dint_driver_params(*cfg, &g->P, &zipf);
    if (cfg->key_dist == 1) {
      if (hipMalloc((void **)&g->d_zipf, zipf.cdf.size() * 4) != hipSuccess) rc = DINT_ENOMEM;
      else if (hipMemcpy(g->d_zipf, zipf.cdf.data(), zipf.cdf.size() * 4, hipMemcpyHostToDevice) != hipSuccess) rc = DINT_EHIP;
      g->P.zipf_cdf = g->d_zipf;
    }
    if (!rc) rc = cfg->workload == DINT_WL_TATP ? upload_clients<TatpTraits>(g) : upload_clients<SbTraits>(g);
  } catch (const std::bad_alloc &) {
    rc = DINT_ENOMEM;
  }
  for (int b = 0; b < 2 && !rc; b++)
    for (int s = 0; s < DINT_N_SHARDS && !rc; s++)
      if (hipMalloc((void **)&g->d_batch[b][s], (size_t)g->cap * g->msg + 64) != hipSuccess) rc = DINT_ENOMEM;
  if (!rc && hipEventCreateWithFlags(&g->ev_pending, hipEventDisableTiming) != hipSuccess) rc = DINT_EHIP;
  if (!rc && (hipMalloc((void **)&g->d_counts, 16) != hipSuccess || hipMalloc((void **)&g->d_ticket, 8) != hipSuccess ||
              hipMalloc((void **)&g->d_pub, (size_t)g->ntiles * 32) != hipSuccess ||
              hipMalloc((void **)&g->d_stats, sizeof(txg_stats)) != hipSuccess))
    rc = DINT_ENOMEM;
  if (!rc && (hipMemset(g->d_counts, 0, 16) != hipSuccess || hipMemset(g->d_ticket, 0, 8) != hipSuccess ||
              hipMemset(g->d_pub, 0, (size_t)g->ntiles * 32) != hipSuccess ||
              hipMemset(g->d_stats, 0, sizeof(txg_stats)) != hipSuccess || hipDeviceSynchronize() != hipSuccess))
    rc = DINT_EHIP;
  if (rc) {
    dint_gdriver_destroy(g);
    return rc;
  }
  *out = g;
  return 0;
}

void dint_gdriver_destroy(dint_gdriver_t *g) {
  if (!g) return;
  hipSetDevice(g->device);
  hipDeviceSynchronize();
  hipFree(g->d_clients);
  hipFree(g->d_store);
  for (auto &b : g->d_batch)
    for (auto p : b) hipFree(p);
  if (g->ev_pending) hipEventDestroy(g->ev_pending);
  hipFree(g->d_counts); hipFree(g->d_pub); hipFree(g->d_ticket); hipFree(g->d_zipf); hipFree(g->d_stats);
  delete g;
}

int dint_gdriver_next(dint_gdriver_t *g, void *stream) {
  if (!g) return DINT_EINVAL;
  if (g->awaiting) return DINT_ESTATE;
  if (hipSetDevice(g->device) != hipSuccess) return DINT_EHIP;
  hipStream_t st = (hipStream_t)stream;
  const bool tatp = g->cfg.workload == DINT_WL_TATP;
  bool fused = false;
  if (g->pending) {
    if (st == g->pending_stream) {
      fused = true;  // the replies are copied by the emit kernel itself
    } else {  // consume was promised on another stream: run it there, and this stream behind it
      if (tatp) launch_consume<TatpTraits>(g, g->pending_stream, g->cur); else launch_consume<SbTraits>(g, g->pending_stream, g->cur);
      if (hipEventRecord(g->ev_pending, g->pending_stream) != hipSuccess || hipStreamWaitEvent(st, g->ev_pending, 0) != hipSuccess)
        return DINT_EHIP;
    }
    g->pending = false;
  }
  g->cur ^= 1u;
  if (tatp) launch_emit<TatpTraits>(g, st, fused); else launch_emit<SbTraits>(g, st, fused);
  if (hipGetLastError() != hipSuccess) return DINT_EHIP;
  g->next_stream = st;
  g->awaiting = true;
  g->epochs++;
  return 0;
}

// The replies are in place in the current batches.  Issued on the stream of the last dint_gdriver_next (the closed
// loop), the copy is deferred into the next emit kernel; the stream must then still exist at the next
// dint_gdriver_next.  On any other stream (or with DINT_TXN_FUSE=0) it runs now, in a kernel of its own.
int dint_gdriver_consume(dint_gdriver_t *g, void *stream) {
  if (!g) return DINT_EINVAL;
  if (!g->awaiting) return DINT_ESTATE;
  if (hipSetDevice(g->device) != hipSuccess) return DINT_EHIP;
  hipStream_t st = (hipStream_t)stream;
  if (g->fuse && st == g->next_stream) {
    g->pending = true;
    g->pending_stream = st;
  } else {
    if (g->cfg.workload == DINT_WL_TATP) launch_consume<TatpTraits>(g, st, g->cur); else launch_consume<SbTraits>(g, st, g->cur);
    if (hipGetLastError() != hipSuccess) return DINT_EHIP;
  }
  g->awaiting = false;
  return 0;
}

// the CURRENT epoch's batch of a shard: the two buffer sets alternate, ask again after every dint_gdriver_next
void *dint_gdriver_batch(dint_gdriver_t *g, uint32_t shard) { return (g && shard < DINT_N_SHARDS) ? g->d_batch[g->cur][shard] : nullptr; }
const void *dint_gdriver_counts(dint_gdriver_t *g) { return g ? g->d_counts : nullptr; }
uint32_t dint_gdriver_cap(const dint_gdriver_t *g) { return g ? g->cap : 0; }

int64_t dint_gdriver_read_batch(dint_gdriver_t *g, uint32_t shard, void *host, uint64_t cap_msgs) {
  if (!g || shard >= DINT_N_SHARDS) return DINT_EINVAL;
  if (hipSetDevice(g->device) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return DINT_EHIP;
  uint32_t cnt[3];
  if (hipMemcpy(cnt, g->d_counts, sizeof cnt, hipMemcpyDeviceToHost) != hipSuccess) return DINT_EHIP;
  const uint64_t n = cnt[shard] < cap_msgs ? cnt[shard] : cap_msgs;
  if (host && n && hipMemcpy(host, g->d_batch[g->cur][shard], n * g->msg, hipMemcpyDeviceToHost) != hipSuccess) return DINT_EHIP;
  return (int64_t)cnt[shard];
}

int dint_gdriver_get_stats(dint_gdriver_t *g, dint_driver_stats *out, uint64_t *overflow) {
  if (!g || !out) return DINT_EINVAL;
  if (hipSetDevice(g->device) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return DINT_EHIP;
  txg_stats s;
  if (hipMemcpy(&s, g->d_stats, sizeof s, hipMemcpyDeviceToHost) != hipSuccess) return DINT_EHIP;
  memset(out, 0, sizeof *out);
  out->txns = s.txns; out->committed = s.committed; out->messages = s.messages; out->epochs = g->epochs;
  for (int k = 0; k < 8; k++) { out->by_type[k] = s.by_type[k]; out->committed_by_type[k] = s.committed_by_type[k]; }
  if (overflow) *overflow = s.overflow;
  return 0;
}

}  // extern "C"
