// k_route.hip -- multi-GPU request routing on gfx950.
//
// The reference shards by key on the CLIENT (`key % 3`, tatp/caladan/client_udp_shard.cc:187,523-531): every
// request is sent to the server that owns its key.  Inside one node the same decision is taken on the GPU that
// ingested the request: home = global slot / bucket % world -- the hash and modulus the engines use -- and the
// batch crosses xGMI once each way (one all-to-all of fixed-size slots; the collective itself is RCCL's).
//
//   k_route_pack    : ONE pass over the batch -- home rank of every request, STABLE partition into the destinations' slots,
//                     slot headers.  A tile of 1,024 requests is read once into LDS; its requests per destination
//                     are published, and the tile adds up what the tiles before it published (tiles are numbered
//                     by a ticket, so every predecessor is running: the wait cannot deadlock).  Request i goes to
//                     position (requests of the same home before i) of its destination's slot, so every destination
//                     receives this rank's requests in index order and the order at the home engine is (source
//                     rank, index): the serial order of the rank-major concatenation of all ingest batches, whatever
//                     the number of GPUs.  (r01-r03: three kernels -- count, scan, scatter -- that read the batch
//                     twice and kept a home[] array between them.)
//   k_route_unpack  : replies[i] = the slot message request i was sent in, after the inverse all-to-all
//
// A slot is `cap` messages; what a destination cannot take is answered "not now, send again" by the sender's own
// unpack (rt_refuse: the eBPF servers' REJECT_* / RETRY replies) and counted in dint_stats.route_overflow: callers size
// `cap` from the recorded or expected maximum, and Router grows it when the counter moves.
#include <algorithm>

#include "../../include/dint_abi.h"
#include "dint_kv.h"

#define RT_TB 256u    // requests per workgroup of the unpack kernel (>= 1000 workgroups per 256k batch)
#define RP_TB 1024u   // requests (= threads) per tile of the pack kernel: <= 1,024 tiles per batch, each adds up its predecessors
#define RP_FLAG 0x80000000u
#define RT_NONE 0xFFFFFFFFu

struct rt_params {
  uint32_t kind;       // 0 = lid workloads (fasst / 2pl), 1 = kv workloads
  uint32_t msg;        // bytes per wire message
  uint32_t key_off, table_off;  // kv: byte offsets (table_off = 0xFFFFFFFF: single table)
  dint_mod slots;      // lid workloads: % n_slots
  const kv_dev *kv;    // kv workloads
  uint32_t world, self;
};

// one batch to route (or to bring back): grid.y of every kernel below selects the item, so the S logical servers of a
// rank (S = 3 for tatp / smallbank) share ONE launch of each kernel -- a step costs 3 + 1 launches on the exchange
// stream, not 3 S + S (at ~10 us per dependent launch the stream, not the engines, was the limit of a step)
struct rt_item {
  const uint8_t *req;
  uint8_t *rep;          // unpack: replies in request order
  uint32_t n, cap;       // requests (with n_dev: their upper bound, which sizes the grid); slot capacity (messages)
  const uint32_t *n_dev; // the live request count in device memory (a batch a kernel produced), or nullptr
  uint8_t *send;         // this item's slot of peer 0 (peer w at + w * stride); unpack: the returned slots
  uint8_t *cnt;          // its u32 live count of peer 0 (peer w at + w * cnt_stride)
  uint64_t cnt_stride;
  uint32_t *slot;        // [n] where each request went
  uint32_t *blk;         // [tiles][world] scratch: requests per tile and destination (RP_FLAG = published) | ticket, extent
  uint32_t *blk_next;    // the other copy (the next call's): zeroed by this one
  dint_dev_stats *stats;
  rt_params p;
};
struct rt_items {
  uint64_t stride;
  rt_item it[DINT_ROUTE_MAXS];
};

// A request that found its destination slot full is not sent; its sender gets the reply the reference's eBPF servers give
// when they cannot take a request right now -- REJECT_READ / REJECT_LOCK / REJECT_COMMIT (tatp), RETRY (smallbank,
// lock_2pl), kReject* (store): every client answers these by sending the request again (dint_refuse in the ABI; e.g.
// tatp/ebpf/shard_kern.c:173-178,289-293,371-376).  Request types those servers never refuse (ABORT, log appends,
// lock_fasst's READ / ABORT / COMMIT) come back unchanged: not answered.  Either way it is counted (route_overflow).
__device__ static inline void rt_refuse(uint8_t *m, uint32_t msg) {
  switch (msg) {
    case 6: m[0] = 4; break;                                                 // lock_2pl RETRY
    case 9: if (m[0] == 1) m[0] = 6; break;                                  // lock_fasst REJECT_LOCK
    case 53: if (m[0] <= 2) m[0] = (uint8_t)(4 + 2 * m[0] + (m[0] == 2)); break;  // store 0 -> 4, 1 -> 6, 2 -> 9
    case 55:
      if (m[1] == 0) m[1] = 5;
      else if (m[1] == 1) m[1] = 8;
      else if (m[1] == 12 || m[1] == 13 || m[1] == 18 || m[1] == 19 || m[1] == 22 || m[1] == 23) m[1] = 11;
      break;
    default: if (m[1] <= 5 || m[1] == 17) m[1] = 16; break;                  // smallbank RETRY
  }
}

__device__ static inline uint32_t rt_n(const rt_item &it) { return it.n_dev ? min(*it.n_dev, it.n) : it.n; }

__device__ static inline uint32_t rt_home(const uint8_t *m, const rt_params &p) {
  uint64_t g;
  if (p.kind == 0) {
    uint32_t lid;
    __builtin_memcpy(&lid, m + 1, 4);  // lid sits at byte 1 of both lock messages
    g = dint_fastmod(dint_hash_lid(lid), p.slots);
  } else {
    const uint32_t table = p.table_off == 0xFFFFFFFFu ? 0 : m[p.table_off];
    if (table >= p.kv->n_tables) return p.self;  // no home: answered (as a bad request) where it was ingested
    uint64_t key;
    __builtin_memcpy(&key, m + p.key_off, 8);
    g = dint_fastmod(dint_hash_key(key), p.kv->mod[table]);
  }
  return (uint32_t)(g % p.world);
}

// copy one wire message (6 .. 55 bytes, unaligned): dwords first, all loads before the first store
__device__ static inline void rt_copy_msg(uint8_t *dst, const uint8_t *src, uint32_t msg) {
  uint32_t w[14];
  const uint32_t nd = msg >> 2;
#pragma unroll
  for (uint32_t k = 0; k < 14; k++)
    if (k < nd) __builtin_memcpy(&w[k], src + 4 * k, 4);
  uint8_t tail[3];
  for (uint32_t k = nd * 4; k < msg; k++) tail[k - nd * 4] = src[k];
#pragma unroll
  for (uint32_t k = 0; k < 14; k++)
    if (k < nd) __builtin_memcpy(dst + 4 * k, &w[k], 4);
  for (uint32_t k = nd * 4; k < msg; k++) dst[k] = tail[k - nd * 4];
}

__global__ void __launch_bounds__(256)
k_route_unpack_simple(rt_items I) {
  const rt_item &it = I.it[blockIdx.y];
  const uint8_t *__restrict__ back = it.send;
  const uint32_t cap = it.cap, n = rt_n(it), msg = it.p.msg;
  const uint64_t stride = I.stride;
  const uint32_t *__restrict__ slot = it.slot;
  const uint8_t *req = it.req;
  uint8_t *rep = it.rep;
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  const uint32_t s = slot[i];
  if (s != RT_NONE) {
    const uint32_t h = s / cap, pos = s - h * cap;
    rt_copy_msg(rep + (size_t)i * msg, back + (size_t)h * stride + (size_t)pos * msg, msg);
  } else {  // not sent (slot overflow): the "not now, send again" reply
    if (rep != req) rt_copy_msg(rep + (size_t)i * msg, req + (size_t)i * msg, msg);
    rt_refuse(rep + (size_t)i * msg, msg);
  }
}


// ---- LDS-staged forms (the common case: 16-byte aligned request / reply arrays) -------------------------------
// A wire message is 6..55 bytes at an odd offset: one lane copying one message touches memory 4 bytes at a time
// across a dozen cache lines, and a wave issues ~28 partially used transactions (measured: 22 us to move 13 MB, 0.6
// TB/s).  Here the workgroup's 1024-message tile is read with 16-byte vectors into LDS, permuted there
// (request order <-> destination-major order), and every destination's run leaves as one contiguous stream.
#define RT_LDS_BYTES (RT_TB * 55u + 32u * DINT_ROUTE_MAXW)  // + up to 31 bytes per run: each run keeps its 16-byte phase

__device__ static inline void rt_lds_load_tile(uint8_t *L, const uint8_t *g, uint32_t nbytes) {  // g 16-byte aligned
  const uint32_t t = threadIdx.x, nv = nbytes >> 4;
  for (uint32_t k = t; k < nv; k += RT_TB) ((uint4 *)L)[k] = ((const uint4 *)g)[k];
  for (uint32_t k = (nv << 4) + t; k < nbytes; k += RT_TB) L[k] = g[k];
}
__device__ static inline void rt_lds_store_tile(uint8_t *g, const uint8_t *L, uint32_t nbytes) {
  const uint32_t t = threadIdx.x, nv = nbytes >> 4;
  for (uint32_t k = t; k < nv; k += RT_TB) ((uint4 *)g)[k] = ((const uint4 *)L)[k];
  for (uint32_t k = (nv << 4) + t; k < nbytes; k += RT_TB) g[k] = L[k];
}
// A run of nb bytes between global memory g and LDS L, where L was placed with g's 16-byte phase ((L - Lb) & 15 ==
// g & 15, Lb 16-byte aligned): head bytes up to the first 16-byte boundary, 16-byte vectors, tail bytes.
template <uint32_t TB = RT_TB>
__device__ static inline void rt_run_store(uint8_t *g, const uint8_t *L, uint32_t nb) {
  const uint32_t t = threadIdx.x, hb = min(nb, (16u - (uint32_t)((uintptr_t)g & 15u)) & 15u), nv = (nb - hb) >> 4;
  if (t < hb) g[t] = L[t];
  for (uint32_t k = t; k < nv; k += TB) ((uint4 *)(g + hb))[k] = ((const uint4 *)(L + hb))[k];
  const uint32_t done = hb + (nv << 4);
  if (t < nb - done) g[done + t] = L[done + t];
}
template <uint32_t TB = RT_TB>
__device__ static inline void rt_run_load(uint8_t *L, const uint8_t *g, uint32_t nb) {
  const uint32_t t = threadIdx.x, hb = min(nb, (16u - (uint32_t)((uintptr_t)g & 15u)) & 15u), nv = (nb - hb) >> 4;
  if (t < hb) L[t] = g[t];
  for (uint32_t k = t; k < nv; k += TB) ((uint4 *)(L + hb))[k] = ((const uint4 *)(g + hb))[k];
  const uint32_t done = hb + (nv << 4);
  if (t < nb - done) L[done + t] = g[done + t];
}
// one message between LDS / memory and registers.  MSG is a compile-time constant (the kernels are instantiated per
// wire format): with a run-time size the array was indexed dynamically and lived in scratch memory -- 64 bytes per
// lane written and read back through HBM in both staged kernels.
template <uint32_t MSG> struct rt_regs { uint32_t w[MSG / 4]; uint8_t tail[MSG % 4 ? MSG % 4 : 1]; };
template <uint32_t MSG> __device__ static inline void rt_get(rt_regs<MSG> &r, const uint8_t *L) {
#pragma unroll
  for (uint32_t k = 0; k < MSG / 4; k++) __builtin_memcpy(&r.w[k], L + 4 * k, 4);
#pragma unroll
  for (uint32_t k = 0; k < MSG % 4; k++) r.tail[k] = L[(MSG / 4) * 4 + k];
}
template <uint32_t MSG> __device__ static inline void rt_put(uint8_t *L, const rt_regs<MSG> &r) {
#pragma unroll
  for (uint32_t k = 0; k < MSG / 4; k++) __builtin_memcpy(L + 4 * k, &r.w[k], 4);
#pragma unroll
  for (uint32_t k = 0; k < MSG % 4; k++) L[(MSG / 4) * 4 + k] = r.tail[k];
}

#define RP_LDS_BYTES (RP_TB * 55u + 32u * DINT_ROUTE_MAXW + 16u)  // the tile at its 16-byte phase; then every run at its own
#define RP_CTL (DINT_ROUTE_BLK_WORDS - 4u)                         // [0] ticket, [1] aggregate words this call published

template <uint32_t MSG>
__global__ void __launch_bounds__(RP_TB)
k_route_pack(rt_items I) {
  const rt_item &it = I.it[blockIdx.y];
  constexpr uint32_t msg = MSG;
  const uint32_t n = rt_n(it), world = it.p.world, cap = it.cap;
  const uint32_t ntiles = (n + RP_TB - 1) / RP_TB;
  const uint8_t *__restrict__ req = it.req;
  uint32_t *blk = it.blk;
  uint8_t *send = it.send;
  const uint64_t stride = I.stride;
  uint32_t *__restrict__ slot = it.slot;
  __shared__ __attribute__((aligned(16))) uint8_t Lb[RP_LDS_BYTES];
  __shared__ uint32_t Wc[RP_TB / 64][DINT_ROUTE_MAXW];
  __shared__ uint32_t Cnt[DINT_ROUTE_MAXW], Loff[DINT_ROUTE_MAXW], Base[DINT_ROUTE_MAXW];  // of the tile per destination: messages, LDS byte offset of the run, messages of earlier tiles
  __shared__ uint32_t Sb, Sext;
  const uint32_t t = threadIdx.x, lane = t & 63, wave = t >> 6;
  if (t == 0) {
    Sb = atomicAdd(&blk[RP_CTL], 1u);  // tiles in the order the workgroups start
    // (read by ONE thread, before the barrier: the housekeeping below clears this word, and a wave that loaded it after
    // wave 0's store would see 0 and skip its share of the zeroing -- ADVICE r04)
    Sext = Sb == 0 ? it.blk_next[RP_CTL + 1] : 0u;
  }
  for (uint32_t k = t; k < (RP_TB / 64) * DINT_ROUTE_MAXW; k += RP_TB) (&Wc[0][0])[k] = 0;
  if (t < DINT_ROUTE_MAXW) Base[t] = 0;
  __syncthreads();
  const uint32_t b = Sb;
  if (b == 0) {  // housekeeping: the other copy of the scratch, which the next call uses, goes back to zero
    uint32_t *nx = it.blk_next;
    const uint32_t ext = min(Sext, RP_CTL);
    for (uint32_t k = t; k < ext; k += RP_TB) nx[k] = 0;
    if (t < 2) nx[RP_CTL + t] = 0;
    if (t == 0) blk[RP_CTL + 1] = ntiles * world;
    if (ntiles == 0 && t < world) *(uint32_t *)(it.cnt + (size_t)t * it.cnt_stride) = 0;  // an empty batch still writes its headers
  }
  if (b >= ntiles) return;  // (workgroup-uniform; the grid is as wide as the largest item's upper bound)
  const uint32_t tile_n = min(RP_TB, n - b * RP_TB), i = b * RP_TB + t;
  const uint8_t *g0 = req + (size_t)b * RP_TB * msg;
  uint8_t *Lt = Lb + (uint32_t)((uintptr_t)g0 & 15u);  // the tile keeps its 16-byte phase (request arrays at any address)
  rt_run_load<RP_TB>(Lt, g0, tile_n * msg);
  __syncthreads();
  const bool valid = t < tile_n;
  const uint32_t h = valid ? rt_home(Lt + t * msg, it.p) : 0;
  uint32_t rank = 0;  // requests of my home before me inside my wave
  for (uint64_t todo = __ballot(valid); todo;) {
    const int l = __ffsll((unsigned long long)todo) - 1;
    const uint32_t hh = (uint32_t)__builtin_amdgcn_readlane(h, l);
    const uint64_t m = __ballot(valid && h == hh);
    if (valid && h == hh) rank = (uint32_t)__popcll(m & lanemask_lt());
    if ((int)lane == l) Wc[wave][hh] = (uint32_t)__popcll(m);
    todo &= ~m;
  }
  rt_regs<MSG> r;
  if (valid) rt_get<MSG>(r, Lt + t * msg);
  __syncthreads();  // every message is in registers: the buffer can be rewritten destination-major
  if (t < world) {
    uint32_t c = 0;
    for (uint32_t k = 0; k < RP_TB / 64; k++) c += Wc[k][t];
    Cnt[t] = c;
    __hip_atomic_store(&blk[(size_t)b * world + t], c | RP_FLAG, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  {  // what the tiles before mine hold, per destination: thread t adds up words t, t + per, ... (all of destination t % world)
    const uint32_t per = (RP_TB / world) * world, tot = b * world;
    if (t < per) {
      uint32_t acc = 0;
      for (uint32_t k = t; k < tot; k += per) {
        uint32_t v;
        while (!((v = __hip_atomic_load(&blk[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) & RP_FLAG)) __builtin_amdgcn_s_sleep(1);
        acc += v & ~RP_FLAG;
      }
      if (acc) atomicAdd(&Base[t % world], acc);
    }
  }
  __syncthreads();
  if (t == 0) {  // every destination's run gets the 16-byte phase of where it goes in memory
    uint32_t o = 0;
    for (uint32_t w = 0; w < world; w++) {
      const uint8_t *g = send + (size_t)w * stride + (size_t)Base[w] * msg;
      o = ((o + 15u) & ~15u) + (uint32_t)((uintptr_t)g & 15u);
      Loff[w] = o;
      o += Cnt[w] * msg;
    }
  }
  __syncthreads();
  uint32_t lrank = rank;
  for (uint32_t k = 0; k < wave; k++) lrank += Wc[k][h];
  if (valid) {
    rt_put<MSG>(Lb + Loff[h] + lrank * msg, r);
    const uint32_t pos = Base[h] + lrank;
    slot[i] = pos < cap ? h * cap + pos : RT_NONE;
  }
  __syncthreads();
  for (uint32_t w = 0; w < world; w++) {  // every destination's run of this tile: one contiguous stream
    const uint32_t base = Base[w];
    const uint32_t cnt = base < cap ? min(Cnt[w], cap - base) : 0;
    rt_run_store<RP_TB>(send + (size_t)w * stride + (size_t)base * msg, Lb + Loff[w], cnt * msg);
  }
  if (b == ntiles - 1 && t < world) {  // the last tile knows the totals: slot headers (live count, clamped to the slot capacity)
    const uint32_t total = Base[t] + Cnt[t];
    *(uint32_t *)(it.cnt + (size_t)t * it.cnt_stride) = min(total, cap);
    if (total > cap) atomicAdd(&it.stats->route_overflow, (unsigned long long)(total - cap));
  }
}

template <uint32_t MSG>
__global__ void __launch_bounds__(RT_TB)
k_route_unpack(rt_items I) {
  const rt_item &it = I.it[blockIdx.y];
  const uint8_t *__restrict__ back = it.send;
  constexpr uint32_t msg = MSG;
  const uint32_t cap = it.cap, n = rt_n(it), world = it.p.world;
  if (blockIdx.x * RT_TB >= n) return;
  const uint64_t stride = I.stride;
  const uint32_t *__restrict__ slot = it.slot;
  const uint8_t *req = it.req;
  uint8_t *rep = it.rep;
  __shared__ __attribute__((aligned(16))) uint8_t Lb[RT_LDS_BYTES];
  __shared__ uint32_t Cnt[DINT_ROUTE_MAXW], Min[DINT_ROUTE_MAXW], Loff[DINT_ROUTE_MAXW];
  const uint32_t t = threadIdx.x, lane = t & 63, i = blockIdx.x * RT_TB + t;
  const uint32_t tile_n = min(RT_TB, n - blockIdx.x * RT_TB);
  if (t < world) { Cnt[t] = 0; Min[t] = 0xFFFFFFFFu; }
  __syncthreads();
  const bool valid = i < n;
  const uint32_t s = valid ? slot[i] : RT_NONE;
  const bool routed = s != RT_NONE;
  const uint32_t h = routed ? s / cap : 0, pos = routed ? s - h * cap : 0;
  // the tile's requests of one home sit at consecutive slot positions (the partition was stable): run = [min, min + count)
  for (uint64_t todo = __ballot(routed); todo;) {
    const int l = __ffsll((unsigned long long)todo) - 1;
    const uint32_t hh = (uint32_t)__builtin_amdgcn_readlane(h, l);
    const uint64_t m = __ballot(routed && h == hh);
    const uint32_t first = (uint32_t)__builtin_amdgcn_readlane(pos, __ffsll((unsigned long long)m) - 1);
    if ((int)lane == l) { atomicAdd(&Cnt[hh], (uint32_t)__popcll(m)); atomicMin(&Min[hh], first); }
    todo &= ~m;
  }
  __syncthreads();
  if (t == 0) {
    uint32_t o = 0;
    for (uint32_t w = 0; w < world; w++) {
      const uint8_t *g = back + (size_t)w * stride + (size_t)(Cnt[w] ? Min[w] : 0u) * msg;
      o = ((o + 15u) & ~15u) + (uint32_t)((uintptr_t)g & 15u);
      Loff[w] = o;
      o += Cnt[w] * msg;
    }
  }
  __syncthreads();
  for (uint32_t w = 0; w < world; w++) {  // every home's run: one contiguous stream into LDS
    const uint32_t nb = Cnt[w] * msg;
    if (!nb) continue;
    rt_run_load(Lb + Loff[w], back + (size_t)w * stride + (size_t)Min[w] * msg, nb);
  }
  __syncthreads();
  rt_regs<MSG> r;
  if (routed) rt_get<MSG>(r, Lb + Loff[h] + (pos - Min[h]) * msg);
  else if (valid) rt_get<MSG>(r, req + (size_t)i * msg);  // not sent (slot overflow): the request, refused below
  __syncthreads();
  if (valid) rt_put<MSG>(Lb + t * msg, r);  // request order
  if (valid && !routed) rt_refuse(Lb + t * msg, msg);
  __syncthreads();
  rt_lds_store_tile(rep + (size_t)blockIdx.x * RT_TB * msg, Lb, tile_n * msg);
}

static rt_params make_params(uint32_t workload, uint32_t msg, dint_mod slots, const dint_kv *kv, dint_shard shard) {
  rt_params p;
  p.msg = msg;
  p.slots = slots;
  p.kv = nullptr;
  p.world = shard.count;
  p.self = shard.index;
  p.key_off = p.table_off = 0;
  if (workload == DINT_WL_FASST || workload == DINT_WL_2PL) {
    p.kind = 0;
  } else {
    const dint_kv_fmt f = dint_kv_format(workload);
    p.kind = 1;
    p.key_off = f.key;
    p.table_off = f.table;
    p.kv = kv->d_dev;
  }
  return p;
}

// items[k] routed with engine k's parameters; one launch, grid.y = item (batches of different wire formats: one launch each)
template <uint32_t MSG> static void launch_pack(const rt_items &I, uint32_t tiles, uint32_t n_items, hipStream_t st) {
  hipLaunchKernelGGL(k_route_pack<MSG>, dim3(std::max(tiles, 1u), n_items), dim3(RP_TB), 0, st, I);
}
static void launch_pack_msg(uint32_t msg, const rt_items &I, uint32_t tiles, uint32_t n_items, hipStream_t st) {
  switch (msg) {
    case 6: launch_pack<6>(I, tiles, n_items, st); break;
    case 9: launch_pack<9>(I, tiles, n_items, st); break;
    case 23: launch_pack<23>(I, tiles, n_items, st); break;
    case 53: launch_pack<53>(I, tiles, n_items, st); break;
    default: launch_pack<55>(I, tiles, n_items, st); break;  // (the engines' wire formats: 6, 9, 23, 53, 55)
  }
}
void dint_launch_route_pack(const dint_route_job *jobs, uint32_t n_jobs, uint64_t stride, hipStream_t st) {
  rt_items I;
  I.stride = stride;
  uint32_t tiles = 0;
  bool one_fmt = true;
  for (uint32_t k = 0; k < n_jobs; k++) {
    const dint_route_job &j = jobs[k];
    rt_item &it = I.it[k];
    it.req = (const uint8_t *)j.d_req;
    it.rep = nullptr;
    it.n = j.n;
    it.n_dev = j.d_n;
    it.cap = j.cap;
    it.send = (uint8_t *)j.d_send;
    it.cnt = (uint8_t *)j.d_cnt;
    it.cnt_stride = j.cnt_stride;
    it.slot = j.d_slot;
    it.blk = j.rs.blk;
    it.blk_next = j.rs.blk_next;
    it.stats = j.stats;
    it.p = make_params(j.workload, j.msg, j.slots, j.kv, j.shard);
    tiles = std::max(tiles, (j.n + RP_TB - 1) / RP_TB);  // <= DINT_ROUTE_MAXN / RP_TB = 1024
    one_fmt = one_fmt && j.msg == jobs[0].msg;
  }
  if (one_fmt) {
    launch_pack_msg(jobs[0].msg, I, tiles, n_jobs, st);
    return;
  }
  for (uint32_t k = 0; k < n_jobs; k++) {
    rt_items J;
    J.stride = stride;
    J.it[0] = I.it[k];
    launch_pack_msg(jobs[k].msg, J, (jobs[k].n + RP_TB - 1) / RP_TB, 1, st);
  }
}

void dint_launch_route_unpack(const dint_route_job *jobs, uint32_t n_jobs, uint64_t stride, hipStream_t st) {
  rt_items I;
  I.stride = stride;
  uint32_t n_max = 0;
  bool aligned = true;
  for (uint32_t k = 0; k < n_jobs; k++) {
    const dint_route_job &j = jobs[k];
    rt_item &it = I.it[k];
    it = rt_item();
    it.req = (const uint8_t *)j.d_req;
    it.rep = (uint8_t *)j.d_rep;
    it.n = j.n;
    it.n_dev = j.d_n;
    it.cap = j.cap;
    it.send = (uint8_t *)j.d_send;  // the slots as they came back
    it.slot = j.d_slot;
    it.p.msg = j.msg;
    it.p.world = j.shard.count;
    n_max = std::max(n_max, j.n);
    aligned = aligned && ((uintptr_t)j.d_rep & 15) == 0;
  }
  if (n_max == 0) return;
  bool staged = aligned;
  for (uint32_t k = 1; k < n_jobs; k++) staged = staged && jobs[k].msg == jobs[0].msg;
  if (staged) {
    const dim3 g((n_max + RT_TB - 1) / RT_TB, n_jobs), b(RT_TB);
    switch (jobs[0].msg) {
      case 6: hipLaunchKernelGGL(k_route_unpack<6>, g, b, 0, st, I); break;
      case 9: hipLaunchKernelGGL(k_route_unpack<9>, g, b, 0, st, I); break;
      case 23: hipLaunchKernelGGL(k_route_unpack<23>, g, b, 0, st, I); break;
      case 53: hipLaunchKernelGGL(k_route_unpack<53>, g, b, 0, st, I); break;
      case 55: hipLaunchKernelGGL(k_route_unpack<55>, g, b, 0, st, I); break;
      default: staged = false;
    }
  }
  if (!staged)
    hipLaunchKernelGGL(k_route_unpack_simple, dim3((n_max + 255) / 256, n_jobs), dim3(256), 0, st, I);
}

static_assert((DINT_ROUTE_MAXN / RP_TB) * DINT_ROUTE_MAXW <= RP_CTL, "aggregates of the largest batch to the widest world, then the control words");
