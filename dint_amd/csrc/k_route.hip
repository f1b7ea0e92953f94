// k_route.hip -- multi-GPU request routing on gfx950.
//
// The reference shards by key on the CLIENT (`key % 3`, tatp/caladan/client_udp_shard.cc:187,523-531): every
// request is sent to the server that owns its key.  Inside one node the same decision is taken on the GPU that
// ingested the request: home = global slot / bucket % world -- the hash and modulus the engines use -- and the
// batch crosses xGMI once each way (one all-to-all of fixed-size slots; the collective itself is RCCL's).
//
//   k_route_count   : home rank of every request (kept in scratch) + requests per destination of each
//                     1024-request block
//   k_route_scan    : exclusive scan of those counts over the blocks, per destination (one workgroup); writes the
//                     slot headers (live count per destination, clamped to the slot capacity)
//   k_route_scatter : STABLE partition -- request i goes to position (requests of the same home before i) of its
//                     destination's slot, so every destination receives this rank's requests in index order and
//                     the order at the home engine is (source rank, index): the serial order of the rank-major
//                     concatenation of all ingest batches, whatever the number of GPUs
//   k_route_unpack  : replies[i] = the slot message request i was sent in, after the inverse all-to-all
//
// A slot is `cap` messages; a destination that would receive more drops the excess (reply = request, counted in
// dint_stats.route_overflow): callers size `cap` from the recorded or expected maximum and check the counter.
#include "../../include/dint_abi.h"
#include "dint_kv.h"

#define RT_TB 1024u
#define RT_NONE 0xFFFFFFFFu

struct rt_params {
  uint32_t kind;       // 0 = lid workloads (fasst / 2pl), 1 = kv workloads
  uint32_t msg;        // bytes per wire message
  uint32_t key_off, table_off;  // kv: byte offsets (table_off = 0xFFFFFFFF: single table)
  dint_mod slots;      // lid workloads: % n_slots
  const kv_dev *kv;    // kv workloads
  uint32_t world, self;
};

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

__global__ void __launch_bounds__(RT_TB)
k_route_count(const uint8_t *__restrict__ req, uint32_t n, rt_params p, uint8_t *__restrict__ home,
              uint32_t *__restrict__ blk) {
  __shared__ uint32_t H[DINT_ROUTE_MAXW];
  const uint32_t t = threadIdx.x, i = blockIdx.x * RT_TB + t;
  if (t < p.world) H[t] = 0;
  __syncthreads();
  const bool valid = i < n;
  uint32_t h = 0;
  if (valid) {
    h = rt_home(req + (size_t)i * p.msg, p);
    home[i] = (uint8_t)h;
  }
  for (uint64_t todo = __ballot(valid); todo;) {  // one LDS atomic per wave and destination
    const int l = __ffsll((unsigned long long)todo) - 1;
    const uint32_t hh = (uint32_t)__builtin_amdgcn_readlane(h, l);
    const uint64_t m = __ballot(valid && h == hh);
    if ((int)lane_id() == l) atomicAdd(&H[hh], (uint32_t)__popcll(m));
    todo &= ~m;
  }
  __syncthreads();
  if (t < p.world) blk[(size_t)blockIdx.x * p.world + t] = H[t];
}

__global__ void __launch_bounds__(RT_TB)
k_route_scan(uint32_t nb, uint32_t world, uint32_t cap, uint32_t *__restrict__ blk, uint8_t *cnt, uint64_t cnt_stride,
             dint_dev_stats *__restrict__ stats) {
  __shared__ uint32_t Sw[RT_TB / 64];
  const uint32_t t = threadIdx.x, lane = t & 63, wave = t >> 6;
  for (uint32_t w = 0; w < world; w++) {
    const uint32_t c = t < nb ? blk[(size_t)t * world + w] : 0;
    uint32_t tot, x = wave_excl_scan_u32(c, &tot);
    __syncthreads();
    if (lane == 0) Sw[wave] = tot;
    __syncthreads();
    uint32_t total = 0;
    for (uint32_t k = 0; k < RT_TB / 64; k++) {
      if (k < wave) x += Sw[k];
      total += Sw[k];
    }
    if (t < nb) blk[(size_t)t * world + w] = x;
    if (t == 0) {
      *(uint32_t *)(cnt + (size_t)w * cnt_stride) = min(total, cap);
      if (total > cap) atomicAdd(&stats->route_overflow, (unsigned long long)(total - cap));
    }
  }
}

__global__ void __launch_bounds__(RT_TB)
k_route_scatter(const uint8_t *__restrict__ req, uint32_t n, uint32_t msg, uint32_t world, uint32_t cap,
                const uint8_t *__restrict__ home, const uint32_t *__restrict__ blk, uint8_t *send, uint64_t stride,
                uint32_t *__restrict__ slot) {
  __shared__ uint32_t Wc[RT_TB / 64][DINT_ROUTE_MAXW];
  const uint32_t t = threadIdx.x, lane = t & 63, wave = t >> 6, i = blockIdx.x * RT_TB + t;
  for (uint32_t k = t; k < (RT_TB / 64) * DINT_ROUTE_MAXW; k += RT_TB) (&Wc[0][0])[k] = 0;
  __syncthreads();
  const bool valid = i < n;
  const uint32_t h = valid ? home[i] : 0;
  uint32_t rank = 0;  // requests of my home before me inside my wave
  for (uint64_t todo = __ballot(valid); todo;) {
    const int l = __ffsll((unsigned long long)todo) - 1;
    const uint32_t hh = (uint32_t)__builtin_amdgcn_readlane(h, l);
    const uint64_t m = __ballot(valid && h == hh);
    if (valid && h == hh) rank = (uint32_t)__popcll(m & lanemask_lt());
    if ((int)lane == l) Wc[wave][hh] = (uint32_t)__popcll(m);
    todo &= ~m;
  }
  __syncthreads();
  if (!valid) return;
  uint32_t pos = blk[(size_t)blockIdx.x * world + h] + rank;
  for (uint32_t k = 0; k < wave; k++) pos += Wc[k][h];
  if (pos < cap) {
    rt_copy_msg(send + (size_t)h * stride + (size_t)pos * msg, req + (size_t)i * msg, msg);
    slot[i] = h * cap + pos;
  } else {
    slot[i] = RT_NONE;
  }
}

__global__ void __launch_bounds__(256)
k_route_unpack(const uint8_t *__restrict__ back, uint32_t cap, uint64_t stride, const uint32_t *__restrict__ slot,
               const uint8_t *req, uint32_t n, uint32_t msg, uint8_t *rep) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  const uint32_t s = slot[i];
  if (s != RT_NONE) {
    const uint32_t h = s / cap, pos = s - h * cap;
    rt_copy_msg(rep + (size_t)i * msg, back + (size_t)h * stride + (size_t)pos * msg, msg);
  } else if (rep != req) {
    rt_copy_msg(rep + (size_t)i * msg, req + (size_t)i * msg, msg);
  }
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

void dint_launch_route_pack(uint32_t workload, uint32_t msg, dint_mod slots, const dint_kv *kv, dint_shard shard,
                            const void *d_req, uint32_t n, void *d_send, uint32_t cap, uint64_t stride, void *d_cnt,
                            uint64_t cnt_stride, uint32_t *d_slot, dint_route_scratch rs, dint_dev_stats *stats,
                            hipStream_t st) {
  const rt_params p = make_params(workload, msg, slots, kv, shard);
  const uint32_t nb = (n + RT_TB - 1) / RT_TB;  // <= DINT_ROUTE_MAXN / RT_TB = 1024; 0 blocks still writes the headers
  if (nb)
    hipLaunchKernelGGL(k_route_count, dim3(nb), dim3(RT_TB), 0, st, (const uint8_t *)d_req, n, p, rs.home, rs.blk);
  hipLaunchKernelGGL(k_route_scan, dim3(1), dim3(RT_TB), 0, st, nb, shard.count, cap, rs.blk, (uint8_t *)d_cnt,
                     cnt_stride, stats);
  if (nb)
    hipLaunchKernelGGL(k_route_scatter, dim3(nb), dim3(RT_TB), 0, st, (const uint8_t *)d_req, n, msg, shard.count, cap,
                       (const uint8_t *)rs.home, (const uint32_t *)rs.blk, (uint8_t *)d_send, stride, d_slot);
}

void dint_launch_route_unpack(const void *d_back, uint32_t cap, uint64_t stride, const uint32_t *d_slot,
                              const void *d_req, uint32_t n, uint32_t msg, void *d_rep, hipStream_t st) {
  if (n == 0) return;
  hipLaunchKernelGGL(k_route_unpack, dim3((n + 255) / 256), dim3(256), 0, st, (const uint8_t *)d_back, cap, stride,
                     d_slot, (const uint8_t *)d_req, n, msg, (uint8_t *)d_rep);
}
