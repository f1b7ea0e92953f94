// k_log.hip -- log_server append on gfx950.
//
// Reference: log_server/udp/server.cc:73-88 -- every COMMIT message is copied into the thread's
// ring at log_entry_cnt, the counter advances modulo kMaxLogEntryNum, the reply is ACK.  A serial
// replay uses one ring (ring 0) and fills it in request order.
//
// GPU formulation: the ring position of request i is  tail + (number of valid log requests below i)
// -- an exclusive scan over the batch, so ring contents are deterministic and identical to the
// serial order (an atomicAdd per request would give an arbitrary order).
//   k_log_count : per-block count of valid requests -> blk_cnt[]
//   k_log_write : block base = sum of the preceding block counts; in-block wave scan; each thread
//                 writes one canonical 64-byte record with four 16-byte stores and patches the reply.
// The tail lives in HBM as {cur, next}: k_log_count publishes next -> cur at the start of a pass,
// k_log_write computes the new next.
#include "dint_kernels.h"

struct __attribute__((packed)) log_msg {  // log_server/udp/net.h:23-30
  uint8_t type;
  uint64_t key;
  uint8_t val[40];
  uint32_t ver;
};

__global__ void __launch_bounds__(256)
k_log_count(const uint8_t *__restrict__ req, uint32_t n, uint32_t *__restrict__ blk_cnt, uint32_t *tail) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (blockIdx.x == 0 && threadIdx.x == 0) tail[0] = tail[1];
  const bool valid = i < n && req[(size_t)i * sizeof(log_msg)] == 0;  // kCommit
  const uint32_t cnt = __syncthreads_count(valid);
  if (threadIdx.x == 0) blk_cnt[blockIdx.x] = cnt;
}

__global__ void __launch_bounds__(256)
k_log_write(const uint8_t *__restrict__ req, uint8_t *rep, uint32_t n, const uint32_t *__restrict__ blk_cnt,
            dint_log log, dint_dev_stats *__restrict__ stats) {
  __shared__ uint32_t red[4];
  __shared__ uint32_t redall[4];
  __shared__ uint32_t wbase[4];
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  // base of this block: sum of the counts of the blocks before it (<= 256 of them)
  uint32_t part = (threadIdx.x < blockIdx.x) ? blk_cnt[threadIdx.x] : 0;
  uint32_t tot;
  wave_excl_scan_u32(part, &tot);
  if (lane == 0) red[wv] = tot;
  // batch total: if the batch alone overflows the ring only its last `cap` records survive
  wave_excl_scan_u32((threadIdx.x < gridDim.x) ? blk_cnt[threadIdx.x] : 0, &tot);
  if (lane == 0) redall[wv] = tot;
  log_msg m;
  bool valid = false;
  if (i < n) {
    m = ((const log_msg *)req)[i];
    valid = m.type == 0;
  }
  const uint64_t vm = __ballot(valid);
  if (lane == 0) wbase[wv] = (uint32_t)__popcll(vm);
  __syncthreads();
  uint32_t base = red[0] + red[1] + red[2] + red[3];
  const uint32_t total_all = redall[0] + redall[1] + redall[2] + redall[3];
  for (uint32_t w = 0; w < wv; w++) base += wbase[w];
  const uint32_t pos_in_batch = base + (uint32_t)__popcll(vm & lanemask_lt());
  if (i < n) {
    if (valid && pos_in_batch + log.cap < total_all) {
      m.type = 1;  // overwritten later in this same batch by a record one ring-lap ahead
    } else if (valid) {
      const uint32_t pos = (uint32_t)(((uint64_t)log.tail[0] + pos_in_batch) % log.cap);
      uint4 *e = (uint4 *)(log.ring + (size_t)pos * 64);
      uint32_t w[16];
      __builtin_memcpy(&w[0], &m.key, 8);
      __builtin_memcpy(&w[2], m.val, 40);
      w[12] = m.ver;
      w[13] = 0;  // is_del = 0, table = 0
      w[14] = 0;
      w[15] = 0;
      e[0] = make_uint4(w[0], w[1], w[2], w[3]);
      e[1] = make_uint4(w[4], w[5], w[6], w[7]);
      e[2] = make_uint4(w[8], w[9], w[10], w[11]);
      e[3] = make_uint4(w[12], w[13], w[14], w[15]);
      m.type = 1;  // kAck
    } else {
      atomicAdd(&stats->bad_requests, 1ULL);
    }
    ((log_msg *)rep)[i] = m;
  }
  // the last thread of the last block knows the batch total
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 255) {
    const uint32_t total = pos_in_batch + (valid ? 1u : 0u);
    log.tail[1] = (uint32_t)(((uint64_t)log.tail[0] + total) % log.cap);
    *(unsigned long long *)(log.tail + 2) += total;  // records ever appended (dint_log_drain)
  }
}

void dint_launch_log(const void *d_req, void *d_rep, uint32_t n, dint_log log, dint_scratch s, hipStream_t st,
                     hipEvent_t *ev) {
  if (n == 0) return;
  const uint32_t nb = (n + 255) / 256;  // <= 256 blocks for n <= DINT_MICRO
  if (ev) hipEventRecord(ev[0], st);
  hipLaunchKernelGGL(k_log_count, dim3(nb), dim3(256), 0, st, (const uint8_t *)d_req, n, s.blk_cnt, log.tail);
  if (ev) hipEventRecord(ev[1], st);
  hipLaunchKernelGGL(k_log_write, dim3(nb), dim3(256), 0, st, (const uint8_t *)d_req, (uint8_t *)d_rep, n,
                     (const uint32_t *)s.blk_cnt, log, s.stats);
  if (ev) hipEventRecord(ev[2], st);
}
