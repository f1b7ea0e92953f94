// k_log.hip -- log_server append on gfx950: a streaming kernel.
//
// Reference: log_server/udp/server.cc:73-88 -- every COMMIT message is copied into the thread's
// ring at log_entry_cnt, the counter advances modulo kMaxLogEntryNum, the reply is ACK.  A serial
// replay uses one ring (ring 0) and fills it in request order.
//
// GPU formulation: the ring position of request i is  tail + (number of valid log requests below i)
// -- an exclusive scan over the batch, so ring contents are deterministic and identical to the
// serial order (an atomicAdd per request would give an arbitrary order).  One launch per pass of up to 2^20 requests
// (never more than the ring holds, so a pass does not lap it):
//   k_log_append : 1024 (256 for small passes) requests per workgroup of as many threads, tiles handed out by ticket.  The tile's 53-byte messages come
//                  in as 16-byte vectors, whole lines, through LDS (a thread reading its own packed struct from HBM
//                  touches two sectors for 53 bytes and shares each with its neighbours' loads); every thread takes its
//                  message out of LDS, the tile publishes its count of valid requests at once and reads the counts of
//                  the tiles before it at the very end (a decoupled look-back: by then they are long there); the canonical
//                  64-byte ring records of a wave are one contiguous 4 KB run of 16-byte stores; the reply -- the message
//                  with its type byte patched in LDS -- leaves as 16-byte vectors again.
//                  The last tile by index publishes the new tail, the last tile to FINISH makes it current (every other
//                  tile has read the old one by then).
// r01-r03: two launches (count, write) of 256-thread workgroups over passes of 65,536, every thread loading and storing
// its packed struct itself: 680 GB/s, 8.5 % of the HBM peak.  (r03c's one-launch form, in which every
// workgroup waited for ALL counts before it wrote anything, lost to the two launches; here nothing waits before its
// own work is done.)
#include "dint_kernels.h"

#define LOG_MSG 53u  // log_server/udp/net.h:23-30: {u8 type; u64 key; u8 val[40]; u32 ver}, packed
// requests per workgroup (= threads): 1024 for the passes that fill the GPU anyway, 256 for the small ones -- a 64k-request
// batch is 64 tiles of 1024, a quarter of the CUs; as 256 tiles of 256 it reaches all of them (17 -> ~10 us per batch)
#define LOG_TB_BIG 1024u
#define LOG_TB_SMALL 256u
#define LOG_SMALL_MAX (LOG_TB_SMALL * 1024u)  // a tile's look-back reads at most 1024 counts

__device__ static inline uint32_t lds_u32(const uint8_t *p) {  // (packed: byte-aligned in LDS)
  return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}

// pub: [0, 1024) valid requests per tile (bit 31 = published), [1024] tiles handed out, [1025] tiles finished
template <uint32_t LOG_TB>
__global__ void __launch_bounds__(LOG_TB)
k_log_append(const uint8_t *__restrict__ req, uint8_t *rep, uint32_t n, uint32_t n_tiles, dint_log log, uint32_t *pub,
             uint32_t *__restrict__ pub_next, dint_dev_stats *__restrict__ stats) {
  static_assert(LOG_TB * LOG_MSG % 16u == 0, "tiles start on a 16-byte boundary");
  __shared__ __attribute__((aligned(16))) uint8_t Sm[LOG_TB * LOG_MSG];
  __shared__ uint32_t Stile, Swc[LOG_TB / 64], Swp[LOG_TB / 64];
  const uint32_t t = threadIdx.x, lane = t & 63, wv = t >> 6;
  if (t == 0) Stile = atomicAdd(&pub[1024], 1u);
  if (blockIdx.x == 0)  // the words the next pass will use (two sets, used alternately)
    for (uint32_t k = t; k < 1024 + 16; k += LOG_TB) pub_next[k] = 0;
  const uint32_t tail0 = log.tail[0];
  __syncthreads();
  const uint32_t tile = Stile;
  const size_t lo = (size_t)tile * LOG_TB;
  const uint32_t cnt = (uint32_t)min((size_t)LOG_TB, (size_t)n - lo), bytes = cnt * LOG_MSG;
  const uint8_t *src = req + lo * LOG_MSG;
  uint8_t *dst = rep + lo * LOG_MSG;
  const bool vec = (((uintptr_t)req | (uintptr_t)rep) & 15) == 0;
  const uint32_t nv = vec ? bytes / 16 : 0;
  for (uint32_t k = t; k < nv; k += LOG_TB) ((uint4 *)Sm)[k] = ((const uint4 *)src)[k];
  for (uint32_t k = nv * 16 + t; k < bytes; k += LOG_TB) Sm[k] = src[k];
  __syncthreads();
  const uint8_t *m = Sm + t * LOG_MSG;
  const bool live = t < cnt, valid = live && m[0] == 0;  // kCommit
  const uint64_t vm = __ballot(valid);
  if (lane == 0) Swc[wv] = (uint32_t)__popcll(vm);
  __syncthreads();
  uint32_t before = 0, tile_total = 0;
  for (uint32_t w = 0; w < LOG_TB / 64; w++) {
    before += w < wv ? Swc[w] : 0;
    tile_total += Swc[w];
  }
  if (t == 0) __hip_atomic_store(&pub[tile], 0x80000000u | tile_total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // my record and my reply, in registers / LDS, before anything waits
  uint32_t w[13];
#pragma unroll
  for (uint32_t k = 0; k < 13; k++) w[k] = valid ? lds_u32(m + 1 + 4 * k) : 0u;  // key (2), val (10), ver (1)
  if (live && !valid) atomicAdd(&stats->bad_requests, 1ULL);
  __syncthreads();  // every thread has taken its message: the type bytes may change
  if (valid) Sm[t * LOG_MSG] = 1;  // kAck
  // ---- the tiles before mine (they were handed out earlier, so they run or are done): one count per thread
  uint32_t part = 0;
  for (uint32_t k = t; k < tile; k += LOG_TB) {
    uint32_t v;
    do { v = __hip_atomic_load(&pub[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while (!(v >> 31));
    part += v & 0x7FFFFFFFu;
  }
  uint32_t tot;
  wave_excl_scan_u32(part, &tot);
  if (lane == 0) Swp[wv] = tot;
  __syncthreads();  // (also: the patched type bytes are in LDS)
  uint32_t base = 0;
  for (uint32_t k = 0; k < LOG_TB / 64; k++) base += Swp[k];
  if (valid) {
    const uint32_t pos = (uint32_t)(((uint64_t)tail0 + base + before + (uint32_t)__popcll(vm & lanemask_lt())) % log.cap);
    uint4 *e = (uint4 *)(log.ring + (size_t)pos * 64);
    e[0] = make_uint4(w[0], w[1], w[2], w[3]);
    e[1] = make_uint4(w[4], w[5], w[6], w[7]);
    e[2] = make_uint4(w[8], w[9], w[10], w[11]);
    e[3] = make_uint4(w[12], 0u, 0u, 0u);  // ver; is_del = 0, table = 0
  }
  for (uint32_t k = t; k < nv; k += LOG_TB) ((uint4 *)dst)[k] = ((const uint4 *)Sm)[k];
  for (uint32_t k = nv * 16 + t; k < bytes; k += LOG_TB) dst[k] = Sm[k];
  if (t == 0) {
    if (tile == n_tiles - 1) {  // the pass's new tail ...
      const uint32_t total = base + tile_total;
      __hip_atomic_store(&log.tail[1], (uint32_t)(((uint64_t)tail0 + total) % log.cap), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      *(unsigned long long *)(log.tail + 2) += total;  // records ever appended (dint_log_drain)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    // ... becomes current when the last tile is done: every tile has read the old one
    if (atomicAdd(&pub[1025], 1u) == n_tiles - 1)
      log.tail[0] = __hip_atomic_load(&log.tail[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

void dint_launch_log(const void *d_req, void *d_rep, uint32_t n, dint_log log, dint_scratch s, hipStream_t st,
                     hipEvent_t *ev) {
  if (n == 0) return;
  if (ev) hipEventRecord(ev[0], st);
  if (n <= LOG_SMALL_MAX) {
    const uint32_t nt = (n + LOG_TB_SMALL - 1) / LOG_TB_SMALL;  // <= 1024 tiles
    hipLaunchKernelGGL((k_log_append<LOG_TB_SMALL>), dim3(nt), dim3(LOG_TB_SMALL), 0, st, (const uint8_t *)d_req, (uint8_t *)d_rep, n, nt,
                       log, s.blk_pub, s.blk_pub_next, s.stats);
  } else {
    const uint32_t nt = (n + LOG_TB_BIG - 1) / LOG_TB_BIG;  // <= 1024 tiles for n <= 2^20
    hipLaunchKernelGGL((k_log_append<LOG_TB_BIG>), dim3(nt), dim3(LOG_TB_BIG), 0, st, (const uint8_t *)d_req, (uint8_t *)d_rep, n, nt,
                       log, s.blk_pub, s.blk_pub_next, s.stats);
  }
  if (ev) hipEventRecord(ev[1], st);
}
