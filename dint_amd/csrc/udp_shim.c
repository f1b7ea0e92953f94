/*
 * udp_shim.c -- the thin host shim that makes the GPU engine a drop-in for the reference servers:
 * it owns the UDP sockets, speaks the reference's wire protocol, and hands fixed-format request
 * batches across the C ABI (include/dint_abi.h).  Plain C; links against libdint.so.
 *
 * What it replaces in the reference (one process per server, same command line spirit):
 *   lock_fasst/udp/server.cc:49-120     server_loop: recvfrom -> switch -> sendto on port 20230
 *   tatp/udp/server_shard.cc:88-211     server_handler (13 request types), :241-274 cpu monitor on 20231
 *   smallbank/udp/server_shard.cc        same shape
 * One datagram = one packed request struct; the reply is the same struct, sent to the datagram's source.
 * Datagrams of the wrong size are dropped (the reference would read garbage).
 *
 * Batching: recvmmsg() drains the socket into a batch; the batch closes when it holds `--batch` requests
 * or `--deadline-us` after its first request, whichever comes first, then dint_submit() runs it on the GPU
 * and sendmmsg() returns the replies.  Arrival order in the socket = request order in the batch, so the
 * engine's serial-equivalence contract gives clients exactly the semantics of a single-threaded reference
 * server.  Port+1 answers the clients' end-of-run CPU-usage query (16 bytes {double ucores, kcores},
 * tatp/caladan/client_udp_shard.cc:75-92) so unmodified clients do not hang in CollectStat.
 *
 *   dint_udp_server --workload {fasst|2pl|log|store|tatp|smallbank} [--rows N] [--slots N] [--populate N]
 *                   [--bind 10.10.1.1] [--port 20230] [--batch 4096] [--deadline-us 100] [--device 0]
 */
#define _GNU_SOURCE
#include <arpa/inet.h>
#include <errno.h>
#include <netinet/in.h>
#include <pthread.h>
#include <signal.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/resource.h>
#include <sys/socket.h>
#include <sys/time.h>
#include <time.h>
#include <unistd.h>

#include "../../include/dint_abi.h"

#define VLEN 1024     /* datagrams per recvmmsg / sendmmsg call */
#define MAX_MSG 64    /* >= the largest wire struct (55 B) */

static volatile sig_atomic_t g_stop = 0;
static void on_signal(int s) { (void)s; g_stop = 1; }

static uint64_t now_us(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (uint64_t)ts.tv_sec * 1000000ull + (uint64_t)ts.tv_nsec / 1000;
}

struct options {
  uint32_t workload;
  uint64_t rows, slots, populate;
  int have_populate;
  const char *bind_ip;
  int port, device;
  uint32_t batch, deadline_us;
};

static int parse_workload(const char *s, uint32_t *out) {
  static const char *names[] = {"fasst", "2pl", "log", "store", "tatp", "smallbank"};
  for (uint32_t i = 0; i < DINT_WL_COUNT; i++)
    if (strcmp(s, names[i]) == 0) { *out = i; return 0; }
  return -1;
}

/* ---- port+1: CPU usage echo (tatp/udp/server_shard.cc:213-274) ------------------------------------ */
struct mon_arg { const char *ip; int port; };
static void *monitor_thread(void *p) {
  struct mon_arg *a = (struct mon_arg *)p;
  int fd = socket(AF_INET, SOCK_DGRAM, 0);
  if (fd < 0) return NULL;
  struct sockaddr_in addr;
  memset(&addr, 0, sizeof addr);
  addr.sin_family = AF_INET;
  addr.sin_port = htons((uint16_t)a->port);
  inet_pton(AF_INET, a->ip, &addr.sin_addr);
  if (bind(fd, (struct sockaddr *)&addr, sizeof addr) < 0) { close(fd); return NULL; }
  struct timeval tv = {0, 200000};
  setsockopt(fd, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof tv);
  struct rusage last;
  getrusage(RUSAGE_SELF, &last);
  uint64_t last_t = now_us();
  while (!g_stop) {
    struct { double ucores, kcores; } msg;
    struct sockaddr_in cli;
    socklen_t len = sizeof cli;
    ssize_t r = recvfrom(fd, &msg, sizeof msg, 0, (struct sockaddr *)&cli, &len);
    if (r < 0) continue;
    /* cores used by this server process since the previous query (the reference samples /proc/stat
     * of its 16 pinned cores once a second; a process-wide rusage delta is the same quantity) */
    struct rusage ru;
    getrusage(RUSAGE_SELF, &ru);
    const uint64_t t = now_us();
    const double dt = (double)(t - last_t) / 1e6 + 1e-9;
    msg.ucores = ((double)(ru.ru_utime.tv_sec - last.ru_utime.tv_sec) + (ru.ru_utime.tv_usec - last.ru_utime.tv_usec) / 1e6) / dt;
    msg.kcores = ((double)(ru.ru_stime.tv_sec - last.ru_stime.tv_sec) + (ru.ru_stime.tv_usec - last.ru_stime.tv_usec) / 1e6) / dt;
    last = ru;
    last_t = t;
    sendto(fd, &msg, sizeof msg, 0, (struct sockaddr *)&cli, len);
  }
  close(fd);
  return NULL;
}

int main(int argc, char **argv) {
  struct options o = {DINT_WL_FASST, 0, 0, 0, 0, "10.10.1.1", 20230, 0, 4096, 100};
  for (int i = 1; i < argc; i++) {
    const char *a = argv[i], *v = (i + 1 < argc) ? argv[i + 1] : NULL;
#define NEED_V if (!v) { fprintf(stderr, "%s needs a value\n", a); return 2; } i++
    if (!strcmp(a, "--workload")) { NEED_V; if (parse_workload(v, &o.workload)) { fprintf(stderr, "unknown workload %s\n", v); return 2; } }
    else if (!strcmp(a, "--rows")) { NEED_V; o.rows = strtoull(v, NULL, 10); }
    else if (!strcmp(a, "--slots")) { NEED_V; o.slots = strtoull(v, NULL, 10); }
    else if (!strcmp(a, "--populate")) { NEED_V; o.populate = strtoull(v, NULL, 10); o.have_populate = 1; }
    else if (!strcmp(a, "--bind")) { NEED_V; o.bind_ip = v; }
    else if (!strcmp(a, "--port")) { NEED_V; o.port = atoi(v); }
    else if (!strcmp(a, "--device")) { NEED_V; o.device = atoi(v); }
    else if (!strcmp(a, "--batch")) { NEED_V; o.batch = (uint32_t)strtoul(v, NULL, 10); }
    else if (!strcmp(a, "--deadline-us")) { NEED_V; o.deadline_us = (uint32_t)strtoul(v, NULL, 10); }
    else { fprintf(stderr, "unknown option %s\n", a); return 2; }
  }
  if (o.batch == 0 || o.batch > DINT_MICRO_BATCH) o.batch = DINT_MICRO_BATCH;

  dint_config cfg;
  memset(&cfg, 0, sizeof cfg);
  cfg.abi_version = DINT_ABI_VERSION;
  cfg.workload = o.workload;
  cfg.device = o.device;
  cfg.n_slots = o.slots;
  cfg.n_rows = o.rows;
  dint_engine_t *eng = NULL;
  if (dint_engine_create(&cfg, &eng)) { fprintf(stderr, "dint_engine_create: %s\n", dint_last_error()); return 1; }
  const int msg_size = dint_msg_size(o.workload);
  if (o.workload >= DINT_WL_STORE) {  /* the reference servers populate at start-up (server_shard.cc:71-85) */
    const uint64_t n = o.have_populate ? o.populate
                     : (o.rows ? o.rows : (o.workload == DINT_WL_STORE ? 2000000ull : o.workload == DINT_WL_TATP ? 7000000ull : 24000000ull));
    if (n && dint_populate(eng, n)) { fprintf(stderr, "dint_populate: %s\n", dint_last_error()); return 1; }
  }

  int fd = socket(AF_INET, SOCK_DGRAM, 0);
  if (fd < 0) { perror("socket"); return 1; }
  int one = 1, buf = 64 << 20;
  setsockopt(fd, SOL_SOCKET, SO_REUSEPORT, &one, sizeof one);  /* as the reference: lock_fasst/udp/server.cc:57-58 */
  setsockopt(fd, SOL_SOCKET, SO_RCVBUF, &buf, sizeof buf);
  setsockopt(fd, SOL_SOCKET, SO_SNDBUF, &buf, sizeof buf);
  struct sockaddr_in addr;
  memset(&addr, 0, sizeof addr);
  addr.sin_family = AF_INET;
  addr.sin_port = htons((uint16_t)o.port);
  if (inet_pton(AF_INET, o.bind_ip, &addr.sin_addr) != 1) { fprintf(stderr, "bad --bind %s\n", o.bind_ip); return 2; }
  if (bind(fd, (struct sockaddr *)&addr, sizeof addr) < 0) { perror("bind"); return 1; }
  struct timeval tv = {0, 100000};  /* wake up to notice signals */
  setsockopt(fd, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof tv);

  signal(SIGINT, on_signal);
  signal(SIGTERM, on_signal);
  pthread_t mon;
  struct mon_arg ma = {o.bind_ip, o.port + 1};
  pthread_create(&mon, NULL, monitor_thread, &ma);

  uint8_t *reqs = (uint8_t *)malloc((size_t)o.batch * MAX_MSG);
  uint8_t *reps = (uint8_t *)malloc((size_t)o.batch * MAX_MSG);
  struct sockaddr_in *peers = (struct sockaddr_in *)malloc((size_t)o.batch * sizeof *peers);
  uint8_t (*rx)[MAX_MSG] = malloc((size_t)VLEN * MAX_MSG);
  struct mmsghdr *mm = (struct mmsghdr *)calloc(VLEN, sizeof *mm);
  struct iovec *iov = (struct iovec *)calloc(VLEN, sizeof *iov);
  struct sockaddr_in *from = (struct sockaddr_in *)calloc(VLEN, sizeof *from);
  if (!reqs || !reps || !peers || !rx || !mm || !iov || !from) { fprintf(stderr, "out of memory\n"); return 1; }

  fprintf(stdout, "dint_udp_server ready workload=%u msg=%d %s:%d batch=%u deadline_us=%u\n", o.workload, msg_size,
          o.bind_ip, o.port, o.batch, o.deadline_us);
  fflush(stdout);

  uint64_t total = 0, batches = 0, dropped = 0;
  while (!g_stop) {
    uint32_t n = 0;
    uint64_t t_first = 0;
    while (n < o.batch && !g_stop) {
      const uint32_t want = (o.batch - n) < VLEN ? (o.batch - n) : VLEN;
      for (uint32_t k = 0; k < want; k++) {
        iov[k].iov_base = rx[k];
        iov[k].iov_len = MAX_MSG;
        memset(&mm[k].msg_hdr, 0, sizeof mm[k].msg_hdr);
        mm[k].msg_hdr.msg_iov = &iov[k];
        mm[k].msg_hdr.msg_iovlen = 1;
        mm[k].msg_hdr.msg_name = &from[k];
        mm[k].msg_hdr.msg_namelen = sizeof from[k];
      }
      /* first datagram of a batch: block; afterwards only take what is already queued */
      const int got = recvmmsg(fd, mm, want, n == 0 ? MSG_WAITFORONE : MSG_DONTWAIT, NULL);
      if (got <= 0) {
        if (n == 0) continue;                                   /* idle: keep waiting */
        if (now_us() - t_first >= o.deadline_us) break;         /* deadline: close the batch */
        continue;
      }
      if (n == 0) t_first = now_us();
      for (int k = 0; k < got; k++) {
        if ((int)mm[k].msg_len != msg_size) { dropped++; continue; }
        memcpy(reqs + (size_t)n * msg_size, rx[k], (size_t)msg_size);
        peers[n] = from[k];
        n++;
      }
      if (now_us() - t_first >= o.deadline_us) break;
    }
    if (n == 0) continue;
    if (dint_submit(eng, reqs, n, reps)) { fprintf(stderr, "dint_submit: %s\n", dint_last_error()); break; }
    for (uint32_t off = 0; off < n;) {
      const uint32_t cnt = (n - off) < VLEN ? (n - off) : VLEN;
      for (uint32_t k = 0; k < cnt; k++) {
        iov[k].iov_base = reps + (size_t)(off + k) * msg_size;
        iov[k].iov_len = (size_t)msg_size;
        memset(&mm[k].msg_hdr, 0, sizeof mm[k].msg_hdr);
        mm[k].msg_hdr.msg_iov = &iov[k];
        mm[k].msg_hdr.msg_iovlen = 1;
        mm[k].msg_hdr.msg_name = &peers[off + k];
        mm[k].msg_hdr.msg_namelen = sizeof peers[off + k];
      }
      int sent = sendmmsg(fd, mm, cnt, 0);
      if (sent < 0) { if (errno == EINTR) continue; perror("sendmmsg"); sent = (int)cnt; }
      off += (uint32_t)sent;
    }
    total += n;
    batches++;
  }
  fprintf(stdout, "dint_udp_server exit requests=%llu batches=%llu dropped=%llu\n", (unsigned long long)total,
          (unsigned long long)batches, (unsigned long long)dropped);
  pthread_join(mon, NULL);
  dint_engine_destroy(eng);
  close(fd);
  return 0;
}
