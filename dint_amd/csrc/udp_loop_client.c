/* udp_loop_client.c -- MEASUREMENT TOOL (built as dint_amd/dint_udp_client; the baseline harness builds the same source).
 *
 * A closed-loop UDP load generator on loopback, for the host shim (dint_udp_server) and for the reference's as-shipped
 * udp/ servers alike: the Caladan clients (lock_fasst/caladan/client.cc) cannot be built offline,
 * so this stands in for them on the wire.
 * Each thread owns one UDP socket and keeps `window` requests of a recorded request stream outstanding: every
 * reply releases the next request (the closed loop of client.cc:183-280, minus the transaction logic -- the stream
 * was recorded from that logic, and any request is legal for the server in any state).  A request whose reply does
 * not arrive within 20 ms is counted lost and its window slot is reused (UDP may drop under overload).
 *
 * usage: udp_loop_client <requests.bin> <msg_size> <port> <threads> <window> <warmup_s> <measure_s> [server ip = 127.0.0.1]
 * stdout: one JSON line {"replies":..., "seconds":..., "ops_per_s":..., "lost":..., "threads":..., "window":...}
 */
#define _GNU_SOURCE
#include <arpa/inet.h>
#include <errno.h>
#include <fcntl.h>
#include <poll.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/socket.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#define VEC 32
#define MAXMSG 128

static unsigned char *g_req;
static size_t g_n, g_msg;
static int g_port, g_threads, g_window;
static uint32_t g_host = 0x7F000001u; /* 127.0.0.1 */
static volatile int g_phase; /* 0 warm-up, 1 measured, 2 stop */

typedef struct { pthread_t th; int id; uint64_t replies, lost; } worker;

static double now_s(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

static void *run(void *arg) {
  worker *w = (worker *)arg;
  int fd = socket(AF_INET, SOCK_DGRAM, 0);
  struct sockaddr_in srv;
  memset(&srv, 0, sizeof srv);
  srv.sin_family = AF_INET;
  srv.sin_port = htons((uint16_t)g_port);
  srv.sin_addr.s_addr = htonl(g_host);
  if (fd < 0 || connect(fd, (struct sockaddr *)&srv, sizeof srv) < 0) { perror("client socket"); return NULL; }
  int sz = 4 << 20;
  if (setsockopt(fd, SOL_SOCKET, SO_RCVBUFFORCE, &sz, sizeof sz) < 0) setsockopt(fd, SOL_SOCKET, SO_RCVBUF, &sz, sizeof sz);  /* (SO_RCVBUF alone is clamped to net.core.rmem_max: ~270 datagrams) */
  if (setsockopt(fd, SOL_SOCKET, SO_SNDBUFFORCE, &sz, sizeof sz) < 0) setsockopt(fd, SOL_SOCKET, SO_SNDBUF, &sz, sizeof sz);

  size_t next = (size_t)w->id, stride = (size_t)g_threads; /* thread t replays requests t, t+T, t+2T, ... */
  struct mmsghdr sm[VEC], rm[VEC];
  struct iovec si[VEC], ri[VEC];
  unsigned char rbuf[VEC][MAXMSG];
  memset(sm, 0, sizeof sm);
  memset(rm, 0, sizeof rm);
  for (int i = 0; i < VEC; i++) {
    ri[i].iov_base = rbuf[i]; ri[i].iov_len = MAXMSG;
    rm[i].msg_hdr.msg_iov = &ri[i]; rm[i].msg_hdr.msg_iovlen = 1;
    sm[i].msg_hdr.msg_iov = &si[i]; sm[i].msg_hdr.msg_iovlen = 1;
  }
  int outstanding = 0;
  uint64_t replies = 0, lost = 0;
  int counted = 0;
  while (g_phase != 2) {
    if (g_phase == 1 && !counted) { replies = lost = 0; counted = 1; }
    int want = g_window - outstanding;
    while (want > 0) {
      int k = want < VEC ? want : VEC;
      for (int i = 0; i < k; i++) {
        si[i].iov_base = g_req + (next % g_n) * g_msg; si[i].iov_len = g_msg;
        next += stride;
      }
      int s = sendmmsg(fd, sm, (unsigned)k, 0);
      if (s <= 0) { if (errno == EINTR || errno == EAGAIN || errno == ENOBUFS || errno == ECONNREFUSED) break; perror("sendmmsg"); return NULL; }
      outstanding += s; want -= s;
      if (s < k) { next -= (size_t)(k - s) * stride; break; }
    }
    struct pollfd p = {fd, POLLIN, 0};
    int pr = poll(&p, 1, 20);
    if (pr == 0) { lost += (uint64_t)outstanding; outstanding = 0; continue; } /* window timed out: refill */
    if (pr < 0) continue;
    int r = recvmmsg(fd, rm, VEC, MSG_DONTWAIT, NULL);
    if (r > 0) { replies += (uint64_t)r; outstanding -= r; if (outstanding < 0) outstanding = 0; }
  }
  w->replies = replies; w->lost = lost;
  close(fd);
  return NULL;
}

int main(int argc, char **argv) {
  if (argc != 8 && argc != 9) { fprintf(stderr, "usage: %s <requests.bin> <msg_size> <port> <threads> <window> <warmup_s> <measure_s> [server ip]\n", argv[0]); return 2; }
  if (argc == 9) {
    struct in_addr ia;
    if (inet_pton(AF_INET, argv[8], &ia) != 1) { fprintf(stderr, "bad server ip %s\n", argv[8]); return 2; }
    g_host = ntohl(ia.s_addr);
  }
  g_msg = (size_t)atoi(argv[2]); g_port = atoi(argv[3]); g_threads = atoi(argv[4]); g_window = atoi(argv[5]);
  double warm = atof(argv[6]), meas = atof(argv[7]);
  if (g_msg == 0 || g_msg > MAXMSG || g_threads < 1 || g_window < 1) return 2;
  int fd = open(argv[1], O_RDONLY);
  struct stat st;
  if (fd < 0 || fstat(fd, &st) < 0) { perror("requests"); return 2; }
  g_n = (size_t)st.st_size / g_msg;
  if (g_n == 0) return 2;
  g_req = (unsigned char *)malloc((size_t)st.st_size);
  for (size_t got = 0; got < (size_t)st.st_size;) {
    ssize_t r = read(fd, g_req + got, (size_t)st.st_size - got);
    if (r <= 0) { perror("read"); return 2; }
    got += (size_t)r;
  }
  close(fd);
  worker *ws = (worker *)calloc((size_t)g_threads, sizeof(worker));
  for (int i = 0; i < g_threads; i++) { ws[i].id = i; pthread_create(&ws[i].th, NULL, run, &ws[i]); }
  struct timespec d = {(time_t)warm, (long)((warm - (time_t)warm) * 1e9)};
  nanosleep(&d, NULL);
  double t0 = now_s();
  g_phase = 1;
  d.tv_sec = (time_t)meas; d.tv_nsec = (long)((meas - (time_t)meas) * 1e9);
  nanosleep(&d, NULL);
  g_phase = 2;
  double dt = now_s() - t0;
  uint64_t rep = 0, lost = 0;
  for (int i = 0; i < g_threads; i++) { pthread_join(ws[i].th, NULL); rep += ws[i].replies; lost += ws[i].lost; }
  printf("{\"replies\": %llu, \"seconds\": %.4f, \"ops_per_s\": %.1f, \"lost\": %llu, \"threads\": %d, \"window\": %d}\n",
         (unsigned long long)rep, dt, rep / dt, (unsigned long long)lost, g_threads, g_window);
  return 0;
}
