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
 * Threads: `--threads` socket threads (default 2), each with its own SO_REUSEPORT socket on the same port exactly as
 * the reference's workers (lock_fasst/udp/server.cc:54-73,136-146); the kernel spreads client flows over them.
 * Batching: recvmmsg() drains a socket into a page-locked batch buffer; the batch closes when it holds `--batch`
 * requests or `--deadline-us` after its first request, whichever comes first, and goes to the GPU with
 * dint_submit_async().  Every thread keeps two batches: while one is on the GPU the next one is being received, and
 * its replies leave with sendmmsg() as soon as dint_wait() returns.  Arrival order in a socket = request order in
 * the batch, and the engine applies submissions in call order, so clients see one serial server.  `--shed`: a batch
 * that closes while both of the thread's batches are still busy is answered at once with the eBPF servers'
 * "not now" replies (dint_refuse: REJECT_READ / REJECT_LOCK / REJECT_COMMIT / RETRY ...) instead of waiting; requests
 * that have no such reply (ABORT, log appends) stay queued.  Port+1 answers the clients' end-of-run CPU-usage query
 * (16 bytes {double ucores, kcores}, tatp/caladan/client_udp_shard.cc:75-92) so unmodified clients do not hang in
 * CollectStat.
 *
 * `--caladan`: the port handshake of the reference's Caladan servers, which the client_caladan* binaries expect
 * (lock_fasst/caladan/server.cc:93-132, proto.h:38-45): the well-known port takes `net_req {int nports}` and answers
 * `net_resp {int nports; u16 ports[nports]}` after opening one data socket per requested port (the reference starts a
 * ServerLoop uthread per port; here the new sockets join the socket threads' epoll sets round robin).  Requests then
 * arrive on the data ports and every reply leaves from the port its request came to.
 *
 *   dint_udp_server --workload {fasst|2pl|log|store|tatp|smallbank} [--rows N] [--slots N] [--populate N]
 *                   [--bind 10.10.1.1] [--port 20230] [--batch 4096] [--deadline-us 100] [--threads 2] [--shed]
 *                   [--caladan [--idle-s 120]] [--device 0]
 */
#define _GNU_SOURCE
#include <arpa/inet.h>
#include <errno.h>
#include <netinet/in.h>
#include <poll.h>
#include <pthread.h>
#include <signal.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/epoll.h>
#include <sys/resource.h>
#include <sys/socket.h>
#include <sys/time.h>
#include <time.h>
#include <unistd.h>

#include "../../include/dint_abi.h"

#define VLEN 1024     /* datagrams per recvmmsg / sendmmsg call */
#define MAX_MSG 64    /* >= the largest wire struct (55 B) */

static volatile sig_atomic_t g_stop = 0;
/* --caladan: when each data socket (indexed by fd) last received a datagram; written by the socket threads, read by the
 * control thread, which closes sockets nobody has used for --idle-s seconds (a stale value only delays a reap) */
static volatile uint64_t *g_fd_seen = NULL;
static uint64_t g_fd_cap = 0;
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
  uint32_t batch, deadline_us, threads;
  int shed, caladan;
  uint32_t idle_s;               /* --caladan: a data socket idle for this long is closed (0 = never) */
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

/* ---- socket threads ------------------------------------------------------------------------------------- */
struct batch_slot {
  uint8_t *reqs, *reps;          /* page-locked (dint_alloc_pinned) */
  struct sockaddr_in *peers;
  int *fds;                      /* --caladan: the data socket each request arrived on (its reply leaves from it) */
  uint32_t n;
  dint_ticket ticket;
  int busy;
};
struct worker {
  pthread_t th;
  int id, fd, msg_size;
  int efd;                       /* --caladan: epoll set of this thread's data sockets (fd is unused then) */
  const struct options *o;
  dint_engine_t *eng;
  struct batch_slot slot[2];
  uint64_t requests, batches, dropped, refused, send_dropped;
};

/* replies [0, n) to their peers; would-block -> wait for the socket, a dead destination -> skip that one reply */
static void send_replies(struct worker *w, const uint8_t *reps, const struct sockaddr_in *peers, const int *fds, uint32_t n,
                         struct mmsghdr *mm, struct iovec *iov) {
  for (uint32_t off = 0; off < n && !g_stop;) {
    uint32_t cnt = (n - off) < VLEN ? (n - off) : VLEN;
    const int fd = fds ? fds[off] : w->fd;
    if (fds) {  /* one sendmmsg per run of replies that leave from the same data socket */
      uint32_t run = 1;
      while (run < cnt && fds[off + run] == fd) run++;
      cnt = run;
    }
    for (uint32_t k = 0; k < cnt; k++) {
      iov[k].iov_base = (void *)(reps + (size_t)(off + k) * w->msg_size);
      iov[k].iov_len = (size_t)w->msg_size;
      memset(&mm[k].msg_hdr, 0, sizeof mm[k].msg_hdr);
      mm[k].msg_hdr.msg_iov = &iov[k];
      mm[k].msg_hdr.msg_iovlen = 1;
      mm[k].msg_hdr.msg_name = (void *)&peers[off + k];
      mm[k].msg_hdr.msg_namelen = sizeof peers[off + k];
    }
    const int sent = sendmmsg(fd, mm, cnt, 0);
    if (sent > 0) { off += (uint32_t)sent; continue; }
    if (sent < 0 && errno == EINTR) continue;
    if (sent == 0 || errno == EAGAIN || errno == EWOULDBLOCK || errno == ENOBUFS) {
      struct pollfd pf = {fd, POLLOUT, 0};
      poll(&pf, 1, 100);
      continue;
    }
    w->send_dropped++;  /* per-destination error (e.g. ECONNREFUSED from an earlier ICMP): give up on this reply only */
    off++;
  }
}

static void complete(struct worker *w, struct batch_slot *b, struct mmsghdr *mm, struct iovec *iov) {
  if (dint_wait(w->eng, b->ticket)) fprintf(stderr, "dint_wait: %s\n", dint_last_error());  /* replies are valid */
  send_replies(w, b->reps, b->peers, w->o->caladan ? b->fds : NULL, b->n, mm, iov);
  w->requests += b->n;
  w->batches++;
  b->busy = 0;
}

static void *socket_thread(void *arg) {
  struct worker *w = (struct worker *)arg;
  const struct options *o = w->o;
  const int msg_size = w->msg_size;
  uint8_t (*rx)[MAX_MSG] = malloc((size_t)VLEN * MAX_MSG);
  struct mmsghdr *mm = (struct mmsghdr *)calloc(VLEN, sizeof *mm);
  struct iovec *iov = (struct iovec *)calloc(VLEN, sizeof *iov);
  struct sockaddr_in *from = (struct sockaddr_in *)calloc(VLEN, sizeof *from);
  uint8_t *shed_tmp = (uint8_t *)malloc((size_t)o->batch * MAX_MSG);
  struct sockaddr_in *shed_peers = (struct sockaddr_in *)malloc((size_t)o->batch * sizeof *shed_peers);
  int *shed_fds = (int *)malloc((size_t)o->batch * sizeof *shed_fds);
  if (!rx || !mm || !iov || !from || !shed_tmp || !shed_peers || !shed_fds) { fprintf(stderr, "out of memory\n"); g_stop = 1; return NULL; }
  int cur = 0;
  while (!g_stop) {
    struct batch_slot *b = &w->slot[cur], *other = &w->slot[cur ^ 1];
    if (b->busy) complete(w, b, mm, iov);
    uint32_t n = 0;
    uint64_t t_first = 0;
    while (n < o->batch && !g_stop) {
      const uint32_t want = (o->batch - n) < VLEN ? (o->batch - n) : VLEN;
      for (uint32_t k = 0; k < want; k++) {
        iov[k].iov_base = rx[k];
        iov[k].iov_len = MAX_MSG;
        memset(&mm[k].msg_hdr, 0, sizeof mm[k].msg_hdr);
        mm[k].msg_hdr.msg_iov = &iov[k];
        mm[k].msg_hdr.msg_iovlen = 1;
        mm[k].msg_hdr.msg_name = &from[k];
        mm[k].msg_hdr.msg_namelen = sizeof from[k];
      }
      /* first datagram of a batch: block -- unless the other batch is on the GPU, whose replies must not wait for
       * new traffic; afterwards only take what is already queued */
      const int flags = (n == 0 && !other->busy) ? MSG_WAITFORONE : MSG_DONTWAIT;
      int got, rfd = w->fd;
      if (o->caladan) {  /* any of this thread's data sockets: take what one ready socket holds */
        struct epoll_event ev;
        const int nr = epoll_wait(w->efd, &ev, 1, (flags & MSG_DONTWAIT) ? 0 : 100);
        rfd = nr > 0 ? ev.data.fd : -1;
        got = nr > 0 ? recvmmsg(rfd, mm, want, MSG_DONTWAIT, NULL) : -1;
        if (got > 0 && g_fd_seen && rfd >= 0 && (uint64_t)rfd < g_fd_cap) g_fd_seen[rfd] = now_us();  /* the control thread reaps idle data sockets */
      } else {
        got = recvmmsg(w->fd, mm, want, flags, NULL);
      }
      if (got <= 0) {
        if (n == 0) {
          if (other->busy) complete(w, other, mm, iov);       /* idle socket: finish what is in flight */
          continue;
        }
        if (now_us() - t_first >= o->deadline_us) break;      /* deadline: close the batch */
        continue;
      }
      if (n == 0) t_first = now_us();
      for (int k = 0; k < got; k++) {
        if ((int)mm[k].msg_len != msg_size) { w->dropped++; continue; }
        memcpy(b->reqs + (size_t)n * msg_size, rx[k], (size_t)msg_size);
        b->peers[n] = from[k];
        b->fds[n] = rfd;
        n++;
      }
      if (now_us() - t_first >= o->deadline_us) break;
    }
    if (n == 0) continue;
    if (o->shed && other->busy) {
      /* both batches would be in flight: tell the senders "not now" (they resend) rather than queue behind the GPU;
       * requests that have no such reply stay in the batch */
      dint_refuse(o->workload, b->reqs, n, shed_tmp);
      uint32_t keep = 0, ref = 0;
      for (uint32_t k = 0; k < n; k++) {
        if (memcmp(shed_tmp + (size_t)k * msg_size, b->reqs + (size_t)k * msg_size, (size_t)msg_size) != 0) {
          memmove(shed_tmp + (size_t)ref * msg_size, shed_tmp + (size_t)k * msg_size, (size_t)msg_size);
          shed_fds[ref] = b->fds[k];
          shed_peers[ref++] = b->peers[k];
        } else {
          memmove(b->reqs + (size_t)keep * msg_size, b->reqs + (size_t)k * msg_size, (size_t)msg_size);
          b->fds[keep] = b->fds[k];
          b->peers[keep++] = b->peers[k];
        }
      }
      send_replies(w, shed_tmp, shed_peers, o->caladan ? shed_fds : NULL, ref, mm, iov);
      w->refused += ref;
      n = keep;
      if (n == 0) continue;
    }
    b->n = n;
    if (dint_submit_async(w->eng, b->reqs, n, b->reps, &b->ticket)) {
      fprintf(stderr, "dint_submit_async: %s\n", dint_last_error());
      g_stop = 1;
      break;
    }
    b->busy = 1;
    cur ^= 1;
  }
  for (int k = 0; k < 2; k++)
    if (w->slot[k].busy) complete(w, &w->slot[k], mm, iov);
  free(rx); free(mm); free(iov); free(from); free(shed_tmp); free(shed_peers); free(shed_fds);
  return NULL;
}

/* ---- --caladan: the control port (lock_fasst/caladan/server.cc:93-132) ------------------------------------------- */
struct control_arg { const struct options *o; struct worker *ws; };
struct dsock { int fd, efd; uint64_t seen; /* g_fd_seen when it was retired */ };
static void dsock_close(struct dsock *d) {
  epoll_ctl(d->efd, EPOLL_CTL_DEL, d->fd, NULL);
  close(d->fd);
}
/* The handshake is unauthenticated and every request opens up to 730 sockets, so the live set is bounded three ways
 * (ADVICE r03): a request that fails half way gives back what it opened; the total stays below the process's file
 * limit (raised to the hard limit at start-up) -- a request that does not fit is refused and logged, not half served;
 * sockets nobody has sent to for --idle-s seconds (clients that left) are closed. */
static void *control_thread(void *p) {
  struct control_arg *c = (struct control_arg *)p;
  const struct options *o = c->o;
  int fd = socket(AF_INET, SOCK_DGRAM, 0);
  struct sockaddr_in addr;
  memset(&addr, 0, sizeof addr);
  addr.sin_family = AF_INET;
  addr.sin_port = htons((uint16_t)o->port);
  inet_pton(AF_INET, o->bind_ip, &addr.sin_addr);
  if (fd < 0 || bind(fd, (struct sockaddr *)&addr, sizeof addr) < 0) { perror("control port"); g_stop = 1; return NULL; }
  struct timeval tv = {0, 100000};
  setsockopt(fd, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof tv);
  const uint64_t max_live = g_fd_cap > 64 + 8 * (uint64_t)o->threads ? g_fd_cap - 64 - 8 * (uint64_t)o->threads : 0;
  struct dsock *live = (struct dsock *)calloc(max_live ? max_live : 1, sizeof *live);
  /* idle sockets are retired in two steps: a socket thread may sit between its
   * epoll_wait() and its recvmmsg() on the descriptor, or hold it in a batch whose replies have not left yet -- closing it
   * there would hand the NUMBER to the next handshake's socket and send replies from the wrong port.  Step one takes the
   * descriptor out of its epoll set (nothing new is picked up from it); step two, a quarter of a second later (a socket
   * thread's loop takes milliseconds), closes it -- unless a datagram was taken from it in between (it arrived at the
   * deadline): then it goes back into service. */
  struct dsock *dying = (struct dsock *)calloc(max_live ? max_live : 1, sizeof *dying);
  uint64_t n_live = 0, n_dying = 0, last_reap = now_us(), t_retired = 0, refused = 0;
  uint32_t next = 0;
  if (!live || !dying) { fprintf(stderr, "out of memory\n"); g_stop = 1; close(fd); return NULL; }
  while (!g_stop) {
    const uint64_t t = now_us();
    if (n_dying && t - t_retired >= 250000ull) {
      for (uint64_t k = 0; k < n_dying; k++) {  /* step two */
        if (g_fd_seen[dying[k].fd] != dying[k].seen) {  /* used after all: back into its thread's epoll set */
          struct epoll_event ev;
          memset(&ev, 0, sizeof ev);
          ev.events = EPOLLIN;
          ev.data.fd = dying[k].fd;
          if (epoll_ctl(dying[k].efd, EPOLL_CTL_ADD, dying[k].fd, &ev) == 0) { live[n_live++] = dying[k]; continue; }
        }
        close(dying[k].fd);
      }
      n_dying = 0;
    }
    if (o->idle_s && !n_dying && t - last_reap >= 1000000ull) {  /* once a second: retire what has been idle for --idle-s */
      uint64_t keep = 0;
      for (uint64_t k = 0; k < n_live; k++) {
        if (t - g_fd_seen[live[k].fd] >= (uint64_t)o->idle_s * 1000000ull) {  /* step one */
          epoll_ctl(live[k].efd, EPOLL_CTL_DEL, live[k].fd, NULL);
          live[k].seen = g_fd_seen[live[k].fd];
          dying[n_dying++] = live[k];
        } else {
          live[keep++] = live[k];
        }
      }
      n_live = keep;
      last_reap = t_retired = t;
    }
    int32_t nports = 0;  /* net_req {int nports}, proto.h:38-40 */
    struct sockaddr_in cli;
    socklen_t len = sizeof cli;
    if (recvfrom(fd, &nports, sizeof nports, 0, (struct sockaddr *)&cli, &len) != (ssize_t)sizeof nports) continue;
    if (nports <= 0 || nports > 730) continue;  /* the answer must fit one datagram (server.cc:121-123: rt::UdpConn::kMaxPayloadSize) */
    if (n_live + n_dying + (uint64_t)nports > max_live) {
      if (refused++ % 64 == 0)
        fprintf(stderr, "handshake refused: %d ports asked, %llu of %llu data sockets live (raise `ulimit -n` or lower --idle-s)\n",
                nports, (unsigned long long)n_live, (unsigned long long)max_live);
      continue;  /* no answer: the client's handshake times out, as against a server that is not there */
    }
    uint8_t resp[4 + 2 * 730];
    memcpy(resp, &nports, 4);  /* net_resp {int nports; uint16_t ports[]}, proto.h:42-45 */
    const uint64_t first = n_live;
    int ok = 1;
    for (int32_t i = 0; i < nports && ok; i++) {
      int dfd = socket(AF_INET, SOCK_DGRAM | SOCK_NONBLOCK, 0);
      struct sockaddr_in da = addr;
      da.sin_port = 0;  /* any free port, as rt::UdpConn::Listen({0, 0}) */
      socklen_t dl = sizeof da;
      int buf = 4 << 20;
      if (dfd < 0 || (uint64_t)dfd >= g_fd_cap || bind(dfd, (struct sockaddr *)&da, sizeof da) < 0 ||
          getsockname(dfd, (struct sockaddr *)&da, &dl) < 0) {
        if (dfd >= 0) close(dfd);
        ok = 0;
        break;
      }
      if (setsockopt(dfd, SOL_SOCKET, SO_RCVBUFFORCE, &buf, sizeof buf) < 0) setsockopt(dfd, SOL_SOCKET, SO_RCVBUF, &buf, sizeof buf);  /* (SO_RCVBUF alone is clamped to net.core.rmem_max: ~270 datagrams) */
      if (setsockopt(dfd, SOL_SOCKET, SO_SNDBUFFORCE, &buf, sizeof buf) < 0) setsockopt(dfd, SOL_SOCKET, SO_SNDBUF, &buf, sizeof buf);
      const uint16_t port = ntohs(da.sin_port);  /* host order: Caladan's netaddr.port is host order (client_caladan.cc:305-308) */
      memcpy(resp + 4 + 2 * i, &port, 2);
      struct epoll_event ev;
      memset(&ev, 0, sizeof ev);
      ev.events = EPOLLIN;
      ev.data.fd = dfd;
      const int efd = c->ws[next++ % o->threads].efd;
      g_fd_seen[dfd] = now_us();
      if (epoll_ctl(efd, EPOLL_CTL_ADD, dfd, &ev) < 0) { close(dfd); ok = 0; break; }
      live[n_live].fd = dfd;
      live[n_live++].efd = efd;
    }
    if (ok) {
      sendto(fd, resp, (size_t)(4 + 2 * nports), 0, (struct sockaddr *)&cli, len);
    } else {  /* give back what this request opened; the client gets no answer and asks again */
      fprintf(stderr, "handshake failed after %llu of %d ports: %s\n", (unsigned long long)(n_live - first), nports, strerror(errno));
      while (n_live > first) dsock_close(&live[--n_live]);
    }
  }
  for (uint64_t k = 0; k < n_live; k++) dsock_close(&live[k]);
  for (uint64_t k = 0; k < n_dying; k++) close(dying[k].fd);
  free(dying);
  free(live);
  close(fd);
  return NULL;
}

int main(int argc, char **argv) {
  struct options o = {DINT_WL_FASST, 0, 0, 0, 0, "10.10.1.1", 20230, 0, 4096, 100, 2, 0, 0, 120};
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
    else if (!strcmp(a, "--threads")) { NEED_V; o.threads = (uint32_t)strtoul(v, NULL, 10); }
    else if (!strcmp(a, "--shed")) { o.shed = 1; }
    else if (!strcmp(a, "--caladan")) { o.caladan = 1; }
    else if (!strcmp(a, "--idle-s")) { NEED_V; o.idle_s = (uint32_t)strtoul(v, NULL, 10); }
    else { fprintf(stderr, "unknown option %s\n", a); return 2; }
  }
  const uint32_t batch_max = o.workload == DINT_WL_LOG ? DINT_MICRO_BATCH : DINT_KV_PASS_MAX;  /* one kernel pass */
  if (o.batch == 0 || o.batch > batch_max) o.batch = batch_max;
  if (o.threads == 0 || o.threads > 64) o.threads = 2;

  dint_config cfg;
  memset(&cfg, 0, sizeof cfg);
  cfg.abi_version = DINT_ABI_VERSION;
  cfg.workload = o.workload;
  cfg.device = o.device;
  cfg.n_slots = o.slots;
  cfg.n_rows = o.rows;
  cfg.flags = DINT_FLAG_COPY_STREAMS;  /* one engine in this process: overlap its copies with its kernels */
  dint_engine_t *eng = NULL;
  if (dint_engine_create(&cfg, &eng)) { fprintf(stderr, "dint_engine_create: %s\n", dint_last_error()); return 1; }
  const int msg_size = dint_msg_size(o.workload);
  if (o.workload >= DINT_WL_STORE) {  /* the reference servers populate at start-up (server_shard.cc:71-85) */
    const uint64_t n = o.have_populate ? o.populate
                     : (o.rows ? o.rows : (o.workload == DINT_WL_STORE ? 2000000ull : o.workload == DINT_WL_TATP ? 7000000ull : 24000000ull));
    if (n && dint_populate(eng, n)) { fprintf(stderr, "dint_populate: %s\n", dint_last_error()); return 1; }
  }
  struct sockaddr_in addr;
  memset(&addr, 0, sizeof addr);
  addr.sin_family = AF_INET;
  addr.sin_port = htons((uint16_t)o.port);
  if (inet_pton(AF_INET, o.bind_ip, &addr.sin_addr) != 1) { fprintf(stderr, "bad --bind %s\n", o.bind_ip); return 2; }

  signal(SIGINT, on_signal);
  signal(SIGTERM, on_signal);
  if (o.caladan) {  /* every client thread gets a data socket: take the file limit the process may have */
    struct rlimit rl;
    if (getrlimit(RLIMIT_NOFILE, &rl) == 0) {
      if (rl.rlim_cur < rl.rlim_max) { rl.rlim_cur = rl.rlim_max; setrlimit(RLIMIT_NOFILE, &rl); getrlimit(RLIMIT_NOFILE, &rl); }
      g_fd_cap = rl.rlim_cur == RLIM_INFINITY || rl.rlim_cur > (1u << 20) ? (1u << 20) : (uint64_t)rl.rlim_cur;
    } else {
      g_fd_cap = 1024;
    }
    g_fd_seen = (volatile uint64_t *)calloc(g_fd_cap, sizeof(uint64_t));
    if (!g_fd_seen) { fprintf(stderr, "out of memory\n"); return 1; }
  }
  struct worker *ws = (struct worker *)calloc(o.threads, sizeof *ws);
  if (!ws) { fprintf(stderr, "out of memory\n"); return 1; }
  for (uint32_t t = 0; t < o.threads; t++) {
    struct worker *w = &ws[t];
    w->id = (int)t; w->o = &o; w->eng = eng; w->msg_size = msg_size;
    w->fd = -1;
    w->efd = -1;
    if (o.caladan) {  /* data sockets appear with the clients' handshakes (control_thread) */
      w->efd = epoll_create1(0);
      if (w->efd < 0) { perror("epoll_create1"); return 1; }
    } else {
      w->fd = socket(AF_INET, SOCK_DGRAM, 0);
      if (w->fd < 0) { perror("socket"); return 1; }
      int one = 1, buf = 64 << 20;
      setsockopt(w->fd, SOL_SOCKET, SO_REUSEPORT, &one, sizeof one);  /* as the reference: lock_fasst/udp/server.cc:57-58 */
      if (setsockopt(w->fd, SOL_SOCKET, SO_RCVBUFFORCE, &buf, sizeof buf) < 0) setsockopt(w->fd, SOL_SOCKET, SO_RCVBUF, &buf, sizeof buf);  /* (SO_RCVBUF alone is clamped to net.core.rmem_max: ~270 datagrams) */
      if (setsockopt(w->fd, SOL_SOCKET, SO_SNDBUFFORCE, &buf, sizeof buf) < 0) setsockopt(w->fd, SOL_SOCKET, SO_SNDBUF, &buf, sizeof buf);
      if (bind(w->fd, (struct sockaddr *)&addr, sizeof addr) < 0) { perror("bind"); return 1; }
      struct timeval tv = {0, 100000};  /* wake up to notice signals */
      setsockopt(w->fd, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof tv);
    }
    for (int k = 0; k < 2; k++) {
      void *a = NULL, *b = NULL;
      if (dint_alloc_pinned((size_t)o.batch * MAX_MSG, &a) || dint_alloc_pinned((size_t)o.batch * MAX_MSG, &b)) {
        fprintf(stderr, "dint_alloc_pinned: %s\n", dint_last_error());
        return 1;
      }
      w->slot[k].reqs = (uint8_t *)a;
      w->slot[k].reps = (uint8_t *)b;
      w->slot[k].peers = (struct sockaddr_in *)malloc((size_t)o.batch * sizeof(struct sockaddr_in));
      w->slot[k].fds = (int *)malloc((size_t)o.batch * sizeof(int));
      if (!w->slot[k].peers || !w->slot[k].fds) { fprintf(stderr, "out of memory\n"); return 1; }
    }
  }
  pthread_t mon;
  struct mon_arg ma = {o.bind_ip, o.port + 1};
  pthread_create(&mon, NULL, monitor_thread, &ma);
  pthread_t ctl;
  struct control_arg ca = {&o, ws};
  if (o.caladan) pthread_create(&ctl, NULL, control_thread, &ca);
  for (uint32_t t = 0; t < o.threads; t++) pthread_create(&ws[t].th, NULL, socket_thread, &ws[t]);

  fprintf(stdout, "dint_udp_server ready workload=%u msg=%d %s:%d batch=%u deadline_us=%u threads=%u shed=%d caladan=%d\n", o.workload,
          msg_size, o.bind_ip, o.port, o.batch, o.deadline_us, o.threads, o.shed, o.caladan);
  fflush(stdout);

  uint64_t total = 0, batches = 0, dropped = 0, refused = 0, send_dropped = 0;
  for (uint32_t t = 0; t < o.threads; t++) {
    pthread_join(ws[t].th, NULL);
    total += ws[t].requests; batches += ws[t].batches; dropped += ws[t].dropped; refused += ws[t].refused;
    send_dropped += ws[t].send_dropped;
  }
  g_stop = 1;
  fprintf(stdout, "dint_udp_server exit requests=%llu batches=%llu dropped=%llu refused=%llu send_dropped=%llu\n",
          (unsigned long long)total, (unsigned long long)batches, (unsigned long long)dropped, (unsigned long long)refused,
          (unsigned long long)send_dropped);
  pthread_join(mon, NULL);
  if (o.caladan) pthread_join(ctl, NULL);
  for (uint32_t t = 0; t < o.threads; t++) {
    if (ws[t].fd >= 0) close(ws[t].fd);
    if (ws[t].efd >= 0) close(ws[t].efd);  /* (the data sockets die with the process) */
    for (int k = 0; k < 2; k++) { dint_free_pinned(ws[t].slot[k].reqs); dint_free_pinned(ws[t].slot[k].reps); free(ws[t].slot[k].peers); free(ws[t].slot[k].fds); }
  }
  dint_engine_destroy(eng);
  free(ws);
  return 0;
}
