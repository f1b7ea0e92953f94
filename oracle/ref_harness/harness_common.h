/*
 * harness_common.h -- replay harness for the UNMODIFIED reference udp/ servers.
 *
 * TEST INFRASTRUCTURE ONLY.  Each ref_<workload>.cc does
 *     #define main ref_main
 *     #include "server.cc"        // found via -I/root/reference/<wl>/udp, not copied
 * and then includes this header, which interposes the libc socket calls the
 * reference server makes (socket/setsockopt/bind/recvfrom/sendto, sched_getcpu):
 * recvfrom() hands out the next record of a trace file, sendto() captures the
 * reply.  Run with one server thread this is the exact serial oracle and the
 * "reference" CPU baseline; zero reference lines are changed or stored here.
 *
 * usage: ref_<wl> <trace.bin> <replies.bin> [state_dump.bin]
 * stdout: one JSON line {"n":..., "seconds":..., "ops_per_s":...}
 *
 * REF_TRACE_WAIT=1 in the environment: the server is started BEFORE its trace exists (tatp / smallbank populate for
 * minutes).  When the server's first recvfrom() arrives the harness creates <replies.bin>.ready, waits for
 * <trace.bin>.go to appear, and only then loads <trace.bin>; the clock starts after the load, as without the option.
 */
#pragma once
#include <fcntl.h>
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/socket.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#ifndef REF_MSG_SIZE
#error "define REF_MSG_SIZE (= sizeof(message) of the included server)"
#endif

static unsigned char *g_trace = nullptr, *g_replies = nullptr;
static size_t g_n = 0, g_rd = 0, g_wr = 0;
static const char *g_reply_path = nullptr, *g_dump_path = nullptr;
static double g_t0 = 0;
static const char *g_trace_path = nullptr;
static bool g_wait = false;

static double now_s() {
  timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

static void ref_dump_state(FILE *f); /* per-workload */

static void finish_and_exit() {
  double dt = now_s() - g_t0;
  FILE *f = fopen(g_reply_path, "wb");
  if (!f || fwrite(g_replies, REF_MSG_SIZE, g_wr, f) != g_wr) { perror("replies"); _Exit(2); }
  fclose(f);
  if (g_dump_path) {
    FILE *d = fopen(g_dump_path, "wb");
    if (!d) { perror("dump"); _Exit(2); }
    ref_dump_state(d);
    fclose(d);
  }
  printf("{\"n\": %zu, \"seconds\": %.6f, \"ops_per_s\": %.1f}\n", g_wr, dt, dt > 0 ? g_wr / dt : 0.0);
  fflush(stdout);
  _Exit(0);
}

static int load_trace() {
  int fd = open(g_trace_path, O_RDONLY);
  if (fd < 0) { perror("trace"); return 2; }
  struct stat st; fstat(fd, &st);
  g_n = st.st_size / REF_MSG_SIZE;
  g_trace = (unsigned char *)malloc(st.st_size ? st.st_size : 1);
  size_t got = 0;
  while (got < (size_t)st.st_size) {
    ssize_t r = read(fd, g_trace + got, st.st_size - got);
    if (r <= 0) { perror("read"); return 2; }
    got += r;
  }
  close(fd);
  g_replies = (unsigned char *)malloc(g_n * REF_MSG_SIZE + 1);
  return 0;
}

static void wait_for_trace() {
  char p[4096];
  snprintf(p, sizeof p, "%s.ready", g_reply_path);
  int fd = open(p, O_CREAT | O_WRONLY, 0644);
  if (fd >= 0) close(fd);
  snprintf(p, sizeof p, "%s.go", g_trace_path);
  struct stat st;
  while (stat(p, &st) != 0) usleep(20000);
  if (load_trace()) _Exit(2);
  g_wait = false;
}

extern "C" {
int socket(int, int, int) noexcept { return 1000; }
int setsockopt(int, int, int, const void *, socklen_t) noexcept { return 0; }
int bind(int, const struct sockaddr *, socklen_t) noexcept { return 0; }
int sched_getcpu(void) noexcept { return 3; } /* log ring 0: (3-3)/2, log_server/udp/server.cc:79-80 */

ssize_t recvfrom(int, void *__restrict buf, size_t len, int, struct sockaddr *__restrict addr,
                 socklen_t *__restrict alen) {
  if (g_wait) wait_for_trace();
  if (g_rd == 0) g_t0 = now_s();
  if (g_rd >= g_n) finish_and_exit();
  memcpy(buf, g_trace + g_rd * REF_MSG_SIZE, len < REF_MSG_SIZE ? len : REF_MSG_SIZE);
  g_rd++;
  if (addr && alen) { memset(addr, 0, *alen < sizeof(sockaddr_in) ? *alen : sizeof(sockaddr_in)); }
  return REF_MSG_SIZE;
}
ssize_t __recvfrom_chk(int fd, void *buf, size_t len, size_t, int flags, struct sockaddr *addr,
                       socklen_t *alen) {
  return recvfrom(fd, buf, len, flags, addr, alen);
}
ssize_t sendto(int, const void *buf, size_t len, int, const struct sockaddr *, socklen_t) {
  memcpy(g_replies + g_wr * REF_MSG_SIZE, buf, REF_MSG_SIZE);
  g_wr++;
  return (ssize_t)len;
}
}

int ref_main(int argc, char **argv);

static int harness_main(int argc, char **argv, int ref_argc, char **ref_argv) {
  if (argc < 3) { fprintf(stderr, "usage: %s <trace.bin> <replies.bin> [dump.bin]\n", argv[0]); return 2; }
  g_reply_path = argv[2];
  g_dump_path = argc > 3 ? argv[3] : nullptr;
  g_trace_path = argv[1];
  const char *w = getenv("REF_TRACE_WAIT");
  g_wait = w && w[0] == '1';
  if (!g_wait && load_trace()) return 2;
  return ref_main(ref_argc, ref_argv);
}
