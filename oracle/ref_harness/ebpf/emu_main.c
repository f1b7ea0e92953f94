/*
 * emu_main.c -- replay harness around the UNMODIFIED reference eBPF servers (DINT proper).  TEST INFRASTRUCTURE ONLY.
 *
 * A DINT server is three programs: the XDP ingress program and the TC egress program of <wl>/ebpf/*_kern.c, and the
 * user-space fallback of <wl>/ebpf/*_user.c that owns the full kvs behind the in-kernel cache.  None of them can be
 * loaded here (no BPF target, no XDP-capable NIC), but all three are plain C: emu_kern.c compiles the kernel side with
 * the host gcc against a stub of bpf_helpers.h (maps = arrays), emu_user.c the user side against a stub of libbpf.
 * This file plays the network between them, one request of a trace at a time -- exactly the path a packet takes
 * (SURVEY.md 3.2):
 *     request -> [eth|ip|udp|message] -> XDP program -> XDP_TX: the reply is the mutated packet
 *                                                     -> XDP_PASS: the (tail-extended) payload goes to the user
 *                                                        thread's recvfrom(); what it sendto()s leaves through the TC
 *                                                        program, which installs the cache line and shrinks the packet
 * Serial, one thread each: the serial semantics of the eBPF flavour, the pin for the codes that only it has
 * (REJECT_LOCK_SAME_KEY, WARMUP_READ, store INSERT).
 *
 * usage: [EMU_HOLD=<1|2|3> [EMU_HOLD_EVERY=<k>]] ref_ebpf_<wl> <trace.bin> <replies.bin> [maps_dump.bin]
 *        (records = packed `struct message`; EMU_HOLD: see emu.h -- the back-pressure replies of a contended entry)
 * stdout: one JSON line {"n":..., "tx":..., "pass":..., "seconds":...}
 */
#define _GNU_SOURCE
#include <arpa/inet.h>
#include <linux/bpf.h>
#include <linux/if_ether.h>
#include <linux/ip.h>
#include <linux/udp.h>
#include <pthread.h>
#include <semaphore.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/socket.h>
#include <time.h>

#include "emu.h"

#define HDR (sizeof(struct ethhdr) + sizeof(struct iphdr) + sizeof(struct udphdr))
#define PORT 20230

/* ---- the user thread's sockets: a one-slot mailbox each way ------------------------------------------------- */
static sem_t g_to_user, g_from_user;
static unsigned char g_mail[512];
static size_t g_mail_len;
static int g_user_on;
static volatile int g_pending;  /* the user thread took a request and has not answered it yet */

int socket(int d, int t, int p) { (void)d; (void)t; (void)p; return 1000; }
int setsockopt(int fd, int l, int o, const void *v, socklen_t n) { (void)fd; (void)l; (void)o; (void)v; (void)n; return 0; }
int bind(int fd, const struct sockaddr *a, socklen_t n) { (void)fd; (void)a; (void)n; return 0; }
ssize_t recvfrom(int fd, void *buf, size_t len, int fl, struct sockaddr *a, socklen_t *al) {
  (void)fd; (void)fl;
  if (g_pending) {  /* the handler looped without a sendto(): it had no answer (its panic() path) */
    g_pending = 0;
    g_mail_len = 0;
    sem_post(&g_from_user);
  }
  sem_wait(&g_to_user);
  g_pending = 1;
  size_t n = g_mail_len < len ? g_mail_len : len;
  memcpy(buf, g_mail, n);
  if (a && al) memset(a, 0, *al);
  return (ssize_t)g_mail_len;
}
ssize_t __recvfrom_chk(int fd, void *buf, size_t len, size_t bl, int fl, struct sockaddr *a, socklen_t *al) {
  (void)bl;
  return recvfrom(fd, buf, len, fl, a, al);
}
ssize_t sendto(int fd, const void *buf, size_t len, int fl, const struct sockaddr *a, socklen_t al) {
  (void)fd; (void)fl; (void)a; (void)al;
  memcpy(g_mail, buf, len < sizeof g_mail ? len : sizeof g_mail);
  g_mail_len = len;
  g_pending = 0;
  sem_post(&g_from_user);
  return (ssize_t)len;
}

static void fill_headers(unsigned char *pkt, size_t payload, int from_server) {
  struct ethhdr *eth = (struct ethhdr *)pkt;
  struct iphdr *ip = (struct iphdr *)(pkt + sizeof *eth);
  struct udphdr *udp = (struct udphdr *)(pkt + sizeof *eth + sizeof *ip);
  memset(pkt, 0, HDR);
  memset(eth->h_dest, from_server ? 0x22 : 0x11, ETH_ALEN);
  memset(eth->h_source, from_server ? 0x11 : 0x22, ETH_ALEN);
  eth->h_proto = htons(ETH_P_IP);
  ip->version = 4; ip->ihl = 5; ip->ttl = 64; ip->protocol = IPPROTO_UDP;
  ip->tot_len = htons((uint16_t)(sizeof *ip + sizeof *udp + payload));
  ip->saddr = htonl(from_server ? 0x0A0A0101 : 0x0A0A0102);
  ip->daddr = htonl(from_server ? 0x0A0A0102 : 0x0A0A0101);
  udp->source = htons(from_server ? PORT : 40000);
  udp->dest = htons(from_server ? 40000 : PORT);
  udp->len = htons((uint16_t)(sizeof *udp + payload));
}

int main(int argc, char **argv) {
  if (argc < 3) { fprintf(stderr, "usage: %s <trace.bin> <replies.bin> [maps_dump.bin]\n", argv[0]); return 2; }
  const size_t msg = emu_msg_size(), ext = emu_ext_size();
  FILE *f = fopen(argv[1], "rb");
  if (!f) { perror("trace"); return 2; }
  fseek(f, 0, SEEK_END);
  const size_t n = (size_t)ftell(f) / msg;
  fseek(f, 0, SEEK_SET);
  unsigned char *trace = malloc(n * msg + 1), *replies = malloc(n * msg + 1);
  if (fread(trace, msg, n, f) != n) { perror("read"); return 2; }
  fclose(f);
  /* packet buffers below 4 GiB: xdp_md / __sk_buff carry 32-bit data pointers */
  unsigned char *pkt = mmap(NULL, 8192, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_32BIT, -1, 0);
  if (pkt == MAP_FAILED) { perror("mmap MAP_32BIT"); return 2; }
  sem_init(&g_to_user, 0, 0);
  sem_init(&g_from_user, 0, 0);
#ifdef EMU_WITH_USER
  emu_user_start();
  g_user_on = 1;
#endif
  size_t n_tx = 0, n_pass = 0, n_other = 0;
  const int hold_bits = getenv("EMU_HOLD") ? atoi(getenv("EMU_HOLD")) : 0;
  const size_t hold_every = getenv("EMU_HOLD_EVERY") ? (size_t)atol(getenv("EMU_HOLD_EVERY")) : (hold_bits ? 1 : 0);
  struct timespec t0, t1;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  for (size_t i = 0; i < n; i++) {
    fill_headers(pkt, msg, 0);
    memcpy(pkt + HDR, trace + i * msg, msg);
    uint32_t len = (uint32_t)(HDR + msg);
    g_emu_hold = (hold_every && i % hold_every == hold_every - 1) ? hold_bits : 0;
    const int rc = emu_xdp(pkt, &len);
    emu_release();  /* the "other packet" is done before the user side / the TC program see this one */
    g_emu_hold = 0;
    if (rc == XDP_TX) {
      n_tx++;
      memcpy(replies + i * msg, pkt + HDR, msg);
    } else if (rc == XDP_PASS && g_user_on) {
      /* the kernel delivers the packet -- tail-extended to an ext_message on a cache miss, as it came for the
         requests the XDP program leaves to user space altogether (tatp DELETE_*) -- to the user thread's socket */
      n_pass++;
      const size_t pl = len - HDR;
      memcpy(g_mail, pkt + HDR, pl);
      g_mail_len = pl;
      sem_post(&g_to_user);
      sem_wait(&g_from_user);
      if (g_mail_len == 0) {  /* nothing came back: no reply on the wire */
        n_pass--;
        n_other++;
        memcpy(replies + i * msg, trace + i * msg, msg);
        continue;
      }
      fill_headers(pkt, g_mail_len, 1);
      memcpy(pkt + HDR, g_mail, g_mail_len);
      len = (uint32_t)(HDR + g_mail_len);
      emu_tc(pkt, &len);
      memcpy(replies + i * msg, pkt + HDR, msg);
    } else {  /* the micro servers have no user side: a packet their XDP program does not answer gets no reply;
                 recorded as the request itself */
      n_other++;
      memcpy(replies + i * msg, trace + i * msg, msg);
    }
  }
  clock_gettime(CLOCK_MONOTONIC, &t1);
  f = fopen(argv[2], "wb");
  if (!f || fwrite(replies, msg, n, f) != n) { perror("replies"); return 2; }
  fclose(f);
  if (argc > 3 && emu_dump_maps(argv[3])) { perror("dump"); return 2; }
  printf("{\"n\": %zu, \"tx\": %zu, \"pass\": %zu, \"unanswered\": %zu, \"seconds\": %.6f}\n", n, n_tx, n_pass, n_other,
         (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec));
  fflush(stdout);
  _Exit(0);
}
