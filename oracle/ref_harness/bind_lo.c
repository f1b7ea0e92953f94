/* bind_lo.c -- TEST / BASELINE INFRASTRUCTURE ONLY.
 *
 * The reference's udp/ servers bind a hard-coded experiment address (lock_fasst/udp/server.cc:45 "10.10.1.1").
 * Linked next to the UNMODIFIED server.cc, this bind() sends every AF_INET bind to 127.0.0.1 (same port), so the
 * as-shipped server -- its own main(), its own thread pinning, real kernel UDP sockets -- runs on a box without
 * that interface.  Nothing else is interposed.
 */
#define _GNU_SOURCE
#include <arpa/inet.h>
#include <netinet/in.h>
#include <stdint.h>
#include <sys/socket.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <stdlib.h>
int bind(int fd, const struct sockaddr *addr, socklen_t len) {
  if (addr && addr->sa_family == AF_INET && len >= sizeof(struct sockaddr_in)) {
    struct sockaddr_in a = *(const struct sockaddr_in *)addr;
    /* BIND_LO_MAP=1: 10.10.1.N -> 127.0.1.N, so that the three shard servers of tatp / smallbank (10.10.1.1 .. 3, one port:
     * tatp/udp/net.h:68-72) stay three addresses on loopback; default: everything to 127.0.0.1 */
    const uint32_t ip = ntohl(a.sin_addr.s_addr);
    if (getenv("BIND_LO_MAP") && (ip >> 8) == 0x0A0A01u) a.sin_addr.s_addr = htonl(0x7F000100u | (ip & 0xFFu));
    else a.sin_addr.s_addr = htonl(INADDR_LOOPBACK);
    return (int)syscall(SYS_bind, fd, &a, (socklen_t)sizeof a);
  }
  return (int)syscall(SYS_bind, fd, addr, len);
}
