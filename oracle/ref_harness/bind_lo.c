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
#include <sys/socket.h>
#include <sys/syscall.h>
#include <unistd.h>

int bind(int fd, const struct sockaddr *addr, socklen_t len) {
  if (addr && addr->sa_family == AF_INET && len >= sizeof(struct sockaddr_in)) {
    struct sockaddr_in a = *(const struct sockaddr_in *)addr;
    a.sin_addr.s_addr = htonl(INADDR_LOOPBACK);
    return (int)syscall(SYS_bind, fd, &a, (socklen_t)sizeof a);
  }
  return (int)syscall(SYS_bind, fd, addr, len);
}
