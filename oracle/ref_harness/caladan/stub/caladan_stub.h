/*
 * caladan_stub.h -- a synchronous, single-threaded stand-in for the slice of the Caladan runtime the reference's
 * clients use (caladan/bindings/cc/{net,thread,sync,timer}.h, caladan/inc/base/log.h ...).  TEST INFRASTRUCTURE ONLY.
 *
 * The Caladan submodules (DPDK, rdma-core, SPDK) are empty in the reference tree, so its clients cannot be linked; but
 * the transaction logic in <wl>/caladan/client_udp_shard.cc is plain C++ over six names.  With this header on the
 * include path the UNMODIFIED client translation unit compiles with g++:
 *   rt::UdpConn::WriteTo   hands the message to ref_client.cc's server hook (a CPU oracle shard) and queues the reply,
 *   rt::UdpConn::ReadFrom  pops it,
 *   rt::Thread             runs its function at once, to completion (the per-shard send-all-then-receive-all workers of a
 *                          transaction phase are independent of each other: three servers),
 * so one ClientLoop runs its transactions one after the other against three serial servers -- the request stream of one
 * client, which is what dint_amd/csrc/txn_clients.h restates.
 */
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/types.h>
#include <time.h>

#include <deque>
#include <functional>
#include <vector>

#ifndef unlikely
#define unlikely(x) __builtin_expect(!!(x), 0)
#define likely(x) __builtin_expect(!!(x), 1)
#endif
#ifndef _unused
#define _unused(x) ((void)(x))
#endif
#include "base/log.h"
#include "net/ip.h"

struct netaddr {
  uint32_t ip;
  uint16_t port;
};

static inline uint64_t microtime(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (uint64_t)ts.tv_sec * 1000000ull + ts.tv_nsec / 1000;
}
struct ref_client_stop {};  /* thrown by the server hook when the message budget is used up: ends ClientLoop */
static inline void init_shutdown(int) { throw ref_client_stop(); }
typedef void (*thread_fn_t)(void *);
static inline int runtime_init(const char *, thread_fn_t fn, void *arg) { fn(arg); return 0; }

/* ref_client.cc: process one message at the server `ip` names; the reply overwrites the message */
void ref_client_server(uint32_t ip, void *msg, size_t len);

namespace rt {
class UdpConn {
 public:
  static constexpr size_t kMaxPayloadSize = 1472;
  static UdpConn *Listen(netaddr) { return new UdpConn(); }
  netaddr LocalAddr() const { return {0, 0}; }
  ssize_t WriteTo(const void *buf, size_t len, const netaddr *raddr) {
    std::vector<uint8_t> m((const uint8_t *)buf, (const uint8_t *)buf + len);
    ref_client_server(raddr->ip, m.data(), len);
    q_.push_back(std::move(m));
    return (ssize_t)len;
  }
  ssize_t ReadFrom(void *buf, size_t len, netaddr *raddr) {
    if (q_.empty()) panic("ReadFrom with nothing outstanding");
    const size_t n = q_.front().size() < len ? q_.front().size() : len;
    memcpy(buf, q_.front().data(), n);
    q_.pop_front();
    if (raddr) *raddr = {0, 0};
    return (ssize_t)n;
  }

 private:
  std::deque<std::vector<uint8_t>> q_;
};
class Thread {
 public:
  Thread() {}
  explicit Thread(std::function<void()> f) { f(); }
  Thread(Thread &&) = default;
  Thread &operator=(Thread &&) = default;
  void Join() {}
  void Detach() {}
};
static inline void Spawn(std::function<void()> f) { f(); }
static inline void Sleep(uint64_t) {}
static inline void SleepUntil(uint64_t) {}
}  // namespace rt
