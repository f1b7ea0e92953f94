// fasst_client.cc -- the lock_fasst load generator (the CALLER of the lock_fasst path), restated as an
// epoch-synchronous closed loop.  Plain host C++.
//
// Reference: lock_fasst/caladan/client.cc:183-280 (ClientLoop) over traces made by lock_fasst/caladan/trace_init.sh
// :6-27 -- per transaction 5..10 distinct keys, sorted; every key is read; each key is also written with probability
// 1 - r_prop (r_prop = 0.8).  One uthread = one worker with ONE request outstanding:
//   READ every read key (remember its version)               client.cc:237-245
//   ACQUIRE_LOCK every write key; on REJECT_LOCK send ABORT for the locks taken so far and restart the
//   transaction from its first read                          :248-270
//   when the transaction's last request is done: re-READ every read key; a changed version -> ABORT every write
//   key and restart the transaction; else COMMIT every write key   :196-235
// W workers run in lock step: one EPOCH = every worker's next request, in worker order (models W concurrent
// clients, so REJECTs, roll-backs and aborts really happen); the epochs laid end to end are the request trace.
// The reference's traces are unseeded Python `random.sample` output, so no fixed trace ships with it:
// here every worker draws its transactions from the reference's own LCG (`fastrand`, tatp/caladan/tatp.h:31-34) seeded
// 0xdeadbeef + worker, keys uniform (the reference) or Zipf(theta) over the key space.  A worker never wraps around
// to its first transaction (the reference replays its 20,000-transaction file in a loop; 24M requests over 4096
// workers are ~320 transactions each).
#include <string.h>

#include <algorithm>
#include <new>
#include <vector>

#include "../../include/dint_abi.h"
#include "../../include/dint_driver.h"
#include "zipf_table.h"

namespace {

#pragma pack(push, 1)
struct FasstMsg {  // lock_fasst/caladan/proto.h:31-36 (= lock_fasst/udp/net.h:23-29)
  uint8_t type;
  uint32_t lid;
  uint32_t ver;
};
#pragma pack(pop)
static_assert(sizeof(FasstMsg) == 9, "packed wire struct");

enum : uint8_t { F_READ = 0, F_ACQ = 1, F_ABORT = 2, F_COMMIT = 3, F_GRANT_READ = 4, F_GRANT_LOCK = 5, F_REJECT_LOCK = 6 };
enum : uint8_t { P_READ, P_ACQ, P_REJ_ABORT, P_VALIDATE, P_RB_ABORT, P_COMMIT };

struct Worker {
  uint64_t rng;
  uint32_t keys[10], vers[10], wkeys[10];
  uint8_t nk, nw, phase, pos, abort_n;
};

}  // namespace

struct dint_fasst_client {
  dint_fasst_client_config cfg{};
  dint_fasst_client_stats st{};
  std::vector<Worker> w;
  std::vector<FasstMsg> out;
  ZipfTable zipf;
  bool awaiting = false;

  uint32_t rnd(Worker &x) {  // fastrand
    x.rng = x.rng * 1103515245ull + 12345ull;
    return (uint32_t)(x.rng >> 32);
  }
  uint32_t pick(Worker &x) {
    if (cfg.key_dist == 1) return (uint32_t)zipf_lookup(zipf.cdf.data(), zipf.n, rnd(x));
    return (uint32_t)(((uint64_t)rnd(x) * cfg.key_space) >> 32);  // uniform over [0, key_space)
  }
  void new_txn(Worker &x) {  // trace_init.sh:12-27
    x.nk = (uint8_t)(5 + rnd(x) % 6);
    for (uint8_t i = 0; i < x.nk;) {  // distinct keys (random.sample)
      const uint32_t k = pick(x);
      bool dup = false;
      for (uint8_t j = 0; j < i; j++) dup |= x.keys[j] == k;
      if (!dup) x.keys[i++] = k;
    }
    std::sort(x.keys, x.keys + x.nk);
    x.nw = 0;
    for (uint8_t i = 0; i < x.nk; i++)
      if (rnd(x) % 100 >= cfg.read_pct) x.wkeys[x.nw++] = x.keys[i];
    x.phase = P_READ;
    x.pos = 0;
  }
  void restart(Worker &x) { x.phase = P_READ; x.pos = 0; }
};

extern "C" {

int dint_fasst_client_create(const dint_fasst_client_config *cfg, dint_fasst_client_t **out) {
  if (!cfg || !out || cfg->n_workers == 0 || cfg->key_space < 16 || cfg->read_pct > 100 || cfg->key_dist > 1) return DINT_EINVAL;
  if (cfg->key_dist == 1 && !(cfg->zipf_theta > 0 && cfg->zipf_theta < 1)) return DINT_EINVAL;
  try {
    dint_fasst_client *c = new dint_fasst_client();
    c->cfg = *cfg;
    c->w.resize(cfg->n_workers);
    c->out.resize(cfg->n_workers);
    if (cfg->key_dist == 1) c->zipf.init(cfg->key_space, cfg->zipf_theta);
    for (uint32_t i = 0; i < cfg->n_workers; i++) {
      c->w[i].rng = 0xdeadbeefull + cfg->first_worker + i;
      c->new_txn(c->w[i]);
    }
    *out = c;
  } catch (const std::bad_alloc &) {
    return DINT_ENOMEM;
  }
  return 0;
}

void dint_fasst_client_destroy(dint_fasst_client_t *c) { delete c; }

// one request per worker, in worker order; returns the batch (n_workers 9-byte messages, valid until the next call)
const void *dint_fasst_client_next(dint_fasst_client_t *c) {
  if (!c || c->awaiting) return nullptr;
  for (size_t i = 0; i < c->w.size(); i++) {
    Worker &x = c->w[i];
    FasstMsg m = {0, 0, 0};
    switch (x.phase) {
      case P_READ: case P_VALIDATE: m.type = F_READ; m.lid = x.keys[x.pos]; break;
      case P_ACQ: m.type = F_ACQ; m.lid = x.wkeys[x.pos]; break;
      case P_REJ_ABORT: case P_RB_ABORT: m.type = F_ABORT; m.lid = x.wkeys[x.pos]; break;
      default: m.type = F_COMMIT; m.lid = x.wkeys[x.pos]; break;
    }
    c->out[i] = m;
  }
  c->st.requests += c->w.size();
  c->st.epochs++;
  c->awaiting = true;
  return c->out.data();
}

int dint_fasst_client_consume(dint_fasst_client_t *c, const void *replies) {
  if (!c || !replies) return DINT_EINVAL;
  if (!c->awaiting) return DINT_ESTATE;
  const FasstMsg *rep = (const FasstMsg *)replies;
  for (size_t i = 0; i < c->w.size(); i++) {
    Worker &x = c->w[i];
    const FasstMsg r = rep[i];
    if (r.lid != c->out[i].lid) c->st.protocol_errors++;  // the asserts of client.cc:205-206,241-242
    switch (x.phase) {
      case P_READ:  // :237-245
        if (r.type != F_GRANT_READ) c->st.protocol_errors++;
        x.vers[x.pos] = r.ver;
        if (++x.pos == x.nk) { x.pos = 0; x.phase = x.nw ? P_ACQ : P_VALIDATE; }
        break;
      case P_ACQ:  // :248-270
        if (r.type == F_GRANT_LOCK) {
          if (++x.pos == x.nw) { x.pos = 0; x.phase = P_VALIDATE; }
        } else if (r.type == F_REJECT_LOCK) {
          c->st.rejects++;
          if (x.pos) { x.abort_n = x.pos; x.pos = 0; x.phase = P_REJ_ABORT; }
          else c->restart(x);
        } else {
          c->st.protocol_errors++;  // "received wrong packet"
        }
        break;
      case P_REJ_ABORT:
        if (++x.pos == x.abort_n) c->restart(x);
        break;
      case P_VALIDATE:  // :196-213
        if (r.ver != x.vers[x.pos]) {
          c->st.rollbacks++;
          if (x.nw) { x.pos = 0; x.phase = P_RB_ABORT; }
          else c->restart(x);
        } else if (++x.pos == x.nk) {
          if (x.nw) { x.pos = 0; x.phase = P_COMMIT; }
          else { c->st.committed++; c->new_txn(x); }
        }
        break;
      case P_RB_ABORT:  // :215-222
        if (++x.pos == x.nw) c->restart(x);
        break;
      default:  // P_COMMIT :224-229
        if (++x.pos == x.nw) { c->st.committed++; c->new_txn(x); }
        break;
    }
  }
  c->awaiting = false;
  return 0;
}

// the transaction worker `worker` is running: its sorted read set (all keys) and its write set, as one transaction of the
// reference's trace files lists them (trace_init.sh:20-27).  For the pin against the unmodified reference client, which
// reads its transactions from such a file (tests/golden/make_golden_clients_micro.py).
int dint_fasst_client_peek(const dint_fasst_client_t *c, uint32_t worker, uint32_t *keys, uint32_t *n_keys, uint32_t *wkeys,
                           uint32_t *n_wkeys) {
  if (!c || !keys || !n_keys || !wkeys || !n_wkeys || worker >= c->w.size()) return DINT_EINVAL;
  const Worker &x = c->w[worker];
  *n_keys = x.nk;
  *n_wkeys = x.nw;
  memcpy(keys, x.keys, sizeof(uint32_t) * x.nk);
  memcpy(wkeys, x.wkeys, sizeof(uint32_t) * x.nw);
  return 0;
}

int dint_fasst_client_get_stats(const dint_fasst_client_t *c, dint_fasst_client_stats *out) {
  if (!c || !out) return DINT_EINVAL;
  *out = c->st;
  return 0;
}

}  // extern "C"
