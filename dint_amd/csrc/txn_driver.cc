// txn_driver.cc -- closed-loop transaction drivers on the HOST (include/dint_driver.h): the reference clients'
// transaction state machines (txn_clients.h, shared with the GPU-resident driver k_txn.hip) as an epoch-synchronous
// generator.  Plain host C++.
//
// One EPOCH = every client emits the messages of its current phase; the messages addressed to shard s form batch s,
// ordered by client id then send order; the three shard servers answer; every client consumes its replies.
#include <stdlib.h>
#include <string.h>

#include <new>
#include <vector>

#include "../../include/dint_abi.h"
#include "../../include/dint_driver.h"
#include "txn_clients.h"

struct dint_driver {
  dint_driver_config cfg{};
  dint_driver_stats st{};
  bool awaiting = false;
  virtual ~dint_driver() {}
  virtual int msg_size() const = 0;
  virtual void next(uint32_t counts[DINT_N_SHARDS]) = 0;
  virtual const void *batch(uint32_t s) const = 0;
  virtual void consume(const void *const rep[DINT_N_SHARDS]) = 0;
};

// parameters of the client state machines from the driver configuration (also used by k_txn.hip)
void dint_driver_params(const dint_driver_config &c, TxParams *P, ZipfTable *zipf) {
  memset(P, 0, sizeof *P);
  P->workload = c.workload;
  P->key_dist = c.key_dist;
  P->n_rows = c.n_rows;
  if (c.workload == DINT_WL_TATP) {
    tatp_workgen(P->workgen);
  } else {
    sb_workgen(P->workgen);
    P->n_hot = c.n_rows * 960000ull / 24000000ull;  // kHotAccountNum / kAccountNum, smallbank.h:17-18
    if (P->n_hot < 2) P->n_hot = c.n_rows < 2 ? c.n_rows : 2;
  }
  if (c.key_dist == 1) {
    zipf->init(c.n_rows, c.zipf_theta);
    P->zipf_cdf = zipf->cdf.data();
  }
}

namespace {

template <class T>
struct HostDriver final : dint_driver {
  typedef typename T::Client Client;
  typedef typename T::Msg Msg;
  std::vector<Client> cl;
  std::vector<Msg> store;  // the clients' working messages: T::NMSG per client
  std::vector<Msg> b[DINT_N_SHARDS];
  TxParams P;
  ZipfTable zipf;

  explicit HostDriver(const dint_driver_config &c) {
    cfg = c;
    dint_driver_params(c, &P, &zipf);
    cl.resize(c.n_clients);
    store.resize((size_t)c.n_clients * T::NMSG);
    memset(store.data(), 0, store.size() * sizeof(Msg));
    for (uint32_t i = 0; i < c.n_clients; i++) {
      memset(&cl[i], 0, sizeof(Client));
      cl[i].rng.s = 0xdeadbeefull + c.first_client + i;  // ClientLoop :1122
      cl[i].m.base = (uint8_t *)&store[(size_t)i * T::NMSG];
      cl[i].m.stride = sizeof(Msg);
    }
  }
  int msg_size() const override { return (int)sizeof(Msg); }
  const void *batch(uint32_t s) const override { return b[s].data(); }

  void next(uint32_t counts[DINT_N_SHARDS]) override {
    for (auto &v : b) v.clear();
    typename T::Out o;
    for (auto &c : cl) {
      T::run(c, P, o);
      for (uint8_t k = 0; k < o.n; k++) {
        auto &v = b[o.shard[k]];
        c.out_pos[k] = (uint32_t)v.size();
        v.push_back(o.materialize(c, k));
      }
      st.messages += o.n;
      for (uint8_t k = 0; k < o.n_fin && k < 2; k++) {
        st.txns++; st.by_type[o.fin_txn[k]]++;
        if (o.fin_ok[k]) { st.committed++; st.committed_by_type[o.fin_txn[k]]++; }
      }
    }
    for (uint32_t s = 0; s < DINT_N_SHARDS; s++) counts[s] = (uint32_t)b[s].size();
    st.epochs++;
  }
  void consume(const void *const rep[DINT_N_SHARDS]) override {
    for (auto &c : cl)
      for (uint32_t k = 0; k < c.n_out; k++) tx_consume_one(c, c.out_dst[k], (const Msg *)rep[c.out_shard[k]] + c.out_pos[k]);
  }
};

}  // namespace

extern "C" {

int dint_driver_create(const dint_driver_config *cfg, dint_driver_t **out) {
  if (!cfg || !out || cfg->n_clients == 0 || cfg->n_rows == 0) return DINT_EINVAL;
  if (cfg->key_dist > 1 || (cfg->key_dist == 1 && !(cfg->zipf_theta > 0 && cfg->zipf_theta < 1))) return DINT_EINVAL;
  try {
    if (cfg->workload == DINT_WL_TATP) *out = new HostDriver<TatpTraits>(*cfg);
    else if (cfg->workload == DINT_WL_SMALLBANK) *out = new HostDriver<SbTraits>(*cfg);
    else return DINT_EINVAL;
  } catch (const std::bad_alloc &) {
    return DINT_ENOMEM;
  }
  return 0;
}
void dint_driver_destroy(dint_driver_t *d) { delete d; }
int dint_driver_msg_size(const dint_driver_t *d) { return d ? d->msg_size() : DINT_EINVAL; }
int dint_driver_next(dint_driver_t *d, uint32_t counts[DINT_N_SHARDS]) {
  if (!d || !counts) return DINT_EINVAL;
  if (d->awaiting) return DINT_ESTATE;
  d->next(counts);
  d->awaiting = true;
  return 0;
}
const void *dint_driver_batch(dint_driver_t *d, uint32_t shard) { return (d && shard < DINT_N_SHARDS) ? d->batch(shard) : nullptr; }
int dint_driver_consume(dint_driver_t *d, const void *const replies[DINT_N_SHARDS]) {
  if (!d || !replies) return DINT_EINVAL;
  if (!d->awaiting) return DINT_ESTATE;
  d->consume(replies);
  d->awaiting = false;
  return 0;
}
int dint_driver_get_stats(const dint_driver_t *d, dint_driver_stats *out) {
  if (!d || !out) return DINT_EINVAL;
  *out = d->st;
  return 0;
}
}
