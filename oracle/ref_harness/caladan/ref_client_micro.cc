/*
 * ref_client_micro.cc -- runs ONE worker of the reference's micro-benchmark load generators, the UNMODIFIED
 * lock_fasst/caladan/client.cc (-DREF_FASST) or lock_2pl/caladan/client.cc (-DREF_2PL) (named by -DCLIENT_SRC, found
 * through -I$(REF)/<wl>/caladan), against a CPU oracle lock server and records the requests it sent and the replies it
 * got.  TEST INFRASTRUCTURE ONLY (the pin of dint_amd/csrc/fasst_client.cc and
 * dint_amd/driver.py::TplClient).  stub/caladan_stub.h stands in for the Caladan runtime.
 *
 * The client reads its transactions from traces/microbenchmarks/lock_24000000_r_0.8/trace_0.csv under the current
 * directory (client.cc GetTraces; the trace_init.sh next to client.cc writes such files from unseeded Python random numbers, so
 * none ships with the reference): the caller writes that file -- the transactions of the restated client it wants to
 * compare -- and starts this program in the directory above `traces/`.
 * So that the abort paths run, one ACQUIRE in `refuse_every` (scattered) is answered REJECT by this harness itself, as if
 * another worker held the lock (the server is not touched).
 *
 * usage: ref_client_<fasst|2pl> <messages> <out prefix> <refuse_every>    -> <prefix>.req / .rep (packed wire structs)
 */
#define main ref_client_main
#include CLIENT_SRC
#undef main

extern "C" {
#include "dint_oracle.h"
}

/* which ACQUIREs the harness refuses: one in `every` on average, but scattered (a hash of the running count) -- with a fixed
 * period a transaction of `every` or more locks could never get all of them and the client would retry it for ever */
static bool refuse_now(uint64_t n, uint64_t every) { return every && (((uint32_t)n * 0x9E3779B1u) >> 20) % every == 0; }

static void *g_srv;
static FILE *g_req, *g_rep;
static uint64_t g_budget, g_sent, g_locks, g_every;

void ref_client_server(uint32_t ip, void *msg, size_t len) {
  (void)ip;
  if (len != sizeof(message)) panic("message of %zu bytes", len);
  if (g_sent >= g_budget) throw ref_client_stop();
  g_sent++;
  fwrite(msg, 1, len, g_req);
  message *m = (message *)msg;
  bool refused = false;
#ifdef REF_FASST
  if (m->type == PktType::kAcquireLock && refuse_now(g_locks++, g_every)) { m->type = PktType::kRejectLock; refused = true; }
  if (!refused) orc_fasst_replay((orc_fasst *)g_srv, msg, 1);
#else
  if (m->action == PktType::kAcquireLock && refuse_now(g_locks++, g_every)) { m->action = PktType::kRejectLock; refused = true; }
  if (!refused) orc_2pl_replay((orc_2pl *)g_srv, msg, 1);
#endif
  fwrite(msg, 1, len, g_rep);
}

int main(int argc, char **argv) {
  if (argc != 4) { fprintf(stderr, "usage: %s <messages> <out prefix> <refuse_every>\n", argv[0]); return 2; }
  g_budget = strtoull(argv[1], nullptr, 10);
  g_every = strtoull(argv[3], nullptr, 10);
  char p[4096];
  snprintf(p, sizeof p, "%s.req", argv[2]);
  g_req = fopen(p, "wb");
  snprintf(p, sizeof p, "%s.rep", argv[2]);
  g_rep = fopen(p, "wb");
  if (!g_req || !g_rep) { perror("output"); return 2; }
  /* what the reference's main() sets up before it starts the runtime (lock_fasst/caladan/client.cc:316-352) */
  machine_id = 1;
  threads = 1;
  mode = "expr";
  trace_tid.resize(1); trace_type.resize(1); trace_lid.resize(1);
#ifdef REF_FASST
  g_srv = orc_fasst_create(kLockHashSize);
  txn_read_l.resize(1); txn_read_r.resize(1); txn_write_l.resize(1); txn_write_r.resize(1);
  txn_read_l[0].resize(kMaxTxnNum); txn_read_r[0].resize(kMaxTxnNum); txn_write_l[0].resize(kMaxTxnNum); txn_write_r[0].resize(kMaxTxnNum);
  ver_table.resize(1);
  ver_table[0].resize(kLockHashSize);
#else
  g_srv = orc_2pl_create(kLockHashSize);
  trace_action.resize(1);
  txn_l.resize(1); txn_r.resize(1);
  txn_l[0].resize(kMaxTxnNum); txn_r[0].resize(kMaxTxnNum);
#endif
  lat_samples.resize(1); pkt_cnt.resize(1); suc_pkt_cnt.resize(1);
  raddr.ip = MAKE_IP_ADDR(10, 10, 1, 1);
  raddr.port = 20230;
  GetTraces(0);
  if (trace_lid[0].empty()) { fprintf(stderr, "no trace under ./traces/microbenchmarks/lock_24000000_r_0.8/trace_0.csv\n"); return 2; }
  try {
    ClientLoop(0);
  } catch (const ref_client_stop &) {
  }
  fclose(g_req); fclose(g_rep);
  printf("{\"messages\": %llu, \"acquires\": %llu, \"trace_lines\": %zu}\n", (unsigned long long)g_sent, (unsigned long long)g_locks,
         trace_lid[0].size());
  return 0;
}
