/*
 * ref_client.cc -- runs ONE client of the reference (the UNMODIFIED <wl>/caladan/client_udp_shard.cc, named by
 * -DCLIENT_SRC and found through -I$(REF)/<wl>/caladan) against three CPU oracle shard servers and records, per shard,
 * the requests it sent and the replies it got.  TEST INFRASTRUCTURE ONLY (the pin of the transaction
 * handlers).  See stub/caladan_stub.h for how the Caladan runtime is stood in for.
 *
 * So that the abort paths run too, every 7th lock request is answered REJECT by this harness itself, as if another
 * client held the lock (the server is not touched; the reference client then releases what it holds and gives up the
 * transaction -- tatp/caladan/client_udp_shard.cc:400-420).
 *
 * usage: ref_client_<wl> <worker gid> <messages> <out prefix>     -> <prefix>.s{0,1,2}.req / .rep (packed wire structs)
 */
#define main ref_client_main
#include CLIENT_SRC
#undef main

extern "C" {
#include "dint_oracle.h"
}

static void *g_shard[3];
static FILE *g_req[3], *g_rep[3];
static uint64_t g_budget, g_sent, g_locks;

void ref_client_server(uint32_t ip, void *msg, size_t len) {
  int s = -1;
  for (int i = 0; i < 3; i++)
    if (servaddr[i].ip == ip) s = i;
  if (s < 0 || len != sizeof(message)) panic("message of %zu bytes to an unknown server", len);
  if (g_sent >= g_budget) throw ref_client_stop();
  g_sent++;
  fwrite(msg, 1, len, g_req[s]);
  message *m = (message *)msg;
  bool refused = false;
#ifdef REF_SMALLBANK
  if (m->type == PktType::kAcquireShared || m->type == PktType::kAcquireExclusive) {
    if (g_locks++ % 7 == 3) { m->type = m->type == PktType::kAcquireShared ? PktType::kRejectShared : PktType::kRejectExclusive; refused = true; }
  }
  if (!refused) orc_sb_replay((orc_sb *)g_shard[s], msg, 1);
#else
  if (m->type == PktType::kAcquireLock) {
    if (g_locks++ % 7 == 3) { m->type = PktType::kRejectLock; refused = true; }
  }
  if (!refused) orc_tatp_replay((orc_tatp *)g_shard[s], msg, 1);
#endif
  fwrite(msg, 1, len, g_rep[s]);
}

int main(int argc, char **argv) {
  if (argc != 4) { fprintf(stderr, "usage: %s <worker gid> <messages> <out prefix>\n", argv[0]); return 2; }
  const int gid = atoi(argv[1]);
  g_budget = strtoull(argv[2], nullptr, 10);
  for (int s = 0; s < 3; s++) {
    char p[4096];
    snprintf(p, sizeof p, "%s.s%d.req", argv[3], s);
    g_req[s] = fopen(p, "wb");
    snprintf(p, sizeof p, "%s.s%d.rep", argv[3], s);
    g_rep[s] = fopen(p, "wb");
    if (!g_req[s] || !g_rep[s]) { perror("output"); return 2; }
#ifdef REF_SMALLBANK
    g_shard[s] = orc_sb_create(kAccountNum, 1000000, kAccountNum);
#else
    g_shard[s] = orc_tatp_create(kSubscriberNum, 1000000, kSubscriberNum);
#endif
    servaddr[s].ip = MAKE_IP_ADDR(10, 10, 1, 1 + s);
    servaddr[s].port = kFasstPort;
  }
  /* what the reference's main() sets up before it starts the runtime (client_udp_shard.cc:1202-1236) */
  machine_id = 0;
  threads = 1 << 30;  /* wrkr_lid = wrkr_gid % threads = gid; the statistics arrays below are never indexed (stat_started stays false) */
  mode = "expr";
  net_intv = 0;
  CreateWorkgenArr();
#ifndef REF_SMALLBANK
  create_map1000();
#endif
  try {
    ClientLoop(gid);
  } catch (const ref_client_stop &) {
  }
  for (int s = 0; s < 3; s++) { fclose(g_req[s]); fclose(g_rep[s]); }
  printf("{\"gid\": %d, \"messages\": %llu, \"locks_refused\": %llu}\n", gid, (unsigned long long)g_sent,
         (unsigned long long)((g_locks + 3) / 7));
  return 0;
}
