/* Replay harness around the unmodified tatp/udp/server_shard.cc (see harness_common.h).
 * Runs as shard 2 so the shard-1-only cpu-monitor threads (server_shard.cc:309-312) are not started. */
#define main ref_main
#include "server_shard.cc"
#undef main
#define REF_MSG_SIZE sizeof(message)
#include "harness_common.h"
#include "kvs_dump.h"

/* dump: 5 tables (kvs_dump format), then per table u32 n + {u32 slot} of held txn locks,
 * then u32 tail, u32 n + n canonical 64-byte log records of ring 0 */
static void ref_dump_state(FILE *f) {
  for (int t = 0; t < kTableNum; t++) dump_kvs(f, tables[t]);
  for (int t = 0; t < kTableNum; t++) {
    uint32_t cnt = 0, lim = (uint32_t)tables[t]->hash_size * kKeysPerEntry;
    for (uint32_t i = 0; i < lim; i++) if (txn_locks[t][i]) cnt++;
    fwrite(&cnt, 4, 1, f);
    for (uint32_t i = 0; i < lim; i++) if (txn_locks[t][i]) fwrite(&i, 4, 1, f);
  }
  uint64_t total = 0; /* log ops = replies of type kCommitLogAck / kDeleteLogAck */
  for (size_t i = 0; i < g_wr; i++) {
    unsigned char ty = g_replies[i * REF_MSG_SIZE + 1];
    if (ty == kCommitLogAck || ty == kDeleteLogAck) total++;
  }
  uint32_t n = total < (uint64_t)kMaxLogEntryNum ? (uint32_t)total : (uint32_t)kMaxLogEntryNum;
  uint32_t tail = (uint32_t)(total % kMaxLogEntryNum);
  fwrite(&tail, 4, 1, f); fwrite(&n, 4, 1, f);
  for (uint32_t i = 0; i < n; i++) {
    unsigned char rec[64] = {0};
    memcpy(rec, &txn_log[0][i].key, 8);
    memcpy(rec + 8, txn_log[0][i].val, kValSize);
    memcpy(rec + 48, &txn_log[0][i].ver, 4);
    rec[52] = txn_log[0][i].is_del; rec[53] = txn_log[0][i].table;
    fwrite(rec, 64, 1, f);
  }
}
int main(int argc, char **argv) {
  char a0[] = "server_shard", a1[] = "2", a2[] = "1";
  char *av[] = {a0, a1, a2, nullptr};
  return harness_main(argc, argv, 3, av);
}
