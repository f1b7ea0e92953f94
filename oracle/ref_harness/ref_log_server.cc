/* Replay harness around the unmodified log_server/udp/server.cc (see harness_common.h). */
#define main ref_main
#include "server.cc"
#undef main
#define REF_MSG_SIZE sizeof(message)
#include "harness_common.h"

/* dump: u32 tail, u32 n_entries, then n canonical 64-byte records
 * {u64 key; u8 val[40]; u32 ver; u8 is_del; u8 table; pad} of ring 0 up to tail
 * (or the whole ring if it wrapped: n = min(total, kMaxLogEntryNum)). */
static void ref_dump_state(FILE *f) {
  uint32_t total = (uint32_t)g_wr;
  uint32_t n = total < (uint32_t)kMaxLogEntryNum ? total : (uint32_t)kMaxLogEntryNum;
  uint32_t tail = total % kMaxLogEntryNum;
  fwrite(&tail, 4, 1, f); fwrite(&n, 4, 1, f);
  for (uint32_t i = 0; i < n; i++) {
    unsigned char rec[64] = {0};
    memcpy(rec, &txn_log[0][i].key, 8);
    memcpy(rec + 8, txn_log[0][i].val, kValSize);
    memcpy(rec + 48, &txn_log[0][i].ver, 4);
    fwrite(rec, 64, 1, f);
  }
}
int main(int argc, char **argv) {
  char a0[] = "server", a1[] = "1";
  char *av[] = {a0, a1, nullptr};
  return harness_main(argc, argv, 2, av);
}
