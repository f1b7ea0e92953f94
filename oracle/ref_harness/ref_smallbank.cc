/* Replay harness around the unmodified smallbank/udp/server_shard.cc (see harness_common.h).
 * Runs as shard 2 so the shard-1-only cpu-monitor threads are not started. */
#define main ref_main
#include "server_shard.cc"
#undef main
#define REF_MSG_SIZE sizeof(message)
#include "harness_common.h"
#include "kvs_dump.h"

/* dump: 2 tables (kvs_dump format), then per table u32 n + {u32 slot, u32 num_ex, u32 num_sh} non-zero */
static void ref_dump_state(FILE *f) {
  for (int t = 0; t < kTableNum; t++) dump_kvs(f, tables[t]);
  for (int t = 0; t < kTableNum; t++) {
    uint32_t cnt = 0, lim = (uint32_t)tables[t]->hash_size * kKeysPerEntry;
    for (uint32_t i = 0; i < lim; i++) if (num_ex[t][i] || num_sh[t][i]) cnt++;
    fwrite(&cnt, 4, 1, f);
    for (uint32_t i = 0; i < lim; i++)
      if (num_ex[t][i] || num_sh[t][i]) {
        uint32_t r[3] = {i, num_ex[t][i], num_sh[t][i]};
        fwrite(r, 4, 3, f);
      }
  }
}
int main(int argc, char **argv) {
  char a0[] = "server_shard", a1[] = "2", a2[] = "1";
  char *av[] = {a0, a1, a2, nullptr};
  return harness_main(argc, argv, 3, av);
}
