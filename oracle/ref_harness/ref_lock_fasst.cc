/* Replay harness around the unmodified lock_fasst/udp/server.cc (see harness_common.h). */
#define main ref_main
#include "server.cc"
#undef main
#define REF_MSG_SIZE sizeof(message)
#include "harness_common.h"

/* dump: u32 count, then {u32 slot, u32 lock, u32 ver} for every non-zero slot */
static void ref_dump_state(FILE *f) {
  uint32_t cnt = 0;
  for (int i = 0; i < kLockHashSize; i++) if (locks[i] || ver_table[i]) cnt++;
  fwrite(&cnt, 4, 1, f);
  for (int i = 0; i < kLockHashSize; i++)
    if (locks[i] || ver_table[i]) {
      uint32_t r[3] = {(uint32_t)i, (uint32_t)locks[i], ver_table[i]};
      fwrite(r, 4, 3, f);
    }
}
int main(int argc, char **argv) {
  char a0[] = "server", a1[] = "1";
  char *av[] = {a0, a1, nullptr};
  return harness_main(argc, argv, 2, av);
}
