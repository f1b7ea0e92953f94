/* Replay harness around the unmodified store/udp/server.cc (see harness_common.h). */
#define main ref_main
#include "server.cc"
#undef main
#define REF_MSG_SIZE sizeof(message)
#include "harness_common.h"
#include "kvs_dump.h"

static void ref_dump_state(FILE *f) { dump_kvs(f, table); }
int main(int argc, char **argv) {
  char a0[] = "server", a1[] = "1";
  char *av[] = {a0, a1, nullptr};
  return harness_main(argc, argv, 2, av);
}
