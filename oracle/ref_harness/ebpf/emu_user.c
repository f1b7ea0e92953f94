/* emu_user.c -- compiles ONE unmodified reference *_user.c (named by -DEMU_USER_SRC; its main() renamed away with
 * -Dmain=ref_user_main and never called) and starts what the emulator needs of it: the table set-up and one
 * server_handler thread, whose socket calls emu_main.c interposes.  TEST INFRASTRUCTURE ONLY. */
#include EMU_USER_SRC
#include "emu.h"

void emu_user_start(void) {
#ifdef EMU_STORE
  table = calloc(1, sizeof(struct kvs));  /* store/ebpf/store_user.c:182-183 (inside its main) */
  kvs_init(table, KVS_HASH_SIZE);
#else
  init_tables();
#endif
  int *id = malloc(sizeof(int));
  *id = 0;
  pthread_t t;
  pthread_create(&t, NULL, server_handler, id);
}
