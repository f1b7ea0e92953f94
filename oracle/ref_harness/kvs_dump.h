/* kvs_dump.h -- dump a reference `kvs` (store|tatp|smallbank /udp/kvs.h) in bucket/chain order.
 * TEST INFRASTRUCTURE ONLY.  Format: u64 nrows, then nrows x {u64 key; u32 ver; u8 val[kValSize]}. */
#pragma once
static void dump_kvs(FILE *f, kvs *t) {
  uint64_t n = 0;
  for (int b = 0; b < t->hash_size; b++)
    for (kvs_entry *e = t->bucket_heads[b]; e; e = e->next)
      for (int i = 0; i < kKeysPerEntry; i++) if (e->valid[i]) n++;
  fwrite(&n, 8, 1, f);
  for (int b = 0; b < t->hash_size; b++)
    for (kvs_entry *e = t->bucket_heads[b]; e; e = e->next)
      for (int i = 0; i < kKeysPerEntry; i++)
        if (e->valid[i]) {
          fwrite(&e->key[i], 8, 1, f);
          fwrite(&e->ver[i], 4, 1, f);
          fwrite(e->val[i], kValSize, 1, f);
        }
}
