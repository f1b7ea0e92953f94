/* emu_kern.c -- compiles ONE unmodified reference *_kern.c (named by -DEMU_KERN_SRC, found through -I$(REF)/<wl>/ebpf)
 * against stub/linux/tools/lib/bpf/bpf_helpers.h and exposes its two programs.  TEST INFRASTRUCTURE ONLY. */
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <sys/mman.h>
#include EMU_KERN_SRC
#include "emu.h"
#ifndef EMU_XDP_FN
#define EMU_XDP_FN tps_prim_xdp_main
#endif

/* maps: found by the address of their definition, storage allocated zeroed on first use (BPF arrays start zeroed) */
#define EMU_MAX_MAPS 64
static struct { const void *def; unsigned char *mem; size_t vsz, n; } g_maps[EMU_MAX_MAPS];
static int g_nmaps;

int g_emu_hold;  /* bit 0: the cache entries' lock words, bit 1: the lock units' */
static uint64_t *g_held[16];
static int g_nheld;
static void emu_hold_word(uint64_t *w) {
  if (*w == 0 && g_nheld < 16) { *w = 1; g_held[g_nheld++] = w; }
}
void emu_release(void) {
  for (int i = 0; i < g_nheld; i++) *g_held[i] = 0;
  g_nheld = 0;
}

void *emu_map_lookup(const void *map, size_t value_size, size_t max_entries, const void *key) {
  int i;
  for (i = 0; i < g_nmaps; i++)
    if (g_maps[i].def == map) break;
  if (i == g_nmaps) {
    if (g_nmaps == EMU_MAX_MAPS) { fprintf(stderr, "emu: too many maps\n"); exit(2); }
    size_t bytes = value_size * max_entries;
    void *m = mmap(NULL, bytes ? bytes : 1, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (m == MAP_FAILED) { perror("emu: map storage"); exit(2); }
    g_maps[i].def = map; g_maps[i].mem = m; g_maps[i].vsz = value_size; g_maps[i].n = max_entries;
    g_nmaps++;
  }
  uint32_t k = *(const uint32_t *)key;  /* BPF_MAP_TYPE_ARRAY / PERCPU_ARRAY (one CPU): u32 index */
  if (k >= g_maps[i].n) return NULL;
  unsigned char *p = g_maps[i].mem + (size_t)k * g_maps[i].vsz;
  /* HOLD MODE (emu.h): "another packet holds this entry's spin lock" -- the one concurrency artefact a serial replay
   * never produces (REJECT_READ / REJECT_COMMIT / REJECT_SET / RETRY ..., e.g. tatp/ebpf/shard_kern.c:173-178).  The
   * emulator owns the map memory: it sets the lock word of every cache entry / lock unit the program looks up during
   * this request and clears it again when the program is back; the UNMODIFIED program answers what it answers. */
#ifndef EMU_NO_TC
  if ((g_emu_hold & 1) && value_size == sizeof(struct cache_entry)) emu_hold_word((uint64_t *)(p + offsetof(struct cache_entry, lock)));
#endif
#ifdef EMU_HAS_LOCK_UNIT
  if ((g_emu_hold & 2) && value_size == sizeof(struct lock_unit)) emu_hold_word((uint64_t *)(p + offsetof(struct lock_unit, lock)));
#endif
  return p;
}

int emu_xdp(void *pkt, uint32_t *len) {
  struct xdp_md ctx;
  memset(&ctx, 0, sizeof ctx);
  ctx.data = (uint32_t)(uintptr_t)pkt;
  ctx.data_end = ctx.data + *len;
  int rc = EMU_XDP_FN(&ctx);
  *len = ctx.data_end - ctx.data;
  return rc;
}
int emu_tc(void *pkt, uint32_t *len) {
#ifdef EMU_NO_TC
  (void)pkt; (void)len;
  return 0;
#else
  struct __sk_buff skb;
  memset(&skb, 0, sizeof skb);
  skb.data = (uint32_t)(uintptr_t)pkt;
  skb.data_end = skb.data + *len;
  skb.len = *len;
  int rc = tps_prim_tc_main(&skb);
  *len = skb.data_end - skb.data;
  return rc;
#endif
}
size_t emu_msg_size(void) { return sizeof(struct message); }
#ifdef EMU_NO_TC  /* the micro servers (lock_fasst, lock_2pl, log_server) have no user-space fallback */
size_t emu_ext_size(void) { return 0; }
#else
size_t emu_ext_size(void) { return sizeof(struct ext_message); }
#endif

/* dump: per map (in order of first use) u64 value_size, u64 count, then {u32 index, value bytes} of every entry that
 * is not all zero */
int emu_dump_maps(const char *path) {
  FILE *f = fopen(path, "wb");
  if (!f) return -1;
  for (int i = 0; i < g_nmaps; i++) {
    uint64_t cnt = 0, vs = g_maps[i].vsz;
    for (size_t k = 0; k < g_maps[i].n; k++) {
      const unsigned char *p = g_maps[i].mem + k * g_maps[i].vsz;
      size_t b = 0;
      while (b < g_maps[i].vsz && !p[b]) b++;
      if (b < g_maps[i].vsz) cnt++;
    }
    fwrite(&vs, 8, 1, f);
    fwrite(&cnt, 8, 1, f);
    for (size_t k = 0; k < g_maps[i].n; k++) {
      const unsigned char *p = g_maps[i].mem + k * g_maps[i].vsz;
      size_t b = 0;
      while (b < g_maps[i].vsz && !p[b]) b++;
      if (b < g_maps[i].vsz) { uint32_t idx = (uint32_t)k; fwrite(&idx, 4, 1, f); fwrite(p, 1, g_maps[i].vsz, f); }
    }
  }
  fclose(f);
  return 0;
}
