/*
 * User-space stand-in for the kernel tree's tools/lib/bpf/bpf_helpers.h -- TEST INFRASTRUCTURE ONLY.
 *
 * The reference's XDP / TC programs (tatp/ebpf/lock_kern.c, smallbank/ebpf/shard_kern.c, store/ebpf/store_kern.c ...)
 * are plain C over a handful of BPF helpers.  With this header on the include path the UNMODIFIED *_kern.c files
 * compile with the host gcc: maps become arrays in process memory, the two helper calls that resize a packet adjust
 * the fake context, SEC() vanishes.  Nothing of the reference is copied; see emu_main.c for how the programs are run.
 */
#ifndef EMU_BPF_HELPERS_H
#define EMU_BPF_HELPERS_H
#include <arpa/inet.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>
#include <linux/bpf.h>
#include <linux/pkt_cls.h>

#define SEC(name)
#define __uint(name, val) int (*name)[val]
#define __type(name, val) __typeof__(val) *name
#define __always_inline inline __attribute__((always_inline))

/* BTF-style map definitions encode everything in pointer-to-array member types: recover sizes with sizeof */
void *emu_map_lookup(const void *map, size_t value_size, size_t max_entries, const void *key);
#define bpf_map_lookup_elem(map, key) \
  emu_map_lookup((const void *)(map), sizeof(*(map)->value), sizeof(*(map)->max_entries) / sizeof(int), (key))

static inline long bpf_xdp_adjust_tail(struct xdp_md *ctx, int delta) {
  ctx->data_end = (__u32)(ctx->data_end + delta);
  return 0;
}
static inline long bpf_skb_change_tail(struct __sk_buff *skb, __u32 len, __u64 flags) {
  (void)flags;
  skb->len = len;
  skb->data_end = skb->data + len;
  return 0;
}
#endif
