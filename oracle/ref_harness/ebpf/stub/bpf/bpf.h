/* User-space stand-in for <bpf/bpf.h> -- TEST INFRASTRUCTURE ONLY (see libbpf.h next to it). */
#ifndef EMU_BPF_H
#define EMU_BPF_H
#include <linux/bpf.h>
static inline int bpf_map_update_elem(int fd, const void *key, const void *value, unsigned long long flags) {
  (void)fd; (void)key; (void)value; (void)flags; return -1;
}
#endif
