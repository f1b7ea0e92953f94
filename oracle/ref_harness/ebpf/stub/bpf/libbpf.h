/* User-space stand-in for <bpf/libbpf.h> -- TEST INFRASTRUCTURE ONLY: lets the reference's unmodified *_user.c loaders
 * compile; their main() (which opens / loads / attaches the BPF object) is never run by the emulator, only their
 * table set-up and the server_handler thread are. */
#ifndef EMU_LIBBPF_H
#define EMU_LIBBPF_H
#include <stdarg.h>
#include <stddef.h>
#include <linux/bpf.h>
struct bpf_object;
struct bpf_program;
enum libbpf_print_level { LIBBPF_WARN, LIBBPF_INFO, LIBBPF_DEBUG };
typedef int (*libbpf_print_fn_t)(enum libbpf_print_level, const char *, va_list);
static inline libbpf_print_fn_t libbpf_set_print(libbpf_print_fn_t fn) { (void)fn; return NULL; }
static inline struct bpf_object *bpf_object__open(const char *p) { (void)p; return NULL; }
static inline int bpf_object__load(struct bpf_object *o) { (void)o; return -1; }
static inline struct bpf_program *bpf_object__find_program_by_name(const struct bpf_object *o, const char *n) { (void)o; (void)n; return NULL; }
static inline int bpf_program__set_type(struct bpf_program *p, enum bpf_prog_type t) { (void)p; (void)t; return 0; }
static inline int bpf_object__find_map_fd_by_name(const struct bpf_object *o, const char *n) { (void)o; (void)n; return -1; }
static inline int bpf_program__fd(const struct bpf_program *p) { (void)p; return -1; }
static inline int bpf_program__pin(struct bpf_program *p, const char *path) { (void)p; (void)path; return -1; }
static inline int bpf_program__unpin(struct bpf_program *p, const char *path) { (void)p; (void)path; return -1; }
static inline int bpf_xdp_attach(int ifindex, int fd, unsigned flags, const void *opts) { (void)ifindex; (void)fd; (void)flags; (void)opts; return -1; }
static inline int bpf_xdp_detach(int ifindex, unsigned flags, const void *opts) { (void)ifindex; (void)flags; (void)opts; return -1; }
static inline int bpf_set_link_xdp_fd(int ifindex, int fd, unsigned flags) { (void)ifindex; (void)fd; (void)flags; return -1; }
#endif
