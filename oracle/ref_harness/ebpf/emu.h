/* emu.h -- the three pieces of the eBPF-flavour emulator (TEST INFRASTRUCTURE ONLY, see emu_main.c) */
#ifndef EMU_H
#define EMU_H
#include <stddef.h>
#include <stdint.h>
/* emu_kern.c: the reference's XDP / TC programs over a packet buffer below 4 GiB (xdp_md carries 32-bit pointers) */
int emu_xdp(void *pkt, uint32_t *len);   /* returns XDP_TX / XDP_PASS / ...; *len may grow (bpf_xdp_adjust_tail) */
int emu_tc(void *pkt, uint32_t *len);    /* egress hook; *len may shrink (bpf_skb_change_tail) */
size_t emu_msg_size(void);               /* sizeof(struct message) / sizeof(struct ext_message) of the included source */
size_t emu_ext_size(void);
int emu_dump_maps(const char *path);     /* every touched map entry: name-less, in definition order */
/* hold mode: while g_emu_hold is set, every cache entry (bit 0) / lock unit (bit 1) the XDP program looks up has its spin
 * lock taken by "another packet"; emu_release() gives them back.  EMU_HOLD=<bits> EMU_HOLD_EVERY=<k> in the environment of
 * a replay holds during every k-th request (emu_main.c). */
extern int g_emu_hold;
void emu_release(void);
/* emu_user.c: the reference's user-space fallback (its kvs + server_handler thread) */
void emu_user_start(void);
/* emu_main.c: the sockets the user thread thinks it has */
#endif
