// txn_clients.h -- the reference clients' transaction state machines, one source for the host driver
// (txn_driver.cc) and the GPU-resident driver (k_txn.hip).
//
// TATP      tatp/caladan/client_udp_shard.cc:177-1117 (7 transactions), mix tatp/caladan/tatp.h:57-63 via
//           CreateWorkgenArr (:63-73), ClientLoop (:1120-1185), keys tatp_nurand (tatp.h:40-43)
// SmallBank smallbank/caladan/client_udp_shard.cc:169-1240 (6 transactions), mix smallbank.h:63-68,
//           account pickers smallbank.h:30-50
// A transaction is a sequence of PHASES; the messages of one phase are sent together (the reference sends them from
// one uthread per shard and joins) and all replies are awaited.  `tx_run` advances one client by one phase: it
// finishes transactions that need no further message, starts the next one, and returns the phase's messages in a
// TxOut (at most TX_MAXOUT; for each the destination shard and the client message that receives the reply).
// Request messages start zeroed (the reference sends uninitialised stack bytes in the fields a request does not set);
// everything a transaction decides on -- reply types, versions, value bytes -- follows the reference client line by
// line (cited at each step).  Everything here is integer / IEEE-single arithmetic with no library calls, so the host
// and the device produce the same bytes.
#pragma once
#include <stdint.h>
#include <string.h>

#include "zipf_table.h"

#if defined(__HIPCC__)
#define TX_HD __host__ __device__
#else
#define TX_HD
#endif

struct TxLcg {  // fastrand, tatp/caladan/tatp.h:31-34
  uint64_t s;
  TX_HD uint32_t next() {
    s = s * 1103515245ull + 12345ull;
    return (uint32_t)(s >> 32);
  }
};

#pragma pack(push, 1)
struct TatpMsg {  // tatp/caladan/proto.h (same layout as tatp/udp/net.h:57-66)
  uint8_t ord, type, table;
  uint64_t key;
  uint8_t val[40];
  uint32_t ver;
};
struct SbMsg {  // smallbank/udp/net.h:41-50
  uint8_t ord, type, table;
  uint64_t key;
  uint8_t val[8];
  uint32_t ver;
};
#pragma pack(pop)
static_assert(sizeof(TatpMsg) == 55 && sizeof(SbMsg) == 23, "packed wire structs");

enum : uint8_t {  // tatp PktType, tatp/udp/net.h:15-52
  T_READ = 0, T_ACQ = 1, T_ABORT = 2, T_GRANT_READ = 4, T_NOT_EXIST = 6, T_GRANT_LOCK = 7, T_REJECT_LOCK = 8,
  T_COMMIT_PRIM = 12, T_COMMIT_BCK = 13, T_COMMIT_LOG = 14, T_INSERT_PRIM = 18, T_INSERT_BCK = 19,
  T_DELETE_PRIM = 22, T_DELETE_BCK = 23, T_DELETE_LOG = 24,
};
enum : uint8_t { TB_SUB = 0, TB_SEC = 1, TB_AI = 2, TB_SF = 3, TB_CF = 4 };
enum : uint8_t {  // smallbank PktType, smallbank/udp/net.h:15-38
  S_ACQ_SH = 0, S_ACQ_EX = 1, S_REL_SH = 2, S_REL_EX = 3, S_COMMIT_PRIM = 4, S_COMMIT_BCK = 5, S_COMMIT_LOG = 6,
  S_GRANT_SH = 7, S_REJECT_SH = 8, S_GRANT_EX = 9, S_REJECT_EX = 10,
};

#define TX_MAXOUT 9      // messages one phase of one client emits at most (smallbank: 3 rows logged on 3 shards)
#define TX_NO_DST 0xFFu  // the reply is discarded
#define TX_DST_MASK 0x3Fu  // out_dst = working message number | flags (TX_FULL, txn_clients.h below); TX_NO_DST is checked first

struct TxParams {
  uint32_t workload;        // DINT_WL_TATP / DINT_WL_SMALLBANK
  uint32_t key_dist;        // 0 = the reference's own distribution, 1 = Zipf (zipf_cdf)
  uint64_t n_rows;
  uint64_t n_hot;           // smallbank: kHotAccountNum scaled to n_rows
  const uint32_t *zipf_cdf; // n_rows thresholds (host or device memory, matching the caller)
  uint8_t workgen[100];     // transaction type per percentile (CreateWorkgenArr)
};

// what one phase emits, and which transactions finished on the way.  A message is always one of the client's working
// messages as it stands when the phase returns, with a type of its own (the reference re-uses one struct for the
// LOG / BCK / PRIM copies of a row): the queue holds {shard, source message, type, reply destination} and the
// caller materialises the wire message -- on the GPU that keeps a phase in registers instead of ~0.5 KB of scratch.
//
// A NEW request (a READ / ACQUIRE of a row: {type, table, key}, every other byte zero) never touches the working
// messages on its way out: send_new() keeps the three fields in the queue, the wire message is built from them, and
// the reply -- the whole message, key and table included -- is what fills the client message it is addressed to.
// (Until r03 the request was first stored into that client message and read back by materialize(): on the GPU one
// scattered 55-byte store and load per message, both dead -- NOTEBOOK.md section 4.)
template <class Msg, int CAP>
struct TxOut {
  uint8_t n;
  uint8_t shard[CAP], dst[CAP], src[CAP], type[CAP], ord[CAP];
  uint8_t n_fin, fin_txn[2], fin_ok[2];
  uint16_t fresh;                // bit k: message k is a new request, built from table[k] / key[k]
  uint8_t table[CAP];
  uint64_t key[CAP];
  TX_HD void clear() { n = 0; n_fin = 0; fresh = 0; }
  // queue client message `src_msg` (sent with type `ty`) for shard s; its reply lands in client message `d`
  TX_HD void send(uint32_t s, uint8_t src_msg, uint8_t ty, uint8_t d) {
    uint8_t j = 0;  // msg->ord = position in this phase's queue for shard s (client_udp_shard.cc:376-380)
    for (uint8_t k = 0; k < n; k++) j += shard[k] == s;
    shard[n] = (uint8_t)s;
    dst[n] = d;
    src[n] = src_msg;
    type[n] = ty;
    ord[n] = j;
    n++;
  }
  // queue a new request {ty, tb, ky} for the key's primary (key % 3, client_udp_shard.cc:187); reply into message `d`
  TX_HD void send_new(uint8_t ty, uint8_t tb, uint64_t ky, uint8_t d) {
    fresh |= (uint16_t)(1u << n);
    table[n] = tb;
    key[n] = ky;
    send((uint32_t)(ky % 3), d, ty, d);
  }
  TX_HD bool is_new(uint8_t k) const { return (fresh >> k) & 1u; }
  template <class Client>
  TX_HD Msg materialize(const Client &c, uint8_t k) const {
    Msg m;
    if (is_new(k)) {
      memset(&m, 0, sizeof m);
      m.table = table[k];
      m.key = key[k];
    } else {
      m = c.m[src[k]];
      m.ver = c.wire_ver(src[k], m.ver);
    }
    m.type = type[k];
    m.ord = ord[k];
    return m;
  }
  TX_HD void finish(uint8_t txn, bool committed) {
    if (n_fin < 2) { fin_txn[n_fin] = txn; fin_ok[n_fin] = committed; }
    n_fin++;
  }
};

// 3 decimal digits -> 3 BCD nibbles (create_map1000); s_id -> sub_nbr (tatp_sid_to_sub_nbr, tatp.h:132-144)
TX_HD static inline uint64_t tx_bcd3(uint32_t v) { return ((uint64_t)(v / 100 % 10) << 8) | ((uint64_t)(v / 10 % 10) << 4) | (v % 10); }
TX_HD static inline uint64_t tx_sub_nbr(uint32_t s) {
  return tx_bcd3(s % 1000) | (tx_bcd3(s / 1000 % 1000) << 12) | (tx_bcd3(s / 1000000 % 1000) << 24);
}

// ======================================================================================================= TATP
enum : uint8_t { TT_GET_SUB = 0, TT_GET_NEW_DEST = 1, TT_GET_ACCESS = 2, TT_UPD_SUB = 3, TT_UPD_LOC = 4, TT_INS_CF = 5, TT_DEL_CF = 6 };
// the working messages of the running transaction (names as in the reference functions)
enum : uint8_t { A_READ = 0, A_LOCK = 1, B_READ = 2, B_LOCK = 3, A_VER = 4, B_VER = 5, TMP0 = 6, TMP1 = 7, TMP2 = 8, TATP_NMSG = 9 };

// A client's working messages: message k at base + k * stride BYTES.  The host driver keeps a client's messages
// together (stride = sizeof(M)).  The device driver keeps message k of ALL clients together, one 64-byte sector each
// (base = store + 64 * client, stride = 64 * number of clients): the lanes of a wave are consecutive clients, one
// access of the wave is one run of consecutive sectors, and a message never straddles two (packed at 55 bytes every
// access touched 1.86 sectors on average).
#define TX_DEV_MSG_STRIDE 64u
template <class M> struct TxMsgs {
  uint8_t *base;
  uint64_t stride;
  TX_HD M &operator[](uint32_t k) const { return *(M *)(base + (size_t)k * stride); }
};

// Client state = a 128-byte header (loaded and stored whole: on the GPU one coalesced access per client and phase)
// + TATP_NMSG working messages in a separate array.  The header carries what the phase logic READS of the replies --
// reply type, version, first value byte per working message (`rt`, `rver`, `rv0`, written when a reply is consumed):
// on the GPU the logic then runs out of registers.  Only the rows a transaction writes back (A_READ / B_READ of the
// four update transactions: their 40 value bytes travel into the LOG / BCK / PRIM messages) are kept whole in the
// working messages (`TX_FULL` on the reply's destination); every other reply leaves three fields behind.  Until r03
// every reply was stored whole and every decision was a scattered load from the message array (NOTEBOOK.md section 4).
#define TX_FULL 0x40u  // out_dst flag: keep the whole reply in the working message, not only its summary
struct TatpClient {
  TxLcg rng;
  uint8_t txn, step, n_out;  // step 0 = idle
  uint8_t sf_type, start_time, end_time;
  uint8_t out_shard[6], out_dst[6];  // a tatp phase emits at most 6 messages; out_dst = working message | TX_FULL, or TX_NO_DST
  uint32_t s_id;
  uint32_t out_pos[6];
  TxMsgs<TatpMsg> m;  // [TATP_NMSG] -- set by the driver before every use (host vector / device array)
  uint8_t rt[TATP_NMSG], rv0[TATP_NMSG];  // per working message: type / val[0] of the last reply (or what the logic set)
  uint8_t pad_[2];
  uint32_t rver[TATP_NMSG - 1];           // ... its version (TMP2 never needs one)
  uint32_t vlr;                           // UpdateLocation: the new vlr_location, parked until the subscriber row is read
  // a consumed reply: the three fields the logic may ask for.  (Unrolled over the message numbers, so that the arrays
  // stay in registers on the GPU.)
  TX_HD void note_reply(uint8_t d, uint8_t type, uint8_t v0, uint32_t ver) {
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (uint8_t j = 0; j < TATP_NMSG; j++)
      if (j == d) {
        rt[j] = type; rv0[j] = v0;
        if (j < TATP_NMSG - 1) rver[j] = ver;
      }
  }
  // version on the wire of working message `src` when it is sent again (the LOG / BCK / PRIM copies of a row)
  TX_HD uint32_t wire_ver(uint8_t src, uint32_t stored) const {
    uint32_t v = stored;
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (uint8_t j = 0; j < TATP_NMSG - 1; j++)
      if (j == src) v = rver[j];
    return v;
  }
};
static_assert(sizeof(TatpClient) == 128, "client header");

TX_HD static inline void tatp_workgen(uint8_t *workgen) {
  // CreateWorkgenArr :63-73 -- note the order: GetSubscriberData, GetAccessData, GetNewDestination, ...
  const uint8_t freq[7][2] = {{TT_GET_SUB, 35}, {TT_GET_ACCESS, 35}, {TT_GET_NEW_DEST, 10}, {TT_UPD_SUB, 2},
                              {TT_UPD_LOC, 14}, {TT_INS_CF, 2}, {TT_DEL_CF, 2}};
  int k = 0;
  for (int f = 0; f < 7; f++)
    for (int j = 0; j < freq[f][1]; j++) workgen[k++] = freq[f][0];
}

TX_HD static inline uint32_t tatp_pick_sid(TxLcg &g, const TxParams &P) {
  if (P.key_dist == 1) return (uint32_t)zipf_lookup(P.zipf_cdf, P.n_rows, g.next());
  const uint32_t n = (uint32_t)P.n_rows;  // tatp_nurand, A = 1048575
  const uint32_t x = g.next() % n, y = g.next() & 1048575u;
  return (x | y) % n;
}

TX_HD static inline uint64_t tatp_sf_key(const TatpClient &c) { return (uint64_t)c.s_id | ((uint64_t)c.sf_type << 32); }
TX_HD static inline uint64_t tatp_cf_key(const TatpClient &c, uint32_t st) { return tatp_sf_key(c) | ((uint64_t)st << 40); }
// The row behind working message i of the running update transaction: the B_* messages are the SPECIAL_FACILITY row,
// the A_* ones the SUBSCRIBER row (UpdateSubscriberData, UpdateLocation) or the CALL_FORWARDING row (Insert / Delete
// CallForwarding).  The reference reads key and table out of the struct the reply came back in; they are the request's.
TX_HD static inline bool tatp_is_b(uint8_t i) { return i == B_READ || i == B_LOCK || i == B_VER; }
TX_HD static inline uint8_t tatp_row_table(const TatpClient &c, uint8_t i) {
  return tatp_is_b(i) ? TB_SF : ((c.txn == TT_INS_CF || c.txn == TT_DEL_CF) ? TB_CF : TB_SUB);
}
TX_HD static inline uint64_t tatp_row_key(const TatpClient &c, uint8_t i) {
  return tatp_is_b(i) ? tatp_sf_key(c) : ((c.txn == TT_INS_CF || c.txn == TT_DEL_CF) ? tatp_cf_key(c, c.start_time) : (uint64_t)c.s_id);
}

typedef TxOut<TatpMsg, 6> TatpOut;  // a tatp phase emits at most 6 messages
// a new request {type, table, key} to the key's primary; the reply's summary goes to working message i (TX_NO_DST:
// nowhere), the whole reply too when the transaction will write the row back (full)
TX_HD static inline void tatp_send_new(TatpOut &o, uint8_t i, uint8_t type, uint8_t table, uint64_t key, bool full = false) {
  o.send_new(type, table, key, (uint8_t)(full ? (i | TX_FULL) : i));
}
// release the lock behind working message i (A_LOCK / B_LOCK): the reference turns the granted lock message into an
// ABORT and sends it back (:402-418); it carries the row's key and table and nothing else
TX_HD static inline void tatp_send_abort(TatpClient &c, TatpOut &o, uint8_t i) {
  o.send_new(T_ABORT, tatp_row_table(c, i), tatp_row_key(c, i), i);
}
// the stored row i (A_READ / B_READ) with type `ty`: to its primary, reply discarded
TX_HD static inline void tatp_send_only(TatpClient &c, TatpOut &o, uint8_t i, uint8_t ty) { o.send((uint32_t)(tatp_row_key(c, i) % 3), i, ty, TX_NO_DST); }
TX_HD static inline void tatp_send_log3(TatpClient &, TatpOut &o, uint8_t i, uint8_t ty) { for (uint32_t s = 0; s < 3; s++) o.send(s, i, ty, TX_NO_DST); }
// backups of rows whose primary is key % 3: first the "+1" copies of every row, then the "+2" copies
TX_HD static inline void tatp_send_bck(TatpClient &c, TatpOut &o, uint8_t ty, uint8_t r0, int n, uint8_t r1 = 0) {
  const uint8_t rows[2] = {r0, r1};
  for (int i = 0; i < n; i++) o.send((uint32_t)((tatp_row_key(c, rows[i]) % 3 + 1) % 3), rows[i], ty, TX_NO_DST);
  for (int i = 0; i < n; i++) o.send((uint32_t)((tatp_row_key(c, rows[i]) % 3 + 2) % 3), rows[i], ty, TX_NO_DST);
}
TX_HD static inline void tatp_finish(TatpClient &c, TatpOut &o, bool committed) {
  o.finish(c.txn, committed);
  c.step = 0;
}

TX_HD static inline void tatp_begin(TatpClient &c, const TxParams &P) {
  c.txn = P.workgen[c.rng.next() % 100];  // ClientLoop :1144
  c.step = 1;
  TxLcg &g = c.rng;
  switch (c.txn) {
    case TT_GET_SUB: c.s_id = tatp_pick_sid(g, P); break;                                                        // :180
    case TT_GET_ACCESS: c.s_id = tatp_pick_sid(g, P); c.sf_type = (uint8_t)((g.next() & 3) + 1); break;          // :308-309 (ai_type)
    case TT_GET_NEW_DEST: case TT_INS_CF:                                                                        // :207-210, :737-740
      c.s_id = tatp_pick_sid(g, P); c.sf_type = (uint8_t)(g.next() % 4 + 1); c.start_time = (uint8_t)(g.next() % 3 * 8);
      c.end_time = (uint8_t)(g.next() % 24);
      break;
    case TT_UPD_SUB: c.s_id = tatp_pick_sid(g, P); c.sf_type = (uint8_t)(g.next() % 4 + 1); break;               // :340-341
    case TT_UPD_LOC: c.s_id = tatp_pick_sid(g, P); c.vlr = g.next(); break;                                      // :579-580
    default: c.s_id = tatp_pick_sid(g, P); c.sf_type = (uint8_t)(g.next() % 4 + 1); c.start_time = (uint8_t)(g.next() % 3 * 8); break;  // DEL_CF :960-962
  }
}

// The step numbers below are this driver's; each case cites the reference lines it restates.
TX_HD static inline void tatp_emit_upd_sub(TatpClient &c, TatpOut &o) {  // TxnUpdateSubscriberData :334-571
  for (;;) {
    switch (c.step) {
      case 1:  // execute: read + lock both rows :345-396
        tatp_send_new(o, A_READ, T_READ, TB_SUB, c.s_id, true); tatp_send_new(o, A_LOCK, T_ACQ, TB_SUB, c.s_id);
        tatp_send_new(o, B_READ, T_READ, TB_SF, tatp_sf_key(c), true); tatp_send_new(o, B_LOCK, T_ACQ, TB_SF, tatp_sf_key(c));
        c.step = 2;
        return;
      case 2:
        if (c.rt[B_READ] == T_NOT_EXIST || c.rt[A_LOCK] == T_REJECT_LOCK || c.rt[B_LOCK] == T_REJECT_LOCK) {  // :400
          c.step = 10;
          continue;
        }
        {  // :425-431
          const uint16_t bits = (uint16_t)c.rng.next();
          memcpy(c.m[A_READ].val + 30, &bits, 2);
          c.m[B_READ].val[2] = (uint8_t)c.rng.next();  // data_a
        }
        tatp_send_new(o, A_VER, T_READ, TB_SUB, c.s_id); tatp_send_new(o, B_VER, T_READ, TB_SF, tatp_sf_key(c));  // verify :433-447
        c.step = 3;
        return;
      case 3:
        if (c.rver[A_READ] != c.rver[A_VER] || c.rver[B_READ] != c.rver[B_VER]) { c.step = 12; continue; }  // :470
        c.rver[A_READ]++; c.rver[B_READ]++;                                                                  // :487-488
        for (uint32_t s = 0; s < 3; s++) { o.send(s, A_READ, T_COMMIT_LOG, TX_NO_DST); o.send(s, B_READ, T_COMMIT_LOG, TX_NO_DST); }  // :493-501
        c.step = 4;
        return;
      case 4: tatp_send_bck(c, o, T_COMMIT_BCK, A_READ, 2, B_READ); c.step = 5; return;  // :521-533
      case 5: tatp_send_only(c, o, A_READ, T_COMMIT_PRIM); tatp_send_only(c, o, B_READ, T_COMMIT_PRIM); c.step = 6; return;  // :552-556
      case 6: tatp_finish(c, o, true); return;
      // abort after a failed execute: release the granted locks one round trip at a time :402-418
      case 10:
        if (c.rt[A_LOCK] == T_GRANT_LOCK) { tatp_send_abort(c, o, A_LOCK); c.step = 11; return; }
        c.step = 11;
        continue;
      case 11:
        if (c.rt[B_LOCK] == T_GRANT_LOCK) { tatp_send_abort(c, o, B_LOCK); c.step = 14; return; }
        tatp_finish(c, o, false);
        return;
      // abort after a failed validation: both locks are held :472-481
      case 12: tatp_send_abort(c, o, A_LOCK); c.step = 13; return;
      case 13: tatp_send_abort(c, o, B_LOCK); c.step = 14; return;
      default: tatp_finish(c, o, false); return;
    }
  }
}

TX_HD static inline void tatp_emit_upd_loc(TatpClient &c, TatpOut &o) {  // TxnUpdateLocation :574-728
  switch (c.step) {
    case 1: tatp_send_new(o, TX_NO_DST, T_READ, TB_SEC, tx_sub_nbr(c.s_id)); c.step = 2; return;  // :583-592 (the reply is only asserted on)
    case 2:
      tatp_send_new(o, A_READ, T_READ, TB_SUB, c.s_id, true); tatp_send_new(o, A_LOCK, T_ACQ, TB_SUB, c.s_id);  // :605-618
      c.step = 3;
      return;
    case 3:
      if (c.rt[A_LOCK] == T_REJECT_LOCK) { tatp_finish(c, o, false); return; }  // :645
      memcpy(c.m[A_READ].val + 36, &c.vlr, 4);                                  // vlr_location :650
      tatp_send_new(o, A_VER, T_READ, TB_SUB, c.s_id);                          // verify :653-660
      c.step = 4;
      return;
    case 4:
      if (c.rver[A_VER] != c.rver[A_READ]) { tatp_send_abort(c, o, A_LOCK); c.step = 8; return; }  // :667-674
      c.rver[A_READ]++;
      tatp_send_log3(c, o, A_READ, T_COMMIT_LOG);  // :677-684
      c.step = 5;
      return;
    case 5: tatp_send_bck(c, o, T_COMMIT_BCK, A_READ, 1); c.step = 6; return;  // :702-708
    case 6: tatp_send_only(c, o, A_READ, T_COMMIT_PRIM); c.step = 7; return;   // :722-723
    case 7: tatp_finish(c, o, true); return;
    default: tatp_finish(c, o, false); return;
  }
}

TX_HD static inline void tatp_emit_ins_cf(TatpClient &c, TatpOut &o) {  // TxnInsertCallForwarding :731-951
  switch (c.step) {
    case 1: tatp_send_new(o, TX_NO_DST, T_READ, TB_SEC, tx_sub_nbr(c.s_id)); c.step = 2; return;  // :743-752 (asserted on only)
    case 2: tatp_send_new(o, B_READ, T_READ, TB_SF, tatp_sf_key(c)); c.step = 3; return;         // :761-769
    case 3:
      if (c.rt[B_READ] == T_NOT_EXIST) { tatp_finish(c, o, false); return; }  // :776
      tatp_send_new(o, A_READ, T_READ, TB_CF, tatp_cf_key(c, c.start_time), true); tatp_send_new(o, A_LOCK, T_ACQ, TB_CF, tatp_cf_key(c, c.start_time));  // :789-799
      c.step = 4;
      return;
    case 4:
      if (c.rt[A_READ] == T_GRANT_READ || c.rt[A_LOCK] == T_REJECT_LOCK) {  // the row exists, or no lock :826
        if (c.rt[A_LOCK] == T_GRANT_LOCK) { tatp_send_abort(c, o, A_LOCK); c.step = 9; return; }
        tatp_finish(c, o, false);
        return;
      }
      c.m[A_READ].val[1] = 101;         // numberx[0] magic :842
      c.m[A_READ].val[0] = c.end_time;  // :843
      tatp_send_new(o, B_VER, T_READ, TB_SF, tatp_sf_key(c)); tatp_send_new(o, A_VER, T_READ, TB_CF, tatp_cf_key(c, c.start_time));  // verify :846-859
      c.step = 5;
      return;
    case 5:
      if (c.rver[B_READ] != c.rver[B_VER] || c.rt[A_VER] == T_GRANT_READ) {  // :884
        tatp_send_abort(c, o, A_LOCK); c.step = 9; return;
      }
      c.rver[A_READ] = 0;  // :896
      tatp_send_log3(c, o, A_READ, T_COMMIT_LOG);
      c.step = 6;
      return;
    case 6: tatp_send_bck(c, o, T_INSERT_BCK, A_READ, 1); c.step = 7; return;  // :921-927
    case 7: tatp_send_only(c, o, A_READ, T_INSERT_PRIM); c.step = 8; return;   // :944-945
    case 8: tatp_finish(c, o, true); return;
    default: tatp_finish(c, o, false); return;
  }
}

TX_HD static inline void tatp_emit_del_cf(TatpClient &c, TatpOut &o) {  // TxnDeleteCallForwarding :954-1117
  switch (c.step) {
    case 1: tatp_send_new(o, TX_NO_DST, T_READ, TB_SEC, tx_sub_nbr(c.s_id)); c.step = 2; return;  // :965-974 (asserted on only)
    case 2:
      tatp_send_new(o, A_READ, T_READ, TB_CF, tatp_cf_key(c, c.start_time), true); tatp_send_new(o, A_LOCK, T_ACQ, TB_CF, tatp_cf_key(c, c.start_time));  // :983-997
      c.step = 3;
      return;
    case 3:
      if (c.rt[A_READ] == T_NOT_EXIST || c.rt[A_LOCK] == T_REJECT_LOCK) {  // :1024
        if (c.rt[A_LOCK] == T_GRANT_LOCK) { tatp_send_abort(c, o, A_LOCK); c.step = 8; return; }
        tatp_finish(c, o, false);
        return;
      }
      tatp_send_new(o, A_VER, T_READ, TB_CF, tatp_cf_key(c, c.start_time));  // verify :1040-1046
      c.step = 4;
      return;
    case 4:
      if (c.rt[A_VER] == T_NOT_EXIST || c.rver[A_VER] != c.rver[A_READ]) {  // :1052
        tatp_send_abort(c, o, A_LOCK); c.step = 8; return;
      }
      tatp_send_log3(c, o, A_READ, T_DELETE_LOG);  // :1063
      c.step = 5;
      return;
    case 5: tatp_send_bck(c, o, T_DELETE_BCK, A_READ, 1); c.step = 6; return;  // :1087-1093
    case 6: tatp_send_only(c, o, A_READ, T_DELETE_PRIM); c.step = 7; return;   // :1110-1111
    case 7: tatp_finish(c, o, true); return;
    default: tatp_finish(c, o, false); return;
  }
}

TX_HD static inline void tatp_emit(TatpClient &c, TatpOut &o) {
  switch (c.txn) {
    case TT_GET_SUB:  // TxnGetSubscriberData :177-199
      // the reply is never looked at (the reference only asserts on it, :187-198): it is not kept
      if (c.step == 1) { tatp_send_new(o, TX_NO_DST, T_READ, TB_SUB, c.s_id); c.step = 2; }
      else tatp_finish(c, o, true);
      return;
    case TT_GET_ACCESS:  // TxnGetAccessData :305-331
      if (c.step == 1) { tatp_send_new(o, A_READ, T_READ, TB_AI, tatp_sf_key(c)); c.step = 2; }
      else tatp_finish(c, o, c.rt[A_READ] != T_NOT_EXIST);
      return;
    case TT_GET_NEW_DEST:  // TxnGetNewDestination :202-302
      if (c.step == 1) { tatp_send_new(o, A_READ, T_READ, TB_SF, tatp_sf_key(c)); c.step = 2; return; }
      if (c.step == 2) {
        if (c.rt[A_READ] == T_NOT_EXIST || c.rv0[A_READ] == 0) { tatp_finish(c, o, false); return; }  // :239,244 (is_active)
        const uint32_t n = c.start_time / 8u + 1;                                                     // cf_to_fetch :212
        for (uint32_t i = 0; i < n; i++) tatp_send_new(o, (uint8_t)(TMP0 + i), T_READ, TB_CF, tatp_cf_key(c, i * 8));
        c.step = 3;
        return;
      }
      {
        bool ok = false;  // :283-297
        const uint32_t n = c.start_time / 8u + 1;
#if defined(__HIPCC__)
#pragma unroll
#endif
        for (uint32_t i = 0; i < 3; i++)
          if (i < n && c.rt[TMP0 + i] != T_NOT_EXIST && i * 8 <= c.start_time && c.end_time < c.rv0[TMP0 + i]) ok = true;
        tatp_finish(c, o, ok);
      }
      return;
    case TT_UPD_SUB: tatp_emit_upd_sub(c, o); return;
    case TT_UPD_LOC: tatp_emit_upd_loc(c, o); return;
    case TT_INS_CF: tatp_emit_ins_cf(c, o); return;
    default: tatp_emit_del_cf(c, o); return;
  }
}

// one phase of one client: emit messages (o.n > 0) -- finishing / starting transactions on the way
TX_HD static inline void tatp_run(TatpClient &c, const TxParams &P, TatpOut &o) {
  o.clear();
  for (;;) {
    if (c.step == 0) tatp_begin(c, P);
    tatp_emit(c, o);
    if (o.n) break;  // waiting for replies
    // the phase emitted nothing: the transaction finished inside emit() -> start the next one
  }
  c.n_out = o.n;
#if defined(__HIPCC__)
#pragma unroll
#endif
  for (uint8_t k = 0; k < 6; k++)
    if (k < o.n) { c.out_shard[k] = o.shard[k]; c.out_dst[k] = o.dst[k]; }
}

// ================================================================================================== SmallBank
// All six transactions are 2PL: acquire every lock (the grant carries the row), compute, log x3, backup x2,
// primary, release.  RETRY replies never occur against a serial server, so the retry loops of the reference
// (:212-246 and siblings) never spin here; a REJECT aborts by releasing whatever was granted, one at a time.
enum : uint8_t { ST_AMALGAMATE = 0, ST_BALANCE = 1, ST_DEPOSIT_CHECKING = 2, ST_SEND_PAYMENT = 3, ST_TRANSACT_SAVING = 4, ST_WRITE_CHECK = 5 };

#define SB_NMSG 3
struct SbClient {
  TxLcg rng;
  uint8_t txn, step, n_out;
  uint8_t n_rows;     // rows of this transaction
  uint8_t n_write, wr[3];  // indices into m[] of the rows written back
  uint8_t rel;        // abort: next row to release
  uint8_t out_shard[TX_MAXOUT], out_dst[TX_MAXOUT];
  uint8_t pad0;
  float amount;
  uint64_t a0, a1;
  uint32_t out_pos[TX_MAXOUT];
  TxMsgs<SbMsg> m;    // [SB_NMSG] the locked rows, in the reference's order
  // every reply is a row the transaction computes on: kept whole (TX_FULL), no summary
  TX_HD void note_reply(uint8_t, uint8_t, uint8_t, uint32_t) {}
  TX_HD uint32_t wire_ver(uint8_t, uint32_t stored) const { return stored; }
};
static_assert(sizeof(SbClient) == 112, "client header");
typedef TxOut<SbMsg, TX_MAXOUT> SbOut;

TX_HD static inline void sb_workgen(uint8_t *workgen) {
  // CreateWorkgenArr, smallbank/caladan/client_udp_shard.cc: same construction as tatp, mix smallbank.h:63-68
  const uint8_t freq[6][2] = {{ST_AMALGAMATE, 15}, {ST_BALANCE, 15}, {ST_DEPOSIT_CHECKING, 15}, {ST_SEND_PAYMENT, 25},
                              {ST_TRANSACT_SAVING, 15}, {ST_WRITE_CHECK, 15}};
  int k = 0;
  for (int f = 0; f < 6; f++)
    for (int j = 0; j < freq[f][1]; j++) workgen[k++] = freq[f][0];
}
TX_HD static inline void sb_get_account(TxLcg &g, const TxParams &P, uint64_t *a) {  // smallbank.h:30-36
  if (P.key_dist == 1) { *a = zipf_lookup(P.zipf_cdf, P.n_rows, g.next()); return; }
  if (g.next() % 100 < 90) *a = g.next() % P.n_hot;
  else *a = g.next() % P.n_rows;
}
TX_HD static inline void sb_get_two_accounts(TxLcg &g, const TxParams &P, uint64_t *a, uint64_t *b) {  // smallbank.h:38-50
  if (P.key_dist == 1) {
    *a = zipf_lookup(P.zipf_cdf, P.n_rows, g.next());
    do { *b = zipf_lookup(P.zipf_cdf, P.n_rows, g.next()); } while (*b == *a && P.n_rows > 1);
    return;
  }
  const uint64_t n = (g.next() % 100 < 90) ? P.n_hot : P.n_rows;
  *a = g.next() % n;
  *b = g.next() % n;
  while (*b == *a && n > 1) *b = g.next() % n;
}
TX_HD static inline float sb_bal(const SbMsg &m) { float f; memcpy(&f, m.val + 4, 4); return f; }
TX_HD static inline void sb_set_bal(SbMsg &m, float f) { memcpy(m.val + 4, &f, 4); }
TX_HD static inline bool sb_granted(const SbMsg &m) { return m.type == S_GRANT_SH || m.type == S_GRANT_EX; }

TX_HD static inline void sb_begin(SbClient &c, const TxParams &P) {
  TxLcg &g = c.rng;
  c.txn = P.workgen[g.next() % 100];
  c.step = 1;
  c.rel = 0;
  c.n_write = 0;
  // lock set of each transaction, in the order the reference pushes the messages (the requests themselves: sb_lock_row)
  switch (c.txn) {
    case ST_AMALGAMATE:  // TxnAmalgamate :169-438: X(sav a0), X(chk a0), X(chk a1)
      sb_get_two_accounts(g, P, &c.a0, &c.a1);
      c.n_rows = 3;
      break;
    case ST_BALANCE:  // TxnBalance :441-578: S(sav), S(chk); read only
      sb_get_account(g, P, &c.a0);
      c.n_rows = 2;
      break;
    case ST_DEPOSIT_CHECKING:  // TxnDepositChecking :581-684: X(chk); bal += 1.3
      sb_get_account(g, P, &c.a0);
      c.amount = 1.3f;
      c.n_rows = 1;
      break;
    case ST_SEND_PAYMENT:  // TxnSendPayment :687-932: X(chk a0), X(chk a1); move 5.0 if funds suffice
      sb_get_two_accounts(g, P, &c.a0, &c.a1);
      c.amount = 5.0f;
      c.n_rows = 2;
      break;
    case ST_TRANSACT_SAVING:  // TxnTransactSaving :935-1038: X(sav); bal += 20.20
      sb_get_account(g, P, &c.a0);
      c.amount = 20.20f;
      c.n_rows = 1;
      break;
    default:  // TxnWriteCheck :1041-1240: S(sav), X(chk); chk -= 5 (+1 penalty when overdrawn)
      sb_get_account(g, P, &c.a0);
      c.amount = 5.0f;
      c.n_rows = 2;
      break;
  }
}

// row i of the transaction's lock set: {lock type, table (0 savings, 1 checking), account}
TX_HD static inline void sb_lock_row(const SbClient &c, uint8_t i, uint8_t *type, uint8_t *table, uint64_t *key) {
  switch (c.txn) {
    case ST_AMALGAMATE: *type = S_ACQ_EX; *table = i == 0 ? 0 : 1; *key = i == 2 ? c.a1 : c.a0; return;
    case ST_BALANCE: *type = S_ACQ_SH; *table = i; *key = c.a0; return;
    case ST_DEPOSIT_CHECKING: *type = S_ACQ_EX; *table = 1; *key = c.a0; return;
    case ST_SEND_PAYMENT: *type = S_ACQ_EX; *table = 1; *key = i == 0 ? c.a0 : c.a1; return;
    case ST_TRANSACT_SAVING: *type = S_ACQ_EX; *table = 0; *key = c.a0; return;
    default: *type = i == 0 ? S_ACQ_SH : S_ACQ_EX; *table = i; *key = c.a0; return;  // WriteCheck
  }
}

// compute phase: returns false when the transaction aborts by its own logic after locking
TX_HD static inline bool sb_compute(SbClient &c) {
  switch (c.txn) {
    case ST_AMALGAMATE:  // :296-299
      sb_set_bal(c.m[2], sb_bal(c.m[2]) + sb_bal(c.m[0]) + sb_bal(c.m[1]));
      sb_set_bal(c.m[0], 0); sb_set_bal(c.m[1], 0);
      c.n_write = 3; c.wr[0] = 0; c.wr[1] = 1; c.wr[2] = 2;
      return true;
    case ST_BALANCE: c.n_write = 0; return true;
    case ST_DEPOSIT_CHECKING: sb_set_bal(c.m[0], sb_bal(c.m[0]) + c.amount); c.n_write = 1; c.wr[0] = 0; return true;
    case ST_SEND_PAYMENT:
      if (sb_bal(c.m[0]) < c.amount) return false;  // insufficient funds: release and abort
      sb_set_bal(c.m[0], sb_bal(c.m[0]) - c.amount); sb_set_bal(c.m[1], sb_bal(c.m[1]) + c.amount);
      c.n_write = 2; c.wr[0] = 0; c.wr[1] = 1;
      return true;
    case ST_TRANSACT_SAVING: sb_set_bal(c.m[0], sb_bal(c.m[0]) + c.amount); c.n_write = 1; c.wr[0] = 0; return true;
    default:
      if (sb_bal(c.m[0]) + sb_bal(c.m[1]) < c.amount) sb_set_bal(c.m[1], sb_bal(c.m[1]) - (c.amount + 1));
      else sb_set_bal(c.m[1], sb_bal(c.m[1]) - c.amount);
      c.n_write = 1; c.wr[0] = 1;
      return true;
  }
}

TX_HD static inline void sb_emit(SbClient &c, SbOut &o) {
  for (;;) {
    switch (c.step) {
      case 1:  // acquire every lock of the transaction in one phase
        for (uint8_t i = 0; i < c.n_rows; i++) {
          uint8_t ty, tb;
          uint64_t ky;
          sb_lock_row(c, i, &ty, &tb, &ky);
          o.send_new(ty, tb, ky, (uint8_t)(i | TX_FULL));
        }
        c.step = 2;
        return;
      case 2: {
        bool all = true;
        for (uint8_t i = 0; i < c.n_rows; i++) all = all && sb_granted(c.m[i]);
        if (!all || !sb_compute(c)) { c.step = 20; continue; }
        if (c.n_write == 0) { c.step = 6; continue; }  // read-only: straight to release
        for (uint8_t k = 0; k < c.n_write; k++) { c.m[c.wr[k]].ver++; }
        for (uint32_t s = 0; s < 3; s++)
          for (uint8_t k = 0; k < c.n_write; k++) o.send(s, c.wr[k], S_COMMIT_LOG, TX_NO_DST);
        c.step = 3;
        return;
      }
      case 3:  // backups: "+1" copies of every row, then "+2" copies
        for (uint32_t d = 1; d <= 2; d++)
          for (uint8_t k = 0; k < c.n_write; k++) o.send((uint32_t)((c.m[c.wr[k]].key % 3 + d) % 3), c.wr[k], S_COMMIT_BCK, TX_NO_DST);
        c.step = 4;
        return;
      case 4:
        for (uint8_t k = 0; k < c.n_write; k++) o.send((uint32_t)(c.m[c.wr[k]].key % 3), c.wr[k], S_COMMIT_PRIM, TX_NO_DST);
        c.step = 6;
        return;
      case 6:  // release every lock in one phase
        for (uint8_t i = 0; i < c.n_rows; i++)
          o.send((uint32_t)(c.m[i].key % 3), i, (c.m[i].type == S_GRANT_SH) ? S_REL_SH : S_REL_EX, TX_NO_DST);
        c.step = 7;
        return;
      case 7: o.finish(c.txn, true); c.step = 0; return;
      case 20:  // abort: release the granted locks one round trip at a time (:248-279)
        while (c.rel < c.n_rows && !sb_granted(c.m[c.rel])) c.rel++;
        if (c.rel < c.n_rows) {
          o.send((uint32_t)(c.m[c.rel].key % 3), c.rel, (c.m[c.rel].type == S_GRANT_SH) ? S_REL_SH : S_REL_EX, TX_NO_DST);
          c.rel++;
          return;
        }
        o.finish(c.txn, false);
        c.step = 0;
        return;
      default: o.finish(c.txn, false); c.step = 0; return;
    }
  }
}

TX_HD static inline void sb_run(SbClient &c, const TxParams &P, SbOut &o) {
  o.clear();
  for (;;) {
    if (c.step == 0) sb_begin(c, P);
    sb_emit(c, o);
    if (o.n) break;
  }
  c.n_out = o.n;
  for (uint8_t k = 0; k < o.n; k++) { c.out_shard[k] = o.shard[k]; c.out_dst[k] = o.dst[k]; }
}

// One awaited reply of a client, shared by the host and the device driver: `d` = out_dst of the message, `r` = the reply
// in the (in place) batch.  The summary always; the whole reply only where the transaction keeps the row.
template <class Client, class Msg>
TX_HD static inline void tx_consume_one(Client &c, uint8_t d, const Msg *r) {
  if (d == TX_NO_DST) return;
  c.note_reply((uint8_t)(d & TX_DST_MASK), r->type, r->val[0], r->ver);
  if (d & TX_FULL) c.m[d & TX_DST_MASK] = *r;
}

// uniform front for templates
struct TatpTraits {
  typedef TatpClient Client; typedef TatpMsg Msg; typedef TatpOut Out;
  static constexpr uint32_t NMSG = TATP_NMSG, MAXOUT = 6;
  TX_HD static void run(Client &c, const TxParams &P, Out &o) { tatp_run(c, P, o); }
};
struct SbTraits {
  typedef SbClient Client; typedef SbMsg Msg; typedef SbOut Out;
  static constexpr uint32_t NMSG = SB_NMSG, MAXOUT = TX_MAXOUT;
  TX_HD static void run(Client &c, const TxParams &P, Out &o) { sb_run(c, P, o); }
};
