// txn_driver.cc -- closed-loop transaction drivers (include/dint_driver.h): the reference clients'
// transaction state machines, restated as an epoch-synchronous generator.  Plain host C++.
//
// TATP      tatp/caladan/client_udp_shard.cc:177-1117 (7 transactions), mix tatp/caladan/tatp.h:57-63 via
//           CreateWorkgenArr (:63-73), ClientLoop (:1120-1185), keys tatp_nurand (tatp.h:40-43)
// SmallBank smallbank/caladan/client_udp_shard.cc:169-1240 (6 transactions), mix smallbank.h:63-68,
//           account pickers smallbank.h:30-50
// Request messages start zeroed (the reference sends uninitialised stack bytes in the fields a request
// does not set); everything a transaction decides on -- reply types, versions, value bytes -- follows the
// reference client line by line (cited at each step).
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <new>
#include <vector>

#include "../../include/dint_abi.h"
#include "../../include/dint_driver.h"
#include "zipf_table.h"

namespace {

struct Lcg {  // fastrand, tatp/caladan/tatp.h:31-34
  uint64_t s;
  uint32_t next() {
    s = s * 1103515245ull + 12345ull;
    return (uint32_t)(s >> 32);
  }
};

// Zipf(theta) over [0, n): exact inverse CDF over an integer threshold table (zipf_table.h), so that the GPU-resident
// driver (k_txn.hip) draws the same keys from the same random words.
struct Zipf {
  ZipfTable t;
  void init(uint64_t n, double th) { t.init(n, th); }
  uint64_t sample(Lcg &g) const { return zipf_lookup(t.cdf.data(), t.n, g.next()); }
};

// ---- wire messages ----------------------------------------------------------------------------------
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

// 3 decimal digits -> 3 BCD nibbles (create_map1000); s_id -> sub_nbr (tatp_sid_to_sub_nbr)
inline uint64_t bcd3(uint32_t v) { return ((uint64_t)(v / 100 % 10) << 8) | ((uint64_t)(v / 10 % 10) << 4) | (v % 10); }
inline uint64_t sub_nbr_of(uint32_t s) { return bcd3(s % 1000) | (bcd3(s / 1000 % 1000) << 12) | (bcd3(s / 1000000 % 1000) << 24); }

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

// ---- generic epoch machinery -------------------------------------------------------------------------
template <class Msg, int MAXM>
struct ClientBase {
  Lcg rng;
  uint8_t txn = 0, step = 0;     // step 0 = idle
  uint8_t n_out = 0;
  uint8_t out_shard[MAXM * 3];   // a log message goes to all three shards
  uint32_t out_pos[MAXM * 3];
  Msg *out_dst[MAXM * 3];        // where the reply is copied (nullptr = discard)
};

template <class Msg>
struct Batches {
  std::vector<Msg> b[DINT_N_SHARDS];
  void clear() { for (auto &v : b) v.clear(); }
};

}  // namespace

// ======================================================================================================
struct dint_driver {
  dint_driver_config cfg{};
  dint_driver_stats st{};
  bool awaiting = false;
  Zipf zipf;
  virtual ~dint_driver() {}
  virtual int msg_size() const = 0;
  virtual void next(uint32_t counts[DINT_N_SHARDS]) = 0;
  virtual const void *batch(uint32_t s) const = 0;
  virtual void consume(const void *const rep[DINT_N_SHARDS]) = 0;
};

namespace {

// ---- TATP ------------------------------------------------------------------------------------------------
struct TatpDriver final : dint_driver {
  enum Txn : uint8_t { GET_SUB = 0, GET_NEW_DEST = 1, GET_ACCESS = 2, UPD_SUB = 3, UPD_LOC = 4, INS_CF = 5, DEL_CF = 6 };
  struct Client : ClientBase<TatpMsg, 6> {
    uint32_t s_id = 0;
    uint8_t sf_type = 0, start_time = 0, end_time = 0;
    // working messages of the running transaction (names as in the reference functions)
    TatpMsg a_read{}, a_lock{}, b_read{}, b_lock{}, a_ver{}, b_ver{}, tmp[3]{};
  };
  std::vector<Client> cl;
  Batches<TatpMsg> bt;
  uint8_t workgen[100];

  explicit TatpDriver(const dint_driver_config &c) {
    cfg = c;
    cl.resize(c.n_clients);
    for (uint32_t i = 0; i < c.n_clients; i++) cl[i].rng.s = 0xdeadbeefull + c.first_client + i;  // ClientLoop :1122
    // CreateWorkgenArr :63-73 -- note the order: GetSubscriberData, GetAccessData, GetNewDestination, ...
    int k = 0;
    const int freq[7][2] = {{GET_SUB, 35}, {GET_ACCESS, 35}, {GET_NEW_DEST, 10}, {UPD_SUB, 2}, {UPD_LOC, 14}, {INS_CF, 2}, {DEL_CF, 2}};
    for (auto &f : freq)
      for (int j = 0; j < f[1]; j++) workgen[k++] = (uint8_t)f[0];
    if (c.key_dist == 1) zipf.init(c.n_rows, c.zipf_theta);
  }
  int msg_size() const override { return 55; }
  const void *batch(uint32_t s) const override { return bt.b[s].data(); }

  uint32_t pick_sid(Lcg &g) {
    if (cfg.key_dist == 1) return (uint32_t)zipf.sample(g);
    const uint32_t n = (uint32_t)cfg.n_rows;  // tatp_nurand, A = 1048575
    const uint32_t x = g.next() % n, y = g.next() & 1048575u;
    return (x | y) % n;
  }

  static TatpMsg mk(uint8_t type, uint8_t table, uint64_t key) {
    TatpMsg m;
    memset(&m, 0, sizeof m);
    m.type = type; m.table = table; m.key = key;
    return m;
  }
  // queue `m` for shard s; its reply lands in *dst
  void send(Client &c, uint32_t s, const TatpMsg &m, TatpMsg *dst) {
    auto &v = bt.b[s];
    uint8_t j = 0;  // msg->ord = position in this phase's queue for shard s (:376-380)
    for (uint8_t k = 0; k < c.n_out; k++) j += c.out_shard[k] == s;
    c.out_shard[c.n_out] = (uint8_t)s;
    c.out_pos[c.n_out] = (uint32_t)v.size();
    c.out_dst[c.n_out] = dst;
    c.n_out++;
    v.push_back(m);
    v.back().ord = j;
    st.messages++;
  }
  void send_prim(Client &c, TatpMsg &m) { send(c, m.key % 3, m, &m); }
  void send_log3(Client &c, const TatpMsg &m) { for (uint32_t s = 0; s < 3; s++) send(c, s, m, nullptr); }
  void finish(Client &c, bool committed) {
    st.txns++; st.by_type[c.txn]++;
    if (committed) { st.committed++; st.committed_by_type[c.txn]++; }
    c.step = 0;
  }

  // one phase of one client: emit messages (n_out > 0) or finish the transaction
  void run(Client &c) {
    for (;;) {
      if (c.step == 0) begin(c);
      c.n_out = 0;
      emit(c);
      if (c.n_out) return;  // waiting for replies
      // the phase emitted nothing: the transaction finished inside emit() -> start the next one
    }
  }

  void begin(Client &c) {
    c.txn = workgen[c.rng.next() % 100];  // ClientLoop :1144
    c.step = 1;
    Lcg &g = c.rng;
    switch (c.txn) {
      case GET_SUB: c.s_id = pick_sid(g); break;                                                         // :180
      case GET_ACCESS: c.s_id = pick_sid(g); c.sf_type = (uint8_t)((g.next() & 3) + 1); break;           // :308-309 (ai_type)
      case GET_NEW_DEST: case INS_CF:                                                                    // :207-210, :737-740
        c.s_id = pick_sid(g); c.sf_type = (uint8_t)(g.next() % 4 + 1); c.start_time = (uint8_t)(g.next() % 3 * 8);
        c.end_time = (uint8_t)(g.next() % 24);
        break;
      case UPD_SUB: c.s_id = pick_sid(g); c.sf_type = (uint8_t)(g.next() % 4 + 1); break;                // :340-341
      case UPD_LOC: {                                                                                    // :579-580
        c.s_id = pick_sid(g);
        const uint32_t vlr = g.next();
        memcpy(c.tmp[2].val, &vlr, 4);  // parked until the subscriber row is read
        break;
      }
      default: c.s_id = pick_sid(g); c.sf_type = (uint8_t)(g.next() % 4 + 1); c.start_time = (uint8_t)(g.next() % 3 * 8); break;  // DEL_CF :960-962
    }
  }

  uint64_t sf_key(const Client &c) const { return (uint64_t)c.s_id | ((uint64_t)c.sf_type << 32); }
  uint64_t cf_key(const Client &c, uint32_t st_) const { return sf_key(c) | ((uint64_t)st_ << 40); }

  // The step numbers below are this driver's; each case cites the reference lines it restates.
  void emit(Client &c) {
    switch (c.txn) {
      case GET_SUB:  // TxnGetSubscriberData :177-199
        if (c.step == 1) { c.a_read = mk(T_READ, TB_SUB, c.s_id); send_prim(c, c.a_read); c.step = 2; }
        else finish(c, true);
        return;
      case GET_ACCESS:  // TxnGetAccessData :305-331
        if (c.step == 1) { c.a_read = mk(T_READ, TB_AI, sf_key(c)); send_prim(c, c.a_read); c.step = 2; }
        else finish(c, c.a_read.type != T_NOT_EXIST);
        return;
      case GET_NEW_DEST:  // TxnGetNewDestination :202-302
        if (c.step == 1) { c.a_read = mk(T_READ, TB_SF, sf_key(c)); send_prim(c, c.a_read); c.step = 2; return; }
        if (c.step == 2) {
          if (c.a_read.type == T_NOT_EXIST || c.a_read.val[0] == 0) { finish(c, false); return; }  // :239,244 (is_active)
          const uint32_t n = c.start_time / 8u + 1;                                               // cf_to_fetch :212
          for (uint32_t i = 0; i < n; i++) { c.tmp[i] = mk(T_READ, TB_CF, cf_key(c, i * 8)); send_prim(c, c.tmp[i]); }
          c.step = 3;
          return;
        }
        {
          bool ok = false;  // :283-297
          const uint32_t n = c.start_time / 8u + 1;
          for (uint32_t i = 0; i < n; i++)
            if (c.tmp[i].type != T_NOT_EXIST && i * 8 <= c.start_time && c.end_time < c.tmp[i].val[0]) ok = true;
          finish(c, ok);
        }
        return;
      case UPD_SUB: emit_upd_sub(c); return;
      case UPD_LOC: emit_upd_loc(c); return;
      case INS_CF: emit_ins_cf(c); return;
      default: emit_del_cf(c); return;
    }
  }

  // backups of a row whose primary is key % 3: first the "+1" copies of every row, then the "+2" copies
  void send_bck(Client &c, TatpMsg *rows, int n) {
    for (int i = 0; i < n; i++) send(c, (rows[i].key % 3 + 1) % 3, rows[i], nullptr);
    for (int i = 0; i < n; i++) send(c, (rows[i].key % 3 + 2) % 3, rows[i], nullptr);
  }

  void emit_upd_sub(Client &c) {  // TxnUpdateSubscriberData :334-571
    switch (c.step) {
      case 1:  // execute: read + lock both rows :345-396
        c.a_read = mk(T_READ, TB_SUB, c.s_id); c.a_lock = mk(T_ACQ, TB_SUB, c.s_id);
        c.b_read = mk(T_READ, TB_SF, sf_key(c)); c.b_lock = mk(T_ACQ, TB_SF, sf_key(c));
        send_prim(c, c.a_read); send_prim(c, c.a_lock); send_prim(c, c.b_read); send_prim(c, c.b_lock);
        c.step = 2;
        return;
      case 2:
        if (c.b_read.type == T_NOT_EXIST || c.a_lock.type == T_REJECT_LOCK || c.b_lock.type == T_REJECT_LOCK) {  // :400
          c.step = 10;
          emit_upd_sub(c);
          return;
        }
        {  // :425-431
          const uint16_t bits = (uint16_t)c.rng.next();
          memcpy(c.a_read.val + 30, &bits, 2);
          c.b_read.val[2] = (uint8_t)c.rng.next();  // data_a
        }
        c.a_ver = mk(T_READ, TB_SUB, c.s_id); c.b_ver = mk(T_READ, TB_SF, sf_key(c));  // verify :433-447
        send_prim(c, c.a_ver); send_prim(c, c.b_ver);
        c.step = 3;
        return;
      case 3:
        if (c.a_read.ver != c.a_ver.ver || c.b_read.ver != c.b_ver.ver) { c.step = 12; emit_upd_sub(c); return; }  // :470
        c.a_read.ver++; c.b_read.ver++;                                                                           // :487-488
        c.a_read.type = c.b_read.type = T_COMMIT_LOG;
        for (uint32_t s = 0; s < 3; s++) { send(c, s, c.a_read, nullptr); send(c, s, c.b_read, nullptr); }        // :493-501
        c.step = 4;
        return;
      case 4: {
        c.a_read.type = c.b_read.type = T_COMMIT_BCK;  // :521-533
        TatpMsg rows[2] = {c.a_read, c.b_read};
        send_bck(c, rows, 2);
        c.step = 5;
        return;
      }
      case 5:
        c.a_read.type = c.b_read.type = T_COMMIT_PRIM;  // :552-556
        send(c, c.a_read.key % 3, c.a_read, nullptr); send(c, c.b_read.key % 3, c.b_read, nullptr);
        c.step = 6;
        return;
      case 6: finish(c, true); return;
      // abort after a failed execute: release the granted locks one round trip at a time :402-418
      case 10:
        if (c.a_lock.type == T_GRANT_LOCK) { c.a_lock.type = T_ABORT; send_prim(c, c.a_lock); c.step = 11; return; }
        c.step = 11;
        [[fallthrough]];
      case 11:
        if (c.b_lock.type == T_GRANT_LOCK) { c.b_lock.type = T_ABORT; send_prim(c, c.b_lock); c.step = 14; return; }
        finish(c, false);
        return;
      // abort after a failed validation: both locks are held :472-481
      case 12: c.a_lock.type = T_ABORT; send_prim(c, c.a_lock); c.step = 13; return;
      case 13: c.b_lock.type = T_ABORT; send_prim(c, c.b_lock); c.step = 14; return;
      default: finish(c, false); return;
    }
  }

  void emit_upd_loc(Client &c) {  // TxnUpdateLocation :574-728
    switch (c.step) {
      case 1: c.b_read = mk(T_READ, TB_SEC, sub_nbr_of(c.s_id)); send_prim(c, c.b_read); c.step = 2; return;  // :583-592
      case 2:
        c.a_read = mk(T_READ, TB_SUB, c.s_id); c.a_lock = mk(T_ACQ, TB_SUB, c.s_id);  // :605-618
        send_prim(c, c.a_read); send_prim(c, c.a_lock);
        c.step = 3;
        return;
      case 3:
        if (c.a_lock.type == T_REJECT_LOCK) { finish(c, false); return; }  // :645
        memcpy(c.a_read.val + 36, c.tmp[2].val, 4);                        // vlr_location :650
        c.a_ver = mk(T_READ, TB_SUB, c.s_id);                              // verify :653-660
        send_prim(c, c.a_ver);
        c.step = 4;
        return;
      case 4:
        if (c.a_ver.ver != c.a_read.ver) { c.a_lock.type = T_ABORT; send_prim(c, c.a_lock); c.step = 8; return; }  // :667-674
        c.a_read.ver++;
        c.a_read.type = T_COMMIT_LOG;
        send_log3(c, c.a_read);  // :677-684
        c.step = 5;
        return;
      case 5: c.a_read.type = T_COMMIT_BCK; send_bck(c, &c.a_read, 1); c.step = 6; return;                        // :702-708
      case 6: c.a_read.type = T_COMMIT_PRIM; send(c, c.a_read.key % 3, c.a_read, nullptr); c.step = 7; return;    // :722-723
      case 7: finish(c, true); return;
      default: finish(c, false); return;
    }
  }

  void emit_ins_cf(Client &c) {  // TxnInsertCallForwarding :731-951
    switch (c.step) {
      case 1: c.tmp[0] = mk(T_READ, TB_SEC, sub_nbr_of(c.s_id)); send_prim(c, c.tmp[0]); c.step = 2; return;  // :743-752
      case 2: c.b_read = mk(T_READ, TB_SF, sf_key(c)); send_prim(c, c.b_read); c.step = 3; return;            // :761-769
      case 3:
        if (c.b_read.type == T_NOT_EXIST) { finish(c, false); return; }  // :776
        c.a_read = mk(T_READ, TB_CF, cf_key(c, c.start_time)); c.a_lock = mk(T_ACQ, TB_CF, cf_key(c, c.start_time));  // :789-799
        send_prim(c, c.a_read); send_prim(c, c.a_lock);
        c.step = 4;
        return;
      case 4:
        if (c.a_read.type == T_GRANT_READ || c.a_lock.type == T_REJECT_LOCK) {  // the row exists, or no lock :826
          if (c.a_lock.type == T_GRANT_LOCK) { c.a_lock.type = T_ABORT; send_prim(c, c.a_lock); c.step = 9; return; }
          finish(c, false);
          return;
        }
        c.a_read.val[1] = 101;         // numberx[0] magic :842
        c.a_read.val[0] = c.end_time;  // :843
        c.b_ver = mk(T_READ, TB_SF, sf_key(c)); c.a_ver = mk(T_READ, TB_CF, cf_key(c, c.start_time));  // verify :846-859
        send_prim(c, c.b_ver); send_prim(c, c.a_ver);
        c.step = 5;
        return;
      case 5:
        if (c.b_read.ver != c.b_ver.ver || c.a_ver.type == T_GRANT_READ) {  // :884
          c.a_lock.type = T_ABORT; send_prim(c, c.a_lock); c.step = 9; return;
        }
        c.a_read.ver = 0;  // :896
        c.a_read.type = T_COMMIT_LOG;
        send_log3(c, c.a_read);
        c.step = 6;
        return;
      case 6: c.a_read.type = T_INSERT_BCK; send_bck(c, &c.a_read, 1); c.step = 7; return;                      // :921-927
      case 7: c.a_read.type = T_INSERT_PRIM; send(c, c.a_read.key % 3, c.a_read, nullptr); c.step = 8; return;  // :944-945
      case 8: finish(c, true); return;
      default: finish(c, false); return;
    }
  }

  void emit_del_cf(Client &c) {  // TxnDeleteCallForwarding :954-1117
    switch (c.step) {
      case 1: c.tmp[0] = mk(T_READ, TB_SEC, sub_nbr_of(c.s_id)); send_prim(c, c.tmp[0]); c.step = 2; return;  // :965-974
      case 2:
        c.a_read = mk(T_READ, TB_CF, cf_key(c, c.start_time)); c.a_lock = mk(T_ACQ, TB_CF, cf_key(c, c.start_time));  // :983-997
        send_prim(c, c.a_read); send_prim(c, c.a_lock);
        c.step = 3;
        return;
      case 3:
        if (c.a_read.type == T_NOT_EXIST || c.a_lock.type == T_REJECT_LOCK) {  // :1024
          if (c.a_lock.type == T_GRANT_LOCK) { c.a_lock.type = T_ABORT; send_prim(c, c.a_lock); c.step = 8; return; }
          finish(c, false);
          return;
        }
        c.a_ver = mk(T_READ, TB_CF, cf_key(c, c.start_time));  // verify :1040-1046
        send_prim(c, c.a_ver);
        c.step = 4;
        return;
      case 4:
        if (c.a_ver.type == T_NOT_EXIST || c.a_ver.ver != c.a_read.ver) {  // :1052
          c.a_lock.type = T_ABORT; send_prim(c, c.a_lock); c.step = 8; return;
        }
        c.a_read.type = T_DELETE_LOG;  // :1063
        send_log3(c, c.a_read);
        c.step = 5;
        return;
      case 5: c.a_read.type = T_DELETE_BCK; send_bck(c, &c.a_read, 1); c.step = 6; return;                      // :1087-1093
      case 6: c.a_read.type = T_DELETE_PRIM; send(c, c.a_read.key % 3, c.a_read, nullptr); c.step = 7; return;  // :1110-1111
      case 7: finish(c, true); return;
      default: finish(c, false); return;
    }
  }

  void next(uint32_t counts[DINT_N_SHARDS]) override {
    bt.clear();
    for (auto &c : cl) run(c);
    for (uint32_t s = 0; s < DINT_N_SHARDS; s++) counts[s] = (uint32_t)bt.b[s].size();
    st.epochs++;
  }
  void consume(const void *const rep[DINT_N_SHARDS]) override {
    for (auto &c : cl)
      for (uint32_t k = 0; k < c.n_out; k++)
        if (c.out_dst[k]) memcpy(c.out_dst[k], (const TatpMsg *)rep[c.out_shard[k]] + c.out_pos[k], sizeof(TatpMsg));
  }
};

// ---- SmallBank ---------------------------------------------------------------------------------------------
// All six transactions are 2PL: acquire every lock (the grant carries the row), compute, log x3, backup x2,
// primary, release.  RETRY replies never occur against a serial server, so the retry loops of the reference
// (:212-246 and siblings) never spin here; a REJECT aborts by releasing whatever was granted, one at a time.
struct SbDriver final : dint_driver {
  enum Txn : uint8_t { AMALGAMATE = 0, BALANCE = 1, DEPOSIT_CHECKING = 2, SEND_PAYMENT = 3, TRANSACT_SAVING = 4, WRITE_CHECK = 5 };
  struct Client : ClientBase<SbMsg, 9> {
    uint64_t a0 = 0, a1 = 0;
    SbMsg m[3]{};       // the locked rows, in the reference's order
    uint8_t n_rows = 0; // rows of this transaction
    uint8_t n_write = 0, wr[3]{};  // indices into m[] of the rows written back
    uint8_t rel = 0;    // abort: next row to release
    float amount = 0;
  };
  std::vector<Client> cl;
  Batches<SbMsg> bt;
  uint8_t workgen[100];
  uint64_t n_hot;

  explicit SbDriver(const dint_driver_config &c) {
    cfg = c;
    cl.resize(c.n_clients);
    for (uint32_t i = 0; i < c.n_clients; i++) cl[i].rng.s = 0xdeadbeefull + c.first_client + i;
    int k = 0;  // CreateWorkgenArr, smallbank/caladan/client_udp_shard.cc: same construction as tatp, mix smallbank.h:63-68
    const int freq[6][2] = {{AMALGAMATE, 15}, {BALANCE, 15}, {DEPOSIT_CHECKING, 15}, {SEND_PAYMENT, 25}, {TRANSACT_SAVING, 15}, {WRITE_CHECK, 15}};
    for (auto &f : freq)
      for (int j = 0; j < f[1]; j++) workgen[k++] = (uint8_t)f[0];
    n_hot = c.n_rows * 960000ull / 24000000ull;  // kHotAccountNum / kAccountNum, smallbank.h:17-18
    if (n_hot < 2) n_hot = c.n_rows < 2 ? c.n_rows : 2;
    if (c.key_dist == 1) zipf.init(c.n_rows, c.zipf_theta);
  }
  int msg_size() const override { return 23; }
  const void *batch(uint32_t s) const override { return bt.b[s].data(); }

  void get_account(Lcg &g, uint64_t *a) {  // smallbank.h:30-36
    if (cfg.key_dist == 1) { *a = zipf.sample(g); return; }
    if (g.next() % 100 < 90) *a = g.next() % n_hot;
    else *a = g.next() % cfg.n_rows;
  }
  void get_two_accounts(Lcg &g, uint64_t *a, uint64_t *b) {  // smallbank.h:38-50
    if (cfg.key_dist == 1) {
      *a = zipf.sample(g);
      do { *b = zipf.sample(g); } while (*b == *a && cfg.n_rows > 1);
      return;
    }
    const uint64_t n = (g.next() % 100 < 90) ? n_hot : cfg.n_rows;
    *a = g.next() % n;
    *b = g.next() % n;
    while (*b == *a && n > 1) *b = g.next() % n;
  }
  static SbMsg mk(uint8_t type, uint8_t table, uint64_t key) {
    SbMsg m;
    memset(&m, 0, sizeof m);
    m.type = type; m.table = table; m.key = key;
    return m;
  }
  void send(Client &c, uint32_t s, const SbMsg &m, SbMsg *dst) {
    auto &v = bt.b[s];
    uint8_t j = 0;  // msg->ord = position in this phase's queue for shard s
    for (uint8_t k = 0; k < c.n_out; k++) j += c.out_shard[k] == s;
    c.out_shard[c.n_out] = (uint8_t)s;
    c.out_pos[c.n_out] = (uint32_t)v.size();
    c.out_dst[c.n_out] = dst;
    c.n_out++;
    v.push_back(m);
    v.back().ord = j;
    st.messages++;
  }
  void finish(Client &c, bool committed) {
    st.txns++; st.by_type[c.txn]++;
    if (committed) { st.committed++; st.committed_by_type[c.txn]++; }
    c.step = 0;
  }
  static float bal(const SbMsg &m) { float f; memcpy(&f, m.val + 4, 4); return f; }
  static void set_bal(SbMsg &m, float f) { memcpy(m.val + 4, &f, 4); }
  static bool granted(const SbMsg &m) { return m.type == S_GRANT_SH || m.type == S_GRANT_EX; }

  void begin(Client &c) {
    Lcg &g = c.rng;
    c.txn = workgen[g.next() % 100];
    c.step = 1;
    c.rel = 0;
    c.n_write = 0;
    // lock set of each transaction, in the order the reference pushes the messages
    switch (c.txn) {
      case AMALGAMATE:  // TxnAmalgamate :169-438: X(sav a0), X(chk a0), X(chk a1)
        get_two_accounts(g, &c.a0, &c.a1);
        c.m[0] = mk(S_ACQ_EX, 0, c.a0); c.m[1] = mk(S_ACQ_EX, 1, c.a0); c.m[2] = mk(S_ACQ_EX, 1, c.a1);
        c.n_rows = 3;
        break;
      case BALANCE:  // TxnBalance :441-578: S(sav), S(chk); read only
        get_account(g, &c.a0);
        c.m[0] = mk(S_ACQ_SH, 0, c.a0); c.m[1] = mk(S_ACQ_SH, 1, c.a0);
        c.n_rows = 2;
        break;
      case DEPOSIT_CHECKING:  // TxnDepositChecking :581-684: X(chk); bal += 1.3
        get_account(g, &c.a0);
        c.amount = 1.3f;
        c.m[0] = mk(S_ACQ_EX, 1, c.a0);
        c.n_rows = 1;
        break;
      case SEND_PAYMENT:  // TxnSendPayment :687-932: X(chk a0), X(chk a1); move 5.0 if funds suffice
        get_two_accounts(g, &c.a0, &c.a1);
        c.amount = 5.0f;
        c.m[0] = mk(S_ACQ_EX, 1, c.a0); c.m[1] = mk(S_ACQ_EX, 1, c.a1);
        c.n_rows = 2;
        break;
      case TRANSACT_SAVING:  // TxnTransactSaving :935-1038: X(sav); bal += 20.20
        get_account(g, &c.a0);
        c.amount = 20.20f;
        c.m[0] = mk(S_ACQ_EX, 0, c.a0);
        c.n_rows = 1;
        break;
      default:  // TxnWriteCheck :1041-1240: S(sav), X(chk); chk -= 5 (+1 penalty when overdrawn)
        get_account(g, &c.a0);
        c.amount = 5.0f;
        c.m[0] = mk(S_ACQ_SH, 0, c.a0); c.m[1] = mk(S_ACQ_EX, 1, c.a0);
        c.n_rows = 2;
        break;
    }
  }

  // compute phase: returns false when the transaction aborts by its own logic after locking
  bool compute(Client &c) {
    switch (c.txn) {
      case AMALGAMATE:  // :296-299
        set_bal(c.m[2], bal(c.m[2]) + bal(c.m[0]) + bal(c.m[1]));
        set_bal(c.m[0], 0); set_bal(c.m[1], 0);
        c.n_write = 3; c.wr[0] = 0; c.wr[1] = 1; c.wr[2] = 2;
        return true;
      case BALANCE: c.n_write = 0; return true;
      case DEPOSIT_CHECKING: set_bal(c.m[0], bal(c.m[0]) + c.amount); c.n_write = 1; c.wr[0] = 0; return true;
      case SEND_PAYMENT:
        if (bal(c.m[0]) < c.amount) return false;  // insufficient funds: release and abort
        set_bal(c.m[0], bal(c.m[0]) - c.amount); set_bal(c.m[1], bal(c.m[1]) + c.amount);
        c.n_write = 2; c.wr[0] = 0; c.wr[1] = 1;
        return true;
      case TRANSACT_SAVING: set_bal(c.m[0], bal(c.m[0]) + c.amount); c.n_write = 1; c.wr[0] = 0; return true;
      default:
        if (bal(c.m[0]) + bal(c.m[1]) < c.amount) set_bal(c.m[1], bal(c.m[1]) - (c.amount + 1));
        else set_bal(c.m[1], bal(c.m[1]) - c.amount);
        c.n_write = 1; c.wr[0] = 1;
        return true;
    }
  }

  void emit(Client &c) {
    switch (c.step) {
      case 1:  // acquire every lock of the transaction in one phase
        for (uint8_t i = 0; i < c.n_rows; i++) send(c, c.m[i].key % 3, c.m[i], &c.m[i]);
        c.step = 2;
        return;
      case 2: {
        bool all = true;
        for (uint8_t i = 0; i < c.n_rows; i++) all = all && granted(c.m[i]);
        if (!all || !compute(c)) { c.step = 20; emit(c); return; }
        if (c.n_write == 0) { c.step = 6; emit(c); return; }  // read-only: straight to release
        for (uint8_t k = 0; k < c.n_write; k++) { c.m[c.wr[k]].ver++; }
        for (uint32_t s = 0; s < 3; s++)
          for (uint8_t k = 0; k < c.n_write; k++) { SbMsg t = c.m[c.wr[k]]; t.type = S_COMMIT_LOG; send(c, s, t, nullptr); }
        c.step = 3;
        return;
      }
      case 3:  // backups: "+1" copies of every row, then "+2" copies
        for (uint32_t d = 1; d <= 2; d++)
          for (uint8_t k = 0; k < c.n_write; k++) { SbMsg t = c.m[c.wr[k]]; t.type = S_COMMIT_BCK; send(c, (t.key % 3 + d) % 3, t, nullptr); }
        c.step = 4;
        return;
      case 4:
        for (uint8_t k = 0; k < c.n_write; k++) { SbMsg t = c.m[c.wr[k]]; t.type = S_COMMIT_PRIM; send(c, t.key % 3, t, nullptr); }
        c.step = 6;
        return;
      case 6:  // release every lock in one phase
        for (uint8_t i = 0; i < c.n_rows; i++) {
          SbMsg t = c.m[i];
          t.type = (c.m[i].type == S_GRANT_SH) ? S_REL_SH : S_REL_EX;
          send(c, t.key % 3, t, nullptr);
        }
        c.step = 7;
        return;
      case 7: finish(c, true); return;
      case 20:  // abort: release the granted locks one round trip at a time (:248-279)
        while (c.rel < c.n_rows && !granted(c.m[c.rel])) c.rel++;
        if (c.rel < c.n_rows) {
          SbMsg t = c.m[c.rel];
          t.type = (c.m[c.rel].type == S_GRANT_SH) ? S_REL_SH : S_REL_EX;
          send(c, t.key % 3, t, nullptr);
          c.rel++;
          return;
        }
        finish(c, false);
        return;
      default: finish(c, false); return;
    }
  }
  void run(Client &c) {
    for (;;) {
      if (c.step == 0) begin(c);
      c.n_out = 0;
      emit(c);
      if (c.n_out) return;
    }
  }
  void next(uint32_t counts[DINT_N_SHARDS]) override {
    bt.clear();
    for (auto &c : cl) run(c);
    for (uint32_t s = 0; s < DINT_N_SHARDS; s++) counts[s] = (uint32_t)bt.b[s].size();
    st.epochs++;
  }
  void consume(const void *const rep[DINT_N_SHARDS]) override {
    for (auto &c : cl)
      for (uint32_t k = 0; k < c.n_out; k++)
        if (c.out_dst[k]) memcpy(c.out_dst[k], (const SbMsg *)rep[c.out_shard[k]] + c.out_pos[k], sizeof(SbMsg));
  }
};

}  // namespace

extern "C" {

int dint_driver_create(const dint_driver_config *cfg, dint_driver_t **out) {
  if (!cfg || !out || cfg->n_clients == 0 || cfg->n_rows == 0) return DINT_EINVAL;
  if (cfg->key_dist > 1 || (cfg->key_dist == 1 && !(cfg->zipf_theta > 0 && cfg->zipf_theta < 1))) return DINT_EINVAL;
  try {
    if (cfg->workload == DINT_WL_TATP) *out = new TatpDriver(*cfg);
    else if (cfg->workload == DINT_WL_SMALLBANK) *out = new SbDriver(*cfg);
    else return DINT_EINVAL;
  } catch (const std::bad_alloc &) {
    return DINT_ENOMEM;
  }
  return 0;
}
void dint_driver_destroy(dint_driver_t *d) { delete d; }
int dint_driver_msg_size(const dint_driver_t *d) { return d ? d->msg_size() : DINT_EINVAL; }
int dint_driver_next(dint_driver_t *d, uint32_t counts[DINT_N_SHARDS]) {
  if (!d || !counts) return DINT_EINVAL;
  if (d->awaiting) return DINT_ESTATE;
  d->next(counts);
  d->awaiting = true;
  return 0;
}
const void *dint_driver_batch(dint_driver_t *d, uint32_t shard) { return (d && shard < DINT_N_SHARDS) ? d->batch(shard) : nullptr; }
int dint_driver_consume(dint_driver_t *d, const void *const replies[DINT_N_SHARDS]) {
  if (!d || !replies) return DINT_EINVAL;
  if (!d->awaiting) return DINT_ESTATE;
  d->consume(replies);
  d->awaiting = false;
  return 0;
}
int dint_driver_get_stats(const dint_driver_t *d, dint_driver_stats *out) {
  if (!d || !out) return DINT_EINVAL;
  *out = d->st;
  return 0;
}
}
