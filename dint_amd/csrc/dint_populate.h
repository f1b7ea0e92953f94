// dint_populate.h -- the reference servers' initial table population, generated on the host and
// streamed to the GPU as bulk-load passes (engine.hip).  Row order per table is the reference's
// insertion order, so chain order inside every bucket comes out identical.
//
//   store      store/udp/tatp.h:44-66      12 rows per subscriber {s_id, sf_type 1..4, start_time 0/8/16}
//   tatp       tatp/udp/tatp.h:283-412     subscriber, secondary subscriber, access info,
//                                          special facility + call forwarding (one PRNG stream for both)
//   smallbank  smallbank/udp/smallbank.h:105-127  savings + checking row per account
// The reference copies partially initialised stack structs into zeroed 40-byte buffers; here every value
// starts as zeros and only the fields the reference assigns are written (SURVEY.md 7 "hard parts").
#pragma once
#include <stdint.h>
#include <string.h>

#include <functional>
#include <vector>

#include "../../include/dint_abi.h"

namespace dint_pop {

struct Lcg {  // fastrand: tatp/udp/tatp.h:32-35
  uint64_t s;
  explicit Lcg(uint64_t seed) : s(seed) {}
  uint32_t next() {
    s = s * 1103515245ull + 12345ull;
    return (uint32_t)(s >> 32);
  }
};

// 3 decimal digits -> 3 BCD nibbles (create_map1000, tatp.h:18-26); 9 digits of s_id -> sub_nbr (tatp.h:132-144)
static inline uint64_t bcd3(uint32_t v) { return ((uint64_t)(v / 100 % 10) << 8) | ((uint64_t)(v / 10 % 10) << 4) | (v % 10); }
static inline uint64_t sub_nbr_of(uint32_t s_id) {
  return bcd3(s_id % 1000) | (bcd3(s_id / 1000 % 1000) << 12) | (bcd3(s_id / 1000000 % 1000) << 24);
}

// a batch of rows for one table, flushed to the sink when full
class RowBatch {
 public:
  using Sink = std::function<int(uint32_t table, const uint64_t *keys, const uint8_t *vals, uint64_t n)>;
  RowBatch(uint32_t table, uint32_t val_size, Sink sink, size_t cap = 65536)
      : table_(table), vs_(val_size), cap_(cap), sink_(std::move(sink)) {
    keys_.reserve(cap);
    vals_.reserve(cap * val_size);
  }
  // returns a zeroed value buffer to fill in
  uint8_t *add(uint64_t key) {
    keys_.push_back(key);
    vals_.resize(vals_.size() + vs_, 0);
    return vals_.data() + vals_.size() - vs_;
  }
  int maybe_flush() { return keys_.size() >= cap_ ? flush() : 0; }
  int flush() {
    int rc = 0;
    if (!keys_.empty()) rc = sink_(table_, keys_.data(), vals_.data(), keys_.size());
    keys_.clear();
    vals_.clear();
    return rc;
  }

 private:
  uint32_t table_, vs_;
  size_t cap_;
  Sink sink_;
  std::vector<uint64_t> keys_;
  std::vector<uint8_t> vals_;
};

// select_between_n_and_m_from(seed, {1,2,3,4}, 1, 4): tatp.h:254-281 -- draw a count, then distinct values by rejection
static inline int pick_types(Lcg &g, uint8_t out[4]) {
  bool used[5] = {false, false, false, false, false};
  const int want = (int)(g.next() % 4u) + 1;
  int got = 0;
  while (got < want) {
    const uint8_t v = (uint8_t)(g.next() % 4u + 1u);
    if (used[v]) continue;
    used[v] = true;
    out[got++] = v;
  }
  return got;
}

static inline int generate(uint32_t workload, uint64_t n, const RowBatch::Sink &sink) {
  int rc = 0;
#define POP_TRY(x) do { rc = (x); if (rc) return rc; } while (0)
  if (workload == DINT_WL_STORE) {
    RowBatch rows(0, 40, sink);
    Lcg g(0xdeadbeef);
    for (uint64_t s = 0; s < n; s++) {
      for (uint64_t sf = 1; sf <= 4; sf++)
        for (uint64_t st = 0; st <= 16; st += 8) {
          uint8_t *v = rows.add(s | (sf << 32) | (st << 40));
          v[0] = (uint8_t)(g.next() % 24u + 1u);  // end_time
          v[1] = 0x5a;                            // numberx[0] = kValMagic
        }
      POP_TRY(rows.maybe_flush());
    }
    return rows.flush();
  }
  if (workload == DINT_WL_SMALLBANK) {
    RowBatch sav(0, 8, sink), chk(1, 8, sink);
    const float bal = 1000000000ull;  // smallbank.h:113,121
    for (uint64_t a = 0; a < n; a++) {
      uint8_t *v = sav.add(a);
      const uint32_t m0 = 97, m1 = 98;  // sb_sav_magic / sb_chk_magic  smallbank.h:71-73
      memcpy(v, &m0, 4);
      memcpy(v + 4, &bal, 4);
      v = chk.add(a);
      memcpy(v, &m1, 4);
      memcpy(v + 4, &bal, 4);
      POP_TRY(sav.maybe_flush());
      POP_TRY(chk.maybe_flush());
    }
    POP_TRY(sav.flush());
    return chk.flush();
  }
  if (workload != DINT_WL_TATP) return DINT_ESTATE;
  {  // SUBSCRIBER  tatp.h:283-309   value = tatp_sub_val_t (tatp.h:160-168)
    RowBatch rows(0, 40, sink);
    Lcg g(0xdeadbeef);
    for (uint64_t s = 0; s < n; s++) {
      uint8_t *v = rows.add(s);
      const uint64_t nbr = sub_nbr_of((uint32_t)s);
      memcpy(v, &nbr, 8);                                         // sub_nbr; bytes 8..14 sub_nbr_unused
      for (int i = 0; i < 5; i++) v[15 + i] = (uint8_t)g.next();  // hex[5]
      for (int i = 0; i < 10; i++) v[20 + i] = (uint8_t)g.next(); // bytes[10]
      const uint16_t bits = (uint16_t)g.next();
      memcpy(v + 30, &bits, 2);
      const uint32_t msc = 97;  // tatp_sub_msc_location_magic
      memcpy(v + 32, &msc, 4);
      const uint32_t vlr = g.next();
      memcpy(v + 36, &vlr, 4);
      POP_TRY(rows.maybe_flush());
    }
    POP_TRY(rows.flush());
  }
  {  // SECONDARY SUBSCRIBER  tatp.h:312-327
    RowBatch rows(1, 40, sink);
    for (uint64_t s = 0; s < n; s++) {
      uint8_t *v = rows.add(sub_nbr_of((uint32_t)s));
      const uint32_t sid = (uint32_t)s;
      memcpy(v, &sid, 4);
      v[4] = 98;  // tatp_sec_sub_magic
      POP_TRY(rows.maybe_flush());
    }
    POP_TRY(rows.flush());
  }
  {  // ACCESS INFO  tatp.h:330-354
    RowBatch rows(2, 40, sink);
    Lcg g(0xdeadbeef);
    for (uint64_t s = 0; s < n; s++) {
      uint8_t ty[4];
      const int k = pick_types(g, ty);
      for (int i = 0; i < k; i++) rows.add(s | ((uint64_t)ty[i] << 32))[0] = 99;  // data1 magic
      POP_TRY(rows.maybe_flush());
    }
    POP_TRY(rows.flush());
  }
  {  // SPECIAL FACILITY + CALL FORWARDING  tatp.h:357-412
    RowBatch sf(3, 40, sink), cf(4, 40, sink);
    Lcg g(0xdeadbeef);
    for (uint64_t s = 0; s < n; s++) {
      uint8_t ty[4];
      const int k = pick_types(g, ty);
      for (int i = 0; i < k; i++) {
        const uint64_t base = s | ((uint64_t)ty[i] << 32);
        uint8_t *v = sf.add(base);
        v[3] = 100;                                 // data_b[0] magic
        v[0] = (g.next() % 100u < 85u) ? 1 : 0;     // is_active
        for (uint64_t st = 0; st <= 16; st += 8) {
          if (g.next() % 2u == 0) continue;         // present with probability 1/2
          uint8_t *w = cf.add(base | (st << 40));
          w[1] = 101;                               // numberx[0] magic
          w[0] = (uint8_t)(g.next() % 24u + 1u);    // end_time
        }
      }
      POP_TRY(sf.maybe_flush());
      POP_TRY(cf.maybe_flush());
    }
    POP_TRY(sf.flush());
    POP_TRY(cf.flush());
  }
#undef POP_TRY
  return 0;
}

}  // namespace dint_pop
