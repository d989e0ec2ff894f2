// Constraint system, random oracle and gadgets of the Nova layer.  Specification: oracle/nova.py, oracle/poseidon.py
// (see r1cs.hpp).  Host code only: O(10^4) sequential field operations per step, no kernel work.
#include "r1cs.hpp"

#include <cstdio>
#include <condition_variable>
#include <stdexcept>
#include <sched.h>
#include <cstdlib>
#include <mutex>
#include <array>
#include <atomic>
#include <chrono>
#include <thread>

namespace vdfnova {

// =============================================================================================================
// CS
// =============================================================================================================
namespace {
struct BufferPool {
  std::mutex mu;
  std::vector<std::vector<Fe>> free_;
};
BufferPool& buffer_pool() { static BufferPool* p = new BufferPool(); return *p; }   // never destroyed (helper threads outlive main)
constexpr size_t POOL_RESERVE = 1 << 14, POOL_KEEP = 16;
}  // namespace
std::vector<Fe> witness_buffer_take() {
  BufferPool& p = buffer_pool();
  {
    std::lock_guard<std::mutex> l(p.mu);
    if (!p.free_.empty()) { std::vector<Fe> v = std::move(p.free_.back()); p.free_.pop_back(); return v; }
  }
  std::vector<Fe> v;
  v.reserve(POOL_RESERVE);
  return v;
}
void witness_buffer_give(std::vector<Fe>&& v) {
  if (v.capacity() < POOL_RESERVE) return;
  v.clear();
  BufferPool& p = buffer_pool();
  std::lock_guard<std::mutex> l(p.mu);
  if (p.free_.size() < POOL_KEEP) p.free_.push_back(std::move(v));
}

CS::CS(int f, bool shape_mode, const RoInstance* r) : ro(r ? r : ro_default()), field_id(f), F(field(f)), shape(shape_mode) { if (!shape) W = witness_buffer_take(); }
CS::~CS() { if (!shape) witness_buffer_give(std::move(W)); }

Num CS::constant(const Fe& k) const {
  Num n;
  n.v = k;
  if (shape && !k.is_zero()) n.lc.push_back(Term{KEY_ONE, k});
  return n;
}

static LC lc_combine(const LC& a, const LC& b, bool negate_b, const Field& F) {
  LC out;
  out.reserve(a.size() + b.size());
  size_t i = 0, j = 0;
  while (i < a.size() || j < b.size()) {
    if (j == b.size() || (i < a.size() && a[i].key < b[j].key)) out.push_back(a[i++]);
    else if (i == a.size() || b[j].key < a[i].key) {
      Term t = b[j++];
      if (negate_b) t.c = neg(t.c, F);
      out.push_back(t);
    } else {
      const Fe c = negate_b ? vdfhost::sub(a[i].c, b[j].c, F) : vdfhost::add(a[i].c, b[j].c, F);
      if (!c.is_zero()) out.push_back(Term{a[i].key, c});
      ++i; ++j;
    }
  }
  return out;
}

Num CS::add(const Num& a, const Num& b) const {
  Num n;
  n.v = vdfhost::add(a.v, b.v, F);
  if (shape) n.lc = lc_combine(a.lc, b.lc, false, F);
  return n;
}
Num CS::sub(const Num& a, const Num& b) const {
  Num n;
  n.v = vdfhost::sub(a.v, b.v, F);
  if (shape) n.lc = lc_combine(a.lc, b.lc, true, F);
  return n;
}
Num CS::scale(const Num& a, const Fe& k) const {
  Num n;
  n.v = vdfhost::mul(a.v, k, F);
  if (shape && !k.is_zero()) {
    n.lc.reserve(a.lc.size());
    for (const Term& t : a.lc) n.lc.push_back(Term{t.key, vdfhost::mul(t.c, k, F)});
  }
  return n;
}
Num CS::scale_small(const Num& a, unsigned k) const {
  Num n;
  Fe acc = vdfhost::zero(), base = a.v;
  for (unsigned e = k; e; e >>= 1) {
    if (e & 1) acc = vdfhost::add(acc, base, F);
    base = vdfhost::add(base, base, F);
  }
  n.v = acc;
  if (shape && k) {
    const Fe kf = from_u64(k, F);
    n.lc.reserve(a.lc.size());
    for (const Term& t : a.lc) n.lc.push_back(Term{t.key, vdfhost::mul(t.c, kf, F)});
  }
  return n;
}
Num CS::alloc(const Fe& v) {
  Num n;
  n.v = v;
  if (shape) n.lc.push_back(Term{(uint32_t)W.size(), one(F)});
  W.push_back(v);
  return n;
}
Num CS::alloc_io(const Fe& v) {
  Num n;
  n.v = v;
  if (shape) n.lc.push_back(Term{KEY_ONE + 1 + (uint32_t)X.size(), one(F)});
  X.push_back(v);
  return n;
}
void CS::skip(size_t n, size_t cons) {
  dev_begin = W.size();
  dev_len = n;
  rows += cons;
}
void CS::enforce(const Num& a, const Num& b, const Num& c) {
  if (shape) cons_.push_back(Row{a.lc, b.lc, c.lc});
  ++rows;
}
Num CS::mul(const Num& a, const Num& b) {
  Num c = alloc(vdfhost::mul(a.v, b.v, F));
  enforce(a, b, c);
  return c;
}
void CS::enforce_equal(const Num& a, const Num& b) { enforce(sub(a, b), constant(one(F)), zero_num()); }

Fe CS::take_inverse(const Fe& den) {
  if (inv_pos < inv_queue.size()) {
    const Fe& c = inv_queue[inv_pos++];
    if (den.is_zero() ? c.is_zero() : vdfhost::mul(den, c, F) == one(F)) return c;
    ++inv_misses;
  }
  return inverse(den, F);
}
Num CS::alloc_inverse_later(const Fe& a) {
  if (shape || a.is_zero()) return alloc(vdfhost::zero());     // shape mode: values are never read
  later_.emplace_back(W.size(), a);
  return alloc(vdfhost::zero());
}
void CS::resolve() {
  if (later_.empty()) return;
  std::vector<Fe> v(later_.size());
  for (size_t k = 0; k < v.size(); ++k) v[k] = later_[k].second;
  batch_inverse(v.data(), v.size(), F);
  for (size_t k = 0; k < v.size(); ++k) W[later_[k].first] = v[k];
  later_.clear();
}

void batch_inverse(Fe* v, size_t n, const Field& F) {
  static thread_local std::vector<Fe> pre;
  pre.resize(n);
  Fe run = one(F);
  for (size_t i = 0; i < n; ++i) {
    pre[i] = run;
    if (!v[i].is_zero()) run = vdfhost::mul(run, v[i], F);
  }
  Fe inv = inverse(run, F);
  for (size_t i = n; i-- > 0;) {
    if (v[i].is_zero()) continue;
    const Fe x = vdfhost::mul(inv, pre[i], F);
    inv = vdfhost::mul(inv, v[i], F);
    v[i] = x;
  }
}

void CS::finish(Coo out[3]) const {
  const uint32_t nv = (uint32_t)W.size();
  auto col = [&](uint32_t key) { return key < KEY_ONE ? key : nv + (key - KEY_ONE); };
  for (size_t r = 0; r < cons_.size(); ++r) {
    const LC* l[3] = {&cons_[r].a, &cons_[r].b, &cons_[r].c};
    for (int k = 0; k < 3; ++k)
      for (const Term& t : *l[k]) { out[k].rows.push_back((uint32_t)r); out[k].cols.push_back(col(t.key)); out[k].vals.push_back(t.c); }
  }
}

// =============================================================================================================
// Random oracle
// =============================================================================================================
static const int M4[4][4] = {{5, 7, 1, 3}, {4, 6, 1, 1}, {1, 3, 5, 7}, {1, 1, 4, 6}};

static RoConstants make_ro(int f) {
  const Field& F = field(f);
  RoConstants rc;
  const unsigned mu[4] = {2, 3, 6, 8};
  for (int i = 0; i < 4; ++i) rc.mu_minus_1[i] = mu[i] - 1;
  Shake256 h;
  h.absorb("vdf-poseidon2-v1", 16);
  const uint8_t hdr[4] = {(uint8_t)f, RO_T, RO_RF, RO_RP};
  h.absorb(hdr, 4);
  Fe r_mont;                                           // Montgomery form of 2^256 mod m
  { Fe o; memcpy(o.l, F.one, 32); r_mont = to_mont(o, F); }
  auto next = [&]() {
    uint64_t w[8];
    h.squeeze(w, 64);
    return vdfhost::add(int_to_fe(w, F), vdfhost::mul(int_to_fe(w + 4, F), r_mont, F), F);
  };
  for (int r = 0; r < RO_RF + RO_RP; ++r) {
    if (r < RO_RF / 2) for (int i = 0; i < RO_T; ++i) rc.ext[r][i] = next();
    else if (r < RO_RF / 2 + RO_RP) rc.in[r - RO_RF / 2] = next();
    else for (int i = 0; i < RO_T; ++i) rc.ext[r - RO_RP][i] = next();
  }
  return rc;
}
const RoConstants& ro_constants(int f) {
  static const RoConstants fp = make_ro(VDF_FIELD_FP), fq = make_ro(VDF_FIELD_FQ);
  return f == VDF_FIELD_FP ? fp : fq;
}

// ---- the parameter block: instances ---------------------------------------------------------------------------
// family 1 [UPSTREAM-RECALL, the Poseidon paper's generate_parameters_grain; oracle/poseidon.py classic_constants is the
// restatement this is tested against]: round constants from an 80-bit Grain LFSR seeded with the instance's own numbers,
// Cauchy MDS matrix 1 / (i + t + j).
static void make_classic(RoInstance& R, int f) {
  const Field& F = field(f);
  const int t = R.spec.width, rf = R.spec.full_rounds, rp = R.spec.partial_rounds, n = 255;
  uint8_t st[80];
  int k = 0;
  auto put = [&](unsigned v, int w) { for (int b = w - 1; b >= 0; --b) st[k++] = (uint8_t)((v >> b) & 1u); };
  put(1, 2); put(0, 4); put((unsigned)n, 12); put((unsigned)t, 12); put((unsigned)rf, 10); put((unsigned)rp, 10);
  while (k < 80) st[k++] = 1;
  int head = 0;                                        // st[(head + i) % 80] = b_i of the paper
  auto step = [&]() -> unsigned {
    auto at = [&](int i) { return st[(head + i) % 80]; };
    const uint8_t b = at(62) ^ at(51) ^ at(38) ^ at(23) ^ at(13) ^ at(0);
    st[head] = b;
    head = (head + 1) % 80;
    return b;
  };
  for (int i = 0; i < 160; ++i) step();
  auto next_bit = [&]() -> unsigned { for (;;) { const unsigned a = step(), b = step(); if (a) return b; } };
  auto next_fe = [&]() -> Fe {
    for (;;) {
      uint64_t v[4] = {0, 0, 0, 0};
      for (int i = 0; i < n; ++i) {                    // most significant bit first
        v[3] = (v[3] << 1) | (v[2] >> 63); v[2] = (v[2] << 1) | (v[1] >> 63); v[1] = (v[1] << 1) | (v[0] >> 63);
        v[0] = (v[0] << 1) | next_bit();
      }
      if (!geq(v, F.m)) { Fe c; memcpy(c.l, v, 32); return to_mont(c, F); }
    }
  };
  R.rc[f].resize((size_t)(rf + rp) * t);
  for (Fe& c : R.rc[f]) c = next_fe();
  R.mds[f].resize((size_t)t * t);
  for (int i = 0; i < t; ++i)
    for (int j = 0; j < t; ++j) R.mds[f][(size_t)i * t + j] = inverse(from_u64((uint64_t)(i + t + j), F), F);
}
const RoInstance* ro_default() {
  static const RoInstance d = [] {
    RoInstance r;
    r.rate = 3; r.is_default = true;
    const char* l = "vdf-poseidon2-v1";
    r.label.assign(l, l + 16);
    return r;
  }();
  return &d;
}
const RoInstance* ro_instance(const RoSpec& sp) {
  if (sp == RoSpec()) return ro_default();
  // what this build supports: x^5, the protocol's 128 / 250-bit truncations; family 0 is the default block only
  if (sp.alpha != 5 || sp.challenge_bits != CHAL_BITS || sp.hash_bits != HASH_BITS || sp.family != 1) return nullptr;
  if (sp.width < 2 || sp.width > RO_MAX_T || sp.full_rounds < 2 || sp.full_rounds > 16 || (sp.full_rounds & 1) || sp.partial_rounds < 0 ||
      sp.partial_rounds > 128)
    return nullptr;
  static std::mutex mu;
  static std::vector<std::unique_ptr<RoInstance>> made;
  std::lock_guard<std::mutex> lock(mu);
  for (const auto& r : made) if (r->spec == sp) return r.get();
  std::unique_ptr<RoInstance> r(new RoInstance());
  r->spec = sp; r->rate = sp.width - 1; r->is_default = false;
  const char* l = "vdf-ro-block-v1:";
  r->label.assign(l, l + 16);
  for (int v : {sp.family, sp.width, sp.full_rounds, sp.partial_rounds, sp.alpha, sp.challenge_bits, sp.hash_bits}) r->label.push_back((uint8_t)v);
  make_classic(*r, VDF_FIELD_FP);
  make_classic(*r, VDF_FIELD_FQ);
  made.push_back(std::move(r));
  return made.back().get();
}

static inline Fe dbl(const Fe& a, const Field& F) { return vdfhost::add(a, a, F); }
// M4 s with 14 additions (the factorisation of the Poseidon2 paper, section 5.1)
static inline void ext_layer(Fe s[4], const Field& F) {
  const Fe t0 = vdfhost::add(s[0], s[1], F), t1 = vdfhost::add(s[2], s[3], F);
  const Fe t2 = vdfhost::add(dbl(s[1], F), t1, F), t3 = vdfhost::add(dbl(s[3], F), t0, F);
  const Fe t4 = vdfhost::add(dbl(dbl(t1, F), F), t3, F), t5 = vdfhost::add(dbl(dbl(t0, F), F), t2, F);
  s[0] = vdfhost::add(t3, t5, F); s[1] = t5; s[2] = vdfhost::add(t2, t4, F); s[3] = t4;
}
static_assert(true, "M4 = [[5,7,1,3],[4,6,1,1],[1,3,5,7],[1,1,4,6]]");
// ones off the diagonal, (2, 3, 6, 8) on it: s_i <- sum + (mu_i - 1) s_i with (mu - 1) = (1, 2, 5, 7)
static inline void int_layer(Fe s[4], const Field& F) {
  const Fe tot = vdfhost::add(vdfhost::add(s[0], s[1], F), vdfhost::add(s[2], s[3], F), F);
  const Fe s2x4 = dbl(dbl(s[2], F), F), s3x8 = dbl(dbl(dbl(s[3], F), F), F);
  s[0] = vdfhost::add(tot, s[0], F);
  s[1] = vdfhost::add(tot, dbl(s[1], F), F);
  s[2] = vdfhost::add(tot, vdfhost::add(s2x4, s[2], F), F);
  s[3] = vdfhost::add(tot, vdfhost::sub(s3x8, s[3], F), F);
}
static inline Fe pow5(const Fe& x, const Field& F) { const Fe x2 = sqr(x, F); return vdfhost::mul(sqr(x2, F), x, F); }

// the original Poseidon permutation (family 1): add constants, x^5 on all lanes / on lane 0, dense matrix
static void classic_permute(Fe* s, int f, const RoInstance& R) {
  const Field& F = field(f);
  const int t = R.spec.width, half = R.spec.full_rounds / 2, rounds = R.spec.full_rounds + R.spec.partial_rounds;
  const Fe* rc = R.rc[f].data();
  const Fe* M = R.mds[f].data();
  Fe tmp[RO_MAX_T];
  for (int r = 0; r < rounds; ++r) {
    for (int i = 0; i < t; ++i) s[i] = vdfhost::add(s[i], rc[(size_t)r * t + i], F);
    if (r < half || r >= half + R.spec.partial_rounds) for (int i = 0; i < t; ++i) s[i] = pow5(s[i], F);
    else s[0] = pow5(s[0], F);
    for (int i = 0; i < t; ++i) {
      Fe acc = vdfhost::mul(M[(size_t)i * t], s[0], F);
      for (int j = 1; j < t; ++j) acc = vdfhost::add(acc, vdfhost::mul(M[(size_t)i * t + j], s[j], F), F);
      tmp[i] = acc;
    }
    for (int i = 0; i < t; ++i) s[i] = tmp[i];
  }
}

void ro_permute(Fe* s, int f, const RoInstance* ro) {
  if (ro && !ro->is_default) { classic_permute(s, f, *ro); return; }
  const Field& F = field(f);
  const RoConstants& rc = ro_constants(f);
  ext_layer(s, F);
  for (int r = 0; r < RO_RF + RO_RP; ++r) {
    if (r < RO_RF / 2 || r >= RO_RF / 2 + RO_RP) {
      const Fe* k = rc.ext[r < RO_RF / 2 ? r : r - RO_RP];
      for (int i = 0; i < 4; ++i) s[i] = pow5(vdfhost::add(s[i], k[i], F), F);
      ext_layer(s, F);
    } else {
      s[0] = pow5(vdfhost::add(s[0], rc.in[r - RO_RF / 2], F), F);
      int_layer(s, F);
    }
  }
}

Fe ro_hash(int f, uint64_t tag, const Fe* xs, size_t n, const RoInstance* ro) {
  const Field& F = field(f);
  if (!ro) ro = ro_default();
  const size_t rate = (size_t)ro->rate;
  Fe s[RO_MAX_T];
  s[0] = from_u64(tag + ((uint64_t)n << 32), F);
  for (int i = 1; i < RO_MAX_T; ++i) s[i] = vdfhost::zero();
  for (size_t k = 0; k < n; k += rate) {
    for (size_t j = 0; j < rate && k + j < n; ++j) s[1 + j] = vdfhost::add(s[1 + j], xs[k + j], F);
    ro_permute(s, f, ro);
  }
  return s[1];
}

// =============================================================================================================
// Gadgets
// =============================================================================================================
Num is_zero(CS& cs, const Num& a) {
  const Field& F = cs.F;
  Num z = cs.alloc(a.v.is_zero() ? one(F) : vdfhost::zero());
  Num inv = cs.alloc_inverse_later(a.v);                // 0 for 0; filled by cs.resolve()
  cs.enforce(a, inv, cs.sub(cs.constant(one(F)), z));
  cs.enforce(a, z, cs.zero_num());
  return z;
}

Num select(CS& cs, const Num& cond, const Num& a, const Num& b) {
  Num out = cs.alloc(cond.v.is_zero() ? b.v : a.v);
  cs.enforce(cond, cs.sub(a, b), cs.sub(out, b));
  return out;
}

std::vector<Num> alloc_bits(CS& cs, const uint64_t v[4], int n) {
  const Field& F = cs.F;
  const Fe o = one(F), z = vdfhost::zero();
  const Num one_n = cs.constant(o);
  std::vector<Num> bits;
  if (!cs.shape) {
    bits.resize(n);
    for (int k = 0; k < n; ++k) {
      bits[k].v = ((v[k / 64] >> (k % 64)) & 1) ? o : z;
      cs.W.push_back(bits[k].v);
    }
    cs.rows += n;
    return bits;
  }
  bits.reserve(n);
  for (int k = 0; k < n; ++k) {
    Num b = cs.alloc(((v[k / 64] >> (k % 64)) & 1) ? o : z);
    cs.enforce(b, cs.sub(one_n, b), cs.zero_num());
    bits.push_back(std::move(b));
  }
  return bits;
}

static const Fe* pow2_table(const Field& F) {          // 2^k in Montgomery form, k < 256
  static Fe tp[256], tq[256];
  static std::once_flag once;
  std::call_once(once, [] {
    for (int f = 0; f < 2; ++f) {
      const Field& G = field(f);
      Fe* t = f == VDF_FIELD_FP ? tp : tq;
      t[0] = one(G);
      for (int k = 1; k < 256; ++k) t[k] = vdfhost::add(t[k - 1], t[k - 1], G);
    }
  });
  return &F == &field(VDF_FIELD_FP) ? tp : tq;
}

// bits are single variables (0 / 1) with ascending keys, as alloc_bits makes them
Num pack(const CS& cs, const Num* bits, size_t n) {
  const Field& F = cs.F;
  const Fe* p2 = pow2_table(F);
  uint64_t v[4] = {0, 0, 0, 0};
  for (size_t k = 0; k < n; ++k) if (!bits[k].v.is_zero()) v[k / 64] |= 1ull << (k % 64);
  Num out;
  out.v = int_to_fe(v, F);
  if (cs.shape) {
    out.lc.reserve(n);
    for (size_t k = 0; k < n; ++k) out.lc.push_back(Term{bits[k].lc[0].key, p2[k]});
  }
  return out;
}

// 1 / k in the field for k = 0 .. 128 (0 for 0), made once per field with one batched inversion
static const Fe& small_inverse(int field_id, uint64_t k) {
  static const std::vector<Fe>* tables[2] = {nullptr, nullptr};
  static std::once_flag once[2];
  const int slot = field_id == VDF_FIELD_FP ? 0 : 1;
  std::call_once(once[slot], [&] {
    const Field& F = field(field_id);
    std::vector<Fe>* t = new std::vector<Fe>(129);
    for (uint64_t v = 0; v <= 128; ++v) (*t)[v] = from_u64(v, F);
    batch_inverse(t->data(), t->size(), F);
    tables[slot] = t;
  });
  return (*tables[slot])[k];
}

std::vector<Num> strict_bits(CS& cs, const Num& a) {
  const Field& F = cs.F;
  uint64_t av[4];
  fe_to_int(a.v, F, av);
  std::vector<Num> bits = alloc_bits(cs, av, 255);
  cs.enforce_equal(pack(cs, bits.data(), 255), a);
  Num mid = cs.zero_num();
  {                                                     // sum of bits 126..253
    uint64_t cnt = 0;
    for (int k = 126; k < 254; ++k) cnt += bits[k].v.is_zero() ? 0 : 1;
    mid.v = from_u64(cnt, F);
    if (cs.shape) for (int k = 126; k < 254; ++k) mid.lc.push_back(Term{bits[k].lc[0].key, one(F)});
  }
  Num mz;
  if (cs.shape) mz = is_zero(cs, mid);
  else {
    // is_zero's two variables; the inverse of a count of at most 128 comes from a table, not from a field inversion
    uint64_t cnt = 0;
    for (int k = 126; k < 254; ++k) cnt += bits[k].v.is_zero() ? 0 : 1;
    mz = cs.alloc(cnt ? vdfhost::zero() : one(F));
    cs.alloc(small_inverse(cs.field_id, cnt));
    cs.rows += 2;
  }
  const Num low = pack(cs, bits.data(), 126);
  // c = m - 2^254; v = low + (2^126 - c) < 2^127, bit 126 clear iff low < c
  uint64_t k126[4] = {0, 1ull << 62, 0, 0}, c[4] = {F.m[0], F.m[1], F.m[2], F.m[3] - (1ull << 62)};
  sub4(k126, c);
  const Num v = cs.add(low, cs.constant(int_to_fe(k126, F)));
  uint64_t vv[4];
  fe_to_int(v.v, F, vv);
  const std::vector<Num> d = alloc_bits(cs, vv, 127);
  cs.enforce_equal(pack(cs, d.data(), 127), v);
  const Num one_n = cs.constant(one(F));
  const Num ok = cs.mul(mz, cs.sub(one_n, d[126]));
  cs.enforce(bits[254], cs.sub(one_n, ok), cs.zero_num());
  return bits;
}

static void ext_layer_num(CS& cs, std::vector<Num>& s) {
  std::vector<Num> o(4);
  for (int i = 0; i < 4; ++i) {
    Num acc = cs.scale_small(s[0], (unsigned)M4[i][0]);
    for (int j = 1; j < 4; ++j) acc = cs.add(acc, cs.scale_small(s[j], (unsigned)M4[i][j]));
    o[i] = std::move(acc);
  }
  s.swap(o);
}
static Num sbox(CS& cs, const Num& x) {
  const Num x2 = cs.mul(x, x);
  const Num x4 = cs.mul(x2, x2);
  return cs.mul(x4, x);
}
// witness mode: the permutation on bare field elements; an S-box leaves x^2, x^4, x^5 in W (three constraints)
static void poseidon_permute_witness(CS& cs, Fe* s) {
  const Field& F = cs.F;
  auto sbox_w = [&](const Fe& x) {
    const Fe x2 = sqr(x, F), x4 = sqr(x2, F), x5 = vdfhost::mul(x4, x, F);
    cs.W.push_back(x2); cs.W.push_back(x4); cs.W.push_back(x5);
    cs.rows += 3;
    return x5;
  };
  if (!cs.ro->is_default) {                           // the original Poseidon permutation (family 1), as classic_permute
    const RoInstance& R = *cs.ro;
    const int t = R.spec.width, half = R.spec.full_rounds / 2, rounds = R.spec.full_rounds + R.spec.partial_rounds;
    const Fe* rcc = R.rc[cs.field_id].data();
    const Fe* M = R.mds[cs.field_id].data();
    Fe tmp[RO_MAX_T];
    for (int r = 0; r < rounds; ++r) {
      for (int i = 0; i < t; ++i) s[i] = vdfhost::add(s[i], rcc[(size_t)r * t + i], F);
      if (r < half || r >= half + R.spec.partial_rounds) for (int i = 0; i < t; ++i) s[i] = sbox_w(s[i]);
      else s[0] = sbox_w(s[0]);
      for (int i = 0; i < t; ++i) {
        Fe acc = vdfhost::mul(M[(size_t)i * t], s[0], F);
        for (int j = 1; j < t; ++j) acc = vdfhost::add(acc, vdfhost::mul(M[(size_t)i * t + j], s[j], F), F);
        tmp[i] = acc;
      }
      for (int i = 0; i < t; ++i) s[i] = tmp[i];
    }
    return;
  }
  const RoConstants& rc = ro_constants(cs.field_id);
  ext_layer(s, F);
  for (int r = 0; r < RO_RF + RO_RP; ++r) {
    if (r < RO_RF / 2 || r >= RO_RF / 2 + RO_RP) {
      const Fe* k = rc.ext[r < RO_RF / 2 ? r : r - RO_RP];
      for (int i = 0; i < 4; ++i) s[i] = sbox_w(vdfhost::add(s[i], k[i], F));
      ext_layer(s, F);
    } else {
      s[0] = sbox_w(vdfhost::add(s[0], rc.in[r - RO_RF / 2], F));
      int_layer(s, F);
    }
  }
}
static void poseidon_permute(CS& cs, std::vector<Num>& s) {
  if (!cs.ro->is_default) {                           // family 1 in R1CS: the matrix as linear combinations (oracle/nova.py poseidon_permute_classic)
    const RoInstance& R = *cs.ro;
    const int t = R.spec.width, half = R.spec.full_rounds / 2, rounds = R.spec.full_rounds + R.spec.partial_rounds;
    const Fe* rcc = R.rc[cs.field_id].data();
    const Fe* M = R.mds[cs.field_id].data();
    for (int r = 0; r < rounds; ++r) {
      for (int i = 0; i < t; ++i) s[i] = cs.add(s[i], cs.constant(rcc[(size_t)r * t + i]));
      if (r < half || r >= half + R.spec.partial_rounds) for (int i = 0; i < t; ++i) s[i] = sbox(cs, s[i]);
      else s[0] = sbox(cs, s[0]);
      std::vector<Num> o(t);
      for (int i = 0; i < t; ++i) {
        Num acc = cs.zero_num();                       // the oracle's lin(): from zero, term by term
        for (int j = 0; j < t; ++j) acc = cs.add(acc, cs.scale(s[j], M[(size_t)i * t + j]));
        o[i] = std::move(acc);
      }
      s.swap(o);
    }
    return;
  }
  const RoConstants& rc = ro_constants(cs.field_id);
  ext_layer_num(cs, s);
  for (int r = 0; r < RO_RF + RO_RP; ++r) {
    if (r < RO_RF / 2 || r >= RO_RF / 2 + RO_RP) {
      const Fe* k = rc.ext[r < RO_RF / 2 ? r : r - RO_RP];
      for (int i = 0; i < 4; ++i) s[i] = sbox(cs, cs.add(s[i], cs.constant(k[i])));
      ext_layer_num(cs, s);
    } else {
      s[0] = sbox(cs, cs.add(s[0], cs.constant(rc.in[r - RO_RF / 2])));
      const Num tot = cs.add(cs.add(s[0], s[1]), cs.add(s[2], s[3]));
      for (int i = 0; i < 4; ++i) s[i] = cs.add(tot, cs.scale_small(s[i], rc.mu_minus_1[i]));
    }
  }
}
Num poseidon_hash(CS& cs, uint64_t tag, const std::vector<Num>& xs) {
  const size_t rate = (size_t)cs.ro->rate, width = (size_t)cs.ro->spec.width;
  if (!cs.shape) {
    const Field& F = cs.F;
    Fe st[RO_MAX_T];
    st[0] = from_u64(tag + ((uint64_t)xs.size() << 32), F);
    for (size_t i = 1; i < width; ++i) st[i] = vdfhost::zero();
    for (size_t k = 0; k < xs.size(); k += rate) {
      for (size_t j = 0; j < rate && k + j < xs.size(); ++j) st[1 + j] = vdfhost::add(st[1 + j], xs[k + j].v, F);
      poseidon_permute_witness(cs, st);
    }
    Num out;
    out.v = st[1];
    return out;
  }
  std::vector<Num> s(width);
  s[0] = cs.constant_u64(tag + ((uint64_t)xs.size() << 32));
  for (size_t i = 1; i < width; ++i) s[i] = cs.zero_num();
  for (size_t k = 0; k < xs.size(); k += rate) {
    for (size_t j = 0; j < rate && k + j < xs.size(); ++j) s[1 + j] = cs.add(s[1 + j], xs[k + j]);
    poseidon_permute(cs, s);
  }
  return s[1];
}

// ---- curve y^2 = x^3 + 5, affine, identity = (0, 0) ------------------------------------------------------------
void check_on_curve(CS& cs, const Num& x, const Num& y, const Num& inf) {
  const Num x2 = cs.mul(x, x);
  const Num x3 = cs.mul(x2, x);
  const Num y2 = cs.mul(y, y);
  cs.enforce(cs.sub(cs.constant(one(cs.F)), inf), cs.sub(y2, cs.add(x3, cs.constant_u64(5))), cs.zero_num());
}

static void ec_double_raw(CS& cs, const Num& x, const Num& y, Num* ox, Num* oy) {
  const Field& F = cs.F;
  const Num x2 = cs.mul(x, x);
  const Num two_y = cs.scale_small(y, 2), three_x2 = cs.scale_small(x2, 3);
  const Num lam = cs.alloc(vdfhost::mul(three_x2.v, cs.take_inverse(two_y.v), F));
  cs.enforce(lam, two_y, three_x2);
  const Num two_x = cs.scale_small(x, 2);
  const Num dx = cs.alloc(vdfhost::sub(sqr(lam.v, F), two_x.v, F));
  cs.enforce(lam, lam, cs.add(dx, two_x));
  const Num dy = cs.alloc(vdfhost::sub(vdfhost::mul(lam.v, vdfhost::sub(x.v, dx.v, F), F), y.v, F));
  cs.enforce(lam, cs.sub(x, dx), cs.add(dy, y));
  *ox = dx; *oy = dy;
}
static void ec_add_raw(CS& cs, const Num& x1, const Num& y1, const Num& x2, const Num& y2, Num* ox, Num* oy) {
  const Field& F = cs.F;
  const Num dxn = cs.sub(x2, x1), dyn = cs.sub(y2, y1);
  const Num lam = cs.alloc(vdfhost::mul(dyn.v, cs.take_inverse(dxn.v), F));
  cs.enforce(lam, dxn, dyn);
  const Num sx = cs.alloc(vdfhost::sub(vdfhost::sub(sqr(lam.v, F), x1.v, F), x2.v, F));
  cs.enforce(lam, lam, cs.add(cs.add(sx, x1), x2));
  const Num sy = cs.alloc(vdfhost::sub(vdfhost::mul(lam.v, vdfhost::sub(x1.v, sx.v, F), F), y1.v, F));
  cs.enforce(lam, cs.sub(x1, sx), cs.add(sy, y1));
  *ox = sx; *oy = sy;
}

void ec_scalar_mul(CS& cs, const std::vector<Num>& bits, const Num& px, const Num& py, const Num& p_inf, Num* rx, Num* ry) {
  const Field& F = cs.F;
  const Num one_n = cs.constant(one(F));
  Num ax = cs.zero_num(), ay = cs.zero_num(), acc_inf = one_n, wx = px, wy = py;
  for (size_t k = 0; k < bits.size(); ++k) {
    Num sx, sy;
    ec_add_raw(cs, ax, ay, wx, wy, &sx, &sy);
    const Num cx = select(cs, acc_inf, wx, sx);
    const Num cy = select(cs, acc_inf, wy, sy);
    ax = select(cs, bits[k], cx, ax);
    ay = select(cs, bits[k], cy, ay);
    acc_inf = cs.mul(acc_inf, cs.sub(one_n, bits[k]));
    if (k + 1 < bits.size()) {
      Num nx, ny;
      ec_double_raw(cs, wx, wy, &nx, &ny);
      wx = nx; wy = ny;
    }
  }
  const Num keep = cs.sub(one_n, p_inf);
  *rx = cs.mul(keep, ax);
  *ry = cs.mul(keep, ay);
}

void ec_add_complete(CS& cs, const Num& x1, const Num& y1, const Num& x2, const Num& y2, Num* ox, Num* oy) {
  const Field& F = cs.F;
  const Num one_n = cs.constant(one(F));
  const Num i1 = is_zero(cs, x1);
  const Num i2 = is_zero(cs, x2);
  const Num same_x = is_zero(cs, cs.sub(x2, x1));
  const Num same_y = is_zero(cs, cs.sub(y2, y1));
  const Num x1sq = cs.mul(x1, x1);
  const Num num = select(cs, same_x, cs.scale_small(x1sq, 3), cs.sub(y2, y1));
  const Num den = select(cs, same_x, cs.scale_small(y1, 2), cs.sub(x2, x1));
  const Num lam = cs.alloc(vdfhost::mul(num.v, cs.take_inverse(den.v), F));
  cs.enforce(lam, den, num);
  const Num x3 = cs.alloc(vdfhost::sub(vdfhost::sub(sqr(lam.v, F), x1.v, F), x2.v, F));
  cs.enforce(lam, lam, cs.add(cs.add(x3, x1), x2));
  const Num y3 = cs.alloc(vdfhost::sub(vdfhost::mul(lam.v, vdfhost::sub(x1.v, x3.v, F), F), y1.v, F));
  cs.enforce(lam, cs.sub(x1, x3), cs.add(y3, y1));
  const Num is_neg = cs.mul(same_x, cs.sub(one_n, same_y));
  const Num keep = cs.sub(one_n, is_neg);
  const Num tx = cs.mul(keep, x3), ty = cs.mul(keep, y3);
  const Num ux = select(cs, i2, x1, tx), uy = select(cs, i2, y1, ty);
  *ox = select(cs, i1, x2, ux);
  *oy = select(cs, i1, y2, uy);
}

// Three helper threads for the independent blocks of a synthesis (the state hash, and one in-circuit fold per folded
// commitment): they run while the calling thread computes the challenge hash and the non-native folds.
// Persistent (created on first use, parked on a condition variable), one pair per process; a synthesis that finds them
// busy (another prover thread) simply does its pre-pass inline.
namespace {
class Helpers {
 public:
  // A process has up to SETS sets of helpers, made when first wanted and never destroyed (their threads outlive main):
  // one per prover running at a time -- two chains proven by two host threads each get their own.  nullptr: all taken,
  // the caller then runs the work itself.
  static Helpers* acquire() {
    static std::atomic<Helpers*> sets[SETS] = {};
    for (int k = 0; k < SETS; ++k) {
      Helpers* h = sets[k].load(std::memory_order_acquire);
      if (!h) {
        Helpers* made = new Helpers();
        made->busy_.store(true);
        if (sets[k].compare_exchange_strong(h, made)) return made;
        made->retire();                                   // lost the race for this slot: h is the winner's set
      }
      bool f = false;
      if (h->busy_.compare_exchange_strong(f, true)) return h;
    }
    return nullptr;
  }
  void release() { busy_.store(false); }
  void start(int k, std::function<void()> fn) {
    Slot& s = slots_[k];
    s.fn = std::move(fn);
    // Dekker-style handshake with loop(): "store state, then load parked" here against "store parked, then load state"
    // there.  Both pairs are seq_cst, so at least one side sees the other's store (release / acquire would allow both to
    // read stale values: a posted job and a helper asleep for good).
    s.state.store(1, std::memory_order_seq_cst);           // 1 = posted
    if (s.parked.load(std::memory_order_seq_cst)) { std::lock_guard<std::mutex> l(s.mu); s.cv.notify_one(); }
  }
  void wait(int k) {
    Slot& s = slots_[k];
    // the job is ~0.1 ms: spin, but hand the core over if it has not even started (an oversubscribed host)
    for (unsigned n = 0; s.state.load(std::memory_order_acquire) != 0; ++n) {
      if (n < 20000) __builtin_ia32_pause(); else std::this_thread::yield();
    }
  }
 private:
  // A helper polls for ~2 ms after its last job before it parks on a condition variable: a prover calls every
  // ~1.3 ms, and waking a parked thread (futex, idle core) costs about as much as the job itself.
  struct Slot {
    std::function<void()> fn;
    std::atomic<int> state{0};
    std::atomic<bool> parked{false};
    std::mutex mu;
    std::condition_variable cv;
  };
  static constexpr int SETS = 2;
  Helpers() { for (int k = 0; k < NSLOT; ++k) std::thread([this, k] { loop(k); }).detach(); }
  void retire() { quit_.store(true); for (int k = 0; k < NSLOT; ++k) { Slot& s = slots_[k]; std::lock_guard<std::mutex> l(s.mu); s.cv.notify_one(); } }   // leaked, threads exit
  void loop(int k) {
    Slot& s = slots_[k];
    for (;;) {
      auto idle_since = std::chrono::steady_clock::now();
      while (s.state.load(std::memory_order_acquire) != 1) {
        if (quit_.load(std::memory_order_relaxed)) return;
        __builtin_ia32_pause();
        if (std::chrono::steady_clock::now() - idle_since > std::chrono::milliseconds(2)) {
          std::unique_lock<std::mutex> l(s.mu);
          s.parked.store(true, std::memory_order_seq_cst);
          // the timeout is a backstop only (the handshake above does not need it): a parked helper looks again every 50 ms.
          // (wait_until on the system clock = pthread_cond_timedwait, which ThreadSanitizer intercepts; wait_for goes through
          // pthread_cond_clockwait, which gcc 11's does not, and it then reports the waiter's mutex as locked twice)
          while (!(s.state.load(std::memory_order_seq_cst) == 1 || quit_.load()))
            s.cv.wait_until(l, std::chrono::system_clock::now() + std::chrono::milliseconds(50));
          s.parked.store(false, std::memory_order_seq_cst);
        }
      }
      s.fn();
      s.fn = nullptr;
      s.state.store(0, std::memory_order_release);
    }
  }
  static constexpr int NSLOT = 3;
  Slot slots_[NSLOT];
  std::atomic<bool> busy_{false}, quit_{false};
};
}  // namespace

// Native pre-pass for  U + [r] P : the true points w_k = 2^k P and a_k = (r mod 2^(k+1)) P in XYZZ coordinates (no
// inversion), then every inverse the slopes will need, straight from those coordinates in ONE batched inversion:
//   chord of a_(k-1) and w_k:   1 / (x_w - x_a) = zz_w zz_a / (X_w zz_a - X_a zz_w)        (a_(k-1) = identity: zz_w / X_w)
//   tangent at w_k:             1 / (2 y_w)     = zzz_w / (2 Y_w)
//   final complete addition:    1 / (x_rP - x_U) = zz / (X - x_U zz)                        (equal x: 1 / (2 y_U), on its own)
// in the order the gadgets meet them: chord_0, tangent_0, chord_1, ..., chord_(bits-1), final.
// Everything on the doubling side needs only P, not r -- the points w_k, the tangents' inverses (a batch of their own), the
// affine doublings and the tangent's witness values: ec_fold_prepare makes them apart, while r is still being hashed
// (synthesize_augmented's late half); after r only the accumulator's chain, the chords' batch and their witness remain.
void ec_fold_prepare(const Field& F, const Aff& P, int bits, FoldPre* pre) {
  const size_t nb = (size_t)bits;
  pre->w.resize(nb);
  pre->tan_inv.resize(nb ? nb - 1 : 0);
  pre->wx.resize(nb); pre->wy.resize(nb);
  pre->x2.resize(nb ? nb - 1 : 0); pre->lam.resize(nb ? nb - 1 : 0);
  Pt cur = pt_from_aff(P, F);
  for (int k = 0; k < bits; ++k) {
    pre->w[k] = cur;
    if (k + 1 < bits) { pre->tan_inv[k] = vdfhost::add(cur.y, cur.y, F); cur = pt_dbl(cur, F); }
  }
  // 1 / (2 y_k) = zzz_k / (2 Y_k): one batched inversion for all tangents
  batch_inverse(pre->tan_inv.data(), pre->tan_inv.size(), F);
  for (size_t k = 0; k + 1 < nb; ++k) pre->tan_inv[k] = vdfhost::mul(pre->tan_inv[k], pre->w[k].zzz, F);
  // the affine doublings and the tangent's witness values, as ec_double_raw computes them; each inverse is checked the
  // way CS::take_inverse checks a queued one (a wrong one is replaced by a real inversion and counted)
  Fe wx = P.x, wy = P.y;
  pre->misses = 0;
  for (size_t k = 0; k < nb; ++k) {
    pre->wx[k] = wx; pre->wy[k] = wy;
    if (k + 1 == nb) break;
    const Fe two_y = vdfhost::add(wy, wy, F);
    Fe inv = pre->tan_inv[k];
    if (!(two_y.is_zero() ? inv.is_zero() : vdfhost::mul(two_y, inv, F) == one(F))) { inv = inverse(two_y, F); pre->tan_inv[k] = inv; ++pre->misses; }
    const Fe x2 = sqr(wx, F);
    const Fe lam = vdfhost::mul(vdfhost::add(vdfhost::add(x2, x2, F), x2, F), inv, F);
    const Fe dx = vdfhost::sub(sqr(lam, F), vdfhost::add(wx, wx, F), F);
    const Fe dy = vdfhost::sub(vdfhost::mul(lam, vdfhost::sub(wx, dx, F), F), wy, F);
    pre->x2[k] = x2; pre->lam[k] = lam;
    wx = dx; wy = dy;
  }
}
void ec_fold_inverses(const Field& F, const Aff& U, const Aff& P, const uint64_t r[4], int bits, std::vector<Fe>* out,
                      const FoldPre* prepared) {
  FoldPre own;
  if (!prepared) { ec_fold_prepare(F, P, bits, &own); prepared = &own; }
  const FoldPre& pre = *prepared;
  const size_t nb = (size_t)bits, base = out->size(), n = 2 * nb;
  out->resize(base + n);
  Fe* q = out->data() + base;                               // chord_0, tangent_0, chord_1, ..., chord_(bits-1), final
  static thread_local std::vector<Fe> d, scale;             // the chords and the final slope: inverted together here
  d.resize(nb + 1); scale.resize(nb + 1);
  // the accumulator's chain as MIXED additions with the affine doublings of `pre`; the chord's denominator
  // x_w - x_a = (x_w zz_a - X_a) / zz_a is the first product of that addition
  Pt acc = pt_identity();
  const bool p_id = P.x.is_zero() && P.y.is_zero();
  for (size_t k = 0; k < nb; ++k) {
    const Fe& wx = pre.wx[k];
    const Fe& wy = pre.wy[k];
    const bool bit = (r[k / 64] >> (k % 64)) & 1;
    if (p_id) { d[k] = vdfhost::zero(); scale[k] = one(F); continue; }
    if (acc.is_id()) {
      d[k] = wx; scale[k] = one(F);
      if (bit) { acc.x = wx; acc.y = wy; acc.zz = one(F); acc.zzz = one(F); }
      continue;
    }
    const Fe pp1 = vdfhost::sub(vdfhost::mul(wx, acc.zz, F), acc.x, F);
    d[k] = pp1; scale[k] = acc.zz;
    if (!bit) continue;
    if (pp1.is_zero()) { acc = pt_add(acc, pt_from_aff(Aff{wx, wy}, F), F); continue; }    // never for a point of prime order
    const Fe rr = vdfhost::sub(vdfhost::mul(wy, acc.zzz, F), acc.y, F);
    const Fe pp = sqr(pp1, F), ppp = vdfhost::mul(pp1, pp, F), qq = vdfhost::mul(acc.x, pp, F);
    const Fe x3 = vdfhost::sub(vdfhost::sub(vdfhost::sub(sqr(rr, F), ppp, F), qq, F), qq, F);
    acc.y = vdfhost::sub(vdfhost::mul(rr, vdfhost::sub(qq, x3, F), F), vdfhost::mul(acc.y, ppp, F), F);
    acc.x = x3;
    acc.zz = vdfhost::mul(acc.zz, pp, F);
    acc.zzz = vdfhost::mul(acc.zzz, ppp, F);
  }
  // acc = [r] P now (the identity also when P is)
  bool same_x = false;
  if (acc.is_id()) { d[nb] = vdfhost::sub(vdfhost::zero(), U.x, F); scale[nb] = one(F); }     // x_rP = 0
  else {
    d[nb] = vdfhost::sub(acc.x, vdfhost::mul(U.x, acc.zz, F), F);
    scale[nb] = acc.zz;
    same_x = d[nb].is_zero();
  }
  if (U.x.is_zero() && acc.is_id()) same_x = true;
  if (same_x) { d[nb] = vdfhost::add(U.y, U.y, F); scale[nb] = one(F); }
  batch_inverse(d.data(), nb + 1, F);
  for (size_t k = 0; k < nb; ++k) {
    q[2 * k] = vdfhost::mul(d[k], scale[k], F);
    if (k + 1 < nb) q[2 * k + 1] = pre.tan_inv[k];
  }
  q[n - 1] = vdfhost::mul(d[nb], scale[nb], F);
}

// Witness of ec_scalar_mul (above) written directly: the same variables in the same order -- per bit the chord's slope and
// sum (3), the two selections by "accumulator still empty" (2), the two by the bit (2), the emptiness flag (1), then the
// tangent's x^2, slope and double (4, not after the last bit); finally keep * acc (2) -- computed with nine field
// multiplications per bit instead of through ~40 Num operations (the tangent's four come ready from ec_fold_prepare).
// Slopes take their inverses from cs.inv_queue (ec_fold_inverses), checked as take_inverse checks them.  tests/test_nova_host.py compares the result with the oracle's
// gadget-by-gadget synthesis.
static void ec_scalar_mul_witness(CS& cs, const uint64_t r[4], int bits, const Aff& P, const FoldPre& pre, Fe* rx, Fe* ry) {
  const Field& F = cs.F;
  const Fe ONE = one(F), ZERO = vdfhost::zero();
  Fe ax = ZERO, ay = ZERO;
  bool acc_inf = true;
  cs.inv_misses += pre.misses;
  const size_t count = (size_t)bits * 8 + (size_t)(bits - 1) * 4 + 2, at = cs.W.size();
  cs.W.resize(at + count);
  Fe* out = cs.W.data() + at;
  for (int k = 0; k < bits; ++k) {
    const bool bit = (r[k / 64] >> (k % 64)) & 1;
    const Fe& wx = pre.wx[k];
    const Fe& wy = pre.wy[k];
    const Fe dxn = vdfhost::sub(wx, ax, F), dyn = vdfhost::sub(wy, ay, F);
    const Fe lam = vdfhost::mul(dyn, cs.take_inverse(dxn), F);
    const Fe sx = vdfhost::sub(vdfhost::sub(sqr(lam, F), ax, F), wx, F);
    const Fe sy = vdfhost::sub(vdfhost::mul(lam, vdfhost::sub(ax, sx, F), F), ay, F);
    const Fe cx = acc_inf ? wx : sx, cy = acc_inf ? wy : sy;
    if (bit) { ax = cx; ay = cy; acc_inf = false; }
    out[0] = lam; out[1] = sx; out[2] = sy; out[3] = cx; out[4] = cy; out[5] = ax; out[6] = ay; out[7] = acc_inf ? ONE : ZERO;
    out += 8;
    if (k + 1 < bits) {                                   // the tangent: prepared, its inverse checked there
      ++cs.inv_pos;
      out[0] = pre.x2[k]; out[1] = pre.lam[k]; out[2] = pre.wx[k + 1]; out[3] = pre.wy[k + 1];
      out += 4;
    }
  }
  cs.rows += count;
  const bool p_inf = P.x.is_zero();
  *rx = p_inf ? ZERO : ax;
  *ry = p_inf ? ZERO : ay;
  out[0] = *rx; out[1] = *ry;
}

// ---- multi-limb integers for the foreign fold (little-endian 64-bit limbs) ---------------------------------------
namespace {
struct Wide { uint64_t l[8]; };                        // up to 512 bits
Wide wide_zero() { Wide w; memset(&w, 0, sizeof(w)); return w; }
Wide wide_from4(const uint64_t v[4]) { Wide w = wide_zero(); memcpy(w.l, v, 32); return w; }
Wide wide_mul(const uint64_t* a, int na, const uint64_t* b, int nb) {
  Wide w = wide_zero();
  for (int i = 0; i < na; ++i) {
    u128 c = 0;
    for (int j = 0; j < nb && i + j < 8; ++j) {
      c += (u128)a[i] * b[j] + w.l[i + j];
      w.l[i + j] = (uint64_t)c;
      c >>= 64;
    }
    for (int k = i + nb; k < 8 && c; ++k) { c += w.l[k]; w.l[k] = (uint64_t)c; c >>= 64; }
  }
  return w;
}
Wide wide_add(const Wide& a, const Wide& b) {
  Wide w;
  u128 c = 0;
  for (int i = 0; i < 8; ++i) { c += (u128)a.l[i] + b.l[i]; w.l[i] = (uint64_t)c; c >>= 64; }
  return w;
}
bool wide_geq(const Wide& a, const Wide& b) {
  for (int i = 7; i >= 0; --i) if (a.l[i] != b.l[i]) return a.l[i] > b.l[i];
  return true;
}
Wide wide_sub(const Wide& a, const Wide& b) {
  Wide w;
  u128 br = 0;
  for (int i = 0; i < 8; ++i) {
    u128 d = (u128)a.l[i] - b.l[i] - (uint64_t)br;
    w.l[i] = (uint64_t)d;
    br = (d >> 64) & 1;
  }
  return w;
}
Wide wide_shl(const Wide& a, int s) {
  Wide w = wide_zero();
  const int ls = s / 64, bs = s % 64;
  for (int i = 7; i >= ls; --i) {
    w.l[i] = a.l[i - ls] << bs;
    if (bs && i - ls - 1 >= 0) w.l[i] |= a.l[i - ls - 1] >> (64 - bs);
  }
  return w;
}
Wide wide_shr(const Wide& a, int s) {
  Wide w = wide_zero();
  const int ls = s / 64, bs = s % 64;
  for (int i = 0; i + ls < 8; ++i) {
    w.l[i] = a.l[i + ls] >> bs;
    if (bs && i + ls + 1 < 8) w.l[i] |= a.l[i + ls + 1] << (64 - bs);
  }
  return w;
}
// quotient (< 2^qbits) and remainder of a / d by shift and subtract
void wide_divmod(const Wide& a, const Wide& d, int qbits, Wide* q, Wide* r) {
  Wide rem = a, quo = wide_zero();
  for (int b = qbits - 1; b >= 0; --b) {
    const Wide ds = wide_shl(d, b);
    if (wide_geq(rem, ds)) { rem = wide_sub(rem, ds); quo.l[b / 64] |= 1ull << (b % 64); }
  }
  *q = quo; *r = rem;
}
void low_bits(const uint64_t v[4], int n, uint64_t out[4]) {       // v mod 2^n, n < 256
  for (int i = 0; i < 4; ++i) {
    const int lo = 64 * i;
    out[i] = n >= lo + 64 ? v[i] : (n > lo ? v[i] & ((1ull << (n - lo)) - 1) : 0);
  }
}
}  // namespace

void fold_foreign(CS& cs, const Num& a_lo, const Num& a_hi, const std::vector<Num>& b_bits, const std::vector<Num>& r_bits,
                  const Field& PF, Num* out_lo, Num* out_hi) {
  const Field& F = cs.F;
  const int L = LIMB_BITS;
  const Num b_lo = pack(cs, b_bits.data(), L), b_all = pack(cs, b_bits.data(), b_bits.size());
  const Num r_lo = pack(cs, r_bits.data(), L), r_all = pack(cs, r_bits.data(), r_bits.size());
  uint64_t alo[4], ahi[4], blo[4], ball[4], rlo[4], rall[4];
  fe_to_int(a_lo.v, F, alo); fe_to_int(a_hi.v, F, ahi);
  fe_to_int(b_lo.v, F, blo); fe_to_int(b_all.v, F, ball);
  fe_to_int(r_lo.v, F, rlo); fe_to_int(r_all.v, F, rall);
  // tot = A + r B, A = a_lo + 2^126 a_hi
  const Wide A = wide_add(wide_from4(alo), wide_shl(wide_from4(ahi), L));
  const Wide tot = wide_add(A, wide_mul(rall, 2, ball, 4));
  Wide kq, R;
  wide_divmod(tot, wide_from4(PF.m), 130, &kq, &R);
  uint64_t Rlo[4], Rhi[4];
  low_bits(R.l, L, Rlo);
  { const Wide h = wide_shr(R, L); memcpy(Rhi, h.l, 32); }
  const std::vector<Num> k_bits = alloc_bits(cs, kq.l, 125);
  const std::vector<Num> rlo_bits = alloc_bits(cs, Rlo, L);
  const std::vector<Num> rhi_bits = alloc_bits(cs, Rhi, 129);
  const Num k = pack(cs, k_bits.data(), 125);
  const Num R_lo = pack(cs, rlo_bits.data(), L), R_hi = pack(cs, rhi_bits.data(), 129);
  // (1) modulo the native field: r B = k p' + R_lo + D R_hi - a_lo - D a_hi
  const Fe D = pow2_table(F)[L];
  const Fe pf_nat = int_to_fe(PF.m, F);
  Num rhs = cs.add(cs.scale(k, pf_nat), R_lo);
  rhs = cs.add(rhs, cs.scale(R_hi, D));
  rhs = cs.sub(rhs, a_lo);
  rhs = cs.sub(rhs, cs.scale(a_hi, D));
  cs.enforce(r_all, b_all, rhs);
  // (2) modulo 2^126: a_lo + r_lo b_lo - k (p' mod D) - R_lo = (c' - 2^127) D
  const Num prod = cs.mul(r_lo, b_lo);
  uint64_t pfl[4];
  low_bits(PF.m, L, pfl);
  Wide pos = wide_add(wide_from4(alo), wide_mul(rlo, 2, blo, 2));
  { Wide off = wide_zero(); off.l[3] = 1ull << 61; pos = wide_add(pos, off); }        // + 2^127 * 2^126 = 2^253
  const Wide negv = wide_add(wide_mul(kq.l, 2, pfl, 2), wide_from4(Rlo));
  const Wide cp = wide_shr(wide_sub(pos, negv), L);
  const std::vector<Num> c_bits = alloc_bits(cs, cp.l, 128);
  Num lhs = cs.add(a_lo, prod);
  lhs = cs.sub(lhs, cs.scale(k, int_to_fe(pfl, F)));
  lhs = cs.sub(lhs, R_lo);
  const Num cnum = cs.sub(pack(cs, c_bits.data(), 128), cs.constant(pow2_table(F)[127]));
  cs.enforce_equal(lhs, cs.scale(cnum, D));
  *out_lo = R_lo; *out_hi = R_hi;
}

// =============================================================================================================
// Step circuits
// =============================================================================================================
std::vector<Num> InverseMinRootCircuit::synthesize(CS& cs, const std::vector<Num>& z) const {
  const Field& F = cs.F;
  Num x = z[0], y = z[1];
  const Num& i_in = z[2];
  if (!cs.shape && device_rounds) {
    // witness mode with the rounds left to the GPU (vdf_minroot_step_segment fills these variables from the forward
    // trace): the outputs are the stored previous state, as StepCircuit::output has them (src/nova/proof.rs:142-152)
    cs.skip(vars_per_round() * t + 1, 3 * t + 1);
    std::vector<Num> out(3);
    out[0].v = input.x; out[1].v = input.y; out[2].v = input.i;
    return out;
  }
  for (uint64_t j = 0; j < t; ++j) {
    const Num new_x_lc = cs.add(cs.sub(y, i_in), cs.constant_u64(j + 1));       // y - (i - 1), i = i_in - j (:162-164)
    Num new_x;
    if (!bound) new_x = cs.alloc(new_x_lc.v);                                     // :167-173
    const Num tmp1 = cs.mul(x, x);                                                // :176
    const Num tmp2 = cs.mul(tmp1, tmp1);                                          // :178
    const Num new_y = cs.alloc(vdfhost::sub(vdfhost::mul(tmp2.v, x.v, F), new_x_lc.v, F));   // :181-189
    cs.enforce(tmp2, x, cs.add(new_y, new_x_lc));                                 // :219-227
    x = bound ? new_x_lc : new_x;
    y = new_y;
  }
  const Num tn = cs.constant_u64(t);
  const Num final_i = cs.alloc(vdfhost::sub(i_in.v, tn.v, F));                    // :122-133
  cs.enforce(final_i, cs.constant(one(F)), cs.sub(i_in, tn));
  return {x, y, final_i};
}
void InverseMinRootCircuit::output(const Fe* z, Fe* out) const {
  (void)z;
  out[0] = input.x; out[1] = input.y; out[2] = input.i;
}

// =============================================================================================================
// Instances, native hashes, the augmented circuit
// =============================================================================================================
static void split126(const uint64_t v[4], uint64_t lo[4], uint64_t hi[4]) {
  low_bits(v, LIMB_BITS, lo);
  const Wide h = wide_shr(wide_from4(v), LIMB_BITS);
  memcpy(hi, h.l, 32);
}

void relaxed_elements(const RelaxedInst& U, const Field& F, Fe out[9]) {
  out[0] = U.comm_W.x; out[1] = U.comm_W.y; out[2] = U.comm_E.x; out[3] = U.comm_E.y;
  out[4] = int_to_fe(U.u, F);
  for (int k = 0; k < 2; ++k) {
    uint64_t lo[4], hi[4];
    split126(U.X[k], lo, hi);
    out[5 + 2 * k] = int_to_fe(lo, F);
    out[6 + 2 * k] = int_to_fe(hi, F);
  }
}

Fe hash_state(int f, const Fe& params, const Fe& i, const std::vector<Fe>& z0, const std::vector<Fe>& zi, const RelaxedInst& U,
              uint64_t out_int[4], const RoInstance* ro) {
  const Field& F = field(f);
  std::vector<Fe> xs = {params, i};
  xs.insert(xs.end(), z0.begin(), z0.end());
  xs.insert(xs.end(), zi.begin(), zi.end());
  Fe ue[9];
  relaxed_elements(U, F, ue);
  xs.insert(xs.end(), ue, ue + 9);
  uint64_t h[4];
  fe_to_int(ro_hash(f, TAG_STATE, xs.data(), xs.size(), ro), F, h);
  low_bits(h, HASH_BITS, out_int);
  return int_to_fe(out_int, F);
}

void hash_challenge(int f, const Fe& params, const RelaxedInst& U, const Aff& u_W, const uint64_t u_X[2][4], const Aff& T,
                    uint64_t r_out[4], const RoInstance* ro) {
  const Field& F = field(f);
  Fe xs[16];
  xs[0] = params;
  relaxed_elements(U, F, xs + 1);
  xs[10] = u_W.x; xs[11] = u_W.y;
  xs[12] = int_to_fe(u_X[0], F); xs[13] = int_to_fe(u_X[1], F);
  xs[14] = T.x; xs[15] = T.y;
  uint64_t h[4];
  fe_to_int(ro_hash(f, TAG_CHAL, xs, 16, ro), F, h);
  low_bits(h, CHAL_BITS, r_out);
}

static thread_local uint64_t g_last_queue = 0, g_last_misses = 0;

// Witness mode, block-parallel.  The augmented circuit's variables come in contiguous runs that depend on each other only
// through a few values: [inputs] [state hash] [challenge hash] [curve checks] [fold of comm_W] [fold of comm_E]
// [non-native folds] [selection, step circuit, output hash].  Each independent run is synthesised into a CS of its own by
// THE SAME gadget code (so its variables come out in the order the shape has them) and spliced into place:
//   helper 0: state hash;            this thread: challenge hash -> r;
//   helpers 1, 2: the two in-circuit folds (native pre-pass + gadgets);     this thread meanwhile: the non-native folds.
// tests/test_nova_host.py compares every variable with the oracle; VDF_NOVA_SEQ_SYNTH=1 selects the sequential path.
namespace {
struct Blk { std::vector<Fe> W; size_t rows = 0; ~Blk() { witness_buffer_give(std::move(W)); } };
inline Num val(const Fe& v) { Num n; n.v = v; return n; }
inline void splice(CS& cs, Blk& b) { cs.W.insert(cs.W.end(), b.W.begin(), b.W.end()); cs.rows += b.rows; }
inline void take(Blk& b, CS& t) { t.resolve(); b.W = std::move(t.W); t.W = std::vector<Fe>(); b.rows = t.rows; }
}  // namespace

// poseidon_hash (above) in witness mode, element by element
namespace {
struct WSponge {
  Fe st[RO_MAX_T];
  size_t fill = 0;
  void init(uint64_t tag, size_t len, const Field& F) {
    st[0] = from_u64(tag + ((uint64_t)len << 32), F);
    for (int i = 1; i < RO_MAX_T; ++i) st[i] = vdfhost::zero();
    fill = 0;
  }
  void absorb(CS& cs, const Fe& x) {
    st[1 + fill] = vdfhost::add(st[1 + fill], x, cs.F);
    if (++fill == (size_t)cs.ro->rate) { poseidon_permute_witness(cs, st); fill = 0; }
  }
  Fe finish(CS& cs) {
    if (fill) { poseidon_permute_witness(cs, st); fill = 0; }
    return st[1];
  }
};
}  // namespace

struct AugEarly {
  int side = 0, fid = 0;
  size_t a = 0;
  AugInputs in;                          // as given early: u_W and T are not read
  Fe ue[9], uX[2];
  Blk b1;                                // the state hash (helper 0)
  std::unique_ptr<CS> chal, foreign, outh;
  WSponge chal_sp, out_sp;
  std::vector<Num> xb[2];
  bool have_out = false;
  std::vector<Fe> z_out;
  Helpers* helpers = nullptr;            // the set this synthesis holds (null: none was free, everything runs inline)
  bool helped = false, pending0 = false;
  ~AugEarly() {
    if (helped) { if (pending0) helpers->wait(0); helpers->release(); }
  }
};
void aug_early_free(AugEarly* e) { delete e; }

AugEarlyPtr synthesize_augmented_early(int side, const AugInputs& in, const StepCircuit& step) {
  static const bool trace = [] { const char* e = env_override("VDF_NOVA_SYNTH_TRACE"); return e && e[0] == '1'; }();
  const auto T0 = std::chrono::steady_clock::now();
  AugEarlyPtr e(new AugEarly(), aug_early_free);
  const int fid = side_field(side);
  const Field& F = field(fid);
  e->side = side; e->fid = fid; e->a = step.arity();
  e->in = in;
  relaxed_elements(in.U, F, e->ue);
  for (int k = 0; k < 2; ++k) e->uX[k] = int_to_fe(in.u_X[k], F);
  const size_t a = e->a;
  // ---- block 1 (helper 0): the hash this step must have been handed
  AugEarly* p = e.get();
  auto run_b1 = [p] {
    CS t(p->fid, false, p->in.ro);
    WSponge sp;
    sp.init(TAG_STATE, 2 + 2 * p->a + 9, t.F);
    sp.absorb(t, p->in.params); sp.absorb(t, p->in.i);
    for (size_t k = 0; k < p->a; ++k) sp.absorb(t, p->in.z0[k]);
    for (size_t k = 0; k < p->a; ++k) sp.absorb(t, p->in.zi[k]);
    for (int k = 0; k < 9; ++k) sp.absorb(t, p->ue[k]);
    strict_bits(t, val(sp.finish(t)));
    take(p->b1, t);
  };
  e->helpers = Helpers::acquire();
  e->helped = e->helpers != nullptr;
  if (e->helped) { e->helpers->start(0, run_b1); e->pending0 = true; } else run_b1();
  // ---- block 2, first half: the challenge hash over what is known (params and the running instance)
  e->chal.reset(new CS(fid, false, in.ro));
  e->chal_sp.init(TAG_CHAL, 16, F);
  e->chal_sp.absorb(*e->chal, in.params);
  for (int k = 0; k < 9; ++k) e->chal_sp.absorb(*e->chal, e->ue[k]);
  // ---- block 5, first half: the bits of u.X
  e->foreign.reset(new CS(fid, false, in.ro));
  for (int k = 0; k < 2; ++k) e->xb[k] = alloc_bits(*e->foreign, in.u_X[k], HASH_BITS);
  e->foreign->rows += 2;                              // the two packings equal u.X[k]
  // ---- the output hash up to z_out, when the step circuit can tell it
  if (step.output_known()) {
    const bool is_base = in.i.is_zero();
    std::vector<Fe> z_in(a);
    for (size_t k = 0; k < a; ++k) z_in[k] = is_base ? in.z0[k] : in.zi[k];
    e->z_out.resize(a);
    step.output(z_in.data(), e->z_out.data());
    e->outh.reset(new CS(fid, false, in.ro));
    e->out_sp.init(TAG_STATE, 2 + 2 * a + 9, F);
    e->out_sp.absorb(*e->outh, in.params);
    e->out_sp.absorb(*e->outh, vdfhost::add(in.i, one(F), F));
    for (size_t k = 0; k < a; ++k) e->out_sp.absorb(*e->outh, in.z0[k]);
    for (size_t k = 0; k < a; ++k) e->out_sp.absorb(*e->outh, e->z_out[k]);
    e->have_out = true;
  }
  if (trace) fprintf(stderr, "synth side %d: early half %.0f us on the calling thread\n", side,
                     std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - T0).count());
  return e;
}

static std::vector<Fe> synthesize_augmented_blocks(CS& cs, int side, const AugInputs& in, const StepCircuit& step, Fe* unew_out,
                                                   uint64_t* r_out, AugEarly* early) {
  const Field& F = cs.F;
  const Field& PF = field(side_field(1 - side));
  const int fid = cs.field_id;
  const size_t a = step.arity();
  const Fe ONE = one(F), ZERO = vdfhost::zero();
  static const bool trace = [] { const char* e = env_override("VDF_NOVA_SYNTH_TRACE"); return e && e[0] == '1'; }();
  const auto T0 = std::chrono::steady_clock::now();
  auto us = [&] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - T0).count(); };
  double tr[8] = {0};
  AugEarlyPtr own(nullptr, aug_early_free);
  if (!early) { own = synthesize_augmented_early(side, in, step); early = own.get(); }
  AugEarly& e = *early;
  if (e.side != side || e.a != a || e.in.ro != in.ro || e.in.params != in.params || e.in.i != in.i || e.in.z0 != in.z0 || e.in.zi != in.zi ||
      memcmp(&e.in.U, &in.U, sizeof(RelaxedInst)) != 0 || memcmp(e.in.u_X, in.u_X, sizeof(in.u_X)) != 0)
    throw std::runtime_error("synthesize_augmented: the early half was made for other inputs");
  const Fe* ue = e.ue;
  const Fe* uX = e.uX;
  const Fe i_new_v = vdfhost::add(in.i, ONE, F);
  tr[0] = us();
  Blk b2, b3, b4, b5, b6;
  uint64_t rv[4] = {0, 0, 0, 0};
  std::atomic<bool> r_ready{false};
  struct alignas(64) FoldOut { Fe x, y; size_t queued = 0, misses = 0; double done = 0, tr[4] = {0, 0, 0, 0}; } fo_w, fo_e;
  auto run_fold = [&](const Aff& Upt, const Aff& P, Blk* b, FoldOut* o) {
    CS t(fid, false, in.ro);
    static thread_local FoldPre pre;                  // a helper thread keeps its scratch: no allocation, no page fault per step
    ec_fold_prepare(F, P, CHAL_BITS, &pre);
    o->tr[0] = us();
    for (unsigned n = 0; !r_ready.load(std::memory_order_acquire); ++n) { if (n < 200000) __builtin_ia32_pause(); else std::this_thread::yield(); }
    o->tr[1] = us();
    ec_fold_inverses(F, Upt, P, rv, CHAL_BITS, &t.inv_queue, &pre);
    o->tr[2] = us();
    Num rx, ry, fx, fy;
    ec_scalar_mul_witness(t, rv, CHAL_BITS, P, pre, &rx.v, &ry.v);
    o->tr[3] = us();
    ec_add_complete(t, val(Upt.x), val(Upt.y), rx, ry, &fx, &fy);
    o->x = fx.v; o->y = fy.v;
    o->queued = t.inv_queue.size();
    o->misses = t.inv_misses + (t.inv_queue.size() - t.inv_pos);
    take(*b, t);
    o->done = us();
  };
  Helpers* Hp = e.helpers;
  bool pending[3] = {e.pending0, false, false};
  // an exception below must not leave a fold spinning for r: the joiner releases them first
  struct Joiner { Helpers* h; bool* p; std::atomic<bool>* go; ~Joiner() { go->store(true, std::memory_order_release); for (int k = 1; k < 3; ++k) if (p[k]) h->wait(k); } } joiner{Hp, pending, &r_ready};   // slot 0: ~AugEarly
  // ---- blocks 3, 4 (helpers 1, 2): U + [r] P, slopes from the native pre-pass.  Started before r is known: the doubling
  // side (ec_fold_prepare) needs only P and runs beside the challenge hash; each fold then waits for r_ready.
  if (e.helped) {
    Hp->start(1, [&] { run_fold(in.U.comm_W, in.u_W, &b3, &fo_w); }); pending[1] = true;
    Hp->start(2, [&] { run_fold(in.U.comm_E, in.T, &b4, &fo_e); }); pending[2] = true;
  }
  // ---- block 2, second half (this thread): the fold challenge
  std::vector<Num> r_bits;
  {
    CS& t = *e.chal;
    for (const Fe& v : {in.u_W.x, in.u_W.y, uX[0], uX[1], in.T.x, in.T.y}) e.chal_sp.absorb(t, v);
    r_bits = strict_bits(t, val(e.chal_sp.finish(t)));
    r_bits.resize(CHAL_BITS);
    fe_to_int(pack(t, r_bits.data(), CHAL_BITS).v, F, rv);
    r_ready.store(true, std::memory_order_release);
    take(b2, t);
  }
  if (in.on_challenge) in.on_challenge(rv);          // (the helpers' folds run meanwhile; this thread has the slack: it joins them below)
  tr[1] = us();
  // ---- block 5, second half (this thread, while the folds run): X' = X + r x in the other field
  Fe f_lo[2], f_hi[2], x_lo[2], x_hi[2];
  {
    CS& t = *e.foreign;
    for (int k = 0; k < 2; ++k) {
      Num lo, hi;
      fold_foreign(t, val(ue[5 + 2 * k]), val(ue[6 + 2 * k]), e.xb[k], r_bits, PF, &lo, &hi);
      f_lo[k] = lo.v; f_hi[k] = hi.v;
      x_lo[k] = pack(t, e.xb[k].data(), LIMB_BITS).v;
      x_hi[k] = pack(t, e.xb[k].data() + LIMB_BITS, HASH_BITS - LIMB_BITS).v;
    }
    take(b5, t);
  }
  tr[2] = us();
  if (!e.helped) { run_fold(in.U.comm_W, in.u_W, &b3, &fo_w); run_fold(in.U.comm_E, in.T, &b4, &fo_e); }
  // ---- assembly, in the shape's order
  const Num params = cs.alloc(in.params);
  const Num i = cs.alloc(in.i);
  std::vector<Num> z0, zi;
  for (size_t k = 0; k < a; ++k) z0.push_back(cs.alloc(in.z0[k]));
  for (size_t k = 0; k < a; ++k) zi.push_back(cs.alloc(in.zi[k]));
  for (int k = 0; k < 9; ++k) cs.alloc(ue[k]);
  const Num uWx = cs.alloc(in.u_W.x), uWy = cs.alloc(in.u_W.y);
  cs.alloc(uX[0]); cs.alloc(uX[1]);
  const Num Tx = cs.alloc(in.T.x), Ty = cs.alloc(in.T.y);
  const Num is_base = is_zero(cs, i);
  if (e.pending0) { Hp->wait(0); e.pending0 = false; pending[0] = false; }
  splice(cs, e.b1);
  cs.rows += 1;                                       // (1 - is_base) (u.X[0] - h_in) = 0
  splice(cs, b2);
  const Num uW_inf = is_zero(cs, uWx);
  check_on_curve(cs, uWx, uWy, uW_inf);
  const Num T_inf = is_zero(cs, Tx);
  check_on_curve(cs, Tx, Ty, T_inf);
  if (pending[1]) { Hp->wait(1); pending[1] = false; }
  if (pending[2]) { Hp->wait(2); pending[2] = false; }
  if (e.helped) { Hp->release(); e.helped = false; }
  tr[3] = us();
  splice(cs, b3);
  splice(cs, b4);
  splice(cs, b5);
  const Fe fu = vdfhost::add(ue[4], int_to_fe(rv, F), F);
  const Fe fold[9] = {fo_w.x, fo_w.y, fo_e.x, fo_e.y, fu, f_lo[0], f_hi[0], f_lo[1], f_hi[1]};
  Fe base[9];
  if (side == 0) for (int k = 0; k < 9; ++k) base[k] = ZERO;
  else { base[0] = in.u_W.x; base[1] = in.u_W.y; base[2] = ZERO; base[3] = ZERO; base[4] = ONE;
         base[5] = x_lo[0]; base[6] = x_hi[0]; base[7] = x_lo[1]; base[8] = x_hi[1]; }
  std::vector<Num> Unew;
  for (int k = 0; k < 9; ++k) Unew.push_back(select(cs, is_base, val(base[k]), val(fold[k])));
  std::vector<Num> z_in;
  for (size_t k = 0; k < a; ++k) z_in.push_back(select(cs, is_base, z0[k], zi[k]));
  cs.step_begin = cs.num_vars();
  const std::vector<Num> z_out = step.synthesize(cs, z_in);
  cs.step_end = cs.num_vars();
  tr[4] = us();
  if (unew_out) for (int k = 0; k < 9; ++k) unew_out[k] = Unew[k].v;
  if (r_out) memcpy(r_out, rv, 32);
  // ---- the output hash: continued from where the early half left it, or from the start
  Fe h_out_v;
  {
    // (a circuit whose synthesis disagrees with what output() announced -- states that are not one evaluation apart --
    // is hashed as synthesised: the early half's prefix is dropped)
    if (e.have_out)
      for (size_t k = 0; k < a; ++k) if (e.z_out[k] != z_out[k].v) e.have_out = false;
    if (!e.have_out) {
      e.outh.reset(new CS(fid, false, in.ro));
      e.out_sp.init(TAG_STATE, 2 + 2 * a + 9, F);
      e.out_sp.absorb(*e.outh, in.params);
      e.out_sp.absorb(*e.outh, i_new_v);
      for (size_t k = 0; k < a; ++k) e.out_sp.absorb(*e.outh, in.z0[k]);
      for (size_t k = 0; k < a; ++k) e.out_sp.absorb(*e.outh, z_out[k].v);
    }
    CS& t = *e.outh;
    for (int k = 0; k < 9; ++k) e.out_sp.absorb(t, Unew[k].v);
    const std::vector<Num> h_out = strict_bits(t, val(e.out_sp.finish(t)));
    h_out_v = pack(t, h_out.data(), HASH_BITS).v;
    take(b6, t);
    splice(cs, b6);
  }
  cs.alloc_io(uX[1]);
  cs.rows += 1;
  cs.alloc_io(h_out_v);
  cs.rows += 1;
  cs.resolve();
  if (trace) fprintf(stderr, "synth side %d: late start %.0f  challenge %.0f  foreign %.0f  fold_w %.0f [prepared %.0f r %.0f inverses %.0f witness %.0f] fold_e %.0f  joined %.0f  step done %.0f  end %.0f us\n",
                     side, tr[0], tr[1], tr[2], fo_w.done, fo_w.tr[0], fo_w.tr[1], fo_w.tr[2], fo_w.tr[3], fo_e.done, tr[3], tr[4], us());
  (void)params; (void)Ty; (void)uWy; (void)T_inf;
  g_last_queue = fo_w.queued + fo_e.queued;
  g_last_misses = fo_w.misses + fo_e.misses;
  std::vector<Fe> out;
  for (const Num& n : z_out) out.push_back(n.v);
  return out;
}

std::vector<Fe> synthesize_augmented(CS& cs, int side, const AugInputs& in, const StepCircuit& step, Fe* unew_out, uint64_t* r_out,
                                     AugEarly* early) {
  static const bool sequential = [] { const char* e = env_override("VDF_NOVA_SEQ_SYNTH"); return e && e[0] == '1'; }();
  if (cs.ro != (in.ro ? in.ro : ro_default())) throw std::runtime_error("synthesize_augmented: the constraint system and the inputs name different RO parameter blocks");
  if (!cs.shape && !sequential) return synthesize_augmented_blocks(cs, side, in, step, unew_out, r_out, early);
  const Field& F = cs.F;
  const Field& PF = field(side_field(1 - side));
  const size_t a = step.arity();
  const Num one_n = cs.constant(one(F));
  const Num params = cs.alloc(in.params);
  const Num i = cs.alloc(in.i);
  std::vector<Num> z0, zi;
  for (size_t k = 0; k < a; ++k) z0.push_back(cs.alloc(in.z0[k]));
  for (size_t k = 0; k < a; ++k) zi.push_back(cs.alloc(in.zi[k]));
  Fe ue[9];
  relaxed_elements(in.U, F, ue);
  std::vector<Num> U;
  for (int k = 0; k < 9; ++k) U.push_back(cs.alloc(ue[k]));
  const Num uWx = cs.alloc(in.u_W.x), uWy = cs.alloc(in.u_W.y);
  const Num uX[2] = {cs.alloc(int_to_fe(in.u_X[0], F)), cs.alloc(int_to_fe(in.u_X[1], F))};
  const Num Tx = cs.alloc(in.T.x), Ty = cs.alloc(in.T.y);
  const Num is_base = is_zero(cs, i);
  // the hash this step must have been handed (checked unless i = 0)
  std::vector<Num> hin = {params, i};
  hin.insert(hin.end(), z0.begin(), z0.end());
  hin.insert(hin.end(), zi.begin(), zi.end());
  hin.insert(hin.end(), U.begin(), U.end());
  const std::vector<Num> h_in = strict_bits(cs, poseidon_hash(cs, TAG_STATE, hin));
  cs.enforce(cs.sub(one_n, is_base), cs.sub(uX[0], pack(cs, h_in.data(), HASH_BITS)), cs.zero_num());
  // fold challenge
  std::vector<Num> hch = {params};
  hch.insert(hch.end(), U.begin(), U.end());
  hch.insert(hch.end(), {uWx, uWy, uX[0], uX[1], Tx, Ty});
  std::vector<Num> r_bits = strict_bits(cs, poseidon_hash(cs, TAG_CHAL, hch));
  r_bits.resize(CHAL_BITS);
  const Num r = pack(cs, r_bits.data(), CHAL_BITS);
  // the two fresh points are on the curve (or the identity)
  const Num uW_inf = is_zero(cs, uWx);
  check_on_curve(cs, uWx, uWy, uW_inf);
  const Num T_inf = is_zero(cs, Tx);
  check_on_curve(cs, Tx, Ty, T_inf);
  if (!cs.shape) {
    uint64_t rv[4];
    fe_to_int(r.v, F, rv);
    cs.inv_queue.clear();
    cs.inv_pos = 0;
    ec_fold_inverses(F, in.U.comm_W, in.u_W, rv, CHAL_BITS, &cs.inv_queue);
    ec_fold_inverses(F, in.U.comm_E, in.T, rv, CHAL_BITS, &cs.inv_queue);
  }
  // comm_W' = U.W + r u.W ; comm_E' = U.E + r T
  Num rWx, rWy, fWx, fWy, rTx, rTy, fEx, fEy;
  ec_scalar_mul(cs, r_bits, uWx, uWy, uW_inf, &rWx, &rWy);
  ec_add_complete(cs, U[0], U[1], rWx, rWy, &fWx, &fWy);
  ec_scalar_mul(cs, r_bits, Tx, Ty, T_inf, &rTx, &rTy);
  ec_add_complete(cs, U[2], U[3], rTx, rTy, &fEx, &fEy);
  const Num fu = cs.add(U[4], r);
  // X' = X + r x in the other field
  std::vector<Num> xb[2];
  for (int k = 0; k < 2; ++k) xb[k] = alloc_bits(cs, in.u_X[k], HASH_BITS);
  for (int k = 0; k < 2; ++k) cs.enforce_equal(pack(cs, xb[k].data(), HASH_BITS), uX[k]);
  Num f0lo, f0hi, f1lo, f1hi;
  fold_foreign(cs, U[5], U[6], xb[0], r_bits, PF, &f0lo, &f0hi);
  fold_foreign(cs, U[7], U[8], xb[1], r_bits, PF, &f1lo, &f1hi);
  const Num fold[9] = {fWx, fWy, fEx, fEy, fu, f0lo, f0hi, f1lo, f1hi};
  Num base[9];
  if (side == 0) {
    for (int k = 0; k < 9; ++k) base[k] = cs.zero_num();
  } else {
    base[0] = uWx; base[1] = uWy; base[2] = cs.zero_num(); base[3] = cs.zero_num(); base[4] = one_n;
    base[5] = pack(cs, xb[0].data(), LIMB_BITS); base[6] = pack(cs, xb[0].data() + LIMB_BITS, HASH_BITS - LIMB_BITS);
    base[7] = pack(cs, xb[1].data(), LIMB_BITS); base[8] = pack(cs, xb[1].data() + LIMB_BITS, HASH_BITS - LIMB_BITS);
  }
  std::vector<Num> Unew;
  for (int k = 0; k < 9; ++k) Unew.push_back(select(cs, is_base, base[k], fold[k]));
  std::vector<Num> z_in;
  for (size_t k = 0; k < a; ++k) z_in.push_back(select(cs, is_base, z0[k], zi[k]));
  cs.step_begin = cs.num_vars();
  const std::vector<Num> z_out = step.synthesize(cs, z_in);
  cs.step_end = cs.num_vars();
  if (unew_out) for (int k = 0; k < 9; ++k) unew_out[k] = Unew[k].v;
  if (r_out) fe_to_int(r.v, F, r_out);
  const Num i_new = cs.add(i, one_n);
  std::vector<Num> hout = {params, i_new};
  hout.insert(hout.end(), z0.begin(), z0.end());
  hout.insert(hout.end(), z_out.begin(), z_out.end());
  hout.insert(hout.end(), Unew.begin(), Unew.end());
  const std::vector<Num> h_out = strict_bits(cs, poseidon_hash(cs, TAG_STATE, hout));
  const Num x0 = cs.alloc_io(uX[1].v);
  cs.enforce_equal(x0, uX[1]);
  const Num hv = pack(cs, h_out.data(), HASH_BITS);
  const Num x1 = cs.alloc_io(hv.v);
  cs.enforce_equal(x1, hv);
  cs.resolve();
  g_last_queue = cs.inv_queue.size();
  g_last_misses = cs.inv_misses + (cs.inv_queue.size() - cs.inv_pos);       // wrong or unused entries
  std::vector<Fe> out;
  for (const Num& n : z_out) out.push_back(n.v);
  return out;
}
void last_synthesis_stats(uint64_t* queued, uint64_t* misses) { *queued = g_last_queue; *misses = g_last_misses; }

}  // namespace vdfnova
