// libvdf_nova.so: the reference crate's MinRoot + Nova proof surface (include/vdf_nova.h) on top of
// the kernel ABI (include/vdf_hip.h).  See the header for the stage this implements.
#include <chrono>
#include <future>
#include <cstdio>
#include <array>
#include <memory>
#include <initializer_list>
#include <string>
#include <vector>
#include "../../../include/vdf_nova.h"
#include "host_math.hpp"

using namespace vdfhost;

namespace {

thread_local std::string g_err;
int fail(int code, const std::string& msg) { g_err = msg; return code; }
#define HIPCALL(ctx, expr)                                                          \
  do {                                                                              \
    int rc__ = (expr);                                                              \
    if (rc__ != VDF_OK) return fail(rc__, std::string(#expr) + ": " + vdf_last_error(ctx)); \
  } while (0)

double now_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// ---------------------------------------------------------------------------------------------
// MinRoot (src/minroot.rs)
// ---------------------------------------------------------------------------------------------
const uint64_t FP_RESCUE_INVALPHA[4] = {0xe0f0f3f0cccccccdull, 0x4e9ee0c9a10a60e2ull, 0x3333333333333333ull,
                                        0x3333333333333333ull};   // src/minroot.rs:273-278
const uint64_t FQ_RESCUE_INVALPHA[4] = {0xd69f2280cccccccdull, 0x4e9ee0c9a143ba4aull, 0x3333333333333333ull,
                                        0x3333333333333333ull};   // src/minroot.rs:280-285

struct Chain {   // the closures of src/minroot.rs:89-92 / :224-227
  const Field& F;
  Fe sq(Fe x, int n) const { for (int i = 0; i < n; ++i) x = sqr(x, F); return x; }
  Fe sqr_mul(const Fe& x, int n, const Fe& y) const { return mul(y, sq(x, n), F); }
};

// PallasVDF::forward_step_ltr_addition_chain, src/minroot.rs:88-127
Fe fwd_ltr_addchain_fq(const Fe& x) {
  const Field& F = field_fq();
  Chain c{F};
  Fe q1 = x, q10 = c.sq(q1, 1), q11 = mul(q10, q1, F), q101 = mul(q10, q11, F), q110 = c.sq(q11, 1);
  Fe q111 = mul(q110, q1, F), q1001 = mul(q111, q10, F), q1111 = mul(q1001, q110, F);
  Fe qr2 = c.sqr_mul(q110, 3, q11), qr4 = c.sqr_mul(qr2, 8, qr2), qr8 = c.sqr_mul(qr4, 16, qr4);
  Fe qr16 = c.sqr_mul(qr8, 32, qr8), qr32 = c.sqr_mul(qr16, 64, qr16);
  Fe v = c.sqr_mul(qr32, 5, q1001);
  struct { int n; const Fe* y; } steps[] = {{8, &q111}, {4, &q1}, {2, &qr4}, {7, &q11}, {6, &q1001}, {3, &q101},
      {7, &q101}, {7, &q111}, {4, &q111}, {5, &q1001}, {5, &q101}, {3, &q11}, {4, &q101}, {3, &q101}, {6, &q1111},
      {4, &q1001}, {6, &q101}, {37, &qr8}, {2, &q1}};
  for (auto& s : steps) v = c.sqr_mul(v, s.n, *s.y);
  return v;
}
// PallasVDF::forward_step_rtl_sequential, src/minroot.rs:130-151
Fe fwd_rtl_fq(const Fe& x) {
  const Field& F = field_fq();
  Fe acc = one(F), s = x;
  for (int count = 0; count < 254; ++count) {
    if ((FQ_RESCUE_INVALPHA[count / 64] >> (count % 64)) & 1) acc = mul(acc, s, F);
    s = sqr(s, F);
  }
  return acc;
}
// PallasVDF::forward_step_sequential_rtl_addition_chain, src/minroot.rs:154-196
Fe fwd_rtl_addchain_fq(const Fe& x) {
  const Field& F = field_fq();
  Fe acc = one(F), s = x, last = x;
  for (int count = 0; count < 128; ++count) {
    last = s;
    if ((FQ_RESCUE_INVALPHA[count / 64] >> (count % 64)) & 1) acc = mul(acc, s, F);
    s = sqr(s, F);
  }
  Fe sa = last;
  sa = mul(sa, sqr(sa, F), F);                                   // :179
  sa = mul(sa, sqr(sqr(sqr(sqr(sa, F), F), F), F), F);           // :180
  for (int count = 1; count <= 122; ++count) {                   // :182-195
    sa = sqr(sa, F);
    if (count % 8 == 1) acc = mul(acc, sa, F);
  }
  return acc;
}
// VestaVDF::forward_step, src/minroot.rs:223-261
Fe fwd_addchain_fp(const Fe& x) {
  const Field& F = field_fp();
  Chain c{F};
  Fe p1 = x, p10 = c.sq(p1, 1), p11 = mul(p10, p1, F), p101 = mul(p10, p11, F), p110 = c.sq(p11, 1);
  Fe p111 = mul(p110, p1, F), p1001 = mul(p111, p10, F), p1111 = mul(p1001, p110, F);
  Fe pr2 = c.sqr_mul(p110, 3, p11), pr4 = c.sqr_mul(pr2, 8, pr2), pr8 = c.sqr_mul(pr4, 16, pr4);
  Fe pr16 = c.sqr_mul(pr8, 32, pr8), pr32 = c.sqr_mul(pr16, 64, pr16);
  Fe v = c.sqr_mul(pr32, 5, p1001);
  struct { int n; const Fe* y; } steps[] = {{8, &p111}, {4, &p1}, {2, &pr4}, {7, &p11}, {6, &p1001}, {3, &p101},
      {5, &p1}, {7, &p101}, {4, &p11}, {8, &p111}, {4, &p1}, {4, &p111}, {9, &p1111}, {8, &p1111}, {6, &p1111},
      {2, &p11}, {34, &pr8}, {2, &p1}};
  for (auto& s : steps) v = c.sqr_mul(v, s.n, *s.y);
  return v;
}

// dispatch of src/minroot.rs:77-84; VestaVDF ignores the mode (:203-205)
Fe forward_step(int field_id, int mode, const Fe& x) {
  if (field_id == VDF_FIELD_FP) return fwd_addchain_fp(x);
  switch (mode) {
    case VDF_MODE_LTR_SEQUENTIAL: return pow_vartime(x, FQ_RESCUE_INVALPHA, field_fq());     // :312-314
    case VDF_MODE_LTR_ADDCHAIN_SEQUENTIAL: return fwd_ltr_addchain_fq(x);
    case VDF_MODE_RTL_SEQUENTIAL: return fwd_rtl_fq(x);
    default: return fwd_rtl_addchain_fq(x);
  }
}
Fe inverse_step(const Fe& x, const Field& F) { return mul(x, sqr(sqr(x, F), F), F); }            // :73-75

struct St { Fe x, y, i; };
St load_state(const vdf_state* s) { St r; memcpy(&r, s, sizeof(St)); return r; }
void store_state(vdf_state* o, const St& s) { memcpy(o, &s, sizeof(St)); }

St round_fwd(int f, int mode, const St& s) {                                                     // :329-335
  const Field& F = field(f);
  St r;
  r.x = forward_step(f, mode, add(s.x, s.y, F));
  r.y = add(s.x, s.i, F);
  r.i = add(s.i, one(F), F);
  return r;
}
St round_inv(int f, const St& s) {                                                               // :338-344
  const Field& F = field(f);
  St r;
  r.i = sub(s.i, one(F), F);
  r.x = sub(s.y, r.i, F);
  r.y = sub(inverse_step(s.x, F), r.x, F);
  return r;
}
bool valid_field(int f) { return f == VDF_FIELD_FP || f == VDF_FIELD_FQ; }
bool valid_mode(int m) { return m >= 0 && m <= 3; }

// ---------------------------------------------------------------------------------------------
// Nova (folding-only stage)
// ---------------------------------------------------------------------------------------------
constexpr int NUM_IO = 6;                  // X = [z_in(3), z_out(3)]
constexpr uint64_t GENS_SEED = 0x4e6f7661; // "Nova": label of the generator family
// Generators by seeded try-and-increment (include/vdf_hip.h): nobody knows their discrete logarithms, which is what
// makes the Pedersen commitments binding -- the [k_i]G family of the kernel tests would not do for a proof system.
constexpr int GENS_FAMILY = VDF_GENS_TRY_AND_INCREMENT;
constexpr int PRIMARY_FIELD = VDF_FIELD_FQ;   // S1 = pallas::Scalar, src/nova/proof.rs:29
constexpr int PRIMARY_CURVE = VDF_CURVE_PALLAS;   // G1, src/nova/proof.rs:26

struct StepRecord { Aff comm_w, comm_T; Fe r; Fe X[NUM_IO]; };

}  // namespace

struct vdf_pp {
  vdf_ctx* ctx = nullptr;
  uint64_t t = 0;
  size_t num_cons = 0, num_vars = 0, ncols = 0, nnz3 = 0, num_gens = 0;
  vdf_shape* shape = nullptr;
  vdf_bases* gens = nullptr;
  uint8_t digest[32];
  Aff gen_u;                // the extra generator U of the inner-product arguments: synthetic generator number num_gens
  void* d_zero = nullptr;   // num_cons zero elements (satisfiability residual)
};

struct Circuit {            // InverseMinRootCircuit<G1>, src/nova/proof.rs:57-66, + the forward trace
  uint64_t inverse_exponent = 5;
  St result, input;
  uint64_t t = 0;
  std::vector<Fe> trace_xy;  // (x, y) of states 0..t: trace[0] = input, trace[t] = result
  void* d_trace = nullptr;   // the same trace in HBM (vdf_nova_circuits_upload)
};
struct vdf_circuits { std::vector<Circuit> v; vdf_ctx* ctx = nullptr; };

struct vdf_proof {
  vdf_pp* pp = nullptr;
  size_t i = 0;              // steps folded so far
  Fe zi[3];                  // current z_i (starts at z0)
  Aff comm_W, comm_E;        // running relaxed instance
  Fe u, X[NUM_IO];
  void* d_z1 = nullptr;      // [W | u | X] of the running instance (W aliases the front)
  void* d_z2 = nullptr;      // [W | 1 | X] of the fresh instance
  void* d_E = nullptr;       // running error vector
  void* d_T = nullptr;
  void* d_abc[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // Az1,Bz1,Cz1,Az2,Bz2,Cz2
  void* d_trace = nullptr;
  vdf_jac* h_comm = nullptr; // pinned, device-mapped result slots: [0] = commitment of W2, [1] = commitment of T
  std::vector<StepRecord> steps;
  double ms[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  // The O(1) instance fold of step k (two 128-bit scalar multiplications on the host) is deferred: step k+1
  // performs it while the GPU works on its commitments; everything else that reads comm_W / comm_E joins first.
  struct Deferred { bool valid = false; Aff cW0, cE0, cw, cT; uint64_t r[4]; };
  mutable Deferred pending;
  void join() const;
};

namespace {

void absorb_fe(Shake256& h, const Fe& a, const Field& F) { Fe c = from_mont(a, F); h.absorb(c.l, 32); }
void absorb_aff(Shake256& h, const Aff& p) {
  const Field& F = field_fp();    // Pallas coordinates live in Fp
  absorb_fe(h, p.x, F);
  absorb_fe(h, p.y, F);
}
// r = SHAKE256(digest | U1 | u2 | comm_T) squeezed to 128 bits
Fe challenge(const vdf_pp* pp, const Aff& cW, const Aff& cE, const Fe& u, const Fe* X, const Aff& cw2, const Fe* X2,
             const Aff& cT, uint64_t r_raw[4]) {
  const Field& F = field(PRIMARY_FIELD);
  Shake256 h;
  h.absorb("vdf-nova-fold-v1", 16);
  h.absorb(pp->digest, 32);
  absorb_aff(h, cW); absorb_aff(h, cE); absorb_fe(h, u, F);
  for (int k = 0; k < NUM_IO; ++k) absorb_fe(h, X[k], F);
  absorb_aff(h, cw2);
  for (int k = 0; k < NUM_IO; ++k) absorb_fe(h, X2[k], F);
  absorb_aff(h, cT);
  r_raw[0] = r_raw[1] = r_raw[2] = r_raw[3] = 0;
  h.squeeze(r_raw, 16);
  Fe r;
  memcpy(r.l, r_raw, 32);
  return to_mont(r, F);
}

// Builds the COO triples of the wrapped step circuit; same layout as the test oracle (step_circuit_shape)
// (constraint order of src/nova/proof.rs:176-178, :219-227, :128-133, then the six IO-binding rows).
struct Coo { std::vector<uint32_t> rows, cols; std::vector<Fe> vals; };
void build_shape(uint64_t t, Coo m[3], size_t* num_cons, size_t* num_vars) {
  const Field& F = field(PRIMARY_FIELD);
  const Fe ONE = one(F), MINUS_ONE = neg(ONE, F);
  const uint32_t nv = (uint32_t)(3 + 4 * t + 1), one_col = nv, io = nv + 1;
  auto push = [](Coo& c, uint32_t r, uint32_t col, const Fe& v) { c.rows.push_back(r); c.cols.push_back(col); c.vals.push_back(v); };
  uint32_t xv = 0, yv = 1, iv = 2, row = 0;
  for (uint64_t j = 0; j < t; ++j) {
    const uint32_t base = (uint32_t)(3 + 4 * j), new_x = base, tmp1 = base + 1, tmp2 = base + 2, new_y = base + 3;
    push(m[0], row, xv, ONE); push(m[1], row, xv, ONE); push(m[2], row, tmp1, ONE); ++row;          // x*x = tmp1
    push(m[0], row, tmp1, ONE); push(m[1], row, tmp1, ONE); push(m[2], row, tmp2, ONE); ++row;      // tmp1*tmp1 = tmp2
    push(m[0], row, tmp2, ONE); push(m[1], row, xv, ONE);                                           // tmp2*x = new_y + y - i + 1
    push(m[2], row, new_y, ONE); push(m[2], row, yv, ONE); push(m[2], row, iv, MINUS_ONE);
    push(m[2], row, one_col, from_u64(j + 1, F)); ++row;                                            // i = z_in.i - j
    xv = new_x; yv = new_y;
  }
  const uint32_t final_i = (uint32_t)(3 + 4 * t);
  push(m[0], row, final_i, ONE); push(m[1], row, one_col, ONE);                                     // final_i*1 = i - t
  push(m[2], row, iv, ONE); push(m[2], row, one_col, neg(from_u64(t, F), F)); ++row;
  const uint32_t outs[6] = {0, 1, 2, xv, yv, final_i};
  for (int k = 0; k < 6; ++k) {
    push(m[0], row, outs[k], ONE); push(m[1], row, one_col, ONE); push(m[2], row, io + k, ONE); ++row;
  }
  *num_cons = row;
  *num_vars = nv;
}

int alloc_proof_buffers(vdf_proof* p) {
  vdf_pp* pp = p->pp;
  vdf_ctx* ctx = pp->ctx;
  HIPCALL(ctx, vdf_dev_alloc(ctx, pp->ncols * 32, &p->d_z1));
  HIPCALL(ctx, vdf_dev_alloc(ctx, pp->ncols * 32, &p->d_z2));
  HIPCALL(ctx, vdf_dev_alloc(ctx, pp->num_cons * 32, &p->d_E));
  HIPCALL(ctx, vdf_dev_alloc(ctx, pp->num_cons * 32, &p->d_T));
  for (int k = 0; k < 6; ++k) HIPCALL(ctx, vdf_dev_alloc(ctx, pp->num_cons * 32, &p->d_abc[k]));
  HIPCALL(ctx, vdf_dev_alloc(ctx, (pp->t + 1) * 64, &p->d_trace));
  HIPCALL(ctx, vdf_host_alloc(ctx, 2 * sizeof(vdf_jac), (void**)&p->h_comm));
  HIPCALL(ctx, vdf_dev_memset(ctx, p->d_E, 0, pp->num_cons * 32));
  return VDF_OK;
}

Aff fold_commitment(const Aff& a, const uint64_t r_raw[4], const Aff& b) {      // a + r*b on Pallas
  const Field& F = field_fp();
  Pt rb = pt_mul(pt_from_aff(b, F), r_raw, 128, F);
  return pt_to_aff(pt_add(pt_from_aff(a, F), rb, F), F);
}

}  // namespace

void vdf_proof::join() const {
  if (!pending.valid) return;
  vdf_proof* self = const_cast<vdf_proof*>(this);
  self->comm_W = fold_commitment(pending.cW0, pending.r, pending.cw);
  self->comm_E = fold_commitment(pending.cE0, pending.r, pending.cT);
  pending.valid = false;
}

extern "C" {

const char* vdf_nova_last_error(void) { return g_err.c_str(); }

// ---- MinRoot -------------------------------------------------------------------------------------
int vdf_minroot_forward_step(int f, int mode, const vdf_fe* x, vdf_fe* out) {
  if (!valid_field(f) || !valid_mode(mode) || !x || !out) return fail(VDF_ERR_BAD_ARG, "bad argument");
  Fe a; memcpy(&a, x, 32);
  Fe r = forward_step(f, mode, a);
  memcpy(out, &r, 32);
  return VDF_OK;
}
int vdf_minroot_inverse_step(int f, const vdf_fe* x, vdf_fe* out) {
  if (!valid_field(f) || !x || !out) return fail(VDF_ERR_BAD_ARG, "bad argument");
  Fe a; memcpy(&a, x, 32);
  Fe r = inverse_step(a, field(f));
  memcpy(out, &r, 32);
  return VDF_OK;
}
int vdf_minroot_round(int f, int mode, const vdf_state* s, vdf_state* out) {
  if (!valid_field(f) || !valid_mode(mode) || !s || !out) return fail(VDF_ERR_BAD_ARG, "bad argument");
  store_state(out, round_fwd(f, mode, load_state(s)));
  return VDF_OK;
}
int vdf_minroot_inverse_round(int f, const vdf_state* s, vdf_state* out) {
  if (!valid_field(f) || !s || !out) return fail(VDF_ERR_BAD_ARG, "bad argument");
  store_state(out, round_inv(f, load_state(s)));
  return VDF_OK;
}
// The sequential loop itself, compiled twice: baseline x86-64 and a BMI2/ADX (Broadwell and later, Zen) clone chosen
// by the dynamic loader, with the field arithmetic flattened into it.  ~285 dependent multiplications per round.
__attribute__((target_clones("default", "arch=broadwell"), flatten, noinline))
void eval_rounds(int f, int mode, St* acc, uint64_t t, vdf_fe* trace_xy) {
  for (uint64_t k = 0; k < t; ++k) {                               // simple_eval, :352-359
    *acc = round_fwd(f, mode, *acc);
    if (trace_xy) { memcpy(&trace_xy[2 * (k + 1)], &acc->x, 32); memcpy(&trace_xy[2 * (k + 1) + 1], &acc->y, 32); }
  }
}
int vdf_minroot_eval(int f, int mode, const vdf_state* s, uint64_t t, vdf_state* out, vdf_fe* trace_xy) {
  if (!valid_field(f) || !valid_mode(mode) || !s || !out) return fail(VDF_ERR_BAD_ARG, "bad argument");
  St acc = load_state(s);
  if (trace_xy) { memcpy(&trace_xy[0], &acc.x, 32); memcpy(&trace_xy[1], &acc.y, 32); }
  eval_rounds(f, mode, &acc, t, trace_xy);
  store_state(out, acc);
  return VDF_OK;
}
int vdf_minroot_inverse_eval(int f, const vdf_state* s, uint64_t t, vdf_state* out) {
  if (!valid_field(f) || !s || !out) return fail(VDF_ERR_BAD_ARG, "bad argument");
  St acc = load_state(s);
  for (uint64_t k = 0; k < t; ++k) acc = round_inv(f, acc);        // :363-365
  store_state(out, acc);
  return VDF_OK;
}
int vdf_minroot_check(int f, const vdf_state* result, uint64_t t, const vdf_state* original) {
  vdf_state back;
  if (vdf_minroot_inverse_eval(f, result, t, &back) != VDF_OK || !original) return 0;
  return memcmp(&back, original, sizeof(back)) == 0;               // :369-371
}
int vdf_minroot_element(int f, uint64_t n, vdf_fe* out) {
  if (!valid_field(f) || !out) return fail(VDF_ERR_BAD_ARG, "bad argument");
  Fe r = from_u64(n, field(f));
  memcpy(out, &r, 32);
  return VDF_OK;
}

// ---- public parameters -------------------------------------------------------------------------------
int vdf_nova_public_params(vdf_ctx* ctx, uint64_t t, vdf_pp** out) {
  if (!ctx || !out || t == 0 || t > (1ull << 24)) return fail(VDF_ERR_BAD_ARG, "bad argument");
  *out = nullptr;
  vdf_pp* pp = new vdf_pp();
  pp->ctx = ctx;
  pp->t = t;
  Coo m[3];
  build_shape(t, m, &pp->num_cons, &pp->num_vars);
  pp->ncols = pp->num_vars + 1 + NUM_IO;
  pp->nnz3 = m[0].rows.size() + m[1].rows.size() + m[2].rows.size();
  const uint32_t* rows[3] = {m[0].rows.data(), m[1].rows.data(), m[2].rows.data()};
  const uint32_t* cols[3] = {m[0].cols.data(), m[1].cols.data(), m[2].cols.data()};
  const vdf_fe* vals[3] = {(const vdf_fe*)m[0].vals.data(), (const vdf_fe*)m[1].vals.data(), (const vdf_fe*)m[2].vals.data()};
  const size_t nnz[3] = {m[0].rows.size(), m[1].rows.size(), m[2].rows.size()};
  int rc = vdf_shape_create(ctx, PRIMARY_FIELD, pp->num_cons, pp->ncols, rows, cols, vals, nnz, &pp->shape);
  if (rc != VDF_OK) { std::string e = vdf_last_error(ctx); delete pp; return fail(rc, "vdf_shape_create: " + e); }
  size_t need = pp->num_vars > pp->num_cons ? pp->num_vars : pp->num_cons;
  size_t g = 1;
  while (g < need) g <<= 1;                                        // next_pow2(max(vars, cons)), SURVEY.md App. C
  pp->num_gens = g;
  rc = vdf_bases_generate_family(ctx, PRIMARY_CURVE, GENS_FAMILY, GENS_SEED, 0, g, &pp->gens);
  if (rc == VDF_OK) rc = vdf_bases_precompute(ctx, pp->gens, 16, 1);
  if (rc == VDF_OK) {
    vdf_bases* ub = nullptr;
    rc = vdf_bases_generate_family(ctx, PRIMARY_CURVE, GENS_FAMILY, GENS_SEED, g, 1, &ub);
    if (rc == VDF_OK) rc = vdf_bases_download(ctx, ub, 0, 1, (vdf_affine*)&pp->gen_u);
    if (ub) vdf_bases_free(ub);
  }
  if (rc == VDF_OK) rc = vdf_dev_alloc(ctx, pp->num_cons * 32, &pp->d_zero);
  if (rc == VDF_OK) rc = vdf_dev_memset(ctx, pp->d_zero, 0, pp->num_cons * 32);
  if (rc != VDF_OK) { std::string e = vdf_last_error(ctx); vdf_nova_pp_free(pp); return fail(rc, "generator setup: " + e); }
  // shape digest: sizes, every COO triple in canonical form, generator family
  Shake256 h;
  h.absorb("vdf-nova-shape-v1", 17);
  uint64_t hdr[6] = {t, (uint64_t)pp->num_cons, (uint64_t)pp->num_vars, (uint64_t)NUM_IO, GENS_SEED, (uint64_t)GENS_FAMILY};
  h.absorb(hdr, sizeof(hdr));
  const Field& F = field(PRIMARY_FIELD);
  for (int k = 0; k < 3; ++k)
    for (size_t e = 0; e < m[k].rows.size(); ++e) {
      uint32_t rc2[2] = {m[k].rows[e], m[k].cols[e]};
      h.absorb(rc2, 8);
      absorb_fe(h, m[k].vals[e], F);
    }
  h.squeeze(pp->digest, 32);
  *out = pp;
  return VDF_OK;
}
void vdf_nova_pp_free(vdf_pp* pp) {
  if (!pp) return;
  if (pp->d_zero) vdf_dev_free(pp->ctx, pp->d_zero);
  if (pp->shape) vdf_shape_free(pp->shape);
  if (pp->gens) vdf_bases_free(pp->gens);
  delete pp;
}
int vdf_nova_pp_sizes(const vdf_pp* pp, uint64_t* num_cons, uint64_t* num_vars, uint64_t* num_io, uint64_t* nnz3,
                      uint64_t* num_gens) {
  if (!pp) return fail(VDF_ERR_BAD_ARG, "null pp");
  if (num_cons) *num_cons = pp->num_cons;
  if (num_vars) *num_vars = pp->num_vars;
  if (num_io) *num_io = NUM_IO;
  if (nnz3) *nnz3 = pp->nnz3;
  if (num_gens) *num_gens = pp->num_gens;
  return VDF_OK;
}

// ---- circuits ----------------------------------------------------------------------------------------
int vdf_nova_eval_and_make_circuits(int mode, uint64_t t, size_t num_steps, const vdf_state* initial_state,
                                    vdf_fe z0_primary[3], vdf_circuits** out) {
  if (!valid_mode(mode) || !initial_state || !z0_primary || !out || t == 0) return fail(VDF_ERR_BAD_ARG, "bad argument");
  if (num_steps == 0) return fail(VDF_ERR_BAD_ARG, "num_steps must be > 0 (assert!, src/nova/proof.rs:268)");
  vdf_circuits* cs = new vdf_circuits();
  St state = load_state(initial_state);
  for (size_t s = 0; s < num_steps; ++s) {                          // :274-279
    Circuit c;
    c.t = t;
    c.input = state;                                                 // previous_state, :285-291
    c.trace_xy.resize(2 * (t + 1));
    vdf_state res;
    vdf_state in;
    store_state(&in, state);
    vdf_minroot_eval(PRIMARY_FIELD, mode, &in, t, &res, (vdf_fe*)c.trace_xy.data());
    c.result = load_state(&res);
    state = c.result;
    cs->v.push_back(std::move(c));
  }
  memcpy(z0_primary, &state, 96);                                    // z0 = final state, :278-281
  std::vector<Circuit> rev(cs->v.rbegin(), cs->v.rend());            // circuits.reverse(), :294
  cs->v.swap(rev);
  *out = cs;
  return VDF_OK;
}
int vdf_nova_circuits_upload(vdf_ctx* ctx, vdf_circuits* c) {
  if (!ctx || !c) return fail(VDF_ERR_BAD_ARG, "null argument");
  c->ctx = ctx;
  for (auto& k : c->v) {
    if (k.d_trace) continue;
    HIPCALL(ctx, vdf_dev_alloc(ctx, k.trace_xy.size() * 32, &k.d_trace));
    HIPCALL(ctx, vdf_dev_memcpy(ctx, k.d_trace, k.trace_xy.data(), k.trace_xy.size() * 32));
  }
  return VDF_OK;
}
size_t vdf_nova_circuits_len(const vdf_circuits* c) { return c ? c->v.size() : 0; }
int vdf_nova_circuit_states(const vdf_circuits* c, size_t k, vdf_state* result, vdf_state* input) {
  if (!c || k >= c->v.size()) return fail(VDF_ERR_BAD_LENGTH, "circuit index out of range");
  if (result) store_state(result, c->v[k].result);
  if (input) store_state(input, c->v[k].input);
  return VDF_OK;
}
void vdf_nova_circuits_free(vdf_circuits* c) {
  if (!c) return;
  if (c->ctx) for (auto& k : c->v) if (k.d_trace) vdf_dev_free(c->ctx, k.d_trace);
  delete c;
}

// ---- prove_step ----------------------------------------------------------------------------------------
static int prove_step_impl(vdf_pp* pp, vdf_proof** proof, const vdf_circuits* circuits, size_t k, const vdf_fe z0[3],
                           vdf_proof** fresh);

int vdf_nova_prove_step(vdf_pp* pp, vdf_proof** proof, const vdf_circuits* circuits, size_t k, const vdf_fe z0[3]) {
  vdf_proof* fresh = nullptr;                     // a proof object this call created (the `None` case)
  const int rc = prove_step_impl(pp, proof, circuits, k, z0, &fresh);
  if (rc != VDF_OK && fresh) vdf_nova_proof_free(fresh);      // never leak a half-built proof
  return rc;
}

static int prove_step_impl(vdf_pp* pp, vdf_proof** proof, const vdf_circuits* circuits, size_t k, const vdf_fe z0[3],
                           vdf_proof** fresh) {
  if (!pp || !proof || !circuits || !z0) return fail(VDF_ERR_BAD_ARG, "null argument");
  if (k >= circuits->v.size()) return fail(VDF_ERR_BAD_LENGTH, "circuit index out of range");
  const Circuit& c = circuits->v[k];
  if (c.t != pp->t) return fail(VDF_ERR_BAD_LENGTH, "circuit t differs from the public parameters");
  vdf_ctx* ctx = pp->ctx;
  const Field& F = field(PRIMARY_FIELD);
  vdf_proof* p = *proof;
  const bool first = (p == nullptr);
  if (first) {
    p = new vdf_proof();
    *fresh = p;
    p->pp = pp;
    memcpy(p->zi, z0, 96);
    int rc = alloc_proof_buffers(p);
    if (rc != VDF_OK) return rc;
  }
  // StepCircuit::output's debug assertion: z_i must be the circuit's result (src/nova/proof.rs:147-149)
  if (memcmp(p->zi, &c.result, 96) != 0)
    return fail(VDF_ERR_BAD_ARG, "z_i does not match the circuit's result state");
  const double t0 = now_ms();
  const size_t nv = pp->num_vars, nc = pp->num_cons;
  // The step is enqueued asynchronously: every call below is stream-ordered, and the host waits only for the
  // two commitments the transcript needs.  The caller's synchronisation mode is restored on the way out.
  int was_async = 0;
  HIPCALL(ctx, vdf_ctx_get_async(ctx, &was_async));
  HIPCALL(ctx, vdf_ctx_set_async(ctx, 1));
  struct Restore { vdf_ctx* c; int a; ~Restore() { if (!a) { vdf_ctx_sync(c); vdf_ctx_set_async(c, 0); } } } restore{ctx, was_async};
  // --- fresh z2 = [z_in | per-round new_x, tmp1, tmp2, new_y | final_i | 1 | X2] in one launch ------------
  const void* d_trace = c.d_trace;
  if (!d_trace) {
    HIPCALL(ctx, vdf_dev_memcpy(ctx, p->d_trace, c.trace_xy.data(), (pp->t + 1) * 64));
    d_trace = p->d_trace;
  }
  Fe X2[NUM_IO] = {c.result.x, c.result.y, c.result.i, c.input.x, c.input.y, c.input.i};
  {
    const Fe head[3] = {c.result.x, c.result.y, c.result.i};
    const Fe u2 = one(F);
    HIPCALL(ctx, vdf_minroot_step_z(ctx, PRIMARY_FIELD, (const vdf_fe*)d_trace, pp->t, (const vdf_fe*)head,
                                    (const vdf_fe*)&c.input.i, (const vdf_fe*)&u2, (const vdf_fe*)X2, (vdf_fe*)p->d_z2));
  }
  const double t1 = now_ms();
  vdf_jac* jw = &p->h_comm[0];
  vdf_jac* jt = &p->h_comm[1];
  Aff comm_w;
  StepRecord rec;
  for (int j = 0; j < NUM_IO; ++j) rec.X[j] = X2[j];
  double t2 = t1, t3 = t1, t4 = t1, t5 = t1, t6 = t1;
  if (first) {
    // running := fresh as a relaxed instance (E = 0, u = 1); the `None` case of prove_step.  A z, B z, C z of
    // the running instance are computed here once and folded from then on (they are linear in z).
    HIPCALL(ctx, vdf_msm(ctx, pp->gens, 0, (const vdf_fe*)p->d_z2, nv, 1, jw));
    HIPCALL(ctx, vdf_dev_memcpy(ctx, p->d_z1, p->d_z2, pp->ncols * 32));
    HIPCALL(ctx, vdf_spmv3(ctx, pp->shape, (const vdf_fe*)p->d_z1, (vdf_fe*)p->d_abc[0], (vdf_fe*)p->d_abc[1], (vdf_fe*)p->d_abc[2]));
    HIPCALL(ctx, vdf_ctx_sync(ctx));
    t2 = now_ms();
    comm_w = jac_to_aff(*jw, field_fp());
    p->comm_W = comm_w;
    p->comm_E.x = p->comm_E.y = zero();
    p->u = one(F);
    for (int j = 0; j < NUM_IO; ++j) p->X[j] = X2[j];
    rec.comm_T.x = rec.comm_T.y = zero();
    rec.r = zero();
    t6 = now_ms();
  } else {
    // --- NIFS.prove: multiply_vec(z2) + cross term (one launch), then the commitments to
    // W2 and T as ONE batched MSM whose two points land in pinned host memory.  While the GPU works the host
    // finishes the previous step's instance fold.
    HIPCALL(ctx, vdf_nifs_cross_term(ctx, pp->shape, (const vdf_fe*)p->d_z2, (const vdf_fe*)p->d_abc[0], (const vdf_fe*)p->d_abc[1],
                                     (const vdf_fe*)p->d_abc[2], (const vdf_fe*)&p->u, (vdf_fe*)p->d_abc[3], (vdf_fe*)p->d_abc[4],
                                     (vdf_fe*)p->d_abc[5], (vdf_fe*)p->d_T));
    t2 = now_ms();
    {
      const size_t off[2] = {0, 0}, len[2] = {nv, nc};
      const vdf_fe* sc[2] = {(const vdf_fe*)p->d_z2, (const vdf_fe*)p->d_T};
      HIPCALL(ctx, vdf_msm_batch(ctx, pp->gens, 2, off, sc, len, 1, p->h_comm));   // h_comm[0] = W2, [1] = T
    }
    t3 = t4 = now_ms();
    p->join();                                                   // the previous step's instance fold
    HIPCALL(ctx, vdf_ctx_sync(ctx));
    t5 = now_ms();
    Aff comm_T;
    jac_to_aff2(*jw, *jt, field_fp(), &comm_w, &comm_T);         // one shared inversion
    uint64_t r_raw[4];
    const Fe r = challenge(pp, p->comm_W, p->comm_E, p->u, p->X, comm_w, X2, comm_T, r_raw);
    // witness fold on the device, one launch: z1 += r*z2 (W, and with it u and X), E += r*T, and the running
    // A z, B z, C z += r * (A z2, B z2, C z2)
    {
      vdf_fe* acc[5] = {(vdf_fe*)p->d_z1, (vdf_fe*)p->d_E, (vdf_fe*)p->d_abc[0], (vdf_fe*)p->d_abc[1], (vdf_fe*)p->d_abc[2]};
      const vdf_fe* add[5] = {(const vdf_fe*)p->d_z2, (const vdf_fe*)p->d_T, (const vdf_fe*)p->d_abc[3],
                              (const vdf_fe*)p->d_abc[4], (const vdf_fe*)p->d_abc[5]};
      const size_t len[5] = {pp->ncols, nc, nc, nc, nc};
      HIPCALL(ctx, vdf_fold_many(ctx, PRIMARY_FIELD, (const vdf_fe*)&r, 5, acc, add, len));
    }
    t6 = now_ms();
    // instance fold on the host (O(1)): u and X now, the two commitments deferred (vdf_proof::join)
    p->pending.cW0 = p->comm_W; p->pending.cE0 = p->comm_E;
    p->pending.cw = comm_w; p->pending.cT = comm_T;
    memcpy(p->pending.r, r_raw, 32);
    p->pending.valid = true;
    p->u = add(p->u, r, F);
    for (int j = 0; j < NUM_IO; ++j) p->X[j] = add(p->X[j], mul(r, X2[j], F), F);
    rec.comm_T = comm_T;
    rec.r = r;
  }
  rec.comm_w = comm_w;
  p->steps.push_back(rec);
  p->i += 1;
  p->zi[0] = c.input.x; p->zi[1] = c.input.y; p->zi[2] = c.input.i;   // c1.output(zi), src/nova/proof.rs:142-152
  const double t7 = now_ms();
  // witness launch | commitments launch (batched) | cross-term launch | - | host fold of the previous step +
  // wait for both commitments | transcript + fold launch | bookkeeping | total
  p->ms[0] = t1 - t0; p->ms[1] = t3 - t2; p->ms[2] = t2 - t1; p->ms[3] = t4 - t3;
  p->ms[4] = t5 - t4; p->ms[5] = t6 - t5; p->ms[6] = t7 - t6; p->ms[7] = t7 - t0;
  *proof = p;
  *fresh = nullptr;                               // handed over to the caller
  return VDF_OK;
}

int vdf_nova_prove_recursively(vdf_pp* pp, const vdf_circuits* circuits, uint64_t num_iters_per_step, const vdf_fe z0[3],
                               vdf_proof** out) {
  if (!pp || !circuits || !out || !z0) return fail(VDF_ERR_BAD_ARG, "null argument");
  if (num_iters_per_step != pp->t) return fail(VDF_ERR_BAD_LENGTH, "num_iters_per_step differs from the public parameters");
  if (circuits->v.empty()) return fail(VDF_ERR_BAD_LENGTH, "no circuits (recursive_snark.unwrap(), src/nova/proof.rs:357)");
  vdf_proof* p = nullptr;
  for (size_t k = 0; k < circuits->v.size(); ++k) {                   // :318-355
    int rc = vdf_nova_prove_step(pp, &p, circuits, k, z0);
    if (rc != VDF_OK) { vdf_nova_proof_free(p); *out = nullptr; return rc; }
  }
  *out = p;
  return VDF_OK;
}

void vdf_nova_proof_free(vdf_proof* p) {
  if (!p) return;
  p->join();
  vdf_ctx* ctx = p->pp ? p->pp->ctx : nullptr;
  if (ctx) {
    void* bufs[] = {p->d_z1, p->d_z2, p->d_E, p->d_T, p->d_abc[0], p->d_abc[1], p->d_abc[2], p->d_abc[3], p->d_abc[4],
                    p->d_abc[5], p->d_trace};
    for (void* b : bufs) if (b) vdf_dev_free(ctx, b);
    if (p->h_comm) vdf_host_free(ctx, p->h_comm);
  }
  delete p;
}
size_t vdf_nova_proof_num_steps(const vdf_proof* p) { return p ? p->i : 0; }

int vdf_nova_proof_instance(const vdf_proof* p, vdf_affine* comm_W, vdf_affine* comm_E, vdf_fe* u, vdf_fe X[6]) {
  if (!p) return fail(VDF_ERR_BAD_ARG, "null proof");
  p->join();
  if (comm_W) memcpy(comm_W, &p->comm_W, 64);
  if (comm_E) memcpy(comm_E, &p->comm_E, 64);
  if (u) memcpy(u, &p->u, 32);
  if (X) memcpy(X, p->X, 32 * NUM_IO);
  return VDF_OK;
}
int vdf_nova_proof_witness_ptrs(const vdf_proof* p, const void** d_W, const void** d_E) {
  if (!p) return fail(VDF_ERR_BAD_ARG, "null proof");
  HIPCALL(p->pp->ctx, vdf_ctx_sync(p->pp->ctx));      // a step may have returned with its fold still in flight
  if (d_W) *d_W = p->d_z1;
  if (d_E) *d_E = p->d_E;
  return VDF_OK;
}
int vdf_nova_proof_step_record(const vdf_proof* p, size_t k, vdf_affine* comm_w, vdf_affine* comm_T, vdf_fe* r, vdf_fe X[6]) {
  if (!p || k >= p->steps.size()) return fail(VDF_ERR_BAD_LENGTH, "step index out of range");
  const StepRecord& s = p->steps[k];
  if (comm_w) memcpy(comm_w, &s.comm_w, 64);
  if (comm_T) memcpy(comm_T, &s.comm_T, 64);
  if (r) memcpy(r, &s.r, 32);
  if (X) memcpy(X, s.X, 32 * NUM_IO);
  return VDF_OK;
}
int vdf_nova_last_step_ms(const vdf_proof* p, double ms[8]) {
  if (!p || !ms) return fail(VDF_ERR_BAD_ARG, "null argument");
  memcpy(ms, p->ms, sizeof(p->ms));
  return VDF_OK;
}

// ---- verify --------------------------------------------------------------------------------------------
int vdf_nova_verify(const vdf_proof* p, vdf_pp* pp, size_t num_steps, const vdf_fe z0[3], const vdf_fe zi[3], int* ok) {
  if (!p || !pp || !z0 || !zi || !ok) return fail(VDF_ERR_BAD_ARG, "null argument");
  *ok = 0;
  if (p->pp != pp) return fail(VDF_ERR_BAD_ARG, "proof was made under other public parameters");
  p->join();
  vdf_ctx* ctx = pp->ctx;
  const Field& F = field(PRIMARY_FIELD);
  if (num_steps == 0 || p->steps.size() != num_steps || p->i != num_steps) return VDF_OK;   // NovaError::ProofVerifyError
  // (1) public-IO chain: X_0.z_in = z0, X_k.z_out = X_{k+1}.z_in; the last z_out is the verified z_i
  if (memcmp(p->steps[0].X, z0, 96) != 0) return VDF_OK;
  for (size_t k = 0; k + 1 < num_steps; ++k)
    if (memcmp(&p->steps[k].X[3], &p->steps[k + 1].X[0], 96) != 0) return VDF_OK;
  // every step must move the counter by exactly t in the inverse direction
  const Fe tfe = from_u64(pp->t, F);
  for (size_t k = 0; k < num_steps; ++k)
    if (sub(p->steps[k].X[2], p->steps[k].X[5], F) != tfe) return VDF_OK;
  // (2) replay the folds of the instances
  Aff cW = p->steps[0].comm_w, cE;
  cE.x = cE.y = zero();
  Fe u = one(F), X[NUM_IO];
  for (int j = 0; j < NUM_IO; ++j) X[j] = p->steps[0].X[j];
  for (size_t k = 1; k < num_steps; ++k) {
    const StepRecord& s = p->steps[k];
    uint64_t r_raw[4];
    const Fe r = challenge(pp, cW, cE, u, X, s.comm_w, s.X, s.comm_T, r_raw);
    if (r != s.r) return VDF_OK;
    cW = fold_commitment(cW, r_raw, s.comm_w);
    cE = fold_commitment(cE, r_raw, s.comm_T);
    u = add(u, r, F);
    for (int j = 0; j < NUM_IO; ++j) X[j] = add(X[j], mul(r, s.X[j], F), F);
  }
  if (memcmp(&cW, &p->comm_W, 64) || memcmp(&cE, &p->comm_E, 64) || u != p->u || memcmp(X, p->X, sizeof(X))) return VDF_OK;
  // (3) the running witness opens the folded instance: commitments and relaxed satisfiability
  const size_t nv = pp->num_vars, nc = pp->num_cons;
  vdf_jac j1, j2;
  HIPCALL(ctx, vdf_msm(ctx, pp->gens, 0, (const vdf_fe*)p->d_z1, nv, 1, &j1));
  HIPCALL(ctx, vdf_msm(ctx, pp->gens, 0, (const vdf_fe*)p->d_E, nc, 1, &j2));
  Aff a1 = jac_to_aff(j1, field_fp()), a2 = jac_to_aff(j2, field_fp());
  if (memcmp(&a1, &p->comm_W, 64) || memcmp(&a2, &p->comm_E, 64)) return VDF_OK;
  // z = (W, u, X) must carry the instance's u and X
  std::vector<Fe> tail(1 + NUM_IO);
  HIPCALL(ctx, vdf_dev_memcpy(ctx, tail.data(), (const char*)p->d_z1 + nv * 32, tail.size() * 32));
  if (tail[0] != p->u || memcmp(&tail[1], p->X, 32 * NUM_IO)) return VDF_OK;
  HIPCALL(ctx, vdf_spmv3(ctx, pp->shape, (const vdf_fe*)p->d_z1, (vdf_fe*)p->d_abc[0], (vdf_fe*)p->d_abc[1], (vdf_fe*)p->d_abc[2]));
  // residual Az*Bz - u*Cz - E through the cross-term kernel with Az2 = Bz1 = 0
  HIPCALL(ctx, vdf_cross_term(ctx, PRIMARY_FIELD, (const vdf_fe*)p->d_abc[0], (const vdf_fe*)pp->d_zero, (const vdf_fe*)p->d_E,
                              (const vdf_fe*)pp->d_zero, (const vdf_fe*)p->d_abc[1], (const vdf_fe*)p->d_abc[2],
                              (const vdf_fe*)&p->u, nc, (vdf_fe*)p->d_T));
  std::vector<uint64_t> res(nc * 4);
  HIPCALL(ctx, vdf_dev_memcpy(ctx, res.data(), p->d_T, nc * 32));
  uint64_t any = 0;
  for (uint64_t w : res) any |= w;
  if (any) return VDF_OK;
  // Ok(zi_primary == zi_primary_verified), src/nova/proof.rs:386
  *ok = memcmp(&p->steps[num_steps - 1].X[3], zi, 96) == 0 ? 1 : 0;
  return VDF_OK;
}

}  // extern "C"

// =============================================================================================================
// Compression SNARK: NovaVDFProof::compress / verification of the compressed proof (src/nova/proof.rs:360-368, :383).
// Protocol "vdf-spartan-v1", restated line by line by the test oracle (spartan.py: prove / verify); every pass over a vector is
// a call through include/vdf_hip.h, the host keeps the transcript, O(log n) field work and O(log n) point work.
// =============================================================================================================
namespace {

struct Transcript {
  uint8_t state[32];
  explicit Transcript(const char* label) {
    Shake256 h;
    h.absorb("vdf-spartan-v1|", 15);
    h.absorb(label, strlen(label));
    h.squeeze(state, 32);
  }
  void absorb(const char* label, const void* data, size_t n) {
    Shake256 h;
    h.absorb(state, 32);
    h.absorb(label, strlen(label));
    h.absorb(":", 1);
    h.absorb(data, n);
    h.squeeze(state, 32);
  }
  void absorb_fe(const char* label, const Fe* v, size_t k, const Field& F) {
    std::vector<uint8_t> b(k * 32);
    for (size_t i = 0; i < k; ++i) { const Fe c = from_mont(v[i], F); memcpy(&b[i * 32], c.l, 32); }
    absorb(label, b.data(), b.size());
  }
  void absorb_pt(const char* label, const Aff* p, size_t k) {
    const Field& F = field_fp();
    std::vector<uint8_t> b(k * 64, 0);
    for (size_t i = 0; i < k; ++i)
      if (!p[i].is_id()) {
        const Fe x = from_mont(p[i].x, F), y = from_mont(p[i].y, F);
        memcpy(&b[i * 64], x.l, 32); memcpy(&b[i * 64 + 32], y.l, 32);
      }
    absorb(label, b.data(), b.size());
  }
  // 128-bit challenge: raw = the integer, return = its Montgomery form in F
  Fe challenge(const char* label, const Field& F, uint64_t raw[4]) {
    Shake256 h;
    h.absorb(state, 32);
    h.absorb(label, strlen(label));
    h.absorb("?", 1);
    uint8_t out[48];
    h.squeeze(out, 48);
    memcpy(state, out + 16, 32);
    raw[0] = raw[1] = raw[2] = raw[3] = 0;
    memcpy(raw, out, 16);
    Fe r;
    memcpy(r.l, raw, 32);
    return to_mont(r, F);
  }
};

struct Ipa { std::vector<Aff> L, R; Fe a; };
struct Spartan {
  std::vector<std::array<Fe, 3>> outer;
  Fe claims[4];
  std::vector<std::array<Fe, 2>> inner;
  Fe w_eval;
  Ipa ipaW, ipaE;
};

size_t pow2_at_least(size_t n) { size_t p = 1; while (p < n) p <<= 1; return p; }
int log2_exact(size_t n) { int k = 0; while (((size_t)1 << k) < n) ++k; return k; }

struct Layout { size_t M, NW, Z; int s, l1; };
Layout layout_of(const vdf_pp* pp) {
  Layout l;
  l.M = pow2_at_least(pp->num_cons); l.NW = pow2_at_least(pp->num_vars); l.Z = 2 * l.NW;
  l.s = log2_exact(l.M); l.l1 = log2_exact(l.Z);
  return l;
}

// device buffers released on scope exit
struct DevBufs {
  vdf_ctx* ctx;
  std::vector<void*> v;
  explicit DevBufs(vdf_ctx* c) : ctx(c) {}
  ~DevBufs() { for (void* p : v) vdf_dev_free(ctx, p); }
  int zeros(size_t elems, void** out) {
    int rc = vdf_dev_alloc(ctx, elems * 32, out);
    if (rc != VDF_OK) return rc;
    v.push_back(*out);
    return vdf_dev_memset(ctx, *out, 0, elems * 32);
  }
};

Fe small(uint64_t k, const Field& F) { return from_u64(k, F); }
// value at r of the polynomial through (0, y0), (1, y1), (2, y2)[, (3, y3)]
Fe interpolate(const Fe* y, int npts, const Fe& r, const Field& F) {
  Fe acc = zero();
  for (int i = 0; i < npts; ++i) {
    Fe num = one(F), den = one(F);
    for (int j = 0; j < npts; ++j)
      if (i != j) {
        num = mul(num, sub(r, small(j, F), F), F);
        den = mul(den, sub(small(i, F), small(j, F), F), F);
      }
    acc = add(acc, mul(y[i], mul(num, inverse(den, F), F), F), F);
  }
  return acc;
}

void instance_bytes(Transcript& tr, const vdf_pp* pp, const Aff& cW, const Aff& cE, const Fe& u, const Fe* X) {
  const Field& F = field(PRIMARY_FIELD);
  tr.absorb("shape", pp->digest, 32);
  const Aff pts[2] = {cW, cE};
  tr.absorb_pt("inst", pts, 2);
  Fe v[1 + NUM_IO];
  v[0] = u;
  for (int j = 0; j < NUM_IO; ++j) v[1 + j] = X[j];
  tr.absorb_fe("inst", v, 1 + NUM_IO, F);
}

// lo = 1 - r, hi = r table of eq(r, .) on the device
int eq_table_dev(vdf_ctx* ctx, const std::vector<Fe>& r, void* out) {
  const Field& F = field(PRIMARY_FIELD);
  std::vector<Fe> lo(r.size());
  for (size_t j = 0; j < r.size(); ++j) lo[j] = sub(one(F), r[j], F);
  HIPCALL(ctx, vdf_pair_table(ctx, PRIMARY_FIELD, (const vdf_fe*)lo.data(), (const vdf_fe*)r.data(), (int)r.size(), (vdf_fe*)out));
  return VDF_OK;
}

// M(y) in the padded layout (W at [0, NW), u at NW, X after it) from eq(r_x, .)
int m_vector_dev(vdf_pp* pp, const Layout& L, const void* d_eq_rx, const Fe& rho, void* d_cols, void* d_mvec) {
  vdf_ctx* ctx = pp->ctx;
  HIPCALL(ctx, vdf_spmv3_t(ctx, pp->shape, (const vdf_fe*)d_eq_rx, (const vdf_fe*)&rho, (vdf_fe*)d_cols));
  HIPCALL(ctx, vdf_dev_memset(ctx, d_mvec, 0, L.Z * 32));
  HIPCALL(ctx, vdf_dev_memcpy(ctx, d_mvec, d_cols, pp->num_vars * 32));
  HIPCALL(ctx, vdf_dev_memcpy(ctx, (char*)d_mvec + L.NW * 32, (const char*)d_cols + pp->num_vars * 32, (1 + NUM_IO) * 32));
  return VDF_OK;
}

Pt pt_mul_fe(const Pt& p, const Fe& k_mont, const Field& Fscalar) {       // k in Montgomery form of the scalar field
  const Fe k = from_mont(k_mont, Fscalar);
  return pt_mul(p, k.l, 255, field_fp());
}

// Fixed-base multiples of one point (the Q of an inner-product argument is multiplied by two fresh scalars per
// round): 4-bit windows, d * 16^w * Q for d = 1..15, so a product is 64 additions and no doubling.
struct FixedBase {
  std::vector<Pt> tab;            // [w * 15 + (d - 1)]
  explicit FixedBase(const Pt& q) : tab(64 * 15) {
    const Field& Fb = field_fp();
    Pt base = q;
    for (int w = 0; w < 64; ++w) {
      Pt acc = base;
      for (int d = 1; d <= 15; ++d) { tab[w * 15 + d - 1] = acc; acc = pt_add(acc, base, Fb); }
      base = acc;                  // 16 * base
    }
  }
  Pt mul(const Fe& k_mont, const Field& Fscalar) const {
    const Field& Fb = field_fp();
    const Fe k = from_mont(k_mont, Fscalar);
    Pt r = pt_identity();
    for (int w = 0; w < 64; ++w) {
      const unsigned d = (unsigned)(k.l[w / 16] >> (4 * (w % 16))) & 15u;
      if (d) r = pt_add(r, tab[w * 15 + d - 1], Fb);
    }
    return r;
  }
};

// <a, b> = v under P = <a, G[0..n)>; a, b, s: device vectors of length n that this function consumes (s = all ones)
int ipa_prove(vdf_pp* pp, Transcript& tr, const char* label, size_t n, void* d_a, void* d_b, void* d_s, void* d_sL, void* d_sR,
              const Fe& v, const Aff& P, vdf_jac* h_lr, Ipa* out) {
  vdf_ctx* ctx = pp->ctx;
  const Field& F = field(PRIMARY_FIELD);
  const Field& Fb = field_fp();
  tr.absorb_pt(label, &P, 1);
  tr.absorb_fe(label, &v, 1, F);
  uint64_t raw[4];
  tr.challenge(label, F, raw);
  const Pt Qp = pt_mul(pt_from_aff(pp->gen_u, Fb), raw, 128, Fb);
  const FixedBase Qtab(Qp);
  out->L.clear(); out->R.clear();
  for (size_t nj = n; nj > 1; nj >>= 1) {
    Fe cross[2];
    const vdf_fe* ab[2] = {(const vdf_fe*)d_a, (const vdf_fe*)d_b};
    HIPCALL(ctx, vdf_reduce(ctx, PRIMARY_FIELD, VDF_REDUCE_IPA_CROSS, ab, nullptr, nj, (vdf_fe*)cross));
    HIPCALL(ctx, vdf_ipa_scalars(ctx, PRIMARY_FIELD, (const vdf_fe*)d_a, (const vdf_fe*)d_s, n, nj, (vdf_fe*)d_sL, (vdf_fe*)d_sR));
    const size_t off[2] = {0, 0}, len[2] = {n, n};
    const vdf_fe* sc[2] = {(const vdf_fe*)d_sL, (const vdf_fe*)d_sR};
    HIPCALL(ctx, vdf_msm_batch(ctx, pp->gens, 2, off, sc, len, 1, h_lr));
    HIPCALL(ctx, vdf_ctx_sync(ctx));
    Aff l0, r0;
    jac_to_aff2(h_lr[0], h_lr[1], Fb, &l0, &r0);
    const Aff Lp = pt_to_aff(pt_add(pt_from_aff(l0, Fb), Qtab.mul(cross[0], F), Fb), Fb);
    const Aff Rp = pt_to_aff(pt_add(pt_from_aff(r0, Fb), Qtab.mul(cross[1], F), Fb), Fb);
    const Aff lr[2] = {Lp, Rp};
    tr.absorb_pt(label, lr, 2);
    const Fe x = tr.challenge(label, F, raw);
    const Fe xi = inverse(x, F);
    vdf_fe* vecs[2] = {(vdf_fe*)d_a, (vdf_fe*)d_b};
    const Fe c_lo[2] = {x, xi}, c_hi[2] = {xi, x};
    HIPCALL(ctx, vdf_fold_halves(ctx, PRIMARY_FIELD, 2, vecs, (const vdf_fe*)c_lo, (const vdf_fe*)c_hi, nj));
    HIPCALL(ctx, vdf_scale_pattern(ctx, PRIMARY_FIELD, (vdf_fe*)d_s, n, nj, (const vdf_fe*)&xi, (const vdf_fe*)&x));
    out->L.push_back(Lp); out->R.push_back(Rp);
  }
  HIPCALL(ctx, vdf_ctx_sync(ctx));
  HIPCALL(ctx, vdf_dev_memcpy(ctx, &out->a, d_a, 32));
  return VDF_OK;
}

// b is eq(rb, .): its fold is a closed form; the coefficient vector of the folded generator is a tensor-product table
int ipa_verify(vdf_pp* pp, Transcript& tr, const char* label, size_t n, const std::vector<Fe>& rb, const Fe& v, const Aff& P,
               const Ipa& proof, void* d_s, bool* ok) {
  vdf_ctx* ctx = pp->ctx;
  const Field& F = field(PRIMARY_FIELD);
  const Field& Fb = field_fp();
  *ok = false;
  const size_t k = proof.L.size();
  if (((size_t)1 << k) != n || proof.R.size() != k || rb.size() != k) return VDF_OK;
  tr.absorb_pt(label, &P, 1);
  tr.absorb_fe(label, &v, 1, F);
  uint64_t raw[4];
  tr.challenge(label, F, raw);
  const Pt Qp = pt_mul(pt_from_aff(pp->gen_u, Fb), raw, 128, Fb);
  Pt acc = pt_add(pt_from_aff(P, Fb), pt_mul_fe(Qp, v, F), Fb);
  std::vector<Fe> xs(k), xis(k);
  Fe bfin = one(F);
  for (size_t j = 0; j < k; ++j) {
    const Aff lr[2] = {proof.L[j], proof.R[j]};
    tr.absorb_pt(label, lr, 2);
    const Fe x = tr.challenge(label, F, raw);
    if (x.is_zero()) return VDF_OK;
    const Fe xi = inverse(x, F);
    acc = pt_add(acc, pt_add(pt_mul_fe(pt_from_aff(proof.L[j], Fb), sqr(x, F), F),
                             pt_mul_fe(pt_from_aff(proof.R[j], Fb), sqr(xi, F), F), Fb), Fb);
    bfin = mul(bfin, add(mul(sub(one(F), rb[j], F), xi, F), mul(rb[j], x, F), F), F);
    xs[j] = x; xis[j] = xi;
  }
  HIPCALL(ctx, vdf_pair_table(ctx, PRIMARY_FIELD, (const vdf_fe*)xis.data(), (const vdf_fe*)xs.data(), (int)k, (vdf_fe*)d_s));
  vdf_jac jg;
  HIPCALL(ctx, vdf_msm(ctx, pp->gens, 0, (const vdf_fe*)d_s, n, 1, &jg));
  const Pt gf = pt_from_aff(jac_to_aff(jg, Fb), Fb);
  const Pt rhs = pt_add(pt_mul_fe(gf, proof.a, F), pt_mul_fe(Qp, mul(proof.a, bfin, F), F), Fb);
  const Aff a1 = pt_to_aff(acc, Fb), a2 = pt_to_aff(rhs, Fb);
  *ok = memcmp(&a1, &a2, sizeof(Aff)) == 0;
  return VDF_OK;
}

int spartan_prove(vdf_pp* pp, const Aff& cW, const Aff& cE, const Fe& u, const Fe* X, const void* d_z, const void* d_E,
                  Spartan* out) {
  vdf_ctx* ctx = pp->ctx;
  const Field& F = field(PRIMARY_FIELD);
  const Layout L = layout_of(pp);
  if (L.l1 > 24 || L.s > 24) return fail(VDF_ERR_BAD_LENGTH, "shape too large for the compression SNARK (2^24 entries)");
  if (L.NW > pp->num_gens) return fail(VDF_ERR_BAD_LENGTH, "not enough generators");
  const size_t nv = pp->num_vars, nc = pp->num_cons;
  DevBufs bufs(ctx);
  void *d_eq, *d_az, *d_bz, *d_cz, *d_e, *d_cols, *d_mvec, *d_zpad, *d_w, *d_s, *d_sL, *d_sR;
  for (void** p : {&d_eq, &d_az, &d_bz, &d_cz, &d_e}) { int rc = bufs.zeros(L.M, p); if (rc != VDF_OK) return fail(rc, vdf_last_error(ctx)); }
  for (void** p : {&d_mvec, &d_zpad}) { int rc = bufs.zeros(L.Z, p); if (rc != VDF_OK) return fail(rc, vdf_last_error(ctx)); }
  for (void** p : {&d_w, &d_s, &d_sL, &d_sR}) { int rc = bufs.zeros(L.NW, p); if (rc != VDF_OK) return fail(rc, vdf_last_error(ctx)); }
  { int rc = bufs.zeros(pp->ncols, &d_cols); if (rc != VDF_OK) return fail(rc, vdf_last_error(ctx)); }
  vdf_jac* h_lr = nullptr;
  HIPCALL(ctx, vdf_host_alloc(ctx, 2 * sizeof(vdf_jac), (void**)&h_lr));
  struct HostFree { vdf_ctx* c; void* p; ~HostFree() { vdf_host_free(c, p); } } hf{ctx, h_lr};

  Transcript tr("compress");
  instance_bytes(tr, pp, cW, cE, u, X);
  HIPCALL(ctx, vdf_spmv3(ctx, pp->shape, (const vdf_fe*)d_z, (vdf_fe*)d_az, (vdf_fe*)d_bz, (vdf_fe*)d_cz));
  HIPCALL(ctx, vdf_dev_memcpy(ctx, d_e, d_E, nc * 32));
  uint64_t raw[4];
  std::vector<Fe> tau(L.s);
  for (int j = 0; j < L.s; ++j) tau[j] = tr.challenge("tau", F, raw);
  { int rc = eq_table_dev(ctx, tau, d_eq); if (rc != VDF_OK) return rc; }
  // ---- outer sum-check -----------------------------------------------------------------------------------
  std::vector<Fe> rx;
  out->outer.clear();
  {
    const vdf_fe* tabs[5] = {(const vdf_fe*)d_eq, (const vdf_fe*)d_az, (const vdf_fe*)d_bz, (const vdf_fe*)d_cz, (const vdf_fe*)d_e};
    vdf_fe* vecs[5] = {(vdf_fe*)d_eq, (vdf_fe*)d_az, (vdf_fe*)d_bz, (vdf_fe*)d_cz, (vdf_fe*)d_e};
    for (size_t n = L.M; n > 1; n >>= 1) {
      std::array<Fe, 3> ev;
      HIPCALL(ctx, vdf_reduce(ctx, PRIMARY_FIELD, VDF_REDUCE_R1CS_ROUND, tabs, (const vdf_fe*)&u, n, (vdf_fe*)ev.data()));
      tr.absorb_fe("outer", ev.data(), 3, F);
      const Fe r = tr.challenge("outer", F, raw);
      const Fe omr = sub(one(F), r, F);
      const Fe c_lo[5] = {omr, omr, omr, omr, omr}, c_hi[5] = {r, r, r, r, r};
      HIPCALL(ctx, vdf_fold_halves(ctx, PRIMARY_FIELD, 5, vecs, (const vdf_fe*)c_lo, (const vdf_fe*)c_hi, n));
      out->outer.push_back(ev);
      rx.push_back(r);
    }
  }
  HIPCALL(ctx, vdf_ctx_sync(ctx));
  HIPCALL(ctx, vdf_dev_memcpy(ctx, &out->claims[0], d_az, 32));
  HIPCALL(ctx, vdf_dev_memcpy(ctx, &out->claims[1], d_bz, 32));
  HIPCALL(ctx, vdf_dev_memcpy(ctx, &out->claims[2], d_cz, 32));
  HIPCALL(ctx, vdf_dev_memcpy(ctx, &out->claims[3], d_e, 32));
  tr.absorb_fe("claims", out->claims, 4, F);
  const Fe rho = tr.challenge("rho", F, raw);
  // ---- inner sum-check -----------------------------------------------------------------------------------
  void* d_eq_rx = d_az;                                       // the outer tables are spent: reuse one as eq(r_x, .)
  { int rc = eq_table_dev(ctx, rx, d_eq_rx); if (rc != VDF_OK) return rc; }
  { int rc = m_vector_dev(pp, L, d_eq_rx, rho, d_cols, d_mvec); if (rc != VDF_OK) return rc; }
  HIPCALL(ctx, vdf_dev_memcpy(ctx, d_zpad, d_z, nv * 32));
  HIPCALL(ctx, vdf_dev_memcpy(ctx, (char*)d_zpad + L.NW * 32, (const char*)d_z + nv * 32, (1 + NUM_IO) * 32));
  HIPCALL(ctx, vdf_dev_memcpy(ctx, d_w, d_z, nv * 32));
  std::vector<Fe> ry;
  out->inner.clear();
  {
    const vdf_fe* tabs[2] = {(const vdf_fe*)d_mvec, (const vdf_fe*)d_zpad};
    vdf_fe* vecs[2] = {(vdf_fe*)d_mvec, (vdf_fe*)d_zpad};
    for (size_t n = L.Z; n > 1; n >>= 1) {
      std::array<Fe, 2> ev;
      HIPCALL(ctx, vdf_reduce(ctx, PRIMARY_FIELD, VDF_REDUCE_QUADRATIC_ROUND, tabs, nullptr, n, (vdf_fe*)ev.data()));
      tr.absorb_fe("inner", ev.data(), 2, F);
      const Fe r = tr.challenge("inner", F, raw);
      const Fe omr = sub(one(F), r, F);
      const Fe c_lo[2] = {omr, omr}, c_hi[2] = {r, r};
      HIPCALL(ctx, vdf_fold_halves(ctx, PRIMARY_FIELD, 2, vecs, (const vdf_fe*)c_lo, (const vdf_fe*)c_hi, n));
      out->inner.push_back(ev);
      ry.push_back(r);
    }
  }
  // ---- openings ----------------------------------------------------------------------------------------------
  void* d_eq_ry = d_mvec;                                     // spent: reuse for eq(r_y[1:], .) (NW entries)
  { std::vector<Fe> rest(ry.begin() + 1, ry.end()); int rc = eq_table_dev(ctx, rest, d_eq_ry); if (rc != VDF_OK) return rc; }
  {
    const vdf_fe* tabs[2] = {(const vdf_fe*)d_w, (const vdf_fe*)d_eq_ry};
    HIPCALL(ctx, vdf_reduce(ctx, PRIMARY_FIELD, VDF_REDUCE_DOT, tabs, nullptr, L.NW, (vdf_fe*)&out->w_eval));
  }
  tr.absorb_fe("weval", &out->w_eval, 1, F);
  std::vector<Fe> ones_lo(24, one(F));
  HIPCALL(ctx, vdf_pair_table(ctx, PRIMARY_FIELD, (const vdf_fe*)ones_lo.data(), (const vdf_fe*)ones_lo.data(), L.l1 - 1, (vdf_fe*)d_s));
  { int rc = ipa_prove(pp, tr, "ipaW", L.NW, d_w, d_eq_ry, d_s, d_sL, d_sR, out->w_eval, cW, h_lr, &out->ipaW); if (rc != VDF_OK) return rc; }
  // E: a = E padded to M (fresh copy: d_e was folded), b = eq(r_x, .)
  HIPCALL(ctx, vdf_dev_memset(ctx, d_e, 0, L.M * 32));
  HIPCALL(ctx, vdf_dev_memcpy(ctx, d_e, d_E, nc * 32));
  HIPCALL(ctx, vdf_pair_table(ctx, PRIMARY_FIELD, (const vdf_fe*)ones_lo.data(), (const vdf_fe*)ones_lo.data(), L.s, (vdf_fe*)d_s));
  { int rc = ipa_prove(pp, tr, "ipaE", L.M, d_e, d_eq_rx, d_s, d_sL, d_sR, out->claims[3], cE, h_lr, &out->ipaE); if (rc != VDF_OK) return rc; }
  return VDF_OK;
}

int spartan_verify(vdf_pp* pp, const Aff& cW, const Aff& cE, const Fe& u, const Fe* X, const Spartan& pf, bool* ok) {
  vdf_ctx* ctx = pp->ctx;
  const Field& F = field(PRIMARY_FIELD);
  const Layout L = layout_of(pp);
  *ok = false;
  if ((int)pf.outer.size() != L.s || (int)pf.inner.size() != L.l1) return VDF_OK;
  Transcript tr("compress");
  instance_bytes(tr, pp, cW, cE, u, X);
  uint64_t raw[4];
  std::vector<Fe> tau(L.s), rx, ry;
  for (int j = 0; j < L.s; ++j) tau[j] = tr.challenge("tau", F, raw);
  // outer
  Fe claim = zero();
  for (const auto& ev : pf.outer) {
    const Fe y[4] = {ev[0], sub(claim, ev[0], F), ev[1], ev[2]};
    tr.absorb_fe("outer", ev.data(), 3, F);
    const Fe r = tr.challenge("outer", F, raw);
    claim = interpolate(y, 4, r, F);
    rx.push_back(r);
  }
  Fe eq_tau = one(F);
  for (int j = 0; j < L.s; ++j)
    eq_tau = mul(eq_tau, add(mul(tau[j], rx[j], F), mul(sub(one(F), tau[j], F), sub(one(F), rx[j], F), F), F), F);
  const Fe a = pf.claims[0], b = pf.claims[1], c = pf.claims[2], e = pf.claims[3];
  if (claim != mul(eq_tau, sub(sub(mul(a, b, F), mul(u, c, F), F), e, F), F)) return VDF_OK;
  tr.absorb_fe("claims", pf.claims, 4, F);
  const Fe rho = tr.challenge("rho", F, raw);
  // inner
  claim = add(a, add(mul(rho, b, F), mul(sqr(rho, F), c, F), F), F);
  for (const auto& ev : pf.inner) {
    const Fe y[3] = {ev[0], sub(claim, ev[0], F), ev[1]};
    tr.absorb_fe("inner", ev.data(), 2, F);
    const Fe r = tr.challenge("inner", F, raw);
    claim = interpolate(y, 3, r, F);
    ry.push_back(r);
  }
  // M(r_y) on the device: eq(r_x, .) -> transposed product -> dot with eq(r_y, .)
  DevBufs bufs(ctx);
  void *d_eq_rx, *d_cols, *d_mvec, *d_eq_ry, *d_s;
  { int rc = bufs.zeros(L.M, &d_eq_rx); if (rc != VDF_OK) return fail(rc, vdf_last_error(ctx)); }
  { int rc = bufs.zeros(pp->ncols, &d_cols); if (rc != VDF_OK) return fail(rc, vdf_last_error(ctx)); }
  for (void** p : {&d_mvec, &d_eq_ry}) { int rc = bufs.zeros(L.Z, p); if (rc != VDF_OK) return fail(rc, vdf_last_error(ctx)); }
  { int rc = bufs.zeros(L.NW, &d_s); if (rc != VDF_OK) return fail(rc, vdf_last_error(ctx)); }
  { int rc = eq_table_dev(ctx, rx, d_eq_rx); if (rc != VDF_OK) return rc; }
  { int rc = m_vector_dev(pp, L, d_eq_rx, rho, d_cols, d_mvec); if (rc != VDF_OK) return rc; }
  { int rc = eq_table_dev(ctx, ry, d_eq_ry); if (rc != VDF_OK) return rc; }
  Fe m_ry;
  {
    const vdf_fe* tabs[2] = {(const vdf_fe*)d_mvec, (const vdf_fe*)d_eq_ry};
    HIPCALL(ctx, vdf_reduce(ctx, PRIMARY_FIELD, VDF_REDUCE_DOT, tabs, nullptr, L.Z, (vdf_fe*)&m_ry));
  }
  // z(r_y) = (1 - r_y[0]) W~(rest) + r_y[0] * (u, X)~(rest); index i of the public half has bits MSB-first over rest
  std::vector<Fe> rest(ry.begin() + 1, ry.end());
  const int k = (int)rest.size();
  Fe pub = zero();
  for (int i = 0; i < 1 + NUM_IO; ++i) {
    Fe w = one(F);
    for (int j = 0; j < k; ++j) w = mul(w, ((i >> (k - 1 - j)) & 1) ? rest[j] : sub(one(F), rest[j], F), F);
    pub = add(pub, mul(w, i == 0 ? u : X[i - 1], F), F);
  }
  const Fe z_ry = add(mul(sub(one(F), ry[0], F), pf.w_eval, F), mul(ry[0], pub, F), F);
  if (claim != mul(m_ry, z_ry, F)) return VDF_OK;
  tr.absorb_fe("weval", &pf.w_eval, 1, F);
  bool ok1 = false, ok2 = false;
  { int rc = ipa_verify(pp, tr, "ipaW", L.NW, rest, pf.w_eval, cW, pf.ipaW, d_s, &ok1); if (rc != VDF_OK) return rc; }
  if (!ok1) return VDF_OK;
  { int rc = ipa_verify(pp, tr, "ipaE", L.M, rx, e, cE, pf.ipaE, d_s, &ok2); if (rc != VDF_OK) return rc; }
  *ok = ok2;
  return VDF_OK;
}

}  // namespace

struct vdf_snark {          // NovaVDFProof::Compressed, src/nova/proof.rs:54
  std::vector<StepRecord> steps;
  Aff comm_W, comm_E;       // the folded instance the argument is about
  Fe u, X[NUM_IO];
  Spartan sp;
};

extern "C" {

int vdf_nova_compress(const vdf_proof* p, vdf_pp* pp, vdf_snark** out) {
  if (!p || !pp || !out) return fail(VDF_ERR_BAD_ARG, "null argument");
  *out = nullptr;
  if (p->pp != pp) return fail(VDF_ERR_BAD_ARG, "proof was made under other public parameters");
  if (p->steps.empty()) return fail(VDF_ERR_BAD_LENGTH, "nothing to compress");
  p->join();
  vdf_ctx* ctx = pp->ctx;
  int was_async = 0;
  HIPCALL(ctx, vdf_ctx_get_async(ctx, &was_async));
  HIPCALL(ctx, vdf_ctx_sync(ctx));
  HIPCALL(ctx, vdf_ctx_set_async(ctx, 1));
  struct Restore { vdf_ctx* c; int a; ~Restore() { vdf_ctx_sync(c); vdf_ctx_set_async(c, a); } } restore{ctx, was_async};
  std::unique_ptr<vdf_snark> s(new vdf_snark());
  s->steps = p->steps;
  s->comm_W = p->comm_W; s->comm_E = p->comm_E; s->u = p->u;
  for (int j = 0; j < NUM_IO; ++j) s->X[j] = p->X[j];
  int rc = spartan_prove(pp, s->comm_W, s->comm_E, s->u, s->X, p->d_z1, p->d_E, &s->sp);
  if (rc != VDF_OK) return rc;
  *out = s.release();
  return VDF_OK;
}

void vdf_nova_snark_free(vdf_snark* s) { delete s; }

// flat canonical encoding of the argument (little-endian, non-Montgomery): outer rounds (3 each), 4 claims, inner
// rounds (2 each), w, then per opening: (L, R) per round as affine (x, y), the final scalar
size_t vdf_nova_snark_size(const vdf_snark* s) {
  if (!s) return 0;
  const Spartan& p = s->sp;
  return 32 * (3 * p.outer.size() + 4 + 2 * p.inner.size() + 1 + 2) + 128 * (p.ipaW.L.size() + p.ipaE.L.size());
}

int vdf_nova_snark_bytes(const vdf_snark* s, uint8_t* out, size_t cap) {
  if (!s || !out) return fail(VDF_ERR_BAD_ARG, "null argument");
  if (cap < vdf_nova_snark_size(s)) return fail(VDF_ERR_BAD_LENGTH, "buffer too small");
  const Field& F = field(PRIMARY_FIELD);
  const Field& Fb = field_fp();
  uint8_t* o = out;
  auto put = [&](const Fe& v, const Field& f) { const Fe c = from_mont(v, f); memcpy(o, c.l, 32); o += 32; };
  auto put_pt = [&](const Aff& a) { if (a.is_id()) { memset(o, 0, 64); o += 64; } else { put(a.x, Fb); put(a.y, Fb); } };
  for (const auto& ev : s->sp.outer) for (const Fe& v : ev) put(v, F);
  for (const Fe& v : s->sp.claims) put(v, F);
  for (const auto& ev : s->sp.inner) for (const Fe& v : ev) put(v, F);
  put(s->sp.w_eval, F);
  for (const Ipa* ip : {&s->sp.ipaW, &s->sp.ipaE}) {
    for (size_t j = 0; j < ip->L.size(); ++j) { put_pt(ip->L[j]); put_pt(ip->R[j]); }
    put(ip->a, F);
  }
  return VDF_OK;
}

// replaces the argument by the given encoding (deserialisation; the tests use it to tamper)
int vdf_nova_snark_set_bytes(vdf_snark* s, const uint8_t* in, size_t len) {
  if (!s || !in) return fail(VDF_ERR_BAD_ARG, "null argument");
  if (len != vdf_nova_snark_size(s)) return fail(VDF_ERR_BAD_LENGTH, "encoding has the wrong length for this shape");
  const Field& F = field(PRIMARY_FIELD);
  const Field& Fb = field_fp();
  const uint8_t* i = in;
  bool canonical = true;
  auto get = [&](Fe& v, const Field& f) { Fe c; memcpy(c.l, i, 32); i += 32; if (geq(c.l, f.m)) canonical = false; v = to_mont(c, f); };
  auto get_pt = [&](Aff& a) { get(a.x, Fb); get(a.y, Fb); };
  for (auto& ev : s->sp.outer) for (Fe& v : ev) get(v, F);
  for (Fe& v : s->sp.claims) get(v, F);
  for (auto& ev : s->sp.inner) for (Fe& v : ev) get(v, F);
  get(s->sp.w_eval, F);
  for (Ipa* ip : {&s->sp.ipaW, &s->sp.ipaE}) {
    for (size_t j = 0; j < ip->L.size(); ++j) { get_pt(ip->L[j]); get_pt(ip->R[j]); }
    get(ip->a, F);
  }
  if (!canonical) return fail(VDF_ERR_NONCANONICAL, "a field element of the encoding is not canonical");
  return VDF_OK;
}

// verification of the compressed proof (src/nova/proof.rs:383): the fold replay of vdf_nova_verify without the
// witness, then the argument that the folded instance is satisfiable
int vdf_nova_verify_compressed(const vdf_snark* s, vdf_pp* pp, size_t num_steps, const vdf_fe z0[3], const vdf_fe zi[3], int* ok) {
  if (!s || !pp || !z0 || !zi || !ok) return fail(VDF_ERR_BAD_ARG, "null argument");
  *ok = 0;
  const Field& F = field(PRIMARY_FIELD);
  if (num_steps == 0 || s->steps.size() != num_steps) return VDF_OK;
  if (memcmp(s->steps[0].X, z0, 96) != 0) return VDF_OK;
  for (size_t k = 0; k + 1 < num_steps; ++k)
    if (memcmp(&s->steps[k].X[3], &s->steps[k + 1].X[0], 96) != 0) return VDF_OK;
  const Fe tfe = from_u64(pp->t, F);
  for (size_t k = 0; k < num_steps; ++k)
    if (sub(s->steps[k].X[2], s->steps[k].X[5], F) != tfe) return VDF_OK;
  Aff cW = s->steps[0].comm_w, cE;
  cE.x = cE.y = zero();
  Fe u = one(F), X[NUM_IO];
  for (int j = 0; j < NUM_IO; ++j) X[j] = s->steps[0].X[j];
  for (size_t k = 1; k < num_steps; ++k) {
    const StepRecord& st = s->steps[k];
    uint64_t r_raw[4];
    const Fe r = challenge(pp, cW, cE, u, X, st.comm_w, st.X, st.comm_T, r_raw);
    if (r != st.r) return VDF_OK;
    cW = fold_commitment(cW, r_raw, st.comm_w);
    cE = fold_commitment(cE, r_raw, st.comm_T);
    u = add(u, r, F);
    for (int j = 0; j < NUM_IO; ++j) X[j] = add(X[j], mul(r, st.X[j], F), F);
  }
  if (memcmp(&cW, &s->comm_W, 64) || memcmp(&cE, &s->comm_E, 64) || u != s->u || memcmp(X, s->X, sizeof(X))) return VDF_OK;
  vdf_ctx* ctx = pp->ctx;
  int was_async = 0;
  HIPCALL(ctx, vdf_ctx_get_async(ctx, &was_async));
  HIPCALL(ctx, vdf_ctx_sync(ctx));
  HIPCALL(ctx, vdf_ctx_set_async(ctx, 1));
  struct Restore { vdf_ctx* c; int a; ~Restore() { vdf_ctx_sync(c); vdf_ctx_set_async(c, a); } } restore{ctx, was_async};
  bool good = false;
  int rc = spartan_verify(pp, s->comm_W, s->comm_E, s->u, s->X, s->sp, &good);
  if (rc != VDF_OK) return rc;
  if (!good) return VDF_OK;
  *ok = memcmp(&s->steps[num_steps - 1].X[3], zi, 96) == 0 ? 1 : 0;
  return VDF_OK;
}

}  // extern "C"
