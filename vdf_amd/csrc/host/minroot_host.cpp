// libvdf_nova.so, part 1: the reference's `minroot` module (src/minroot.rs) -- the forward (slow) and inverse
// (fast) MinRoot rounds, the four forward-step chains, eval / check -- on the host, where the reference keeps them.
#include "nova_internal.hpp"

using namespace vdfnova;

namespace {

// ---------------------------------------------------------------------------------------------
// MinRoot (src/minroot.rs)
// ---------------------------------------------------------------------------------------------
const uint64_t FP_RESCUE_INVALPHA[4] = {0xe0f0f3f0cccccccdull, 0x4e9ee0c9a10a60e2ull, 0x3333333333333333ull,
                                        0x3333333333333333ull};   // src/minroot.rs:273-278
const uint64_t FQ_RESCUE_INVALPHA[4] = {0xd69f2280cccccccdull, 0x4e9ee0c9a143ba4aull, 0x3333333333333333ull,
                                        0x3333333333333333ull};   // src/minroot.rs:280-285

// The chains run in the lazy domain [0, 2m) and canonicalise once, at the end (host_math.hpp).
inline Fe lmul(const Fe& a, const Fe& b, const Field& F) { return mul<true>(a, b, F); }
inline Fe lsqr(const Fe& a, const Field& F) { return sqr<true>(a, F); }

struct Chain {   // the closures of src/minroot.rs:89-92 / :224-227
  const Field& F;
  Fe sq(Fe x, int n) const { for (int i = 0; i < n; ++i) x = lsqr(x, F); return x; }
  Fe sqr_mul(const Fe& x, int n, const Fe& y) const { return lmul(y, sq(x, n), F); }
};

// PallasVDF::forward_step_ltr_addition_chain, src/minroot.rs:88-127
Fe fwd_ltr_addchain_fq(const Fe& x) {
  const Field& F = field_fq();
  Chain c{F};
  Fe q1 = x, q10 = c.sq(q1, 1), q11 = lmul(q10, q1, F), q101 = lmul(q10, q11, F), q110 = c.sq(q11, 1);
  Fe q111 = lmul(q110, q1, F), q1001 = lmul(q111, q10, F), q1111 = lmul(q1001, q110, F);
  Fe qr2 = c.sqr_mul(q110, 3, q11), qr4 = c.sqr_mul(qr2, 8, qr2), qr8 = c.sqr_mul(qr4, 16, qr4);
  Fe qr16 = c.sqr_mul(qr8, 32, qr8), qr32 = c.sqr_mul(qr16, 64, qr16);
  Fe v = c.sqr_mul(qr32, 5, q1001);
  struct { int n; const Fe* y; } steps[] = {{8, &q111}, {4, &q1}, {2, &qr4}, {7, &q11}, {6, &q1001}, {3, &q101},
      {7, &q101}, {7, &q111}, {4, &q111}, {5, &q1001}, {5, &q101}, {3, &q11}, {4, &q101}, {3, &q101}, {6, &q1111},
      {4, &q1001}, {6, &q101}, {37, &qr8}, {2, &q1}};
  for (auto& s : steps) v = c.sqr_mul(v, s.n, *s.y);
  return canon(v, F);
}
// PallasVDF::forward_step_rtl_sequential, src/minroot.rs:130-151
Fe fwd_rtl_fq(const Fe& x) {
  const Field& F = field_fq();
  Fe acc = one(F), s = x;
  for (int count = 0; count < 254; ++count) {
    if ((FQ_RESCUE_INVALPHA[count / 64] >> (count % 64)) & 1) acc = lmul(acc, s, F);
    s = lsqr(s, F);
  }
  return canon(acc, F);
}
// PallasVDF::forward_step_sequential_rtl_addition_chain, src/minroot.rs:154-196
Fe fwd_rtl_addchain_fq(const Fe& x) {
  const Field& F = field_fq();
  Fe acc = one(F), s = x, last = x;
  for (int count = 0; count < 128; ++count) {
    last = s;
    if ((FQ_RESCUE_INVALPHA[count / 64] >> (count % 64)) & 1) acc = lmul(acc, s, F);
    s = lsqr(s, F);
  }
  Fe sa = last;
  sa = lmul(sa, lsqr(sa, F), F);                                   // :179
  sa = lmul(sa, lsqr(lsqr(lsqr(lsqr(sa, F), F), F), F), F);           // :180
  for (int count = 1; count <= 122; ++count) {                   // :182-195
    sa = lsqr(sa, F);
    if (count % 8 == 1) acc = lmul(acc, sa, F);
  }
  return canon(acc, F);
}
// VestaVDF::forward_step, src/minroot.rs:223-261
Fe fwd_addchain_fp(const Fe& x) {
  const Field& F = field_fp();
  Chain c{F};
  Fe p1 = x, p10 = c.sq(p1, 1), p11 = lmul(p10, p1, F), p101 = lmul(p10, p11, F), p110 = c.sq(p11, 1);
  Fe p111 = lmul(p110, p1, F), p1001 = lmul(p111, p10, F), p1111 = lmul(p1001, p110, F);
  Fe pr2 = c.sqr_mul(p110, 3, p11), pr4 = c.sqr_mul(pr2, 8, pr2), pr8 = c.sqr_mul(pr4, 16, pr4);
  Fe pr16 = c.sqr_mul(pr8, 32, pr8), pr32 = c.sqr_mul(pr16, 64, pr16);
  Fe v = c.sqr_mul(pr32, 5, p1001);
  struct { int n; const Fe* y; } steps[] = {{8, &p111}, {4, &p1}, {2, &pr4}, {7, &p11}, {6, &p1001}, {3, &p101},
      {5, &p1}, {7, &p101}, {4, &p11}, {8, &p111}, {4, &p1}, {4, &p111}, {9, &p1111}, {8, &p1111}, {6, &p1111},
      {2, &p11}, {34, &pr8}, {2, &p1}};
  for (auto& s : steps) v = c.sqr_mul(v, s.n, *s.y);
  return canon(v, F);
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

}  // namespace

extern "C" {

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
#if defined(__SANITIZE_THREAD__)
__attribute__((flatten, noinline))               // an ifunc resolver runs before ThreadSanitizer's runtime is up
#else
__attribute__((target_clones("default", "arch=broadwell"), flatten, noinline))
#endif
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

}  // extern "C"
