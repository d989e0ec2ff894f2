/* pasta_ref.c -- plain-C CPU restatement of the protocol/vdf Nova/MinRoot hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/pasta.py for the full statement).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this; the product
 * (libvdf_hip.so, vdf_amd/) never does.  Kind: "port" -- the Rust reference cannot be built
 * here (no cargo/rustc), and the algorithms on this path live in third-party crates that are
 * absent from /root/reference (pasta_curves 0.4.0, pasta-msm 0.1.1, nova-snark 0.8.0;
 * Cargo.toml:15-18).  PARITY UNPINNED for known-answer values: the reference's tests hold
 * none; this file is pinned against oracle/pasta.py (Python big integers) and the
 * reference's property tests (tests/test_oracle.py).
 *
 * Independent of the device code on purpose: 4 x 64-bit limbs with unsigned __int128 CIOS
 * Montgomery (the device uses 32/29-bit limbs), so a shared bug is unlikely.
 *
 * Reference lines restated: src/minroot.rs:73-75, :88-196, :223-261, :273-285, :312-344,
 * :352-365; src/nova/proof.rs:107-126, :162-189.
 */
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;
typedef struct { uint64_t l[4]; } fe;
typedef struct { fe x, y; } affine;
typedef struct { fe x, y, zz, zzz; } xyzz;
typedef struct { fe x, y, z; } jac;

typedef struct {
  uint64_t m[4];
  uint64_t inv;     /* -m^-1 mod 2^64 */
  uint64_t one[4];  /* R mod m */
  uint64_t r2[4];   /* R^2 mod m */
} field;

/* SURVEY.md Appendix A */
static const field FP = {
    {0x992d30ed00000001ull, 0x224698fc094cf91bull, 0x0000000000000000ull, 0x4000000000000000ull},
    0x992d30ecffffffffull,
    {0x34786d38fffffffdull, 0x992c350be41914adull, 0xffffffffffffffffull, 0x3fffffffffffffffull},
    {0x8c78ecb30000000full, 0xd7d30dbd8b0de0e7ull, 0x7797a99bc3c95d18ull, 0x096d41af7b9cb714ull}};
static const field FQ = {
    {0x8c46eb2100000001ull, 0x224698fc0994a8ddull, 0x0000000000000000ull, 0x4000000000000000ull},
    0x8c46eb20ffffffffull,
    {0x5b2b3e9cfffffffdull, 0x992c350be3420567ull, 0xffffffffffffffffull, 0x3fffffffffffffffull},
    {0xfc9678ff0000000full, 0x67bb433d891a16e3ull, 0x7fae231004ccf590ull, 0x096d41af7ccfdaa9ull}};

static const field* fld(int f) { return f == 0 ? &FP : &FQ; }

static int fe_is_zero(const fe* a) { return (a->l[0] | a->l[1] | a->l[2] | a->l[3]) == 0; }
static int fe_eq(const fe* a, const fe* b) {
  return ((a->l[0] ^ b->l[0]) | (a->l[1] ^ b->l[1]) | (a->l[2] ^ b->l[2]) | (a->l[3] ^ b->l[3])) == 0;
}
static int geq_m(const uint64_t* t, const field* F) {
  for (int i = 3; i >= 0; --i) {
    if (t[i] > F->m[i]) return 1;
    if (t[i] < F->m[i]) return 0;
  }
  return 1;
}
static void sub_m(uint64_t* t, const field* F) {
  u128 b = 0;
  for (int i = 0; i < 4; ++i) {
    u128 d = (u128)t[i] - F->m[i] - (uint64_t)b;
    t[i] = (uint64_t)d;
    b = (d >> 64) & 1;
  }
}
static void fe_add(fe* r, const fe* a, const fe* b, const field* F) {
  u128 c = 0;
  uint64_t t[4];
  for (int i = 0; i < 4; ++i) { c += (u128)a->l[i] + b->l[i]; t[i] = (uint64_t)c; c >>= 64; }
  if (geq_m(t, F)) sub_m(t, F);
  memcpy(r->l, t, 32);
}
static void fe_sub(fe* r, const fe* a, const fe* b, const field* F) {
  u128 br = 0;
  uint64_t t[4];
  for (int i = 0; i < 4; ++i) {
    u128 d = (u128)a->l[i] - b->l[i] - (uint64_t)br;
    t[i] = (uint64_t)d;
    br = (d >> 64) & 1;
  }
  if (br) {
    u128 c = 0;
    for (int i = 0; i < 4; ++i) { c += (u128)t[i] + F->m[i]; t[i] = (uint64_t)c; c >>= 64; }
  }
  memcpy(r->l, t, 32);
}
static void fe_neg(fe* r, const fe* a, const field* F) {
  fe z = {{0, 0, 0, 0}};
  if (fe_is_zero(a)) { *r = *a; return; }
  fe_sub(r, &z, a, F);
}
/* CIOS Montgomery multiplication, R = 2^256 */
static void fe_mul(fe* r, const fe* a, const fe* b, const field* F) {
  uint64_t t[6] = {0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 4; ++i) {
    u128 c = 0;
    for (int j = 0; j < 4; ++j) {
      c += (u128)a->l[j] * b->l[i] + t[j];
      t[j] = (uint64_t)c;
      c >>= 64;
    }
    c += t[4];
    t[4] = (uint64_t)c;
    t[5] = (uint64_t)(c >> 64);
    uint64_t q = t[0] * F->inv;
    c = (u128)q * F->m[0] + t[0];
    c >>= 64;
    for (int j = 1; j < 4; ++j) {
      c += (u128)q * F->m[j] + t[j];
      t[j - 1] = (uint64_t)c;
      c >>= 64;
    }
    c += t[4];
    t[3] = (uint64_t)c;
    t[4] = t[5] + (uint64_t)(c >> 64);
  }
  if (t[4] || geq_m(t, F)) sub_m(t, F);
  memcpy(r->l, t, 32);
}
static void fe_sqr(fe* r, const fe* a, const field* F) { fe_mul(r, a, a, F); }
static void fe_from_mont(fe* r, const fe* a, const field* F) {
  fe one = {{1, 0, 0, 0}};
  fe_mul(r, a, &one, F);
}
static void fe_to_mont(fe* r, const fe* a, const field* F) {
  fe r2;
  memcpy(r2.l, F->r2, 32);
  fe_mul(r, a, &r2, F);
}
static void fe_set_one(fe* r, const field* F) { memcpy(r->l, F->one, 32); }
static void fe_set_u64(fe* r, uint64_t v, const field* F) {
  fe t = {{v, 0, 0, 0}};
  fe_to_mont(r, &t, F);
}
/* left-to-right square and multiply, as ff::Field::pow_vartime (src/minroot.rs:312-314) */
static void fe_pow(fe* r, const fe* a, const uint64_t e[4], const field* F) {
  fe acc;
  fe_set_one(&acc, F);
  for (int i = 3; i >= 0; --i)
    for (int b = 63; b >= 0; --b) {
      fe_sqr(&acc, &acc, F);
      if ((e[i] >> b) & 1) fe_mul(&acc, &acc, a, F);
    }
  *r = acc;
}
static void fe_inv(fe* r, const fe* a, const field* F) {
  uint64_t e[4] = {F->m[0] - 2, F->m[1], F->m[2], F->m[3]};
  fe_pow(r, a, e, F);
}

/* ---- exported element-wise helpers (n elements, Montgomery form in and out) --------------- */
void ref_fe_mul(int f, const fe* a, const fe* b, size_t n, fe* out) {
  for (size_t i = 0; i < n; ++i) fe_mul(&out[i], &a[i], &b[i], fld(f));
}
void ref_fe_to_mont(int f, const fe* a, size_t n, fe* out) {
  for (size_t i = 0; i < n; ++i) fe_to_mont(&out[i], &a[i], fld(f));
}
void ref_fe_from_mont(int f, const fe* a, size_t n, fe* out) {
  for (size_t i = 0; i < n; ++i) fe_from_mont(&out[i], &a[i], fld(f));
}
/* a + r*b  (nova-snark RelaxedR1CSWitness::fold) */
void ref_axpy(int f, const fe* a, const fe* r, const fe* b, size_t n, fe* out) {
  const field* F = fld(f);
  for (size_t i = 0; i < n; ++i) {
    fe t;
    fe_mul(&t, r, &b[i], F);
    fe_add(&out[i], &a[i], &t, F);
  }
}
/* T = AZ1*BZ2 + AZ2*BZ1 - u1*CZ2 - CZ1  (nova-snark commit_T) */
void ref_cross_term(int f, const fe* az1, const fe* bz1, const fe* cz1, const fe* az2, const fe* bz2, const fe* cz2,
                    const fe* u1, size_t n, fe* T) {
  const field* F = fld(f);
  for (size_t i = 0; i < n; ++i) {
    fe t0, t1, t2;
    fe_mul(&t0, &az1[i], &bz2[i], F);
    fe_mul(&t1, &az2[i], &bz1[i], F);
    fe_add(&t0, &t0, &t1, F);
    fe_mul(&t2, u1, &cz2[i], F);
    fe_sub(&t0, &t0, &t2, F);
    fe_sub(&T[i], &t0, &cz1[i], F);
  }
}
/* COO sparse mat-vec (nova-snark R1CSShape::multiply_vec) */
void ref_spmv(int f, const uint32_t* rows, const uint32_t* cols, const fe* vals, size_t nnz, const fe* z,
              size_t num_rows, fe* out) {
  const field* F = fld(f);
  memset(out, 0, num_rows * sizeof(fe));
  for (size_t k = 0; k < nnz; ++k) {
    fe t;
    fe_mul(&t, &vals[k], &z[cols[k]], F);
    fe_add(&out[rows[k]], &out[rows[k]], &t, F);
  }
}

/* ---- MinRoot (src/minroot.rs) ---------------------------------------------------------------- */
static const uint64_t FP_RESCUE_INVALPHA[4] = {0xe0f0f3f0cccccccdull, 0x4e9ee0c9a10a60e2ull, 0x3333333333333333ull,
                                               0x3333333333333333ull}; /* :273-278 */
static const uint64_t FQ_RESCUE_INVALPHA[4] = {0xd69f2280cccccccdull, 0x4e9ee0c9a143ba4aull, 0x3333333333333333ull,
                                               0x3333333333333333ull}; /* :280-285 */

static void sqr_n(fe* x, int n, const field* F) { for (int i = 0; i < n; ++i) fe_sqr(x, x, F); }
/* sqr_mul(x, n, y) = y * x^(2^n)  (:92, :227) */
static void sqr_mul(fe* r, const fe* x, int n, const fe* y, const field* F) {
  fe t = *x;
  sqr_n(&t, n, F);
  fe_mul(r, y, &t, F);
}

/* src/minroot.rs:88-127 */
static void fwd_ltr_addchain_fq(fe* r, const fe* x) {
  const field* F = &FQ;
  fe q1 = *x, q10, q11, q101, q110, q111, q1001, q1111, qr2, qr4, qr8, qr16, qr32, v;
  q10 = q1; sqr_n(&q10, 1, F);
  fe_mul(&q11, &q10, &q1, F);
  fe_mul(&q101, &q10, &q11, F);
  q110 = q11; sqr_n(&q110, 1, F);
  fe_mul(&q111, &q110, &q1, F);
  fe_mul(&q1001, &q111, &q10, F);
  fe_mul(&q1111, &q1001, &q110, F);
  sqr_mul(&qr2, &q110, 3, &q11, F);
  sqr_mul(&qr4, &qr2, 8, &qr2, F);
  sqr_mul(&qr8, &qr4, 16, &qr4, F);
  sqr_mul(&qr16, &qr8, 32, &qr8, F);
  sqr_mul(&qr32, &qr16, 64, &qr16, F);
  sqr_mul(&v, &qr32, 5, &q1001, F);
  const int ns[19] = {8, 4, 2, 7, 6, 3, 7, 7, 4, 5, 5, 3, 4, 3, 6, 4, 6, 37, 2};
  const fe* ys[19] = {&q111, &q1, &qr4, &q11, &q1001, &q101, &q101, &q111, &q111, &q1001,
                      &q101, &q11, &q101, &q101, &q1111, &q1001, &q101, &qr8, &q1};
  for (int i = 0; i < 19; ++i) sqr_mul(&v, &v, ns[i], ys[i], F);
  *r = v;
}
/* src/minroot.rs:130-151 */
static void fwd_rtl_fq(fe* r, const fe* x) {
  const field* F = &FQ;
  fe acc, sq = *x;
  fe_set_one(&acc, F);
  for (int count = 0; count < 254; ++count) {
    if ((FQ_RESCUE_INVALPHA[count / 64] >> (count % 64)) & 1) fe_mul(&acc, &acc, &sq, F);
    fe_sqr(&sq, &sq, F);
  }
  *r = acc;
}
/* src/minroot.rs:154-196 */
static void fwd_rtl_addchain_fq(fe* r, const fe* x) {
  const field* F = &FQ;
  fe acc, sq = *x, last = *x, s, t;
  fe_set_one(&acc, F);
  for (int count = 0; count < 128; ++count) {
    last = sq;
    if ((FQ_RESCUE_INVALPHA[count / 64] >> (count % 64)) & 1) fe_mul(&acc, &acc, &sq, F);
    fe_sqr(&sq, &sq, F);
  }
  s = last;
  fe_sqr(&t, &s, F); fe_mul(&s, &s, &t, F);                 /* :179 */
  t = s; sqr_n(&t, 4, F); fe_mul(&s, &s, &t, F);            /* :180 */
  for (int count = 1; count <= 122; ++count) {             /* :182-195 */
    fe_sqr(&s, &s, F);
    if (count % 8 == 1) fe_mul(&acc, &acc, &s, F);
  }
  *r = acc;
}
/* src/minroot.rs:223-261 */
static void fwd_addchain_fp(fe* r, const fe* x) {
  const field* F = &FP;
  fe p1 = *x, p10, p11, p101, p110, p111, p1001, p1111, pr2, pr4, pr8, pr16, pr32, v;
  p10 = p1; sqr_n(&p10, 1, F);
  fe_mul(&p11, &p10, &p1, F);
  fe_mul(&p101, &p10, &p11, F);
  p110 = p11; sqr_n(&p110, 1, F);
  fe_mul(&p111, &p110, &p1, F);
  fe_mul(&p1001, &p111, &p10, F);
  fe_mul(&p1111, &p1001, &p110, F);
  sqr_mul(&pr2, &p110, 3, &p11, F);
  sqr_mul(&pr4, &pr2, 8, &pr2, F);
  sqr_mul(&pr8, &pr4, 16, &pr4, F);
  sqr_mul(&pr16, &pr8, 32, &pr8, F);
  sqr_mul(&pr32, &pr16, 64, &pr16, F);
  sqr_mul(&v, &pr32, 5, &p1001, F);
  const int ns[18] = {8, 4, 2, 7, 6, 3, 5, 7, 4, 8, 4, 4, 9, 8, 6, 2, 34, 2};
  const fe* ys[18] = {&p111, &p1, &pr4, &p11, &p1001, &p101, &p1, &p101, &p11,
                      &p111, &p1, &p111, &p1111, &p1111, &p1111, &p11, &pr8, &p1};
  for (int i = 0; i < 18; ++i) sqr_mul(&v, &v, ns[i], ys[i], F);
  *r = v;
}
/* dispatch of src/minroot.rs:77-84; mode 0..3 = EvalMode order (:15-20); Vesta ignores it (:203-205) */
static void forward_step(fe* r, const fe* x, int f, int mode) {
  if (f == 0) { fwd_addchain_fp(r, x); return; }
  switch (mode) {
    case 0: fe_pow(r, x, FQ_RESCUE_INVALPHA, &FQ); break;
    case 1: fwd_ltr_addchain_fq(r, x); break;
    case 2: fwd_rtl_fq(r, x); break;
    default: fwd_rtl_addchain_fq(r, x); break;
  }
}
void ref_forward_step(int f, int mode, const fe* x, fe* out) { forward_step(out, x, f, mode); }
void ref_forward_step_pow(int f, const fe* x, fe* out) {   /* default trait method, :312-314 */
  fe_pow(out, x, f == 0 ? FP_RESCUE_INVALPHA : FQ_RESCUE_INVALPHA, fld(f));
}
/* src/minroot.rs:73-75 */
static void inverse_step(fe* r, const fe* x, const field* F) {
  fe t;
  fe_sqr(&t, x, F);
  fe_sqr(&t, &t, F);
  fe_mul(r, x, &t, F);
}
/* state = {x, y, i}.  t forward rounds (:329-335, :352-359); trace (optional) receives the
 * (x, y) of every state 0..t. */
void ref_minroot_eval(int f, int mode, const fe* state_in, uint64_t t, fe* state_out, fe* trace_xy) {
  const field* F = fld(f);
  fe x = state_in[0], y = state_in[1], i = state_in[2], one;
  fe_set_one(&one, F);
  if (trace_xy) { trace_xy[0] = x; trace_xy[1] = y; }
  for (uint64_t k = 0; k < t; ++k) {
    fe s, nx, ny;
    fe_add(&s, &x, &y, F);
    forward_step(&nx, &s, f, mode);
    fe_add(&ny, &x, &i, F);
    fe_add(&i, &i, &one, F);
    x = nx; y = ny;
    if (trace_xy) { trace_xy[2 * (k + 1)] = x; trace_xy[2 * (k + 1) + 1] = y; }
  }
  state_out[0] = x; state_out[1] = y; state_out[2] = i;
}
/* t inverse rounds (:338-344, :363-365) */
void ref_minroot_inverse_eval(int f, const fe* state_in, uint64_t t, fe* state_out) {
  const field* F = fld(f);
  fe x = state_in[0], y = state_in[1], i = state_in[2], one;
  fe_set_one(&one, F);
  for (uint64_t k = 0; k < t; ++k) {
    fe ni, nx, ny;
    fe_sub(&ni, &i, &one, F);
    fe_sub(&nx, &y, &ni, F);
    inverse_step(&ny, &x, F);
    fe_sub(&ny, &ny, &nx, F);
    x = nx; y = ny; i = ni;
  }
  state_out[0] = x; state_out[1] = y; state_out[2] = i;
}
/* The 4t+1 aux values of InverseMinRootCircuit::synthesize, computed the way the circuit does
 * (sequentially from `result`; src/nova/proof.rs:107-126, :162-189). */
void ref_step_witness(int f, const fe* result, uint64_t t, fe* W) {
  const field* F = fld(f);
  fe x = result[0], y = result[1], i = result[2], one;
  fe_set_one(&one, F);
  for (uint64_t k = 0; k < t; ++k) {
    fe ni, nx, t1, t2, ny;
    fe_sub(&ni, &i, &one, F);          /* :162-164 */
    fe_sub(&nx, &y, &ni, F);           /* :167-173 */
    fe_sqr(&t1, &x, F);                /* :176 */
    fe_sqr(&t2, &t1, F);               /* :178 */
    fe_mul(&ny, &t2, &x, F);           /* :181-189 */
    fe_sub(&ny, &ny, &nx, F);
    W[4 * k] = nx; W[4 * k + 1] = t1; W[4 * k + 2] = t2; W[4 * k + 3] = ny;
    x = nx; y = ny; i = ni;
  }
  W[4 * t] = i;                        /* final_i, :122-126 */
}

/* ---- curve: y^2 = x^3 + 5, XYZZ (EFD madd-2008-s / add-2008-s / dbl-2008-s-1) ------------------ */
static int aff_is_id(const affine* a) { return fe_is_zero(&a->x) && fe_is_zero(&a->y); }
static void xyzz_set_id(xyzz* r) { memset(r, 0, sizeof(*r)); }
static void xyzz_dbl(xyzz* r, const xyzz* a, const field* F) {
  if (fe_is_zero(&a->zz)) { *r = *a; return; }
  fe U, V, W, S, M, t, X3, Y3;
  fe_add(&U, &a->y, &a->y, F);
  fe_sqr(&V, &U, F);
  fe_mul(&W, &U, &V, F);
  fe_mul(&S, &a->x, &V, F);
  fe_sqr(&t, &a->x, F);
  fe_add(&M, &t, &t, F);
  fe_add(&M, &M, &t, F);
  fe_sqr(&X3, &M, F);
  fe_sub(&X3, &X3, &S, F);
  fe_sub(&X3, &X3, &S, F);
  fe_sub(&t, &S, &X3, F);
  fe_mul(&Y3, &M, &t, F);
  fe_mul(&t, &W, &a->y, F);
  fe_sub(&Y3, &Y3, &t, F);
  fe zz, zzz;
  fe_mul(&zz, &V, &a->zz, F);
  fe_mul(&zzz, &W, &a->zzz, F);
  r->x = X3; r->y = Y3; r->zz = zz; r->zzz = zzz;
}
static void xyzz_add(xyzz* acc, const xyzz* b, const field* F) {
  if (fe_is_zero(&b->zz)) return;
  if (fe_is_zero(&acc->zz)) { *acc = *b; return; }
  fe U1, U2, S1, S2, P, R, PP, PPP, Q, X3, Y3, t;
  fe_mul(&U1, &acc->x, &b->zz, F);
  fe_mul(&U2, &b->x, &acc->zz, F);
  fe_mul(&S1, &acc->y, &b->zzz, F);
  fe_mul(&S2, &b->y, &acc->zzz, F);
  fe_sub(&P, &U2, &U1, F);
  fe_sub(&R, &S2, &S1, F);
  if (fe_is_zero(&P)) {
    if (fe_is_zero(&R)) { xyzz d; xyzz_dbl(&d, acc, F); *acc = d; }
    else xyzz_set_id(acc);
    return;
  }
  fe_sqr(&PP, &P, F);
  fe_mul(&PPP, &P, &PP, F);
  fe_mul(&Q, &U1, &PP, F);
  fe_sqr(&X3, &R, F);
  fe_sub(&X3, &X3, &PPP, F);
  fe_sub(&X3, &X3, &Q, F);
  fe_sub(&X3, &X3, &Q, F);
  fe_sub(&t, &Q, &X3, F);
  fe_mul(&Y3, &R, &t, F);
  fe_mul(&t, &S1, &PPP, F);
  fe_sub(&Y3, &Y3, &t, F);
  fe_mul(&t, &acc->zz, &b->zz, F);
  fe_mul(&acc->zz, &t, &PP, F);
  fe_mul(&t, &acc->zzz, &b->zzz, F);
  fe_mul(&acc->zzz, &t, &PPP, F);
  acc->x = X3; acc->y = Y3;
}
static void xyzz_madd(xyzz* acc, const affine* b, int negate, const field* F) {
  if (aff_is_id(b)) return;
  xyzz t;
  t.x = b->x;
  if (negate) fe_neg(&t.y, &b->y, F); else t.y = b->y;
  fe_set_one(&t.zz, F);
  fe_set_one(&t.zzz, F);
  xyzz_add(acc, &t, F);   /* general add with zz = 1: same group result as the mixed formula */
}
static void xyzz_to_affine(affine* r, const xyzz* a, const field* F) {
  if (fe_is_zero(&a->zz)) { memset(r, 0, sizeof(*r)); return; }
  fe izz, izzz;
  fe_inv(&izz, &a->zz, F);
  fe_inv(&izzz, &a->zzz, F);
  fe_mul(&r->x, &a->x, &izz, F);
  fe_mul(&r->y, &a->y, &izzz, F);
}
static const field* curve_field(int curve) { return curve == 0 ? &FP : &FQ; }        /* coordinates */
static const field* curve_scalar_field(int curve) { return curve == 0 ? &FQ : &FP; } /* scalars */

/* affine normalisation of a Jacobian point (x/z^2, y/z^3) -- the parity canonical form */
void ref_jac_to_affine(int curve, const jac* p, affine* out) {
  const field* F = curve_field(curve);
  if (fe_is_zero(&p->z)) { memset(out, 0, sizeof(*out)); return; }
  fe zi, zi2, zi3;
  fe_inv(&zi, &p->z, F);
  fe_sqr(&zi2, &zi, F);
  fe_mul(&zi3, &zi2, &zi, F);
  fe_mul(&out->x, &p->x, &zi2, F);
  fe_mul(&out->y, &p->y, &zi3, F);
}
int ref_on_curve(int curve, const affine* p) {
  const field* F = curve_field(curve);
  if (aff_is_id(p)) return 1;
  fe y2, x3, five;
  fe_sqr(&y2, &p->y, F);
  fe_sqr(&x3, &p->x, F);
  fe_mul(&x3, &x3, &p->x, F);
  fe_set_u64(&five, 5, F);
  fe_add(&x3, &x3, &five, F);
  return fe_eq(&y2, &x3);
}
size_t ref_count_off_curve(int curve, const affine* p, size_t n) {
  size_t bad = 0;
  for (size_t i = 0; i < n; ++i) bad += !ref_on_curve(curve, &p[i]);
  return bad;
}

static uint64_t splitmix64(uint64_t x) {
  uint64_t z = x + 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
/* synthetic generators P_i = [k_i] G, G = (-1, 2)  (oracle/pasta.py base_dlog) */
void ref_synthetic_bases(int curve, uint64_t seed, size_t start, size_t n, affine* out) {
  const field* F = curve_field(curve);
  affine g;
  fe one, two;
  fe_set_one(&one, F);
  fe_neg(&g.x, &one, F);
  fe_add(&two, &one, &one, F);
  g.y = two;
  for (size_t i = 0; i < n; ++i) {
    uint64_t k = splitmix64(seed * 0xD1342543DE82EF95ull + (start + i)) | 1ull;
    xyzz acc;
    xyzz_set_id(&acc);
    for (int b = 63; b >= 0; --b) {
      xyzz d;
      xyzz_dbl(&d, &acc, F);
      acc = d;
      if ((k >> b) & 1) xyzz_madd(&acc, &g, 0, F);
    }
    xyzz_to_affine(&out[i], &acc, F);
  }
}

/* Tonelli-Shanks square root (2-adicity 32 in both fields); returns 0 for a non-residue */
static int fe_sqrt(fe* r, const fe* a, const field* F) {
  if (fe_is_zero(a)) { *r = *a; return 1; }
  /* m - 1 = 2^32 * t */
  uint64_t t[4], e[4];
  memcpy(t, F->m, 32);
  t[0] -= 1;
  for (int i = 0; i < 4; ++i) e[i] = (t[i] >> 1) | (i < 3 ? t[i + 1] << 63 : 0);        /* (m - 1) / 2 */
  fe one, leg;
  fe_set_one(&one, F);
  fe_pow(&leg, a, e, F);
  if (!fe_eq(&leg, &one)) return 0;
  uint64_t tt[4], th[4];
  for (int i = 0; i < 4; ++i) tt[i] = (t[i] >> 32) | (i < 3 ? t[i + 1] << 32 : 0);        /* t = (m - 1) >> 32, odd */
  /* (t + 1) / 2 */
  { u128 c = (u128)tt[0] + 1; th[0] = (uint64_t)c; c >>= 64; for (int i = 1; i < 4; ++i) { c += tt[i]; th[i] = (uint64_t)c; c >>= 64; } }
  for (int i = 0; i < 4; ++i) th[i] = (th[i] >> 1) | (i < 3 ? th[i + 1] << 63 : 0);
  fe z, c, x, b, mone;
  fe_neg(&mone, &one, F);
  for (uint64_t k = 2;; ++k) {                        /* smallest non-residue, as oracle/pasta.py sqrt_mod */
    fe_set_u64(&z, k, F);
    fe_pow(&leg, &z, e, F);
    if (fe_eq(&leg, &mone)) break;
  }
  fe_pow(&c, &z, tt, F);
  fe_pow(&x, a, th, F);
  fe_pow(&b, a, tt, F);
  int s = 32;
  while (!fe_eq(&b, &one)) {
    int k = 0;
    fe q = b;
    while (!fe_eq(&q, &one)) { fe_sqr(&q, &q, F); ++k; }
    fe g = c;
    for (int i = 0; i < s - k - 1; ++i) fe_sqr(&g, &g, F);
    fe_mul(&x, &x, &g, F);
    fe_sqr(&c, &g, F);
    fe_mul(&b, &b, &c, F);
    s = k;
  }
  *r = x;
  return 1;
}
static uint64_t rotl64(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
/* generator family 1 (oracle/pasta.py tai_base): seeded try-and-increment, even y */
void ref_tai_bases(int curve, uint64_t seed, size_t start, size_t n, affine* out) {
  const field* F = curve_field(curve);
  fe five;
  fe_set_u64(&five, 5, F);
  for (size_t i = 0; i < n; ++i) {
    uint64_t sm = seed * 0xD1342543DE82EF95ull + (uint64_t)(start + i) * 0x9E3779B97F4A7C15ull, st[4];
    for (int k = 0; k < 4; ++k) { st[k] = splitmix64(sm); sm += 0x9E3779B97F4A7C15ull; }
    for (;;) {
      uint64_t w[4];
      for (int k = 0; k < 4; ++k) {
        w[k] = rotl64(st[1] * 5, 7) * 9;
        const uint64_t t = st[1] << 17;
        st[2] ^= st[0]; st[3] ^= st[1]; st[1] ^= st[2]; st[0] ^= st[3];
        st[2] ^= t;
        st[3] = rotl64(st[3], 45);
      }
      while (geq_m(w, F)) sub_m(w, F);
      fe xc, x, rhs, y, yc;
      memcpy(xc.l, w, 32);
      fe_to_mont(&x, &xc, F);
      fe_sqr(&rhs, &x, F);
      fe_mul(&rhs, &rhs, &x, F);
      fe_add(&rhs, &rhs, &five, F);
      if (!fe_sqrt(&y, &rhs, F)) continue;
      fe_from_mont(&yc, &y, F);
      if (yc.l[0] & 1) fe_neg(&y, &y, F);
      out[i].x = x; out[i].y = y;
      break;
    }
  }
}

/* ---- Pippenger MSM (pasta-msm 0.1.1's published algorithm: signed windows, XYZZ buckets with
 * mixed addition, running-sum bucket reduction, windows spread over a thread pool) ------------ */
typedef struct {
  int curve, c, w;
  size_t n;
  const affine* pts;
  const fe* scal;   /* canonical (non-Montgomery) scalars */
  xyzz result;      /* sum_b b * bucket_b of this window */
} win_job;

static int32_t signed_digit(const fe* s, int c, int w, int* carry_io) {
  /* digits are produced in window order by the caller (carry chain) */
  int bit = w * c;
  uint64_t raw = 0;
  int l = bit / 64, sh = bit % 64;
  if (l < 4) {
    raw = s->l[l] >> sh;
    if (sh && l + 1 < 4) raw |= s->l[l + 1] << (64 - sh);
  }
  raw &= ((1ull << c) - 1);
  raw += (uint64_t)*carry_io;
  if (raw > (1ull << (c - 1))) { *carry_io = 1; return (int32_t)((int64_t)raw - (int64_t)(1ull << c)); }
  *carry_io = 0;
  return (int32_t)raw;
}

typedef struct { win_job* jobs; int njobs; int next; pthread_mutex_t mu; int8_t* unused; int32_t* digits; } pool_t;

static void run_window(win_job* j, const int32_t* digits) {
  const field* F = curve_field(j->curve);
  size_t nb = (size_t)1 << (j->c - 1);
  xyzz* buckets = (xyzz*)calloc(nb, sizeof(xyzz));
  for (size_t i = 0; i < j->n; ++i) {
    int32_t d = digits[(size_t)j->w * j->n + i];
    if (d > 0) xyzz_madd(&buckets[d - 1], &j->pts[i], 0, F);
    else if (d < 0) xyzz_madd(&buckets[-d - 1], &j->pts[i], 1, F);
  }
  xyzz run, tot;
  xyzz_set_id(&run);
  xyzz_set_id(&tot);
  for (size_t b = nb; b-- > 0;) {
    xyzz_add(&run, &buckets[b], F);
    xyzz_add(&tot, &run, F);
  }
  j->result = tot;
  free(buckets);
}
static void* worker(void* arg) {
  pool_t* p = (pool_t*)arg;
  for (;;) {
    pthread_mutex_lock(&p->mu);
    int k = p->next++;
    pthread_mutex_unlock(&p->mu);
    if (k >= p->njobs) break;
    run_window(&p->jobs[k], p->digits);
  }
  return NULL;
}

/* out (Jacobian, z = 1 or identity) = sum scalars[i] * pts[i];  threads <= 0 -> 1 */
void ref_msm(int curve, const affine* pts, const fe* scalars, size_t n, int is_mont, int threads, int window_bits,
             jac* out) {
  const field* F = curve_field(curve);
  const field* SF = curve_scalar_field(curve);
  memset(out, 0, sizeof(*out));
  if (n == 0) return;
  int c = window_bits;
  if (c <= 0) {
    int lg = 0;
    while (((size_t)1 << (lg + 1)) <= n) ++lg;
    c = lg > 6 ? lg - 3 : 3;
    if (c > 16) c = 16;
  }
  int W = (256 + c - 1) / c;
  fe* canon = (fe*)malloc(n * sizeof(fe));
  for (size_t i = 0; i < n; ++i) {
    if (is_mont) fe_from_mont(&canon[i], &scalars[i], SF); else canon[i] = scalars[i];
  }
  int32_t* digits = (int32_t*)malloc((size_t)W * n * sizeof(int32_t));
  for (size_t i = 0; i < n; ++i) {
    int carry = 0;
    for (int w = 0; w < W; ++w) digits[(size_t)w * n + i] = signed_digit(&canon[i], c, w, &carry);
  }
  win_job* jobs = (win_job*)calloc((size_t)W, sizeof(win_job));
  for (int w = 0; w < W; ++w) { jobs[w].curve = curve; jobs[w].c = c; jobs[w].w = w; jobs[w].n = n; jobs[w].pts = pts; jobs[w].scal = canon; }
  pool_t pool;
  pool.jobs = jobs; pool.njobs = W; pool.next = 0; pool.digits = digits; pool.unused = NULL;
  pthread_mutex_init(&pool.mu, NULL);
  if (threads < 1) threads = 1;
  if (threads > W) threads = W;
  pthread_t* th = (pthread_t*)malloc((size_t)threads * sizeof(pthread_t));
  for (int t = 1; t < threads; ++t) pthread_create(&th[t], NULL, worker, &pool);
  worker(&pool);
  for (int t = 1; t < threads; ++t) pthread_join(th[t], NULL);
  pthread_mutex_destroy(&pool.mu);
  /* Horner over windows */
  xyzz acc;
  xyzz_set_id(&acc);
  for (int w = W - 1; w >= 0; --w) {
    for (int k = 0; k < c; ++k) { xyzz d; xyzz_dbl(&d, &acc, F); acc = d; }
    xyzz_add(&acc, &jobs[w].result, F);
  }
  affine a;
  xyzz_to_affine(&a, &acc, F);
  if (!aff_is_id(&a)) { out->x = a.x; out->y = a.y; fe_set_one(&out->z, F); }
  free(th); free(jobs); free(digits); free(canon);
}

/* naive double-and-add MSM (independent of the Pippenger code above; small n only) */
void ref_msm_naive(int curve, const affine* pts, const fe* scalars, size_t n, int is_mont, jac* out) {
  const field* F = curve_field(curve);
  const field* SF = curve_scalar_field(curve);
  xyzz acc;
  xyzz_set_id(&acc);
  for (size_t i = 0; i < n; ++i) {
    fe s;
    if (is_mont) fe_from_mont(&s, &scalars[i], SF); else s = scalars[i];
    xyzz r;
    xyzz_set_id(&r);
    for (int b = 255; b >= 0; --b) {
      xyzz d;
      xyzz_dbl(&d, &r, F);
      r = d;
      if ((s.l[b / 64] >> (b % 64)) & 1) xyzz_madd(&r, &pts[i], 0, F);
    }
    xyzz_add(&acc, &r, F);
  }
  affine a;
  xyzz_to_affine(&a, &acc, F);
  memset(out, 0, sizeof(*out));
  if (!aff_is_id(&a)) { out->x = a.x; out->y = a.y; fe_set_one(&out->z, F); }
}
