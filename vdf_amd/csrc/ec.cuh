// Pallas / Vesta group law on the device: y^2 = x^3 + 5, XYZZ accumulators with
// mixed (affine) addition.  Replaces the curve layer the reference reaches through
// pasta_curves 0.4.0 / pasta-msm 0.1.1 (Cargo.toml:17-18) for every Pedersen
// commitment on the prove_step path (src/nova/proof.rs:342-349).
//
// Memory encodings (C ABI, include/vdf_hip.h):
//   affine  = {x, y}, 64 B, Montgomery limbs, identity = (0, 0)
//   jac     = {x, y, z}, 96 B, Jacobian (x/z^2, y/z^3), identity has z = 0
//   xyzz    = {x, y, zz, zzz}, 128 B, device-internal; identity has zz = 0
// Formulas: EFD "xyzz" madd-2008-s, add-2008-s, dbl-2008-s-1, mdbl-2008-s-1 (a = 0).
#pragma once
#include "fe.cuh"

namespace vdf {

template <class P> struct Affine { Fe<P> x, y; };
template <class P> struct XYZZ { Fe<P> x, y, zz, zzz; };
template <class P> struct Jac { Fe<P> x, y, z; };

template <class P> VDF_HD bool affine_is_identity(const Affine<P>& a) { return fe_is_zero(a.x) && fe_is_zero(a.y); }
template <class P> VDF_HD bool xyzz_is_identity(const XYZZ<P>& a) { return fe_is_zero(a.zz); }

template <class P> VDF_HD XYZZ<P> xyzz_identity() {
  XYZZ<P> r;
  r.x = fe_zero<P>(); r.y = fe_zero<P>(); r.zz = fe_zero<P>(); r.zzz = fe_zero<P>();
  return r;
}

template <class P> VDF_HD XYZZ<P> xyzz_from_affine(const Affine<P>& a) {
  XYZZ<P> r;
  if (affine_is_identity(a)) return xyzz_identity<P>();
  r.x = a.x; r.y = a.y; r.zz = fe_one<P>(); r.zzz = fe_one<P>();
  return r;
}

// 2 * (affine point), mdbl-2008-s-1.  y = 0 cannot occur on a prime-order curve.
template <class P> VDF_HD XYZZ<P> xyzz_dbl_affine(const Affine<P>& a) {
  XYZZ<P> r;
  Fe<P> U = fe_dbl(a.y);
  Fe<P> V = fe_sqr(U);
  Fe<P> W = fe_mul(U, V);
  Fe<P> S = fe_mul(a.x, V);
  Fe<P> X2 = fe_sqr(a.x);
  Fe<P> M = fe_add(fe_dbl(X2), X2);
  r.x = fe_sub(fe_sub(fe_sqr(M), S), S);
  r.y = fe_sub(fe_mul(M, fe_sub(S, r.x)), fe_mul(W, a.y));
  r.zz = V;
  r.zzz = W;
  return r;
}

// 2 * (xyzz point), dbl-2008-s-1.
template <class P> VDF_HD XYZZ<P> xyzz_dbl(const XYZZ<P>& a) {
  if (xyzz_is_identity(a)) return a;
  XYZZ<P> r;
  Fe<P> U = fe_dbl(a.y);
  Fe<P> V = fe_sqr(U);
  Fe<P> W = fe_mul(U, V);
  Fe<P> S = fe_mul(a.x, V);
  Fe<P> X2 = fe_sqr(a.x);
  Fe<P> M = fe_add(fe_dbl(X2), X2);
  r.x = fe_sub(fe_sub(fe_sqr(M), S), S);
  r.y = fe_sub(fe_mul(M, fe_sub(S, r.x)), fe_mul(W, a.y));
  r.zz = fe_mul(V, a.zz);
  r.zzz = fe_mul(W, a.zzz);
  return r;
}

// acc += affine (madd-2008-s), all exceptional cases handled.  INL selects inlined multiplies
// (the MSM bucket loop) or out-of-line ones (everything else).
template <class P, bool INL = false> VDF_HD void xyzz_madd(XYZZ<P>& acc, const Affine<P>& b) {
  if (affine_is_identity(b)) return;
  if (xyzz_is_identity(acc)) { acc = xyzz_from_affine(b); return; }
  Fe<P> U2 = fe_mul_sel<INL>(b.x, acc.zz);
  Fe<P> S2 = fe_mul_sel<INL>(b.y, acc.zzz);
  Fe<P> Pp = fe_sub(U2, acc.x);
  Fe<P> Rr = fe_sub(S2, acc.y);
  if (fe_is_zero(Pp)) {
    if (fe_is_zero(Rr)) acc = xyzz_dbl_affine(b);
    else acc = xyzz_identity<P>();
    return;
  }
  Fe<P> PP = fe_mul_sel<INL>(Pp, Pp);
  Fe<P> PPP = fe_mul_sel<INL>(Pp, PP);
  Fe<P> Qq = fe_mul_sel<INL>(acc.x, PP);
  Fe<P> X3 = fe_sub(fe_sub(fe_sub(fe_mul_sel<INL>(Rr, Rr), PPP), Qq), Qq);
  Fe<P> Y3 = fe_sub(fe_mul_sel<INL>(Rr, fe_sub(Qq, X3)), fe_mul_sel<INL>(acc.y, PPP));
  acc.x = X3;
  acc.y = Y3;
  acc.zz = fe_mul_sel<INL>(acc.zz, PP);
  acc.zzz = fe_mul_sel<INL>(acc.zzz, PPP);
}

// acc += b (add-2008-s), all exceptional cases handled.
template <class P> VDF_HD void xyzz_add(XYZZ<P>& acc, const XYZZ<P>& b) {
  if (xyzz_is_identity(b)) return;
  if (xyzz_is_identity(acc)) { acc = b; return; }
  Fe<P> U1 = fe_mul(acc.x, b.zz);
  Fe<P> U2 = fe_mul(b.x, acc.zz);
  Fe<P> S1 = fe_mul(acc.y, b.zzz);
  Fe<P> S2 = fe_mul(b.y, acc.zzz);
  Fe<P> Pp = fe_sub(U2, U1);
  Fe<P> Rr = fe_sub(S2, S1);
  if (fe_is_zero(Pp)) {
    if (fe_is_zero(Rr)) acc = xyzz_dbl(acc);
    else acc = xyzz_identity<P>();
    return;
  }
  Fe<P> PP = fe_sqr(Pp);
  Fe<P> PPP = fe_mul(Pp, PP);
  Fe<P> Qq = fe_mul(U1, PP);
  Fe<P> X3 = fe_sub(fe_sub(fe_sub(fe_sqr(Rr), PPP), Qq), Qq);
  Fe<P> Y3 = fe_sub(fe_mul(Rr, fe_sub(Qq, X3)), fe_mul(S1, PPP));
  acc.x = X3;
  acc.y = Y3;
  acc.zz = fe_mul(fe_mul(acc.zz, b.zz), PP);
  acc.zzz = fe_mul(fe_mul(acc.zzz, b.zzz), PPP);
}

template <class P> VDF_HD Affine<P> affine_neg(const Affine<P>& a) {
  Affine<P> r;
  r.x = a.x;
  r.y = fe_neg(a.y);
  return r;
}

// XYZZ -> Jacobian without inversion: z = zz*zzz (= Z^5), x' = x*zz*zzz^2, y' = y*zz^3*zzz^2.
template <class P> VDF_HD Jac<P> xyzz_to_jac(const XYZZ<P>& a) {
  Jac<P> r;
  if (xyzz_is_identity(a)) {
    r.x = fe_zero<P>(); r.y = fe_zero<P>(); r.z = fe_zero<P>();
    return r;
  }
  Fe<P> zzz2 = fe_sqr(a.zzz);
  Fe<P> t = fe_mul(a.zz, zzz2);            // zz * zzz^2
  r.x = fe_mul(a.x, t);
  r.y = fe_mul(fe_mul(a.y, fe_sqr(a.zz)), t);
  r.z = fe_mul(a.zz, a.zzz);
  return r;
}

template <class P> VDF_HD Affine<P> xyzz_to_affine(const XYZZ<P>& a) {
  Affine<P> r;
  if (xyzz_is_identity(a)) { r.x = fe_zero<P>(); r.y = fe_zero<P>(); return r; }
  // 1/zzz, then 1/zz = (1/zzz)^2 * zz^2  (since zz^3 = zzz^2)
  Fe<P> izzz = fe_inv(a.zzz);
  Fe<P> izz = fe_mul(fe_sqr(izzz), fe_sqr(a.zz));
  r.x = fe_mul(a.x, izz);
  r.y = fe_mul(a.y, izzz);
  return r;
}

template <class P> VDF_HD XYZZ<P> jac_to_xyzz(const Jac<P>& a) {
  XYZZ<P> r;
  if (fe_is_zero(a.z)) return xyzz_identity<P>();
  r.x = a.x; r.y = a.y;
  r.zz = fe_sqr(a.z);
  r.zzz = fe_mul(r.zz, a.z);
  return r;
}

// [k] * affine for a 64-bit scalar (synthetic base generation, small multiples).
template <class P> VDF_HD XYZZ<P> xyzz_mul_u64(const Affine<P>& a, uint64_t k) {
  XYZZ<P> r = xyzz_identity<P>();
  for (int b = 63; b >= 0; --b) {
    r = xyzz_dbl(r);
    if ((k >> b) & 1ull) xyzz_madd(r, a);
  }
  return r;
}

template <class P> VDF_HD Affine<P> affine_load(const void* p) {
  Affine<P> r;
  r.x = fe_load<P>(p);
  r.y = fe_load<P>(reinterpret_cast<const char*>(p) + 32);
  return r;
}
template <class P> VDF_HD void affine_store(void* p, const Affine<P>& a) {
  fe_store<P>(p, a.x);
  fe_store<P>(reinterpret_cast<char*>(p) + 32, a.y);
}
template <class P> VDF_HD XYZZ<P> xyzz_load(const void* p) {
  const char* c = reinterpret_cast<const char*>(p);
  XYZZ<P> r;
  r.x = fe_load<P>(c); r.y = fe_load<P>(c + 32); r.zz = fe_load<P>(c + 64); r.zzz = fe_load<P>(c + 96);
  return r;
}
template <class P> VDF_HD void xyzz_store(void* p, const XYZZ<P>& a) {
  char* c = reinterpret_cast<char*>(p);
  fe_store<P>(c, a.x); fe_store<P>(c + 32, a.y); fe_store<P>(c + 64, a.zz); fe_store<P>(c + 96, a.zzz);
}

}  // namespace vdf
