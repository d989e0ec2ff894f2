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

// Mixed addition in the lazy domain (fe.cuh) for the bucket loops of msm.hip and msm_direct.hip, with the SIGN of the
// accumulator tracked beside it: the point a lane holds is  sigma * (x/zz, y/zzz),  sigma = -1 when `flip` is set.
// madd-2008-s needs P = U2 - X1, R = S2 - Y1 and Y3 = R (Q - X3) - Y1 PPP: a difference of two products.  With the
// operands of the FIRST subtraction swapped (P' = X1 - U2 = -P: free) PPP' = P' PP = -PPP and
//     X3 = R^2 + PPP' - 2Q          Y3 = R (Q - X3) + Y1 PPP'
// is a SUM of two products -- one column scan, one Montgomery reduction (fe_mul2_lazy) -- while ZZZ1 PPP' = -ZZZ3.
// Slack (how far above 2m a lazy coordinate may sit, fe.cuh): a difference inherits its MINUEND's slack, a product gives
// eps + half the sum of its factors' slacks, the product pair 2 eps + half the sum of all four.  R = S2 - Y1 has the
// minuend S2 = y_b * ZZZ1 with y_b canonical: below 2m, slack 0, whatever Y1 is -- so X3 (minuend R^2) has slack eps,
// Q - X3 (minuend Q = X1 PP) 1.5 eps + d_x <= 2.5 eps, PPP' 1.5 eps + d_x, and
//     d_y' <= 2 eps + (0 + 2.5 eps)/2 + (d_y + 2.5 eps)/2 = 4.5 eps + d_y / 2      d_zz' <= 2 eps + d_zz / 2      d_zzz' <= 2.25 eps + d_zzz / 2
// CONTRACT: every coordinate stays below 2m + 9 eps for a chain of any length (tools/ubench/madd_check.hip runs 10^4-long
// chains with cancellations and checks the bound at every step).  (With R' = Y1 - S2, as this function first had it, the
// minuend was Y1 and d_y' <= 3.25 eps + 1.5 d_y + 0.5 d_x: safe only with overwhelming probability, not by construction.)
// (X3, Y3, ZZ3, -ZZZ3) is the negated sum: instead of negating a coordinate (8+ instructions per addition) the sign
// moves into `flip`, the NEXT point is negated before it is added (the loops negate by the digit's sign anyway:
// sigma (A + sigma b) = sigma A + b), and a pending sign is applied once when the accumulator is flushed.
// Per addition: 8 products + 1 product pair + 6 subtractions, ~2,490 VALU instructions against ~2,595 for 10 + 7.
// `b` is the point to add to the STORED accumulator (the caller has applied digit sign XOR flip), never the identity.
// the two squarings of an addition (P^2, R^2): 43 limb products instead of 64, the same value bit for bit (fe.cuh fe_sqr_lazy)
template <class P> __device__ __forceinline__ Fe<P> fe_sqr_madd(const Fe<P>& a) {
#ifdef VDF_MADD_NO_SQR  // A/B build only: the general product
  return fe_mul_lazy(a, a);
#else
  return fe_sqr_lazy(a);
#endif
}
template <class P>
__device__ __forceinline__ void xyzz_madd_lazy(XYZZ<P>& acc, bool& have, bool& flip, const Affine<P>& b) {
  if (!have) { acc = xyzz_from_affine(b); have = true; flip = false; return; }
#ifdef VDF_MADD_V1     // A/B build only (tools/ab_madd.sh): round 3's formulas, ten products and seven subtractions, no sign tracking
  {
    const Fe<P> U2 = fe_mul_lazy(b.x, acc.zz);
    const Fe<P> S2 = fe_mul_lazy(b.y, acc.zzz);
    const Fe<P> Pp = fe_sub_lazy(U2, acc.x);
    const Fe<P> Rr = fe_sub_lazy(S2, acc.y);
    if (Pp.v[0] <= 2u && fe_is_zero(fe_canon(Pp))) {
      if (fe_is_zero(fe_canon(Rr))) acc = xyzz_dbl_affine(b);
      else have = false;
      return;
    }
    const Fe<P> PP = fe_mul_lazy(Pp, Pp);
    const Fe<P> PPP = fe_mul_lazy(Pp, PP);
    const Fe<P> Qq = fe_mul_lazy(acc.x, PP);
    const Fe<P> X3 = fe_sub_lazy(fe_sub_lazy(fe_sub_lazy(fe_mul_lazy(Rr, Rr), PPP), Qq), Qq);
    const Fe<P> Y3 = fe_sub_lazy(fe_mul_lazy(Rr, fe_sub_lazy(Qq, X3)), fe_mul_lazy(acc.y, PPP));
    acc.x = X3;
    acc.y = Y3;
    acc.zz = fe_mul_lazy(acc.zz, PP);
    acc.zzz = fe_mul_lazy(acc.zzz, PPP);
    return;
  }
#endif
  const Fe<P> U2 = fe_mul_lazy(b.x, acc.zz);
  const Fe<P> S2 = fe_mul_lazy(b.y, acc.zzz);
  const Fe<P> Pn = fe_sub_lazy(acc.x, U2);         // P' = -P
#ifdef VDF_MADD_R4     // A/B build only: round 4's operand order (R' = Y1 - S2; same instruction count, no slack contraction)
  const Fe<P> Rr = fe_sub_lazy(acc.y, S2);
#else
  const Fe<P> Rr = fe_sub_lazy(S2, acc.y);         // R itself: the minuend is a product with a canonical factor, below 2m (no slack)
#endif
  // P == 0 (mod m) means P in {0, m, 2m}; m == 1 (mod 2^32), so the low limb is 0, 1 or 2: cheap filter
  if (Pn.v[0] <= 2u && fe_is_zero(fe_canon(Pn))) {
    if (fe_is_zero(fe_canon(Rr))) acc = xyzz_dbl_affine(b);          // same point: double (canonical output), sign unchanged
    else have = false;                                                // opposite points: identity
    return;
  }
  const Fe<P> PP = fe_sqr_madd(Pn);
  const Fe<P> PPPn = fe_mul_lazy(Pn, PP);
  const Fe<P> Qq = fe_mul_lazy(acc.x, PP);
  const Fe<P> X3 = fe_sub_lazy(fe_sub_lazy(fe_sqr_madd(Rr), Qq), fe_sub_lazy(Qq, PPPn));
#ifdef VDF_MADD_R4
  acc.y = fe_mul2_lazy(Rr, fe_sub_lazy(X3, Qq), acc.y, PPPn);
#else
  acc.y = fe_mul2_lazy(Rr, fe_sub_lazy(Qq, X3), acc.y, PPPn);
#endif
  acc.x = X3;
  acc.zz = fe_mul_lazy(acc.zz, PP);
  acc.zzz = fe_mul_lazy(acc.zzz, PPPn);
  flip = !flip;
}
// the accumulator as a plain XYZZ point (lazy coordinates): a pending sign goes into y
template <class P>
__device__ __forceinline__ XYZZ<P> xyzz_lazy_resolve(XYZZ<P> acc, bool have, bool flip) {
  if (!have) return xyzz_identity<P>();
  if (flip) acc.y = fe_neg_lazy(acc.y);
  return acc;
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

#ifdef __HIPCC__
// acc + b for two XYZZ points by ONE lane in the lazy domain, sign-tracked like xyzz_madd_lazy (the lane-serial fix-up of
// msm.hip: 12 products + 1 product pair + 6 subtractions, ~3,300 instructions, where the quad-cooperative law spends 4 lanes x
// ~2,070).  `b`: CANONICAL coordinates, not the identity, already negated by the caller when `flip` is set.  add-2008-s with
// P' = U1 - U2 = -P:  X3 = R^2 + PPP' - 2Q,  Y3 = R (Q - X3) + S1 PPP' (a SUM: fe_mul2_lazy),  ZZ3 = ZZ1 ZZ2 PP,
// ZZZ1 ZZZ2 PPP' = -ZZZ3: the negated sum, so `flip` toggles.  Every product has a factor that is canonical or a product with
// a canonical factor (below 2m: no slack), so the stored coordinates stay below 2m + 4 eps whatever the chain's length.
template <class P>
__device__ __forceinline__ void xyzz_add_lazy(XYZZ<P>& acc, bool& have, bool& flip, const XYZZ<P>& b) {
  if (!have) { acc = b; have = true; flip = false; return; }
  const Fe<P> U1 = fe_mul_lazy(acc.x, b.zz);
  const Fe<P> U2 = fe_mul_lazy(b.x, acc.zz);
  const Fe<P> S1 = fe_mul_lazy(acc.y, b.zzz);
  const Fe<P> S2 = fe_mul_lazy(b.y, acc.zzz);
  const Fe<P> Pn = fe_sub_lazy(U1, U2);
  const Fe<P> Rr = fe_sub_lazy(S2, S1);
  if (Pn.v[0] <= 2u && fe_is_zero(fe_canon(Pn))) {                     // same x: the canonical law decides (double / identity)
    XYZZ<P> a = xyzz_lazy_resolve<P>(acc, have, flip);
    a.x = fe_canon(a.x); a.y = fe_canon(a.y); a.zz = fe_canon(a.zz); a.zzz = fe_canon(a.zzz);
    XYZZ<P> bb = b;
    if (flip) bb.y = fe_neg(bb.y);                                     // the caller negated it for the stored sign: undo
    xyzz_add(a, bb);
    acc = a; flip = false; have = !xyzz_is_identity(a);
    return;
  }
  const Fe<P> PP = fe_sqr_madd(Pn);
  const Fe<P> PPPn = fe_mul_lazy(Pn, PP);
  const Fe<P> Qq = fe_mul_lazy(U1, PP);
  const Fe<P> X3 = fe_sub_lazy(fe_sub_lazy(fe_sqr_madd(Rr), Qq), fe_sub_lazy(Qq, PPPn));
  acc.y = fe_mul2_lazy(Rr, fe_sub_lazy(Qq, X3), S1, PPPn);
  acc.x = X3;
  acc.zz = fe_mul_lazy(fe_mul_lazy(acc.zz, b.zz), PP);
  acc.zzz = fe_mul_lazy(fe_mul_lazy(acc.zzz, b.zzz), PPPn);
  flip = !flip;
}
#endif

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
