// Quad-cooperative XYZZ group law for the latency-bound tail of the MSM (bucket fix-up, bucket
// reduction, Horner, point sum).
//
// Those kernels are chains of 30-50 dependent point operations run by a handful of waves: what
// matters is the latency of ONE addition, 14 dependent field multiplications when a lane does it
// alone (~1500 cycles each at one wave per SIMD).  Here FOUR adjacent lanes (a quad) own one point,
// lane q holding coordinate q (0 = X, 1 = Y, 2 = ZZ, 3 = ZZZ).  The add-2008-s / dbl-2008-s-1
// formulas have at most four independent multiplications per dependency level, so a quad evaluates
// an addition in 4 multiplication stages (a doubling in 3), exchanging operands with DPP
// quad_perm moves (one VALU instruction per 32-bit register, no LDS).  Same instruction stream in
// all four lanes; per-lane operand choice is v_cndmask on the lane's quad position.
//
// Memory format is unchanged (x, y, zz, zzz, 32 bytes each): lane q of a quad loads / stores the
// 32 bytes at offset 32*q, so a quad moves one 128-byte point as one contiguous line.
// Control flow around these calls must be QUAD-UNIFORM (all four lanes take the same branch).
#pragma once
#include "ec.cuh"

namespace vdf {

template <class P> struct QPoint {
  Fe<P> c;      // this lane's coordinate
  bool inf;     // identity flag, identical in the four lanes of the quad
};

// quad permute: lane i of every quad reads lane SEL_i of the same quad
template <int S0, int S1, int S2, int S3>
__device__ __forceinline__ uint32_t quad_perm_u32(uint32_t v) {
  constexpr int ctrl = S0 | (S1 << 2) | (S2 << 4) | (S3 << 6);
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, ctrl, 0xf, 0xf, false);
}
template <int S0, int S1, int S2, int S3, class P>
__device__ __forceinline__ Fe<P> quad_perm(const Fe<P>& a) {
  Fe<P> r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r.v[i] = quad_perm_u32<S0, S1, S2, S3>(a.v[i]);
  return r;
}
template <int S, class P> __device__ __forceinline__ Fe<P> quad_bcast(const Fe<P>& a) { return quad_perm<S, S, S, S>(a); }
template <int S> __device__ __forceinline__ bool quad_bcast_flag(bool f) { return quad_perm_u32<S, S, S, S>(f ? 1u : 0u) != 0; }

template <class P> __device__ __forceinline__ Fe<P> fe_select(bool c, const Fe<P>& a, const Fe<P>& b) {
  Fe<P> r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r.v[i] = c ? a.v[i] : b.v[i];
  return r;
}

__device__ __forceinline__ int quad_pos() { return (int)(threadIdx.x & 3u); }

template <class P> __device__ __forceinline__ QPoint<P> qpoint_identity() {
  QPoint<P> r;
  r.c = fe_zero<P>();
  r.inf = true;
  return r;
}

// gather a quad-distributed point into every lane / scatter it back (slow paths only)
template <class P> __device__ __forceinline__ XYZZ<P> qpoint_gather(const QPoint<P>& a) {
  XYZZ<P> r;
  r.x = quad_bcast<0>(a.c); r.y = quad_bcast<1>(a.c); r.zz = quad_bcast<2>(a.c); r.zzz = quad_bcast<3>(a.c);
  if (a.inf) r = xyzz_identity<P>();
  return r;
}
template <class P> __device__ __forceinline__ QPoint<P> qpoint_scatter(const XYZZ<P>& a) {
  QPoint<P> r;
  const int q = quad_pos();
  r.c = fe_select(q == 0, a.x, fe_select(q == 1, a.y, fe_select(q == 2, a.zz, a.zzz)));
  r.inf = xyzz_is_identity(a);
  return r;
}

// -a: the y coordinate (lane 1) changes sign
template <class P> __device__ __forceinline__ QPoint<P> qpoint_neg(QPoint<P> a) {
  a.c = fe_select(quad_pos() == 1, fe_neg(a.c), a.c);
  return a;
}

// 2 * a  (dbl-2008-s-1, a = 0) in THREE stages -- a squaring and two multiplications: M^2 needs only X^2, so it rides in the spare
// lane of the second stage (S, W, ZZ3 take three) instead of opening a stage of its own; the first stage is squarings in every
// lane (43 limb products instead of 64, fe_sqr_inl).  The Horner chain of the table-less path is 128 of these in a row.
template <class P> __device__ __attribute__((noinline)) QPoint<P> qpoint_dbl(QPoint<P> a) {
  if (a.inf) return a;
  const int q = quad_pos();
  const Fe<P> c2 = fe_dbl(a.c);                                   // lane 1: U = 2Y
  const Fe<P> A1 = fe_select(q == 1, c2, a.c);                    // lane 0: X, 1: U, 2: ZZ, 3: ZZZ
  const Fe<P> t1 = fe_sqr_inl(A1);                                // lane 0: X^2, lane 1: V = U^2
  const Fe<P> V = quad_bcast<1>(t1);
  const Fe<P> xx = quad_bcast<0>(t1);
  const Fe<P> M = fe_add(fe_dbl(xx), xx);                         // 3 X^2 (all lanes)
  const Fe<P> A2 = fe_select(q == 3, M, A1);
  const Fe<P> B2 = fe_select(q == 3, M, V);
  const Fe<P> t2 = fe_mul_inl(A2, B2);                            // lane 0: S = X*V, 1: W = U*V, 2: ZZ3 = ZZ*V, 3: M^2
  const Fe<P> S = quad_bcast<0>(t2);
  const Fe<P> W = quad_bcast<1>(t2);
  const Fe<P> mm = quad_bcast<3>(t2);
  const Fe<P> x3 = fe_sub(fe_sub(mm, S), S);                      // all lanes
  const Fe<P> A3 = fe_select(q == 0, M, W);
  const Fe<P> B3 = fe_select(q == 0, fe_sub(S, x3), a.c);
  const Fe<P> t3 = fe_mul_inl(A3, B3);                            // lane 0: M*(S - X3), 1: W*Y, 3: ZZZ3 = W*ZZZ
  const Fe<P> y3 = fe_sub(quad_bcast<0>(t3), quad_bcast<1>(t3));
  QPoint<P> r;
  r.c = fe_select(q == 0, x3, fe_select(q == 1, y3, fe_select(q == 2, t2, t3)));
  r.inf = false;
  return r;
}

// acc + b  (add-2008-s) in four multiplication stages, all exceptional cases handled
template <class P> __device__ __attribute__((noinline)) QPoint<P> qpoint_add(QPoint<P> acc, QPoint<P> b) {
  if (b.inf) return acc;
  if (acc.inf) return b;
  const int q = quad_pos();
  const Fe<P> bx = quad_perm<2, 3, 0, 1>(b.c);                    // lane 0: ZZ2, 1: ZZZ2, 2: X2, 3: Y2
  const Fe<P> t1 = fe_mul_inl(acc.c, bx);                         // lane 0: U1, 1: S1, 2: U2, 3: S2
  const Fe<P> d = fe_sub(quad_perm<2, 3, 0, 1>(t1), t1);          // lane 0: P = U2-U1, 1: R = S2-S1
  const bool dz = fe_is_zero(d);
  if (quad_bcast_flag<0>(dz)) {                                   // same x: doubling or opposite points
    if (quad_bcast_flag<1>(dz)) return qpoint_dbl<P>(acc);
    return qpoint_identity<P>();
  }
  const Fe<P> A2 = fe_select(q < 2, d, acc.c);
  const Fe<P> B2 = fe_select(q < 2, d, b.c);
  const Fe<P> t2 = fe_mul_inl(A2, B2);                            // lane 0: PP, 1: RR, 2: ZZ1*ZZ2, 3: ZZZ1*ZZZ2
  const Fe<P> pp = quad_bcast<0>(t2);
  const Fe<P> u1 = quad_bcast<0>(t1);
  const Fe<P> A3 = fe_select(q == 0, d, fe_select(q == 1, u1, t2));
  const Fe<P> t3 = fe_mul_inl(A3, pp);                            // lane 0: PPP, 1: Q = U1*PP, 2: ZZ3
  const Fe<P> ppp = quad_bcast<0>(t3);
  const Fe<P> qq = quad_bcast<1>(t3);
  const Fe<P> rr = quad_bcast<1>(t2);
  const Fe<P> x3 = fe_sub(fe_sub(fe_sub(rr, ppp), qq), qq);       // all lanes
  const Fe<P> T = fe_sub(qq, x3);
  const Fe<P> s1 = quad_bcast<1>(t1);
  const Fe<P> A4 = fe_select(q == 1, d, fe_select(q == 2, s1, t2));
  const Fe<P> B4 = fe_select(q == 1, T, ppp);
  const Fe<P> t4 = fe_mul_inl(A4, B4);                            // lane 1: R*T, 2: S1*PPP, 3: ZZZ3
  const Fe<P> y3 = fe_sub(t4, quad_bcast<2>(t4));                 // lane 1: Y3
  QPoint<P> r;
  r.c = fe_select(q == 0, x3, fe_select(q == 1, y3, fe_select(q == 2, t3, t4)));
  r.inf = false;
  return r;
}

// memory: 128-byte XYZZ (identity = all-zero zz); lane q touches bytes [32q, 32q+32)
template <class P> __device__ __forceinline__ QPoint<P> qpoint_load(const char* p) {
  QPoint<P> r;
  r.c = fe_load<P>(p + 32 * quad_pos());
  const bool z = fe_is_zero(r.c);
  r.inf = quad_bcast_flag<2>(z);                                  // zz == 0
  return r;
}
// the same for a point k_accumulate flushed in its lazy domain (coordinates in [0, 2m + eps), identity all-zero):
// every lane brings its own coordinate to canonical form, one fe_canon per lane instead of four per flush
template <class P> __device__ __forceinline__ QPoint<P> qpoint_load_lazy(const char* p) {
  QPoint<P> r;
  r.c = fe_canon(fe_load<P>(p + 32 * quad_pos()));
  const bool z = fe_is_zero(r.c);
  r.inf = quad_bcast_flag<2>(z);                                  // zz == 0
  return r;
}
template <class P> __device__ __forceinline__ void qpoint_store(char* p, const QPoint<P>& a) {
  fe_store<P>(p + 32 * quad_pos(), a.inf ? fe_zero<P>() : a.c);
}

// exchange with the quad `mask` lanes away (mask a multiple of 4): same coordinate, other point
template <class P> __device__ __forceinline__ QPoint<P> qpoint_shfl_xor(const QPoint<P>& a, int mask) {
  QPoint<P> r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r.c.v[i] = __shfl_xor(a.c.v[i], mask, 64);
  r.inf = __shfl_xor((int)a.inf, mask, 64) != 0;
  return r;
}
// sum over the 16 quads of a wavefront (4 butterfly steps); result in every quad
template <class P> __device__ __forceinline__ QPoint<P> qpoint_wave_sum(QPoint<P> v) {
#pragma unroll 1
  for (int m = 32; m >= 4; m >>= 1) {
    QPoint<P> o = qpoint_shfl_xor(v, m);
    v = qpoint_add<P>(v, o);
  }
  return v;
}

}  // namespace vdf
