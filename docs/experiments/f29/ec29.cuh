// XYZZ group law over the unsaturated-limb field (f29.cuh) for the MSM back end: bucket fix-up,
// bucket reduction and the final Horner all run at one or two waves per SIMD, where the 9 x 29-bit
// multiply has ~1.5x lower latency than the saturated 8 x 32-bit one (943 vs 1462 cycles per
// dependent multiply, tools/gpu_mul_rate.py).
//
// Invariants of a stored / passed XYZZ29 (all coordinates normalized, i.e. limbs < 2^29 but the top):
//   X < 10m, Y < 6.3m, ZZ, ZZZ < 2m.   Identity: `inf` (stored as ZZ = 0 exactly).
// Every function below restores them; the bound of each intermediate is in the comments (multiples
// of m; a product of bounds a*b gives a mont29 output < a*b/128 + 1).
#pragma once
#include "f29.cuh"
#include "ec.cuh"

namespace vdf {

template <class Q> struct XYZZ29 {
  F29<Q> x, y, zz, zzz;
  bool inf;
};

template <class Q> __device__ __forceinline__ F29<Q> f29_zero() {
  F29<Q> r;
#pragma unroll
  for (int i = 0; i < 9; ++i) r.v[i] = 0;
  return r;
}
template <class Q> __device__ __forceinline__ XYZZ29<Q> xyzz29_identity() {
  XYZZ29<Q> r;
  r.x = f29_zero<Q>(); r.y = f29_zero<Q>(); r.zz = f29_zero<Q>(); r.zzz = f29_zero<Q>();
  r.inf = true;
  return r;
}

// exact v == 0 (mod m) for a normalized value < K*m (rare paths only)
template <class Q, int K> __device__ __forceinline__ bool f29_is_zero_mod_k(F29<Q> a) {
  for (int k = 0; k < K; ++k) f29_cond_sub(a);
  uint32_t o = 0;
#pragma unroll
  for (int i = 0; i < 9; ++i) o |= a.v[i];
  return o == 0;
}

// 2 * a  (dbl-2008-s-1, a = 0)
template <class Q> __device__ __forceinline__ XYZZ29<Q> xyzz29_dbl_inl(const XYZZ29<Q>& a) {
  if (a.inf) return a;
  XYZZ29<Q> r;
  F29<Q> U;                                                 // 2Y < 12.6m  (Y normalized: limbs < 2^30)
#pragma unroll
  for (int i = 0; i < 9; ++i) U.v[i] = 2u * a.y.v[i];
  f29_normalize(U);
  const F29<Q> V = f29_mul(U, U);                           // < 2.3m
  const F29<Q> W = f29_mul(U, V);                           // < 1.3m
  const F29<Q> S = f29_mul(a.x, V);                         // < 1.2m
  const F29<Q> X2 = f29_mul(a.x, a.x);                      // < 1.8m
  F29<Q> M;                                                 // 3*X^2 < 5.4m
#pragma unroll
  for (int i = 0; i < 9; ++i) M.v[i] = 3u * X2.v[i];
  f29_normalize(M);
  const F29<Q> MM = f29_mul(M, M);                          // < 1.3m
  F29<Q> X3;                                                // M^2 - 2S + 4m < 5.3m   (2S < 2.4m < 3m)
#pragma unroll
  for (int i = 0; i < 9; ++i) X3.v[i] = MM.v[i] + Q::B4[i] - 2u * S.v[i];
  f29_normalize(X3);
  const F29<Q> T = f29_sub(S, X3, Q::B16);                  // < 17.2m, limbs < 2^31
  const F29<Q> MT = f29_mul(M, T);                          // < 1.8m
  const F29<Q> WY = f29_mul(W, a.y);                        // < 1.1m
  r.x = X3;
  r.y = f29_sub(MT, WY, Q::B4);                             // < 5.8m
  f29_normalize(r.y);
  r.zz = f29_mul(V, a.zz);
  r.zzz = f29_mul(W, a.zzz);
  r.inf = false;
  return r;
}

// acc += b  (add-2008-s) with the exceptional cases
template <class Q> __device__ __forceinline__ void xyzz29_add_inl(XYZZ29<Q>& acc, const XYZZ29<Q>& b) {
  if (b.inf) return;
  if (acc.inf) { acc = b; return; }
  const F29<Q> U1 = f29_mul(acc.x, b.zz);                   // < 2m
  const F29<Q> U2 = f29_mul(b.x, acc.zz);
  const F29<Q> S1 = f29_mul(acc.y, b.zzz);
  const F29<Q> S2 = f29_mul(b.y, acc.zzz);
  F29<Q> Pp = f29_sub(U2, U1, Q::B4);                       // < 6m
  f29_normalize(Pp);
  F29<Q> Rr = f29_sub(S2, S1, Q::B4);                       // < 6m
  f29_normalize(Rr);
  const F29<Q> PP = f29_mul(Pp, Pp);                        // < 1.3m
  if (PP.v[0] <= 1u && f29_is_zero_mod_k<Q, 2>(PP)) {       // same x
    if (f29_is_zero_mod_k<Q, 6>(Rr)) acc = xyzz29_dbl_inl(acc);
    else acc = xyzz29_identity<Q>();
    return;
  }
  const F29<Q> PPP = f29_mul(Pp, PP);                       // < 1.1m
  const F29<Q> Qq = f29_mul(U1, PP);                        // < 1.1m
  const F29<Q> RR = f29_mul(Rr, Rr);                        // < 1.3m
  F29<Q> X3 = f29_sub3(RR, PPP, Qq, Q::B8_31);              // < 9.3m
  f29_normalize(X3);
  const F29<Q> T = f29_sub(Qq, X3, Q::B16);                 // < 17.1m
  const F29<Q> RT = f29_mul(Rr, T);                         // < 1.9m
  const F29<Q> SP = f29_mul(S1, PPP);                       // < 1.1m
  F29<Q> Y3 = f29_sub(RT, SP, Q::B4);                       // < 5.9m
  f29_normalize(Y3);
  acc.x = X3;
  acc.y = Y3;
  acc.zz = f29_mul(f29_mul(acc.zz, b.zz), PP);
  acc.zzz = f29_mul(f29_mul(acc.zzz, b.zzz), PPP);
}

// Out-of-line entry points: the multiplies inside stay inlined (no per-multiply call overhead), the
// group operation itself is one function per field, so a kernel with many additions stays small.
template <class Q> __device__ __attribute__((noinline)) void xyzz29_add(XYZZ29<Q>& acc, const XYZZ29<Q>& b) { xyzz29_add_inl(acc, b); }
template <class Q> __device__ __attribute__((noinline)) void xyzz29_dbl_to(XYZZ29<Q>& r, const XYZZ29<Q>& a) { r = xyzz29_dbl_inl(a); }
template <class Q> __device__ __forceinline__ XYZZ29<Q> xyzz29_dbl(const XYZZ29<Q>& a) { XYZZ29<Q> r; xyzz29_dbl_to(r, a); return r; }

// raw 9-limb storage (no conversion): 144 B per point, identity = all-zero ZZ
template <class Q> __device__ __forceinline__ F29<Q> f29_load_raw(const char* p) {
  F29<Q> r;
  const uint4 a = *reinterpret_cast<const uint4*>(p);
  const uint4 b = *reinterpret_cast<const uint4*>(p + 16);
  r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w;
  r.v[4] = b.x; r.v[5] = b.y; r.v[6] = b.z; r.v[7] = b.w;
  r.v[8] = *reinterpret_cast<const uint32_t*>(p + 32);
  return r;
}
template <class Q> __device__ __forceinline__ void f29_store_raw(char* p, const F29<Q>& a) {
  *reinterpret_cast<uint4*>(p) = make_uint4(a.v[0], a.v[1], a.v[2], a.v[3]);
  *reinterpret_cast<uint4*>(p + 16) = make_uint4(a.v[4], a.v[5], a.v[6], a.v[7]);
  *reinterpret_cast<uint32_t*>(p + 32) = a.v[8];
}
// A stored point takes a 192-byte slot: each 36-byte coordinate starts on a 16-byte boundary
// (offsets 0 / 48 / 96 / 144) so that it moves as two dwordx4 and one dword.
static constexpr int XYZZ29_SLOT = 192;
template <class Q> __device__ __forceinline__ XYZZ29<Q> xyzz29_load(const char* p) {
  XYZZ29<Q> r;
  r.x = f29_load_raw<Q>(p);
  r.y = f29_load_raw<Q>(p + 48);
  r.zz = f29_load_raw<Q>(p + 96);
  r.zzz = f29_load_raw<Q>(p + 144);
  uint32_t o = 0;
#pragma unroll
  for (int i = 0; i < 9; ++i) o |= r.zz.v[i];
  r.inf = (o == 0);
  return r;
}
template <class Q> __device__ __forceinline__ void xyzz29_store(char* p, const XYZZ29<Q>& a) {
  if (a.inf) {
    const F29<Q> z = f29_zero<Q>();
    f29_store_raw<Q>(p, z); f29_store_raw<Q>(p + 48, z); f29_store_raw<Q>(p + 96, z); f29_store_raw<Q>(p + 144, z);
    return;
  }
  f29_store_raw<Q>(p, a.x); f29_store_raw<Q>(p + 48, a.y); f29_store_raw<Q>(p + 96, a.zz); f29_store_raw<Q>(p + 144, a.zzz);
}

template <class Q> __device__ __forceinline__ XYZZ29<Q> xyzz29_shfl_xor(const XYZZ29<Q>& a, int mask) {
  XYZZ29<Q> r;
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    r.x.v[i] = __shfl_xor(a.x.v[i], mask, 64);
    r.y.v[i] = __shfl_xor(a.y.v[i], mask, 64);
    r.zz.v[i] = __shfl_xor(a.zz.v[i], mask, 64);
    r.zzz.v[i] = __shfl_xor(a.zzz.v[i], mask, 64);
  }
  r.inf = __shfl_xor((int)a.inf, mask, 64) != 0;
  return r;
}
// wavefront all-reduce (6 butterfly steps of XYZZ additions)
template <class Q> __device__ __forceinline__ XYZZ29<Q> xyzz29_wave_sum(XYZZ29<Q> v) {
#pragma unroll 1
  for (int m = 32; m >= 1; m >>= 1) {
    XYZZ29<Q> o = xyzz29_shfl_xor(v, m);
    xyzz29_add(v, o);
  }
  return v;
}

// ordinary (R = 2^256, 8 x 32) Jacobian point -> XYZZ29 (the other ranks' partials in point_sum)
template <class P> __device__ __forceinline__ XYZZ29<typename P29Of<P>::type> xyzz29_from_jac(const Jac<P>& j) {
  using Q = typename P29Of<P>::type;
  if (fe_is_zero(j.z)) return xyzz29_identity<Q>();
  const F29<Q> ONE = f29_const<Q>(Q::ONE);
  XYZZ29<Q> r;
  const F29<Q> z = f29_mul(f29_from_fe_x32<P>(j.z), ONE);
  r.x = f29_mul(f29_from_fe_x32<P>(j.x), ONE);
  r.y = f29_mul(f29_from_fe_x32<P>(j.y), ONE);
  r.zz = f29_mul(z, z);
  r.zzz = f29_mul(r.zz, z);
  r.inf = false;
  return r;
}
// XYZZ29 -> ordinary Jacobian: z = zz*zzz, x' = x*zz*zzz^2, y' = y*zz^3*zzz^2 (no inversion)
template <class P> __device__ __forceinline__ Jac<P> xyzz29_to_jac(const XYZZ29<typename P29Of<P>::type>& a) {
  using Q = typename P29Of<P>::type;
  Jac<P> r;
  if (a.inf) { r.x = fe_zero<P>(); r.y = fe_zero<P>(); r.z = fe_zero<P>(); return r; }
  const F29<Q> zzz2 = f29_mul(a.zzz, a.zzz);
  const F29<Q> t = f29_mul(a.zz, zzz2);
  const F29<Q> zz2 = f29_mul(a.zz, a.zz);
  r.x = f29_to_fe<P>(f29_mul(a.x, t));
  r.y = f29_to_fe<P>(f29_mul(f29_mul(a.y, zz2), t));
  r.z = f29_to_fe<P>(f29_mul(a.zz, a.zzz));
  return r;
}

}  // namespace vdf
