// Diagnostic (not product code): the bucket loop's sign-tracked mixed addition (ec.cuh xyzz_madd_lazy, fe_mul2_lazy,
// fe_neg_lazy) against the plain canonical formulas, lane by lane, on random points.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -I vdf_amd/csrc tools/ubench/madd_check.hip -o tools/ubench/madd_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#include "fe.cuh"
#include "ec.cuh"
using namespace vdf;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)
typedef FpParams P;

__device__ Fe<P> rnd_fe(uint32_t& s) {
  Fe<P> r;
  for (int i = 0; i < 8; ++i) { s = s * 1664525u + 1013904223u; r.v[i] = s ^ (s >> 13); }
  r.v[7] &= 0x3fffffffu;
  return r;
}
// a point on y^2 = x^3 + 5 from a seed: [k] G by double-and-add with the canonical formulas
__device__ Affine<P> rnd_pt(uint32_t& s) {
  Affine<P> g; g.x = fe_neg(fe_one<P>()); g.y = fe_from_u64<P>(2);
  s = s * 1664525u + 1013904223u;
  return xyzz_to_affine(xyzz_mul_u64(g, ((uint64_t)s << 20) | 12345u));
}

// (0) fe_sqr_lazy against fe_mul_lazy(a, a), BIT FOR BIT (before any canonicalisation: both compute (a^2 + q m) / 2^256), over
// all of [0, 2^256): random words, values around 2^255 and 2^256 - 1 (the doubled number's ninth limb), words of all ones / top
// bits only (every a_j >> 31 carry into the next limb of 2a), small values.
__global__ void k_sqr(uint32_t* bad) {
  uint32_t s = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 71u;
  for (int it = 0; it < 512; ++it) {
    Fe<P> a;
    for (int i = 0; i < 8; ++i) { s = s * 1664525u + 1013904223u; a.v[i] = s ^ (s >> 13); }
    const int kind = it & 15;
    if (kind == 1) for (int i = 0; i < 8; ++i) a.v[i] = 0xFFFFFFFFu;
    if (kind == 2) for (int i = 0; i < 8; ++i) a.v[i] = 0x80000000u;
    if (kind == 3) for (int i = 0; i < 8; ++i) a.v[i] = (a.v[i] & 1u) ? 0xFFFFFFFFu : 0u;
    if (kind == 4) { for (int i = 0; i < 7; ++i) a.v[i] = 0; a.v[7] = 0x80000000u; }
    if (kind == 5) { for (int i = 0; i < 7; ++i) a.v[i] = 0xFFFFFFFFu; a.v[7] = 0x7FFFFFFFu; }
    if (kind == 6) for (int i = 1; i < 8; ++i) a.v[i] = 0;
    if (kind == 7) for (int i = 0; i < 8; ++i) a.v[i] |= 0x80000000u;
    if (kind == 8) for (int i = 0; i < 8; ++i) a.v[i] &= 0x7FFFFFFFu;
    if (kind == 9) for (int i = 0; i < 8; ++i) a.v[i] = (it >> 4) == i ? 0xFFFFFFFFu : 0u;
    const Fe<P> want = fe_mul_lazy(a, a), got = fe_sqr_lazy(a);
    bool same = true;
    for (int i = 0; i < 8; ++i) same = same && want.v[i] == got.v[i];
    if (!same) atomicAdd(&bad[15], 1u);
  }
}

__global__ void k_check(uint32_t* bad, int steps) {
  uint32_t s = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 17u;
  // (1) fe_mul2_lazy vs two products and an addition
  for (int it = 0; it < 64; ++it) {
    Fe<P> a = rnd_fe(s), b = rnd_fe(s), c = rnd_fe(s), d = rnd_fe(s);
    Fe<P> want = fe_add(fe_mul_inl(a, b), fe_mul_inl(c, d));
    Fe<P> got = fe_canon(fe_mul2_lazy(a, b, c, d));
    if (!fe_eq(want, got)) atomicAdd(&bad[0], 1u);
    Fe<P> n = fe_canon(fe_neg_lazy(a));
    if (!fe_eq(n, fe_neg(a))) atomicAdd(&bad[1], 1u);
  }
  // (2) chains of mixed additions with random signs
  XYZZ<P> ref = xyzz_identity<P>();
  XYZZ<P> acc = xyzz_identity<P>();
  bool have = false, flip = false;
  for (int k = 0; k < steps; ++k) {
    Affine<P> pt = rnd_pt(s);
    s = s * 1664525u + 1013904223u;
    const bool neg = (s >> 9) & 1u;
    Affine<P> t = pt;
    if (neg) t.y = fe_neg(t.y);
    xyzz_madd<P, true>(ref, t);
    Affine<P> u = pt;
    if (neg != (have && flip)) u.y = fe_neg(u.y);
    xyzz_madd_lazy<P>(acc, have, flip, u);
    XYZZ<P> r = xyzz_lazy_resolve<P>(acc, have, flip);
    r.x = fe_canon(r.x); r.y = fe_canon(r.y); r.zz = fe_canon(r.zz); r.zzz = fe_canon(r.zzz);
    const Affine<P> A = xyzz_to_affine(ref), B = xyzz_to_affine(r);
    if (!fe_eq(A.x, B.x) || !fe_eq(A.y, B.y)) { atomicAdd(&bad[2], 1u); if (k < 8) atomicAdd(&bad[3 + k], 1u); }
  }
}

// value < 2m + 2^130 (2^130 ~ 29 eps; the proven bound is 2m + 9 eps)?  t = a - 2m: a borrow means a < 2m; otherwise t < 2^130.
__device__ bool below_bound(const Fe<P>& a) {
  constexpr uint64_t D1 = 2ull * P::MOD[1], D2 = 2ull * P::MOD[2] + (D1 >> 32), D3 = 2ull * P::MOD[3] + (D2 >> 32);
  const uint32_t T[8] = {2u, (uint32_t)D1, (uint32_t)D2, (uint32_t)D3, (uint32_t)(D3 >> 32), 0u, 0u, 0x80000000u};
  uint32_t t[8];
  uint64_t borrow = 0;
  for (int i = 0; i < 8; ++i) {
    const uint64_t d = (uint64_t)a.v[i] - T[i] - borrow;
    t[i] = (uint32_t)d;
    borrow = (d >> 63) & 1u;
  }
  if (borrow) return true;
  return t[7] == 0 && t[6] == 0 && t[5] == 0 && t[4] < 4u;
}

// (3) LONG chains: `steps` additions per lane into ONE accumulator -- a walk k G, (k+1) G, ... with random signs, so the
// partial sums do not repeat -- and every `cancel_every` additions the negated running sum is added (cancel to the identity:
// `have` drops, the next point restarts the accumulator with `flip` whatever it was).  After EVERY addition each stored
// coordinate must be below 2m + 2^130 and the resolved point must equal the canonical reference.
__global__ void k_long(uint32_t* bad, int steps, int cancel_every) {
  uint32_t s = (blockIdx.x * blockDim.x + threadIdx.x) * 2246822519u + 99u;
  Affine<P> g; g.x = fe_neg(fe_one<P>()); g.y = fe_from_u64<P>(2);
  s = s * 1664525u + 1013904223u;
  XYZZ<P> walk = xyzz_mul_u64(g, ((uint64_t)s << 8) | 77u);
  XYZZ<P> ref = xyzz_identity<P>();
  XYZZ<P> acc = xyzz_identity<P>();
  bool have = false, flip = false;
  for (int k = 0; k < steps; ++k) {
    Affine<P> pt;
    bool neg;
    if (cancel_every && k % cancel_every == cancel_every - 1 && !xyzz_is_identity(ref)) {
      pt = xyzz_to_affine(ref); neg = true;                   // add -(running sum): the identity
    } else {
      xyzz_madd<P, true>(walk, g);
      pt = xyzz_to_affine(walk);
      s = s * 1664525u + 1013904223u;
      neg = (s >> 11) & 1u;
    }
    Affine<P> t = pt;
    if (neg) t.y = fe_neg(t.y);
    xyzz_madd<P, true>(ref, t);
    Affine<P> u = pt;
    if (neg != (have && flip)) u.y = fe_neg(u.y);
    xyzz_madd_lazy<P>(acc, have, flip, u);
    if (have && !(below_bound(acc.x) && below_bound(acc.y) && below_bound(acc.zz) && below_bound(acc.zzz))) atomicAdd(&bad[12], 1u);
    if ((k & 63) == 63 || k + 1 == steps || !have) {          // the comparison costs an inversion: every 64 steps, at cancellations, at the end
      XYZZ<P> r = xyzz_lazy_resolve<P>(acc, have, flip);
      r.x = fe_canon(r.x); r.y = fe_canon(r.y); r.zz = fe_canon(r.zz); r.zzz = fe_canon(r.zzz);
      if (xyzz_is_identity(ref) != !have) atomicAdd(&bad[13], 1u);
      else if (have) {
        const Affine<P> A = xyzz_to_affine(ref), B = xyzz_to_affine(r);
        if (!fe_eq(A.x, B.x) || !fe_eq(A.y, B.y)) atomicAdd(&bad[13], 1u);
      }
    }
    if (!have) atomicAdd(&bad[14], 1u);                        // cancellations seen (must be > 0)
  }
}

int main(int argc, char** argv) {
  const int long_steps = argc > 1 ? atoi(argv[1]) : 10240;
  uint32_t* d; CK(hipMalloc(&d, 64)); CK(hipMemset(d, 0, 64));
  hipLaunchKernelGGL(k_sqr, dim3(16), dim3(64), 0, 0, d);
  hipLaunchKernelGGL(k_check, dim3(8), dim3(64), 0, 0, d, 24);
  CK(hipDeviceSynchronize());
  uint32_t h[16]; CK(hipMemcpy(h, d, 64, hipMemcpyDeviceToHost));
  printf("fe_sqr_lazy vs fe_mul_lazy(a, a), bit for bit (524288 values): mismatches %u\n", h[15]);
  printf("fe_mul2_lazy mismatches %u, fe_neg_lazy mismatches %u, chain mismatches %u (first steps:", h[0], h[1], h[2]);
  for (int k = 0; k < 8; ++k) printf(" %u", h[3 + k]);
  printf(")\n");
  hipLaunchKernelGGL(k_long, dim3(4), dim3(64), 0, 0, d, long_steps, 997);
  CK(hipDeviceSynchronize());
  CK(hipMemcpy(h, d, 64, hipMemcpyDeviceToHost));
  printf("long chains (%d additions x 256 lanes, cancel every 997): coordinates above 2m + 2^130: %u, mismatches %u, cancellations %u\n",
         long_steps, h[12], h[13], h[14]);
  if (h[14] == 0) { printf("no cancellation was exercised\n"); return 1; }
  return (h[0] | h[1] | h[2] | h[12] | h[13] | h[15]) ? 1 : 0;
}
