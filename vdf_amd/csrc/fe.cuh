// Pasta field arithmetic for gfx950: 8 x 32-bit limbs, Montgomery form (R = 2^256).
//
// Replaces (on the device) the field layer the reference reaches through
// pasta_curves 0.4.0 (Cargo.toml:17): Fp/Fq `mul`, `square`, `add`, `sub`
// (used at src/minroot.rs:74, :221, :331-333, :339-342 and by every third-party
// call on the prove_step path, src/nova/proof.rs:342-349).  In-memory layout is
// the same 4 x u64 little-endian Montgomery limbs pasta_curves' `repr-c` uses, so
// buffers cross the C ABI unchanged.
//
// The work-horse is v_mad_u64_u32 (32x32+64 -> 64).  Both moduli are 2^254 + c,
// c < 2^126, == 1 (mod 2^32):  -m^-1 mod 2^32 = 0xFFFFFFFF so the Montgomery
// quotient digit is just the negated low limb, modulus limb 0 is 1 (no multiply),
// limbs 4..6 are zero and limb 7 is 2^30 (a shift): 3 real multiplies per
// reduction round instead of 8.
//
// The same code compiles for the host (the C++ Nova layer uses it for its O(1)
// scalar work); there is no separate CPU implementation in the product.
#pragma once
#include <stdint.h>
#include "pasta_constants.h"

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define VDF_HD __host__ __device__ __forceinline__
#else
#define VDF_HD inline
#endif

namespace vdf {

template <class P>
struct Fe {
  uint32_t v[8];
};

using Fp = Fe<FpParams>;
using Fq = Fe<FqParams>;

template <class P> VDF_HD Fe<P> fe_zero() {
  Fe<P> r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r.v[i] = 0;
  return r;
}
template <class P> VDF_HD Fe<P> fe_one() {
  Fe<P> r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r.v[i] = P::ONE[i];
  return r;
}
template <class P> VDF_HD bool fe_is_zero(const Fe<P>& a) {
  uint32_t o = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) o |= a.v[i];
  return o == 0;
}
template <class P> VDF_HD bool fe_eq(const Fe<P>& a, const Fe<P>& b) {
  uint32_t o = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) o |= a.v[i] ^ b.v[i];
  return o == 0;
}

// r = a - m if a >= m (a < 2m assumed).
#if defined(__HIP_DEVICE_COMPILE__)
// 8 subtract-with-borrow + 8 selects; modulus limbs 0, 4..7 are inline constants (1, 0, 0, 0, 2^30).
template <class P> __device__ __forceinline__ void fe_cond_sub(uint32_t t[8]) {
  uint32_t d0, d1, d2, d3, d4, d5, d6, d7;
  asm("v_subrev_co_u32_e32 %0, vcc, 1, %8\n\t"
      "v_subbrev_co_u32_e32 %1, vcc, %16, %9, vcc\n\t"
      "v_subbrev_co_u32_e32 %2, vcc, %17, %10, vcc\n\t"
      "v_subbrev_co_u32_e32 %3, vcc, %18, %11, vcc\n\t"
      "v_subbrev_co_u32_e32 %4, vcc, 0, %12, vcc\n\t"
      "v_subbrev_co_u32_e32 %5, vcc, 0, %13, vcc\n\t"
      "v_subbrev_co_u32_e32 %6, vcc, 0, %14, vcc\n\t"
      "v_subbrev_co_u32_e32 %7, vcc, 2.0, %15, vcc\n\t"
      "v_cndmask_b32_e32 %8, %0, %8, vcc\n\t"
      "v_cndmask_b32_e32 %9, %1, %9, vcc\n\t"
      "v_cndmask_b32_e32 %10, %2, %10, vcc\n\t"
      "v_cndmask_b32_e32 %11, %3, %11, vcc\n\t"
      "v_cndmask_b32_e32 %12, %4, %12, vcc\n\t"
      "v_cndmask_b32_e32 %13, %5, %13, vcc\n\t"
      "v_cndmask_b32_e32 %14, %6, %14, vcc\n\t"
      "v_cndmask_b32_e32 %15, %7, %15, vcc"
      : "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3), "=&v"(d4), "=&v"(d5), "=&v"(d6), "=&v"(d7),
        "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]), "+v"(t[4]), "+v"(t[5]), "+v"(t[6]), "+v"(t[7])
      : "v"(P::MOD[1]), "v"(P::MOD[2]), "v"(P::MOD[3])     // VGPRs: VCC carry-in already uses the constant bus
      : "vcc");
}
static_assert(FpParams::MOD[0] == 1 && FpParams::MOD[4] == 0 && FpParams::MOD[5] == 0 && FpParams::MOD[6] == 0 &&
              FpParams::MOD[7] == 0x40000000u, "fe_cond_sub hard-codes the sparse limbs");
static_assert(FqParams::MOD[0] == 1 && FqParams::MOD[4] == 0 && FqParams::MOD[5] == 0 && FqParams::MOD[6] == 0 &&
              FqParams::MOD[7] == 0x40000000u, "fe_cond_sub hard-codes the sparse limbs");

template <class P> __device__ __forceinline__ Fe<P> fe_add(const Fe<P>& a, const Fe<P>& b) {
  Fe<P> r = a;
  asm("v_add_co_u32_e32 %0, vcc, %0, %8\n\t"
      "v_addc_co_u32_e32 %1, vcc, %1, %9, vcc\n\t"
      "v_addc_co_u32_e32 %2, vcc, %2, %10, vcc\n\t"
      "v_addc_co_u32_e32 %3, vcc, %3, %11, vcc\n\t"
      "v_addc_co_u32_e32 %4, vcc, %4, %12, vcc\n\t"
      "v_addc_co_u32_e32 %5, vcc, %5, %13, vcc\n\t"
      "v_addc_co_u32_e32 %6, vcc, %6, %14, vcc\n\t"
      "v_addc_co_u32_e32 %7, vcc, %7, %15, vcc"
      : "+v"(r.v[0]), "+v"(r.v[1]), "+v"(r.v[2]), "+v"(r.v[3]), "+v"(r.v[4]), "+v"(r.v[5]), "+v"(r.v[6]), "+v"(r.v[7])
      : "v"(b.v[0]), "v"(b.v[1]), "v"(b.v[2]), "v"(b.v[3]), "v"(b.v[4]), "v"(b.v[5]), "v"(b.v[6]), "v"(b.v[7])
      : "vcc");
  fe_cond_sub<P>(r.v);        // a, b < m < 2^255: no carry out of limb 7
  return r;
}

// a - b, plus m when the subtraction borrows (mask from the final borrow, then a second carry chain)
template <class P> __device__ __forceinline__ Fe<P> fe_sub(const Fe<P>& a, const Fe<P>& b) {
  Fe<P> r = a;
  uint32_t mask;
  asm("v_sub_co_u32_e32 %0, vcc, %0, %9\n\t"
      "v_subb_co_u32_e32 %1, vcc, %1, %10, vcc\n\t"
      "v_subb_co_u32_e32 %2, vcc, %2, %11, vcc\n\t"
      "v_subb_co_u32_e32 %3, vcc, %3, %12, vcc\n\t"
      "v_subb_co_u32_e32 %4, vcc, %4, %13, vcc\n\t"
      "v_subb_co_u32_e32 %5, vcc, %5, %14, vcc\n\t"
      "v_subb_co_u32_e32 %6, vcc, %6, %15, vcc\n\t"
      "v_subb_co_u32_e32 %7, vcc, %7, %16, vcc\n\t"
      "v_cndmask_b32_e64 %8, 0, -1, vcc"
      : "+v"(r.v[0]), "+v"(r.v[1]), "+v"(r.v[2]), "+v"(r.v[3]), "+v"(r.v[4]), "+v"(r.v[5]), "+v"(r.v[6]), "+v"(r.v[7]),
        "=&v"(mask)
      : "v"(b.v[0]), "v"(b.v[1]), "v"(b.v[2]), "v"(b.v[3]), "v"(b.v[4]), "v"(b.v[5]), "v"(b.v[6]), "v"(b.v[7])
      : "vcc");
  const uint32_t m0 = mask & 1u, m1 = mask & P::MOD[1], m2 = mask & P::MOD[2], m3 = mask & P::MOD[3],
                 m7 = mask & 0x40000000u;
  asm("v_add_co_u32_e32 %0, vcc, %0, %8\n\t"
      "v_addc_co_u32_e32 %1, vcc, %1, %9, vcc\n\t"
      "v_addc_co_u32_e32 %2, vcc, %2, %10, vcc\n\t"
      "v_addc_co_u32_e32 %3, vcc, %3, %11, vcc\n\t"
      "v_addc_co_u32_e32 %4, vcc, 0, %4, vcc\n\t"
      "v_addc_co_u32_e32 %5, vcc, 0, %5, vcc\n\t"
      "v_addc_co_u32_e32 %6, vcc, 0, %6, vcc\n\t"
      "v_addc_co_u32_e32 %7, vcc, %7, %12, vcc"
      : "+v"(r.v[0]), "+v"(r.v[1]), "+v"(r.v[2]), "+v"(r.v[3]), "+v"(r.v[4]), "+v"(r.v[5]), "+v"(r.v[6]), "+v"(r.v[7])
      : "v"(m0), "v"(m1), "v"(m2), "v"(m3), "v"(m7)
      : "vcc");
  return r;
}
#else
template <class P> VDF_HD void fe_cond_sub(uint32_t t[8]) {
  uint32_t d[8];
  uint32_t borrow = 0;
  for (int i = 0; i < 8; ++i) {
    uint64_t s = (uint64_t)t[i] - P::MOD[i] - borrow;
    d[i] = (uint32_t)s;
    borrow = (uint32_t)(s >> 63);
  }
  for (int i = 0; i < 8; ++i) t[i] = borrow ? t[i] : d[i];
}

template <class P> VDF_HD Fe<P> fe_add(const Fe<P>& a, const Fe<P>& b) {
  Fe<P> r;
  uint32_t c = 0;
  for (int i = 0; i < 8; ++i) {
    uint64_t s = (uint64_t)a.v[i] + b.v[i] + c;
    r.v[i] = (uint32_t)s;
    c = (uint32_t)(s >> 32);
  }
  fe_cond_sub<P>(r.v);        // a, b < m < 2^255 so no carry out of limb 7
  return r;
}

template <class P> VDF_HD Fe<P> fe_sub(const Fe<P>& a, const Fe<P>& b) {
  Fe<P> r;
  uint32_t borrow = 0;
  for (int i = 0; i < 8; ++i) {
    uint64_t s = (uint64_t)a.v[i] - b.v[i] - borrow;
    r.v[i] = (uint32_t)s;
    borrow = (uint32_t)(s >> 63);
  }
  uint32_t c = 0;
  for (int i = 0; i < 8; ++i) {
    uint64_t s = (uint64_t)r.v[i] + (borrow ? P::MOD[i] : 0u) + c;
    r.v[i] = (uint32_t)s;
    c = (uint32_t)(s >> 32);
  }
  return r;
}
#endif

template <class P> VDF_HD Fe<P> fe_neg(const Fe<P>& a) {
  return fe_is_zero(a) ? a : fe_sub(fe_zero<P>(), a);
}

template <class P> VDF_HD Fe<P> fe_dbl(const Fe<P>& a) { return fe_add(a, a); }

// One Montgomery reduction round on t[0..8]: t = (t + q*m) / 2^32 with q = -t[0].
template <class P> VDF_HD void mont_round(uint32_t t[9]) {
  const uint32_t q = 0u - t[0];
  uint64_t s;
  uint32_t c = (t[0] != 0u);                       // t0 + q*1 = 0 or 2^32
  s = (uint64_t)q * P::MOD[1] + t[1] + c; t[0] = (uint32_t)s; c = (uint32_t)(s >> 32);
  s = (uint64_t)q * P::MOD[2] + t[2] + c; t[1] = (uint32_t)s; c = (uint32_t)(s >> 32);
  s = (uint64_t)q * P::MOD[3] + t[3] + c; t[2] = (uint32_t)s; c = (uint32_t)(s >> 32);
  s = (uint64_t)t[4] + c;                 t[3] = (uint32_t)s; c = (uint32_t)(s >> 32);
  s = (uint64_t)t[5] + c;                 t[4] = (uint32_t)s; c = (uint32_t)(s >> 32);
  s = (uint64_t)t[6] + c;                 t[5] = (uint32_t)s; c = (uint32_t)(s >> 32);
  // q * 2^30 split over limbs 7 and 8
  s = (uint64_t)t[7] + (uint32_t)(q << 30) + c; t[6] = (uint32_t)s; c = (uint32_t)(s >> 32);
  s = (uint64_t)t[8] + (q >> 2) + c;      t[7] = (uint32_t)s; t[8] = (uint32_t)(s >> 32);
}

// Montgomery product a*b/R mod m, inputs and output in [0, m): portable form (host, and the
// reference the device form is tested against).
template <class P> VDF_HD Fe<P> fe_mul_generic(const Fe<P>& a, const Fe<P>& b) {
  uint32_t t[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) t[i] = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    uint32_t c = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      uint64_t s = (uint64_t)a.v[j] * b.v[i] + t[j] + c;
      t[j] = (uint32_t)s;
      c = (uint32_t)(s >> 32);
    }
    t[8] += c;                                    // t < 2m*2^32-ish: no overflow of limb 8
    mont_round<P>(t);
  }
  fe_cond_sub<P>(t);
  Fe<P> r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r.v[i] = t[i];
  return r;
}

#if defined(__HIP_DEVICE_COMPILE__)
// ---- gfx950 form: product scanning with a 96-bit column accumulator ---------------------------
// acc = (lo, mid) in an aligned VGPR pair + hi.  One product costs v_mad_u64_u32 (64-bit
// accumulate, carry to VCC) + v_addc_co_u32 (carry into hi): 2 instructions, no v_mov glue.
// hipcc's own lowering of the portable form is 563 instructions for 88 multiplies; this is ~260.
__device__ __forceinline__ void madc(uint64_t& acc, uint32_t& hi, uint32_t a, uint32_t b) {
  asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32_e32 %1, vcc, 0, %1, vcc"
      : "+v"(acc), "+v"(hi) : "v"(a), "v"(b) : "vcc");
}
// same with a scalar (SGPR / inline constant) multiplier: modulus limbs are wave-uniform constants
__device__ __forceinline__ void madc_s(uint64_t& acc, uint32_t& hi, uint32_t a, uint32_t k) {
  asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32_e32 %1, vcc, 0, %1, vcc"
      : "+v"(acc), "+v"(hi) : "v"(a), "s"(k) : "vcc");
}
// Montgomery step of a low column: adding q*m0 (m0 = 1, q = -lo) makes the low word 0 and carries
// (lo != 0); this also performs the one-limb shift: (lo, mid, hi) -> (mid + c, hi + c', 0).
__device__ __forceinline__ void col_shift_q(uint64_t& acc, uint32_t& hi) {
  uint32_t lo = (uint32_t)acc, mid = (uint32_t)(acc >> 32), nlo, nmid;
  asm("v_cmp_ne_u32_e32 vcc, 0, %2\n\tv_addc_co_u32_e32 %0, vcc, 0, %3, vcc\n\tv_addc_co_u32_e32 %1, vcc, 0, %4, vcc"
      : "=&v"(nlo), "=&v"(nmid) : "v"(lo), "v"(mid), "v"(hi) : "vcc");
  acc = ((uint64_t)nmid << 32) | nlo;
}
__device__ __forceinline__ void col_shift(uint64_t& acc, uint32_t& hi) {
  acc = (acc >> 32) | ((uint64_t)hi << 32);
  hi = 0;
}

// One column of the product scan as a single asm statement (fewer asm boundaries = fewer of the
// s_nop hazard pads hipcc puts between consecutive asm blocks).  NP products a[i]*b[K-i] and the
// reduction terms q[K-1]*m1, q[K-2]*m2, q[K-3]*m3, q[K-7]*m7 are accumulated into (acc, hi).
#define VDF_MADC "v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32_e32 %1, vcc, 0, %1, vcc"
__device__ __forceinline__ void madc2(uint64_t& acc, uint32_t& hi, uint32_t a0, uint32_t b0, uint32_t a1, uint32_t b1) {
  asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32_e32 %1, vcc, 0, %1, vcc\n\t"
      "v_mad_u64_u32 %0, vcc, %4, %5, %0\n\tv_addc_co_u32_e32 %1, vcc, 0, %1, vcc"
      : "+v"(acc), "+v"(hi) : "v"(a0), "v"(b0), "v"(a1), "v"(b1) : "vcc");
}
__device__ __forceinline__ void madc4(uint64_t& acc, uint32_t& hi, uint32_t a0, uint32_t b0, uint32_t a1, uint32_t b1,
                                      uint32_t a2, uint32_t b2, uint32_t a3, uint32_t b3) {
  asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32_e32 %1, vcc, 0, %1, vcc\n\t"
      "v_mad_u64_u32 %0, vcc, %4, %5, %0\n\tv_addc_co_u32_e32 %1, vcc, 0, %1, vcc\n\t"
      "v_mad_u64_u32 %0, vcc, %6, %7, %0\n\tv_addc_co_u32_e32 %1, vcc, 0, %1, vcc\n\t"
      "v_mad_u64_u32 %0, vcc, %8, %9, %0\n\tv_addc_co_u32_e32 %1, vcc, 0, %1, vcc"
      : "+v"(acc), "+v"(hi) : "v"(a0), "v"(b0), "v"(a1), "v"(b1), "v"(a2), "v"(b2), "v"(a3), "v"(b3) : "vcc");
}
// reduction terms of one column: up to four q*m products with scalar (wave-uniform) modulus limbs
template <int N>
__device__ __forceinline__ void madc_red(uint64_t& acc, uint32_t& hi, uint32_t q1, uint32_t q2, uint32_t q3, uint32_t q7,
                                         uint32_t m1, uint32_t m2, uint32_t m3, uint32_t m7) {
  if constexpr (N == 4)
    asm("v_mad_u64_u32 %0, vcc, %2, %6, %0\n\tv_addc_co_u32_e32 %1, vcc, 0, %1, vcc\n\t"
        "v_mad_u64_u32 %0, vcc, %3, %7, %0\n\tv_addc_co_u32_e32 %1, vcc, 0, %1, vcc\n\t"
        "v_mad_u64_u32 %0, vcc, %4, %8, %0\n\tv_addc_co_u32_e32 %1, vcc, 0, %1, vcc\n\t"
        "v_mad_u64_u32 %0, vcc, %5, %9, %0\n\tv_addc_co_u32_e32 %1, vcc, 0, %1, vcc"
        : "+v"(acc), "+v"(hi) : "v"(q1), "v"(q2), "v"(q3), "v"(q7), "s"(m1), "s"(m2), "s"(m3), "s"(m7) : "vcc");
  else if constexpr (N == 3)
    asm("v_mad_u64_u32 %0, vcc, %2, %5, %0\n\tv_addc_co_u32_e32 %1, vcc, 0, %1, vcc\n\t"
        "v_mad_u64_u32 %0, vcc, %3, %6, %0\n\tv_addc_co_u32_e32 %1, vcc, 0, %1, vcc\n\t"
        "v_mad_u64_u32 %0, vcc, %4, %7, %0\n\tv_addc_co_u32_e32 %1, vcc, 0, %1, vcc"
        : "+v"(acc), "+v"(hi) : "v"(q1), "v"(q2), "v"(q3), "s"(m1), "s"(m2), "s"(m3) : "vcc");
}

template <class P> __device__ __forceinline__ Fe<P> fe_mul_inl(const Fe<P>& a, const Fe<P>& b) {
  constexpr uint32_t M1 = P::MOD[1], M2 = P::MOD[2], M3 = P::MOD[3], M7 = P::MOD[7];
  const uint32_t* A = a.v;
  const uint32_t* B = b.v;
  uint32_t r[8];
#include "fe_mul_gfx950.inc"
  fe_cond_sub<P>(r);                      // result < 2m < 2^256: the carry word is zero
  Fe<P> o;
#pragma unroll
  for (int i = 0; i < 8; ++i) o.v[i] = r[i];
  return o;
}

// ---- lazy domain [0, 2m + eps) for the MSM bucket loop ------------------------------------------------
// Montgomery closure: with a, b < 2m + k*eps the unreduced product (a*b + q*m)/2^256 is below
// 2m + (k+1)*eps, eps = m*(m/2^254 - 1) ~ 2^126, so the final conditional subtraction can be dropped as long
// as every consumer accepts [0, 2m + small): a chain of thousands of multiplications stays below
// 2m + 2^150 < 2^256.  Subtraction corrects a borrow with 2m instead of m (same cost), which keeps the
// range.  Values are made canonical again (two conditional subtractions) only when a bucket is flushed.
template <class P> __device__ __forceinline__ Fe<P> fe_mul_lazy(const Fe<P>& a, const Fe<P>& b) {
  constexpr uint32_t M1 = P::MOD[1], M2 = P::MOD[2], M3 = P::MOD[3], M7 = P::MOD[7];
  const uint32_t* A = a.v;
  const uint32_t* B = b.v;
  uint32_t r[8];
#include "fe_mul_gfx950.inc"
  Fe<P> o;
#pragma unroll
  for (int i = 0; i < 8; ++i) o.v[i] = r[i];
  return o;
}

// a*a in the lazy domain: the integer (a^2 + q m) / 2^256 of fe_mul_lazy(a, a), bit for bit, from 43 limb products instead of 64.
// a^2 = sum_i a_i E_i B^(2i) with E_i = a_i + 2 (a div B^(i+1)) B: its limbs are a_i, then (a_(i+1) << 1) mod 2^32, then limbs
// i+2 .. 8 of the doubled number 2a (whose limb i+2 carries a_(i+1)'s top bit in) -- eight diagonal products, seven with a
// shifted neighbour, 28 with limbs of 2a; 15 shifts build those.  Needs a < 2^256 (every lazy value is).
template <class P> __device__ __forceinline__ Fe<P> fe_sqr_lazy(const Fe<P>& a) {
  constexpr uint32_t M1 = P::MOD[1], M2 = P::MOD[2], M3 = P::MOD[3], M7 = P::MOD[7];
  const uint32_t* A = a.v;
  uint32_t S[8], D[9];
#pragma unroll
  for (int j = 1; j < 8; ++j) S[j] = a.v[j] << 1;
#pragma unroll
  for (int j = 2; j < 8; ++j) D[j] = __builtin_amdgcn_alignbit(a.v[j], a.v[j - 1], 31);
  D[8] = a.v[7] >> 31;
  uint32_t r[8];
#include "fe_sqr_gfx950.inc"
  Fe<P> o;
#pragma unroll
  for (int i = 0; i < 8; ++i) o.v[i] = r[i];
  return o;
}

// a - b (+ 2m on borrow) for a, b in [0, 2m + eps).  Three-operand form: the difference goes to registers of its own
// (early-clobber outputs), so an operand that stays live afterwards -- X1, Y1, Q in the bucket addition -- is not copied first
// (the in-place form cost eight v_mov per live operand: 42 moves per addition in round 4's first version of the loop).
template <class P> __device__ __forceinline__ Fe<P> fe_sub_lazy(const Fe<P>& a, const Fe<P>& b) {
  Fe<P> r;
  uint32_t mask;
  asm("v_sub_co_u32_e32 %0, vcc, %9, %17\n\t"
      "v_subb_co_u32_e32 %1, vcc, %10, %18, vcc\n\t"
      "v_subb_co_u32_e32 %2, vcc, %11, %19, vcc\n\t"
      "v_subb_co_u32_e32 %3, vcc, %12, %20, vcc\n\t"
      "v_subb_co_u32_e32 %4, vcc, %13, %21, vcc\n\t"
      "v_subb_co_u32_e32 %5, vcc, %14, %22, vcc\n\t"
      "v_subb_co_u32_e32 %6, vcc, %15, %23, vcc\n\t"
      "v_subb_co_u32_e32 %7, vcc, %16, %24, vcc\n\t"
      "v_cndmask_b32_e64 %8, 0, -1, vcc"
      : "=&v"(r.v[0]), "=&v"(r.v[1]), "=&v"(r.v[2]), "=&v"(r.v[3]), "=&v"(r.v[4]), "=&v"(r.v[5]), "=&v"(r.v[6]), "=&v"(r.v[7]),
        "=&v"(mask)
      : "v"(a.v[0]), "v"(a.v[1]), "v"(a.v[2]), "v"(a.v[3]), "v"(a.v[4]), "v"(a.v[5]), "v"(a.v[6]), "v"(a.v[7]),
        "v"(b.v[0]), "v"(b.v[1]), "v"(b.v[2]), "v"(b.v[3]), "v"(b.v[4]), "v"(b.v[5]), "v"(b.v[6]), "v"(b.v[7])
      : "vcc");
  // 2m = {2, 2*m1 (33 bits: low word + carry into limb 2), ...}: computed limb-wise at compile time
  constexpr uint64_t D1 = 2ull * P::MOD[1], D2 = 2ull * P::MOD[2] + (D1 >> 32), D3 = 2ull * P::MOD[3] + (D2 >> 32);
  constexpr uint32_t T0 = 2u, T1 = (uint32_t)D1, T2 = (uint32_t)D2, T3 = (uint32_t)D3, T4 = (uint32_t)(D3 >> 32), T7 = 0x80000000u;
  const uint32_t m0 = mask & T0, m1 = mask & T1, m2 = mask & T2, m3 = mask & T3, m4 = mask & T4, m7 = mask & T7;
  asm("v_add_co_u32_e32 %0, vcc, %0, %8\n\t"
      "v_addc_co_u32_e32 %1, vcc, %1, %9, vcc\n\t"
      "v_addc_co_u32_e32 %2, vcc, %2, %10, vcc\n\t"
      "v_addc_co_u32_e32 %3, vcc, %3, %11, vcc\n\t"
      "v_addc_co_u32_e32 %4, vcc, %4, %12, vcc\n\t"
      "v_addc_co_u32_e32 %5, vcc, 0, %5, vcc\n\t"
      "v_addc_co_u32_e32 %6, vcc, 0, %6, vcc\n\t"
      "v_addc_co_u32_e32 %7, vcc, %7, %13, vcc"
      : "+v"(r.v[0]), "+v"(r.v[1]), "+v"(r.v[2]), "+v"(r.v[3]), "+v"(r.v[4]), "+v"(r.v[5]), "+v"(r.v[6]), "+v"(r.v[7])
      : "v"(m0), "v"(m1), "v"(m2), "v"(m3), "v"(m4), "v"(m7)
      : "vcc");
  return r;
}

// a*b + c*d with ONE shared Montgomery reduction, lazy domain in and out: the two products' partial products go into
// the same column accumulators (128 multiply-adds) and are reduced once (32 more), against 2 x (64 + 32) for two
// products and the 23-instruction addition between them -- the pair costs ~395 instructions instead of ~515.
// Bound: inputs below 2m + d give (ab + cd)/R < 2(2m + d)^2 / 2^256 = 2m + 2eps + 2d(1 + ...) and q*m/R < m, so the
// column scan ends below 3m + 2eps + 2d < 2^256 (the ninth word is zero).  Back into the lazy domain with the top bit:
// bit 255 clear -> below 2^255 = 2m - 2c, nothing to do; set -> at least 2^255 > m, subtract m: the result is at least
// 2^255 - m = m - 2c > 0 and below 2m + 2eps + 2d.  (14 instructions; a compare-and-subtract of 2m would underflow for
// values in [2^255, 2m).)  Exactly: out < 2m + 2 eps + (d_a + d_b + d_c + d_d) / 2 -- each factor's slack is halved, because the
// OTHER factor is below 2m + d and 2m / 2^256 = 1/2 (+ 2^-130).  The pair alone does not contract (four halves); the mixed
// addition feeds it one slack-free factor and two bounded ones, which does (ec.cuh xyzz_madd_lazy: d_y' <= 4.5 eps + d_y / 2).
template <class P> __device__ __forceinline__ Fe<P> fe_mul2_lazy(const Fe<P>& a, const Fe<P>& b, const Fe<P>& c, const Fe<P>& d) {
  constexpr uint32_t M1 = P::MOD[1], M2 = P::MOD[2], M3 = P::MOD[3], M7 = P::MOD[7];
  const uint32_t* A = a.v;
  const uint32_t* B = b.v;
  const uint32_t* Cc = c.v;
  const uint32_t* Dd = d.v;
  uint32_t r[8];
#include "fe_mul2_gfx950.inc"
  const uint32_t mask = (uint32_t)((int32_t)r[7] >> 31);
  const uint32_t m0 = mask & 1u, m1 = mask & P::MOD[1], m2 = mask & P::MOD[2], m3 = mask & P::MOD[3], m7 = mask & 0x40000000u;
  asm("v_sub_co_u32_e32 %0, vcc, %0, %8\n\t"
      "v_subb_co_u32_e32 %1, vcc, %1, %9, vcc\n\t"
      "v_subb_co_u32_e32 %2, vcc, %2, %10, vcc\n\t"
      "v_subb_co_u32_e32 %3, vcc, %3, %11, vcc\n\t"
      "v_subbrev_co_u32_e32 %4, vcc, 0, %4, vcc\n\t"
      "v_subbrev_co_u32_e32 %5, vcc, 0, %5, vcc\n\t"
      "v_subbrev_co_u32_e32 %6, vcc, 0, %6, vcc\n\t"
      "v_subb_co_u32_e32 %7, vcc, %7, %12, vcc"
      : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7])
      : "v"(m0), "v"(m1), "v"(m2), "v"(m3), "v"(m7)
      : "vcc");
  Fe<P> o;
#pragma unroll
  for (int i = 0; i < 8; ++i) o.v[i] = r[i];
  return o;
}

// -a in the lazy domain: 3m - a, exact for every a below 3m (fe_sub_lazy(0, a) would wrap for a in (2m, 2m + eps));
// the result lies in (m - eps, 3m], so it goes ONLY where the consumer canonicalises: a bucket accumulator flushed with
// its sign pending (ec.cuh xyzz_lazy_resolve; the tail kernels load through fe_canon, two conditional subtractions).
// Precondition a != 0 (mod m) -- a = 0 would give 3m, which fe_canon's two subtractions leave at m, not 0 -- holds for the
// one caller: the y of a point of a curve of odd prime order is never 0 (y = 0 is a point of order 2).
template <class P> __device__ __forceinline__ Fe<P> fe_neg_lazy(const Fe<P>& a) {
  constexpr uint64_t D1 = 3ull * P::MOD[1], D2 = 3ull * P::MOD[2] + (D1 >> 32), D3 = 3ull * P::MOD[3] + (D2 >> 32);
  constexpr uint32_t T1 = (uint32_t)D1, T2 = (uint32_t)D2, T3 = (uint32_t)D3, T4 = (uint32_t)(D3 >> 32), T7 = 0xC0000000u;   // limb 0 = 3, limbs 5, 6 = 0
  Fe<P> r = a;
  asm("v_sub_co_u32_e32 %0, vcc, 3, %0\n\t"
      "v_subb_co_u32_e32 %1, vcc, %8, %1, vcc\n\t"
      "v_subb_co_u32_e32 %2, vcc, %9, %2, vcc\n\t"
      "v_subb_co_u32_e32 %3, vcc, %10, %3, vcc\n\t"
      "v_subb_co_u32_e32 %4, vcc, %11, %4, vcc\n\t"
      "v_subb_co_u32_e32 %5, vcc, 0, %5, vcc\n\t"
      "v_subb_co_u32_e32 %6, vcc, 0, %6, vcc\n\t"
      "v_subb_co_u32_e32 %7, vcc, %12, %7, vcc"
      : "+v"(r.v[0]), "+v"(r.v[1]), "+v"(r.v[2]), "+v"(r.v[3]), "+v"(r.v[4]), "+v"(r.v[5]), "+v"(r.v[6]), "+v"(r.v[7])
      : "v"(T1), "v"(T2), "v"(T3), "v"(T4), "v"(T7)
      : "vcc");
  return r;
}

// m - a for a canonical, NON-ZERO a (the y of a curve point: neither curve has a point with y = 0): eight instructions, where
// fe_neg pays a zero test and a masked correction (~45) -- the bucket loops negate a point per addition
template <class P> __device__ __forceinline__ Fe<P> fe_neg_nz(const Fe<P>& a) {
  Fe<P> r;
  asm("v_sub_co_u32_e32 %0, vcc, 1, %8\n\t"
      "v_subb_co_u32_e32 %1, vcc, %16, %9, vcc\n\t"
      "v_subb_co_u32_e32 %2, vcc, %17, %10, vcc\n\t"
      "v_subb_co_u32_e32 %3, vcc, %18, %11, vcc\n\t"
      "v_subb_co_u32_e32 %4, vcc, 0, %12, vcc\n\t"
      "v_subb_co_u32_e32 %5, vcc, 0, %13, vcc\n\t"
      "v_subb_co_u32_e32 %6, vcc, 0, %14, vcc\n\t"
      "v_subb_co_u32_e32 %7, vcc, 2.0, %15, vcc"
      : "=&v"(r.v[0]), "=&v"(r.v[1]), "=&v"(r.v[2]), "=&v"(r.v[3]), "=&v"(r.v[4]), "=&v"(r.v[5]), "=&v"(r.v[6]), "=&v"(r.v[7])
      : "v"(a.v[0]), "v"(a.v[1]), "v"(a.v[2]), "v"(a.v[3]), "v"(a.v[4]), "v"(a.v[5]), "v"(a.v[6]), "v"(a.v[7]),
        "v"(P::MOD[1]), "v"(P::MOD[2]), "v"(P::MOD[3])
      : "vcc");
  return r;
}

// [0, 2m + eps) -> [0, m)
template <class P> __device__ __forceinline__ Fe<P> fe_canon(Fe<P> a) {
  fe_cond_sub<P>(a.v);
  fe_cond_sub<P>(a.v);
  return a;
}
// canonical a^2 for canonical a: the lazy squaring's result is below 2m (as fe_mul_inl's before its subtraction)
template <class P> __device__ __forceinline__ Fe<P> fe_sqr_inl(const Fe<P>& a) {
  Fe<P> r = fe_sqr_lazy(a);
  fe_cond_sub<P>(r.v);
  return r;
}
#else
template <class P> VDF_HD Fe<P> fe_sqr_inl(const Fe<P>& a) { return fe_mul_generic(a, a); }
template <class P> VDF_HD Fe<P> fe_mul_inl(const Fe<P>& a, const Fe<P>& b) { return fe_mul_generic(a, b); }
// host pass: canonical arithmetic is a valid instance of the lazy interface
template <class P> VDF_HD Fe<P> fe_mul_lazy(const Fe<P>& a, const Fe<P>& b) { return fe_mul_generic(a, b); }
template <class P> VDF_HD Fe<P> fe_sqr_lazy(const Fe<P>& a) { return fe_mul_generic(a, a); }
template <class P> VDF_HD Fe<P> fe_canon(Fe<P> a) { return a; }
template <class P> VDF_HD Fe<P> fe_sub_lazy(const Fe<P>& a, const Fe<P>& b) { return fe_sub(a, b); }
template <class P> VDF_HD Fe<P> fe_mul2_lazy(const Fe<P>& a, const Fe<P>& b, const Fe<P>& c, const Fe<P>& d) {
  return fe_add(fe_mul_generic(a, b), fe_mul_generic(c, d));
}
template <class P> VDF_HD Fe<P> fe_neg_lazy(const Fe<P>& a) { return fe_neg(a); }
template <class P> VDF_HD Fe<P> fe_neg_nz(const Fe<P>& a) { return fe_neg(a); }
#endif

// Out-of-line multiply (by-value arguments travel in VGPRs): one copy per field per TU.
#if defined(__HIP_DEVICE_COMPILE__)
template <class P> __device__ __attribute__((noinline)) Fe<P> fe_mul_call(Fe<P> a, Fe<P> b) { return fe_mul_inl(a, b); }
template <class P> VDF_HD Fe<P> fe_mul(const Fe<P>& a, const Fe<P>& b) { return fe_mul_call<P>(a, b); }
#else
template <class P> VDF_HD Fe<P> fe_mul(const Fe<P>& a, const Fe<P>& b) { return fe_mul_inl(a, b); }
#endif
template <class P> VDF_HD Fe<P> fe_sqr(const Fe<P>& a) { return fe_mul(a, a); }
// compile-time choice between the two
template <bool INL, class P> VDF_HD Fe<P> fe_mul_sel(const Fe<P>& a, const Fe<P>& b) {
  if constexpr (INL) return fe_mul_inl(a, b); else return fe_mul(a, b);
}

// Out of Montgomery form: a/R mod m (eight reduction rounds of a alone).
template <class P> VDF_HD Fe<P> fe_from_mont(const Fe<P>& a) {
  uint32_t t[9];
#pragma unroll
  for (int i = 0; i < 8; ++i) t[i] = a.v[i];
  t[8] = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) mont_round<P>(t);
  fe_cond_sub<P>(t);
  Fe<P> r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r.v[i] = t[i];
  return r;
}

template <class P> VDF_HD Fe<P> fe_to_mont(const Fe<P>& a) {
  Fe<P> r2;
#pragma unroll
  for (int i = 0; i < 8; ++i) r2.v[i] = P::R2[i];
  return fe_mul(a, r2);
}

// true iff the 256-bit value is a canonical residue (< m).
template <class P> VDF_HD bool fe_is_canonical(const Fe<P>& a) {
  uint32_t borrow = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    uint64_t s = (uint64_t)a.v[i] - P::MOD[i] - borrow;
    borrow = (uint32_t)(s >> 63);
  }
  return borrow != 0;
}

// a^e for a 256-bit exponent given as 8 x u32 LE limbs, left-to-right square and
// multiply (same shape as ff::Field::pow_vartime, src/minroot.rs:312-314).
template <class P> VDF_HD Fe<P> fe_pow(const Fe<P>& a, const uint32_t e[8]) {
  Fe<P> r = fe_one<P>();
  bool started = false;
  for (int i = 7; i >= 0; --i) {
    for (int b = 31; b >= 0; --b) {
      if (started) r = fe_sqr(r);
      if ((e[i] >> b) & 1u) {
        r = started ? fe_mul(r, a) : a;
        started = true;
      }
    }
  }
  return r;
}

template <class P> VDF_HD Fe<P> fe_inv(const Fe<P>& a) {   // a^(m-2); inv(0) = 0
  uint32_t e[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) e[i] = P::MOD_MINUS_2[i];
  return fe_pow(a, e);
}

template <class P> VDF_HD Fe<P> fe_from_u64(uint64_t x) {     // Montgomery form of a small integer
  Fe<P> r = fe_zero<P>();
  r.v[0] = (uint32_t)x;
  r.v[1] = (uint32_t)(x >> 32);
  return fe_to_mont(r);
}

// Montgomery form of a SMALL integer without a Montgomery product: k * (2^256 mod m) is below 2^288, and with
// m = 2^254 + c (c < 2^126: limbs 0..3 of the modulus) the quotient is its bits from 254 up (below 2^34... 2^32 for k < 2^30):
// r = (x mod 2^254) - q c, plus m if that went negative -- about 50 instructions instead of the ~270 of fe_to_mont.
// Used where a kernel needs the constant j + 1 of round j (vecops.hip k_nifs_cross_minroot).  k < 2^30.
template <class P> VDF_HD Fe<P> fe_from_small(uint32_t k) {
  uint32_t x[9];
  uint64_t carry = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    carry += (uint64_t)k * P::ONE[i];
    x[i] = (uint32_t)carry;
    carry >>= 32;
  }
  x[8] = (uint32_t)carry;                                         // k < 2^30: x < 2^286, x[8] < 2^30
  const uint32_t q = (x[8] << 2) | (x[7] >> 30);                  // floor(x / 2^254)
  x[7] &= 0x3FFFFFFFu;
  uint32_t qc[5];
  carry = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    carry += (uint64_t)q * P::MOD[i];
    qc[i] = (uint32_t)carry;
    carry >>= 32;
  }
  qc[4] = (uint32_t)carry;
  Fe<P> r;
  uint32_t borrow = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const uint64_t d = (uint64_t)x[i] - (i < 5 ? qc[i] : 0u) - borrow;
    r.v[i] = (uint32_t)d;
    borrow = (uint32_t)(d >> 63);
  }
  if (borrow) {                                                   // negative: add m (the result is then within 2^158 of m, below it)
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const uint64_t sum = (uint64_t)r.v[i] + P::MOD[i] + c;
      r.v[i] = (uint32_t)sum;
      c = (uint32_t)(sum >> 32);
    }
  }
  return r;
}

// 16-byte vectorised load/store of one 32-byte element (two dwordx4 per lane).
template <class P> VDF_HD Fe<P> fe_load(const void* p) {
  Fe<P> r;
#if defined(__HIP_DEVICE_COMPILE__)
  const uint4* q = reinterpret_cast<const uint4*>(p);
  uint4 lo = q[0], hi = q[1];
  r.v[0] = lo.x; r.v[1] = lo.y; r.v[2] = lo.z; r.v[3] = lo.w;
  r.v[4] = hi.x; r.v[5] = hi.y; r.v[6] = hi.z; r.v[7] = hi.w;
#else
  const uint32_t* q = reinterpret_cast<const uint32_t*>(p);
  for (int i = 0; i < 8; ++i) r.v[i] = q[i];
#endif
  return r;
}
template <class P> VDF_HD void fe_store(void* p, const Fe<P>& a) {
#if defined(__HIP_DEVICE_COMPILE__)
  uint4* q = reinterpret_cast<uint4*>(p);
  q[0] = make_uint4(a.v[0], a.v[1], a.v[2], a.v[3]);
  q[1] = make_uint4(a.v[4], a.v[5], a.v[6], a.v[7]);
#else
  uint32_t* q = reinterpret_cast<uint32_t*>(p);
  for (int i = 0; i < 8; ++i) q[i] = a.v[i];
#endif
}

}  // namespace vdf
