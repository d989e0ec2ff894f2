// Unsaturated-limb field arithmetic for the MSM bucket loop on gfx950: 9 limbs of 29 bits in
// 32-bit registers, Montgomery radix R' = 2^261.
//
// Why a second representation: on gfx950 every carry-writing VALU op (v_addc_co_u32 ...) issues as
// slowly as v_mad_u64_u32 (~3.3 cycles per wave-instruction per SIMD, tools/ubench/valu_rates.hip),
// so a saturated 8 x 32-bit multiply spends more issue slots on carries than on multiplies.  With
// 29-bit limbs the 64-bit column sums of a 9 x 9 product plus its reduction terms never overflow
// (9 * 2^31 * 2^29 + 5 * 2^58 < 2^64), so a multiply is 126 plain v_mad_u64_u32 and a handful of
// shifts -- no carry chain at all -- and add / sub are nine limb-wise VALU ops with lazy reduction.
//
// Values: an F29 holds x * 2^261 mod m, only lazily reduced (the comments give the bound of every
// intermediate as a multiple of m); "normalized" means every limb but the top one is < 2^29.
// mont29 needs one operand normalized and the other with limbs < 2^31; its output is normalized
// and < a*b/2^261 + m.  Used only inside k_accumulate; buckets are stored in the ordinary
// 8 x 32-bit, R = 2^256 form (f29_to_fe).
#pragma once
#include "fe.cuh"

namespace vdf {

template <class P29> struct F29 { uint32_t v[9]; };
static constexpr uint32_t MASK29 = 0x1FFFFFFFu;

template <class P> struct P29Of;
template <> struct P29Of<FpParams> { using type = Fp29; };
template <> struct P29Of<FqParams> { using type = Fq29; };

template <class Q> __device__ __forceinline__ F29<Q> f29_const(const uint32_t (&c)[9]) {
  F29<Q> r;
#pragma unroll
  for (int i = 0; i < 9; ++i) r.v[i] = c[i];
  return r;
}

// carry propagation: limbs < 2^32 in, normalized out (top limb takes the excess)
template <class Q> __device__ __forceinline__ void f29_normalize(F29<Q>& a) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    a.v[i + 1] += a.v[i] >> 29;
    a.v[i] &= MASK29;
  }
}

// (hi:lo) >> 29 with 32-bit ops (v_alignbit_b32 + v_lshrrev_b32): 64-bit shifts are slow VALU ops
__device__ __forceinline__ uint64_t shr29(uint64_t c) {
  const uint32_t lo = (uint32_t)c, hi = (uint32_t)(c >> 32);
  const uint32_t rlo = __builtin_amdgcn_alignbit(hi, lo, 29);
  const uint32_t rhi = hi >> 29;
  return ((uint64_t)rhi << 32) | rlo;
}

// a*b/2^261 mod m (lazy).  One operand normalized, the other with limbs < 2^31.
template <class Q> __device__ __forceinline__ F29<Q> f29_mul(const F29<Q>& a, const F29<Q>& b) {
  uint64_t c[18];
#pragma unroll
  for (int i = 0; i < 18; ++i) c[i] = 0;
  uint32_t m8;                                   // 2^22 hidden from the optimiser: keeps q*m8 a single mad
  asm("s_mov_b32 %0, 0x400000" : "=s"(m8));
#pragma unroll
  for (int i = 0; i < 9; ++i) {
#pragma unroll
    for (int j = 0; j < 9; ++j) c[i + j] += (uint64_t)a.v[j] * b.v[i];
    // Montgomery digit: m == 1 (mod 2^29) so q = -c[i] mod 2^29; m limbs 5..7 are zero, limb 8 = 2^22
    const uint32_t q = (0u - (uint32_t)c[i]) & MASK29;
    c[i + 1] += (uint64_t)q * Q::M[1];
    c[i + 2] += (uint64_t)q * Q::M[2];
    c[i + 3] += (uint64_t)q * Q::M[3];
    c[i + 4] += (uint64_t)q * Q::M[4];
    c[i + 8] += (uint64_t)q * m8;
    c[i + 1] += shr29(c[i] + q);                  // c[i] + q*1 is a multiple of 2^29
  }
  F29<Q> r;
#pragma unroll
  for (int k = 9; k < 17; ++k) {
    c[k + 1] += shr29(c[k]);
    r.v[k - 9] = (uint32_t)c[k] & MASK29;
  }
  r.v[8] = (uint32_t)c[17];
  return r;
}

// Out-of-line copy for the tail kernels (dozens of multiplies per kernel: keeps compile time sane)
template <class Q> __device__ __attribute__((noinline)) F29<Q> f29_mul_call(F29<Q> a, F29<Q> b) { return f29_mul(a, b); }
template <class Q> __device__ __forceinline__ F29<Q> f29_mulc(const F29<Q>& a, const F29<Q>& b) { return f29_mul_call<Q>(a, b); }

// limb-wise a + bias - b; `bias` is a multiple of m with borrow-proof limbs (pasta_constants.h)
template <class Q> __device__ __forceinline__ F29<Q> f29_sub(const F29<Q>& a, const F29<Q>& b, const uint32_t (&bias)[9]) {
  F29<Q> r;
#pragma unroll
  for (int i = 0; i < 9; ++i) r.v[i] = a.v[i] + bias[i] - b.v[i];
  return r;
}
// a + bias - b - 2c
template <class Q> __device__ __forceinline__ F29<Q> f29_sub3(const F29<Q>& a, const F29<Q>& b, const F29<Q>& c, const uint32_t (&bias)[9]) {
  F29<Q> r;
#pragma unroll
  for (int i = 0; i < 9; ++i) r.v[i] = a.v[i] + bias[i] - b.v[i] - 2u * c.v[i];
  return r;
}

// 8 x 32-bit value v (< m, R = 2^256 Montgomery form) -> the limbs of 32*v, i.e. the same field
// element in the R' = 2^261 domain, unreduced (< 32m) but normalized: a pure bit repack.
template <class P> __device__ __forceinline__ F29<typename P29Of<P>::type> f29_from_fe_x32(const Fe<P>& a) {
  F29<typename P29Of<P>::type> r;
  r.v[0] = (a.v[0] << 5) & MASK29;
#pragma unroll
  for (int k = 1; k < 8; ++k) {
    const int o = 29 * k - 5, w = o >> 5, s = o & 31;
    const uint64_t two = ((uint64_t)a.v[w + 1] << 32) | a.v[w];
    r.v[k] = (uint32_t)(two >> s) & MASK29;
  }
  r.v[8] = a.v[7] >> 3;            // bits 227..255
  return r;
}

// normalized value < 2m -> canonical (< m): subtract m once if it does not borrow
template <class Q> __device__ __forceinline__ void f29_cond_sub(F29<Q>& a) {
  uint32_t d[9];
  uint32_t borrow = 0;
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    const uint32_t t = a.v[i] - Q::M[i] - borrow;
    borrow = t >> 31;
    d[i] = (i < 8) ? (t & MASK29) : t;
  }
#pragma unroll
  for (int i = 0; i < 9; ++i) a.v[i] = borrow ? a.v[i] : d[i];
}

// canonical 9 x 29 -> 8 x 32
template <class P> __device__ __forceinline__ Fe<P> f29_pack(const F29<typename P29Of<P>::type>& a) {
  Fe<P> r;
#pragma unroll
  for (int w = 0; w < 8; ++w) {
    const int i = (32 * w) / 29, s = 32 * w - 29 * i;
    uint64_t t = (uint64_t)a.v[i] >> s;
    t |= (uint64_t)a.v[i + 1] << (29 - s);
    if (i + 2 < 9) t |= (uint64_t)a.v[i + 2] << (58 - s);
    r.v[w] = (uint32_t)t;
  }
  return r;
}

// R' domain (lazy, one operand rule satisfied by the normalized constant) -> canonical R = 2^256 form
template <class P> __device__ __forceinline__ Fe<P> f29_to_fe(const F29<typename P29Of<P>::type>& a) {
  using Q = typename P29Of<P>::type;
  F29<Q> t = f29_mul(a, f29_const<Q>(Q::TO_R256));      // a * 2^-5, < 2m, normalized
  f29_cond_sub(t);
  return f29_pack<P>(t);
}

// exact test v == 0 (mod m) for a normalized value < 4m (rare path)
template <class Q> __device__ __forceinline__ bool f29_is_zero_mod(F29<Q> a) {
  f29_cond_sub(a);
  f29_cond_sub(a);
  f29_cond_sub(a);
  uint32_t o = 0;
#pragma unroll
  for (int i = 0; i < 9; ++i) o |= a.v[i];
  return o == 0;
}

}  // namespace vdf
