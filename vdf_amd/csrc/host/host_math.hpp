// Host-side arithmetic of the Nova layer: what the reference also keeps on the CPU -- the sequential
// MinRoot evaluation (src/minroot.rs:329-359), O(1) instance folds and transcript hashing.  4 x 64-bit
// Montgomery limbs (R = 2^256), the in-memory form of pasta_curves' Fp/Fq (Cargo.toml:17), so values
// cross the C ABI unchanged.  No kernel work lives here.
#pragma once
#include <stdint.h>
#include <string.h>
#include <x86intrin.h>
#include <vector>
#include "../../../include/vdf_hip.h"

namespace vdfhost {

typedef unsigned __int128 u128;

struct Field {
  uint64_t m[4], inv, one[4], r2[4];
};

inline uint64_t neg_inv64(uint64_t m0) {          // -m^-1 mod 2^64 by Newton iteration
  uint64_t x = 1;
  for (int i = 0; i < 6; ++i) x *= 2 - m0 * x;
  return 0 - x;
}

inline bool geq(const uint64_t* a, const uint64_t* b) {
  for (int i = 3; i >= 0; --i) {
    if (a[i] != b[i]) return a[i] > b[i];
  }
  return true;
}
inline void sub4(uint64_t* a, const uint64_t* b) {
  u128 br = 0;
  for (int i = 0; i < 4; ++i) {
    u128 d = (u128)a[i] - b[i] - (uint64_t)br;
    a[i] = (uint64_t)d;
    br = (d >> 64) & 1;
  }
}

struct Fe {
  uint64_t l[4];
  bool is_zero() const { return (l[0] | l[1] | l[2] | l[3]) == 0; }
  bool operator==(const Fe& o) const { return memcmp(l, o.l, 32) == 0; }
  bool operator!=(const Fe& o) const { return !(*this == o); }
};
static_assert(sizeof(Fe) == sizeof(vdf_fe), "layout");

// Branch-free: whether a sum wraps past m is a coin flip for random operands, and the linear layers of the hash are
// hundreds of additions per permutation -- a mispredicted branch each would cost more than the multiplications.
inline Fe add(const Fe& a, const Fe& b, const Field& F) {
  unsigned long long r0, r1, r2, r3, t0, t1, t2, t3;
  unsigned char c = _addcarry_u64(0, a.l[0], b.l[0], &r0);
  c = _addcarry_u64(c, a.l[1], b.l[1], &r1);
  c = _addcarry_u64(c, a.l[2], b.l[2], &r2);
  c = _addcarry_u64(c, a.l[3], b.l[3], &r3);                     // operands below 2^255: no carry out
  unsigned char br = _subborrow_u64(0, r0, F.m[0], &t0);
  br = _subborrow_u64(br, r1, F.m[1], &t1);
  br = _subborrow_u64(br, r2, F.m[2], &t2);
  br = _subborrow_u64(br, r3, F.m[3], &t3);
  const uint64_t keep = 0 - (uint64_t)(br & (unsigned char)(c ^ 1));    // all ones: the sum was below m already
  Fe r;
  r.l[0] = (r0 & keep) | (t0 & ~keep); r.l[1] = (r1 & keep) | (t1 & ~keep);
  r.l[2] = (r2 & keep) | (t2 & ~keep); r.l[3] = (r3 & keep) | (t3 & ~keep);
  return r;
}
inline Fe sub(const Fe& a, const Fe& b, const Field& F) {
  unsigned long long r0, r1, r2, r3;
  unsigned char br = _subborrow_u64(0, a.l[0], b.l[0], &r0);
  br = _subborrow_u64(br, a.l[1], b.l[1], &r1);
  br = _subborrow_u64(br, a.l[2], b.l[2], &r2);
  br = _subborrow_u64(br, a.l[3], b.l[3], &r3);
  const uint64_t fix = 0 - (uint64_t)br;                          // all ones: add m back
  Fe r;
  unsigned long long o0, o1, o2, o3;
  unsigned char c = _addcarry_u64(0, r0, F.m[0] & fix, &o0);
  c = _addcarry_u64(c, r1, F.m[1] & fix, &o1);
  c = _addcarry_u64(c, r2, F.m[2] & fix, &o2);
  c = _addcarry_u64(c, r3, F.m[3] & fix, &o3);
  (void)c;
  r.l[0] = o0; r.l[1] = o1; r.l[2] = o2; r.l[3] = o3;
  return r;
}
inline Fe neg(const Fe& a, const Field& F) {
  Fe z = {{0, 0, 0, 0}};
  return a.is_zero() ? a : sub(z, a, F);
}
// Montgomery product for the two Pasta moduli, m = 2^254 + c with c < 2^126: in 64-bit limbs m[2] = 0 and
// m[3] = 2^62 (checked in host_math.cpp), so a reduction step is two multiplications and a shift instead of four.
// Inline on purpose: the forward MinRoot evaluation -- the delay itself, ~285 of these per iteration, strictly
// sequential (src/minroot.rs:329-359) -- is the end-to-end wall-clock floor of a prover, and its entry points are
// compiled twice (baseline x86-64 and a BMI2/ADX clone picked at load time) with everything inlined into them.
// LAZY: operands and result in [0, 2m) instead of [0, m): with R = 2^256 and m < 2^254 (1 + 2^-128), a, b < 2m give
// (a b + q m) / R < (4 m^2 + R m) / R < 2m, and a b + q m < 2^512 -- so a chain of multiplications needs neither the
// comparison nor the subtraction (a ~20 % mispredicted branch per operation) until its end (canon).
template <bool LAZY = false>
inline Fe mont_reduce(uint64_t t[8], const Field& F) {
  uint64_t top = 0;                                   // carry out of limb 7
  for (int i = 0; i < 4; ++i) {
    const uint64_t q = t[i] * F.inv;
    // t += q * (m0 + m1 * 2^64 + 2^254) * 2^(64 i)
    u128 c = (u128)q * F.m[0] + t[i];
    c = (c >> 64) + (u128)q * F.m[1] + t[i + 1];
    t[i + 1] = (uint64_t)c;
    c = (c >> 64) + t[i + 2];
    t[i + 2] = (uint64_t)c;
    c = (c >> 64) + t[i + 3] + (q << 62);
    t[i + 3] = (uint64_t)c;
    c = (c >> 64) + (q >> 2);
    for (int j = i + 4; j < 8; ++j) { c += t[j]; t[j] = (uint64_t)c; c >>= 64; }
    top += (uint64_t)c;
  }
  Fe r;
  memcpy(r.l, t + 4, 32);
  if (!LAZY && (top || geq(r.l, F.m))) sub4(r.l, F.m);
  return r;
}
inline Fe canon(Fe r, const Field& F) {               // [0, 2m) -> [0, m)
  if (geq(r.l, F.m)) sub4(r.l, F.m);
  return r;
}
// The same product as one block of mulx / adcx / adox where the CPU has BMI2 and ADX (every x86-64 server core since
// Broadwell / Zen): generated by tools/gen_host_mul.py.  The compiler's own code for the portable form below keeps two
// serial carry chains in flags and spends ~50 cycles per product; this one overlaps them (g_has_adx: host_math.cpp, cpuid
// at load time; VDF_HOST_NO_ADX=1 forces the portable form).
extern const bool g_has_adx;
template <bool LAZY = false>
inline Fe mul_adx(const Fe& a, const Fe& b, const Field& F) {
  uint64_t t0, t1, t2, t3, t4, t5, lo, hi;
  Fe r;
#include "fe_mul_x86_adx.inc"
  if (!LAZY && geq(r.l, F.m)) sub4(r.l, F.m);
  return r;
}
template <bool LAZY = false>
inline Fe mul(const Fe& a, const Fe& b, const Field& F) {
  if (g_has_adx) return mul_adx<LAZY>(a, b, F);
  uint64_t t[8];
  {
    u128 c = 0;
    for (int j = 0; j < 4; ++j) { c += (u128)a.l[j] * b.l[0]; t[j] = (uint64_t)c; c >>= 64; }
    t[4] = (uint64_t)c;
  }
  for (int i = 1; i < 4; ++i) {
    u128 c = 0;
    for (int j = 0; j < 4; ++j) { c += (u128)a.l[j] * b.l[i] + t[i + j]; t[i + j] = (uint64_t)c; c >>= 64; }
    t[i + 4] = (uint64_t)c;
  }
  return mont_reduce<LAZY>(t, F);
}
// Dedicated squaring: the six cross products once, doubled, plus the four squares -- 10 multiplications instead of
// 16.  (On the GPU the same trick does not pay, DESIGN.md 4.2: there a carry costs what a multiplication does.)
template <bool LAZY = false>
inline Fe sqr(const Fe& a, const Field& F) {
  if (g_has_adx) return mul_adx<LAZY>(a, a, F);
  uint64_t t[8];
  u128 c = (u128)a.l[0] * a.l[1];
  t[1] = (uint64_t)c;
  c = (c >> 64) + (u128)a.l[0] * a.l[2];
  t[2] = (uint64_t)c;
  c = (c >> 64) + (u128)a.l[0] * a.l[3];
  t[3] = (uint64_t)c;
  t[4] = (uint64_t)(c >> 64);
  c = (u128)a.l[1] * a.l[2] + t[3];
  t[3] = (uint64_t)c;
  c = (c >> 64) + (u128)a.l[1] * a.l[3] + t[4];
  t[4] = (uint64_t)c;
  t[5] = (uint64_t)(c >> 64);
  c = (u128)a.l[2] * a.l[3] + t[5];
  t[5] = (uint64_t)c;
  t[6] = (uint64_t)(c >> 64);
  t[7] = t[6] >> 63;
  for (int i = 6; i > 1; --i) t[i] = (t[i] << 1) | (t[i - 1] >> 63);
  t[1] <<= 1;
  c = (u128)a.l[0] * a.l[0];
  t[0] = (uint64_t)c;
  c = (c >> 64) + t[1];
  t[1] = (uint64_t)c;
  for (int i = 1; i < 4; ++i) {
    c = (c >> 64) + (u128)a.l[i] * a.l[i] + t[2 * i];
    t[2 * i] = (uint64_t)c;
    c = (c >> 64) + t[2 * i + 1];
    t[2 * i + 1] = (uint64_t)c;
  }
  return mont_reduce<LAZY>(t, F);
}
inline Fe one(const Field& F) { Fe r; memcpy(r.l, F.one, 32); return r; }
inline Fe zero() { Fe r = {{0, 0, 0, 0}}; return r; }
inline Fe to_mont(const Fe& a, const Field& F) { Fe r2; memcpy(r2.l, F.r2, 32); return mul(a, r2, F); }
inline Fe from_mont(const Fe& a, const Field& F) { Fe o = {{1, 0, 0, 0}}; return mul(a, o, F); }
inline Fe from_u64(uint64_t v, const Field& F) { Fe t = {{v, 0, 0, 0}}; return to_mont(t, F); }
// left-to-right square and multiply: the shape of ff::Field::pow_vartime (src/minroot.rs:312-314)
inline Fe pow_vartime(const Fe& a, const uint64_t e[4], const Field& F) {
  Fe acc = one(F);
  for (int i = 3; i >= 0; --i)
    for (int b = 63; b >= 0; --b) {
      acc = sqr(acc, F);
      if ((e[i] >> b) & 1) acc = mul(acc, a, F);
    }
  return acc;
}
inline Fe inverse(const Fe& a, const Field& F) {
  uint64_t e[4] = {F.m[0] - 2, F.m[1], F.m[2], F.m[3]};
  return pow_vartime(a, e, F);
}

const Field& field_fp();
const Field& field_fq();
inline const Field& field(int f) { return f == VDF_FIELD_FP ? field_fp() : field_fq(); }

// ---- curve y^2 = x^3 + 5 on the host (instance folds: two 128-bit scalar multiplications per step) --
struct Aff { Fe x, y; bool is_id() const { return x.is_zero() && y.is_zero(); } };
struct Pt { Fe x, y, zz, zzz; bool is_id() const { return zz.is_zero(); } };   // XYZZ

inline Pt pt_identity() { Pt p; p.x = p.y = p.zz = p.zzz = zero(); return p; }
inline Pt pt_from_aff(const Aff& a, const Field& F) {
  if (a.is_id()) return pt_identity();
  Pt p; p.x = a.x; p.y = a.y; p.zz = one(F); p.zzz = one(F);
  return p;
}
Pt pt_dbl(const Pt& a, const Field& F);
Pt pt_add(const Pt& a, const Pt& b, const Field& F);
Pt pt_mul(const Pt& a, const uint64_t k[4], int bits, const Field& F);
Aff pt_to_aff(const Pt& a, const Field& F);
void pt_to_aff2(const Pt& pa, const Pt& pb, const Field& F, Aff* a, Aff* b);     // one inversion for both
Pt pt_from_jac(const vdf_jac& j, const Field& F);                                // (X, Y, Z) -> (X, Y, Z^2, Z^3): no inversion
Aff jac_to_aff(const vdf_jac& j, const Field& F);
void jac_to_aff2(const vdf_jac& ja, const vdf_jac& jb, const Field& F, Aff* a, Aff* b);

// Square root (Tonelli-Shanks; both Pasta moduli are 1 mod 2^32): false when `a` is not a square.
bool fe_sqrt(const Fe& a, const Field& F, Fe* out);
// 32-byte point encoding of the wire formats: canonical little-endian x with the parity of y in bit 255; the
// identity is 32 zero bytes (x = 0 is on neither curve: 5 is not a square).  decompress: false unless the bytes
// are exactly what compress would write for some point.
void pt_compress(const Aff& a, const Field& F, uint8_t out[32]);
bool pt_decompress(const uint8_t in[32], const Field& F, Aff* out);

// ---- SHAKE256 (FIPS 202) for the transcript -----------------------------------------------------
struct Shake256 {
  uint64_t st[25];
  uint8_t buf[136];
  size_t pos;
  bool squeezing;
  Shake256();
  void absorb(const void* data, size_t n);
  void squeeze(void* out, size_t n);
};

}  // namespace vdfhost
