#include "host_math.hpp"
#include "../pasta_constants.h"

#include <cstdlib>

namespace vdfhost {

// BMI2 (mulx) and ADX (adcx / adox) by cpuid, once, before main: selects the product of fe_mul_x86_adx.inc
static bool detect_adx() {
  if (const char* e = std::getenv("VDF_HOST_NO_ADX")) if (e[0] == '1') return false;
  __builtin_cpu_init();
  return __builtin_cpu_supports("bmi2") && __builtin_cpu_supports("adx");
}
const bool g_has_adx = detect_adx();

static Field make_field(const uint32_t mod[8], const uint32_t one_[8], const uint32_t r2_[8]) {
  static_assert(FpParams::MOD[4] == 0 && FpParams::MOD[5] == 0 && FpParams::MOD[6] == 0 && FpParams::MOD[7] == 0x40000000u &&
                FqParams::MOD[4] == 0 && FqParams::MOD[5] == 0 && FqParams::MOD[6] == 0 && FqParams::MOD[7] == 0x40000000u,
                "mont_reduce hard-codes m[2] = 0, m[3] = 2^62");
  Field F;
  for (int i = 0; i < 4; ++i) {
    F.m[i] = ((uint64_t)mod[2 * i + 1] << 32) | mod[2 * i];
    F.one[i] = ((uint64_t)one_[2 * i + 1] << 32) | one_[2 * i];
    F.r2[i] = ((uint64_t)r2_[2 * i + 1] << 32) | r2_[2 * i];
  }
  F.inv = neg_inv64(F.m[0]);
  return F;
}
const Field& field_fp() {
  static const Field F = make_field(FpParams::MOD, FpParams::ONE, FpParams::R2);
  return F;
}
const Field& field_fq() {
  static const Field F = make_field(FqParams::MOD, FqParams::ONE, FqParams::R2);
  return F;
}

// EFD xyzz dbl-2008-s-1 (a = 0)
Pt pt_dbl(const Pt& a, const Field& F) {
  if (a.is_id()) return a;
  Fe U = add(a.y, a.y, F), V = sqr(U, F), W = mul(U, V, F), S = mul(a.x, V, F);
  Fe x2 = sqr(a.x, F), M = add(add(x2, x2, F), x2, F);
  Pt r;
  r.x = sub(sub(sqr(M, F), S, F), S, F);
  r.y = sub(mul(M, sub(S, r.x, F), F), mul(W, a.y, F), F);
  r.zz = mul(V, a.zz, F);
  r.zzz = mul(W, a.zzz, F);
  return r;
}
// EFD xyzz add-2008-s with the exceptional cases
Pt pt_add(const Pt& a, const Pt& b, const Field& F) {
  if (b.is_id()) return a;
  if (a.is_id()) return b;
  Fe U1 = mul(a.x, b.zz, F), U2 = mul(b.x, a.zz, F), S1 = mul(a.y, b.zzz, F), S2 = mul(b.y, a.zzz, F);
  Fe P = sub(U2, U1, F), R = sub(S2, S1, F);
  if (P.is_zero()) return R.is_zero() ? pt_dbl(a, F) : pt_identity();
  Fe PP = sqr(P, F), PPP = mul(P, PP, F), Q = mul(U1, PP, F);
  Pt r;
  r.x = sub(sub(sub(sqr(R, F), PPP, F), Q, F), Q, F);
  r.y = sub(mul(R, sub(Q, r.x, F), F), mul(S1, PPP, F), F);
  r.zz = mul(mul(a.zz, b.zz, F), PP, F);
  r.zzz = mul(mul(a.zzz, b.zzz, F), PPP, F);
  return r;
}
Pt pt_mul(const Pt& a, const uint64_t k[4], int bits, const Field& F) {
  Pt r = pt_identity();
  for (int b = bits - 1; b >= 0; --b) {
    r = pt_dbl(r, F);
    if ((k[b / 64] >> (b % 64)) & 1) r = pt_add(r, a, F);
  }
  return r;
}
Aff pt_to_aff(const Pt& a, const Field& F) {
  Aff r;
  if (a.is_id()) { r.x = r.y = zero(); return r; }
  Fe izzz = inverse(a.zzz, F);
  Fe izz = mul(sqr(izzz, F), sqr(a.zz, F), F);     // zz^3 = zzz^2
  r.x = mul(a.x, izz, F);
  r.y = mul(a.y, izzz, F);
  return r;
}
void pt_to_aff2(const Pt& pa, const Pt& pb, const Field& F, Aff* a, Aff* b) {
  if (pa.is_id() || pb.is_id()) { *a = pt_to_aff(pa, F); *b = pt_to_aff(pb, F); return; }
  const Fe inv_ab = inverse(mul(pa.zzz, pb.zzz, F), F);
  auto fin = [&](const Pt& q, const Fe& izzz, Aff* o) {
    o->x = mul(q.x, mul(sqr(izzz, F), sqr(q.zz, F), F), F);       // 1 / zz = zz^2 / zzz^2
    o->y = mul(q.y, izzz, F);
  };
  fin(pa, mul(inv_ab, pb.zzz, F), a);
  fin(pb, mul(inv_ab, pa.zzz, F), b);
}
Pt pt_from_jac(const vdf_jac& j, const Field& F) {
  Pt p;
  Fe Z;
  memcpy(p.x.l, j.x.l, 32); memcpy(p.y.l, j.y.l, 32); memcpy(Z.l, j.z.l, 32);
  if (Z.is_zero()) return pt_identity();
  p.zz = sqr(Z, F);
  p.zzz = mul(p.zz, Z, F);
  return p;
}
Aff jac_to_aff(const vdf_jac& j, const Field& F) {
  Fe X, Y, Z;
  memcpy(X.l, j.x.l, 32); memcpy(Y.l, j.y.l, 32); memcpy(Z.l, j.z.l, 32);
  Aff r;
  if (Z.is_zero()) { r.x = r.y = zero(); return r; }
  Fe zi = inverse(Z, F), zi2 = sqr(zi, F);
  r.x = mul(X, zi2, F);
  r.y = mul(Y, mul(zi2, zi, F), F);
  return r;
}

// Two Jacobian points to affine with one shared field inversion (Montgomery's trick).
void jac_to_aff2(const vdf_jac& ja, const vdf_jac& jb, const Field& F, Aff* a, Aff* b) {
  Fe Za, Zb;
  memcpy(Za.l, ja.z.l, 32); memcpy(Zb.l, jb.z.l, 32);
  if (Za.is_zero() || Zb.is_zero()) { *a = jac_to_aff(ja, F); *b = jac_to_aff(jb, F); return; }
  const Fe inv_ab = inverse(mul(Za, Zb, F), F);
  const Fe zia = mul(inv_ab, Zb, F), zib = mul(inv_ab, Za, F);
  auto fin = [&](const vdf_jac& j, const Fe& zi, Aff* o) {
    Fe X, Y;
    memcpy(X.l, j.x.l, 32); memcpy(Y.l, j.y.l, 32);
    const Fe zi2 = sqr(zi, F);
    o->x = mul(X, zi2, F);
    o->y = mul(Y, mul(zi2, zi, F), F);
  };
  fin(ja, zia, a);
  fin(jb, zib, b);
}

// ---- square roots and the 32-byte point encoding --------------------------------------------------
// m - 1 = 2^32 * T with T odd.  5 generates the multiplicative group of both fields, so z = 5^T has order 2^32.
bool fe_sqrt(const Fe& a, const Field& F, Fe* out) {
  if (a.is_zero()) { *out = a; return true; }
  uint64_t T[4] = {(F.m[0] >> 32) | (F.m[1] << 32), (F.m[1] >> 32) | (F.m[2] << 32), (F.m[2] >> 32) | (F.m[3] << 32), F.m[3] >> 32};
  uint64_t Th[4] = {(T[0] >> 1) | (T[1] << 63), (T[1] >> 1) | (T[2] << 63), (T[2] >> 1) | (T[3] << 63), T[3] >> 1};   // (T - 1) / 2
  Fe z = pow_vartime(from_u64(5, F), T, F);
  const Fe w = pow_vartime(a, Th, F);
  Fe x = mul(a, w, F);            // a^((T+1)/2)
  Fe b = mul(x, w, F);            // a^T
  const Fe o = one(F);
  int v = 32;
  while (b != o) {
    int k = 0;
    for (Fe t = b; t != o; t = sqr(t, F)) if (++k == v) return false;     // order 2^v: not a square
    for (int j = 0; j < v - k - 1; ++j) z = sqr(z, F);
    x = mul(x, z, F);
    z = sqr(z, F);
    b = mul(b, z, F);
    v = k;
  }
  *out = x;
  return true;
}

void pt_compress(const Aff& a, const Field& F, uint8_t out[32]) {
  if (a.is_id()) { memset(out, 0, 32); return; }
  const Fe x = from_mont(a.x, F), y = from_mont(a.y, F);
  memcpy(out, x.l, 32);
  out[31] |= (uint8_t)((y.l[0] & 1) << 7);
}

bool pt_decompress(const uint8_t in[32], const Field& F, Aff* out) {
  Fe x;
  memcpy(x.l, in, 32);
  const uint64_t odd = x.l[3] >> 63;
  x.l[3] &= ~(1ull << 63);
  if (geq(x.l, F.m)) return false;
  if (x.is_zero()) {
    out->x = out->y = zero();
    return !odd;                  // the identity has one encoding
  }
  const Fe xm = to_mont(x, F);
  Fe y;
  if (!fe_sqrt(add(mul(sqr(xm, F), xm, F), from_u64(5, F), F), F, &y)) return false;
  if ((from_mont(y, F).l[0] & 1) != odd) y = neg(y, F);
  out->x = xm; out->y = y;
  return true;
}

// ---- Keccak-f[1600] / SHAKE256 (FIPS 202) ---------------------------------------------------------
static inline uint64_t rotl(uint64_t x, int n) { return (x << n) | (x >> (64 - n)); }
static void keccak_f(uint64_t s[25]) {
  static const uint64_t RC[24] = {
      0x0000000000000001ull, 0x0000000000008082ull, 0x800000000000808aull, 0x8000000080008000ull, 0x000000000000808bull,
      0x0000000080000001ull, 0x8000000080008081ull, 0x8000000000008009ull, 0x000000000000008aull, 0x0000000000000088ull,
      0x0000000080008009ull, 0x000000008000000aull, 0x000000008000808bull, 0x800000000000008bull, 0x8000000000008089ull,
      0x8000000000008003ull, 0x8000000000008002ull, 0x8000000000000080ull, 0x000000000000800aull, 0x800000008000000aull,
      0x8000000080008081ull, 0x8000000000008080ull, 0x0000000080000001ull, 0x8000000080008008ull};
  static const int ROT[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};
  for (int round = 0; round < 24; ++round) {
    uint64_t C[5], D[5], B[25];
    for (int x = 0; x < 5; ++x) C[x] = s[x] ^ s[x + 5] ^ s[x + 10] ^ s[x + 15] ^ s[x + 20];
    for (int x = 0; x < 5; ++x) D[x] = C[(x + 4) % 5] ^ rotl(C[(x + 1) % 5], 1);
    for (int i = 0; i < 25; ++i) s[i] ^= D[i % 5];
    for (int x = 0; x < 5; ++x)
      for (int y = 0; y < 5; ++y) {
        int i = x + 5 * y;
        uint64_t v = ROT[i] ? rotl(s[i], ROT[i]) : s[i];
        B[y + 5 * ((2 * x + 3 * y) % 5)] = v;
      }
    for (int y = 0; y < 5; ++y)
      for (int x = 0; x < 5; ++x) s[x + 5 * y] = B[x + 5 * y] ^ ((~B[(x + 1) % 5 + 5 * y]) & B[(x + 2) % 5 + 5 * y]);
    s[0] ^= RC[round];
  }
}
Shake256::Shake256() : pos(0), squeezing(false) { memset(st, 0, sizeof(st)); memset(buf, 0, sizeof(buf)); }
static void xor_block(uint64_t st[25], const uint8_t* b) {
  for (int i = 0; i < 17; ++i) {
    uint64_t w;
    memcpy(&w, b + 8 * i, 8);
    st[i] ^= w;
  }
}
void Shake256::absorb(const void* data, size_t n) {
  const uint8_t* p = (const uint8_t*)data;
  while (n) {
    size_t take = 136 - pos < n ? 136 - pos : n;
    memcpy(buf + pos, p, take);
    pos += take; p += take; n -= take;
    if (pos == 136) { xor_block(st, buf); keccak_f(st); pos = 0; }
  }
}
void Shake256::squeeze(void* out, size_t n) {
  uint8_t* o = (uint8_t*)out;
  if (!squeezing) {
    memset(buf + pos, 0, 136 - pos);
    buf[pos] ^= 0x1F;
    buf[135] ^= 0x80;
    xor_block(st, buf);
    keccak_f(st);
    squeezing = true;
    pos = 0;
  }
  while (n) {
    if (pos == 136) { keccak_f(st); pos = 0; }
    size_t take = 136 - pos < n ? 136 - pos : n;
    memcpy(o, (const uint8_t*)st + pos, take);
    pos += take; o += take; n -= take;
  }
}

}  // namespace vdfhost
