#!/usr/bin/env python3
"""Emit vdf_amd/csrc/pasta_constants.h: Montgomery constants of the two Pasta fields as
8 x 32-bit little-endian limbs (the device representation).
Values are derived here from the two moduli only."""
P = 0x40000000000000000000000000000000224698FC094CF91B992D30ED00000001
Q = 0x40000000000000000000000000000000224698FC0994A8DD8C46EB2100000001
R = 1 << 256


def limbs32(x):
    return ", ".join("0x%08xu" % ((x >> (32 * i)) & 0xFFFFFFFF) for i in range(8))


def cube_roots(m):
    g = 2
    while True:
        z = pow(g, (m - 1) // 3, m)
        if z != 1:
            return z, z * z % m
        g += 1


def curve_mul(k, pt, p):
    """[k] pt on y^2 = x^3 + 5 over F_p, affine (None = identity)."""
    def add(a, b):
        if a is None: return b
        if b is None: return a
        if a[0] == b[0]:
            if (a[1] + b[1]) % p == 0: return None
            lam = 3 * a[0] * a[0] * pow(2 * a[1], -1, p) % p
        else:
            lam = (b[1] - a[1]) * pow(b[0] - a[0], -1, p) % p
        x = (lam * lam - a[0] - b[0]) % p
        return (x, (lam * (a[0] - x) - a[1]) % p)
    acc = None
    while k:
        if k & 1: acc = add(acc, pt)
        pt = add(pt, pt)
        k >>= 1
    return acc


def glv(p, r):
    """The endomorphism phi(x, y) = (zeta x, y) = [lambda] of y^2 = x^3 + 5 over F_p (order r), and a reduced basis
    (a1, b1), (a2, b2) of the lattice a + b lambda = 0 (mod r) with a1, a2, b2 > 0 > b1 and a1 b2 - a2 b1 = r."""
    import math
    G = ((-1) % p, 2)
    zeta = lam = None
    for l in cube_roots(r):
        pt = curve_mul(l, G, p)
        for z in cube_roots(p):
            if pt == (z * G[0] % p, G[1]):
                zeta, lam = z, l
    assert zeta is not None
    s0, t0, r0, s1, t1, r1, rows = 1, 0, r, 0, 1, lam, []
    while r1:
        q = r0 // r1
        r0, r1 = r1, r0 - q * r1
        s0, s1 = s1, s0 - q * s1
        t0, t1 = t1, t0 - q * t1
        rows.append((r0, -t0))
    idx = next(i for i, (rr, _) in enumerate(rows) if rr < math.isqrt(r))
    a1, b1 = rows[idx]
    a2, b2 = min((rows[idx - 1], rows[idx + 1]), key=lambda v: v[0] * v[0] + v[1] * v[1])
    assert (a1 + b1 * lam) % r == 0 and (a2 + b2 * lam) % r == 0 and a1 > 0 and a2 > 0 and b1 < 0 < b2 and a1 * b2 - a2 * b1 == r
    return zeta, lam, a1, b1, a2, b2


def emit(name, m):
    assert m % (1 << 32) == 1 and (m >> 128) == 1 << 126
    inv32 = (-pow(m, -1, 1 << 32)) % (1 << 32)
    assert inv32 == 0xFFFFFFFF
    out = []
    out.append("struct %s {" % name)
    out.append("  static constexpr uint32_t MOD[8] = {%s};" % limbs32(m))
    out.append("  static constexpr uint32_t ONE[8] = {%s};   /* R mod m */" % limbs32(R % m))
    out.append("  static constexpr uint32_t R2[8]  = {%s};   /* R^2 mod m */" % limbs32(R * R % m))
    out.append("  static constexpr uint32_t THREE_B[8] = {%s}; /* 15*R mod m (3b, b = 5) */" % limbs32(15 * R % m))
    out.append("  static constexpr uint32_t FIVE[8] = {%s};  /* 5*R mod m (curve b) */" % limbs32(5 * R % m))
    out.append("  static constexpr uint32_t GEN_X[8] = {%s}; /* (-1)*R mod m */" % limbs32((m - 1) * R % m))
    out.append("  static constexpr uint32_t GEN_Y[8] = {%s}; /* 2*R mod m */" % limbs32(2 * R % m))
    T = (m - 1) >> 32
    assert (m - 1) == T << 32 and T & 1
    out.append("  /* Tonelli-Shanks: m - 1 = 2^32 * T.  TS_EXP = (T - 1) / 2; TS_Z = 5^T (5 generates the multiplicative")
    out.append("     group of both fields), a primitive 2^32-th root of unity, Montgomery form */")
    out.append("  static constexpr uint32_t TS_EXP[8] = {%s};" % limbs32((T - 1) // 2))
    assert pow(5, (m - 1) // 2, m) == m - 1
    out.append("  static constexpr uint32_t TS_Z[8] = {%s};" % limbs32(pow(5, T, m) * R % m))
    out.append("  /* (m - 2) for Fermat inversion */")
    out.append("  static constexpr uint32_t MOD_MINUS_2[8] = {%s};" % limbs32(m - 2))
    # GLV (msm.hip k_glv_split / k_glv_points).  This field is the BASE field of one curve and the SCALAR field of the other:
    other = Q if m == P else P
    zeta, _, _, _, _, _ = glv(m, other)            # the curve over F_m: phi(x, y) = (ZETA x, y)
    out.append("  /* GLV.  As the BASE field of its curve: phi(x, y) = (ZETA x, y) = [lambda] (x, y); ZETA in Montgomery form */")
    out.append("  static constexpr uint32_t GLV_ZETA[8] = {%s};" % limbs32(zeta * R % m))
    _, lam, a1, b1, a2, b2 = glv(other, m)         # the curve whose scalars live here
    g1, g2 = (b2 << 382) // m, ((-b1) << 382) // m
    assert g1 < R and g2 < R and max(a1, a2, -b1, b2) < 1 << 129
    out.append("  /* ... and as the SCALAR field of the other curve: k = k1 + lambda k2 with c1 = (k G1) >> 382, c2 = (k G2) >> 382,")
    out.append("     k1 = k - c1 A1 - c2 A2, k2 = c1 NB1 - c2 B2, |k1|, |k2| < 2^129 (lattice a + b lambda = 0: (A1, -NB1), (A2, B2)) */")
    out.append("  static constexpr uint32_t GLV_G1[8] = {%s};" % limbs32(g1))
    out.append("  static constexpr uint32_t GLV_G2[8] = {%s};" % limbs32(g2))
    out.append("  static constexpr uint32_t GLV_A1[8] = {%s};" % limbs32(a1))
    out.append("  static constexpr uint32_t GLV_A2[8] = {%s};" % limbs32(a2))
    out.append("  static constexpr uint32_t GLV_NB1[8] = {%s};" % limbs32(-b1))
    out.append("  static constexpr uint32_t GLV_B2[8] = {%s};" % limbs32(b2))
    out.append("  static constexpr uint32_t GLV_LAMBDA[8] = {%s};   /* plain integer (tests) */" % limbs32(lam))
    out.append("};")
    return "\n".join(out)


hdr = """// GENERATED by tools/gen_constants.py -- do not edit.
// Montgomery constants (R = 2^256) of the Pasta fields, 8 x u32 little-endian limbs.
// Both moduli are 2^254 + c with c < 2^126 and are == 1 (mod 2^32), so
// -m^-1 mod 2^32 = 0xFFFFFFFF, limbs 4..6 are zero and limb 7 is 0x40000000.
#pragma once
#include <stdint.h>

%s

%s
""" % (emit("FpParams", P), emit("FqParams", Q))
open("vdf_amd/csrc/pasta_constants.h", "w").write(hdr)
print(hdr)
