"""TEST INFRASTRUCTURE -- never imported by the product (vdf_amd/).

The random oracle of the Nova layer ("vdf-ro-v1"): a Poseidon2-style permutation (Grassi, Khovratovich, Schofnegger,
"Poseidon2", 2023) of width 4 over either Pasta field, S-box x^5 (gcd(5, m - 1) = 1 in both fields -- the MinRoot
exponent, src/minroot.rs:273-285), 8 external + 56 internal rounds, in a sponge of rate 3 / capacity 1.

The reference reaches its RO through nova-snark 0.8.0 -> neptune 7.2.0 (Cargo.toml:14-15), neither of which is in
/root/reference; their constants are unknowable here, so the constants below are this
build's own, derived from SHAKE256 of a label: PARITY UNPINNED against nova-snark, pinned product-vs-oracle.

External linear layer: the 4 x 4 matrix M4 of the paper.  Internal linear layer: ones off the diagonal, MU[i] on it;
`internal_matrix_ok` checks the paper's condition for it (the characteristic polynomial of M_I^k is irreducible for
k = 1..8, so no invariant subspace trail survives), and tests/test_oracle_nova.py runs that check.
"""
from __future__ import annotations

import hashlib
from functools import lru_cache
from typing import List, Sequence

from contextlib import contextmanager
from dataclasses import dataclass

from . import pasta as o


# ---- the random oracle as a PARAMETER BLOCK -----------------------------------------------
# Everything the RO is made of, as data that the parameters' digest covers (digest_shapes absorbs `label()`, and the
# shapes themselves change with it): the permutation family, width, round numbers, S-box exponent, and the two
# truncations of the protocol.  Two instances ship:
#   DEFAULT         this build's own: Poseidon2-style, width 4, 8 + 56 rounds, constants from SHAKE256 (below);
#   NEPTUNE_SHAPED  [UPSTREAM-RECALL] the shape nova-snark 0.8.0 reaches through neptune 7.2.0 (Cargo.toml:14-15): the
#                   original Poseidon permutation (Grassi et al., 2019) of width 25 (a sponge of rate 24), x^5, 8 full and 57
#                   partial rounds, Cauchy MDS matrix M[i][j] = 1 / (x_i + y_j) with x_i = i, y_j = t + j, round constants
#                   from the paper's Grain LFSR.  Neither crate is in /root/reference: the numbers are recalled, nothing
#                   pins them, and interoperability is NOT claimed -- what the instance proves is that the day the upstream
#                   constants are available, adopting them is a change of DATA (this block), not of code.
@dataclass(frozen=True)
class RoSpec:
    family: int = 0              # 0 = Poseidon2-style (this build), 1 = original Poseidon (Cauchy MDS, Grain constants)
    width: int = 4
    full_rounds: int = 8
    partial_rounds: int = 56
    alpha: int = 5
    challenge_bits: int = 128
    hash_bits: int = 250


DEFAULT = RoSpec()
NEPTUNE_SHAPED = RoSpec(family=1, width=25, full_rounds=8, partial_rounds=57)
_current = DEFAULT


@contextmanager
def using(spec: RoSpec):
    """Every hash of the oracle (native and in-circuit) under `spec` inside the block."""
    global _current
    prev, _current = _current, spec
    try:
        yield spec
    finally:
        _current = prev


def current() -> RoSpec:
    return _current


def label() -> bytes:
    """What the parameters' digest absorbs for the RO: the historical label for the default, the whole block otherwise."""
    s = _current
    if s == DEFAULT:
        return LABEL
    return b"vdf-ro-block-v1:" + bytes([s.family, s.width, s.full_rounds, s.partial_rounds, s.alpha, s.challenge_bits, s.hash_bits])


def width() -> int:
    return _current.width


def rate() -> int:
    return _current.width - 1


@lru_cache(maxsize=None)
def classic_constants(field: int, spec: RoSpec):
    """(round constants [R][t], MDS [t][t]) of the original Poseidon over `field` [UPSTREAM-RECALL: the paper's
    generate_parameters_grain]: an 80-bit Grain LFSR seeded with (field type 1 : 2 bits, S-box 0 : 4, n = 255 : 12, t : 12,
    R_F : 10, R_P : 10, thirty ones), taps b[i+80] = b[i+62] ^ b[i+51] ^ b[i+38] ^ b[i+23] ^ b[i+13] ^ b[i], the first 160
    bits discarded, then bits taken in pairs (first bit 1: keep the second); a constant = 255 bits, most significant first,
    rejected when not below the modulus.  MDS: Cauchy, M[i][j] = 1 / (i + t + j)."""
    m = o.modulus(field)
    t, rf, rp, n = spec.width, spec.full_rounds, spec.partial_rounds, 255
    bits = []
    for v, w in ((1, 2), (0, 4), (n, 12), (t, 12), (rf, 10), (rp, 10)):
        bits += [(v >> (w - 1 - k)) & 1 for k in range(w)]
    bits += [1] * 30
    assert len(bits) == 80
    st = bits

    def step():
        nonlocal st
        b = st[62] ^ st[51] ^ st[38] ^ st[23] ^ st[13] ^ st[0]
        st = st[1:] + [b]
        return b
    for _ in range(160):
        step()

    def next_bit():
        while True:
            a, b = step(), step()
            if a:
                return b

    def next_fe():
        while True:
            v = 0
            for _ in range(n):
                v = (v << 1) | next_bit()
            if v < m:
                return v
    rc = [[next_fe() for _ in range(t)] for _ in range(rf + rp)]
    mds = [[pow((i + t + j) % m, -1, m) for j in range(t)] for i in range(t)]
    return rc, mds


def _classic_permute(state: Sequence[int], field: int, spec: RoSpec) -> List[int]:
    m = o.modulus(field)
    rc, mds = classic_constants(field, spec)
    t, half = spec.width, spec.full_rounds // 2
    s = list(state)
    for r in range(spec.full_rounds + spec.partial_rounds):
        s = [(s[i] + rc[r][i]) % m for i in range(t)]
        if r < half or r >= half + spec.partial_rounds:
            s = [pow(x, 5, m) for x in s]
        else:
            s[0] = pow(s[0], 5, m)
        s = [sum(mds[i][j] * s[j] for j in range(t)) % m for i in range(t)]
    return s


T = 4
RATE = 3
RF = 8
RP = 56
M4 = ((5, 7, 1, 3), (4, 6, 1, 1), (1, 3, 5, 7), (1, 1, 4, 6))
MU = {o.FIELD_FP: (2, 3, 6, 8), o.FIELD_FQ: (2, 3, 6, 8)}      # diagonal of the internal matrix: the first
# increasing 4-subset of 2..11 that passes internal_matrix_ok in both fields
LABEL = b"vdf-poseidon2-v1"


@lru_cache(maxsize=None)
def round_constants(field: int) -> List[List[int]]:
    """RF/2 external rounds x 4, RP internal rounds x 1, RF/2 external rounds x 4: 64-byte SHAKE256 chunks mod m."""
    m = o.modulus(field)
    n = RF * T + RP
    raw = hashlib.shake_256(LABEL + bytes([field, T, RF, RP])).digest(64 * n)
    vals = [int.from_bytes(raw[64 * k:64 * k + 64], "little") % m for k in range(n)]
    out, k = [], 0
    for r in range(RF + RP):
        w = T if (r < RF // 2 or r >= RF // 2 + RP) else 1
        out.append(vals[k:k + w])
        k += w
    return out


def ext_layer(s: Sequence[int], m: int) -> List[int]:
    return [sum(M4[i][j] * s[j] for j in range(T)) % m for i in range(T)]


def int_layer(s: Sequence[int], field: int, m: int) -> List[int]:
    tot = sum(s)
    mu = MU[field]
    return [(tot + (mu[i] - 1) * s[i]) % m for i in range(T)]


def permute(state: Sequence[int], field: int) -> List[int]:
    if _current.family == 1:
        return _classic_permute(state, field, _current)
    m = o.modulus(field)
    rc = round_constants(field)
    s = ext_layer(state, m)
    for r in range(RF + RP):
        if r < RF // 2 or r >= RF // 2 + RP:
            s = [pow((s[i] + rc[r][i]) % m, 5, m) for i in range(T)]
            s = ext_layer(s, m)
        else:
            s = [pow((s[0] + rc[r][0]) % m, 5, m)] + list(s[1:])
            s = int_layer(s, field, m)
    return s


def hash_elements(tag: int, xs: Sequence[int], field: int) -> int:
    """Sponge: capacity lane 0 starts at tag + 2^32 * len(xs); the inputs are added to lanes 1..3 three at a time,
    one permutation per chunk; the output is lane 1 (a full field element; truncation is the caller's)."""
    m = o.modulus(field)
    w, r = width(), rate()                       # (4, 3) for the default block
    s = [(tag + (len(xs) << 32)) % m] + [0] * (w - 1)
    for k in range(0, len(xs), r):
        chunk = xs[k:k + r]
        for j, v in enumerate(chunk):
            s[1 + j] = (s[1 + j] + v) % m
        s = permute(s, field)
    return s[1]


# ---- the condition on the internal matrix ---------------------------------------------------------------------
def _matmul(a, b, m):
    n = len(a)
    return [[sum(a[i][k] * b[k][j] for k in range(n)) % m for j in range(n)] for i in range(n)]


def _charpoly(a, m):
    """Faddeev-LeVerrier: coefficients c[0..n] of det(xI - A), c[n] = 1."""
    n = len(a)
    c = [0] * (n + 1)
    c[n] = 1
    mk = [[0] * n for _ in range(n)]
    for k in range(1, n + 1):
        # M_k = A M_{k-1} + c_{n-k+1} I
        mk = _matmul(a, mk, m)
        for i in range(n):
            mk[i][i] = (mk[i][i] + c[n - k + 1]) % m
        amk = _matmul(a, mk, m)
        tr = sum(amk[i][i] for i in range(n)) % m
        c[n - k] = (-tr * pow(k, -1, m)) % m
    return c


def _polmulmod(a, b, f, m):
    n = len(f) - 1
    prod = [0] * (2 * n - 1)
    for i, x in enumerate(a):
        if x:
            for j, y in enumerate(b):
                prod[i + j] = (prod[i + j] + x * y) % m
    for d in range(2 * n - 2, n - 1, -1):           # f is monic
        q = prod[d]
        if q:
            for k in range(n + 1):
                prod[d - n + k] = (prod[d - n + k] - q * f[k]) % m
    return prod[:n]


def _polgcd_is_one(a, f, m):
    a, b = list(f), list(a)

    def trim(p):
        while p and p[-1] == 0:
            p.pop()
        return p
    a, b = trim(a), trim(b)
    while b:
        inv = pow(b[-1], -1, m)
        while len(a) >= len(b):
            q = a[-1] * inv % m
            sh = len(a) - len(b)
            for k in range(len(b)):
                a[sh + k] = (a[sh + k] - q * b[k]) % m
            trim(a)
            if not a:
                break
        a, b = b, a
    return len(a) == 1


def _irreducible_deg4(f, m):
    """Rabin: x^(m^4) = x mod f and gcd(x^(m^2) - x, f) = 1."""
    x = [0, 1, 0, 0]

    def frob(p):                                  # p(x) -> p(x)^m mod f
        res, base, e = [1, 0, 0, 0], p, m
        while e:
            if e & 1:
                res = _polmulmod(res, base, f, m)
            base = _polmulmod(base, base, f, m)
            e >>= 1
        return res
    x1 = frob(x)
    x2 = frob(x1)
    d = [(x2[i] - x[i]) % m for i in range(4)]
    if not any(d) or not _polgcd_is_one(d, f, m):
        return False
    x4 = frob(frob(x2))
    return x4 == x


def internal_matrix_ok(mu: Sequence[int], field: int) -> bool:
    m = o.modulus(field)
    mi = [[(mu[i] if i == j else 1) % m for j in range(T)] for i in range(T)]
    pw = mi
    for _ in range(2 * T):
        if not _irreducible_deg4(_charpoly(pw, m), m):
            return False
        pw = _matmul(mi, pw, m)
    return True
