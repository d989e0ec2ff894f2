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

from . import pasta as o

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
    s = [(tag + (len(xs) << 32)) % m, 0, 0, 0]
    for k in range(0, len(xs), RATE):
        chunk = xs[k:k + RATE]
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
