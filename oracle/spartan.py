"""CPU oracle (TEST INFRASTRUCTURE ONLY) for the compression SNARK of the folding proof: a Spartan-style argument
for relaxed R1CS with inner-product-argument openings, the shape of nova-snark 0.8.0's `CompressedSNARK`
(`RelaxedR1CSSNARK` over `spartan_with_ipa_pc`) that /root/reference/src/nova/proof.rs:360-368 (`compress`) and
:383 (verification of the compressed proof) reach.  PARITY UNPINNED: nova-snark is a third-party crate absent from
/root/reference (Cargo.toml:15), the reference's only test of this path asserts `is_ok()` (src/nova/proof.rs:446-450),
and every constant below (transcript, challenge width, padding) is this build's own.  What pins it is mathematics:
completeness and soundness checks in tests/, and bit-exact agreement between this restatement and the product.

Protocol "vdf-spartan-v3" for an instance (comm_W, comm_E, u, X) of shape (A, B, C) with witness (W, E),
(A z) o (B z) = u (C z) + E,  z = (W, u, X):

  transcript   hash chain over SHAKE256: state' = H(state | label | ':' | data); a challenge is the first 16 bytes
               (little-endian, 128 bits) of H(state | label | '?'), the next 32 bytes become the state.
  tables       multilinear in MSB-first order: index i = sum x_j 2^(k-j); round j of a sum-check binds x_j,
               i.e. folds the upper half of every table onto the lower half.
  outer        tau in F^s (s = log2 of the padded constraint count);  0 = sum_x eq(tau,x) (Az(x) Bz(x) - u Cz(x) - E(x)),
               cubic rounds sent as g(0), g(2), g(3); ends with claims a, b, c, e at r_x.
  inner        rho;  a + rho b + rho^2 c = sum_y M(y) z(y),  M(y) = sum_x eq(r_x,x) (A + rho B + rho^2 C)[x,y], over the
               padded layout  y < 2^l : W (l = log2 of the padded variable count),  y = 2^l : u,  y = 2^l + 1 + i : X_i;
               quadratic rounds sent as g(0), g(2); ends with the claim w = W~(r_y[1:]).
  openings     two inner-product arguments under the Pedersen generators: <W, eq(r_y[1:])> = w against comm_W and
               <E, eq(r_x)> = e against comm_E.  L = <a_lo, G_hi> + <a_lo, b_hi> Q,  R = <a_hi, G_lo> + <a_hi, b_lo> Q,
               Q = x0 U;  a' = a_lo x + a_hi x^-1,  b' = b_lo x^-1 + b_hi x,  G' = G_lo x^-1 + G_hi x.
"""
from __future__ import annotations

import hashlib
from dataclasses import dataclass, field
from typing import Callable, List, Optional, Sequence, Tuple

from . import pasta as o

Point = Optional[Tuple[int, int]]


# ---- transcript -----------------------------------------------------------------------------------------
def _h(data: bytes, n: int) -> bytes:
    return hashlib.shake_256(data).digest(n)


class Transcript:
    def __init__(self, label: bytes):
        self.state = _h(b"vdf-spartan-v3|" + label, 32)

    def absorb(self, label: bytes, data: bytes) -> None:
        self.state = _h(self.state + label + b":" + data, 32)

    def absorb_fe(self, label: bytes, vals: Sequence[int]) -> None:
        self.absorb(label, b"".join(int(v).to_bytes(32, "little") for v in vals))

    def absorb_pt(self, label: bytes, pts: Sequence[Point]) -> None:
        self.absorb(label, b"".join(b"\0" * 64 if p is None else p[0].to_bytes(32, "little") + p[1].to_bytes(32, "little")
                                    for p in pts))

    def challenge(self, label: bytes) -> int:
        out = _h(self.state + label + b"?", 48)
        self.state = out[16:48]
        return int.from_bytes(out[:16], "little")


# ---- multilinear helpers --------------------------------------------------------------------------------
def eq_table(r: Sequence[int], m: int) -> List[int]:
    """eq(r, x) for all x in {0,1}^k, x_1 the most significant bit of the index."""
    t = [1]
    for rj in r:
        t = [v for e in t for v in (e * (1 - rj) % m, e * rj % m)]
    return t


def bind(table: Sequence[int], r: int, m: int) -> List[int]:
    h = len(table) // 2
    return [(table[i] + r * (table[h + i] - table[i])) % m for i in range(h)]


def mle_eval(table: Sequence[int], r: Sequence[int], m: int) -> int:
    t = list(table)
    for rj in r:
        t = bind(t, rj, m)
    return t[0]


def interpolate(evals_at: Sequence[Tuple[int, int]], r: int, m: int) -> int:
    """Lagrange interpolation of the points (x_i, y_i) at r."""
    acc = 0
    for i, (xi, yi) in enumerate(evals_at):
        num, den = 1, 1
        for j, (xj, _) in enumerate(evals_at):
            if i != j:
                num = num * (r - xj) % m
                den = den * (xi - xj) % m
        acc = (acc + yi * num * pow(den, -1, m)) % m
    return acc


def sumcheck_prove(tr: Transcript, label: bytes, tables: List[List[int]], comb: Callable[..., int], degree: int, m: int):
    """Returns (round messages [g(0), g(2), .., g(degree)], challenges, bound values of the tables)."""
    msgs, rs = [], []
    pts = [0] + list(range(2, degree + 1))
    while len(tables[0]) > 1:
        h = len(tables[0]) // 2
        ev = []
        for t in pts:
            s = 0
            for i in range(h):
                s += comb(*[(tb[i] + t * (tb[h + i] - tb[i])) % m for tb in tables])
            ev.append(s % m)
        tr.absorb_fe(label, ev)
        r = tr.challenge(label)
        tables = [bind(tb, r, m) for tb in tables]
        msgs.append(ev)
        rs.append(r)
    return msgs, rs, [tb[0] for tb in tables]


def sumcheck_verify(tr: Transcript, label: bytes, claim: int, msgs: Sequence[Sequence[int]], degree: int, m: int):
    """Returns (final claim, challenges); the caller checks the final claim against the bound polynomial."""
    rs = []
    for ev in msgs:
        if len(ev) != degree:
            raise ValueError("malformed sum-check message")
        g1 = (claim - ev[0]) % m
        pts = [(0, ev[0]), (1, g1)] + [(t, ev[k + 1]) for k, t in enumerate(range(2, degree + 1))]
        tr.absorb_fe(label, ev)
        r = tr.challenge(label)
        claim = interpolate(pts, r, m)
        rs.append(r)
    return claim, rs


# ---- inner-product argument -----------------------------------------------------------------------------
IPA_STOP = 16      # the halving stops at this length: the prover sends the remaining vector instead of four more rounds


@dataclass
class IpaProof:
    L: List[Point] = field(default_factory=list)
    R: List[Point] = field(default_factory=list)
    a: List[int] = field(default_factory=list)     # the folded vector, min(n, IPA_STOP) elements


class Gens:
    """A generator table held as the C restatement's array (uint64[n, 8], Montgomery affine): sliceable, and its MSMs
    run in oracle/pasta_ref.c -- the augmented circuits have ~10^4 variables, far beyond msm_naive."""

    def __init__(self, arr, curve: int):
        self.arr, self.curve = arr, curve

    def __len__(self) -> int:
        return self.arr.shape[0]

    def __getitem__(self, sl) -> "Gens":
        assert isinstance(sl, slice)
        return Gens(self.arr[sl], self.curve)

    def msm(self, scalars: Sequence[int]) -> Point:
        import numpy as np
        from . import cref
        L = cref.lib()
        n = len(scalars)
        assert n == len(self)
        pts = np.ascontiguousarray(self.arr)
        sc = np.frombuffer(b"".join(int(x).to_bytes(32, "little") for x in scalars), dtype="<u8").reshape(n, 4).copy()
        out, aff = np.zeros(12, dtype="<u8"), np.zeros(8, dtype="<u8")
        L.ref_msm(self.curve, cref.p(pts), cref.p(sc), n, 0, 4, 0, cref.p(out))
        L.ref_jac_to_affine(self.curve, cref.p(out), cref.p(aff))
        bm = o.curve_base_modulus(self.curve)
        raw = aff.tobytes()
        x, y = o.from_mont(int.from_bytes(raw[:32], "little"), bm), o.from_mont(int.from_bytes(raw[32:], "little"), bm)
        return None if (x, y) == (0, 0) else (x, y)


def _msm(s: Sequence[int], G, curve: int) -> Point:
    if isinstance(G, Gens):
        return G.msm(s)
    return o.msm_naive(list(s), list(G), curve)


class _IpaProver:
    """One inner-product argument as a state machine, so that several can advance in lockstep: a round of all of them
    is one batch of MSMs for the product (ipa_prove_many).  The folded generators G' = G_lo x^-1 + G_hi x are never
    formed: s[t] is the coefficient of original generator t in its folded generator G'_(t mod size), and a round's
    <a_lo, G'_hi>, <a_hi, G'_lo> are MSMs over the ORIGINAL generators with scalars s[t] a[...] (as the product does it)."""

    def __init__(self, tr, label, G, U, a, b, v, P, curve):
        self.q, self.pm, self.curve, self.label = o.curve_scalar_modulus(curve), o.curve_base_modulus(curve), curve, label
        tr.absorb_pt(label, [P]); tr.absorb_fe(label, [v])
        self.Q = o.pt_mul(tr.challenge(label), U, self.pm)
        self.a, self.b, self.G = list(a), list(b), G
        self.n = len(self.a)
        self.s = [1] * self.n
        self.proof = IpaProof()

    def active(self) -> bool:
        return len(self.a) > IPA_STOP

    def round_points(self) -> None:
        a, b, q, pm, s = self.a, self.b, self.q, self.pm, self.s
        size = len(a)
        h = size // 2
        cL = sum(x * y for x, y in zip(a[:h], b[h:])) % q
        cR = sum(x * y for x, y in zip(a[h:], b[:h])) % q
        sL = [s[t] * a[(t % size) - h] % q if (t % size) >= h else 0 for t in range(self.n)]
        sR = [s[t] * a[(t % size) + h] % q if (t % size) < h else 0 for t in range(self.n)]
        self.L = o.pt_add(_msm(sL, self.G, self.curve), o.pt_mul(cL, self.Q, pm), pm)
        self.R = o.pt_add(_msm(sR, self.G, self.curve), o.pt_mul(cR, self.Q, pm), pm)

    def round_fold(self, tr) -> None:
        a, b, q = self.a, self.b, self.q
        size = len(a)
        h = size // 2
        tr.absorb_pt(self.label, [self.L, self.R])
        x = tr.challenge(self.label)
        xi = pow(x, -1, q)
        self.a = [(a[i] * x + a[h + i] * xi) % q for i in range(h)]
        self.b = [(b[i] * xi + b[h + i] * x) % q for i in range(h)]
        self.s = [self.s[t] * (x if (t % size) >= h else xi) % q for t in range(self.n)]
        self.proof.L.append(self.L); self.proof.R.append(self.R)

    def finish(self) -> IpaProof:
        self.proof.a = list(self.a)
        return self.proof


def ipa_prove_many(tr: Transcript, jobs, curve: int) -> List[IpaProof]:
    """jobs: (label, G, U, a, b, v, P) each.  Statements and values are absorbed job by job; then the arguments advance
    in lockstep: every round, all still-active jobs compute their (L, R) -- before any challenge of that round is drawn --
    and then, job by job, absorb them, draw their challenge and fold.  A shorter vector simply finishes earlier."""
    ps = [_IpaProver(tr, *job, curve) for job in jobs]
    while any(p.active() for p in ps):
        act = [p for p in ps if p.active()]
        for p in act:
            p.round_points()
        for p in act:
            p.round_fold(tr)
    return [p.finish() for p in ps]


def ipa_prove(tr: Transcript, label: bytes, G: Sequence[Point], U: Point, a: Sequence[int], b: Sequence[int], v: int,
              P: Point, curve: int) -> IpaProof:
    return ipa_prove_many(tr, [(label, G, U, a, b, v, P)], curve)[0]


class _IpaVerifier:
    def __init__(self, tr, label, G, U, b, v, P, proof, curve):
        self.q, self.pm, self.curve, self.label = o.curve_scalar_modulus(curve), o.curve_base_modulus(curve), curve, label
        self.G, self.proof, self.n = G, proof, len(G)
        self.m = min(self.n, IPA_STOP)
        self.ok = (self.m << len(proof.L)) == self.n and len(proof.L) == len(proof.R) and len(proof.a) == self.m
        tr.absorb_pt(label, [P]); tr.absorb_fe(label, [v])
        self.Q = o.pt_mul(tr.challenge(label), U, self.pm)
        self.acc = o.pt_add(P, o.pt_mul(v, self.Q, self.pm), self.pm)
        self.s = [1] * self.n             # G'_i = sum over t = i (mod m) of s_t G_t
        self.b = list(b)
        self.size, self.idx = self.n, 0

    def active(self) -> bool:
        return self.ok and self.idx < len(self.proof.L)

    def round(self, tr) -> None:
        q, pm = self.q, self.pm
        L, R = self.proof.L[self.idx], self.proof.R[self.idx]
        tr.absorb_pt(self.label, [L, R])
        x = tr.challenge(self.label)
        if x == 0:
            self.ok = False
            return
        xi = pow(x, -1, q)
        self.acc = o.pt_add(self.acc, o.pt_add(o.pt_mul(x * x % q, L, pm), o.pt_mul(xi * xi % q, R, pm), pm), pm)
        h = self.size // 2
        for t in range(self.n):
            self.s[t] = self.s[t] * (x if (t % self.size) >= h else xi) % q
        self.b = [(self.b[i] * xi + self.b[h + i] * x) % q for i in range(h)]
        self.size, self.idx = h, self.idx + 1

    def final(self) -> bool:
        if not self.ok:
            return False
        q, pm, m = self.q, self.pm, self.m
        ab = sum(x * y for x, y in zip(self.proof.a, self.b)) % q
        rhs = o.pt_add(_msm([self.s[t] * self.proof.a[t % m] % q for t in range(self.n)], self.G, self.curve),
                       o.pt_mul(ab, self.Q, pm), pm)
        return self.acc == rhs


def ipa_verify_many(tr: Transcript, jobs, curve: int) -> bool:
    """jobs: (label, G, U, b, v, P, proof) each; the transcript order of ipa_prove_many."""
    vs = [_IpaVerifier(tr, *job, curve) for job in jobs]
    if not all(v.ok for v in vs):
        return False
    while any(v.active() for v in vs):
        for v in [v for v in vs if v.active()]:
            v.round(tr)
    return all(v.final() for v in vs)


def ipa_verify(tr: Transcript, label: bytes, G: Sequence[Point], U: Point, b: Sequence[int], v: int, P: Point,
               proof: IpaProof, curve: int) -> bool:
    return ipa_verify_many(tr, [(label, G, U, b, v, P, proof)], curve)


# ---- the SNARK ---------------------------------------------------------------------------------------------
def _pow2_at_least(n: int) -> int:
    p = 1
    while p < n:
        p <<= 1
    return p


@dataclass
class SpartanProof:
    outer: List[List[int]]
    claims: Tuple[int, int, int, int]        # Az, Bz, Cz, E at r_x
    inner: List[List[int]]
    w_eval: int
    ipa_W: IpaProof
    ipa_E: IpaProof


def _layout(shape: o.R1CSShape):
    M = _pow2_at_least(shape.num_cons)
    NW = _pow2_at_least(shape.num_vars)
    return M, NW, 2 * NW


def _col(shape: o.R1CSShape, NW: int, c: int) -> int:
    return c if c < shape.num_vars else NW + (c - shape.num_vars)


def _instance_bytes(tr: Transcript, digest: bytes, comm_W: Point, comm_E: Point, u: int, X: Sequence[int]) -> None:
    tr.absorb(b"shape", digest)
    tr.absorb_pt(b"inst", [comm_W, comm_E])
    tr.absorb_fe(b"inst", [u] + list(X))


def m_vector(shape: o.R1CSShape, eq_rx: Sequence[int], rho: int, NW: int, q: int) -> List[int]:
    out = [0] * (2 * NW)
    for coef, mat in ((1, shape.A), (rho, shape.B), (rho * rho % q, shape.C)):
        for r, c, v in mat:
            out[_col(shape, NW, c)] = (out[_col(shape, NW, c)] + coef * v % q * eq_rx[r]) % q
    return out


def prove(shape: o.R1CSShape, digest: bytes, G: Sequence[Point], U: Point, comm_W: Point, comm_E: Point, u: int,
          X: Sequence[int], W: Sequence[int], E: Sequence[int], curve: int = o.CURVE_PALLAS) -> SpartanProof:
    q = o.curve_scalar_modulus(curve)
    M, NW, Z = _layout(shape)
    s, l1 = M.bit_length() - 1, Z.bit_length() - 1
    tr = Transcript(b"compress")
    _instance_bytes(tr, digest, comm_W, comm_E, u, X)
    z = list(W) + [u] + list(X)
    az, bz, cz = o.multiply_vec(shape, z, q)
    pad = lambda v, n: list(v) + [0] * (n - len(v))
    tau = [tr.challenge(b"tau") for _ in range(s)]
    tables = [eq_table(tau, q), pad(az, M), pad(bz, M), pad(cz, M), pad(E, M)]
    outer, rx, fin = sumcheck_prove(tr, b"outer", tables, lambda e, a, b, c, d: e * ((a * b - u * c - d) % q), 3, q)
    claims = (fin[1], fin[2], fin[3], fin[4])
    tr.absorb_fe(b"claims", claims)
    rho = tr.challenge(b"rho")
    eq_rx = eq_table(rx, q)
    zpad = pad(W, NW) + pad([u] + list(X), NW)
    inner, ry, _ = sumcheck_prove(tr, b"inner", [m_vector(shape, eq_rx, rho, NW, q), zpad], lambda a, b: a * b, 2, q)
    eq_ry = eq_table(ry[1:], q)
    w_eval = sum(x * y for x, y in zip(pad(W, NW), eq_ry)) % q
    tr.absorb_fe(b"weval", [w_eval])
    ipa_W, ipa_E = ipa_prove_many(tr, [(b"ipaW", G[:NW], U, pad(W, NW), eq_ry, w_eval, comm_W),
                                       (b"ipaE", G[:M], U, pad(E, M), eq_rx, claims[3], comm_E)], curve)
    return SpartanProof(outer, claims, inner, w_eval, ipa_W, ipa_E)


def verify(shape: o.R1CSShape, digest: bytes, G: Sequence[Point], U: Point, comm_W: Point, comm_E: Point, u: int,
           X: Sequence[int], proof: SpartanProof, curve: int = o.CURVE_PALLAS) -> bool:
    q = o.curve_scalar_modulus(curve)
    M, NW, Z = _layout(shape)
    s, l1 = M.bit_length() - 1, Z.bit_length() - 1
    if len(proof.outer) != s or len(proof.inner) != l1:
        return False
    tr = Transcript(b"compress")
    _instance_bytes(tr, digest, comm_W, comm_E, u, X)
    tau = [tr.challenge(b"tau") for _ in range(s)]
    try:
        claim, rx = sumcheck_verify(tr, b"outer", 0, proof.outer, 3, q)
    except ValueError:
        return False
    a, b, c, e = proof.claims
    if claim != mle_eval(eq_table(tau, q), rx, q) * ((a * b - u * c - e) % q) % q:
        return False
    tr.absorb_fe(b"claims", proof.claims)
    rho = tr.challenge(b"rho")
    try:
        claim, ry = sumcheck_verify(tr, b"inner", (a + rho * b + rho * rho % q * c) % q, proof.inner, 2, q)
    except ValueError:
        return False
    eq_rx = eq_table(rx, q)
    m_ry = mle_eval(m_vector(shape, eq_rx, rho, NW, q), ry, q)
    eq_rest = eq_table(ry[1:], q)
    pub = sum(v * eq_rest[i] for i, v in enumerate([u] + list(X))) % q
    z_ry = ((1 - ry[0]) * proof.w_eval + ry[0] * pub) % q
    if claim != m_ry * z_ry % q:
        return False
    tr.absorb_fe(b"weval", [proof.w_eval])
    return ipa_verify_many(tr, [(b"ipaW", G[:NW], U, eq_rest, proof.w_eval, comm_W, proof.ipa_W),
                                (b"ipaE", G[:M], U, eq_rx, e, comm_E, proof.ipa_E)], curve)
