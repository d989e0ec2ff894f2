"""TEST INFRASTRUCTURE -- never imported by the product (vdf_amd/).

CPU restatement of the Nova IVC layer of libvdf_nova.so, protocol "vdf-nova-ivc-v1": what the reference reaches
through nova-snark 0.8.0 (Cargo.toml:15) at src/nova/proof.rs:232-237 (PublicParams::setup), :342-349
(RecursiveSNARK::prove_step on the Pallas/Vesta cycle, primary circuit = the MinRoot step circuit, secondary =
TrivialTestCircuit, :26-43, :258-260) and :370-392 (verify, zi_secondary == [0]).

nova-snark's source is not in /root/reference and the reference's tests hold no known-answer value for anything in it
(SURVEY.md 8c), so every implementation-defined choice below -- the random oracle (oracle/poseidon.py), the order of
allocations in the augmented circuit, the non-native arithmetic, the transcript layout, the shape digest -- is this
build's own: PARITY UNPINNED against nova-snark; the product is pinned bit-for-bit against THIS file.  What is taken
from the published scheme (Kothapalli, Setty, Tzialla, "Nova", 2021) is the structure: relaxed
R1CS, NIFS folding with a 128-bit challenge, an augmented circuit per curve that checks the previous output hash,
folds the other curve's instance in-circuit, runs the step circuit and outputs the next hash (250-bit truncation so
that a hash fits both fields), and a verifier that recomputes the two hashes and checks three satisfiability claims.

Sides: side 0 = primary (commitments on Pallas, circuit over Fq), side 1 = secondary (Vesta, circuit over Fp).  The
circuit of side s folds instances of side 1 - s, whose commitment coordinates are native to it.

Plain big-integer Python; commitments go through an injected `commit(side, vector)` (tests use the C restatement's MSM).
"""
from __future__ import annotations

import hashlib
from dataclasses import dataclass, field as dc_field
from typing import Callable, Dict, List, Optional, Sequence, Tuple

from . import pasta as o
from . import poseidon as ps

ONE = -1                      # the constant column (u of a relaxed instance)
NUM_IO = 2                    # public IO of an augmented circuit: X[0] = the other side's X[1], X[1] = the output hash
HASH_BITS = 250               # a hash that travels between the fields
CHAL_BITS = 128               # fold challenge
LIMB = 126                    # X of a running instance is carried as (low 126 bits, the rest)
TAG_STATE, TAG_CHAL = 1, 2    # domain tags of the two hashes
B_CURVE = 5

SIDE_FIELD = (o.FIELD_FQ, o.FIELD_FP)            # circuit field of side s = scalar field of its curve
SIDE_CURVE = (o.CURVE_PALLAS, o.CURVE_VESTA)


def io_var(k: int) -> int:
    return -(2 + k)


# ---------------------------------------------------------------------------------------------------------------
# Constraint system: R1CS over z = (W, u, X); a Num carries a linear combination and its value
# ---------------------------------------------------------------------------------------------------------------
class Num:
    __slots__ = ("lc", "v")

    def __init__(self, lc: Dict[int, int], v: int):
        self.lc, self.v = lc, v


class CS:
    def __init__(self, field: int):
        self.field = field
        self.m = o.modulus(field)
        self.W: List[int] = []
        self.X: List[int] = []
        self.A: List[Tuple[int, int, int]] = []
        self.B: List[Tuple[int, int, int]] = []
        self.C: List[Tuple[int, int, int]] = []
        self.rows = 0

    # -- linear combinations (no constraints) --
    def const(self, k: int) -> Num:
        k %= self.m
        return Num({ONE: k} if k else {}, k)

    def add(self, a: Num, b: Num) -> Num:
        lc = dict(a.lc)
        for var, c in b.lc.items():
            s = (lc.get(var, 0) + c) % self.m
            if s:
                lc[var] = s
            else:
                lc.pop(var, None)
        return Num(lc, (a.v + b.v) % self.m)

    def scale(self, a: Num, k: int) -> Num:
        k %= self.m
        if k == 0:
            return Num({}, 0)
        return Num({var: c * k % self.m for var, c in a.lc.items()}, a.v * k % self.m)

    def sub(self, a: Num, b: Num) -> Num:
        return self.add(a, self.scale(b, -1))

    def lin(self, terms: Sequence[Tuple[int, Num]]) -> Num:
        acc = Num({}, 0)
        for k, n in terms:
            acc = self.add(acc, self.scale(n, k))
        return acc

    # -- variables and constraints --
    def alloc(self, v: int) -> Num:
        self.W.append(v % self.m)
        return Num({len(self.W) - 1: 1}, v % self.m)

    def alloc_io(self, v: int) -> Num:
        self.X.append(v % self.m)
        return Num({io_var(len(self.X) - 1): 1}, v % self.m)

    def enforce(self, a: Num, b: Num, c: Num) -> None:
        r = self.rows
        for mat, n in ((self.A, a), (self.B, b), (self.C, c)):
            for var in sorted(n.lc, key=lambda x: (x < 0, x if x >= 0 else -x)):      # W ascending, then ONE, then IO
                mat.append((r, var, n.lc[var]))
        self.rows += 1

    def mul(self, a: Num, b: Num) -> Num:
        c = self.alloc(a.v * b.v)
        self.enforce(a, b, c)
        return c

    def enforce_equal(self, a: Num, b: Num) -> None:
        self.enforce(self.sub(a, b), self.const(1), Num({}, 0))

    def shape(self) -> o.R1CSShape:
        nv = len(self.W)

        def col(var):
            return var if var >= 0 else (nv if var == ONE else nv + 1 + (-var - 2))
        conv = lambda mat: [(r, col(var), c) for r, var, c in mat]
        return o.R1CSShape(self.rows, nv, len(self.X), conv(self.A), conv(self.B), conv(self.C))


# ---------------------------------------------------------------------------------------------------------------
# Gadgets.  Allocation and constraint ORDER is part of the specification (the product must produce the same W).
# ---------------------------------------------------------------------------------------------------------------
def inv_or_zero(v: int, m: int) -> int:
    return pow(v, -1, m) if v % m else 0


def is_zero(cs: CS, a: Num) -> Num:
    """z = 1 iff a = 0.  alloc z, inv;  a * inv = 1 - z;  a * z = 0."""
    z = cs.alloc(1 if a.v == 0 else 0)
    inv = cs.alloc(inv_or_zero(a.v, cs.m))
    cs.enforce(a, inv, cs.sub(cs.const(1), z))
    cs.enforce(a, z, Num({}, 0))
    return z


def select(cs: CS, cond: Num, a: Num, b: Num) -> Num:
    """cond ? a : b = b + cond (a - b): one constraint."""
    out = cs.alloc(a.v if cond.v else b.v)
    cs.enforce(cond, cs.sub(a, b), cs.sub(out, b))
    return out


def alloc_bits(cs: CS, v: int, n: int) -> List[Num]:
    """n boolean variables, least significant first (one constraint each)."""
    bits = []
    for k in range(n):
        b = cs.alloc((v >> k) & 1)
        cs.enforce(b, cs.sub(cs.const(1), b), Num({}, 0))
        bits.append(b)
    return bits


def pack(cs: CS, bits: Sequence[Num], shift: int = 0) -> Num:
    return cs.lin([(1 << (shift + k), b) for k, b in enumerate(bits)])


def strict_bits(cs: CS, a: Num) -> List[Num]:
    """The 255 bits of the CANONICAL representative of a (m = 2^254 + c, c < 2^126): value < m iff bit 254 is clear,
    or bits 126..253 are clear and the low 126 bits are below c."""
    m = cs.m
    c = m - (1 << 254)
    assert 0 < c < (1 << 126)
    bits = alloc_bits(cs, a.v, 255)
    cs.enforce_equal(pack(cs, bits), a)
    mid = cs.lin([(1, b) for b in bits[126:254]])
    mz = is_zero(cs, mid)
    low = pack(cs, bits[:126])
    v = cs.add(low, cs.const((1 << 126) - c))                 # < 2^127; bit 126 clear iff low < c
    d = alloc_bits(cs, v.v, 127)
    cs.enforce_equal(pack(cs, d), v)
    ok = cs.mul(mz, cs.sub(cs.const(1), d[126]))
    cs.enforce(bits[254], cs.sub(cs.const(1), ok), Num({}, 0))
    return bits


def poseidon_permute_classic(cs: CS, s: List[Num]) -> List[Num]:
    """The original Poseidon permutation in R1CS (ps.RoSpec family 1): add the round constants, x^5 on every lane (full rounds)
    or on lane 0 (partial rounds: the other lanes stay linear combinations), then the dense MDS matrix as linear combinations."""
    spec = ps.current()
    rc, mds = ps.classic_constants(cs.field, spec)
    t, half = spec.width, spec.full_rounds // 2

    def sbox(x):
        x2 = cs.mul(x, x)
        x4 = cs.mul(x2, x2)
        return cs.mul(x4, x)
    for r in range(spec.full_rounds + spec.partial_rounds):
        s = [cs.add(s[i], cs.const(rc[r][i])) for i in range(t)]
        if r < half or r >= half + spec.partial_rounds:
            s = [sbox(x) for x in s]
        else:
            s = [sbox(s[0])] + s[1:]
        s = [cs.lin([(mds[i][j], s[j]) for j in range(t)]) for i in range(t)]
    return s


def poseidon_permute(cs: CS, s: List[Num]) -> List[Num]:
    if ps.current().family == 1:
        return poseidon_permute_classic(cs, s)
    f, m = cs.field, cs.m
    rc = ps.round_constants(f)
    mu = ps.MU[f]

    def ext(st):
        return [cs.lin([(ps.M4[i][j], st[j]) for j in range(ps.T)]) for i in range(ps.T)]

    def sbox(x):
        x2 = cs.mul(x, x)
        x4 = cs.mul(x2, x2)
        return cs.mul(x4, x)
    s = ext(s)
    for r in range(ps.RF + ps.RP):
        if r < ps.RF // 2 or r >= ps.RF // 2 + ps.RP:
            s = [sbox(cs.add(s[i], cs.const(rc[r][i]))) for i in range(ps.T)]
            s = ext(s)
        else:
            s = [sbox(cs.add(s[0], cs.const(rc[r][0])))] + s[1:]
            tot = cs.lin([(1, x) for x in s])
            s = [cs.add(tot, cs.scale(s[i], mu[i] - 1)) for i in range(ps.T)]
    return s


def poseidon_hash(cs: CS, tag: int, xs: Sequence[Num]) -> Num:
    s = [cs.const(tag + (len(xs) << 32))] + [cs.const(0)] * (ps.width() - 1)
    for k in range(0, len(xs), ps.rate()):
        for j, x in enumerate(xs[k:k + ps.rate()]):
            s[1 + j] = cs.add(s[1 + j], x)
        s = poseidon_permute(cs, s)
    return s[1]


# ---- curve y^2 = x^3 + 5 in affine coordinates; the identity is (0, 0) (x = 0 is on neither curve) --------------
def check_on_curve(cs: CS, x: Num, y: Num, inf: Num) -> None:
    """(1 - inf) (y^2 - x^3 - 5) = 0."""
    x2 = cs.mul(x, x)
    x3 = cs.mul(x2, x)
    y2 = cs.mul(y, y)
    cs.enforce(cs.sub(cs.const(1), inf), cs.sub(y2, cs.add(x3, cs.const(B_CURVE))), Num({}, 0))


def ec_double_raw(cs: CS, x: Num, y: Num) -> Tuple[Num, Num]:
    """Doubling of a point that is not the identity; for (0, 0) every value is 0 (lambda = 0 when 2y = 0)."""
    m = cs.m
    x2 = cs.mul(x, x)
    lam = cs.alloc(3 * x2.v * inv_or_zero(2 * y.v, m))
    cs.enforce(lam, cs.scale(y, 2), cs.scale(x2, 3))
    dx = cs.alloc(lam.v * lam.v - 2 * x.v)
    cs.enforce(lam, lam, cs.add(dx, cs.scale(x, 2)))
    dy = cs.alloc(lam.v * (x.v - dx.v) - y.v)
    cs.enforce(lam, cs.sub(x, dx), cs.add(dy, y))
    return dx, dy


def ec_add_raw(cs: CS, x1: Num, y1: Num, x2: Num, y2: Num) -> Tuple[Num, Num]:
    """Chord addition, valid when x1 != x2; lambda = 0 when x1 = x2."""
    m = cs.m
    lam = cs.alloc((y2.v - y1.v) * inv_or_zero(x2.v - x1.v, m))
    cs.enforce(lam, cs.sub(x2, x1), cs.sub(y2, y1))
    sx = cs.alloc(lam.v * lam.v - x1.v - x2.v)
    cs.enforce(lam, lam, cs.lin([(1, sx), (1, x1), (1, x2)]))
    sy = cs.alloc(lam.v * (x1.v - sx.v) - y1.v)
    cs.enforce(lam, cs.sub(x1, sx), cs.add(sy, y1))
    return sx, sy


def ec_scalar_mul(cs: CS, bits: Sequence[Num], px: Num, py: Num, p_inf: Num) -> Tuple[Num, Num]:
    """[sum bits 2^k] P, least significant bit first: acc += 2^k P when bit k is set.  P is on the curve or the identity
    (checked by the caller) and the group has prime order > 2^254, so with fewer than 254 bits acc = (r mod 2^k) P is
    never +-2^k P: the chord formula is complete here except for acc = identity, which a flag tracks."""
    ax, ay = cs.const(0), cs.const(0)
    acc_inf = cs.const(1)
    wx, wy = px, py
    for k, b in enumerate(bits):
        sx, sy = ec_add_raw(cs, ax, ay, wx, wy)
        cx = select(cs, acc_inf, wx, sx)
        cy = select(cs, acc_inf, wy, sy)
        ax = select(cs, b, cx, ax)
        ay = select(cs, b, cy, ay)
        acc_inf = cs.mul(acc_inf, cs.sub(cs.const(1), b))
        if k + 1 < len(bits):
            wx, wy = ec_double_raw(cs, wx, wy)
    keep = cs.sub(cs.const(1), p_inf)                          # the identity times anything is the identity
    return cs.mul(keep, ax), cs.mul(keep, ay)


def ec_add_complete(cs: CS, x1: Num, y1: Num, x2: Num, y2: Num) -> Tuple[Num, Num]:
    m = cs.m
    i1 = is_zero(cs, x1)
    i2 = is_zero(cs, x2)
    same_x = is_zero(cs, cs.sub(x2, x1))
    same_y = is_zero(cs, cs.sub(y2, y1))
    x1sq = cs.mul(x1, x1)
    num = select(cs, same_x, cs.scale(x1sq, 3), cs.sub(y2, y1))
    den = select(cs, same_x, cs.scale(y1, 2), cs.sub(x2, x1))
    lam = cs.alloc(num.v * inv_or_zero(den.v, m))
    cs.enforce(lam, den, num)
    x3 = cs.alloc(lam.v * lam.v - x1.v - x2.v)
    cs.enforce(lam, lam, cs.lin([(1, x3), (1, x1), (1, x2)]))
    y3 = cs.alloc(lam.v * (x1.v - x3.v) - y1.v)
    cs.enforce(lam, cs.sub(x1, x3), cs.add(y3, y1))
    is_neg = cs.mul(same_x, cs.sub(cs.const(1), same_y))       # P + (-P)
    keep = cs.sub(cs.const(1), is_neg)
    tx, ty = cs.mul(keep, x3), cs.mul(keep, y3)
    ux, uy = select(cs, i2, x1, tx), select(cs, i2, y1, ty)
    return select(cs, i1, x2, ux), select(cs, i1, y2, uy)


# ---- non-native fold of one public-IO element:  R = A + r B  mod p'  (p' = the other field's modulus) ----------
def fold_foreign(cs: CS, a_lo: Num, a_hi: Num, b_bits: Sequence[Num], r_bits: Sequence[Num], pf: int) -> Tuple[Num, Num]:
    """A = a_lo + 2^126 a_hi (trusted ranges: bound by the previous output hash), B = 250 bits, r = 128 bits.
    The prover supplies the quotient k (125 bits) and R (126 + 129 bits); the integer identity A + r B = k p' + R is
    checked modulo the native modulus and modulo 2^126 -- both sides are below 2^379.1 < native * 2^126."""
    m = cs.m
    D = 1 << LIMB
    b_lo, b_all = pack(cs, b_bits[:LIMB]), pack(cs, b_bits)
    r_lo, r_all = pack(cs, r_bits[:LIMB]), pack(cs, r_bits)
    A = a_lo.v + D * a_hi.v
    tot = A + r_all.v * b_all.v
    kq, R = divmod(tot, pf)
    assert kq < (1 << 125) and a_lo.v < D
    k_bits = alloc_bits(cs, kq, 125)
    rlo_bits = alloc_bits(cs, R % D, LIMB)
    rhi_bits = alloc_bits(cs, R >> LIMB, 129)
    k = pack(cs, k_bits)
    R_lo, R_hi = pack(cs, rlo_bits), pack(cs, rhi_bits)
    # (1) modulo the native field
    rhs = cs.lin([(pf % m, k), (1, R_lo), (D, R_hi), (-1, a_lo), (-D, a_hi)])
    cs.enforce(r_all, b_all, rhs)
    # (2) modulo 2^126: a_lo + r_lo b_lo - k (p' mod D) - R_lo = (c' - 2^127) D
    prod = cs.mul(r_lo, b_lo)
    low = a_lo.v + (r_lo.v * b_lo.v) - kq * (pf % D) - (R % D)
    assert low % D == 0
    cprime = low // D + (1 << 127)
    assert 0 <= cprime < (1 << 128)
    c_bits = alloc_bits(cs, cprime, 128)
    lhs = cs.lin([(1, a_lo), (1, prod), (-(pf % D), k), (-1, R_lo)])
    cs.enforce_equal(lhs, cs.scale(cs.sub(pack(cs, c_bits), cs.const(1 << 127)), D))
    return R_lo, R_hi


# ---------------------------------------------------------------------------------------------------------------
# Step circuits (the seam of src/nova/proof.rs:79-153: arity / synthesize / output)
# ---------------------------------------------------------------------------------------------------------------
class InverseMinRootCircuit:
    """src/nova/proof.rs:57-230.  `bound` = False restates the reference's circuit exactly (4 aux per round, new_x
    allocated at :167-173 but bound by no constraint -- the third constraint uses y - i + 1 directly, :219-227);
    `bound` = True is the sound variant the product offers as an option (VDF_CIRCUIT_MINROOT_BOUND): new_x is not a variable at all, the next round's x
    is the linear combination y - i + 1 itself (3 aux per round, the same three constraints)."""

    def __init__(self, t: int, result: Optional[o.State], inp: Optional[o.State], bound: bool = False):
        self.t, self.result, self.input, self.bound = t, result, inp, bound

    def arity(self) -> int:
        return 3

    def synthesize(self, cs: CS, z: Sequence[Num]) -> List[Num]:
        m = cs.m
        x, y, i_in = z                                        # z[2] is the counter; i of round j = i_in - j (:162-164)
        for j in range(self.t):
            new_x_lc = cs.lin([(1, y), (-1, i_in), (j + 1, cs.const(1))])       # y - (i - 1)
            if not self.bound:
                new_x = cs.alloc(new_x_lc.v)                  # :167-173 (unconstrained in the reference)
            tmp1 = cs.mul(x, x)                               # :176
            tmp2 = cs.mul(tmp1, tmp1)                         # :178
            new_y = cs.alloc(tmp2.v * x.v - new_x_lc.v)       # :181-189
            cs.enforce(tmp2, x, cs.add(new_y, new_x_lc))      # :219-227: tmp2 * x = new_y + y - i + 1
            x, y = (new_x_lc if self.bound else new_x), new_y
        final_i = cs.alloc(i_in.v - self.t)                   # :122-133
        cs.enforce(final_i, cs.const(1), cs.sub(i_in, cs.const(self.t)))
        return [x, y, final_i]

    def output(self, z: Sequence[int]) -> List[int]:          # :142-152
        if self.result is not None:
            assert list(z) == [self.result.x, self.result.y, self.result.i]
        return [self.input.x, self.input.y, self.input.i]


class CubicCircuit:
    """A second primary step circuit for the seam's tests (nova-snark's own example): arity 1, z -> z^3 + z + 5."""

    def arity(self) -> int:
        return 1

    def synthesize(self, cs: CS, z: Sequence[Num]) -> List[Num]:
        x = z[0]
        x2 = cs.mul(x, x)
        x3 = cs.mul(x2, x)
        rhs = cs.add(cs.add(x3, x), cs.const(5))
        y = cs.alloc(rhs.v)
        cs.enforce(rhs, cs.const(1), y)
        return [y]

    def output(self, z: Sequence[int]) -> List[int]:
        return [(z[0] ** 3 + z[0] + 5) % o.Q]


class TrivialTestCircuit:
    """nova-snark's TrivialTestCircuit (src/nova/proof.rs:258-260): arity 1, z_out = z_in, no constraint."""

    def arity(self) -> int:
        return 1

    def synthesize(self, cs: CS, z: Sequence[Num]) -> List[Num]:
        return list(z)

    def output(self, z: Sequence[int]) -> List[int]:
        return list(z)


# ---------------------------------------------------------------------------------------------------------------
# Instances
# ---------------------------------------------------------------------------------------------------------------
Aff = Tuple[int, int]                                         # (0, 0) = identity


@dataclass
class Relaxed:                                                # running instance + witness of one side
    comm_W: Aff
    comm_E: Aff
    u: int
    X: List[int]
    W: List[int]
    E: List[int]


@dataclass
class Fresh:                                                  # strict instance (u = 1, E = 0) + witness
    comm_W: Aff
    X: List[int]
    W: List[int]


def split126(v: int) -> Tuple[int, int]:
    return v & ((1 << LIMB) - 1), v >> LIMB


def relaxed_elements(U) -> List[int]:
    """The nine numbers a running instance is hashed as (by the OTHER side's circuit, natively)."""
    return [U.comm_W[0], U.comm_W[1], U.comm_E[0], U.comm_E[1], U.u, *split126(U.X[0]), *split126(U.X[1])]


def trunc(v: int, bits: int) -> int:
    return v & ((1 << bits) - 1)


def hash_state(field: int, params: int, i: int, z0: Sequence[int], zi: Sequence[int], U) -> int:
    return trunc(ps.hash_elements(TAG_STATE, [params, i, *z0, *zi, *relaxed_elements(U)], field), HASH_BITS)


def hash_challenge(field: int, params: int, U, u_W: Aff, u_X: Sequence[int], T: Aff) -> int:
    xs = [params, *relaxed_elements(U), u_W[0], u_W[1], u_X[0], u_X[1], T[0], T[1]]
    return trunc(ps.hash_elements(TAG_CHAL, xs, field), CHAL_BITS)


# ---------------------------------------------------------------------------------------------------------------
# The augmented circuit of side `side` (over SIDE_FIELD[side]); it folds instances of the other side
# ---------------------------------------------------------------------------------------------------------------
@dataclass
class AugInputs:
    params: int
    i: int
    z0: List[int]
    zi: List[int]
    U: Relaxed                 # running instance of the other side (W, E unused)
    u_W: Aff                   # fresh instance of the other side
    u_X: List[int]
    T: Aff


def synthesize_augmented(cs: CS, side: int, inp: AugInputs, step) -> List[int]:
    """Returns z_{i+1}.  Allocation order: inputs, base flag, state hash, challenge, curve checks, the two scalar
    multiplications and additions, the two foreign folds, base-case selection, the step circuit, the output hash."""
    m = cs.m
    pf = o.modulus(SIDE_FIELD[1 - side])                     # modulus of the folded instance's scalars
    a = step.arity()
    params = cs.alloc(inp.params)
    i = cs.alloc(inp.i)
    z0 = [cs.alloc(v) for v in inp.z0]
    zi = [cs.alloc(v) for v in inp.zi]
    U = [cs.alloc(v) for v in relaxed_elements(inp.U)]       # Wx Wy Ex Ey u X0lo X0hi X1lo X1hi
    uWx, uWy = cs.alloc(inp.u_W[0]), cs.alloc(inp.u_W[1])
    uX = [cs.alloc(inp.u_X[0]), cs.alloc(inp.u_X[1])]
    Tx, Ty = cs.alloc(inp.T[0]), cs.alloc(inp.T[1])
    is_base = is_zero(cs, i)
    # the hash this step must have been handed (checked unless i = 0)
    h_in = strict_bits(cs, poseidon_hash(cs, TAG_STATE, [params, i, *z0, *zi, *U]))
    cs.enforce(cs.sub(cs.const(1), is_base), cs.sub(uX[0], pack(cs, h_in[:HASH_BITS])), Num({}, 0))
    # fold challenge
    r_bits = strict_bits(cs, poseidon_hash(cs, TAG_CHAL, [params, *U, uWx, uWy, uX[0], uX[1], Tx, Ty]))[:CHAL_BITS]
    r = pack(cs, r_bits)
    # the two fresh points are on the curve (or the identity)
    uW_inf = is_zero(cs, uWx)
    check_on_curve(cs, uWx, uWy, uW_inf)
    T_inf = is_zero(cs, Tx)
    check_on_curve(cs, Tx, Ty, T_inf)
    # comm_W' = U.W + r u.W ; comm_E' = U.E + r T
    rWx, rWy = ec_scalar_mul(cs, r_bits, uWx, uWy, uW_inf)
    fWx, fWy = ec_add_complete(cs, U[0], U[1], rWx, rWy)
    rTx, rTy = ec_scalar_mul(cs, r_bits, Tx, Ty, T_inf)
    fEx, fEy = ec_add_complete(cs, U[2], U[3], rTx, rTy)
    fu = cs.add(U[4], r)
    # X' = X + r x  in the other field
    xb = [alloc_bits(cs, inp.u_X[k], HASH_BITS) for k in range(2)]
    for k in range(2):
        cs.enforce_equal(pack(cs, xb[k]), uX[k])
    f0 = fold_foreign(cs, U[5], U[6], xb[0], r_bits, pf)
    f1 = fold_foreign(cs, U[7], U[8], xb[1], r_bits, pf)
    fold = [fWx, fWy, fEx, fEy, fu, f0[0], f0[1], f1[0], f1[1]]
    if side == 0:
        base = [cs.const(0)] * 9                              # primary base case: the default instance
    else:                                                     # secondary base case: the first primary instance, relaxed
        base = [uWx, uWy, cs.const(0), cs.const(0), cs.const(1),
                pack(cs, xb[0][:LIMB]), pack(cs, xb[0][LIMB:], 0), pack(cs, xb[1][:LIMB]), pack(cs, xb[1][LIMB:], 0)]
    Unew = [select(cs, is_base, base[k], fold[k]) for k in range(9)]
    z_in = [select(cs, is_base, z0[k], zi[k]) for k in range(a)]
    z_out = step.synthesize(cs, z_in)
    i_new = cs.add(i, cs.const(1))
    h_out = strict_bits(cs, poseidon_hash(cs, TAG_STATE, [params, i_new, *z0, *z_out, *Unew]))
    x0 = cs.alloc_io(uX[1].v)
    cs.enforce_equal(x0, uX[1])
    hv = pack(cs, h_out[:HASH_BITS])
    x1 = cs.alloc_io(hv.v)
    cs.enforce_equal(x1, hv)
    return [n.v for n in z_out]


# ---------------------------------------------------------------------------------------------------------------
# Public parameters, NIFS, RecursiveSNARK
# ---------------------------------------------------------------------------------------------------------------
def default_relaxed(num_vars: int = 0, num_cons: int = 0) -> Relaxed:
    return Relaxed((0, 0), (0, 0), 0, [0, 0], [0] * num_vars, [0] * num_cons)


def dummy_inputs(arity: int) -> AugInputs:
    return AugInputs(0, 0, [0] * arity, [0] * arity, default_relaxed(), (0, 0), [0, 0], (0, 0))


def digest_shapes(t: int, shapes: Sequence[o.R1CSShape], gens_seed: int, gens_family: int) -> int:
    """`params`: SHAKE256 over both shapes, the generator family and the RO label, truncated to 250 bits."""
    h = hashlib.shake_256()
    h.update(b"vdf-nova-ivc-v1" + ps.label())
    h.update(int(t).to_bytes(8, "little") + int(gens_seed).to_bytes(8, "little") + int(gens_family).to_bytes(8, "little"))
    for sh in shapes:
        for v in (sh.num_cons, sh.num_vars, sh.num_io):
            h.update(int(v).to_bytes(8, "little"))
        for mat in (sh.A, sh.B, sh.C):
            h.update(len(mat).to_bytes(8, "little"))
            for r, c, v in mat:
                h.update(int(r).to_bytes(4, "little") + int(c).to_bytes(4, "little") + int(v).to_bytes(32, "little"))
    return trunc(int.from_bytes(h.digest(32), "little"), HASH_BITS)


@dataclass
class PublicParams:
    t: int
    shapes: List[o.R1CSShape]
    params: int
    bound: bool
    commit: Callable[[int, Sequence[int]], Aff]               # commit(side, vector) under that side's generators


def public_params(t: int, commit, gens_seed: int, gens_family: int, bound: bool = False, primary=None) -> PublicParams:
    """src/nova/proof.rs:232-237: both augmented circuits synthesised once for their shapes.  `primary`: another step
    circuit than InverseMinRootCircuit on the primary side (anything with arity / synthesize / output; t is then 0)."""
    shapes = []
    for side, step in ((0, primary or InverseMinRootCircuit(t, None, None, bound)), (1, TrivialTestCircuit())):
        cs = CS(SIDE_FIELD[side])
        synthesize_augmented(cs, side, dummy_inputs(step.arity()), step)
        shapes.append(cs.shape())
    return PublicParams(t, shapes, digest_shapes(t, shapes, gens_seed, gens_family), bound, commit)


def nifs_fold(pp: PublicParams, side: int, run: Relaxed, fr: Fresh) -> Tuple[Relaxed, Aff, int]:
    """NIFS.prove for an instance of `side`; the challenge is the one the OTHER side's circuit
    recomputes, hence over that circuit's field."""
    sh, m = pp.shapes[side], o.modulus(SIDE_FIELD[side])
    a1, b1, c1 = o.multiply_vec(sh, run.W + [run.u] + run.X, m)
    a2, b2, c2 = o.multiply_vec(sh, fr.W + [1] + fr.X, m)
    T = o.cross_term(a1, b1, c1, a2, b2, c2, run.u, m)
    cT = pp.commit(side, T)
    r = hash_challenge(SIDE_FIELD[1 - side], pp.params, run, fr.comm_W, fr.X, cT)
    cW = ec_fold(side, run.comm_W, r, fr.comm_W)
    cE = ec_fold(side, run.comm_E, r, cT)
    out = Relaxed(cW, cE, (run.u + r) % m, o.axpy(run.X, r, fr.X, m), o.axpy(run.W, r, fr.W, m), o.axpy(run.E, r, T, m))
    return out, cT, r


def ec_fold(side: int, a: Aff, r: int, b: Aff) -> Aff:
    bm = o.curve_base_modulus(SIDE_CURVE[side])
    pt = lambda q: None if q == (0, 0) else q
    return o.pt_add(pt(a), o.pt_mul(r, pt(b), bm), bm) or (0, 0)


@dataclass
class RecursiveSNARK:
    i: int
    zi: List[List[int]]                                       # zi[side]
    r: List[Relaxed]                                          # running instance + witness per side
    l2: Optional[Fresh]                                       # the last (unfolded) secondary instance
    trace: list = dc_field(default_factory=list)              # per step: dict of what the product must reproduce


def synth_fresh(pp: PublicParams, side: int, inp: AugInputs, step) -> Tuple[Fresh, List[int]]:
    cs = CS(SIDE_FIELD[side])
    z_next = synthesize_augmented(cs, side, inp, step)
    assert cs.rows == pp.shapes[side].num_cons and len(cs.W) == pp.shapes[side].num_vars
    return Fresh(pp.commit(side, cs.W), list(cs.X), list(cs.W)), z_next


def prove_step(pp: PublicParams, snark: Optional[RecursiveSNARK], c1, z0_1: Sequence[int],
               z0_2: Sequence[int] = (0,)) -> RecursiveSNARK:
    """RecursiveSNARK::prove_step (src/nova/proof.rs:342-349)."""
    c2 = TrivialTestCircuit()
    sh = pp.shapes
    if snark is None:
        f1, z1 = synth_fresh(pp, 0, AugInputs(pp.params, 0, list(z0_1), list(z0_1), default_relaxed(), (0, 0), [0, 0], (0, 0)), c1)
        f2, z2 = synth_fresh(pp, 1, AugInputs(pp.params, 0, list(z0_2), list(z0_2), default_relaxed(), f1.comm_W, f1.X, (0, 0)), c2)
        r1 = Relaxed(f1.comm_W, (0, 0), 1, list(f1.X), list(f1.W), [0] * sh[0].num_cons)
        r2 = default_relaxed(sh[1].num_vars, sh[1].num_cons)
        out = RecursiveSNARK(1, [z1, z2], [r1, r2], f2)
        out.trace.append(dict(l1=f1, l2=f2))
        return out
    s = snark
    r2_new, T2, ch2 = nifs_fold(pp, 1, s.r[1], s.l2)
    f1, z1 = synth_fresh(pp, 0, AugInputs(pp.params, s.i, list(z0_1), s.zi[0], s.r[1], s.l2.comm_W, s.l2.X, T2), c1)
    r1_new, T1, ch1 = nifs_fold(pp, 0, s.r[0], f1)
    f2, z2 = synth_fresh(pp, 1, AugInputs(pp.params, s.i, list(z0_2), s.zi[1], s.r[0], f1.comm_W, f1.X, T1), c2)
    out = RecursiveSNARK(s.i + 1, [z1, z2], [r1_new, r2_new], f2, s.trace)
    out.trace.append(dict(l1=f1, l2=f2, T1=T1, T2=T2, r1=ch1, r2=ch2))
    return out


def verify(pp: PublicParams, s: RecursiveSNARK, num_steps: int, z0_1: Sequence[int], z0_2: Sequence[int] = (0,)):
    """RecursiveSNARK::verify: (zi_primary, zi_secondary) or None (src/nova/proof.rs:381-386 compares them)."""
    if num_steps == 0 or s.i != num_steps:
        return None
    if len(s.l2.X) != 2 or len(s.r[0].X) != 2 or len(s.r[1].X) != 2:
        return None
    if s.l2.X[0] != hash_state(SIDE_FIELD[0], pp.params, num_steps, z0_1, s.zi[0], s.r[1]):
        return None
    if s.l2.X[1] != hash_state(SIDE_FIELD[1], pp.params, num_steps, z0_2, s.zi[1], s.r[0]):
        return None
    for side in (0, 1):
        R, m = s.r[side], o.modulus(SIDE_FIELD[side])
        if pp.commit(side, R.W) != R.comm_W or pp.commit(side, R.E) != R.comm_E:
            return None
        if not o.is_sat_relaxed(pp.shapes[side], R.W, R.E, R.u, R.X, m):
            return None
    m2 = o.modulus(SIDE_FIELD[1])
    if pp.commit(1, s.l2.W) != s.l2.comm_W:
        return None
    if not o.is_sat_relaxed(pp.shapes[1], s.l2.W, [0] * pp.shapes[1].num_cons, 1, s.l2.X, m2):
        return None
    return s.zi[0], s.zi[1]


# ---------------------------------------------------------------------------------------------------------------
# Commitments through the C restatement (oracle/pasta_ref.c): Pedersen over a seeded generator family
# ---------------------------------------------------------------------------------------------------------------
GENS_SEED = 0x4E6F7661              # "Nova"
FAMILY_KNOWN_DLOG, FAMILY_TRY_AND_INCREMENT = 0, 1


class CCommit:
    """commit(side, v) = sum v_i G_i with G = generator family `family` of that side's curve (family 0: [k_i]G with
    known k_i -- tests only; family 1: try-and-increment, what public_params uses)."""

    def __init__(self, family: int = FAMILY_TRY_AND_INCREMENT, seed: int = GENS_SEED, threads: int = 4):
        from . import cref
        self.cref, self.L = cref, cref.lib()
        self.family, self.seed, self.threads = family, seed, threads
        self.gens = {0: None, 1: None}

    def _gens(self, side: int, n: int):
        import numpy as np
        have = self.gens[side]
        if have is None or have.shape[0] < n:
            g = np.zeros((n, 8), dtype="<u8")
            fn = self.L.ref_tai_bases if self.family == FAMILY_TRY_AND_INCREMENT else self.L.ref_synthetic_bases
            fn(SIDE_CURVE[side], self.seed, 0, n, self.cref.p(g))
            self.gens[side] = have = g
        return have

    def __call__(self, side: int, v: Sequence[int]) -> Aff:
        import numpy as np
        n = len(v)
        if n == 0:
            return (0, 0)
        g = self._gens(side, n)
        sc = np.frombuffer(b"".join(int(x).to_bytes(32, "little") for x in v), dtype="<u8").reshape(n, 4).copy()
        out, aff = np.zeros(12, dtype="<u8"), np.zeros(8, dtype="<u8")
        self.L.ref_msm(SIDE_CURVE[side], self.cref.p(g), self.cref.p(sc), n, 0, self.threads, 0, self.cref.p(out))
        self.L.ref_jac_to_affine(SIDE_CURVE[side], self.cref.p(out), self.cref.p(aff))
        bm = o.curve_base_modulus(SIDE_CURVE[side])
        raw = aff.tobytes()
        return (o.from_mont(int.from_bytes(raw[:32], "little"), bm), o.from_mont(int.from_bytes(raw[32:], "little"), bm))


# ---------------------------------------------------------------------------------------------------------------
# CompressedSNARK (src/nova/proof.rs:360-368 compress, :383 verify): fold the last secondary instance, then one
# Spartan-style argument per curve (oracle/spartan.py, "vdf-spartan-v3") that the two running instances are satisfiable
# ---------------------------------------------------------------------------------------------------------------
def num_gens(sh: o.R1CSShape) -> int:
    g = 1
    while g < max(sh.num_vars, sh.num_cons):
        g <<= 1
    return g


def spartan_setup(pp: PublicParams, side: int):
    """(generator table, the extra generator U = generator number num_gens of the family) of one side."""
    from . import spartan
    import numpy as np
    com = pp.commit
    g = num_gens(pp.shapes[side])
    arr = com._gens(side, g)[:g]
    one = np.zeros((1, 8), dtype="<u8")
    fn = com.L.ref_tai_bases if com.family == FAMILY_TRY_AND_INCREMENT else com.L.ref_synthetic_bases
    fn(SIDE_CURVE[side], com.seed, g, 1, com.cref.p(one))
    bm = o.curve_base_modulus(SIDE_CURVE[side])
    raw = one.tobytes()
    U = (o.from_mont(int.from_bytes(raw[:32], "little"), bm), o.from_mont(int.from_bytes(raw[32:64], "little"), bm))
    return spartan.Gens(arr, SIDE_CURVE[side]), U


@dataclass
class CompressedSNARK:
    r_U1: Relaxed              # instances only (W, E empty)
    r_U2: Relaxed              # the running secondary instance BEFORE the last fold
    l_u2: Fresh
    T2: Aff
    snark1: object             # spartan.SpartanProof for r_U1
    snark2: object             # for fold(r_U2, l_u2)
    zi: List[List[int]]


def _inst_only(R: Relaxed) -> Relaxed:
    return Relaxed(R.comm_W, R.comm_E, R.u, list(R.X), [], [])


def _pt(a: Aff):
    return None if a == (0, 0) else a


def compress(pp: PublicParams, s: RecursiveSNARK) -> CompressedSNARK:
    from . import spartan
    f2, T2, _ = nifs_fold(pp, 1, s.r[1], s.l2)
    digest = int(pp.params).to_bytes(32, "little")
    proofs = []
    for side, R in ((0, s.r[0]), (1, f2)):
        G, U = spartan_setup(pp, side)
        proofs.append(spartan.prove(pp.shapes[side], digest, G, U, _pt(R.comm_W), _pt(R.comm_E), R.u, R.X, R.W, R.E, SIDE_CURVE[side]))
    return CompressedSNARK(_inst_only(s.r[0]), _inst_only(s.r[1]), Fresh(s.l2.comm_W, list(s.l2.X), []), T2, proofs[0], proofs[1],
                           [list(s.zi[0]), list(s.zi[1])])


def verify_compressed(pp: PublicParams, c: CompressedSNARK, num_steps: int, z0_1: Sequence[int], z0_2: Sequence[int] = (0,)):
    from . import spartan
    if num_steps == 0:
        return None
    if c.l_u2.X[0] != hash_state(SIDE_FIELD[0], pp.params, num_steps, z0_1, c.zi[0], c.r_U2):
        return None
    if c.l_u2.X[1] != hash_state(SIDE_FIELD[1], pp.params, num_steps, z0_2, c.zi[1], c.r_U1):
        return None
    m2 = o.modulus(SIDE_FIELD[1])
    r = hash_challenge(SIDE_FIELD[0], pp.params, c.r_U2, c.l_u2.comm_W, c.l_u2.X, c.T2)
    f2 = Relaxed(ec_fold(1, c.r_U2.comm_W, r, c.l_u2.comm_W), ec_fold(1, c.r_U2.comm_E, r, c.T2), (c.r_U2.u + r) % m2,
                 o.axpy(c.r_U2.X, r, c.l_u2.X, m2), [], [])
    digest = int(pp.params).to_bytes(32, "little")
    for side, R, pf in ((0, c.r_U1, c.snark1), (1, f2, c.snark2)):
        G, U = spartan_setup(pp, side)
        if not spartan.verify(pp.shapes[side], digest, G, U, _pt(R.comm_W), _pt(R.comm_E), R.u, R.X, pf, SIDE_CURVE[side]):
            return None
    return c.zi[0], c.zi[1]
