"""Big-integer oracle for the protocol/vdf Nova/MinRoot hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the shipped
product: only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` may import it, and there only as the checker.

PARITY STATUS: **parity unpinned for known-answer values.**  The reference
(`/root/reference`, Rust) cannot be compiled or run in this environment and its
own tests hold no golden vector, proof byte, commitment or digest for this path
(SURVEY.md section 8c).  What the reference does pin, and what this oracle is
checked against in ``tests/test_oracle.py``:

* the exponent constants ``FP_RESCUE_INVALPHA`` / ``FQ_RESCUE_INVALPHA``
  (src/minroot.rs:273-285): ``5 * e == 1 (mod m-1)``;
* ``test_exponents``  (src/minroot.rs:449-458): inverse exponent is 5;
* ``test_steps``      (src/minroot.rs:460-477): inverse_step(forward_step(x)) == x;
* ``test_eval``       (src/minroot.rs:479-510): all four EvalModes agree and
  inverse_eval(eval(s, t), t) == s, check(result, t, s);
* ``test_vanilla_proof`` (src/minroot.rs:512-542): chaining, result.i == n*t;
* the circuit's own debug assertions (src/nova/proof.rs:194-216) and the three
  constraints per round (src/nova/proof.rs:176-178, 219-227).

Everything else is exact prime-field / prime-order-group arithmetic, whose
results are canonical (unique in [0, m) / unique affine point), restated from
the published definitions of the Pasta curves (pasta_curves 0.4.0), Pippenger
MSM (pasta-msm 0.1.1) and the Nova folding scheme (nova-snark 0.8.0), none of
whose sources are present under /root/reference.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Iterable, List, Optional, Sequence, Tuple

# ---------------------------------------------------------------------------
# Constants (pasta_curves 0.4.0 published parameters)
# ---------------------------------------------------------------------------
P = 0x40000000000000000000000000000000224698FC094CF91B992D30ED00000001  # Fp: Pallas base / Vesta scalar
Q = 0x40000000000000000000000000000000224698FC0994A8DD8C46EB2100000001  # Fq: Vesta base / Pallas scalar
R_BITS = 256
R = 1 << R_BITS
CURVE_B = 5  # y^2 = x^3 + 5 on both curves

FIELD_FP = 0
FIELD_FQ = 1
CURVE_PALLAS = 0  # points over Fp, order Q, scalars in Fq
CURVE_VESTA = 1   # points over Fq, order P, scalars in Fp


def modulus(field: int) -> int:
    return P if field == FIELD_FP else Q


def curve_base_modulus(curve: int) -> int:
    return P if curve == CURVE_PALLAS else Q


def curve_scalar_modulus(curve: int) -> int:
    return Q if curve == CURVE_PALLAS else P


# src/minroot.rs:273-285 (little-endian u64 limbs)
FP_RESCUE_INVALPHA_LIMBS = [0xE0F0F3F0CCCCCCCD, 0x4E9EE0C9A10A60E2, 0x3333333333333333, 0x3333333333333333]
FQ_RESCUE_INVALPHA_LIMBS = [0xD69F2280CCCCCCCD, 0x4E9EE0C9A143BA4A, 0x3333333333333333, 0x3333333333333333]


def limbs_to_int(limbs: Sequence[int]) -> int:
    return sum(int(l) << (64 * i) for i, l in enumerate(limbs))


def int_to_limbs(x: int, n: int = 4) -> List[int]:
    return [(x >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(n)]


FP_RESCUE_INVALPHA = limbs_to_int(FP_RESCUE_INVALPHA_LIMBS)
FQ_RESCUE_INVALPHA = limbs_to_int(FQ_RESCUE_INVALPHA_LIMBS)


# ---------------------------------------------------------------------------
# Montgomery representation (pasta_curves stores 4 x u64 LE limbs of x*R mod m)
# ---------------------------------------------------------------------------
def to_mont(x: int, m: int) -> int:
    return (x % m) * R % m


def from_mont(xm: int, m: int) -> int:
    return xm * pow(R, -1, m) % m


def fe_to_bytes(x: int) -> bytes:
    return int(x).to_bytes(32, "little")


def fe_from_bytes(b: bytes) -> int:
    return int.from_bytes(b, "little")


# ---------------------------------------------------------------------------
# Deterministic input generator shared by the oracle, the C restatement and the
# device (splitmix64; constants from Vigna's public-domain reference).
# ---------------------------------------------------------------------------
MASK64 = (1 << 64) - 1


def splitmix64(x: int) -> int:
    z = (x + 0x9E3779B97F4A7C15) & MASK64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK64
    return z ^ (z >> 31)


def base_dlog(seed: int, i: int) -> int:
    """64-bit discrete log of synthetic base i:  P_i = [k_i] G, k_i != 0."""
    k = splitmix64((seed * 0xD1342543DE82EF95 + i) & MASK64)
    return k | 1


def rand_fe(seed: int, i: int, m: int) -> int:
    """Field element from four splitmix64 words, reduced mod m."""
    v = 0
    for j in range(4):
        v |= splitmix64((seed * 0x2545F4914F6CDD1D + 4 * i + j) & MASK64) << (64 * j)
    return v % m


# ---------------------------------------------------------------------------
# Short-Weierstrass y^2 = x^3 + 5, affine, identity = None.  Memory encoding of
# the identity is (0, 0) as in pasta_curves' repr-c affine (header).
# ---------------------------------------------------------------------------
Point = Optional[Tuple[int, int]]
GEN = (-1, 2)  # (-1, 2) lies on both curves: 4 == -1 + 5


def generator(curve: int) -> Point:
    m = curve_base_modulus(curve)
    return (GEN[0] % m, GEN[1] % m)


def on_curve(pt: Point, curve: int) -> bool:
    if pt is None:
        return True
    m = curve_base_modulus(curve)
    x, y = pt
    return (y * y - (x * x * x + CURVE_B)) % m == 0


def pt_neg(pt: Point, m: int) -> Point:
    if pt is None:
        return None
    return (pt[0], (-pt[1]) % m)


def pt_add(a: Point, b: Point, m: int) -> Point:
    if a is None:
        return b
    if b is None:
        return a
    x1, y1 = a
    x2, y2 = b
    if x1 == x2:
        if (y1 + y2) % m == 0:
            return None
        lam = 3 * x1 * x1 * pow(2 * y1, -1, m) % m
    else:
        lam = (y2 - y1) * pow(x2 - x1, -1, m) % m
    x3 = (lam * lam - x1 - x2) % m
    y3 = (lam * (x1 - x3) - y1) % m
    return (x3, y3)


def pt_mul(k: int, pt: Point, m: int) -> Point:
    if k < 0:
        return pt_mul(-k, pt_neg(pt, m), m)
    acc: Point = None
    add = pt
    while k:
        if k & 1:
            acc = pt_add(acc, add, m)
        add = pt_add(add, add, m)
        k >>= 1
    return acc


def msm_naive(scalars: Sequence[int], bases: Sequence[Point], curve: int) -> Point:
    """sum_i s_i * B_i  (the Pedersen commitment of SURVEY.md K1/K2)."""
    m = curve_base_modulus(curve)
    acc: Point = None
    for s, b in zip(scalars, bases):
        acc = pt_add(acc, pt_mul(s % curve_scalar_modulus(curve), b, m), m)
    return acc


def synthetic_bases(curve: int, seed: int, n: int, start: int = 0) -> List[Point]:
    m = curve_base_modulus(curve)
    g = generator(curve)
    return [pt_mul(base_dlog(seed, start + i), g, m) for i in range(n)]


def sqrt_mod(a: int, m: int) -> Optional[int]:
    """A square root of a mod the prime m (Tonelli-Shanks), or None for a non-residue."""
    a %= m
    if a == 0:
        return 0
    if pow(a, (m - 1) // 2, m) != 1:
        return None
    s, t = 0, m - 1
    while t % 2 == 0:
        s, t = s + 1, t // 2
    z = 2
    while pow(z, (m - 1) // 2, m) != m - 1:
        z += 1
    c, x, b = pow(z, t, m), pow(a, (t + 1) // 2, m), pow(a, t, m)
    while b != 1:
        k, tt = 0, b
        while tt != 1:
            tt, k = tt * tt % m, k + 1
        g = pow(c, 1 << (s - k - 1), m)
        x, c = x * g % m, g * g % m
        b, s = b * c % m, k
    return x


def tai_base(curve: int, seed: int, i: int) -> Point:
    """Generator family 1 (include/vdf_hip.h VDF_GENS_TRY_AND_INCREMENT): xoshiro256**
    seeded by four splitmix64 words of (seed, index); candidates x = 256 stream bits mod m until x^3 + 5 is a
    square; y = the even root."""
    m = curve_base_modulus(curve)
    sm = (seed * 0xD1342543DE82EF95 + i * 0x9E3779B97F4A7C15) & MASK64
    st = []
    for _ in range(4):
        st.append(splitmix64(sm))
        sm = (sm + 0x9E3779B97F4A7C15) & MASK64

    def rotl(x, k):
        return ((x << k) | (x >> (64 - k))) & MASK64

    def nxt():
        r = (rotl(st[1] * 5 & MASK64, 7) * 9) & MASK64
        t = (st[1] << 17) & MASK64
        st[2] ^= st[0]; st[3] ^= st[1]; st[1] ^= st[2]; st[0] ^= st[3]
        st[2] ^= t
        st[3] = rotl(st[3], 45)
        return r

    while True:
        x = sum(nxt() << (64 * k) for k in range(4)) % m
        y = sqrt_mod(x * x * x + 5, m)
        if y is None:
            continue
        if y & 1:
            y = m - y
        return (x, y)


def label_base(curve: int, label: bytes, i: int) -> Point:
    """Generator family 2 (include/vdf_hip.h vdf_bases_generate_label): derived from a label through SHAKE256, the way
    nova-snark derives CommitGens (its own encoding is absent from /root/reference: this one is the build's own)."""
    import hashlib
    assert len(label) <= 64
    m = curve_base_modulus(curve)
    ctr = 0
    while True:
        msg = b"vdf-gens-v1" + bytes([curve, len(label)]) + label + int(i).to_bytes(8, "little") + ctr.to_bytes(4, "little")
        x = int.from_bytes(hashlib.shake_256(msg).digest(64), "little") % m
        ctr += 1
        y = sqrt_mod(x * x * x + 5, m) if x else None
        if y is None:
            continue
        return (x, m - y if y & 1 else y)


def tai_bases(curve: int, seed: int, n: int, start: int = 0) -> List[Point]:
    return [tai_base(curve, seed, start + i) for i in range(n)]


def msm_by_dlog(scalars: Iterable[int], curve: int, seed: int, start: int = 0) -> Point:
    """Expected MSM over synthetic_bases in O(n) field work: [sum s_i k_i mod r] G."""
    r = curve_scalar_modulus(curve)
    acc = 0
    for i, s in enumerate(scalars):
        acc += (s % r) * base_dlog(seed, start + i)
    return pt_mul(acc % r, generator(curve), curve_base_modulus(curve))


def msm_by_dlog_limbs(scalars, curve: int, seed: int, start: int = 0) -> Point:
    """msm_by_dlog for a uint64[n, 4] little-endian limb array (values below the scalar modulus), vectorised so that
    2^24 terms take seconds: sum s_i k_i is accumulated exactly as 16 x 4 dot products of 16-bit pieces in uint64
    (each product < 2^32, n <= 2^24 of them per sum < 2^56)."""
    import numpy as np
    s = np.ascontiguousarray(scalars, dtype="<u8").reshape(-1, 4)
    n = s.shape[0]
    assert n <= 1 << 24
    r = curve_scalar_modulus(curve)
    with np.errstate(over="ignore"):
        z = np.uint64((seed * 0xD1342543DE82EF95 + start) & MASK64) + np.arange(n, dtype=np.uint64)
        z = z + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        k = (z ^ (z >> np.uint64(31))) | np.uint64(1)
    s16 = s.view("<u2").reshape(n, 16).astype(np.uint64)
    k16 = k.view("<u2").reshape(n, 4).astype(np.uint64)
    acc = 0
    for a in range(16):
        col = np.ascontiguousarray(s16[:, a])
        for b in range(4):
            acc += int(np.dot(col, k16[:, b])) << (16 * (a + b))
    return pt_mul(acc % r, generator(curve), curve_base_modulus(curve))


def point_to_affine_ints(pt: Point) -> Tuple[int, int]:
    return (0, 0) if pt is None else pt


# ---------------------------------------------------------------------------
# MinRoot VDF  (src/minroot.rs)
# ---------------------------------------------------------------------------
@dataclass(frozen=True)
class State:  # src/minroot.rs:267-272
    x: int
    y: int
    i: int


EVAL_MODES = ("LTRSequential", "LTRAddChainSequential", "RTLSequential", "RTLAddChainSequential")  # :14-31


def _sqr_n(x: int, n: int, m: int) -> int:
    for _ in range(n):
        x = x * x % m
    return x


def forward_step_ltr_sequential(x: int, field: int) -> int:
    """src/minroot.rs:312-314: x.pow_vartime(exponent)."""
    m = modulus(field)
    e = FP_RESCUE_INVALPHA if field == FIELD_FP else FQ_RESCUE_INVALPHA
    return pow(x, e, m)


def forward_step_ltr_addition_chain_fq(x: int) -> int:
    """src/minroot.rs:88-127 (PallasVDF, field Fq)."""
    m = Q
    sqr = lambda v, n: _sqr_n(v, n, m)
    mul = lambda a, b: a * b % m
    sqr_mul = lambda v, n, y: y * sqr(v, n) % m
    q1 = x
    q10 = sqr(q1, 1)
    q11 = mul(q10, q1)
    q101 = mul(q10, q11)
    q110 = sqr(q11, 1)
    q111 = mul(q110, q1)
    q1001 = mul(q111, q10)
    q1111 = mul(q1001, q110)
    qr2 = sqr_mul(q110, 3, q11)
    qr4 = sqr_mul(qr2, 8, qr2)
    qr8 = sqr_mul(qr4, 16, qr4)
    qr16 = sqr_mul(qr8, 32, qr8)
    qr32 = sqr_mul(qr16, 64, qr16)
    v = sqr_mul(qr32, 5, q1001)
    for n, y in ((8, q111), (4, q1), (2, qr4), (7, q11), (6, q1001), (3, q101), (7, q101), (7, q111),
                 (4, q111), (5, q1001), (5, q101), (3, q11), (4, q101), (3, q101), (6, q1111), (4, q1001),
                 (6, q101), (37, qr8), (2, q1)):
        v = sqr_mul(v, n, y)
    return v


def forward_step_rtl_sequential_fq(x: int) -> int:
    """src/minroot.rs:130-151: right-to-left square-and-multiply over 254 bits."""
    m = Q
    acc, sq, e = 1, x, FQ_RESCUE_INVALPHA
    for count in range(254):
        if (e >> count) & 1:
            acc = acc * sq % m
        sq = sq * sq % m
    return acc


def forward_step_rtl_addition_chain_fq(x: int) -> int:
    """src/minroot.rs:154-196: RTL over the low 128 bits, then the 0x33.. tail."""
    m = Q
    acc, sq, e = 1, x, FQ_RESCUE_INVALPHA
    last = 0
    for count in range(128):
        last = sq
        if (e >> count) & 1:
            acc = acc * sq % m
        sq = sq * sq % m
    s = last
    s = s * (s * s % m) % m                      # :179
    s = s * _sqr_n(s, 4, m) % m                  # :180
    for count in range(1, 123):                  # :182-195
        s = s * s % m
        if count % 8 == 1:
            acc = acc * s % m
    return acc


def forward_step_addition_chain_fp(x: int) -> int:
    """src/minroot.rs:223-261 (VestaVDF, field Fp)."""
    m = P
    sqr = lambda v, n: _sqr_n(v, n, m)
    mul = lambda a, b: a * b % m
    sqr_mul = lambda v, n, y: y * sqr(v, n) % m
    p1 = x
    p10 = sqr(p1, 1)
    p11 = mul(p10, p1)
    p101 = mul(p10, p11)
    p110 = sqr(p11, 1)
    p111 = mul(p110, p1)
    p1001 = mul(p111, p10)
    p1111 = mul(p1001, p110)
    pr2 = sqr_mul(p110, 3, p11)
    pr4 = sqr_mul(pr2, 8, pr2)
    pr8 = sqr_mul(pr4, 16, pr4)
    pr16 = sqr_mul(pr8, 32, pr8)
    pr32 = sqr_mul(pr16, 64, pr16)
    v = sqr_mul(pr32, 5, p1001)
    for n, y in ((8, p111), (4, p1), (2, pr4), (7, p11), (6, p1001), (3, p101), (5, p1), (7, p101),
                 (4, p11), (8, p111), (4, p1), (4, p111), (9, p1111), (8, p1111), (6, p1111), (2, p11),
                 (34, pr8), (2, p1)):
        v = sqr_mul(v, n, y)
    return v


def forward_step(x: int, field: int, mode: str = "LTRSequential") -> int:
    """Dispatch of src/minroot.rs:77-84 (Pallas) and :223 (Vesta ignores the mode, :203-205)."""
    if field == FIELD_FP:
        return forward_step_addition_chain_fp(x)
    if mode == "LTRSequential":
        return forward_step_ltr_sequential(x, field)
    if mode == "LTRAddChainSequential":
        return forward_step_ltr_addition_chain_fq(x)
    if mode == "RTLSequential":
        return forward_step_rtl_sequential_fq(x)
    if mode == "RTLAddChainSequential":
        return forward_step_rtl_addition_chain_fq(x)
    raise ValueError(mode)


def inverse_step(x: int, field: int) -> int:
    """src/minroot.rs:73-75, :220-222: x * (x^2)^2."""
    m = modulus(field)
    x2 = x * x % m
    return x * (x2 * x2 % m) % m


def minroot_round(s: State, field: int, mode: str = "LTRSequential") -> State:
    """src/minroot.rs:329-335."""
    m = modulus(field)
    return State(forward_step((s.x + s.y) % m, field, mode), (s.x + s.i) % m, (s.i + 1) % m)


def minroot_inverse_round(s: State, field: int) -> State:
    """src/minroot.rs:338-344."""
    m = modulus(field)
    i = (s.i - 1) % m
    x = (s.y - i) % m
    y = (inverse_step(s.x, field) - x) % m
    return State(x, y, i)


def minroot_eval(s: State, t: int, field: int, mode: str = "LTRSequential") -> State:
    """src/minroot.rs:352-359 (simple_eval)."""
    for _ in range(t):
        s = minroot_round(s, field, mode)
    return s


def minroot_eval_trace(s: State, t: int, field: int, mode: str = "LTRSequential") -> List[State]:
    """simple_eval keeping every state (trace[0] = input ... trace[t] = result).

    The reference discards the trace (src/minroot.rs:352-359); the GPU witness
    kernel consumes it."""
    out = [s]
    for _ in range(t):
        s = minroot_round(s, field, mode)
        out.append(s)
    return out


def minroot_inverse_eval(s: State, t: int, field: int) -> State:
    """src/minroot.rs:363-365."""
    for _ in range(t):
        s = minroot_inverse_round(s, field)
    return s


def minroot_check(result: State, t: int, original: State, field: int) -> bool:
    """src/minroot.rs:369-371."""
    return original == minroot_inverse_eval(result, t, field)


# ---------------------------------------------------------------------------
# Step circuit witness + R1CS shape  (src/nova/proof.rs:87-230)
# ---------------------------------------------------------------------------
def step_witness_segment(result: State, t: int, field: int) -> List[int]:
    """The 4t+1 aux values `InverseMinRootCircuit::synthesize` allocates, in
    allocation order: per round new_x, tmp1, tmp2, new_y (src/nova/proof.rs:167,
    176, 178, 181), then final_i (:122)."""
    m = modulus(field)
    x, y, i = result.x, result.y, result.i
    out: List[int] = []
    for _ in range(t):
        new_i = (i - 1) % m                       # :162-164
        new_x = (y - new_i) % m                   # :167-173
        tmp1 = x * x % m                          # :176
        tmp2 = tmp1 * tmp1 % m                    # :178
        new_y = (tmp2 * x - new_x) % m            # :181-189
        assert tmp2 * x % m == (new_y + y - i + 1) % m   # :209-216
        out += [new_x, tmp1, tmp2, new_y]
        x, y, i = new_x, new_y, new_i
    out.append(i)                                 # final_i, :122-126
    return out


def step_witness_from_trace(trace_xy: Sequence[Tuple[int, int]], i0: int, t: int, field: int) -> List[int]:
    """Same 4t+1 values computed round-parallel from the forward trace
    (trace_xy[k] = (x, y) after k forward rounds, k = 0..t, starting at i = i0).

    Inverse round j (0-based) starts from forward state t-j and lands on t-j-1,
    so new_x, new_y are simply the forward state t-j-1 and only x^2, x^4 of state
    t-j need computing.  This is the layout `vdf_minroot_witness` fills."""
    m = modulus(field)
    out: List[int] = []
    for j in range(t):
        x, _y = trace_xy[t - j]
        nx, ny = trace_xy[t - j - 1]
        tmp1 = x * x % m
        out += [nx % m, tmp1, tmp1 * tmp1 % m, ny % m]
    out.append(i0 % m)
    return out


@dataclass
class R1CSShape:
    """COO triples per matrix over z = (W, u, X)  (SURVEY.md Appendix C)."""
    num_cons: int
    num_vars: int
    num_io: int
    A: List[Tuple[int, int, int]]
    B: List[Tuple[int, int, int]]
    C: List[Tuple[int, int, int]]


def step_circuit_shape(t: int, field: int) -> R1CSShape:
    """R1CS of the exposed-IO wrapper around `InverseMinRootCircuit::synthesize`.

    Variables W = [z_in x, y, i | per round new_x, tmp1, tmp2, new_y | final_i]
    (3 + 4t + 1); constant column u at index num_vars; public IO X = [z_in(3),
    z_out(3)] after it.  Constraint order follows src/nova/proof.rs: per round
    x*x=tmp1 (:176), tmp1*tmp1=tmp2 (:178), tmp2*x = new_y + y - i + 1 (:219-227);
    then final_i*1 = i - t (:128-133); then 6 wrapper rows binding z_in / z_out
    to X.  `i` is carried as a linear combination z_in.i - j (:162-164)."""
    m = modulus(field)
    nv = 3 + 4 * t + 1
    ONE = nv
    A: List[Tuple[int, int, int]] = []
    B: List[Tuple[int, int, int]] = []
    C: List[Tuple[int, int, int]] = []
    x_var, y_var, i_var = 0, 1, 2
    row = 0
    for j in range(t):
        base = 3 + 4 * j
        new_x, tmp1, tmp2, new_y = base, base + 1, base + 2, base + 3
        A.append((row, x_var, 1)); B.append((row, x_var, 1)); C.append((row, tmp1, 1)); row += 1
        A.append((row, tmp1, 1)); B.append((row, tmp1, 1)); C.append((row, tmp2, 1)); row += 1
        # tmp2 * x = new_y + y - (i_in - j) + 1
        A.append((row, tmp2, 1)); B.append((row, x_var, 1))
        C.append((row, new_y, 1)); C.append((row, y_var, 1)); C.append((row, i_var, m - 1))
        C.append((row, ONE, (j + 1) % m)); row += 1
        x_var, y_var = new_x, new_y
    final_i = 3 + 4 * t
    A.append((row, final_i, 1)); B.append((row, ONE, 1))
    C.append((row, i_var, 1)); C.append((row, ONE, (-t) % m)); row += 1
    io = ONE + 1
    for k, v in enumerate((0, 1, 2, x_var, y_var, final_i)):
        A.append((row, v, 1)); B.append((row, ONE, 1)); C.append((row, io + k, 1)); row += 1
    return R1CSShape(row, nv, 6, A, B, C)


def spmv(entries: Sequence[Tuple[int, int, int]], z: Sequence[int], rows: int, m: int) -> List[int]:
    out = [0] * rows
    for r, c, v in entries:
        out[r] = (out[r] + v * z[c]) % m
    return out


def multiply_vec(shape: R1CSShape, z: Sequence[int], m: int):
    return (spmv(shape.A, z, shape.num_cons, m), spmv(shape.B, z, shape.num_cons, m),
            spmv(shape.C, z, shape.num_cons, m))


def cross_term(az1, bz1, cz1, az2, bz2, cz2, u1: int, m: int) -> List[int]:
    """T = AZ1 o BZ2 + AZ2 o BZ1 - u1*CZ2 - u2*CZ1 with u2 = 1."""
    return [(a1 * b2 + a2 * b1 - u1 * c2 - c1) % m
            for a1, b1, c1, a2, b2, c2 in zip(az1, bz1, cz1, az2, bz2, cz2)]


def axpy(a: Sequence[int], r: int, b: Sequence[int], m: int) -> List[int]:
    """a + r*b element-wise (witness / error fold)."""
    return [(x + r * y) % m for x, y in zip(a, b)]


def is_sat_relaxed(shape: R1CSShape, W: Sequence[int], E: Sequence[int], u: int, X: Sequence[int], m: int) -> bool:
    z = list(W) + [u] + list(X)
    az, bz, cz = multiply_vec(shape, z, m)
    return all((a * b - u * c - e) % m == 0 for a, b, c, e in zip(az, bz, cz, E))
