"""TEST INFRASTRUCTURE -- never imported by the product (vdf_amd/).

CPU restatement of the folding-only proof layer of libvdf_nova.so ("vdf-nova-fold-v1"): the Pedersen generators of
`public_params`, the shape digest, the SHAKE256 challenge, NIFS.prove / the verifier's fold replay over the exposed-IO
MinRoot step circuit (reference call sites src/nova/proof.rs:302-358 prove_recursively,
:370-387 verify).  nova-snark's own transcript and constants are not in /root/reference (crate dependency,
Cargo.toml:15-18), so this pins the product against THIS restatement, not against nova-snark: parity unpinned
(DESIGN.md section 2).  Plain big-integer Python: use at small t only.
"""
from __future__ import annotations

import hashlib
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

from . import pasta as o

Point = Optional[Tuple[int, int]]
GENS_SEED = 0x4E6F7661              # "Nova"
GENS_FAMILY = 1                     # VDF_GENS_TRY_AND_INCREMENT
_GENS = {}


def gens(n: int, start: int = 0) -> List[Tuple[int, int]]:
    """Generators start .. start+n-1 of public_params (cached)."""
    for i in range(start, start + n):
        if i not in _GENS:
            _GENS[i] = o.tai_base(o.CURVE_PALLAS, GENS_SEED, i)
    return [_GENS[i] for i in range(start, start + n)]


def commit(v: Sequence[int]) -> Tuple[int, int]:
    """Pedersen commitment sum v_i G_i; (0, 0) for the identity, as the product's affine encoding has it."""
    return o.msm_naive(list(v), gens(len(v)), o.CURVE_PALLAS) or (0, 0)


def le32(v: int) -> bytes:
    return int(v).to_bytes(32, "little")


def shape_digest(sh: o.R1CSShape, t: int) -> bytes:
    h = hashlib.shake_256()
    h.update(b"vdf-nova-shape-v1")
    for v in (t, sh.num_cons, sh.num_vars, sh.num_io, GENS_SEED, GENS_FAMILY):
        h.update(int(v).to_bytes(8, "little"))
    for mat in (sh.A, sh.B, sh.C):
        for r, c, v in mat:
            h.update(int(r).to_bytes(4, "little") + int(c).to_bytes(4, "little") + le32(v))
    return h.digest(32)


def challenge(digest: bytes, cW, cE, u: int, X: Sequence[int], cw2, X2: Sequence[int], cT) -> int:
    """128-bit fold challenge r = SHAKE256(label | digest | U1 | u2 | comm_T)."""
    h = hashlib.shake_256()
    h.update(b"vdf-nova-fold-v1" + digest)
    for p in (cW, cE):
        h.update(le32(p[0]) + le32(p[1]))
    h.update(le32(u))
    for v in X:
        h.update(le32(v))
    h.update(le32(cw2[0]) + le32(cw2[1]))
    for v in X2:
        h.update(le32(v))
    h.update(le32(cT[0]) + le32(cT[1]))
    return int.from_bytes(h.digest(16), "little")


def _pt(a) -> Point:
    return None if a == (0, 0) else a


def fold_point(a, r: int, b):
    """a + r b with (0, 0) as the identity."""
    return o.pt_add(_pt(a), o.pt_mul(r, _pt(b), o.P), o.P) or (0, 0)


@dataclass
class Step:
    comm_w: Tuple[int, int]
    comm_T: Tuple[int, int]
    r: int
    X: List[int]


@dataclass
class Proof:
    """Running relaxed instance, its witness, and one record per step."""
    W: List[int]
    E: List[int]
    u: int
    X: List[int]
    comm_W: Tuple[int, int]
    comm_E: Tuple[int, int]
    steps: List[Step] = field(default_factory=list)


def forward_states(initial: o.State, t: int, n: int) -> List[o.State]:
    """states[k] after k * t forward rounds; step k of the proof runs from states[n - k] back to states[n - k - 1]."""
    states = [initial]
    for _ in range(n):
        states.append(o.minroot_eval(states[-1], t, o.FIELD_FQ))
    return states


def prove_chain(initial: o.State, t: int, n: int) -> Tuple[Proof, o.R1CSShape, bytes]:
    """prove_recursively (src/nova/proof.rs:302-358) of the folding-only layer, on the CPU."""
    m = o.Q
    sh = o.step_circuit_shape(t, o.FIELD_FQ)
    digest = shape_digest(sh, t)
    states = forward_states(initial, t, n)
    proof = None
    for k in range(n):
        res, inp = states[n - k], states[n - k - 1]
        W2 = [res.x, res.y, res.i] + o.step_witness_segment(res, t, o.FIELD_FQ)
        X2 = [res.x, res.y, res.i, inp.x, inp.y, inp.i]
        cw2 = commit(W2)
        if proof is None:
            proof = Proof(W2, [0] * sh.num_cons, 1, X2, cw2, (0, 0), [Step(cw2, (0, 0), 0, X2)])
            continue
        a1, b1, c1 = o.multiply_vec(sh, proof.W + [proof.u] + proof.X, m)
        a2, b2, c2 = o.multiply_vec(sh, W2 + [1] + X2, m)
        T = o.cross_term(a1, b1, c1, a2, b2, c2, proof.u, m)
        cT = commit(T)
        r = challenge(digest, proof.comm_W, proof.comm_E, proof.u, proof.X, cw2, X2, cT)
        proof.W, proof.E = o.axpy(proof.W, r, W2, m), o.axpy(proof.E, r, T, m)
        proof.u, proof.X = (proof.u + r) % m, o.axpy(proof.X, r, X2, m)
        proof.comm_W, proof.comm_E = fold_point(proof.comm_W, r, cw2), fold_point(proof.comm_E, r, cT)
        proof.steps.append(Step(cw2, cT, r, X2))
    return proof, sh, digest


def replay(digest: bytes, steps: Sequence[Step]):
    """The verifier's fold replay: (comm_W, comm_E, u, X) the records fold to, or None if a challenge is wrong."""
    m = o.Q
    cW, cE, u, X = steps[0].comm_w, (0, 0), 1, list(steps[0].X)
    for s in steps[1:]:
        r = challenge(digest, cW, cE, u, X, s.comm_w, s.X, s.comm_T)
        if r != s.r:
            return None
        cW, cE = fold_point(cW, r, s.comm_w), fold_point(cE, r, s.comm_T)
        u, X = (u + r) % m, o.axpy(X, r, s.X, m)
    return cW, cE, u, X


def verify(proof: Proof, sh: o.R1CSShape, digest: bytes, t: int, z0: Sequence[int], zi: Sequence[int]) -> bool:
    """NovaVDFProof::verify (src/nova/proof.rs:370-387) for the folding-only layer."""
    m = o.Q
    st = proof.steps
    if not st or list(st[0].X[:3]) != list(z0) or list(st[-1].X[3:]) != list(zi):
        return False
    for a, b in zip(st, st[1:]):
        if a.X[3:] != b.X[:3]:
            return False
    if any((s.X[2] - s.X[5]) % m != t % m for s in st):
        return False
    folded = replay(digest, st)
    if folded is None or folded != (proof.comm_W, proof.comm_E, proof.u, proof.X):
        return False
    if commit(proof.W) != proof.comm_W or commit(proof.E) != proof.comm_E:
        return False
    return o.is_sat_relaxed(sh, proof.W, proof.E, proof.u, proof.X, m)
