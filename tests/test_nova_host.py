"""CPU tests of the host side of the Nova layer (libvdf_nova.so: random oracle, R1CS shapes, augmented-circuit synthesis)
against its specification, oracle/nova.py.  No device call is made: these entry points are host arithmetic only.
Reference anchors: src/nova/proof.rs:79-153 (step circuit), :232-237 (setup), :342-349 (prove_step)."""
import copy
import random

import numpy as np
import pytest

from oracle import nova as nv, pasta as o, poseidon as ps
from util import limbs, ints

import vdf_amd.nova as vn
from vdf_amd.minroot import State


def mont(vals, field):
    return limbs([o.to_mont(v % o.modulus(field), o.modulus(field)) for v in vals])


def unmont(arr, field):
    m = o.modulus(field)
    return [o.from_mont(v, m) for v in ints(arr)]


@pytest.mark.parametrize("field", [o.FIELD_FP, o.FIELD_FQ])
def test_random_oracle_equals_the_restatement(field):
    rng = random.Random(field)
    m = o.modulus(field)
    for n in (0, 1, 2, 3, 4, 7, 16, 17):
        xs = [rng.randrange(m) for _ in range(n)]
        for tag in (1, 2, 99):
            got = unmont(vn.ro_hash(field, tag, mont(xs, field) if n else np.zeros((0, 4), dtype="<u8")), field)[0]
            assert got == ps.hash_elements(tag, xs, field)


@pytest.mark.parametrize("t,kind", [(1, 0), (5, 0), (5, 1), (64, 0)])
def test_shape_digest_equals_the_restatement(t, kind):
    """Both R1CS shapes (every triple of A, B, C on both sides) hash to the oracle's `params`."""
    pp = nv.public_params(t, None, nv.GENS_SEED, nv.FAMILY_TRY_AND_INCREMENT, bound=(kind == 0))
    digest, sizes = vn.shape_digest(t, kind, 1)
    assert digest == pp.params
    for s in (0, 1):
        sh = pp.shapes[s]
        assert sizes[s] == [sh.num_cons, sh.num_vars, len(sh.A) + len(sh.B) + len(sh.C)]
    # the generator family is part of the digest
    assert vn.shape_digest(t, kind, 0)[0] != digest


def c_inputs(side, inp):
    """oracle AugInputs -> vdf_nova_aug_inputs (the folded side's scalars in that side's Montgomery form)."""
    f, pf = nv.SIDE_FIELD[side], nv.SIDE_FIELD[1 - side]
    a = vn.AugInputs()

    def put(dst, vals, field):
        arr = mont(vals, field)
        if isinstance(dst, vn._Fe):
            C = __import__("ctypes")
            C.memmove(C.addressof(dst), arr.ctypes.data, 32)
        else:
            C = __import__("ctypes")
            C.memmove(C.addressof(dst), arr.ctypes.data, 32 * len(vals))
    put(a.params, [inp.params], f); put(a.i, [inp.i], f)
    put(a.z0, list(inp.z0) + [0] * (3 - len(inp.z0)), f); put(a.zi, list(inp.zi) + [0] * (3 - len(inp.zi)), f)
    put(a.U_comm_W, inp.U.comm_W, f); put(a.U_comm_E, inp.U.comm_E, f)
    put(a.U_u, [inp.U.u], pf); put(a.U_X, inp.U.X, pf)
    put(a.u_comm_W, inp.u_W, f); put(a.u_X, inp.u_X, pf); put(a.T, inp.T, f)
    return a


def st(s, field=o.FIELD_FQ):
    return State.from_ints(field, s.x, s.y, s.i)


@pytest.fixture(scope="module")
def oracle_run():
    """Three oracle steps at t = 5 with every circuit's inputs and outputs recorded."""
    t, n = 5, 3
    com = nv.CCommit()
    rec = []
    orig = nv.synth_fresh

    def spy(pp, side, inp, step):
        fresh, z = orig(pp, side, inp, step)
        rec.append((side, copy.deepcopy(inp), step, fresh, z))
        return fresh, z
    nv.synth_fresh = spy
    try:
        pp = nv.public_params(t, com, nv.GENS_SEED, 1)
        init = o.State(0x1234, 0, 1)
        states = [init]
        for _ in range(n):
            states.append(o.minroot_eval(states[-1], t, o.FIELD_FQ))
        z0 = [states[n].x, states[n].y, states[n].i]
        s = None
        for k in range(n):
            s = nv.prove_step(pp, s, nv.InverseMinRootCircuit(t, states[n - k], states[n - k - 1]), z0)
    finally:
        nv.synth_fresh = orig
    return t, pp, rec


def test_augmented_circuit_witness_equals_the_restatement(oracle_run):
    """Every variable of both augmented circuits, base step and two folding steps: W, X and z_{i+1} bit for bit."""
    t, pp, rec = oracle_run
    assert len(rec) == 6
    for side, inp, step, fresh, z_next in rec:
        f = nv.SIDE_FIELD[side]
        res = st(step.result) if side == 0 else None
        inn = st(step.input) if side == 0 else None
        W, X, zn, nc = vn.aug_synthesize(side, t, 1, c_inputs(side, inp), res, inn)        # 1: the reference's circuit (the oracle's default)
        assert nc == pp.shapes[side].num_cons and W.shape[0] == pp.shapes[side].num_vars
        assert unmont(X, f) == fresh.X
        assert unmont(zn, f) == z_next
        got = unmont(W, f)
        bad = [k for k in range(len(got)) if got[k] != fresh.W[k]]
        assert not bad, (side, inp.i, bad[:5])
        # all 2 x (128 chords + 127 tangents + 1 final slope) inverses came from the batched pre-pass
        assert vn.synthesis_stats() == (512, 0)


def test_bound_circuit_witness(oracle_run):
    """The bound form (3 variables per round: new_x as the linear combination y - i + 1) through the same seam."""
    t, pp, rec = oracle_run
    side, inp, step, fresh, z_next = rec[2]                     # a folding step of the primary side
    bound_step = nv.InverseMinRootCircuit(t, step.result, step.input, bound=True)
    cs = nv.CS(nv.SIDE_FIELD[0])
    z = nv.synthesize_augmented(cs, 0, inp, bound_step)
    W, X, zn, nc = vn.aug_synthesize(0, t, 0, c_inputs(0, inp), st(step.result), st(step.input))
    assert unmont(W, o.FIELD_FQ) == cs.W and unmont(zn, o.FIELD_FQ) == z and nc == cs.rows
    assert len(cs.W) == len(fresh.W) - t                        # the reference's allocation (4 per round, src/nova/proof.rs:167-181) has t more


def test_committed_known_answers(golden):
    """tests/golden/vectors.json: the random oracle on (1..5) in both fields and the parameters digest at t = 1."""
    for field in (o.FIELD_FP, o.FIELD_FQ):
        want = int(golden["ro"][str(field)], 16)
        assert ps.hash_elements(1, [1, 2, 3, 4, 5], field) == want
        assert unmont(vn.ro_hash(field, 1, mont([1, 2, 3, 4, 5], field)), field)[0] == want
    assert vn.shape_digest(1, 0, 1)[0] == int(golden["params_t1"], 16)
    assert vn.shape_digest(1, 1, 1)[0] == int(golden["params_t1_reference"], 16) == vn.shape_digest(1)[0]      # the default: the reference's circuit


def test_custom_step_circuit_shape_through_the_seam():
    """The C ABI's vdf_step_circuit (src/nova/proof.rs:79-153): a host-written circuit's shapes hash to the oracle's
    `params` for the same circuit; a circuit that fails while its shape is recorded is an error."""
    Q = o.Q
    fe = lambda v: limbs([o.to_mont(v % Q, Q)]).tobytes()

    class Cubic(vn.StepCircuit):
        arity = 1

        def synthesize(self, cs, z):
            x = z[0]
            x3 = cs.mul(cs.mul(x, x), x)
            rhs = cs.add(cs.add(x3, x), cs.const(fe(5)))
            y = cs.alloc(cs.value(rhs) if cs.is_witness else None)
            cs.enforce(rhs, cs.const(fe(1)), y)
            return [y]

    class Broken(vn.StepCircuit):
        arity = 1

        def synthesize(self, cs, z):
            raise RuntimeError("no shape")
    digest, sizes = vn.shape_digest_custom(Cubic())
    opp = nv.public_params(0, None, nv.GENS_SEED, nv.FAMILY_TRY_AND_INCREMENT, primary=nv.CubicCircuit())
    assert digest == opp.params and sizes[0][:2] == [opp.shapes[0].num_cons, opp.shapes[0].num_vars]
    with pytest.raises(RuntimeError):
        vn.shape_digest_custom(Broken())


@pytest.mark.parametrize("kind,per", [(1, 4), (0, 3)], ids=["reference", "bound"])
def test_early_rows_are_the_minroot_stencil(kind, per):
    """What vdf_nifs_cross_term_minroot (include/vdf_hip.h) computes without the sparse matrices is compared, triple by
    triple, with the rows InverseMinRootCircuit::synthesize records (src/nova/proof.rs:107-133, :219-227): the host layer
    reports the stencil for both forms of the circuit at every t (host only), and the oracle's shape has those rows --
    the same triples, read off oracle/nova.py's own R1CS -- where the host says they are."""
    for t in (1, 2, 3, 7, 64, 100):
        got_per, row0, nrows, seg = vn.shape_stencil(t, kind)
        assert got_per == per and nrows == 3 * t + 1
        sh = nv.public_params(t, None, nv.GENS_SEED, nv.FAMILY_TRY_AND_INCREMENT, bound=(kind == 0)).shapes[0] if t <= 7 else None
        if sh is None:
            continue
        Q, one = o.Q, sh.num_vars
        rows = {k: {} for k in range(3)}
        for k, mat in enumerate((sh.A, sh.B, sh.C)):
            for r, c, v in mat:
                if row0 <= r < row0 + nrows:
                    rows[k].setdefault(r - row0, {})[c] = v % Q
        for j in range(t):
            rd = seg + per * j
            t1 = rd + per - 3
            y_j = rd - 1 if j else seg - 2
            if per == 4 or j == 0:
                x = {(rd - per if j else seg - 3): 1}
            else:
                x = {(rd - per - 1 if j > 1 else seg - 2): 1, seg - 1: Q - 1, one: j}
            assert rows[0][3 * j] == x and rows[1][3 * j] == x and rows[2][3 * j] == {t1: 1}
            assert rows[0][3 * j + 1] == {t1: 1} == rows[1][3 * j + 1] and rows[2][3 * j + 1] == {t1 + 1: 1}
            assert rows[0][3 * j + 2] == {t1 + 1: 1} and rows[1][3 * j + 2] == x
            assert rows[2][3 * j + 2] == {t1 + 2: 1, y_j: 1, seg - 1: Q - 1, one: j + 1}
        assert rows[0][3 * t] == {seg + per * t: 1} and rows[1][3 * t] == {one: 1} and rows[2][3 * t] == {seg - 1: 1, one: (Q - t) % Q}


# ---- the random oracle as a parameter block (vdf_nova_ro_params; oracle/poseidon.py RoSpec) -------------------------
RO_SMALL = (ps.RoSpec(family=1, width=9, full_rounds=8, partial_rounds=30), dict(width=9, partial_rounds=30))


@pytest.mark.parametrize("field", [o.FIELD_FP, o.FIELD_FQ])
def test_ro_block_native_hash_equals_the_restatement(field):
    """The original Poseidon permutation (family 1: Grain-LFSR constants, Cauchy MDS) in the host library against the oracle's,
    for the neptune-shaped block (width 25, 8 + 57 rounds) and a smaller instance; the default block is untouched by it."""
    rng = random.Random(100 + field)
    m = o.modulus(field)
    for spec, ro in ((ps.NEPTUNE_SHAPED, vn.ro_preset(vn.RO_NEPTUNE_SHAPED)), (RO_SMALL[0], vn.ro_preset(vn.RO_NEPTUNE_SHAPED, **RO_SMALL[1]))):
        assert ro.as_dict() == dict(family=spec.family, width=spec.width, full_rounds=spec.full_rounds, partial_rounds=spec.partial_rounds,
                                    alpha=5, challenge_bits=128, hash_bits=250)
        for n in (0, 1, spec.width - 2, spec.width - 1, spec.width, 2 * spec.width + 3):
            xs = [rng.randrange(m) for _ in range(n)]
            with ps.using(spec):
                want = ps.hash_elements(7, xs, field)
            got = unmont(vn.ro_hash(field, 7, mont(xs, field) if n else np.zeros((0, 4), dtype="<u8"), ro), field)[0]
            assert got == want
            assert n == 0 or got != unmont(vn.ro_hash(field, 7, mont(xs, field)), field)[0]         # another oracle than the default
    xs = [1, 2, 3, 4, 5]
    assert unmont(vn.ro_hash(field, 1, mont(xs, field), vn.ro_preset(vn.RO_DEFAULT)), field)[0] == ps.hash_elements(1, xs, field)
    for bad in (dict(alpha=3), dict(challenge_bits=120), dict(hash_bits=248), dict(width=26), dict(width=1), dict(full_rounds=7), dict(family=2)):
        with pytest.raises(Exception):
            vn.ro_hash(field, 1, mont(xs, field), vn.ro_preset(vn.RO_NEPTUNE_SHAPED, **bad))
    with pytest.raises(Exception):                                          # family 0 supports the default numbers only
        vn.ro_hash(field, 1, mont(xs, field), vn.ro_preset(vn.RO_DEFAULT, partial_rounds=57))


def test_ro_block_changes_shapes_and_digest_as_the_restatement_says():
    """A parameter set under another block: both augmented circuits hold that block's permutations (other sizes, other
    triples), and the digest -- which absorbs the block itself -- equals the oracle's under the same block."""
    t, kind = 3, 1
    spec, fields = RO_SMALL
    ro = vn.ro_preset(vn.RO_NEPTUNE_SHAPED, **fields)
    with ps.using(spec):
        pp = nv.public_params(t, None, nv.GENS_SEED, nv.FAMILY_TRY_AND_INCREMENT)
    digest, sizes = vn.shape_digest(t, kind, 1, ro)
    assert digest == pp.params
    for s in (0, 1):
        sh = pp.shapes[s]
        assert sizes[s] == [sh.num_cons, sh.num_vars, len(sh.A) + len(sh.B) + len(sh.C)]
    d0, s0 = vn.shape_digest(t, kind, 1)
    assert d0 != digest and s0 != sizes
    assert vn.shape_digest(t, kind, 1, vn.ro_preset(vn.RO_DEFAULT)) == (d0, s0)


def test_ro_block_augmented_witness_equals_the_restatement():
    """Two oracle steps at t = 3 under the small classic-Poseidon block: every variable of the four augmented circuits."""
    t, n = 3, 2
    spec, fields = RO_SMALL
    ro = vn.ro_preset(vn.RO_NEPTUNE_SHAPED, **fields)
    com = nv.CCommit()
    rec = []
    orig = nv.synth_fresh

    def spy(pp, side, inp, step):
        fresh, z = orig(pp, side, inp, step)
        rec.append((side, copy.deepcopy(inp), step, fresh, z))
        return fresh, z
    nv.synth_fresh = spy
    try:
        with ps.using(spec):
            pp = nv.public_params(t, com, nv.GENS_SEED, 1)
            states = [o.State(0x4321, 0, 1)]
            for _ in range(n):
                states.append(o.minroot_eval(states[-1], t, o.FIELD_FQ))
            z0 = [states[n].x, states[n].y, states[n].i]
            s = None
            for k in range(n):
                s = nv.prove_step(pp, s, nv.InverseMinRootCircuit(t, states[n - k], states[n - k - 1]), z0)
            assert nv.verify(pp, s, n, z0) is not None
    finally:
        nv.synth_fresh = orig
    assert len(rec) == 4
    for side, inp, step, fresh, z_next in rec:
        f = nv.SIDE_FIELD[side]
        res = st(step.result) if side == 0 else None
        inn = st(step.input) if side == 0 else None
        W, X, zn, nc = vn.aug_synthesize(side, t, 1, c_inputs(side, inp), res, inn, ro=ro)
        assert nc == pp.shapes[side].num_cons and W.shape[0] == pp.shapes[side].num_vars
        assert unmont(X, f) == fresh.X and unmont(zn, f) == z_next
        got = unmont(W, f)
        bad = [k for k in range(len(got)) if got[k] != fresh.W[k]]
        assert not bad, (side, inp.i, bad[:5])
