"""CPU tests of the Nova IVC oracle (oracle/nova.py, oracle/poseidon.py): the specification the product is pinned to.
Reference anchors: src/nova/proof.rs:403-451 (test_nova_proof: 5 iterations per step, 3 steps, x random, y = 0, i = 1;
verify returns true against the initial state), :386 (zi_secondary == [0])."""
import copy
import random

import pytest

from oracle import nova as nv, pasta as o, poseidon as ps


def sat(cs, W=None):
    sh = cs.shape()
    return o.is_sat_relaxed(sh, cs.W if W is None else W, [0] * sh.num_cons, 1, cs.X, cs.m)


@pytest.mark.parametrize("field", [o.FIELD_FP, o.FIELD_FQ])
def test_internal_matrix_condition_and_permutation(field):
    assert ps.internal_matrix_ok(ps.MU[field], field)
    assert not ps.internal_matrix_ok((2, 3, 4, 6), field)          # the check does reject
    m = o.modulus(field)
    rc = ps.round_constants(field)
    assert len(rc) == ps.RF + ps.RP and sum(len(r) for r in rc) == ps.RF * ps.T + ps.RP
    assert all(0 <= v < m for r in rc for v in r)
    a, b = ps.permute([1, 2, 3, 4], field), ps.permute([1, 2, 3, 5], field)
    assert a != b and len(set(a)) == 4
    # the sponge separates tags and lengths
    assert len({ps.hash_elements(1, [7, 8], field), ps.hash_elements(2, [7, 8], field), ps.hash_elements(1, [7, 8, 0], field)}) == 3


@pytest.mark.parametrize("field", [o.FIELD_FP, o.FIELD_FQ])
def test_poseidon_gadget_equals_native(field):
    cs = nv.CS(field)
    xs = [cs.alloc(v) for v in (5, 0, o.modulus(field) - 1, 1 << 200, 77)]
    h = nv.poseidon_hash(cs, 9, xs)
    assert h.v == ps.hash_elements(9, [x.v for x in xs], field)
    assert sat(cs) and cs.rows == 2 * (ps.RF * ps.T + ps.RP) * 3


@pytest.mark.parametrize("field", [o.FIELD_FP, o.FIELD_FQ])
def test_strict_bits_accepts_only_the_canonical_representative(field):
    m = o.modulus(field)
    c = m - (1 << 254)
    for v in (0, 1, c - 1, c, (1 << 126) - 1, 1 << 126, (1 << 254) - 1, 1 << 254, (1 << 254) + c - 1, m - 1):
        cs = nv.CS(field)
        a = cs.alloc(v)
        bits = nv.strict_bits(cs, a)
        assert sat(cs) and sum(b.v << k for k, b in enumerate(bits)) == v
    # v + m fits 255 bits for small v: the non-canonical bit pattern must violate a constraint however the helper
    # variables are chosen honestly for it
    v = 12345
    cs = nv.CS(field)
    a = cs.alloc(v)
    nv.strict_bits(cs, a)
    W = list(cs.W)
    alt = v + m
    for k in range(255):
        W[1 + k] = (alt >> k) & 1
    assert not sat(cs, W)
    # recompute the helpers for the alternative pattern: still rejected (bit 254 set and low bits >= c)
    cs2 = nv.CS(field)
    a2 = cs2.alloc(v)
    a2.v = alt                                   # make the gadget derive every helper from the alternative integer
    nv.strict_bits(cs2, a2)
    assert not sat(cs2)


def rand_point(rng, curve):
    bm, sm = o.curve_base_modulus(curve), o.curve_scalar_modulus(curve)
    return o.pt_mul(rng.randrange(1, sm), o.generator(curve), bm)


@pytest.mark.parametrize("side", [0, 1])
def test_curve_gadgets(side):
    """The circuit of `side` does arithmetic on the OTHER side's curve (its coordinates are native)."""
    rng = random.Random(5 + side)
    field, curve = nv.SIDE_FIELD[side], nv.SIDE_CURVE[1 - side]
    bm = o.curve_base_modulus(curve)
    assert bm == o.modulus(field)
    P, Q = rand_point(rng, curve), rand_point(rng, curve)
    neg = lambda p: (p[0], (-p[1]) % bm)
    enc = lambda p: (0, 0) if p is None else p
    for a, b in ((P, Q), (P, P), (P, neg(P)), (None, Q), (P, None), (None, None)):
        cs = nv.CS(field)
        ax, ay = (cs.alloc(v) for v in enc(a))
        bx, by = (cs.alloc(v) for v in enc(b))
        x3, y3 = nv.ec_add_complete(cs, ax, ay, bx, by)
        assert (x3.v, y3.v) == enc(o.pt_add(a, b, bm)) and sat(cs)
    for r, pt in ((0, P), (1, P), (2, P), (rng.getrandbits(128), P), ((1 << 128) - 1, Q), (rng.getrandbits(128), None), (1 << 127, Q)):
        cs = nv.CS(field)
        bits = nv.alloc_bits(cs, r, 128)
        px, py = (cs.alloc(v) for v in enc(pt))
        inf = nv.is_zero(cs, px)
        nv.check_on_curve(cs, px, py, inf)
        rx, ry = nv.ec_scalar_mul(cs, bits, px, py, inf)
        assert (rx.v, ry.v) == enc(o.pt_mul(r, pt, bm)) and sat(cs)
    # a point off the curve fails the curve check
    cs = nv.CS(field)
    px, py = cs.alloc(P[0]), cs.alloc((P[1] + 1) % bm)
    nv.check_on_curve(cs, px, py, nv.is_zero(cs, px))
    assert not sat(cs)


@pytest.mark.parametrize("side", [0, 1])
def test_foreign_fold(side):
    rng = random.Random(17 + side)
    field = nv.SIDE_FIELD[side]
    pf = o.modulus(nv.SIDE_FIELD[1 - side])
    for A, B, r in ((0, 0, 0), (pf - 1, (1 << 250) - 1, (1 << 128) - 1), (rng.randrange(pf), rng.getrandbits(250), rng.getrandbits(128)),
                    (pf - 1, 1, 1), (1, (1 << 250) - 1, 0)):
        cs = nv.CS(field)
        lo, hi = nv.split126(A)
        a_lo, a_hi = cs.alloc(lo), cs.alloc(hi)
        bb, rb = nv.alloc_bits(cs, B, 250), nv.alloc_bits(cs, r, 128)
        R_lo, R_hi = nv.fold_foreign(cs, a_lo, a_hi, bb, rb, pf)
        assert R_lo.v + (R_hi.v << 126) == (A + r * B) % pf and sat(cs)
        # a wrong remainder is rejected: flip the lowest bit of R (the first of the 126 + 129 remainder bits)
        W = list(cs.W)
        first_r = 2 + 250 + 128 + 125
        W[first_r] ^= 1
        assert not sat(cs, W)


def chain(t, n, x0=0x1234):
    init = o.State(x0, 0, 1)                      # y = 0, i = 1 as src/nova/proof.rs:417-421
    states = [init]
    for _ in range(n):
        states.append(o.minroot_eval(states[-1], t, o.FIELD_FQ))
    return init, states


@pytest.fixture(scope="module")
def proof_5_3():
    t, n = 5, 3
    pp = nv.public_params(t, nv.CCommit(), nv.GENS_SEED, nv.FAMILY_TRY_AND_INCREMENT)
    init, states = chain(t, n)
    z0 = [states[n].x, states[n].y, states[n].i]
    s = None
    snaps = []
    for k in range(n):
        s = nv.prove_step(pp, s, nv.InverseMinRootCircuit(t, states[n - k], states[n - k - 1]), z0)
        snaps.append(copy.deepcopy(s))
    return pp, init, z0, snaps


def test_nova_proof_5_iterations_3_steps(proof_5_3):
    """test_nova_proof (src/nova/proof.rs:403-451)."""
    pp, init, z0, snaps = proof_5_3
    assert nv.verify(pp, snaps[2], 3, z0) == ([init.x, init.y, init.i], [0])     # :386: zi_secondary == [0]
    # every prefix is itself a valid proof of fewer steps
    for k in (0, 1):
        zi = nv.verify(pp, snaps[k], k + 1, z0)
        assert zi is not None and zi[1] == [0]
    assert nv.verify(pp, snaps[2], 2, z0) is None                                 # wrong step count
    assert nv.verify(pp, snaps[2], 3, [z0[0], z0[1], (z0[2] + 1) % o.Q]) is None   # other z0
    # shapes: the reference's step circuit contributes 3t + 1 constraints and 4t + 1 variables (src/nova/proof.rs:155-230, :122-133);
    # the bound form 3t + 1 and 3t + 1
    pp1 = nv.public_params(1, pp.commit, nv.GENS_SEED, 1)
    assert pp.shapes[0].num_cons - pp1.shapes[0].num_cons == 3 * 4 and pp.shapes[0].num_vars - pp1.shapes[0].num_vars == 4 * 4
    ppb, ppb1 = (nv.public_params(t_, pp.commit, nv.GENS_SEED, 1, bound=True) for t_ in (5, 1))
    assert ppb.shapes[0].num_cons - ppb1.shapes[0].num_cons == 3 * 4 and ppb.shapes[0].num_vars - ppb1.shapes[0].num_vars == 3 * 4


def test_tampered_proofs_are_rejected(proof_5_3):
    pp, init, z0, snaps = proof_5_3
    good = snaps[2]
    muts = []
    s = copy.deepcopy(good); s.r[0].W[7] = (s.r[0].W[7] + 1) % o.Q; muts.append(s)
    s = copy.deepcopy(good); s.r[1].E[3] = (s.r[1].E[3] + 1) % o.P; muts.append(s)
    s = copy.deepcopy(good); s.r[0].u = (s.r[0].u + 1) % o.Q; muts.append(s)
    s = copy.deepcopy(good); s.r[1].X[0] ^= 1; muts.append(s)
    s = copy.deepcopy(good); s.l2.X[1] ^= 1; muts.append(s)
    s = copy.deepcopy(good); s.l2.W[100] = (s.l2.W[100] + 1) % o.P; muts.append(s)
    s = copy.deepcopy(good); s.zi[0][0] = (s.zi[0][0] + 1) % o.Q; muts.append(s)
    s = copy.deepcopy(good); s.zi[1][0] = 1; muts.append(s)
    s = copy.deepcopy(good); s.r[0].comm_E = s.r[0].comm_W; muts.append(s)
    for s in muts:
        assert nv.verify(pp, s, 3, z0) is None


def test_reference_circuit_has_the_unbound_new_x_and_the_bound_form_does_not():
    """src/nova/proof.rs:167-173 allocates new_x, :219-227 never uses it: in the reference's shape a prover may set
    every new_x freely (here: round 0's), which changes the step's output x.  The bound form has no such variable."""
    t = 4
    init, states = chain(t, 1)
    res, inp = states[1], states[0]
    for bound in (False, True):
        cs = nv.CS(o.FIELD_FQ)
        z = [cs.alloc(v) for v in (res.x, res.y, res.i)]
        out = nv.InverseMinRootCircuit(t, res, inp, bound).synthesize(cs, z)
        assert [n.v for n in out] == [inp.x, inp.y, inp.i] and sat(cs)
        assert len(cs.W) == 3 + (3 if bound else 4) * t + 1 and cs.rows == 3 * t + 1
        if not bound:
            # forge: pick new_x of round 0, then solve the remaining rounds honestly from it
            W = list(cs.W)
            m = o.Q
            x, y, i = (res.x * 0 + 99991), W[3 + 3], (res.i - 1) % m       # new_x := 99991, y = round 0's new_y
            W[3] = x
            for j in range(1, t):
                b = 3 + 4 * j
                nx = (y - i + 1) % m
                t1 = x * x % m; t2 = t1 * t1 % m
                W[b], W[b + 1], W[b + 2], W[b + 3] = nx, t1, t2, (t2 * x - nx) % m
                x, y, i = nx, W[b + 3], (i - 1) % m
            assert sat(cs, W) and W[3 + 4 * (t - 1)] != inp.x


def test_known_dlog_family_commitments_match_the_identity():
    com = nv.CCommit(nv.FAMILY_KNOWN_DLOG, 7)
    v = [3, 0, 1 << 200, o.Q - 1, 5]
    exp = o.msm_by_dlog(v, o.CURVE_PALLAS, 7)
    assert com(0, v) == exp
