"""CPU: the compression SNARK restatement (oracle/spartan.py) is complete and rejects tampering, on a relaxed
instance obtained by folding two MinRoot step instances at t = 3 (the shape the product proves at t = 2^16)."""
import copy

import pytest

from oracle import pasta as o
from oracle import spartan as sp

Q = o.Q


def _fresh(shape, t, x0, i0):
    st = o.State(x0 % Q, 0, i0)
    res = o.minroot_eval(st, t, o.FIELD_FQ)
    W = [res.x, res.y, res.i] + o.step_witness_segment(res, t, o.FIELD_FQ)
    X = [res.x, res.y, res.i, st.x, st.y, st.i]
    assert o.is_sat_relaxed(shape, W, [0] * shape.num_cons, 1, X, Q)
    return W, X


@pytest.fixture(scope="module")
def relaxed():
    t = 6                                    # 32 padded variables / constraints: a halving round, then the 16-vector
    shape = o.step_circuit_shape(t, o.FIELD_FQ)
    N = 1
    while N < max(shape.num_vars, shape.num_cons):
        N <<= 1
    G = o.synthetic_bases(o.CURVE_PALLAS, 0x4E6F7661, N)
    U = o.synthetic_bases(o.CURVE_PALLAS, 0x4E6F7661, 1, start=N)[0]
    W1, X1 = _fresh(shape, t, 123456789, 5)
    W2, X2 = _fresh(shape, t, 987654321, 9)
    z1, z2 = W1 + [1] + X1, W2 + [1] + X2
    abc1, abc2 = o.multiply_vec(shape, z1, Q), o.multiply_vec(shape, z2, Q)
    T = o.cross_term(*abc1, *abc2, 1, Q)
    r = 0x1234567890ABCDEF1234567890ABCDEF
    W = o.axpy(W1, r, W2, Q)
    E = o.axpy([0] * shape.num_cons, r, T, Q)
    u = (1 + r) % Q
    X = o.axpy(X1, r, X2, Q)
    assert o.is_sat_relaxed(shape, W, E, u, X, Q)
    comm_W = o.msm_naive(W, G[:len(W)], o.CURVE_PALLAS)
    comm_E = o.msm_naive(E, G[:len(E)], o.CURVE_PALLAS)
    return dict(shape=shape, G=G, U=U, W=W, E=E, u=u, X=X, comm_W=comm_W, comm_E=comm_E, digest=b"\x07" * 32)


def _prove(r):
    return sp.prove(r["shape"], r["digest"], r["G"], r["U"], r["comm_W"], r["comm_E"], r["u"], r["X"], r["W"], r["E"])


def _verify(r, proof, **over):
    a = dict(r, **over)
    return sp.verify(a["shape"], a["digest"], a["G"], a["U"], a["comm_W"], a["comm_E"], a["u"], a["X"], proof)


def test_multilinear_helpers():
    r = [3, 5, 7]
    eq = sp.eq_table(r, Q)
    assert sum(eq) % Q == 1
    # eq(r, x) at the corner x = (1, 0, 1): index 0b101, x_1 = most significant bit
    assert eq[0b101] == 3 * (1 - 5) * 7 % Q
    f = [o.rand_fe(1, i, Q) for i in range(8)]
    assert sp.mle_eval(f, [1, 0, 1], Q) == f[0b101]
    assert sp.mle_eval(f, r, Q) == sum(a * b for a, b in zip(f, eq)) % Q


def test_interpolate_is_exact_on_a_cubic():
    g = lambda t: (2 * t * t * t + 3 * t * t + 5 * t + 7) % Q
    pts = [(k, g(k)) for k in range(4)]
    for r in (4, 10, Q - 3):
        assert sp.interpolate(pts, r, Q) == g(r)


def test_complete_and_sound_against_tampering(relaxed):
    proof = _prove(relaxed)
    assert _verify(relaxed, proof)
    # the same inputs give the same proof (deterministic transcript)
    assert _prove(relaxed).outer == proof.outer
    bad = copy.deepcopy(proof); bad.outer[1][0] = (bad.outer[1][0] + 1) % Q
    assert not _verify(relaxed, bad)
    bad = copy.deepcopy(proof); bad.claims = (proof.claims[0], proof.claims[1], (proof.claims[2] + 1) % Q, proof.claims[3])
    assert not _verify(relaxed, bad)
    bad = copy.deepcopy(proof); bad.inner[0][1] = (bad.inner[0][1] + 1) % Q
    assert not _verify(relaxed, bad)
    bad = copy.deepcopy(proof); bad.w_eval = (bad.w_eval + 1) % Q
    assert not _verify(relaxed, bad)
    assert len(proof.ipa_W.L) == 1 and len(proof.ipa_W.a) == sp.IPA_STOP == len(proof.ipa_E.a)
    for k in (0, 7, 15):
        bad = copy.deepcopy(proof); bad.ipa_W.a[k] = (bad.ipa_W.a[k] + 1) % Q
        assert not _verify(relaxed, bad)
    bad = copy.deepcopy(proof); bad.ipa_E.a = bad.ipa_E.a[:-1]
    assert not _verify(relaxed, bad)
    bad = copy.deepcopy(proof); bad.ipa_E.L[0] = relaxed["U"]
    assert not _verify(relaxed, bad)
    bad = copy.deepcopy(proof); bad.ipa_E.L.append(relaxed["U"]); bad.ipa_E.R.append(relaxed["U"])     # one round too many
    assert not _verify(relaxed, bad)
    # a different instance: u, X, commitments
    assert not _verify(relaxed, proof, u=(relaxed["u"] + 1) % Q)
    assert not _verify(relaxed, proof, X=[(relaxed["X"][0] + 1) % Q] + relaxed["X"][1:])
    assert not _verify(relaxed, proof, comm_E=relaxed["comm_W"])
    assert not _verify(relaxed, proof, digest=b"\x08" * 32)


def test_unsatisfied_witness_cannot_be_proved(relaxed):
    r = dict(relaxed)
    r["W"] = list(r["W"]); r["W"][4] = (r["W"][4] + 1) % Q
    r["comm_W"] = o.msm_naive(r["W"], r["G"][:len(r["W"])], o.CURVE_PALLAS)
    assert not o.is_sat_relaxed(r["shape"], r["W"], r["E"], r["u"], r["X"], Q)
    assert not _verify(r, _prove(r))


def test_against_the_committed_golden_argument():
    """tests/golden/vectors.json pins the restatement (generator family, transcript, encoding) against regressions."""
    import json, os
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "vectors.json")))
    for c in (o.CURVE_PALLAS, o.CURVE_VESTA):
        assert [[("%064x" % p[0]), ("%064x" % p[1])] for p in o.tai_bases(c, 7, 4)] == g["tai_bases_seed7"][str(c)]
    t = 6
    shape = o.step_circuit_shape(t, o.FIELD_FQ)
    W1, X1 = _fresh(shape, t, 123456789, 5)
    W2, X2 = _fresh(shape, t, 987654321, 9)
    abc1, abc2 = o.multiply_vec(shape, W1 + [1] + X1, Q), o.multiply_vec(shape, W2 + [1] + X2, Q)
    T = o.cross_term(*abc1, *abc2, 1, Q)
    r = 0x1234567890ABCDEF1234567890ABCDEF
    W, E, u, X = o.axpy(W1, r, W2, Q), [r * v % Q for v in T], (1 + r) % Q, o.axpy(X1, r, X2, Q)
    G = o.tai_bases(o.CURVE_PALLAS, 0x4E6F7661, 32)
    U = o.tai_base(o.CURVE_PALLAS, 0x4E6F7661, 32)
    cW, cE = o.msm_naive(W, G[:len(W)], o.CURVE_PALLAS), o.msm_naive(E, G[:len(E)], o.CURVE_PALLAS)
    gs = g["spartan_t6"]
    assert ["%064x" % cW[0], "%064x" % cW[1]] == gs["comm_W"] and "%064x" % u == gs["u"]
    pf = sp.prove(shape, b"\x07" * 32, G, U, cW, cE, u, X, W, E)
    fe = lambda v: int(v).to_bytes(32, "little")
    pt = lambda q: b"\0" * 64 if q is None else fe(q[0]) + fe(q[1])
    enc = b"".join(fe(v) for ev in pf.outer for v in ev) + b"".join(fe(v) for v in pf.claims)
    enc += b"".join(fe(v) for ev in pf.inner for v in ev) + fe(pf.w_eval)
    for ipa in (pf.ipa_W, pf.ipa_E):
        enc += b"".join(pt(L) + pt(R) for L, R in zip(ipa.L, ipa.R)) + b"".join(fe(v) for v in ipa.a)
        assert len(ipa.L) == 1 and len(ipa.a) == sp.IPA_STOP
    assert enc.hex() == gs["argument_hex"]
