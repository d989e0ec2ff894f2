"""CPU tests: the oracle against everything the reference's own tests pin for this path
(SURVEY.md section 4 / 8c), against the committed golden vectors, and C restatement vs Python."""
import numpy as np
import pytest

from oracle import pasta as o
from util import limbs, ints, mont, unmont, hexes, jac_to_affine, affine_array


# ---- constants: src/minroot.rs:273-285 ------------------------------------------------------
def test_exponent_constants():
    assert 5 * o.FP_RESCUE_INVALPHA % (o.P - 1) == 1
    assert 5 * o.FQ_RESCUE_INVALPHA % (o.Q - 1) == 1
    # the cross pairing fails: PallasVDF works in Fq, VestaVDF in Fp (src/minroot.rs:38, :199)
    assert 5 * o.FP_RESCUE_INVALPHA % (o.Q - 1) != 1


# ---- test_exponents, src/minroot.rs:449-458 ----------------------------------------------------
@pytest.mark.parametrize("field", [o.FIELD_FP, o.FIELD_FQ])
def test_inverse_exponent_is_five(field):
    m = o.modulus(field)
    x = o.rand_fe(1, 0, m)
    assert o.inverse_step(x, field) == pow(x, 5, m)


# ---- test_steps, src/minroot.rs:460-477: 100 random x, both curves, default mode ---------------
@pytest.mark.parametrize("field", [o.FIELD_FP, o.FIELD_FQ])
def test_steps(field):
    m = o.modulus(field)
    for i in range(100):
        x = o.rand_fe(42, i, m)
        assert o.inverse_step(o.forward_step(x, field), field) == x


# ---- test_eval, src/minroot.rs:479-510: Pallas, all four modes, 10 random (x, y), i = 0, t = 10 --
@pytest.mark.parametrize("mode", o.EVAL_MODES)
def test_eval_roundtrip(mode):
    for k in range(10):
        s = o.State(o.rand_fe(42, 2 * k, o.Q), o.rand_fe(42, 2 * k + 1, o.Q), 0)
        r = o.minroot_eval(s, 10, o.FIELD_FQ, mode)
        assert o.minroot_inverse_eval(r, 10, o.FIELD_FQ) == s
        assert o.minroot_check(r, 10, s, o.FIELD_FQ)
        assert r == o.minroot_eval(s, 10, o.FIELD_FQ, "LTRSequential")


# ---- test_vanilla_proof, src/minroot.rs:512-542: t = 4, n = 3, both curves ----------------------
@pytest.mark.parametrize("field", [o.FIELD_FP, o.FIELD_FQ])
def test_vanilla_chain(field):
    m = o.modulus(field)
    s0 = o.State(o.rand_fe(42, 0, m), 0, 0)
    s, t, n = s0, 4, 3
    for _ in range(n):
        nxt = o.minroot_eval(s, t, field)
        assert o.minroot_check(nxt, t, s, field)      # Evaluation::append verifies the second proof
        s = nxt
    assert s.i == n * t                               # V::element(final_proof.t) == result.i
    assert o.minroot_check(s, n * t, s0, field)


# ---- circuit: src/nova/proof.rs:155-230, sizes of SURVEY.md Appendix B ---------------------------
@pytest.mark.parametrize("t", [5, 10, 1000])
def test_step_circuit_shape_sizes(t):
    sh = o.step_circuit_shape(t, o.FIELD_FQ)
    assert sh.num_vars == 3 + 4 * t + 1
    assert sh.num_cons == 3 * t + 1 + 6
    # per-round rows: A, B have one entry each; C has 1, 1 and 4 entries
    assert len(sh.A) == 3 * t + 1 + 6 and len(sh.B) == 3 * t + 1 + 6
    assert len(sh.C) == 6 * t + 2 + 6


def test_step_circuit_satisfied_and_tamper_rejected():
    t = 5
    s0 = o.State(o.rand_fe(42, 0, o.Q), 0, 1)          # test_nova_proof: y = 0, i = 1 (proof.rs:417-421)
    tr = o.minroot_eval_trace(s0, t, o.FIELD_FQ)
    res = tr[-1]
    seg = o.step_witness_segment(res, t, o.FIELD_FQ)
    assert seg == o.step_witness_from_trace([(s.x, s.y) for s in tr], s0.i, t, o.FIELD_FQ)
    sh = o.step_circuit_shape(t, o.FIELD_FQ)
    W = [res.x, res.y, res.i] + seg
    X = [res.x, res.y, res.i, s0.x, s0.y, s0.i]
    assert o.is_sat_relaxed(sh, W, [0] * sh.num_cons, 1, X, o.Q)
    bad = list(W); bad[5] = (bad[5] + 1) % o.Q            # tmp2 of round 0
    assert not o.is_sat_relaxed(sh, bad, [0] * sh.num_cons, 1, X, o.Q)
    # new_x is never tied to y - new_i by a constraint of its own, but it is the
    # next round's x, so tampering it alone still breaks that round's x*x = tmp1
    bad = list(W); bad[3] = (bad[3] + 1) % o.Q            # new_x of round 0
    assert not o.is_sat_relaxed(sh, bad, [0] * sh.num_cons, 1, X, o.Q)


def test_folding_preserves_satisfiability():
    """NIFS algebra: folding two satisfied instances with the cross term
    yields a satisfied relaxed instance."""
    t, m = 3, o.Q
    sh = o.step_circuit_shape(t, o.FIELD_FQ)

    def instance(seed):
        s0 = o.State(o.rand_fe(seed, 0, m), o.rand_fe(seed, 1, m), 7)
        tr = o.minroot_eval_trace(s0, t, o.FIELD_FQ)
        res = tr[-1]
        W = [res.x, res.y, res.i] + o.step_witness_segment(res, t, o.FIELD_FQ)
        X = [res.x, res.y, res.i, s0.x, s0.y, s0.i]
        return W, X

    W1, X1 = instance(1)
    W2, X2 = instance(2)
    z1, z2 = W1 + [1] + X1, W2 + [1] + X2
    a1, b1, c1 = o.multiply_vec(sh, z1, m)
    a2, b2, c2 = o.multiply_vec(sh, z2, m)
    T = o.cross_term(a1, b1, c1, a2, b2, c2, 1, m)
    r = o.rand_fe(3, 0, m) >> 126
    W = o.axpy(W1, r, W2, m)
    E = o.axpy([0] * sh.num_cons, r, T, m)
    X = o.axpy(X1, r, X2, m)
    assert o.is_sat_relaxed(sh, W, E, (1 + r) % m, X, m)


# ---- curves -----------------------------------------------------------------------------------------
@pytest.mark.parametrize("curve", [o.CURVE_PALLAS, o.CURVE_VESTA])
def test_curve_order_and_generator(curve):
    g = o.generator(curve)
    assert o.on_curve(g, curve)
    assert o.pt_mul(o.curve_scalar_modulus(curve), g, o.curve_base_modulus(curve)) is None


# ---- golden vectors -----------------------------------------------------------------------------------
def test_golden_survey_appendix_b(golden):
    """SURVEY.md Appendix B known answers (surveyor-derived), PallasVDF t = 1, 10 and circuit round 0."""
    g = golden["minroot_eval_123_321_0"][str(o.FIELD_FQ)]
    assert g["1"][0] == "3f3bcbd00f12f95e040a87ddc7f03f29ad2b4ac49fa84e909a7728bf07b98819"
    assert g["10"][0] == "29c3861ffe89dfc67e84472737766fdc2e4e353529dbd68515d576c958c6a5c6"
    assert golden["minroot_eval_123_321_0"][str(o.FIELD_FP)]["10"][0] == \
        "082724179fb830c8c27cf88a9cd8ff0a2bbcf818d222a657bf43ad1d7a749a6e"
    assert golden["circuit_round0_on_pallas_t10"][3] == \
        "0bc588bd21ddf28369626b23f23f7e1198b9e05d9b1e7270c6ad762094df75b0"


def test_golden_regenerates(golden):
    for f in (o.FIELD_FP, o.FIELD_FQ):
        for t, row in golden["minroot_eval_123_321_0"][str(f)].items():
            r = o.minroot_eval(o.State(123, 321, 0), int(t), f)
            assert [r.x, r.y, r.i] == hexes(row)
    for curve, cases in golden["msm_seed7"].items():
        for n, c in cases.items():
            pts = [tuple(hexes(p)) for p in c["bases"]]
            res = o.msm_naive(hexes(c["scalars"]), pts, int(curve))
            assert list(o.point_to_affine_ints(res)) == hexes(c["result"])


# ---- C restatement vs Python oracle and golden vectors ---------------------------------------------------
@pytest.mark.parametrize("field", [o.FIELD_FP, o.FIELD_FQ])
def test_c_field_mul_golden(cref, golden, field):
    L, m = cref.lib(), o.modulus(field)
    g = golden["field_mul"][str(field)]
    a, b = mont(hexes(g["a"]), m), mont(hexes(g["b"]), m)
    out = cref.fe_array(len(a))
    L.ref_fe_mul(field, cref.p(a), cref.p(b), len(a), cref.p(out))
    assert unmont(out, m) == hexes(g["mul"])


@pytest.mark.parametrize("mode", range(4))
def test_c_minroot_modes(cref, golden, mode):
    L = cref.lib()
    st = mont([123, 321, 0], o.Q)
    so, tr = cref.fe_array(3), cref.fe_array(22)
    L.ref_minroot_eval(o.FIELD_FQ, mode, cref.p(st), 10, cref.p(so), cref.p(tr))
    assert unmont(so, o.Q) == hexes(golden["minroot_eval_123_321_0"][str(o.FIELD_FQ)]["10"])
    back = cref.fe_array(3)
    L.ref_minroot_inverse_eval(o.FIELD_FQ, cref.p(so), 10, cref.p(back))
    assert unmont(back, o.Q) == [123, 321, 0]
    exp = o.minroot_eval_trace(o.State(123, 321, 0), 10, o.FIELD_FQ)
    assert unmont(tr, o.Q) == [v for s in exp for v in (s.x, s.y)]


def test_c_minroot_vesta(cref, golden):
    L = cref.lib()
    st, so = mont([123, 321, 0], o.P), cref.fe_array(3)
    L.ref_minroot_eval(o.FIELD_FP, 0, cref.p(st), 10, cref.p(so), None)
    assert unmont(so, o.P) == hexes(golden["minroot_eval_123_321_0"][str(o.FIELD_FP)]["10"])


def test_c_step_witness_golden(cref, golden):
    L = cref.lib()
    g = golden["witness_t5"]
    res = mont(hexes(g["trace_xy"][-1]) + [1 + 5], o.Q)
    W = cref.fe_array(21)
    L.ref_step_witness(o.FIELD_FQ, cref.p(res), 5, cref.p(W))
    assert unmont(W, o.Q) == hexes(g["W"])


def test_c_fold_ops_golden(cref, golden):
    L, m, g = cref.lib(), o.Q, golden["fold_t5"]
    sh = g["shape"]
    z1, z2 = mont(hexes(g["z1"]), m), mont(hexes(g["z2"]), m)
    outs = {}
    for name, key in (("A", "az"), ("B", "bz"), ("C", "cz")):
        rows = np.array([e[0] for e in sh[name]], dtype=np.uint32)
        cols = np.array([e[1] for e in sh[name]], dtype=np.uint32)
        vals = mont([int(e[2], 16) for e in sh[name]], m)
        for tag, z in (("1", z1), ("2", z2)):
            out = cref.fe_array(sh["num_cons"])
            L.ref_spmv(o.FIELD_FQ, cref.p(rows), cref.p(cols), cref.p(vals), len(rows), cref.p(z), sh["num_cons"], cref.p(out))
            assert unmont(out, m) == hexes(g[key + tag])
            outs[key + tag] = out
    T = cref.fe_array(sh["num_cons"])
    u1 = mont([hexes(g["z1"])[sh["num_vars"]]], m)
    L.ref_cross_term(o.FIELD_FQ, *(cref.p(outs[k]) for k in ("az1", "bz1", "cz1", "az2", "bz2", "cz2")), cref.p(u1),
                     sh["num_cons"], cref.p(T))
    assert unmont(T, m) == hexes(g["T"])
    nv = sh["num_vars"]
    Wf = cref.fe_array(nv)
    a, rr, b = z1[:nv].copy(), mont([int(g["r"], 16)], m), z2[:nv].copy()     # keep the buffers alive
    L.ref_axpy(o.FIELD_FQ, cref.p(a), cref.p(rr), cref.p(b), nv, cref.p(Wf))
    assert unmont(Wf, m) == hexes(g["W_fold"])


@pytest.mark.parametrize("curve", [o.CURVE_PALLAS, o.CURVE_VESTA])
def test_c_msm_golden_and_dlog(cref, golden, curve):
    L = cref.lib()
    sm = o.curve_scalar_modulus(curve)
    for n, c in golden["msm_seed7"][str(curve)].items():
        n = int(n)
        pts = np.zeros((n, 8), dtype="<u8")
        L.ref_synthetic_bases(curve, 7, 0, n, cref.p(pts))
        assert np.array_equal(pts, affine_array([tuple(hexes(p)) for p in c["bases"]], curve))
        for is_mont in (0, 1):
            sc = mont(hexes(c["scalars"]), sm) if is_mont else limbs(hexes(c["scalars"]))
            for fn in ("pip", "naive"):
                out = np.zeros(12, dtype="<u8")
                if fn == "pip":
                    L.ref_msm(curve, cref.p(pts), cref.p(sc), n, is_mont, 2, 0, cref.p(out))
                else:
                    L.ref_msm_naive(curve, cref.p(pts), cref.p(sc), n, is_mont, cref.p(out))
                got = jac_to_affine(out, curve)
                exp = tuple(hexes(c["result"]))
                assert (got or (0, 0)) == exp
    # a larger one through the discrete-log identity, several window sizes
    n = 3000
    pts = np.zeros((n, 8), dtype="<u8")
    L.ref_synthetic_bases(curve, 9, 0, n, cref.p(pts))
    assert L.ref_count_off_curve(curve, cref.p(pts), n) == 0
    sc = [o.rand_fe(5, i, sm) for i in range(n)]
    exp = o.msm_by_dlog(sc, curve, 9)
    for c_bits in (0, 4, 9, 16):
        out = np.zeros(12, dtype="<u8")
        sc_arr = limbs(sc)
        L.ref_msm(curve, cref.p(pts), cref.p(sc_arr), n, 0, 4, c_bits, cref.p(out))
        assert jac_to_affine(out, curve) == exp


def test_try_and_increment_generators_are_curve_points_with_even_y():
    """Generator family 1: on the curve, even y, deterministic per (seed, index)."""
    for curve in (o.CURVE_PALLAS, o.CURVE_VESTA):
        m = o.curve_base_modulus(curve)
        pts = o.tai_bases(curve, 7, 40, start=1000)
        assert len(set(pts)) == 40
        for x, y in pts:
            assert (y * y - x * x * x - 5) % m == 0 and y % 2 == 0 and 0 <= x < m
        assert o.tai_base(curve, 7, 1003) == pts[3]
        assert o.tai_base(curve, 8, 1003) != pts[3]
    for a in (0, 1, 4, 5, 12345678901234567890):
        r = o.sqrt_mod(a * a, o.P)
        assert r in (a % o.P, (-a) % o.P)


def test_vectorised_dlog_identity_equals_the_plain_one():
    """msm_by_dlog_limbs (what the 2^24 GPU test is checked against) is the same function as msm_by_dlog."""
    import numpy as np
    from util import rand_limbs, ints
    rng = np.random.default_rng(11)
    for curve, seed, start, n in ((o.CURVE_PALLAS, 7, 0, 1000), (o.CURVE_VESTA, 3, 12345, 257), (o.CURVE_PALLAS, 9, 1 << 23, 1)):
        sc = rand_limbs(rng, n)
        assert o.msm_by_dlog_limbs(sc, curve, seed, start) == o.msm_by_dlog(ints(sc), curve, seed, start)


@pytest.mark.parametrize("t", [1, 2, 5, 17])
def test_packed_commitment_identity_of_the_reference_rounds(t):
    """The identity behind vdf_minroot_step_segment_packed (include/vdf_hip.h) and libvdf_nova.so's make_packed_generators,
    in the oracle's own integers: the reference allocates new_x in every round (src/nova/proof.rs:167-173) although
    new_x_j = y_j - (i_in - (j + 1)) with y_j = new_y_(j-1) (:162-173), so the Pedersen commitment to the 4t + 1 round
    variables equals an MSM of 3t + 4 scalars [tmp1, tmp2, new_y per round | final_i | y_0 | i_in | 1] over the derived
    generators [G[4j+1], G[4j+2], G[4j+3] + G[4j+4] (last round: G[4t-1]) | G[4t] | G[0] | -S1 | S2]."""
    field, curve, m, bm = o.FIELD_FQ, o.CURVE_PALLAS, o.Q, o.P
    init = o.State(o.rand_fe(77, 0, m), 0, 5)
    result = o.minroot_eval(init, t, field)
    W = o.step_witness_segment(result, t, field)                 # new_x, tmp1, tmp2, new_y per round, then final_i
    assert len(W) == 4 * t + 1 and W[4 * t] == init.i
    G = o.tai_bases(curve, 9, 4 * t + 1)
    want = o.msm_naive(W, G, curve)
    add = lambda a, b: o.pt_add(a, b, bm)
    packed, D = [], []
    for j in range(t):
        packed += [W[4 * j + 1], W[4 * j + 2], W[4 * j + 3]]
        D += [G[4 * j + 1], G[4 * j + 2], add(G[4 * j + 3], G[4 * j + 4]) if j + 1 < t else G[4 * t - 1]]
    s1 = s2 = None
    for j in range(t - 1, -1, -1):
        s1 = add(s1, G[4 * j]); s2 = add(s2, s1)
    packed += [W[4 * t], result.y, result.i, 1]                  # final_i, the y and the i the first round reads, one
    D += [G[4 * t], G[0], o.pt_neg(s1, bm), s2]
    assert o.msm_naive(packed, D, curve) == want
    for j in range(t):                                           # the relation itself
        y_j = result.y if j == 0 else W[4 * (j - 1) + 3]
        assert W[4 * j] == (y_j - (result.i - (j + 1))) % m
