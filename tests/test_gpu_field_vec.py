"""GPU parity: field multiply / Montgomery conversion and the folding vector kernels, through
the C ABI, bit-exact against the golden vectors and the C restatement."""
import numpy as np
import pytest

from oracle import pasta as o
from util import limbs, ints, mont, unmont, hexes, rand_limbs

pytestmark = pytest.mark.gpu
FIELDS = [o.FIELD_FP, o.FIELD_FQ]


@pytest.mark.parametrize("field", FIELDS)
def test_fe_mul_golden_and_edges(ctx, golden, field):
    m = o.modulus(field)
    g = golden["field_mul"][str(field)]
    a, b = mont(hexes(g["a"]), m), mont(hexes(g["b"]), m)
    out = np.zeros_like(a)
    ctx.fe_mul(field, a, b, len(a), out)
    assert unmont(out, m) == hexes(g["mul"])


@pytest.mark.parametrize("field", FIELDS)
def test_fe_mul_random_vs_c(ctx, cref, field):
    n = 1 << 18
    rng = np.random.default_rng(field + 1)
    a, b = rand_limbs(rng, n), rand_limbs(rng, n)
    a[0] = 0; b[1] = 0
    a[2] = limbs([o.modulus(field) - 1])[0]; b[2] = a[2]
    out, exp = np.zeros_like(a), np.zeros_like(a)
    ctx.fe_mul(field, a, b, n, out)
    cref.lib().ref_fe_mul(field, cref.p(a), cref.p(b), n, cref.p(exp))
    assert np.array_equal(out, exp)


@pytest.mark.parametrize("field", FIELDS)
def test_mont_roundtrip(ctx, cref, field):
    n = 10000
    a = rand_limbs(np.random.default_rng(3), n)
    am, back, exp = np.zeros_like(a), np.zeros_like(a), np.zeros_like(a)
    ctx.fe_to_mont(field, a, n, am)
    cref.lib().ref_fe_to_mont(field, cref.p(a), n, cref.p(exp))
    assert np.array_equal(am, exp)
    ctx.fe_from_mont(field, am, n, back)
    assert np.array_equal(back, a)


@pytest.mark.parametrize("field", FIELDS)
@pytest.mark.parametrize("n", [0, 1, 255, 256, 257, 100003])
def test_axpy_vs_c(ctx, cref, field, n):
    rng = np.random.default_rng(n + 7)
    a, b, r = rand_limbs(rng, n), rand_limbs(rng, n), rand_limbs(rng, 1)
    out, exp = np.zeros_like(a), np.zeros_like(a)
    ctx.axpy(field, a, r, b, n, out)
    cref.lib().ref_axpy(field, cref.p(a), cref.p(r), cref.p(b), n, cref.p(exp))
    assert np.array_equal(out, exp)


@pytest.mark.parametrize("field", FIELDS)
def test_axpy_in_place_on_device(ctx, cref, field):
    import torch
    n = 70001
    rng = np.random.default_rng(11)
    a, b, r = rand_limbs(rng, n), rand_limbs(rng, n), rand_limbs(rng, 1)
    exp = np.zeros_like(a)
    cref.lib().ref_axpy(field, cref.p(a), cref.p(r), cref.p(b), n, cref.p(exp))
    da, db, dr = (torch.from_numpy(x.view(np.int64)).cuda() for x in (a, b, r))
    ctx.axpy(field, da, dr, db, n, da)          # out aliases a, all buffers device-resident
    ctx.sync()
    assert np.array_equal(da.cpu().numpy().view("<u8"), exp)


@pytest.mark.parametrize("field", FIELDS)
@pytest.mark.parametrize("n", [1, 1000, 196609])
def test_cross_term_vs_c(ctx, cref, field, n):
    rng = np.random.default_rng(n)
    v = [rand_limbs(rng, n) for _ in range(6)]
    u1 = rand_limbs(rng, 1)
    out, exp = np.zeros_like(v[0]), np.zeros_like(v[0])
    ctx.cross_term(field, *v, u1, n, out)
    cref.lib().ref_cross_term(field, *(cref.p(x) for x in v), cref.p(u1), n, cref.p(exp))
    assert np.array_equal(out, exp)


def test_witness_golden_t5(ctx, golden):
    g = golden["witness_t5"]
    tr = mont([int(h, 16) for pair in g["trace_xy"] for h in pair], o.Q)
    i0 = mont([int(g["i0"], 16)], o.Q)
    W = np.zeros((21, 4), dtype="<u8")
    ctx.minroot_witness(o.FIELD_FQ, tr, i0, 5, W)
    assert unmont(W, o.Q) == hexes(g["W"])


@pytest.mark.parametrize("field", FIELDS)
@pytest.mark.parametrize("t", [1, 10, 1024, 65536])
def test_witness_vs_sequential_circuit(ctx, cref, field, t):
    """The round-parallel kernel must equal the circuit's own sequential computation
    (src/nova/proof.rs:107-126) on a real forward trace."""
    L, m = cref.lib(), o.modulus(field)
    st = mont([o.rand_fe(t, 0, m), 0, 1], m)
    so, tr = cref.fe_array(3), cref.fe_array(2 * (t + 1))
    L.ref_minroot_eval(field, 1, cref.p(st), t, cref.p(so), cref.p(tr))
    exp = cref.fe_array(4 * t + 1)
    L.ref_step_witness(field, cref.p(so), t, cref.p(exp))
    W = np.zeros_like(exp)
    i0 = st[2:3].copy()
    ctx.minroot_witness(field, tr, i0, t, W)
    assert np.array_equal(W, exp)


def _shape_arrays(entries, m):
    rows = np.array([e[0] for e in entries], dtype=np.uint32)
    cols = np.array([e[1] for e in entries], dtype=np.uint32)
    vals = mont([e[2] for e in entries], m)
    return rows, cols, vals


def test_spmv_golden_t5(ctx, golden):
    g, m = golden["fold_t5"], o.Q
    sh = g["shape"]
    mats = [_shape_arrays([(a, b, int(c, 16)) for a, b, c in sh[k]], m) for k in "ABC"]
    ncols = sh["num_vars"] + 1 + sh["num_io"]
    shape = ctx.shape_create(o.FIELD_FQ, sh["num_cons"], ncols, mats)
    for tag in ("1", "2"):
        z = mont(hexes(g["z" + tag]), m)
        az, bz, cz = (np.zeros((sh["num_cons"], 4), dtype="<u8") for _ in range(3))
        ctx.spmv3(shape, z, az, bz, cz)
        assert unmont(az, m) == hexes(g["az" + tag])
        assert unmont(bz, m) == hexes(g["bz" + tag])
        assert unmont(cz, m) == hexes(g["cz" + tag])
    shape.free()


@pytest.mark.parametrize("t", [64, 4096])
def test_spmv_step_circuit_vs_c(ctx, cref, t):
    m = o.Q
    sh = o.step_circuit_shape(t, o.FIELD_FQ)
    mats = [_shape_arrays(e, m) for e in (sh.A, sh.B, sh.C)]
    ncols = sh.num_vars + 1 + sh.num_io
    shape = ctx.shape_create(o.FIELD_FQ, sh.num_cons, ncols, mats)
    z = rand_limbs(np.random.default_rng(t), ncols)
    outs = [np.zeros((sh.num_cons, 4), dtype="<u8") for _ in range(3)]
    ctx.spmv3(shape, z, *outs)
    for (rows, cols, vals), got in zip(mats, outs):
        exp = cref.fe_array(sh.num_cons)
        cref.lib().ref_spmv(o.FIELD_FQ, cref.p(rows), cref.p(cols), cref.p(vals), len(rows), cref.p(z), sh.num_cons, cref.p(exp))
        assert np.array_equal(got, exp)
    shape.free()


def test_fold_linearity_property_full_size(ctx):
    """Size-independent property at the BASELINE size (t = 2^16 rows): axpy(a, r, b) - a == r*b."""
    import torch
    n = 196609
    rng = np.random.default_rng(5)
    a, b, r = rand_limbs(rng, n), rand_limbs(rng, n), rand_limbs(rng, 1)
    zero = np.zeros_like(a)
    s1, s2 = np.zeros_like(a), np.zeros_like(a)
    ctx.axpy(o.FIELD_FQ, a, r, b, n, s1)
    ctx.axpy(o.FIELD_FQ, zero, r, b, n, s2)
    rb = np.zeros_like(a)
    ctx.fe_mul(o.FIELD_FQ, np.broadcast_to(r, a.shape).copy(), b, n, rb)
    assert np.array_equal(s2, rb)
    # s1 - a == s2: check through axpy with r = -1:  s1 + (-1)*a
    minus_one = mont([o.Q - 1], o.Q)
    d = np.zeros_like(a)
    ctx.axpy(o.FIELD_FQ, s1, minus_one, a, n, d)
    assert np.array_equal(d, s2)
