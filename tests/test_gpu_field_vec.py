"""GPU parity: field multiply / Montgomery conversion and the folding vector kernels, through
the C ABI, bit-exact against the golden vectors and the C restatement."""
import numpy as np
import pytest

import vdf_amd
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


# ---- fused step operations: must equal the composition of the unfused references, bit for bit ----------
def _dev(x):
    import torch
    return torch.from_numpy(np.ascontiguousarray(x).view(np.int64)).cuda()


def _host(t):
    return t.cpu().numpy().view("<u8")


@pytest.mark.parametrize("field", FIELDS)
@pytest.mark.parametrize("t", [1, 10, 1000, 65536])
def test_step_z_equals_witness_plus_scalar_stores(ctx, cref, field, t):
    L, m = cref.lib(), o.modulus(field)
    st = mont([o.rand_fe(t + 3, 0, m), 0, 7], m)
    so, tr = cref.fe_array(3), cref.fe_array(2 * (t + 1))
    L.ref_minroot_eval(field, 1, cref.p(st), t, cref.p(so), cref.p(tr))
    W = cref.fe_array(4 * t + 1)
    L.ref_step_witness(field, cref.p(so), t, cref.p(W))
    rng = np.random.default_rng(t)
    z_in, u, X = rand_limbs(rng, 3), rand_limbs(rng, 1), rand_limbs(rng, 6)
    i0 = st[2:3].copy()
    exp = np.concatenate([z_in, W, u, X])
    assert np.array_equal(W[-1], i0[0])
    z = _dev(np.zeros_like(exp))
    ctx.minroot_step_z(field, _dev(tr), t, z_in, i0, u, X, z)
    ctx.sync()
    assert np.array_equal(_host(z), exp)


@pytest.mark.parametrize("field", FIELDS)
@pytest.mark.parametrize("t", [1, 7, 1000])
def test_step_z_packed_is_the_witness_without_new_x(ctx, cref, field, t):
    """w_packed = [z_in | tmp1, tmp2, new_y per round | i0]; and the relation that makes new_x redundant in a commitment
    (src/nova/proof.rs:162-173): new_x_j = y_j - (i_j - 1), y_0 = z_in.y, y_j = new_y_{j-1}, i_j = z_in.i - j."""
    L, m = cref.lib(), o.modulus(field)
    st = mont([o.rand_fe(t + 5, 0, m), 0, 11], m)
    so, tr = cref.fe_array(3), cref.fe_array(2 * (t + 1))
    L.ref_minroot_eval(field, 1, cref.p(st), t, cref.p(so), cref.p(tr))
    rng = np.random.default_rng(t)
    u, X = rand_limbs(rng, 1), rand_limbs(rng, 6)
    z_in, i0 = so.copy(), st[2:3].copy()                 # the step starts from the evaluation's result
    z = _dev(np.zeros((4 * t + 11, 4), dtype="<u8"))
    wp = _dev(np.zeros((3 * t + 4, 4), dtype="<u8"))
    ctx.minroot_step_z_packed(field, _dev(tr), t, z_in, i0, u, X, z, wp)
    ctx.sync()
    zz, ww = _host(z), _host(wp)
    rounds = zz[3:3 + 4 * t].reshape(t, 4, 4)
    exp = np.concatenate([zz[:3], rounds[:, 1:, :].reshape(3 * t, 4), zz[3 + 4 * t:4 + 4 * t]])
    assert np.array_equal(ww, exp)
    W = unmont(zz[:4 * t + 4], m)
    y, i = W[1], W[2]
    for j in range(t):
        new_x, new_y = W[3 + 4 * j], W[6 + 4 * j]
        assert new_x == (y - (i - j - 1)) % m
        y = new_y
    with pytest.raises(Exception):
        ctx.minroot_step_z_packed(field, _dev(tr), t, z_in, i0, u, X, z, np.zeros((3 * t + 4, 4), dtype="<u8"))   # host buffer


@pytest.mark.parametrize("per", [4, 3])
@pytest.mark.parametrize("t", [5, 64, 4096])
def test_stencil_with_the_fold_on_the_way(ctx, cref, t, per):
    """vdf_nifs_cross_term_minroot_fold = the fold of the stencil rows (ref_axpy of the C restatement, row by row: X1 += r X2prev,
    E1 += r Tprev) followed by vdf_nifs_cross_term_minroot over the folded rows -- every vector byte-identical, the rows behind
    the stencil untouched, for a 128-bit challenge (the plain-integer product) and a full-width one, with and without E."""
    field, m = o.FIELD_FQ, o.Q
    L = cref.lib()
    S = 3 + 11                                                     # some variables in front of z_in, as an augmented circuit has
    nvars = S + per * t + 1 + 9
    ncols, nc, row0 = nvars + 3, 3 * t + 1 + 13, 6
    nr = 3 * t + 1
    rng = np.random.default_rng(1000 * per + t)
    z2 = rand_limbs(rng, ncols)
    z2[nvars] = limbs([o.to_mont(1, m)])[0]                        # a fresh instance: the constant is one
    u1 = rand_limbs(rng, 1)
    for bits in (128, 254):
        r_int = int(rng.integers(1, 2**63)) ** 4 % (1 << bits) | 1
        r = limbs([o.to_mont(r_int % m, m)])
        for with_e in (True, False):
            run = [rand_limbs(rng, nc) for _ in range(4)]           # A z1, B z1, C z1, E1
            prev = [rand_limbs(rng, nc) for _ in range(4)]          # A z2, B z2, C z2, T of the previous step
            # expected: fold of rows [row0, row0 + nr) on the host's C restatement, then the plain stencil on the device
            want_run = [x.copy() for x in run]
            for k in range(4 if with_e else 3):
                a, b = np.ascontiguousarray(run[k][row0:row0 + nr]), np.ascontiguousarray(prev[k][row0:row0 + nr])
                out = cref.fe_array(nr)
                L.ref_axpy(field, cref.p(a), cref.p(r), cref.p(b), nr, cref.p(out))
                want_run[k][row0:row0 + nr] = out
            w2 = [_dev(x) for x in prev[:3]]
            wT = _dev(prev[3])
            ctx.nifs_cross_term_minroot(field, per, t, S, nvars, row0, _dev(z2), *[_dev(x) for x in want_run[:3]], u1, *w2, wT)
            d_run = [_dev(x) for x in run]
            d_prev = [_dev(x) for x in prev]
            ctx.nifs_cross_term_minroot_fold(field, per, t, S, nvars, row0, _dev(z2), r, d_run[0], d_run[1], d_run[2],
                                             d_run[3] if with_e else None, d_prev[3] if with_e else None, u1, d_prev[0], d_prev[1],
                                             d_prev[2], d_prev[3])
            ctx.sync()
            for k in range(4):
                assert np.array_equal(_host(d_run[k]), want_run[k]), (bits, with_e, k)
            for k in range(3):
                assert np.array_equal(_host(d_prev[k]), _host(w2[k])), (bits, with_e, k)
            assert np.array_equal(_host(d_prev[3]), _host(wT)), (bits, with_e)
            assert np.array_equal(_host(d_prev[3])[:row0], prev[3][:row0]) and np.array_equal(_host(d_prev[3])[row0 + nr:], prev[3][row0 + nr:])
    with pytest.raises(Exception):
        ctx.nifs_cross_term_minroot_fold(field, per, t, S, nvars, row0, _dev(z2), _dev(r), *d_run[:3], None, None, u1, *d_prev)   # r on the device
    with pytest.raises(Exception):
        ctx.nifs_cross_term_minroot_fold(field, per, t, S, nvars, row0, _dev(z2), r, *d_run[:3], d_run[3], None, u1, *d_prev)      # E without T_prev


@pytest.mark.parametrize("t", [5, 64, 4096])
def test_nifs_cross_term_equals_spmv_then_cross(ctx, cref, t):
    field, m = o.FIELD_FQ, o.Q
    sh = o.step_circuit_shape(t, field)
    mats = [_shape_arrays(e, m) for e in (sh.A, sh.B, sh.C)]
    ncols, nc = sh.num_vars + 1 + sh.num_io, sh.num_cons
    shape = ctx.shape_create(field, nc, ncols, mats)
    rng = np.random.default_rng(t + 100)
    z2 = rand_limbs(rng, ncols)
    abc1 = [rand_limbs(rng, nc) for _ in range(3)]
    u1 = rand_limbs(rng, 1)
    abc2 = []
    for rows, cols, vals in mats:
        e = cref.fe_array(nc)
        cref.lib().ref_spmv(field, cref.p(rows), cref.p(cols), cref.p(vals), len(rows), cref.p(z2), nc, cref.p(e))
        abc2.append(e)
    expT = cref.fe_array(nc)
    cref.lib().ref_cross_term(field, *(cref.p(x) for x in abc1 + abc2), cref.p(u1), nc, cref.p(expT))
    d1 = [_dev(x) for x in abc1]
    d2 = [_dev(np.zeros((nc, 4), dtype="<u8")) for _ in range(3)]
    dT = _dev(np.zeros((nc, 4), dtype="<u8"))
    ctx.nifs_cross_term(shape, _dev(z2), *d1, u1, *d2, dT)
    ctx.sync()
    for got, e in zip(d2, abc2):
        assert np.array_equal(_host(got), e)
    assert np.array_equal(_host(dT), expT)
    # the same in two calls over complementary row ranges (vdf_nifs_cross_term_rows): the rows of a range first, from a
    # z2 whose other columns are still garbage where those rows do not read them -- here simply: inside, then outside
    for rb, rn in ((0, nc), (nc, 0), (1, nc // 2), (nc // 3, nc - nc // 3)):
        e2 = [_dev(np.zeros((nc, 4), dtype="<u8")) for _ in range(3)]
        eT = _dev(np.zeros((nc, 4), dtype="<u8"))
        ctx.nifs_cross_term_rows(shape, rb, rn, 1, _dev(z2), *d1, u1, *e2, eT)
        ctx.sync()
        part = _host(eT).copy()
        assert np.array_equal(part[rb:rb + rn], expT[rb:rb + rn]) and not part[:rb].any() and not part[rb + rn:].any()
        ctx.nifs_cross_term_rows(shape, rb, rn, 2, _dev(z2), *d1, u1, *e2, eT)
        ctx.sync()
        assert np.array_equal(_host(eT), expT)
        for got, e in zip(e2, abc2):
            assert np.array_equal(_host(got), e)
    # the MinRoot rows by STENCIL (vdf_nifs_cross_term_minroot: no sparse matrix at all): rows 0 .. 3t of this shape are the
    # reference's rounds with seg_begin = 3 (z_in in front) and the constant at num_vars; every vector byte-identical to
    # the sparse kernel's on a RANDOM z2 (constant column included), the six wrapper rows behind them untouched
    e2 = [_dev(np.zeros((nc, 4), dtype="<u8")) for _ in range(3)]
    eT = _dev(np.zeros((nc, 4), dtype="<u8"))
    ctx.nifs_cross_term_minroot(field, 4, t, 3, sh.num_vars, 0, _dev(z2), *d1, u1, *e2, eT)
    ctx.sync()
    nr = 3 * t + 1
    assert np.array_equal(_host(eT)[:nr], expT[:nr]) and not _host(eT)[nr:].any()
    for got, e in zip(e2, abc2):
        assert np.array_equal(_host(got)[:nr], e[:nr]) and not _host(got)[nr:].any()
    # the constant's column holding ONE (every fresh instance): the kernel's small-integer path, no product for (j + 1) * one --
    # against the sparse kernel on the same z2
    z2u = z2.copy()
    z2u[sh.num_vars] = limbs([o.to_mont(1, m)])[0]
    g2 = [_dev(np.zeros((nc, 4), dtype="<u8")) for _ in range(3)]
    gT = _dev(np.zeros((nc, 4), dtype="<u8"))
    ctx.nifs_cross_term(shape, _dev(z2u), *d1, u1, *g2, gT)
    e2 = [_dev(np.zeros((nc, 4), dtype="<u8")) for _ in range(3)]
    eT = _dev(np.zeros((nc, 4), dtype="<u8"))
    ctx.nifs_cross_term_minroot(field, 4, t, 3, sh.num_vars, 0, _dev(z2u), *d1, u1, *e2, eT)
    ctx.sync()
    assert np.array_equal(_host(eT)[:nr], _host(gT)[:nr])
    for got, want in zip(e2, g2):
        assert np.array_equal(_host(got)[:nr], _host(want)[:nr])
    with pytest.raises(Exception):
        ctx.nifs_cross_term_minroot(field, 5, t, 3, sh.num_vars, 0, _dev(z2), *d1, u1, *e2, eT)      # 3 or 4 variables per round
    with pytest.raises(Exception):
        ctx.nifs_cross_term_minroot(field, 4, t, 3, sh.num_vars - 1, 0, _dev(z2), *d1, u1, *e2, eT)  # the constant inside the rounds
    with pytest.raises(Exception):
        ctx.nifs_cross_term_rows(shape, 1, nc, 1, _dev(z2), *d1, u1, *d2, dT)      # range beyond the shape
    with pytest.raises(Exception):
        ctx.nifs_cross_term_rows(shape, 0, 1, 3, _dev(z2), *d1, u1, *d2, dT)       # unknown selection
    # scalars of fused calls must be host memory; vectors device memory
    with pytest.raises(Exception):
        ctx.nifs_cross_term(shape, z2, *d1, u1, *d2, dT)
    with pytest.raises(Exception):
        ctx.nifs_cross_term(shape, _dev(z2), *d1, _dev(u1), *d2, dT)
    shape.free()


def test_vec_is_zero(ctx):
    for n in (0, 1, 255, 256, 257, 70001):
        v = np.zeros((n, 4), dtype="<u8")
        assert ctx.vec_is_zero(_dev(v) if n else v, n)
        for pos in ([0, n // 2, n - 1] if n else []):
            w = v.copy()
            w[pos, int(pos) % 4] = 1 << (pos % 64)
            assert not ctx.vec_is_zero(_dev(w), n)
            assert not ctx.vec_is_zero(w, n)                       # host memory is staged
    with pytest.raises(Exception):
        from vdf_amd._lib import lib
        ctx._check(lib.vdf_vec_is_zero(ctx.handle, None, 0, None))


@pytest.mark.parametrize("field", FIELDS)
def test_fold_many_equals_axpy_per_vector(ctx, cref, field):
    rng = np.random.default_rng(42)
    sizes = [262155, 196615, 1, 0, 255, 256, 257, 70000]
    r = rand_limbs(rng, 1)
    acc = [rand_limbs(rng, n) for n in sizes]
    add = [rand_limbs(rng, n) for n in sizes]
    exp = []
    for a, b, n in zip(acc, add, sizes):
        e = np.zeros_like(a)
        cref.lib().ref_axpy(field, cref.p(a), cref.p(r), cref.p(b), n, cref.p(e))
        exp.append(e)
    dacc = [_dev(a) if len(a) else None for a in acc]
    dadd = [_dev(b) if len(b) else None for b in add]
    ctx.fold_many(field, r, dacc, dadd, sizes)
    ctx.sync()
    for d, e, n in zip(dacc, exp, sizes):
        if n:
            assert np.array_equal(_host(d), e)
    with pytest.raises(Exception):
        ctx.fold_many(field, r, dacc * 2, dadd * 2, sizes * 2)      # more than 8 vectors


@pytest.mark.parametrize("field", [o.FIELD_FP, o.FIELD_FQ])
def test_fold_many_with_challenges_below_2_128(ctx, cref, field):
    """A fold challenge is a 128-bit integer (CHAL_BITS): vdf_fold_many then multiplies by its PLAIN value without a Montgomery
    reduction (fe_mul_u128: y r = Ph 2^254 + Pl = Pl - c Ph mod m).  Edge scalars (0, 1, 2^128 - 1, 2^127, the first value that
    takes the general path again: 2^128) and edge elements (0, 1, m - 1, 2^254, c, values whose product lands on Ph = 2^128),
    against the C restatement; and the same inputs with the fast path switched off."""
    from vdf_amd import hip as _hip
    m = o.modulus(field)
    rng = np.random.default_rng(128 + field)
    n = 4099
    edge = [0, 1, m - 1, m - 2, 1 << 254, (1 << 254) - 1, m - (1 << 254), (m - 1) // 2, 3, (1 << 128) - 1]
    vals = edge + [int(rng.integers(0, 1 << 62)) * int(rng.integers(0, 1 << 62)) * int(rng.integers(1, 1 << 62)) *
                   int(rng.integers(1, 1 << 62)) * int(rng.integers(1, 1 << 8)) % m for _ in range(n - len(edge))]
    add = mont(vals, m)
    add[:len(edge)] = limbs(edge)                  # the edge patterns as STORED limbs too (the kernel multiplies those)
    add[len(edge):2 * len(edge)] = mont(edge, m)
    acc0 = rand_limbs(rng, n)
    acc0[:3] = limbs([0, m - 1, 1])
    scalars = [0, 1, (1 << 128) - 1, 1 << 127, (1 << 128) - 0x1234567, 1 << 128, int(rng.integers(1, 1 << 62)) * int(rng.integers(1, 1 << 62)),
               int(rng.integers(1, 1 << 62)) << 66 | 5]
    was = _hip.tuning_get().fold_u128
    try:
        for fast in (1, 0):
            _hip.tuning_set(fold_u128=fast)
            for rint in scalars:
                r = mont([rint % m], m)
                e = np.zeros_like(acc0)
                cref.lib().ref_axpy(field, cref.p(acc0), cref.p(r), cref.p(add), n, cref.p(e))
                d = _dev(acc0.copy())
                ctx.fold_many(field, r, [d], [_dev(add)], [n])
                ctx.sync()
                assert np.array_equal(_host(d), e), (fast, hex(rint))
    finally:
        _hip.tuning_set(fold_u128=was)


def test_ctx_wait_orders_two_contexts(ctx):
    """vdf_ctx_wait: work on the second context sees the results of the first without a host sync."""
    n = 1 << 16
    rng = np.random.default_rng(9)
    a, b = rand_limbs(rng, n), rand_limbs(rng, n)
    ctx2 = vdf_amd.Context(ctx.device)
    was = ctx.get_async()
    ctx.set_async(True); ctx2.set_async(True)
    try:
        da, db = _dev(a), _dev(b)
        d1, d2 = _dev(np.zeros_like(a)), _dev(np.zeros_like(a))
        ctx.fe_mul(o.FIELD_FQ, da, db, n, d1)          # d1 = a*b on ctx
        ctx2.wait(ctx)
        ctx2.fe_mul(o.FIELD_FQ, d1, db, n, d2)         # d2 = d1*b on ctx2, ordered after the first
        ctx2.sync()
        exp1, exp2 = np.zeros_like(a), np.zeros_like(a)
        ctx.set_async(False)
        ctx.fe_mul(o.FIELD_FQ, a, b, n, exp1)
        ctx.fe_mul(o.FIELD_FQ, exp1, b, n, exp2)
        assert np.array_equal(_host(d2), exp2)
    finally:
        ctx.set_async(was)
        ctx2.close()


def test_ctx_marks(ctx):
    """vdf_ctx_mark / vdf_ctx_sync_mark: the host waits for what was enqueued before the mark, no more."""
    n = 1 << 16
    rng = np.random.default_rng(10)
    a, b = rand_limbs(rng, n), rand_limbs(rng, n)
    was = ctx.get_async()
    ctx.set_async(True)
    try:
        fresh = vdf_amd.Context(0)                    # (the session's context has had its marks set by earlier proofs)
        with pytest.raises(Exception):
            fresh.sync_mark(3)                        # never set
        fresh.close()
        for bad in (-1, 16):
            with pytest.raises(Exception):
                ctx.mark(bad)
        da, db = _dev(a), _dev(b)
        d1, d2 = _dev(np.zeros_like(a)), _dev(np.zeros_like(a))
        ctx.fe_mul(o.FIELD_FQ, da, db, n, d1)
        ctx.mark(2)
        ctx.fe_mul_chain(o.FIELD_FQ, da, n, 2000, d2) # long-running work behind the mark
        ctx.sync_mark(2)
        exp1 = np.zeros_like(a)
        got1 = _host(d1).copy()                       # complete although the chain may still run
        ctx.sync()
        ctx.set_async(False)
        ctx.fe_mul(o.FIELD_FQ, a, b, n, exp1)
        assert np.array_equal(got1, exp1)
        # a second context waits for a mark of the first, not for the long work behind it
        other = vdf_amd.Context(0)
        other.set_async(True)
        ctx.set_async(True)
        d3, d4 = _dev(np.zeros_like(a)), _dev(np.zeros_like(a))
        ctx.fe_mul(o.FIELD_FQ, da, db, n, d3)
        ctx.mark(1)
        ctx.fe_mul_chain(o.FIELD_FQ, da, n, 2000, d2)
        other.wait_mark(ctx, 1)
        other.fe_mul(o.FIELD_FQ, d3, db, n, d4)        # reads what was written before the mark
        other.sync()
        ctx.sync()
        ctx.set_async(False)
        exp4 = np.zeros_like(a)
        ctx.fe_mul(o.FIELD_FQ, exp1, b, n, exp4)
        assert np.array_equal(_host(d4), exp4)
        with pytest.raises(Exception):
            ctx.wait_mark(other, 3)                    # never set (on `other`: the session's context has had its marks set by earlier proofs)
        other.close()
    finally:
        ctx.set_async(was)


@pytest.mark.parametrize("field", [o.FIELD_FP, o.FIELD_FQ])
def test_spmv_and_fused_cross_term_on_skewed_rows(ctx, cref, field):
    """Rows of 0 .. 300 entries with arbitrary coefficients, as an augmented circuit has them (bit packings of 255 terms,
    Poseidon states of ~60): rows beyond VDF_LONG_ROW = 8 entries are summed by a wavefront each (k_spmv_long ahead of
    the lane-per-row kernels, or inside the cross term's launch: k_nifs_cross_f).  Against the C restatement, for vdf_spmv3 and for vdf_nifs_cross_term."""
    m = o.modulus(field)
    rng = np.random.default_rng(77 + field)
    lens = [0, 1, 2, 7, 8, 9, 10, 33, 63, 64, 65, 127, 128, 129, 255, 300] * 9 + [1] * 500 + [60] * 200
    rng.shuffle(lens)
    nc, ncols = len(lens), 700
    mats = []
    for k in range(3):
        rows, cols, vals = [], [], []
        for r, ln in enumerate(np.roll(lens, 17 * k)):
            cs = rng.choice(ncols, size=int(ln), replace=False) if ln <= ncols else rng.integers(0, ncols, size=int(ln))
            for c in cs:
                rows.append(r); cols.append(int(c))
                kind = rng.integers(0, 4)
                vals.append(1 if kind == 0 else (m - 1 if kind == 1 else int(rng.integers(2, 1 << 62)) * int(rng.integers(1, 1 << 62)) % m))
        mats.append((np.array(rows, dtype=np.uint32), np.array(cols, dtype=np.uint32), mont(vals, m) if vals else np.zeros((0, 4), dtype="<u8")))
    shape = ctx.shape_create(field, nc, ncols, mats)
    z = rand_limbs(rng, ncols)
    outs = [np.zeros((nc, 4), dtype="<u8") for _ in range(3)]
    ctx.spmv3(shape, z, *outs)
    exp = []
    for (rows, cols, vals), got in zip(mats, outs):
        e = cref.fe_array(nc)
        cref.lib().ref_spmv(field, cref.p(rows), cref.p(cols), cref.p(vals), len(rows), cref.p(z), nc, cref.p(e))
        assert np.array_equal(got, e)
        exp.append(e)
    abc1 = [rand_limbs(rng, nc) for _ in range(3)]
    u1 = rand_limbs(rng, 1)
    expT = cref.fe_array(nc)
    cref.lib().ref_cross_term(field, *(cref.p(x) for x in abc1 + exp), cref.p(u1), nc, cref.p(expT))
    # the long rows inside the cross term's own launch (k_nifs_cross_f, the default) and by k_spmv_long ahead of it
    from vdf_amd import hip as _hip
    was = _hip.tuning_get().nifs_fused
    try:
        for fused in (1, 0):
            _hip.tuning_set(nifs_fused=fused)
            d2 = [_dev(np.zeros((nc, 4), dtype="<u8")) for _ in range(3)]
            dT = _dev(np.zeros((nc, 4), dtype="<u8"))
            ctx.nifs_cross_term(shape, _dev(z), *[_dev(x) for x in abc1], u1, *d2, dT)
            ctx.sync()
            for got, e in zip(d2, exp):
                assert np.array_equal(_host(got), e), fused
            assert np.array_equal(_host(dT), expT), fused
    finally:
        _hip.tuning_set(nifs_fused=was)
    # a row range must not hold a long row (they are the OUTSIDE call's work); a range of short rows is fine
    long_rows = [r for r in range(nc) if any(np.roll(lens, 17 * k)[r] > 8 for k in range(3))]
    with pytest.raises(Exception):
        ctx.nifs_cross_term_rows(shape, long_rows[0], 1, 1, _dev(z), *[_dev(x) for x in abc1], u1, *d2, dT)
    short = next(r for r in range(nc) if r not in set(long_rows))
    e2 = [_dev(np.zeros((nc, 4), dtype="<u8")) for _ in range(3)]
    eT = _dev(np.zeros((nc, 4), dtype="<u8"))
    ctx.nifs_cross_term_rows(shape, short, 1, 1, _dev(z), *[_dev(x) for x in abc1], u1, *e2, eT)
    ctx.nifs_cross_term_rows(shape, short, 1, 2, _dev(z), *[_dev(x) for x in abc1], u1, *e2, eT)
    ctx.sync()
    assert np.array_equal(_host(eT), expT)
    shape.free()


@pytest.mark.parametrize("field", [o.FIELD_FP, o.FIELD_FQ])
@pytest.mark.parametrize("t,per", [(1, 3), (5, 4), (1024, 3), (65536, 3), (4097, 4)])
def test_minroot_step_segment(ctx, cref, field, t, per):
    """vdf_minroot_step_segment: the step circuit's own variables inside an augmented circuit -- per = 4 is the reference's
    allocation (src/nova/proof.rs:167-181), per = 3 the bound form without new_x -- against the C restatement's witness."""
    m = o.modulus(field)
    st = mont([o.rand_fe(5, 0, m), 0, 1], m)
    so, tr = cref.fe_array(3), cref.fe_array(2 * (t + 1))
    cref.lib().ref_minroot_eval(field, 1, cref.p(st), t, cref.p(so), cref.p(tr))
    W = cref.fe_array(4 * t + 1)
    cref.lib().ref_step_witness(field, cref.p(so), t, cref.p(W))
    out = _dev(np.zeros((per * t + 1, 4), dtype="<u8"))
    i0 = st[2:3].copy()
    ctx.minroot_step_segment(field, _dev(tr), t, i0, per, out)
    ctx.sync()
    got = _host(out)
    rounds = W[:4 * t].reshape(t, 4, 4)
    want = rounds if per == 4 else rounds[:, 1:, :]
    assert np.array_equal(got[:per * t], want.reshape(per * t, 4)) and np.array_equal(got[per * t], W[4 * t])


@pytest.mark.parametrize("t", [1, 2, 7, 1024, 4097])
def test_packed_commitment_of_the_reference_rounds(ctx, cref, t):
    """vdf_minroot_step_segment_packed (include/vdf_hip.h): the reference's 4t + 1 round variables (src/nova/proof.rs:167-189)
    and the 3t + 4 scalars whose MSM over the DERIVED generators is the same Pedersen commitment -- new_x_j = y_j - (i_in - j - 1)
    folds into the generators of the new_y's.  Checked three ways: `out` equals the C restatement's witness, `packed` equals its
    definition, and the two commitments (C restatement's MSM over 4t + 1 generators, device MSM over the 3t + 4 derived ones)
    are the same point."""
    field, curve, m, bm = o.FIELD_FQ, o.CURVE_PALLAS, o.Q, o.P
    L = cref.lib()
    st = mont([o.rand_fe(15, 0, m), 0, 3], m)
    so, tr = cref.fe_array(3), cref.fe_array(2 * (t + 1))
    L.ref_minroot_eval(field, 1, cref.p(st), t, cref.p(so), cref.p(tr))
    W = cref.fe_array(4 * t + 1)
    L.ref_step_witness(field, cref.p(so), t, cref.p(W))
    out, packed = _dev(np.zeros((4 * t + 1, 4), dtype="<u8")), _dev(np.zeros((3 * t + 4, 4), dtype="<u8"))
    i0, i_in = st[2:3].copy(), so[2:3].copy()                  # final_i = the input state's i; the first round reads the result's i
    ctx.minroot_step_segment_packed(field, _dev(tr), t, i0, i_in, out, packed)
    ctx.sync()
    got, pk = _host(out), _host(packed)
    assert np.array_equal(got, W)
    rounds = W[:4 * t].reshape(t, 4, 4)
    assert np.array_equal(pk[:3 * t], rounds[:, 1:, :].reshape(3 * t, 4)) and np.array_equal(pk[3 * t], W[4 * t])
    assert np.array_equal(pk[3 * t + 1], tr[2 * t + 1]) and np.array_equal(pk[3 * t + 2], so[2]) and unmont(pk[3 * t + 3:3 * t + 4], m) == [1]
    # the commitment both ways
    seed, n = 21, 4 * t + 1
    G = cref.fe_array(2 * n).reshape(n, 8)
    L.ref_tai_bases(curve, seed, 0, n, cref.p(G))
    want = np.zeros(12, dtype="<u8")
    L.ref_msm(curve, cref.p(G), cref.p(W), n, 1, 4, 0, cref.p(want))
    pt = lambda row: None if not row.any() else tuple(unmont(row.reshape(2, 4), bm))
    Gi = [pt(G[k]) for k in range(n)]
    add = lambda a, b: o.pt_add(a, b, bm)
    neg = lambda a: None if a is None else (a[0], (-a[1]) % bm)
    D = []
    for j in range(t):
        D += [Gi[4 * j + 1], Gi[4 * j + 2], add(Gi[4 * j + 3], Gi[4 * j + 4]) if j + 1 < t else Gi[4 * t - 1]]
    s1, s2 = None, None
    for j in range(t - 1, -1, -1):
        s1 = add(s1, Gi[4 * j]); s2 = add(s2, s1)
    D += [Gi[4 * t], Gi[0], neg(s1), s2]
    Dl = mont([c for p_ in D for c in (p_ or (0, 0))], bm).reshape(3 * t + 4, 8)
    bases = ctx.bases_upload(curve, Dl)
    got_pt = np.ascontiguousarray(ctx.msm(bases, packed, n=3 * t + 4, is_mont=True))
    ga, wa = np.zeros(8, dtype="<u8"), np.zeros(8, dtype="<u8")
    L.ref_jac_to_affine(curve, cref.p(got_pt), cref.p(ga))
    L.ref_jac_to_affine(curve, cref.p(want), cref.p(wa))
    assert np.array_equal(ga, wa)
    bases.free()


def test_kernel_events_and_the_accumulate_gate(ctx, cref):
    """vdf_ctx_set_kernel_timing / vdf_ctx_kernel_events: every launch of a bucket-method MSM shows up once, in order, with the
    pipeline's algorithmic bytes on its accumulation kernel; vdf_ctx_gate_accumulate holds that kernel of the NEXT MSM of a
    context behind another context's mark (results unchanged), one-shot."""
    n = 1 << 14
    rng = np.random.default_rng(3)
    sc = rand_limbs(rng, n)
    bases = ctx.bases_generate(o.CURVE_PALLAS, 5, n)
    bases.precompute(13, 1)
    want = np.ascontiguousarray(ctx.msm(bases, sc))
    other = vdf_amd.Context(0)
    was = ctx.get_async()
    try:
        ctx.set_kernel_timing(True)
        ctx.kernel_events()
        dsc, dout = _dev(sc), _dev(np.zeros(12, dtype="<u8"))
        ctx.set_async(True); other.set_async(True)
        da = _dev(rand_limbs(rng, 1 << 14))
        dlong = _dev(np.zeros((1 << 14, 4), dtype="<u8"))
        other.fe_mul_chain(o.FIELD_FQ, da, 1 << 14, 3000, dlong)   # ~ms of work on the other queue
        other.mark(2)
        ctx.gate_accumulate(other, 2)
        ctx.msm(bases, dsc, n=n, out=dout)
        ctx.msm(bases, dsc, n=n, out=dout)                          # the gate was for one MSM only
        other.sync(); ctx.sync()
        ev = ctx.kernel_events()
        names = [e[0] for e in ev]
        assert names == ["msm_sort(5 launches)", "k_accumulate", "msm_tail(fixup+reduce)"] * 2
        assert ev[1][1] == 96.0 * n and ev[0][1] == 0.0
        assert all(e[3] >= e[2] for e in ev) and all(ev[i + 1][2] >= ev[i][2] for i in range(len(ev) - 1))
        # the gated accumulation started after its own sort by about the other queue's work; the second MSM's did not wait
        assert ev[1][2] - ev[0][3] > 5 * max(ev[4][2] - ev[3][3], 0.01)
        ga, wa = np.zeros(8, dtype="<u8"), np.zeros(8, dtype="<u8")         # (the Jacobian representative is free: compare affine)
        got = np.ascontiguousarray(_host(dout)).copy()                      # named: the pointer must outlive the call
        cref.lib().ref_jac_to_affine(o.CURVE_PALLAS, cref.p(got), cref.p(ga))
        cref.lib().ref_jac_to_affine(o.CURVE_PALLAS, cref.p(want), cref.p(wa))
        assert np.array_equal(ga, wa)
        assert ctx.kernel_events() == []
        with pytest.raises(Exception):
            ctx.gate_accumulate(other, 3)                           # never set
    finally:
        ctx.set_kernel_timing(False)
        ctx.set_async(was)
        other.close()
        bases.free()
