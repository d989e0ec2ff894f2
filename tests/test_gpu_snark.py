"""GPU parity of the compression-SNARK building blocks (include/vdf_hip.h) against oracle/spartan.py, bit-exact."""
import numpy as np
import pytest

from oracle import pasta as o
from oracle import spartan as sp
from util import limbs, ints, mont, unmont, rand_limbs

pytestmark = pytest.mark.gpu
F, Q = o.FIELD_FQ, o.Q


def _dev(x):
    import torch
    return torch.from_numpy(np.ascontiguousarray(x).view(np.int64)).cuda()


def _host(t):
    return t.cpu().numpy().view("<u8")


def _rand(seed, n):
    return [o.rand_fe(seed, i, Q) for i in range(n)]


@pytest.mark.parametrize("k", [0, 1, 5, 11])
def test_pair_table_is_the_eq_table(ctx, k):
    r = _rand(k + 1, k)
    out = _dev(np.zeros((1 << k, 4), dtype="<u8"))
    ctx.pair_table(F, mont([(1 - x) % Q for x in r], Q) if k else None, mont(r, Q) if k else None, k, out)
    ctx.sync()
    assert unmont(_host(out), Q) == sp.eq_table(r, Q)


def test_fold_halves_binds_and_folds(ctx):
    n = 1 << 12
    vs = [_rand(10 + t, n) for t in range(3)]
    r, x = o.rand_fe(99, 0, Q), o.rand_fe(99, 1, Q)
    xi = pow(x, -1, Q)
    dv = [_dev(mont(v, Q)) for v in vs]
    ctx.fold_halves(F, dv, mont([(1 - r) % Q, x, xi], Q), mont([r, xi, x], Q), n)
    ctx.sync()
    h = n // 2
    assert unmont(_host(dv[0])[:h], Q) == sp.bind(vs[0], r, Q)
    assert unmont(_host(dv[1])[:h], Q) == [(vs[1][i] * x + vs[1][h + i] * xi) % Q for i in range(h)]
    assert unmont(_host(dv[2])[:h], Q) == [(vs[2][i] * xi + vs[2][h + i] * x) % Q for i in range(h)]
    assert unmont(_host(dv[2])[h:], Q) == vs[2][h:]                    # upper half untouched


@pytest.mark.parametrize("n", [2, 256, 1 << 13, 3 << 12])
def test_reductions(ctx, n):
    tabs = [_rand(20 + t, n) for t in range(5)]
    d = [_dev(mont(t, Q)) for t in tabs]
    u = o.rand_fe(5, 5, Q)
    assert unmont(ctx.reduce(F, 0, d[:2], n), Q) == [sum(a * b for a, b in zip(tabs[0], tabs[1])) % Q]
    if n & (n - 1):
        return                                                         # the round reductions need a power of two
    h = n // 2
    at = lambda f, i, t: (f[i] + t * (f[h + i] - f[i])) % Q
    quad = [sum(at(tabs[0], i, t) * at(tabs[1], i, t) for i in range(h)) % Q for t in (0, 2)]
    assert unmont(ctx.reduce(F, 1, d[:2], n), Q) == quad
    cubic = [sum(at(tabs[0], i, t) * ((at(tabs[1], i, t) * at(tabs[2], i, t) - u * at(tabs[3], i, t) - at(tabs[4], i, t)) % Q)
                 for i in range(h)) % Q for t in (0, 2, 3)]
    assert unmont(ctx.reduce(F, 2, d, n, u=mont([u], Q)), Q) == cubic
    cross = [sum(tabs[0][i] * tabs[1][h + i] for i in range(h)) % Q, sum(tabs[0][h + i] * tabs[1][i] for i in range(h)) % Q]
    assert unmont(ctx.reduce(F, 3, d[:2], n), Q) == cross


def _shape_arrays(entries):
    rows = np.array([e[0] for e in entries], dtype=np.uint32)
    cols = np.array([e[1] for e in entries], dtype=np.uint32)
    return rows, cols, mont([e[2] for e in entries], Q)


@pytest.mark.parametrize("t", [3, 64, 300])
def test_spmv3_transposed(ctx, t):
    """M(y) over the shape's own column order; t = 300 makes the constant column heavy (> 64 entries)."""
    sh = o.step_circuit_shape(t, F)
    ncols = sh.num_vars + 1 + sh.num_io
    shape = ctx.shape_create(F, sh.num_cons, ncols, [_shape_arrays(e) for e in (sh.A, sh.B, sh.C)])
    eq = _rand(t, sh.num_cons)
    rho = o.rand_fe(t, 777, Q)
    out = _dev(np.zeros((ncols, 4), dtype="<u8"))
    ctx.spmv3_t(shape, _dev(mont(eq, Q)), mont([rho], Q), out)
    ctx.sync()
    exp = [0] * ncols
    for coef, mat in ((1, sh.A), (rho, sh.B), (rho * rho % Q, sh.C)):
        for r, c, v in mat:
            exp[c] = (exp[c] + coef * v % Q * eq[r]) % Q
    assert unmont(_host(out), Q) == exp
    shape.free()


def test_ipa_round_helpers(ctx):
    n, nj = 1 << 10, 1 << 7
    a, s = _rand(1, nj), _rand(2, n)
    x = o.rand_fe(3, 0, Q)
    xi = pow(x, -1, Q)
    sL, sR = _dev(np.zeros((n, 4), dtype="<u8")), _dev(np.zeros((n, 4), dtype="<u8"))
    ds = _dev(mont(s, Q))
    ctx.ipa_scalars(F, _dev(mont(a, Q)), ds, n, nj, sL, sR)
    ctx.scale_pattern(F, ds, n, nj, mont([xi], Q), mont([x], Q))
    ctx.sync()
    h = nj // 2
    eL = [s[t] * a[(t % nj) - h] % Q if (t % nj) >= h else 0 for t in range(n)]
    eR = [s[t] * a[(t % nj) + h] % Q if (t % nj) < h else 0 for t in range(n)]
    assert unmont(_host(sL), Q) == eL and unmont(_host(sR), Q) == eR
    assert unmont(_host(ds), Q) == [s[t] * (x if (t % nj) >= h else xi) % Q for t in range(n)]


def test_building_blocks_in_the_other_field(ctx):
    """The same kernels instantiated for Fp (a Vesta-side argument would use them)."""
    Fp, P = o.FIELD_FP, o.P
    r = [o.rand_fe(7, i, P) for i in range(6)]
    out = _dev(np.zeros((64, 4), dtype="<u8"))
    ctx.pair_table(Fp, mont([(1 - x) % P for x in r], P), mont(r, P), 6, out)
    ctx.sync()
    assert unmont(_host(out), P) == sp.eq_table(r, P)
    a, b = [o.rand_fe(8, i, P) for i in range(64)], [o.rand_fe(9, i, P) for i in range(64)]
    da, db = _dev(mont(a, P)), _dev(mont(b, P))
    assert unmont(ctx.reduce(Fp, 0, [da, db], 64), P) == [sum(x * y for x, y in zip(a, b)) % P]
    assert unmont(ctx.reduce(Fp, 3, [da, db], 64), P) == [sum(a[i] * b[32 + i] for i in range(32)) % P,
                                                           sum(a[32 + i] * b[i] for i in range(32)) % P]
    ctx.fold_halves(Fp, [da], mont([(1 - r[0]) % P], P), mont([r[0]], P), 64)
    ctx.sync()
    assert unmont(_host(da)[:32], P) == sp.bind(a, r[0], P)


def test_scalar_and_vector_placement_is_checked(ctx):
    v = _dev(np.zeros((8, 4), dtype="<u8"))
    one = mont([1], Q)
    with pytest.raises(Exception):
        ctx.pair_table(F, _dev(one), _dev(one), 1, v)                 # scalars must be host memory
    with pytest.raises(Exception):
        ctx.fold_halves(F, [np.zeros((8, 4), dtype="<u8")], one, one, 8)      # vectors must be device memory
    with pytest.raises(Exception):
        ctx.fold_halves(F, [v], one, one, 6)                          # power of two


@pytest.mark.parametrize("field", [o.FIELD_FP, o.FIELD_FQ])
@pytest.mark.parametrize("k,log_m", [(0, 0), (0, 4), (3, 2), (9, 4), (14, 4)])
def test_pair_table_pattern(ctx, field, k, log_m):
    """out[i] = (pair table over the top k bits of i) * pattern[low log_m bits of i]."""
    import torch
    m = o.modulus(field)
    rng = np.random.default_rng(100 * k + log_m)
    lo, hi = [int(x) % m for x in ints(rand_limbs(rng, max(k, 1)))], [int(x) % m for x in ints(rand_limbs(rng, max(k, 1)))]
    pat = [int(x) % m for x in ints(rand_limbs(rng, 1 << log_m))]
    n = 1 << (k + log_m)
    out = torch.zeros((n, 4), dtype=torch.int64, device="cuda")
    ctx.pair_table_pattern(field, mont(lo, m), mont(hi, m), k, mont(pat, m), log_m, out)
    ctx.sync()
    got = unmont(out.cpu().numpy().view("<u8"), m)
    idx = sorted(i for i in set([0, 1, n - 1, n // 2, n // 3] + [int(x) for x in rng.integers(0, n, size=40)]) if i < n)
    for i in idx:
        e = pat[i & ((1 << log_m) - 1)]
        top = i >> log_m
        for j in range(k):
            e = e * (hi[j] if (top >> (k - 1 - j)) & 1 else lo[j]) % m
        assert got[i] == e, i
    with pytest.raises(Exception):
        ctx.pair_table_pattern(field, mont(lo, m), mont(hi, m), k, mont(pat + [1] * 16, m), 5, out)      # pattern too long
