"""GPU parity: Pippenger MSM through the C ABI against the golden vectors, the C restatement
and (at full BASELINE sizes) the discrete-log identity of the synthetic generators."""
import os
import numpy as np
import pytest

from oracle import pasta as o
from util import limbs, ints, mont, unmont, hexes, jac_to_affine, affine_array, rand_limbs

pytestmark = pytest.mark.gpu
CURVES = [o.CURVE_PALLAS, o.CURVE_VESTA]


def cpu_msm(cref, curve, pts, sc, is_mont=0):
    out = np.zeros(12, dtype="<u8")
    cref.lib().ref_msm(curve, cref.p(pts), cref.p(sc), len(sc), is_mont, 8, 0, cref.p(out))
    return jac_to_affine(out, curve)


@pytest.mark.parametrize("curve", CURVES)
def test_golden_small(ctx, golden, curve):
    sm = o.curve_scalar_modulus(curve)
    for n, c in golden["msm_seed7"][str(curve)].items():
        pts = affine_array([tuple(hexes(p)) for p in c["bases"]], curve)
        bases = ctx.bases_upload(curve, pts)
        exp = tuple(hexes(c["result"]))
        for is_mont in (False, True):
            sc = mont(hexes(c["scalars"]), sm) if is_mont else limbs(hexes(c["scalars"]))
            got = jac_to_affine(ctx.msm(bases, sc, is_mont=is_mont), curve)
            assert (got or (0, 0)) == exp, (curve, n, is_mont)
        bases.free()


@pytest.mark.parametrize("curve", CURVES)
def test_generated_bases_match_oracle(ctx, cref, curve):
    n = 2000
    bases = ctx.bases_generate(curve, 7, n)
    host = bases.download()
    exp = np.zeros((n, 8), dtype="<u8")
    cref.lib().ref_synthetic_bases(curve, 7, 0, n, cref.p(exp))
    assert np.array_equal(host, exp)
    assert cref.lib().ref_count_off_curve(curve, cref.p(host), n) == 0
    bases.free()


@pytest.mark.parametrize("curve", CURVES)
@pytest.mark.parametrize("n", [0, 1, 2, 127, 128, 129, 1 << 10, 5000, 1 << 14, (1 << 16) + 3])
def test_sizes_vs_c(ctx, cref, curve, n):
    bases = ctx.bases_generate(curve, 21, max(n, 1))
    pts = bases.download()
    sc = rand_limbs(np.random.default_rng(n), n)
    got = jac_to_affine(ctx.msm(bases, sc, n=n), curve)
    exp = cpu_msm(cref, curve, pts[:n], sc) if n else None
    assert got == exp
    bases.free()


def _distributions(n, sm, rng):
    full = rand_limbs(rng, n)
    d = {}
    d["all_zero"] = np.zeros((n, 4), dtype="<u8")
    d["all_one"] = limbs([1] * n)
    d["all_q_minus_1"] = limbs([sm - 1] * n)
    d["all_equal_random"] = np.repeat(full[:1], n, axis=0).copy()
    single = np.zeros((n, 4), dtype="<u8"); single[n // 2] = full[0]
    d["single_nonzero"] = single
    half = full.copy(); half[rng.random(n) < 0.5] = 0
    d["half_zero"] = half
    small = np.zeros((n, 4), dtype="<u8"); small[:, 0] = rng.integers(0, 1 << 16, size=n, dtype=np.uint64)
    d["le_16_bit"] = small
    d["one_bit"] = limbs([1 << int(b) for b in rng.integers(0, 254, size=n)])
    d["window_edges"] = limbs([(1 << 15) | (1 << 31) | (0x8000 << 48) | (0xFFFF << 100)] * n)
    return d


@pytest.mark.parametrize("curve", CURVES)
def test_scalar_distributions(ctx, cref, curve):
    n = 6000
    sm = o.curve_scalar_modulus(curve)
    bases = ctx.bases_generate(curve, 33, n)
    pts = bases.download()
    rng = np.random.default_rng(99)
    for name, sc in _distributions(n, sm, rng).items():
        got = jac_to_affine(ctx.msm(bases, sc), curve)
        assert got == cpu_msm(cref, curve, pts, sc), name
    bases.free()


@pytest.mark.parametrize("curve", CURVES)
def test_duplicate_and_identity_bases(ctx, cref, curve):
    """Duplicated bases force the P + P doubling branch of the mixed addition; (0, 0) bases are
    the identity and must be skipped; P and -P in one bucket must cancel."""
    n = 4096
    bases0 = ctx.bases_generate(curve, 5, n)
    pts = bases0.download()
    pts[1::2] = pts[0::2]                      # every point twice
    pts[10] = 0                                # identity bases
    pts[11] = 0
    m = o.curve_base_modulus(curve)
    neg = unmont(pts[20].reshape(2, 4), m)
    pts[21] = limbs([o.to_mont(neg[0], m), o.to_mont((-neg[1]) % m, m)]).reshape(8)   # -P next to P
    bases = ctx.bases_upload(curve, pts)
    rng = np.random.default_rng(1)
    sc = rand_limbs(rng, n)
    sc[1::2] = sc[0::2]                        # equal scalars on equal points: same bucket in every window
    for arr in (sc, limbs([3] * n), limbs([1] * n)):
        got = jac_to_affine(ctx.msm(bases, arr), curve)
        assert got == cpu_msm(cref, curve, pts, arr)
    bases.free(); bases0.free()


@pytest.mark.parametrize("curve", CURVES)
@pytest.mark.parametrize("c", [4, 7, 11, 13, 16, 17, 19, 20])
def test_window_sizes(ctx, cref, curve, c):
    n = 3000
    bases = ctx.bases_generate(curve, 8, n)
    pts = bases.download()
    sc = rand_limbs(np.random.default_rng(c), n)
    ctx.set_msm_window(c)
    try:
        got = jac_to_affine(ctx.msm(bases, sc), curve)
    finally:
        ctx.set_msm_window(0)
    assert got == cpu_msm(cref, curve, pts, sc)
    bases.free()


@pytest.mark.parametrize("curve", CURVES)
@pytest.mark.parametrize("c,sets", [(16, 1), (16, 4), (13, 1), (8, 2), (16, 16), (17, 1), (18, 1), (19, 2), (20, 1)])
def test_fixed_base_tables(ctx, cref, curve, c, sets):
    n = 5000
    bases = ctx.bases_generate(curve, 13, n)
    pts = bases.download()
    bases.precompute(c, sets)
    rng = np.random.default_rng(c * 100 + sets)
    sc = rand_limbs(rng, n)
    assert jac_to_affine(ctx.msm(bases, sc), curve) == cpu_msm(cref, curve, pts, sc)
    # offset + shorter length against the same table (how commit_T reuses the generator table)
    off, k = 777, 3000
    got = jac_to_affine(ctx.msm(bases, sc[:k].copy(), n=k, offset=off), curve)
    assert got == cpu_msm(cref, curve, pts[off:off + k].copy(), sc[:k].copy())
    # skewed scalars through the table path (one heavy bucket)
    eq = np.repeat(sc[:1], n, axis=0).copy()
    assert jac_to_affine(ctx.msm(bases, eq), curve) == cpu_msm(cref, curve, pts, eq)
    bases.free()


@pytest.mark.parametrize("curve", CURVES)
def test_heavy_bucket_path(ctx, curve):
    """All-equal scalars at 2^17 put every point of a window into one bucket: exercises the
    queued wavefront reduce.  Expected value through the discrete-log identity."""
    n = 1 << 17
    sm = o.curve_scalar_modulus(curve)
    bases = ctx.bases_generate(curve, 3, n)
    s = o.rand_fe(1, 0, sm)
    sc = limbs([s] * n)
    exp = o.msm_by_dlog([s] * n, curve, 3)
    assert jac_to_affine(ctx.msm(bases, sc), curve) == exp
    bases.precompute(16, 1)
    assert jac_to_affine(ctx.msm(bases, sc), curve) == exp
    bases.free()


@pytest.mark.parametrize("mode", ["plain", "table", "table_recommended"])
def test_full_size_2_20_dlog_identity(ctx, mode):
    """BASELINE config 2 (2^20 Pallas points), device-resident scalars, checked bit-exactly
    through sum s_i*[k_i]G = [sum s_i*k_i]G."""
    import torch
    n, curve = 1 << 20, o.CURVE_PALLAS
    bases = ctx.bases_generate(curve, 7, n)
    if mode == "table":
        bases.precompute(16, 1)
    elif mode == "table_recommended":                 # the bench's configuration: window 0 = the library's choice
        bases.precompute(0, 1)
        assert bases.window == 17
    else:
        assert bases.window == 0
    sc = rand_limbs(np.random.default_rng(2020), n)
    d = torch.from_numpy(sc.view(np.int64)).cuda()
    out = torch.zeros(12, dtype=torch.int64, device="cuda")
    ctx.msm(bases, d, n=n, out=out)
    ctx.sync()
    got = jac_to_affine(out.cpu().numpy().view("<u8"), curve)
    assert got == o.msm_by_dlog(ints(sc), curve, 7)
    bases.free()


def test_maximum_size_2_24_additivity(ctx):
    """BASELINE config 4's total size on one GPU (2^24 Pallas points, 16 GiB fixed-base table): checked against the
    oracle by the discrete-log identity (oracle.pasta.msm_by_dlog_limbs), and by the size-independent property that
    MSM(all) equals the sum of the quarter MSMs taken at offsets into the same table (even and ragged splits)."""
    import torch
    curve, n = o.CURVE_PALLAS, 1 << 24
    bases = ctx.bases_generate(curve, 7, n)
    bases.precompute(0, 1)
    g = torch.Generator(device="cuda"); g.manual_seed(24)
    sc = torch.randint(-(2**63), 2**63 - 1, (n, 4), dtype=torch.int64, device="cuda", generator=g)
    sc[:, 3] &= 0x3FFFFFFFFFFFFFFF
    was = ctx.get_async()
    ctx.set_async(False)
    try:
        out = torch.zeros(12, dtype=torch.int64, device="cuda")
        ctx.msm(bases, sc, n=n, out=out)
        q = n // 4
        parts = torch.zeros((4, 12), dtype=torch.int64, device="cuda")
        for k in range(4):
            ctx.msm(bases, sc[k * q:(k + 1) * q], n=q, offset=k * q, out=parts[k])
        tot = torch.zeros(12, dtype=torch.int64, device="cuda")
        ctx.point_sum(curve, parts, 4, out=tot)
        aff = lambda t: jac_to_affine(t.cpu().numpy().view("<u8"), curve)
        assert aff(out) == aff(tot) and aff(out) is not None
        # and against the oracle itself: sum s_i [k_i] G = [sum s_i k_i] G, 2^24 exact multiply-adds on the host
        assert aff(out) == o.msm_by_dlog_limbs(sc.cpu().numpy().view("<u8"), curve, 7)
        # ragged split as well: 3 + (n - 3)
        ctx.msm(bases, sc[:3], n=3, out=parts[0])
        ctx.msm(bases, sc[3:], n=n - 3, offset=3, out=parts[1])
        ctx.point_sum(curve, parts, 2, out=tot)
        assert aff(out) == aff(tot)
    finally:
        ctx.set_async(was)
        bases.free()
        del sc
        torch.cuda.empty_cache()


def test_linearity_property(ctx):
    """msm(a + r*b) == msm(a) + r*msm(b): ties the MSM and axpy kernels together at 2^16."""
    n, curve = 1 << 16, o.CURVE_PALLAS
    bases = ctx.bases_generate(curve, 4, n)
    rng = np.random.default_rng(8)
    a, b = rand_limbs(rng, n), rand_limbs(rng, n)
    r = 0x1234567
    am, bm, rm = (np.zeros_like(a) for _ in range(3))
    ctx.fe_to_mont(o.FIELD_FQ, a, n, am); ctx.fe_to_mont(o.FIELD_FQ, b, n, bm)
    rmont = mont([r], o.Q)
    cm = np.zeros_like(a)
    ctx.axpy(o.FIELD_FQ, am, rmont, bm, n, cm)
    pa = jac_to_affine(ctx.msm(bases, am, is_mont=True), curve)
    pb = jac_to_affine(ctx.msm(bases, bm, is_mont=True), curve)
    pc = jac_to_affine(ctx.msm(bases, cm, is_mont=True), curve)
    assert pc == o.pt_add(pa, o.pt_mul(r, pb, o.P), o.P)
    bases.free()


def test_pippenger_shims(cref):
    """Drop-in shims with the upstream pasta-msm shape (host pointers, upload on call)."""
    from vdf_amd._lib import lib
    for curve, fn in ((o.CURVE_PALLAS, lib.mult_pippenger_pallas), (o.CURVE_VESTA, lib.mult_pippenger_vesta)):
        n = 1500
        pts = np.zeros((n, 8), dtype="<u8")
        cref.lib().ref_synthetic_bases(curve, 2, 0, n, cref.p(pts))
        sm = o.curve_scalar_modulus(curve)
        sc = [o.rand_fe(6, i, sm) for i in range(n)]
        scm = mont(sc, sm)
        out = np.zeros(12, dtype="<u8")
        fn(out.ctypes.data, pts.ctypes.data, n, scm.ctypes.data, True)
        assert jac_to_affine(out, curve) == o.msm_by_dlog(sc, curve, 2)


def test_pippenger_shim_generator_cache(cref):
    """vdf_shim_set_cache: the same generator array on later calls is served from HBM (plain, then with its table);
    results stay those of the uncached shim; an array rewritten in place ANYWHERE (the cache hashes every point), another
    length and eviction are handled."""
    import time
    from vdf_amd._lib import lib
    curve, n = o.CURVE_PALLAS, 1 << 16
    pts = np.zeros((n, 8), dtype="<u8")
    cref.lib().ref_synthetic_bases(curve, 3, 0, n, cref.p(pts))
    rng = np.random.default_rng(21)
    sm = o.curve_scalar_modulus(curve)
    out = np.zeros(12, dtype="<u8")

    def run(points, count, seed):
        sc = rand_limbs(np.random.default_rng(seed), count)
        t0 = time.perf_counter()
        lib.mult_pippenger_pallas(out.ctypes.data, points.ctypes.data, count, sc.ctypes.data, False)
        dt = time.perf_counter() - t0
        return jac_to_affine(out, curve), ints(sc), dt

    assert lib.vdf_shim_set_cache(65) != 0 and lib.vdf_shim_set_cache(-1) != 0
    assert lib.vdf_shim_set_cache(2) == 0
    # ONE knob: the call is reflected by vdf_hip_tuning_get, and vdf_hip_tuning_set(shim_cache = ...) is in force
    # on the next shim call -- nothing is latched at first use
    from vdf_amd.hip import tuning_get, tuning_set
    assert tuning_get().shim_cache == 2
    tuning_set(shim_cache=3)
    assert tuning_get().shim_cache == 3
    assert lib.vdf_shim_set_cache(2) == 0 and tuning_get().shim_cache == 2
    try:
        times = []
        for call in range(4):                      # upload, cached + table build, cached table, cached table
            got, sc, dt = run(pts, n, 100 + call)
            assert got == o.msm_by_dlog(sc, curve, 3)
            times.append(dt)
        print("shim call times (ms):", [round(1e3 * t, 2) for t in times])
        got, sc, _ = run(pts, 5000, 7)             # a prefix is another set
        assert got == o.msm_by_dlog(sc, curve, 3)
        # the array rewritten in place where the fingerprint looks: first generator doubled
        g0 = o.synthetic_bases(curve, 3, 1)[0]
        pts2 = pts.copy()
        pts[0] = affine_array([o.pt_add(g0, g0, o.P)], curve)[0]
        got, sc, _ = run(pts, n, 9)
        want = o.pt_add(o.msm_by_dlog(sc, curve, 3), o.pt_mul(sc[0], g0, o.P), o.P)      # + one more s_0 * G_0
        assert got == want
        # ... and rewritten at an arbitrary index (a sampled fingerprint would have missed it): generator 12345 doubled
        g1 = o.synthetic_bases(curve, 3, 1, start=12345)[0]
        pts[12345] = affine_array([o.pt_add(g1, g1, o.P)], curve)[0]
        got, sc, _ = run(pts, n, 19)
        want = o.pt_add(o.pt_add(o.msm_by_dlog(sc, curve, 3), o.pt_mul(sc[0], g0, o.P), o.P), o.pt_mul(sc[12345], g1, o.P), o.P)
        assert got == want
        pts[12345] = pts2[12345]
        # eviction: two other sets push the first one out; it still computes correctly afterwards
        for k, arr in enumerate((pts2, pts2[:30000].copy())):
            got, sc, _ = run(arr, len(arr), 50 + k)
            assert got == o.msm_by_dlog(sc, curve, 3)
        got, sc, _ = run(pts, n, 10)
        assert got == o.pt_add(o.msm_by_dlog(sc, curve, 3), o.pt_mul(sc[0], g0, o.P), o.P)
    finally:
        assert lib.vdf_shim_set_cache(0) == 0      # frees every cached set


def test_pippenger_shim_cache_under_concurrent_callers(cref):
    """Four threads, four generator arrays, a cache of two: entries are evicted while other threads compute.  rayon
    workers call pasta-msm concurrently (SURVEY.md 8b threading), so the shim must stay correct under exactly this."""
    import threading
    from vdf_amd._lib import lib
    curve = o.CURVE_PALLAS
    sets = []
    for k in range(4):
        n = 3000 + 500 * k
        pts = np.zeros((n, 8), dtype="<u8")
        cref.lib().ref_synthetic_bases(curve, 40 + k, 0, n, cref.p(pts))
        sets.append((pts, n, 40 + k))
    assert lib.vdf_shim_set_cache(2) == 0
    failures = []

    def worker(k):
        pts, n, seed = sets[k]
        rng = np.random.default_rng(k)
        for rep in range(12):
            sc = rand_limbs(rng, n)
            out = np.zeros(12, dtype="<u8")
            lib.mult_pippenger_pallas(out.ctypes.data, pts.ctypes.data, n, sc.ctypes.data, False)
            if jac_to_affine(out, curve) != o.msm_by_dlog(ints(sc), curve, seed):
                failures.append((k, rep))

    try:
        threads = [threading.Thread(target=worker, args=(k,)) for k in range(4)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        assert not failures
    finally:
        assert lib.vdf_shim_set_cache(0) == 0


def test_wide_batch_with_a_large_tableless_window(ctx, cref):
    """Found by tools/gpu_msm_fuzz.py: a table-less batch of 3 at window 19 is 42 bucket sets of 2^18 buckets -- more sort
    partitions than the pipeline has, which used to run pass B with 11 fine bits (out-of-bounds writes).  The window of a
    table-less call is now lowered to what fits; results are unchanged."""
    curve = o.CURVE_PALLAS
    bases = ctx.bases_generate(curve, 5, 14524)
    pts = bases.download()
    rng = np.random.default_rng(23)
    offs, lens = [5473, 10447, 12204], [6006, 3444, 1067]
    scs = [rand_limbs(rng, n) for n in lens]
    try:
        for w in (19, 20):
            ctx.set_msm_window(w)
            out = ctx.msm_batch(bases, scs, lens, offs)
            for g in range(3):
                assert jac_to_affine(out[g], curve) == cpu_msm(cref, curve, pts[offs[g]:offs[g] + lens[g]], scs[g])
    finally:
        ctx.set_msm_window(0)
        bases.free()


def test_error_paths(ctx):
    import vdf_amd
    bases = ctx.bases_generate(o.CURVE_PALLAS, 1, 16)
    sc = np.zeros((32, 4), dtype="<u8")
    with pytest.raises(vdf_amd.VdfError) as e:
        ctx.msm(bases, sc, n=32)                       # more scalars than generators
    assert e.value.code == vdf_amd._lib.VDF_ERR_BAD_LENGTH
    with pytest.raises(vdf_amd.VdfError):
        ctx.set_msm_window(3)
    with pytest.raises(vdf_amd.VdfError):
        ctx.axpy(7, sc, sc, sc, 4, sc)                 # unknown field
    bases.free()


@pytest.mark.parametrize("curve", CURVES)
def test_point_sum_and_sharded_ranges(ctx, cref, curve):
    """The multi-GPU combine on one device: MSMs over generator ranges [start, start+count) (what each
    rank owns) summed with vdf_point_sum equal the single MSM over the whole range."""
    from vdf_amd.dist import shard_range
    n, world = 5000, 3
    whole = ctx.bases_generate(curve, 17, n)
    pts = whole.download()
    sc = rand_limbs(np.random.default_rng(4), n)
    exp = cpu_msm(cref, curve, pts, sc)
    partials = np.zeros((world, 12), dtype="<u8")
    for r in range(world):
        start, count = shard_range(n, r, world)
        b = ctx.bases_generate(curve, 17, count, start=start)
        assert np.array_equal(b.download(), pts[start:start + count])
        partials[r] = ctx.msm(b, sc[start:start + count].copy())
        b.free()
    assert jac_to_affine(ctx.point_sum(curve, partials, world), curve) == exp
    # identities and a single point
    zero = np.zeros((4, 12), dtype="<u8")
    assert jac_to_affine(ctx.point_sum(curve, zero, 4), curve) is None
    assert jac_to_affine(ctx.point_sum(curve, partials[:1].copy(), 1), curve) == jac_to_affine(partials[0], curve)
    whole.free()


def test_stage_timing_api(ctx):
    n = 1 << 14
    bases = ctx.bases_generate(o.CURVE_PALLAS, 2, n)
    sc = rand_limbs(np.random.default_rng(1), n)
    ctx.set_timing(True)
    ctx.msm_timing()
    for _ in range(3):
        ctx.msm(bases, sc)
    sort_ms, acc_ms, tail_ms, total_ms, calls = ctx.msm_timing()
    ctx.set_timing(False)
    assert calls == 3 and acc_ms > 0 and total_ms >= acc_ms
    assert abs(total_ms - (sort_ms + acc_ms + tail_ms)) < 0.2 * total_ms + 0.05
    assert ctx.msm_timing()[4] == 0
    bases.free()


@pytest.mark.parametrize("curve", CURVES)
def test_point_sum_exceptional_cases(ctx, curve):
    """The quad-cooperative addition of the tail kernels on its exceptional branches: P + P (doubling),
    P + (-P) (identity), identity operands, and points with z != 1."""
    m = o.curve_base_modulus(curve)
    g = o.generator(curve)
    P1 = o.pt_mul(12345, g, m)
    P2 = o.pt_mul(777, g, m)

    def jac(pt, z=1):
        if pt is None:
            return [0, 0, 0]
        x, y = pt
        return [x * z * z % m, y * z * z * z % m, z % m]

    def run(points):
        flat = [o.to_mont(v, m) for p in points for v in p]
        arr = limbs(flat).reshape(len(points), 12)
        return jac_to_affine(ctx.point_sum(curve, arr, len(points)), curve)

    neg = (P1[0], (-P1[1]) % m)
    assert run([jac(P1), jac(P1)]) == o.pt_add(P1, P1, m)
    assert run([jac(P1, 5), jac(P1, 9)]) == o.pt_add(P1, P1, m)                  # same point, different z
    assert run([jac(P1), jac(neg)]) is None
    assert run([jac(P1, 3), jac(neg, 11), jac(P2)]) == P2
    assert run([jac(None), jac(P2, 7), jac(None)]) == P2
    many = [jac(o.pt_mul(k + 1, g, m), k + 2) for k in range(40)]
    assert run(many) == o.pt_mul(sum(range(1, 41)), g, m)
    assert run([jac(P1)] * 33) == o.pt_mul(33 * 12345, g, m)                     # equal partials across quads


# ---- batched MSM: k vectors over one generator table, one pipeline -----------------------------------------
@pytest.mark.parametrize("curve", CURVES)
@pytest.mark.parametrize("mode", ["plain", "tbl16x1", "tbl13x2"])
def test_msm_batch_equals_separate_calls(ctx, cref, curve, mode):
    nb = 6000
    bases = ctx.bases_generate(curve, 21, nb)
    pts = bases.download()
    if mode != "plain":
        c, s = mode[3:].split("x")
        bases.precompute(int(c), int(s))
    rng = np.random.default_rng(77)
    sizes = [6000, 4099, 1, 0]
    offs = [0, 1901, 5999, 17]
    sc = [rand_limbs(rng, n) for n in sizes]
    sc[1][:50] = sc[1][0]                       # a heavy bucket inside one group only
    for k in (1, 2, 3, 4):
        got = ctx.msm_batch(bases, sc[:k], offsets=offs[:k])
        for g in range(k):
            exp = cpu_msm(cref, curve, pts[offs[g]:offs[g] + sizes[g]].copy(), sc[g]) if sizes[g] else None
            a = jac_to_affine(got[g], curve)
            assert a == exp, (mode, k, g)
            if sizes[g]:
                assert a == jac_to_affine(ctx.msm(bases, sc[g], offset=offs[g]), curve)
    with pytest.raises(Exception):
        ctx.msm_batch(bases, sc + sc[:1], offsets=offs + [0])          # more than 4 groups
    with pytest.raises(Exception):
        ctx.msm_batch(bases, sc[:2], offsets=[0, 5000])                # offset + n beyond the table
    bases.free()


def test_msm_batch_fold_shapes_dlog_identity(ctx):
    """The two commitments of one fold at t = 2^16 (|W| = 262148, |T| = 196615) in one batch, device-resident,
    checked with the discrete-log identity of the synthetic generators."""
    import torch
    curve, seed = o.CURVE_PALLAS, 0x4E6F7661
    nb = 1 << 19
    bases = ctx.bases_generate(curve, seed, nb)
    bases.precompute(16, 1)
    sizes = [262148, 196615]
    g = torch.Generator(device="cuda"); g.manual_seed(5)
    scs = []
    for n in sizes:
        t = torch.randint(0, 2**63 - 1, (n, 4), dtype=torch.int64, device="cuda", generator=g)
        t[:, 3] &= 0x3FFFFFFFFFFFFFFF
        scs.append(t)
    out = torch.zeros((2, 12), dtype=torch.int64, device="cuda")
    ctx.msm_batch(bases, scs, out=out)
    ctx.sync()
    res = out.cpu().numpy().view("<u8")
    for k, n in enumerate(sizes):
        vals = ints(scs[k].cpu().numpy().view("<u8"))
        assert jac_to_affine(res[k], curve) == o.msm_by_dlog(vals, curve, seed)
    bases.free()


# ---- generator family 1: try-and-increment (no known discrete logs) --------------------------------------
@pytest.mark.parametrize("curve", CURVES)
def test_try_and_increment_bases_match_oracle(ctx, cref, curve):
    import vdf_amd
    n, start, seed = 300, 12345, 99
    bases = ctx.bases_generate(curve, seed, n, start=start, family=vdf_amd.GENS_TRY_AND_INCREMENT)
    got = bases.download()
    exp = affine_array(o.tai_bases(curve, seed, n, start), curve)
    assert np.array_equal(got, exp)
    m = o.curve_base_modulus(curve)
    for x, y in (tuple(unmont(got[i].reshape(2, 4), m)) for i in range(0, n, 37)):
        assert (y * y - x * x * x - 5) % m == 0 and y % 2 == 0
    # an MSM over them against the C restatement (plain and table paths)
    sc = rand_limbs(np.random.default_rng(3), n)
    want = cpu_msm(cref, curve, got, sc)
    assert jac_to_affine(ctx.msm(bases, sc), curve) == want
    bases.precompute(16, 1)
    assert jac_to_affine(ctx.msm(bases, sc), curve) == want
    with pytest.raises(Exception):
        ctx.bases_generate(curve, seed, 4, family=7)
    bases.free()


# ---- MSM jobs: vectors pushed one at a time, one shared bucket reduction ------------------------------------
@pytest.mark.parametrize("curve", CURVES)
def test_msm_job_equals_separate_calls(ctx, cref, curve):
    import torch
    nb = 9000
    bases = ctx.bases_generate(curve, 31, nb)
    pts = bases.download()
    rng = np.random.default_rng(5)
    sizes, offs = [9000, 5003, 1, 777], [0, 1234, 8999, 100]
    sc = [rand_limbs(rng, n) for n in sizes]
    dev = [torch.from_numpy(x.view(np.int64)).cuda() for x in sc]
    with pytest.raises(Exception):
        ctx.msm_job(bases, sizes[:2], offs[:2])                 # no fixed-base table yet
    bases.precompute(16, 1)
    was = ctx.get_async()
    ctx.set_async(True)
    try:
        for k in (1, 2, 4):
            job = ctx.msm_job(bases, sizes[:k], offs[:k])
            with pytest.raises(Exception):
                ctx.msm_job(bases, sizes[:1], offs[:1])         # one job per context at a time
            with pytest.raises(Exception):
                ctx.msm(bases, dev[0], n=sizes[0])              # nor any other MSM: the job owns the workspace
            for g in reversed(range(k)):                        # any push order
                job.push(g, dev[g])
            got = job.finish()
            for g in range(k):
                assert jac_to_affine(got[g], curve) == cpu_msm(cref, curve, pts[offs[g]:offs[g] + sizes[g]].copy(), sc[g]), (k, g)
        # a vector written by a kernel enqueued just before the push (stream order, no host sync)
        a, b = rand_limbs(rng, sizes[1]), rand_limbs(rng, sizes[1])
        prod = np.zeros_like(a)
        da, db = torch.from_numpy(a.view(np.int64)).cuda(), torch.from_numpy(b.view(np.int64)).cuda()
        dprod = torch.zeros_like(da)
        field = o.FIELD_FQ if curve == o.CURVE_PALLAS else o.FIELD_FP
        job = ctx.msm_job(bases, [sizes[0], sizes[1]], [0, 0], is_mont=True)
        job.push(0, dev[0])
        ctx.fe_mul(field, da, db, sizes[1], dprod)               # producer of vector 1, on the context's stream
        job.push(1, dprod)
        got = job.finish()
        cref.lib().ref_fe_mul(field, cref.p(a), cref.p(b), sizes[1], cref.p(prod))
        assert jac_to_affine(got[0], curve) == cpu_msm(cref, curve, pts[:sizes[0]].copy(), sc[0], is_mont=1)
        assert jac_to_affine(got[1], curve) == cpu_msm(cref, curve, pts[:sizes[1]].copy(), prod, is_mont=1)
        # unfinished pushes are an error, and the job still ends
        job = ctx.msm_job(bases, sizes[:2], offs[:2])
        job.push(0, dev[0])
        with pytest.raises(Exception):
            job.finish()
        job = ctx.msm_job(bases, sizes[:1], offs[:1])
        job.push(0, dev[0])
        job.finish()
    finally:
        ctx.set_async(was)
    bases.free()


def test_bases_validate(ctx):
    """Setup-time check of an uploaded generator table: canonical coordinates, curve membership, identity allowed."""
    import vdf_amd
    from vdf_amd.hip import VdfError
    curve, m = o.CURVE_PALLAS, o.P
    good = affine_array(o.synthetic_bases(curve, 3, 50) + [None], curve)           # 50 points + the identity
    b = ctx.bases_upload(curve, good)
    b.validate()
    b.free()
    bad = good.copy()
    bad[17, 4:8] = limbs([o.to_mont(5, m)])[0]                                      # y replaced: off the curve
    b = ctx.bases_upload(curve, bad)
    with pytest.raises(VdfError) as e:
        b.validate()
    assert e.value.code == vdf_amd._lib.VDF_ERR_BAD_ARG and "index 17" in str(e.value)
    b.free()
    bad = good.copy()
    bad[9, 0:4] = limbs([m])[0]                                                     # x = m: not canonical
    bad[30, 4:8] = limbs([o.to_mont(5, m)])[0]
    b = ctx.bases_upload(curve, bad)
    with pytest.raises(VdfError) as e:
        b.validate()
    assert e.value.code == vdf_amd._lib.VDF_ERR_NONCANONICAL and "index 9" in str(e.value)
    b.free()
    g = ctx.bases_generate(curve, 3, 1000, family=vdf_amd.GENS_TRY_AND_INCREMENT)
    g.validate()
    g.free()


def test_one_context_from_several_threads(ctx, cref):
    """Calls are re-entrant across threads on one context (per-context mutex; include/vdf_hip.h): four threads issue
    host-buffer MSMs of different sizes concurrently and every result is the right one."""
    import threading
    curve = o.CURVE_PALLAS
    nb = 6000
    bases = ctx.bases_generate(curve, 17, nb)
    pts = bases.download()
    bases.precompute(16, 1)
    rng = np.random.default_rng(123)
    jobs = []
    for k in range(4):
        n = [6000, 4000, 1234, 77][k]
        sc = rand_limbs(rng, n)
        jobs.append((n, sc, cpu_msm(cref, curve, pts[:n].copy(), sc)))
    errors = []

    def worker(k):
        n, sc, want = jobs[k]
        try:
            for _ in range(8):
                got = jac_to_affine(ctx.msm(bases, sc, n=n), curve)
                if got != want:
                    errors.append((k, "mismatch"))
        except Exception as e:          # noqa: BLE001
            errors.append((k, repr(e)))

    ths = [threading.Thread(target=worker, args=(k,)) for k in range(4)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert not errors, errors
    bases.free()


def test_sharded_and_multi_context_msm_through_the_c_abi(ctx):
    """vdf_msm_sharded with an injected collective (here: two 'ranks' played in turn on one GPU, the gather done by hand)
    and vdf_msm_multi (one process, a context per device -- two contexts of device 0 here): both equal the plain MSM."""
    import torch
    import vdf_amd
    from vdf_amd.hip import msm_multi
    from vdf_amd.dist import shard_range
    curve, n, world = o.CURVE_PALLAS, 50000, 2
    sc = rand_limbs(np.random.default_rng(12), n)
    want = o.msm_by_dlog(ints(sc), curve, 5)
    d = torch.from_numpy(sc.view(np.int64)).cuda()
    shards = []
    for r in range(world):
        start, count = shard_range(n, r, world)
        b = ctx.bases_generate(curve, 5, count, start=start)
        b.precompute(0, 1)
        shards.append((b, start, count))
    partials = [torch.zeros(12, dtype=torch.int64, device="cuda") for _ in range(world)]
    gathered = torch.zeros(world * 12, dtype=torch.int64, device="cuda")
    out = torch.zeros(12, dtype=torch.int64, device="cuda")
    # rank 1's partial first (as its own process would), then rank 0 runs the whole sharded call with a gather that
    # places both partials
    b1, s1, c1 = shards[1]
    ctx.msm(b1, d[s1:s1 + c1], n=c1, out=partials[1])
    calls = []

    def all_gather(dst, src):
        calls.append(1)
        dst[:12].copy_(src)
        dst[12:].copy_(partials[1])
    b0, s0, c0 = shards[0]
    ctx.msm_sharded(b0, d[s0:s0 + c0], c0, 0, world, all_gather, partials[0], gathered, out)
    ctx.sync()
    assert calls == [1] and jac_to_affine(out.cpu().numpy().view("<u8"), curve) == want
    # world of one: no collective unless asked for
    ball = ctx.bases_generate(curve, 5, n)
    ctx.msm_sharded(ball, d, n, 0, 1, None, partials[0], gathered[:12], out)
    ctx.sync()
    assert jac_to_affine(out.cpu().numpy().view("<u8"), curve) == want
    with pytest.raises(vdf_amd.VdfError):
        ctx.msm_sharded(b0, d[s0:s0 + c0], c0, 2, world, all_gather, partials[0], gathered, out)       # rank out of range

    def failing(dst, src):
        raise RuntimeError("link down")
    with pytest.raises(RuntimeError):
        ctx.msm_sharded(b0, d[s0:s0 + c0], c0, 0, world, failing, partials[0], gathered, out)
    # one process, two contexts: scalars in host memory for one, device memory for the other
    with vdf_amd.Context(0) as c2:
        b2 = c2.bases_generate(curve, 5, c1, start=s1)
        got = msm_multi([ctx, c2], [b0, b2], [d[s0:s0 + c0], sc[s1:s1 + c1]], [c0, c1])
        assert jac_to_affine(got, curve) == want
        b2.free()
    for b, _, _ in shards:
        b.free()
    ball.free()


@pytest.mark.parametrize("curve", CURVES)
def test_label_derived_bases_match_oracle(ctx, curve):
    """Generator family 2: label -> SHAKE256 -> curve points (nova-snark derives CommitGens this way),
    against oracle/pasta.py label_base: the first points, a range far out, another label, the empty label."""
    m = o.curve_base_modulus(curve)
    for label, start, n in ((b"vdf-nova-ivc-v1 gens", 0, 300), (b"vdf-nova-ivc-v1 gens", (1 << 40) + 17, 70), (b"x", 5, 40), (b"", 0, 9),
                            (bytes(range(64)), 3, 33)):
        b = ctx.bases_generate_label(curve, label, n, start=start)
        pts = b.download()
        got = [tuple(unmont(pts[k].reshape(2, 4), m)) for k in range(n)]
        assert got == [o.label_base(curve, label, start + k) for k in range(n)]
        for x, y in got:
            assert (y * y - x * x * x - 5) % m == 0 and y % 2 == 0 and x != 0
        b.free()
    with pytest.raises(Exception):
        ctx.bases_generate_label(curve, b"a" * 65, 4)


# ---- direct-sum MSM over a digit table (msm_direct.hip): the small commitments on a prover's critical path ------------
def _digit_edge_scalars(sm, c, n, rng):
    """Scalars whose signed digits sit at the edges of [-2^(c-1), 2^(c-1)): all-ones, runs of 2^(c-1) and 2^(c-1) - 1
    in every window, carries running to the top, the largest scalars of the field."""
    half = 1 << (c - 1)
    vals = [0, 1, 2, half - 1, half, half + 1, (1 << c) - 1, 1 << c, sm - 1, sm - 2, sm >> 1, (sm >> 1) + 1]
    vals += [sum(half << (c * j) for j in range(0, 254 // c)) % sm, sum((half - 1) << (c * j) for j in range(0, 254 // c)) % sm,
             ((1 << 254) - 1) % sm, (1 << 253), (1 << 254) % sm]
    vals += [int(x) for x in rng.integers(0, 2, size=max(0, n - len(vals)))]           # a witness's bits
    return limbs(vals[:n])


@pytest.mark.parametrize("curve", CURVES)
@pytest.mark.parametrize("c", [8, 9, 10, 11, 12])
def test_digit_table_msm_equals_the_bucket_method(ctx, cref, curve, c):
    nb = 700
    sm = o.curve_scalar_modulus(curve)
    bases = ctx.bases_generate(curve, 5, nb)
    pts = bases.download()
    bases.precompute(15, 1)
    ranges = [(0, 300), (450, 250)]
    bases.precompute_digits(ranges, c)
    assert bases.digit_window == c
    rng = np.random.default_rng(c)
    cases = {"random": rand_limbs(rng, 300), "edges": _digit_edge_scalars(sm, c, 300, rng)}
    cases.update({k: v for k, v in _distributions(300, sm, rng).items()})
    for name, sc in cases.items():
        for is_mont in (False, True):
            s_in = mont(ints(sc), sm) if is_mont else sc
            got = jac_to_affine(ctx.msm(bases, s_in, is_mont=is_mont), curve)
            assert got == cpu_msm(cref, curve, pts[:300].copy(), sc), (name, is_mont)
    # sub-ranges, the second range, a batch across both; a vector that leaves the ranges takes the bucket method
    sc = [rand_limbs(rng, n) for n in (250, 120, 1, 300)]
    offs = [450, 37, 699, 0]
    got = ctx.msm_batch(bases, sc, offsets=offs)
    for g in range(4):
        assert jac_to_affine(got[g], curve) == cpu_msm(cref, curve, pts[offs[g]:offs[g] + len(sc[g])].copy(), sc[g]), g
    wide = rand_limbs(rng, 500)
    assert jac_to_affine(ctx.msm(bases, wide, offset=100), curve) == cpu_msm(cref, curve, pts[100:600].copy(), wide)
    mixed = ctx.msm_batch(bases, [sc[0], wide], offsets=[450, 100])
    assert jac_to_affine(mixed[0], curve) == cpu_msm(cref, curve, pts[450:700].copy(), sc[0])
    assert jac_to_affine(mixed[1], curve) == cpu_msm(cref, curve, pts[100:600].copy(), wide)
    # empty vectors and an all-zero vector give the identity
    z = ctx.msm_batch(bases, [np.zeros((0, 4), dtype="<u8"), np.zeros((9, 4), dtype="<u8")], offsets=[0, 460])
    assert jac_to_affine(z[0], curve) is None and jac_to_affine(z[1], curve) is None
    with pytest.raises(Exception):
        bases.precompute_digits([(0, 300), (299, 10)], c)              # overlapping ranges
    with pytest.raises(Exception):
        bases.precompute_digits([(0, nb + 1)], c)                      # beyond the generators
    with pytest.raises(Exception):
        bases.precompute_digits([(0, 10)], 13)                         # window out of range
    bases.precompute_digits([])
    assert bases.digit_window == 0
    bases.free()


def test_digit_table_msm_at_witness_size_dlog_identity(ctx):
    """The shape a Nova step waits for: two ~10^4-term vectors in one call (a witness full of bits and small values, and a
    random cross term), device-resident, through the digit table; the discrete-log identity of the synthetic generators
    needs no CPU MSM."""
    import torch
    curve, n = o.CURVE_VESTA, 10049
    sm = o.curve_scalar_modulus(curve)
    bases = ctx.bases_generate(curve, 0x4E6F7661, n)
    bases.precompute_digits([(0, n)])
    rng = np.random.default_rng(3)
    w = rand_limbs(rng, n)
    bits = rng.random(n) < 0.5
    w[bits] = 0
    w[bits, 0] = rng.integers(0, 2, size=int(bits.sum()), dtype=np.uint64)
    t = rand_limbs(rng, n)
    dw, dt = (torch.from_numpy(x.view(np.int64)).cuda() for x in (w, t))
    got = ctx.msm_batch(bases, [dw, dt])
    ctx.sync()
    for vec, res in ((w, got[0]), (t, got[1])):
        assert jac_to_affine(res, curve) == o.msm_by_dlog_limbs(vec, curve, 0x4E6F7661)
    bases.free()


def test_lazy_addition_chains_stay_inside_their_bound():
    """ec.cuh xyzz_madd_lazy (the bucket loops' sign-tracked mixed addition): fe_mul2_lazy / fe_neg_lazy against the canonical
    operations, short chains step by step, and 10,240-long chains per lane -- with cancellations to the identity every 997
    additions and further additions behind them -- in which every stored coordinate must stay below 2m + 2^130 (the proven
    bound is 2m + 9 eps, eps ~ 2^125) and the resolved sum must equal the canonical one.  tools/ubench/madd_check.hip, run as
    a child process (built by vdf_amd/csrc/Makefile)."""
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "ubench", "madd_check")
    assert os.path.exists(exe), "make -C vdf_amd/csrc"
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "coordinates above 2m + 2^130: 0, mismatches 0" in r.stdout, r.stdout
    assert "bit for bit (524288 values): mismatches 0" in r.stdout, r.stdout           # fe_sqr_lazy (the additions' two squarings)


@pytest.mark.parametrize("curve", CURVES)
def test_endomorphism_split_on_edge_scalars(ctx, cref, curve):
    """The table-less path over a whole generator set of 2^12 points and more splits every scalar k = k1 + lambda k2 (msm.hip
    k_glv_split; constants generated by tools/gen_constants.py) and runs 2n points [P | phi(P)] with 127-bit half-scalars:
    scalars at the edges of the split's arithmetic -- 0, 1, 2, r - 1, r - 2, lambda, lambda +- 1, (r - 1) / 2, powers of two
    around 2^127 / 2^128 / 2^254, all-ones words -- against the C restatement, and the same vector with the endomorphism off."""
    from vdf_amd.hip import tuning_get, tuning_set
    r = o.curve_scalar_modulus(curve)
    n = 4096 + 37
    lam = pow(5, (r - 1) // 3, r)                               # a primitive cube root of unity in the scalar field
    edges = [0, 1, 2, r - 1, r - 2, lam, lam + 1, lam - 1, (lam * lam) % r, (r - 1) // 2, (r + 1) // 2, 1 << 126, 1 << 127, (1 << 127) - 1,
             (1 << 127) + 1, 1 << 128, (1 << 128) - 1, 1 << 129, 1 << 253, (1 << 254) - 1, 1 << 254, (1 << 254) + 1, r // 3, 2 * r // 3,
             (1 << 64) - 1, ((1 << 192) - 1) % r]
    rng = np.random.default_rng(7 + curve)
    sc = rand_limbs(rng, n)
    sc[:len(edges)] = limbs([e % r for e in edges])
    sc[n - len(edges):] = limbs([(r - e) % r for e in edges])
    bases = ctx.bases_generate(curve, 45, n)
    pts = bases.download()
    want = cpu_msm(cref, curve, pts, sc)
    assert tuning_get().glv == 1
    assert jac_to_affine(ctx.msm(bases, sc), curve) == want
    assert jac_to_affine(ctx.msm(bases, mont(ints(sc), r), is_mont=True), curve) == want         # Montgomery-form scalars
    tuning_set(glv=0)
    try:
        assert jac_to_affine(ctx.msm(bases, sc), curve) == want
    finally:
        tuning_set(glv=1)
    # a prefix of the set does not take the endomorphism (its point array is the whole set's): same answer
    assert jac_to_affine(ctx.msm(bases, sc[:4100], n=4100), curve) == cpu_msm(cref, curve, pts[:4100], sc[:4100])
    bases.free()
