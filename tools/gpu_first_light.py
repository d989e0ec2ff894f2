"""First-light check on a GPU box: field ops, vector ops, MSM vs the Python oracle."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from oracle import pasta as o
import vdf_amd as v

ctx = v.Context(0)
print(v._lib.lib.vdf_version())
ok = True
for field, m in ((v.FIELD_FP, o.P), (v.FIELD_FQ, o.Q)):
    n = 1000
    a = [o.rand_fe(1, i, m) for i in range(n)]; b = [o.rand_fe(2, i, m) for i in range(n)]
    a[0], b[0] = 0, 5; a[1], b[1] = m - 1, m - 1; a[2], b[2] = 1, m - 1
    am = v.ints_to_limbs([o.to_mont(x, m) for x in a]); bm = v.ints_to_limbs([o.to_mont(x, m) for x in b])
    out = np.zeros_like(am)
    ctx.fe_mul(field, am, bm, n, out)
    got = [o.from_mont(x, m) for x in v.limbs_to_ints(out)]
    exp = [x * y % m for x, y in zip(a, b)]
    print("fe_mul field", field, got == exp); ok &= got == exp
    r = v.ints_to_limbs([o.to_mont(12345678901234567890123, m)])
    ctx.axpy(field, am, r, bm, n, out)
    got = [o.from_mont(x, m) for x in v.limbs_to_ints(out)]
    exp = o.axpy(a, 12345678901234567890123, b, m)
    print("axpy", got == exp); ok &= got == exp

# MSM small sizes, both curves
for curve in (v.CURVE_PALLAS, v.CURVE_VESTA):
    bm_, sm = o.curve_base_modulus(curve), o.curve_scalar_modulus(curve)
    for n in (1, 2, 127, 1000, 5000):
        bases = ctx.bases_generate(curve, 7, n)
        if n <= 127:
            host = bases.download()
            pts = v.limbs_to_ints(host.reshape(-1, 4))
            exp_pts = o.synthetic_bases(curve, 7, n)
            got_pts = [(o.from_mont(pts[2 * i], bm_), o.from_mont(pts[2 * i + 1], bm_)) for i in range(n)]
            print("bases", curve, n, got_pts == exp_pts); ok &= got_pts == exp_pts
        sc = [o.rand_fe(3, i, sm) for i in range(n)]
        if n >= 4: sc[0] = 0; sc[1] = 1; sc[2] = sm - 1; sc[3] = sc[4 % n]
        exp = o.msm_by_dlog(sc, curve, 7)
        for is_mont in (False, True):
            arr = v.ints_to_limbs([o.to_mont(s, sm) if is_mont else s for s in sc])
            t0 = time.time()
            j = v.limbs_to_ints(ctx.msm(bases, arr, is_mont=is_mont).reshape(3, 4))
            dt = time.time() - t0
            X, Y, Z = (o.from_mont(t, bm_) for t in j)
            if Z == 0: got = None
            else:
                zi = pow(Z, -1, bm_); got = (X * zi * zi % bm_, Y * zi * zi * zi % bm_)
            print("msm", curve, n, is_mont, got == exp, "%.1f ms" % (dt * 1e3)); ok &= got == exp
print("ALL OK" if ok else "FAILURES")
sys.exit(0 if ok else 1)
