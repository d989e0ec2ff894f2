"""Randomised R1CS shapes against big-integer arithmetic: vdf_spmv3, vdf_nifs_cross_term and vdf_spmv3_t over matrices
with empty rows, repeated entries, heavy rows and heavy columns, few and many distinct coefficients, both fields.
A one-off hunt (like gpu_msm_fuzz.py), not part of the suite."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
from oracle import pasta as o
import vdf_amd as v
from util import mont, unmont, rand_limbs, ints

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = v.Context(0)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).cuda()
host = lambda t: t.cpu().numpy().view("<u8")
bad = 0
t0 = time.time()
for it in range(iters):
    field = int(rng.integers(0, 2))
    m = o.modulus(field)
    rows_n = int(rng.choice([1, 3, 40, 700, 5000]))
    cols_n = int(rng.choice([1, 2, 50, 900, 6000]))
    dict_small = [1, m - 1, 2, int(rng.integers(3, 1 << 60))]
    mats, trip = [], []
    for k in range(3):
        nnz = int(rng.integers(0, 3 * rows_n + 2))
        r = rng.integers(0, rows_n, size=nnz)
        c = rng.integers(0, cols_n, size=nnz)
        if nnz and rng.random() < 0.4:                      # a heavy column and a heavy row
            c[rng.random(nnz) < 0.5] = int(rng.integers(0, cols_n))
            r[rng.random(nnz) < 0.2] = int(rng.integers(0, rows_n))
        if rng.random() < 0.6:
            vals = [dict_small[int(x)] for x in rng.integers(0, 4, size=nnz)]
        else:
            vals = [int(x) for x in ints(rand_limbs(rng, nnz))] if nnz else []
        trip.append(list(zip(r.tolist(), c.tolist(), vals)))
        mats.append((r.astype(np.uint32), c.astype(np.uint32), mont(vals, m) if nnz else np.zeros((0, 4), dtype="<u8")))
    shape = ctx.shape_create(field, rows_n, cols_n, mats)
    z = [int(x) % m for x in ints(rand_limbs(rng, cols_n))]
    zd = dev(mont(z, m))
    outs = [dev(np.zeros((rows_n, 4), dtype="<u8")) for _ in range(3)]
    ctx.spmv3(shape, zd, *outs)
    ctx.sync()
    exp = []
    for k in range(3):
        acc = [0] * rows_n
        for (r_, c_, v_) in trip[k]:
            acc[r_] = (acc[r_] + v_ * z[c_]) % m
        exp.append(acc)
    ok = all(unmont(host(outs[k]), m) == exp[k] for k in range(3))
    # transposed product with eq weights and powers of rho
    eq = [int(x) % m for x in ints(rand_limbs(rng, rows_n))]
    rho = int(rng.integers(1, 1 << 62))
    out_t = dev(np.zeros((cols_n, 4), dtype="<u8"))
    ctx.spmv3_t(shape, dev(mont(eq, m)), mont([rho], m), out_t)
    ctx.sync()
    acc = [0] * cols_n
    for k in range(3):
        pw = pow(rho, k, m)
        for (r_, c_, v_) in trip[k]:
            acc[c_] = (acc[c_] + eq[r_] * v_ % m * pw) % m
    ok = ok and unmont(host(out_t), m) == acc
    # the fused multiply_vec + cross term
    a1, b1, c1 = (dev(mont([int(x) % m for x in ints(rand_limbs(rng, rows_n))], m)) for _ in range(3))
    u1 = int(rng.integers(1, 1 << 62))
    o2 = [dev(np.zeros((rows_n, 4), dtype="<u8")) for _ in range(4)]
    ctx.nifs_cross_term(shape, zd, a1, b1, c1, mont([u1], m), *o2)
    ctx.sync()
    A1, B1, C1 = (unmont(host(x), m) for x in (a1, b1, c1))
    T = [(A1[i] * exp[1][i] + exp[0][i] * B1[i] - u1 * exp[2][i] - C1[i]) % m for i in range(rows_n)]
    ok = ok and all(unmont(host(o2[k]), m) == exp[k] for k in range(3)) and unmont(host(o2[3]), m) == T
    if not ok:
        bad += 1
        print("MISMATCH", dict(it=it, field=field, rows=rows_n, cols=cols_n, nnz=[len(t) for t in trip]), flush=True)
    shape.free()
    if it % 25 == 24:
        print(f"{it + 1} shapes, {bad} mismatches, {time.time() - t0:.0f} s", flush=True)
print("shape fuzz done:", iters, "shapes,", bad, "mismatches")
sys.exit(1 if bad else 0)
