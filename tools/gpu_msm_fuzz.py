"""Randomised MSM configurations against the plain-C CPU restatement (test infrastructure: oracle/): sizes, windows,
bucket sets, plain / table, batches of 1-4 with offsets, scalar distributions, both curves.  A one-off hunt for rare
geometry bugs (bucket-reduction shapes, heavy / giant queues, tiny sets), not part of the suite."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from oracle import pasta as o
from oracle import cref
import vdf_amd as v
from util import jac_to_affine, rand_limbs, limbs

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
only = int(sys.argv[3]) if len(sys.argv) > 3 else -1          # run just this configuration (the others only advance the RNG)
BIG = bool(os.environ.get("FUZZ_BIG"))                         # production-like: 2^17 .. 1.1 M points, table windows 16 / 17, one set
rng = np.random.default_rng(seed)
ctx = v.Context(0)
L = cref.lib()


def cpu_msm(curve, pts, sc):
    out = np.zeros(12, dtype="<u8")
    L.ref_msm(curve, cref.p(np.ascontiguousarray(pts)), cref.p(np.ascontiguousarray(sc)), len(sc), 0, 8, 0, cref.p(out))
    return jac_to_affine(out, curve)


def scalars(n, kind, sm):
    full = rand_limbs(rng, n)
    if kind == 0: return full
    if kind == 1:
        s = np.zeros((n, 4), dtype="<u8"); s[:, 0] = rng.integers(0, 1 << 16, size=n, dtype=np.uint64); return s
    if kind == 2:
        full[rng.random(n) < 0.6] = 0; return full
    if kind == 3: return np.repeat(full[:1], n, axis=0).copy()
    if kind == 4: return limbs([sm - 1 - int(x) for x in rng.integers(0, 5, size=n)])
    return limbs([1 << int(b) for b in rng.integers(0, 254, size=n)])


t0 = time.time()
bad = 0
for it in range(iters):
    curve = int(rng.integers(0, 2))
    sm = o.curve_scalar_modulus(curve)
    nmax = int(rng.choice([7, 60, 500, 3000, 20000, 20000, 300000]))
    ntot = int(rng.integers(1, nmax + 1))
    if BIG:
        ntot = int(rng.integers(1 << 17, 1100000))
    bseed = int(rng.integers(1, 1000))
    c = int(rng.integers(4, 21))
    table = rng.random() < 0.6
    if BIG:
        c, table = int(rng.choice([16, 17, 0])), True
    sets, win = 0, 0
    if table:
        windows = (256 + c - 1) // c if c else 16
        sets = 1 if BIG else int(rng.choice([1, 1, 2, windows]))
    else:
        win = c if rng.random() < 0.7 else 0
    k = int(rng.integers(1, 5))
    offs, lens, scs = [], [], []
    for g in range(k):
        ln = int(rng.integers(1, ntot + 1))
        off = int(rng.integers(0, ntot - ln + 1))
        offs.append(off); lens.append(ln); scs.append(scalars(ln, int(rng.integers(0, 6)), sm))
    single = k == 1 and rng.random() < 0.5
    # digit table (msm_direct.hip) over 1-3 random ranges of a small generator set, random window: vectors inside the ranges
    # take the direct sum, the others (and mixed batches) the bucket method
    digits = None
    if not BIG and ntot <= 3000 and rng.random() < 0.5:
        cuts = sorted(set(int(x) for x in rng.integers(0, ntot + 1, size=int(rng.integers(2, 7)))))
        rngs = [(cuts[i], cuts[i + 1] - cuts[i]) for i in range(0, len(cuts) - 1, 2)][:3]
        rngs = [r for r in rngs if r[1] > 0]
        if rngs:
            digits = (int(rng.integers(4, 12)), rngs)
            if rng.random() < 0.7:                       # mostly: put every vector inside a range
                offs, lens, scs = [], [], []
                for g in range(k):
                    b, cnt = rngs[int(rng.integers(0, len(rngs)))]
                    ln = int(rng.integers(1, cnt + 1)); off = b + int(rng.integers(0, cnt - ln + 1))
                    offs.append(off); lens.append(ln); scs.append(scalars(ln, int(rng.integers(0, 6)), sm))
    if only >= 0 and it != only:
        continue
    if only >= 0 or os.environ.get("FUZZ_VERBOSE"):
        print("config", it, dict(curve=curve, ntot=ntot, c=c, table=table, sets=sets, win=win, k=k, offs=offs, lens=lens, single=single), flush=True)
    bases = ctx.bases_generate(curve, bseed, ntot)
    pts = bases.download()
    if table:
        bases.precompute(c, sets)
        ctx.set_msm_window(0)
    else:
        ctx.set_msm_window(0 if digits else win)
    if digits:
        bases.precompute_digits(digits[1], digits[0])
    if single:
        got = [jac_to_affine(ctx.msm(bases, scs[0], n=lens[0], offset=offs[0]), curve)]
    else:
        out = ctx.msm_batch(bases, scs, lens, offs)
        got = [jac_to_affine(out[g], curve) for g in range(k)]
    exp = [cpu_msm(curve, pts[offs[g]:offs[g] + lens[g]], scs[g]) for g in range(k)]
    if got != exp:
        bad += 1
        print("MISMATCH", dict(it=it, curve=curve, ntot=ntot, c=c, table=table, k=k, offs=offs, lens=lens, digits=digits), flush=True)
    bases.free()
    if it % 50 == 49:
        print(f"{it + 1} configurations, {bad} mismatches, {time.time() - t0:.0f} s", flush=True)
ctx.set_msm_window(0)
print("fuzz done:", iters, "configurations,", bad, "mismatches")
sys.exit(1 if bad else 0)
