"""Randomised sizes through the element-wise and folding kernels against big-integer arithmetic: fe_mul, to/from
Montgomery, axpy, cross_term, fold_many (1-8 vectors of different lengths), fold_halves, dot / IPA-cross reductions,
ipa_scalars, scale_pattern, pair_table.  A one-off hunt for boundary bugs (n = 0, 1, odd, just past a workgroup)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
from oracle import pasta as o
import vdf_amd as v
from util import mont, unmont, rand_limbs, ints

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 150
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = v.Context(0)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).cuda()
host = lambda t: t.cpu().numpy().view("<u8")
SIZES = [0, 1, 2, 3, 63, 64, 65, 255, 256, 257, 511, 1000, 4097, 70001]


def vec(n, m):
    vals = [int(x) % m for x in ints(rand_limbs(rng, n))] if n else []
    return vals, dev(mont(vals, m) if n else np.zeros((0, 4), dtype="<u8"))


bad = 0
t0 = time.time()
for it in range(iters):
    field = int(rng.integers(0, 2)); m = o.modulus(field)
    n = int(rng.choice(SIZES))
    ok = True
    a, da = vec(n, m); b, db = vec(n, m); c, dc = vec(n, m)
    out = dev(np.zeros((max(n, 1), 4), dtype="<u8"))
    ctx.fe_mul(field, da, db, n, out); ctx.sync()
    ok &= unmont(host(out)[:n], m) == [x * y % m for x, y in zip(a, b)]
    r = int(rng.integers(0, 1 << 62)) if rng.random() < 0.7 else int(ints(rand_limbs(rng, 1))[0]) % m
    ctx.axpy(field, da, mont([r], m), db, n, out); ctx.sync()
    ok &= unmont(host(out)[:n], m) == [(x + r * y) % m for x, y in zip(a, b)]
    # fold_many: k vectors of different lengths, acc += r * add
    k = int(rng.integers(1, 9))
    lens = [int(rng.choice(SIZES[:-1])) for _ in range(k)]
    accs, adds = [vec(l, m) for l in lens], [vec(l, m) for l in lens]
    ctx.fold_many(field, mont([r], m), [x[1] for x in accs], [x[1] for x in adds], lens); ctx.sync()
    for (av, ad), (bv, bd), l in zip(accs, adds, lens):
        ok &= unmont(host(ad)[:l], m) == [(x + r * y) % m for x, y in zip(av, bv)]
    # power-of-two kernels
    lg = int(rng.integers(1, 15)); p2 = 1 << lg
    x, dx = vec(p2, m); y, dy = vec(p2, m)
    got = ctx.reduce(field, 0, [dx, dy], p2)
    ok &= unmont(got, m) == [sum(p * q for p, q in zip(x, y)) % m]
    h = p2 // 2
    got = ctx.reduce(field, 3, [dx, dy], p2)
    ok &= unmont(got, m) == [sum(p * q for p, q in zip(x[:h], y[h:])) % m, sum(p * q for p, q in zip(x[h:], y[:h])) % m]
    clo, chi = [int(z) % m for z in ints(rand_limbs(rng, 2))], [int(z) % m for z in ints(rand_limbs(rng, 2))]
    ctx.fold_halves(field, [dx, dy], mont(clo, m), mont(chi, m), p2); ctx.sync()
    ok &= unmont(host(dx)[:h], m) == [(clo[0] * x[i] + chi[0] * x[h + i]) % m for i in range(h)]
    ok &= unmont(host(dy)[:h], m) == [(clo[1] * y[i] + chi[1] * y[h + i]) % m for i in range(h)]
    # ipa_scalars / scale_pattern on a vector of n_tot with a sub-period nj
    lgn = int(rng.integers(1, 13)); nt = 1 << lgn; nj = 1 << int(rng.integers(1, lgn + 1)); hj = nj // 2
    s_, ds = vec(nt, m); aa, daa = vec(nj, m)
    sL, sR = dev(np.zeros((nt, 4), dtype="<u8")), dev(np.zeros((nt, 4), dtype="<u8"))
    ctx.ipa_scalars(field, daa, ds, nt, nj, sL, sR); ctx.sync()
    eL = [s_[t] * aa[(t % nj) - hj] % m if (t % nj) >= hj else 0 for t in range(nt)]
    eR = [s_[t] * aa[(t % nj) + hj] % m if (t % nj) < hj else 0 for t in range(nt)]
    ok &= unmont(host(sL), m) == eL and unmont(host(sR), m) == eR
    xl, xh = int(ints(rand_limbs(rng, 1))[0]) % m, int(ints(rand_limbs(rng, 1))[0]) % m
    ctx.scale_pattern(field, ds, nt, nj, mont([xl], m), mont([xh], m)); ctx.sync()
    ok &= unmont(host(ds), m) == [s_[t] * (xh if (t % nj) >= hj else xl) % m for t in range(nt)]
    # cross term
    vs = [vec(n, m) for _ in range(6)]
    u1 = int(ints(rand_limbs(rng, 1))[0]) % m
    ctx.cross_term(field, *[z[1] for z in vs], mont([u1], m), n, out); ctx.sync()
    A1, B1, C1, A2, B2, C2 = (z[0] for z in vs)
    ok &= unmont(host(out)[:n], m) == [(A1[i] * B2[i] + A2[i] * B1[i] - u1 * C2[i] - C1[i]) % m for i in range(n)]
    if not ok:
        bad += 1
        print("MISMATCH", dict(it=it, field=field, n=n, k=k, lens=lens, lg=lg, nt=nt, nj=nj), flush=True)
    if it % 25 == 24:
        print(f"{it + 1} rounds, {bad} mismatches, {time.time() - t0:.0f} s", flush=True)
print("vector fuzz done:", iters, "rounds,", bad, "mismatches")
sys.exit(1 if bad else 0)
