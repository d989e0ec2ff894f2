"""Throughput probe: independent 2^20 MSMs over one table issued as batches of `b` (vdf_msm_batch: one sort, one
accumulate grid, one tail for the batch) on `depth` contexts, against one MSM per call."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import pasta as o
import vdf_amd as v

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << lg
K = 24
ctxs = [v.Context(0) for _ in range(3)]
bases = ctxs[0].bases_generate(v.CURVE_PALLAS, 7, n); bases.precompute(0, 1)
g = torch.Generator(device="cuda"); g.manual_seed(1)
scs = []
for k in range(4):
    t = torch.randint(-(2**63), 2**63 - 1, (n, 4), dtype=torch.int64, device="cuda", generator=g); t[:, 3] &= 0x3FFFFFFFFFFFFFFF
    scs.append(t)
torch.cuda.synchronize()
for c in ctxs: c.set_async(True)
def aff(t):
    j = v.limbs_to_ints(np.ascontiguousarray(t).view("<u8").reshape(3, 4))
    X, Y, Z = (o.from_mont(x, o.P) for x in j)
    zi = pow(Z, -1, o.P)
    return (X * zi * zi % o.P, Y * zi * zi * zi % o.P)
exp = [o.msm_by_dlog(v.limbs_to_ints(s.cpu().numpy().view("<u8")), v.CURVE_PALLAS, 7) for s in scs]
for b in (1, 2, 4):
    for depth in (1, 2, 3):
        outs = [torch.zeros((b, 12), dtype=torch.int64, device="cuda") for _ in range(depth)]
        def issue(k):
            if b == 1:
                ctxs[k].msm(bases, scs[0], n=n, out=outs[k][0])
            else:
                ctxs[k].msm_batch(bases, [scs[j] for j in range(b)], [n] * b, [0] * b, out=outs[k])
        for k in range(depth): issue(k)
        for k in range(depth): ctxs[k].sync()
        calls = K // b
        t0 = time.perf_counter()
        for i in range(calls): issue(i % depth)
        for k in range(depth): ctxs[k].sync()
        dt = (time.perf_counter() - t0) / (calls * b)
        ok = all(aff(outs[k][j].cpu().numpy()) == exp[j] for k in range(depth) for j in range(b))
        print(f"batch {b} depth {depth}: {dt*1e3:.3f} ms per MSM = {n/dt/1e9:.3f} GPoints/s; correct: {ok}", flush=True)
