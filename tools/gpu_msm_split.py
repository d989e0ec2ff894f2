"""One MSM at a time, as one pipeline (vdf_msm) and as an MSM job of k point-chunks (vdf_msm_job_*: chunk g + 1 is sorted under
chunk g's bucket accumulation, one shared bucket reduction; the k partial points are summed by vdf_point_sum), at 2^k points
over a fixed-base table.  usage: gpu_msm_split.py [log2n ...]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import pasta as o
import vdf_amd as v

ctx = v.Context(0)
for lg in [int(a) for a in sys.argv[1:]] or [20, 22, 24]:
    n = 1 << lg
    bases = ctx.bases_generate(v.CURVE_PALLAS, 7, n, family=v.GENS_KNOWN_DLOG)      # (the fast family: seconds at 2^24)
    bases.precompute(0, 1)
    g = torch.Generator(device="cuda"); g.manual_seed(lg)
    sc = torch.randint(-(2**63), 2**63 - 1, (n, 4), dtype=torch.int64, device="cuda", generator=g); sc[:, 3] &= 0x3FFFFFFFFFFFFFFF
    torch.cuda.synchronize()
    def aff(t):
        j = v.limbs_to_ints(np.ascontiguousarray(t).view("<u8").reshape(3, 4))
        X, Y, Z = (o.from_mont(x, o.P) for x in j)
        zi = pow(Z, -1, o.P)
        return (X * zi * zi % o.P, Y * zi * zi * zi % o.P)
    ctx.set_async(False)
    ref = None
    def plain():
        return ctx.msm(bases, sc, n=n)
    def split(k):
        ch = n // k
        job = ctx.msm_job(bases, [ch] * k, [i * ch for i in range(k)])
        for i in range(k):
            job.push(i, sc[i * ch:(i + 1) * ch])
        outs = job.finish()
        return ctx.point_sum(v.CURVE_PALLAS, outs, k)
    for name, fn in (("vdf_msm", plain), ("job x2", lambda: split(2)), ("job x4", lambda: split(4))):
        r = fn(); r = fn()
        reps = 8 if lg <= 22 else 4
        t0 = time.perf_counter()
        for _ in range(reps): r = fn()
        dt = (time.perf_counter() - t0) / reps
        a = aff(r)
        if ref is None: ref = a
        print(f"2^{lg} {name:8s}: {dt*1e3:8.3f} ms = {n/dt/1e9:.3f} GPoints/s; same point: {a == ref}", flush=True)
    bases.free()
