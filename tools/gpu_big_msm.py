"""Large single-GPU MSMs (2^22, 2^24 points, fixed-base table, recommended window): timing, and parity through
size-independent identities -- the discrete-log identity of the [k_i]G family at 2^22, and at 2^24
MSM(all) == sum of the four quarter MSMs (offsets into the same table) via vdf_point_sum."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import pasta as o
import vdf_amd as v

ctx = v.Context(0)
curve = v.CURVE_PALLAS
for lg in [int(a) for a in sys.argv[1:]] or [22, 24]:
    n = 1 << lg
    t0 = time.time(); bases = ctx.bases_generate(curve, 7, n); bases.precompute(0, 1)
    print(f"2^{lg}: bases + table {time.time() - t0:.1f} s ({16 * n * 64 / 2**30:.1f} GiB table)", flush=True)
    g = torch.Generator(device="cuda"); g.manual_seed(lg)
    sc = torch.randint(-(2**63), 2**63 - 1, (n, 4), dtype=torch.int64, device="cuda", generator=g)
    sc[:, 3] &= 0x3FFFFFFFFFFFFFFF
    out = torch.zeros(12, dtype=torch.int64, device="cuda")
    ctx.set_async(False)
    ctx.msm(bases, sc, n=n, out=out)
    full = v.hip.jac_words_to_affine(out.cpu().numpy().view("<u8"), curve) if hasattr(v.hip, "jac_words_to_affine") else None
    ctx.set_async(True)
    st = torch.cuda.ExternalStream(ctx.stream)
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    reps = 3
    with torch.cuda.stream(st):
        e0.record(st)
        for _ in range(reps): ctx.msm(bases, sc, n=n, out=out)
        e1.record(st)
    ctx.sync(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"2^{lg}: {ms:.3f} ms/MSM = {n / ms / 1e6:.3f} GPoints/s", flush=True)
    ctx.set_async(False)
    q = n // 4
    parts = torch.zeros((4, 12), dtype=torch.int64, device="cuda")
    for k in range(4):
        ctx.msm(bases, sc[k * q:(k + 1) * q], n=q, offset=k * q, out=parts[k])
    tot = torch.zeros(12, dtype=torch.int64, device="cuda")
    ctx.point_sum(curve, parts, 4, out=tot)
    def aff(t):
        j = v.limbs_to_ints(t.cpu().numpy().view("<u8").reshape(3, 4))
        X, Y, Z = (o.from_mont(x, o.P) for x in j)
        zi = pow(Z, -1, o.P)
        return (X * zi * zi % o.P, Y * zi * zi * zi % o.P)
    print(f"2^{lg}: MSM(all) == sum of quarter MSMs: {aff(out) == aff(tot)}", flush=True)
    if lg <= 22:
        t0 = time.time()
        ints = v.limbs_to_ints(sc.cpu().numpy().view("<u8"))
        print(f"2^{lg}: discrete-log identity: {aff(out) == o.msm_by_dlog(ints, curve, 7)} ({time.time() - t0:.0f} s of host big-int work)", flush=True)
    bases.free(); del sc
    torch.cuda.empty_cache()
