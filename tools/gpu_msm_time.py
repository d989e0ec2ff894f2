"""Timing probe: MSM at 2^k with device-resident scalars (torch), plain and fixed-base-table modes."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import pasta as o
import vdf_amd as v

arg = sys.argv[1] if len(sys.argv) > 1 else "20"
n = int(arg[1:]) if arg.startswith("n") else 1 << int(arg)       # "18" = 2^18 points, "n196615" = that many
lg = n.bit_length() - 1
modes = sys.argv[2].split(",") if len(sys.argv) > 2 else ["plain", "tbl16x1", "tbl16x4"]
ctx = v.Context(0)
curve = v.CURVE_PALLAS
sm, bm_ = o.Q, o.P
t0 = time.time(); bases = ctx.bases_generate(curve, 7, n); print("bases_generate %.1f ms" % ((time.time() - t0) * 1e3))
g = torch.Generator(device="cuda"); g.manual_seed(1)
sc = torch.randint(0, 2**63 - 1, (n, 4), dtype=torch.int64, device="cuda", generator=g)
sc = sc * 2 + torch.randint(0, 2, (n, 4), dtype=torch.int64, device="cuda", generator=g)   # full 64-bit limbs
sc[:, 3] &= 0x3FFFFFFFFFFFFFFF   # < 2^254 < q
torch.cuda.synchronize()
sc_host = sc.cpu().numpy().view("<u8")
ints = v.limbs_to_ints(sc_host)
t0 = time.time(); exp = o.msm_by_dlog(ints, curve, 7); print("oracle dlog check %.1f s" % (time.time() - t0))
out = torch.zeros(12, dtype=torch.int64, device="cuda")

def check(tag):
    j = v.limbs_to_ints(out.cpu().numpy().view("<u8").reshape(3, 4))
    X, Y, Z = (o.from_mont(t, bm_) for t in j)
    zi = pow(Z, -1, bm_); got = (X * zi * zi % bm_, Y * zi * zi * zi % bm_)
    print(tag, "parity", got == exp)

for mode in modes:
    if mode.startswith("tbl"):
        c, s = mode[3:].split("x")
        t0 = time.time(); bases.precompute(int(c), int(s)); print(mode, "precompute %.1f ms" % ((time.time() - t0) * 1e3))
        ctx.set_msm_window(0)
    else:
        if mode.startswith("plain") and len(mode) > 5: ctx.set_msm_window(int(mode[5:]))
        bases.precompute(16, 16)   # tables == 1 -> drops the table
    ctx.set_async(False)
    ctx.msm(bases, sc, n=n, out=out); check(mode)
    ctx.set_async(True)
    st = torch.cuda.ExternalStream(ctx.stream)
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    reps = 5
    with torch.cuda.stream(st):
        e0.record(st)
        for _ in range(reps): ctx.msm(bases, sc, n=n, out=out)
        e1.record(st)
    ctx.sync(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"{mode}: n={n} {ms:.3f} ms/MSM  {n / ms / 1e6:.3f} GPoints/s  HBM-alg {96 * n / ms / 1e6:.1f} GB/s")
    ctx.set_timing(True); ctx.msm_timing()
    for _ in range(reps): ctx.msm(bases, sc, n=n, out=out)
    ctx.sync()
    s_, a_, t_, tot_, calls = ctx.msm_timing()
    ctx.set_timing(False)
    print(f"{mode}: stages sort {s_/calls:.3f} accumulate {a_/calls:.3f} tail {t_/calls:.3f} ms")
