"""Stage times (sort / accumulate / tail, HIP events) of ONE fixed-base MSM at a time for several sizes and windows, every
result checked exactly against [sum s_i k_i] G.  usage: gpu_msm_window_sweep.py "20,22,24" "16,17,18,19,20" [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vdf_amd
import bench as B

sizes = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "20,22").split(",")]
windows = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "16,17,18,19,20").split(",")]
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
ctx = vdf_amd.Context(0)
curve = vdf_amd.CURVE_PALLAS
for lg in sizes:
    n = 1 << lg
    g = torch.Generator(device="cuda"); g.manual_seed(100 + lg)
    sc = torch.randint(-(2**63), 2**63 - 1, (n, 4), dtype=torch.int64, device="cuda", generator=g)
    sc[:, 3] &= 0x3FFFFFFFFFFFFFFF
    want = B._scalar_mul_generator(B._sum_s_k(sc.cpu().numpy().view("<u8"), B._dlogs(11, 0, n)) % B._Q, B._P)
    res = torch.zeros(12, dtype=torch.int64, device="cuda")
    for c in windows:
        bases = ctx.bases_generate(curve, 11, n)
        try:
            bases.precompute(c, 1)
        except Exception as ex:
            print("2^%d window %d: no table (%s)" % (lg, c, ex), flush=True)
            bases.free()
            continue
        ctx.set_async(True)
        try:
            for _ in range(2):
                ctx.msm(bases, sc, n=n, out=res)
            ctx.sync()
        except Exception as ex:
            print("2^%d window %d: %s" % (lg, c, ex), flush=True)
            ctx.set_async(False); bases.free()
            continue
        ctx.set_timing(True); ctx.msm_timing()
        a = time.perf_counter()
        for _ in range(reps):
            ctx.msm(bases, sc, n=n, out=res)
            ctx.sync()
        wall = (time.perf_counter() - a) / reps * 1e3
        st = ctx.msm_timing()
        ctx.set_timing(False); ctx.set_async(False)
        cnt = max(st[4], 1)
        ok = B._jac_to_affine_ints(res.cpu().numpy().view("<u8").tobytes(), B._P) == want
        print("2^%d window %2d: wall %8.3f ms  sort %7.3f  accumulate %7.3f  tail %7.3f  pipeline %7.3f  = %.4f GPoints/s  exact %s" %
              (lg, c, wall, st[0] / cnt, st[1] / cnt, st[2] / cnt, st[3] / cnt, n / wall / 1e6, ok), flush=True)
        bases.free()
    del sc
    torch.cuda.empty_cache()
