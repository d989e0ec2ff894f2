"""A table-less (variable-base) MSM of 2^lg points, `reps` times, for a kernel trace of its tail:
   rocprofv3 --kernel-trace --stats -d gpurun_out/r5/tl -- python3 tools/gpu_tableless_run.py 20 10"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, vdf_amd
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 20
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
n = 1 << lg
ctx = vdf_amd.Context(0)
bases = ctx.bases_generate(vdf_amd.CURVE_PALLAS, 11, n)
g = torch.Generator(device="cuda"); g.manual_seed(100 + lg)
sc = torch.randint(-(2**63), 2**63 - 1, (n, 4), dtype=torch.int64, device="cuda", generator=g); sc[:, 3] &= 0x3FFFFFFFFFFFFFFF
res = torch.zeros(12, dtype=torch.int64, device="cuda")
ctx.set_async(True)
for _ in range(3): ctx.msm(bases, sc, n=n, out=res)
ctx.sync()
a = time.perf_counter()
for _ in range(reps): ctx.msm(bases, sc, n=n, out=res); ctx.sync()
print("table-less 2^%d: %.3f ms per call" % (lg, (time.perf_counter() - a) / reps * 1e3))
