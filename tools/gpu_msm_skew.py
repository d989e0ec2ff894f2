"""MSM time at 2^k for skewed scalar distributions (fixed-base table c = 16), checked with the discrete-log identity."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import pasta as o
import vdf_amd as v
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << lg
ctx = v.Context(0)
bases = ctx.bases_generate(v.CURVE_PALLAS, 7, n); bases.precompute(16, 1)
rng = np.random.default_rng(1)
def uniform():
    a = rng.integers(0, 2**64, size=(n, 4), dtype=np.uint64); a[:, 3] &= np.uint64(0x3FFFFFFFFFFFFFFF); return a
cases = {"uniform": uniform()}
a = uniform(); a[:] = a[0]; cases["all equal"] = a
a = np.zeros((n, 4), dtype=np.uint64); a[:, 0] = rng.integers(0, 2, size=n, dtype=np.uint64); cases["bits (0/1)"] = a
a = np.zeros((n, 4), dtype=np.uint64); a[:, 0] = rng.integers(0, 1 << 16, size=n, dtype=np.uint64); cases["16-bit"] = a
a = uniform(); a[: n // 2] = 0; cases["half zeros"] = a
a = uniform(); a[:, 0] &= np.uint64(0xFFFFFFFFFFFF0000); a[:, 0] |= np.uint64(5); cases["same low digit"] = a
def aff(words):
    j = v.limbs_to_ints(np.ascontiguousarray(words).view("<u8").reshape(3, 4))
    X, Y, Z = (o.from_mont(x, o.P) for x in j)
    if Z == 0: return None
    zi = pow(Z, -1, o.P); return (X * zi * zi % o.P, Y * zi * zi * zi % o.P)
for name, sc in cases.items():
    d = torch.from_numpy(sc.view(np.int64)).cuda()
    out = torch.zeros(12, dtype=torch.int64, device="cuda")
    ctx.msm(bases, d, n=n, out=out); ctx.sync()
    t0 = time.perf_counter()
    for _ in range(3): ctx.msm(bases, d, n=n, out=out)
    ctx.sync()
    ms = (time.perf_counter() - t0) / 3 * 1e3
    ok = aff(out.cpu().numpy()) == o.msm_by_dlog(v.limbs_to_ints(sc), v.CURVE_PALLAS, 7) if lg <= 18 else "(not checked)"
    print(f"{name:16s} {ms:8.3f} ms  parity {ok}", flush=True)
