"""The zero-change drop-in (mult_pippenger_pallas with host pointers) with and without the generator cache."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import vdf_amd
from vdf_amd._lib import lib

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 19
n = 1 << lg
ctx = vdf_amd.Context(0)
b = ctx.bases_generate(vdf_amd.CURVE_PALLAS, 7, n)
pts = b.download()
rng = np.random.default_rng(1)
out = np.zeros(12, dtype="<u8")
for cache in (0, 4):
    assert lib.vdf_shim_set_cache(cache) == 0
    ts = []
    for k in range(6):
        sc = rng.integers(0, 2**64, size=(n, 4), dtype=np.uint64); sc[:, 3] &= np.uint64(0x3FFFFFFFFFFFFFFF)
        t0 = time.perf_counter()
        lib.mult_pippenger_pallas(out.ctypes.data, pts.ctypes.data, n, sc.ctypes.data, False)
        ts.append(1e3 * (time.perf_counter() - t0))
    print(f"2^{lg} points, cache {cache}: call times ms", [round(t, 2) for t in ts])
lib.vdf_shim_set_cache(0)
