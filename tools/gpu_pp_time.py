"""Wall time of public_params(t): generators by try-and-increment + fixed-base table + R1CS shape upload."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vdf_amd
from vdf_amd.nova import public_params

ctx = vdf_amd.Context(0)
for t in (1 << 10, 1 << 16, 1 << 16):
    t0 = time.perf_counter()
    pp = public_params(ctx, t)
    ctx.sync()
    print(f"t={t}: public_params {1e3 * (time.perf_counter() - t0):.1f} ms, sizes {pp.sizes()}")
    pp.free()
