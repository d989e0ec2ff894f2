"""Throughput probe: dependent Montgomery multiplications per second (whole chip)."""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import vdf_amd as v
ctx = v.Context(0)
for field in (0, 1):
    for nthreads, iters in ((256 * 4 * 64 * 4, 2000), (256 * 4 * 64 * 8, 2000), (256 * 4 * 64 * 1, 2000)):
        a = torch.randint(0, 2**62, (nthreads, 4), dtype=torch.int64, device="cuda")
        out = torch.zeros_like(a)
        ctx.set_async(True)
        st = torch.cuda.ExternalStream(ctx.stream)
        ctx.fe_mul_chain(field, a, nthreads, 10, out); ctx.sync()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(st):
            e0.record(st); ctx.fe_mul_chain(field, a, nthreads, iters, out); e1.record(st)
        ctx.sync(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        rate = nthreads * iters / (ms * 1e-3)
        print(f"field {field} waves/SIMD {nthreads // (256*4*64)}: {ms:.3f} ms  {rate/1e9:.1f} Gmul/s  => {1024*64*2.4e9/rate:.0f} cycles/mul/SIMD-wave")
