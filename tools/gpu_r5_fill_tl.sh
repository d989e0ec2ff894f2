#!/bin/bash
# The table-less path by accumulation fill (automatic = three without a table) and size; then the MSM tests.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/r5
for e in "VDF_MSM_ACC_WG=2" "VDF_MSM_ACC_WG=0" "VDF_MSM_ACC_WG=2" "VDF_MSM_ACC_WG=0"; do
  echo "== $e"
  env $e timeout -k 10 200 python3 - <<'P'
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch, vdf_amd, bench as B
ctx = vdf_amd.Context(0)
for lg in (12, 14, 16, 18, 20, 21, 22):
    n = 1 << lg
    bases = ctx.bases_generate(vdf_amd.CURVE_PALLAS, 11, n)
    g = torch.Generator(device="cuda"); g.manual_seed(100 + lg)
    sc = torch.randint(-(2**63), 2**63 - 1, (n, 4), dtype=torch.int64, device="cuda", generator=g); sc[:, 3] &= 0x3FFFFFFFFFFFFFFF
    res = torch.zeros(12, dtype=torch.int64, device="cuda")
    ctx.set_async(True)
    for _ in range(3): ctx.msm(bases, sc, n=n, out=res)
    ctx.sync()
    a = time.perf_counter()
    for _ in range(8): ctx.msm(bases, sc, n=n, out=res)
    ctx.sync()
    wall = (time.perf_counter() - a) / 8 * 1e3
    ctx.set_async(False)
    print("table-less 2^%d: %.3f ms back to back = %.4f GPoints/s" % (lg, wall, n / wall / 1e6), flush=True)
    bases.free()
P
done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5/fill_tableless.txt
timeout -k 10 600 python -m pytest tests/test_gpu_msm.py tests/test_gpu_tuning.py -x -q > gpurun_out/r5/pytest_fill.txt 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/r5/pytest_fill.txt
