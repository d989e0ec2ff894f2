#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r5
mkdir -p $OUT
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_msm.py tests/test_gpu_tuning.py -x -q > $OUT/pytest_glv.txt 2>&1; rc=$?; echo "pytest msm rc=$rc"; tail -3 $OUT/pytest_glv.txt
true
for g in 0 1; do
  echo "== VDF_MSM_GLV=$g"
  VDF_MSM_GLV=$g timeout -k 10 200 python3 - <<'P'
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch, vdf_amd, bench as B
ctx = vdf_amd.Context(0)
for lg in (14, 16, 18, 20, 22):
    n = 1 << lg
    bases = ctx.bases_generate(vdf_amd.CURVE_PALLAS, 11, n)
    g = torch.Generator(device="cuda"); g.manual_seed(100 + lg)
    sc = torch.randint(-(2**63), 2**63 - 1, (n, 4), dtype=torch.int64, device="cuda", generator=g); sc[:, 3] &= 0x3FFFFFFFFFFFFFFF
    res = torch.zeros(12, dtype=torch.int64, device="cuda")
    want = B._scalar_mul_generator(B._sum_s_k(sc.cpu().numpy().view("<u8"), B._dlogs(11, 0, n)) % B._Q, B._P)
    ctx.set_async(True)
    for _ in range(3): ctx.msm(bases, sc, n=n, out=res)
    ctx.sync(); ctx.set_timing(True); ctx.msm_timing()
    a = time.perf_counter()
    for _ in range(5): ctx.msm(bases, sc, n=n, out=res); ctx.sync()
    wall = (time.perf_counter() - a) / 5 * 1e3
    st = ctx.msm_timing(); ctx.set_timing(False); ctx.set_async(False)
    ok = B._jac_to_affine_ints(res.cpu().numpy().view("<u8").tobytes(), B._P) == want
    print("table-less 2^%d: wall %.3f ms  sort %.3f acc %.3f tail %.3f  = %.4f GPoints/s exact %s" % (lg, wall, st[0]/st[4], st[1]/st[4], st[2]/st[4], n / wall / 1e6, ok), flush=True)
    bases.free()
P
done 2>&1 | grep -v amdgpu.ids | tee $OUT/glv_tableless.txt
