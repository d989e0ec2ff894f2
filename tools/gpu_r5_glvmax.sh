#!/bin/bash
# The endomorphism beyond 2^21 points (an A/B build with -DVDF_GLV_MAX_LOG2=23 against the shipped limit), table-less MSMs back to back.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/r5
for round in 1 2; do for which in shipped ab; do
  if [ $which = ab ]; then export VDF_HIP_LIB=$R/vdf_amd/csrc/build/ab/libvdf_hip.so; else unset VDF_HIP_LIB; fi
  echo "== round $round $which"
  timeout -k 10 200 python3 - <<'P'
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch, vdf_amd, bench as B
ctx = vdf_amd.Context(0)
for lg in (21, 22, 23):
    n = 1 << lg
    bases = ctx.bases_generate(vdf_amd.CURVE_PALLAS, 11, n)
    g = torch.Generator(device="cuda"); g.manual_seed(100 + lg)
    sc = torch.randint(-(2**63), 2**63 - 1, (n, 4), dtype=torch.int64, device="cuda", generator=g); sc[:, 3] &= 0x3FFFFFFFFFFFFFFF
    res = torch.zeros(12, dtype=torch.int64, device="cuda")
    ctx.set_async(True)
    for _ in range(3): ctx.msm(bases, sc, n=n, out=res)
    ctx.sync()
    a = time.perf_counter()
    for _ in range(6): ctx.msm(bases, sc, n=n, out=res)
    ctx.sync()
    wall = (time.perf_counter() - a) / 6 * 1e3
    ctx.set_async(False)
    want = B._scalar_mul_generator(B._sum_s_k(sc.cpu().numpy().view("<u8"), B._dlogs(11, 0, n)) % B._Q, B._P) if lg <= 22 else None
    ok = (B._jac_to_affine_ints(res.cpu().numpy().view("<u8").tobytes(), B._P) == want) if want is not None else "-"
    print("table-less 2^%d: %.3f ms back to back = %.4f GPoints/s exact %s" % (lg, wall, n / wall / 1e6, ok), flush=True)
    bases.free()
P
done; done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5/glv_max.txt
