"""The early rows' cross term alone on the device at t = 2^k: the generic sparse kernel (vdf_nifs_cross_term_rows over the
primary shape's CSR) against the MinRoot stencil (vdf_nifs_cross_term_minroot), same rows, same vectors; HIP events around the
launches (vdf_ctx_kernel_events), results compared."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import vdf_amd
from vdf_amd.nova import shape_export, shape_stencil, shape_digest

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 16
t = 1 << lg
ctx = vdf_amd.Context(0)
for kind, name in ((1, "reference"), (0, "bound")):
    per, row0, nrows, seg = shape_stencil(t, kind)
    _, sizes = shape_digest(t, kind, 1)
    nc, nv = sizes[0][0], sizes[0][1]
    mats = shape_export(t, kind, 0)
    shape = ctx.shape_create(vdf_amd.FIELD_FQ, nc, nv + 3, mats)
    g = torch.Generator(device="cuda"); g.manual_seed(kind)
    def rnd(n):
        v = torch.randint(-(2**63), 2**63 - 1, (n, 4), dtype=torch.int64, device="cuda", generator=g)
        v[:, 3] &= 0x0FFFFFFFFFFFFFFF
        return v
    z2, a1, b1, c1 = rnd(nv + 3), rnd(nc), rnd(nc), rnd(nc)
    # the constant's column holds ONE (Montgomery form of 1 in Fq), as in every fresh instance: the prover's case
    z2[nv] = torch.tensor(np.array([0x5b2b3e9cfffffffd, 0x992c350be3420567, 0xffffffffffffffff, 0x3fffffffffffffff], dtype="<u8").view(np.int64), device="cuda")
    u1 = np.array([[5, 6, 7, 8]], dtype="<u8")
    outs = [[torch.zeros((nc, 4), dtype=torch.int64, device="cuda") for _ in range(4)] for _ in range(2)]
    ctx.set_async(True)
    res = {}
    for which in (0, 1):
        o = outs[which]
        call = (lambda: ctx.nifs_cross_term_rows(shape, row0, nrows, 1, z2, a1, b1, c1, u1, o[0], o[1], o[2], o[3])) if which == 0 else \
               (lambda: ctx.nifs_cross_term_minroot(vdf_amd.FIELD_FQ, per, t, seg, nv, row0, z2, a1, b1, c1, u1, o[0], o[1], o[2], o[3]))
        for _ in range(3): call()
        ctx.sync(); ctx.set_kernel_timing(True); ctx.kernel_events()
        for _ in range(10): call()
        ctx.sync()
        ev = [e for e in ctx.kernel_events() if e[0].startswith("k_nifs_cross")]
        ctx.set_kernel_timing(False)
        us = sorted((e[3] - e[2]) * 1e3 for e in ev)
        res[which] = (ev[0][0], us[len(us) // 2], ev[0][1])
    same = all(torch.equal(outs[0][k][row0:row0 + nrows], outs[1][k][row0:row0 + nrows]) for k in range(4))
    for which in (0, 1):
        nm, us, nb = res[which]
        print(f"{name} circuit, t = 2^{lg}, {nrows} rows: {nm:24s} {us:7.1f} us alone (median of 10), {nb / 1e6:.1f} MB algorithmic = {nb / us / 1e6:.2f} TB/s")
    print("  identical results:", same)
    shape.free()
