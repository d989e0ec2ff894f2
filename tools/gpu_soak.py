"""Soak: many folds at a small t with verification, and repeated MSMs of random sizes / modes against the
discrete-log identity -- looks for rare failures (races in the sort, heavy-bucket paths, stream ordering)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import pasta as o
import vdf_amd as v
from vdf_amd.minroot import PallasVDF, State, FIELD_FQ, EvalMode
from vdf_amd.nova import InverseMinRootCircuit, NovaVDFProof, public_params

ctx = v.Context(0)
t0 = time.time()
# 1. 120 folds at t = 256, verified
t, n = 256, 120
pp = public_params(ctx, t)
initial = State.from_ints(FIELD_FQ, 987654321, 0, 0)
z0, circuits = InverseMinRootCircuit.eval_and_make_circuits(PallasVDF.new_with_mode(EvalMode.LTRAddChainSequential), t, n, initial)
circuits.upload(ctx)
ctx.set_async(True)
proof = NovaVDFProof.prove_recursively(pp, circuits, t, z0)
ctx.sync(); ctx.set_async(False)
print("120 folds verify:", proof.verify(pp, n, z0, [initial.x, initial.y, initial.i]), f"{time.time()-t0:.1f}s", flush=True)
proof.free(); pp.free()
# 2. MSMs of random sizes, three paths, vs the discrete-log identity
rng = np.random.default_rng(2024)
nb = 1 << 17
bases = ctx.bases_generate(v.CURVE_PALLAS, 11, nb)
def aff(words):
    j = v.limbs_to_ints(np.ascontiguousarray(words).view("<u8").reshape(3, 4))
    X, Y, Z = (o.from_mont(x, o.P) for x in j)
    if Z == 0: return None
    zi = pow(Z, -1, o.P)
    return (X * zi * zi % o.P, Y * zi * zi * zi % o.P)
bad = 0
for it in range(60):
    if it == 20: bases.precompute(16, 1)
    if it == 40: bases.precompute(13, 2)
    k = int(rng.integers(1, 4))
    sizes = [int(rng.integers(1, nb // 2)) for _ in range(k)]
    offs = [int(rng.integers(0, nb - s)) for s in sizes]
    scs = []
    for s in sizes:
        a = rng.integers(0, 2**64, size=(s, 4), dtype=np.uint64); a[:, 3] &= np.uint64(0x3FFFFFFFFFFFFFFF)
        if it % 7 == 0: a[: s // 2] = a[0]          # heavy bucket
        if it % 11 == 0: a[:, 1:] = 0               # small scalars
        scs.append(a)
    got = ctx.msm_batch(bases, scs, offsets=offs)
    for g in range(k):
        exp = o.msm_by_dlog(v.limbs_to_ints(scs[g]), v.CURVE_PALLAS, 11, start=offs[g])
        if aff(got[g]) != exp:
            bad += 1; print("MISMATCH", it, g, sizes[g], offs[g], flush=True)
print(f"60 random batched MSMs: {bad} mismatches, {time.time()-t0:.1f}s", flush=True)
sys.exit(1 if bad else 0)
