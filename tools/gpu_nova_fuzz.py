"""Random iteration counts and chain lengths through the whole proof layer: prove, verify, compress, both wire formats,
resume from a checkpoint.  One-off hunt for shape-dependent bugs (t is usually a power of two in the tests)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import pasta as o
import vdf_amd
from vdf_amd.minroot import PallasVDF, State, FIELD_FQ, EvalMode
from vdf_amd.nova import InverseMinRootCircuit, NovaVDFProof, CompressedNovaVDFProof, public_params

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = vdf_amd.Context(0)
t0 = time.time()
for it in range(iters):
    t = int(rng.choice([1, 2, 3, 7, 31, 100, 257, 1000, 1023, 4097, 12345]))
    n = int(rng.integers(1, 6))
    i0 = int(rng.choice([0, 1, 5, (1 << 64) - 3, o.Q - 2]))
    pp = public_params(ctx, t)
    initial = State.from_ints(FIELD_FQ, o.rand_fe(it + 1, 0, o.Q), int(rng.integers(0, 2)) * 12345, i0)
    z0, circuits = InverseMinRootCircuit.eval_and_make_circuits(PallasVDF.new_with_mode(EvalMode(int(rng.integers(0, 4)))), t, n, initial)
    if rng.random() < 0.5:
        circuits.upload(ctx)
    zi = [initial.x, initial.y, initial.i]
    cut = int(rng.integers(1, n + 1))
    proof = None
    for k in range(cut):
        proof = NovaVDFProof.prove_step(pp, proof, circuits, k, z0)
    blob = proof.serialize()
    restored = NovaVDFProof.deserialize(pp, blob)
    for k in range(cut, n):
        proof = NovaVDFProof.prove_step(pp, proof, circuits, k, z0)
        restored = NovaVDFProof.prove_step(pp, restored, circuits, k, z0)
    ok = proof.verify(pp, n, z0, zi) and restored.verify(pp, n, z0, zi) and proof.serialize() == restored.serialize()
    snark = proof.compress(pp)
    wire = snark.serialize()
    ok = ok and snark.verify(pp, n, z0, zi) and CompressedNovaVDFProof.deserialize(pp, wire).verify(pp, n, z0, zi)
    ok = ok and not snark.verify(pp, n, z0, [zi[0], zi[1], zi[0]])
    print(f"{it}: t={t} steps={n} i0={i0 if i0 < 1 << 70 else 'q-2'} checkpoint after {cut}: {'ok' if ok else 'FAILED'}", flush=True)
    if not ok:
        sys.exit(1)
    for h in (snark, proof, restored, circuits, pp):
        h.free()
print("nova fuzz done:", iters, "chains,", f"{time.time() - t0:.0f} s")
