"""Long chains at several sizes: every chain must verify, compress and verify again (the ring of fresh-witness slots and
the lookahead are exercised hardest when steps are short and the host runs far ahead of the GPU)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pasta as o
import vdf_amd
from vdf_amd.minroot import PallasVDF, State, FIELD_FQ, EvalMode
from vdf_amd.nova import InverseMinRootCircuit, NovaVDFProof, public_params

ctx = vdf_amd.Context(0)
for t, n, upload in ((8, 400, True), (64, 300, False), (1024, 200, True), (1 << 14, 60, False), (1 << 16, 24, True)):
    pp = public_params(ctx, t)
    initial = State.from_ints(FIELD_FQ, o.rand_fe(t, 0, o.Q), 0, 3)
    z0, circuits = InverseMinRootCircuit.eval_and_make_circuits(PallasVDF.new_with_mode(EvalMode.LTRAddChainSequential), t, n, initial)
    if upload:
        circuits.upload(ctx)
    ctx.set_async(True)
    t0 = time.perf_counter()
    proof = NovaVDFProof.prove_recursively(pp, circuits, t, z0)
    ctx.sync()
    dt = time.perf_counter() - t0
    ctx.set_async(False)
    zi = [initial.x, initial.y, initial.i]
    ok = proof.verify(pp, n, z0, zi)
    snark = proof.compress(pp)
    ok2 = snark.verify(pp, n, z0, zi)
    print(f"t={t} steps={n} uploaded={upload}: {1e3 * dt / n:.3f} ms/step  verify {ok}  compressed verify {ok2}", flush=True)
    assert ok and ok2
    snark.free(); proof.free(); circuits.free(); pp.free()
print("soak ok")
