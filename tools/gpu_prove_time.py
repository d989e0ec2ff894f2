"""prove_step timing at t = 2^k (BASELINE config 3): per-stage breakdown."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pasta as o
import vdf_amd
from vdf_amd.minroot import PallasVDF, State, FIELD_FQ, EvalMode
from vdf_amd.nova import InverseMinRootCircuit, NovaVDFProof, public_params, CIRCUIT_MINROOT_BOUND, CIRCUIT_MINROOT_REFERENCE
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 16
n = int(sys.argv[2]) if len(sys.argv) > 2 else 6
kind = CIRCUIT_MINROOT_REFERENCE if len(sys.argv) > 3 and sys.argv[3].startswith("ref") else CIRCUIT_MINROOT_BOUND
t = 1 << lg
ctx = vdf_amd.Context(0)
print("step circuit:", "reference (4 variables per round)" if kind else "bound (3 variables per round)")
t0 = time.time(); pp = public_params(ctx, t, kind); print("memory", pp.memory()); print("public_params %.2f s" % (time.time() - t0), pp.sizes(0), pp.sizes(1), "early rows", pp.early_rows())
initial = State.from_ints(FIELD_FQ, o.rand_fe(1, 0, o.Q), 0, 0)
t0 = time.time(); z0, circuits = InverseMinRootCircuit.eval_and_make_circuits(PallasVDF.new_with_mode(EvalMode.LTRAddChainSequential), t, n, initial)
print("forward evaluation of %d x 2^%d rounds: %.2f s (host, sequential)" % (n, lg, time.time() - t0))
circuits.upload(ctx)
ctx.set_async(True)
proof = None
for k in range(n):
    t0 = time.perf_counter(); proof = NovaVDFProof.prove_step(pp, proof, circuits, k, z0); dt = time.perf_counter() - t0
    print("step %d: %.2f ms " % (k, dt * 1e3), {a: round(b, 3) for a, b in proof.last_step_ms().items()})
ctx.sync()
proof.free()
# back to back (no print between steps: the device's hidden queues get no pause to catch up); the base case and the first
# fold (cold lookahead, workspaces growing) are a warm-up, as in bench.py
proof = NovaVDFProof.prove_step(pp, None, circuits, 0, z0)
first = 2 if n > 3 else 1
for k in range(1, first): proof = NovaVDFProof.prove_step(pp, proof, circuits, k, z0)
ctx.sync()
t0 = time.perf_counter()
for k in range(first, n): proof = NovaVDFProof.prove_step(pp, proof, circuits, k, z0)
ctx.sync(); dt = (time.perf_counter() - t0) / (n - first)
print("steady state: %.3f ms/step = %.1f prove_step/s" % (dt * 1e3, 1 / dt))
t0 = time.time(); ok = proof.verify(pp, n, z0, [initial.x, initial.y, initial.i]); print("verify", ok, "%.1f ms" % ((time.time() - t0) * 1e3))
