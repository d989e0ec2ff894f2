"""Time line of every launch of two back-to-back steady-state prove_steps (the prover's three queues on the device's common
clock, vdf_nova_proof_kernel_events), for DESIGN.md 4.3.  usage: gpu_step_events.py [log2t] [ref|bound]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
from oracle import pasta as o
import vdf_amd
from vdf_amd.minroot import PallasVDF, State, FIELD_FQ, EvalMode
from vdf_amd.nova import InverseMinRootCircuit, NovaVDFProof, public_params, INST_FRESH_SECONDARY
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 16
kind = 0 if len(sys.argv) > 2 and sys.argv[2].startswith("b") else 1
t, n = 1 << lg, 14
ctx = vdf_amd.Context(0)
pp = public_params(ctx, t, kind)
initial = State.from_ints(FIELD_FQ, o.rand_fe(1, 0, o.Q), 0, 0)
z0, circuits = InverseMinRootCircuit.eval_and_make_circuits(PallasVDF.new_with_mode(EvalMode.LTRAddChainSequential), t, n, initial)
circuits.upload(ctx)
ctx.set_async(True)
proof = None
for k in range(4):
    proof = NovaVDFProof.prove_step(pp, proof, circuits, k, z0)
ctx.sync()
proof.set_kernel_timing(True)
proof.kernel_events()
marks = []
a = time.perf_counter()
for k in range(4, n):
    proof = NovaVDFProof.prove_step(pp, proof, circuits, k, z0)
    marks.append(proof.last_step_ms())
proof.instance(INST_FRESH_SECONDARY)
ctx.sync()
print("step circuit:", "reference" if kind else "bound", " %.3f ms per step with the events on" % ((time.perf_counter() - a) / (n - 4) * 1e3))
ev = sorted(proof.kernel_events(), key=lambda e: e[3])
t0 = ev[0][3]
span = (ev[-1][4] - t0) / (n - 4)
print("device span per step %.3f ms" % span)
# two steps from the middle
lo, hi = t0 + 4 * span, t0 + 6 * span
names = ("chain", "lookahead", "early_rows")
print("%-9s %-10s %-26s %9s %9s" % ("start us", "queue", "kernel", "dur us", "GB/s"))
for q, name, nbytes, s, e in ev:
    if lo <= s < hi:
        print("%9.1f %-10s %-26s %9.1f %9s" % ((s - lo) * 1e3, names[q], name, (e - s) * 1e3, ("%.0f" % (nbytes / (e - s) * 1e-6)) if nbytes else ""))
print("host stage_ms of the last step:", {k: round(v, 3) for k, v in marks[-1].items()})
