"""Single-chain prove_step rate as a function of how many idle streams were opened BEFORE / BETWEEN the prover's queues (the
stream -> hardware queue mapping is by creation order).  usage: gpu_single_chain_vs_padding.py <idle before ctx> <idle after ctx> [steps]"""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vdf_amd
from vdf_amd.minroot import PallasVDF, State, FIELD_FQ, EvalMode
from vdf_amd.nova import InverseMinRootCircuit, NovaVDFProof, public_params, INST_FRESH_SECONDARY
before, after = int(sys.argv[1]), int(sys.argv[2])
n = (int(sys.argv[3]) if len(sys.argv) > 3 else 48) + 2
t = 1 << 16
pad0 = [vdf_amd.Context(0) for _ in range(before)]
ctx = vdf_amd.Context(0)
pad1 = [vdf_amd.Context(0) for _ in range(after)]
initial = State.from_ints(FIELD_FQ, 0x1234567890ABCDEF, 0, 0)
z0, circuits = InverseMinRootCircuit.eval_and_make_circuits(PallasVDF.new_with_mode(EvalMode.LTRAddChainSequential), t, n, initial)
circuits.upload(ctx)
pp = public_params(ctx, t)
ctx.set_async(True)
gc.disable()
rates = []
for rep in range(5):
    p = NovaVDFProof.prove_step(pp, None, circuits, 0, z0)
    p = NovaVDFProof.prove_step(pp, p, circuits, 1, z0)
    ctx.sync()
    a = time.perf_counter()
    for k in range(2, n):
        p = NovaVDFProof.prove_step(pp, p, circuits, k, z0)
    p.instance(INST_FRESH_SECONDARY)
    ctx.sync()
    rates.append((n - 2) / (time.perf_counter() - a))
    p.free()
print("idle streams before the context %d, between it and the prover's queues %d: %s  median %.0f prove_step/s" %
      (before, after, " ".join("%.0f" % r for r in rates), sorted(rates[1:])[2]), flush=True)
