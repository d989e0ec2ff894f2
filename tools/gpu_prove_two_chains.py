"""Aggregate prove_step throughput of TWO independent VDF chains proven concurrently on one GPU (two host
threads, two contexts): one chain's bucket reduction and host transcript run under the other's accumulation."""
import sys, os, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vdf_amd
from vdf_amd.minroot import PallasVDF, State, FIELD_FQ, EvalMode
from vdf_amd.nova import InverseMinRootCircuit, NovaVDFProof, public_params

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 16
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
chains = int(sys.argv[3]) if len(sys.argv) > 3 else 2
t = 1 << lg
work = []
for c in range(chains):
    ctx = vdf_amd.Context(0)
    pp = public_params(ctx, t)
    initial = State.from_ints(FIELD_FQ, 1000 + c, 0, 0)
    z0, circuits = InverseMinRootCircuit.eval_and_make_circuits(PallasVDF.new_with_mode(EvalMode.LTRAddChainSequential), t, n, initial)
    circuits.upload(ctx)
    ctx.set_async(True)
    proof = NovaVDFProof.prove_step(pp, None, circuits, 0, z0)
    proof = NovaVDFProof.prove_step(pp, proof, circuits, 1, z0)       # warm-up fold
    ctx.sync()
    work.append([ctx, pp, circuits, z0, proof, initial])
print("setup done", flush=True)

def run(w):
    ctx, pp, circuits, z0, proof, _ = w
    for k in range(2, n):
        proof = NovaVDFProof.prove_step(pp, proof, circuits, k, z0)
    ctx.sync()
    w[4] = proof

for label, group in (("one chain", work[:1]), (f"{chains} chains", work)):
    if label == "one chain":
        import copy
        # time chain 0 alone on a copy of its state is not possible (proof advances); use the last chain for the solo run
        group = work[-1:]
    ths = [threading.Thread(target=run, args=(w,)) for w in group]
    t0 = time.perf_counter()
    for th in ths: th.start()
    for th in ths: th.join()
    dt = time.perf_counter() - t0
    steps = len(group) * (n - 2)
    print(f"{label}: {steps} folds in {dt*1e3:.1f} ms = {steps/dt:.1f} prove_step/s aggregate", flush=True)
    if label == "one chain":
        # re-arm the solo chain so it can take part in the concurrent run: rebuild its proof
        ctx, pp, circuits, z0, proof, initial = work[-1]
        proof.free()
        p = NovaVDFProof.prove_step(pp, None, circuits, 0, z0); p = NovaVDFProof.prove_step(pp, p, circuits, 1, z0); ctx.sync()
        work[-1][4] = p
for w in work:
    ctx, pp, circuits, z0, proof, initial = w
    ctx.set_async(False)
    print("verify", proof.verify(pp, n, z0, [initial.x, initial.y, initial.i]))
