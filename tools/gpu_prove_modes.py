"""Is the run-to-run spread of a single chain's rate (1,125 ... 1,190 prove_step/s between processes on one box) a property of the PROCESS
or of the prover object?  One process, one parameter set: `reps` chains of `n` steps one after the other, each a new proof (new side
contexts and streams, new workspaces), with the box's clock probe between them.
   python3 tools/gpu_prove_modes.py [reps] [n]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pasta as o
import vdf_amd
if os.environ.get('VDF_TEST_TORCH'):                      # what bench.py has open before its prove_step leg: torch's streams
    import torch
    _streams = [torch.cuda.Stream() for _ in range(int(os.environ['VDF_TEST_TORCH']))]
    for _s in _streams:
        with torch.cuda.stream(_s): torch.zeros(16, device='cuda').add_(1)
    torch.cuda.synchronize()
from vdf_amd.minroot import PallasVDF, State, FIELD_FQ, EvalMode
from vdf_amd.nova import InverseMinRootCircuit, NovaVDFProof, public_params, CIRCUIT_MINROOT_REFERENCE
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100
t = 1 << 16
ctx = vdf_amd.Context(0)
pp = public_params(ctx, t, CIRCUIT_MINROOT_REFERENCE)
initial = State.from_ints(FIELD_FQ, o.rand_fe(1, 0, o.Q), 0, 0)
z0, circuits = InverseMinRootCircuit.eval_and_make_circuits(PallasVDF.new_with_mode(EvalMode.LTRAddChainSequential), t, n, initial)
circuits.upload(ctx)
ctx.set_async(True)
for r in range(reps):
    proof = NovaVDFProof.prove_step(pp, None, circuits, 0, z0)
    proof = NovaVDFProof.prove_step(pp, proof, circuits, 1, z0)
    ctx.sync()
    rates = []
    for seg in range(4):                                   # four quarters of the chain: does the rate move INSIDE a chain?
        lo, hi = 2 + seg * (n - 2) // 4, 2 + (seg + 1) * (n - 2) // 4
        t0 = time.perf_counter()
        for k in range(lo, hi): proof = NovaVDFProof.prove_step(pp, proof, circuits, k, z0)
        ctx.sync(); rates.append((hi - lo) / (time.perf_counter() - t0))
    mhz, kms = ctx.clock_probe(6000)
    print("chain %d: quarters %s  = %.1f prove_step/s | clock probe %.0f MHz %.3f ms | queues %s" % (
        r, " ".join("%.0f" % x for x in rates), len(rates) / sum(1 / x for x in rates), mhz, kms, ctx.queue_info()), flush=True)
    proof.free()
