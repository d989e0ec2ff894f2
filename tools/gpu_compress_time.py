"""compress / verify_compressed timing at t = 2^k (default 16)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vdf_amd
from vdf_amd.minroot import PallasVDF, State, FIELD_FQ, EvalMode
from vdf_amd.nova import InverseMinRootCircuit, NovaVDFProof, public_params
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 16
t, n = 1 << lg, 2
ctx = vdf_amd.Context(0)
pp = public_params(ctx, t)
initial = State.from_ints(FIELD_FQ, 4242, 0, 0)
z0, circuits = InverseMinRootCircuit.eval_and_make_circuits(PallasVDF.new_with_mode(EvalMode.LTRAddChainSequential), t, n, initial)
proof = NovaVDFProof.prove_recursively(pp, circuits, t, z0)
zi = [initial.x, initial.y, initial.i]
for rep in range(3):
    a = time.perf_counter(); s = proof.compress(pp); b = time.perf_counter(); ok = s.verify(pp, n, z0, zi); c = time.perf_counter()
    print(f"compress {1e3*(b-a):.1f} ms, verify {1e3*(c-b):.1f} ms, ok {ok}, {len(s.to_bytes())} bytes", flush=True)
    s.free()
