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
# where the time goes: HIP events around every launch of one more compression (a pass of its own)
ctx.sync(); ctx.set_kernel_timing(True); ctx.kernel_events()
a = time.perf_counter(); s = proof.compress(pp); ctx.sync(); wall = (time.perf_counter() - a) * 1e3
ev = ctx.kernel_events(); ctx.set_kernel_timing(False); s.free()
agg = {}
for name, nbytes, s0, s1 in ev:
    e = agg.setdefault(name, [0, 0.0, 0.0]); e[0] += 1; e[1] += s1 - s0; e[2] += nbytes
dev = sum(e[1] for e in agg.values())
print(f"timed pass {wall:.1f} ms, device (sum of launches) {dev:.1f} ms, host + idle {wall - dev:.1f} ms")
print("%-26s %6s %9s %9s %9s %8s" % ("kernel", "calls", "ms", "avg us", "MB", "GB/s"))
for name, (calls, ms, nb) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-26s %6d %9.3f %9.1f %9.1f %8s" % (name, calls, ms, ms / calls * 1e3, nb / 1e6, ("%.0f" % (nb / ms / 1e6)) if nb else "-"))
t0 = ev[0][2]
print("first 60 launches (start ms, dur us):")
for name, nbytes, s0, s1 in ev[:60]:
    print("  %8.3f %8.1f  %s" % (s0 - t0, (s1 - s0) * 1e3, name))
# three rounds of the inner-product arguments from the middle of the pass: the gaps are the host between the launches
mid = [i for i, e in enumerate(ev) if e[0] == "k_accumulate"]
if len(mid) >= 8:
    lo, hi = mid[5], mid[8]
    print("rounds 6-8 of the primary side's arguments (start ms, dur us, gap to the previous launch's end us):")
    prev = ev[lo - 1][3] if lo else ev[0][2]
    for name, nbytes, s0, s1 in ev[lo - 4:hi + 1]:
        print("  %8.3f %8.1f %8.1f  %s" % (s0 - t0, (s1 - s0) * 1e3, (s0 - prev) * 1e3, name))
        prev = s1
# every gap of more than 40 us on the caller's queue (host turns, synchronisations, the other queues' work it waits for)
print("gaps over 40 us on the caller's queue (at ms, gap us, after -> before):")
tot = 0.0
for (n0, _, _, e0), (n1, _, s1, _) in zip(ev, ev[1:]):
    g = (s1 - e0) * 1e3
    if g > 40:
        tot += g
        print("  %8.3f %8.1f  %s -> %s" % (e0 - t0, g, n0, n1))
print("  total %.2f ms" % (tot / 1e3))
