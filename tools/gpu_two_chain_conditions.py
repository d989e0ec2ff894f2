"""Two concurrent chains under controlled conditions, one process per condition (what moves the two-chain aggregate?).
usage: gpu_two_chain_conditions.py <extra idle caller contexts> <1 = a third parameter set with a compression done first> [steps]"""
import gc, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vdf_amd
from vdf_amd.minroot import PallasVDF, State, FIELD_FQ, EvalMode
from vdf_amd.nova import InverseMinRootCircuit, NovaVDFProof, public_params, INST_FRESH_SECONDARY

extra, with_c = int(sys.argv[1]), int(sys.argv[2])
n = (int(sys.argv[3]) if len(sys.argv) > 3 else 24) + 2
t = 1 << 16


def chain(seed, steps):
    initial = State.from_ints(FIELD_FQ, 0x1234567890ABCDEF + seed, 0, 0)
    z0, circuits = InverseMinRootCircuit.eval_and_make_circuits(PallasVDF.new_with_mode(EvalMode.LTRAddChainSequential), t, steps, initial)
    return z0, circuits


ctxs = [vdf_amd.Context(0), vdf_amd.Context(0)]
idle = [vdf_amd.Context(0) for _ in range(extra)]
work = []
for k, c in enumerate(ctxs):
    z0, circuits = chain(k + 1, n)
    circuits.upload(c)
    work.append((c, public_params(c, t), circuits, z0))
if with_c:
    cC = vdf_amd.Context(0)
    ppC = public_params(cC, t)
    z0c, circ_c = chain(3, 3)
    pc = NovaVDFProof.prove_recursively(ppC, circ_c, t, z0c)
    pc.compress(ppC).free()


def prove(i, spans, delay):
    c, pp, circuits, z0 = work[i]
    c.set_async(True)
    p = NovaVDFProof.prove_step(pp, None, circuits, 0, z0)
    p = NovaVDFProof.prove_step(pp, p, circuits, 1, z0)
    c.sync()
    if delay:
        time.sleep(delay)
    a = time.perf_counter()
    for k in range(2, n):
        p = NovaVDFProof.prove_step(pp, p, circuits, k, z0)
    p.instance(INST_FRESH_SECONDARY)
    c.sync()
    spans[i] = (a, time.perf_counter())
    p.free()


gc.disable()
singles = []
for rep in range(4):
    sp = {}
    prove(0, sp, 0)
    singles.append((n - 2) / (sp[0][1] - sp[0][0]))
single = sorted(singles[1:])[1]
aggs = []
for rep in range(6):
    sp = {}
    ths = [threading.Thread(target=prove, args=(i, sp, i * 0.5 / single)) for i in range(2)]
    for th in ths: th.start()
    for th in ths: th.join()
    a_, b_ = max(s[0] for s in sp.values()), min(s[1] for s in sp.values())
    aggs.append(sum((n - 2) * (b_ - a_) / (s[1] - s[0]) for s in sp.values()) / (b_ - a_))
print("extra idle contexts %d, third set + compression %d, GPU_MAX_HW_QUEUES=%s: single %.0f/s (%s); two chains median %.0f/s (%s); streams %d" %
      (extra, with_c, os.environ.get("GPU_MAX_HW_QUEUES"), single, " ".join("%.0f" % x for x in singles), sorted(aggs[1:])[2],
       " ".join("%.0f" % x for x in aggs), ctxs[0].queue_info()["device_streams"]), flush=True)
