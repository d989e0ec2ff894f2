"""Interleaved A/B of prove_step configurations INSIDE ONE PROCESS (one box, one clock state, one set of circuits): every round
proves the same chain once per configuration, `steps` steady-state folds each; the per-configuration medians are compared.
A configuration is  name[:field=value,...]  where a field of vdf_hip_tuning is set process-wide before that configuration's
run and a field of vdf_nova_tuning goes into its own parameter set.
usage: gpu_prove_ab_inproc.py <rounds> <steps> base fused:fold_fused=1 serial:fixup_serial=1 ..."""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import vdf_amd
from vdf_amd import hip
from vdf_amd.minroot import PallasVDF, State, FIELD_FQ, EvalMode
from vdf_amd.nova import InverseMinRootCircuit, NovaVDFProof, NovaTuning, public_params, INST_FRESH_SECONDARY

rounds, steps = int(sys.argv[1]), int(sys.argv[2])
lg = int(os.environ.get("AB_LOG2T", "16"))
t, n = 1 << lg, steps + 2
hip_fields = dict(hip.HipTuning._fields_)
nova_fields = dict(NovaTuning._fields_)
configs = []
for spec in sys.argv[3:]:
    name, _, rest = spec.partition(":")
    h, v = {}, {}
    for kv in filter(None, rest.split(",")):
        k, val = kv.split("=")
        (h if k in hip_fields else v)[k] = int(val)
        assert k in hip_fields or k in nova_fields, k
    configs.append((name, h, v))
ctx = vdf_amd.Context(0)
base_hip = hip.tuning_get()
defaults = {k: getattr(base_hip, k) for k in hip_fields if k != "struct_size"}
initial = State.from_ints(FIELD_FQ, 0x1234567890ABCDEF1234567890ABCDEF, 0, 0)
z0, circuits = InverseMinRootCircuit.eval_and_make_circuits(PallasVDF.new_with_mode(EvalMode.LTRAddChainSequential), t, n, initial)
circuits.upload(ctx)
pps = {}
for name, h, v in configs:
    key = tuple(sorted(v.items()))
    if key not in pps:
        pps[key] = public_params(ctx, t, 1, **v) if v else public_params(ctx, t, 1)
ctx.set_async(True)
res = {name: [] for name, _, _ in configs}
digest = None
for rnd in range(rounds + 1):                                  # round 0: settle, not rated
    for name, h, v in (configs if rnd % 2 == 0 else configs[::-1]):
        hip.tuning_set(**{**defaults, **h})
        pp = pps[tuple(sorted(v.items()))]
        gc.collect(); gc.disable()
        proof = NovaVDFProof.prove_step(pp, None, circuits, 0, z0)
        proof = NovaVDFProof.prove_step(pp, proof, circuits, 1, z0)
        ctx.sync()
        a = time.perf_counter()
        for k in range(2, n):
            proof = NovaVDFProof.prove_step(pp, proof, circuits, k, z0)
        proof.instance(INST_FRESH_SECONDARY)
        ctx.sync()
        dt = (time.perf_counter() - a) / steps * 1e3
        gc.enable()
        if rnd == 0:
            assert proof.verify(pp, n, z0, [initial.x, initial.y, initial.i]), name
            data = proof.serialize()
            digest = digest or data
            assert data == digest, "configuration %s changes the proof" % name
        else:
            res[name].append(dt)
        proof.free()
hip.tuning_set(**defaults)
med = lambda xs: sorted(xs)[len(xs) // 2]
base = med(res[configs[0][0]])
for name, _, _ in configs:
    xs = res[name]
    print("%-28s median %.4f ms/step (%7.1f /s)  %+5.1f %% vs %s   runs: %s" %
          (name, med(xs), 1e3 / med(xs), (med(xs) / base - 1) * 100, configs[0][0], " ".join("%.3f" % x for x in xs)), flush=True)
