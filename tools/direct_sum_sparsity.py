"""How much of a direct sum's work is spent on null digits, and how much of THAT a static schedule could remove (VERDICT r4
item 3).  CPU only: one augmented-circuit witness per side from the oracle (oracle/nova.py, three steps at t = 4: the wrapper's
~10^4 variables do not depend on t), the digit table's offset digits (msm_direct.hip: e_j = window_j(k + H) - 2^(c-1), c = 10,
26 windows), and the kernel's dealing of entries to wavefronts (entry e = j * n + s; a wavefront's 64 lanes hold 64 consecutive
entries; the addition is skipped by the whole wavefront when all 64 digits are null).
usage: python3 tools/direct_sum_sparsity.py > profiles/r05_direct_sum_sparsity.txt"""
import collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import nova as nv, pasta as o

t, n = 4, 3
pp = nv.public_params(t, nv.CCommit(threads=4), nv.GENS_SEED, nv.FAMILY_TRY_AND_INCREMENT, bound=False)
states = [o.State(0x5678, 0, 1)]
for _ in range(n):
    states.append(o.minroot_eval(states[-1], t, o.FIELD_FQ))
z0 = [states[n].x, states[n].y, states[n].i]
s = None
for k in range(n):
    s = nv.prove_step(pp, s, nv.InverseMinRootCircuit(t, states[n - k], states[n - k - 1], False), z0)
tr = s.trace[-1]
c, Wn = 10, 26
H = sum(1 << (c * j + c - 1) for j in range(Wn))
for name in ("l1", "l2"):
    W = tr[name].W
    nvar = len(W)
    hist = collections.Counter("0/1" if v <= 1 else "<= 2^64" if v < (1 << 64) else "<= 2^128" if v < (1 << 128) else "<= 2^250" if v < (1 << 250) else "full" for v in W)
    runs, cur = [], 0
    for v in W:
        if v <= 1:
            cur += 1
        else:
            if cur:
                runs.append(cur)
            cur = 0
    if cur:
        runs.append(cur)
    flat = [((k + H) >> (c * j)) & ((1 << c) - 1) != (1 << (c - 1)) for j in range(Wn) for k in W]
    E, useful = len(flat), sum(flat)
    blocks = (E + 63) // 64
    active = sum(1 for b in range(blocks) if any(flat[64 * b:64 * b + 64]))
    ideal = (useful + 63) // 64
    print("%s side (%s): %d variables %s" % ("primary" if name == "l1" else "secondary", name, nvar, dict(hist)))
    print("  0/1 variables in runs of >= 64 consecutive: %d of %d (runs: %s)" % (sum(r for r in runs if r >= 64), hist["0/1"], sorted(r for r in runs if r >= 64)))
    print("  entries (variable, window) %d, non-null %d (%.1f %%)" % (E, useful, 100.0 * useful / E))
    print("  64-entry blocks %d; executed today (any lane non-null) %d (%.1f %%); with a perfect static schedule %d (%.1f %%): "
          "%.1f %% of the executed blocks saved" % (blocks, active, 100.0 * active / blocks, ideal, 100.0 * ideal / blocks, 100.0 * (1 - ideal / active)))
print("a step's direct sums also commit ~10^4 rows of T per side (full-size scalars: nothing to skip), so the saving is about half of "
      "the above per kernel: ~5 % of k_direct_sum's 53 M wave-instructions per step = ~0.7 % of a step's ~370 M")
