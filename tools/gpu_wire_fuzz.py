"""Mutated wire bytes: every decode must end in an error or in a proof that does not verify -- never in a crash, a hang
or an accepted forgery.  Both formats ("VDFSNK02" compressed proof, "VDFRSK01" running proof)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import pasta as o
import vdf_amd
from vdf_amd.minroot import PallasVDF, State, FIELD_FQ
from vdf_amd.nova import InverseMinRootCircuit, NovaVDFProof, CompressedNovaVDFProof, public_params

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = vdf_amd.Context(0)
t, n = 32, 3
pp = public_params(ctx, t)
initial = State.from_ints(FIELD_FQ, o.rand_fe(9, 0, o.Q), 0, 1)
z0, circuits = InverseMinRootCircuit.eval_and_make_circuits(PallasVDF.new(), t, n, initial)
proof = NovaVDFProof.prove_recursively(pp, circuits, t, z0)
zi = [initial.x, initial.y, initial.i]
good_c = proof.compress(pp).serialize()
good_r = proof.serialize()
assert CompressedNovaVDFProof.deserialize(pp, good_c).verify(pp, n, z0, zi)
assert NovaVDFProof.deserialize(pp, good_r).verify(pp, n, z0, zi)


def mutate(b: bytes) -> bytes:
    a = bytearray(b)
    kind = int(rng.integers(0, 6))
    if kind == 0:                                   # a few random bytes
        for _ in range(int(rng.integers(1, 6))):
            a[int(rng.integers(0, len(a)))] = int(rng.integers(0, 256))
    elif kind == 1:                                 # one flipped bit
        i = int(rng.integers(0, len(a))); a[i] ^= 1 << int(rng.integers(0, 8))
    elif kind == 2:                                 # truncation / extension
        cut = int(rng.integers(0, len(a) + 40))
        a = a[:cut] if cut <= len(a) else a + bytes(rng.integers(0, 256, size=cut - len(a), dtype=np.uint8))
    elif kind == 3:                                 # header fields: t, step count
        off = int(rng.choice([8, 16]))
        vals = [0, 1, 2, 4, 1 << 20, (1 << 63) - 1, (1 << 64) - 1]
        a[off:off + 8] = vals[int(rng.integers(0, len(vals)))].to_bytes(8, "little")
    elif kind == 4:                                 # a 32-byte element replaced by a non-canonical or random one
        off = 56 + 32 * int(rng.integers(0, (len(a) - 56) // 32))
        a[off:off + 32] = bytes([0xFF] * 32) if rng.random() < 0.5 else bytes(rng.integers(0, 256, size=32, dtype=np.uint8))
    else:                                           # two elements swapped
        k = (len(a) - 56) // 32
        i, j = (56 + 32 * int(x) for x in rng.integers(0, k, size=2))
        a[i:i + 32], a[j:j + 32] = a[j:j + 32], a[i:i + 32]
    return bytes(a)


accepted = errors = rejected = 0
t0 = time.time()
for it in range(iters):
    comp = it % 4 != 3                              # the running proof is 100x larger: every fourth
    good = good_c if comp else good_r
    bad = mutate(good)
    if bad == good:
        continue
    try:
        p = (CompressedNovaVDFProof if comp else NovaVDFProof).deserialize(pp, bad)
    except vdf_amd.VdfError:
        errors += 1
        continue
    ok = p.verify(pp, n, z0, zi)
    p.free()
    if ok:
        accepted += 1
        print("ACCEPTED a mutated", "compressed" if comp else "running", "proof at iteration", it, flush=True)
    else:
        rejected += 1
print(f"wire fuzz done: {iters} mutations, {errors} refused at decoding, {rejected} decoded and rejected, {accepted} ACCEPTED, "
      f"{time.time() - t0:.0f} s")
sys.exit(1 if accepted else 0)
