"""Stress for one observed flake (a parameter set made right after a proof had a digest no parameter choice explains): parameter
sets with flags, made between proofs, many times; every digest against the host-only vdf_nova_shape_digest."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vdf_amd
from vdf_amd.minroot import PallasVDF, State, FIELD_FQ
from vdf_amd.nova import (InverseMinRootCircuit, NovaVDFProof, public_params, shape_digest, PP_NO_DIGIT_TABLES, PP_NO_EARLY_ROWS,
                          GENS_TRY_AND_INCREMENT)
ctx = vdf_amd.Context(0)
t, n, bad = 96, 4, 0
for kind in (1, 0):
    want = shape_digest(t, kind)[0]
    initial = State.from_ints(FIELD_FQ, 0x1234 + kind, 0, 1)
    z0, circuits = InverseMinRootCircuit.eval_and_make_circuits(PallasVDF.new(), t, n, initial)
    pp = public_params(ctx, t, kind)
    assert pp.digest() == want
    for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 12):
        a = NovaVDFProof.prove_recursively(pp, circuits, t, z0)
        for flags in (PP_NO_DIGIT_TABLES, PP_NO_EARLY_ROWS, PP_NO_DIGIT_TABLES | PP_NO_EARLY_ROWS):
            pp1 = public_params(ctx, t, kind, GENS_TRY_AND_INCREMENT, flags)
            if pp1.digest() != want:
                bad += 1
                print("MISMATCH kind %d iteration %d flags %d: %x" % (kind, it, flags, pp1.digest()), flush=True)
            b = NovaVDFProof.prove_recursively(pp1, circuits, t, z0)
            assert b.verify(pp1, n, z0, [initial.x, initial.y, initial.i])
            b.free(); pp1.free()
        a.free()
    pp.free()
print("digest mismatches:", bad)
