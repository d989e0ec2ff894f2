"""GPU: the step-circuit seam (src/nova/proof.rs:79-153, `impl StepCircuit for InverseMinRootCircuit`: arity / synthesize /
output).  A circuit the library has never seen -- z -> z^3 + z + 5, arity 1, written here in Python over the C ABI's
vdf_step_circuit / vdf_cs_* -- is set up, proven for three steps, verified, compressed and verified again, every
quantity against the oracle's run of the same circuit (oracle/nova.py CubicCircuit through the same seam)."""
import numpy as np
import pytest

from oracle import nova as nv, pasta as o, wire as w
from util import limbs, unmont
from test_gpu_nova import aff_ints, check_instance
from vdf_amd.nova import (NovaVDFProof, StepCircuit, public_params_custom, INST_RUNNING_PRIMARY, INST_RUNNING_SECONDARY,
                          INST_FRESH_SECONDARY)

pytestmark = pytest.mark.gpu
Q = o.Q


def fe(v):
    return limbs([o.to_mont(v % Q, Q)]).tobytes()


class Cubic(StepCircuit):
    arity = 1

    def synthesize(self, cs, z):
        x = z[0]
        x2 = cs.mul(x, x)
        x3 = cs.mul(x2, x)
        rhs = cs.add(cs.add(x3, x), cs.const(fe(5)))
        y = cs.alloc(cs.value(rhs) if cs.is_witness else None)
        cs.enforce(rhs, cs.const(fe(1)), y)
        return [y]


def test_a_host_written_step_circuit_through_the_seam(ctx):
    n, x0 = 3, 0x1234567
    circuit = Cubic()
    pp = public_params_custom(ctx, circuit)
    opp = nv.public_params(0, nv.CCommit(), nv.GENS_SEED, nv.FAMILY_TRY_AND_INCREMENT, primary=nv.CubicCircuit())
    assert pp.digest() == opp.params
    assert pp.segment() == (0, 0)                 # nothing for the device to fill: every variable comes from the host
    z0 = [fe(x0)]
    proof, want, x = None, None, x0
    for k in range(n):
        proof = NovaVDFProof.prove_step_custom(pp, proof, circuit, z0)
        want = nv.prove_step(opp, want, nv.CubicCircuit(), [x0])
        x = (x ** 3 + x + 5) % Q
        ls, tr = proof.last_step(), want.trace[-1]
        assert aff_ints(ls["comm_W1"], 0) == tuple(tr["l1"].comm_W) and unmont(ls["X1"], Q) == tr["l1"].X
        check_instance(proof, INST_RUNNING_PRIMARY, 0, want.r[0])
        check_instance(proof, INST_RUNNING_SECONDARY, 1, want.r[1])
        check_instance(proof, INST_FRESH_SECONDARY, 1, want.l2)
        zp, zs = proof.zi()
        assert unmont(zp, Q) == [x] == want.zi[0] and unmont(zs, o.P) == [0]
    assert proof.verify(pp, n, z0, [fe(x)]) is True
    assert proof.verify(pp, n, z0, [fe(x + 1)]) is False
    assert proof.verify(pp, n - 1, z0, [fe(x)]) is False
    snark = proof.compress(pp)
    assert snark.serialize() == w.encode_compressed_proof(0, opp.params, nv.compress(opp, want))
    assert snark.verify(pp, n, z0, [fe(x)]) is True
    assert snark.verify(pp, n, z0, [fe(x + 1)]) is False
    # checkpoint and resume with another arity than MinRoot's
    blob = proof.serialize()
    assert blob == w.encode_running_proof(0, opp.params, want, [x0])
    again = NovaVDFProof.deserialize(pp, blob)
    again.pp = pp
    again = NovaVDFProof.prove_step_custom(pp, again, circuit, z0)
    assert again.verify(pp, n + 1, z0, [fe((x ** 3 + x + 5) % Q)])


def test_a_failing_circuit_is_an_error_not_a_crash(ctx):
    class Broken(StepCircuit):
        arity = 1
        calls = 0

        def synthesize(self, cs, z):
            Broken.calls += 1
            if cs.is_witness:
                raise RuntimeError("no witness today")
            return [cs.mul(z[0], z[0])]
    c = Broken()
    pp = public_params_custom(ctx, c)
    with pytest.raises(RuntimeError):
        NovaVDFProof.prove_step_custom(pp, None, c, [fe(3)])
