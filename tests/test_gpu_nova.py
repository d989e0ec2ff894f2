"""GPU tests of the Nova proof layer (vdf_amd/nova.py over libvdf_nova.so + libvdf_hip.so):
the reference's test_nova_proof (src/nova/proof.rs:403-451) at its own size, BASELINE config 1
(t = 1024, 3 steps), and a full replay of the folds by the oracle (witness, error vector,
instance, commitments through the discrete-log identity, SHAKE256 transcript)."""
import numpy as np
import pytest

from oracle import pasta as o
from oracle import nifs
from util import ints, unmont
from vdf_amd.minroot import PallasVDF, State, FIELD_FQ
from vdf_amd.nova import InverseMinRootCircuit, NovaVDFProof, public_params

pytestmark = pytest.mark.gpu
# the oracle side of the proof layer lives in oracle/nifs.py; these names are what the other GPU tests import
GENS_SEED, GENS_FAMILY = nifs.GENS_SEED, nifs.GENS_FAMILY
gens, commit, le32, shape_digest, challenge = nifs.gens, nifs.commit, nifs.le32, nifs.shape_digest, nifs.challenge


def aff_ints(arr):
    x, y = unmont(np.asarray(arr).reshape(2, 4), o.P)
    return (x, y)


def make(ctx, t, n, seed=42, i0=1):
    x = o.rand_fe(seed, 0, o.Q)
    initial = State.from_ints(FIELD_FQ, x, 0, i0)             # y = 0, i = 1: src/nova/proof.rs:417-421
    pp = public_params(ctx, t)
    z0, circuits = InverseMinRootCircuit.eval_and_make_circuits(PallasVDF.new(), t, n, initial)
    return pp, z0, circuits, initial, (x, 0, i0)


def test_nova_proof(ctx):
    """test_nova_proof_aux(5, 3), src/nova/proof.rs:403-451, including its compress leg (:446-450)."""
    t, n = 5, 3
    pp, z0, circuits, initial, init_ints = make(ctx, t, n)
    zi = [initial.x, initial.y, initial.i]
    proof = NovaVDFProof.prove_recursively(pp, circuits, t, z0)
    assert proof.verify(pp, n, z0, zi) is True
    # Ok(false) legs: wrong expected z_i, wrong step count
    wrong = [initial.y, initial.x, initial.i]
    assert proof.verify(pp, n, z0, wrong) is False
    assert proof.verify(pp, n + 1, z0, zi) is False
    assert proof.verify(pp, n, zi, zi) is False
    compressed = proof.compress(pp)                              # :446-448
    assert compressed.verify(pp, n, z0, zi) is True              # :449-450
    assert compressed.verify(pp, n, z0, wrong) is False


def test_eval_and_make_circuits_order(ctx):
    """Circuits come back reversed and z0 is the FINAL state (src/nova/proof.rs:278-294)."""
    t, n = 4, 3
    x = o.rand_fe(9, 0, o.Q)
    initial = State.from_ints(FIELD_FQ, x, 0, 0)
    z0, circuits = InverseMinRootCircuit.eval_and_make_circuits(PallasVDF.new(), t, n, initial)
    states = [o.State(x, 0, 0)]
    for _ in range(n):
        states.append(o.minroot_eval(states[-1], t, o.FIELD_FQ))
    assert State(*z0).to_ints(FIELD_FQ) == (states[-1].x, states[-1].y, states[-1].i)
    assert len(circuits) == n
    for k in range(n):
        res, inp = circuits.states(k)
        assert res.to_ints(FIELD_FQ) == (states[n - k].x, states[n - k].y, states[n - k].i)
        assert inp.to_ints(FIELD_FQ) == (states[n - k - 1].x, states[n - k - 1].y, states[n - k - 1].i)
    with pytest.raises(AssertionError):
        InverseMinRootCircuit.eval_and_make_circuits(PallasVDF.new(), t, 0, initial)


@pytest.mark.parametrize("t,n", [(5, 3), (24, 3)])
def test_prove_steps_replayed_by_the_oracle(ctx, t, n):
    """Every quantity of every fold, bit-exact against the Python oracle."""
    m = o.Q
    pp, z0, circuits, initial, init_ints = make(ctx, t, n, seed=77)
    sh = o.step_circuit_shape(t, o.FIELD_FQ)
    sizes = pp.sizes()
    assert (sizes["num_cons"], sizes["num_vars"], sizes["num_io"]) == (sh.num_cons, sh.num_vars, 6)
    assert sizes["nnz"] == len(sh.A) + len(sh.B) + len(sh.C)
    digest = shape_digest(sh, t)
    # the whole chain by the oracle, then the product step by step against its prefixes
    states = nifs.forward_states(o.State(*init_ints), t, n)
    proof = None
    for k in range(n):
        proof = NovaVDFProof.prove_step(pp, proof, circuits, k, z0)
        want, sh2, dg = nifs.prove_chain(states[n - k - 1], t, k + 1)      # the first k + 1 steps end at states[n-k-1]
        assert dg == digest
        for j in range(k + 1):
            rec, ws = proof.step_record(j), want.steps[j]
            assert aff_ints(rec["comm_w"]) == ws.comm_w and aff_ints(rec["comm_T"]) == ws.comm_T
            assert unmont(rec["r"].reshape(1, 4), m) == [ws.r] and unmont(rec["X"], m) == ws.X
        inst = proof.instance()
        gW, gE = proof.witness()
        assert unmont(gW, m) == want.W
        assert unmont(gE, m) == want.E
        assert unmont(inst["u"].reshape(1, 4), m) == [want.u]
        assert unmont(inst["X"], m) == want.X
        assert aff_ints(inst["comm_W"]) == want.comm_W and aff_ints(inst["comm_E"]) == want.comm_E
        assert o.is_sat_relaxed(sh, want.W, want.E, want.u, want.X, m)
    zi = [State.from_ints(FIELD_FQ, *init_ints).x, State.from_ints(FIELD_FQ, *init_ints).y, State.from_ints(FIELD_FQ, *init_ints).i]
    assert proof.verify(pp, n, z0, zi)


def test_tampered_witness_is_rejected(ctx):
    """Corrupting one word of the running witness on the device must fail verification."""
    import ctypes as C
    from vdf_amd._lib import lib
    from vdf_amd.nova import nova_lib
    t, n = 8, 2
    pp, z0, circuits, initial, _ = make(ctx, t, n, seed=5)
    zi = [initial.x, initial.y, initial.i]
    proof = NovaVDFProof.prove_recursively(pp, circuits, t, z0)
    assert proof.verify(pp, n, z0, zi)
    dW, dE = C.c_void_p(), C.c_void_p()
    nova_lib.vdf_nova_proof_witness_ptrs(proof.handle, C.byref(dW), C.byref(dE))
    word = np.zeros(4, dtype="<u8")
    ctx._check(lib.vdf_dev_memcpy(ctx.handle, word.ctypes.data, dW.value + 7 * 32, 32))
    bad = word.copy(); bad[0] ^= 1
    ctx._check(lib.vdf_dev_memcpy(ctx.handle, dW.value + 7 * 32, bad.ctypes.data, 32))
    assert proof.verify(pp, n, z0, zi) is False
    ctx._check(lib.vdf_dev_memcpy(ctx.handle, dW.value + 7 * 32, word.ctypes.data, 32))
    assert proof.verify(pp, n, z0, zi) is True


def _same_proof(a, b, n):
    ia, ib = a.instance(), b.instance()
    assert all(np.array_equal(ia[k], ib[k]) for k in ia)
    for k in range(n):
        ra, rb = a.step_record(k), b.step_record(k)
        assert all(np.array_equal(ra[f], rb[f]) for f in ra)
    (wa, ea), (wb, eb) = a.witness(), b.witness()
    assert np.array_equal(wa, wb) and np.array_equal(ea, eb)


def test_lookahead_is_invisible(ctx):
    """prove_step enqueues the next step's fresh witness and commitment ahead of time (include/vdf_nova.h).  Whatever
    the caller does next -- the expected step, the same step from another circuits object, a refused call in between,
    a second proof interleaved -- the proofs are the ones a step-at-a-time prover makes."""
    t, n = 32, 5
    pp, z0, circuits, initial, init_ints = make(ctx, t, n, seed=77)
    zi = [initial.x, initial.y, initial.i]
    ref = NovaVDFProof.prove_recursively(pp, circuits, t, z0)
    assert ref.verify(pp, n, z0, zi)
    # (a) the same states from a second circuits object: every lookahead is for the wrong object and is dropped
    _, twin = InverseMinRootCircuit.eval_and_make_circuits(PallasVDF.new(), t, n, initial)
    p = None
    for k in range(n):
        p = NovaVDFProof.prove_step(pp, p, circuits if k % 2 == 0 else twin, k, z0)
    _same_proof(ref, p, n)
    assert p.verify(pp, n, z0, zi)
    # (b) a refused call (wrong step) between two good ones leaves the proof and its lookahead intact
    q = None
    for k in range(n):
        q = NovaVDFProof.prove_step(pp, q, circuits, k, z0)
        if k == 1:
            with pytest.raises(Exception):
                NovaVDFProof.prove_step(pp, q, circuits, 4, z0)
            with pytest.raises(Exception):
                NovaVDFProof.prove_step(pp, q, circuits, n, z0)
    _same_proof(ref, q, n)
    # (c) two proofs advanced in turns over the same circuits
    a = b = None
    for k in range(n):
        a = NovaVDFProof.prove_step(pp, a, circuits, k, z0)
        b = NovaVDFProof.prove_step(pp, b, circuits, k, z0)
    _same_proof(ref, a, n)
    _same_proof(ref, b, n)
    # (d) circuits whose traces were never uploaded (staged per step) and uploaded ones give the same proof
    twin.upload(ctx)
    u = NovaVDFProof.prove_recursively(pp, twin, t, z0)
    _same_proof(ref, u, n)
    # (e) freeing the circuits right after the last call is safe: nothing in flight reads them
    _, gone = InverseMinRootCircuit.eval_and_make_circuits(PallasVDF.new(), t, n, initial)
    gone.upload(ctx)
    g = None
    for k in range(3):
        g = NovaVDFProof.prove_step(pp, g, gone, k, z0)
    gone.free()
    g.free()
    assert ref.compress(pp).verify(pp, n, z0, zi)


def test_mismatched_z0_is_an_error(ctx):
    import vdf_amd
    t, n = 4, 2
    pp, z0, circuits, initial, _ = make(ctx, t, n, seed=6)
    with pytest.raises(vdf_amd.VdfError):
        NovaVDFProof.prove_recursively(pp, circuits, t, [initial.x, initial.y, initial.i])   # not the final state
    with pytest.raises(vdf_amd.VdfError):
        NovaVDFProof.prove_recursively(pp, circuits, t + 1, z0)


def test_config1_t1024_three_steps(ctx):
    """BASELINE config 1: 1024 iterations per step, 3 recursive steps; both i = 0 (benches/nova.rs:24-26)
    and i = 1 (src/nova/proof.rs:419)."""
    for i0 in (0, 1):
        t, n = 1024, 3
        pp, z0, circuits, initial, _ = make(ctx, t, n, seed=11, i0=i0)
        proof = NovaVDFProof.prove_recursively(pp, circuits, t, z0)
        assert proof.num_steps() == n
        assert proof.verify(pp, n, z0, [initial.x, initial.y, initial.i])
        ms = proof.last_step_ms()
        assert ms["total"] > 0


def test_plain_c_client_of_both_abis():
    """examples/prove_chain.c: eval -> prove -> verify -> compress -> verify, from C, in a process of its own."""
    import os, subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "prove_chain")
    assert os.path.exists(exe), "build it with `make -C vdf_amd/csrc` (part of __graft_entry__.build())"
    r = subprocess.run([exe, "8", "3"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "verify: true" in r.stdout and "verify (compressed): true" in r.stdout
