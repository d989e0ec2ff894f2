"""GPU tests of the Nova proof layer (vdf_amd/nova.py over libvdf_nova.so + libvdf_hip.so): the reference's
test_nova_proof (src/nova/proof.rs:403-451) at its own size, every quantity of every step against the oracle
(oracle/nova.py), BASELINE config 1 (t = 1024, 3 steps) and config 3 (t = 2^16) checked by the C restatement and the
discrete-log identity, and negative cases."""
import ctypes as C

import numpy as np
import pytest

from oracle import nova as nv, pasta as o
from util import ints, unmont, limbs
from vdf_amd.minroot import PallasVDF, State, FIELD_FQ, EvalMode
from vdf_amd.nova import (InverseMinRootCircuit, NovaVDFProof, public_params, CIRCUIT_MINROOT_BOUND, CIRCUIT_MINROOT_REFERENCE,
                          GENS_KNOWN_DLOG, GENS_TRY_AND_INCREMENT, INST_RUNNING_PRIMARY, INST_RUNNING_SECONDARY,
                          INST_FRESH_SECONDARY)

pytestmark = pytest.mark.gpu
BASE = (o.P, o.Q)                      # coordinate modulus of side 0 / 1 commitments
SCAL = (o.Q, o.P)                      # scalar modulus of side 0 / 1 instances


def aff_ints(arr, side):
    return tuple(unmont(np.asarray(arr).reshape(2, 4), BASE[side]))


def make(ctx, t, n, seed=42, i0=1, kind=CIRCUIT_MINROOT_REFERENCE, family=GENS_TRY_AND_INCREMENT, mode=None):
    x = o.rand_fe(seed, 0, o.Q)
    initial = State.from_ints(FIELD_FQ, x, 0, i0)             # y = 0, i = 1: src/nova/proof.rs:417-421
    pp = public_params(ctx, t, kind, family)
    vdf = PallasVDF.new() if mode is None else PallasVDF.new_with_mode(mode)
    z0, circuits = InverseMinRootCircuit.eval_and_make_circuits(vdf, t, n, initial)
    return pp, z0, circuits, initial, (x, 0, i0)


def check_instance(proof, which, side, want):
    """instance + witness of the product against an oracle Relaxed / Fresh."""
    m = SCAL[side]
    inst = proof.instance(which)
    z, E = proof.witness(which)
    assert aff_ints(inst["comm_W"], side) == tuple(want.comm_W)
    assert unmont(inst["X"], m) == list(want.X)
    zz = unmont(z, m)
    nvars = len(want.W)
    assert zz[:nvars] == list(want.W)
    if which == INST_FRESH_SECONDARY:
        assert zz[nvars:] == [1] + list(want.X) and E is None
        assert unmont(inst["u"].reshape(1, 4), m) == [1] and aff_ints(inst["comm_E"], side) == (0, 0)
    else:
        assert aff_ints(inst["comm_E"], side) == tuple(want.comm_E)
        assert unmont(inst["u"].reshape(1, 4), m) == [want.u]
        assert zz[nvars:] == [want.u] + list(want.X)
        assert unmont(E, m) == list(want.E)


def test_nova_proof(ctx):
    """test_nova_proof_aux(5, 3), src/nova/proof.rs:403-451 (its compress leg: tests/test_gpu_compress.py)."""
    t, n = 5, 3
    pp, z0, circuits, initial, init_ints = make(ctx, t, n)
    zi = [initial.x, initial.y, initial.i]
    proof = NovaVDFProof.prove_recursively(pp, circuits, t, z0)
    assert proof.num_steps() == n
    assert proof.verify(pp, n, z0, zi) is True
    # Ok(false) legs: wrong expected z_i, wrong step count, another z0
    wrong = [initial.y, initial.x, initial.i]
    assert proof.verify(pp, n, z0, wrong) is False
    assert proof.verify(pp, n + 1, z0, zi) is False
    assert proof.verify(pp, n, zi, zi) is False
    # zi_secondary == [0] (:386, :389-391)
    zp, zs = proof.zi()
    assert unmont(zs, o.P) == [0] and [bytes(zp[k]) for k in range(3)] == zi


@pytest.mark.parametrize("t,n", [(10, 200), (100, 20), (1000, 2)])
def test_reference_bench_cases(ctx, t, n):
    """The only measurement cases the reference defines (benches/nova.rs:62-66): (num_iters_per_step, num_steps) = (10, 200),
    (100, 20), (1000, 2), initial state x = element, y = 0, i = 0 (:24-26), the whole n-step proof (:28-59) -- here as a
    correctness case over the reference's own step circuit: prove_recursively, verify (Ok(true), and Ok(false) for a wrong z_i
    or step count), the long chain also through compress / verify / the wire (bench.py times the same three cases)."""
    x = o.rand_fe(4, t, o.Q)
    initial = State.from_ints(FIELD_FQ, x, 0, 0)
    pp = public_params(ctx, t)                                              # the default: the reference's circuit
    z0, circuits = InverseMinRootCircuit.eval_and_make_circuits(PallasVDF.new(), t, n, initial)
    zi = [initial.x, initial.y, initial.i]
    proof = NovaVDFProof.prove_recursively(pp, circuits, t, z0)
    assert proof.num_steps() == n and proof.verify(pp, n, z0, zi) is True
    assert proof.verify(pp, n - 1, z0, zi) is False and proof.verify(pp, n, z0, [zi[1], zi[0], zi[2]]) is False
    # the chain's end state is the forward evaluation's (src/minroot.rs:352-359), the proof walks it backwards to the start
    s = o.State(x, 0, 0)
    for _ in range(min(n, 3)):
        s = o.minroot_eval(s, t, o.FIELD_FQ)
    if n <= 3:
        assert State(*z0).to_ints(FIELD_FQ) == (s.x, s.y, s.i)
    if n == 200:
        from vdf_amd.nova import CompressedNovaVDFProof
        snark = proof.compress(pp)
        assert snark.verify(pp, n, z0, zi)
        assert CompressedNovaVDFProof.deserialize(pp, snark.serialize()).verify(pp, n, z0, zi)


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


@pytest.mark.parametrize("t,n,kind", [(5, 3, CIRCUIT_MINROOT_REFERENCE), (24, 3, CIRCUIT_MINROOT_REFERENCE), (7, 2, CIRCUIT_MINROOT_BOUND),
                                      (24, 3, CIRCUIT_MINROOT_BOUND)], ids=["t5-reference", "t24-reference", "t7-bound", "t24-bound"])
def test_prove_steps_replayed_by_the_oracle(ctx, t, n, kind):
    """Every quantity of every step, bit-exact against oracle/nova.py: parameters digest, the fresh primary instance,
    both cross-term commitments, both challenges, the three instances a proof carries and their witnesses."""
    pp, z0, circuits, initial, init_ints = make(ctx, t, n, seed=77, kind=kind)
    com = nv.CCommit()
    opp = nv.public_params(t, com, nv.GENS_SEED, nv.FAMILY_TRY_AND_INCREMENT, bound=(kind == CIRCUIT_MINROOT_BOUND))
    assert pp.digest() == opp.params
    for side in (0, 1):
        sz, sh = pp.sizes(side), opp.shapes[side]
        assert (sz["num_cons"], sz["num_vars"], sz["num_io"], sz["nnz"]) == (sh.num_cons, sh.num_vars, 2, len(sh.A) + len(sh.B) + len(sh.C))
    states = [o.State(*init_ints)]
    for _ in range(n):
        states.append(o.minroot_eval(states[-1], t, o.FIELD_FQ))
    z0i = [states[n].x, states[n].y, states[n].i]
    assert [State(*z0).to_ints(FIELD_FQ)[k] for k in range(3)] == z0i
    proof, want = None, None
    for k in range(n):
        proof = NovaVDFProof.prove_step(pp, proof, circuits, k, z0)
        want = nv.prove_step(opp, want, nv.InverseMinRootCircuit(t, states[n - k], states[n - k - 1], kind == CIRCUIT_MINROOT_BOUND), z0i)
        tr, ls = want.trace[-1], proof.last_step()
        assert aff_ints(ls["comm_W1"], 0) == tuple(tr["l1"].comm_W) and unmont(ls["X1"], o.Q) == tr["l1"].X
        if k:
            assert aff_ints(ls["comm_T1"], 0) == tuple(tr["T1"]) and aff_ints(ls["comm_T2"], 1) == tuple(tr["T2"])
            assert (ls["r1"], ls["r2"]) == (tr["r1"], tr["r2"])
        check_instance(proof, INST_RUNNING_PRIMARY, 0, want.r[0])
        check_instance(proof, INST_RUNNING_SECONDARY, 1, want.r[1])
        check_instance(proof, INST_FRESH_SECONDARY, 1, want.l2)
        zp, zs = proof.zi()
        assert unmont(zp, o.Q) == want.zi[0] and unmont(zs, o.P) == want.zi[1]
        assert nv.verify(opp, want, k + 1, z0i) is not None
    zi = [initial.x, initial.y, initial.i]
    assert proof.verify(pp, n, z0, zi)


@pytest.mark.parametrize("which", ["neptune-shaped", "small"])
def test_prove_steps_under_another_random_oracle_block(ctx, which):
    """The random oracle is a parameter block covered by the digest (vdf_nova_ro_params): under the
    neptune-shaped block (original Poseidon, width 25, 8 + 57 rounds: [UPSTREAM-RECALL], unpinned) and under a smaller
    instance of the same family, every quantity of every step is the oracle's under the same block, bit for bit -- shapes,
    digest, commitments, challenges, instances, witnesses -- the proof verifies, compresses and survives the wire; and the
    parameters differ from the default block's."""
    from oracle import poseidon as ps
    from vdf_amd.nova import ro_preset, RO_NEPTUNE_SHAPED
    t, n, kind = 5, 2, CIRCUIT_MINROOT_REFERENCE
    if which == "small":
        spec, ro = ps.RoSpec(family=1, width=9, full_rounds=8, partial_rounds=30), ro_preset(RO_NEPTUNE_SHAPED, width=9, partial_rounds=30)
    else:
        spec, ro = ps.NEPTUNE_SHAPED, ro_preset(RO_NEPTUNE_SHAPED)
    x = o.rand_fe(78, 0, o.Q)
    initial = State.from_ints(FIELD_FQ, x, 0, 1)
    pp = public_params(ctx, t, kind, GENS_TRY_AND_INCREMENT, ro=ro)
    assert pp.ro() == ro.as_dict()
    pp0 = public_params(ctx, t, kind, GENS_TRY_AND_INCREMENT)
    assert pp0.digest() != pp.digest() and pp0.sizes(0) != pp.sizes(0) and pp0.ro()["family"] == 0
    pp0.free()
    z0, circuits = InverseMinRootCircuit.eval_and_make_circuits(PallasVDF.new(), t, n, initial)
    com = nv.CCommit()
    with ps.using(spec):
        opp = nv.public_params(t, com, nv.GENS_SEED, nv.FAMILY_TRY_AND_INCREMENT)
        assert pp.digest() == opp.params
        for side in (0, 1):
            sz, sh = pp.sizes(side), opp.shapes[side]
            assert (sz["num_cons"], sz["num_vars"], sz["nnz"]) == (sh.num_cons, sh.num_vars, len(sh.A) + len(sh.B) + len(sh.C))
        states = [o.State(x, 0, 1)]
        for _ in range(n):
            states.append(o.minroot_eval(states[-1], t, o.FIELD_FQ))
        z0i = [states[n].x, states[n].y, states[n].i]
        proof, want = None, None
        for k in range(n):
            proof = NovaVDFProof.prove_step(pp, proof, circuits, k, z0)
            want = nv.prove_step(opp, want, nv.InverseMinRootCircuit(t, states[n - k], states[n - k - 1]), z0i)
            tr, ls = want.trace[-1], proof.last_step()
            assert aff_ints(ls["comm_W1"], 0) == tuple(tr["l1"].comm_W) and unmont(ls["X1"], o.Q) == tr["l1"].X
            if k:
                assert aff_ints(ls["comm_T1"], 0) == tuple(tr["T1"]) and aff_ints(ls["comm_T2"], 1) == tuple(tr["T2"])
                assert (ls["r1"], ls["r2"]) == (tr["r1"], tr["r2"])
            check_instance(proof, INST_RUNNING_PRIMARY, 0, want.r[0])
            check_instance(proof, INST_RUNNING_SECONDARY, 1, want.r[1])
            check_instance(proof, INST_FRESH_SECONDARY, 1, want.l2)
        assert nv.verify(opp, want, n, z0i) is not None
    zi = [initial.x, initial.y, initial.i]
    assert proof.verify(pp, n, z0, zi)
    snark = proof.compress(pp)
    assert snark.verify(pp, n, z0, zi)
    from vdf_amd.nova import CompressedNovaVDFProof
    again = CompressedNovaVDFProof.deserialize(pp, snark.serialize())
    assert again.verify(pp, n, z0, zi)


def test_tampered_witness_is_rejected(ctx):
    """Corrupting one word of any of the five witness vectors on the device must fail verification."""
    from vdf_amd._lib import lib
    from vdf_amd.nova import nova_lib
    t, n = 8, 2
    pp, z0, circuits, initial, _ = make(ctx, t, n, seed=5)
    zi = [initial.x, initial.y, initial.i]
    proof = NovaVDFProof.prove_recursively(pp, circuits, t, z0)
    assert proof.verify(pp, n, z0, zi)
    for which in (INST_RUNNING_PRIMARY, INST_RUNNING_SECONDARY, INST_FRESH_SECONDARY):
        dz, dE = C.c_void_p(), C.c_void_p()
        assert nova_lib.vdf_nova_proof_witness_ptrs(proof.handle, which, C.byref(dz), C.byref(dE)) == 0
        seg_b, seg_n = pp.segment()
        spots = [(dz.value, 32 * 40), (dE.value, 32 * 11)]
        if which == INST_RUNNING_PRIMARY:
            spots.append((dz.value, 32 * (seg_b + 7)))       # a MinRoot round variable (tmp2 of round 2) of the folded witness
        for base, off in spots:
            if not base:
                continue
            word = np.zeros(1, dtype="<u8")
            ctx._check(lib.vdf_dev_memcpy(ctx.handle, word.ctypes.data, base + off, 8))
            bad = word ^ np.uint64(1)
            ctx._check(lib.vdf_dev_memcpy(ctx.handle, base + off, bad.ctypes.data, 8))
            assert proof.verify(pp, n, z0, zi) is False
            ctx._check(lib.vdf_dev_memcpy(ctx.handle, base + off, word.ctypes.data, 8))
            assert proof.verify(pp, n, z0, zi) is True


def test_mismatched_z0_is_an_error(ctx):
    """StepCircuit::output's assertion (src/nova/proof.rs:147-149): z_i must be the circuit's result."""
    from vdf_amd.hip import VdfError
    t = 4
    pp, z0, circuits, initial, _ = make(ctx, t, 2, seed=3)
    with pytest.raises(VdfError):
        NovaVDFProof.prove_step(pp, None, circuits, 0, [initial.x, initial.y, initial.i])
    with pytest.raises(VdfError):
        NovaVDFProof.prove_step(pp, None, circuits, 1, z0)          # step 1's circuit first: result != z0


def test_lookahead_is_invisible(ctx):
    """Steps proven out of the order the lookahead expected, and from another circuits object, give the same proof."""
    t, n = 16, 4
    pp, z0, circuits, initial, _ = make(ctx, t, n, seed=11)
    a = NovaVDFProof.prove_recursively(pp, circuits, t, z0)
    z0b, circuits_b = InverseMinRootCircuit.eval_and_make_circuits(PallasVDF.new(), t, n, initial)
    b = None
    for k in range(n):
        b = NovaVDFProof.prove_step(pp, b, circuits_b if k % 2 else circuits, k, z0)     # alternate the source
    for which in (INST_RUNNING_PRIMARY, INST_RUNNING_SECONDARY, INST_FRESH_SECONDARY):
        ia, ib = a.instance(which), b.instance(which)
        for key in ia:
            assert np.array_equal(ia[key], ib[key]), (which, key)
    assert b.verify(pp, n, z0, [initial.x, initial.y, initial.i])


def test_early_rows_of_the_cross_term_are_invisible(ctx, monkeypatch):
    """prove_step commits the MinRoot rounds' share of T ahead of the rest of a step (vdf_nova_pp_early_rows): the same
    proof as with T in one piece, and the early run covers the rounds' constraints (3 per round, all but the first
    round's, which read more of the witness)."""
    t, n = 64, 4
    pp, z0, circuits, initial, _ = make(ctx, t, n, seed=12)
    if pp.tuning()["early_rows"] != 2:
        pytest.skip("the environment overrides the default under test (tools/gpu_env_matrix.sh)")
    rb, rn = pp.early_rows()
    assert 3 * t - 8 <= rn <= 3 * t + 2 and rb > 0
    a = NovaVDFProof.prove_recursively(pp, circuits, t, z0)
    pp1 = public_params(ctx, t, CIRCUIT_MINROOT_REFERENCE, GENS_TRY_AND_INCREMENT, early_rows=0)       # vdf_nova_tuning, not the environment
    assert pp1.early_rows() == (0, 0) and pp1.digest() == pp.digest()
    b = NovaVDFProof.prove_recursively(pp1, circuits, t, z0)
    for which in (INST_RUNNING_PRIMARY, INST_RUNNING_SECONDARY, INST_FRESH_SECONDARY):
        ia, ib = a.instance(which), b.instance(which)
        for key in ia:
            assert np.array_equal(ia[key], ib[key]), (which, key)
        for va, vb in zip(a.witness(which), b.witness(which)):
            assert (va is None and vb is None) or np.array_equal(va, vb)
    assert a.verify(pp, n, z0, [initial.x, initial.y, initial.i])


@pytest.mark.parametrize("kind", [CIRCUIT_MINROOT_REFERENCE, CIRCUIT_MINROOT_BOUND], ids=["reference", "bound"])
def test_early_rows_by_stencil_equal_the_sparse_kernel(ctx, kind, monkeypatch):
    """The early rows of the built-in circuits run WITHOUT the sparse matrices (vdf_nifs_cross_term_minroot): public_params
    compares the stencil with the shape's triples and reports it (vdf_nova_pp_stencil = variables per round); the proof --
    every instance, every witness, both running A z / B z / C z through the folds that follow -- is the one the generic
    sparse kernel gives (VDF_NOVA_STENCIL=0), at a t that is no multiple of the workgroup size."""
    t, n = 100, 5
    pp, z0, circuits, initial, _ = make(ctx, t, n, seed=21, kind=kind)
    tn = pp.tuning()
    if not (tn["stencil"] == 1 and tn["early_rows"] != 0):
        pytest.skip("the environment overrides the defaults under test (tools/gpu_env_matrix.sh)")
    assert pp.stencil() == (4 if kind == CIRCUIT_MINROOT_REFERENCE else 3)
    assert pp.early_rows()[1] == 3 * t + 1
    a = NovaVDFProof.prove_recursively(pp, circuits, t, z0)
    pp1 = public_params(ctx, t, kind, GENS_TRY_AND_INCREMENT, stencil=0)
    assert pp1.stencil() == 0 and pp1.early_rows() == pp.early_rows() and pp1.digest() == pp.digest()
    b = NovaVDFProof.prove_recursively(pp1, circuits, t, z0)
    for which in (INST_RUNNING_PRIMARY, INST_RUNNING_SECONDARY, INST_FRESH_SECONDARY):
        ia, ib = a.instance(which), b.instance(which)
        for key in ia:
            assert np.array_equal(ia[key], ib[key]), (which, key)
        for va, vb in zip(a.witness(which), b.witness(which)):
            assert (va is None and vb is None) or np.array_equal(va, vb)
    assert a.verify(pp, n, z0, [initial.x, initial.y, initial.i])
    # the compressed proofs agree byte for byte too (the running A z, B z, C z feed the sum-checks)
    sa, sb = a.compress(pp), b.compress(pp1)
    assert sa.to_bytes() == sb.to_bytes()
    assert sa.verify(pp, n, z0, [initial.x, initial.y, initial.i])


@pytest.mark.parametrize("kind", [CIRCUIT_MINROOT_REFERENCE, CIRCUIT_MINROOT_BOUND], ids=["reference", "bound"])
def test_public_params_flags_decline_the_accelerators_and_change_nothing(ctx, kind, monkeypatch):
    """vdf_nova_public_params_flags (include/vdf_nova.h): VDF_PP_NO_DIGIT_TABLES / VDF_PP_NO_EARLY_ROWS decline the HBM-hungry
    digit tables and the early rows of T; vdf_nova_pp_memory reports what a parameter set holds.  Parameters (digest) and
    every instance and witness of a proof are the same with and without them -- they are accelerators, not protocol."""
    from vdf_amd.nova import PP_NO_DIGIT_TABLES, PP_NO_EARLY_ROWS
    t, n = 96, 4
    pp, z0, circuits, initial, _ = make(ctx, t, n, seed=14, kind=kind)
    tn = pp.tuning()
    if tn["early_rows"] == 0 or tn["digit_window"] == -1 or tn["flags"]:
        pytest.skip("the environment overrides the defaults under test (tools/gpu_env_matrix.sh)")
    mem = pp.memory()
    assert mem["digit_tables_skipped"] == 0 and min(mem["digit_table_bytes"]) > 0 and pp.early_rows()[1] > 0
    # the primary side's figures include the derived generators of the packed commitment (reference circuit) and their table
    extra = (3 * t + 4) if (kind == CIRCUIT_MINROOT_REFERENCE and tn["packed_commit"]) else 0
    assert mem["gens_bytes"][0] == 64 * (pp.sizes(0)["num_gens"] + extra) and mem["table_bytes"][0] >= 15 * 64 * pp.sizes(0)["num_gens"]
    assert mem["table_bytes"][1] % (64 * pp.sizes(1)["num_gens"]) == 0          # whole tables, as the library holds them
    a = NovaVDFProof.prove_recursively(pp, circuits, t, z0)
    for flags in (PP_NO_DIGIT_TABLES, PP_NO_EARLY_ROWS, PP_NO_DIGIT_TABLES | PP_NO_EARLY_ROWS):
        pp1 = public_params(ctx, t, kind, GENS_TRY_AND_INCREMENT, flags)
        assert pp1.digest() == pp.digest()
        m1 = pp1.memory()
        assert (m1["digit_table_bytes"] == [0, 0]) == bool(flags & PP_NO_DIGIT_TABLES)
        assert (pp1.early_rows() == (0, 0)) == bool(flags & PP_NO_EARLY_ROWS)
        b = NovaVDFProof.prove_recursively(pp1, circuits, t, z0)
        for which in (INST_RUNNING_PRIMARY, INST_RUNNING_SECONDARY, INST_FRESH_SECONDARY):
            ia, ib = a.instance(which), b.instance(which)
            for key in ia:
                assert np.array_equal(ia[key], ib[key]), (flags, which, key)
            for va, vb in zip(a.witness(which), b.witness(which)):
                assert (va is None and vb is None) or np.array_equal(va, vb)
        assert b.verify(pp1, n, z0, [initial.x, initial.y, initial.i])
        b.free(); pp1.free()
    with pytest.raises(Exception):
        public_params(ctx, t, kind, GENS_TRY_AND_INCREMENT, 64)          # unknown flag


def test_two_chains_proven_concurrently(ctx):
    """Two host threads, two contexts, two chains at once (the bench's aggregate leg): the helper threads of the witness
    synthesis are taken by one prover at a time and the other synthesises inline -- both must give the proofs they give
    when run alone."""
    import threading
    import vdf_amd
    t, n = 64, 5
    ctxs = [ctx, vdf_amd.Context(0)]
    work = [make(c, t, n, seed=21 + k) for k, c in enumerate(ctxs)]
    alone = [NovaVDFProof.prove_recursively(w[0], w[2], t, w[1]) for w in work]
    both, errs = [None, None], []

    def run(k):
        try:
            pp, z0, circuits, _, _ = work[k]
            pr = None
            for j in range(n):
                pr = NovaVDFProof.prove_step(pp, pr, circuits, j, z0)
            both[k] = pr
        except Exception as e:                                  # noqa: BLE001
            errs.append(e)

    ths = [threading.Thread(target=run, args=(k,)) for k in range(2)]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    assert not errs, errs
    for k in range(2):
        pp, z0, _, initial, _ = work[k]
        assert both[k].verify(pp, n, z0, [initial.x, initial.y, initial.i])
        for which in (INST_RUNNING_PRIMARY, INST_RUNNING_SECONDARY, INST_FRESH_SECONDARY):
            ia, ib = alone[k].instance(which), both[k].instance(which)
            for key in ia:
                assert np.array_equal(ia[key], ib[key]), (k, which, key)
    for k in range(2):
        alone[k].free(); both[k].free(); work[k][0].free()
    ctxs[1].close()


def _canon(cref, field, arr):
    out = np.zeros_like(arr)
    cref.lib().ref_fe_from_mont(field, cref.p(np.ascontiguousarray(arr)), arr.shape[0], cref.p(out))
    return out


@pytest.mark.parametrize("t,n,kind", [(1024, 3, CIRCUIT_MINROOT_REFERENCE), (1 << 16, 2, CIRCUIT_MINROOT_REFERENCE),
                                      (1024, 3, CIRCUIT_MINROOT_BOUND), (1 << 16, 2, CIRCUIT_MINROOT_BOUND)],
                         ids=["t1024-n3-reference", "t65536-n2-reference", "t1024-n3-bound", "t65536-n2-bound"])
def test_one_fold_replayed_by_the_c_oracle_at_baseline_sizes(ctx, cref, t, n, kind):
    """BASELINE config 1 (t = 1024, 3 steps) and config 3 (t = 2^16) checked against the CPU restatements instead of the
    product's own verifier: generators with known discrete logarithms (tests only), then for the LAST step --
      * both R1CS shapes: the parameters digest equals the oracle's (every triple of both matrices triples);
      * the MinRoot rounds of the fresh witness = the C restatement's ref_step_witness (src/nova/proof.rs:162-189) -- for the
        reference's circuit (4 variables per round, :167-189) its 4t + 1 values AS THEY ARE, for the bound form without new_x;
      * commitments of the fresh witness, of the cross term T and of the folded W, E = [sum s_i k_i] G (discrete-log identity);
      * A z, B z, C z (C restatement ref_spmv over the ORACLE's shape), T (ref_cross_term), W' = W + r W2 and E' = E + r T
        (ref_axpy) element for element, u' = u + r, X' = X + r X2.
    Then the product's own verify, and zi."""
    from vdf_amd.nova import INST_FRESH_PRIMARY_LAST
    from vdf_amd.nova import nova_lib
    L, fld, m = cref.lib(), o.FIELD_FQ, o.Q
    reference = kind == CIRCUIT_MINROOT_REFERENCE
    pp, z0, circuits, initial, init_ints = make(ctx, t, n, seed=3, kind=kind, family=GENS_KNOWN_DLOG,
                                                mode=EvalMode.LTRAddChainSequential)
    opp = nv.public_params(t, None, nv.GENS_SEED, nv.FAMILY_KNOWN_DLOG, bound=not reference)
    assert pp.digest() == opp.params
    sh = opp.shapes[0]
    nvar, nc = sh.num_vars, sh.num_cons
    seg_b, seg_n = pp.segment()
    assert seg_n == (4 if reference else 3) * t + 1 and seg_b + seg_n <= nvar
    proof = None
    for k in range(n - 1):
        proof = NovaVDFProof.prove_step(pp, proof, circuits, k, z0)
    z_old, E_old = proof.witness(INST_RUNNING_PRIMARY)
    inst_old = proof.instance(INST_RUNNING_PRIMARY)
    proof = NovaVDFProof.prove_step(pp, proof, circuits, n - 1, z0)
    ls = proof.last_step()
    z2, _ = proof.witness(INST_FRESH_PRIMARY_LAST)
    z_new, E_new = proof.witness(INST_RUNNING_PRIMARY)
    inst_new = proof.instance(INST_RUNNING_PRIMARY)
    # MinRoot rounds of the fresh witness: the C restatement's 4t + 1 values -- unstripped for the reference's circuit
    # (new_x, tmp1, tmp2, new_y per round, then final_i: src/nova/proof.rs:167-189, :122-126), without new_x for the bound form
    res, _inp = circuits.states(n - 1)
    st = np.frombuffer(res.x + res.y + res.i, dtype="<u8").reshape(3, 4).copy()
    Wref = cref.fe_array(4 * t + 1)
    L.ref_step_witness(fld, cref.p(st), t, cref.p(Wref))
    want_seg = Wref if reference else np.concatenate([Wref[:4 * t].reshape(t, 4, 4)[:, 1:, :].reshape(3 * t, 4), Wref[4 * t:]])
    assert np.array_equal(z2[seg_b:seg_b + seg_n], want_seg)
    assert unmont(z2[nvar:nvar + 1], m) == [1] and np.array_equal(z2[nvar + 1:], ls["X1"])
    # commitments by the discrete-log identity
    dl = lambda vec: o.msm_by_dlog_limbs(_canon(cref, fld, np.ascontiguousarray(vec)), o.CURVE_PALLAS, nv.GENS_SEED) or (0, 0)
    assert aff_ints(ls["comm_W1"], 0) == dl(z2[:nvar])
    # the fold through the C restatement over the oracle's shape
    def coo(mat):
        rows = np.array([e[0] for e in mat], dtype=np.uint32)
        cols = np.array([e[1] for e in mat], dtype=np.uint32)
        vals = limbs([o.to_mont(e[2], m) for e in mat])
        return rows, cols, vals
    mats = [coo(x) for x in (sh.A, sh.B, sh.C)]

    def mv(z):
        out = []
        for rows, cols, vals in mats:
            e = cref.fe_array(nc)
            L.ref_spmv(fld, cref.p(rows), cref.p(cols), cref.p(vals), len(rows), cref.p(np.ascontiguousarray(z)), nc, cref.p(e))
            out.append(e)
        return out
    abc1, abc2 = mv(z_old), mv(z2)
    T = cref.fe_array(nc)
    L.ref_cross_term(fld, *(cref.p(x) for x in abc1 + abc2), cref.p(np.ascontiguousarray(inst_old["u"].reshape(1, 4))), nc, cref.p(T))
    assert aff_ints(ls["comm_T1"], 0) == dl(T)
    r = limbs([o.to_mont(ls["r1"], m)])
    W_exp, E_exp = cref.fe_array(nvar + 3), cref.fe_array(nc)
    L.ref_axpy(fld, cref.p(np.ascontiguousarray(z_old)), cref.p(r), cref.p(np.ascontiguousarray(z2)), nvar + 3, cref.p(W_exp))
    L.ref_axpy(fld, cref.p(np.ascontiguousarray(E_old)), cref.p(r), cref.p(T), nc, cref.p(E_exp))
    assert np.array_equal(z_new, W_exp)            # W', and with it u' = u + r and X' = X + r X2 (they ride in z)
    assert np.array_equal(E_new, E_exp)
    assert np.array_equal(z_new[nvar], inst_new["u"]) and np.array_equal(z_new[nvar + 1:], inst_new["X"])
    assert aff_ints(inst_new["comm_W"], 0) == dl(z_new[:nvar]) and aff_ints(inst_new["comm_E"], 0) == dl(E_new)
    # relaxed satisfiability of the folded instance, by the C restatement
    a, b, c = mv(z_new)
    ab, uc = cref.fe_array(nc), cref.fe_array(nc)
    L.ref_fe_mul(fld, cref.p(a), cref.p(b), nc, cref.p(ab))
    uvec = np.ascontiguousarray(np.broadcast_to(inst_new["u"].reshape(1, 4), (nc, 4)))
    L.ref_fe_mul(fld, cref.p(uvec), cref.p(c), nc, cref.p(uc))
    lhs, zero1 = cref.fe_array(nc), limbs([o.to_mont(m - 1, m)])
    L.ref_axpy(fld, cref.p(ab), cref.p(zero1), cref.p(uc), nc, cref.p(lhs))          # ab - u c
    assert np.array_equal(lhs, E_new)
    assert proof.verify(pp, n, z0, [initial.x, initial.y, initial.i])


def test_parameters_from_a_label(ctx):
    """gens_family = VDF_GENS_LABEL_SHAKE: Pedersen generators derived from a label through SHAKE256 (as nova-snark's
    CommitGens are; the points themselves are checked against the oracle in tests/test_gpu_msm.py).  Same shapes, another
    digest; a proof under them verifies, compresses and verifies compressed."""
    from vdf_amd.nova import GENS_LABEL_SHAKE
    t, n = 9, 3
    pp, z0, circuits, initial, _ = make(ctx, t, n, seed=8, family=GENS_LABEL_SHAKE)
    pp_tai = public_params(ctx, t)
    assert pp.digest() != pp_tai.digest() and pp.sizes(0) == pp_tai.sizes(0) and pp.sizes(1) == pp_tai.sizes(1)
    zi = [initial.x, initial.y, initial.i]
    proof = NovaVDFProof.prove_recursively(pp, circuits, t, z0)
    assert proof.verify(pp, n, z0, zi)
    snark = proof.compress(pp)
    assert snark.verify(pp, n, z0, zi)
