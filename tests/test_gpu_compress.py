"""GPU: NovaVDFProof::compress and verification of the compressed proof (src/nova/proof.rs:360-368, :383; the
reference's own test at :446-450): one argument per side of the curve cycle (SS1, SS2 of :32-33).  The arguments the
product makes must equal, byte for byte, the ones the Python restatement (oracle/nova.py compress over
oracle/spartan.py) makes for the same proof; at larger sizes they must verify, and tampering must be rejected."""
import time

import numpy as np
import pytest

from oracle import nova as nv, pasta as o, wire as w
from test_gpu_nova import make
from vdf_amd.minroot import State, FIELD_FQ
from vdf_amd.nova import NovaVDFProof, CIRCUIT_MINROOT_BOUND, CIRCUIT_MINROOT_REFERENCE

pytestmark = pytest.mark.gpu


def _zi(init_ints):
    s = State.from_ints(FIELD_FQ, *init_ints)
    return [s.x, s.y, s.i]


def oracle_proof(t, n, init_ints, bound=False):
    """The same chain proven by the oracle: (public parameters, RecursiveSNARK, z0 as integers)."""
    opp = nv.public_params(t, nv.CCommit(), nv.GENS_SEED, nv.FAMILY_TRY_AND_INCREMENT, bound=bound)
    states = [o.State(*init_ints)]
    for _ in range(n):
        states.append(o.minroot_eval(states[-1], t, o.FIELD_FQ))
    z0 = [states[n].x, states[n].y, states[n].i]
    s = None
    for k in range(n):
        s = nv.prove_step(opp, s, nv.InverseMinRootCircuit(t, states[n - k], states[n - k - 1], bound), z0)
    return opp, s, z0


@pytest.mark.parametrize("t,n,kind", [(3, 2, CIRCUIT_MINROOT_BOUND), (5, 3, CIRCUIT_MINROOT_BOUND), (5, 3, CIRCUIT_MINROOT_REFERENCE),
                                      (1024, 2, CIRCUIT_MINROOT_REFERENCE), (1024, 2, CIRCUIT_MINROOT_BOUND)],
                         ids=["t3-bound", "t5-bound", "t5-reference", "t1024-reference", "t1024-bound"])
def test_compressed_arguments_equal_the_oracles(ctx, t, n, kind):
    """compress (src/nova/proof.rs:360-368) byte for byte against oracle/nova.py compress over oracle/spartan.py, up to
    BASELINE config 1's step size (t = 1024: 2^14-row sum-checks, the oracle's MSMs in the C restatement) and for both
    forms of the step circuit."""
    pp, z0, circuits, initial, init_ints = make(ctx, t, n, seed=31, kind=kind)
    proof = NovaVDFProof.prove_recursively(pp, circuits, t, z0)
    opp, want_s, z0i = oracle_proof(t, n, init_ints, bound=(kind == CIRCUIT_MINROOT_BOUND))
    assert pp.digest() == opp.params
    want = nv.compress(opp, want_s)
    assert nv.verify_compressed(opp, want, n, z0i) is not None
    snark = proof.compress(pp)
    assert snark.to_bytes() == w.encode_flat_arguments(want)
    assert snark.serialize() == w.encode_compressed_proof(t, opp.params, want)
    assert snark.verify(pp, n, z0, _zi(init_ints))
    # compressing does not disturb the running proof
    assert proof.verify(pp, n, z0, _zi(init_ints))
    assert proof.compress(pp).serialize() == snark.serialize()


def test_two_queue_openings_change_no_byte(ctx):
    """vdf_nova_tuning.compress_queues: the primary side's two openings on two queues half a round apart (the default) and in
    lockstep on one produce the same proof -- only launches move, the transcript sees the same sequence -- at a size where
    the two openings have different numbers of rounds (t = 1024: 2^13 / 2^12 entries)."""
    from vdf_amd.nova import public_params, GENS_TRY_AND_INCREMENT
    t, n = 1024, 2
    _, z0, circuits, initial, init_ints = make(ctx, t, n, seed=77)
    pp = public_params(ctx, t, CIRCUIT_MINROOT_REFERENCE, GENS_TRY_AND_INCREMENT, compress_queues=1)   # (explicit: the environment may override the default)
    assert pp.tuning()["compress_queues"] == 1
    proof = NovaVDFProof.prove_recursively(pp, circuits, t, z0)
    a = proof.compress(pp)
    pp0 = public_params(ctx, t, CIRCUIT_MINROOT_REFERENCE, GENS_TRY_AND_INCREMENT, compress_queues=0)
    assert pp0.tuning()["compress_queues"] == 0 and pp0.digest() == pp.digest()
    proof0 = NovaVDFProof.prove_recursively(pp0, circuits, t, z0)
    b = proof0.compress(pp0)
    assert a.serialize() == b.serialize()
    assert a.verify(pp, n, z0, _zi(init_ints)) and b.verify(pp0, n, z0, _zi(init_ints))
    # again on the same parameter set: the queues made by the first call are reused
    assert proof.compress(pp).serialize() == a.serialize()


def test_nova_proof_compress_leg(ctx):
    """test_nova_proof_aux(5, 3), src/nova/proof.rs:446-450: compress succeeds and the compressed proof verifies."""
    t, n = 5, 3
    pp, z0, circuits, initial, init_ints = make(ctx, t, n)
    zi = _zi(init_ints)
    compressed = NovaVDFProof.prove_recursively(pp, circuits, t, z0).compress(pp)
    assert compressed.verify(pp, n, z0, zi) is True
    assert compressed.verify(pp, n, z0, [zi[1], zi[0], zi[2]]) is False
    assert compressed.verify(pp, n + 1, z0, zi) is False


def test_compress_and_verify_at_t_1024_and_tampering(ctx):
    """BASELINE config 1 sizes (t = 1024, 3 steps): the reference's test_nova_proof flow including compress."""
    t, n = 1024, 3
    pp, z0, circuits, initial, init_ints = make(ctx, t, n, seed=11)
    proof = NovaVDFProof.prove_recursively(pp, circuits, t, z0)
    zi = _zi(init_ints)
    assert proof.verify(pp, n, z0, zi)
    snark = proof.compress(pp)
    assert snark.verify(pp, n, z0, zi)
    good = snark.to_bytes()
    size = 0
    sect = []
    for side in (0, 1):
        sz = pp.sizes(side)
        s = (sz["num_cons"] - 1).bit_length()
        l1 = (sz["num_vars"] - 1).bit_length() + 1
        kW, kE = (l1 - 1) - 4, s - 4                   # halving rounds: the arguments stop at 16 elements
        sect.append((size, s, l1, kW, kE))
        size += 32 * (3 * s + 4 + 2 * l1 + 1 + 16 + 16) + 128 * (kW + kE)
    assert len(good) == size
    assert not snark.verify(pp, n, z0, [zi[1], zi[0], zi[2]])
    assert not snark.verify(pp, n - 1, z0, zi)
    # every section of both arguments: one flipped low bit must be rejected (or refused as non-canonical / off the curve)
    for side, (base, s, l1, kW, kE) in enumerate(sect):
        head = base + 32 * (3 * s + 4 + 2 * l1 + 1)
        offsets = {"outer": base + 40, "claims": base + 32 * 3 * s + 33, "inner": base + 32 * (3 * s + 4) + 64 + 1,
                   "w_eval": base + 32 * (3 * s + 4 + 2 * l1), "ipaW.L": head + 3, "ipaW.a[0]": head + 128 * kW,
                   "ipaW.a[9]": head + 128 * kW + 32 * 9 + 2, "ipaE.R": head + 128 * kW + 32 * 16 + 64 + 5,
                   "ipaE.a[15]": head + 128 * (kW + kE) + 32 * 31}
        for name, off in offsets.items():
            bad = bytearray(good)
            bad[off] ^= 1
            try:
                snark.set_bytes(bytes(bad))
            except Exception:
                continue
            assert not snark.verify(pp, n, z0, zi), (side, name)
    snark.set_bytes(good)
    assert snark.verify(pp, n, z0, zi)
    with pytest.raises(Exception):
        snark.set_bytes(good[:-1])
    with pytest.raises(Exception):
        snark.set_bytes(b"\xff" * len(good))            # not canonical


@pytest.mark.parametrize("kind", [CIRCUIT_MINROOT_REFERENCE, CIRCUIT_MINROOT_BOUND], ids=["reference", "bound"])
def test_compress_full_size_t_2_16(ctx, kind):
    """t = 2^16 (BASELINE config 3 shape; the reference's circuit: 2^18 constraints, 2^19 variables and generators on the
    primary side): completes and verifies."""
    t, n = 1 << 16, 2
    pp, z0, circuits, initial, init_ints = make(ctx, t, n, seed=5, kind=kind)
    proof = NovaVDFProof.prove_recursively(pp, circuits, t, z0)
    t0 = time.perf_counter()
    snark = proof.compress(pp)
    t1 = time.perf_counter()
    assert snark.verify(pp, n, z0, _zi(init_ints))
    t2 = time.perf_counter()
    print(f"compress {1e3 * (t1 - t0):.1f} ms, verify {1e3 * (t2 - t1):.1f} ms, arguments {len(snark.to_bytes())} bytes, "
          f"on the wire {len(snark.serialize())} bytes")
    bad = bytearray(snark.to_bytes())
    bad[100] ^= 1
    snark.set_bytes(bytes(bad))
    assert not snark.verify(pp, n, z0, _zi(init_ints))


def test_config_5_compressed_snark_of_2_20_folded_iterations(ctx):
    """BASELINE config 5's workload on one GPU: 16 steps of 2^16 MinRoot iterations (2^20 folded iterations) of the reference's
    circuit -> compress (src/nova/proof.rs:360-368) -> verify the compressed proof (:383, the reference's test at :446-450) ->
    wire round trip in another context, which must accept it; a wrong z_i, a wrong step count and a flipped bit are rejected."""
    import vdf_amd
    from vdf_amd.minroot import EvalMode
    from vdf_amd.nova import CompressedNovaVDFProof as CompressedSNARK, public_params
    t, n = 1 << 16, 16
    pp, z0, circuits, initial, init_ints = make(ctx, t, n, seed=55, i0=0, kind=CIRCUIT_MINROOT_REFERENCE, mode=EvalMode.LTRAddChainSequential)
    circuits.upload(ctx)
    zi = _zi(init_ints)
    t0 = time.perf_counter()
    proof = NovaVDFProof.prove_recursively(pp, circuits, t, z0)
    t1 = time.perf_counter()
    assert proof.num_steps() == n and proof.verify(pp, n, z0, zi)
    t2 = time.perf_counter()
    snark = proof.compress(pp)
    t3 = time.perf_counter()
    assert snark.verify(pp, n, z0, zi)
    t4 = time.perf_counter()
    wire = snark.serialize()
    print(f"2^20 iterations in {n} steps: prove {1e3 * (t1 - t0):.1f} ms, compress {1e3 * (t3 - t2):.1f} ms, "
          f"verify compressed {1e3 * (t4 - t3):.1f} ms, {len(wire)} bytes on the wire")
    assert not snark.verify(pp, n, z0, [zi[1], zi[0], zi[2]]) and not snark.verify(pp, n - 1, z0, zi)
    ctx_v = vdf_amd.Context(0)                       # a verifier of its own: parameters derived again, the proof from bytes
    pp_v = public_params(ctx_v, t, CIRCUIT_MINROOT_REFERENCE)
    assert pp_v.digest() == pp.digest()
    got = CompressedSNARK.deserialize(pp_v, wire)
    assert got.verify(pp_v, n, z0, zi) and got.serialize() == wire
    bad = bytearray(wire)
    bad[len(bad) // 2] ^= 4
    try:
        assert not CompressedSNARK.deserialize(pp_v, bytes(bad)).verify(pp_v, n, z0, zi)
    except vdf_amd.VdfError:
        pass                                         # refused at decoding: non-canonical or off the curve
    got.free(); pp_v.free(); ctx_v.close()
