"""GPU: the wire formats of include/vdf_nova.h.  "VDFSNK02" -- the compressed proof a prover ships to a verifier in
another process; "VDFRSK01" -- the running proof as a checkpoint that prove_step resumes from.  The reference keeps
proofs in memory only (src/nova/proof.rs:52-55): the expected bytes are those of the restatement oracle/wire.py,
and the behaviour asked of a decoded proof is the reference's own test flow (:403-451)."""
import time

import numpy as np
import pytest

import vdf_amd
from oracle import pasta as o
from oracle import spartan as sp
from oracle import wire as w
from util import unmont
from test_gpu_nova import make, shape_digest, aff_ints, gens
from test_gpu_compress import _zi, _pt
from vdf_amd.nova import NovaVDFProof, CompressedNovaVDFProof, public_params

pytestmark = pytest.mark.gpu
Q = o.Q


def _chain(proof, n):
    recs = [proof.step_record(k) for k in range(n)]
    z = [unmont(recs[0]["X"][:3], Q)] + [unmont(r["X"][3:], Q) for r in recs]
    cw = [_pt(aff_ints(r["comm_w"])) for r in recs]
    cT = [_pt(aff_ints(r["comm_T"])) for r in recs]
    return z, cw, cT


@pytest.mark.parametrize("t,n", [(3, 3), (5, 1)])
def test_compressed_proof_bytes_equal_the_oracles(ctx, t, n):
    pp, z0, circuits, initial, init_ints = make(ctx, t, n, seed=31)
    proof = NovaVDFProof.prove_recursively(pp, circuits, t, z0)
    sh = o.step_circuit_shape(t, o.FIELD_FQ)
    inst = proof.instance()
    gW, gE = proof.witness()
    W, E = unmont(gW, Q), unmont(gE, Q)
    u, X = unmont(inst["u"].reshape(1, 4), Q)[0], unmont(inst["X"], Q)
    cW, cE = _pt(aff_ints(inst["comm_W"])), _pt(aff_ints(inst["comm_E"]))
    N = pp.sizes()["num_gens"]
    digest = shape_digest(sh, t)
    want = sp.prove(sh, digest, gens(N), gens(1, start=N)[0], cW, cE, u, X, W, E)
    z, cw, cT = _chain(proof, n)
    snark = proof.compress(pp)
    got = snark.serialize()
    assert got == w.encode_compressed_proof(t, digest, z, cw, cT, want)
    assert len(got) == w.chain_size(n) + len(w.encode_argument(want))
    # running proof: same chain under its own magic, then the witness
    assert proof.serialize() == w.encode_running_proof(t, digest, z, cw, cT, W, E)
    # and back
    again = CompressedNovaVDFProof.deserialize(pp, got)
    assert again.serialize() == got and again.to_bytes() == snark.to_bytes()
    assert again.verify(pp, n, z0, _zi(init_ints))


def test_product_bytes_equal_the_committed_vector(ctx, golden):
    """tests/golden/vectors.json "wire_t6" (made on the CPU by the oracle alone): the product's bytes for the same chain."""
    import hashlib
    g = golden["wire_t6"]
    t, n = g["t"], g["steps"]
    pp, z0, circuits, initial, init_ints = make(ctx, t, n, seed=g["seed"], i0=g["i0"])
    proof = NovaVDFProof.prove_recursively(pp, circuits, t, z0)
    assert proof.compress(pp).serialize().hex() == g["compressed_proof_hex"]
    running = proof.serialize()
    assert len(running) == g["running_proof_len"] and hashlib.sha256(running).hexdigest() == g["running_proof_sha256"]
    assert CompressedNovaVDFProof.deserialize(pp, bytes.fromhex(g["compressed_proof_hex"])).verify(pp, n, z0, _zi(init_ints))


def test_a_verifier_in_its_own_context_accepts_the_bytes_and_nothing_else(ctx):
    """The prover's objects never reach the verifier: fresh context, public parameters derived again, bytes only."""
    t, n = 64, 3
    pp, z0, circuits, initial, init_ints = make(ctx, t, n, seed=17)
    zi = _zi(init_ints)
    good = NovaVDFProof.prove_recursively(pp, circuits, t, z0).compress(pp).serialize()
    with vdf_amd.Context(0) as vctx:
        vpp = public_params(vctx, t)
        snark = CompressedNovaVDFProof.deserialize(vpp, good)
        assert snark.verify(vpp, n, z0, zi)
        assert not snark.verify(vpp, n, z0, [zi[1], zi[0], zi[2]])
        assert not snark.verify(vpp, n - 1, z0, zi)
        # one flipped bit anywhere: refused at decoding, or decoded and rejected
        chain = w.chain_size(n)
        offsets = {"magic": 3, "t": 8, "steps": 16, "digest": 30, "z0": 56 + 40, "z1": 56 + 96 + 5, "comm_w0": 56 + 96 + 96 + 7,
                   "z2": 56 + 96 + 128 + 64, "comm_T1": 56 + 96 + 128 + 128 + 1, "sign bit": 56 + 96 + 96 + 31,
                   "outer": chain + 33, "ipa point": len(good) - 32 * 16 - 64 + 9, "ipa a": len(good) - 1}
        for name, off in offsets.items():
            bad = bytearray(good)
            bad[off] ^= 0x80 if name == "sign bit" else 1
            try:
                s2 = CompressedNovaVDFProof.deserialize(vpp, bytes(bad))
            except vdf_amd.VdfError:
                continue
            assert not s2.verify(vpp, n, z0, zi), name
            s2.free()
        for cut in (good[:-1], good + b"\0", good[:chain], good[:40], b""):
            with pytest.raises(vdf_amd.VdfError):
                CompressedNovaVDFProof.deserialize(vpp, cut)
        with pytest.raises(vdf_amd.VdfError):
            NovaVDFProof.deserialize(vpp, good)                       # the other format's magic
        # other public parameters (another t): refused by the digest
        vpp2 = public_params(vctx, 32)
        with pytest.raises(vdf_amd.VdfError):
            CompressedNovaVDFProof.deserialize(vpp2, good)
        vpp2.free()
        snark.free()
        vpp.free()


def test_checkpoint_and_resume_gives_the_same_proof(ctx):
    """Two steps, checkpoint, then two more steps from the original and from the restored proof (in its own context):
    identical records, identical folded witness, both verify."""
    t, n = 16, 4
    pp, z0, circuits, initial, init_ints = make(ctx, t, n, seed=23)
    zi = _zi(init_ints)
    proof = None
    for k in range(2):
        proof = NovaVDFProof.prove_step(pp, proof, circuits, k, z0)
    blob = proof.serialize()
    assert len(blob) == w.chain_size(2) + 32 * (pp.sizes()["num_vars"] + pp.sizes()["num_cons"])
    with vdf_amd.Context(0) as c2:
        pp2 = public_params(c2, t)
        restored = NovaVDFProof.deserialize(pp2, blob)
        assert restored.num_steps() == 2
        assert restored.serialize() == blob
        for k in range(2, n):
            proof = NovaVDFProof.prove_step(pp, proof, circuits, k, z0)
            restored = NovaVDFProof.prove_step(pp2, restored, circuits, k, z0)
        assert restored.serialize() == proof.serialize()
        assert restored.verify(pp2, n, z0, zi) and proof.verify(pp, n, z0, zi)
        assert restored.compress(pp2).serialize() == proof.compress(pp).serialize()
        # a witness that does not open the folded commitments is refused at load
        bad = bytearray(blob)
        bad[w.chain_size(2) + 32 * 5] ^= 1
        with pytest.raises(vdf_amd.VdfError):
            NovaVDFProof.deserialize(pp2, bytes(bad))
        bad = bytearray(blob)
        bad[-1] = 0xFF                                                  # not canonical
        with pytest.raises(vdf_amd.VdfError):
            NovaVDFProof.deserialize(pp2, bytes(bad))
        with pytest.raises(vdf_amd.VdfError):
            NovaVDFProof.deserialize(pp2, blob[:-32])
        restored.free()
        pp2.free()


def test_wire_formats_at_t_2_16(ctx):
    """Full-size shape: sizes and timings of both encodings."""
    t, n = 1 << 16, 3
    pp, z0, circuits, initial, init_ints = make(ctx, t, n, seed=5)
    zi = _zi(init_ints)
    proof = None
    for k in range(2):
        proof = NovaVDFProof.prove_step(pp, proof, circuits, k, z0)
    t0 = time.perf_counter()
    blob = proof.serialize()
    t1 = time.perf_counter()
    restored = NovaVDFProof.deserialize(pp, blob)
    t2 = time.perf_counter()
    restored = NovaVDFProof.prove_step(pp, restored, circuits, 2, z0)
    assert restored.verify(pp, n, z0, zi)
    snark = restored.compress(pp)
    t3 = time.perf_counter()
    wire = snark.serialize()
    t4 = time.perf_counter()
    back = CompressedNovaVDFProof.deserialize(pp, wire)
    t5 = time.perf_counter()
    assert back.verify(pp, n, z0, zi)
    print(f"checkpoint {len(blob) / 2**20:.1f} MiB: save {1e3 * (t1 - t0):.1f} ms, load {1e3 * (t2 - t1):.1f} ms; "
          f"compressed proof {len(wire)} bytes (argument {len(wire) - w.chain_size(n)}): "
          f"encode {1e3 * (t4 - t3):.2f} ms, decode {1e3 * (t5 - t4):.2f} ms")
