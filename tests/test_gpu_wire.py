"""GPU: the wire formats of include/vdf_nova.h.  "VDFSNK03" -- the compressed proof a prover ships to a verifier in
another process; "VDFRSK02" -- the running proof as a checkpoint that prove_step resumes from.  The reference keeps
proofs in memory only (src/nova/proof.rs:52-55): the expected bytes are those of the restatement oracle/wire.py,
and the behaviour asked of a decoded proof is the reference's own test flow (:403-451)."""
import hashlib
import time

import numpy as np
import pytest

import vdf_amd
from oracle import nova as nv, pasta as o, wire as w
from test_gpu_nova import make
from test_gpu_compress import _zi, oracle_proof
from vdf_amd.nova import NovaVDFProof, CompressedNovaVDFProof, public_params

pytestmark = pytest.mark.gpu
HEADER = 48 + 5 * 32 * 2 + 3 * 32 + 32 + 128          # magic, t, digest | three instances | T | both z_i


def test_proof_bytes_equal_the_oracles(ctx):
    t, n = 4, 3
    pp, z0, circuits, initial, init_ints = make(ctx, t, n, seed=31)
    proof = NovaVDFProof.prove_recursively(pp, circuits, t, z0)
    opp, want_s, z0i = oracle_proof(t, n, init_ints)
    running = proof.serialize()
    assert running == w.encode_running_proof(t, opp.params, want_s, z0i)
    snark = proof.compress(pp)
    got = snark.serialize()
    assert got == w.encode_compressed_proof(t, opp.params, nv.compress(opp, want_s))
    again = CompressedNovaVDFProof.deserialize(pp, got)
    assert again.serialize() == got and again.to_bytes() == snark.to_bytes()
    assert again.verify(pp, n, z0, _zi(init_ints))


@pytest.mark.parametrize("key", ["wire_ivc_t2_reference", "wire_ivc_t2"])
def test_product_bytes_equal_the_committed_vector(ctx, golden, key):
    """tests/golden/vectors.json "wire_ivc_t2_reference" / "wire_ivc_t2" (made on the CPU by the oracle alone, for the
    reference's step circuit and for the bound form): the product's bytes for that chain."""
    g = golden[key]
    t, n = g["t"], g["steps"]
    pp, z0, circuits, initial, init_ints = make(ctx, t, n, seed=g["seed"], i0=g["i0"], kind=0 if g["bound"] else 1)
    assert pp.digest() == int(g["params"], 16)
    proof = NovaVDFProof.prove_recursively(pp, circuits, t, z0)
    wire = proof.compress(pp).serialize()
    assert len(wire) == g["compressed_proof_len"] and wire[:HEADER].hex() == g["compressed_proof_head_hex"]
    assert hashlib.sha256(wire).hexdigest() == g["compressed_proof_sha256"]
    running = proof.serialize()
    assert len(running) == g["running_proof_len"] and hashlib.sha256(running).hexdigest() == g["running_proof_sha256"]


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
        offsets = {"magic": 3, "t": 8, "digest": 30, "r_U1.comm_W": 48 + 5, "r_U1.comm_E sign": 48 + 32 + 31, "r_U1.u": 48 + 64 + 1,
                   "r_U1.X1": 48 + 128 + 3, "r_U2.comm_W": 48 + 160 + 7, "r_U2.X0": 48 + 160 + 96 + 2, "l_u2.comm_W": 48 + 320 + 9,
                   "l_u2.X1": 48 + 320 + 64 + 4, "T2": 48 + 416 + 6, "zi1": 48 + 448 + 40, "zi2": 48 + 448 + 96,
                   "outer": HEADER + 33, "ipa point": len(good) - 32 * 16 - 64 + 9, "ipa a": len(good) - 1}
        for name, off in offsets.items():
            bad = bytearray(good)
            bad[off] ^= 0x80 if name.endswith("sign") else 1
            try:
                s2 = CompressedNovaVDFProof.deserialize(vpp, bytes(bad))
            except vdf_amd.VdfError:
                continue
            assert not s2.verify(vpp, n, z0, zi), name
            s2.free()
        for cut in (good[:-1], good + b"\0", good[:HEADER], good[:40], b""):
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
    identical running proofs, both verify, identical compressed proofs."""
    t, n = 16, 4
    pp, z0, circuits, initial, init_ints = make(ctx, t, n, seed=23)
    zi = _zi(init_ints)
    proof = None
    for k in range(2):
        proof = NovaVDFProof.prove_step(pp, proof, circuits, k, z0)
    blob = proof.serialize()
    s1, s2 = pp.sizes(0), pp.sizes(1)
    head = 56 + 96 + 96 + 32 + 160 + 160 + 96
    assert len(blob) == head + 32 * (s1["num_vars"] + s1["num_cons"] + 2 * s2["num_vars"] + s2["num_cons"])
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
        # a witness that does not open its commitment is refused at load
        bad = bytearray(blob)
        bad[head + 32 * 5] ^= 1
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


@pytest.mark.parametrize("kind", [1, 0], ids=["reference", "bound"])
def test_wire_formats_at_t_2_16(ctx, kind):
    """Full-size shape (the reference's circuit and the bound form): sizes and timings of both encodings; a checkpoint
    taken after two steps resumes (the restored proof proves the third step and verifies)."""
    t, n = 1 << 16, 3
    pp, z0, circuits, initial, init_ints = make(ctx, t, n, seed=5, kind=kind)
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
          f"compressed proof {len(wire)} bytes (statement {HEADER}): encode {1e3 * (t4 - t3):.2f} ms, decode {1e3 * (t5 - t4):.2f} ms")
