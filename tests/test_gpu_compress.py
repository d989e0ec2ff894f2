"""GPU: NovaVDFProof::compress and verification of the compressed proof (src/nova/proof.rs:360-368, :383; the
reference's own test at :446-450).  The argument produced by the product must equal, byte for byte, the one the
Python restatement (oracle/spartan.py) produces for the same folded instance and witness; at larger sizes it must
verify, and every kind of tampering must be rejected."""
import numpy as np
import pytest

from oracle import pasta as o
from oracle import spartan as sp
from util import unmont
from test_gpu_nova import make, shape_digest, aff_ints, gens
from vdf_amd.minroot import State, FIELD_FQ
from vdf_amd.nova import NovaVDFProof

pytestmark = pytest.mark.gpu
Q = o.Q


def _encode(proof: sp.SpartanProof) -> bytes:
    fe = lambda v: int(v).to_bytes(32, "little")
    pt = lambda p: b"\0" * 64 if p is None else fe(p[0]) + fe(p[1])
    out = b"".join(fe(v) for ev in proof.outer for v in ev)
    out += b"".join(fe(v) for v in proof.claims)
    out += b"".join(fe(v) for ev in proof.inner for v in ev)
    out += fe(proof.w_eval)
    for ipa in (proof.ipa_W, proof.ipa_E):
        out += b"".join(pt(L) + pt(R) for L, R in zip(ipa.L, ipa.R)) + b"".join(fe(v) for v in ipa.a)
    return out


def _zi(init_ints):
    s = State.from_ints(FIELD_FQ, *init_ints)
    return [s.x, s.y, s.i]


def _pt(a):
    return None if a == (0, 0) else a


@pytest.mark.parametrize("t,n", [(3, 3), (5, 2), (12, 2)])       # 0, 1 and 2 halving rounds before the 16-vector
def test_compressed_argument_equals_the_oracles(ctx, t, n):
    pp, z0, circuits, initial, init_ints = make(ctx, t, n, seed=31)
    proof = NovaVDFProof.prove_recursively(pp, circuits, t, z0)
    sh = o.step_circuit_shape(t, o.FIELD_FQ)
    inst = proof.instance()
    gW, gE = proof.witness()
    W, E = unmont(gW, Q), unmont(gE, Q)
    u, X = unmont(inst["u"].reshape(1, 4), Q)[0], unmont(inst["X"], Q)
    cW, cE = _pt(aff_ints(inst["comm_W"])), _pt(aff_ints(inst["comm_E"]))
    assert o.is_sat_relaxed(sh, W, E, u, X, Q)
    N = pp.sizes()["num_gens"]
    G = gens(N)
    U = gens(1, start=N)[0]
    digest = shape_digest(sh, t)
    want = sp.prove(sh, digest, G, U, cW, cE, u, X, W, E)
    assert sp.verify(sh, digest, G, U, cW, cE, u, X, want)
    snark = proof.compress(pp)
    got = snark.to_bytes()
    assert got == _encode(want)
    assert snark.verify(pp, n, z0, _zi(init_ints))


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
    sizes = pp.sizes()
    s = (sizes["num_cons"] - 1).bit_length()
    l1 = (sizes["num_vars"] - 1).bit_length() + 1
    kW, kE = (l1 - 1) - 4, s - 4                       # halving rounds: the arguments stop at 16 elements
    assert len(good) == 32 * (3 * s + 4 + 2 * l1 + 1 + 16 + 16) + 128 * (kW + kE)
    # wrong statement
    assert not snark.verify(pp, n, z0, [zi[1], zi[0], zi[2]])
    assert not snark.verify(pp, n - 1, z0, zi)
    # every section of the encoding: one flipped low bit must be rejected (or refused as non-canonical)
    head = 32 * (3 * s + 4 + 2 * l1 + 1)
    offsets = {"outer": 40, "claims": 32 * 3 * s + 33, "inner": 32 * (3 * s + 4) + 64 + 1, "w_eval": 32 * (3 * s + 4 + 2 * l1),
               "ipaW.L": head + 3, "ipaW.a[0]": head + 128 * kW, "ipaW.a[9]": head + 128 * kW + 32 * 9 + 2,
               "ipaE.R": head + 128 * kW + 32 * 16 + 64 + 5, "ipaE.a[15]": len(good) - 32}
    for name, off in offsets.items():
        bad = bytearray(good)
        bad[off] ^= 1
        try:
            snark.set_bytes(bytes(bad))
        except Exception:
            continue
        assert not snark.verify(pp, n, z0, zi), name
    snark.set_bytes(good)
    assert snark.verify(pp, n, z0, zi)
    with pytest.raises(Exception):
        snark.set_bytes(good[:-1])
    with pytest.raises(Exception):
        snark.set_bytes(b"\xff" * len(good))            # not canonical


def test_compress_full_size_t_2_16(ctx):
    """t = 2^16 (BASELINE config 3 shape: 2^18 constraints, 2^19 padded variables): completes and verifies."""
    import time
    t, n = 1 << 16, 2
    pp, z0, circuits, initial, init_ints = make(ctx, t, n, seed=5)
    proof = NovaVDFProof.prove_recursively(pp, circuits, t, z0)
    t0 = time.perf_counter()
    snark = proof.compress(pp)
    t1 = time.perf_counter()
    assert snark.verify(pp, n, z0, _zi(init_ints))
    t2 = time.perf_counter()
    print(f"compress {1e3 * (t1 - t0):.1f} ms, verify {1e3 * (t2 - t1):.1f} ms, argument {len(snark.to_bytes())} bytes")
    bad = bytearray(snark.to_bytes())
    bad[100] ^= 1
    snark.set_bytes(bytes(bad))
    assert not snark.verify(pp, n, z0, _zi(init_ints))
