"""TEST INFRASTRUCTURE -- never imported by the product (vdf_amd/).

Python restatement of the wire formats of include/vdf_nova.h ("VDFSNK02" compressed proof, "VDFRSK01" running
proof) and of the 32-byte point encoding.  The reference serialises nothing (src/nova/proof.rs:52-55 keeps
proofs in memory), so there are no reference vectors for these: parity is product bytes == these bytes, plus the
round trips and the rejection cases in tests/test_wire.py and tests/test_gpu_wire.py.  Parity unpinned against the
reference, like the rest of the proof layer.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

from . import pasta as o

Point = Optional[Tuple[int, int]]
MAGIC_SNARK = b"VDFSNK02"
MAGIC_PROOF = b"VDFRSK01"


def fe(v: int) -> bytes:
    return int(v).to_bytes(32, "little")


def compress_point(pt: Point) -> bytes:
    """x canonical little-endian, parity of y in bit 255; identity = 32 zero bytes."""
    if pt is None:
        return b"\0" * 32
    x, y = pt
    return (x | ((y & 1) << 255)).to_bytes(32, "little")


def decompress_point(data: bytes, curve: int = o.CURVE_PALLAS) -> Point:
    """Inverse of compress_point; ValueError unless `data` is exactly an encoding."""
    if len(data) != 32:
        raise ValueError("32 bytes")
    m = o.curve_base_modulus(curve)
    v = int.from_bytes(data, "little")
    odd, x = v >> 255, v & ((1 << 255) - 1)
    if x >= m:
        raise ValueError("x is not canonical")
    if x == 0:
        if odd:
            raise ValueError("the identity has one encoding")
        return None
    y = o.sqrt_mod((x * x * x + 5) % m, m)
    if y is None:
        raise ValueError("x is on no point of the curve")
    if (y & 1) != odd:
        y = m - y
    return (x, y)


def encode_chain(magic: bytes, t: int, digest: bytes, z: Sequence[Sequence[int]], comm_w: Sequence[Point],
                 comm_T: Sequence[Point]) -> bytes:
    """z: the n + 1 states z_0 .. z_n (3 integers each); comm_w: n points; comm_T: n points, [0] ignored."""
    n = len(comm_w)
    assert len(z) == n + 1 and len(comm_T) == n and len(digest) == 32 and len(magic) == 8
    out = magic + t.to_bytes(8, "little") + n.to_bytes(8, "little") + digest
    out += b"".join(fe(v) for v in z[0])
    for k in range(n):
        out += b"".join(fe(v) for v in z[k + 1]) + compress_point(comm_w[k])
        if k:
            out += compress_point(comm_T[k])
    return out


def encode_argument(proof) -> bytes:
    """oracle.spartan.SpartanProof with 32-byte points (the argument section of "VDFSNK02")."""
    out = b"".join(fe(v) for ev in proof.outer for v in ev)
    out += b"".join(fe(v) for v in proof.claims)
    out += b"".join(fe(v) for ev in proof.inner for v in ev)
    out += fe(proof.w_eval)
    for ipa in (proof.ipa_W, proof.ipa_E):
        out += b"".join(compress_point(L) + compress_point(R) for L, R in zip(ipa.L, ipa.R)) + b"".join(fe(v) for v in ipa.a)
    return out


def encode_compressed_proof(t, digest, z, comm_w, comm_T, proof) -> bytes:
    return encode_chain(MAGIC_SNARK, t, digest, z, comm_w, comm_T) + encode_argument(proof)


def encode_running_proof(t, digest, z, comm_w, comm_T, W: Sequence[int], E: Sequence[int]) -> bytes:
    return encode_chain(MAGIC_PROOF, t, digest, z, comm_w, comm_T) + b"".join(fe(v) for v in W) + b"".join(fe(v) for v in E)


def chain_size(n: int) -> int:
    return 8 + 8 + 8 + 32 + 96 + n * 128 + (n - 1) * 32
