"""TEST INFRASTRUCTURE -- never imported by the product (vdf_amd/).

Python restatement of the wire formats of include/vdf_nova.h ("VDFSNK03" compressed proof, "VDFRSK02" running
proof) and of the 32-byte point encoding.  The reference serialises nothing (src/nova/proof.rs:52-55 keeps
proofs in memory), so there are no reference vectors for these: parity is product bytes == these bytes, plus the
round trips and the rejection cases in tests/test_wire.py and tests/test_gpu_wire.py.  Parity unpinned against the
reference, like the rest of the proof layer.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

from . import pasta as o

Point = Optional[Tuple[int, int]]
MAGIC_SNARK = b"VDFSNK03"
MAGIC_PROOF = b"VDFRSK02"


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


def encode_argument(proof, points=None) -> bytes:
    """oracle.spartan.SpartanProof; points = compress_point for the wire (32 bytes), None for the flat encoding (64)."""
    flat = lambda p: b"\0" * 64 if p is None else fe(p[0]) + fe(p[1])
    pt = points or flat
    out = b"".join(fe(v) for ev in proof.outer for v in ev)
    out += b"".join(fe(v) for v in proof.claims)
    out += b"".join(fe(v) for ev in proof.inner for v in ev)
    out += fe(proof.w_eval)
    for ipa in (proof.ipa_W, proof.ipa_E):
        out += b"".join(pt(L) + pt(R) for L, R in zip(ipa.L, ipa.R)) + b"".join(fe(v) for v in ipa.a)
    return out


def _pt(a):
    return None if tuple(a) == (0, 0) else tuple(a)


def encode_instance(inst, relaxed: bool) -> bytes:
    """oracle.nova.Relaxed / Fresh: comm_W [, comm_E, u], X."""
    out = compress_point(_pt(inst.comm_W))
    if relaxed:
        out += compress_point(_pt(inst.comm_E)) + fe(inst.u)
    return out + b"".join(fe(v) for v in inst.X)


def encode_flat_arguments(c) -> bytes:
    """vdf_nova_snark_bytes: both arguments of an oracle.nova.CompressedSNARK, 64-byte points."""
    return encode_argument(c.snark1) + encode_argument(c.snark2)


def encode_compressed_proof(t: int, params: int, c) -> bytes:
    """ "VDFSNK03" (include/vdf_nova.h) for an oracle.nova.CompressedSNARK."""
    out = MAGIC_SNARK + int(t).to_bytes(8, "little") + int(params).to_bytes(32, "little")
    out += encode_instance(c.r_U1, True) + encode_instance(c.r_U2, True) + encode_instance(c.l_u2, False)
    out += compress_point(_pt(c.T2))
    out += b"".join(fe(v) for v in c.zi[0]) + b"".join(fe(v) for v in c.zi[1])
    return out + encode_argument(c.snark1, compress_point) + encode_argument(c.snark2, compress_point)


def encode_running_proof(t: int, params: int, s, z0) -> bytes:
    """ "VDFRSK02" for an oracle.nova.RecursiveSNARK."""
    out = MAGIC_PROOF + int(t).to_bytes(8, "little") + int(s.i).to_bytes(8, "little") + int(params).to_bytes(32, "little")
    out += b"".join(fe(v) for v in z0) + b"".join(fe(v) for v in s.zi[0]) + b"".join(fe(v) for v in s.zi[1])
    out += encode_instance(s.r[0], True) + encode_instance(s.r[1], True) + encode_instance(s.l2, False)
    for vec in (s.r[0].W, s.r[0].E, s.r[1].W, s.r[1].E, s.l2.W):
        out += b"".join(fe(v) for v in vec)
    return out
