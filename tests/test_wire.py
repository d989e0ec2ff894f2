"""CPU: the 32-byte point encoding of the wire formats (include/vdf_nova.h) -- the product's host code
(libvdf_nova.so, no device involved) against the Python restatement oracle/wire.py, plus the rejection cases.
The reference serialises nothing (src/nova/proof.rs:52-55), so the expectations are the restatement's."""
import random

import numpy as np
import pytest

from oracle import pasta as o
from oracle import wire as w
from util import affine_array, unmont
from vdf_amd.nova import point_compress, point_decompress

P = o.P


def _points(n, seed):
    pts = o.tai_bases(o.CURVE_PALLAS, seed, n // 2) + o.synthetic_bases(o.CURVE_PALLAS, seed, n - n // 2)
    return pts + [o.pt_neg(p, P) for p in pts[:4]] + [None]


def test_oracle_point_codec_round_trips():
    for p in _points(12, seed=3):
        enc = w.compress_point(p)
        assert len(enc) == 32 and w.decompress_point(enc) == p
    assert w.compress_point(None) == b"\0" * 32


def test_five_is_not_a_square_so_x_zero_is_free_for_the_identity():
    assert o.sqrt_mod(5, o.P) is None and o.sqrt_mod(5, o.Q) is None


def test_product_compress_equals_the_oracles():
    pts = _points(24, seed=11)
    arr = affine_array(pts, o.CURVE_PALLAS)
    for p, a in zip(pts, arr):
        assert point_compress(a) == w.compress_point(p)


def test_product_decompress_equals_the_oracles():
    pts = _points(24, seed=5)
    for p in pts:
        got = point_decompress(w.compress_point(p))
        x, y = unmont(got.reshape(2, 4), P)
        assert ((x, y) == (0, 0) and p is None) or (x, y) == p


def test_decompress_rejects_everything_that_is_not_an_encoding():
    rng = random.Random(9)
    bad = []
    bad.append((P).to_bytes(32, "little"))                             # x = p: not canonical
    bad.append(((1 << 255) - 1).to_bytes(32, "little"))                # x far above p
    bad.append((1 << 255).to_bytes(32, "little"))                      # identity with the sign bit set
    while len(bad) < 8:                                                # x with x^3 + 5 a non-residue
        x = rng.randrange(1, P)
        if o.sqrt_mod((x ** 3 + 5) % P, P) is None:
            bad.append(x.to_bytes(32, "little"))
    for enc in bad:
        with pytest.raises(ValueError):
            w.decompress_point(enc)
        with pytest.raises(Exception):
            point_decompress(enc)
    with pytest.raises(ValueError):
        point_decompress(b"\0" * 31)


def test_vesta_points_use_the_other_coordinate_field():
    """Commitments of the secondary side are Vesta points (coordinates in Fq)."""
    pts = o.tai_bases(o.CURVE_VESTA, 4, 6) + [None]
    arr = affine_array(pts, o.CURVE_VESTA)
    for p, a in zip(pts, arr):
        enc = point_compress(a, o.CURVE_VESTA)
        assert enc == w.compress_point(p) and w.decompress_point(enc, o.CURVE_VESTA) == p
        got = point_decompress(enc, o.CURVE_VESTA)
        x, y = unmont(got.reshape(2, 4), o.Q)
        assert ((x, y) == (0, 0) and p is None) or (x, y) == p


def test_both_roots_are_reachable():
    """The sign bit selects the root: flipping it negates the point."""
    p = o.tai_bases(o.CURVE_PALLAS, 2, 1)[0]
    enc = bytearray(w.compress_point(p))
    enc[31] ^= 0x80
    got = point_decompress(bytes(enc))
    assert tuple(unmont(got.reshape(2, 4), P)) == o.pt_neg(p, P)
