"""Shared helpers for the tests: data reshaping between Python ints and limb arrays, and the
affine normalisation used as the canonical form of a point."""
import numpy as np

from oracle import pasta as o


def limbs(vals):
    return np.frombuffer(b"".join(int(v).to_bytes(32, "little") for v in vals), dtype="<u8").reshape(-1, 4).copy()


def ints(arr):
    raw = np.ascontiguousarray(arr).tobytes()
    return [int.from_bytes(raw[32 * i:32 * i + 32], "little") for i in range(len(raw) // 32)]


def mont(vals, m):
    return limbs([o.to_mont(v, m) for v in vals])


def unmont(arr, m):
    return [o.from_mont(v, m) for v in ints(arr)]


def jac_to_affine(jac_arr, curve):
    """uint64[12] Jacobian (Montgomery) -> canonical affine ints, identity = None."""
    m = o.curve_base_modulus(curve)
    X, Y, Z = unmont(np.asarray(jac_arr).reshape(3, 4), m)
    if Z == 0:
        return None
    zi = pow(Z, -1, m)
    return (X * zi * zi % m, Y * zi * zi * zi % m)


def affine_array(points, curve):
    """list of oracle points (None = identity) -> uint64[n, 8] Montgomery affine array."""
    m = o.curve_base_modulus(curve)
    flat = []
    for p in points:
        x, y = (0, 0) if p is None else p
        flat += [o.to_mont(x, m), o.to_mont(y, m)]
    return limbs(flat).reshape(-1, 8)


def rand_limbs(rng, n, top_mask=0x3FFFFFFFFFFFFFFF):
    """n uniformly random 254-bit values as uint64[n, 4] (always < p, q)."""
    a = rng.integers(0, 2**64, size=(n, 4), dtype=np.uint64)
    a[:, 3] &= np.uint64(top_mask)
    return a


def hexes(lst):
    return [int(h, 16) for h in lst]
