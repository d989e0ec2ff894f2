"""ctypes loader of oracle/build/libpasta_ref.so (the plain-C CPU restatement).
TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never from vdf_amd/."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(_HERE, "build", "libpasta_ref.so")


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "pasta_ref.c")
    if force or not os.path.exists(SO) or os.path.getmtime(SO) < os.path.getmtime(src):
        # -march=native is resolved on the machine that builds; rebuild on the GPU box if the ISA differs
        subprocess.check_call(["make", "-s", "-C", _HERE, "all"])
    return SO


def load() -> C.CDLL:
    build()
    try:
        return C.CDLL(SO)
    except OSError:
        build(force=True)
        return C.CDLL(SO)


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        _lib = load()
        vp, sz, i, u64 = C.c_void_p, C.c_size_t, C.c_int, C.c_uint64
        sig = {
            "ref_fe_mul": [i, vp, vp, sz, vp], "ref_fe_to_mont": [i, vp, sz, vp], "ref_fe_from_mont": [i, vp, sz, vp],
            "ref_axpy": [i, vp, vp, vp, sz, vp],
            "ref_cross_term": [i, vp, vp, vp, vp, vp, vp, vp, sz, vp],
            "ref_spmv": [i, vp, vp, vp, sz, vp, sz, vp],
            "ref_forward_step": [i, i, vp, vp], "ref_forward_step_pow": [i, vp, vp],
            "ref_minroot_eval": [i, i, vp, u64, vp, vp], "ref_minroot_inverse_eval": [i, vp, u64, vp],
            "ref_step_witness": [i, vp, u64, vp],
            "ref_jac_to_affine": [i, vp, vp], "ref_synthetic_bases": [i, u64, sz, sz, vp], "ref_tai_bases": [i, u64, sz, sz, vp],
            "ref_msm": [i, vp, vp, sz, i, i, i, vp], "ref_msm_naive": [i, vp, vp, sz, i, vp],
        }
        for name, args in sig.items():
            fn = getattr(_lib, name)
            fn.argtypes = args
            fn.restype = None
        _lib.ref_on_curve.argtypes = [i, vp]; _lib.ref_on_curve.restype = i
        _lib.ref_count_off_curve.argtypes = [i, vp, sz]; _lib.ref_count_off_curve.restype = sz
    return _lib


def p(a: np.ndarray) -> int:
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data


def fe_array(n: int) -> np.ndarray:
    return np.zeros((n, 4), dtype="<u8")
