"""Python mirror of the reference's Nova proof API (/root/reference/src/nova/proof.rs:232-392) over
libvdf_nova.so: `public_params`, `InverseMinRootCircuit.{circuits, eval_and_make_circuits}`,
`NovaVDFProof.{prove_recursively, verify, compress}`.  Every heavy step runs as HIP kernels through
the C ABI of libvdf_hip.so; see include/vdf_nova.h (Nova IVC on the Pallas / Vesta cycle, protocol "vdf-nova-ivc-v1")."""
from __future__ import annotations

import ctypes as C
import weakref
from typing import List, Sequence, Tuple

import numpy as np

from .hip import Context, VdfError
from .minroot import State, _State, _Fe, nova_lib, EvalMode, MinRootVDF, PallasVDF  # noqa: F401

_vp, _i, _u64, _sz = C.c_void_p, C.c_int, C.c_uint64, C.c_size_t
for _name, _res, _args in [
    ("vdf_nova_public_params", _i, [_vp, _u64, C.POINTER(_vp)]),
    ("vdf_nova_public_params_ex", _i, [_vp, _u64, _i, _i, C.POINTER(_vp)]),
    ("vdf_nova_public_params_flags", _i, [_vp, _u64, _i, _i, C.c_uint, C.POINTER(_vp)]),
    ("vdf_nova_public_params_tuned", _i, [_vp, _u64, _i, _i, _vp, C.POINTER(_vp)]),
    ("vdf_nova_tuning_default", None, [_vp]),
    ("vdf_nova_ro_preset", _i, [_i, _vp]),
    ("vdf_nova_public_params_ro", _i, [_vp, _u64, _i, _i, _vp, _vp, C.POINTER(_vp)]),
    ("vdf_nova_pp_ro", _i, [_vp, _vp]),
    ("vdf_nova_ro_hash_ro", _i, [_vp, _i, _u64, _vp, _sz, _vp]),
    ("vdf_nova_shape_digest_ro", _i, [_vp, _u64, _i, _i, _vp, _vp]),
    ("vdf_nova_aug_synthesize_ro", _i, [_vp, _i, _u64, _i, _vp, _vp, _vp, _vp, _sz, C.POINTER(_sz), C.POINTER(_sz), _vp, _vp]),
    ("vdf_nova_pp_tuning", _i, [_vp, _vp]),
    ("vdf_nova_pp_setup_ms", _i, [_vp, C.POINTER(C.c_double * 7)]),
    ("vdf_nova_pp_memory", _i, [_vp, _vp, _vp, _vp, C.POINTER(C.c_uint)]),
    ("vdf_nova_pp_free", None, [_vp]),
    ("vdf_nova_pp_sizes", _i, [_vp, _i] + [C.POINTER(_u64)] * 5),
    ("vdf_nova_pp_digest", _i, [_vp, _vp]),
    ("vdf_nova_pp_segment", _i, [_vp, C.POINTER(_u64), C.POINTER(_u64)]),
    ("vdf_nova_pp_early_rows", _i, [_vp, C.POINTER(_u64), C.POINTER(_u64)]),
    ("vdf_nova_pp_stencil", _i, [_vp]),
    ("vdf_nova_shape_stencil", _i, [_u64, _i, C.POINTER(_u64), C.POINTER(_u64), C.POINTER(_u64)]),
    ("vdf_nova_eval_and_make_circuits", _i, [_i, _u64, _sz, C.POINTER(_State), C.POINTER(_Fe * 3), C.POINTER(_vp)]),
    ("vdf_nova_circuits_len", _sz, [_vp]),
    ("vdf_nova_circuits_upload", _i, [_vp, _vp]),
    ("vdf_nova_circuit_states", _i, [_vp, _sz, C.POINTER(_State), C.POINTER(_State)]),
    ("vdf_nova_circuits_free", None, [_vp]),
    ("vdf_nova_prove_recursively", _i, [_vp, _vp, _u64, C.POINTER(_Fe * 3), C.POINTER(_vp)]),
    ("vdf_nova_prove_step", _i, [_vp, C.POINTER(_vp), _vp, _sz, C.POINTER(_Fe * 3)]),
    ("vdf_nova_verify", _i, [_vp, _vp, _sz, C.POINTER(_Fe * 3), C.POINTER(_Fe * 3), C.POINTER(_i)]),
    ("vdf_nova_proof_free", None, [_vp]),
    ("vdf_nova_proof_num_steps", _sz, [_vp]),
    ("vdf_nova_proof_instance", _i, [_vp, _i, _vp, _vp, _vp, _vp]),
    ("vdf_nova_proof_witness_ptrs", _i, [_vp, _i, C.POINTER(_vp), C.POINTER(_vp)]),
    ("vdf_nova_proof_zi", _i, [_vp, _vp, _vp]),
    ("vdf_nova_proof_last_step", _i, [_vp, _vp]),
    ("vdf_nova_last_step_ms", _i, [_vp, C.POINTER(C.c_double * 8)]),
    ("vdf_nova_proof_set_kernel_timing", _i, [_vp, _i]),
    ("vdf_nova_proof_kernel_events", _i, [_vp, _vp, _vp, _sz, C.POINTER(_sz)]),
    ("vdf_nova_ro_hash", _i, [_i, _u64, _vp, _sz, _vp]),
    ("vdf_nova_shape_digest", _i, [_u64, _i, _i, _vp, _vp]),
    ("vdf_nova_shape_digest_custom", _i, [_vp, _i, _vp, _vp]),
    ("vdf_nova_shape_export", _i, [_u64, _i, _i, _vp, _vp, _vp, _vp]),
    ("vdf_nova_aug_synthesize", _i, [_i, _u64, _i, _vp, _vp, _vp, _vp, _sz, C.POINTER(_sz), C.POINTER(_sz), _vp, _vp]),
    ("vdf_nova_public_params_custom", _i, [_vp, _vp, _i, C.POINTER(_vp)]),
    ("vdf_nova_prove_step_custom", _i, [_vp, C.POINTER(_vp), _vp, _vp]),
    ("vdf_nova_verify_custom", _i, [_vp, _vp, _sz, _vp, _vp, C.POINTER(_i)]),
    ("vdf_cs_is_witness", _i, [_vp]),
    ("vdf_cs_const", C.c_uint32, [_vp, _vp]),
    ("vdf_cs_add", C.c_uint32, [_vp, C.c_uint32, C.c_uint32]),
    ("vdf_cs_sub", C.c_uint32, [_vp, C.c_uint32, C.c_uint32]),
    ("vdf_cs_scale", C.c_uint32, [_vp, C.c_uint32, _vp]),
    ("vdf_cs_alloc", C.c_uint32, [_vp, _vp]),
    ("vdf_cs_mul", C.c_uint32, [_vp, C.c_uint32, C.c_uint32]),
    ("vdf_cs_enforce", _i, [_vp, C.c_uint32, C.c_uint32, C.c_uint32]),
    ("vdf_cs_value", _i, [_vp, C.c_uint32, _vp]),
    ("vdf_nova_synthesis_stats", _i, [C.POINTER(_u64), C.POINTER(_u64)]),
    ("vdf_nova_compress", _i, [_vp, _vp, C.POINTER(_vp)]),
    ("vdf_nova_verify_compressed", _i, [_vp, _vp, _sz, C.POINTER(_Fe * 3), C.POINTER(_Fe * 3), C.POINTER(_i)]),
    ("vdf_nova_snark_free", None, [_vp]),
    ("vdf_nova_snark_size", _sz, [_vp]),
    ("vdf_nova_snark_bytes", _i, [_vp, _vp, _sz]),
    ("vdf_nova_snark_set_bytes", _i, [_vp, _vp, _sz]),
    ("vdf_nova_point_compress", _i, [_i, _vp, _vp]),
    ("vdf_nova_point_decompress", _i, [_i, _vp, _vp]),
    ("vdf_nova_snark_serialized_size", _sz, [_vp]),
    ("vdf_nova_snark_serialize", _i, [_vp, _vp, _sz]),
    ("vdf_nova_snark_deserialize", _i, [_vp, _vp, _sz, C.POINTER(_vp)]),
    ("vdf_nova_proof_serialized_size", _sz, [_vp]),
    ("vdf_nova_proof_serialize", _i, [_vp, _vp, _sz]),
    ("vdf_nova_proof_deserialize", _i, [_vp, _vp, _sz, C.POINTER(_vp)]),
]:
    if not hasattr(nova_lib, _name):          # AttributeError at call time names the missing entry point
        continue
    getattr(nova_lib, _name).argtypes = _args
    getattr(nova_lib, _name).restype = _res

CIRCUIT_MINROOT_BOUND, CIRCUIT_MINROOT_REFERENCE = 0, 1
SIDE_PRIMARY, SIDE_SECONDARY = 0, 1
INST_RUNNING_PRIMARY, INST_RUNNING_SECONDARY, INST_FRESH_SECONDARY, INST_FRESH_PRIMARY_LAST = 0, 1, 2, 3
GENS_KNOWN_DLOG, GENS_TRY_AND_INCREMENT, GENS_LABEL_SHAKE = 0, 1, 2
PP_NO_DIGIT_TABLES, PP_NO_EARLY_ROWS = 1, 2
KERNEL_EVENT_DTYPE = np.dtype([("name", "S24"), ("bytes", "<f8"), ("start_ms", "<f8"), ("end_ms", "<f8")])     # vdf_kernel_event


class NovaTuning(C.Structure):
    """vdf_nova_tuning (include/vdf_nova.h): everything tunable about a parameter set and the prover over it."""
    _fields_ = [("struct_size", C.c_uint32), ("flags", C.c_uint32), ("digit_budget_bytes", C.c_uint64)] + [(k, C.c_int32) for k in (
        "digit_window", "early_rows", "stencil", "small_window", "big_window", "packed_commit", "lookahead_early", "gate_accumulate",
        "fold_on_rows", "nifs_ahead", "early_row_parts", "lookahead_priority", "side_accumulate_fill", "verbose", "compress_queues", "rows_at_challenge", "fold_fused")]

    def as_dict(self) -> dict:
        return {k: getattr(self, k) for k, _ in self._fields_}


class RoParams(C.Structure):
    """vdf_nova_ro_params (include/vdf_nova.h): the random oracle as a parameter block covered by the parameters' digest."""
    _fields_ = [("struct_size", C.c_uint32)] + [(k, C.c_int32) for k in (
        "family", "width", "full_rounds", "partial_rounds", "alpha", "challenge_bits", "hash_bits")]

    def as_dict(self) -> dict:
        return {k: getattr(self, k) for k, _ in self._fields_ if k != "struct_size"}


RO_DEFAULT, RO_NEPTUNE_SHAPED = 0, 1


def ro_preset(which: int = RO_DEFAULT, **fields) -> RoParams:
    """0: this build's default block (Poseidon2-style, width 4); 1: the neptune-shaped block (original Poseidon, width 25,
    8 + 57 rounds; [UPSTREAM-RECALL], unpinned).  Keyword fields replace members (e.g. width=9 for a smaller instance)."""
    r = RoParams()
    _check(nova_lib.vdf_nova_ro_preset(which, C.byref(r)))
    for k, v in fields.items():
        if k not in dict(RoParams._fields_):
            raise KeyError(k)
        setattr(r, k, int(v))
    return r


def _ro_ptr(ro):
    return C.byref(ro) if ro is not None else None


def tuning_default(**fields) -> NovaTuning:
    """The library's defaults (environment overrides applied once per process), with the named fields replaced."""
    t = NovaTuning()
    nova_lib.vdf_nova_tuning_default(C.byref(t))
    for k, v in fields.items():
        if k not in dict(NovaTuning._fields_):
            raise KeyError(k)
        setattr(t, k, int(v))
    return t


def _check(rc: int) -> None:
    if rc != 0:
        raise VdfError(rc, (nova_lib.vdf_nova_last_error() or b"").decode())


def point_compress(aff: np.ndarray, curve: int = 0) -> bytes:
    """32-byte encoding of a point of `curve` given as 8 Montgomery words (x, y); host arithmetic only."""
    a = np.ascontiguousarray(aff, dtype="<u8").reshape(8)
    out = (C.c_uint8 * 32)()
    _check(nova_lib.vdf_nova_point_compress(curve, a.ctypes.data, out))
    return bytes(out)


def point_decompress(data: bytes, curve: int = 0) -> np.ndarray:
    if len(data) != 32:
        raise ValueError("32 bytes")
    out = np.zeros(8, dtype="<u8")
    _check(nova_lib.vdf_nova_point_decompress(curve, (C.c_uint8 * 32).from_buffer_copy(data), out.ctypes.data))
    return out


def ro_hash(field: int, tag: int, xs: np.ndarray, ro: "RoParams | None" = None) -> np.ndarray:
    """The random oracle's sponge (host only): lane 1 after absorbing xs (Montgomery limbs in and out); `ro`: another block."""
    xs = np.ascontiguousarray(xs, dtype="<u8").reshape(-1, 4)
    out = np.zeros(4, dtype="<u8")
    _check(nova_lib.vdf_nova_ro_hash_ro(_ro_ptr(ro), field, tag, xs.ctypes.data, xs.shape[0], out.ctypes.data))
    return out


def shape_digest(t: int, circuit_kind: int = 1, gens_family: int = 1, ro: "RoParams | None" = None):
    """(digest as an integer, sizes[side] = (num_cons, num_vars, nnz)) of the parameters public_params(t) would make."""
    d = (C.c_uint8 * 32)()
    sizes = np.zeros((2, 3), dtype="<u8")
    _check(nova_lib.vdf_nova_shape_digest_ro(_ro_ptr(ro), t, circuit_kind, gens_family, d, sizes.ctypes.data))
    return int.from_bytes(bytes(d), "little"), sizes.tolist()


def shape_stencil(t: int, circuit_kind: int = 1):
    """(variables per round of the MinRoot stencil the early rows match, or 0; first early row; early rows; first round variable)
    -- host only: what public_params(t) would use for the early rows' cross term."""
    b, n, s_ = C.c_uint64(), C.c_uint64(), C.c_uint64()
    per = nova_lib.vdf_nova_shape_stencil(t, circuit_kind, C.byref(b), C.byref(n), C.byref(s_))
    if per < 0:
        _check(-per)
    return per, b.value, n.value, s_.value


def shape_export(t: int, circuit_kind: int = 1, side: int = 0):
    """[(rows uint32[nnz], cols uint32[nnz], vals uint64[nnz, 4])] x 3 (A, B, C) of the shape public_params(t) makes (host only)."""
    nnz = np.zeros(3, dtype="<u8")
    _check(nova_lib.vdf_nova_shape_export(t, circuit_kind, side, nnz.ctypes.data, None, None, None))
    mats = [(np.zeros(int(z), dtype=np.uint32), np.zeros(int(z), dtype=np.uint32), np.zeros((int(z), 4), dtype="<u8")) for z in nnz]
    arr = lambda k: (C.c_void_p * 3)(*[m[k].ctypes.data for m in mats])
    _check(nova_lib.vdf_nova_shape_export(t, circuit_kind, side, nnz.ctypes.data, arr(0), arr(1), arr(2)))
    return mats


# ---- the step-circuit seam (include/vdf_nova.h vdf_step_circuit; src/nova/proof.rs:79-153) ---------------------------
_SYNTH = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32))


class _StepCircuitC(C.Structure):
    _fields_ = [("arity", C.c_size_t), ("synthesize", _SYNTH), ("self", C.c_void_p)]


class ConstraintSystem:
    """The vdf_cs a step circuit's synthesize receives: numbers are opaque handles; field elements cross as 32-byte
    Montgomery limbs (Fq: the primary circuit's field)."""

    def __init__(self, handle):
        self.h = handle

    @property
    def is_witness(self) -> bool:
        return bool(nova_lib.vdf_cs_is_witness(self.h))

    def const(self, k: bytes) -> int:
        return nova_lib.vdf_cs_const(self.h, C.byref(_Fe.from_buffer_copy(k)))

    def add(self, a: int, b: int) -> int:
        return nova_lib.vdf_cs_add(self.h, a, b)

    def sub(self, a: int, b: int) -> int:
        return nova_lib.vdf_cs_sub(self.h, a, b)

    def scale(self, a: int, k: bytes) -> int:
        return nova_lib.vdf_cs_scale(self.h, a, C.byref(_Fe.from_buffer_copy(k)))

    def alloc(self, value: bytes = None) -> int:
        return nova_lib.vdf_cs_alloc(self.h, C.byref(_Fe.from_buffer_copy(value)) if value is not None else None)

    def mul(self, a: int, b: int) -> int:
        return nova_lib.vdf_cs_mul(self.h, a, b)

    def enforce(self, a: int, b: int, c: int) -> None:
        _check(nova_lib.vdf_cs_enforce(self.h, a, b, c))

    def value(self, a: int) -> bytes:
        out = _Fe()
        _check(nova_lib.vdf_cs_value(self.h, a, C.byref(out)))
        return bytes(out)


class StepCircuit:
    """trait StepCircuit (src/nova/proof.rs:79-153) for a circuit written by the host: subclass with `arity` and
    `synthesize(cs, z_in) -> z_out` (lists of handles).  The instance is handed to the library as a vdf_step_circuit."""
    arity = 1

    def synthesize(self, cs: ConstraintSystem, z_in):
        raise NotImplementedError

    def _c(self):
        def cb(_self, cs, zin, zout):
            try:
                out = self.synthesize(ConstraintSystem(cs), [zin[k] for k in range(self.arity)])
                for k in range(self.arity):
                    zout[k] = out[k]
                return 0
            except Exception as e:            # must not unwind through the C frames
                self._error = e
                return 1
        self._cb = _SYNTH(cb)                 # kept alive with the circuit
        self._struct = _StepCircuitC(self.arity, self._cb, None)
        return self._struct


def shape_digest_custom(circuit: StepCircuit, gens_family: int = 1):
    """Host only: (digest, sizes) of the parameters public_params_custom would make for this circuit."""
    d = (C.c_uint8 * 32)()
    sizes = np.zeros((2, 3), dtype="<u8")
    rc = nova_lib.vdf_nova_shape_digest_custom(C.byref(circuit._c()), gens_family, d, sizes.ctypes.data)
    _reraise(circuit)
    _check(rc)
    return int.from_bytes(bytes(d), "little"), sizes.tolist()


def _reraise(circuit) -> None:
    e = getattr(circuit, "_error", None)
    if e is not None:
        circuit._error = None
        raise e


def public_params_custom(ctx: Context, circuit: StepCircuit, gens_family: int = GENS_TRY_AND_INCREMENT) -> "NovaVDFPublicParams":
    h = C.c_void_p()
    rc = nova_lib.vdf_nova_public_params_custom(ctx.handle, C.byref(circuit._c()), gens_family, C.byref(h))
    _reraise(circuit)
    _check(rc)
    pp = NovaVDFPublicParams(ctx, h.value, 0)
    pp.arity = circuit.arity
    return pp


def _zn(vals: Sequence[bytes]):
    return (_Fe * len(vals))(*[_Fe.from_buffer_copy(v) for v in vals])


def synthesis_stats() -> Tuple[int, int]:
    """(slope inverses queued by the batched pre-pass, queue misses) of this thread's last augmented-circuit synthesis."""
    q, m = C.c_uint64(), C.c_uint64()
    _check(nova_lib.vdf_nova_synthesis_stats(C.byref(q), C.byref(m)))
    return q.value, m.value


class AugInputs(C.Structure):     # vdf_nova_aug_inputs
    _fields_ = [("params", _Fe), ("i", _Fe), ("z0", _Fe * 3), ("zi", _Fe * 3),
                ("U_comm_W", _Fe * 2), ("U_comm_E", _Fe * 2), ("U_u", _Fe), ("U_X", _Fe * 2),
                ("u_comm_W", _Fe * 2), ("u_X", _Fe * 2), ("T", _Fe * 2)]


def aug_synthesize(side: int, t: int, circuit_kind: int, inputs: AugInputs, result=None, inp=None, cap: int = 1 << 16, ro=None):
    """One augmented circuit synthesised on the host: (W, X, z_next, num_cons)."""
    W = np.zeros((cap, 4), dtype="<u8")
    X, zn = np.zeros((2, 4), dtype="<u8"), np.zeros((3, 4), dtype="<u8")
    nv, nc = C.c_size_t(), C.c_size_t()
    r = C.byref(result._c()) if result is not None else None
    i = C.byref(inp._c()) if inp is not None else None
    _check(nova_lib.vdf_nova_aug_synthesize_ro(_ro_ptr(ro), side, t, circuit_kind, C.addressof(inputs), C.cast(r, _vp) if r else None,
                                               C.cast(i, _vp) if i else None, W.ctypes.data, cap, C.byref(nv), C.byref(nc),
                                               X.ctypes.data, zn.ctypes.data))
    return W[:nv.value].copy(), X, zn[:3 if side == 0 else 1].copy(), nc.value


def _z(vals: Sequence[bytes]):
    return (_Fe * 3)(*[_Fe.from_buffer_copy(v) for v in vals])


class NovaVDFPublicParams:        # src/nova/proof.rs:38-43
    def __init__(self, ctx: Context, handle: int, t: int):
        self.ctx, self.handle, self.num_iters_per_step = ctx, handle, t
        self._proofs = weakref.WeakSet()         # proofs made under these parameters: freed before them
        ctx._children.add(self)

    def sizes(self, side: int = 0) -> dict:
        v = [C.c_uint64() for _ in range(5)]
        _check(nova_lib.vdf_nova_pp_sizes(self.handle, side, *[C.byref(x) for x in v]))
        return dict(zip(("num_cons", "num_vars", "num_io", "nnz", "num_gens"), [x.value for x in v]))

    def digest(self) -> int:
        d = (C.c_uint8 * 32)()
        _check(nova_lib.vdf_nova_pp_digest(self.handle, d))
        return int.from_bytes(bytes(d), "little")

    def segment(self) -> Tuple[int, int]:
        """(first variable, count) of the primary witness's run that the GPU fills: the MinRoot rounds."""
        b, n = C.c_uint64(), C.c_uint64()
        _check(nova_lib.vdf_nova_pp_segment(self.handle, C.byref(b), C.byref(n)))
        return b.value, n.value

    def early_rows(self) -> Tuple[int, int]:
        """(first constraint, count) of the primary rows whose share of T and comm_T a step makes ahead of the rest."""
        b, n = C.c_uint64(), C.c_uint64()
        _check(nova_lib.vdf_nova_pp_early_rows(self.handle, C.byref(b), C.byref(n)))
        return b.value, n.value

    def stencil(self) -> int:
        """4 / 3: the early rows run as the MinRoot stencil (reference / bound rounds), 0: through the sparse kernel."""
        return int(nova_lib.vdf_nova_pp_stencil(self.handle))

    def ro(self) -> dict:
        """The random oracle's parameter block this set was made under (vdf_nova_pp_ro)."""
        r = RoParams()
        _check(nova_lib.vdf_nova_pp_ro(self.handle, C.byref(r)))
        return r.as_dict()

    def tuning(self) -> dict:
        t = NovaTuning()
        _check(nova_lib.vdf_nova_pp_tuning(self.handle, C.byref(t)))
        return t.as_dict()

    def setup_ms(self) -> dict:
        """Wall-clock of the stages of the public_params call that made this set (vdf_nova_pp_setup_ms)."""
        ms = (C.c_double * 7)()
        _check(nova_lib.vdf_nova_pp_setup_ms(self.handle, C.byref(ms)))
        return dict(zip(("shapes_and_digest_host", "shapes_to_device", "generators", "fixed_base_tables", "digit_tables", "other", "total"),
                        [float(x) for x in ms]))

    def memory(self) -> dict:
        """HBM held per side (bytes): generators, fixed-base table, digit table; `skipped` = sides whose digit table did not fit."""
        g, t, d = (np.zeros(2, dtype="<u8") for _ in range(3))
        sk = C.c_uint()
        _check(nova_lib.vdf_nova_pp_memory(self.handle, g.ctypes.data, t.ctypes.data, d.ctypes.data, C.byref(sk)))
        return {"gens_bytes": g.tolist(), "table_bytes": t.tolist(), "digit_table_bytes": d.tolist(), "digit_tables_skipped": sk.value}

    def free(self) -> None:
        if self.handle and self.ctx.handle:      # a dead context took the device memory with it (see hip.Bases.free)
            for proof in list(self._proofs):     # a proof holds device buffers of this context and points at pp
                proof.free()
            nova_lib.vdf_nova_pp_free(self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def public_params(ctx: Context, num_iters_per_step: int, circuit_kind: int = CIRCUIT_MINROOT_REFERENCE,
                  gens_family: int = GENS_TRY_AND_INCREMENT, flags: int = 0, tuning: "NovaTuning | None" = None,
                  ro: "RoParams | None" = None, **tune) -> NovaVDFPublicParams:      # :232-237
    """`tuning` / keyword fields of vdf_nova_tuning (e.g. digit_window=12, early_rows=0): vdf_nova_public_params_tuned;
    `ro`: the random oracle's parameter block (ro_preset): vdf_nova_public_params_ro."""
    h = C.c_void_p()
    if ro is not None:
        t = tuning if tuning is not None else tuning_default()
        for k, v in tune.items():
            if k not in dict(NovaTuning._fields_):
                raise KeyError(k)
            setattr(t, k, int(v))
        t.flags |= flags
        _check(nova_lib.vdf_nova_public_params_ro(ctx.handle, num_iters_per_step, circuit_kind, gens_family, C.byref(ro), C.byref(t), C.byref(h)))
    elif tuning is not None or tune:
        t = tuning if tuning is not None else tuning_default()
        for k, v in tune.items():
            if k not in dict(NovaTuning._fields_):
                raise KeyError(k)
            setattr(t, k, int(v))
        t.flags |= flags
        _check(nova_lib.vdf_nova_public_params_tuned(ctx.handle, num_iters_per_step, circuit_kind, gens_family, C.byref(t), C.byref(h)))
    else:
        _check(nova_lib.vdf_nova_public_params_flags(ctx.handle, num_iters_per_step, circuit_kind, gens_family, flags, C.byref(h)))
    pp = NovaVDFPublicParams(ctx, h.value, num_iters_per_step)
    pp.circuit_kind = circuit_kind
    return pp


class Circuits:
    """Vec<InverseMinRootCircuit<G1>> in proving order (already reversed, :294)."""

    def __init__(self, handle: int, t: int):
        self.handle, self.t = handle, t

    def __len__(self) -> int:
        return nova_lib.vdf_nova_circuits_len(self.handle)

    def upload(self, ctx: Context) -> None:
        """Move the forward traces into HBM (an input of proving; outside the timed region)."""
        _check(nova_lib.vdf_nova_circuits_upload(ctx.handle, self.handle))
        self._ctx = ctx          # keep the context alive until the traces are freed ...
        ctx._children.add(self)  # ... and let it free them first if it is closed earlier

    def states(self, k: int) -> Tuple[State, State]:
        """(result, input) of circuit k: InverseMinRootCircuit.result / .input (:63-64)."""
        r, i = _State(), _State()
        _check(nova_lib.vdf_nova_circuit_states(self.handle, k, C.byref(r), C.byref(i)))
        return State._from_c(r), State._from_c(i)

    def free(self) -> None:
        ctx = getattr(self, "_ctx", None)
        if self.handle and (ctx is None or ctx.handle):      # uploaded traces need a live context to be released
            nova_lib.vdf_nova_circuits_free(self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class InverseMinRootCircuit:      # src/nova/proof.rs:57-66, :239-299
    arity = 3                     # :83-85

    @staticmethod
    def eval_and_make_circuits(v: MinRootVDF, num_iters_per_step: int, num_steps: int,
                               initial_state: State) -> Tuple[List[bytes], Circuits]:   # :262-299
        if num_steps <= 0:
            raise AssertionError("num_steps > 0")                                       # :268
        z0 = (_Fe * 3)()
        h = C.c_void_p()
        _check(nova_lib.vdf_nova_eval_and_make_circuits(int(v.eval_mode), num_iters_per_step, num_steps,
                                                        C.byref(initial_state._c()), C.byref(z0), C.byref(h)))
        return [bytes(z0[k]) for k in range(3)], Circuits(h.value, num_iters_per_step)


class NovaVDFProof:               # enum NovaVDFProof { Recursive, Compressed }, :51-55
    def __init__(self, handle: int, pp: NovaVDFPublicParams):
        self.handle, self.pp = handle, pp
        pp.ctx._children.add(self)
        pp._proofs.add(self)

    @staticmethod
    def prove_recursively(pp: NovaVDFPublicParams, circuits: Circuits, num_iters_per_step: int,
                          z0: Sequence[bytes]) -> "NovaVDFProof":                       # :302-358
        h = C.c_void_p()
        _check(nova_lib.vdf_nova_prove_recursively(pp.handle, circuits.handle, num_iters_per_step, C.byref(_z(z0)), C.byref(h)))
        return NovaVDFProof(h.value, pp)

    @staticmethod
    def prove_step(pp: NovaVDFPublicParams, proof: "NovaVDFProof | None", circuits: Circuits, k: int,
                   z0: Sequence[bytes]) -> "NovaVDFProof":                              # RecursiveSNARK::prove_step, :342-349
        h = C.c_void_p(proof.handle if proof is not None else None)
        _check(nova_lib.vdf_nova_prove_step(pp.handle, C.byref(h), circuits.handle, k, C.byref(_z(z0))))
        if proof is None:
            return NovaVDFProof(h.value, pp)
        return proof

    def verify(self, pp: NovaVDFPublicParams, num_steps: int, z0: Sequence[bytes], zi: Sequence[bytes]) -> bool:   # :370-387
        ok = C.c_int(0)
        _check(nova_lib.vdf_nova_verify_custom(self.handle, pp.handle, num_steps, _zn(z0), _zn(zi), C.byref(ok)))
        return bool(ok.value)

    @staticmethod
    def prove_step_custom(pp: NovaVDFPublicParams, proof: "NovaVDFProof | None", circuit: "StepCircuit",
                          z0: Sequence[bytes]) -> "NovaVDFProof":
        """prove_step for a host-written primary step circuit (public_params_custom)."""
        h = C.c_void_p(proof.handle if proof is not None else None)
        rc = nova_lib.vdf_nova_prove_step_custom(pp.handle, C.byref(h), C.byref(circuit._c()), _zn(z0))
        _reraise(circuit)
        _check(rc)
        return NovaVDFProof(h.value, pp) if proof is None else proof

    def compress(self, pp: NovaVDFPublicParams) -> "CompressedNovaVDFProof":            # :360-368
        h = C.c_void_p()
        _check(nova_lib.vdf_nova_compress(self.handle, pp.handle, C.byref(h)))
        return CompressedNovaVDFProof(h.value, pp)

    # ---- checkpoint: the running proof as bytes ("VDFRSK01", include/vdf_nova.h) ----
    def serialize(self) -> bytes:
        n = nova_lib.vdf_nova_proof_serialized_size(self.handle)
        buf = (C.c_uint8 * n)()
        _check(nova_lib.vdf_nova_proof_serialize(self.handle, buf, n))
        return bytes(buf)

    @staticmethod
    def deserialize(pp: NovaVDFPublicParams, data: bytes) -> "NovaVDFProof":
        """Rebuilds the device-resident running proof; prove_step continues from it."""
        buf = (C.c_uint8 * len(data)).from_buffer_copy(data)
        h = C.c_void_p()
        _check(nova_lib.vdf_nova_proof_deserialize(pp.handle, buf, len(data), C.byref(h)))
        return NovaVDFProof(h.value, pp)

    # ---- introspection used by the parity tests and the bench ----
    def num_steps(self) -> int:
        return nova_lib.vdf_nova_proof_num_steps(self.handle)

    def instance(self, which: int = INST_RUNNING_PRIMARY) -> dict:
        cw, ce = np.zeros(8, dtype="<u8"), np.zeros(8, dtype="<u8")
        u, X = np.zeros(4, dtype="<u8"), np.zeros((2, 4), dtype="<u8")
        _check(nova_lib.vdf_nova_proof_instance(self.handle, which, cw.ctypes.data, ce.ctypes.data, u.ctypes.data, X.ctypes.data))
        return {"comm_W": cw, "comm_E": ce, "u": u, "X": X}

    def witness(self, which: int = INST_RUNNING_PRIMARY) -> Tuple[np.ndarray, np.ndarray]:
        """Downloads (z = [W | u | X], E) of one of the three instances for parity checks (E is None for the fresh one)."""
        from ._lib import lib
        s = self.pp.sizes(0 if which in (INST_RUNNING_PRIMARY, INST_FRESH_PRIMARY_LAST) else 1)
        dz, dE = C.c_void_p(), C.c_void_p()
        _check(nova_lib.vdf_nova_proof_witness_ptrs(self.handle, which, C.byref(dz), C.byref(dE)))
        z = np.zeros((s["num_vars"] + 3, 4), dtype="<u8")
        self.pp.ctx._check(lib.vdf_dev_memcpy(self.pp.ctx.handle, z.ctypes.data, dz.value, z.nbytes))
        E = None
        if dE.value:
            E = np.zeros((s["num_cons"], 4), dtype="<u8")
            self.pp.ctx._check(lib.vdf_dev_memcpy(self.pp.ctx.handle, E.ctypes.data, dE.value, E.nbytes))
        return z, E

    def zi(self) -> Tuple[np.ndarray, np.ndarray]:
        a, b = np.zeros((getattr(self.pp, "arity", 3), 4), dtype="<u8"), np.zeros((1, 4), dtype="<u8")
        _check(nova_lib.vdf_nova_proof_zi(self.handle, a.ctypes.data, b.ctypes.data))
        return a, b

    def last_step(self) -> dict:
        """By-products of the last prove_step: fresh primary instance, cross-term commitments, fold challenges."""
        raw = np.zeros(8 + 8 + 8 + 8 + 4 + 4, dtype="<u8")
        _check(nova_lib.vdf_nova_proof_last_step(self.handle, raw.ctypes.data))
        return {"comm_W1": raw[0:8], "X1": raw[8:16].reshape(2, 4), "comm_T1": raw[16:24], "comm_T2": raw[24:32],
                "r1": int.from_bytes(raw[32:36].tobytes(), "little"), "r2": int.from_bytes(raw[36:40].tobytes(), "little")}

    def last_step_ms(self) -> dict:
        ms = (C.c_double * 8)()
        _check(nova_lib.vdf_nova_last_step_ms(self.handle, C.byref(ms)))
        return dict(zip(("secondary_nifs", "primary_synthesis", "primary_launch", "primary_wait", "secondary_synthesis", "secondary_launch", "lookahead", "total"), list(ms)))

    def set_kernel_timing(self, flag: bool) -> None:
        """HIP events around every launch of this prover's three queues (include/vdf_nova.h); measure rates with it off."""
        _check(nova_lib.vdf_nova_proof_set_kernel_timing(self.handle, int(flag)))

    def kernel_events(self) -> list:
        """Drains the timed launches: [(queue, kernel, algorithmic bytes, start_ms, end_ms)] on the device's common time line."""
        n = C.c_size_t()
        _check(nova_lib.vdf_nova_proof_kernel_events(self.handle, None, None, 0, C.byref(n)))
        cap = n.value + 64
        ev = np.zeros(cap, dtype=KERNEL_EVENT_DTYPE)
        q = np.zeros(cap, dtype=np.int32)
        _check(nova_lib.vdf_nova_proof_kernel_events(self.handle, ev.ctypes.data, q.ctypes.data, cap, C.byref(n)))
        return [(int(q[i]), ev["name"][i].decode(), float(ev["bytes"][i]), float(ev["start_ms"][i]), float(ev["end_ms"][i]))
                for i in range(n.value)]

    def free(self) -> None:
        if self.handle and self.pp.handle and self.pp.ctx.handle:      # needs live parameters and a live context
            nova_lib.vdf_nova_proof_free(self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class CompressedNovaVDFProof:     # NovaVDFProof::Compressed, src/nova/proof.rs:54
    def __init__(self, handle: int, pp: NovaVDFPublicParams):
        self.handle, self.pp = handle, pp
        pp.ctx._children.add(self)
        pp._proofs.add(self)

    def verify(self, pp: NovaVDFPublicParams, num_steps: int, z0: Sequence[bytes], zi: Sequence[bytes]) -> bool:   # :370-387
        ok = C.c_int(0)
        _check(nova_lib.vdf_nova_verify_compressed(self.handle, pp.handle, num_steps, C.cast(_zn(z0), C.POINTER(_Fe * 3)),
                                                   C.cast(_zn(zi), C.POINTER(_Fe * 3)), C.byref(ok)))
        return bool(ok.value)

    def to_bytes(self) -> bytes:
        """The argument in its flat canonical encoding (include/vdf_nova.h)."""
        n = nova_lib.vdf_nova_snark_size(self.handle)
        buf = (C.c_uint8 * n)()
        _check(nova_lib.vdf_nova_snark_bytes(self.handle, buf, n))
        return bytes(buf)

    def set_bytes(self, data: bytes) -> None:
        buf = (C.c_uint8 * len(data)).from_buffer_copy(data)
        _check(nova_lib.vdf_nova_snark_set_bytes(self.handle, buf, len(data)))

    def serialize(self) -> bytes:
        """The whole compressed proof (step chain + argument, 32-byte points): "VDFSNK01", include/vdf_nova.h."""
        n = nova_lib.vdf_nova_snark_serialized_size(self.handle)
        buf = (C.c_uint8 * n)()
        _check(nova_lib.vdf_nova_snark_serialize(self.handle, buf, n))
        return bytes(buf)

    @staticmethod
    def deserialize(pp: NovaVDFPublicParams, data: bytes) -> "CompressedNovaVDFProof":
        buf = (C.c_uint8 * len(data)).from_buffer_copy(data)
        h = C.c_void_p()
        _check(nova_lib.vdf_nova_snark_deserialize(pp.handle, buf, len(data), C.byref(h)))
        return CompressedNovaVDFProof(h.value, pp)

    def free(self) -> None:
        if self.handle and self.pp.handle and self.pp.ctx.handle:
            nova_lib.vdf_nova_snark_free(self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
