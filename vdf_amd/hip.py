"""Thin Python handle over the C ABI (include/vdf_hip.h).  Plumbing only: every arithmetic
operation is a HIP kernel behind `libvdf_hip.so`.  Buffers may be numpy arrays (host, staged by
the library over PCIe) or torch CUDA tensors (device, used in place)."""
from __future__ import annotations

import ctypes as C
import weakref
from typing import Optional, Sequence

import numpy as np

from . import _lib
from ._lib import lib, CURVE_PALLAS, CURVE_VESTA, FIELD_FP, FIELD_FQ  # noqa: F401


class HipTuning(C.Structure):
    """vdf_hip_tuning (include/vdf_hip.h): process-wide tuning of the kernels."""
    _fields_ = [("struct_size", C.c_uint32)] + [(k, C.c_int32) for k in (
        "msm_direct", "direct_priority", "direct_fused", "light_priority", "accumulate_fill", "accumulate_lds", "slice_len", "part_bits",
        "reduction", "reduction_quads", "heavy_min", "giant_span", "nifs_lanes", "shim_cache", "nifs_fused", "fold_u128", "fixup_serial", "sort_staged", "glv")]


def tuning_get() -> HipTuning:
    t = HipTuning()
    rc = lib.vdf_hip_tuning_get(C.byref(t))
    if rc != 0:
        raise VdfError(rc, "vdf_hip_tuning_get")
    return t


def tuning_set(**fields) -> HipTuning:
    """Changes the named fields of the process-wide tuning; returns the values now in force."""
    t = tuning_get()
    for k, v in fields.items():
        if k not in dict(HipTuning._fields_):
            raise KeyError(k)
        setattr(t, k, int(v))
    rc = lib.vdf_hip_tuning_set(C.byref(t))
    if rc != 0:
        raise VdfError(rc, "vdf_hip_tuning_set: a field is out of range")
    return t


class VdfError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"vdf_hip error {code}: {msg}")
        self.code = code


def ints_to_limbs(vals: Sequence[int]) -> np.ndarray:
    """Python ints -> uint64[n, 4] little-endian limbs (data reshaping, no arithmetic)."""
    buf = b"".join(int(v).to_bytes(32, "little") for v in vals)
    return np.frombuffer(buf, dtype="<u8").reshape(-1, 4).copy()


def limbs_to_ints(arr: np.ndarray) -> list:
    a = np.ascontiguousarray(arr, dtype="<u8").reshape(-1, 4)
    raw = a.tobytes()
    return [int.from_bytes(raw[32 * i:32 * i + 32], "little") for i in range(a.shape[0])]


def _ptr(x) -> Optional[int]:
    """Address of a numpy array or torch tensor (None stays None)."""
    if x is None:
        return None
    if isinstance(x, np.ndarray):
        if not x.flags["C_CONTIGUOUS"]:
            raise ValueError("numpy buffers must be C-contiguous")
        return x.ctypes.data
    if hasattr(x, "data_ptr"):
        if not x.is_contiguous():
            raise ValueError("tensors must be contiguous")
        return x.data_ptr()
    if isinstance(x, int):
        return x
    raise TypeError(type(x))


class Bases:
    def __init__(self, ctx: "Context", handle: int, curve: int):
        self.ctx, self.handle, self.curve = ctx, handle, curve
        ctx._children.add(self)

    def __len__(self) -> int:
        return lib.vdf_bases_len(self.handle)

    def validate(self) -> None:
        """Raises VdfError (NONCANONICAL / BAD_ARG, with the first offending index) unless every point is valid."""
        bad = C.c_size_t()
        rc = lib.vdf_bases_validate(self.ctx.handle, self.handle, C.byref(bad))
        if rc != 0:
            raise VdfError(rc, (lib.vdf_last_error(self.ctx.handle) or b"").decode() + f" (index {bad.value})")

    @property
    def device_ptr(self) -> int:
        return lib.vdf_bases_device_ptr(self.handle)

    def precompute(self, window_bits: int = 16, sets: int = 1) -> None:
        self.ctx._check(lib.vdf_bases_precompute(self.ctx.handle, self.handle, window_bits, sets))

    @property
    def window(self) -> int:
        """Window of the fixed-base table (0 without one)."""
        return lib.vdf_bases_window(self.handle)

    def precompute_digits(self, ranges, window_bits: int = 0) -> None:
        """Digit table over the generator ranges [(begin, count), ...] (at most 4): MSMs inside them of up to 2^17 scalars
        become a plain sum of gathered multiples (vdf_hip.h vdf_bases_precompute_digits).  [] drops the table."""
        k = len(ranges)
        b = (C.c_size_t * max(k, 1))(*[int(r[0]) for r in ranges])
        n = (C.c_size_t * max(k, 1))(*[int(r[1]) for r in ranges])
        self.ctx._check(lib.vdf_bases_precompute_digits(self.ctx.handle, self.handle, window_bits, k, b, n))

    @property
    def digit_window(self) -> int:
        return lib.vdf_bases_digit_window(self.handle)

    def download(self, offset: int = 0, n: Optional[int] = None) -> np.ndarray:
        n = len(self) - offset if n is None else n
        out = np.zeros((n, 8), dtype="<u8")
        self.ctx._check(lib.vdf_bases_download(self.ctx.handle, self.handle, offset, n, _ptr(out)))
        return out

    def free(self) -> None:
        # if the context is already gone (finalisers run in no particular order at interpreter shutdown) the
        # device memory went with it: calling into the library with a dead context would be a use after free
        if self.handle and self.ctx.handle:
            lib.vdf_bases_free(self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


_ALLGATHER = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)     # vdf_allgather_fn


def msm_multi(ctxs, bases, scalars, n, offsets=None, is_mont: bool = False):
    """vdf_msm_multi: one process, one Context / generator shard / scalar slice per device; returns uint64[12]."""
    k = len(ctxs)
    out = np.zeros(12, dtype="<u8")
    offs = (C.c_size_t * k)(*(offsets or [0] * k))
    rc = lib.vdf_msm_multi((C.c_void_p * k)(*[c.handle for c in ctxs]), (C.c_void_p * k)(*[b.handle for b in bases]), offs,
                           (C.c_void_p * k)(*[_ptr(x) for x in scalars]), (C.c_size_t * k)(*[int(x) for x in n]), k,
                           int(is_mont), out.ctypes.data)
    ctxs[0]._check(rc)
    return out


class Shape:
    def __init__(self, ctx: "Context", handle: int, field: int, num_cons: int, num_cols: int):
        self.ctx, self.handle, self.field, self.num_cons, self.num_cols = ctx, handle, field, num_cons, num_cols
        ctx._children.add(self)

    def free(self) -> None:
        if self.handle and self.ctx.handle:      # see Bases.free
            lib.vdf_shape_free(self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class MsmJob:
    """Handle of an open MSM job; finish() ends it in every case."""

    def __init__(self, ctx: "Context", handle: int, k: int):
        self.ctx, self.handle, self.k = ctx, handle, k

    def push(self, g: int, scalars) -> None:
        self.ctx._check(lib.vdf_msm_job_push(self.handle, g, _ptr(scalars)))

    def finish(self, out=None):
        if out is None:
            out = np.zeros((self.k, 12), dtype="<u8")
        h, self.handle = self.handle, None
        self.ctx._check(lib.vdf_msm_job_finish(h, _ptr(out)))
        return out


QUEUE_CRITICAL, QUEUE_SIDE = 1, 2


class Context:
    """One context per GPU (one process per GPU)."""

    def __init__(self, device: int = 0, pooled_role: int = 0):
        """pooled_role: 0 = a context with a stream of its own (vdf_ctx_create); QUEUE_CRITICAL / QUEUE_SIDE = a stream from the
        device's pool of hardware queues, shared once the budget is spent (vdf_ctx_create_pooled)."""
        h = C.c_void_p()
        dev = C.c_int(device)
        rc = (lib.vdf_ctx_create_pooled(C.byref(dev), 1, pooled_role, C.byref(h)) if pooled_role
              else lib.vdf_ctx_create(C.byref(dev), 1, C.byref(h)))
        if rc != _lib.VDF_OK:
            raise VdfError(rc, (lib.vdf_last_error(None) or b"").decode())
        self.handle = h.value
        self.device = device
        self._children = weakref.WeakSet()      # handles that own device memory of this context

    def _check(self, rc: int) -> None:
        if rc != _lib.VDF_OK:
            raise VdfError(rc, (lib.vdf_last_error(self.handle) or b"").decode())

    def __enter__(self) -> "Context":
        return self

    def __exit__(self, *exc) -> None:
        self.close()

    def close(self) -> None:
        if self.handle:
            for child in list(self._children):   # bases / shapes must go before their context
                child.free()
            lib.vdf_ctx_destroy(self.handle)
            self.handle = None

    def queue_info(self) -> dict:
        """{"pooled", "sharers", "device_streams"}: vdf_ctx_queue_info."""
        a, b, c = C.c_int(), C.c_int(), C.c_int()
        self._check(lib.vdf_ctx_queue_info(self.handle, C.byref(a), C.byref(b), C.byref(c)))
        return {"pooled": bool(a.value), "sharers": b.value, "device_streams": c.value}

    def set_stream(self, stream_ptr) -> None:
        """A foreign hipStream_t (e.g. a torch stream's .cuda_stream); None / 0 = back to the context's own."""
        self._check(lib.vdf_ctx_set_stream(self.handle, stream_ptr or None))

    @property
    def stream(self) -> int:
        return lib.vdf_ctx_get_stream(self.handle)

    def set_async(self, flag: bool) -> None:
        self._check(lib.vdf_ctx_set_async(self.handle, int(flag)))

    def get_async(self) -> bool:
        v = C.c_int()
        self._check(lib.vdf_ctx_get_async(self.handle, C.byref(v)))
        return bool(v.value)

    def sync(self) -> None:
        self._check(lib.vdf_ctx_sync(self.handle))

    def wait(self, other: "Context") -> None:
        """Work enqueued on this context from now on starts after everything enqueued on `other` so far."""
        self._check(lib.vdf_ctx_wait(self.handle, other.handle))

    def mark(self, slot: int = 0) -> None:
        """Remembers the current end of this context's queue under `slot` (0..3)."""
        self._check(lib.vdf_ctx_mark(self.handle, slot))

    def sync_mark(self, slot: int = 0) -> None:
        """Waits for everything enqueued before mark `slot`; later work keeps running."""
        self._check(lib.vdf_ctx_sync_mark(self.handle, slot))

    def wait_mark(self, other: "Context", slot: int = 0) -> None:
        """Work enqueued on this context from now on starts after `other` has reached its mark `slot` (not after what
        `other` was given since)."""
        self._check(lib.vdf_ctx_wait_mark(self.handle, other.handle, slot))

    def set_accumulate_fill(self, workgroups_per_cu: int) -> None:
        """Resident accumulation workgroups per CU this context's bucket-method MSMs fill (1..3; 0 = the process-wide value)."""
        self._check(lib.vdf_ctx_set_accumulate_fill(self.handle, workgroups_per_cu))

    def set_msm_window(self, c: int) -> None:
        self._check(lib.vdf_ctx_set_msm_window(self.handle, c))

    # ---- bases -------------------------------------------------------------------------
    def bases_upload(self, curve: int, pts) -> Bases:
        n = pts.shape[0]
        h = C.c_void_p()
        self._check(lib.vdf_bases_upload(self.handle, curve, _ptr(pts), n, C.byref(h)))
        return Bases(self, h.value, curve)

    def bases_generate(self, curve: int, seed: int, n: int, start: int = 0, family: int = 0) -> Bases:
        """family 0: P_i = [k_i]G (known discrete logs); 1: try-and-increment (unknown discrete logs)."""
        h = C.c_void_p()
        self._check(lib.vdf_bases_generate_family(self.handle, curve, family, seed, start, n, C.byref(h)))
        return Bases(self, h.value, curve)

    def bases_generate_label(self, curve: int, label: bytes, n: int, start: int = 0) -> Bases:
        """Generators derived from a label (SHAKE256 -> curve points): include/vdf_hip.h vdf_bases_generate_label."""
        h = C.c_void_p()
        buf = (C.c_uint8 * max(len(label), 1)).from_buffer_copy(label or b"\0")
        self._check(lib.vdf_bases_generate_label(self.handle, curve, buf, len(label), start, n, C.byref(h)))
        return Bases(self, h.value, curve)

    # ---- msm ---------------------------------------------------------------------------
    def msm(self, bases: Bases, scalars, n: Optional[int] = None, offset: int = 0, is_mont: bool = False, out=None):
        """Returns the Jacobian result as uint64[12] (numpy) unless `out` (device tensor) is given."""
        n = scalars.shape[0] if n is None else n
        host_out = out is None
        if host_out:
            out = np.zeros(12, dtype="<u8")
        self._check(lib.vdf_msm(self.handle, bases.handle, offset, _ptr(scalars), n, int(is_mont), _ptr(out)))
        return out

    def msm_batch(self, bases: Bases, scalars, n=None, offsets=None, is_mont: bool = False, out=None):
        """k <= 4 MSMs over the same generators in one pipeline; returns uint64[k, 12] unless `out` is given."""
        k = len(scalars)
        n = [s.shape[0] for s in scalars] if n is None else list(n)
        offsets = [0] * k if offsets is None else list(offsets)
        if out is None:
            out = np.zeros((k, 12), dtype="<u8")
        sc = (C.c_void_p * k)(*[_ptr(x) for x in scalars])
        self._check(lib.vdf_msm_batch(self.handle, bases.handle, k, (C.c_size_t * k)(*offsets), sc, (C.c_size_t * k)(*n),
                                      int(is_mont), _ptr(out)))
        return out

    def msm_job(self, bases: Bases, n, offsets=None, is_mont: bool = False) -> "MsmJob":
        """Start an MSM job over k = len(n) vectors (see include/vdf_hip.h): push them one at a time, then finish."""
        k = len(n)
        offsets = [0] * k if offsets is None else list(offsets)
        h = C.c_void_p()
        self._check(lib.vdf_msm_job_begin(self.handle, bases.handle, k, (C.c_size_t * k)(*offsets), (C.c_size_t * k)(*n),
                                          int(is_mont), C.byref(h)))
        return MsmJob(self, h.value, k)

    def msm_sharded(self, bases: Bases, scalars, n: int, rank: int, world: int, all_gather, partial, gathered, out,
                    is_mont: bool = False, always_gather: bool = False, offset: int = 0):
        """vdf_msm_sharded: this rank's partial, the host's all-gather of the world partials, the local point sum.
        all_gather(dst, src) is the collective on the caller's buffers `gathered` (world x 12 words) and `partial` (12),
        both device memory; it is called once, from inside the library, ordered after the partial.  The header's contract
        is that the collective runs ON the stream the library hands over (the one the partial was produced on and the
        point sum will run on): when the buffers are torch tensors the callback therefore makes that stream torch's
        current one for the duration of the call, whatever stream the caller happens to be on."""
        err = []

        def cb(_user, _send, _recv, _bytes, _stream):
            try:
                if hasattr(partial, "data_ptr"):
                    import torch
                    with torch.cuda.stream(torch.cuda.ExternalStream(int(_stream or 0), device=partial.device)):
                        all_gather(gathered, partial)
                else:
                    all_gather(gathered, partial)
                return 0
            except Exception as e:            # an exception must not unwind through the C frames
                err.append(e)
                return 1
        fn = _ALLGATHER(cb)
        rc = lib.vdf_msm_sharded(self.handle, bases.handle, offset, _ptr(scalars), n, int(is_mont), rank, world, fn, None,
                                 1 if always_gather else 0, _ptr(partial), _ptr(gathered), _ptr(out))
        if err:
            raise err[0]
        self._check(rc)
        return out

    def point_sum(self, curve: int, points, n: int, out=None):
        host_out = out is None
        if host_out:
            out = np.zeros(12, dtype="<u8")
        self._check(lib.vdf_point_sum(self.handle, curve, _ptr(points), n, _ptr(out)))
        return out

    def set_timing(self, flag: bool) -> None:
        self._check(lib.vdf_ctx_set_timing(self.handle, int(flag)))

    def msm_timing(self):
        """(sort_ms, accumulate_ms, tail_ms, total_ms, calls) summed since the last query."""
        ms = (C.c_float * 4)()
        calls = C.c_int()
        self._check(lib.vdf_msm_timing(self.handle, ms, C.byref(calls)))
        return (*[float(x) for x in ms], calls.value)

    # ---- shape / spmv --------------------------------------------------------------------
    def shape_create(self, field: int, num_cons: int, num_cols: int, mats) -> Shape:
        """mats: three (rows uint32[nnz], cols uint32[nnz], vals uint64[nnz,4]) triples (host)."""
        rows = (C.c_void_p * 3)(*[_ptr(m[0]) for m in mats])
        cols = (C.c_void_p * 3)(*[_ptr(m[1]) for m in mats])
        vals = (C.c_void_p * 3)(*[_ptr(m[2]) for m in mats])
        nnz = (C.c_size_t * 3)(*[int(m[0].shape[0]) for m in mats])
        h = C.c_void_p()
        self._check(lib.vdf_shape_create(self.handle, field, num_cons, num_cols, rows, cols, vals, nnz, C.byref(h)))
        self._keep = mats
        return Shape(self, h.value, field, num_cons, num_cols)

    def spmv3(self, shape: Shape, z, az, bz, cz) -> None:
        self._check(lib.vdf_spmv3(self.handle, shape.handle, _ptr(z), _ptr(az), _ptr(bz), _ptr(cz)))

    # ---- vector ops ------------------------------------------------------------------------
    def cross_term(self, field, az1, bz1, cz1, az2, bz2, cz2, u1, n, out) -> None:
        self._check(lib.vdf_cross_term(self.handle, field, _ptr(az1), _ptr(bz1), _ptr(cz1), _ptr(az2), _ptr(bz2),
                                       _ptr(cz2), _ptr(u1), n, _ptr(out)))

    def axpy(self, field, a, r, b, n, out) -> None:
        self._check(lib.vdf_axpy(self.handle, field, _ptr(a), _ptr(r), _ptr(b), n, _ptr(out)))

    def minroot_witness(self, field, trace_xy, i0, t, out) -> None:
        self._check(lib.vdf_minroot_witness(self.handle, field, _ptr(trace_xy), _ptr(i0), t, _ptr(out)))

    # ---- fused step operations: scalar operands are HOST arrays, vectors are device buffers ----
    def minroot_step_z(self, field, trace_xy, t, z_in, i0, u, X, z) -> None:
        self._check(lib.vdf_minroot_step_z(self.handle, field, _ptr(trace_xy), t, _ptr(z_in), _ptr(i0), _ptr(u), _ptr(X),
                                           _ptr(z)))

    def minroot_step_z_packed(self, field, trace_xy, t, z_in, i0, u, X, z, w_packed) -> None:
        self._check(lib.vdf_minroot_step_z_packed(self.handle, field, _ptr(trace_xy), t, _ptr(z_in), _ptr(i0), _ptr(u),
                                                  _ptr(X), _ptr(z), _ptr(w_packed)))

    def minroot_step_segment(self, field, trace_xy, t, i0, vars_per_round, out) -> None:
        self._check(lib.vdf_minroot_step_segment(self.handle, field, _ptr(trace_xy), t, _ptr(i0), vars_per_round, _ptr(out)))

    def minroot_step_segment_packed(self, field, trace_xy, t, i0, i_in, out, packed) -> None:
        """The reference's allocation (4 variables per round) and the 3t + 4 scalars of its commitment without new_x."""
        self._check(lib.vdf_minroot_step_segment_packed(self.handle, field, _ptr(trace_xy), t, _ptr(i0), _ptr(i_in), _ptr(out), _ptr(packed)))

    def gate_accumulate(self, other: "Context", slot: int) -> None:
        """The next bucket-method MSM on this context accumulates only after `other`'s mark `slot` (vdf_ctx_gate_accumulate)."""
        self._check(lib.vdf_ctx_gate_accumulate(self.handle, other.handle, slot))

    def set_kernel_timing(self, flag: bool) -> None:
        self._check(lib.vdf_ctx_set_kernel_timing(self.handle, int(flag)))

    def kernel_events(self) -> list:
        """Drains the timed launches of this context: [(kernel, algorithmic bytes, start_ms, end_ms)]."""
        n = C.c_size_t()
        self._check(lib.vdf_ctx_kernel_events(self.handle, None, 0, C.byref(n)))
        ev = np.zeros(n.value + 16, dtype=np.dtype([("name", "S24"), ("bytes", "<f8"), ("start_ms", "<f8"), ("end_ms", "<f8")]))
        self._check(lib.vdf_ctx_kernel_events(self.handle, ev.ctypes.data, ev.shape[0], C.byref(n)))
        return [(ev["name"][i].decode(), float(ev["bytes"][i]), float(ev["start_ms"][i]), float(ev["end_ms"][i])) for i in range(n.value)]

    def vec_is_zero(self, v, n: int) -> bool:
        """True iff all n field elements are zero (decided on the device)."""
        out = C.c_int(0)
        self._check(lib.vdf_vec_is_zero(self.handle, _ptr(v), n, C.byref(out)))
        return bool(out.value)

    def nifs_cross_term(self, shape: Shape, z2, az1, bz1, cz1, u1, az2, bz2, cz2, T) -> None:
        self._check(lib.vdf_nifs_cross_term(self.handle, shape.handle, _ptr(z2), _ptr(az1), _ptr(bz1), _ptr(cz1), _ptr(u1),
                                            _ptr(az2), _ptr(bz2), _ptr(cz2), _ptr(T)))

    def nifs_cross_term_rows(self, shape: Shape, row_begin: int, row_count: int, part: int, z2, az1, bz1, cz1, u1, az2, bz2, cz2, T) -> None:
        """part: 1 = only the rows [row_begin, row_begin + row_count), 2 = every row but those (vdf_hip.h VDF_ROWS_*)."""
        self._check(lib.vdf_nifs_cross_term_rows(self.handle, shape.handle, row_begin, row_count, part, _ptr(z2), _ptr(az1), _ptr(bz1),
                                                 _ptr(cz1), _ptr(u1), _ptr(az2), _ptr(bz2), _ptr(cz2), _ptr(T)))

    def nifs_cross_term_minroot(self, field, per, t, seg_begin, one_col, row_begin, z2, az1, bz1, cz1, u1, az2, bz2, cz2, T) -> None:
        """The 3t + 1 rows of a built-in MinRoot step circuit from row_begin on, by stencil (vdf_hip.h)."""
        self._check(lib.vdf_nifs_cross_term_minroot(self.handle, field, per, t, seg_begin, one_col, row_begin, _ptr(z2), _ptr(az1),
                                                    _ptr(bz1), _ptr(cz1), _ptr(u1), _ptr(az2), _ptr(bz2), _ptr(cz2), _ptr(T)))

    def nifs_cross_term_minroot_fold(self, field, per, t, seg_begin, one_col, row_begin, z2, r, az1, bz1, cz1, e1, t_prev, u1, az2, bz2, cz2, T) -> None:
        """The same rows with the previous fold of those rows applied on the way (vdf_hip.h); e1 / t_prev may be None."""
        self._check(lib.vdf_nifs_cross_term_minroot_fold(self.handle, field, per, t, seg_begin, one_col, row_begin, _ptr(z2), _ptr(r), _ptr(az1),
                                                         _ptr(bz1), _ptr(cz1), _ptr(e1) if e1 is not None else None,
                                                         _ptr(t_prev) if t_prev is not None else None, _ptr(u1), _ptr(az2), _ptr(bz2),
                                                         _ptr(cz2), _ptr(T)))

    def fold_many(self, field, r, acc, add, n) -> None:
        k = len(acc)
        a = (C.c_void_p * k)(*[_ptr(x) for x in acc])
        b = (C.c_void_p * k)(*[_ptr(x) for x in add])
        ln = (C.c_size_t * k)(*[int(x) for x in n])
        self._check(lib.vdf_fold_many(self.handle, field, _ptr(r), k, a, b, ln))

    # ---- compression SNARK building blocks: vectors on the device, scalars as host arrays ----
    def pair_table(self, field, lo, hi, k, out) -> None:
        self._check(lib.vdf_pair_table(self.handle, field, _ptr(lo), _ptr(hi), k, _ptr(out)))

    def pair_table_pattern(self, field, lo, hi, k, pattern, log_m, out) -> None:
        self._check(lib.vdf_pair_table_pattern(self.handle, field, _ptr(lo), _ptr(hi), k, _ptr(pattern), log_m, _ptr(out)))

    def fold_halves(self, field, vectors, c_lo, c_hi, n) -> None:
        k = len(vectors)
        v = (C.c_void_p * k)(*[_ptr(x) for x in vectors])
        self._check(lib.vdf_fold_halves(self.handle, field, k, v, _ptr(c_lo), _ptr(c_hi), n))

    def reduce(self, field, kind, tables, n, u=None):
        nout = 1 if kind == 0 else 3 if kind == 2 else 2
        out = np.zeros((nout, 4), dtype="<u8")
        t = (C.c_void_p * len(tables))(*[_ptr(x) for x in tables])
        self._check(lib.vdf_reduce(self.handle, field, kind, t, _ptr(u), n, _ptr(out)))
        return out

    def spmv3_t(self, shape: Shape, eq, rho, out) -> None:
        self._check(lib.vdf_spmv3_t(self.handle, shape.handle, _ptr(eq), _ptr(rho), _ptr(out)))

    def ipa_scalars(self, field, a, s, n, nj, sL, sR) -> None:
        self._check(lib.vdf_ipa_scalars(self.handle, field, _ptr(a), _ptr(s), n, nj, _ptr(sL), _ptr(sR)))

    def scale_pattern(self, field, s, n, nj, x_lo, x_hi) -> None:
        self._check(lib.vdf_scale_pattern(self.handle, field, _ptr(s), n, nj, _ptr(x_lo), _ptr(x_hi)))

    def fe_mul(self, field, a, b, n, out) -> None:
        self._check(lib.vdf_fe_mul(self.handle, field, _ptr(a), _ptr(b), n, _ptr(out)))

    def fe_to_mont(self, field, a, n, out) -> None:
        self._check(lib.vdf_fe_to_mont(self.handle, field, _ptr(a), n, _ptr(out)))

    def fe_from_mont(self, field, a, n, out) -> None:
        self._check(lib.vdf_fe_from_mont(self.handle, field, _ptr(a), n, _ptr(out)))

    def clock_probe(self, iters: int = 6000):
        """(shader MHz sustained under a multiply-bound load on every SIMD, the probe kernel's duration in ms)."""
        mhz, ms = C.c_double(), C.c_double()
        self._check(lib.vdf_ctx_clock_probe(self.handle, iters, C.byref(mhz), C.byref(ms)))
        return mhz.value, ms.value

    def fe_mul_chain(self, field, a, n, iters, out) -> None:
        self._check(lib.vdf_fe_mul_chain(self.handle, field, _ptr(a), n, iters, _ptr(out)))

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
