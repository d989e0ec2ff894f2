"""ctypes binding of libvdf_hip.so (include/vdf_hip.h).

The HIP library is the product; this module only loads it and declares prototypes.  There is no
Python or CPU fallback: if the library is missing the import raises, and without a GPU
`vdf_ctx_create` fails with VDF_ERR_NO_DEVICE.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# VDF_HIP_LIB=/path/to/other.so loads ANOTHER build of the same ABI (A/B runs of kernels: tools/ab_*.sh) without touching the
# shipped binary; libvdf_nova.so then binds to it too (same SONAME, loaded first with RTLD_GLOBAL).  Unset = the product.
LIB_PATH = os.environ.get("VDF_HIP_LIB") or os.path.join(_HERE, "libvdf_hip.so")
IS_OVERRIDE = bool(os.environ.get("VDF_HIP_LIB"))

VDF_OK, VDF_ERR_BAD_ARG, VDF_ERR_BAD_LENGTH, VDF_ERR_NONCANONICAL, VDF_ERR_DEVICE, VDF_ERR_OOM, VDF_ERR_NO_DEVICE = range(7)
CURVE_PALLAS, CURVE_VESTA = 0, 1
FIELD_FP, FIELD_FQ = 0, 1
GENS_KNOWN_DLOG, GENS_TRY_AND_INCREMENT = 0, 1

# every symbol include/vdf_hip.h declares: (name, restype, argtypes)
_vp, _sz, _i, _u64 = C.c_void_p, C.c_size_t, C.c_int, C.c_uint64
PROTOTYPES = {
    "vdf_ctx_create": (_i, [C.POINTER(_i), _i, C.POINTER(_vp)]),
    "vdf_ctx_destroy": (None, [_vp]),
    "vdf_ctx_create_pooled": (_i, [C.POINTER(_i), _i, _i, C.POINTER(_vp)]),
    "vdf_ctx_create_pooled_near": (_i, [_vp, _i, C.POINTER(_vp)]),
    "vdf_ctx_queue_info": (_i, [_vp, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)]),
    "vdf_ctx_set_stream": (_i, [_vp, _vp]),
    "vdf_ctx_get_stream": (_vp, [_vp]),
    "vdf_ctx_set_async": (_i, [_vp, _i]),
    "vdf_ctx_get_async": (_i, [_vp, C.POINTER(_i)]),
    "vdf_ctx_sync": (_i, [_vp]),
    "vdf_ctx_wait": (_i, [_vp, _vp]),
    "vdf_shim_set_cache": (_i, [_i]),
    "vdf_ctx_mark": (_i, [_vp, _i]),
    "vdf_ctx_sync_mark": (_i, [_vp, _i]),
    "vdf_ctx_wait_mark": (_i, [_vp, _vp, _i]),
    "vdf_ctx_gate_accumulate": (_i, [_vp, _vp, _i]),
    "vdf_ctx_set_light_priority": (_i, [_vp, _i]),
    "vdf_ctx_device": (_i, [_vp]),
    "vdf_last_error": (C.c_char_p, [_vp]),
    "vdf_bases_upload": (_i, [_vp, _i, _vp, _sz, C.POINTER(_vp)]),
    "vdf_bases_validate": (_i, [_vp, _vp, C.POINTER(_sz)]),
    "vdf_bases_generate": (_i, [_vp, _i, _u64, _sz, C.POINTER(_vp)]),
    "vdf_bases_generate_range": (_i, [_vp, _i, _u64, _sz, _sz, C.POINTER(_vp)]),
    "vdf_bases_generate_family": (_i, [_vp, _i, _i, _u64, _sz, _sz, C.POINTER(_vp)]),
    "vdf_bases_generate_label": (_i, [_vp, _i, _vp, _sz, _sz, _sz, C.POINTER(_vp)]),
    "vdf_bases_precompute": (_i, [_vp, _vp, _i, _i]),
    "vdf_bases_download": (_i, [_vp, _vp, _sz, _sz, _vp]),
    "vdf_bases_window": (_i, [_vp]),
    "vdf_bases_precompute_digits": (_i, [_vp, _vp, _i, _i, _vp, _vp]),
    "vdf_bases_digit_window": (_i, [_vp]),
    "vdf_bases_digit_table_bytes": (C.c_size_t, [_vp]),
    "vdf_digit_table_bytes": (C.c_size_t, [_i, _sz]),
    "vdf_bases_table_bytes": (C.c_size_t, [_vp]),
    "vdf_bases_len": (_sz, [_vp]),
    "vdf_bases_device_ptr": (_vp, [_vp]),
    "vdf_bases_free": (None, [_vp]),
    "vdf_msm": (_i, [_vp, _vp, _sz, _vp, _sz, _i, _vp]),
    "vdf_msm_batch": (_i, [_vp, _vp, _i, C.POINTER(_sz), C.POINTER(_vp), C.POINTER(_sz), _i, _vp]),
    "vdf_msm_job_begin": (_i, [_vp, _vp, _i, C.POINTER(_sz), C.POINTER(_sz), _i, C.POINTER(_vp)]),
    "vdf_msm_job_push": (_i, [_vp, _i, _vp]),
    "vdf_msm_job_finish": (_i, [_vp, _vp]),
    "vdf_ctx_set_msm_window": (_i, [_vp, _i]),
    "vdf_point_sum": (_i, [_vp, _i, _vp, _sz, _vp]),
    "vdf_ctx_set_timing": (_i, [_vp, _i]),
    "vdf_msm_sharded": (_i, [_vp, _vp, _sz, _vp, _sz, _i, _i, _i, _vp, _vp, _i, _vp, _vp, _vp]),
    "vdf_msm_multi": (_i, [C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_sz), C.POINTER(_vp), C.POINTER(_sz), _i, _i, _vp]),
    "vdf_msm_timing": (_i, [_vp, C.POINTER(C.c_float), C.POINTER(_i)]),
    "vdf_ctx_set_kernel_timing": (_i, [_vp, _i]),
    "vdf_ctx_kernel_events": (_i, [_vp, _vp, _sz, C.POINTER(_sz)]),
    "mult_pippenger_pallas": (None, [_vp, _vp, _sz, _vp, C.c_bool]),
    "mult_pippenger_vesta": (None, [_vp, _vp, _sz, _vp, C.c_bool]),
    "vdf_shape_create": (_i, [_vp, _i, _sz, _sz, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_sz), C.POINTER(_vp)]),
    "vdf_shape_free": (None, [_vp]),
    "vdf_spmv3": (_i, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "vdf_cross_term": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "vdf_axpy": (_i, [_vp, _i, _vp, _vp, _vp, _sz, _vp]),
    "vdf_minroot_witness": (_i, [_vp, _i, _vp, _vp, _u64, _vp]),
    "vdf_minroot_step_z": (_i, [_vp, _i, _vp, _u64, _vp, _vp, _vp, _vp, _vp]),
    "vdf_minroot_step_z_packed": (_i, [_vp, _i, _vp, _u64, _vp, _vp, _vp, _vp, _vp, _vp]),
    "vdf_minroot_step_segment": (_i, [_vp, _i, _vp, _u64, _vp, _i, _vp]),
    "vdf_minroot_step_segment_packed": (_i, [_vp, _i, _vp, _u64, _vp, _vp, _vp, _vp]),
    "vdf_vec_is_zero": (_i, [_vp, _vp, _sz, C.POINTER(C.c_int)]),
    "vdf_nifs_cross_term": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "vdf_nifs_cross_term_rows": (_i, [_vp, _vp, C.c_size_t, C.c_size_t, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "vdf_nifs_cross_term_minroot": (_i, [_vp, _i, _i, _u64, _sz, _sz, _sz, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "vdf_nifs_cross_term_minroot_fold": (_i, [_vp, _i, _i, _u64, _sz, _sz, _sz, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "vdf_fold_many": (_i, [_vp, _i, _vp, _i, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_sz)]),
    "vdf_pair_table": (_i, [_vp, _i, _vp, _vp, _i, _vp]),
    "vdf_pair_table_pattern": (_i, [_vp, _i, _vp, _vp, _i, _vp, _i, _vp]),
    "vdf_fold_halves": (_i, [_vp, _i, _i, C.POINTER(_vp), _vp, _vp, _sz]),
    "vdf_reduce": (_i, [_vp, _i, _i, C.POINTER(_vp), _vp, _sz, _vp]),
    "vdf_spmv3_t": (_i, [_vp, _vp, _vp, _vp, _vp]),
    "vdf_ipa_scalars": (_i, [_vp, _i, _vp, _vp, _sz, _sz, _vp, _vp]),
    "vdf_scale_pattern": (_i, [_vp, _i, _vp, _sz, _sz, _vp, _vp]),
    "vdf_fe_mul": (_i, [_vp, _i, _vp, _vp, _sz, _vp]),
    "vdf_fe_to_mont": (_i, [_vp, _i, _vp, _sz, _vp]),
    "vdf_fe_from_mont": (_i, [_vp, _i, _vp, _sz, _vp]),
    "vdf_fe_mul_chain": (_i, [_vp, _i, _vp, _sz, _i, _vp]),
    "vdf_hip_tuning_get": (_i, [_vp]),
    "vdf_hip_tuning_set": (_i, [_vp]),
    "vdf_ctx_set_accumulate_fill": (_i, [_vp, _i]),
    "vdf_ctx_clock_probe": (_i, [_vp, _i, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "vdf_dev_alloc": (_i, [_vp, _sz, C.POINTER(_vp)]),
    "vdf_dev_free": (_i, [_vp, _vp]),
    "vdf_dev_mem_info": (_i, [_vp, C.POINTER(_sz), C.POINTER(_sz)]),
    "vdf_dev_memcpy": (_i, [_vp, _vp, _vp, _sz]),
    "vdf_dev_memset": (_i, [_vp, _vp, _i, _sz]),
    "vdf_host_alloc": (_i, [_vp, _sz, C.POINTER(_vp)]),
    "vdf_host_free": (_i, [_vp, _vp]),
    "vdf_version": (C.c_char_p, []),
}


def load() -> C.CDLL:
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). vdf_amd has no CPU fallback.")
    if IS_OVERRIDE:
        import sys
        print("vdf_amd: VDF_HIP_LIB override in force: loading %s (not the shipped library)" % LIB_PATH, file=sys.stderr)
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)          # AttributeError if the ABI drifted from the header
        fn.restype = res
        fn.argtypes = args
    return lib


lib = load()
