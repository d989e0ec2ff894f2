"""vdf_amd: MI355X-native hot path of the protocol/vdf Nova prover for the MinRoot VDF.

`vdf_amd.hip` binds the C ABI of `libvdf_hip.so` (include/vdf_hip.h); `vdf_amd.nova` mirrors the
reference crate's proof API (src/nova/proof.rs) on top of it.  Importing the package loads the HIP
library and fails loudly if it has not been built: there is no CPU fallback.
"""
import os as _os

# The prover drives three device queues at once; the HIP runtime multiplexes streams onto 4 hardware queues
# unless told otherwise, and kernels of one stream then wait behind another's.  Only effective before HIP initialises.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

from . import _lib  # noqa: F401,E402  (raises ImportError when libvdf_hip.so is missing)
from .hip import Context, Bases, Shape, VdfError, ints_to_limbs, limbs_to_ints  # noqa: F401,E402
from ._lib import CURVE_PALLAS, CURVE_VESTA, FIELD_FP, FIELD_FQ, GENS_KNOWN_DLOG, GENS_TRY_AND_INCREMENT  # noqa: F401,E402
