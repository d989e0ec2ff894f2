"""Python mirror of the reference's `minroot` module (/root/reference/src/minroot.rs) over
libvdf_nova.so.  Same names and argument meaning: EvalMode, State, PallasVDF, VestaVDF,
MinRootVDF.{round, inverse_round, eval, inverse_eval, check, forward_step, inverse_step, element,
exponent, inverse_exponent}, Evaluation.{eval, eval_with_mode, result, verify, append}.

Field elements cross this module as 32-byte little-endian Montgomery limbs (the pasta_curves
`repr-c` memory form); `State.from_ints` / `State.to_ints` convert for convenience."""
from __future__ import annotations

import ctypes as C
import enum
import os
from dataclasses import dataclass
from typing import List, Optional, Tuple

from . import _lib as _hip  # noqa: F401  (libvdf_nova.so depends on libvdf_hip.so)

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "libvdf_nova.so")
if not os.path.exists(_PATH):
    raise ImportError(f"{_PATH} is missing: run __graft_entry__.build()")
nova_lib = C.CDLL(_PATH, mode=C.RTLD_GLOBAL)

FIELD_FP, FIELD_FQ = 0, 1
P = 0x40000000000000000000000000000000224698FC094CF91B992D30ED00000001
Q = 0x40000000000000000000000000000000224698FC0994A8DD8C46EB2100000001
_R = 1 << 256
# src/minroot.rs:273-285
FP_RESCUE_INVALPHA = [0xE0F0F3F0CCCCCCCD, 0x4E9EE0C9A10A60E2, 0x3333333333333333, 0x3333333333333333]
FQ_RESCUE_INVALPHA = [0xD69F2280CCCCCCCD, 0x4E9EE0C9A143BA4A, 0x3333333333333333, 0x3333333333333333]


class _Fe(C.Structure):
    _fields_ = [("l", C.c_uint64 * 4)]


class _State(C.Structure):
    _fields_ = [("x", _Fe), ("y", _Fe), ("i", _Fe)]


_vp, _i, _u64, _sz = C.c_void_p, C.c_int, C.c_uint64, C.c_size_t
for _name, _args in {
    "vdf_minroot_forward_step": [_i, _i, C.POINTER(_Fe), C.POINTER(_Fe)],
    "vdf_minroot_inverse_step": [_i, C.POINTER(_Fe), C.POINTER(_Fe)],
    "vdf_minroot_round": [_i, _i, C.POINTER(_State), C.POINTER(_State)],
    "vdf_minroot_inverse_round": [_i, C.POINTER(_State), C.POINTER(_State)],
    "vdf_minroot_eval": [_i, _i, C.POINTER(_State), _u64, C.POINTER(_State), _vp],
    "vdf_minroot_inverse_eval": [_i, C.POINTER(_State), _u64, C.POINTER(_State)],
    "vdf_minroot_check": [_i, C.POINTER(_State), _u64, C.POINTER(_State)],
    "vdf_minroot_element": [_i, _u64, C.POINTER(_Fe)],
}.items():
    getattr(nova_lib, _name).argtypes = _args
    getattr(nova_lib, _name).restype = _i
nova_lib.vdf_nova_last_error.restype = C.c_char_p
nova_lib.vdf_nova_last_error.argtypes = []


def _modulus(field: int) -> int:
    return P if field == FIELD_FP else Q


def _fe(b: bytes) -> _Fe:
    return _Fe.from_buffer_copy(b)


class EvalMode(enum.IntEnum):     # src/minroot.rs:14-31
    LTRSequential = 0
    LTRAddChainSequential = 1
    RTLSequential = 2
    RTLAddChainSequential = 3

    @classmethod
    def all(cls) -> List["EvalMode"]:
        return [cls.LTRSequential, cls.LTRAddChainSequential, cls.RTLSequential, cls.RTLAddChainSequential]


@dataclass(frozen=True)
class State:                      # src/minroot.rs:267-272
    x: bytes
    y: bytes
    i: bytes

    @staticmethod
    def from_ints(field: int, x: int, y: int, i: int) -> "State":
        m = _modulus(field)
        return State(*[((v % m) * _R % m).to_bytes(32, "little") for v in (x, y, i)])

    def to_ints(self, field: int) -> Tuple[int, int, int]:
        m = _modulus(field)
        rinv = pow(_R, -1, m)
        return tuple(int.from_bytes(b, "little") * rinv % m for b in (self.x, self.y, self.i))

    def _c(self) -> _State:
        return _State.from_buffer_copy(self.x + self.y + self.i)

    @staticmethod
    def _from_c(s: _State) -> "State":
        raw = bytes(s)
        return State(raw[0:32], raw[32:64], raw[64:96])


class MinRootVDF:                 # trait MinRootVDF<G>, src/minroot.rs:287-374
    FIELD: int = FIELD_FQ

    def __init__(self, eval_mode: Optional[EvalMode] = None):
        self.eval_mode = self.default_mode() if eval_mode is None else EvalMode(eval_mode)

    @classmethod
    def new(cls):                                   # :291-296
        return cls(cls.default_mode())

    @classmethod
    def new_with_mode(cls, eval_mode: EvalMode):    # :298
        return cls(eval_mode)

    @staticmethod
    def default_mode() -> EvalMode:                 # :300-302
        return EvalMode.LTRSequential

    @staticmethod
    def inverse_exponent() -> int:                  # :68-70, :215-217
        return 5

    @classmethod
    def element(cls, n: int) -> bytes:              # :60-62, :207-209
        out = _Fe()
        assert nova_lib.vdf_minroot_element(cls.FIELD, n, C.byref(out)) == 0
        return bytes(out)

    def forward_step(self, x: bytes) -> bytes:      # :77-84, :223
        out = _Fe()
        assert nova_lib.vdf_minroot_forward_step(self.FIELD, int(self.eval_mode), C.byref(_fe(x)), C.byref(out)) == 0
        return bytes(out)

    @classmethod
    def inverse_step(cls, x: bytes) -> bytes:       # :73-75, :220-222
        out = _Fe()
        assert nova_lib.vdf_minroot_inverse_step(cls.FIELD, C.byref(_fe(x)), C.byref(out)) == 0
        return bytes(out)

    def round(self, s: State) -> State:             # :329-335
        out = _State()
        assert nova_lib.vdf_minroot_round(self.FIELD, int(self.eval_mode), C.byref(s._c()), C.byref(out)) == 0
        return State._from_c(out)

    @classmethod
    def inverse_round(cls, s: State) -> State:      # :338-344
        out = _State()
        assert nova_lib.vdf_minroot_inverse_round(cls.FIELD, C.byref(s._c()), C.byref(out)) == 0
        return State._from_c(out)

    def eval(self, x: State, t: int) -> State:      # :348-359
        out = _State()
        assert nova_lib.vdf_minroot_eval(self.FIELD, int(self.eval_mode), C.byref(x._c()), t, C.byref(out), None) == 0
        return State._from_c(out)

    simple_eval = eval

    @classmethod
    def inverse_eval(cls, x: State, t: int) -> State:   # :363-365
        out = _State()
        assert nova_lib.vdf_minroot_inverse_eval(cls.FIELD, C.byref(x._c()), t, C.byref(out)) == 0
        return State._from_c(out)

    @classmethod
    def check(cls, result: State, t: int, original: State) -> bool:   # :369-371
        return bool(nova_lib.vdf_minroot_check(cls.FIELD, C.byref(result._c()), t, C.byref(original._c())))


class PallasVDF(MinRootVDF):      # src/minroot.rs:38-197: modulus of Fq (scalar field of Pallas)
    FIELD = FIELD_FQ

    @staticmethod
    def exponent() -> List[int]:
        return list(FQ_RESCUE_INVALPHA)


class VestaVDF(MinRootVDF):       # src/minroot.rs:199-262: modulus of Fp; ignores the mode (:203-205)
    FIELD = FIELD_FP

    @staticmethod
    def exponent() -> List[int]:
        return list(FP_RESCUE_INVALPHA)


TargetVDF = PallasVDF             # src/minroot.rs:265


@dataclass
class Evaluation:                 # src/minroot.rs:376-439
    V: type
    result_state: State
    t: int

    @staticmethod
    def eval(V: type, x: State, t: int) -> Tuple[List[bytes], "Evaluation"]:          # :394-408
        result = V.new().eval(x, t)
        return [result.x, result.y, result.i], Evaluation(V, result, t)

    @staticmethod
    def eval_with_mode(V: type, eval_mode: EvalMode, x: State, t: int) -> "Evaluation":   # :410-418
        return Evaluation(V, V.new_with_mode(eval_mode).eval(x, t), t)

    def result(self) -> State:                                                        # :420-422
        return self.result_state

    def verify(self, original: State) -> bool:                                        # :424-426
        return self.V.check(self.result_state, self.t, original)

    def append(self, other: "Evaluation") -> Optional["Evaluation"]:                  # :428-438
        if other.verify(self.result_state):
            return Evaluation(self.V, other.result_state, self.t + other.t)
        return None
