"""CPU tests of the host mirror of the reference's `minroot` module (vdf_amd/minroot.py over
libvdf_nova.so): the reference's own unit tests (src/minroot.rs:441-543) restated at the same
sizes, plus agreement with the oracle."""
import pytest

from oracle import pasta as o
from vdf_amd.minroot import (EvalMode, Evaluation, PallasVDF, State, VestaVDF, FIELD_FP, FIELD_FQ,
                             FP_RESCUE_INVALPHA, FQ_RESCUE_INVALPHA)

VDFS = [PallasVDF, VestaVDF]


def rand_state(V, seed, k=0, y_zero=False, i=0):
    m = o.modulus(V.FIELD)
    return State.from_ints(V.FIELD, o.rand_fe(seed, 2 * k, m), 0 if y_zero else o.rand_fe(seed, 2 * k + 1, m), i)


@pytest.mark.parametrize("V", VDFS)
def test_exponents(V):                                   # src/minroot.rs:449-458
    assert V.inverse_exponent() == 5
    e = sum(l << (64 * k) for k, l in enumerate(V.exponent()))
    assert 5 * e % (o.modulus(V.FIELD) - 1) == 1
    assert V.exponent() == (FQ_RESCUE_INVALPHA if V is PallasVDF else FP_RESCUE_INVALPHA)


@pytest.mark.parametrize("V", VDFS)
def test_steps(V):                                       # src/minroot.rs:460-477
    vdf = V.new()
    m = o.modulus(V.FIELD)
    for k in range(100):
        x = o.rand_fe(42, k, m)
        xb = State.from_ints(V.FIELD, x, 0, 0).x
        y = vdf.forward_step(xb)
        assert V.inverse_step(y) == xb
        assert State(y, xb, xb).to_ints(V.FIELD)[0] == o.forward_step(x, V.FIELD)


@pytest.mark.parametrize("mode", EvalMode.all())
def test_eval(mode):                                     # src/minroot.rs:479-510 (Pallas only, :482)
    vdf = PallasVDF.new_with_mode(mode)
    for k in range(10):
        x = rand_state(PallasVDF, 42, k)
        result = vdf.eval(x, 10)
        assert PallasVDF.inverse_eval(result, 10) == x
        assert PallasVDF.check(result, 10, x)
        xi = x.to_ints(FIELD_FQ)
        exp = o.minroot_eval(o.State(*xi), 10, o.FIELD_FQ)
        assert result.to_ints(FIELD_FQ) == (exp.x, exp.y, exp.i)


@pytest.mark.parametrize("V", VDFS)
def test_vanilla_proof(V):                               # src/minroot.rs:512-542
    x = rand_state(V, 42, 0, y_zero=True)
    t, n = 4, 3
    _z0, first = Evaluation.eval(V, x, t)
    final = first
    for _ in range(1, n):
        _, new = Evaluation.eval(V, final.result(), t)
        final = final.append(new)
        assert final is not None, "failed to append proof"
    assert V.element(final.t) == final.result().i
    assert n * t == final.t
    assert final.verify(x)
    # append must refuse an evaluation that does not chain
    _, stray = Evaluation.eval(V, rand_state(V, 7, 1), t)
    assert first.append(stray) is None


def test_round_and_inverse_round_are_inverse():
    for V in VDFS:
        s = rand_state(V, 5, 0, i=17)
        assert V.inverse_round(V.new().round(s)) == s


def test_default_mode_and_vesta_ignores_mode():
    assert PallasVDF.default_mode() == EvalMode.LTRSequential          # src/minroot.rs:300-302
    s = rand_state(VestaVDF, 3)
    ref = VestaVDF.new().eval(s, 3)
    for mode in EvalMode.all():                                         # src/minroot.rs:203-205
        assert VestaVDF.new_with_mode(mode).eval(s, 3) == ref
