"""GPU: the tuning surface as STRUCTS (vdf_hip_tuning / vdf_nova_tuning) instead of environment variables:
every switch changes scheduling, table sizes or kernel choice and NOTHING ELSE -- the proof made under each variant is
byte-identical to the default's (instances, witnesses through the wire format), and verifies.  Also: the digit tables are
budgeted (default 20 GiB; a budget that buys nothing leaves the bucket method), and out-of-range fields are refused."""
import ctypes

import pytest

import vdf_amd
from vdf_amd import hip
from vdf_amd.minroot import PallasVDF, State, FIELD_FQ
from vdf_amd.nova import (InverseMinRootCircuit, NovaVDFProof, public_params, tuning_default, CIRCUIT_MINROOT_REFERENCE,
                          CIRCUIT_MINROOT_BOUND, GENS_TRY_AND_INCREMENT)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = vdf_amd.Context(0)
    yield c
    c.close()


def chain(t, n, seed=5):
    initial = State.from_ints(FIELD_FQ, 0x1234567 + seed, 0, 1)
    z0, circuits = InverseMinRootCircuit.eval_and_make_circuits(PallasVDF.new(), t, n, initial)
    return z0, circuits, [initial.x, initial.y, initial.i]


def proof_bytes(pp, circuits, t, n, z0, zi):
    p = NovaVDFProof.prove_recursively(pp, circuits, t, z0)
    assert p.verify(pp, n, z0, zi)
    data = p.serialize()
    snark = p.compress(pp)
    assert snark.verify(pp, n, z0, zi)
    out = (data, snark.to_bytes())
    snark.free(); p.free()
    return out


NOVA_VARIANTS = [dict(early_rows=0), dict(early_rows=1), dict(stencil=0), dict(digit_window=-1), dict(digit_window=8), dict(digit_window=12),
                 dict(packed_commit=0), dict(lookahead_early=0), dict(gate_accumulate=0), dict(fold_on_rows=0), dict(nifs_ahead=0),
                 dict(early_row_parts=2), dict(lookahead_priority=3), dict(side_accumulate_fill=2), dict(small_window=12, big_window=14),
                 dict(digit_budget_bytes=1 << 20), dict(rows_at_challenge=0), dict(compress_queues=0), dict(fold_fused=1),
                 dict(fold_fused=1, fold_on_rows=0), dict(fold_fused=1, rows_at_challenge=0)]


@pytest.mark.parametrize("kind", [CIRCUIT_MINROOT_REFERENCE, CIRCUIT_MINROOT_BOUND], ids=["reference", "bound"])
def test_nova_tuning_changes_scheduling_only(ctx, kind):
    t, n = 80, 4
    z0, circuits, zi = chain(t, n)
    pp = public_params(ctx, t, kind, GENS_TRY_AND_INCREMENT, tuning=tuning_default())
    want = proof_bytes(pp, circuits, t, n, z0, zi)
    digest = pp.digest()
    pp.free()
    for v in NOVA_VARIANTS:
        pp1 = public_params(ctx, t, kind, GENS_TRY_AND_INCREMENT, **v)
        tn = pp1.tuning()
        for k, val in v.items():
            assert tn[k] == val
        assert pp1.digest() == digest, v
        if v.get("digit_window") == -1 or v.get("digit_budget_bytes") == 1 << 20:
            assert pp1.memory()["digit_table_bytes"] == [0, 0]
        if v.get("digit_budget_bytes") == 1 << 20:
            assert pp1.memory()["digit_tables_skipped"] == 3             # asked for, did not fit the budget: reported
        assert proof_bytes(pp1, circuits, t, n, z0, zi) == want, v
        pp1.free()


def test_digit_tables_are_budgeted(ctx):
    """The window is the widest of 12 .. 8 whose tables (both sides together) fit vdf_nova_tuning.digit_budget_bytes.  The
    ~2 x 10^4 generators that carry tables are the two augmented wrappers', whatever t is: the default 20 GiB buy the 10-bit
    tables (18-19 GB), 72 GiB the 12-bit ones (60-65 GB: an opt-in), a budget between two sizes the smaller window, and a
    fixed window is taken as given."""
    t = 64
    per = lambda c, g: int(hip.lib.vdf_digit_table_bytes(c, g))
    pp = public_params(ctx, t, CIRCUIT_MINROOT_REFERENCE, GENS_TRY_AND_INCREMENT, digit_window=0, digit_budget_bytes=20 << 30)
    total = sum(pp.memory()["digit_table_bytes"])
    gens = total // per(10, 1)
    assert total == per(10, gens) and 15000 < gens < 30000 and pp.memory()["digit_tables_skipped"] == 0   # 10-bit tables under 20 GiB
    assert per(10, gens) <= 20 << 30 < per(11, gens)
    pp.free()
    for budget, c in ((72 << 30, 12), (per(11, gens) + 1024, 11), (per(9, gens) + 1024, 9)):
        pp = public_params(ctx, t, CIRCUIT_MINROOT_REFERENCE, GENS_TRY_AND_INCREMENT, digit_window=0, digit_budget_bytes=budget)
        assert sum(pp.memory()["digit_table_bytes"]) == per(c, gens), (budget, c)
        pp.free()
    pp = public_params(ctx, t, CIRCUIT_MINROOT_REFERENCE, GENS_TRY_AND_INCREMENT, digit_window=8, digit_budget_bytes=1 << 20)
    assert sum(pp.memory()["digit_table_bytes"]) == per(8, gens)            # a window the caller fixed is not second-guessed by the budget
    pp.free()


def test_out_of_range_tuning_is_refused(ctx):
    for bad in (dict(digit_window=5), dict(early_rows=3), dict(small_window=3), dict(early_row_parts=4), dict(side_accumulate_fill=0), dict(flags=64)):
        with pytest.raises(Exception):
            public_params(ctx, 16, CIRCUIT_MINROOT_REFERENCE, GENS_TRY_AND_INCREMENT, **bad)
    t = hip.tuning_get()
    assert t.struct_size > 0
    with pytest.raises(Exception):
        hip.tuning_set(direct_priority=7)
    with pytest.raises(Exception):
        hip.tuning_set(nifs_lanes=3)
    assert hip.tuning_get().direct_priority == t.direct_priority           # a refused set changes nothing


HIP_VARIANTS = [dict(msm_direct=0), dict(direct_fused=0), dict(direct_priority=0), dict(light_priority=1), dict(accumulate_fill=3),
                dict(accumulate_fill=1), dict(accumulate_fill=2), dict(accumulate_lds=55296), dict(slice_len=16), dict(reduction=0), dict(nifs_lanes=1),
                dict(nifs_lanes=4), dict(nifs_fused=0), dict(fold_u128=0), dict(heavy_min=2, giant_span=16), dict(fixup_serial=0), dict(sort_staged=0), dict(glv=0),
                dict(fixup_serial=0, heavy_min=2, giant_span=16)]


def test_hip_tuning_changes_scheduling_only(ctx):
    t, n = 80, 3
    z0, circuits, zi = chain(t, n, seed=9)
    base = hip.tuning_get()
    pp = public_params(ctx, t, CIRCUIT_MINROOT_REFERENCE, GENS_TRY_AND_INCREMENT)
    want = proof_bytes(pp, circuits, t, n, z0, zi)
    try:
        for v in HIP_VARIANTS:
            hip.tuning_set(**v)
            got = hip.tuning_get()
            for k, val in v.items():
                assert getattr(got, k) == val
            assert proof_bytes(pp, circuits, t, n, z0, zi) == want, v
            # restore before the next variant
            hip.lib.vdf_hip_tuning_set(ctypes.byref(base))
    finally:
        hip.lib.vdf_hip_tuning_set(ctypes.byref(base))
    pp.free()
