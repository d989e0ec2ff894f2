#!/usr/bin/env python3
"""bench.py -- headline measurement of the MI355X-native Nova/MinRoot hot path.

Workload (BASELINE.json configs[1], weak scaling for N > 1 as configs[3]): one Pippenger MSM over
2^20 Pallas points PER GPU.  A "step" = one pass of the hot path over one batch of synthetic input
already resident in HBM: scalars -> signed digits -> LDS-staged counting sort -> bucket accumulation ->
bucket reduction -> one Jacobian point; with N GPUs the global MSM has N * 2^20 points, sharded by
point-chunk, and each step ends with the all-gather of N 96-byte partials (RCCL over xGMI) and a
local point-sum.  value = points processed by all ranks / max-over-ranks wall time.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  `roofline` prices the dominant kernel (bucket accumulation) against the
HBM roofline with its algorithmic bytes (96 B per (base, scalar) pair) over its average
duration measured with HIP events on the library's stream.  `cpu_baseline` times the plain-C CPU
restatement (oracle/pasta_ref.c, kind "port") on this box's host cores on the same inputs and checks
the GPU result against it bit-for-bit (only this leg touches oracle/).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# A prove_step keeps three queues busy at once (the step's chain, the next step's rounds, the early rows of T) beside the
# two of the MSM leg and torch's own: with the runtime's default of 4 hardware queues they share, and the chain's
# kernels then wait behind another queue's 0.25 ms bucket accumulation (measured: 1.5 ms per step instead of 1.05).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

import numpy as np  # noqa: E402
import torch  # noqa: E402


MAX_LINE_BYTES = 6144        # the driver keeps the last 8 KB of stdout; the line stays well inside that


def _r(x, nd=4):
    """Round a float to `nd` significant-enough decimals for the line (None and non-floats pass through)."""
    return round(x, nd) if isinstance(x, float) else x


def flat_summary(line, failures=()):
    """The two halves of BASELINE's metric and what qualifies them as flat scalars: the LAST key of the line."""
    ps = line.get("prove_step") or {}
    rl = line.get("roofline") or {}
    s = {"msm_gpoints_per_s": _r(line.get("value")), "msm_ms_per_step": _r(line.get("ms_per_step")),
         "msm_single_gpoints_per_s": _r((line.get("single_msm") or {}).get("value")), "msm_roofline_frac": _r(rl.get("frac"), 5),
         "msm_valu_issue_frac": _r((rl.get("valu") or {}).get("valu_issue_frac"))}
    for k_, v_ in (line.get("msm_sizes") or {}).items():
        if isinstance(v_, dict):
            s["msm_" + k_ + "_gpoints_per_s"] = _r(v_.get("GPoints_per_s"))
            if v_.get("two_in_flight_GPoints_per_s"):
                s["msm_" + k_ + "_two_in_flight_gpoints_per_s"] = _r(v_["two_in_flight_GPoints_per_s"])
    if ps:
        s.update({"prove_step_per_s": _r(ps.get("value"), 2), "prove_step_ms_median": _r(ps.get("ms_per_step")),
                  "prove_roofline_frac": _r((ps.get("roofline") or {}).get("frac"), 5),
                  "prove_step_bound_form_per_s": _r((ps.get("bound_form") or {}).get("value"), 2),
                  "prove_step_two_chains_per_s": _r((ps.get("aggregate_over_concurrent_chains") or {}).get("value"), 2),
                  "compress_ms": _r((ps.get("compress") or {}).get("compress_ms"), 2),
                  "cpu_prove_step_per_s": _r((ps.get("cpu_baseline") or {}).get("value")),
                  "public_params_s": _r(ps.get("public_params_s"), 3),
                  "digit_table_GB": round(sum((ps.get("hbm") or {}).get("digit_table_bytes", [0])) / 1e9, 1)})
    rep = line.get("prove_step_replicas")
    if rep:                                       # N > 1: one independent chain per GPU; the whole-job rate is their sum
        s.update({"prove_step_per_s": _r(rep["value"], 2), "prove_step_per_gpu_min": _r(rep["per_gpu_min"], 2),
                  "prove_step_per_gpu_max": _r(rep["per_gpu_max"], 2), "prove_step_what": "sum over one chain per GPU (replicas)"})
    for k_, v_ in line.items():
        if k_.startswith("strong_2_") and isinstance(v_, dict):
            s["msm_sharded_" + k_[7:] + "_gpoints_per_s"] = _r(v_.get("value"))
    s["cpu_msm_gpoints_per_s"] = _r((line.get("cpu_baseline") or {}).get("value"), 6)
    s["parity"] = "bit-exact vs oracle/ in this run: %s; unpinned vs nova-snark" % (not failures)
    return s


def contract_line(full):
    """The ONE stdout line: the contract's keys, `roofline` and `cpu_baseline` as numbers (no prose), a prove_step of under
    1 KB, msm_sizes as four numbers, `summary` last.  Everything else lives in bench_detail.json.  Pure function of the full
    record so that tests/test_bench_contract.py can hold its size and key set without a GPU."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data")
    out = {k: _r(full.get(k), 6) for k in keep}
    cfg = full.get("config") or {}
    out["config"] = {"workload": str(cfg.get("workload", ""))[:240]}
    for k in ("points_per_gpu", "window_bits", "bucket_sets", "steps_in_flight"):
        if k in cfg:
            out["config"][k] = cfg[k]
    if full.get("invalid"):
        out["invalid"] = [str(x)[:120] for x in full["invalid"]][:6]
    rl = full.get("roofline") or {}
    out["roofline"] = {"bound": rl.get("bound"), "achieved": _r(rl.get("achieved"), 3), "peak": rl.get("peak"), "unit": rl.get("unit"),
                       "frac": _r(rl.get("frac"), 6), "traffic": rl.get("traffic"), "kernel": rl.get("kernel"),
                       "algorithmic_bytes_per_launch": rl.get("algorithmic_bytes_per_launch"),
                       "avg_launch_ms": _r(rl.get("avg_launch_ms"), 5),
                       "valu_issue_frac": _r((rl.get("valu") or {}).get("valu_issue_frac"))}
    cb = full.get("cpu_baseline")
    if cb:
        out["cpu_baseline"] = {"value": _r(cb.get("value"), 7), "unit": cb.get("unit"), "cores": cb.get("cores"), "kind": cb.get("kind"),
                               "sample": str(cb.get("sample", ""))[:160], "parity_bit_exact": cb.get("parity_bit_exact")}
    if full.get("self_check"):
        out["self_check"] = {"ok": full["self_check"].get("ok"), "points": full["self_check"].get("points")}
    ps = full.get("prove_step")
    if ps:
        pr = ps.get("roofline") or {}
        pc = ps.get("cpu_baseline") or {}
        p = {"metric": ps.get("metric"), "value": _r(ps.get("value"), 2), "unit": ps.get("unit"), "ms_per_step": _r(ps.get("ms_per_step")),
             "ms_by_repeat": (ps.get("ms_per_step_by_repeat") or {}).get("in_order"), "timed_steps": ps.get("timed_steps"),
             "circuit": "reference (src/nova/proof.rs:155-230)" if str(ps.get("circuit", "")).startswith("reference") else "bound",
             "verified": ps.get("verified"),
             "roofline": {"bound": pr.get("bound"), "achieved": _r(pr.get("achieved"), 2), "peak": pr.get("peak"), "unit": pr.get("unit"),
                          "frac": _r(pr.get("frac"), 5), "algorithmic_bytes_per_step": pr.get("algorithmic_bytes_per_step"),
                          "device_busy_frac": _r(pr.get("device_busy_frac"))},
             "cpu_baseline": {"value": _r(pc.get("value")), "unit": pc.get("unit"), "cores": pc.get("cores"), "kind": pc.get("kind"),
                              "parity_bit_exact": pc.get("parity_bit_exact")},
             "parity": pc.get("parity"),
             "two_chains_per_s": _r((ps.get("aggregate_over_concurrent_chains") or {}).get("value"), 2),
             "bound_form_per_s": _r((ps.get("bound_form") or {}).get("value"), 2),
             "compress_ms": _r((ps.get("compress") or {}).get("compress_ms"), 2),
             "compress_verified": (ps.get("compress") or {}).get("verified"),
             "cpu_config1_per_s": _r((ps.get("cpu_baseline_config1") or {}).get("value"))}
        out["prove_step"] = {k: v for k, v in p.items() if v is not None}
    ms = full.get("msm_sizes")
    if ms:
        out["msm_sizes"] = {k: _r(v.get("GPoints_per_s")) for k, v in ms.items() if isinstance(v, dict)}
        out["msm_sizes"].update({k + "_two_in_flight": _r(v["two_in_flight_GPoints_per_s"]) for k, v in ms.items()
                                 if isinstance(v, dict) and v.get("two_in_flight_GPoints_per_s")})
        out["msm_sizes"]["exact"] = all(v.get("exact", False) for v in ms.values() if isinstance(v, dict))
    for k, v in full.items():
        if k.startswith("strong_2_") and isinstance(v, dict):
            out[k] = {kk: _r(v.get(kk)) for kk in ("value", "unit", "n_gpus", "scaling", "total_points", "points_per_gpu", "ms_per_msm",
                                                   "steps", "exact", "path")}
    rep = full.get("prove_step_replicas")
    if rep:
        out["prove_step_replicas"] = {kk: _r(rep.get(kk), 2) for kk in ("value", "unit", "n_gpus", "per_gpu_min", "per_gpu_max", "verified",
                                                                        "scaling")}
    out["detail"] = "bench_detail.json (beside bench.py) and stderr"
    out["summary"] = full.get("summary") or flat_summary(full, full.get("invalid") or ())

    def clamp(o, depth=0):
        """No prose and no tables on the line, whatever the legs put in their records: strings to 200 characters, lists to 12
        items, dictionaries to 24 keys, nothing deeper than three levels."""
        if isinstance(o, str):
            return o[:200]
        if isinstance(o, dict):
            return {str(k)[:48]: clamp(v, depth + 1) for k, v in list(o.items())[:24]} if depth < 3 else None
        if isinstance(o, (list, tuple)):
            return [clamp(v, depth + 1) for v in list(o)[:12]] if depth < 3 else None
        return o
    out = clamp(out)
    # the size is a contract too: shed the optional sub-records, least important first, rather than ever exceed it
    for drop in ("msm_sizes", "prove_step_replicas", "self_check", "detail"):
        if len(json.dumps(out, separators=(",", ":"))) <= MAX_LINE_BYTES:
            break
        out.pop(drop, None)
    if len(json.dumps(out, separators=(",", ":"))) > MAX_LINE_BYTES:
        out["summary"] = {k: v for k, v in out["summary"].items() if k in ("msm_gpoints_per_s", "prove_step_per_s", "msm_roofline_frac")}
    return out


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100, help="timed MSM steps (100 x 1.3 ms: a region long enough for +-1 %)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--regions", type=int, default=5, help="the timed region of `steps` steps is run this many times; the median region is reported")
    ap.add_argument("--no-sizes", action="store_true", help="skip the msm_sizes leg (2^20, 2^22, 2^24 one at a time; variable-base 2^20)")
    ap.add_argument("--settle", type=int, default=40, help="untimed MSM steps during set-up, before the warm-up steps (clock ramp, workspaces)")
    ap.add_argument("--log2n", type=int, default=20, help="points per GPU (2^k)")
    ap.add_argument("--window", type=int, default=0, help="fixed-base window bits; 0 = the library's recommendation for this size")
    ap.add_argument("--sets", type=int, default=1, help="bucket sets of the fixed-base table (1 = no Horner tail)")
    ap.add_argument("--bases", choices=["tai", "dlog"], default="tai",
                    help="generator family: seeded try-and-increment (SURVEY.md 8d config 2) or [k_i]G with known k_i")
    ap.add_argument("--depth", type=int, default=0, help="independent MSM steps in flight (contexts / streams); 0 = two for timed regions of fewer than "
                    "40 steps, three for longer ones.  Steps started together sort, accumulate and reduce together for a few rounds before they drift "
                    "into overlap, and a region ends with as many tails as are in flight: a region's two ends cost ~1.7 ms at three in flight and "
                    "~0.6 ms at two, a steady step 1.08 ms at three and 1.11 at two (profiles/r05_chained_accumulations_not_adopted.txt, one box, "
                    "interleaved: 20 steps 0.908-0.919 at three against 0.919-0.934 at two; 100 steps 0.962-0.971 against 0.938-0.947; four: 0.83-0.87)")
    ap.add_argument("--rehearse-collective", action="store_true",
                    help="take the N > 1 code path (process group, all-gather on the step's stream) with a world of one")
    ap.add_argument("--strong-log2n", type=int, default=24,
                    help="BASELINE config 4: one MSM of 2^k points TOTAL, sharded over the ranks (strong scaling), as a sub-record; "
                         "runs for N > 1 (and for N = 1 with --rehearse-collective or --strong); 0 = skip")
    ap.add_argument("--strong", action="store_true", help="run the strong-scaling sub-record on one GPU too")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-prove", action="store_true", help="skip the prove_step leg (BASELINE config 3)")
    ap.add_argument("--prove-log2t", type=int, default=16, help="MinRoot iterations per prove_step (2^k)")
    ap.add_argument("--prove-chains", type=int, default=2,
                    help="also report the aggregate rate of this many independent chains proven concurrently (1 = skip)")
    ap.add_argument("--prove-steps", type=int, default=52, help="1 base case + 1 warm-up fold + timed steady-state folds (50: a timed repeat opens with an empty "
                    "lookahead and closes with a drain -- about 0.5 ms whatever its length, 2.4 % of 24 steps, 1.2 % of 50; tools/gpu_prove_modes.py: 100 steps)")
    ap.add_argument("--prove-repeats", type=int, default=7, help="the steady state is timed this many times (a fresh proof each): 7 x 24 = 168 timed steps; the median repeat is reported (a host stall of a few ms -- other tenants of the node -- lands in one or two)")
    ap.add_argument("--digit-budget-gib", type=int, default=20,
                    help="HBM the prove_step leg lets the digit tables of a parameter set take (vdf_nova_tuning.digit_budget_bytes).  The "
                         "default is the LIBRARY's default, 20 GiB (10-bit tables, 19 GB at t = 2^16): the headline is what a host gets "
                         "without asking.  40 GiB buys the 11-bit tables (35 GB) for ~1 % (r4: 1,130-1,145 against 1,123 prove_step/s), 72 GiB "
                         "the 12-bit ones (65 GB) for nothing more; with another value the library's default is measured beside it as "
                         "prove_step.library_default_budget (a leg that follows the main one in the same process runs ~10 % below its own "
                         "process's rate: DESIGN.md 4.3.4)")
    ap.add_argument("--no-bound-form", action="store_true", help="skip the bound-form sub-record of the prove_step leg")
    ap.add_argument("--no-reference-cases", action="store_true", help="skip the reference's own bench cases (benches/nova.rs:62-66)")
    return ap.parse_args()


def usable_cores():
    """Host cores this process may actually use: the cgroup CPU quota where one is set (the GPU box shows 256 logical CPUs
    and grants a share of them), else the affinity mask."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // p))
        except Exception:
            pass
    return n


def box_fingerprint(ctx, tag):
    """What distinguishes one leased box from another (a 12 % spread between boxes, unexplained): the host CPU
    and the share of it this process is granted, the shader clock the device sustains under a multiply-bound load on every
    SIMD (vdf_ctx_clock_probe: ~5 ms), how long the driver takes to hand out and take back 4 GiB of HBM (page-table work:
    a fragmented VRAM pool shows here and in public_params_s) and the rate of 64-byte random gathers over that block (TLB
    reach: the digit tables and fixed-base tables are gathered at random)."""
    fp = {"at": tag}
    try:
        fp["shader_mhz"], fp["clock_probe_ms"] = (round(x, 2) for x in ctx.clock_probe(6000))
    except Exception as ex:                       # the fingerprint never fails the bench
        fp["shader_mhz_error"] = str(ex)
    return fp


def host_fingerprint():
    model = None
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                model = ln.split(":", 1)[1].strip()
                break
    except Exception:
        pass
    fp = {"cpu_model": model, "logical_cpus": os.cpu_count(), "granted_cores": usable_cores()}
    try:
        fp["loadavg_1m"] = float(open("/proc/loadavg").read().split()[0])
    except Exception:
        pass
    return fp


def hbm_fingerprint(ctx, gib=4):
    """hipMalloc + hipFree of `gib` GiB through the library, and 2^22 random 64-byte gathers over the block (torch's gather
    kernel: a fingerprint of the box's address translation, not a product kernel)."""
    out = {}
    try:
        torch.cuda.synchronize()
        a = time.perf_counter()
        buf = torch.empty((gib << 30) // 64, 8, dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        out["alloc_ms_per_GiB"] = round((time.perf_counter() - a) * 1e3 / gib, 3)
        buf[::4096].zero_()                                       # touch every 256 KiB
        g = torch.Generator(device="cuda"); g.manual_seed(5)
        idx = torch.randint(0, buf.shape[0], (1 << 22,), device="cuda", generator=g)
        torch.index_select(buf, 0, idx)                           # warm-up
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4):
            torch.index_select(buf, 0, idx)
        e1.record(); torch.cuda.synchronize()
        out["gather64_GB_per_s"] = round(4 * (1 << 22) * 64 / (e0.elapsed_time(e1) * 1e-3) / 1e9, 1)
        out["gather_span_GiB"] = gib
        del buf, idx
        a = time.perf_counter()
        torch.cuda.empty_cache()
        torch.cuda.synchronize()
        out["free_ms_per_GiB"] = round((time.perf_counter() - a) * 1e3 / gib, 3)
    except Exception as ex:
        out["error"] = str(ex)
    return out


def median(xs):
    s_ = sorted(xs)
    return s_[len(s_) // 2] if len(s_) % 2 else 0.5 * (s_[len(s_) // 2 - 1] + s_[len(s_) // 2])


def msm_sizes_leg(ctx, curve, sizes=(20, 22, 24), reps=5):
    """north_star's range in the driver's record: ONE MSM at a time (no second MSM in flight) of 2^20, 2^22 and 2^24 Pallas
    points on this GPU over a fixed-base table (the library's window for the size), `reps` timed MSMs each, and the
    VARIABLE-BASE path at 2^20 (no table: device-resident points as pasta-msm's mult_pippenger receives them, seam B2 --
    bucket sets per window + on-device Horner).  Generators with known discrete logarithms, so every size is checked
    EXACTLY against [sum s_i k_i] G (host integers; no oracle code)."""
    import vdf_amd
    bm, sm = (_P, _Q) if curve == vdf_amd.CURVE_PALLAS else (_Q, _P)
    out = {}
    ctx2 = vdf_amd.Context(ctx.device)

    def timed(bases, sc, n, res):
        ctx.set_async(True)
        for _ in range(2):
            ctx.msm(bases, sc, n=n, out=res)
        ctx.sync()
        ctx.set_timing(True); ctx.msm_timing()
        per = []
        for _ in range(reps):                     # one at a time: the latency of a lone MSM, stage times from HIP events
            a = time.perf_counter()
            ctx.msm(bases, sc, n=n, out=res)
            ctx.sync()
            per.append((time.perf_counter() - a) * 1e3)
        st = ctx.msm_timing()
        ctx.set_timing(False)
        a = time.perf_counter()
        for _ in range(reps):                     # back to back on one stream: the sustained rate of one queue
            ctx.msm(bases, sc, n=n, out=res)
        ctx.sync()
        b2b = (time.perf_counter() - a) * 1e3 / reps
        ctx.set_async(False)
        cnt = max(st[4], 1)
        return per, b2b, {"sort": st[0] / cnt, "accumulate": st[1] / cnt, "tail": st[2] / cnt, "pipeline": st[3] / cnt}

    for lg in sizes:
        n = 1 << lg
        a = time.perf_counter()
        bases = ctx.bases_generate(curve, 11, n)                  # family 0: [k_i] G
        g = torch.Generator(device="cuda"); g.manual_seed(100 + lg)
        sc = torch.randint(-(2**63), 2**63 - 1, (n, 4), dtype=torch.int64, device="cuda", generator=g)
        sc[:, 3] &= 0x3FFFFFFFFFFFFFFF
        res = torch.zeros(12, dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        want = None
        rec = {}
        if lg == sizes[0]:
            per, b2b, st = timed(bases, sc, n, res)                # no table yet: the variable-base path
            want = _scalar_mul_generator(_sum_s_k(sc.cpu().numpy().view("<u8"), _dlogs(11, 0, n)) % sm, bm)
            ok = _jac_to_affine_ints(res.cpu().numpy().view("<u8").tobytes(), bm) == want
            out["variable_base_2_%d" % lg] = {
                "ms": median(per), "ms_back_to_back": b2b, "GPoints_per_s": n / b2b / 1e6, "stage_ms": st, "exact": bool(ok),
                "what": "no fixed-base table: points resident in HBM, bucket sets per window + on-device Horner (the path "
                        "behind mult_pippenger_pallas once its inputs are on the device)"}
        bases.precompute(0, 1)
        ctx.sync()
        setup_s = time.perf_counter() - a
        per, b2b, st = timed(bases, sc, n, res)
        if want is None:
            want = _scalar_mul_generator(_sum_s_k(sc.cpu().numpy().view("<u8"), _dlogs(11, 0, n)) % sm, bm)
        ok = _jac_to_affine_ints(res.cpu().numpy().view("<u8").tobytes(), bm) == want
        acc = st["accumulate"]
        # two MSMs in flight on two contexts over the one table (what `value` measures at 2^20), for the sizes north_star names
        two = None
        if lg >= 22:
            ctx2.set_async(True); ctx.set_async(True)
            res2 = torch.zeros(12, dtype=torch.int64, device="cuda")
            for _ in range(2):
                ctx.msm(bases, sc, n=n, out=res); ctx2.msm(bases, sc, n=n, out=res2)
            ctx.sync(); ctx2.sync()
            a2 = time.perf_counter()
            for _ in range(reps):
                ctx.msm(bases, sc, n=n, out=res); ctx2.msm(bases, sc, n=n, out=res2)
            ctx.sync(); ctx2.sync()
            two = 2 * reps * n / (time.perf_counter() - a2) / 1e9
            # (the two results are the same POINT; their Jacobian representatives differ with the order the sort's atomics took)
            ok = ok and _jac_to_affine_ints(res2.cpu().numpy().view("<u8").tobytes(), bm) == want
            ctx2.set_async(False); ctx.set_async(False)
        rec = {"ms": median(per), "ms_by_rep": [round(x, 4) for x in per], "ms_back_to_back": b2b, "two_in_flight_GPoints_per_s": two,
               "GPoints_per_s": n / b2b / 1e6, "window_bits": bases.window, "table_GiB": round(((255 + bases.window - 1) // bases.window) * n * 64 / 2**30, 2),
               "stage_ms": st, "k_accumulate_GB_per_s": 96.0 * n / (acc * 1e-3) / 1e9 if acc else None,
               "k_accumulate_frac": 96.0 * n / (acc * 1e-3) / 8e12 if acc else None, "exact": bool(ok),
               "generators_and_table_s": round(setup_s, 2)}
        out["table_2_%d" % lg] = rec
        bases.free()
        del sc
        torch.cuda.empty_cache()
    ctx2.close()
    out["what"] = ("one MSM at a time on one stream (ms = median latency of %d stream-synchronised calls; GPoints_per_s from the "
                   "same calls back to back), [k_i]G generators, exact = the result equals [sum s_i k_i mod q] G; `value` of this "
                   "line is the 2^20 case with the independent MSMs of --depth in flight (pipelined)" % reps)
    return out


def cpu_baseline_leg(ctx, bases, scalars_dev, n, curve, gpu_jac):
    """Rank 0, N = 1 only.  The ONLY place bench.py touches oracle/: times the C restatement on the
    host cores and checks the GPU result against it."""
    import threading
    from oracle import cref, pasta as o
    L = cref.lib()
    # The restatement spreads the windows of one MSM (16 at 2^16+ points) over a thread pool; to use every host core the
    # points are cut into chunks, each chunk an independent MSM with its own pool (ctypes calls release the GIL), and
    # the chunk results are added: chunks x 16 threads.
    cores = usable_cores()
    per_chunk = 16
    chunks = max(1, min(cores // per_chunk, 32))
    while n % chunks:
        chunks -= 1
    threads_used = chunks * min(per_chunk, cores)
    pts = bases.download(0, n)
    sc = scalars_dev.cpu().numpy().view("<u8").copy()
    outs = [np.zeros(12, dtype="<u8") for _ in range(chunks)]
    q = n // chunks

    def work(k):
        L.ref_msm(curve, cref.p(pts[k * q:(k + 1) * q]), cref.p(sc[k * q:(k + 1) * q]), q, 0, min(per_chunk, cores), 0, cref.p(outs[k]))
    t0 = time.perf_counter()
    ths = [threading.Thread(target=work, args=(k,)) for k in range(chunks)]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    bm = o.curve_base_modulus(curve)
    acc = None
    for k in range(chunks):
        aff = np.zeros(8, dtype="<u8")
        L.ref_jac_to_affine(curve, cref.p(outs[k]), cref.p(aff))
        raw = aff.tobytes()
        pt = (o.from_mont(int.from_bytes(raw[:32], "little"), bm), o.from_mont(int.from_bytes(raw[32:], "little"), bm))
        acc = o.pt_add(acc, None if pt == (0, 0) else pt, bm)
    dt = time.perf_counter() - t0
    # one thread, on the first 2^17 points of the same inputs (SURVEY.md 8d asks for a 1-thread figure beside it)
    n1 = min(n, 1 << 17)
    out1 = np.zeros(12, dtype="<u8")
    t0 = time.perf_counter()
    L.ref_msm(curve, cref.p(pts), cref.p(sc), n1, 0, 1, 0, cref.p(out1))
    dt1 = time.perf_counter() - t0
    aff_gpu = np.zeros(8, dtype="<u8")
    g = np.ascontiguousarray(gpu_jac, dtype="<u8")
    L.ref_jac_to_affine(curve, cref.p(g), cref.p(aff_gpu))
    raw = aff_gpu.tobytes()
    gpu_pt = (o.from_mont(int.from_bytes(raw[:32], "little"), bm), o.from_mont(int.from_bytes(raw[32:], "little"), bm))
    return {
        "value": n / dt / 1e9, "unit": "GPoints/s", "cores": cores, "threads_used": threads_used, "kind": "port",
        "logical_cpus_visible": os.cpu_count(),
        "sample": f"the full 2^{n.bit_length() - 1}-point MSM of the timed workload (same bases and scalars), "
                  f"{dt:.2f} s wall: {chunks} point-chunks x {min(per_chunk, cores)} window threads (oracle/pasta_ref.c), chunk results added",
        "parity_bit_exact": bool((acc or (0, 0)) == gpu_pt),
        "single_thread": {"value": n1 / dt1 / 1e9, "unit": "GPoints/s", "sample": f"first 2^{n1.bit_length() - 1} points, {dt1:.2f} s"},
    }


def cpu_prove_baseline_leg(log2t=10, nsteps=3):
    """BASELINE config 1 (MinRoot Nova prove, 1024 iterations per step, 3 recursive steps, CPU): the restatement
    oracle/nova.py -- Python big integers for the circuits and the folds, MSMs in oracle/pasta_ref.c -- timed on this
    box's host cores.  A port, and a slow one (the reference's Rust prover cannot be built here): reported, never the target."""
    from oracle import nova as nv, pasta as o
    t = 1 << log2t
    cores = usable_cores()
    com = nv.CCommit(threads=min(16, cores))
    pp = nv.public_params(t, com, nv.GENS_SEED, nv.FAMILY_TRY_AND_INCREMENT)
    states = [o.State(0x1234567890ABCDEF1234567890ABCDEF, 0, 0)]
    for _ in range(nsteps):
        states.append(o.minroot_eval(states[-1], t, o.FIELD_FQ))
    z0 = [states[nsteps].x, states[nsteps].y, states[nsteps].i]
    com(0, [1] * pp.shapes[0].num_vars); com(1, [1] * pp.shapes[1].num_vars)      # generators made outside the timed region
    s, per = None, []
    for k in range(nsteps):
        a = time.perf_counter()
        s = nv.prove_step(pp, s, nv.InverseMinRootCircuit(t, states[nsteps - k], states[nsteps - k - 1]), z0)
        per.append(time.perf_counter() - a)
    ok = nv.verify(pp, s, nsteps, z0) is not None
    steady = per[1:] if nsteps > 1 else per
    return {"value": len(steady) / sum(steady), "unit": "prove_step/s", "cores": cores, "threads_used": min(16, cores), "kind": "port",
            "sample": f"oracle/nova.py, t = 2^{log2t}, {nsteps} steps (base case {per[0]:.2f} s apart; steady-state steps timed): Python big "
                      f"integers for synthesis and folds on one core, the four commitments per step on {min(16, cores)} threads in C",
            "verified": bool(ok), "seconds_per_step": [round(x, 3) for x in per]}


_P = 0x40000000000000000000000000000000224698FC094CF91B992D30ED00000001
_Q = 0x40000000000000000000000000000000224698FC0994A8DD8C46EB2100000001


def _dlogs(seed, start, n):
    """k_i of the known-dlog generator family (include/vdf_hip.h VDF_GENS_KNOWN_DLOG), i = start .. start + n - 1."""
    with np.errstate(over="ignore"):
        z = np.uint64((seed * 0xD1342543DE82EF95 + start) & ((1 << 64) - 1)) + np.arange(n, dtype=np.uint64)
        z = z + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return (z ^ (z >> np.uint64(31))) | np.uint64(1)


def _sum_s_k(scalars_u64, k):
    """sum s_i k_i as a Python integer: 16 x 4 exact uint64 dot products of 16-bit pieces (n <= 2^24)."""
    n = scalars_u64.shape[0]
    s16 = scalars_u64.view("<u2").reshape(n, 16).astype(np.uint64)
    k16 = k.view("<u2").reshape(n, 4).astype(np.uint64)
    acc = 0
    for a in range(16):
        col = np.ascontiguousarray(s16[:, a])
        for b in range(4):
            acc += int(np.dot(col, k16[:, b])) << (16 * (a + b))
    return acc


def _scalar_mul_generator(k, bm):
    """[k] (-1, 2) on y^2 = x^3 + 5 over F_bm, affine, Python integers (None = identity)."""
    def add(a, b):
        if a is None: return b
        if b is None: return a
        if a[0] == b[0]:
            if (a[1] + b[1]) % bm == 0: return None
            lam = 3 * a[0] * a[0] * pow(2 * a[1], -1, bm) % bm
        else:
            lam = (b[1] - a[1]) * pow(b[0] - a[0], -1, bm) % bm
        x = (lam * lam - a[0] - b[0]) % bm
        return (x, (lam * (a[0] - x) - a[1]) % bm)
    want, g = None, ((-1) % bm, 2)
    while k:
        if k & 1: want = add(want, g)
        g = add(g, g)
        k >>= 1
    return want


def _jac_to_affine_ints(raw, bm):
    R = 1 << 256
    X, Y, Z = (int.from_bytes(raw[32 * k:32 * k + 32], "little") * pow(R, -1, bm) % bm for k in range(3))
    return None if Z == 0 else (X * pow(Z, -2, bm) % bm, Y * pow(Z, -3, bm) % bm)


def strong_scaling_leg(ctx, curve, log2n, rank, world, steps, dist, always_gather):
    """BASELINE config 4: ONE MSM of 2^log2n Pallas points sharded by point-chunk over the ranks (rank g owns generators and
    scalars [start_g, start_g + count_g)), all-gather of the 96-byte partials over RCCL, local point sum -- through the
    C ABI's vdf_msm_sharded.  Generators with known discrete logarithms, so the result is checked EXACTLY, whatever N is:
    every rank adds up s_i k_i over its slice on the host, the sums are all-gathered, and rank 0 compares the GPU's point
    with [sum s_i k_i mod q] G."""
    import vdf_amd
    from vdf_amd.dist import ShardedMsm
    ntot = 1 << log2n
    sh = ShardedMsm(ctx, curve, seed=11, n_total=ntot, rank=rank, world=world, table=(0, 1), family=vdf_amd.GENS_KNOWN_DLOG)
    g = torch.Generator(device="cuda")
    g.manual_seed(4321 + rank)
    # two scalar vectors, alternated step by step: a step that read a stale partial or gathered buffer (an unordered
    # collective) would return the OTHER vector's point and fail the exactness check below
    scs = []
    for _ in range(2):
        v = torch.randint(-(2**63), 2**63 - 1, (sh.count, 4), dtype=torch.int64, device="cuda", generator=g)
        v[:, 3] &= 0x3FFFFFFFFFFFFFFF
        scs.append(v)
    partial = torch.zeros(12, dtype=torch.int64, device="cuda")
    gathered = torch.zeros(world * 12, dtype=torch.int64, device="cuda")
    result = torch.zeros(12, dtype=torch.int64, device="cuda")
    collective = world > 1 or always_gather
    torch.cuda.synchronize()                 # the inputs were made on torch's default stream; the MSM runs on the context's
    run_stream = torch.cuda.ExternalStream(ctx.stream) if ctx.stream else torch.cuda.current_stream()

    def all_gather(dst, src):
        dist.all_gather_into_tensor(dst, src)

    def fence():
        torch.cuda.synchronize()
        if collective:
            dist.barrier()
            torch.cuda.synchronize()
    # partial MSM, all-gather and point sum all on the context's stream (vdf_hip.h: the collective is ordered on the stream
    # the library hands to the callback; hip.py makes it torch's current stream for the call)
    with torch.cuda.stream(run_stream):
        for i in range(2):
            sh.run(scs[i & 1], partial, gathered, all_gather, out=result, always_gather=always_gather)
    fence()
    t0 = time.perf_counter()
    with torch.cuda.stream(run_stream):
        for i in range(steps):
            sh.run(scs[i & 1], partial, gathered, all_gather, out=result, always_gather=always_gather)
    fence()
    elapsed = time.perf_counter() - t0
    sc = scs[(steps - 1) & 1]                # the vector of the last step: `result` must be ITS point
    # exactness: sum s_i k_i over all ranks
    mine = _sum_s_k(sc.cpu().numpy().view("<u8"), _dlogs(11, sh.start, sh.count))
    words = torch.tensor(list(int(mine).to_bytes(48, "little")), dtype=torch.uint8, device="cuda")
    allw = torch.zeros(world * 48, dtype=torch.uint8, device="cuda")
    tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
    if collective:
        dist.all_gather_into_tensor(allw, words)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    else:
        allw = words
    sh.bases.free()
    if rank != 0:
        return None
    bm, sm = (_P, _Q) if curve == vdf_amd.CURVE_PALLAS else (_Q, _P)
    raw = allw.cpu().numpy().tobytes()
    total = sum(int.from_bytes(raw[48 * r:48 * r + 48], "little") for r in range(world)) % sm
    ok = _jac_to_affine_ints(result.cpu().numpy().view("<u8").tobytes(), bm) == _scalar_mul_generator(total, bm)
    dt = float(tmax.item()) / steps
    return {"metric": "MSM GPoints/s, 2^%d Pallas points in total sharded over the GPUs (BASELINE config 4)" % log2n,
            "value": ntot / dt / 1e9, "unit": "GPoints/s", "n_gpus": world, "scaling": "strong", "total_points": ntot,
            "points_per_gpu": ntot // world, "ms_per_msm": dt * 1e3, "steps": steps, "exact": bool(ok),
            "check": "sum s_i [k_i] G = [sum s_i k_i] G with the per-rank sums all-gathered (known-dlog generators)",
            "path": "vdf_msm_sharded (C ABI): partial -> all-gather of 96-byte partials (RCCL) -> local point sum"}


def msm_dlog_self_check(ctx, curve, scalars_dev, n):
    """A GPU-side parity check that needs no oracle: n scalars of the timed workload over [k_i]G generators (k_i from
    splitmix64, include/vdf_hip.h VDF_GENS_KNOWN_DLOG) must give [sum s_i k_i mod q] G.  The right-hand side is one
    host scalar multiplication done with Python integers here (y^2 = x^3 + 5, affine)."""
    import vdf_amd
    P_ = 0x40000000000000000000000000000000224698FC094CF91B992D30ED00000001
    Q_ = 0x40000000000000000000000000000000224698FC0994A8DD8C46EB2100000001
    bm, sm = (P_, Q_) if curve == vdf_amd.CURVE_PALLAS else (Q_, P_)
    M64 = (1 << 64) - 1

    def splitmix(x):
        z = (x + 0x9E3779B97F4A7C15) & M64
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
        return z ^ (z >> 31)

    def add(a, b):
        if a is None: return b
        if b is None: return a
        if a[0] == b[0]:
            if (a[1] + b[1]) % bm == 0: return None
            lam = 3 * a[0] * a[0] * pow(2 * a[1], -1, bm) % bm
        else:
            lam = (b[1] - a[1]) * pow(b[0] - a[0], -1, bm) % bm
        x = (lam * lam - a[0] - b[0]) % bm
        return (x, (lam * (a[0] - x) - a[1]) % bm)
    seed = 99
    bases = ctx.bases_generate(curve, seed, n)
    out = torch.zeros(12, dtype=torch.int64, device="cuda")
    ctx.msm(bases, scalars_dev[:n], n=n, out=out)
    ctx.sync()
    raw = out.cpu().numpy().view("<u8").tobytes()
    R = 1 << 256
    X, Y, Z = (int.from_bytes(raw[32 * k:32 * k + 32], "little") * pow(R, -1, bm) % bm for k in range(3))
    got = None if Z == 0 else (X * pow(Z, -2, bm) % bm, Y * pow(Z, -3, bm) % bm)
    sc = scalars_dev[:n].cpu().numpy().view("<u8").tobytes()
    acc = 0
    for i in range(n):
        acc += int.from_bytes(sc[32 * i:32 * i + 32], "little") * (splitmix((seed * 0xD1342543DE82EF95 + i) & M64) | 1)
    k, want, g = acc % sm, None, ((-1) % bm, 2)
    while k:
        if k & 1: want = add(want, g)
        g = add(g, g)
        k >>= 1
    bases.free()
    return {"ok": bool(got == want), "points": n, "what": "sum s_i [k_i] G = [sum s_i k_i] G over known-dlog generators"}


CIRCUIT_NAMES = {1: "reference (4 variables per round: src/nova/proof.rs:155-230 as written)", 0: "bound (3 variables per round: new_x as the linear combination y - i + 1)"}


def step_algorithmic_bytes(sizes, t, vars_per_round):
    """SURVEY.md 8d's per-unit figures over the ACTUAL shapes of both sides (augmented circuits included): commitments 96 B per
    (generator, scalar) pair, cross term 224 B per row, folds 96 B per element (W and E), MinRoot rounds 64 B read + 32 B per
    variable written, one multiply_vec = sum over A, B, C of nnz * 36 + rows * 36 + cols * 32 (raw 32-byte coefficients).
    `as_reference_computes`: two multiply_vec per fold (running and fresh z), what nova-snark does and SURVEY's 0.245 GB
    counts; `one_multiply_vec`: A z, B z, C z of the running instance folded along instead (what this library does)."""
    parts = {"msm_W": 0.0, "msm_T": 0.0, "cross_term": 0.0, "folds": 0.0, "multiply_vec": 0.0}
    for side in ("primary", "secondary"):
        z = sizes[side]
        cols = z["num_vars"] + 1 + z["num_io"]
        parts["msm_W"] += 96.0 * z["num_vars"]
        parts["msm_T"] += 96.0 * z["num_cons"]
        parts["cross_term"] += 224.0 * z["num_cons"]
        parts["folds"] += 96.0 * (cols + z["num_cons"])
        parts["multiply_vec"] += 36.0 * z["nnz"] + 3 * (36.0 * z["num_cons"] + 32.0 * cols)
    parts["minroot_rounds"] = (64.0 + 32.0 * vars_per_round) * t
    fixed = sum(v for k, v in parts.items() if k != "multiply_vec")
    return {"parts": parts, "as_reference_computes": fixed + 2 * parts["multiply_vec"], "one_multiply_vec": fixed + parts["multiply_vec"]}


def kernel_report(events, nsteps):
    """Per-kernel table and the device's busy fraction from the launches of `nsteps` steps (queue, name, bytes, start, end)."""
    agg = {}
    for q, name, nbytes, a, b in events:
        e = agg.setdefault((name, q), {"kernel": name, "queue": q, "calls": 0, "us": 0.0, "bytes": 0.0})
        e["calls"] += 1; e["us"] += (b - a) * 1e3; e["bytes"] += nbytes
    rows = []
    for e in sorted(agg.values(), key=lambda e: -e["us"]):
        rows.append({"kernel": e["kernel"], "queue": ("chain", "lookahead", "early_rows")[e["queue"]],
                     "calls_per_step": round(e["calls"] / nsteps, 2), "avg_us": e["us"] / e["calls"],
                     "us_per_step": e["us"] / nsteps, "bytes_per_call": e["bytes"] / e["calls"],
                     "GB_per_s": (e["bytes"] / e["us"] * 1e-3) if e["bytes"] and e["us"] else None})
    iv = sorted((a, b) for _, _, _, a, b in events)
    busy, cur_a, cur_b = 0.0, None, None
    for a, b in iv:
        if cur_b is None or a > cur_b:
            if cur_b is not None:
                busy += cur_b - cur_a
            cur_a, cur_b = a, b
        else:
            cur_b = max(cur_b, b)
    if cur_b is not None:
        busy += cur_b - cur_a
    span = (max(b for _, b in iv) - min(a for a, _ in iv)) if iv else 0.0
    return rows, (busy / span if span else None), span / nsteps if nsteps else None


def prove_step_leg(ctx, log2t, nsteps, kind=1, repeats=5, chains=2, with_compress=True, with_roofline=True, seed_offset=0,
                   circuits_in=None, chain2=None, digit_budget_gib=20):
    """BASELINE config 3: Nova prove_step for MinRoot at 2^16 iterations per step on one GPU -- a full IVC step on the
    Pallas / Vesta cycle (both augmented circuits, in-circuit NIFS verifier, see include/vdf_nova.h) over the step circuit
    `kind`.  Forward evaluation and public parameters are outside the timed region (benches/nova.rs:28-59); step 0 (base
    case) and the first fold (cold lookahead, workspaces growing) are reported apart from the steady-state steps.  The
    steady state is timed `repeats` times (a fresh proof over the same circuits each time): value = all timed steps / all
    timed seconds, with the per-repeat minimum / median / maximum beside it."""
    from vdf_amd.minroot import PallasVDF, State, FIELD_FQ, EvalMode
    from vdf_amd.nova import InverseMinRootCircuit, NovaVDFProof, public_params, INST_FRESH_SECONDARY
    t = 1 << log2t
    t0 = time.perf_counter()
    pp = public_params(ctx, t, kind, digit_budget_bytes=digit_budget_gib << 30)
    pp_s = time.perf_counter() - t0
    initial = State.from_ints(FIELD_FQ, 0x1234567890ABCDEF1234567890ABCDEF + (seed_offset << 64), 0, 0)   # y = 0, i = 0: benches/nova.rs:24-26
    if circuits_in is None:
        t0 = time.perf_counter()
        z0, circuits = InverseMinRootCircuit.eval_and_make_circuits(
            PallasVDF.new_with_mode(EvalMode.LTRAddChainSequential), t, nsteps, initial)
        eval_s = time.perf_counter() - t0
        circuits.upload(ctx)                     # the forward trace is an input: resident in HBM before timing starts
    else:
        z0, circuits, eval_s = circuits_in
    # A step's MinRoot rounds and their share of the commitment are in flight one step ahead on a second context; the
    # steady state is timed as one region closed by the commitment the last step leaves pending and a synchronisation.
    was_async = ctx.get_async()
    ctx.set_async(True)
    first_timed = 2 if nsteps > 2 else 1
    nsteady = max(nsteps - first_timed, 1)
    rates, stages, base_case, first_fold, proof, outliers = [], [], 0.0, 0.0, None, []
    fp_before = box_fingerprint(ctx, "before the timed repeats")
    # Settle (untimed): one whole pass over the chain -- base case + every fold -- before the first timed repeat: the device
    # reaches the clock it then holds, every workspace has its final size, the helper threads are awake (the
    # first repeat of a cold leg ran 10 % slow and the mean carried it)
    # The stage times (vdf_nova_last_step_ms: host wall-clock per stage of a step) are read in THIS pass, after every step: inside
    # a timed repeat the harness does nothing between two steps -- whatever it does there is on the chain's critical path (the
    # next step's first launches wait for it; reading the stages after every step cost 3 %, after every fourth 1 %).
    settle_steps = 0
    trace = bool(os.environ.get("VDF_BENCH_TRACE"))
    if nsteps > 2:
        for k in range(nsteps):
            proof = NovaVDFProof.prove_step(pp, proof, circuits, k, z0)
            if k >= first_timed:
                stages.append(proof.last_step_ms())
                if trace:
                    print("step %d" % k, {a_: round(b_, 3) for a_, b_ in stages[-1].items()}, file=sys.stderr)
        proof.instance(INST_FRESH_SECONDARY)
        ctx.sync()
        settle_steps = nsteps
    import gc
    for rep in range(max(1, repeats)):
        if proof is not None:
            proof.free()
        # the interpreter's cyclic collector runs when it pleases: one full collection of this process's objects (torch is
        # imported) is ~45 ms -- fifty steps' worth -- inside whichever repeat it lands in (r4: per_repeat_slowest_call showed one
        # such call per run).  It is the harness's, not the prover's: collected here, switched off for the timed steps.
        gc.collect()
        gc.disable()
        a = time.perf_counter()
        proof = NovaVDFProof.prove_step(pp, None, circuits, 0, z0)
        ctx.sync()
        if rep == 0:
            base_case = time.perf_counter() - a
        if nsteps > 2:
            a = time.perf_counter()
            proof = NovaVDFProof.prove_step(pp, proof, circuits, 1, z0)
            ctx.sync()
            if rep == 0:
                first_fold = time.perf_counter() - a
        a = time.perf_counter()
        prev, worst = a, (0.0, -1)
        for k in range(first_timed, nsteps):
            proof = NovaVDFProof.prove_step(pp, proof, circuits, k, z0)
            now = time.perf_counter()
            if now - prev > worst[0]:
                worst = (now - prev, k)
            prev = now
            if nsteps <= 2:
                stages.append(proof.last_step_ms())
        proof.instance(INST_FRESH_SECONDARY)     # the last secondary commitment (it rides in the NEXT step's batch otherwise)
        ctx.sync()
        end = time.perf_counter()
        gc.enable()
        rates.append((end - a) / nsteady)
        outliers.append({"slowest_call_ms": round(worst[0] * 1e3, 3), "at_step": worst[1], "closing_sync_ms": round((end - prev) * 1e3, 3)})
    fp_after = box_fingerprint(ctx, "after the timed repeats")
    ok = proof.verify(pp, nsteps, z0, [initial.x, initial.y, initial.i])
    avg = median(rates) if nsteps > 1 else base_case          # the MEDIAN repeat: one cold or disturbed repeat does not move it
    srt = sorted(rates)
    stage_avg = {k: sum(s_[k] for s_ in stages) / len(stages) for k in stages[-1]} if stages else {}
    sizes = {"primary": pp.sizes(0), "secondary": pp.sizes(1)}
    per = 4 if kind == 1 else 3
    out = {"metric": "Nova prove_step/sec (MinRoot, 2^%d iters/step)" % log2t, "value": 1.0 / avg, "unit": "prove_step/s",
           "circuit": CIRCUIT_NAMES[kind], "ms_per_step": avg * 1e3,
           "timed_steps": nsteady * len(rates) if nsteps > 1 else 0, "repeats": len(rates),
           "ms_per_step_by_repeat": {"min": srt[0] * 1e3, "median": median(rates) * 1e3, "max": srt[-1] * 1e3,
                                     "mean": sum(rates) / len(rates) * 1e3, "in_order": [round(x * 1e3, 4) for x in rates]},
           "value_is": "1 / median over the repeats of (wall time of the timed folds / their number); every repeat a fresh proof",
           "settle_steps_untimed": settle_steps, "box": [fp_before, fp_after], "per_repeat_slowest_call": outliers,
           "base_case_ms": base_case * 1e3, "first_fold_ms_untimed_warmup": first_fold * 1e3,
           "steady_state_steps": nsteady if nsteps > 1 else 0,
           "stage_ms": stage_avg, "verified": bool(ok), "shape": sizes, "public_params_s": pp_s,
           "public_params_ms_split": {k_: round(v_, 1) for k_, v_ in pp.setup_ms().items()}, "hbm": pp.memory(),
           "tuning": {k_: v_ for k_, v_ in pp.tuning().items() if k_ != "struct_size"}, "early_rows_stencil": pp.stencil(),
           "segment_commitment": "the MinRoot rounds' 4t + 1 variables committed as an MSM of 3t + 4 terms over derived generators "
                                 "(new_x is an affine image of new_y: same group element; vdf_hip.h vdf_minroot_step_segment_packed)" if kind == 1 else
                                 "the MinRoot rounds' 3t + 1 variables, one MSM",
           "forward_eval_s_per_step_host": eval_s / nsteps,
           "stage": "Nova IVC step on the Pallas/Vesta cycle: NIFS of the previous secondary instance, synthesis + commitment + NIFS "
                    "of the primary augmented circuit (MinRoot step circuit inside), synthesis of the secondary augmented circuit "
                    "(TrivialTestCircuit); constant-size proof, hash-checking verifier",
           "stage_ms_meaning": "read in the untimed settle pass over the same chain; host wall-clock per stage (vdf_nova_last_step_ms): *_synthesis = host field arithmetic of an augmented "
                               "circuit (hashes, in-circuit curve arithmetic); secondary_nifs / primary_wait = GPU cross term + commitments"}
    if with_roofline and nsteps > 3:
        # a pass of its own with HIP events around every launch of the prover's three queues (two event records per launch:
        # not part of `value`); the same kernels, names and order as the rocprofv3 --kernel-trace summary under profiles/
        proof.free()
        proof = NovaVDFProof.prove_step(pp, None, circuits, 0, z0)
        proof = NovaVDFProof.prove_step(pp, proof, circuits, 1, z0)
        ctx.sync()
        proof.set_kernel_timing(True)
        proof.kernel_events()
        nt = min(nsteps - 2, 12)
        a = time.perf_counter()
        for k in range(2, 2 + nt):
            proof = NovaVDFProof.prove_step(pp, proof, circuits, k, z0)
        proof.instance(INST_FRESH_SECONDARY)
        ctx.sync()
        timed_pass_ms = (time.perf_counter() - a) / nt * 1e3
        ev = proof.kernel_events()
        proof.set_kernel_timing(False)
        rows, busy, span_ms = kernel_report(ev, nt)
        ab = step_algorithmic_bytes(sizes, t, per)
        achieved = ab["as_reference_computes"] / avg / 1e9
        out["roofline"] = {
            "bound": "hbm", "peak": 8000.0, "unit": "GB/s",
            "algorithmic_bytes_per_step": ab["as_reference_computes"],
            "algorithmic_bytes_which": "SURVEY.md 8d's per-unit figures, 'as the reference computes' variant (two multiply_vec per fold; "
                                       "its 0.245 GB counts the bare step circuit, this the actual shapes of both augmented circuits)",
            "algorithmic_bytes_one_multiply_vec": ab["one_multiply_vec"], "algorithmic_bytes_parts": ab["parts"],
            "achieved": achieved, "frac": achieved / 8000.0,
            "achieved_meaning": "algorithmic bytes of a whole step / ms_per_step: a step is a chain through host and device, "
                                "not one kernel; the per-kernel lines below price each launch by its own bytes and duration",
            "per_kernel": rows, "device_busy_frac": busy, "device_span_ms_per_step": span_ms,
            "per_kernel_timing": "HIP events on the launching stream around every launch (vdf_ctx_kernel_events), %d steady-state "
                                 "steps in a pass of their own at %.3f ms per step (the events' overhead); kernels of the three queues "
                                 "overlap, so a duration includes the time a launch shares the SIMDs with another queue's" % (nt, timed_pass_ms)}
    if not with_compress:
        proof.free()
        pp.free()
        return out
    proof.free()
    # Aggregate rate of TWO independent chains proven concurrently on this GPU (two host threads, two contexts):
    # one chain's bucket reduction and host transcript run under the other's accumulation.  The headline `value`
    # above stays the single chain the reference's bench runs; a prover serving several VDFs gets this rate.
    if chains > 1 and nsteps > 3:
        import threading
        import vdf_amd
        work = [(ctx, pp, circuits, z0)]
        for c in range(1, chains):
            ctx2 = vdf_amd.Context(ctx.device)
            pp2 = public_params(ctx2, t, kind, digit_budget_bytes=digit_budget_gib << 30)
            if chain2 is not None and c == 1:
                z02, circ2 = chain2()
            else:
                init2 = State.from_ints(FIELD_FQ, 0x1234567890ABCDEF1234567890ABCDEF + c, 0, 0)
                z02, circ2 = InverseMinRootCircuit.eval_and_make_circuits(
                    PallasVDF.new_with_mode(EvalMode.LTRAddChainSequential), t, nsteps, init2)
            circ2.upload(ctx2)
            work.append((ctx2, pp2, circ2, z02))
        def warm():
            ps = []
            for cx, p_, cs, z_ in work:
                cx.set_async(True)
                pr = NovaVDFProof.prove_step(p_, None, cs, 0, z_)
                pr = NovaVDFProof.prove_step(p_, pr, cs, 1, z_)
                cx.sync()
                ps.append(pr)
            return ps

        def run(i, delay):
            cx, p_, cs, z_ = work[i]
            pr = proofs[i]
            if delay:
                time.sleep(delay)                     # chains started half a step apart: they settle into complementary phases at once
            t_ = time.perf_counter()
            for k in range(2, nsteps):
                pr = NovaVDFProof.prove_step(p_, pr, cs, k, z_)
            pr.instance(INST_FRESH_SECONDARY)
            cx.sync()
            spans[i] = (t_, time.perf_counter())

        rates2 = []
        for rep in range(4):                          # repeat 0 is the settle pass of this leg: run, not rated
            proofs = warm()
            gc.collect()
            gc.disable()
            spans = [None] * chains
            ths = [threading.Thread(target=run, args=(i, i * avg / chains)) for i in range(chains)]
            for th in ths:
                th.start()
            for th in ths:
                th.join()
            gc.enable()
            # aggregate over the window in which ALL chains were proving (the stagger leaves a chain alone at either end)
            a_, b_ = max(sp[0] for sp in spans), min(sp[1] for sp in spans)
            done = sum((nsteps - 2) * (b_ - a_) / (sp[1] - sp[0]) for sp in spans)
            if rep > 0:
                rates2.append(done / (b_ - a_))
            for pr in proofs:
                pr.free()
        proofs = []
        agg = median(rates2)
        out["aggregate_over_concurrent_chains"] = {"chains": chains, "value": agg, "unit": "prove_step/s",
                                                   "folds_per_chain": nsteps - 2, "repeats": len(rates2), "by_repeat": rates2,
                                                   "vs_single_chain": agg * avg,
                                                   "what": "independent chains proven by two host threads on this GPU, started half a step apart; "
                                                           "rate over the window in which all chains were proving (median of the repeats, after one unrated settle pass); "
                                                           "tools/gpu_prove_two_chains.py measures the same over 300 steps per chain"}
        for pr in proofs:
            pr.free()
        for cx, p_, cs, z_ in work[1:]:
            p_.free(); cs.free(); cx.close()
        ctx.set_async(was_async)
    # compress (src/nova/proof.rs:360-368) and verification of the compressed proof, once, outside `value`
    # (after the two-chain leg: a compression opens a second queue on this parameter set, and with two provers' six queues
    # beside the MSM leg's the process would go past the hardware queues the runtime maps streams onto)
    proof = NovaVDFProof.prove_recursively(pp, circuits, t, z0)
    # the first compression of a parameter set creates its second queue and grows two MSM workspaces: timed apart, the figure
    # reported is the second call's (as every other timed region here follows its own warm-up)
    a = time.perf_counter()
    snark = proof.compress(pp)
    compress_first_ms = (time.perf_counter() - a) * 1e3
    snark.free()
    a = time.perf_counter()
    snark = proof.compress(pp)
    compress_ms = (time.perf_counter() - a) * 1e3
    a = time.perf_counter()
    ok_c = snark.verify(pp, nsteps, z0, [initial.x, initial.y, initial.i])
    verify_c_ms = (time.perf_counter() - a) * 1e3
    # the same compression once more with HIP events around every launch (a pass of its own, not the figure above): where
    # the time goes, pass by pass, each priced by its own bytes against the 8 TB/s roof (the sum-check passes are
    # the genuinely HBM-bound work of this library)
    ctx.sync()
    ctx.set_kernel_timing(True); ctx.kernel_events()
    a = time.perf_counter()
    snark2 = proof.compress(pp)
    ctx.sync()
    timed_ms = (time.perf_counter() - a) * 1e3
    cev = ctx.kernel_events()
    ctx.set_kernel_timing(False)
    snark2.free()
    agg = {}
    for name, nbytes, s0, s1 in cev:
        e = agg.setdefault(name, {"kernel": name, "calls": 0, "ms": 0.0, "bytes": 0.0})
        e["calls"] += 1; e["ms"] += s1 - s0; e["bytes"] += nbytes
    crow = []
    for e in sorted(agg.values(), key=lambda e: -e["ms"]):
        gbs = e["bytes"] / (e["ms"] * 1e-3) / 1e9 if e["bytes"] and e["ms"] else None
        crow.append({"kernel": e["kernel"], "calls": e["calls"], "ms": round(e["ms"], 3), "avg_us": round(e["ms"] / e["calls"] * 1e3, 1),
                     "MB": round(e["bytes"] / 1e6, 1), "GB_per_s": gbs and round(gbs, 1), "frac_of_8TBs": gbs and round(gbs / 8000.0, 4)})
    dev_ms = sum(e["ms"] for e in agg.values())
    msm_ms = sum(e["ms"] for e in agg.values() if e["kernel"].startswith(("msm_", "k_accumulate", "k_direct")))
    out["compress"] = {"compress_ms": compress_ms, "verify_compressed_ms": verify_c_ms, "verified": bool(ok_c),
                       "per_kernel": crow, "per_kernel_pass_ms": round(timed_ms, 2), "device_ms_sum_of_launches": round(dev_ms, 2),
                       "msm_share_of_device_time": round(msm_ms / dev_ms, 3) if dev_ms else None,
                       "host_and_idle_ms": round(timed_ms - dev_ms, 2), "compress_first_call_ms": compress_first_ms,
                       "per_kernel_covers": "the caller's queue: the fold of the last secondary instance, the PRIMARY side's sum-checks and its W "
                                            "opening; the primary side's E opening runs half a round behind on a third queue and the secondary "
                                            "side's argument on a second (their launches are not in this table)",
                       "argument_bytes": len(snark.to_bytes()),
                       "wire_bytes": len(snark.serialize()),
                       "what": "one Spartan-style argument with inner-product-argument openings per side of the cycle (vdf_nova.h)"}
    snark.free()
    proof.free()
    pp.free()
    return out


def reference_bench_cases_leg(ctx, cases=((10, 200), (100, 20), (1000, 2))):
    """The only measurement cases the reference defines (benches/nova.rs:62-66): (t, n) = (10, 200), (100, 20), (1000, 2); the
    whole n-step proof is timed, shape + generators and the forward evaluation made outside the closure (:28-59).  The
    reference's own step circuit; x = a seeded element, y = 0, i = 0 (:24-26)."""
    from vdf_amd.minroot import PallasVDF, State, FIELD_FQ
    from vdf_amd.nova import InverseMinRootCircuit, NovaVDFProof, public_params, CIRCUIT_MINROOT_REFERENCE
    out = []
    for t, n in cases:
        pp = public_params(ctx, t, CIRCUIT_MINROOT_REFERENCE)
        initial = State.from_ints(FIELD_FQ, 0x1234567890ABCDEF1234567890ABCDEF, 0, 0)
        z0, circuits = InverseMinRootCircuit.eval_and_make_circuits(PallasVDF.new(), t, n, initial)     # default mode, as the bench's V::default
        circuits.upload(ctx)
        best, proof = None, None
        for rep in range(3):                                                    # criterion samples 10; three here, the best and all reported
            if proof is not None:
                proof.free()
            a = time.perf_counter()
            proof = NovaVDFProof.prove_recursively(pp, circuits, t, z0)
            ctx.sync()
            dt = time.perf_counter() - a
            best = dt if best is None else min(best, dt)
        ok = proof.verify(pp, n, z0, [initial.x, initial.y, initial.i])
        out.append({"num_iters_per_step": t, "num_steps": n, "whole_proof_ms": best * 1e3, "ms_per_step": best * 1e3 / n,
                    "prove_step_per_s": n / best, "verified": bool(ok), "primary_num_vars": pp.sizes(0)["num_vars"]})
        proof.free(); circuits.free(); pp.free()
    return {"cases": out, "what": "benches/nova.rs:62-66: whole n-step proof timed (prove_recursively), parameters and forward "
                                  "evaluation outside; best of 3 runs; the reference's own step circuit",
            "circuit": CIRCUIT_NAMES[1]}


def cpu_prove_baseline_c_leg(ctx, log2t, kind=1):
    """The CPU figure for BASELINE config 3 AT ITS OWN SIZE (t = 2^16): what one prove_step computes over vectors -- witness
    of the MinRoot rounds, commitment of the fresh witness, both multiply_vec, the cross term and its commitment, the folds
    of W and E, on both sides of the cycle -- run by the plain-C restatement (oracle/pasta_ref.c: ref_step_witness, ref_msm,
    ref_spmv, ref_cross_term, ref_axpy) on this box's host cores, on the inputs of a real step of the GPU prover (its running
    and fresh witnesses downloaded), and compared with what the GPU made of them.  The synthesis of the two augmented
    circuits (~10^4 variables each: hashes, in-circuit curve arithmetic) is host work in the product too and is stated apart."""
    import threading
    from concurrent.futures import ThreadPoolExecutor
    from oracle import cref, nova as nv, pasta as o
    from vdf_amd.minroot import PallasVDF, State, FIELD_FQ, FIELD_FP, EvalMode
    from vdf_amd.nova import (InverseMinRootCircuit, NovaVDFProof, public_params, shape_export, INST_RUNNING_PRIMARY,
                              INST_RUNNING_SECONDARY, INST_FRESH_SECONDARY, INST_FRESH_PRIMARY_LAST)
    L = cref.lib()
    t, n = 1 << log2t, 3
    cores = usable_cores()
    nthr = max(1, min(cores, 32))
    pp = public_params(ctx, t, kind)
    initial = State.from_ints(FIELD_FQ, 0xABCDEF, 0, 0)
    z0, circuits = InverseMinRootCircuit.eval_and_make_circuits(PallasVDF.new_with_mode(EvalMode.LTRAddChainSequential), t, n, initial)
    proof = None
    for k in range(n - 1):
        proof = NovaVDFProof.prove_step(pp, proof, circuits, k, z0)
    old = {0: proof.witness(INST_RUNNING_PRIMARY), 1: proof.witness(INST_RUNNING_SECONDARY)}
    old_inst = {0: proof.instance(INST_RUNNING_PRIMARY), 1: proof.instance(INST_RUNNING_SECONDARY)}
    fresh2 = proof.witness(INST_FRESH_SECONDARY)[0]            # the secondary instance the last step folds
    proof = NovaVDFProof.prove_step(pp, proof, circuits, n - 1, z0)
    ls, stage = proof.last_step(), proof.last_step_ms()
    fresh1 = proof.witness(INST_FRESH_PRIMARY_LAST)[0]
    new = {0: proof.witness(INST_RUNNING_PRIMARY), 1: proof.witness(INST_RUNNING_SECONDARY)}
    res, _ = circuits.states(n - 1)
    st = np.frombuffer(res.x + res.y + res.i, dtype="<u8").reshape(3, 4).copy()
    sizes = [pp.sizes(0), pp.sizes(1)]
    pool = ThreadPoolExecutor(nthr)
    # setup (untimed, like public_params): the shapes' triples cut into row ranges for the threads, the generators
    field = {0: FIELD_FQ, 1: FIELD_FP}
    curve = {0: 0, 1: 1}
    mats, gens = {}, {}
    for side in (0, 1):
        nc = sizes[side]["num_cons"]
        chunks = []
        for rows, cols, vals in shape_export(t, kind, side):
            cut = np.searchsorted(rows, np.linspace(0, nc, nthr + 1).astype(np.int64))
            for a, b in zip(cut[:-1], cut[1:]):
                if b > a:
                    r0 = int(rows[a]); r1 = int(rows[b - 1]) + 1
                    chunks.append((len(chunks) // 1, np.ascontiguousarray(rows[a:b] - r0), np.ascontiguousarray(cols[a:b]),
                                   np.ascontiguousarray(vals[a:b]), r0, r1))
            mats.setdefault(side, []).append(chunks)
            chunks = []
        ng = max(sizes[side]["num_vars"], nc)
        pts = np.zeros((ng, 8), dtype="<u8")
        step = (ng + nthr - 1) // nthr
        list(pool.map(lambda a: L.ref_tai_bases(curve[side], nv.GENS_SEED, a, min(step, ng - a), cref.p(pts[a:a + min(step, ng - a)])), range(0, ng, step)))
        gens[side] = pts

    def par(fn, total):
        step_ = (total + nthr - 1) // nthr
        list(pool.map(lambda a: fn(a, min(a + step_, total)), range(0, total, step_)))

    def msm(side, scalars, count):
        # windows of one MSM on up to 16 threads inside the C code; point chunks on top of that to use every core
        per_chunk = min(16, nthr)
        nchunk = max(1, nthr // per_chunk)
        outs = [np.zeros(12, dtype="<u8") for _ in range(nchunk)]
        q = (count + nchunk - 1) // nchunk
        sc = np.ascontiguousarray(scalars[:count])

        def one(c):
            a, b = c * q, min(count, (c + 1) * q)
            if b > a:
                L.ref_msm(curve[side], cref.p(gens[side][a:b]), cref.p(sc[a:b]), b - a, 1, per_chunk, 0, cref.p(outs[c]))
        list(pool.map(one, range(nchunk)))
        bm = o.curve_base_modulus(curve[side])
        acc = None
        for c in range(nchunk):
            aff = np.zeros(8, dtype="<u8")
            L.ref_jac_to_affine(curve[side], cref.p(outs[c]), cref.p(aff))
            raw = aff.tobytes()
            pt = (o.from_mont(int.from_bytes(raw[:32], "little"), bm), o.from_mont(int.from_bytes(raw[32:], "little"), bm))
            acc = o.pt_add(acc, None if pt == (0, 0) else pt, bm)
        return acc or (0, 0)

    def mv(side, z):
        nc = sizes[side]["num_cons"]
        outs = [cref.fe_array(nc) for _ in range(3)]
        jobs = [(k, c) for k in range(3) for c in mats[side][k]]
        list(pool.map(lambda j: L.ref_spmv(field[side], cref.p(j[1][1]), cref.p(j[1][2]), cref.p(j[1][3]), len(j[1][1]), cref.p(z),
                                           j[1][5] - j[1][4], cref.p(outs[j[0]][j[1][4]:j[1][5]])), jobs))
        return outs

    def one_side(side, z_old, E_old, u_old, z_fresh, r_int):
        f, nv_, nc = field[side], sizes[side]["num_vars"], sizes[side]["num_cons"]
        cW = msm(side, z_fresh, nv_)
        abc1, abc2 = mv(side, z_old), mv(side, z_fresh)
        T = cref.fe_array(nc)
        u1 = np.ascontiguousarray(u_old.reshape(1, 4))
        par(lambda a, b: L.ref_cross_term(f, *(cref.p(x[a:b]) for x in abc1 + abc2), cref.p(u1), b - a, cref.p(T[a:b])), nc)
        cT = msm(side, T, nc)
        r = np.frombuffer(int(o.to_mont(r_int, o.Q if side == 0 else o.P)).to_bytes(32, "little"), dtype="<u8").reshape(1, 4).copy()
        W2, E2 = cref.fe_array(nv_ + 3), cref.fe_array(nc)
        par(lambda a, b: L.ref_axpy(f, cref.p(z_old[a:b]), cref.p(r), cref.p(z_fresh[a:b]), b - a, cref.p(W2[a:b])), nv_ + 3)
        par(lambda a, b: L.ref_axpy(f, cref.p(E_old[a:b]), cref.p(r), cref.p(T[a:b]), b - a, cref.p(E2[a:b])), nc)
        return cW, cT, W2, E2

    def cpu_step():
        W = cref.fe_array(4 * t + 1)
        L.ref_step_witness(FIELD_FQ, cref.p(st), t, cref.p(W))
        sec = one_side(1, old[1][0], old[1][1], old_inst[1]["u"], fresh2, ls["r2"])
        pri = one_side(0, old[0][0], old[0][1], old_inst[0]["u"], fresh1, ls["r1"])
        return W, pri, sec
    cpu_step()                                                   # warm-up (page faults, thread pool)
    reps, a = 3, time.perf_counter()
    for _ in range(reps):
        W, pri, sec = cpu_step()
    dt = (time.perf_counter() - a) / reps
    # parity: the CPU's results on these inputs against what the GPU prover made of them
    aff = lambda arr, m: tuple(o.from_mont(int.from_bytes(np.asarray(arr).reshape(2, 4)[k].tobytes(), "little"), m) for k in range(2))
    seg_b, seg_n = pp.segment()
    want_seg = W if kind == 1 else np.concatenate([W[:4 * t].reshape(t, 4, 4)[:, 1:, :].reshape(3 * t, 4), W[4 * t:]])
    checks = {"minroot_rounds": bool(np.array_equal(fresh1[seg_b:seg_b + seg_n], want_seg)),
              "comm_W_primary": aff(ls["comm_W1"], o.P) == tuple(pri[0]), "comm_T_primary": aff(ls["comm_T1"], o.P) == tuple(pri[1]),
              "comm_T_secondary": aff(ls["comm_T2"], o.Q) == tuple(sec[1]),
              "folded_W_E_primary": bool(np.array_equal(new[0][0], pri[2]) and np.array_equal(new[0][1], pri[3])),
              "folded_W_E_secondary": bool(np.array_equal(new[1][0], sec[2]) and np.array_equal(new[1][1], sec[3]))}
    synth_ms = stage["primary_synthesis"] + stage["secondary_synthesis"]
    proof.free(); circuits.free(); pp.free()
    pool.shutdown()
    return {"value": 1.0 / (dt + synth_ms * 1e-3), "unit": "prove_step/s", "cores": cores, "threads_used": nthr, "kind": "port",
            "vector_work_s_per_step": dt, "synthesis_ms_per_step_stated_apart": synth_ms,
            "value_without_synthesis": 1.0 / dt, "parity_bit_exact": all(checks.values()), "parity": checks,
            "circuit": CIRCUIT_NAMES[kind],
            "sample": f"one steady-state step at t = 2^{log2t} (primary {sizes[0]['num_vars']} variables / {sizes[0]['num_cons']} constraints), "
                      f"average of {reps} runs after a warm-up: oracle/pasta_ref.c ref_step_witness + per side ref_msm(W) + 2 x 3 ref_spmv + "
                      f"ref_cross_term + ref_msm(T) + 2 ref_axpy on {nthr} threads; the synthesis of the two augmented circuits is the "
                      f"product's own host code (same on both legs), its time from the GPU prover's step added to the CPU figure"}


def main():
    args = parse()
    # stdout carries exactly ONE line, the JSON record: libraries that announce themselves there (RCCL prints a version
    # banner when the process group starts) are sent to stderr for the whole run
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N > 1 with torch.distributed.run (one process per GPU)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path is the product and there is no CPU fallback")
    torch.cuda.set_device(local_rank)
    import torch.distributed as dist
    if world > 1 or args.rehearse_collective:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    import vdf_amd
    from vdf_amd.dist import ShardedMsm

    curve = vdf_amd.CURVE_PALLAS
    n = 1 << args.log2n
    # `depth` contexts (stream + MSM workspace each) share one generator table; step i runs on context i % depth.
    # Steps are independent MSMs, so two in flight let one pipeline's sort and latency-bound tail run under the
    # other's ALU-bound bucket accumulation (light kernels carry a raised wave priority for exactly this).
    depth = args.depth if args.depth > 0 else (2 if args.steps < 40 else 3)
    ctxs = [vdf_amd.Context(local_rank) for _ in range(depth)]
    ctx = ctxs[0]
    family = vdf_amd.GENS_TRY_AND_INCREMENT if args.bases == "tai" else vdf_amd.GENS_KNOWN_DLOG
    shs = [ShardedMsm(ctx, curve, seed=7, n_total=n * world, rank=rank, world=world, table=(args.window, args.sets),
                      family=family)]
    for k in range(1, depth):
        shs.append(ShardedMsm(ctxs[k], curve, seed=7, n_total=n * world, rank=rank, world=world, bases=shs[0].bases))
    sh = shs[0]

    # synthetic scalars, uniform 254-bit (< q), generated on the device
    g = torch.Generator(device="cuda")
    g.manual_seed(1234 + rank)
    sc = torch.randint(-(2**63), 2**63 - 1, (n, 4), dtype=torch.int64, device="cuda", generator=g)
    sc[:, 3] &= 0x3FFFFFFFFFFFFFFF
    partial = [torch.zeros(12, dtype=torch.int64, device="cuda") for _ in range(depth)]
    gathered = [torch.zeros(world * 12, dtype=torch.int64, device="cuda") for _ in range(depth)]
    results = [torch.zeros(12, dtype=torch.int64, device="cuda") for _ in range(depth)]
    result = results[0]
    torch.cuda.synchronize()

    # each context runs on a torch stream of its own, so the collective of a step is ordered with its kernels
    streams = [torch.cuda.Stream() for _ in range(depth)]
    for c, s_ in zip(ctxs, streams):
        c.set_stream(s_.cuda_stream)
        c.set_async(True)

    def all_gather(dst, src):
        dist.all_gather_into_tensor(dst, src)

    def step(i):
        k = i % depth
        with torch.cuda.stream(streams[k]):
            shs[k].run(sc, partial[k], gathered[k], all_gather, out=results[k], always_gather=args.rehearse_collective)

    def fence():
        torch.cuda.synchronize()
        if world > 1 or args.rehearse_collective:
            dist.barrier()
            torch.cuda.synchronize()

    # Setup, before the W warm-up steps the contract asks for: bring the device to its sustained clock and grow every
    # workspace (a GPU that has idled through the generator set-up runs its first ~30 ms below the clock it then holds:
    # 20 timed steps after 3 warm-up steps measured 0.79 GPoints/s, the same 20 steps after this 0.86)
    import gc
    gc.collect()
    gc.disable()                            # (the harness's collector stays out of the timed regions -- and out of the gap between the
                                            # settle steps and them: a collection is ~50 ms of idle device; see prove_step_leg)
    fp_msm = [box_fingerprint(ctx, "before the settle steps of the MSM leg")]
    for i in range(args.settle):
        step(i)
    fence()
    for i in range(max(args.warmup, depth)):
        step(i)
    fence()
    for c in ctxs:
        c.set_timing(True)
        c.msm_timing()                    # clear
    # The contract's timed region -- EXACTLY `steps` steps between two fences -- is run `regions` times back to back and the
    # MEDIAN region is reported (at the driver's --steps 20 one region is 25 ms: a single one cannot be read finer than
    # +-3 %); ms_per_step x steps is the duration of that one region.
    region_s = []
    for _ in range(max(1, args.regions)):
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(i)
        fence()
        region_s.append(time.perf_counter() - t0)
    gc.enable()
    fp_msm.append(box_fingerprint(ctx, "after the MSM regions"))
    elapsed = median(region_s)
    sort_ms = acc_ms = tail_ms = total_ms = 0.0
    calls = 0
    for c in ctxs:
        a_, b_, c_, d_, n_ = c.msm_timing()
        sort_ms += a_; acc_ms += b_; tail_ms += c_; total_ms += d_; calls += n_
        c.set_timing(False)
    # One more short pass, one step at a time on context 0 (not part of `value`): with steps in flight a kernel's
    # HIP-event (and rocprof) duration includes the time its workgroups queue behind the other stream's, so the
    # isolated duration of the dominant kernel is measured separately for the issue-rate figure.
    iso = None
    if depth > 1:
        ctx.set_timing(True)
        ctx.msm_timing()
        for _ in range(5):
            step(0)
            ctx.sync()
        a_, b_, c_, d_, n_ = ctx.msm_timing()
        ctx.set_timing(False)
        iso = {"sort": a_ / n_, "accumulate": b_ / n_, "tail": c_ / n_, "pipeline": d_ / n_}
    if world > 1:
        t = torch.tensor(region_s, dtype=torch.float64, device="cuda")       # MAX over the ranks, region by region
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        region_s = [float(x) for x in t.tolist()]
        elapsed = median(region_s)
    strong = None
    if args.strong_log2n and (world > 1 or args.rehearse_collective or args.strong):
        for c in ctxs:
            c.set_async(True)
        strong = strong_scaling_leg(ctx, curve, args.strong_log2n, rank, world, max(3, args.steps // 4), dist,
                                    args.rehearse_collective)
    # prove_step across GPUs: one chain does not shard (each step needs the previous challenge), so N GPUs prove N
    # independent chains -- replicas, no collective on the data path; only the rates are combined here
    replicas = None
    if (world > 1 or args.rehearse_collective) and not args.no_prove:
        ctx.set_async(False)
        for k in range(1, depth):                 # the MSM legs are done: the prover needs the hardware queues (three per chain)
            shs[k] = None
            ctxs[k].close()
        ctx.sync()
        ctx.set_stream(None)                      # the chain on the context's own stream: its neighbours are the prover's side queues
        mine = prove_step_leg(ctx, args.prove_log2t, args.prove_steps, kind=1, repeats=1, chains=1, with_compress=False,
                              with_roofline=False, seed_offset=rank, digit_budget_gib=args.digit_budget_gib)
        r = torch.tensor([mine["value"], -mine["value"], mine["value"], 1.0 if mine["verified"] else 0.0],
                         dtype=torch.float64, device="cuda")
        agg = r.clone()
        dist.all_reduce(agg[0:1], op=dist.ReduceOp.SUM)
        mx = r[1:3].clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        ok = r[3:4].clone()
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        replicas = {"metric": mine["metric"], "value": float(agg[0].item()), "unit": "prove_step/s", "n_gpus": world,
                    "per_gpu_min": float(-mx[0].item()), "per_gpu_max": float(mx[1].item()), "verified": bool(ok.item() > 0.5),
                    "steady_state_steps_per_gpu": mine["steady_state_steps"], "scaling": "weak",
                    "circuit": mine["circuit"],
                    "what": "independent chains, one per GPU (replicas; a single chain does not shard); rank 0's stage times follow",
                    "stage_ms": mine["stage_ms"]}

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = (n * world) / (elapsed / args.steps) / 1e9
        acc_avg_ms = acc_ms / max(calls, 1)
        acc_iso_ms = iso["accumulate"] if iso else acc_avg_ms
        alg_bytes = 96.0 * n
        # roofline.achieved prices the kernel's own duration: the launch alone on the device (isolated).  With steps in
        # flight a launch's HIP-event duration also contains the time its workgroups queue behind the other step's; that
        # figure is kept beside it (overlapped_*), never used for frac.
        achieved = alg_bytes / (acc_iso_ms * 1e-3) / 1e9
        achieved_overlapped = alg_bytes / (acc_avg_ms * 1e-3) / 1e9
        window = shs[0].bases.window or args.window or 16
        windows_eff = (255 + window - 1) // window                    # entries per point: scalars are below 2^255, zero digits are rare
        ctx_num_simds = 4 * torch.cuda.get_device_properties(local_rank).multi_processor_count
        # measured inputs of the two derived figures come from profiles/ (tracked, with their provenance), not from literals
        def latest(prefix):
            import glob
            c = sorted(glob.glob(os.path.join(ROOT, "profiles", prefix + "_r*.json")))
            try:
                return (json.load(open(c[-1])), os.path.basename(c[-1])) if c else (None, None)
            except Exception:
                return None, None
        tj, tfile = latest("traffic")
        traffic = tj.get("k_accumulate_hbm_bytes_per_launch") if tj else None
        vm, vfile = latest("valu_model")
        valu = None
        if vm:
            per_add, cyc, ghz = vm["valu_per_bucket_addition"], vm["cycles_per_valu_wave_instruction_per_simd"], vm["sustained_shader_clock_ghz"]
            issue = lambda ms: (per_add * n * windows_eff / 64.0) / (ctx_num_simds * ms * 1e-3 * ghz * 1e9 / cyc)
            valu = {"valu_issue_frac": issue(acc_iso_ms), "of_timed_region_accumulate_only": issue(elapsed / args.steps * 1e3),
                    "valu_per_bucket_addition": per_add, "cycles_per_valu_wave_instruction_per_simd": cyc,
                    "sustained_shader_clock_ghz": ghz, "source": "profiles/" + vfile, "produced_at_commit": vm.get("produced_at_commit"), "provenance": vm.get("provenance")}
        line = {
            "metric": "MSM GPoints/s at 2^20 (Pallas, Pedersen-commitment MSM of Nova prove_step); prove_step/s in `prove_step`",
            "value": value, "unit": "GPoints/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u32", "dtype_note": "255-bit Montgomery residues in 8 x 32-bit limbs, v_mad_u64_u32; exact integer arithmetic",
            "data": "synthetic",
            "config": {"workload": f"BASELINE configs[1]: Pippenger MSM, 2^{args.log2n} Pallas points per GPU, uniform 254-bit scalars resident in HBM, "
                                   f"fixed-base table c={window}, {depth} MSMs in flight; N>1: point-chunk shards + all-gather of 96-B partials",
                       "workload_detail": f"{'seeded try-and-increment' if args.bases == 'tai' else '[k_i]G'} generators (seed 7), scalars from torch Philox "
                                          f"(seed 1234+rank), bucket sets {args.sets}, {depth} independent steps in flight on {depth} streams",
                       "points_per_gpu": n, "window_bits": window, "bucket_sets": args.sets, "steps_in_flight": depth},
            # the two ways to run the same MSM, side by side: `value` is the pipelined throughput
            "single_msm": {"latency_ms": iso["pipeline"] if iso else total_ms / max(calls, 1),
                           "value": n / ((iso["pipeline"] if iso else total_ms / max(calls, 1)) * 1e-3) / 1e9, "unit": "GPoints/s",
                           "what": "one MSM at a time on an idle device (stream-synchronised between calls)"},
            "pipelined": {"ms_per_msm": ms_per_step, "value": value, "unit": "GPoints/s", "steps_in_flight": depth},
            "timed_regions": {"regions": len(region_s), "steps_per_region": args.steps, "reported": "median region",
                              "ms_per_step_by_region": [round(x / args.steps * 1e3, 4) for x in region_s]},
            "stage_ms": {"sort": sort_ms / max(calls, 1), "accumulate": acc_avg_ms, "tail": tail_ms / max(calls, 1),
                         "pipeline": total_ms / max(calls, 1)},
            "stage_ms_one_step_at_a_time": iso,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                         "frac": achieved / 8000.0, "traffic": traffic, "traffic_source": ("profiles/" + tfile) if tfile else None,
                         "traffic_produced_at_commit": (tj or {}).get("produced_at_commit"),
                         "kernel": "k_accumulate", "algorithmic_bytes_per_launch": alg_bytes,
                         "avg_launch_ms": acc_iso_ms, "launch_timing": "HIP events on the launching stream, one step at a time",
                         "overlapped_avg_launch_ms": acc_avg_ms, "overlapped_achieved": achieved_overlapped,
                         "valu": valu,
                         "note": "k_accumulate is integer-VALU-issue bound, neither HBM nor MFMA: frac is the contract's HBM "
                                 "figure from the isolated launch.  overlapped_* is the HIP-event duration inside the timed region, "
                                 "where a launch also waits for the other in-flight step's workgroups to retire (it can exceed "
                                 "ms_per_step and is not kernel time).  valu.* is the bound that holds (DESIGN.md 4.1, 4.2)"},
        }
        if strong is not None:
            line["strong_2_%d" % args.strong_log2n] = strong
        if replicas is not None:
            line["prove_step_replicas"] = replicas
        line["box"] = {"host": host_fingerprint(), "device": fp_msm}
        if world == 1 and not args.no_sizes and not args.rehearse_collective:
            for c in ctxs:
                c.sync()
            line["box"]["hbm"] = hbm_fingerprint(ctx)
            line["msm_sizes"] = msm_sizes_leg(ctx, curve)
            for c in ctxs:
                c.set_async(True)
        if world == 1 and not args.no_prove and not args.rehearse_collective:
            ctx.set_async(False)
            for c in ctxs[1:]:                    # the MSM leg's other queues are done: a prover needs the hardware queues (three per chain)
                c.close()
            # the MSM leg ran the context on a torch stream (its collective is ordered there); the prover's chain runs on the
            # context's OWN stream again, whose two neighbours in creation order are its look-ahead and early-rows queues
            # (vdf_hip.h vdf_ctx_create_pooled_near: worth 10 % of the rate, profiles/r05_single_chain_vs_padding.txt)
            ctx.sync()
            ctx.set_stream(None)
            # one forward evaluation serves both forms of the step circuit (the circuits hold states and traces, not shapes);
            # the second chain of the two-chain leg is evaluated on another host thread meanwhile (ctypes releases the GIL)
            import threading
            from vdf_amd.minroot import PallasVDF, State, FIELD_FQ, EvalMode
            from vdf_amd.nova import InverseMinRootCircuit
            tt, second = 1 << args.prove_log2t, {}

            def eval_chain(seed, into):
                init = State.from_ints(FIELD_FQ, 0x1234567890ABCDEF1234567890ABCDEF + seed, 0, 0)
                a_ = time.perf_counter()
                into["z0"], into["circuits"] = InverseMinRootCircuit.eval_and_make_circuits(
                    PallasVDF.new_with_mode(EvalMode.LTRAddChainSequential), tt, args.prove_steps, init)
                into["eval_s"] = time.perf_counter() - a_
            th2 = None
            if args.prove_chains > 1:
                th2 = threading.Thread(target=eval_chain, args=(1, second))
                th2.start()
            first_chain = {}
            eval_chain(0, first_chain)
            first_chain["circuits"].upload(ctx)
            shared = (first_chain["z0"], first_chain["circuits"], first_chain["eval_s"])

            def chain2():
                th2.join()
                return second["z0"], second["circuits"]
            # the headline: the REFERENCE's step circuit (src/nova/proof.rs:155-230: 4 variables per round, 2^19 generators)
            line["prove_step"] = prove_step_leg(ctx, args.prove_log2t, args.prove_steps, kind=1, repeats=args.prove_repeats,
                                                chains=args.prove_chains, circuits_in=shared, chain2=chain2 if th2 else None,
                                                digit_budget_gib=args.digit_budget_gib)
            line["prove_step"]["config"] = {"digit_budget_gib": args.digit_budget_gib,
                                            "note": ("the library's default budget (vdf_nova.h)" if args.digit_budget_gib == 20 else
                                                     "digit tables opted in beyond the library's 20 GiB default (vdf_nova.h); "
                                                     "library_default_budget is the same leg on the default")}
            if args.digit_budget_gib != 20:
                dflt = prove_step_leg(ctx, args.prove_log2t, args.prove_steps, kind=1, repeats=3, chains=1, with_compress=False,
                                      with_roofline=False, circuits_in=shared, digit_budget_gib=20)
                line["prove_step"]["library_default_budget"] = {k_: dflt[k_] for k_ in ("value", "unit", "ms_per_step", "ms_per_step_by_repeat", "hbm",
                                                                                       "verified", "public_params_s", "public_params_ms_split")}
            if not args.no_bound_form:
                line["prove_step"]["bound_form"] = prove_step_leg(ctx, args.prove_log2t, args.prove_steps, kind=0, repeats=3, chains=1,
                                                                  with_compress=False, with_roofline=False, circuits_in=shared,
                                                                  digit_budget_gib=args.digit_budget_gib)
                line["prove_step"]["bound_form"]["note"] = ("the sound variant of the step circuit (3 variables per round), measured after the "
                                                            "reference's circuit in the same process on parameters of its own")
            if not args.no_reference_cases:
                line["prove_step"]["reference_bench_cases"] = reference_bench_cases_leg(ctx)
            if "hbm" in line["box"]:
                # the same probe after the proving legs' allocations (65 GB of digit tables made and freed several times): a
                # drop in the gather rate or a jump in the allocation time is this process's own fragmentation of the pool
                line["box"]["hbm_after_prove_legs"] = hbm_fingerprint(ctx)
        failures = []
        if world == 1 and not args.no_cpu:
            ctx.set_async(False)
            line["cpu_baseline"] = cpu_baseline_leg(ctx, sh.bases, sc, n, curve, result.cpu().numpy().view("<u8"))
            if not line["cpu_baseline"]["parity_bit_exact"]:
                failures.append("MSM result differs from the CPU restatement")
            if "prove_step" in line:
                # config 3 at its own size in C; config 1 (t = 1024, 3 steps) as the Python restatement runs it
                line["prove_step"]["cpu_baseline"] = cpu_prove_baseline_c_leg(ctx, args.prove_log2t, kind=1)
                if not line["prove_step"]["cpu_baseline"]["parity_bit_exact"]:
                    failures.append("prove_step: a GPU result differs from the C restatement at t = 2^%d" % args.prove_log2t)
                line["prove_step"]["cpu_baseline_config1"] = cpu_prove_baseline_leg()
        elif world == 1:
            # without the CPU leg the result is still checked: the same scalars against generators with known discrete
            # logarithms, sum s_i [k_i] G = [sum s_i k_i] G (host big integers only; no oracle code involved)
            line["self_check"] = msm_dlog_self_check(ctx, curve, sc, min(n, 1 << 16))
            if not line["self_check"]["ok"]:
                failures.append("MSM result violates the discrete-log identity")
        if strong is not None and not strong["exact"]:
            failures.append("the sharded 2^%d MSM violates the discrete-log identity" % args.strong_log2n)
        for key in ("prove_step", "prove_step_replicas"):
            if key in line and not line[key].get("verified", True):
                failures.append(key + ": the proof did not verify")
        if "prove_step" in line:
            if not line["prove_step"].get("bound_form", {}).get("verified", True):
                failures.append("prove_step.bound_form: the proof did not verify")
            for c_ in line["prove_step"].get("reference_bench_cases", {}).get("cases", []):
                if not c_["verified"]:
                    failures.append("reference bench case (%d, %d): the proof did not verify" % (c_["num_iters_per_step"], c_["num_steps"]))
        if "prove_step" in line and "compress" in line["prove_step"] and not line["prove_step"]["compress"]["verified"]:
            failures.append("the compressed proof did not verify")
        if failures:
            line["value"] = None
            line["invalid"] = failures
        line["summary"] = flat_summary(line, failures)
        if "prove_step" in line:
            line["prove_step_per_s"] = line["summary"].get("prove_step_per_s")
            line["prove_step_ms_median"] = line["summary"].get("prove_step_ms_median")
        # The WHOLE record (per-kernel tables, the compress table, tunings, box fingerprints, notes) goes to bench_detail.json
        # beside this script and to stderr; stdout carries ONE line of at most MAX_LINE_BYTES built from it (the
        # driver keeps the last 8 KB of stdout, and a 24 KB line left its record unparsed)
        detail = json.dumps(line)
        try:
            with open(os.path.join(ROOT, "bench_detail.json"), "w") as f:
                f.write(detail + "\n")
        except OSError as ex:
            print("bench.py: bench_detail.json not written: %s" % ex, file=sys.stderr)
        print("bench_detail " + detail, file=sys.stderr)
        sys.stderr.flush()
        json_out.write(json.dumps(contract_line(line), separators=(",", ":")) + "\n")
        json_out.flush()
        if failures:
            raise SystemExit("bench.py: " + "; ".join(failures))
    if world > 1 or args.rehearse_collective:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
