"""CPU: the ONE line bench.py puts on stdout stays inside what the driver keeps of stdout (its last 8 KB) and carries the
contract's keys, whatever the full record holds (a 24 KB line left BENCH_r04.parsed null).  The full record of
round 4 (profiles/r04_bench_line.json: data, 24 KB) and an inflated copy of it are the canned inputs."""
import copy
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline")


def full_record():
    rec = json.load(open(os.path.join(ROOT, "profiles", "r04_bench_line.json")))
    rec.pop("summary", None)
    return rec


def check(line):
    text = json.dumps(line, separators=(",", ":"))
    assert len(text) <= bench.MAX_LINE_BYTES <= 8192, len(text)
    assert "\n" not in text
    for k in CONTRACT:
        assert k in line, k
    assert set(line["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"}
    assert set(line["cpu_baseline"]) >= {"value", "unit", "cores", "kind", "sample"}
    assert "workload" in line["config"] and not any(k in line["config"] for k in ("model", "seq_len", "global_batch"))
    assert list(line)[-1] == "summary"
    return text


def test_line_of_a_real_record_fits_and_keeps_the_contract():
    rec = full_record()
    line = bench.contract_line(rec)
    text = check(line)
    assert len(text) < 4096                                     # today's line: under half of what the driver keeps
    assert line["roofline"]["frac"] == round(rec["roofline"]["frac"], 6)
    assert abs(line["roofline"]["achieved"] / line["roofline"]["peak"] - line["roofline"]["frac"]) < 1e-5
    ps = line["prove_step"]
    assert len(json.dumps(ps, separators=(",", ":"))) <= 1200
    assert ps["value"] == round(rec["prove_step"]["value"], 2) and ps["cpu_baseline"]["kind"] == "port"
    assert ps["roofline"]["frac"] > 0 and all(ps["parity"].values())
    assert len(line["msm_sizes"]) == 5 and line["msm_sizes"]["exact"] is True
    assert line["summary"]["prove_step_per_s"] == ps["value"] and line["summary"]["msm_gpoints_per_s"] == round(rec["value"], 4)
    assert line["value"] == round(rec["value"], 6) and line["dtype"] == rec["dtype"]


def test_line_fits_whatever_the_record_grows_to():
    rec = full_record()

    def inflate(o):
        if isinstance(o, dict):
            d = {k: inflate(v) for k, v in o.items()}
            d.update({"extra_%d" % i: "x" * 400 for i in range(8)})
            return d
        if isinstance(o, list):
            return [inflate(v) for v in o] * 3
        return o * 12 if isinstance(o, str) else o
    big = inflate(copy.deepcopy(rec))
    for k in ("unit", "data", "dtype", "scaling", "metric"):     # the contract's own strings are bench.py's literals
        big[k] = rec[k]
    assert len(json.dumps(big)) > 200_000
    check(bench.contract_line(big))


def test_invalid_run_says_so_on_the_line():
    rec = full_record()
    rec["value"], rec["invalid"] = None, ["MSM result differs from the CPU restatement"]
    line = bench.contract_line(rec)
    check(line)
    assert line["value"] is None and line["invalid"] and "False" in line["summary"]["parity"]


def test_multi_gpu_record_keeps_its_sub_records():
    rec = json.load(open(os.path.join(ROOT, "profiles", "r04_bench_line_collective_rehearsal.json")))
    rec.pop("summary", None)
    rec.setdefault("cpu_baseline", {"value": 0.002, "unit": "GPoints/s", "cores": 16, "kind": "port", "sample": "s"})
    line = bench.contract_line(rec)
    check(line)
    strong = [k for k in line if k.startswith("strong_2_")]
    assert strong and line[strong[0]]["exact"] is True and "vdf_msm_sharded" in line[strong[0]]["path"]
