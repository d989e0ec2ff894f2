"""GPU: bench.py's N > 1 code path on one GPU, so that it cannot rot while no 8-GPU node runs it: a process group over RCCL
with a world of one, the all-gather of 96-byte partials inside vdf_msm_sharded (the host's collective ordered on the
library's stream), the local point sum, the strong-scaling sub-record with its exactness check and the JSON contract."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_rehearses_the_collective_path():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29571", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    # a fresh child process (never an exec of this one: the test process has initialised the GPU)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--rehearse-collective", "--strong", "--strong-log2n", "20",
                        "--no-prove", "--no-cpu", "--steps", "3", "--warmup", "2"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [x for x in r.stdout.splitlines() if x.strip()]
    assert len(lines) == 1, "stdout carries exactly one JSON line"
    line = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline"):
        assert key in line, key
    assert line["n_gpus"] == 1 and line["scaling"] == "weak" and line["value"] > 0 and "workload" in line["config"]
    s = line["strong_2_20"]
    assert s["exact"] is True                       # [sum s_i k_i] G, through partial -> RCCL all-gather -> point sum
    assert s["n_gpus"] == 1 and s["scaling"] == "strong" and s["points_per_gpu"] == 1 << 20 and s["total_points"] == 1 << 20
    assert "vdf_msm_sharded" in s["path"] and s["value"] > 0
    assert line["self_check"]["ok"] is True
