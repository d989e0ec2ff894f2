"""GPU: a world of TWO for the sharded MSM with real GPU partials.  No multi-GPU node is
available to the tests, so both ranks run on cuda:0 as fresh child processes over a gloo process group (RCCL refuses two
ranks on one device): tests/world2_rank.py.  What this covers that tests/test_dist.py (gloo, injected CPU backend) and
tests/test_gpu_bench.py (RCCL, world of one) do not: two processes, two different shards and partials, a collective that
really exchanges them, and the point sum of both -- through the C ABI's vdf_msm_sharded on each rank.  Still no scaling
measurement: the multi-GPU path stays unmeasured on hardware."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_sharded_msm_world_of_two_on_one_gpu():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(2):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK="0",
                   HSA_ENABLE_IPC_MODE_LEGACY="0", GLOO_SOCKET_IFNAME="lo")
        # fresh children (never an exec of this process, which has initialised the GPU)
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "world2_rank.py")], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    try:
        for p in procs:
            out, err = p.communicate(timeout=600)
            outs.append((p.returncode, out, err))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for rank, (rc, out, err) in enumerate(outs):
        assert rc == 0, "rank %d: %s\n%s" % (rank, out[-1500:], err[-2500:])
        line = json.loads([x for x in out.splitlines() if x.strip().startswith("{")][-1])
        assert line["ok"] and line["rank"] == rank and line["world"] == 2
        assert len(line["cases"]) == 6
        for c in line["cases"]:
            assert c["equals_one_gpu_msm"] and c["collective_calls"] == 1
            assert c["equals_c_restatement"] if c["log2n"] == 16 else c["dlog_identity"]
