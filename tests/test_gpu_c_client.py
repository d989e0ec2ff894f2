"""GPU: the plain-C client of both C ABIs (examples/prove_chain.c: no Python, no torch, no HIP headers) RUN as a fresh child
process -- the flow the reference's own test walks (src/nova/proof.rs:403-451: eval -> prove_recursively -> verify ->
compress -> verify) through the boundary exactly as a cgo / Rust FFI host would drive it.  Its digest and the bytes of its
compressed proof must equal (a) the committed vector tests/golden/vectors.json "wire_ivc_t2_reference" (made by the oracle
alone on the CPU) and (b) what vdf_amd.nova (ctypes) produces for the same inputs at a second size."""
import hashlib
import os
import subprocess

import pytest

from oracle import pasta as o
from test_gpu_nova import make
from vdf_amd.nova import NovaVDFProof

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "examples", "prove_chain")


def run_client(log2t, steps, x_int, i0, out_path):
    assert os.path.exists(EXE), "examples/prove_chain is built by vdf_amd/csrc/Makefile (all)"
    x_hex = int(o.to_mont(x_int, o.Q)).to_bytes(32, "little").hex()          # a vdf_fe: 4 x u64 little-endian, Montgomery form
    # a fresh child process (never an exec of this one: the test process has initialised the GPU)
    r = subprocess.run([EXE, str(log2t), str(steps), x_hex, str(i0), out_path], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = dict(ln.split(": ", 1) for ln in r.stdout.splitlines() if ": " in ln)
    assert lines["verify"].startswith("true") and lines["verify (compressed)"].startswith("true")
    assert "decoded and verified: true" in r.stdout
    wire = open(out_path, "rb").read()
    fnv = 0xCBF29CE484222325
    for b in wire:
        fnv = ((fnv ^ b) * 0x100000001B3) & ((1 << 64) - 1)
    assert lines["wire fnv1a64"] == "%016x" % fnv                             # the file IS what the client hashed
    return int(lines["digest"], 16), wire


def test_c_client_reproduces_the_committed_vector(golden, tmp_path):
    g = golden["wire_ivc_t2_reference"]
    assert g["t"] == 2 and not g["bound"]
    digest, wire = run_client(1, g["steps"], o.rand_fe(g["seed"], 0, o.Q), g["i0"], str(tmp_path / "wire.bin"))
    assert digest == int(g["params"], 16)
    assert len(wire) == g["compressed_proof_len"] and hashlib.sha256(wire).hexdigest() == g["compressed_proof_sha256"]


def test_c_client_and_the_ctypes_host_agree(ctx, tmp_path):
    t, n, seed, i0 = 64, 3, 77, 0
    digest, wire = run_client(6, n, o.rand_fe(seed, 0, o.Q), i0, str(tmp_path / "wire.bin"))
    pp, z0, circuits, initial, init_ints = make(ctx, t, n, seed=seed, i0=i0)
    assert digest == pp.digest()
    proof = NovaVDFProof.prove_recursively(pp, circuits, t, z0)
    assert proof.compress(pp).serialize() == wire
    proof.free(); pp.free()
