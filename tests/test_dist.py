"""CPU tests of the multi-GPU path: point-chunk sharding + all-gather of 96-byte
partials + local point-sum, exercised with world_size 2 and 3 over gloo.  The per-rank compute is
injected (the oracle's C restatement stands in for the HIP context, which needs a GPU); what is
under test is vdf_amd.dist's partitioning, buffer layout and collective call pattern."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import pasta as o
from vdf_amd.dist import ShardedMsm, shard_range


def test_shard_range_partitions_exactly():
    for n in (0, 1, 7, 1 << 20, (1 << 24) + 5):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0
            assert sum(c for _, c in spans) == n
            for (s0, c0), (s1, _) in zip(spans, spans[1:]):
                assert s0 + c0 == s1
            assert max(c for _, c in spans) - min(c for _, c in spans) <= 1
    with pytest.raises(ValueError):
        shard_range(10, 2, 2)


class _OracleBases:
    def __init__(self, pts):
        self.pts = pts

    def precompute(self, *a):
        pass


class OracleBackend:
    """Test-only stand-in with the Context surface ShardedMsm uses."""

    def __init__(self):
        from oracle import cref
        self.c = cref
        self.L = cref.lib()

    def bases_generate(self, curve, seed, n, start=0):
        pts = np.zeros((n, 8), dtype="<u8")
        self.L.ref_synthetic_bases(curve, seed, start, n, self.c.p(pts))
        b = _OracleBases(pts)
        b.curve = curve
        return b

    def msm(self, bases, scalars, n=None, is_mont=False, out=None):
        sc = np.ascontiguousarray(scalars.numpy().view("<u8"))
        res = np.zeros(12, dtype="<u8")
        self.L.ref_msm(bases.curve, self.c.p(bases.pts), self.c.p(sc), n, int(is_mont), 1, 0, self.c.p(res))
        out.copy_(torch.from_numpy(res.view(np.int64)))
        return out

    def msm_sharded(self, bases, scalars, n, rank, world, all_gather, partial, gathered, out, is_mont=False,
                    always_gather=False, offset=0):
        """The protocol of include/vdf_hip.h vdf_msm_sharded with the C restatement as the device: partial into the
        caller's buffer, the caller's collective called once on its two buffers, then the sum of the world partials."""
        assert 0 <= rank < world and partial.numel() == 12 and gathered.numel() == 12 * world
        self.calls = getattr(self, "calls", 0) + 1
        self.msm(bases, scalars, n=n, is_mont=is_mont, out=partial)
        all_gather(gathered, partial)
        pt = self.point_sum(bases.curve, gathered, world)
        m = o.curve_base_modulus(bases.curve)
        enc = [0, 0, 0] if pt is None else [o.to_mont(pt[0], m), o.to_mont(pt[1], m), o.to_mont(1, m)]
        out.copy_(torch.from_numpy(np.frombuffer(b"".join(int(v).to_bytes(32, "little") for v in enc), dtype=np.int64).copy()))
        return pt

    def point_sum(self, curve, points, n, out=None):
        m = o.curve_base_modulus(curve)
        arr = points.numpy().view("<u8").reshape(n, 3, 4)
        acc = None
        for k in range(n):
            X, Y, Z = (o.from_mont(int.from_bytes(arr[k, j].tobytes(), "little"), m) for j in range(3))
            if Z:
                zi = pow(Z, -1, m)
                acc = o.pt_add(acc, (X * zi * zi % m, Y * zi ** 3 % m), m)
        return acc


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_total, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        curve = o.CURVE_PALLAS
        sm = o.curve_scalar_modulus(curve)
        sh = ShardedMsm(OracleBackend(), curve, seed=7, n_total=n_total, rank=rank, world=world, table=None)
        all_scalars = [o.rand_fe(5, i, sm) for i in range(n_total)]          # same on every rank
        mine = all_scalars[sh.start: sh.start + sh.count]
        sc = np.frombuffer(b"".join(int(v).to_bytes(32, "little") for v in mine), dtype="<u8").reshape(-1, 4).copy()
        sc_t = torch.from_numpy(sc.view(np.int64))
        partial = torch.zeros(12, dtype=torch.int64)
        gathered = torch.zeros(world * 12, dtype=torch.int64)

        def all_gather(dst, src):
            dist.all_gather_into_tensor(dst, src)

        got = sh.run(sc_t, partial, gathered, all_gather)
        exp = o.msm_by_dlog(all_scalars, curve, 7)
        # the same MSM through the sharded entry point's protocol (what the product's Context.msm_sharded binds)
        result = torch.zeros(12, dtype=torch.int64)
        got2 = sh.run(sc_t, partial, gathered, all_gather, out=result)
        X, Y, Z = (o.from_mont(int.from_bytes(result.numpy().view("<u8")[4 * j:4 * j + 4].tobytes(), "little"), o.P) for j in range(3))
        ret[rank] = (got == exp and got2 == exp and sh.backend.calls == 1 and (X, Y, Z) == (exp[0], exp[1], 1), sh.start, sh.count)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_total", [(2, 600), (3, 1001)])
def test_sharded_msm_gloo(world, n_total):
    mgr = mp.Manager()
    ret = mgr.dict()
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n_total, ret), nprocs=world, join=True)
    assert len(ret) == world
    covered = 0
    for r in range(world):
        ok, start, count = ret[r]
        assert ok, f"rank {r}: sharded MSM differs from the single MSM"
        assert start == covered
        covered += count
    assert covered == n_total
