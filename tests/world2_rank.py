"""One rank of tests/test_gpu_dist2.py: a REAL world of two for vdf_msm_sharded on one GPU.  Two of these run as fresh child
processes, both on cuda:0 (RCCL refuses two ranks on one device, so the process group is gloo): each builds its point-chunk
shard of the generators (with its fixed-base table), runs the sharded entry point of the C ABI with a callback that stages
its 96-byte partial to the host on the stream it is handed, all-gathers over gloo and copies the gathered partials back on
that stream, and compares the summed point with the one-GPU MSM of the WHOLE vector -- against the C restatement at 2^16
points and against the discrete-log identity at 2^20.  Scalars differ per iteration (a stale partial fails).  Prints one JSON
line.  Test infrastructure: may use oracle/."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

P_ = 0x40000000000000000000000000000000224698FC094CF91B992D30ED00000001
Q_ = 0x40000000000000000000000000000000224698FC0994A8DD8C46EB2100000001


def dlogs(seed, start, n):
    with np.errstate(over="ignore"):
        z = np.uint64((seed * 0xD1342543DE82EF95 + start) & ((1 << 64) - 1)) + np.arange(n, dtype=np.uint64)
        z = z + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return (z ^ (z >> np.uint64(31))) | np.uint64(1)


def sum_s_k(scalars_u64, k):
    n = scalars_u64.shape[0]
    s16 = scalars_u64.view("<u2").reshape(n, 16).astype(np.uint64)
    k16 = k.view("<u2").reshape(n, 4).astype(np.uint64)
    acc = 0
    for a in range(16):
        col = np.ascontiguousarray(s16[:, a])
        for b in range(4):
            acc += int(np.dot(col, k16[:, b])) << (16 * (a + b))
    return acc


def scalar_mul_generator(k, bm):
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


def jac_affine(raw, bm):
    R = 1 << 256
    X, Y, Z = (int.from_bytes(raw[32 * k:32 * k + 32], "little") * pow(R, -1, bm) % bm for k in range(3))
    return None if Z == 0 else (X * pow(Z, -2, bm) % bm, Y * pow(Z, -3, bm) % bm)


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    import vdf_amd
    from vdf_amd.dist import ShardedMsm
    ctx = vdf_amd.Context(0)
    curve = vdf_amd.CURVE_PALLAS
    out_line = {"rank": rank, "world": world, "cases": []}
    calls = {"n": 0}

    def make_gather(partial, gathered):
        def all_gather(dst, src):
            # torch's current stream IS the stream the library produced the partial on (hip.py msm_sharded): the D->H copy is
            # ordered behind the partial, the H->D copy in front of the point sum
            calls["n"] += 1
            host = src.cpu()                                   # 96 bytes, synchronous on the current stream
            parts = [torch.zeros_like(host) for _ in range(world)]
            dist.all_gather(parts, host)                       # gloo, host memory
            dst.copy_(torch.cat(parts).to(dst.device, non_blocking=False))
        return all_gather

    for lg, family in ((16, vdf_amd.GENS_TRY_AND_INCREMENT), (20, vdf_amd.GENS_KNOWN_DLOG)):
        n = 1 << lg
        sh = ShardedMsm(ctx, curve, seed=23, n_total=n, rank=rank, world=world, table=(0, 1), family=family)
        whole = ShardedMsm(ctx, curve, seed=23, n_total=n, rank=0, world=1, table=(0, 1), family=family)      # the one-GPU MSM of the full vector
        partial = torch.zeros(12, dtype=torch.int64, device="cuda")
        gathered = torch.zeros(world * 12, dtype=torch.int64, device="cuda")
        result = torch.zeros(12, dtype=torch.int64, device="cuda")
        single = torch.zeros(12, dtype=torch.int64, device="cuda")
        gather = make_gather(partial, gathered)
        for it in range(3):
            rng = np.random.default_rng(1000 * lg + it)                       # the SAME full vector on every rank
            full = rng.integers(0, 2**64, size=(n, 4), dtype=np.uint64)
            full[:, 3] &= np.uint64(0x3FFFFFFFFFFFFFFF)
            mine = torch.from_numpy(full[sh.start:sh.start + sh.count].view(np.int64).copy()).cuda()
            before = calls["n"]
            sh.run(mine, partial, gathered, gather, out=result)
            ctx.sync()
            whole.local_partial(torch.from_numpy(full.view(np.int64).copy()).cuda(), single)
            ctx.sync()
            got = jac_affine(result.cpu().numpy().view("<u8").tobytes(), P_)
            one_gpu = jac_affine(single.cpu().numpy().view("<u8").tobytes(), P_)
            case = {"log2n": lg, "iteration": it, "collective_calls": calls["n"] - before, "equals_one_gpu_msm": got == one_gpu}
            if lg <= 16:
                from oracle import cref, pasta as o
                L = cref.lib()
                pts = whole.bases.download()
                exp, aff = np.zeros(12, dtype="<u8"), np.zeros(8, dtype="<u8")
                L.ref_msm(curve, cref.p(pts), cref.p(full), n, 0, 8, 0, cref.p(exp))
                L.ref_jac_to_affine(curve, cref.p(exp), cref.p(aff))
                raw = aff.tobytes()
                want = (o.from_mont(int.from_bytes(raw[:32], "little"), P_), o.from_mont(int.from_bytes(raw[32:], "little"), P_))
                case["equals_c_restatement"] = got == want
            else:
                want = scalar_mul_generator(sum_s_k(full, dlogs(23, 0, n)) % Q_, P_)
                case["dlog_identity"] = got == want
            out_line["cases"].append(case)
        sh.bases.free(); whole.bases.free()
    out_line["ok"] = all(c["equals_one_gpu_msm"] and c["collective_calls"] == 1 and c.get("equals_c_restatement", True)
                         and c.get("dlog_identity", True) for c in out_line["cases"])
    dist.barrier()
    dist.destroy_process_group()
    print(json.dumps(out_line))
    sys.exit(0 if out_line["ok"] else 1)


if __name__ == "__main__":
    main()
