"""Point-chunk-sharded MSM across the GPUs of one node.

One process per GPU.  Rank g owns generators [start_g, start_g + count_g) permanently (fixed at
setup, with their fixed-base table) and the matching slice of scalars; it runs the single-GPU
Pippenger on its slice and contributes ONE 96-byte Jacobian partial.  Elliptic-curve addition is
not an RCCL reduction operator, so the exchange is an all-gather of the partials (world x 96 B
over xGMI: latency-bound, never bandwidth-bound) followed by a local point_sum kernel on every
rank.  No other data-path collective exists; raw buckets are never shipped.
"""
from __future__ import annotations

from typing import Callable, Tuple


def shard_range(n_total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous balanced partition of [0, n_total): returns (start, count) of `rank`."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    base, rem = divmod(n_total, world)
    start = rank * base + min(rank, rem)
    return start, base + (1 if rank < rem else 0)


class ShardedMsm:
    """`backend` is a vdf_amd.Context in the product; anything with the same
    bases_generate / msm / point_sum surface can be injected by a test."""

    def __init__(self, backend, curve: int, seed: int, n_total: int, rank: int, world: int,
                 table: Tuple[int, int] | None = (16, 1), family: int = 0, bases=None):
        self.backend, self.curve, self.rank, self.world = backend, curve, rank, world
        self.n_total = n_total
        self.start, self.count = shard_range(n_total, rank, world)
        # family 0: [k_i]G (known discrete logs), 1: try-and-increment; both are index-addressed, so a rank
        # generates exactly its own range.  `bases`: reuse another instance's generators (and table) -- several
        # contexts of one device can run MSMs over the same table concurrently.
        if bases is not None:
            self.bases = bases
            return
        if family:
            self.bases = backend.bases_generate(curve, seed, self.count, start=self.start, family=family)
        else:
            self.bases = backend.bases_generate(curve, seed, self.count, start=self.start)
        if table is not None and self.count:
            self.bases.precompute(*table)

    def local_partial(self, scalars_local, out, is_mont: bool = False):
        """MSM of this rank's slice into `out` (12 x int64 / uint64 buffer)."""
        return self.backend.msm(self.bases, scalars_local, n=self.count, is_mont=is_mont, out=out)

    def combine(self, gathered, out=None):
        """Sum of the world partials (gathered: world x 12 words, rank-major)."""
        return self.backend.point_sum(self.curve, gathered, self.world, out=out)

    def run(self, scalars_local, partial_buf, gathered_buf, all_gather: Callable, out=None, is_mont: bool = False,
            always_gather: bool = False):
        """One sharded MSM.  all_gather(dst, src) fills dst (world x 12) from every rank's src (12).
        always_gather: take the collective path even for world == 1 (rehearsal of the N > 1 code on one GPU)."""
        if self.world == 1 and not always_gather:
            # one rank: its partial is the result (nothing to gather, nothing to sum)
            return self.local_partial(scalars_local, partial_buf if out is None else out, is_mont=is_mont)
        if out is not None and hasattr(self.backend, "msm_sharded"):
            # the product path: one C-ABI call (include/vdf_hip.h vdf_msm_sharded) that runs the partial, calls the
            # collective handed in here, and sums -- what a non-Python host does with an RCCL communicator
            return self.backend.msm_sharded(self.bases, scalars_local, self.count, self.rank, self.world, all_gather,
                                            partial_buf, gathered_buf, out, is_mont=is_mont, always_gather=always_gather)
        self.local_partial(scalars_local, partial_buf, is_mont=is_mont)
        all_gather(gathered_buf, partial_buf)
        return self.combine(gathered_buf, out=out)
