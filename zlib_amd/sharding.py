"""Chunk-range sharding across ranks (one process per GPU).

Chunks are independent given the input, so rank r takes the contiguous range
``shard_chunks(n_chunks, r, world)`` plus a 32 KiB input halo.  The one exchange
step of the path is an all-gather of per-chunk compressed lengths (RCCL on GPUs,
gloo in the CPU tests) followed by an exclusive scan that places every chunk in
the global output."""
from __future__ import annotations

from typing import Tuple

import torch
import torch.distributed as dist


def shard_chunks(n_chunks: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous chunk range [c0, c1) of `rank`."""
    base, extra = divmod(n_chunks, world)
    c0 = rank * base + min(rank, extra)
    return c0, c0 + base + (1 if rank < extra else 0)


def gather_offsets(local_lens: torch.Tensor, n_chunks: int, rank: int, world: int) -> Tuple[torch.Tensor, int]:
    """All-gathers per-chunk compressed lengths (int64, on the tensor's device) and
    returns (global byte offsets of this rank's chunks, global total)."""
    if world == 1:
        all_lens = local_lens
    else:
        counts = [shard_chunks(n_chunks, r, world) for r in range(world)]
        width = max(c1 - c0 for c0, c1 in counts)
        padded = torch.zeros(width, dtype=torch.int64, device=local_lens.device)
        padded[: local_lens.numel()] = local_lens
        out = torch.empty(world * width, dtype=torch.int64, device=local_lens.device)
        dist.all_gather_into_tensor(out, padded)
        all_lens = torch.cat([out[r * width: r * width + (c1 - c0)] for r, (c0, c1) in enumerate(counts)])
    csum = torch.cumsum(all_lens, 0)
    excl = csum - all_lens
    c0, c1 = shard_chunks(n_chunks, rank, world)
    return excl[c0:c1], int(csum[-1].item()) if csum.numel() else 0
