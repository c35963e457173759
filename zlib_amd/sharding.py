"""Chunk-range sharding across ranks (one process per GPU).

Chunks are independent given the input, so rank r takes the contiguous range
``shard_chunks(n_chunks, r, world)`` plus a 32 KiB input halo.  The one exchange
step of the path is an all-gather of per-chunk compressed lengths (RCCL on GPUs,
gloo in the CPU tests) followed by an exclusive scan that places every chunk in
the global output.  Inflate shards the same way; there the window in front of a
shard is the previous rank's last 32 KiB of OUTPUT, handed from rank to rank
point to point (``inflate_shard_handover``)."""
from __future__ import annotations

from typing import Tuple

import torch
import torch.distributed as dist


def shard_chunks(n_chunks: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous chunk range [c0, c1) of `rank`."""
    base, extra = divmod(n_chunks, world)
    c0 = rank * base + min(rank, extra)
    return c0, c0 + base + (1 if rank < extra else 0)


def gather_offsets(local_lens: torch.Tensor, n_chunks: int, rank: int, world: int, engine=None, want_total: bool = True):
    """All-gathers per-chunk compressed lengths (int64, on the tensor's device) and
    returns (global byte offsets of this rank's chunks, global total).

    With ``engine`` (a zlib_amd.Engine) and lengths on the GPU the exclusive scan is the library's own
    (``zamd_chunk_offsets_dev``); ``want_total=False`` leaves the total on the device (a 0-d tensor), so
    that a step of a timed loop ends without a copy to the host."""
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
    c0, c1 = shard_chunks(n_chunks, rank, world)
    if engine is not None and all_lens.is_cuda and all_lens.numel():
        all_lens = all_lens.contiguous()
        offs = torch.empty(all_lens.numel() + 1, dtype=torch.int64, device=all_lens.device)
        torch.cuda.current_stream().synchronize()     # the engine works on streams of its own: the lengths must have landed
        engine.chunk_offsets(all_lens.data_ptr(), all_lens.numel(), offs.data_ptr())
        total = offs[-1]
        return offs[c0:c1], (int(total.item()) if want_total else total)
    csum = torch.cumsum(all_lens, 0)
    excl = csum - all_lens
    total = csum[-1] if csum.numel() else torch.zeros((), dtype=torch.int64)
    return excl[c0:c1], (int(total.item()) if want_total else total)


def inflate_shard_handover(engine, rank: int, world: int, win_in: torch.Tensor, win_out: torch.Tensor, *begin_args, **begin_kw) -> None:
    """Sharded inflate of consecutive shards of one stream, one shard per rank.

    ``engine`` has the three steps of include/zamd.h (``inflate_shard_begin`` /
    ``inflate_shard_tail`` / ``inflate_shard_finish``); ``win_in`` / ``win_out`` are
    32768-byte uint8 tensors (device tensors with RCCL, CPU tensors with gloo).
    Every rank decodes at once; only the 32 KiB tail windows travel, rank r -> r+1,
    each a gather over the window that arrived, so the chain costs world-1 small
    sends and not world-1 decodes."""
    engine.inflate_shard_begin(*begin_args, has_head=rank > 0, **begin_kw)
    if rank > 0:
        dist.recv(win_in, src=rank - 1)
        if win_in.is_cuda:          # the engine works on streams of its own: the received bytes must have landed
            torch.cuda.current_stream().synchronize()
    if rank + 1 < world:
        engine.inflate_shard_tail(win_in.data_ptr() if rank > 0 else 0, win_out.data_ptr())
        dist.send(win_out, dst=rank + 1)
    engine.inflate_shard_finish(win_in.data_ptr() if rank > 0 else 0)
