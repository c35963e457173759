"""world_size-2 gloo test of the sharded path's host logic: contiguous chunk-range
sharding, all-gather of per-chunk compressed lengths, exclusive scan to global
offsets.  The per-chunk compression here is the oracle (CPU box); on the GPU the
same sharding code in bench.py drives the engine.  CPU only."""
import os
import socket
import sys
import zlib

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _worker(rank, world, port, n, chunk, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import _oracle as O
    from zlib_amd.sharding import shard_chunks, gather_offsets
    data = O.corpus(n, 4, 5)
    nchunks = (n + chunk - 1) // chunk
    c0, c1 = shard_chunks(nchunks, rank, world)
    lens, pieces = [], []
    for c in range(c0, c1):
        s = c * chunk
        ln = min(chunk, n - s)
        d = min(32768, s)
        out = O.deflate_chunk(data[s - d:s + ln], d, ln, 6, 0, 8, final=(s + ln >= n))
        lens.append(len(out)); pieces.append(out)
    offs, total = gather_offsets(torch.tensor(lens, dtype=torch.int64), nchunks, rank, world)
    # place the shard at its global offset in a shared result (gather to rank 0 for the check)
    gathered = [None] * world
    dist.all_gather_object(gathered, (c0, offs.tolist(), b"".join(pieces)))
    if rank == 0:
        buf = bytearray(total)
        for (cc0, o, blob) in gathered:
            buf[o[0]:o[0] + len(blob)] = blob
        q.put((bytes(buf), total))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_matches_single_rank():
    import _oracle as O
    n, chunk = 5 * 65536 + 1234, 65536
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, chunk, q)) for r in range(2)]
    for p in procs:
        p.start()
    blob, total = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    data = O.corpus(n, 4, 5)
    want, lens = O.deflate_chunked(data, chunk, 6, 0, 8)
    assert total == len(want) and blob == want
    assert zlib.decompress(blob, -15) == data
