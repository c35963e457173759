#!/usr/bin/env python3
"""Benchmark of the chunked-DEFLATE hot path on MI355X.

One step = one pass of the engine over this rank's shard of the synthetic corpus
(inputs already resident in HBM): level-6 (deflate_slow) chunked raw deflate,
1 MiB chunks, 32 KiB dictionary priming, Z_SYNC_FLUSH per chunk — the 64 GiB
mixed-corpus configuration cut to its per-GPU share (8 GiB per GPU, weak
scaling), followed by the batch inflate of what was produced.  With N > 1 ranks every rank compresses its contiguous chunk
range and the per-chunk lengths are all-gathered over RCCL and scanned into
global output offsets inside the timed step.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CHUNK = 1 << 20
PIECE = 16 << 20
WORKLOADS = {
    # name: (corpus kind, seed, level, strategy, mem_level)
    "l6_mixed": (4, 5, 6, 0, 8),      # the headline configuration: "deflate GB/s @level 6"
    "l9_lowent": (2, 3, 9, 0, 8),
    "l6_text": (0, 1, 6, 0, 8),
    "l1_text": (0, 1, 1, 0, 8),       # deflate_fast
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="l6_mixed", choices=sorted(WORKLOADS))
    ap.add_argument("--chunks-per-gpu", type=int, default=int(os.environ.get("ZAMD_BENCH_CHUNKS", "8192")))
    ap.add_argument("--cpu-sample-mib", type=int, default=int(os.environ.get("ZAMD_BENCH_CPU_MIB", "384")))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-inflate", action="store_true", help="skip the batch-inflate leg (reported beside the deflate value)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import zlib_amd
    from zlib_amd.sharding import gather_offsets

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU path")
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    kind, seed, level, strategy, mem = WORKLOADS[args.workload]
    nchunks = args.chunks_per_gpu
    n = nchunks * CHUNK
    total_chunks = nchunks * world

    # engine workspace: 144 GiB of the 288 GB of HBM (all of the shard's intermediate data resident, four batches
    # in flight at once); the library's own default is 48 GiB
    ws_gib = int(os.environ.get("ZAMD_BENCH_WS_GIB", "144"))
    eng = zlib_amd.Engine(local, ws_gib << 30)
    lib = eng.lib
    # this rank's shard of the corpus: piece i of the global corpus is corpus(kind, seed + i)
    import ctypes as C
    threads = min(16, os.cpu_count() or 1)
    pieces_per_shard = (n + PIECE - 1) // PIECE
    piece0 = rank * pieces_per_shard
    host = torch.empty(n, dtype=torch.uint8, pin_memory=True)
    t0 = time.perf_counter()
    lib.zamd_corpus_fill_pieces(host.data_ptr(), n, kind, seed + piece0, PIECE, threads)
    gen_s = time.perf_counter() - t0
    head = 0
    d_buf = torch.empty(32768 + n, dtype=torch.uint8, device="cuda")
    if rank > 0:
        prev = torch.empty(PIECE, dtype=torch.uint8)
        lib.zamd_corpus_fill(prev.data_ptr(), PIECE, kind, seed + piece0 - 1)
        d_buf[:32768].copy_(prev[-32768:])
        head = 32768
    d_buf[32768:].copy_(host, non_blocking=False)
    d_in = d_buf.data_ptr() + 32768
    cap = lib.zamd_deflate_bound(n, CHUNK, mem)
    d_out = torch.empty(cap, dtype=torch.uint8, device="cuda")
    d_off = torch.empty(nchunks + 1, dtype=torch.int64, device="cuda")
    finish = rank == world - 1

    def step():
        total = eng.deflate_shard_dev(d_in, n, head, d_out.data_ptr(), cap, level=level, strategy=strategy, mem_level=mem,
                                      chunk_size=CHUNK, finish=finish, d_chunk_off=d_off.data_ptr())
        lens = d_off[1:] - d_off[:-1]
        offs, gtotal = gather_offsets(lens, total_chunks, rank, world)   # RCCL all-gather + exclusive scan
        return total, gtotal

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        total, gtotal = step()
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = 1e3 * elapsed / max(1, args.steps)
    in_bytes_all = n * world
    value = in_bytes_all / (elapsed / max(1, args.steps)) / 1e9

    # kernel timing of the last call (HIP events on the engine's own streams)
    names = ["chain", "match", "parse", "plan", "layout", "emit", "call"]
    kt = {nm: eng.timing_ms(i) for i, nm in enumerate(names)}
    dominant = "match" if kt["match"][1] else "parse"     # levels 1-3 have no separate match kernel
    match_ms, match_launches = kt[dominant]
    ratio = n / max(1, total)
    roofline = None
    if match_launches:
        avg_ms = match_ms / match_launches
        alg_bytes = (n / match_launches) * (1.0 + 1.0 / ratio)    # N_in + N_out of one launch's batch
        ach = alg_bytes / (avg_ms * 1e-3) / 1e9
        # HBM traffic per launch from the committed PMC passes (FETCH_SIZE / WRITE_SIZE, separate
        # rocprofv3 runs, gfx950 correction applied): bytes per input byte x the input bytes of one launch
        traffic = None
        try:
            prof = json.load(open(os.path.join(ROOT, "profiles", "r01_hbm_traffic.json")))
            per_byte = [v["hbm_bytes_per_input_byte"] for k, v in prof["kernels"].items() if k.startswith("k_match")][0]
            traffic = round(per_byte * (n / match_launches) / 1e9, 3) if dominant == "match" else None    # no counter pass for k_fast
        except Exception:
            pass
        roofline = {"bound": "hbm", "kernel": "k_match" if dominant == "match" else "k_fast", "achieved": round(ach, 3), "peak": 8000.0, "unit": "GB/s",
                    "frac": round(ach / 8000.0, 6), "traffic": traffic, "traffic_unit": "GB per launch",
                    "algorithmic_gb_per_launch": round(alg_bytes / 1e9, 3),
                    "avg_launch_ms": round(avg_ms, 3), "launches": match_launches,
                    "note": "the engine keeps up to four batches in flight on four streams, so launches of this kernel overlap "
                            "each other and the other kernels; a launch's duration is its span, not its share of the GPU"}

    # ---- inflate leg of the metric (outside the timed deflate region): every rank inflates the shard it
    # just produced, chunk-indexed, window of its first chunk = the 32 KiB in front of the shard
    inflate = None
    if not args.no_inflate:
        d_back = torch.empty(n, dtype=torch.uint8, device="cuda")
        def inf_step():
            eng.inflate_dev(d_out.data_ptr(), d_off.data_ptr(), nchunks, d_back.data_ptr(), n, chunk_size=CHUNK,
                            comp_bytes=int(total), d_head_window=d_buf.data_ptr() if head else 0)
        inf_step()
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            inf_step()
        fence()
        inf_elapsed = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([inf_elapsed], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            inf_elapsed = float(t.item())
        ok = bool(torch.equal(d_back, d_buf[32768:]))
        dec_ms, dec_n = eng.timing_ms(7)
        res_ms, _ = eng.timing_ms(8)
        inf_s = inf_elapsed / max(1, args.steps)
        inflate = {"value": round(n * world / inf_s / 1e9, 4), "unit": "GB/s of inflated output", "ms_per_step": round(1e3 * inf_s, 3),
                   "verified_equal_to_input": ok, "decode_ms": round(dec_ms, 3), "resolve_ms": round(res_ms, 3),
                   "roofline": {"bound": "hbm", "kernel": "k_inf_decode", "achieved": round((n + total) / (dec_ms * 1e-3) / 1e9, 3) if dec_ms else None,
                                "peak": 8000.0, "unit": "GB/s", "frac": round((n + total) / (dec_ms * 1e-3) / 1e9 / 8000.0, 6) if dec_ms else None}}
        del d_back

    if rank == 0:
        # cheap self-check outside the timed region: the first chunks inflate to the input
        import zlib
        off = d_off[:5].cpu().tolist()
        blob = bytes(d_out[: off[4]].cpu().numpy().tobytes())
        back = zlib.decompressobj(-15).decompress(blob)
        assert back == bytes(host[: 4 * CHUNK].numpy().tobytes()), "self-check failed: output does not inflate to the input"

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        exe = os.path.join(ROOT, "oracle", "_ref", "cpu_baseline")
        sample = args.cpu_sample_mib << 20
        cores = min(16, os.cpu_count() or 1)
        if os.path.exists(exe):
            cmd = [exe, str(kind), str(seed), str(sample), str(CHUNK), str(level), str(strategy), str(mem), str(cores), str(PIECE)]
            try:
                r = json.loads(subprocess.run(cmd, capture_output=True, text=True, timeout=600, check=True).stdout)
                cpu = {"value": round(r["bytes"] / r["seconds"] / 1e9, 4), "unit": "GB/s", "cores": cores, "kind": r["kind"],
                       "sample": "first %d MiB of the same corpus, same chunk recipe, reference zlib %s built from source, one chunk per task"
                                 % (args.cpu_sample_mib, r["zlib"]),
                       "ratio": round(r["bytes"] / r["out_bytes"], 4),
                       "inflate_value": round(r["bytes"] / r["inflate_seconds"] / 1e9, 4) if r.get("inflate_seconds") else None,
                       "single_stream_ratio": round(r["single_bytes"] / r["single_out"], 5) if r.get("single_out") else None,
                       "chunked_ratio_same_prefix": round(r["single_bytes"] / r["chunked_out_same_prefix"], 5) if r.get("chunked_out_same_prefix") else None,
                       "ratio_delta_vs_ref": round(r["single_out"] / r["chunked_out_same_prefix"] - 1.0, 6) if r.get("chunked_out_same_prefix") else None}
            except Exception as ex:          # the baseline is reported, never required
                cpu = {"value": None, "unit": "GB/s", "cores": cores, "kind": "reference", "sample": "failed: %r" % (ex,)}

    if rank == 0:
        line = {
            "metric": "deflate GB/s @level 6, 1/2/4/8 GPUs; inflate GB/s; ratio delta vs ref" if args.workload == "l6_mixed" else "deflate GB/s",
            "value": round(value, 4), "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": "%s: level %d strategy %d memLevel %d, 1 MiB chunks + 32 KiB dictionary + Z_SYNC_FLUSH, %d chunks (%.1f GiB) per GPU, "
                                   "corpus kind %d in 16 MiB pieces seeded %d+i" % (args.workload, level, strategy, mem, nchunks, n / 2**30, kind, seed),
                       "chunk_size": CHUNK, "chunks_per_gpu": nchunks, "sharding": "contiguous chunk ranges, RCCL all-gather of lengths",
                       "engine_workspace_gib": ws_gib},
            "ratio": round(ratio, 4), "out_bytes_rank0": int(total), "out_bytes_all": int(gtotal),
            "kernel_ms_last_step": {k: round(v[0], 3) for k, v in kt.items()},
            "corpus_gen_s": round(gen_s, 2),
            "inflate": inflate,
            "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()
