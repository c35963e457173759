#!/usr/bin/env python3
"""Benchmark of the chunked-DEFLATE hot path on MI355X.

One step = one pass of the engine over this rank's shard of the synthetic corpus
(inputs already resident in HBM): level-6 (deflate_slow) chunked raw deflate,
1 MiB chunks, 32 KiB dictionary priming, Z_SYNC_FLUSH per chunk — the 64 GiB
mixed-corpus configuration cut to its per-GPU share (8 GiB per GPU, weak
scaling), followed by the batch inflate of what was produced.  With N > 1 ranks
every rank compresses its contiguous chunk range and the per-chunk lengths are
all-gathered over RCCL and scanned into global output offsets inside the timed
step; the inflate leg hands each rank's last 32 KiB of output to the next rank.

`python bench.py --gpus N` with no WORLD_SIZE in the environment starts the N
ranks itself (fresh child processes, before this process touches the GPU) and
relays rank 0's line.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import socket
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
    "l2_text": (0, 1, 2, 0, 8),
    "l3_text": (0, 1, 3, 0, 8),
    # the strategies of BASELINE configs[3] on its corpora
    "l9_image_filtered": (3, 4, 9, 1, 8),
    "l6_lowent_huff": (2, 3, 6, 2, 8),
    "l6_image_rle": (3, 4, 6, 3, 8),
}
METRIC = "deflate GB/s @level 6, 1/2/4/8 GPUs; inflate GB/s; ratio delta vs ref"


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="l6_mixed", choices=sorted(WORKLOADS))
    ap.add_argument("--chunks-per-gpu", type=int, default=int(os.environ.get("ZAMD_BENCH_CHUNKS", "8192")))
    ap.add_argument("--cpu-sample-mib", type=int, default=int(os.environ.get("ZAMD_BENCH_CPU_MIB", "384")))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-inflate", action="store_true", help="skip the batch-inflate leg (reported beside the deflate value)")
    ap.add_argument("--no-serial-pass", action="store_true", help="skip the extra untimed step that times every kernel alone on the GPU")
    ap.add_argument("--backend", default="nccl", choices=("nccl", "gloo"))
    ap.add_argument("--dry", action="store_true",
                    help="no GPU: run the rank launcher, the rendezvous, the length all-gather and the window hand-over on "
                         "synthetic lengths (CPU test of the multi-rank plumbing)")
    return ap.parse_args(argv)


def launch_ranks(args, argv):
    """Parent of a multi-rank run: starts one fresh process per GPU and relays rank 0's output.  Nothing here may
    touch the GPU (no torch.cuda call, no engine): the children initialise it themselves."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=(r == 0)))
    out, _ = procs[0].communicate()
    rc = procs[0].returncode
    for p in procs[1:]:
        p.wait()
        rc = rc or p.returncode
    sys.stdout.write(out)
    sys.stdout.flush()
    return rc


def progress(rank, what):
    """One line per stage on stderr: a run that writes nothing for minutes is taken to be hung."""
    if rank == 0:
        sys.stderr.write("[bench] %s\n" % what)
        sys.stderr.flush()


def host_info():
    model = None
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                model = ln.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    try:
        affinity = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        affinity = os.cpu_count() or 1
    return model, os.cpu_count() or 1, affinity


def dry_run(args, rank, world):
    """The multi-rank plumbing without a GPU: rendezvous, all-gather of lengths, scan, window hand-over."""
    import torch
    import torch.distributed as dist
    from zlib_amd.sharding import gather_offsets, shard_chunks, inflate_shard_handover
    if world > 1:
        dist.init_process_group(args.backend)
    nchunks = args.chunks_per_gpu
    total_chunks = nchunks * world
    c0, c1 = shard_chunks(total_chunks, rank, world)
    lens = torch.arange(c0, c1, dtype=torch.int64) % 977 + 1000          # synthetic chunk lengths
    offs, gtotal = gather_offsets(lens, total_chunks, rank, world)
    want_total = int((torch.arange(total_chunks, dtype=torch.int64) % 977 + 1000).sum())
    want_first = int((torch.arange(c0, dtype=torch.int64) % 977 + 1000).sum())
    ok = gtotal == want_total and (offs.numel() == 0 or int(offs[0]) == want_first)

    class Fake:                     # the three steps of the sharded inflate, on 32 KiB of CPU bytes: tail = head + 1
        def inflate_shard_begin(self, has_head): self.has_head = has_head
        def inflate_shard_tail(self, d_head, d_tail): win_out.copy_((win_in + 1) if self.has_head else torch.ones_like(win_out))
        def inflate_shard_finish(self, d_head): self.done = True
    win_in = torch.zeros(32768, dtype=torch.uint8); win_out = torch.zeros(32768, dtype=torch.uint8)
    fk = Fake()
    if world > 1:
        inflate_shard_handover(fk, rank, world, win_in, win_out)
        ok = ok and fk.done and (rank == 0 or bool((win_in == rank).all()))
        seen = [None] * world
        dist.all_gather_object(seen, ok)
        ok = all(seen)
        ranks_seen = dist.get_world_size()
    else:
        ranks_seen = 1
    if rank == 0:
        print(json.dumps({"metric": METRIC, "value": None, "unit": "GB/s", "n_gpus": world, "dry": True, "backend": args.backend,
                          "ranks_in_group": ranks_seen, "plumbing_ok": bool(ok), "out_bytes_all": gtotal}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0 if ok else 1


def main():
    argv = sys.argv[1:]
    args = parse_args(argv)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args, argv))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE is %d\n" % (args.gpus, world))
        sys.exit(2)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if args.dry:
        sys.exit(dry_run(args, rank, world))

    import torch
    import torch.distributed as dist
    import zlib_amd
    from zlib_amd.sharding import gather_offsets, inflate_shard_handover

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU path")
    if os.environ.get("ZAMD_BENCH_ONE_GPU"):      # rehearsal of N ranks on a one-GPU box: every rank on device 0
        local = 0
    torch.cuda.set_device(local)
    if world > 1:
        dist.init_process_group(args.backend, device_id=torch.device("cuda", local) if args.backend == "nccl" else None)
    kind, seed, level, strategy, mem = WORKLOADS[args.workload]
    nchunks = args.chunks_per_gpu
    n = nchunks * CHUNK
    total_chunks = nchunks * world

    # engine workspace: 156 GiB of the 288 GB of HBM (all of the shard's intermediate data resident, four batches
    # in flight at once); the library's own default is 48 GiB
    ws_gib = int(os.environ.get("ZAMD_BENCH_WS_GIB", "156"))
    eng = zlib_amd.Engine(local, ws_gib << 30)
    lib = eng.lib
    cpu_model, host_cpus, affinity = host_info()
    # worker threads on the host: the CPUs this process may use, at most the share the box gives one GPU's job
    # (OMP_NUM_THREADS is that share where the launcher sets it)
    share = int(os.environ.get("OMP_NUM_THREADS", "0") or 0)
    threads = max(1, min(affinity, share) if share > 0 else affinity)
    # this rank's shard of the corpus: piece i of the global corpus is corpus(kind, seed + i)
    pieces_per_shard = (n + PIECE - 1) // PIECE
    piece0 = rank * pieces_per_shard
    host = torch.empty(n, dtype=torch.uint8, pin_memory=True)
    t0 = time.perf_counter()
    lib.zamd_corpus_fill_pieces(host.data_ptr(), n, kind, seed + piece0, PIECE, threads)
    gen_s = time.perf_counter() - t0
    progress(rank, "corpus generated in %.1f s" % gen_s)
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
    # the deflate call also records block boundaries inside every chunk (8 bytes each, SPLIT_PARTS - 1 per chunk): the
    # inflate leg puts several wavefronts on a chunk where the chunks alone do not fill the GPU (the library decides)
    SPLIT_PARTS = 8
    d_splits = torch.zeros(nchunks * (SPLIT_PARTS - 1), dtype=torch.int64, device="cuda")
    finish = rank == world - 1

    def step():
        eng.record_splits(d_splits.data_ptr(), SPLIT_PARTS)
        total = eng.deflate_shard_dev(d_in, n, head, d_out.data_ptr(), cap, level=level, strategy=strategy, mem_level=mem,
                                      chunk_size=CHUNK, finish=finish, d_chunk_off=d_off.data_ptr())
        lens = d_off[1:] - d_off[:-1]
        if args.backend == "gloo" and world > 1:
            lens = lens.cpu()
        # RCCL all-gather + exclusive scan (the library's zamd_chunk_offsets_dev); the total stays on the device
        offs, gtotal = gather_offsets(lens, total_chunks, rank, world, engine=eng if lens.is_cuda else None, want_total=False)
        return total, gtotal

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    for _ in range(args.warmup):
        step()
    fence()
    progress(rank, "warm-up done")
    t0 = time.perf_counter()
    for _ in range(args.steps):
        total, gtotal = step()
    fence()
    elapsed = max_over_ranks(time.perf_counter() - t0)
    gtotal = int(gtotal.item())
    ms_per_step = 1e3 * elapsed / max(1, args.steps)
    progress(rank, "timed steps done: %.1f ms per step" % ms_per_step)
    in_bytes_all = n * world
    value = in_bytes_all / (elapsed / max(1, args.steps)) / 1e9
    ratio = n / max(1, total)

    # kernel timing (HIP events on the engine's own streams).  In the timed steps up to four batches are in flight on
    # four streams, so an event pair around a launch measures a span that overlaps other kernels.  The roofline figure
    # therefore comes from one more, untimed, step with a single stream: every kernel alone on the GPU, event pairs
    # that add up to the step.
    names = {"chain": 0, "match": 1, "parse": 2, "plan": 3, "layout": 4, "emit": 5, "call": 6, "reach": 10, "first3": 11}
    kt = {nm: eng.timing_ms(i) for nm, i in names.items()}
    dominant = "match" if kt["match"][1] else "parse"     # levels 1-3 have no separate match kernel
    if not kt["match"][1] and kt["plan"][0] > kt["parse"][0]:
        dominant = "plan"                                   # Z_HUFFMAN_ONLY: no search and no parse to speak of
    # chain budgets from 256 on (levels 7-9) search along the four-byte chains: k_reach and k_first3 ran, the matcher is k_match4
    kernel_name = ("k_match4" if kt["reach"][1] else "k_match") if dominant == "match" else "k_plan" if dominant == "plan" else \
                  ("k_fastw" if level == 1 else "k_fastr" if level in (2, 3) else "k_parse")
    roofline = None
    kt_serial = None
    if not args.no_serial_pass:
        eng.set_streams(4, serialize=True)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        step()
        torch.cuda.synchronize()
        serial_ms = 1e3 * (time.perf_counter() - t1)
        eng.set_streams(4, serialize=False)
        progress(rank, "serial pass done: %.1f ms" % serial_ms)
        kt_serial = {nm: eng.timing_ms(i) for nm, i in names.items()}
        sum_ms, launches = kt_serial[dominant]
        if launches:
            avg_ms = sum_ms / launches
            alg_bytes = (n / launches) * (1.0 + 1.0 / ratio)    # N_in + N_out of one launch's batch
            ach = alg_bytes / (avg_ms * 1e-3) / 1e9
            traffic, traffic_src = None, None
            try:                    # HBM bytes of this kernel from the committed PMC passes (the file names the workload and the commit it was taken at)
                prof = json.load(open(os.path.join(ROOT, "profiles", "r03_hbm_traffic.json")))
                ent = [v for k, v in prof["kernels"].items() if k.startswith(kernel_name)][0]
                if prof.get("workload", "l6_mixed") == args.workload:
                    traffic = round(ent["hbm_bytes_per_input_byte"] * (n / launches) / 1e9, 3)
                    traffic_src = "from_profile: profiles/r03_hbm_traffic.json at commit %s (separate rocprofv3 --pmc passes, gfx950 FETCH_SIZE correction)" % prof.get("commit", "?")
            except Exception:
                pass
            roofline = {"bound": "hbm", "kernel": kernel_name, "achieved": round(ach, 3), "peak": 8000.0, "unit": "GB/s",
                        "frac": round(ach / 8000.0, 6), "traffic": traffic, "traffic_unit": "GB per launch", "traffic_source": traffic_src,
                        "algorithmic_gb_per_launch": round(alg_bytes / 1e9, 3),
                        "avg_launch_ms": round(avg_ms, 3), "launches": launches, "sum_launch_ms": round(sum_ms, 3),
                        "serial_step_ms": round(serial_ms, 3),
                        "measured": "one untimed step with the same batches issued to a single stream (every kernel alone on the GPU), HIP events around each launch",
                        "overlapped_span_ms_in_timed_step": round(kt[dominant][0] / max(1, kt[dominant][1]), 3),
                        "whole_path": {"achieved": round(in_bytes_all * (1.0 + 1.0 / ratio) / (ms_per_step * 1e-3) / 1e9 / world, 3), "unit": "GB/s per GPU",
                                       "frac": round(in_bytes_all * (1.0 + 1.0 / ratio) / (ms_per_step * 1e-3) / 1e9 / world / 8000.0, 6)}}

    # ---- inflate leg of the metric (outside the timed deflate region): every rank inflates the shard it just
    # produced, chunk-indexed.  The window of a rank's first chunk is the previous rank's last 32 KiB of OUTPUT and
    # arrives from that rank (point to point), as it would in a real sharded decode.
    inflate = None
    if not args.no_inflate:
        d_back = torch.empty(n, dtype=torch.uint8, device="cuda")
        wdev = "cuda" if args.backend == "nccl" else "cpu"
        win_in = torch.zeros(32768, dtype=torch.uint8, device="cuda"); win_out = torch.zeros(32768, dtype=torch.uint8, device="cuda")

        def inf_step():
            if world == 1:
                eng.inflate_dev(d_out.data_ptr(), d_off.data_ptr(), nchunks, d_back.data_ptr(), n, chunk_size=CHUNK, comp_bytes=int(total),
                                d_splits=d_splits.data_ptr(), parts=SPLIT_PARTS)
            elif wdev == "cuda":
                inflate_shard_handover(eng, rank, world, win_in, win_out, d_out.data_ptr(), int(total), d_off.data_ptr(), nchunks,
                                       d_back.data_ptr(), n, chunk_size=CHUNK)
            else:                   # gloo: the windows travel through host memory
                eng.inflate_shard_begin(d_out.data_ptr(), int(total), d_off.data_ptr(), nchunks, d_back.data_ptr(), n, chunk_size=CHUNK, has_head=rank > 0)
                if rank > 0:
                    h = torch.empty(32768, dtype=torch.uint8); dist.recv(h, src=rank - 1); win_in.copy_(h)
                if rank + 1 < world:
                    eng.inflate_shard_tail(win_in.data_ptr() if rank > 0 else 0, win_out.data_ptr())
                    dist.send(win_out.cpu(), dst=rank + 1)
                eng.inflate_shard_finish(win_in.data_ptr() if rank > 0 else 0)
        inf_step()
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            inf_step()
        fence()
        inf_elapsed = max_over_ranks(time.perf_counter() - t0)
        ok = bool(torch.equal(d_back, d_buf[32768:]))
        if world > 1:               # every rank's shard must have come back
            flags = [None] * world
            dist.all_gather_object(flags, ok)
            ok = all(flags)
        dec_ms, dec_n = eng.timing_ms(7)
        res_ms, _ = eng.timing_ms(8)
        inf_s = inf_elapsed / max(1, args.steps)
        inflate = {"value": round(n * world / inf_s / 1e9, 4), "unit": "GB/s of inflated output", "ms_per_step": round(1e3 * inf_s, 3),
                   "verified_equal_to_input": ok, "decode_ms": round(dec_ms, 3), "resolve_ms": round(res_ms, 3),
                   "window_handover": "rank r sends its last 32 KiB of output to rank r+1" if world > 1 else None,
                   "roofline": {"bound": "hbm", "kernel": "k_inf_decode", "achieved": round((n + total) / (dec_ms * 1e-3) / 1e9, 3) if dec_ms else None,
                                "peak": 8000.0, "unit": "GB/s", "frac": round((n + total) / (dec_ms * 1e-3) / 1e9 / 8000.0, 6) if dec_ms else None}}
        del d_back
        progress(rank, "inflate leg done")

    if rank == 0:
        # cheap self-check outside the timed region: the first chunks inflate to the input
        import zlib
        off = d_off[:5].cpu().tolist()
        blob = bytes(d_out[: off[4]].cpu().numpy().tobytes())
        back = zlib.decompressobj(-15).decompress(blob)
        assert back == bytes(host[: 4 * CHUNK].numpy().tobytes()), "self-check failed: output does not inflate to the input"

    cpu = None
    if rank == 0 and not args.no_cpu_baseline:
        exe = os.path.join(ROOT, "oracle", "_ref", "cpu_baseline")
        sample = args.cpu_sample_mib << 20
        cores = threads
        if os.path.exists(exe):
            cmd = [exe, str(kind), str(seed), str(sample), str(CHUNK), str(level), str(strategy), str(mem), str(cores), str(PIECE)]
            try:
                r = json.loads(subprocess.run(cmd, capture_output=True, text=True, timeout=600, check=True).stdout)
                cpu = {"value": round(r["bytes"] / r["seconds"] / 1e9, 4), "unit": "GB/s", "cores": cores, "kind": r["kind"],
                       "cpu_model": cpu_model, "host_cpus": host_cpus, "cpus_allowed": affinity,
                       "cores_note": "threads used = the CPUs this job may use (its share of the host: OMP_NUM_THREADS where set, else its affinity mask)",
                       "sample": "first %d MiB of the same corpus, same chunk recipe, reference zlib %s built from source, one chunk per task"
                                 % (args.cpu_sample_mib, r["zlib"]),
                       "ratio": round(r["bytes"] / r["out_bytes"], 4),
                       "inflate_value": round(r["bytes"] / r["inflate_seconds"] / 1e9, 4) if r.get("inflate_seconds") else None,
                       "single_stream_ratio": round(r["single_bytes"] / r["single_out"], 5) if r.get("single_out") else None,
                       "chunked_ratio_same_prefix": round(r["single_bytes"] / r["chunked_out_same_prefix"], 5) if r.get("chunked_out_same_prefix") else None,
                       "ratio_delta_vs_ref": round(r["single_out"] / r["chunked_out_same_prefix"] - 1.0, 6) if r.get("chunked_out_same_prefix") else None}
            except Exception as ex:          # the baseline is reported, never required
                cpu = {"value": None, "unit": "GB/s", "cores": cores, "kind": "reference", "cpu_model": cpu_model, "sample": "failed: %r" % (ex,)}

    if rank == 0:
        line = {
            "metric": METRIC if args.workload == "l6_mixed" else "deflate GB/s",
            "value": round(value, 4), "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": "%s: level %d strategy %d memLevel %d, 1 MiB chunks + 32 KiB dictionary + Z_SYNC_FLUSH, %d chunks (%.1f GiB) per GPU, "
                                   "corpus kind %d generated in 16 MiB pieces, piece i seeded %d+i (the configuration's seed-%d stream cut into "
                                   "independently seeded pieces so that every rank can generate its own shard)"
                                   % (args.workload, level, strategy, mem, nchunks, n / 2**30, kind, seed, seed),
                       "chunk_size": CHUNK, "chunks_per_gpu": nchunks, "sharding": "contiguous chunk ranges, RCCL all-gather of lengths",
                       "engine_workspace_gib": ws_gib, "backend": args.backend},
            "ratio": round(ratio, 4), "out_bytes_rank0": int(total), "out_bytes_all": int(gtotal),
            "kernel_ms_last_step_overlapped_spans": {k: round(v[0], 3) for k, v in kt.items()},
            "kernel_ms_serial_pass": {k: {"sum_ms": round(v[0], 3), "launches": v[1]} for k, v in kt_serial.items()} if kt_serial else None,
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
