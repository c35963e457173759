"""Manual tool: times batch inflate of the engine's own stream, one wavefront to a chunk and several.
usage: gpu_inflate_bench.py [MiB] [kind] [level] [strategy] [parts ...]"""
import ctypes as C
import sys
import time

import zlib_amd

mib = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
kind = int(sys.argv[2]) if len(sys.argv) > 2 else 4
level = int(sys.argv[3]) if len(sys.argv) > 3 else 6
strategy = int(sys.argv[4]) if len(sys.argv) > 4 else 0
parts_list = [int(x) for x in sys.argv[5:]] or [1, 2, 4, 8]
n = mib << 20
chunk = 1 << 20
eng = zlib_amd.Engine(0)
data = zlib_amd.corpus(n, kind, 5, piece_bytes=16 << 20, threads=16)
nch = n // chunk
cap = eng.lib.zamd_deflate_bound(n, chunk, 8)
d_in, d_out, d_off, d_back = eng.alloc(n), eng.alloc(cap), eng.alloc(8 * (nch + 1)), eng.alloc(n)
d_in.upload(data)
for parts in parts_list:
    d_sp = eng.alloc(8 * nch * max(parts - 1, 1))
    if parts > 1:
        eng.record_splits(d_sp.ptr, parts)
    total = eng.deflate_dev(d_in.ptr, n, d_out.ptr, cap, level=level, strategy=strategy, d_chunk_off=d_off.ptr)
    print("parts %d: deflate %d -> %d (ratio %.3f)" % (parts, n, total, n / total), flush=True)
    eng.lib.zamd_dev_memset(eng.handle, d_back.ptr, 0, n)
    for it in range(3):
        t0 = time.time()
        eng.inflate_dev(d_out.ptr, d_off.ptr, nch, d_back.ptr, n, comp_bytes=total, d_splits=d_sp.ptr if parts > 1 else 0, parts=parts)
        dt = time.time() - t0
        dec, res = eng.timing_ms(7)[0], eng.timing_ms(8)[0]
        print("  inflate: %.1f ms wall = %.2f GB/s out; decode %.1f ms, resolve %.1f ms" % (dt * 1e3, n / dt / 1e9, dec, res), flush=True)
    back = d_back.download(n)
    if back == data:
        print("  round trip OK", flush=True)
    else:
        import numpy as np
        a, b = np.frombuffer(back, dtype=np.uint8), np.frombuffer(data, dtype=np.uint8)
        bad = np.flatnonzero(a != b)
        print("  round trip MISMATCH: %d bytes differ, first at %d (chunk %d + %d), last at %d" % (
            len(bad), bad[0], bad[0] // chunk, bad[0] % chunk, bad[-1]), flush=True)
        print("  differing bytes per chunk (first 8 chunks with any):", [(int(c), int(k)) for c, k in zip(*np.unique(bad // chunk, return_counts=True))][:8], flush=True)
    d_sp.free()
