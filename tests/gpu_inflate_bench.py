"""Manual tool: times batch inflate of the engine's own level-6 stream.  usage: gpu_inflate_bench.py [MiB] [kind]"""
import ctypes as C
import sys
import time

import zlib_amd

mib = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
kind = int(sys.argv[2]) if len(sys.argv) > 2 else 4
n = mib << 20
chunk = 1 << 20
eng = zlib_amd.Engine(0)
data = zlib_amd.corpus(n, kind, 5, piece_bytes=16 << 20, threads=16)
nch = n // chunk
cap = eng.lib.zamd_deflate_bound(n, chunk, 8)
d_in, d_out, d_off, d_back = eng.alloc(n), eng.alloc(cap), eng.alloc(8 * (nch + 1)), eng.alloc(n)
d_in.upload(data)
total = eng.deflate_dev(d_in.ptr, n, d_out.ptr, cap, level=6, d_chunk_off=d_off.ptr)
print("deflate: %d -> %d (ratio %.3f)" % (n, total, n / total))
for it in range(3):
    t0 = time.time()
    eng.inflate_dev(d_out.ptr, d_off.ptr, nch, d_back.ptr, n)
    dt = time.time() - t0
    dec, res = eng.timing_ms(7)[0], eng.timing_ms(8)[0]
    print("inflate: %.1f ms wall = %.2f GB/s out; decode %.1f ms, resolve %.1f ms" % (dt * 1e3, n / dt / 1e9, dec, res))
back = d_back.download(n)
print("round trip", "OK" if back == data else "MISMATCH")
