"""Manual tool: deflate() of one large buffer against the same bytes cut into T streams compressed by T threads at once
(what overlapping the batches of one call could gain at most).  usage: gpu_api_pair_bench.py [MiB] [threads]"""
import ctypes as C
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _zapi as Z
import zlib_amd

mib = int(sys.argv[1]) if len(sys.argv) > 1 else 3584
nt = int(sys.argv[2]) if len(sys.argv) > 2 else 2
n = mib << 20
data = zlib_amd.corpus(n, 4, 5, piece_bytes=16 << 20, threads=16)
src = (C.c_char * n).from_buffer(data)
l = Z.lib()
cap = l.compressBound(n)
out = C.create_string_buffer(cap)


def one(off, cnt, res, k):
    s = Z.ZStream()
    assert l.deflateInit2_(C.byref(s), 6, 8, 31, 8, 0, Z.VERSION, Z.SZ) == 0
    s.next_in, s.avail_in, s.next_out, s.avail_out = C.addressof(src) + off, cnt, C.addressof(out) + l.compressBound(off), l.compressBound(cnt)
    rc = l.deflate(C.byref(s), Z.Z_FINISH)
    assert rc == Z.Z_STREAM_END, rc
    res[k] = s.total_out
    l.deflateEnd(C.byref(s))


for it in range(2):
    res = [0]
    t0 = time.time()
    one(0, n, res, 0)
    dt = time.time() - t0
    print("one stream: %.1f ms = %.2f GB/s" % (dt * 1e3, n / dt / 1e9), flush=True)
per = (n // nt) & ~((1 << 20) - 1)
for it in range(3):
    res = [0] * nt
    ths = [threading.Thread(target=one, args=(k * per, per, res, k)) for k in range(nt)]
    t0 = time.time()
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    dt = time.time() - t0
    print("%d streams at once: %.1f ms = %.2f GB/s" % (nt, dt * 1e3, per * nt / dt / 1e9), flush=True)
