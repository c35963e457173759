"""Manual tool: six threads, each with streams of its own (compress2 + uncompress of the stream API), one after the other
and all at once.  usage: gpu_threads_bench.py [MiB per thread] [threads]"""
import ctypes as C
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _zapi as Z
import zlib_amd

mib = int(sys.argv[1]) if len(sys.argv) > 1 else 64
nt = int(sys.argv[2]) if len(sys.argv) > 2 else 6
l = Z.lib()
l.compress2.argtypes = [C.c_void_p, C.POINTER(C.c_ulong), C.c_void_p, C.c_ulong, C.c_int]
l.uncompress.argtypes = [C.c_void_p, C.POINTER(C.c_ulong), C.c_void_p, C.c_ulong]
n = mib << 20
datas = [zlib_amd.corpus(n, k % 5, 40 + k, piece_bytes=16 << 20, threads=8) for k in range(nt)]
srcs = [(C.c_char * n).from_buffer(d) for d in datas]
cap = l.compressBound(n)
outs = [C.create_string_buffer(cap) for _ in range(nt)]
backs = [C.create_string_buffer(n) for _ in range(nt)]
ok = [False] * nt


def work(k, reps=2):
    for _ in range(reps):
        c = C.c_ulong(cap)
        assert l.compress2(outs[k], C.byref(c), srcs[k], n, 6) == 0
        m = C.c_ulong(n)
        assert l.uncompress(backs[k], C.byref(m), outs[k], c.value) == 0 and m.value == n


for k in range(nt):
    work(k, 1)                  # warm: the first lane's engine, the buffers
ths = [threading.Thread(target=work, args=(k, 1)) for k in range(nt)]
for t in ths:
    t.start()                   # warm: the engines of the other lanes, which only threads at once bring up
for t in ths:
    t.join()
t0 = time.time()
for k in range(nt):
    work(k)
serial = time.time() - t0
ths = [threading.Thread(target=work, args=(k,)) for k in range(nt)]
t0 = time.time()
for t in ths:
    t.start()
for t in ths:
    t.join()
wall = time.time() - t0
for k in range(nt):
    ok[k] = backs[k].raw == bytes(datas[k])
print("%d threads x %d MiB: one after the other %.1f ms, all at once %.1f ms: ratio %.2f; lanes %s; round trips %s" % (
    nt, mib, serial * 1e3, wall * 1e3, wall / serial, os.environ.get("ZAMD_Z_LANES", "4 (default)"), "OK" if all(ok) else "MISMATCH"))
