"""Manual tool: the gz layer on a large file: gzwrite() and gzread() in pieces of the given size.
usage: gpu_gz_bench.py [MiB] [piece MiB] [mode]"""
import ctypes as C
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _zapi as Z
import zlib_amd

mib = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
piece = (int(sys.argv[2]) if len(sys.argv) > 2 else 64) << 20
mode = sys.argv[3] if len(sys.argv) > 3 else "wb6I"
n = mib << 20
data = zlib_amd.corpus(n, 4, 5, piece_bytes=16 << 20, threads=16)
src = (C.c_char * n).from_buffer(data)
l = Z.lib()
l.gzwrite.argtypes = [C.c_void_p, C.c_void_p, C.c_uint]
path = os.path.join(tempfile.gettempdir(), "zamd_gz_bench.gz").encode()
back = C.create_string_buffer(n)
for it in range(2):
    t0 = time.time()
    g = l.gzopen(path, mode.encode())
    at = 0
    while at < n:
        k = min(piece, n - at)
        assert l.gzwrite(g, C.addressof(src) + at, k) == k
        at += k
    assert l.gzclose(g) == 0
    dt = time.time() - t0
    print("gzwrite %s: %d MiB in pieces of %d MiB: %.1f ms = %.2f GB/s (file %d bytes)" % (mode, mib, piece >> 20, dt * 1e3, n / dt / 1e9, os.path.getsize(path)), flush=True)
for it in range(2):
    t0 = time.time()
    g = l.gzopen(path, b"rb")
    at = 0
    while at < n:
        k = min(piece, n - at)
        r = l.gzread(g, C.addressof(back) + at, k)
        assert r == k, (r, k)
        at += k
    l.gzclose(g)
    dt = time.time() - t0
    print("gzread: %.1f ms = %.2f GB/s" % (dt * 1e3, n / dt / 1e9), flush=True)
print("round trip", "OK" if back.raw == bytes(data) else "MISMATCH")
os.remove(path)
