"""Manual tool: inflate() of this library on a stream the system zlib wrote (one gzip member, no sync markers),
host buffers.  usage: gpu_foreign_bench.py [MiB] [kind] [level]"""
import ctypes as C
import sys
import time
import zlib

sys.path.insert(0, "tests")
import _zapi as Z
import zlib_amd

mib = int(sys.argv[1]) if len(sys.argv) > 1 else 512
kind = int(sys.argv[2]) if len(sys.argv) > 2 else 0
level = int(sys.argv[3]) if len(sys.argv) > 3 else 6
n = mib << 20
data = bytes(zlib_amd.corpus(n, kind, 5, piece_bytes=16 << 20, threads=16))
t0 = time.time()
c = zlib.compressobj(level, zlib.DEFLATED, 31)
comp = c.compress(data) + c.flush()
t_def = time.time() - t0
t0 = time.time()
assert zlib.decompress(comp, 47) == data
t_cpu = time.time() - t0
print("system zlib, one core: deflate %.2f s, inflate %.0f ms = %.2f GB/s; %d -> %d bytes" % (t_def, t_cpu * 1e3, n / t_cpu / 1e9, n, len(comp)), flush=True)
l = Z.lib()
src = C.create_string_buffer(comp, len(comp))
back = C.create_string_buffer(n)
for it in range(3):
    s = Z.ZStream()
    assert l.inflateInit2_(C.byref(s), 47, Z.VERSION, Z.SZ) == 0
    s.next_in, s.avail_in, s.next_out, s.avail_out = C.addressof(src), len(comp), C.addressof(back), n
    t0 = time.time()
    rc = l.inflate(C.byref(s), Z.Z_FINISH)
    dt = time.time() - t0
    assert rc == Z.Z_STREAM_END, (rc, s.msg)
    l.inflateEnd(C.byref(s))
    print("inflate() of the foreign stream: %.1f ms = %.2f GB/s of output (host buffers)" % (dt * 1e3, n / dt / 1e9), flush=True)
print("round trip", "OK" if back.raw == data else "MISMATCH")
