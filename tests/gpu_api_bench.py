"""Manual tool: host-buffer (PCIe-inclusive) rate of the stream API.  usage: gpu_api_bench.py [MiB] [level] [pinned]"""
import ctypes as C
import sys
import time

import os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _zapi as Z
import zlib_amd

mib = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
level = int(sys.argv[2]) if len(sys.argv) > 2 else 6
n = mib << 20
data = zlib_amd.corpus(n, 4, 5, piece_bytes=16 << 20, threads=16)
src = (C.c_char * n).from_buffer(data)
l = Z.lib()
cap = l.compressBound(n)
out = C.create_string_buffer(cap)
back = C.create_string_buffer(n)
if len(sys.argv) > 3 and sys.argv[3] == "pinned":      # the caller's buffers registered with the GPU beforehand
    tiny = C.create_string_buffer(1 << 16); tl = C.c_ulong(1 << 16)
    l.compress.argtypes = [C.c_void_p, C.POINTER(C.c_ulong), C.c_void_p, C.c_ulong]
    assert l.compress(tiny, C.byref(tl), src, 1000) == 0        # brings the GPU runtime up
    hip = C.CDLL("libamdhip64.so")
    hip.hipHostRegister.argtypes = [C.c_void_p, C.c_size_t, C.c_uint]
    for buf, sz in ((src, n), (out, cap), (back, n)):
        rc = hip.hipHostRegister(C.addressof(buf), sz, 0)
        print("hipHostRegister", sz, "->", rc, flush=True)
for it in range(2):
    s = Z.ZStream()
    assert l.deflateInit2_(C.byref(s), level, 8, 31, 8, 0, Z.VERSION, Z.SZ) == 0
    assert l.zamd_deflate_index(C.byref(s), 1) == 0
    s.next_in, s.avail_in, s.next_out, s.avail_out = C.addressof(src), n, C.addressof(out), cap
    t0 = time.time()
    rc = l.deflate(C.byref(s), Z.Z_FINISH)
    dt = time.time() - t0
    assert rc == Z.Z_STREAM_END, rc
    total = s.total_out
    pairs, cnt = C.POINTER(C.c_uint)(), C.c_uint(0)
    assert l.zamd_deflate_index_get(C.byref(s), C.byref(pairs), C.byref(cnt)) == 0
    index = (C.c_uint * (2 * cnt.value))(*pairs[:2 * cnt.value])
    recs, per = C.POINTER(C.c_ulonglong)(), C.c_uint(0)
    assert l.zamd_deflate_index_splits(C.byref(s), C.byref(recs), C.byref(per)) == 0
    splits = (C.c_ulonglong * (per.value * cnt.value))(*recs[:per.value * cnt.value])
    l.deflateEnd(C.byref(s))
    print("deflate(): %d -> %d in %.1f ms = %.2f GB/s (host buffers, gzip wrapper)" % (n, total, dt * 1e3, n / dt / 1e9), flush=True)
for it in range(6):
    s = Z.ZStream()
    assert l.inflateInit2_(C.byref(s), 47, Z.VERSION, Z.SZ) == 0
    if it >= 2:
        assert l.zamd_inflate_index_hint(C.byref(s), index, len(index) // 2) == 0
    if it >= 4:
        assert l.zamd_inflate_index_hint_splits(C.byref(s), splits, per.value) == 0
    s.next_in, s.avail_in, s.next_out, s.avail_out = C.addressof(out), total, C.addressof(back), n
    t0 = time.time()
    rc = l.inflate(C.byref(s), Z.Z_FINISH)
    dt = time.time() - t0
    assert rc == Z.Z_STREAM_END, (rc, s.msg)
    l.inflateEnd(C.byref(s))
    print("inflate()%s: %.1f ms = %.2f GB/s of output (host buffers)" % (" with the chunk index and its split points" if it >= 4 else " with the chunk index" if it >= 2 else "", dt * 1e3, n / dt / 1e9), flush=True)
print("round trip", "OK" if back.raw == bytes(data) else "MISMATCH")
