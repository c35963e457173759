"""Manual tool: the same input through deflate() with different engine batch sizes; the streams must be equal."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import _zapi as Z
import zlib_amd

mib = int(sys.argv[1]) if len(sys.argv) > 1 else 1536
n = mib << 20
data = zlib_amd.corpus(n, 4, 5, piece_bytes=16 << 20, threads=16)
src = (C.c_char * n).from_buffer(data)
l = Z.lib()
cap = l.compressBound(n)
outs = {}
for b in sys.argv[2:] or ["512", "1024"]:
    os.environ["ZAMD_Z_BATCH_MIB"] = b
    out = C.create_string_buffer(cap)
    s = Z.ZStream()
    assert l.deflateInit2_(C.byref(s), 6, 8, 31, 8, 0, Z.VERSION, Z.SZ) == 0
    s.next_in, s.avail_in, s.next_out, s.avail_out = C.addressof(src), n, C.addressof(out), cap
    assert l.deflate(C.byref(s), Z.Z_FINISH) == Z.Z_STREAM_END
    outs[b] = out.raw[:s.total_out]
    l.deflateEnd(C.byref(s))
    print("batch", b, "MiB ->", len(outs[b]), flush=True)
keys = list(outs)
a = outs[keys[0]]
for k in keys[1:]:
    b = outs[k]
    if a == b:
        print(keys[0], "==", k)
        continue
    i = next(i for i in range(min(len(a), len(b))) if a[i] != b[i])
    print("first difference between", keys[0], "and", k, "at byte", i, "of", len(a), len(b))
    print(a[i - 16:i + 24].hex()); print(b[i - 16:i + 24].hex())
