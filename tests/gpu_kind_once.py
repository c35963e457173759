"""Manual tool: one deflate call over one corpus kind (for counter passes under rocprofv3).  usage: gpu_kind_once.py KIND [MiB] [level]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zlib_amd

kind = int(sys.argv[1])
mib = int(sys.argv[2]) if len(sys.argv) > 2 else 256
level = int(sys.argv[3]) if len(sys.argv) > 3 else 6
n = mib << 20
eng = zlib_amd.Engine(0)
data = zlib_amd.corpus(n, kind, 5, piece_bytes=16 << 20, threads=16)
cap = eng.lib.zamd_deflate_bound(n, 1 << 20, 8)
d_in, d_out = eng.alloc(n), eng.alloc(cap)
d_in.upload(data)
total = eng.deflate_dev(d_in.ptr, n, d_out.ptr, cap, level=level)
names = ["chain", "match", "parse", "plan", "layout", "emit", "call"]
t = {nm: eng.timing_ms(i)[0] for i, nm in enumerate(names)}
t["match4"] = eng.timing_ms(10)[0]
print("kind %d ratio %.3f fallback hops %d  " % (kind, n / total, int(eng.timing_ms(9)[0])) + "  ".join("%s %.1f" % (k, v) for k, v in t.items()), flush=True)
