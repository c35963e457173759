"""Manual tool: per-kernel times of the deflate pipeline per corpus kind.  usage: gpu_deflate_kinds.py [MiB] [level]"""
import sys

import zlib_amd

mib = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
level = int(sys.argv[2]) if len(sys.argv) > 2 else 6
n = mib << 20
chunk = 1 << 20
eng = zlib_amd.Engine(0)
names = ["chain", "match", "parse", "plan", "layout", "emit", "call"]
idx = {"chain": 0, "match": 1, "parse": 2, "plan": 3, "layout": 4, "emit": 5, "call": 6, "reach": 10, "first3": 11}
for kind in range(5):
    data = zlib_amd.corpus(n, kind, 5, piece_bytes=16 << 20, threads=16)
    cap = eng.lib.zamd_deflate_bound(n, chunk, 8)
    d_in, d_out = eng.alloc(n), eng.alloc(cap)
    d_in.upload(data)
    for it in range(2):
        total = eng.deflate_dev(d_in.ptr, n, d_out.ptr, cap, level=level)
    t = {nm: eng.timing_ms(i)[0] for nm, i in idx.items()}
    print("kind %d ratio %.3f  %.2f GB/s  " % (kind, n / total, n / t["call"] / 1e6) + "  ".join("%s %.1f" % (k, v) for k, v in t.items()), flush=True)
    d_in.free(); d_out.free()
