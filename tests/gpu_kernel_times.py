"""Manual tool: per-kernel times of one deflate call with every kernel alone on the GPU.  usage: gpu_kernel_times.py [MiB] [level] [kind]"""
import sys

import zlib_amd

mib = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
level = int(sys.argv[2]) if len(sys.argv) > 2 else 6
kinds = [int(sys.argv[3])] if len(sys.argv) > 3 else range(5)
n = mib << 20
eng = zlib_amd.Engine(0)
eng.set_streams(4, serialize=True)
idx = {"chain": 0, "match": 1, "parse": 2, "plan": 3, "layout": 4, "emit": 5, "call": 6, "reach": 10, "first3": 11}
for kind in kinds:
    data = zlib_amd.corpus(n, kind, 5, piece_bytes=16 << 20, threads=16)
    cap = eng.lib.zamd_deflate_bound(n, 1 << 20, 8)
    d_in, d_out = eng.alloc(n), eng.alloc(cap)
    d_in.upload(data)
    for it in range(2):
        total = eng.deflate_dev(d_in.ptr, n, d_out.ptr, cap, level=level)
    t = {nm: eng.timing_ms(i) for nm, i in idx.items()}
    print("kind %d level %d %d MiB ratio %.3f: " % (kind, level, mib, n / total) + "  ".join("%s %.1f/%d" % (k, v[0], v[1]) for k, v in t.items()), flush=True)
    d_in.free(); d_out.free()
