"""Manual tool: the stream API from several threads at once, every thread with streams of its own (zlib.h:143-145 allows
that): random inputs, steps and flush points per thread; every stream must be the one the same call sequence gives when
it runs alone, and must inflate to its input.  usage: gpu_threads_fuzz.py [threads] [cases per thread] [seed]"""
import os
import random
import sys
import threading
import zlib

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["ZAMD_Z_CHUNK"] = "262144"
os.environ["ZAMD_Z_BATCH_MIB"] = "2"
import _zapi as Z
import zlib_amd

nt = int(sys.argv[1]) if len(sys.argv) > 1 else 6
per = int(sys.argv[2]) if len(sys.argv) > 2 else 20
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 1
pool = bytes(zlib_amd.corpus(16 << 20, 4, 81))
FL = [Z.Z_SYNC_FLUSH, Z.Z_FULL_FLUSH, Z.Z_PARTIAL_FLUSH, Z.Z_BLOCK]


def plan(rng):
    n = rng.choice([rng.randrange(1, 9 << 20), rng.randrange(1, 8) << 20, rng.randrange(1, 100000)])
    off = rng.randrange(0, len(pool) - n + 1)
    flushes = {rng.randrange(1, n + 1): rng.choice(FL) for _ in range(rng.choice([0, 0, 1, 2]))}
    return dict(off=off, n=n, wbits=rng.choice([15, -15, 31]), level=rng.choice([1, 6, 9]), flushes=flushes,
                in_step=rng.choice([None, rng.randrange(1 << 16, 4 << 20)]), out_step=rng.choice([9 << 20, 70000, rng.randrange(1 << 12, 1 << 22)]),
                i_in=rng.choice([None, rng.randrange(1 << 15, 3 << 20)]), i_out=rng.choice([9 << 20, 70000, rng.randrange(1 << 12, 1 << 22)]))


def run(p):
    data = pool[p["off"]:p["off"] + p["n"]]
    got, tot = Z.deflate_stream(data, p["level"], p["wbits"], in_step=p["in_step"], out_step=p["out_step"], flushes=p["flushes"] or None)
    rc, out, msg, unused, totals = Z.inflate_stream(got, p["wbits"] if p["wbits"] != 31 else 47, in_step=p["i_in"], out_step=p["i_out"])
    return got, tot, rc, bytes(out) == data and zlib.decompressobj(p["wbits"]).decompress(got) == data


plans = [[plan(random.Random(seed * 1000 + t * 100 + i)) for i in range(per)] for t in range(nt)]
alone = [[run(p) for p in row] for row in plans]
together = [[None] * per for _ in range(nt)]


def work(t):
    for i, p in enumerate(plans[t]):
        together[t][i] = run(p)


ths = [threading.Thread(target=work, args=(t,)) for t in range(nt)]
for th in ths:
    th.start()
for th in ths:
    th.join()
bad = 0
for t in range(nt):
    for i in range(per):
        a, b = alone[t][i], together[t][i]
        if b is None or a[0] != b[0] or a[1] != b[1] or not (a[3] and b[3]) or a[2] != Z.Z_STREAM_END or b[2] != Z.Z_STREAM_END:
            bad += 1
            print("MISMATCH thread", t, "case", i, plans[t][i], flush=True)
print("threads", nt, "cases", nt * per, "bad", bad)
sys.exit(1 if bad else 0)
