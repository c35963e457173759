"""Manual tool: streams of another producer (Python's zlib: the system's libz) through inflate() and uncompress().  Random
levels, strategies, window sizes, wrappers, flush points of every kind and sizes from nothing to 20 MB; random input and
output steps.  usage: gpu_foreign_fuzz.py [cases] [seed]"""
import os
import random
import sys
import zlib

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _zapi as Z
import zlib_amd

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
pool = bytes(zlib_amd.corpus(20 << 20, 4, 79)) + bytes(zlib_amd.corpus(4 << 20, 1, 80))
bad = 0
for case in range(cases):
    n = rng.choice([0, 1, rng.randrange(0, 2000), rng.randrange(0, 300000), rng.randrange(0, len(pool))])
    off = rng.randrange(0, len(pool) - n + 1)
    data = pool[off:off + n]
    wb = rng.choice([9, 12, 15])
    wrap = rng.choice(["zlib", "raw", "gzip"])
    wbits = wb if wrap == "zlib" else -wb if wrap == "raw" else 16 + wb
    level = rng.choice([0, 1, 3, 6, 9])
    strategy = rng.choice([zlib.Z_DEFAULT_STRATEGY, zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED])
    c = zlib.compressobj(level, zlib.DEFLATED, wbits, rng.choice([1, 8, 9]), strategy)
    comp = bytearray()
    at = 0
    while at < n:
        k = min(n - at, rng.choice([n, rng.randrange(1, 1 << 20), rng.randrange(1, 5000)]))
        comp += c.compress(data[at:at + k])
        at += k
        if rng.random() < 0.3:
            comp += c.flush(rng.choice([zlib.Z_SYNC_FLUSH, zlib.Z_FULL_FLUSH, zlib.Z_PARTIAL_FLUSH, zlib.Z_BLOCK]))
    comp += c.flush()
    comp = bytes(comp)
    desc = dict(case=case, n=n, wbits=wbits, level=level, strategy=strategy, comp=len(comp))
    read_bits = wbits if wrap != "zlib" else rng.choice([wb, 15, 47])
    rc, out, msg, unused, totals = Z.inflate_stream(comp + b"tail", read_bits, in_step=rng.choice([None, None, rng.randrange(1, 1 << 20)]),
                                                    out_step=rng.choice([max(n, 1) + 16, 70000, rng.randrange(1, 1 << 22)]))
    if not (rc == Z.Z_STREAM_END and bytes(out) == data and unused == 4):
        bad += 1
        print("MISMATCH", desc, rc, msg, len(out), unused, flush=True)
    elif case % 10 == 0:
        print("ok", desc, flush=True)
print("cases", cases, "bad", bad)
sys.exit(1 if bad else 0)
