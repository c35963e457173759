"""Manual tool: random parity cases, engine against the oracle (the CPU restatement that is pinned to the reference
build): corpus kind, length, level, strategy, memLevel and chunk size at random, plus inputs made of spliced pieces of
different kinds, runs and repeats.  Bytes and chunk lengths must be equal.  usage: gpu_parity_fuzz.py [cases] [seed]"""
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _oracle as O
import zlib_amd

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
eng = zlib_amd.Engine(0)
bad = 0
for case in range(cases):
    level = rng.choice([0, 1, 2, 3, 4, 5, 6, 6, 7, 8, 9])
    strategy = rng.choice([0, 0, 0, 1, 2, 3, 4])
    if level == 0 and strategy in (2, 3):
        strategy = 0                              # refused: the reference overruns its own buffer there
    mem = rng.choice([1, 2, 5, 8, 8, 9])
    chunk = rng.choice([32768, 49152, 65536, 131072, 1 << 18, 1 << 20])
    n = rng.choice([rng.randrange(0, 600), rng.randrange(0, 200000), rng.randrange(0, 1500000), chunk * rng.randrange(1, 4) + rng.choice([-1, 0, 1, 262])])
    if level >= 8:
        n = min(n, 400000)
    n = max(0, n)
    if rng.random() < 0.6:
        data = bytes(O.corpus(n, rng.randrange(0, 5), rng.randrange(1, 1000)))
    else:                                          # spliced: pieces of different kinds, runs, copies of earlier stretches
        parts = bytearray()
        while len(parts) < n:
            k = rng.randrange(1, 70000)
            what = rng.random()
            if what < 0.5:
                parts += O.corpus(k, rng.randrange(0, 5), rng.randrange(1, 1000))
            elif what < 0.7:
                parts += bytes([rng.randrange(256)]) * k
            elif parts:
                a = rng.randrange(0, len(parts))
                parts += parts[a:a + k]
            else:
                parts += bytes(rng.randrange(256) for _ in range(min(k, 300)))
        data = bytes(parts[:n])
    desc = dict(case=case, n=n, level=level, strategy=strategy, mem=mem, chunk=chunk)
    want, wlens = O.deflate_chunked(data, chunk, level, strategy, mem)
    out, offs = eng.deflate_chunks(data, level=level, strategy=strategy, mem_level=mem, chunk_size=chunk)
    lens = [offs[i + 1] - offs[i] for i in range(len(offs) - 1)]
    if out != want or lens != list(wlens):
        bad += 1
        print("MISMATCH", desc, len(out), len(want), flush=True)
    elif case % 10 == 0:
        print("ok", desc, flush=True)
print("cases", cases, "bad", bad)
sys.exit(1 if bad else 0)
