"""Manual tool: random calls of the stream API against each other.  Every case compresses one input with random batch
and chunk sizes, input steps, output steps and flush points; the stream must be the one a single large batch gives with the
same flush points, Python's zlib must inflate it to the input, and inflate() must do the same with random steps, with and
without the chunk index.  usage: gpu_api_fuzz.py [cases] [seed]"""
import os
import random
import sys
import zlib

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _zapi as Z
import zlib_amd

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
pool = bytes(zlib_amd.corpus(14 << 20, 4, 77))
FL = [Z.Z_SYNC_FLUSH, Z.Z_FULL_FLUSH, Z.Z_PARTIAL_FLUSH, Z.Z_BLOCK]
bad = 0
for case in range(cases):
    chunk = rng.choice([65536, 262144, 1 << 20])
    batch = rng.choice([1, 2, 3])
    n = rng.choice([rng.randrange(1, 13 << 20), rng.randrange(1, 12) << 20, (rng.randrange(1, 12) << 20) + rng.choice([-1, 1, 77, chunk, -chunk])])
    n = max(1, min(n, len(pool)))
    off = rng.randrange(0, len(pool) - n + 1)
    data = pool[off:off + n]
    wbits = rng.choice([15, -15, 31])
    level = rng.choice([1, 6, 6, 9])
    flushes = {}
    for _ in range(rng.choice([0, 0, 1, 2])):
        at = rng.choice([rng.randrange(1, n + 1), min(n, rng.randrange(1, 12) << 20)])
        flushes[at] = rng.choice(FL)
    in_step = rng.choice([None, None, rng.randrange(1 << 16, 4 << 20)])
    out_step = rng.choice([n + (n >> 3) + (1 << 17), 70000, rng.randrange(1 << 12, 1 << 22)])
    desc = dict(case=case, n=n, chunk=chunk, batch=batch, wbits=wbits, level=level, flushes=flushes, in_step=in_step, out_step=out_step)
    os.environ["ZAMD_Z_CHUNK"] = str(chunk)
    os.environ["ZAMD_Z_BATCH_MIB"] = "64"
    want, tot_w = Z.deflate_stream(data, level, wbits, out_step=n + (n >> 3) + (1 << 17), flushes=flushes or None)
    os.environ["ZAMD_Z_BATCH_MIB"] = str(batch)
    idx, spl = [], []
    got, tot_g = Z.deflate_stream(data, level, wbits, in_step=in_step, out_step=out_step, flushes=flushes or None, index=idx, splits=spl)
    ok = got == want and tot_g == tot_w
    try:
        back = zlib.decompressobj(wbits).decompress(got)
    except zlib.error as ex:
        back = repr(ex)
    ok = ok and back == data
    # inflate() with random steps; the index only describes streams without odd flushes
    hint = idx[1:] if idx and idx[0] == 0 else None
    for use_hint in (False, True):
        if use_hint and hint is None:
            continue
        rc, out, msg, unused, totals = Z.inflate_stream(got, wbits if wbits != 31 else 47, in_step=rng.choice([None, rng.randrange(1 << 15, 3 << 20)]),
                                                        out_step=rng.choice([n + 16, 70000, rng.randrange(1 << 12, 1 << 22)]),
                                                        hint=hint if use_hint else None, hint_splits=spl if use_hint and spl and spl[0] else None)
        if not (rc == Z.Z_STREAM_END and bytes(out) == data):
            ok = False
            desc["inflate"] = (use_hint, rc, msg, len(out))
    if not ok:
        bad += 1
        print("MISMATCH", desc, len(got), len(want), flush=True)
    elif case % 5 == 0:
        print("ok", case, n, flush=True)
print("cases", cases, "bad", bad)
sys.exit(1 if bad else 0)
