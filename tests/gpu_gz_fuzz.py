"""Manual tool: random use of the gz* layer.  Every case writes a file with random gzwrite() sizes, modes (plain, with the
chunk index, with the windows for random access) and chunk sizes, reads it back with Python's gzip and with gzread() in
random pieces, and seeks to random offsets.  usage: gpu_gz_fuzz.py [cases] [seed]"""
import ctypes as C
import gzip
import os
import random
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["ZAMD_GZ_AHEAD_AFTER_KIB"] = "1024"       # (read once, when the library loads)
import _zapi as Z
import zlib_amd

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
pool = bytes(zlib_amd.corpus(24 << 20, 4, 78))
l = Z.lib()
bad = 0
tmp = tempfile.mkdtemp(prefix="zamd_gzfuzz_")
for case in range(cases):
    chunk = rng.choice([65536, 262144, 1 << 20])
    os.environ["ZAMD_Z_CHUNK"] = str(chunk)
    os.environ["ZAMD_Z_BATCH_MIB"] = str(rng.choice([1, 2, 512]))
    os.environ["ZAMD_GZ_BUF_MIB"] = rng.choice(["1", "32"])       # (1 MiB with batches of 2: writes collected behind the caller fill their buffers)
    os.environ["ZAMD_GZ_BEHIND"] = rng.choice(["1", "1", "0"])
    os.environ["ZAMD_GZ_AHEAD_MIB"] = rng.choice(["3", "5", "256", "0"])  # (with the small buffer and AHEAD_AFTER below: readers get their data inflated ahead)
    n = rng.choice([rng.randrange(1, len(pool)), rng.randrange(1, 20) << 20, 1, 100])
    off = rng.randrange(0, len(pool) - n + 1)
    data = pool[off:off + n]
    mode = rng.choice(["wb", "wb6I", "wb6W", "wb1W", "wb9I"])
    path = os.path.join(tmp, "f%d.gz" % case).encode()
    desc = dict(case=case, n=n, chunk=chunk, mode=mode, batch=os.environ["ZAMD_Z_BATCH_MIB"], buf=os.environ["ZAMD_GZ_BUF_MIB"], behind=os.environ["ZAMD_GZ_BEHIND"], ahead=os.environ["ZAMD_GZ_AHEAD_MIB"])
    g = l.gzopen(path, mode.encode())
    assert g
    at = 0
    while at < n:
        k = min(n - at, rng.choice([1, 4096, rng.randrange(1, 1 << 20), rng.randrange(1, 8 << 20)]))
        what = rng.random()
        if what < 0.03:
            assert l.gzputc(g, data[at]) == data[at], desc
            k = 1
        else:
            w = l.gzwrite(g, data[at:at + k], k)
            assert w == k, (desc, w, k)
        at += k
        assert l.gztell(g) == at, (desc, at)
        what = rng.random()
        if what < 0.05:
            assert l.gzflush(g, Z.Z_SYNC_FLUSH) == 0
        elif what < 0.08 and mode == "wb":
            assert l.gzsetparams(g, rng.choice([1, 6, 9]), 0) == 0
        elif what < 0.11:
            assert l.gzoffset(g) >= 0
    assert l.gzclose(g) == 0
    ok = True
    try:
        ok = gzip.open(path.decode(), "rb").read() == data
    except Exception as ex:          # noqa: BLE001 - whatever it is, it is a mismatch
        ok = False
        desc["gzip"] = repr(ex)
    g = l.gzopen(path, b"rb")
    assert g
    got = bytearray()
    buf = C.create_string_buffer(9 << 20)
    while True:
        k = rng.choice([1, 4096, rng.randrange(1, 1 << 20), rng.randrange(1, 9 << 20)])
        if got and rng.random() < 0.05:                 # push the last byte back and read it again
            assert l.gzungetc(got[-1], g) == got[-1], desc
            del got[-1]
        if rng.random() < 0.03:
            c = l.gzgetc(g)
            if c >= 0:
                got.append(c)
            l.gzoffset(g), l.gzeof(g), l.gzdirect(g)
        r = l.gzread(g, buf, k)
        if r < 0:
            ok = False
            desc["gzread"] = r
            break
        got += buf.raw[:r]
        if r < k:
            break
    ok = ok and bytes(got) == data
    for _ in range(6):
        to = rng.randrange(0, n)
        k = min(n - to, rng.randrange(1, 300000))
        p = l.gzseek(g, to, 0)
        r = l.gzread(g, buf, k) if p == to else -1
        if p != to or r != k or buf.raw[:k] != data[to:to + k]:
            ok = False
            desc.setdefault("seek", []).append((to, k, p, r))
    l.gzclose(g)
    os.remove(path)
    if not ok:
        bad += 1
        print("MISMATCH", desc, len(got), flush=True)
    elif case % 5 == 0:
        print("ok", case, n, mode, flush=True)
os.rmdir(tmp)
print("cases", cases, "bad", bad)
sys.exit(1 if bad else 0)
