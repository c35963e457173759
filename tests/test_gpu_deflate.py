"""Parity of the HIP engine (through the C ABI) with the CPU oracle and with the
golden vectors of the reference build.  All integer work: the bar is bit-exact."""
import json
import os
import zlib

import pytest

import _oracle as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "deflate_golden.json")))
GPU_LEVELS = range(0, 10)   # deflate_stored (0), deflate_fast (1-3) and deflate_slow (4-9)


def _gpu(engine, data, **kw):
    out, offs = engine.deflate_chunks(data, **kw)
    return out, [offs[i + 1] - offs[i] for i in range(len(offs) - 1)]


@pytest.mark.parametrize("row", GOLD["chunked"],
                         ids=lambda r: "k%d_L%d_S%d_M%d_n%d" % (r["kind"], r["level"], r["strategy"], r["mem_level"], r["n"]))
def test_golden_rows_from_reference_build(engine, row):
    data = O.corpus(row["n"], row["kind"], row["seed"])
    out, lens = _gpu(engine, data, level=row["level"], strategy=row["strategy"], mem_level=row["mem_level"], chunk_size=row["chunk"])
    assert lens == row["chunk_len"]
    assert "%08x" % zlib.crc32(out) == row["out_crc"]
    off = 0
    for l, c in zip(lens, row["chunk_crc"]):
        assert "%08x" % zlib.crc32(out[off:off + l]) == c
        off += l


CASES = []
for kind in range(5):
    CASES.append((kind, (1 << 20) + 4321, 6, 0, 8, 1 << 19))
for lvl in GPU_LEVELS:
    CASES.append((4, 300000, lvl, 0, 8, 65536))
for strat in (1, 2, 3, 4):
    CASES.append((3, 700001, 6, strat, 8, 1 << 18))
    CASES.append((2, 150000, 9, strat, 8, 65536))
CASES.append((4, 400000, 6, 3, 8, 131072))      # Z_RLE over long runs (the mixed corpus has stretches of one byte value)
for mem in (1, 2, 4, 7, 9):
    CASES.append((4, 400000, 6, 0, mem, 131072))
for mem in (1, 3, 8):
    for lvl in (1, 2, 3):
        CASES.append((0, 250000, lvl, 0, mem, 65536))
        CASES.append((4, 250000, lvl, 4, mem, 131072))
for kind in range(5):
    CASES.append((kind, (1 << 20) + 4321, 1, 0, 8, 1 << 19))
# ragged / edge sizes: empty, below MIN_MATCH, around MIN_LOOKAHEAD, block-split boundary, tail-mode slide geometry
for n in (0, 1, 2, 3, 4, 261, 262, 263, 16383, 16384, 32767, 32768, 32769, 65274, 65277, 65374, 65535, 65536, 65537, 98304 + 32600):
    CASES.append((0, n, 6, 0, 8, 65536))
    CASES.append((4, n, 9, 0, 8, 32768))
    CASES.append((0, n, 1, 0, 8, 65536))
    CASES.append((4, n, 3, 0, 8, 32768))
    CASES.append((1, n, 0, 0, 8, 65536))
# chunks of 4 MiB (a-coordinates beyond 2^21), and sizes around the 240- and 256-byte reloads of the input registers of levels 1-3
for lvl in (1, 2, 6):
    CASES.append((4, 5 * (1 << 20) + 123, lvl, 0, 8, 1 << 22))
for n in (188, 189, 239, 240, 241, 255, 256, 257, 479, 480, 481, 735):
    for lvl in (1, 2, 3):
        CASES.append((0, n, lvl, 0, 8, 32768))
        CASES.append((2, 32768 + n, lvl, 0, 8, 32768))
for mem in (1, 5, 9):
    CASES.append((4, 300000, 0, 0, mem, 131072))
    CASES.append((1, (1 << 20) + 70000, 0, 4, mem, 1 << 20))


@pytest.mark.parametrize("kind,n,level,strategy,mem,chunk", CASES)
def test_bit_exact_against_oracle(engine, kind, n, level, strategy, mem, chunk):
    data = O.corpus(n, kind, 17 + kind)
    want, wlens = O.deflate_chunked(data, chunk, level, strategy, mem)
    out, lens = _gpu(engine, data, level=level, strategy=strategy, mem_level=mem, chunk_size=chunk)
    assert lens == wlens
    assert out == want


def test_pathological_inputs(engine):
    import random
    rnd = random.Random(99)
    inputs = [b"\0" * 300000, b"a" * 70000 + b"b" * 70000, bytes(rnd.getrandbits(8) for _ in range(200000)),
              (b"abc" * 50000)[:140001], bytes((i * 7) & 0xff for i in range(100000)),
              bytes(rnd.choice(b"ab") for _ in range(150000))]
    for per in (1, 2, 3, 5, 258, 259, 300, 32505, 32506, 32507):
        base = bytes(rnd.getrandbits(8) for _ in range(per))
        inputs.append((base * (120000 // per + 2))[:120000])
    for data in inputs:
        for level in (1, 3, 4, 6, 9):
            want, wlens = O.deflate_chunked(data, 65536, level, 0, 8)
            out, lens = _gpu(engine, data, level=level, chunk_size=65536)
            assert out == want


def test_full_size_chunks_round_trip_and_sampled_parity(engine):
    """Benchmark-sized chunks (1 MiB, 256 of them): size-independent properties over the
    whole stream, byte parity with the oracle on sampled chunks."""
    n = 256 << 20
    import zlib_amd
    data = bytes(zlib_amd.corpus(n, 4, 5, piece_bytes=16 << 20, threads=8))
    out, offs = engine.deflate_chunks(data, level=6, chunk_size=1 << 20)
    assert offs[0] == 0 and all(b > a for a, b in zip(offs, offs[1:])) and offs[-1] == len(out)
    assert zlib.decompress(out, -15) == data                         # one valid raw deflate stream
    assert engine.inflate_chunks(out, offs, n, chunk_size=1 << 20) == data   # and the GPU's own batch inflate agrees
    for c in range(len(offs) - 2):
        assert out[offs[c + 1] - 4:offs[c + 1]] == b"\x00\x00\xff\xff"  # every non-final chunk ends on a sync marker
    for c in (0, 1, 77, 128, 255):
        s = c << 20
        d = min(32768, s)
        want = O.deflate_chunk(data[s - d:s + (1 << 20)], d, 1 << 20, 6, 0, 8, final=(c == 255))
        assert out[offs[c]:offs[c + 1]] == want


def test_shard_entry_point_continues_a_stream(engine):
    """A shard with a 32 KiB leading dictionary produces the same chunks as the
    whole-buffer call (what each rank of a multi-GPU run computes)."""
    import ctypes as C
    n, chunk = 6 * 131072 + 555, 131072
    data = O.corpus(n, 4, 21)
    whole, offs = engine.deflate_chunks(data, level=6, chunk_size=chunk)
    split = 3 * chunk
    d_in = engine.alloc(n); d_in.upload(data)
    cap = engine.lib.zamd_deflate_bound(n, chunk, 8)
    d_out = engine.alloc(cap)
    tot = C.c_uint64(0)
    rc = engine.lib.zamd_deflate_shard_dev(engine.handle, d_in.ptr, split, 0, chunk, 6, 0, 8, 0, d_out.ptr, cap, None, C.byref(tot))
    assert rc == 0
    first = bytes(d_out.download(tot.value))
    rc = engine.lib.zamd_deflate_shard_dev(engine.handle, d_in.ptr + split, n - split, 32768, chunk, 6, 0, 8, 1, d_out.ptr, cap, None, C.byref(tot))
    assert rc == 0
    second = bytes(d_out.download(tot.value))
    assert first + second == whole


def test_checksums(engine):
    for n in (0, 1, 15, 16, 17, 5551, 5552, 5553, 65521, (1 << 20) - 1, 1 << 20, (5 << 20) + 3):
        d = O.corpus(n, 4, 3)
        assert engine.adler32(d) == zlib.adler32(d)
        assert engine.crc32(d) == zlib.crc32(d)
        assert engine.adler32(d, 0x12345) == zlib.adler32(d, 0x12345)
        assert engine.crc32(d, 0xdeadbeef) == zlib.crc32(d, 0xdeadbeef)
    # checksum of checksums: per-chunk values combined equal the whole-buffer value (64 MiB)
    import ctypes as C
    import zlib_amd
    n = 64 << 20
    data = bytes(zlib_amd.corpus(n, 4, 8, piece_bytes=16 << 20, threads=8))
    buf = engine.alloc(n); buf.upload(data)
    nch = n >> 20
    da, dc = engine.alloc(4 * nch), engine.alloc(4 * nch)
    assert engine.lib.zamd_checksums_dev(engine.handle, buf.ptr, n, 1 << 20, da.ptr, dc.ptr) == 0
    ad = memoryview(da.download()).cast("I"); cr = memoryview(dc.download()).cast("I")
    a, c = 1, 0
    for i in range(nch):
        assert ad[i] == zlib.adler32(data[i << 20:(i + 1) << 20])
        a = zlib_amd.adler32_combine(a, ad[i], 1 << 20); c = zlib_amd.crc32_combine(c, cr[i], 1 << 20)
    assert a == zlib.adler32(data) and c == zlib.crc32(data)


def test_errors_are_loud(engine):
    import ctypes as C
    import zlib_amd
    d = engine.alloc(1 << 16); o = engine.alloc(1 << 17)
    tot = C.c_uint64(0)
    lib = engine.lib
    assert lib.zamd_deflate_dev(engine.handle, d.ptr, 1 << 16, 1 << 16, 10, 0, 8, 1, o.ptr, 1 << 17, None, C.byref(tot)) == zlib_amd.Z_STREAM_ERROR
    assert lib.zamd_deflate_dev(engine.handle, d.ptr, 1 << 16, 1 << 16, 0, 3, 8, 1, o.ptr, 1 << 17, None, C.byref(tot)) == zlib_amd.Z_STREAM_ERROR
    assert lib.zamd_deflate_dev(engine.handle, d.ptr, 1 << 16, 1000, 6, 0, 8, 1, o.ptr, 1 << 17, None, C.byref(tot)) == zlib_amd.Z_STREAM_ERROR
    assert lib.zamd_deflate_dev(engine.handle, d.ptr, 1 << 16, 1 << 16, 6, 7, 8, 1, o.ptr, 1 << 17, None, C.byref(tot)) == zlib_amd.Z_STREAM_ERROR
    d.upload(bytes(os.urandom(1 << 16)))
    assert lib.zamd_deflate_dev(engine.handle, d.ptr, 1 << 16, 1 << 16, 6, 0, 8, 1, o.ptr, 1024, None, C.byref(tot)) == zlib_amd.Z_BUF_ERROR


@pytest.mark.parametrize("head", [16, 4096, 32752])
def test_shard_entry_point_with_a_short_head_dictionary(engine, head):
    data = O.corpus(head + 200000, 0, 44)
    chunk = 65536
    n = len(data) - head
    nch = (n + chunk - 1) // chunk
    cap = engine.lib.zamd_deflate_bound(n, chunk, 8)
    d_in, d_out, d_off = engine.alloc(32768 + n), engine.alloc(cap), engine.alloc(8 * (nch + 1))
    d_in.upload(data, offset=32768 - head)
    total = engine.deflate_shard_dev(d_in.ptr + 32768, n, head, d_out.ptr, cap, level=6, chunk_size=chunk, d_chunk_off=d_off.ptr)
    out = bytes(d_out.download(total))
    offs = list(memoryview(d_off.download(8 * (nch + 1))).cast("Q"))
    want = b""
    for c in range(nch):
        start = head + c * chunk
        dl = min(start, 32768)
        ln = min(chunk, len(data) - start)
        want += O.deflate_chunk(data[start - dl:start + ln], dl, ln, 6, 0, 8, c == nch - 1)
    assert out == want and offs[-1] == len(want)


@pytest.mark.parametrize("level,tune", [(6, (4, 4, 16, 8)), (6, (8, 258, 258, 300)), (9, (32, 16, 100, 2000)), (4, (1, 0, 3, 1)),
                                        (5, (40, 300, 1000, 17)), (1, (4, 8, 32, 64)), (2, (4, 3, 9, 1)), (3, (4, 258, 258, 40)),
                                        (1, (2, 4, 8, 16)), (3, (1, 6, 32, 32)), (1, (4, 258, 5, 3))])
def test_tuned_parameters_match_the_oracle(engine, level, tune):
    """deflateTune values instead of the level's table (deflate.c:531-547)."""
    data = O.corpus(3 * 65536 + 1234, 4, 61) + O.corpus(100000, 0, 62)
    chunk = 65536
    engine.tune(*tune)
    try:
        out, offs = engine.deflate_chunks(data, level=level, chunk_size=chunk)
    finally:
        engine.tune()
    want = b""
    nch = (len(data) + chunk - 1) // chunk
    for c in range(nch):
        start = c * chunk
        dl, ln = min(start, 32768), min(chunk, len(data) - start)
        want += O.deflate_chunk(data[start - dl:start + ln], dl, ln, level, 0, 8, c == nch - 1, tune=tune)
    assert out == want
    plain, _ = engine.deflate_chunks(data, level=level, chunk_size=chunk)
    assert plain == O.deflate_chunked(data, chunk, level, 0, 8)[0]        # and the table values are back


@pytest.mark.parametrize("level,kind,bits", [(6, 4, 3), (6, 1, 5), (1, 0, 7), (0, 0, 1), (9, 2, 4), (6, 4, 0)])
def test_primed_bits_shift_the_first_chunk_only(engine, level, kind, bits):
    """deflatePrime residue (deflate.c:464-487): the call's first chunk starts inside byte 0."""
    data = O.corpus(2 * 65536 + 999, kind, 71)
    chunk = 65536
    engine.prime(bits, 0x5a)
    out, offs = engine.deflate_chunks(data, level=level, chunk_size=chunk)
    want = O.deflate_chunk(data[:chunk], 0, chunk, level, 0, 8, False, prime=(bits, 0x5a))
    want += O.deflate_chunk(data[chunk - 32768:2 * chunk], 32768, chunk, level, 0, 8, False)
    want += O.deflate_chunk(data[2 * chunk - 32768:], 32768, 999, level, 0, 8, True)
    assert out == want
    again, _ = engine.deflate_chunks(data, level=level, chunk_size=chunk)      # one call only
    assert again == O.deflate_chunked(data, chunk, level, 0, 8)[0]


def test_parse_driven_matcher_is_exact_too():
    """The matcher that follows the parse (k_follow) instead of searching every position: same bytes, also for
    deflateTune values only it can express (nice_length below max_lazy, a quarter budget of zero)."""
    import zlib_amd
    data = O.corpus(3 * 65536 + 77, 4, 81) + O.corpus(150000, 0, 82) + bytes(70000) + O.corpus(100000, 3, 83)
    chunk = 131072
    os.environ["ZAMD_MATCH_FOLLOW"] = "1"
    try:
        eng = zlib_amd.Engine(0)
    finally:
        del os.environ["ZAMD_MATCH_FOLLOW"]
    for level, strategy in ((4, 0), (6, 0), (6, 1), (6, 4), (9, 0)):
        out, _ = eng.deflate_chunks(data, level=level, strategy=strategy, chunk_size=chunk)
        assert out == O.deflate_chunked(data, chunk, level, strategy, 8)[0], (level, strategy)
    eng.close()


def test_level_1_build_for_eight_wavefronts_per_simd_is_exact_too():
    """k_fastw exists in two builds; the one capped at 64 scalar registers is picked for calls of more than 24 chunks per CU.
    ZAMD_FASTW8=1 picks it for any call (read once per process, hence the child process)."""
    import subprocess, sys
    code = r"""
import sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import _oracle as O, zlib_amd
eng = zlib_amd.Engine(0)
data = O.corpus(5 * 65536 + 4321, 0, 91) + O.corpus(200000, 4, 92) + bytes(40000) + O.corpus(65536, 1, 93)
for level, strategy, mem, chunk in ((1, 0, 8, 65536), (1, 0, 9, 131072), (1, 4, 1, 65536), (1, 1, 8, 32768)):
    out, _ = eng.deflate_chunks(data, level=level, strategy=strategy, mem_level=mem, chunk_size=chunk)
    assert out == O.deflate_chunked(data, chunk, level, strategy, mem)[0], (level, strategy, mem, chunk)
eng.tune(4, 258, 5, 3)
out, _ = eng.deflate_chunks(data, level=2, chunk_size=65536)
want = b""
nch = (len(data) + 65535) // 65536
for c in range(nch):
    start = c * 65536
    dl, ln = min(start, 32768), min(65536, len(data) - start)
    want += O.deflate_chunk(data[start - dl:start + ln], dl, ln, 2, 0, 8, c == nch - 1, tune=(4, 258, 5, 3))
assert out == want
eng.close()
print("ok")
""" % (os.path.join(ROOT, "tests"), ROOT)
    env = dict(os.environ, ZAMD_FASTW8="1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout + r.stderr


def test_search_along_four_byte_chains_is_exact_at_every_level():
    """Chain budgets from 256 on (levels 7-9) search along the four-byte chains (k_reach, k_first3, k_match4), smaller
    ones along the hash chains (k_match).  ZAMD_MATCH4_MIN_CHAIN=4 sends every budget the former way (read once per
    process, hence the child process): levels 4-9, every strategy that searches, hash sizes, small windows, budgets
    that are no power of two, ragged sizes, runs and periods up to MAX_DIST."""
    import subprocess, sys
    code = r"""
import sys, random
sys.path.insert(0, %r); sys.path.insert(0, %r)
import _oracle as O, zlib_amd
eng = zlib_amd.Engine(0)
rnd = random.Random(5)
cases = []
for kind in range(5):
    cases.append((O.corpus((1 << 20) + 4321, kind, 17 + kind), 6, 0, 8, 1 << 19))
for lvl in (4, 5, 6, 7, 8, 9):
    cases.append((O.corpus(300000, 4, 21), lvl, 0, 8, 65536))
for strat in (1, 4):
    cases.append((O.corpus(700001, 3, 22), 6, strat, 8, 1 << 18))
    cases.append((O.corpus(150000, 2, 23), 9, strat, 8, 65536))
for mem in (1, 2, 4, 7, 9):
    cases.append((O.corpus(400000, 4, 24), 6, 0, mem, 131072))
for n in (0, 1, 2, 3, 4, 5, 261, 262, 263, 32767, 32768, 32769, 65274, 65277, 65374, 65535, 65536, 65537, 98304 + 32600):
    cases.append((O.corpus(n, 0, 25), 6, 0, 8, 65536))
    cases.append((O.corpus(n, 4, 26), 4, 0, 8, 32768))
cases.append((bytes(300000), 6, 0, 8, 65536))
cases.append((bytes(rnd.choice(b"ab") for _ in range(150000)), 5, 0, 8, 65536))
for per in (1, 2, 3, 4, 5, 258, 259, 300, 32505, 32506, 32507):
    base = bytes(rnd.getrandbits(8) for _ in range(per))
    cases.append(((base * (120000 // per + 2))[:120000], 6, 0, 8, 65536))
for data, level, strategy, mem, chunk in cases:
    out, _ = eng.deflate_chunks(data, level=level, strategy=strategy, mem_level=mem, chunk_size=chunk)
    assert out == O.deflate_chunked(data, chunk, level, strategy, mem)[0], (len(data), level, strategy, mem, chunk)
data = O.corpus(3 * 65536 + 1234, 4, 61) + O.corpus(100000, 0, 62)
for level, tune in ((6, (4, 4, 16, 8)), (6, (8, 258, 258, 300)), (9, (32, 16, 100, 2000)), (5, (40, 300, 1000, 17)), (6, (8, 16, 128, 100)), (4, (4, 4, 4, 5))):
    eng.tune(*tune)
    out, _ = eng.deflate_chunks(data, level=level, chunk_size=65536)
    eng.tune()
    want = b""
    nch = (len(data) + 65535) // 65536
    for c in range(nch):
        start = c * 65536
        dl, ln = min(start, 32768), min(65536, len(data) - start)
        want += O.deflate_chunk(data[start - dl:start + ln], dl, ln, level, 0, 8, c == nch - 1, tune=tune)
    assert out == want, (level, tune)
data = O.corpus(400000, 0, 63) + O.corpus(100000, 3, 64)
for wbits in (9, 10, 12, 14):
    eng.window_bits(wbits)
    out, _ = eng.deflate_chunks(data, level=6, chunk_size=65536)
    eng.window_bits(15)
    want = b""
    nch = (len(data) + 65535) // 65536
    for c in range(nch):
        start = c * 65536
        dl, ln = min(start, 1 << wbits), min(65536, len(data) - start)
        want += O.deflate_chunk(data[start - dl:start + ln], dl, ln, 6, 0, 8, c == nch - 1, wbits=wbits)
    assert out == want, wbits
eng.close()
print("ok")
""" % (os.path.join(ROOT, "tests"), ROOT)
    env = dict(os.environ, ZAMD_MATCH4_MIN_CHAIN="4")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout + r.stderr


@pytest.mark.parametrize("level,tune", [(6, (4, 64, 8, 128)), (5, (20, 258, 9, 3)), (8, (1, 16, 4, 2)), (6, (33, 1, 8, 2)), (6, (14, 2, 9, 1))])
def test_tuned_parameters_outside_the_tables_shape(engine, level, tune):
    """nice_length below max_lazy, chain budgets whose quarter is zero, lazy limits below 3."""
    data = O.corpus(2 * 65536 + 1234, 0, 85) + O.corpus(60000, 3, 86)
    chunk = 65536
    engine.tune(*tune)
    try:
        out, offs = engine.deflate_chunks(data, level=level, chunk_size=chunk)
    finally:
        engine.tune()
    want = b""
    nch = (len(data) + chunk - 1) // chunk
    for c in range(nch):
        start = c * chunk
        dl, ln = min(start, 32768), min(chunk, len(data) - start)
        want += O.deflate_chunk(data[start - dl:start + ln], dl, ln, level, 0, 8, c == nch - 1, tune=tune)
    assert out == want


from test_oracle_golden import boundary_input, _ZFLUSH


@pytest.mark.parametrize("r", GOLD["boundary"], ids=lambda r: "k%d_s%d_n%d_d%d_L%d_S%d_W%d_F%d" % (
    r["kind"], r["seed"], r["n"], r["dict"], r["level"], r["strategy"], r["wbits"], r["flush"]))
def test_boundary_rows_from_reference_build(engine, r):
    """One chunk as the stream API's corner cases make it: a dictionary of any length in front (deflate.c:323-389),
    windowBits 9..14 (deflate.c:271-291), ended by any flush value (deflate.c:923-945, trees.c:892-901), with the bits
    the reference keeps in its bit buffer and strm->data_type (trees.c:1120-1147) — against the reference's own results."""
    buf = boundary_input(r)
    fl = _ZFLUSH[r["flush"]]
    engine.window_bits(r["wbits"])
    try:
        if fl in (O.FLUSH_PARTIAL, O.FLUSH_BLOCK):
            engine.tail(1 if fl == O.FLUSH_PARTIAL else 2)
        out, _ = engine.deflate_chunks(buf[r["dict"]:], level=r["level"], strategy=r["strategy"], chunk_size=1 << 20,
                                       finish=fl == O.FLUSH_FINISH, dictionary=buf[:r["dict"]])
        pb, pv = engine.pending_bits()
        dt = engine.data_type()
    finally:
        engine.window_bits(15)
    assert len(out) == r["out_len"] and "%08x" % zlib.crc32(out) == r["out_crc"]
    assert (pb, pv, dt) == (r["pend_bits"], r["pend_val"], r["data_type"])


@pytest.mark.parametrize("wbits", [9, 10, 11, 12, 13, 14])
def test_small_windows_over_many_chunks(engine, wbits):
    """windowBits below 15 over several chunks (each primed with the previous 1 << wbits bytes), all parsers."""
    engine.window_bits(wbits)
    try:
        for kind, n, level, strategy, mem, chunk in ((4, 300001, 6, 0, 8, 65536), (0, 200000, 1, 0, 8, 32768), (3, 150000, 9, 1, 8, 65536),
                                                     (2, 100000, 4, 3, 3, 32768), (4, 250000, 3, 0, 9, 131072), (1, 140000, 0, 0, 8, 65536),
                                                     (0, 65536 + 65274 - (1 << wbits), 6, 0, 8, 65536), (4, 90000, 6, 2, 8, 32768)):
            data = O.corpus(n, kind, 200 + wbits)
            out, lens = _gpu(engine, data, level=level, strategy=strategy, mem_level=mem, chunk_size=chunk)
            want, wlens = O.deflate_chunked(data, chunk, level, strategy, mem, wbits=wbits)
            assert lens == wlens and out == want, (kind, n, level, strategy, mem, chunk)
            assert zlib.decompress(out, -15) == data
    finally:
        engine.window_bits(15)


@pytest.mark.parametrize("dlen", [1, 2, 3, 15, 17, 1001, 12345, 32767, 32768])
def test_dictionary_of_any_length_then_several_chunks(engine, dlen):
    """A preset dictionary of any length primes the first chunk; the later chunks see (dictionary ++ data) as history."""
    d = O.corpus(dlen, 0, 300)
    data = O.corpus(3 * 65536 + 99, 0, 301)
    for level in (1, 2, 3, 6):
        out, lens = _gpu(engine, data, level=level, chunk_size=65536, dictionary=d)
        hist = d + data
        want = b""
        for c in range(4):
            start = dlen + c * 65536
            ln = min(65536, len(hist) - start)
            dl = min(start, 32768)
            want += O.deflate_chunk(hist[start - dl:start + ln], dl, ln, level, 0, 8, final=(c == 3))
        assert out == want, (dlen, level)


@pytest.mark.parametrize("flush", [O.FLUSH_PARTIAL, O.FLUSH_BLOCK])
def test_partial_flush_and_block_in_the_middle_of_a_stream(engine, flush):
    """A call of several chunks ended by Z_PARTIAL_FLUSH / Z_BLOCK: the chunks inside still end on sync markers, the last
    one as the flush value says; the bits left over lead the next call (zamd_engine_prime), and the whole is one stream."""
    data = O.corpus(2 * 65536 + 5000, 0, 310) + O.corpus(70000, 4, 311)
    cut = 2 * 65536 + 5000
    engine.tail(1 if flush == O.FLUSH_PARTIAL else 2)
    a, _ = engine.deflate_chunks(data[:cut], level=6, chunk_size=65536, finish=False)
    pb, pv = engine.pending_bits()
    want = b""
    for c in range(3):
        s0 = c * 65536
        ln = min(65536, cut - s0)
        dl = min(s0, 32768)
        if c < 2:
            want += O.deflate_chunk(data[s0 - dl:s0 + ln], dl, ln, 6, 0, 8, final=False)
        else:
            w, wpb, wpv, _ = O.deflate_chunk(data[s0 - dl:s0 + ln], dl, ln, 6, 0, 8, flush=flush, want_state=True)
            want += w
    assert a == want and (pb, pv) == (wpb, wpv)
    if pb:
        engine.prime(pb, pv)
    b, _ = engine.deflate_chunks(data[cut:], level=6, chunk_size=65536, finish=True, dictionary=data[cut - 32768:cut])
    # the second call's first byte holds the leftover bits in its low end: it replaces nothing of the first call
    assert zlib.decompress(a + b, -15) == data


@pytest.mark.parametrize("n,base", [(1, 0), (7, 5), (1024, 0), (1025, 1 << 40), (70001, 123)])
def test_chunk_offsets_scan(engine, n, base):
    """zamd_chunk_offsets_dev: where the chunks of ranks that compress side by side go (the total_out a serial writer
    accumulates, deflate.c:642-662)."""
    import numpy as np
    rng = np.random.default_rng(n)
    lens = rng.integers(0, 1 << 21, size=n, dtype=np.uint64)
    d_len = engine.alloc(8 * n); d_len.upload(lens.tobytes())
    d_off = engine.alloc(8 * (n + 1))
    engine.chunk_offsets(d_len.ptr, n, d_off.ptr, base)
    got = np.frombuffer(bytes(d_off.download(8 * (n + 1))), dtype=np.uint64)
    want = np.concatenate(([0], np.cumsum(lens))).astype(np.uint64) + np.uint64(base)
    assert (got == want).all()
    d_len.free(); d_off.free()


def test_random_parity_cases():
    """tests/gpu_parity_fuzz.py: corpus kind, length, level, strategy, memLevel and chunk size at random, and inputs spliced
    from pieces of different kinds, runs and repeats; bytes and chunk lengths equal the oracle's.  (920 cases were run when
    the tool was written; forty here.)"""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "gpu_parity_fuzz.py"), "40", "31"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "bad 0" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
