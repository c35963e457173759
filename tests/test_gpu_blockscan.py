"""GPU tests: parallel inflate of streams without sync markers (what every other producer writes).
zamd_find_blocks_dev guesses starts of dynamic blocks; segments in ZAMD_INF_BLOCKSTOP mode confirm them; inflate()
of libzamd_z.so builds its segment list from them.  The producer here is the system zlib (Python's zlib module):
one stream, no flushes.  Byte work: the bar is bit-exact."""
import time
import zlib

import pytest

import _oracle as O
import _zapi as Z
import zlib_amd

pytestmark = pytest.mark.gpu
STEP = 128 * 8192          # bits between targets


def _raw(data, level):
    c = zlib.compressobj(level, zlib.DEFLATED, -15)
    return c.compress(data) + c.flush()


@pytest.mark.parametrize("kind,level", [(0, 6), (3, 9), (4, 6), (2, 1)])
def test_guessed_starts_are_block_starts_and_segments_chain(engine, kind, level):
    data = O.corpus(24 * (1 << 20) + 4321, kind, 300 + kind)
    comp = _raw(data, level)
    end = 8 * len(comp)
    targets = list(range(STEP, end - STEP // 2, STEP))
    found = engine.find_blocks(comp, targets, STEP)
    hits = sorted({f for f in found if f is not None})
    assert all(t <= f < t + STEP for t, f in zip(targets, found) if f is not None)
    if kind != 4:                                       # the mixed corpus has long stretches of stored blocks
        assert len(hits) >= 0.8 * len(targets)
    # counting pass: a segment stops at the first block boundary at or behind the next guess; a guess that is
    # hit exactly is a block start, one that is passed is not (rare: dropped here, repaired by inflate())
    starts = [0] + hits
    wrong = 0
    for attempt in range(8):
        segs = []
        for i, b in enumerate(starts):
            last = i + 1 == len(starts)
            segs.append(dict(in_bit=b, in_end_bit=end, out_off=0, out_cap=1 << 40, win_avail=0 if i == 0 else 32768,
                             mode=0 if last else zlib_amd.INF_BLOCKSTOP, stop_bit=0 if last else starts[i + 1]))
        res, _ = engine.inflate_segments(comp, segs)
        bad = set()
        i = 0
        while i + 1 < len(starts):                       # follow the chain from the true start
            r = res[i]
            assert r["status"] == 2, (i, r)
            if r["end_bit"] != starts[i + 1]:
                bad.add(starts[i + 1])
                break
            i += 1
        if not bad:
            break
        wrong += len(bad)
        starts = [b for b in starts if b not in bad]
    else:
        raise AssertionError("too many wrong guesses")
    assert wrong <= max(1, len(hits) // 20), (wrong, len(hits))
    assert res[-1]["status"] == 1
    assert sum(r["out_len"] for r in res) == len(data)
    # placing pass
    off = 0
    for g, r in zip(segs, res):
        g.update(in_end_bit=r["end_bit"], out_off=off, out_cap=r["out_len"], mode=zlib_amd.INF_EXACT, win_avail=min(off, 32768))
        off += r["out_len"]
    res2, out = engine.inflate_segments(comp, segs, out_bytes=len(data))
    assert all(r["status"] in (0, 1, 2) for r in res2), [r for r in res2 if r["status"] >= 3][:2]
    assert out == data


def test_nothing_is_found_where_nothing_is(engine):
    noise = O.corpus(1 << 20, 1, 77)
    assert engine.find_blocks(noise, [0, 1 << 20, 3 << 20], 1 << 20) == [None, None, None]
    stored = _raw(noise, 0)                              # stored blocks only
    assert engine.find_blocks(stored, [0, 4 << 20], 1 << 20) == [None, None]


@pytest.mark.parametrize("wbits,level", [(15, 6), (31, 9), (-15, 1)])
def test_inflate_of_a_foreign_stream(wbits, level):
    data = O.corpus(96 * (1 << 20) + 12345, 0 if level != 9 else 3, 310 + level)
    c = zlib.compressobj(level, zlib.DEFLATED, wbits)
    comp = c.compress(data) + c.flush()
    t0 = time.perf_counter()
    rc, out, msg, rest, _ = Z.inflate_stream(comp, wbits=wbits, out_step=len(data) + 64)
    dt = time.perf_counter() - t0
    assert rc == Z.Z_STREAM_END
    assert out == data
    print("inflate() of a %d MiB foreign stream (level %d, windowBits %d): %.0f ms" % (len(data) >> 20, level, wbits, dt * 1e3))


def test_inflate_of_a_foreign_stream_in_pieces():
    data = O.corpus(20 * (1 << 20) + 999, 0, 320)
    comp = zlib.compress(data, 6)
    rc, out, msg, rest, _ = Z.inflate_stream(comp, in_step=3 * (1 << 20) + 17, out_step=1 << 20)
    assert rc == Z.Z_STREAM_END and out == data
    # damage in the middle: what lies in front of it is delivered unharmed, and the stream ends in an error (a
    # wrong symbol need not break the format: the reference, too, may only notice at the check value)
    bad = bytearray(comp)
    bad[len(bad) // 2] ^= 0x40
    rc, out, msg, rest, _ = Z.inflate_stream(bytes(bad), out_step=len(data) + 64)
    assert rc == Z.Z_DATA_ERROR and msg
    want_rc, want_msg = None, None
    try:
        zlib.decompress(bytes(bad))
    except zlib.error as ex:
        want_msg = str(ex)
    assert want_msg is not None and msg in want_msg            # the system zlib words its errors as the reference does
    k = len(data) * 2 // 5
    assert out[:k] == data[:k]


def test_planted_block_headers_are_guessed_and_repaired(engine):
    """Stored blocks that carry the bytes of a real dynamic-block header: the search takes them for block starts,
    the segments in front of them end elsewhere, and inflate() mends the chain in a further pass."""
    text = O.corpus(12 * (1 << 20), 0, 330)
    header = _raw(text[:1 << 20], 6)[:256]               # a dynamic block's first bytes (BFINAL 0: the input is long)
    assert header[0] & 7 == 4
    noise = bytearray(O.corpus(6 * (1 << 20), 1, 331))
    planted = list(range(5000, len(noise) - 4096, 40000))
    for at in planted:
        noise[at:at + len(header)] = header
    data = text[:6 << 20] + bytes(noise) + text[6 << 20:]
    comp = _raw(data, 6)
    # the planted bytes are in the stream verbatim (stored blocks), and the search finds some of them
    first = comp.find(header, 4096)                       # (the stream itself begins with these bytes)
    assert first > 0
    hits = engine.find_blocks(comp, [first * 8 - 4096], 1 << 16)
    assert hits == [first * 8]
    rc, out, msg, rest, _ = Z.inflate_stream(b"\x78\x9c" + comp + zlib.adler32(data).to_bytes(4, "big"), out_step=len(data) + 64)
    assert rc == Z.Z_STREAM_END, msg
    assert out == data
