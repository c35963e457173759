"""Batch inflate on the GPU (through the C ABI): round trips of the engine's own
chunked streams, foreign single streams, the reference's raw-inflate test vectors
and the segment interface.  Byte work: the bar is bit-exact."""
import json
import os
import zlib

import pytest

import _oracle as O
import zlib_amd

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VEC = json.load(open(os.path.join(ROOT, "tests", "golden", "inflate_vectors.json")))["raw"]


def _one_segment(engine, comp, out_cap=1 << 22, **kw):
    seg = dict(in_bit=0, in_end_bit=8 * len(comp), out_off=0, out_cap=out_cap, win_avail=0, mode=0)
    seg.update(kw)
    return seg


@pytest.mark.parametrize("v", VEC, ids=lambda v: v["hex"][:20].replace(" ", ""))
def test_reference_raw_vectors(engine, v):
    comp = bytes.fromhex(v["hex"].replace(" ", ""))
    rc, want, used, msg = O.inflate_raw(comp, b"", 1 << 20)
    assert msg == v["msg"]
    res, _ = engine.inflate_segments(comp, [_one_segment(engine, comp)])
    r = res[0]
    if v["msg"] is None:
        assert r["status"] == 1 and r["final_seen"] == 1
        assert r["out_len"] == len(want)
        res, out = engine.inflate_segments(comp, [_one_segment(engine, comp, out_cap=len(want))], out_bytes=max(16, len(want)))
        assert res[0]["status"] == 1
        assert out[:len(want)] == want
    else:
        assert r["status"] >= 16
        assert r["msg"] == v["msg"]


CASES = []
for kind in range(5):
    CASES.append((kind, (1 << 20) * 3 + 4321, 6, 0, 1 << 20))
    CASES.append((kind, 700001, 9, 0, 65536))
for strat in (1, 2, 3, 4):
    CASES.append((3, 500001, 6, strat, 1 << 18))
for n in (0, 1, 5, 258, 32768, 32769, 65536, 65537, 100000):
    CASES.append((0, n, 6, 0, 32768))
    CASES.append((4, n, 4, 0, 65536))


@pytest.mark.parametrize("kind,n,level,strategy,chunk", CASES)
def test_round_trip_of_chunked_streams(engine, kind, n, level, strategy, chunk):
    data = O.corpus(n, kind, 23 + kind)
    stream, offs = engine.deflate_chunks(data, level=level, strategy=strategy, chunk_size=chunk)
    out = engine.inflate_chunks(stream, offs, n, chunk_size=chunk)
    assert out == data


SPLIT_CASES = [(kind, (1 << 20) * 3 + 4321, 6, 0, 1 << 20, parts) for kind in range(5) for parts in (2, 5)]
SPLIT_CASES += [(4, 700001, 9, 0, 65536, 3), (0, 700001, 1, 0, 131072, 16), (3, 500001, 6, 2, 1 << 18, 4), (3, 500001, 6, 3, 1 << 18, 4),
                (1, 400000, 0, 0, 1 << 18, 4), (2, (1 << 20) + 5, 6, 0, 1 << 20, 8)]
SPLIT_CASES += [(0, n, 6, 0, 32768, 4) for n in (0, 1, 5, 32768, 32769, 100000)]


@pytest.mark.parametrize("kind,n,level,strategy,chunk,parts", SPLIT_CASES)
def test_round_trip_with_several_wavefronts_to_a_chunk(engine, kind, n, level, strategy, chunk, parts):
    # the deflate call records block boundaries inside every chunk; the inflate call starts a wavefront at each
    data = O.corpus(n, kind, 31 + kind)
    stream, offs, splits = engine.deflate_chunks(data, level=level, strategy=strategy, chunk_size=chunk, parts=parts)
    plain, offs2 = engine.deflate_chunks(data, level=level, strategy=strategy, chunk_size=chunk)
    assert stream == plain and offs == offs2                      # recording changes nothing in the stream
    nch = len(offs) - 1
    assert len(splits) == nch * (parts - 1)
    for c in range(nch):
        recs = [r for r in splits[c * (parts - 1):(c + 1) * (parts - 1)]]
        live = [r for r in recs if r]
        assert recs == live + [0] * (len(recs) - len(live))          # packed to the front
        assert live == sorted(live) and len(set(live)) == len(live)
        clen = min(chunk, n - c * chunk)
        for r in live:
            bit, off = r >> 32, r & 0xffffffff
            assert 0 < off < clen and 0 < bit < 8 * (offs[c + 1] - offs[c])
            # a serial decoder that starts at the recorded bit with the data in front of it as window produces the rest
            rc, want, used, msg = O.inflate_raw_at(stream[offs[c]:offs[c + 1]], bit, data[max(0, c * chunk + off - 32768):c * chunk + off], clen - off)
            assert want == data[c * chunk + off:c * chunk + clen], (c, r, msg)
    if n >= 3 * chunk and chunk >= (1 << 20) and level >= 4 and kind != 1:
        assert any(splits), "chunks of several blocks must come with split points"
    assert engine.inflate_chunks(stream, offs, n, chunk_size=chunk, splits=splits, parts=parts) == data
    if n > chunk + 40000:
        # a shard of the stream with the window in front of it
        k = 1
        out = engine.inflate_chunks(stream[offs[k]:], [o - offs[k] for o in offs[k:]], n - k * chunk, chunk_size=chunk,
                                    head_window=data[k * chunk - 32768:k * chunk], splits=splits[k * (parts - 1):], parts=parts)
        assert out == data[k * chunk:]


def test_split_records_are_hints(engine):
    # records that are out of order or outside the chunk end the chunk's list; a record that is not a block boundary
    # makes the chunk fail like other bad data; nothing reads or writes outside its buffers
    data = O.corpus(4 * 262144 + 999, 0, 6)
    chunk, parts = 262144, 4
    stream, offs, splits = engine.deflate_chunks(data, level=6, chunk_size=chunk, parts=parts)
    assert all(splits[c * 3] for c in range(4))
    junk = list(splits)
    junk[0], junk[1] = splits[1], splits[0]                          # out of order: the second ends the list
    junk[3] = (0xfffffff0 << 32) | 5                                   # a bit offset far outside chunk 1
    junk[6] = (splits[6] & ~0xffffffff) | (chunk + 7)                # an output offset outside chunk 2
    assert engine.inflate_chunks(stream, offs, len(data), chunk_size=chunk, splits=junk, parts=parts) == data
    wrong = list(splits)
    wrong[4] += 3 << 32                                              # three bits behind a block boundary of chunk 1
    with pytest.raises(zlib_amd.ZamdError) as ei:
        engine.inflate_chunks(stream, offs, len(data), chunk_size=chunk, splits=wrong, parts=parts)
    assert ei.value.code == zlib_amd.Z_DATA_ERROR and ei.value.segment == 1


def test_long_range_window_propagation(engine):
    # every chunk starts by copying from the previous one, and runs of one byte cross many chunks
    data = b"\0" * 200000 + O.corpus(100000, 0, 3) + b"\xff" * 150000 + O.corpus(70000, 0, 3) * 3
    for chunk in (32768, 65536):
        stream, offs = engine.deflate_chunks(data, level=9, chunk_size=chunk)
        assert engine.inflate_chunks(stream, offs, len(data), chunk_size=chunk) == data
    # parts shorter than the window: a byte may come from the part before the part before
    stream, offs, splits = engine.deflate_chunks(data, level=1, mem_level=1, chunk_size=131072, parts=16)
    assert sum(1 for r in splits if r) > 40
    assert engine.inflate_chunks(stream, offs, len(data), chunk_size=131072, splits=splits, parts=16) == data


def test_more_segments_than_wavefront_slots_and_chains_as_long_as_the_data(engine):
    """More segments than the GPU holds wavefronts of the decode kernel (5 120): the launch takes them longest first
    (k_inf_order), and references that run back through every segment — a run of one byte from the first chunk to the
    last — are followed by the jump rounds, eight steps a round.  Zeros, a period, and text between runs."""
    chunk, n_chunks = 32768, 6000
    n = chunk * n_chunks
    text = O.corpus(3 << 20, 4, 9)
    cases = [
        b"\0" * n,
        (b"abcdefg" * (n // 7 + 1))[:n],
        (b"\x55" * (40 << 20) + text + b"\x55" * n)[:n],
        (text * (n // len(text) + 1))[:n],
    ]
    for data in cases:
        for level in (1, 6):
            stream, offs = engine.deflate_chunks(data, level=level, chunk_size=chunk)
            assert len(offs) - 1 == n_chunks
            assert engine.inflate_chunks(stream, offs, n, chunk_size=chunk) == data


@pytest.mark.parametrize("level,wbits_data", [(1, 0), (6, 1), (9, 4)])
def test_foreign_single_stream(engine, level, wbits_data):
    # a stream that was not cut into chunks (system zlib as the producer): one segment, one wavefront
    data = O.corpus(600000, wbits_data, 5)
    c = zlib.compressobj(level, zlib.DEFLATED, -15)
    comp = c.compress(data) + c.flush()
    res, _ = engine.inflate_segments(comp, [_one_segment(engine, comp)])
    assert res[0]["status"] == 1 and res[0]["out_len"] == len(data)
    assert res[0]["end_bit"] <= 8 * len(comp) and res[0]["end_bit"] > 8 * (len(comp) - 1)
    res, out = engine.inflate_segments(comp, [_one_segment(engine, comp, out_cap=len(data))], out_bytes=len(data))
    assert out == data


def test_segments_stop_at_sync_markers_and_chain(engine):
    data = O.corpus(300000, 4, 9)
    c = zlib.compressobj(6, zlib.DEFLATED, -15)
    pieces = [data[0:1000], data[1000:120000], data[120000:120001], data[120001:]]
    comp, cuts = b"", [0]
    for i, p in enumerate(pieces):
        comp += c.compress(p) + c.flush(zlib.Z_SYNC_FLUSH if i < len(pieces) - 1 else zlib.Z_FINISH)
        cuts.append(len(comp))
    segs = [dict(in_bit=8 * cuts[i], in_end_bit=8 * len(comp), out_off=0, out_cap=1 << 30, win_avail=min(32768, sum(map(len, pieces[:i]))),
                 mode=zlib_amd.INF_SYNCSTOP) for i in range(len(pieces))]
    res, _ = engine.inflate_segments(comp, segs)
    for i, r in enumerate(res):
        assert r["status"] == (2 if i < len(pieces) - 1 else 1), r
        assert r["out_len"] == len(pieces[i])
        assert r["end_bit"] == 8 * cuts[i + 1] or i == len(pieces) - 1
    off = 0
    for s, r in zip(segs, res):
        s["out_off"], s["out_cap"], s["in_end_bit"] = off, r["out_len"], r["end_bit"]
        off += r["out_len"]
    res, out = engine.inflate_segments(comp, segs, out_bytes=len(data))
    assert out == data


@pytest.mark.parametrize("n,step", [(3000000, 131072), (3000000, 4096), (200000, 1 << 20), (9000000, 65536)])
def test_counting_pass_reports_block_marks(engine, n, step):
    """The block marks of a counting pass (zamd_inflate_block_marks) are boundaries between blocks: cut there, the
    segment inflates part by part to the same bytes (the parts' windows are the output in front of them).  A segment with
    more blocks than a row holds reports marks spread over all of it; the request holds for one call."""
    data = O.corpus(n, 4, 77)
    c = zlib.compressobj(6, zlib.DEFLATED, -15)
    half = n // 2
    comp = c.compress(data[:half]) + c.flush(zlib.Z_SYNC_FLUSH)
    cut = len(comp)
    comp += c.compress(data[half:]) + c.flush()
    segs = [dict(in_bit=0, in_end_bit=8 * len(comp), out_off=0, out_cap=1 << 30, win_avail=0, mode=zlib_amd.INF_SYNCSTOP),
            dict(in_bit=8 * cut, in_end_bit=8 * len(comp), out_off=0, out_cap=1 << 30, win_avail=32768, mode=zlib_amd.INF_SYNCSTOP)]
    res, _ = engine.inflate_segments(comp, segs, mark_step=step)
    assert [r["status"] for r in res] == [2, 1] and [r["out_len"] for r in res] == [half, n - half]
    parts, base = [], 0
    for s, r in zip(segs, res):
        marks = r["marks"]
        assert len(marks) <= zlib_amd.INF_MARKS - 1
        assert marks == sorted(set(marks)) and all(0 < o <= r["out_len"] and 0 < b < r["end_bit"] - s["in_bit"] for b, o in marks)
        marks = [m for m in marks if m[1] < r["out_len"]]        # the boundary in front of a closing empty block cuts nothing off
        if r["out_len"] >= 4 * step and step >= 65536:
            assert len(marks) >= 3
        if marks and len(marks) < zlib_amd.INF_MARKS - 1 and step >= 65536:        # nothing dropped: one mark per step of output at least
            assert marks[0][1] >= step and all(b[1] - a[1] >= step for a, b in zip(marks, marks[1:]))
        if len(marks) >= 7:                            # spread over the whole segment
            assert marks[-1][1] > r["out_len"] // 2
        edges = [(0, 0)] + marks + [(r["end_bit"] - s["in_bit"], r["out_len"])]
        for (b0, o0), (b1, o1) in zip(edges, edges[1:]):
            parts.append(dict(in_bit=s["in_bit"] + b0, in_end_bit=s["in_bit"] + b1, out_off=base + o0, out_cap=o1 - o0,
                              win_avail=min(32768, base + o0), mode=zlib_amd.INF_EXACT))
        base += r["out_len"]
    res2, out = engine.inflate_segments(comp, parts, out_bytes=n)
    assert all(r["status"] in (0, 1) for r in res2), res2
    assert out == data
    # the next counting pass was not asked for marks
    res3, _ = engine.inflate_segments(comp, segs)
    assert "marks" not in res3[0] and [r["out_len"] for r in res3] == [half, n - half]


@pytest.mark.parametrize("mode", ["0", "1"])
def test_literal_pairs_always_and_never(mode):
    """The speculative decode takes two literals in a row as one token where a segment's tokens are short (the other
    tests run that way).  ZAMD_INF_PAIRS=1 does it everywhere, 0 nowhere (read once per process, hence the child
    process): the same bytes, the same lengths and block boundaries, the same messages for the reference's bad streams."""
    import subprocess, sys
    code = r"""
import json, os, sys, zlib
sys.path.insert(0, %r); sys.path.insert(0, %r)
import _oracle as O, zlib_amd
eng = zlib_amd.Engine(0)
for kind in range(5):
    for level, strategy, chunk, parts in ((6, 0, 1 << 20, 1), (1, 0, 65536, 4), (9, 0, 262144, 4), (6, 2, 262144, 4), (6, 3, 262144, 1), (0, 0, 65536, 1)):
        n = 1500000 + 777 * kind
        data = O.corpus(n, kind, 50 + kind)
        if parts > 1:
            stream, offs, splits = eng.deflate_chunks(data, level=level, strategy=strategy, chunk_size=chunk, parts=parts)
            assert eng.inflate_chunks(stream, offs, n, chunk_size=chunk, splits=splits, parts=parts) == data, (kind, level, strategy)
        else:
            stream, offs = eng.deflate_chunks(data, level=level, strategy=strategy, chunk_size=chunk)
        assert eng.inflate_chunks(stream, offs, n, chunk_size=chunk) == data, (kind, level, strategy)
one = lambda comp, **kw: dict(dict(in_bit=0, in_end_bit=8 * len(comp), out_off=0, out_cap=1 << 22, win_avail=0, mode=0), **kw)
for v in json.load(open(os.path.join(%r, "tests", "golden", "inflate_vectors.json")))["raw"]:
    comp = bytes.fromhex(v["hex"].replace(" ", ""))
    rc, want, used, msg = O.inflate_raw(comp, b"", 1 << 20)
    r = eng.inflate_segments(comp, [one(comp)])[0][0]
    if v["msg"] is None:
        assert r["status"] == 1 and r["out_len"] == len(want)
        res, out = eng.inflate_segments(comp, [one(comp, out_cap=len(want))], out_bytes=max(16, len(want)))
        assert out[:len(want)] == want
    else:
        assert r["status"] >= 16 and r["msg"] == v["msg"], (v, r)
# fixed-code blocks, short codes in rows, input that ends inside a block
for data in (bytes(range(256)) * 40, b"ab" * 30000 + bytes([7, 7, 9]) * 5000, O.corpus(400000, 2, 2)):
    for lvl, strat in ((6, zlib.Z_FIXED), (6, zlib.Z_HUFFMAN_ONLY), (6, 0)):
        c = zlib.compressobj(lvl, zlib.DEFLATED, -15, 8, strat)
        comp = c.compress(data) + c.flush()
        r = eng.inflate_segments(comp, [one(comp)])[0][0]
        assert r["status"] == 1 and r["out_len"] == len(data) and (r["end_bit"] + 7) // 8 == len(comp)
        res, out = eng.inflate_segments(comp, [one(comp, out_cap=len(data))], out_bytes=len(data))
        assert out == data
        cut = len(comp) * 2 // 3
        r = eng.inflate_segments(comp[:cut], [one(comp[:cut])])[0][0]
        assert r["status"] == 3 and r["end_bit"] <= 8 * cut
        if r["out_len"]:
            res, out = eng.inflate_segments(comp[:cut], [one(comp[:cut], out_cap=r["out_len"], in_end_bit=r["end_bit"])], out_bytes=r["out_len"])
            assert res[0]["status"] == 0 and out == data[:r["out_len"]]
print("ok")
""" % (os.path.join(ROOT, "tests"), ROOT, ROOT)
    env = dict(os.environ, ZAMD_INF_PAIRS=mode)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout + r.stderr


def test_truncated_input_reports_last_complete_block(engine):
    data = O.corpus(400000, 0, 2)
    c = zlib.compressobj(6, zlib.DEFLATED, -15)
    comp = c.compress(data) + c.flush()
    cut = len(comp) * 2 // 3
    res, _ = engine.inflate_segments(comp[:cut], [_one_segment(engine, comp[:cut])])
    r = res[0]
    assert r["status"] == 3
    assert 0 < r["out_len"] < len(data) and r["end_bit"] <= 8 * cut
    # the reported prefix is exactly what a serial decoder produces up to that block boundary
    seg = _one_segment(engine, comp[:cut], out_cap=r["out_len"], in_end_bit=r["end_bit"])
    res2, out = engine.inflate_segments(comp[:cut], [seg], out_bytes=r["out_len"])
    assert res2[0]["status"] == 0
    assert out == data[:r["out_len"]]


def test_corrupt_chunk_is_reported(engine):
    data = O.corpus(300000, 0, 8)
    stream, offs = engine.deflate_chunks(data, level=6, chunk_size=65536)
    hits = 0
    for delta in range(0, 40, 3):
        bad = bytearray(stream)
        mid = (offs[2] + offs[3]) // 2 + delta
        bad[mid] ^= 0x55
        try:
            out = engine.inflate_chunks(bytes(bad), offs, len(data), chunk_size=65536)
        except zlib_amd.ZamdError as e:
            assert e.code == zlib_amd.Z_DATA_ERROR and e.segment == 2 and e.msg
            hits += 1
            continue
        # raw deflate carries no checksum: a flipped literal can still be a valid stream of the same length,
        # but then only chunk 2 and what copies from it may differ
        assert out[:2 * 65536] == data[:2 * 65536] and out != data
    assert hits > 0


def test_shard_with_head_window(engine):
    data = O.corpus(500000, 4, 12)
    chunk = 65536
    stream, offs = engine.deflate_chunks(data, level=6, chunk_size=chunk)
    k = 3                                   # second shard starts at chunk k
    sub = stream[offs[k]:]
    sub_offs = [o - offs[k] for o in offs[k:]]
    head = data[k * chunk - 32768:k * chunk]
    out = engine.inflate_chunks(sub, sub_offs, len(data) - k * chunk, chunk_size=chunk, head_window=head)
    assert out == data[k * chunk:]


def test_results_do_not_depend_on_stale_device_memory(engine):
    # workspace buffers are reused between calls: poison freed memory, then run small and ragged cases
    for fill in (0x55, 0xff):
        junk = [engine.alloc(96 << 20) for _ in range(4)]
        for jbuf in junk:
            engine.lib.zamd_dev_memset(engine.handle, jbuf.ptr, fill, jbuf.nbytes)
        for jbuf in junk:
            jbuf.free()
        for n, chunk in ((1, 32768), (40000, 32768), (100001, 65536), (3 * 65536 + 7, 65536)):
            data = O.corpus(n, 0, 77)
            stream, offs = engine.deflate_chunks(data, level=6, chunk_size=chunk)
            assert engine.inflate_chunks(stream, offs, n, chunk_size=chunk) == data
            comp = zlib.compressobj(6, zlib.DEFLATED, -15)
            blob = comp.compress(data) + comp.flush()
            res, out = engine.inflate_segments(blob, [_one_segment(engine, blob, out_cap=n, mode=zlib_amd.INF_EXACT)], out_bytes=max(n, 16))
            assert res[0]["status"] == 1 and out[:n] == data


def test_sync_marker_search_on_the_gpu(engine):
    """zamd_find_sync_dev: every 00 00 FF FF of a buffer, also overlapping ones and those at the very ends."""
    import re
    import random
    rnd = random.Random(7)
    buf = bytearray(rnd.getrandbits(8) for _ in range(300000))
    marks = [0, 4, 17, 1000, 65530, 65536, 131071, len(buf) - 4]
    for m in marks:
        buf[m:m + 4] = b"\x00\x00\xff\xff"
    buf[200000:200006] = b"\x00\x00\x00\xff\xff\xff"
    data = bytes(buf)
    want = [m.start() + 4 for m in re.finditer(b"(?=\x00\x00\xff\xff)", data)]
    assert set(m + 4 for m in marks) <= set(want) and 200005 in want
    assert engine.find_sync(data) == want
    assert engine.find_sync(data, start=65534) == [w for w in want if w - 4 >= 65534]
    assert engine.find_sync(data[:3]) == [] and engine.find_sync(b"\x00\x00\xff\xff") == [4]
    # the chunk boundaries of an engine stream are among them
    src = O.corpus(5 * 65536 + 99, 4, 201)
    stream, offs = engine.deflate_chunks(src, level=6, chunk_size=65536)
    found = set(engine.find_sync(stream))
    assert all(o in found for o in offs[1:-1])


@pytest.mark.parametrize("kind,n,chunk,cuts", [(4, 7 * 65536 + 1234, 65536, (3, 5)), (0, 5 * 32768, 32768, (1, 2, 4)),
                                               (-1, 6 * 65536 + 9, 65536, (2, 4)), (3, 2 * (1 << 20) + 77, 1 << 20, (1,)),
                                               (-2, 4 * 32768 + 100, 32768, (1, 2, 3))])
def test_sharded_inflate_hands_the_window_from_shard_to_shard(kind, n, chunk, cuts):
    """Consecutive shards of one stream on engines of their own, as ranks would hold them: every shard is decoded
    without the window in front of it (begin), the 32 KiB tail windows go down the line (tail), then the shards
    complete (finish).  kind -1: zeros, every byte a copy of the one before, so the last window of every shard
    depends on the first byte of the stream; kind -2: a period of 40 000, longer than the window, ragged end."""
    import struct
    if kind == -1:
        data = bytes(n)
    elif kind == -2:
        per = O.corpus(40000, 1, 77)
        data = (per * (n // 40000 + 1))[:n]
    else:
        data = O.corpus(n, kind, 31)
    e0 = zlib_amd.Engine(0)
    stream, offs = e0.deflate_chunks(data, level=6, chunk_size=chunk)
    nchunks = len(offs) - 1
    bounds = [0] + list(cuts) + [nchunks]
    engines, bufs, wins = [], [], []
    try:
        # begin on every shard before any window exists
        for r in range(len(bounds) - 1):
            c0, c1 = bounds[r], bounds[r + 1]
            e = zlib_amd.Engine(0)
            comp = stream[offs[c0]:offs[c1]]
            rel = [o - offs[c0] for o in offs[c0:c1 + 1]]
            n_out = min(n, c1 * chunk) - c0 * chunk
            d_comp = e.alloc(len(comp) + 16); d_comp.upload(comp)
            d_off = e.alloc(8 * len(rel)); d_off.upload(struct.pack("<%dQ" % len(rel), *rel))
            d_out = e.alloc(max(16, n_out)); d_win_in = e.alloc(32768); d_win_out = e.alloc(32768)
            e.inflate_shard_begin(d_comp.ptr, len(comp), d_off.ptr, c1 - c0, d_out.ptr, n_out, chunk_size=chunk, has_head=r > 0)
            engines.append(e); bufs.append((d_comp, d_off, d_out, d_win_in, d_win_out, n_out, c0))
        # the windows travel
        prev = None
        for r, e in enumerate(engines):
            d_comp, d_off, d_out, d_win_in, d_win_out, n_out, c0 = bufs[r]
            if r > 0:
                d_win_in.upload(prev)
            e.inflate_shard_tail(d_win_in.ptr if r > 0 else 0, d_win_out.ptr)
            prev = bytes(d_win_out.download(32768))
            end = c0 * chunk + n_out
            want = data[max(0, end - 32768):end]
            assert prev[32768 - len(want):] == want, r
            wins.append(prev)
        for r, e in enumerate(engines):
            d_comp, d_off, d_out, d_win_in, d_win_out, n_out, c0 = bufs[r]
            e.inflate_shard_finish(d_win_in.ptr if r > 0 else 0)
            assert bytes(d_out.download(n_out)) == data[c0 * chunk:c0 * chunk + n_out], r
    finally:
        for b in bufs:
            for d in b[:5]:
                d.free()
        for e in engines:
            e.close()
        e0.close()
