"""The zlib 1.2.8 stream API exported by libzamd_z.so, driven like the reference's own
test/example.c drives libz: one-shot helpers, streaming with small buffers, flushes,
gzip headers, error returns.  The system zlib (Python's) is the independent reader/writer."""
import ctypes as C
import gzip
import io
import json
import os
import zlib

import pytest

import _oracle as O
import _zapi as Z
import zlib_amd

pytestmark = pytest.mark.gpu


def test_compress_of_one_chunk_equals_reference_bytes():
    # inputs of at most one chunk are a single Z_FINISH stream: byte-identical to the reference's compress2
    for data, level in ((b"hello, hello!", 6), (O.corpus(300000, 0, 1), 6), (O.corpus(1 << 20, 4, 5), 9), (b"", 6), (O.corpus(70000, 3, 4), 4),
                        (O.corpus(200000, 0, 1), 1), (O.corpus(200000, 1, 9), 0)):
        rc, comp = Z.compress2(data, level)
        assert rc == 0
        flg = {0: 0x01, 1: 0x01, 4: 0x5e, 6: 0x9c, 9: 0xda}[level]
        want = bytes([0x78, flg]) + O.deflate_chunk(data, 0, len(data), level, 0, 8, True) + zlib.adler32(data).to_bytes(4, "big")
        assert comp == want
        assert zlib.decompress(comp) == data


def test_compress_uncompress_multi_chunk_round_trip():
    data = O.corpus(3 * (1 << 20) + 12345, 4, 5)
    rc, comp = Z.compress2(data, 6)
    assert rc == 0
    assert zlib.decompress(comp) == data
    rc, back = Z.uncompress(comp, len(data))
    assert rc == 0 and back == data
    # too little room / truncated input (uncompr.c:49-54)
    assert Z.uncompress(comp, len(data) - 1)[0] == Z.Z_BUF_ERROR
    assert Z.uncompress(comp[:-5], len(data))[0] == Z.Z_DATA_ERROR
    assert Z.compress2(data, 6, cap=1000)[0] == Z.Z_BUF_ERROR


@pytest.mark.parametrize("wbits", [15, -15, 31])
@pytest.mark.parametrize("in_step,out_step", [(None, 1 << 20), (70001, 4096), (1 << 20, 1 << 16)])
def test_deflate_streaming_is_a_valid_stream(wbits, in_step, out_step):
    data = O.corpus(2 * (1 << 20) + 777, 0, 11)
    comp, (tin, tout, adler) = Z.deflate_stream(data, 6, wbits, in_step=in_step, out_step=out_step)
    assert tin == len(data) and tout == len(comp)
    assert zlib.decompressobj(wbits).decompress(comp) == data
    if wbits == 15:
        assert adler == zlib.adler32(data)
    if wbits == 31:
        assert adler == zlib.crc32(data)
        assert comp[:10] == bytes([0x1f, 0x8b, 8, 0, 0, 0, 0, 0, 0, 3])


def test_flush_points_are_decodable_prefixes():
    data = O.corpus(400000, 4, 3)
    cuts = {1000: Z.Z_SYNC_FLUSH, 150000: Z.Z_FULL_FLUSH, 150001: Z.Z_PARTIAL_FLUSH, 300000: Z.Z_BLOCK}
    comp, _ = Z.deflate_stream(data, 6, -15, flushes=cuts)
    assert zlib.decompressobj(-15).decompress(comp) == data
    # after a Z_SYNC_FLUSH everything fed so far can be decoded from the bytes produced so far
    l = Z.lib()
    s = Z.ZStream()
    assert l.deflateInit2_(C.byref(s), 6, 8, -15, 8, 0, Z.VERSION, Z.SZ) == 0
    src = C.create_string_buffer(data, len(data))
    obuf = C.create_string_buffer(1 << 20)
    s.next_in, s.avail_in, s.next_out, s.avail_out = C.addressof(src), 5000, C.addressof(obuf), 1 << 20
    assert l.deflate(C.byref(s), Z.Z_SYNC_FLUSH) == Z.Z_OK
    part = obuf.raw[:s.total_out]
    assert part.endswith(b"\x00\x00\xff\xff")
    assert zlib.decompressobj(-15).decompress(part) == data[:5000]
    # nothing new and the same flush again: Z_BUF_ERROR (deflate.c:887-890), not fatal
    assert l.deflate(C.byref(s), Z.Z_SYNC_FLUSH) == Z.Z_BUF_ERROR
    s.avail_in = 0
    assert l.deflate(C.byref(s), Z.Z_FINISH) == Z.Z_STREAM_END
    assert zlib.decompressobj(-15).decompress(obuf.raw[:s.total_out]) == data[:5000]
    assert l.deflateEnd(C.byref(s)) == Z.Z_OK


def test_gzip_header_fields_round_trip():
    data = O.corpus(100000, 0, 4)
    h = Z.GzHeader()
    name, comment, extra = C.create_string_buffer(b"file.txt"), C.create_string_buffer(b"a comment"), C.create_string_buffer(b"\x01\x02\x03\x04\x05", 5)
    h.text, h.time, h.os, h.hcrc = 1, 0x12345678, 7, 1
    h.name, h.comment, h.extra, h.extra_len = C.addressof(name), C.addressof(comment), C.addressof(extra), 5
    comp, _ = Z.deflate_stream(data, 9, 31, header=h)
    g = gzip.GzipFile(fileobj=io.BytesIO(comp))
    assert g.read() == data
    assert g.mtime == 0x12345678
    # and back through our inflate with inflateGetHeader
    h2 = Z.GzHeader()
    nbuf, cbuf, ebuf = C.create_string_buffer(64), C.create_string_buffer(64), C.create_string_buffer(64)
    h2.name, h2.name_max, h2.comment, h2.comm_max, h2.extra, h2.extra_max = C.addressof(nbuf), 64, C.addressof(cbuf), 64, C.addressof(ebuf), 64
    rc, out, msg, unused, _ = Z.inflate_stream(comp, 47, header=h2)
    assert rc == Z.Z_STREAM_END and out == data and unused == 0
    assert (h2.done, h2.text, h2.time, h2.os, h2.hcrc, h2.extra_len, h2.xflags) == (1, 1, 0x12345678, 7, 1, 5, 2)
    assert nbuf.value == b"file.txt" and cbuf.value == b"a comment" and ebuf.raw[:5] == b"\x01\x02\x03\x04\x05"


@pytest.mark.parametrize("level", [1, 6, 9])
@pytest.mark.parametrize("in_step,out_step", [(None, 1 << 22), (5000, 3000), (64 << 10, 1 << 20)])
def test_inflate_of_foreign_streams(level, in_step, out_step):
    data = O.corpus(700000, 4, 21)
    for wbits, comp in ((15, zlib.compress(data, level)), (31, gzip.compress(data, level)),
                        (-15, zlib.compressobj(level, 8, -15).compress(data) + zlib.compressobj(level, 8, -15).flush())):
        if wbits == -15:
            c = zlib.compressobj(level, 8, -15)
            comp = c.compress(data) + c.flush()
        rc, out, msg, unused, (tin, tout, adler) = Z.inflate_stream(comp + b"TRAILING", wbits, in_step=in_step, out_step=out_step)
        assert rc == Z.Z_STREAM_END, msg
        assert out == data
        assert unused == 8 and tin == len(comp) and tout == len(data)
        if wbits == 15:
            assert adler == zlib.adler32(data)
        if wbits == 31:
            assert adler == zlib.crc32(data)


def test_inflate_of_own_chunked_stream_uses_sync_markers():
    data = O.corpus(5 * (1 << 20) + 99, 4, 6)
    comp, _ = Z.deflate_stream(data, 6, 31)
    rc, out, msg, unused, _ = Z.inflate_stream(comp, 31, out_step=1 << 20)
    assert rc == Z.Z_STREAM_END and out == data


HEADER_VECTORS = [   # test/infcover.c:399-411: (hex, windowBits, expected return, expected msg)
    ("1f 8b 00 00", 31, Z.Z_DATA_ERROR, "unknown compression method"),
    ("1f 8b 08 80", 31, Z.Z_DATA_ERROR, "unknown header flags set"),
    ("77 85", 15, Z.Z_DATA_ERROR, "unknown compression method"),
    ("78 9c", 8, Z.Z_DATA_ERROR, "invalid window size"),
    ("78 9c 63 00 00 00 01 00 01", 15, Z.Z_STREAM_END, None),
    ("1f 8b 08 1e 00 00 00 00 00 00 01 00 00 00 00 00 00", 47, Z.Z_DATA_ERROR, "header crc mismatch"),
    ("1f 8b 08 02 00 00 00 00 00 00 1d 26 03 00 00 00 00 00 00 00 00 00", 47, Z.Z_STREAM_END, None),
    ("78 90", 47, Z.Z_DATA_ERROR, "incorrect header check"),
    ("08 b8 00 00 00 01", 8, Z.Z_NEED_DICT, None),
    ("1f 8b 08 00 00 00 00 00 00 00 03 00 00 00 00 01", 47, Z.Z_DATA_ERROR, "incorrect data check"),
    ("1f 8b 08 00 00 00 00 00 00 00 03 00 00 00 00 00 00 00 00 01", 47, Z.Z_DATA_ERROR, "incorrect length check"),
]


@pytest.mark.parametrize("hexs,wbits,want,msg", HEADER_VECTORS)
def test_wrapper_vectors_of_the_reference_test(hexs, wbits, want, msg):
    comp = bytes.fromhex(hexs.replace(" ", ""))
    rc, out, got_msg, unused, _ = Z.inflate_stream(comp, wbits)
    assert rc == want
    assert got_msg == msg


def test_data_errors_carry_the_reference_messages():
    import json, os
    vec = json.load(open(os.path.join(Z.ROOT, "tests", "golden", "inflate_vectors.json")))["raw"]
    for v in vec:
        comp = bytes.fromhex(v["hex"].replace(" ", ""))
        rc, out, msg, unused, _ = Z.inflate_stream(comp, -15)
        if v["msg"] is None:
            assert rc == Z.Z_STREAM_END, v
            assert out == zlib.decompressobj(-15).decompress(comp)
        else:
            assert rc == Z.Z_DATA_ERROR and msg == v["msg"], (v, rc, msg)


def test_preset_dictionary_on_inflate():
    d = O.corpus(20000, 0, 30)
    data = O.corpus(50000, 0, 30)
    c = zlib.compressobj(6, 8, 15, 8, 0, d)
    comp = c.compress(data) + c.flush()
    rc, out, msg, unused, _ = Z.inflate_stream(comp, 15, dictionary=d)
    assert rc == Z.Z_STREAM_END and out == data
    rc, out, msg, unused, _ = Z.inflate_stream(comp, 15)
    assert rc == Z.Z_NEED_DICT


def test_checksum_entry_points():
    l = Z.lib()
    data = O.corpus(3000000, 1, 8)
    assert l.adler32(0, None, 0) == 1 and l.crc32(0, None, 0) == 0
    assert l.adler32(1, data, len(data)) == zlib.adler32(data)
    assert l.crc32(0, data, len(data)) == zlib.crc32(data)
    a = l.adler32(1, data[:1000], 1000)
    assert l.adler32(a, data[1000:], len(data) - 1000) == zlib.adler32(data)
    c1, c2 = zlib.crc32(data[:12345]), zlib.crc32(data[12345:])
    assert l.crc32_combine(c1, c2, len(data) - 12345) == zlib.crc32(data)
    a1, a2 = zlib.adler32(data[:12345]), zlib.adler32(data[12345:])
    assert l.adler32_combine64(a1, a2, len(data) - 12345) == zlib.adler32(data)


def test_params_change_mid_stream():
    data = O.corpus(300000, 0, 2)
    l = Z.lib()
    s = Z.ZStream()
    assert l.deflateInit2_(C.byref(s), 9, 8, 15, 8, 0, Z.VERSION, Z.SZ) == 0
    src = C.create_string_buffer(data, len(data))
    obuf = C.create_string_buffer(1 << 20)
    s.next_in, s.avail_in, s.next_out, s.avail_out = C.addressof(src), 100000, C.addressof(obuf), 1 << 20
    assert l.deflate(C.byref(s), Z.Z_NO_FLUSH) == Z.Z_OK
    assert l.deflateParams(C.byref(s), 4, 3) == Z.Z_OK      # to level 4, Z_RLE
    s.avail_in = 200000
    assert l.deflate(C.byref(s), Z.Z_FINISH) == Z.Z_STREAM_END
    assert zlib.decompress(obuf.raw[:s.total_out]) == data
    assert l.deflateEnd(C.byref(s)) == Z.Z_OK


@pytest.mark.parametrize("first,then,cut,total,kind,exact", [
    ((9, 0), (4, 3), 100000, 300000, 0, True), ((6, 0), (6, 1), 33333, 150000, 3, True), ((6, 0), (9, 4), 5, 90000, 2, True),
    ((6, 2), (6, 0), 65536, 400000, 4, True), ((7, 1), (7, 0), 20000, 250000, 0, True),
    ((5, 0), (8, 0), 131072, 500000, 0, False), ((2, 0), (1, 0), 60000, 180000, 0, False),
    ((4, 0), (0, 0), 50000, 120000, 4, False), ((0, 0), (6, 0), 70000, 200000, 4, False),
    ((6, 0), (1, 0), 40000, 100000, 0, False), ((1, 0), (6, 0), 70001, 200000, 4, False)])
def test_params_change_mid_stream_is_the_reference_byte_for_byte(first, then, cut, total, kind, exact):
    """deflateParams in the middle of a stream (deflate.c:490-528).  Where the change makes the reference end the block
    (a new strategy, or a level that parses with another function: it then calls deflate(Z_BLOCK), :512-518) and both
    sides parse with deflate_slow / deflate_rle / deflate_huff, the bytes are the reference's for inputs of at most one
    chunk: what was taken so far is compressed with the old values and ends on a block boundary, the rest goes on behind
    it, inside the same byte, with the new ones.
    Not reproduced, and checked by round trip and check value only:
    * a new level with the same function and strategy (5 -> 8, 2 -> 1): the reference ends no block there, and the last
      lookahead of the first part (< MIN_LOOKAHEAD bytes it has not looked at yet, :1744-1747) is parsed with the new values;
    * from or to deflate_fast / deflate_stored: the second part goes on with the hash chains the first part's function
      left behind — every string (deflate_slow), some (deflate_fast, :1680-1697), none (deflate_stored) — and with its
      prev_length, which longest_match takes as every later search's starting length (:1156) when deflate_fast, which
      never sets it, takes over; deflate_stored cuts its blocks by where the stream stands in the window (:1571-1609).
      A chunk primed with its history has every string entered and starts at window index w_size."""
    ref = Z.ref_lib()
    if ref is None:
        pytest.skip("reference build not present")
    P = C.POINTER(Z.ZStream)
    ref.deflateInit2_.argtypes = [P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_int]
    ref.deflate.argtypes = [P, C.c_int]; ref.deflateParams.argtypes = [P, C.c_int, C.c_int]; ref.deflateEnd.argtypes = [P]
    data = O.corpus(total, kind, 210 + cut % 7)
    src = C.create_string_buffer(data, len(data))

    def run(l):
        s = Z.ZStream()
        obuf = C.create_string_buffer(2 * total + 4096)
        assert l.deflateInit2_(C.byref(s), first[0], 8, 15, 8, first[1], Z.VERSION, Z.SZ) == 0
        s.next_in, s.avail_in, s.next_out, s.avail_out = C.addressof(src), cut, C.addressof(obuf), 2 * total + 4096
        assert l.deflate(C.byref(s), Z.Z_NO_FLUSH) == Z.Z_OK
        assert l.deflateParams(C.byref(s), then[0], then[1]) == Z.Z_OK
        assert s.avail_in == 0
        s.avail_in = total - cut
        assert l.deflate(C.byref(s), Z.Z_FINISH) == Z.Z_STREAM_END
        out = obuf.raw[:s.total_out]
        adler = s.adler
        assert l.deflateEnd(C.byref(s)) == Z.Z_OK
        return out, adler

    want, wadler = run(ref)
    got, gadler = run(Z.lib())
    assert zlib.decompress(got) == data
    assert gadler == wadler
    if exact:
        assert got == want


def test_tuned_stream_equals_the_tuned_reference():
    """deflateTune (deflate.c:531-547): values hold until deflateReset or a level change."""
    data = O.corpus(400000, 4, 77)
    l = Z.lib()
    s = Z.ZStream()
    src = C.create_string_buffer(data, len(data))
    cap = len(data) + 4096
    obuf = C.create_string_buffer(cap)

    def one_shot():
        s.next_in, s.avail_in, s.next_out, s.avail_out = C.addressof(src), len(data), C.addressof(obuf), cap
        s.total_out = 0
        assert l.deflate(C.byref(s), Z.Z_FINISH) == Z.Z_STREAM_END
        return obuf.raw[:s.total_out]

    assert l.deflateInit2_(C.byref(s), 6, 8, -15, 8, 0, Z.VERSION, Z.SZ) == 0
    tune = (4, 5, 24, 12)
    assert l.deflateTune(C.byref(s), *tune) == Z.Z_OK
    tuned = one_shot()
    assert tuned == O.deflate_chunk(data, 0, len(data), 6, 0, 8, True, tune=tune)
    plain = O.deflate_chunk(data, 0, len(data), 6, 0, 8, True)
    assert tuned != plain
    assert l.deflateReset(C.byref(s)) == Z.Z_OK                 # back to the level's table
    assert one_shot() == plain
    assert l.deflateReset(C.byref(s)) == Z.Z_OK
    assert l.deflateTune(C.byref(s), *tune) == Z.Z_OK
    assert l.deflateParams(C.byref(s), 7, 0) == Z.Z_OK          # a new level brings its own values
    assert one_shot() == O.deflate_chunk(data, 0, len(data), 7, 0, 8, True)
    assert l.deflateReset(C.byref(s)) == Z.Z_OK
    assert l.deflateTune(C.byref(s), 4, 4, 8, 0) == Z.Z_STREAM_ERROR       # no chain budget
    assert l.deflateParams(C.byref(s), 1, 0) == Z.Z_OK
    assert l.deflateTune(C.byref(s), 4, 4, 8, 65) == Z.Z_STREAM_ERROR      # levels 1-3: at most 64
    assert l.deflateTune(C.byref(s), 4, 6, 8, 64) == Z.Z_OK
    assert one_shot() == O.deflate_chunk(data, 0, len(data), 1, 0, 8, True, tune=(4, 6, 8, 64))
    assert l.deflateEnd(C.byref(s)) == Z.Z_OK


def _deflate_with_dict(data, d, level=6, wbits=15):
    l = Z.lib()
    s = Z.ZStream()
    assert l.deflateInit2_(C.byref(s), level, 8, wbits, 8, 0, Z.VERSION, Z.SZ) == 0
    assert l.deflateSetDictionary(C.byref(s), d, len(d)) == Z.Z_OK
    src = C.create_string_buffer(data, max(len(data), 1))
    cap = len(data) + len(data) // 8 + 4096
    obuf = C.create_string_buffer(cap)
    s.next_in, s.avail_in, s.next_out, s.avail_out = C.addressof(src), len(data), C.addressof(obuf), cap
    assert l.deflate(C.byref(s), Z.Z_FINISH) == Z.Z_STREAM_END
    out = obuf.raw[:s.total_out]
    adler = s.adler
    assert l.deflateEnd(C.byref(s)) == Z.Z_OK
    return out, adler


def test_preset_dictionary_on_deflate():
    data = O.corpus(200000, 0, 31)
    # dictionaries of any length, also longer than the window (the last 32 KiB count, the id covers all of it,
    # deflate.c:323-389): the reference's bytes exactly
    for dl in (1, 15, 1001, 16000, 32768, 50000):
        d = O.corpus(dl, 0, 32)
        out, adler = _deflate_with_dict(data, d)
        k = d[-32768:]
        want = bytes([0x78, 0xbb]) + zlib.adler32(d).to_bytes(4, "big") + O.deflate_chunk(k + data, len(k), len(data), 6, 0, 8, True) \
            + zlib.adler32(data).to_bytes(4, "big")
        assert out == want, dl
        assert adler == zlib.adler32(data)
        assert zlib.decompressobj(15, d).decompress(out) == data
        raw, _ = _deflate_with_dict(data, d, wbits=-15)
        assert raw == O.deflate_chunk(k + data, len(k), len(data), 6, 0, 8, True), dl
    # our own inflate takes it back through Z_NEED_DICT
    rc, back, msg, unused, _ = Z.inflate_stream(out, 15, dictionary=d)
    assert rc == Z.Z_STREAM_END and back == data


@pytest.mark.parametrize("wbits", [8, 9, 10, 11, 12, 13, 14])
def test_deflate_with_a_small_window(wbits):
    """windowBits 8..14 (deflate.c:271-291; 8 becomes 9, :276): header, body and trailer as the reference writes them."""
    data = O.corpus(150000, 4, 40 + wbits)
    w = max(wbits, 9)
    out, totals = Z.deflate_stream(data, 6, wbits)
    hdr = (8 + ((w - 8) << 4)) << 8 | (2 << 6)
    hdr += 31 - hdr % 31
    want = hdr.to_bytes(2, "big") + O.deflate_chunk(data, 0, len(data), 6, 0, 8, True, wbits=w) + zlib.adler32(data).to_bytes(4, "big")
    assert out == want
    assert zlib.decompress(out) == data
    raw, _ = Z.deflate_stream(data, 9, -wbits)
    assert raw == O.deflate_chunk(data, 0, len(data), 9, 0, 8, True, wbits=w)
    gz, _ = Z.deflate_stream(data, 1, 16 + wbits)
    assert gz[10:-8] == O.deflate_chunk(data, 0, len(data), 1, 0, 8, True, wbits=w)
    # several chunks and a dictionary with the same window
    big = O.corpus(3 * (1 << 20) // 8, 0, 41)
    comp, _ = Z.deflate_stream(big, 6, -wbits, in_step=100000)
    assert zlib.decompressobj(-15).decompress(comp) == big


@pytest.mark.parametrize("flush", [Z.Z_PARTIAL_FLUSH, Z.Z_BLOCK])
def test_partial_flush_and_block_are_the_references_bits(flush):
    """Z_PARTIAL_FLUSH ends the data with _tr_align's 10 bits (trees.c:892-901), Z_BLOCK with nothing (deflate.c:923-925);
    both leave up to seven bits in the bit buffer (deflatePending), which lead the next block."""
    data = O.corpus(120000, 0, 50)
    l = Z.lib()
    s = Z.ZStream()
    assert l.deflateInit2_(C.byref(s), 6, 8, -15, 8, 0, Z.VERSION, Z.SZ) == 0
    src = C.create_string_buffer(data, len(data))
    obuf = C.create_string_buffer(1 << 20)
    s.next_in, s.avail_in, s.next_out, s.avail_out = C.addressof(src), 70000, C.addressof(obuf), 1 << 20
    assert l.deflate(C.byref(s), flush) == Z.Z_OK
    first = obuf.raw[:s.total_out]
    want, pb, pv, dt = O.deflate_chunk(data[:70000], 0, 70000, 6, 0, 8, flush=O.FLUSH_PARTIAL if flush == Z.Z_PARTIAL_FLUSH else O.FLUSH_BLOCK,
                                       want_state=True)
    assert first == want
    pend, bits = C.c_uint(99), C.c_int(99)
    assert l.deflatePending(C.byref(s), C.byref(pend), C.byref(bits)) == Z.Z_OK
    assert (pend.value, bits.value) == (0, pb)
    assert s.data_type == dt == 1
    # the rest of the input: a chunk of its own, primed with the history and led by the bits left over
    s.avail_in = 50000
    assert l.deflate(C.byref(s), Z.Z_FINISH) == Z.Z_STREAM_END
    whole = obuf.raw[:s.total_out]
    rest = O.deflate_chunk(data[70000 - 32768:], 32768, 50000, 6, 0, 8, True, prime=(pb, pv))
    assert whole == first + rest
    assert zlib.decompressobj(-15).decompress(whole) == data
    assert l.deflateEnd(C.byref(s)) == Z.Z_OK


def test_data_type_is_set_from_the_first_block():
    """strm->data_type after deflate (trees.c:920-921, 1120-1147): Z_TEXT / Z_BINARY from the first block's literals at
    level > 0, untouched (Z_UNKNOWN) at level 0."""
    l = Z.lib()
    for kind, level, want in ((0, 6, 1), (1, 6, 0), (4, 1, O.deflate_chunk(O.corpus(90000, 4, 60), 0, 90000, 1, want_state=True)[3]), (0, 0, 2), (3, 9, 0)):
        data = O.corpus(90000, kind, 60)
        s = Z.ZStream()
        assert l.deflateInit2_(C.byref(s), level, 8, 15, 8, 0, Z.VERSION, Z.SZ) == 0
        assert s.data_type == 2
        src = C.create_string_buffer(data, len(data))
        obuf = C.create_string_buffer(1 << 20)
        s.next_in, s.avail_in, s.next_out, s.avail_out = C.addressof(src), len(data), C.addressof(obuf), 1 << 20
        assert l.deflate(C.byref(s), Z.Z_FINISH) == Z.Z_STREAM_END
        assert s.data_type == want, (kind, level)
        assert l.deflateEnd(C.byref(s)) == Z.Z_OK


def test_preloaded_under_a_program_linked_against_libz():
    """Python's own zlib module is linked against the system libz; with libzamd_z.so preloaded the same
    module runs on the GPU library (tests/gpu_preload_check.py prints one line per check)."""
    import os
    import subprocess
    import sys
    env = dict(os.environ)
    ours = os.path.join(Z.ROOT, "zlib_amd", "libzamd_z.so")
    env["LD_PRELOAD"] = (env.get("LD_PRELOAD", "") + " " + ours).strip()
    r = subprocess.run([sys.executable, os.path.join(Z.ROOT, "tests", "gpu_preload_check.py")], cwd=Z.ROOT, env=env,
                       capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if ":" in l]
    assert "runtime version: 1.2.8" in r.stdout
    assert "error text: Error -3 while decompressing data: invalid block type" in r.stdout
    checks = [l for l in lines if l.split(":")[0] not in ("runtime version", "error text")]
    assert len(checks) == 5 and all("False" not in l for l in checks), r.stdout


def test_false_sync_markers_inside_the_data():
    # 00 00 FF FF inside stored blocks and inside Huffman-coded data looks like a flush point: the guesses
    # that start there must not change the result
    import random
    rnd = random.Random(5)
    raw = bytearray(rnd.getrandbits(8) for _ in range(400000))
    for i in range(0, len(raw) - 8, 997):
        raw[i:i + 4] = b"\x00\x00\xff\xff"
    raw = bytes(raw)
    for level in (0, 1, 6):
        c = zlib.compressobj(level, zlib.DEFLATED, 15)
        comp = c.compress(raw[:150000]) + c.flush(zlib.Z_SYNC_FLUSH) + c.compress(raw[150000:]) + c.flush()
        assert comp.count(b"\x00\x00\xff\xff") > (50 if level == 0 else 1)
        for in_step in (None, 70000):
            rc, out, msg, unused, _ = Z.inflate_stream(comp, 15, in_step=in_step, out_step=1 << 20)
            assert rc == Z.Z_STREAM_END, msg
            assert out == raw


def test_inflate_prime_starts_inside_a_byte():
    """inflatePrime (inflate.c:230-250) as examples/zran.c:296-300 uses it: the stream starts at a bit
    inside a byte whose upper bits are handed over first."""
    data = O.corpus(300000, 4, 91)
    l = Z.lib()
    for k in (1, 3, 7):
        shifted = O.deflate_chunk(data, 0, len(data), 6, 0, 8, True, prime=(k, 0x7f))    # k foreign bits in front
        s = Z.ZStream()
        assert l.inflateInit2_(C.byref(s), -15, Z.VERSION, Z.SZ) == 0
        if k == 3:      # in two portions
            assert l.inflatePrime(C.byref(s), 2, (shifted[0] >> k) & 3) == Z.Z_OK
            assert l.inflatePrime(C.byref(s), 8 - k - 2, shifted[0] >> (k + 2)) == Z.Z_OK
        else:
            assert l.inflatePrime(C.byref(s), 8 - k, shifted[0] >> k) == Z.Z_OK
        src = C.create_string_buffer(shifted[1:], len(shifted) - 1)
        obuf = C.create_string_buffer(len(data) + 16)
        s.next_in, s.avail_in, s.next_out, s.avail_out = C.addressof(src), len(shifted) - 1, C.addressof(obuf), len(data) + 16
        assert l.inflate(C.byref(s), Z.Z_FINISH) == Z.Z_STREAM_END
        assert obuf.raw[:s.total_out] == data
        assert l.inflatePrime(C.byref(s), 17, 0) == Z.Z_STREAM_ERROR
        assert l.inflateEnd(C.byref(s)) == Z.Z_OK
    # a wrapped stream takes no bits in front of its header
    s = Z.ZStream()
    assert l.inflateInit2_(C.byref(s), 15, Z.VERSION, Z.SZ) == 0
    assert l.inflatePrime(C.byref(s), 3, 1) == Z.Z_STREAM_ERROR
    assert l.inflatePrime(C.byref(s), -1, 0) == Z.Z_OK
    assert l.inflateEnd(C.byref(s)) == Z.Z_OK


def test_inflate_prime_in_the_middle_of_a_stream(engine):
    """inflatePrime at any time (inflate.c:230-250): bytes up to the one a block ends in, that block's last bits through
    inflatePrime, the rest of the byte the same way, then the remaining bytes.  The reference build takes the same calls."""
    import zlib_amd
    data = O.corpus(600000, 0, 17)
    c = zlib.compressobj(6, zlib.DEFLATED, -15)
    comp = c.compress(data) + c.flush()
    # a block boundary inside a byte, from the engine's own segment interface
    P = L1 = None
    for frac in (3, 4, 5, 6, 7):
        seg = dict(in_bit=0, in_end_bit=8 * len(comp), out_off=0, out_cap=1 << 22, win_avail=0, mode=zlib_amd.INF_BLOCKSTOP, stop_bit=len(comp) * frac)
        res, _ = engine.inflate_segments(comp, [seg])
        if res[0]["status"] == 2 and res[0]["end_bit"] % 8:
            P, L1 = res[0]["end_bit"], res[0]["out_len"]
            break
    assert P is not None and 0 < L1 < len(data)
    K, r = P // 8, P % 8
    results = []
    for l in (Z.lib(), Z.ref_lib()):
        if l is None:
            continue
        l.inflatePrime.argtypes = [C.POINTER(Z.ZStream), C.c_int, C.c_int]
        s = Z.ZStream()
        assert l.inflateInit2_(C.byref(s), -15, Z.VERSION, Z.SZ) == 0
        src = C.create_string_buffer(comp, len(comp))
        obuf = C.create_string_buffer(len(data) + 16)
        s.next_out, s.avail_out = C.addressof(obuf), len(data) + 16
        s.next_in, s.avail_in = C.addressof(src), K
        assert l.inflate(C.byref(s), Z.Z_SYNC_FLUSH) in (Z.Z_OK, Z.Z_BUF_ERROR)
        assert s.avail_in == 0 and s.total_out <= L1
        assert l.inflatePrime(C.byref(s), r, comp[K] & ((1 << r) - 1)) == Z.Z_OK            # the block's last bits
        assert l.inflate(C.byref(s), Z.Z_SYNC_FLUSH) in (Z.Z_OK, Z.Z_BUF_ERROR)
        assert s.total_out == L1 and obuf.raw[:L1] == data[:L1]                               # exactly up to the block's end
        assert l.inflatePrime(C.byref(s), 8 - r, comp[K] >> r) == Z.Z_OK                      # the rest of that byte
        s.next_in, s.avail_in = C.addressof(src) + K + 1, len(comp) - K - 1
        assert l.inflate(C.byref(s), Z.Z_FINISH) == Z.Z_STREAM_END
        assert s.total_out == len(data) and obuf.raw[:len(data)] == data
        results.append((s.total_in, s.total_out))
        assert l.inflateEnd(C.byref(s)) == Z.Z_OK
    assert len(set(results)) == 1
    # bits that leave a byte unfinished, then input: one buffer cannot hold that, and says so
    l = Z.lib()
    s = Z.ZStream()
    assert l.inflateInit2_(C.byref(s), -15, Z.VERSION, Z.SZ) == 0
    src = C.create_string_buffer(comp, len(comp))
    obuf = C.create_string_buffer(len(data) + 16)
    s.next_out, s.avail_out, s.next_in, s.avail_in = C.addressof(obuf), len(data) + 16, C.addressof(src), K
    assert l.inflate(C.byref(s), Z.Z_SYNC_FLUSH) in (Z.Z_OK, Z.Z_BUF_ERROR)
    assert l.inflatePrime(C.byref(s), 3, 5) == Z.Z_OK
    s.next_in, s.avail_in = C.addressof(src) + K, 10
    assert l.inflate(C.byref(s), Z.Z_SYNC_FLUSH) == Z.Z_STREAM_ERROR and b"inflatePrime" in s.msg
    assert l.inflateEnd(C.byref(s)) == Z.Z_OK


def test_deflate_prime_puts_bits_in_front():
    """deflatePrime (deflate.c:464-487): whole bytes go out at once, the rest shifts the first block."""
    data = O.corpus(200000, 4, 93)
    l = Z.lib()
    for bits, value in ((3, 5), (11, 0x5a5), (8, 0xc3), (16, 0xbeef)):
        s = Z.ZStream()
        assert l.deflateInit2_(C.byref(s), 6, 8, -15, 8, 0, Z.VERSION, Z.SZ) == 0
        assert l.deflatePrime(C.byref(s), bits, value) == Z.Z_OK
        pend, pbits = C.c_uint(0), C.c_int(0)
        assert l.deflatePending(C.byref(s), C.byref(pend), C.byref(pbits)) == Z.Z_OK
        assert (pend.value, pbits.value) == (bits // 8, bits % 8)
        src = C.create_string_buffer(data, len(data))
        cap = len(data) + 4096
        obuf = C.create_string_buffer(cap)
        s.next_in, s.avail_in, s.next_out, s.avail_out = C.addressof(src), len(data), C.addressof(obuf), cap
        assert l.deflate(C.byref(s), Z.Z_FINISH) == Z.Z_STREAM_END
        out = obuf.raw[:s.total_out]
        assert l.deflateEnd(C.byref(s)) == Z.Z_OK
        whole, rest = bits // 8, bits % 8
        want = (value & ((1 << (8 * whole)) - 1)).to_bytes(whole, "little") + \
            O.deflate_chunk(data, 0, len(data), 6, 0, 8, True, prime=(rest, value >> (8 * whole)))
        assert out == want
        # and the reader's counterpart takes it back
        s = Z.ZStream()
        assert l.inflateInit2_(C.byref(s), -15, Z.VERSION, Z.SZ) == 0
        body = out[whole:]
        if rest:
            assert l.inflatePrime(C.byref(s), 8 - rest, body[0] >> rest) == Z.Z_OK
            body = body[1:]
        src = C.create_string_buffer(body, len(body))
        back = C.create_string_buffer(len(data) + 16)
        s.next_in, s.avail_in, s.next_out, s.avail_out = C.addressof(src), len(body), C.addressof(back), len(data) + 16
        assert l.inflate(C.byref(s), Z.Z_FINISH) == Z.Z_STREAM_END
        assert back.raw[:s.total_out] == data
        assert l.inflateEnd(C.byref(s)) == Z.Z_OK


def _inflate_back(stream, piece, fail_output=False):
    """inflateBack over `stream`, handed out by the in() call-back `piece` bytes at a time."""
    l = Z.lib()
    s = Z.ZStream()
    win = C.create_string_buffer(32768)
    assert l.inflateBackInit_(C.byref(s), 15, win, Z.VERSION, Z.SZ) == Z.Z_OK
    src = C.create_string_buffer(stream, max(len(stream), 1))
    state = {"pos": 0}
    got = bytearray()

    def pull(desc, bufp):
        n = min(piece, len(stream) - state["pos"])
        if n:
            bufp[0] = C.addressof(src) + state["pos"]
            state["pos"] += n
        return n

    def push(desc, buf, n):
        got.extend(C.string_at(buf, n))
        return 1 if fail_output else 0

    rc = l.inflateBack(C.byref(s), Z.IN_FUNC(pull), None, Z.OUT_FUNC(push), None)
    msg, left = s.msg, s.avail_in
    assert l.inflateBackEnd(C.byref(s)) == Z.Z_OK
    return rc, bytes(got), msg, left


def test_inflate_back_call_back_interface():
    """infback.c through the cases of the reference's test (test/infcover.c:470-504, 507-560)."""
    l = Z.lib()
    win = C.create_string_buffer(32768)
    assert l.inflateBackInit_(None, 0, win, None, 0) == Z.Z_VERSION_ERROR
    assert l.inflateBackInit_(None, 0, win, Z.VERSION, Z.SZ) == Z.Z_STREAM_ERROR
    assert l.inflateBackEnd(None) == Z.Z_STREAM_ERROR
    s = Z.ZStream()
    assert l.inflateBackInit_(C.byref(s), 7, win, Z.VERSION, Z.SZ) == Z.Z_STREAM_ERROR
    assert _inflate_back(bytes([3, 0]), 2)[0] == Z.Z_STREAM_END
    data = O.corpus(500000, 4, 111)
    raw = O.deflate_chunked(data, 131072, 6, 0, 8)[0]
    for piece in (len(raw) + 10, 65536, 1000):
        rc, got, msg, left = _inflate_back(raw + b"xyz", piece)
        assert rc == Z.Z_STREAM_END and got == data
    assert _inflate_back(raw, 1 << 20)[3] == 0 and _inflate_back(raw + b"xyz", 1 << 20)[3] == 3      # unused input is handed back
    assert _inflate_back(raw[:len(raw) // 2], 65536)[0] == Z.Z_BUF_ERROR                               # in() runs dry
    assert _inflate_back(raw, 65536, fail_output=True)[0] == Z.Z_BUF_ERROR                             # out() fails
    vec = json.load(open(os.path.join(Z.ROOT, "tests", "golden", "inflate_vectors.json")))["raw"]
    for v in vec:
        rc, got, msg, left = _inflate_back(bytes.fromhex(v["hex"].replace(" ", "")), 1 << 16)
        assert rc != Z.Z_STREAM_ERROR
        if v["msg"] is not None:
            assert rc == Z.Z_DATA_ERROR and msg == v["msg"].encode(), (v, rc, msg)


def test_state_goes_through_the_callers_allocator():
    """zalloc/zfree hooks (deflate.c:240-253, 277, 997-1003; the bookkeeping idea of test/infcover.c:55-234):
    every block comes from the caller's allocator and goes back to it; a failing allocator is Z_MEM_ERROR."""
    l = Z.lib()
    libc = C.CDLL(None)
    libc.malloc.restype = C.c_void_p; libc.malloc.argtypes = [C.c_size_t]
    libc.free.argtypes = [C.c_void_p]
    live, stats = set(), {"allocs": 0, "fail": False}

    def zalloc(opaque, items, size):
        if stats["fail"]:
            return None
        p = libc.malloc(items * size)
        live.add(p); stats["allocs"] += 1
        return p

    def zfree(opaque, p):
        live.discard(p)
        libc.free(p)

    za, zf = Z.ALLOC(zalloc), Z.FREE(zfree)
    data = O.corpus(200000, 0, 121)
    src = C.create_string_buffer(data, len(data))
    obuf = C.create_string_buffer(len(data) + 4096)
    s = Z.ZStream()
    s.zalloc, s.zfree = C.cast(za, C.c_void_p), C.cast(zf, C.c_void_p)
    assert l.deflateInit2_(C.byref(s), 6, 8, 15, 8, 0, Z.VERSION, Z.SZ) == Z.Z_OK
    assert stats["allocs"] >= 1 and live
    s.next_in, s.avail_in, s.next_out, s.avail_out = C.addressof(src), len(data), C.addressof(obuf), len(data) + 4096
    d = Z.ZStream()
    assert l.deflateCopy(C.byref(d), C.byref(s)) == Z.Z_OK               # the copy's state comes from the same hooks
    assert l.deflate(C.byref(s), Z.Z_FINISH) == Z.Z_STREAM_END
    packed = obuf.raw[:s.total_out]
    assert l.deflateEnd(C.byref(s)) == Z.Z_OK and l.deflateEnd(C.byref(d)) == Z.Z_OK
    assert not live
    i = Z.ZStream()
    i.zalloc, i.zfree = C.cast(za, C.c_void_p), C.cast(zf, C.c_void_p)
    assert l.inflateInit2_(C.byref(i), 15, Z.VERSION, Z.SZ) == Z.Z_OK and live
    psrc = C.create_string_buffer(packed, len(packed))
    back = C.create_string_buffer(len(data) + 16)
    i.next_in, i.avail_in, i.next_out, i.avail_out = C.addressof(psrc), len(packed), C.addressof(back), len(data) + 16
    assert l.inflate(C.byref(i), Z.Z_FINISH) == Z.Z_STREAM_END and back.raw[:i.total_out] == data
    assert l.inflateEnd(C.byref(i)) == Z.Z_OK and not live
    stats["fail"] = True
    for init in (lambda z: l.deflateInit2_(C.byref(z), 6, 8, 15, 8, 0, Z.VERSION, Z.SZ), lambda z: l.inflateInit2_(C.byref(z), 15, Z.VERSION, Z.SZ)):
        z = Z.ZStream()
        z.zalloc, z.zfree = C.cast(za, C.c_void_p), C.cast(zf, C.c_void_p)
        assert init(z) == Z.Z_MEM_ERROR and not z.state


def test_a_callers_memory_limit_bounds_the_buffers_too():
    """The mem_limit idea of test/infcover.c:55-234: an allocator that refuses beyond a budget.  The state's buffers
    (input waiting for a full batch, output waiting for room) come from zalloc as well, so a stream that would
    have to buffer more than the budget reports Z_MEM_ERROR instead of growing behind the caller's back, and
    everything it did get goes back through zfree."""
    l = Z.lib()
    libc = C.CDLL(None)
    libc.malloc.restype = C.c_void_p; libc.malloc.argtypes = [C.c_size_t]
    libc.free.argtypes = [C.c_void_p]
    live, budget = {}, {"left": 3 << 20, "refused": 0}

    def zalloc(opaque, items, size):
        n = items * size
        if n > budget["left"]:
            budget["refused"] += 1
            return None
        p = libc.malloc(n)
        live[p] = n; budget["left"] -= n
        return p

    def zfree(opaque, p):
        budget["left"] += live.pop(p)
        libc.free(p)

    za, zf = Z.ALLOC(zalloc), Z.FREE(zfree)
    data = O.corpus(8 << 20, 1, 141)                     # noise: the output is as large as the input
    src = C.create_string_buffer(data, len(data))
    obuf = C.create_string_buffer(1 << 16)
    s = Z.ZStream()
    s.zalloc, s.zfree = C.cast(za, C.c_void_p), C.cast(zf, C.c_void_p)
    assert l.deflateInit2_(C.byref(s), 6, 8, -15, 8, 0, Z.VERSION, Z.SZ) == Z.Z_OK
    rc, pos = Z.Z_OK, 0
    while rc == Z.Z_OK and pos < len(data):               # 8 MiB of input in pieces: they wait in the state for a full batch
        s.next_in, s.avail_in = C.addressof(src) + pos, 1 << 20
        s.next_out, s.avail_out = C.addressof(obuf), 1 << 16
        rc = l.deflate(C.byref(s), Z.Z_NO_FLUSH)
        pos += (1 << 20) - s.avail_in
    assert rc == Z.Z_MEM_ERROR and budget["refused"] >= 1 and pos < len(data)
    l.deflateEnd(C.byref(s))
    assert not live and budget["left"] == 3 << 20


def test_streams_of_different_threads():
    """Distinct streams may be used from different threads at once (zlib.h:143-145): each call takes one of the
    library's engines (lanes) for as long as it runs."""
    import threading
    l = Z.lib()
    results = {}

    def work(k):
        try:
            data = O.corpus(700000 + 12345 * k, k % 5, 130 + k)
            for rep in range(3):
                cap = C.c_ulong(l.compressBound(len(data)))
                out = C.create_string_buffer(cap.value)
                assert l.compress2(out, C.byref(cap), data, len(data), 1 + (k + rep) % 9) == Z.Z_OK
                assert zlib.decompress(out.raw[:cap.value]) == data
                n = C.c_ulong(len(data))
                back = C.create_string_buffer(len(data))
                assert l.uncompress(back, C.byref(n), out.raw[:cap.value], cap.value) == Z.Z_OK and back.raw == data
                rc, got, msg, unused, _ = Z.inflate_stream(zlib.compress(data, 6), 15, in_step=50000, out_step=200000)
                assert rc == Z.Z_STREAM_END and got == data
                assert l.crc32(0, data, len(data)) == zlib.crc32(data)
            results[k] = True
        except BaseException as e:          # noqa: BLE001 - reported by the main thread
            results[k] = e

    threads = [threading.Thread(target=work, args=(k,)) for k in range(6)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert all(results.get(k) is True for k in range(6)), results


def test_threads_run_side_by_side_on_the_gpu():
    """Six threads with streams of their own take less time together than one after the other: the calls hold a lane
    each, not one lock for the whole library.  (Measured: 0.5 to 0.6 of the serial time with the default four lanes; the
    bound here leaves room for a busy host.)"""
    import subprocess, sys
    r = subprocess.run([sys.executable, os.path.join(Z.ROOT, "tests", "gpu_threads_bench.py"), "4", "6"], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, PYTHONPATH=Z.ROOT))
    assert r.returncode == 0 and "round trips OK" in r.stdout, r.stdout + r.stderr
    ratio = float(r.stdout.split("ratio ")[1].split(";")[0])
    assert ratio < 0.9, r.stdout


@pytest.mark.parametrize("in_step", [None, 3000, 700])
def test_bytes_behind_a_stream_go_back_to_the_call_that_brought_them(in_step, monkeypatch):
    """Two gzip members of highly compressible data with a small output batch: decoding stops again and again with
    output waiting while the end of the first member and the whole second member are already buffered.  After
    Z_STREAM_END next_in must stand exactly behind the first member's trailer (inflate.c:1198-1208), whatever the
    feeding pattern, so that the second member is not lost."""
    import gzip
    monkeypatch.setenv("ZAMD_Z_OUT_BATCH_MIB", "8")     # a block of zeros from level 6 inflates to 4 MiB: two blocks a round
    a = bytes(40 << 20)
    b = O.corpus(300000, 0, 71)
    m1, m2 = gzip.compress(a, 6), gzip.compress(b, 6)
    comp = m1 + m2
    l = Z.lib()
    s = Z.ZStream()
    assert l.inflateInit2_(C.byref(s), 31, Z.VERSION, Z.SZ) == 0
    src = C.create_string_buffer(comp, len(comp))
    obuf = C.create_string_buffer(1 << 20)
    outs, cur = [], bytearray()
    pos, n = 0, len(comp)
    step = in_step or n
    guard = 0
    while len(outs) < 2:
        guard += 1
        assert guard < 100000
        avail = min(step, n - pos)
        s.next_in, s.avail_in = C.addressof(src) + pos, avail
        s.next_out, s.avail_out = C.addressof(obuf), 1 << 20
        rc = l.inflate(C.byref(s), Z.Z_NO_FLUSH)
        assert rc in (Z.Z_OK, Z.Z_STREAM_END, Z.Z_BUF_ERROR), (rc, s.msg)
        used = avail - s.avail_in
        assert 0 <= used <= avail and s.next_in == C.addressof(src) + pos + used
        pos += used
        cur += obuf.raw[:(1 << 20) - s.avail_out]
        if rc == Z.Z_STREAM_END:
            outs.append(bytes(cur)); cur = bytearray()
            if len(outs) == 1:
                assert pos == len(m1), (pos, len(m1))
                assert s.total_in == len(m1)
                assert l.inflateReset(C.byref(s)) == 0
    assert outs[0] == a and outs[1] == b and pos == n
    assert l.inflateEnd(C.byref(s)) == 0


INF_VEC = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "inflate_vectors.json")))["inf"]


@pytest.mark.parametrize("v", INF_VEC, ids=lambda v: v["what"].replace(" ", "_"))
def test_inf_harness_vectors_of_the_reference_test(v):
    """The vectors of the reference's inf() harness (test/infcover.c:367-371, 611-613, 643-658) through the harness's own
    loop: the first inflate() call gives the stated code; a stream that ends gives the reference's bytes, a broken one
    the reference's message (both taken from the reference build where it is present)."""
    rcs, out, msg = Z.inf_loop(Z.lib(), v["hex"], v["step"], v["win"], v["len"])
    assert rcs[0] == v["err"], (rcs, msg)
    ref = Z.ref_lib()
    if ref is not None:
        rrcs, rout, rmsg = Z.inf_loop(ref, v["hex"], v["step"], v["win"], v["len"])
        assert rrcs[0] == v["err"]
        if rrcs[-1] == Z.Z_STREAM_END:
            assert rcs[-1] == Z.Z_STREAM_END and out == rout
        if rrcs[-1] == Z.Z_DATA_ERROR:
            assert rcs[-1] == Z.Z_DATA_ERROR and msg == rmsg


def _bits_lsb(fields):
    """(value, nbits, is_huffman_code) -> bytes, LSB first; Huffman codes go in most significant bit first."""
    acc, n = 0, 0
    for val, nb, code in fields:
        if code:
            val = int(("{:0%db}" % nb).format(val)[::-1], 2)
        acc |= val << n
        n += nb
    return acc.to_bytes((n + 7) // 8, "little")


@pytest.mark.parametrize("win,ok", [(-8, False), (-9, True), (-15, True)])
def test_a_distance_beyond_the_declared_window_is_too_far_back(win, ok):
    """A stored block of 300 bytes in one inflate() call, then, in the next call, a match at distance 300: with a
    256-byte window (windowBits 8) the bytes are gone (inflate.c:379-427, 1122-1134), with 512 bytes or more they are
    there.  The expectation comes from the reference build where it is present."""
    body = O.corpus(300, 1, 99)
    b1 = bytes([0]) + (300).to_bytes(2, "little") + (300 ^ 0xffff).to_bytes(2, "little") + body
    # final fixed block: length 3 (code 257, 7 bits), distance 300 (code 16, 5 bits, 7 extra bits = 300 - 257), end of block
    b2 = _bits_lsb([(1, 1, False), (1, 2, False), (1, 7, True), (16, 5, True), (43, 7, False), (0, 7, True)])
    stream = b1 + b2
    rcs, out, msg = Z.inf_loop(Z.lib(), stream, 0, win, 400, pieces=[len(b1), len(b2)])
    ref = Z.ref_lib()
    if ref is not None:
        rrcs, rout, rmsg = Z.inf_loop(ref, stream, 0, win, 400, pieces=[len(b1), len(b2)])
        assert (rrcs[-1] == Z.Z_STREAM_END) == ok and (ok or rmsg == "invalid distance too far back")
    if ok:
        assert rcs[-1] == Z.Z_STREAM_END and out == body + body[:3]
    else:
        assert rcs[-1] == Z.Z_DATA_ERROR and msg == "invalid distance too far back" and out == body


def test_get_crc_table_hands_out_all_eight_tables():
    """crc32.c:190-199 returns crc_table[8][256] (crc32.c:90-125); callers that slice by four read tables 1-3."""
    l = Z.lib()
    t = l.get_crc_table()
    tab = [[t[k * 256 + n] for n in range(256)] for k in range(8)]
    for n in (0, 1, 2, 127, 128, 255):
        for k in range(4):
            want = zlib.crc32(bytes([n]) + bytes(k), 0xffffffff) ^ 0xffffffff      # raw register: byte n then k zero bytes
            assert tab[k][n] == want, (k, n)
            assert tab[k + 4][n] == int.from_bytes(want.to_bytes(4, "little"), "big")
    # a slice-by-four CRC over the tables equals crc32()
    data = O.corpus(1001, 4, 5)
    c = 0xffffffff
    i = 0
    while i + 4 <= len(data):
        c ^= int.from_bytes(data[i:i + 4], "little")
        c = tab[3][c & 0xff] ^ tab[2][(c >> 8) & 0xff] ^ tab[1][(c >> 16) & 0xff] ^ tab[0][c >> 24]
        i += 4
    for b in data[i:]:
        c = tab[0][(c ^ b) & 0xff] ^ (c >> 8)
    assert c ^ 0xffffffff == zlib.crc32(data) == l.crc32(0, data, len(data))
    assert l.zamd_z_last_error() == 0


@pytest.mark.parametrize("wbits,first_flush", [(31, None), (15, Z.Z_SYNC_FLUSH), (-15, Z.Z_FULL_FLUSH), (15, Z.Z_PARTIAL_FLUSH), (-15, Z.Z_BLOCK)])
def test_overlapped_batches_equal_batches_in_a_row(monkeypatch, wbits, first_flush):
    """deflate() with several whole batches of input and room for their output overlaps upload, kernels and download of
    consecutive batches (the flush's own batch included).  The stream must be the one the same call produces batch by
    batch, and the one a caller with a small output buffer gets."""
    import zlib as pyz
    monkeypatch.setenv("ZAMD_Z_BATCH_MIB", "1")
    monkeypatch.setenv("ZAMD_Z_CHUNK", "262144")
    data = bytes(zlib_amd.corpus((5 << 20) + 300001, 4, 11))
    cut = (4 << 20) + 77
    flushes = {cut: first_flush} if first_flush is not None else None
    big = len(data) + (len(data) >> 3) + (1 << 16)
    idx_a, idx_b = [], []
    piped, tot_p = Z.deflate_stream(data, 6, wbits, out_step=big, flushes=flushes, index=idx_a)
    monkeypatch.setenv("ZAMD_Z_NO_PIPELINE", "1")
    plain, tot_q = Z.deflate_stream(data, 6, wbits, out_step=big, flushes=flushes, index=idx_b)
    monkeypatch.delenv("ZAMD_Z_NO_PIPELINE")
    small, tot_s = Z.deflate_stream(data, 6, wbits, out_step=70000, flushes=flushes)
    assert piped == plain and tot_p == tot_q and idx_a == idx_b
    assert small == plain and tot_s == tot_q
    assert pyz.decompressobj(wbits).decompress(piped) == data


def test_inflate_places_large_rounds_in_overlapped_groups():
    """With a chunk index that carries split points, a round of inflate() that can write straight into the caller's
    buffer is placed in groups whose uploads and downloads overlap the next group's kernels.  ZAMD_Z_INF_GROUP_MIB=16
    makes 16 MiB groups (read once per process, hence the child process): the bytes, the check value and the totals are
    those of one pass; an index or split records that lie, and a stream that is damaged in a late group, behave as
    they do without the groups."""
    import subprocess, sys
    code = r"""
import ctypes as C, sys, zlib
sys.path.insert(0, %r); sys.path.insert(0, %r)
import _zapi as Z, zlib_amd
n = 100 << 20
data = bytes(zlib_amd.corpus(n, 4, 31, piece_bytes=16 << 20, threads=8))
idx, spl = [], []
comp, totals = Z.deflate_stream(data, level=6, wbits=31, out_step=n + (1 << 20), index=idx, splits=spl)
assert idx[0] == 0 and len(idx) - 1 == n >> 20
pairs = idx[1:]
per = spl[0]
assert per == 7 and len(spl) - 1 == per * len(pairs) and sum(1 for r in spl[1:] if r) > 3 * len(pairs)
l = Z.lib()
def inflate_all(comp, hint, splits=None):
    s = Z.ZStream()
    assert l.inflateInit2_(C.byref(s), 47, Z.VERSION, Z.SZ) == 0
    if hint is not None:
        flat = (C.c_uint * (2 * len(hint)))(*[v for p in hint for v in p])
        assert l.zamd_inflate_index_hint(C.byref(s), flat, len(hint)) == 0
    if splits is not None:
        recs = (C.c_ulonglong * (len(splits) - 1))(*splits[1:])
        assert l.zamd_inflate_index_hint_splits(C.byref(s), recs, splits[0]) == 0
    src = C.create_string_buffer(comp, len(comp)); back = C.create_string_buffer(n + 16)
    s.next_in, s.avail_in, s.next_out, s.avail_out = C.addressof(src), len(comp), C.addressof(back), n + 16
    rc = l.inflate(C.byref(s), Z.Z_FINISH)
    out = (rc, back.raw[:s.total_out], s.msg, s.total_in, s.adler)
    l.inflateEnd(C.byref(s))
    return out
for hint, splits in ((pairs, spl), (pairs, None), (None, None)):
    rc, out, msg, tin, adler = inflate_all(comp, hint, splits)
    assert rc == Z.Z_STREAM_END and out == data and tin == len(comp) and adler == zlib.crc32(data), (rc, msg, len(out))
# an index whose chunk sizes are wrong from chunk 40 on: the chunks in front of it are placed by it, the rest without it
lying = list(pairs); lying[40] = (lying[40][0] + 1, lying[40][1]); lying[41] = (lying[41][0] - 1, lying[41][1])
rc, out, msg, tin, adler = inflate_all(comp, lying, spl)
assert rc == Z.Z_STREAM_END and out == data and adler == zlib.crc32(data), (rc, msg)
# split records that are no block boundaries (chunk 50), out of order (chunk 51) and outside their chunk (chunk 52)
wrong = list(spl)
wrong[1 + 50 * per] += 3 << 32
wrong[1 + 51 * per], wrong[2 + 51 * per] = wrong[2 + 51 * per], wrong[1 + 51 * per]
wrong[1 + 52 * per] = (0xfffffff0 << 32) | 77
rc, out, msg, tin, adler = inflate_all(comp, pairs, wrong)
assert rc == Z.Z_STREAM_END and out == data and adler == zlib.crc32(data), (rc, msg)
# damaged data in chunk 70: what lies in front of it arrives, then the error (of the block, or of the check value at the end)
bad = bytearray(comp); at = sum(p[0] for p in pairs[:70]) + 5000; bad[at] ^= 0x55
for hint, splits in ((pairs, spl), (pairs, None), (None, None)):
    rc, out, msg, tin, adler = inflate_all(bytes(bad), hint, splits)
    assert rc == Z.Z_DATA_ERROR and msg, (rc, msg)
    assert len(out) >= (60 << 20) and out[:60 << 20] == data[:60 << 20]
print("ok")
""" % (os.path.join(Z.ROOT, "tests"), Z.ROOT)
    env = dict(os.environ, ZAMD_Z_INF_GROUP_MIB="16")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout + r.stderr


@pytest.mark.parametrize("wbits,first_flush", [(31, None), (15, Z.Z_SYNC_FLUSH), (-15, Z.Z_FULL_FLUSH), (-15, Z.Z_PARTIAL_FLUSH)])
def test_input_of_whole_batches_ends_like_any_other(monkeypatch, wbits, first_flush):
    """Input that is exactly so many engine batches long: the flush belongs to the last batch that has input (the final
    block holds data, a sync flush writes one marker), so the stream does not depend on the batch size (deflate.c:873-945
    knows no batches).  Overlapped, batch by batch and with a small output buffer."""
    import zlib as pyz
    monkeypatch.setenv("ZAMD_Z_CHUNK", "262144")
    data = bytes(zlib_amd.corpus(5 << 20, 4, 12))
    cut = 2 << 20
    flushes = {cut: first_flush} if first_flush is not None else None
    big = len(data) + (len(data) >> 3) + (1 << 16)
    monkeypatch.setenv("ZAMD_Z_BATCH_MIB", "16")
    want, tot_w = Z.deflate_stream(data, 6, wbits, out_step=big, flushes=flushes)          # every call is one batch
    for batch in ("1", "5"):
        monkeypatch.setenv("ZAMD_Z_BATCH_MIB", batch)
        piped, tot_p = Z.deflate_stream(data, 6, wbits, out_step=big, flushes=flushes)
        monkeypatch.setenv("ZAMD_Z_NO_PIPELINE", "1")
        plain, tot_q = Z.deflate_stream(data, 6, wbits, out_step=big, flushes=flushes)
        monkeypatch.delenv("ZAMD_Z_NO_PIPELINE")
        small, tot_s = Z.deflate_stream(data, 6, wbits, out_step=70000, flushes=flushes)
        fed, tot_f = Z.deflate_stream(data, 6, wbits, out_step=big, flushes=flushes, in_step=(1 << 20) - 4096)
        assert piped == want and plain == want and small == want and fed == want, (batch, len(piped), len(plain), len(small), len(fed), len(want))
        assert tot_p == tot_w and tot_q == tot_w and tot_s == tot_w and tot_f == tot_w
    assert pyz.decompressobj(wbits).decompress(want) == data
    # sizes at which the row of overlapped engine calls has nothing, or less than a chunk, left for its call before the last
    for n in ((3 << 20) + 77, (3 << 20), (5 << 20) - 262144 + 5, (3 << 20) + 262144):
        monkeypatch.setenv("ZAMD_Z_BATCH_MIB", "16")
        want, tot_w = Z.deflate_stream(data[:n], 6, wbits, out_step=big)
        monkeypatch.setenv("ZAMD_Z_BATCH_MIB", "1")
        piped, tot_p = Z.deflate_stream(data[:n], 6, wbits, out_step=big)
        assert piped == want and tot_p == tot_w, (n, len(piped), len(want))


def test_random_calls_of_the_stream_api_agree():
    """tests/gpu_api_fuzz.py: random batch and chunk sizes, input and output steps and flush points; the stream is the one
    a single batch gives, Python's zlib inflates it, and so does inflate() with random steps, with and without the index.
    (A child process: the tool sets the sizes through the environment.)"""
    import subprocess, sys
    r = subprocess.run([sys.executable, os.path.join(Z.ROOT, "tests", "gpu_api_fuzz.py"), "30", "11"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "bad 0" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_random_streams_of_another_producer_inflate():
    """tests/gpu_foreign_fuzz.py: streams of Python's zlib (the system's libz) with random levels, strategies, window sizes,
    wrappers and flush points of every kind, through inflate() with random steps; trailing bytes stay with the caller.
    (660 cases were run when the tool was written; thirty here.)"""
    import subprocess, sys
    r = subprocess.run([sys.executable, os.path.join(Z.ROOT, "tests", "gpu_foreign_fuzz.py"), "30", "41"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "bad 0" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
