"""GPU tests: the chunk index — per chunk of the deflate data its compressed and inflated length — as the
stream API hands it out (zamd_deflate_index*), as inflate() takes it (zamd_inflate_index_hint), and as the gz
layer keeps it in a file (trailing empty gzip members, found from the end of the file).  The reference has no
such index (examples/zran.c:67-72 builds access points by inflating once); what is checked is that the index
describes the stream exactly (every chunk inflates on its own, primed with the 32 KiB before it, to the stated
length), that hinted inflation gives the same bytes as unhinted, that wrong hints cost nothing but time, and
that foreign gzip readers are not disturbed by the index members."""
import ctypes as C
import gzip
import os
import struct
import zlib

import pytest

import _oracle as O
import _zapi as Z

pytestmark = pytest.mark.gpu
MIB = 1 << 20


def test_index_of_a_stream_describes_every_chunk():
    data = O.corpus(5 * MIB + 12345, 4, 201)
    idx = []
    comp, totals = Z.deflate_stream(data, level=6, wbits=-15, in_step=3 * MIB + 17, out_step=1 << 20, index=idx)
    assert idx[0] == Z.Z_OK
    pairs = idx[1:]
    assert [r for _, r in pairs] == [MIB] * 5 + [12345]
    assert sum(c for c, _ in pairs) == len(comp)
    # every chunk is a deflate stream of its own once the window is primed
    cpos = rpos = 0
    for c, r in pairs:
        d = zlib.decompressobj(-15, zdict=data[max(0, rpos - 32768):rpos]) if rpos else zlib.decompressobj(-15)
        assert d.decompress(comp[cpos:cpos + c]) == data[rpos:rpos + r]
        cpos += c; rpos += r
    # and the bytes are the oracle's for the same chunk recipe
    assert comp == O.deflate_chunked(data, MIB, 6, 0, 8)[0]


@pytest.mark.parametrize("wbits", [-15, 15, 31])
def test_hinted_inflate_gives_the_same_bytes(wbits):
    data = O.corpus(9 * MIB + 4321, 4, 203)
    idx = []
    comp, _ = Z.deflate_stream(data, level=6, wbits=wbits, out_step=4 << 20, index=idx)
    pairs = idx[1:]
    for in_step, out_step in ((None, 16 << 20), (3 * MIB + 5, 1 << 20), (70000, 5 * MIB)):
        rc, out, msg, unused, totals = Z.inflate_stream(comp, wbits=wbits, in_step=in_step, out_step=out_step, hint=pairs)
        assert (rc, msg, unused) == (Z.Z_STREAM_END, None, 0), (in_step, rc, msg)
        assert out == data


def test_wrong_hints_are_dropped():
    data = O.corpus(4 * MIB + 999, 0, 205)
    idx = []
    comp, _ = Z.deflate_stream(data, level=6, wbits=15, out_step=4 << 20, index=idx)
    pairs = idx[1:]
    shifted = [(c + 7, r) for c, r in pairs]
    swapped = [(c, r - 1) for c, r in pairs]
    short = pairs[:2]
    for hint in (shifted, swapped, short, [(123, 456)]):
        rc, out, msg, unused, _ = Z.inflate_stream(comp, wbits=15, out_step=8 << 20, hint=hint)
        assert (rc, msg, unused) == (Z.Z_STREAM_END, None, 0)
        assert out == data
    # damaged data under a correct index is still an error
    bad = bytearray(comp); bad[len(bad) // 2] ^= 0x55
    rc, out, msg, _, _ = Z.inflate_stream(bytes(bad), wbits=15, out_step=8 << 20, hint=pairs)
    assert rc == Z.Z_DATA_ERROR


def test_deflate_prime_leaves_no_index():
    l = Z.lib()
    s = Z.ZStream()
    assert l.deflateInit2_(C.byref(s), 6, 8, -15, 8, 0, Z.VERSION, Z.SZ) == 0
    assert l.zamd_deflate_index(C.byref(s), 1) == 0
    assert l.deflatePrime(C.byref(s), 3, 5) == 0
    pairs, cnt = C.POINTER(C.c_uint)(), C.c_uint(0)
    assert l.zamd_deflate_index_get(C.byref(s), C.byref(pairs), C.byref(cnt)) == Z.Z_DATA_ERROR
    assert l.deflateEnd(C.byref(s)) in (0, Z.Z_DATA_ERROR)


def _parse_index(blob):
    """An independent reader of the file layout: the last 22 bytes lead to the first index member."""
    assert blob[-14:-10] == b"ZAMI" and blob[-10:] == b"\x03\x00" + bytes(8)
    at = struct.unpack("<Q", blob[-22:-14])[0]
    first_index = at
    pairs, member = [], None
    while True:
        assert blob[at:at + 4] == b"\x1f\x8b\x08\x04" and blob[at + 12:at + 14] == b"ZI"
        xlen, = struct.unpack("<H", blob[at + 10:at + 12])
        plen, = struct.unpack("<H", blob[at + 14:at + 16])
        ver, flags, _, n, moff, first = struct.unpack("<BBHIQQ", blob[at + 16:at + 40])
        assert ver == 1 and xlen == plen + 4 and first == len(pairs)
        member = moff if member is None else member
        assert member == moff
        pairs += [struct.unpack("<II", blob[at + 40 + 8 * i:at + 48 + 8 * i]) for i in range(n)]
        at += 12 + xlen + 10
        if flags & 1:
            break
    assert at == len(blob)
    return member, pairs, first_index


def test_gz_file_with_index(tmp_path):
    l = Z.lib()
    data = O.corpus(6 * MIB + 777, 4, 207)
    name = str(tmp_path / "i.gz")
    g = l.gzopen(name.encode(), b"wb6I")
    assert g
    assert l.gzwrite(g, data[:2 * MIB + 5], 2 * MIB + 5) == 2 * MIB + 5
    rest = data[2 * MIB + 5:]
    assert l.gzwrite(g, rest, len(rest)) == len(rest)
    assert l.gzclose(g) == Z.Z_OK
    blob = open(name, "rb").read()
    # foreign readers see the data and nothing else
    assert gzip.decompress(blob) == data
    member, pairs, first_index = _parse_index(blob)
    assert member == 0 and sum(r for _, r in pairs) == len(data)
    # header, deflate data, trailer, then one member with the chunks' split records ('ZS': 7 x u64 a chunk), then the index
    data_end = 10 + sum(c for c, _ in pairs) + 8
    assert blob[data_end:data_end + 4] == b"\x1f\x8b\x08\x04" and blob[data_end + 12:data_end + 14] == b"ZS" and blob[data_end + 17] == 7
    assert data_end + 12 + 4 + 16 + 8 * 7 * len(pairs) + 10 == first_index
    cpos, rpos = 10, 0
    for c, r in pairs:
        d = zlib.decompressobj(-15, zdict=data[max(0, rpos - 32768):rpos]) if rpos else zlib.decompressobj(-15)
        assert d.decompress(blob[cpos:cpos + c]) == data[rpos:rpos + r]
        cpos += c; rpos += r
    # the library's loader agrees with the independent parse
    ix = Z.GzIndex()
    fd = os.open(name, os.O_RDONLY)
    assert l.zamd_gz_index_load(fd, C.byref(ix)) == 0
    assert ix.member_offset == 0 and [(ix.pairs[2 * i], ix.pairs[2 * i + 1]) for i in range(ix.n_chunks)] == pairs
    assert ix.splits_per_chunk == 7
    assert [ix.splits[k] for k in range(7 * ix.n_chunks)] == [int.from_bytes(blob[data_end + 32 + 8 * k:data_end + 40 + 8 * k], "little") for k in range(7 * len(pairs))]
    l.zamd_gz_index_free(C.byref(ix))
    os.close(fd)
    # reading back with the index, in ragged pieces, with a rewind in between
    g = l.gzopen(name.encode(), b"rb")
    buf = C.create_string_buffer(3 * MIB + 1)
    got = bytearray()
    while True:
        n = l.gzread(g, buf, len(buf))
        assert n >= 0
        if n == 0:
            break
        got += buf.raw[:n]
    assert bytes(got) == data and l.gzeof(g) == 1
    assert l.gzseek(g, MIB + 3, 0) == MIB + 3
    assert l.gzread(g, buf, 1000) == 1000 and buf.raw[:1000] == data[MIB + 3:MIB + 1003]
    assert l.gzclose(g) == Z.Z_OK
    # a file without index has none to load, and an index behind appended data is not mistaken for one of the new file
    plain = str(tmp_path / "p.gz")
    open(plain, "wb").write(gzip.compress(data[:100000]))
    fd = os.open(plain, os.O_RDONLY)
    assert l.zamd_gz_index_load(fd, C.byref(ix)) == -1
    os.close(fd)
    g = l.gzopen(plain.encode(), b"abI")                               # append an indexed member behind a foreign one
    assert l.gzwrite(g, data[:MIB + 9], MIB + 9) == MIB + 9
    assert l.gzclose(g) == Z.Z_OK
    blob2 = open(plain, "rb").read()
    member2, pairs2, _ = _parse_index(blob2)
    assert member2 == len(gzip.compress(data[:100000])) and sum(r for _, r in pairs2) == MIB + 9
    assert gzip.decompress(blob2) == data[:100000] + data[:MIB + 9]
    g = l.gzopen(plain.encode(), b"rb")
    got = bytearray()
    while True:
        n = l.gzread(g, buf, len(buf))
        if n <= 0:
            break
        got += buf.raw[:n]
    assert bytes(got) == data[:100000] + data[:MIB + 9]
    assert l.gzclose(g) == Z.Z_OK


def test_no_index_unless_asked(tmp_path):
    l = Z.lib()
    name = str(tmp_path / "n.gz")
    g = l.gzopen(name.encode(), b"wb")
    assert l.gzwrite(g, b"x" * 100000, 100000) == 100000
    assert l.gzclose(g) == Z.Z_OK
    blob = open(name, "rb").read()
    assert blob[-14:-10] != b"ZAMI" and gzip.decompress(blob) == b"x" * 100000
