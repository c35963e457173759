"""The entry point that allocates nothing (zamd_checksums_noalloc_dev, include/zamd.h): its per-slice partial
checksums live in a fixed array of the library's code object.  Called through the C ABI and checked against the
entry points it is a variant of (zamd_adler32_dev / zamd_crc32_dev) and against the oracle, on two calls in a row,
the second with a smaller input, so that what the first call left in the array must not leak into the second."""
import ctypes as C
import struct

import pytest

import _oracle as O
import zlib_amd

pytestmark = pytest.mark.gpu


def _both(engine, d_in, n, d_res, adler=1, crc=0):
    rc = engine.lib.zamd_checksums_noalloc_dev(engine.handle, adler, crc, d_in.ptr, n, d_res.ptr)
    assert rc == 0, rc
    return struct.unpack("<2I", bytes(d_res.download(8)))


def test_no_alloc_checksums_match_the_allocating_entry_points(engine):
    big = O.corpus(5 * (1 << 20) + 12345, 4, 901)          # six slices, the last one ragged
    small = O.corpus((1 << 20) + 77, 1, 902)               # two slices: four of the first call's partials stay behind
    d_in = engine.alloc(len(big) + 16)
    d_res = engine.alloc(16)
    try:
        for data, seeds in ((big, (1, 0)), (small, (1, 0)), (small[:1000], (0x12345, 0xdeadbeef)), (b"", (7, 9)), (big, (0x00010001, 0xffffffff))):
            n = len(data)
            if n:
                d_in.upload(data)
            got = _both(engine, d_in, n, d_res, *seeds)
            want = (engine.adler32_dev(d_in.ptr, n, seeds[0]) if n else seeds[0], engine.crc32_dev(d_in.ptr, n, seeds[1]) if n else seeds[1])
            assert got == want, (n, seeds)
            assert got == (O.adler32(data, seeds[0]), O.crc32(data, seeds[1])), (n, seeds)
    finally:
        d_in.free(); d_res.free()


def test_no_alloc_refuses_what_its_array_cannot_hold(engine):
    d_res = engine.alloc(16)
    try:
        rc = engine.lib.zamd_checksums_noalloc_dev(engine.handle, 1, 0, d_res.ptr, (16 << 30) + 1, d_res.ptr)
        assert rc == zlib_amd.Z_BUF_ERROR
        assert engine.lib.zamd_checksums_noalloc_dev(engine.handle, 1, 0, d_res.ptr, 8, None) == zlib_amd.Z_STREAM_ERROR
    finally:
        d_res.free()
