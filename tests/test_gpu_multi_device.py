"""One deflate call spread over a list of devices inside the library (zamd_deflate_multi, ZAMD_DEVICES): the bytes are
those of one engine.  One GPU is all a test box has: the list [0] takes the RCCL path (ncclAllGather in a communicator
of one rank), the lists [0, 0] and [0, 0, 0] name a device twice, which RCCL refuses, and take the exchange through host
memory; both end in the same table of chunk offsets."""
import hashlib
import os
import subprocess
import sys
import zlib

import pytest

import _oracle as O
import zlib_amd

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def multis():
    ms = {k: zlib_amd.MultiEngine(list(k)) for k in ((0,), (0, 0), (0, 0, 0))}
    yield ms
    for m in ms.values():
        m.close()


def test_exchange_paths(multis):
    assert multis[(0,)].uses_rccl, "librccl.so is part of the ROCm image: a list of distinct devices must use it"
    assert not multis[(0, 0)].uses_rccl and not multis[(0, 0, 0)].uses_rccl


CASES = [(0, 5 * 65536 + 123, 6, 0, 65536), (4, 1 << 20, 9, 0, 131072), (2, 300000, 1, 0, 32768), (1, 200000, 0, 0, 65536),
         (3, 7 * 32768, 6, 3, 32768), (0, 40000, 6, 0, 65536), (0, 0, 6, 0, 65536), (4, 3 * (1 << 20) + 5, 4, 0, 1 << 20)]


@pytest.mark.parametrize("kind,n,level,strategy,chunk", CASES)
def test_bytes_are_those_of_one_engine(engine, multis, kind, n, level, strategy, chunk):
    data = O.corpus(n, kind, 77 + kind)
    want, want_offs = engine.deflate_chunks(data, level=level, strategy=strategy, chunk_size=chunk)
    for devs, m in multis.items():
        got, offs, a, c = m.deflate(data, level=level, strategy=strategy, chunk_size=chunk)
        assert got == want, (devs, len(got), len(want))
        assert offs == want_offs
        assert a == zlib.adler32(data) and c == zlib.crc32(data)
    assert zlib.decompressobj(-15).decompress(want) == data


def test_dictionary_and_unfinished_streams(engine, multis):
    data = O.corpus(6 * 65536 + 99, 0, 5)
    dictionary = O.corpus(20000, 0, 6)
    want, want_offs = engine.deflate_chunks(data, level=6, chunk_size=65536, dictionary=dictionary, finish=False)
    for devs, m in multis.items():
        got, offs, a, c = m.deflate(data, level=6, chunk_size=65536, dictionary=dictionary, finish=False, adler=12345, crc=999)
        assert got == want and offs == want_offs, devs
        assert a == zlib.adler32(data, 12345) and c == zlib.crc32(data, 999)


def test_deflate_of_the_stream_api_uses_the_device_list():
    # a one-shot deflate() of several batches with ZAMD_DEVICES set: the same stream as without, and the trace names the path
    code = r"""
import ctypes as C, hashlib, sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import _zapi as Z, zlib_amd
n = 40 << 20
data = bytes(zlib_amd.corpus(n, 4, 9, piece_bytes=16 << 20, threads=8))
idx = []
comp, totals = Z.deflate_stream(data, level=6, wbits=31, out_step=n + (1 << 20), index=idx)
print(hashlib.sha256(comp).hexdigest(), totals[2], idx[0], len(idx) - 1)
rc, out, msg, unused, t = Z.inflate_stream(comp, wbits=47, out_step=n + 16, hint=idx[1:])
assert rc == Z.Z_STREAM_END and out == data
""" % (os.path.join(ROOT, "tests"), ROOT)
    outs = []
    for devices in (None, "0,0"):
        env = dict(os.environ, ZAMD_Z_BATCH_MIB="8", ZAMD_Z_TRACE="1")
        env.pop("ZAMD_DEVICES", None)
        if devices:
            env["ZAMD_DEVICES"] = devices
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout + r.stderr
        assert ("deflate, several devices" in r.stderr) == bool(devices), r.stderr[-2000:]
        outs.append(r.stdout.strip().splitlines()[-1])
    assert outs[0] == outs[1]
