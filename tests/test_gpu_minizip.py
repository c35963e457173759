"""GPU tests: the reference's own zip tools (contrib/minizip/minizip.c, miniunz.c with zip.c, unzip.c, ioapi.c),
compiled unmodified from where they lie in the reference tree and linked with libzamd_z.so instead of libz
(oracle/Makefile, target ref).  They use raw deflate (-MAX_WBITS, zip.c:1238), crc32 over 16 KiB pieces
(zip.c:1421), deflate() with 64 KiB output buffers (zip.c:1435-1470) and inflate() with 16 KiB input buffers
(unzip.c:1698-1840): an application's view of the drop-in claim.  Python's zipfile (system zlib) is the
second reader and writer."""
import os
import subprocess
import zipfile

import pytest

import _oracle as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MINIZIP = os.path.join(ROOT, "oracle", "_ref", "minizip_zamd")
MINIUNZ = os.path.join(ROOT, "oracle", "_ref", "miniunz_zamd")


def _tools():
    if not (os.path.exists(MINIZIP) and os.path.exists(MINIUNZ)):
        pytest.skip("the reference's minizip was not built against this library (make -C oracle ref)")


def _files(tmp_path):
    files = {
        "text.txt": O.corpus(3 * (1 << 20) + 777, 0, 201),
        "mixed.bin": O.corpus(2 * (1 << 20) + 12345, 4, 202),
        "noise.bin": O.corpus(300000, 1, 203),
        "small.txt": b"hello, hello!\n",
        "empty": b"",
    }
    for name, data in files.items():
        (tmp_path / name).write_bytes(data)
    return files


@pytest.mark.parametrize("level", [1, 6, 9])
def test_minizip_writes_archives_other_readers_accept(tmp_path, level):
    _tools()
    files = _files(tmp_path)
    r = subprocess.run([MINIZIP, "-o", "-%d" % level, "a.zip"] + sorted(files), cwd=tmp_path, check=True, timeout=300,
                       stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=dict(os.environ, ZAMD_Z_TRACE="1"))
    assert b"zamd_z deflate" in r.stderr                              # the GPU library did the work, not a libz further down the link order
    with zipfile.ZipFile(tmp_path / "a.zip") as z:
        assert z.testzip() is None                                   # every member's crc32 as this library computed it
        assert sorted(z.namelist()) == sorted(files)
        for name, data in files.items():
            assert z.read(name) == data
            info = z.getinfo(name)
            assert info.file_size == len(data)
            if len(data) > 100000 and name != "noise.bin":
                assert info.compress_type == zipfile.ZIP_DEFLATED and info.compress_size < len(data) * 0.7
    # and back through the reference's reader on this library
    out = tmp_path / "out"
    out.mkdir()
    subprocess.run([MINIUNZ, "-o", str(tmp_path / "a.zip"), "-d", str(out)], check=True, timeout=300,
                   stdout=subprocess.DEVNULL)
    for name, data in files.items():
        assert (out / name).read_bytes() == data


def test_miniunz_reads_archives_of_the_system_zlib(tmp_path):
    _tools()
    files = _files(tmp_path)
    with zipfile.ZipFile(tmp_path / "p.zip", "w", zipfile.ZIP_DEFLATED, compresslevel=6) as z:
        for name in sorted(files):
            z.write(tmp_path / name, name)
        z.writestr("stored.bin", files["noise.bin"], compress_type=zipfile.ZIP_STORED)
    out = tmp_path / "out"
    out.mkdir()
    r = subprocess.run([MINIUNZ, "-o", str(tmp_path / "p.zip"), "-d", str(out)], check=True, timeout=300,
                       stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=dict(os.environ, ZAMD_Z_TRACE="1"))
    assert b"zamd_z inflate" in r.stderr
    for name, data in files.items():
        assert (out / name).read_bytes() == data
    assert (out / "stored.bin").read_bytes() == files["noise.bin"]


def test_miniunz_reports_a_damaged_member(tmp_path):
    _tools()
    data = O.corpus(1 << 20, 0, 205)
    with zipfile.ZipFile(tmp_path / "d.zip", "w", zipfile.ZIP_DEFLATED) as z:
        z.writestr("t.txt", data)
    blob = bytearray((tmp_path / "d.zip").read_bytes())
    blob[len(blob) // 2] ^= 0x55                                      # inside the deflate data
    (tmp_path / "d.zip").write_bytes(bytes(blob))
    out = tmp_path / "out"
    out.mkdir()
    r = subprocess.run([MINIUNZ, "-o", str(tmp_path / "d.zip"), "-d", str(out)], timeout=300,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    got = (out / "t.txt").read_bytes() if (out / "t.txt").exists() else b""
    assert got != data
    assert b"error" in r.stdout.lower()                                # unzReadCurrentFile < 0 or the crc check of unzCloseCurrentFile
