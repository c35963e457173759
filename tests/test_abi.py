"""The C-ABI shared library loads without a GPU and exports every symbol that
include/*.h declares; calls that need a device fail loudly instead of falling
back to a CPU path.  CPU only (no compute calls)."""
import ctypes
import subprocess
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    names = set()
    for fn in os.listdir(os.path.join(ROOT, "include")):
        if not fn.endswith(".h") or fn == "zamd_zlib.h":       # that header is libzamd_z.so's, checked below
            continue
        text = open(os.path.join(ROOT, "include", fn)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        for m in re.finditer(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\(", text):
            name = m.group(1)
            if name.startswith(("zamd_",)) and not name.isupper():
                names.add(name)
    return sorted(names)


def test_library_exports_every_declared_symbol():
    import zlib_amd
    lib = zlib_amd.load_library()
    names = _declared_symbols()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/ but not exported by libzamd.so"


def test_host_side_arithmetic_entry_points():
    import zlib as pyz
    import zlib_amd
    a, b = b"hello, hello!" * 1000, bytes(range(256)) * 77
    assert zlib_amd.crc32_combine(pyz.crc32(a), pyz.crc32(b), len(b)) == pyz.crc32(a + b)
    assert zlib_amd.adler32_combine(pyz.adler32(a), pyz.adler32(b), len(b)) == pyz.adler32(a + b)
    assert zlib_amd.crc32_combine(123, 456, 0) == 123
    lib = zlib_amd.load_library()
    assert lib.zamd_deflate_bound(1 << 20, 1 << 20, 8) >= (1 << 20) + 5 * 65
    # corpus generator is the fixture definition
    import _oracle as O
    for kind in range(5):
        assert bytes(zlib_amd.corpus(200000, kind, kind + 1)) == O.corpus(200000, kind, kind + 1)


def test_no_cpu_fallback_without_a_device():
    import torch
    import zlib_amd
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(zlib_amd.ZamdError):
        zlib_amd.Engine(0)


def _zlib_api_symbols():
    text = open(os.path.join(ROOT, "include", "zamd_zlib.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = "\n".join(l for l in text.splitlines() if not l.lstrip().startswith("#") and "typedef" not in l)
    names = set(m.group(1) for m in re.finditer(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\(", text))
    return sorted(n for n in names if n[0].islower() and n not in ("sizeof",))


def test_zlib_api_library_exports_every_declared_function():
    import _zapi as Z
    lib = Z.lib()
    names = _zlib_api_symbols()
    assert len(names) >= 44, names
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/zamd_zlib.h but not exported by libzamd_z.so"


def test_zlib_api_host_logic_without_compute():
    """Argument checking of the stream API (deflate.c:233-275, inflate.c:189-192, zutil.c:30-40, 136)."""
    import ctypes as C
    import _zapi as Z
    lib = Z.lib()
    assert lib.zlibVersion() == b"1.2.8"
    assert lib.zlibCompileFlags() == 0xa9
    assert lib.zError(-3) == b"data error" and lib.zError(2) == b"need dictionary" and lib.zError(0) == b""
    assert C.sizeof(Z.ZStream) == 112
    s = Z.ZStream()
    assert lib.deflateInit_(C.byref(s), 6, b"2.0", Z.SZ) == Z.Z_VERSION_ERROR
    assert lib.deflateInit_(C.byref(s), 6, Z.VERSION, Z.SZ - 8) == Z.Z_VERSION_ERROR
    assert lib.inflateInit_(C.byref(s), None, Z.SZ) == Z.Z_VERSION_ERROR
    assert lib.deflateInit_(None, 6, Z.VERSION, Z.SZ) == Z.Z_STREAM_ERROR
    for bad in ((10, 8, 15, 8, 0), (6, 7, 15, 8, 0), (6, 8, 16 + 16, 8, 0), (6, 8, 7, 8, 0), (6, 8, 15, 0, 0), (6, 8, 15, 10, 0), (6, 8, 15, 8, 5)):
        assert lib.deflateInit2_(C.byref(s), *bad, Z.VERSION, Z.SZ) == Z.Z_STREAM_ERROR, bad
    assert lib.inflateInit2_(C.byref(s), 7, Z.VERSION, Z.SZ) == Z.Z_STREAM_ERROR
    # a stream can be set up and torn down without touching the GPU
    assert lib.deflateInit2_(C.byref(s), -1, 8, 31, 9, 1, Z.VERSION, Z.SZ) == Z.Z_OK
    assert s.zalloc and s.zfree and s.adler == 0 and s.total_in == 0
    assert lib.deflate(C.byref(s), 9) == Z.Z_STREAM_ERROR            # bad flush value
    assert lib.deflate(C.byref(s), 0) == Z.Z_STREAM_ERROR            # next_out == NULL
    assert lib.deflateSetDictionary(C.byref(s), b"abc", 3) == Z.Z_STREAM_ERROR
    assert lib.deflateEnd(C.byref(s)) == Z.Z_OK and not s.state
    assert lib.deflateEnd(C.byref(s)) == Z.Z_STREAM_ERROR
    assert lib.deflateInit2_(C.byref(s), 6, 8, -15, 8, 0, Z.VERSION, Z.SZ) == Z.Z_OK
    assert lib.deflateEnd(C.byref(s)) == Z.Z_DATA_ERROR              # raw streams start BUSY (deflate.c:413, 1006)
    assert lib.inflateInit2_(C.byref(s), 47, Z.VERSION, Z.SZ) == Z.Z_OK
    assert lib.inflate(C.byref(s), 0) == Z.Z_STREAM_ERROR            # next_out == NULL
    assert lib.inflateUndermine(C.byref(s), 1) == Z.Z_DATA_ERROR
    assert lib.inflateMark(C.byref(s)) == -65536
    assert lib.deflate(C.byref(s), 0) == Z.Z_STREAM_ERROR            # an inflate state is not a deflate state
    assert lib.inflateEnd(C.byref(s)) == Z.Z_OK
    assert lib.compressBound(1 << 20) >= (1 << 20) + (1 << 8) + (1 << 6) + 13
    tab = lib.get_crc_table()
    assert tab[1] == 0x77073096 and tab[255] == 0x2d02ef8d


def test_symbol_versions_and_soname():
    """libzamd_z.so carries the reference's symbol versions (zlib.map:1-83), so that programs linked against a versioned
    libz find e.g. deflatePending@ZLIB_1.2.5.1 when it is preloaded.  Its SONAME is its own: as libz.so.1 it would also be
    handed to libelf inside the ROCm runtime, whose inflate() calls during HSA start-up would re-enter this library."""
    so = os.path.join(ROOT, "zlib_amd", "libzamd_z.so")
    dyn = subprocess.run(["readelf", "-d", so], capture_output=True, text=True).stdout
    assert "Library soname: [libzamd_z.so]" in dyn
    ver = subprocess.run(["readelf", "-V", so], capture_output=True, text=True).stdout
    for node in ("ZLIB_1.2.0", "ZLIB_1.2.2.3", "ZLIB_1.2.5.1", "ZLIB_1.2.7.1"):
        assert node in ver, node
    syms = subprocess.run(["nm", "-D", so], capture_output=True, text=True).stdout
    assert "deflatePending@@ZLIB_1.2.5.1" in syms and "inflateGetDictionary@@ZLIB_1.2.7.1" in syms
    assert " T _Z" not in syms                       # no C++ internals exported
    exe = os.path.join(ROOT, "examples", "zamd_gzip")
    if os.path.exists(exe):
        ldd = subprocess.run(["ldd", exe], capture_output=True, text=True).stdout
        line = [ln for ln in ldd.splitlines() if "libzamd_z.so" in ln]
        assert line and os.path.realpath(line[0].split("=>")[1].split("(")[0].strip()) == os.path.realpath(so), ldd


def test_a_copy_that_runs_out_of_memory_leaves_no_shared_state():
    """deflateCopy / inflateCopy with an allocator that refuses in the middle of the copy (the mem_limit of
    test/infcover.c:55-234): Z_MEM_ERROR, dest->state is NULL (deflate.c:1036-1053) or dest untouched
    (inflate.c:1455-1466), everything the copy got went back, and the source still works.  No GPU: input shorter
    than a batch only waits in the state."""
    import ctypes as C
    import _zapi as Z
    l = Z.lib()
    libc = C.CDLL(None)
    libc.malloc.restype = C.c_void_p; libc.malloc.argtypes = [C.c_size_t]
    libc.free.argtypes = [C.c_void_p]
    live, ctl = {}, {"allow": 1 << 30}

    def zalloc(opaque, items, size):
        if ctl["allow"] <= 0:
            return None
        ctl["allow"] -= 1
        p = libc.malloc(items * size)
        live[p] = items * size
        return p

    def zfree(opaque, p):
        assert p in live, "freed a block that did not come from zalloc (or twice)"
        del live[p]
        libc.free(p)

    za, zf = Z.ALLOC(zalloc), Z.FREE(zfree)
    data = bytes(range(256)) * 1200
    src = C.create_string_buffer(data, len(data))
    obuf = C.create_string_buffer(1 << 16)
    s = Z.ZStream()
    s.zalloc, s.zfree = C.cast(za, C.c_void_p), C.cast(zf, C.c_void_p)
    assert l.deflateInit2_(C.byref(s), 6, 8, 15, 8, 0, Z.VERSION, Z.SZ) == Z.Z_OK
    s.next_in, s.avail_in, s.next_out, s.avail_out = C.addressof(src), len(data), C.addressof(obuf), 1 << 16
    assert l.deflate(C.byref(s), Z.Z_NO_FLUSH) == Z.Z_OK and s.avail_in == 0      # buffered, mid-stream
    before = dict(live)
    src_state = s.state
    for allow in (0, 1, 2):                  # the state block itself, then the copy's buffers
        d = Z.ZStream()
        ctl["allow"] = allow
        rc = l.deflateCopy(C.byref(d), C.byref(s))
        ctl["allow"] = 1 << 30
        if rc == Z.Z_OK:
            assert d.state and d.state != src_state
            assert l.deflateEnd(C.byref(d)) in (Z.Z_OK, Z.Z_DATA_ERROR)
        else:
            assert rc == Z.Z_MEM_ERROR and not d.state, (allow, rc, d.state)
        assert live == before, "a failed copy kept memory"
    assert l.deflateEnd(C.byref(s)) == Z.Z_DATA_ERROR and not live            # still BUSY: deflate.c:1006

    i = Z.ZStream()
    i.zalloc, i.zfree = C.cast(za, C.c_void_p), C.cast(zf, C.c_void_p)
    assert l.inflateInit2_(C.byref(i), 15, Z.VERSION, Z.SZ) == Z.Z_OK
    half = bytes.fromhex("789c")                                              # a zlib header and nothing more
    hsrc = C.create_string_buffer(half, len(half))
    i.next_in, i.avail_in, i.next_out, i.avail_out = C.addressof(hsrc), len(half), C.addressof(obuf), 1 << 16
    assert l.inflate(C.byref(i), Z.Z_NO_FLUSH) in (Z.Z_OK, Z.Z_BUF_ERROR)
    before = dict(live)
    for allow in (0, 1):
        d = Z.ZStream()
        d.total_in = 777                                                       # must survive a failed copy
        ctl["allow"] = allow
        rc = l.inflateCopy(C.byref(d), C.byref(i))
        ctl["allow"] = 1 << 30
        if rc == Z.Z_OK:
            assert d.state and d.state != i.state
            assert l.inflateEnd(C.byref(d)) == Z.Z_OK
        else:
            assert rc == Z.Z_MEM_ERROR and not d.state and d.total_in == 777
        assert live == before
    assert l.inflateEnd(C.byref(i)) == Z.Z_OK and not live
