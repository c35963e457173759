"""The C-ABI shared library loads without a GPU and exports every symbol that
include/*.h declares; calls that need a device fail loudly instead of falling
back to a CPU path.  CPU only (no compute calls)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    names = set()
    for fn in os.listdir(os.path.join(ROOT, "include")):
        if not fn.endswith(".h"):
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
