"""Manual/automated check: run with LD_PRELOAD=zlib_amd/libzamd_z.so so that Python's own zlib module
(linked against the system libz) is served by the GPU library instead."""
import sys
import zlib

sys.path.insert(0, "tests")
import _oracle as O

data = O.corpus(3 * (1 << 20) + 4567, 4, 5)
small = O.corpus(200000, 0, 1)
c = zlib.compress(small, 6)
want = bytes([0x78, 0x9c]) + O.deflate_chunk(small, 0, len(small), 6, 0, 8, True) + O.adler32(small).to_bytes(4, "big")
print("one-chunk compress identical to the reference bytes:", c == want)
big = zlib.compress(data, 6)
print("multi-chunk compress has sync markers:", big.count(b"\x00\x00\xff\xff") >= 3, len(big))
print("decompress round trip:", zlib.decompress(big) == data)
co = zlib.compressobj(9, zlib.DEFLATED, 31)
g = co.compress(data[:1000000]) + co.flush(zlib.Z_SYNC_FLUSH) + co.compress(data[1000000:]) + co.flush()
do = zlib.decompressobj(47)
back = do.decompress(g[:70000]) + do.decompress(g[70000:]) + do.flush()
print("gzip streaming round trip:", back == data, do.eof)
print("crc32/adler32:", zlib.crc32(data) == O.crc32(data), zlib.adler32(data) == O.adler32(data))
try:
    zlib.decompress(b"\x78\x9c\x06")
except zlib.error as e:
    print("error text:", e)
print("runtime version:", zlib.ZLIB_RUNTIME_VERSION)
