"""Quick GPU sanity + timing of the match kernel variants (manual tool)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _oracle as O
import zlib_amd
e = zlib_amd.Engine(0)
data = O.corpus((4 << 20) + 12345, 4, 5)
want, _ = O.deflate_chunked(data, 1 << 20, 6, 0, 8)
got, _ = e.deflate_chunks(data, level=6, chunk_size=1 << 20)
print("walks", os.environ.get("ZAMD_WALKS"), "parity", got == want, "timing", [round(e.timing_ms(k)[0], 2) for k in range(7)], flush=True)
big = bytes(zlib_amd.corpus(512 << 20, 4, 5, piece_bytes=16 << 20, threads=8))
d_in = e.alloc(len(big)); d_in.upload(big)
cap = e.lib.zamd_deflate_bound(len(big), 1 << 20, 8)
d_out = e.alloc(cap)
for _ in range(2):
    tot = e.deflate_dev(d_in.ptr, len(big), d_out.ptr, cap)
print("  512MiB timing", [round(e.timing_ms(k)[0], 2) for k in range(7)], "out", tot, flush=True)
e.close()
