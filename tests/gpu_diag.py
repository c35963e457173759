"""Manual GPU diagnostic: runs cases through the HIP engine, compares with the
oracle and, on a mismatch, prints where the pipelines diverge (block table,
symbol stream, match records).  usage: python tests/gpu_diag.py [quick|full]"""
import struct
import sys
import time
import os

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _oracle as O
import zlib_amd


def diagnose(e, data, chunk_size, level, strategy, mem, got, want_chunks):
    n = len(data)
    off = 0
    for ci, wc in enumerate(want_chunks):
        cs = ci * chunk_size
        ln = min(chunk_size, n - cs)
        dict_len = min(32768, cs)
        final = cs + ln >= n
        buf = data[cs - dict_len: cs + ln]
        wbytes, wsyms, wblocks = O.deflate_chunk(buf, dict_len, ln, level, strategy, mem, final, want_trace=True)
        gsyms = list(memoryview(e.debug_dump(2, ci, ln + 1, 4)).cast("I"))
        graw = e.debug_dump(3, ci, 4096, 40)
        gblocks = [struct.unpack_from("<8IQ", graw, 40 * i) for i in range(len(graw) // 40)]
        print(f" chunk {ci}: oracle {len(wsyms)} syms {len(wblocks)} blocks; gpu {len(gsyms)} syms {len(gblocks)} blocks")
        bad = False
        for i, (g, w) in enumerate(zip(gblocks, wblocks)):
            gt = (g[0], g[1], g[2], g[3], g[5], g[4] & 1, (g[4] >> 1) & 1)
            wt = (w["sym_start"], w["sym_count"], w["in_start"], w["in_len"], w["type"], w["last"], w["can_store"])
            wbits = w["bit_end"] - w["bit_start"]
            if gt != wt or (g[5] != 0 and g[6] != wbits - (7 & -w["bit_end"] if False else 0) and not w["last"] and g[6] != wbits):
                print(f"  block {i}: gpu (sym_start,count,in_start,in_len,type,last,can_store)={gt} bits={g[6]} hdr={g[7]} off={g[8]}; oracle {wt} bits={wbits} start={w['bit_start']}")
                bad = True
                if i > 3:
                    break
        ws = [(p if d == 0 else (0x80000000 | (lc << 16) | d)) for (p, d, lc) in wsyms]
        if not any(b[4] & 4 for b in gblocks):
            for i, (g, w) in enumerate(zip(gsyms, ws)):
                if g != w:
                    print(f"  first differing symbol {i}: gpu {g:#x} oracle {w:#x} (oracle pos {wsyms[i][0]})")
                    p = wsyms[i][0] - dict_len
                    lo = max(0, p - 3)
                    rec = list(memoryview(e.debug_dump(1, ci, ln, 8)).cast("I"))
                    pd = list(memoryview(e.debug_dump(0, ci, ln, 2)).cast("H"))
                    for q in range(lo, min(ln, p + 4)):
                        print(f"    pos {q}: rec full len={rec[2*q] & 0xffff} dist={rec[2*q] >> 16}  quarter len={rec[2*q+1] & 0xffff} dist={rec[2*q+1] >> 16}  link={pd[q]}")
                    bad = True
                    break
        if bad:
            return
        off += wc


def run_case(e, name, data, chunk_size=1 << 20, level=6, strategy=0, mem=8):
    t0 = time.time()
    got, offs = e.deflate_chunks(data, level=level, strategy=strategy, mem_level=mem, chunk_size=chunk_size)
    t1 = time.time()
    want, lens = O.deflate_chunked(data, chunk_size, level, strategy, mem)
    ok = got == want
    glens = [offs[i + 1] - offs[i] for i in range(len(offs) - 1)]
    tm = [e.timing_ms(k) for k in range(7)]
    print(f"{'OK ' if ok else 'BAD'} {name}: n={len(data)} chunk={chunk_size} L{level} S{strategy} M{mem} out={len(got)} want={len(want)} "
          f"gpu_call={1e3*(t1-t0):.1f}ms kernels(ms)=" + " ".join(f"{m:.2f}" for m, _ in tm), flush=True)
    if not ok:
        d = next((i for i in range(min(len(got), len(want))) if got[i] != want[i]), min(len(got), len(want)))
        print(f"  first differing byte {d}; gpu chunk lens {glens[:8]} oracle {lens[:8]}")
        diagnose(e, data, chunk_size, level, strategy, mem, got, lens)
    return ok


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "quick"
    e = zlib_amd.Engine(0)
    print(zlib_amd.load_library().zamd_version().decode(), flush=True)
    ok = True
    ok &= run_case(e, "tiny-text", O.corpus(100000, 0, 1), 32768)
    ok &= run_case(e, "empty", b"", 32768)
    ok &= run_case(e, "3bytes", b"abc", 32768)
    ok &= run_case(e, "a40", b"a" * 40, 32768)
    ok &= run_case(e, "mixed-1M", O.corpus(1 << 20, 4, 5), 1 << 20)
    ok &= run_case(e, "mixed-4M+", O.corpus((4 << 20) + 12345, 4, 5), 1 << 20)
    for strat in (1, 2, 3, 4):
        ok &= run_case(e, f"image-S{strat}", O.corpus((2 << 20) + 777, 3, 4), 1 << 20, 6, strat)
    for lvl in (4, 5, 7, 8, 9):
        ok &= run_case(e, f"text-L{lvl}", O.corpus((1 << 20) + 4000, 0, 1), 1 << 19, lvl)
    for mem in (1, 5, 9):
        ok &= run_case(e, f"mixed-M{mem}", O.corpus((1 << 20) + 99, 4, 7), 1 << 18, 6, 0, mem)
    ok &= run_case(e, "tail-corner", O.corpus(65374, 0, 3), 65536)
    if mode == "full":
        for kind in range(5):
            ok &= run_case(e, f"kind{kind}-16M", O.corpus(16 << 20, kind, kind + 1), 1 << 20)
    # checksums
    import zlib
    for n in (0, 1, 100, 5552, 65521, 1 << 20, (3 << 20) + 17):
        d = O.corpus(n, 4, 11)
        a, c = e.adler32(d), e.crc32(d)
        good = a == zlib.adler32(d) and c == zlib.crc32(d)
        print(f"{'OK ' if good else 'BAD'} checksums n={n}: adler {a:08x}/{zlib.adler32(d):08x} crc {c:08x}/{zlib.crc32(d):08x}", flush=True)
        ok &= good
    print("ALL OK" if ok else "FAILURES", flush=True)
    e.close()
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
