"""ctypes access to the CPU oracle (oracle/liboracle.so) — test infrastructure only."""
import ctypes as C
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ORACLE_DIR, "liboracle.so")


class Sym(C.Structure):
    _fields_ = [("pos", C.c_uint32), ("dist", C.c_uint16), ("lc", C.c_uint16)]


class Block(C.Structure):
    _fields_ = [("sym_start", C.c_uint32), ("sym_count", C.c_uint32), ("in_start", C.c_uint32), ("in_len", C.c_uint32),
                ("type", C.c_uint32), ("last", C.c_uint32), ("can_store", C.c_uint32),
                ("opt_len", C.c_uint64), ("static_len", C.c_uint64), ("bit_start", C.c_uint64), ("bit_end", C.c_uint64)]


class Trace(C.Structure):
    _fields_ = [("syms", C.POINTER(Sym)), ("n_syms", C.c_size_t), ("cap_syms", C.c_size_t),
                ("blocks", C.POINTER(Block)), ("n_blocks", C.c_size_t), ("cap_blocks", C.c_size_t)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            subprocess.check_call(["make", "-C", ORACLE_DIR, "liboracle.so"])
        l = C.CDLL(LIB)
        l.zo_deflate_chunk.argtypes = [C.c_char_p, C.c_size_t, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                       C.c_void_p, C.c_size_t, C.POINTER(Trace)]
        l.zo_deflate_chunk.restype = C.c_size_t
        l.zo_deflate_chunked.argtypes = [C.c_char_p, C.c_size_t, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t,
                                         C.POINTER(C.c_uint32), C.c_size_t]
        l.zo_deflate_chunked.restype = C.c_size_t
        l.zo_deflate_chunk_tuned.argtypes = [C.c_char_p, C.c_size_t, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                             C.POINTER(C.c_int), C.c_void_p, C.c_size_t, C.POINTER(Trace)]
        l.zo_deflate_chunk_tuned.restype = C.c_size_t
        l.zo_deflate_chunk_primed.argtypes = [C.c_char_p, C.c_size_t, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                              C.POINTER(C.c_int), C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(Trace)]
        l.zo_deflate_chunk_primed.restype = C.c_size_t
        l.zo_deflate_chunk_ex.argtypes = [C.c_char_p, C.c_size_t, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                          C.POINTER(C.c_int), C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(Trace),
                                          C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        l.zo_deflate_chunk_ex.restype = C.c_size_t
        l.zo_deflate_chunked_w.argtypes = [C.c_char_p, C.c_size_t, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t,
                                           C.POINTER(C.c_uint32), C.c_size_t]
        l.zo_deflate_chunked_w.restype = C.c_size_t
        l.zo_trace_free.argtypes = [C.POINTER(Trace)]
        l.zo_adler32.argtypes = [C.c_uint32, C.c_char_p, C.c_size_t]; l.zo_adler32.restype = C.c_uint32
        l.zo_crc32.argtypes = [C.c_uint32, C.c_char_p, C.c_size_t]; l.zo_crc32.restype = C.c_uint32
        l.zo_adler32_combine.argtypes = [C.c_uint32, C.c_uint32, C.c_int64]; l.zo_adler32_combine.restype = C.c_uint32
        l.zo_crc32_combine.argtypes = [C.c_uint32, C.c_uint32, C.c_int64]; l.zo_crc32_combine.restype = C.c_uint32
        l.zo_inflate_raw.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                     C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.POINTER(C.c_char_p), C.c_int]
        l.zo_inflate_raw.restype = C.c_int
        l.zo_inflate_raw_at.argtypes = [C.c_char_p, C.c_size_t, C.c_uint64, C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                        C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.POINTER(C.c_char_p), C.c_int]
        l.zo_inflate_raw_at.restype = C.c_int
        l.zo_corpus_fill.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_uint64]
        _lib = l
    return _lib


def corpus(n, kind, seed):
    buf = C.create_string_buffer(n + 1)
    lib().zo_corpus_fill(buf, n, kind, seed)
    return buf.raw[:n]


FLUSH_SYNC, FLUSH_FINISH, FLUSH_PARTIAL, FLUSH_BLOCK = 0, 1, 2, 3


def deflate_chunk(buf, dict_len, n, level=6, strategy=0, mem_level=8, final=True, want_trace=False, tune=None, prime=(0, 0),
                  wbits=15, flush=None, want_state=False):
    """buf = dictionary + chunk bytes.  Returns bytes, or (bytes, syms, blocks) with want_trace, or
    (bytes, pending bit count, pending bit value, data_type) with want_state.  flush (FLUSH_*) overrides final."""
    cap = n + n // 4 + 65536
    out = C.create_string_buffer(cap)
    tr = Trace()
    tv = (C.c_int * 4)(*tune) if tune else None
    fl = flush if flush is not None else (FLUSH_FINISH if final else FLUSH_SYNC)
    pb, pv, dt = C.c_int(0), C.c_int(0), C.c_int(2)
    r = lib().zo_deflate_chunk_ex(bytes(buf), dict_len, n, level, strategy, mem_level, wbits, fl, tv, prime[0], prime[1],
                                  out, cap, C.byref(tr) if want_trace else None, C.byref(pb), C.byref(pv), C.byref(dt))
    if r == C.c_size_t(-1).value:
        raise RuntimeError("oracle deflate failed")
    data = out.raw[:r]
    if want_state:
        return data, pb.value, pv.value, dt.value
    if not want_trace:
        return data
    syms = [(tr.syms[i].pos, tr.syms[i].dist, tr.syms[i].lc) for i in range(tr.n_syms)]
    blocks = [{f[0]: getattr(tr.blocks[i], f[0]) for f in Block._fields_} for i in range(tr.n_blocks)]
    lib().zo_trace_free(C.byref(tr))
    return data, syms, blocks


def deflate_chunked(data, chunk_size=1 << 20, level=6, strategy=0, mem_level=8, wbits=15):
    n = len(data)
    cap = n + n // 4 + 65536 + (n // chunk_size + 1) * 1024
    out = C.create_string_buffer(cap)
    maxc = n // chunk_size + 2
    lens = (C.c_uint32 * maxc)()
    r = lib().zo_deflate_chunked_w(bytes(data), n, chunk_size, level, strategy, mem_level, wbits, out, cap, lens, maxc)
    if r == C.c_size_t(-1).value:
        raise RuntimeError("oracle deflate failed")
    nch = max(1, (n + chunk_size - 1) // chunk_size)
    return out.raw[:r], list(lens[:nch])


def adler32(data, value=1):
    return lib().zo_adler32(value, bytes(data), len(data))


def crc32(data, value=0):
    return lib().zo_crc32(value, bytes(data), len(data))


def inflate_raw_at(comp, start_bit, dict_bytes=b"", out_cap=0, stop_at_sync=True):
    """Serial decode from a block that starts at bit start_bit of comp, with dict_bytes as the window in front of it."""
    out = C.create_string_buffer(out_cap + 1)
    used, made, msg = C.c_size_t(0), C.c_size_t(0), C.c_char_p()
    rc = lib().zo_inflate_raw_at(bytes(comp), len(comp), start_bit, bytes(dict_bytes), len(dict_bytes), out, out_cap,
                                 C.byref(used), C.byref(made), C.byref(msg), int(stop_at_sync))
    return rc, out.raw[:made.value], used.value, (msg.value.decode() if msg.value else None)


def inflate_raw(comp, dict_bytes=b"", out_cap=0, stop_at_sync=False):
    out = C.create_string_buffer(out_cap + 1)
    used, made, msg = C.c_size_t(0), C.c_size_t(0), C.c_char_p()
    rc = lib().zo_inflate_raw(bytes(comp), len(comp), bytes(dict_bytes), len(dict_bytes), out, out_cap,
                              C.byref(used), C.byref(made), C.byref(msg), int(stop_at_sync))
    return rc, out.raw[:made.value], used.value, (msg.value.decode() if msg.value else None)
