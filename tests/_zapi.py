"""ctypes view of zlib_amd/libzamd_z.so — the zlib 1.2.8 stream API of include/zamd_zlib.h —
written the way a ctypes-based zlib binding declares it."""
import ctypes as C
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "zlib_amd", "libzamd_z.so")

Z_NO_FLUSH, Z_PARTIAL_FLUSH, Z_SYNC_FLUSH, Z_FULL_FLUSH, Z_FINISH, Z_BLOCK = 0, 1, 2, 3, 4, 5
Z_OK, Z_STREAM_END, Z_NEED_DICT = 0, 1, 2
Z_STREAM_ERROR, Z_DATA_ERROR, Z_MEM_ERROR, Z_BUF_ERROR, Z_VERSION_ERROR = -2, -3, -4, -5, -6

ALLOC = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_uint, C.c_uint)
FREE = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p)
IN_FUNC = C.CFUNCTYPE(C.c_uint, C.c_void_p, C.POINTER(C.c_void_p))        # zlib.h:1003-1004
OUT_FUNC = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_uint)


class ZStream(C.Structure):        # zlib.h:85-104
    _fields_ = [("next_in", C.c_void_p), ("avail_in", C.c_uint), ("total_in", C.c_ulong),
                ("next_out", C.c_void_p), ("avail_out", C.c_uint), ("total_out", C.c_ulong),
                ("msg", C.c_char_p), ("state", C.c_void_p),
                ("zalloc", C.c_void_p), ("zfree", C.c_void_p), ("opaque", C.c_void_p),
                ("data_type", C.c_int), ("adler", C.c_ulong), ("reserved", C.c_ulong)]


class GzHeader(C.Structure):       # zlib.h:109-129
    _fields_ = [("text", C.c_int), ("time", C.c_ulong), ("xflags", C.c_int), ("os", C.c_int),
                ("extra", C.c_void_p), ("extra_len", C.c_uint), ("extra_max", C.c_uint),
                ("name", C.c_void_p), ("name_max", C.c_uint), ("comment", C.c_void_p), ("comm_max", C.c_uint),
                ("hcrc", C.c_int), ("done", C.c_int)]


_lib = None


class GzIndex(C.Structure):
    _fields_ = [("member_offset", C.c_ulonglong), ("n_chunks", C.c_uint), ("pairs", C.POINTER(C.c_uint)),
                ("window_offset", C.c_ulonglong), ("window_every", C.c_uint), ("n_windows", C.c_uint),
                ("splits", C.POINTER(C.c_ulonglong)), ("splits_per_chunk", C.c_uint)]


def lib():
    global _lib
    if _lib is None:
        l = C.CDLL(LIB)
        P = C.POINTER(ZStream)
        l.zlibVersion.restype = C.c_char_p
        l.zlibCompileFlags.restype = C.c_ulong
        l.zError.argtypes = [C.c_int]; l.zError.restype = C.c_char_p
        l.deflateInit_.argtypes = [P, C.c_int, C.c_char_p, C.c_int]
        l.deflateInit2_.argtypes = [P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_int]
        l.inflateInit_.argtypes = [P, C.c_char_p, C.c_int]
        l.inflateInit2_.argtypes = [P, C.c_int, C.c_char_p, C.c_int]
        for name in ("deflate", "inflate"):
            getattr(l, name).argtypes = [P, C.c_int]
        for name in ("deflateEnd", "deflateReset", "deflateResetKeep", "inflateEnd", "inflateReset", "inflateResetKeep",
                     "inflateSync", "inflateSyncPoint"):
            getattr(l, name).argtypes = [P]
        l.inflateReset2.argtypes = [P, C.c_int]
        l.deflateParams.argtypes = [P, C.c_int, C.c_int]
        l.deflateBound.argtypes = [P, C.c_ulong]; l.deflateBound.restype = C.c_ulong
        l.deflatePending.argtypes = [P, C.POINTER(C.c_uint), C.POINTER(C.c_int)]
        l.deflateSetHeader.argtypes = [P, C.POINTER(GzHeader)]
        l.inflateGetHeader.argtypes = [P, C.POINTER(GzHeader)]
        l.deflateCopy.argtypes = [P, P]; l.inflateCopy.argtypes = [P, P]
        l.deflateSetDictionary.argtypes = [P, C.c_char_p, C.c_uint]
        l.inflateSetDictionary.argtypes = [P, C.c_char_p, C.c_uint]
        l.inflateGetDictionary.argtypes = [P, C.c_void_p, C.POINTER(C.c_uint)]
        l.deflateTune.argtypes = [P, C.c_int, C.c_int, C.c_int, C.c_int]
        l.deflatePrime.argtypes = [P, C.c_int, C.c_int]; l.inflatePrime.argtypes = [P, C.c_int, C.c_int]
        l.inflateUndermine.argtypes = [P, C.c_int]
        l.inflateMark.argtypes = [P]; l.inflateMark.restype = C.c_long
        l.compress.argtypes = [C.c_void_p, C.POINTER(C.c_ulong), C.c_char_p, C.c_ulong]
        l.compress2.argtypes = [C.c_void_p, C.POINTER(C.c_ulong), C.c_char_p, C.c_ulong, C.c_int]
        l.uncompress.argtypes = [C.c_void_p, C.POINTER(C.c_ulong), C.c_char_p, C.c_ulong]
        l.compressBound.argtypes = [C.c_ulong]; l.compressBound.restype = C.c_ulong
        for name in ("adler32", "crc32"):
            getattr(l, name).argtypes = [C.c_ulong, C.c_char_p, C.c_uint]; getattr(l, name).restype = C.c_ulong
        for name in ("adler32_combine", "crc32_combine", "adler32_combine64", "crc32_combine64"):
            getattr(l, name).argtypes = [C.c_ulong, C.c_ulong, C.c_long]; getattr(l, name).restype = C.c_ulong
        l.get_crc_table.restype = C.POINTER(C.c_uint)
        l.inflateBackInit_.argtypes = [P, C.c_int, C.c_void_p, C.c_char_p, C.c_int]
        l.inflateBack.argtypes = [P, IN_FUNC, C.c_void_p, OUT_FUNC, C.c_void_p]
        l.inflateBackEnd.argtypes = [P]
        # gz* file layer (zlib.h:1224-1552)
        G = C.c_void_p
        for name in ("gzopen", "gzopen64"):
            getattr(l, name).argtypes = [C.c_char_p, C.c_char_p]; getattr(l, name).restype = G
        l.gzdopen.argtypes = [C.c_int, C.c_char_p]; l.gzdopen.restype = G
        l.gzbuffer.argtypes = [G, C.c_uint]
        l.gzsetparams.argtypes = [G, C.c_int, C.c_int]
        l.gzread.argtypes = [G, C.c_void_p, C.c_uint]
        l.gzwrite.argtypes = [G, C.c_char_p, C.c_uint]
        l.gzputs.argtypes = [G, C.c_char_p]
        l.gzgets.argtypes = [G, C.c_void_p, C.c_int]; l.gzgets.restype = C.c_void_p
        l.gzputc.argtypes = [G, C.c_int]
        l.gzgetc.argtypes = [G]; l.gzgetc_.argtypes = [G]
        l.gzungetc.argtypes = [C.c_int, G]
        l.gzflush.argtypes = [G, C.c_int]
        l.gzseek.argtypes = [G, C.c_long, C.c_int]; l.gzseek.restype = C.c_long
        l.gzseek64.argtypes = [G, C.c_longlong, C.c_int]; l.gzseek64.restype = C.c_longlong
        for name in ("gztell", "gzoffset"):
            getattr(l, name).argtypes = [G]; getattr(l, name).restype = C.c_long
        for name in ("gztell64", "gzoffset64"):
            getattr(l, name).argtypes = [G]; getattr(l, name).restype = C.c_longlong
        for name in ("gzrewind", "gzeof", "gzdirect", "gzclose", "gzclose_r", "gzclose_w"):
            getattr(l, name).argtypes = [G]
        l.gzerror.argtypes = [G, C.POINTER(C.c_int)]; l.gzerror.restype = C.c_char_p
        l.gzclearerr.argtypes = [G]; l.gzclearerr.restype = None
        l.gzprintf.restype = C.c_int
        l.zamd_deflate_index.argtypes = [P, C.c_int]
        l.zamd_deflate_index_get.argtypes = [P, C.POINTER(C.POINTER(C.c_uint)), C.POINTER(C.c_uint)]
        l.zamd_inflate_index_hint.argtypes = [P, C.POINTER(C.c_uint), C.c_uint]
        l.zamd_deflate_index_splits.argtypes = [P, C.POINTER(C.POINTER(C.c_ulonglong)), C.POINTER(C.c_uint)]
        l.zamd_inflate_index_hint_splits.argtypes = [P, C.POINTER(C.c_ulonglong), C.c_uint]
        l.zamd_gz_index_load.argtypes = [C.c_int, C.POINTER(GzIndex)]
        l.zamd_gz_index_free.argtypes = [C.POINTER(GzIndex)]; l.zamd_gz_index_free.restype = None
        l.zamd_gz_seek_stats.argtypes = [G, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)]
        _lib = l
    return _lib


VERSION = b"1.2.8"
SZ = C.sizeof(ZStream)


def compress2(data, level=-1, cap=None):
    l = lib()
    cap = l.compressBound(len(data)) if cap is None else cap
    buf = C.create_string_buffer(max(cap, 1))
    n = C.c_ulong(cap)
    rc = l.compress2(buf, C.byref(n), bytes(data), len(data), level)
    return rc, buf.raw[:n.value] if rc == 0 else b""


def uncompress(comp, cap):
    l = lib()
    buf = C.create_string_buffer(max(cap, 1))
    n = C.c_ulong(cap)
    rc = l.uncompress(buf, C.byref(n), bytes(comp), len(comp))
    return rc, buf.raw[:n.value] if rc == 0 else b""


def deflate_stream(data, level=6, wbits=15, mem=8, strategy=0, in_step=None, out_step=65536, flushes=None, header=None, index=None, splits=None):
    """Feeds `data` through deflate() in in_step slices with out_step output space per call.
    flushes: {offset: flush} applied when the input cursor reaches offset.  Returns (rc_list_tail, bytes, strm)."""
    l = lib()
    s = ZStream()
    rc = l.deflateInit2_(C.byref(s), level, 8, wbits, mem, strategy, VERSION, SZ)
    assert rc == 0, rc
    if header is not None:
        assert l.deflateSetHeader(C.byref(s), C.byref(header)) == 0
    if index is not None:                      # a list that receives (compressed bytes, inflated bytes) per chunk
        assert l.zamd_deflate_index(C.byref(s), 1) == 0
    src = C.create_string_buffer(bytes(data), max(len(data), 1))
    out = bytearray()
    obuf = C.create_string_buffer(out_step)
    pos, n = 0, len(data)
    in_step = in_step or max(n, 1)
    cuts = sorted((flushes or {}).items())
    while True:
        nxt = min(n, pos + in_step)
        fl = Z_FINISH if nxt == n else Z_NO_FLUSH
        for off, f in cuts:
            if pos < off <= nxt:
                nxt, fl = off, (f if off < n or f == Z_FINISH else f)
                break
        s.next_in = C.addressof(src) + pos
        s.avail_in = nxt - pos
        while True:
            s.next_out = C.addressof(obuf)
            s.avail_out = out_step
            rc = l.deflate(C.byref(s), fl)
            assert rc in (Z_OK, Z_STREAM_END, Z_BUF_ERROR), rc
            out += obuf.raw[:out_step - s.avail_out]
            if rc == Z_STREAM_END or (s.avail_out != 0 and s.avail_in == 0):
                break
        pos = nxt
        if rc == Z_STREAM_END:
            break
        if pos == n and fl != Z_FINISH:
            cuts = [(n, Z_FINISH)]
            in_step = 1
            # one more round with Z_FINISH and no input
            s.next_in = C.addressof(src) + n
            s.avail_in = 0
            while True:
                s.next_out = C.addressof(obuf); s.avail_out = out_step
                rc = l.deflate(C.byref(s), Z_FINISH)
                out += obuf.raw[:out_step - s.avail_out]
                if rc == Z_STREAM_END:
                    break
                assert rc == Z_OK, rc
            break
    totals = (s.total_in, s.total_out, s.adler)
    if index is not None:
        pairs, cnt = C.POINTER(C.c_uint)(), C.c_uint(0)
        rc = l.zamd_deflate_index_get(C.byref(s), C.byref(pairs), C.byref(cnt))
        index.append(rc)
        index.extend((pairs[2 * i], pairs[2 * i + 1]) for i in range(cnt.value))
        if splits is not None and rc == 0:     # a list that receives the records per chunk, then all the split records
            recs, per = C.POINTER(C.c_ulonglong)(), C.c_uint(0)
            assert l.zamd_deflate_index_splits(C.byref(s), C.byref(recs), C.byref(per)) == 0
            splits.append(per.value)
            splits.extend(recs[:per.value * cnt.value])
    assert l.deflateEnd(C.byref(s)) == 0
    return bytes(out), totals


def inflate_stream(comp, wbits=15, in_step=None, out_step=65536, flush=Z_NO_FLUSH, header=None, dictionary=None, hint=None, hint_splits=None):
    """Feeds `comp` through inflate().  Returns (rc, bytes, msg, unused_input, totals)."""
    l = lib()
    s = ZStream()
    rc = l.inflateInit2_(C.byref(s), wbits, VERSION, SZ)
    assert rc == 0, rc
    if header is not None:
        assert l.inflateGetHeader(C.byref(s), C.byref(header)) == 0
    if hint is not None:                       # [(compressed bytes, inflated bytes), ...] of the deflate data
        flat = (C.c_uint * (2 * len(hint)))(*[v for pair in hint for v in pair])
        assert l.zamd_inflate_index_hint(C.byref(s), flat, len(hint)) == 0
        if hint_splits:                        # [records per chunk, record, record, ...] as deflate_stream(splits=) filled it
            recs = (C.c_ulonglong * (len(hint_splits) - 1))(*hint_splits[1:])
            assert l.zamd_inflate_index_hint_splits(C.byref(s), recs, hint_splits[0]) == 0
    src = C.create_string_buffer(bytes(comp), max(len(comp), 1))
    obuf = C.create_string_buffer(out_step)
    out = bytearray()
    pos, n = 0, len(comp)
    in_step = in_step or max(n, 1)
    rc = Z_OK
    idle = 0
    while True:
        nxt = min(n, pos + in_step)
        s.next_in = C.addressof(src) + pos
        s.avail_in = nxt - pos
        while True:
            s.next_out = C.addressof(obuf)
            s.avail_out = out_step
            rc = l.inflate(C.byref(s), flush)
            if rc == Z_NEED_DICT and dictionary is not None:
                assert l.inflateSetDictionary(C.byref(s), dictionary, len(dictionary)) == 0
                continue
            got = out_step - s.avail_out
            out += obuf.raw[:got]
            if rc not in (Z_OK, Z_BUF_ERROR) or (s.avail_out != 0 and s.avail_in == 0) or (rc == Z_BUF_ERROR and got == 0):
                break
        pos = nxt - s.avail_in
        if rc not in (Z_OK, Z_BUF_ERROR):
            break
        if nxt == n:
            idle += 1
            if idle > 2:
                break
    msg = s.msg.decode() if s.msg else None
    totals = (s.total_in, s.total_out, s.adler)
    assert l.inflateEnd(C.byref(s)) == 0
    return rc, bytes(out), msg, n - pos, totals


REF_LIB = os.path.join(ROOT, "oracle", "_ref", "libzref.so")


def ref_lib():
    """The unmodified reference build (oracle/_ref/libzref.so), or None where it was not built.  A checker only."""
    if not os.path.exists(REF_LIB):
        return None
    # Python's own zlib module has the system libz in the process: the reference build must bind its internal calls to itself
    l = C.CDLL(REF_LIB, mode=os.RTLD_LOCAL | os.RTLD_DEEPBIND)
    P = C.POINTER(ZStream)
    l.inflateInit2_.argtypes = [P, C.c_int, C.c_char_p, C.c_int]
    l.inflate.argtypes = [P, C.c_int]
    l.inflateEnd.argtypes = [P]
    return l


def inf_loop(l, hexs, step, win, length, pieces=None):
    """The loop of the reference's inf() harness (test/infcover.c:284-347): the input goes in `step` bytes at a time
    (0 = all at once; `pieces` = explicit piece sizes instead), every inflate() call gets a fresh output buffer of
    `length` bytes.  Returns (return codes of all calls, output of all calls, last msg), or the init error."""
    data = bytes.fromhex("".join("%02x" % int(t, 16) for t in hexs.split())) if isinstance(hexs, str) else bytes(hexs)
    s = ZStream()
    rc = l.inflateInit2_(C.byref(s), win, VERSION, SZ)
    if rc != Z_OK:
        return [rc], b"", None
    src = C.create_string_buffer(data, max(len(data), 1))
    obuf = C.create_string_buffer(max(length, 1))
    have = len(data)
    sizes = list(pieces) if pieces else None
    if step == 0 or step > have:
        step = have
    nxt = sizes.pop(0) if sizes else step
    s.next_in = C.addressof(src)
    s.avail_in = nxt
    have -= nxt
    rcs, out = [], bytearray()
    while True:
        s.avail_out = length
        s.next_out = C.addressof(obuf)
        rc = l.inflate(C.byref(s), Z_NO_FLUSH)
        rcs.append(rc)
        out += obuf.raw[:length - s.avail_out]
        if rc not in (Z_OK, Z_BUF_ERROR):
            break
        have += s.avail_in
        nxt = sizes.pop(0) if sizes else step
        s.avail_in = min(nxt, have)
        have -= s.avail_in
        if s.avail_in == 0:
            break
        if len(rcs) > 10000:
            break
    msg = s.msg.decode() if s.msg else None
    l.inflateEnd(C.byref(s))
    return rcs, bytes(out), msg
