"""zlib_amd — MI355X-native chunked DEFLATE behind the zlib 1.2.8 API.

Python side of the C ABI in ``include/zamd.h`` (ctypes, no torch types cross the
boundary).  The compute lives in ``zlib_amd/libzamd.so`` (hand-written HIP for
gfx950); this module only loads it and mirrors the call shapes of the reference
(``deflateInit2``/``deflate`` chunk recipe, ``adler32``/``crc32`` with
``*_combine``).  There is no CPU fallback: if the library or a GPU is missing the
calls raise.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Tuple

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libzamd.so")

Z_OK, Z_STREAM_ERROR, Z_DATA_ERROR, Z_MEM_ERROR, Z_BUF_ERROR = 0, -2, -3, -4, -5
Z_DEFAULT_STRATEGY, Z_FILTERED, Z_HUFFMAN_ONLY, Z_RLE, Z_FIXED = 0, 1, 2, 3, 4

_lib = None


class ZamdError(RuntimeError):
    def __init__(self, code: int, what: str, msg: str = "", segment: int = -1):
        super().__init__(f"{what} failed with code {code}" + (f": {msg} (segment {segment})" if msg else ""))
        self.code, self.msg, self.segment = code, msg, segment


INF_MARKS = 16      # ZAMD_INF_MARKS: entries in a segment's row of block marks


class InflateSeg(C.Structure):
    """zamd_inflate_seg of include/zamd.h."""
    _fields_ = [("in_bit", C.c_uint64), ("in_end_bit", C.c_uint64), ("out_off", C.c_uint64), ("out_cap", C.c_uint64),
                ("win_avail", C.c_uint32), ("mode", C.c_uint32), ("out_len", C.c_uint64), ("end_bit", C.c_uint64),
                ("status", C.c_uint32), ("final_seen", C.c_uint32), ("stop_bit", C.c_uint64)]


INF_SYNCSTOP, INF_EXACT, INF_BLOCKSTOP = 1, 2, 4


def load_library() -> C.CDLL:
    """Loads libzamd.so (built in-tree by ``make`` / ``__graft_entry__.build()``)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build it with `make zlib_amd/libzamd.so` (hipcc, gfx950)")
    lib = C.CDLL(LIB_PATH, mode=getattr(os, "RTLD_LOCAL", 0))
    u64, u32, i32, vp = C.c_uint64, C.c_uint32, C.c_int, C.c_void_p
    lib.zamd_engine_create.argtypes = [C.POINTER(vp), i32, u64]
    lib.zamd_engine_create.restype = i32
    lib.zamd_engine_destroy.argtypes = [vp]
    lib.zamd_engine_destroy.restype = None
    lib.zamd_deflate_bound.argtypes = [u64, u32, i32]
    lib.zamd_deflate_bound.restype = u64
    lib.zamd_deflate_dev.argtypes = [vp, vp, u64, u32, i32, i32, i32, i32, vp, u64, vp, C.POINTER(u64)]
    lib.zamd_deflate_dev.restype = i32
    lib.zamd_deflate_shard_dev.argtypes = [vp, vp, u64, u32, u32, i32, i32, i32, i32, vp, u64, vp, C.POINTER(u64)]
    lib.zamd_deflate_shard_dev.restype = i32
    lib.zamd_chunk_offsets_dev.argtypes = [vp, vp, u64, u64, vp]
    lib.zamd_chunk_offsets_dev.restype = i32
    lib.zamd_engine_tune.argtypes = [vp, i32, i32, i32, i32]
    lib.zamd_engine_tune.restype = i32
    lib.zamd_find_sync_dev.argtypes = [vp, vp, u64, u64, C.POINTER(u64), C.c_uint32, C.POINTER(C.c_uint32)]
    lib.zamd_find_sync_dev.restype = i32
    lib.zamd_find_blocks_dev.argtypes = [vp, vp, u64, C.POINTER(u64), C.c_uint32, u64, C.POINTER(u64)]
    lib.zamd_find_blocks_dev.restype = i32
    lib.zamd_engine_window_bits.argtypes = [vp, i32]
    lib.zamd_engine_window_bits.restype = i32
    lib.zamd_engine_tail.argtypes = [vp, i32]
    lib.zamd_engine_tail.restype = i32
    lib.zamd_engine_pending_bits.argtypes = [vp, C.POINTER(i32), C.POINTER(i32)]
    lib.zamd_engine_pending_bits.restype = i32
    lib.zamd_engine_data_type.argtypes = [vp]
    lib.zamd_engine_data_type.restype = i32
    lib.zamd_engine_prime.argtypes = [vp, i32, i32]
    lib.zamd_engine_prime.restype = i32
    lib.zamd_checksums_dev.argtypes = [vp, vp, u64, u32, vp, vp]
    lib.zamd_checksums_dev.restype = i32
    lib.zamd_adler32_dev.argtypes = [vp, u32, vp, u64, C.POINTER(u32)]
    lib.zamd_adler32_dev.restype = i32
    lib.zamd_crc32_dev.argtypes = [vp, u32, vp, u64, C.POINTER(u32)]
    lib.zamd_crc32_dev.restype = i32
    lib.zamd_checksums_noalloc_dev.argtypes = [vp, u32, u32, vp, u64, vp]
    lib.zamd_checksums_noalloc_dev.restype = i32
    lib.zamd_adler32_combine.argtypes = [u32, u32, C.c_int64]
    lib.zamd_adler32_combine.restype = u32
    lib.zamd_crc32_combine.argtypes = [u32, u32, C.c_int64]
    lib.zamd_crc32_combine.restype = u32
    lib.zamd_inflate_dev.argtypes = [vp, vp, vp, u64, u32, vp, u64, C.POINTER(C.c_int64)]
    lib.zamd_inflate_dev.restype = i32
    lib.zamd_inflate_shard_dev.argtypes = [vp, vp, u64, vp, u64, u32, vp, vp, u64, C.POINTER(C.c_int64)]
    lib.zamd_inflate_shard_dev.restype = i32
    lib.zamd_inflate_split_dev.argtypes = [vp, vp, u64, vp, u64, u32, vp, u32, vp, vp, u64, C.POINTER(C.c_int64)]
    lib.zamd_inflate_split_dev.restype = i32
    lib.zamd_engine_splits.argtypes = [vp, vp, u32]
    lib.zamd_engine_splits.restype = i32
    lib.zamd_inflate_shard_begin_dev.argtypes = [vp, vp, u64, vp, u64, u32, i32, vp, u64, C.POINTER(C.c_int64)]
    lib.zamd_inflate_shard_begin_dev.restype = i32
    lib.zamd_inflate_shard_tail_dev.argtypes = [vp, vp, vp]
    lib.zamd_inflate_shard_tail_dev.restype = i32
    lib.zamd_inflate_shard_finish_dev.argtypes = [vp, vp]
    lib.zamd_inflate_shard_finish_dev.restype = i32
    lib.zamd_engine_set_streams.argtypes = [vp, i32, i32]
    lib.zamd_engine_set_streams.restype = i32
    lib.zamd_inflate_segments_dev.argtypes = [vp, vp, u64, C.POINTER(InflateSeg), u64, vp, vp, u64]
    lib.zamd_inflate_segments_dev.restype = i32
    lib.zamd_inflate_block_marks.argtypes = [vp, C.POINTER(C.c_ulonglong), C.c_uint32]
    lib.zamd_inflate_block_marks.restype = i32
    lib.zamd_inflate_message.argtypes = [u32]
    lib.zamd_inflate_message.restype = C.c_char_p
    lib.zamd_inflate_last_error.argtypes = [vp, C.POINTER(C.c_int64), C.POINTER(u32)]
    lib.zamd_inflate_last_error.restype = i32
    lib.zamd_last_timing_ms.argtypes = [vp, i32, C.POINTER(i32)]
    lib.zamd_last_timing_ms.restype = C.c_double
    lib.zamd_corpus_fill.argtypes = [vp, u64, i32, u64]
    lib.zamd_corpus_fill.restype = None
    lib.zamd_corpus_fill_pieces.argtypes = [vp, u64, i32, u64, u64, i32]
    lib.zamd_corpus_fill_pieces.restype = None
    lib.zamd_debug_dump.argtypes = [vp, i32, u32, vp, u64]
    lib.zamd_debug_dump.restype = C.c_int64
    for name in ("zamd_dev_alloc",):
        getattr(lib, name).argtypes = [vp, u64, C.POINTER(vp)]
        getattr(lib, name).restype = i32
    lib.zamd_dev_free.argtypes = [vp, vp]
    lib.zamd_dev_free.restype = i32
    lib.zamd_dev_upload.argtypes = [vp, vp, vp, u64]
    lib.zamd_dev_upload.restype = i32
    lib.zamd_dev_download.argtypes = [vp, vp, vp, u64]
    lib.zamd_dev_download.restype = i32
    lib.zamd_dev_memset.argtypes = [vp, vp, i32, u64]
    lib.zamd_dev_memset.restype = i32
    lib.zamd_version.argtypes = []
    lib.zamd_version.restype = C.c_char_p
    lib.zamd_chunk_lengths_dev.argtypes = [vp, vp, u64, vp, u64]
    lib.zamd_chunk_lengths_dev.restype = i32
    lib.zamd_multi_create.argtypes = [C.POINTER(vp), C.POINTER(i32), i32, u64]
    lib.zamd_multi_create.restype = i32
    lib.zamd_multi_destroy.argtypes = [vp]
    lib.zamd_multi_destroy.restype = None
    lib.zamd_multi_devices.argtypes = [vp]
    lib.zamd_multi_devices.restype = i32
    lib.zamd_multi_engine.argtypes = [vp, i32]
    lib.zamd_multi_engine.restype = vp
    lib.zamd_multi_uses_rccl.argtypes = [vp]
    lib.zamd_multi_uses_rccl.restype = i32
    lib.zamd_multi_partition.argtypes = [u64, i32, i32, C.POINTER(u64), C.POINTER(u64), C.POINTER(u64)]
    lib.zamd_multi_partition.restype = i32
    lib.zamd_multi_offsets_host.argtypes = [C.POINTER(u64), u64, u64, C.POINTER(u64)]
    lib.zamd_multi_offsets_host.restype = i32
    lib.zamd_deflate_multi.argtypes = [vp, vp, u64, vp, u32, u32, i32, i32, i32, i32, vp, u64, C.POINTER(u64), C.POINTER(u64),
                                       C.POINTER(u32), C.POINTER(u32)]
    lib.zamd_deflate_multi.restype = i32
    _lib = lib
    return lib


def _chk(code: int, what: str) -> None:
    if code != 0:
        raise ZamdError(code, what)


def corpus(n: int, kind: int, seed: int, piece_bytes: int = 0, threads: int = 1) -> bytearray:
    """Synthetic corpus (kinds 0 text, 1 binary, 2 lowent, 3 image, 4 mixed)."""
    lib = load_library()
    buf = bytearray(n)
    if n:
        ptr = (C.c_char * n).from_buffer(buf)
        if piece_bytes:
            lib.zamd_corpus_fill_pieces(C.addressof(ptr), n, kind, seed, piece_bytes, threads)
        else:
            lib.zamd_corpus_fill(C.addressof(ptr), n, kind, seed)
    return buf


def adler32_combine(a1: int, a2: int, len2: int) -> int:
    return load_library().zamd_adler32_combine(a1, a2, len2)


def crc32_combine(c1: int, c2: int, len2: int) -> int:
    return load_library().zamd_crc32_combine(c1, c2, len2)


class DeviceBuffer:
    """hipMalloc'ed bytes on the engine's GPU."""

    def __init__(self, engine: "Engine", nbytes: int):
        self.engine, self.nbytes = engine, nbytes
        p = C.c_void_p()
        _chk(engine.lib.zamd_dev_alloc(engine.handle, nbytes, C.byref(p)), "zamd_dev_alloc")
        self.ptr = p.value

    def upload(self, data, offset: int = 0) -> None:
        mv = memoryview(data).cast("B")
        n = mv.nbytes
        if n == 0:
            return
        if mv.readonly:
            src = C.c_char_p(bytes(mv))
            _chk(self.engine.lib.zamd_dev_upload(self.engine.handle, self.ptr + offset, src, n), "zamd_dev_upload")
        else:
            arr = (C.c_char * n).from_buffer(mv)
            _chk(self.engine.lib.zamd_dev_upload(self.engine.handle, self.ptr + offset, C.addressof(arr), n), "zamd_dev_upload")

    def download(self, nbytes: Optional[int] = None, offset: int = 0) -> bytearray:
        n = self.nbytes - offset if nbytes is None else nbytes
        out = bytearray(n)
        if n:
            arr = (C.c_char * n).from_buffer(out)
            _chk(self.engine.lib.zamd_dev_download(self.engine.handle, C.addressof(arr), self.ptr + offset, n), "zamd_dev_download")
        return out

    def free(self) -> None:
        if self.ptr:
            self.engine.lib.zamd_dev_free(self.engine.handle, self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def multi_partition(total_chunks: int, n_devices: int, index: int) -> Tuple[int, int, int]:
    """(first chunk, chunk count, chunks per device) of device `index` (zamd_multi_partition; needs no GPU)."""
    lib = load_library()
    first, count, per = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
    _chk(lib.zamd_multi_partition(total_chunks, n_devices, index, C.byref(first), C.byref(count), C.byref(per)), "zamd_multi_partition")
    return first.value, count.value, per.value


def multi_offsets_host(lengths, base: int = 0) -> List[int]:
    """Exclusive scan of chunk lengths on the host (zamd_multi_offsets_host; needs no GPU): len(lengths) + 1 offsets."""
    lib = load_library()
    n = len(lengths)
    src = (C.c_uint64 * max(n, 1))(*lengths)
    dst = (C.c_uint64 * (n + 1))()
    _chk(lib.zamd_multi_offsets_host(src, n, base, dst), "zamd_multi_offsets_host")
    return list(dst)


class MultiEngine:
    """One deflate call on several GPUs of this node (zamd_multi_create): one engine per entry of `devices`."""

    def __init__(self, devices, workspace_bytes: int = 0):
        self.lib = load_library()
        h = C.c_void_p()
        arr = (C.c_int32 * len(devices))(*devices)
        _chk(self.lib.zamd_multi_create(C.byref(h), arr, len(devices), workspace_bytes), "zamd_multi_create")
        self.handle = h
        self.devices = list(devices)

    def close(self) -> None:
        if self.handle:
            self.lib.zamd_multi_destroy(self.handle)
            self.handle = None

    @property
    def uses_rccl(self) -> bool:
        return bool(self.lib.zamd_multi_uses_rccl(self.handle))

    def deflate(self, data, *, level: int = 6, strategy: int = 0, mem_level: int = 8, chunk_size: int = 1 << 20,
                finish: bool = True, dictionary: bytes = b"", adler: int = 1, crc: int = 0):
        """Host bytes in, (stream, chunk offsets + total, adler32, crc32) out."""
        n = len(data)
        nchunks = max(1, (n + chunk_size - 1) // chunk_size)
        cap = self.lib.zamd_deflate_bound(n, chunk_size, mem_level) + 64
        out = C.create_string_buffer(cap)
        offs = (C.c_uint64 * (nchunks + 1))()
        total, a, c = C.c_uint64(0), C.c_uint32(adler), C.c_uint32(crc)
        src = C.create_string_buffer(bytes(data), max(n, 1))
        _chk(self.lib.zamd_deflate_multi(self.handle, src, n, bytes(dictionary) if dictionary else None, len(dictionary), chunk_size,
                                         level, strategy, mem_level, int(finish), out, cap, offs, C.byref(total), C.byref(a), C.byref(c)),
             "zamd_deflate_multi")
        return out.raw[:total.value], list(offs), a.value, c.value


class Engine:
    """One engine per process and GPU (zamd_engine_create)."""

    def __init__(self, device: int = 0, workspace_bytes: int = 0):
        self.lib = load_library()
        h = C.c_void_p()
        _chk(self.lib.zamd_engine_create(C.byref(h), device, workspace_bytes), "zamd_engine_create")
        self.handle = h
        self.device = device

    def close(self) -> None:
        if self.handle:
            self.lib.zamd_engine_destroy(self.handle)
            self.handle = None

    def tune(self, good_length: int = 0, max_lazy: int = 0, nice_length: int = 0, max_chain: int = 0) -> None:
        """deflateTune for the following deflate calls; no arguments = back to the level's table."""
        _chk(self.lib.zamd_engine_tune(self.handle, good_length, max_lazy, nice_length, max_chain), "zamd_engine_tune")

    def prime(self, bits: int, value: int) -> None:
        """deflatePrime for the next deflate call: `bits` (0..7) bits of `value` in front of its first block."""
        _chk(self.lib.zamd_engine_prime(self.handle, bits, value), "zamd_engine_prime")

    def window_bits(self, bits: int = 15) -> None:
        """windowBits (9..15) of the following deflate calls."""
        _chk(self.lib.zamd_engine_window_bits(self.handle, bits), "zamd_engine_window_bits")

    def tail(self, mode: int) -> None:
        """How the next deflate call ends when finish is false: 0 sync marker, 1 _tr_align (Z_PARTIAL_FLUSH), 2 nothing (Z_BLOCK)."""
        _chk(self.lib.zamd_engine_tail(self.handle, mode), "zamd_engine_tail")

    def pending_bits(self) -> Tuple[int, int]:
        """(count, value) of the bits the last deflate call left behind its last whole byte."""
        b, v = C.c_int(0), C.c_int(0)
        _chk(self.lib.zamd_engine_pending_bits(self.handle, C.byref(b), C.byref(v)), "zamd_engine_pending_bits")
        return b.value, v.value

    def data_type(self) -> int:
        """strm->data_type after the last deflate call: 0 Z_BINARY, 1 Z_TEXT, 2 Z_UNKNOWN."""
        return self.lib.zamd_engine_data_type(self.handle)

    def alloc(self, nbytes: int) -> DeviceBuffer:
        return DeviceBuffer(self, nbytes)

    # -- device-resident entry points -------------------------------------------------
    def deflate_dev(self, d_in: int, n: int, d_out: int, out_cap: int, *, level: int = 6, strategy: int = 0,
                    mem_level: int = 8, chunk_size: int = 1 << 20, finish: bool = True, d_chunk_off: int = 0) -> int:
        total = C.c_uint64(0)
        _chk(self.lib.zamd_deflate_dev(self.handle, d_in, n, chunk_size, level, strategy, mem_level, int(finish),
                                       d_out, out_cap, d_chunk_off or None, C.byref(total)), "zamd_deflate_dev")
        return total.value

    def deflate_shard_dev(self, d_in: int, n: int, head_dict_len: int, d_out: int, out_cap: int, *, level: int = 6,
                          strategy: int = 0, mem_level: int = 8, chunk_size: int = 1 << 20, finish: bool = True,
                          d_chunk_off: int = 0) -> int:
        """One shard of a larger input: the head_dict_len bytes before d_in are its first chunk's dictionary."""
        total = C.c_uint64(0)
        _chk(self.lib.zamd_deflate_shard_dev(self.handle, d_in, n, head_dict_len, chunk_size, level, strategy, mem_level,
                                             int(finish), d_out, out_cap, d_chunk_off or None, C.byref(total)),
             "zamd_deflate_shard_dev")
        return total.value

    def timing_ms(self, which: int) -> Tuple[float, int]:
        n = C.c_int(0)
        ms = self.lib.zamd_last_timing_ms(self.handle, which, C.byref(n))
        return ms, n.value

    def adler32_dev(self, d_in: int, n: int, value: int = 1) -> int:
        r = C.c_uint32(0)
        _chk(self.lib.zamd_adler32_dev(self.handle, value, d_in, n, C.byref(r)), "zamd_adler32_dev")
        return r.value

    def crc32_dev(self, d_in: int, n: int, value: int = 0) -> int:
        r = C.c_uint32(0)
        _chk(self.lib.zamd_crc32_dev(self.handle, value, d_in, n, C.byref(r)), "zamd_crc32_dev")
        return r.value

    # -- host convenience (upload, run on the GPU, download) ---------------------------
    def deflate_chunks(self, data, *, level: int = 6, strategy: int = 0, mem_level: int = 8,
                       chunk_size: int = 1 << 20, finish: bool = True, dictionary: bytes = b"", parts: int = 0):
        """Chunked raw deflate of host bytes; returns (stream, chunk offsets + total).  `dictionary` (at most the
        window size) primes the first chunk as deflateSetDictionary does.  parts >= 2: a third value, the
        parts - 1 split records of every chunk (see record_splits)."""
        n = len(data)
        nchunks = max(1, (n + chunk_size - 1) // chunk_size)
        d_sp = self.alloc(8 * nchunks * (parts - 1)) if parts >= 2 else None
        if d_sp is not None:
            self.record_splits(d_sp.ptr, parts)
        cap = self.lib.zamd_deflate_bound(n, chunk_size, mem_level)
        d_stage = self.alloc(32768 + max(n, 16))
        d_in = DeviceBuffer.__new__(DeviceBuffer)
        d_in.engine, d_in.nbytes, d_in.ptr = self, max(n, 16), d_stage.ptr + 32768
        d_in.free = lambda: d_stage.free()
        d_out = self.alloc(cap)
        d_off = self.alloc(8 * (nchunks + 1))
        try:
            d_in.upload(data)
            if dictionary:
                d_stage.upload(dictionary, 32768 - len(dictionary))
                total = self.deflate_shard_dev(d_in.ptr, n, len(dictionary), d_out.ptr, cap, level=level, strategy=strategy,
                                               mem_level=mem_level, chunk_size=chunk_size, finish=finish, d_chunk_off=d_off.ptr)
            else:
                total = self.deflate_dev(d_in.ptr, n, d_out.ptr, cap, level=level, strategy=strategy, mem_level=mem_level,
                                         chunk_size=chunk_size, finish=finish, d_chunk_off=d_off.ptr)
            out = bytes(d_out.download(total))
            offs = list(memoryview(d_off.download(8 * (nchunks + 1))).cast("Q"))
            splits = list(memoryview(d_sp.download(8 * nchunks * (parts - 1))).cast("Q")) if d_sp is not None else None
        finally:
            self.record_splits(0, 0)
            d_in.free(); d_out.free(); d_off.free()
            if d_sp is not None:
                d_sp.free()
        return (out, offs, splits) if splits is not None else (out, offs)

    def adler32(self, data, value: int = 1) -> int:
        n = len(data)
        if n == 0:
            return value
        d = self.alloc(n)
        try:
            d.upload(data)
            return self.adler32_dev(d.ptr, n, value)
        finally:
            d.free()

    def crc32(self, data, value: int = 0) -> int:
        n = len(data)
        if n == 0:
            return value
        d = self.alloc(n)
        try:
            d.upload(data)
            return self.crc32_dev(d.ptr, n, value)
        finally:
            d.free()

    # -- inflate ------------------------------------------------------------------------
    def _inflate_error(self, rc: int, what: str) -> ZamdError:
        seg, st = C.c_int64(-1), C.c_uint32(0)
        self.lib.zamd_inflate_last_error(self.handle, C.byref(seg), C.byref(st))
        return ZamdError(rc, what, self.lib.zamd_inflate_message(st.value).decode(), seg.value)

    def record_splits(self, d_splits: int, parts: int) -> None:
        """The next deflate call also records parts - 1 split points per chunk into d_splits (uint64 each)."""
        _chk(self.lib.zamd_engine_splits(self.handle, d_splits or None, parts), "zamd_engine_splits")

    def inflate_dev(self, d_comp: int, d_chunk_off: int, n_chunks: int, d_out: int, n_out: int, *,
                    chunk_size: int = 1 << 20, comp_bytes: Optional[int] = None, d_head_window: int = 0,
                    d_splits: int = 0, parts: int = 0) -> None:
        """Batch inflate of a chunk-indexed raw deflate stream (device pointers); with d_splits (as record_splits made
        the deflate call write them) `parts` wavefronts share a chunk."""
        bad = C.c_int64(-1)
        if d_splits:
            rc = self.lib.zamd_inflate_split_dev(self.handle, d_comp, comp_bytes, d_chunk_off, n_chunks, chunk_size,
                                                 d_splits, parts, d_head_window or None, d_out, n_out, C.byref(bad))
        elif comp_bytes is None:
            rc = self.lib.zamd_inflate_dev(self.handle, d_comp, d_chunk_off, n_chunks, chunk_size, d_out, n_out, C.byref(bad))
        else:
            rc = self.lib.zamd_inflate_shard_dev(self.handle, d_comp, comp_bytes, d_chunk_off, n_chunks, chunk_size,
                                                 d_head_window or None, d_out, n_out, C.byref(bad))
        if rc == Z_DATA_ERROR:
            raise self._inflate_error(rc, "zamd_inflate_dev")
        _chk(rc, "zamd_inflate_dev")

    def chunk_offsets(self, d_lengths: int, n: int, d_offsets: int, base: int = 0) -> None:
        """Exclusive scan of n uint64 chunk lengths on the device into n + 1 offsets (the last one is the total)."""
        _chk(self.lib.zamd_chunk_offsets_dev(self.handle, d_lengths, n, base, d_offsets), "zamd_chunk_offsets_dev")

    def set_streams(self, nstreams: int = 4, serialize: bool = False) -> None:
        """Batches a deflate call keeps in flight; serialize: same batches on one stream (per-kernel timing)."""
        _chk(self.lib.zamd_engine_set_streams(self.handle, nstreams, int(serialize)), "zamd_engine_set_streams")

    def inflate_shard_begin(self, d_comp: int, comp_bytes: int, d_chunk_off: int, n_chunks: int, d_out: int, n_out: int, *,
                            chunk_size: int = 1 << 20, has_head: bool = False) -> None:
        """First step of the sharded inflate: decode without the window in front of the shard."""
        bad = C.c_int64(-1)
        rc = self.lib.zamd_inflate_shard_begin_dev(self.handle, d_comp, comp_bytes, d_chunk_off, n_chunks, chunk_size,
                                                   int(has_head), d_out, n_out, C.byref(bad))
        if rc == Z_DATA_ERROR:
            raise self._inflate_error(rc, "zamd_inflate_shard_begin_dev")
        _chk(rc, "zamd_inflate_shard_begin_dev")

    def inflate_shard_tail(self, d_head_window: int, d_tail_window: int) -> None:
        """Second step: this shard's last 32 KiB (the next shard's window) from the window in front of it."""
        _chk(self.lib.zamd_inflate_shard_tail_dev(self.handle, d_head_window or None, d_tail_window), "zamd_inflate_shard_tail_dev")

    def inflate_shard_finish(self, d_head_window: int) -> None:
        _chk(self.lib.zamd_inflate_shard_finish_dev(self.handle, d_head_window or None), "zamd_inflate_shard_finish_dev")

    def inflate_chunks(self, stream, offsets, n_out: int, *, chunk_size: int = 1 << 20, head_window=None,
                       splits=None, parts: int = 0) -> bytes:
        """Host convenience around inflate_dev: offsets = chunk offsets + total, as deflate_chunks returns; splits and
        parts as deflate_chunks(parts=...) returned them."""
        import struct
        n_chunks = len(offsets) - 1
        d_sp = self.alloc(8 * max(1, len(splits))) if splits is not None else None
        d_comp = self.alloc(len(stream) + 16)
        d_off = self.alloc(8 * len(offsets))
        d_out = self.alloc(max(n_out, 16))
        d_head = self.alloc(32768) if head_window is not None else None
        try:
            d_comp.upload(stream)
            d_off.upload(struct.pack(f"<{len(offsets)}Q", *offsets))
            if d_head is not None:
                d_head.upload(head_window)
            if d_sp is not None:
                d_sp.upload(struct.pack(f"<{len(splits)}Q", *splits))
            self.inflate_dev(d_comp.ptr, d_off.ptr, n_chunks, d_out.ptr, n_out, chunk_size=chunk_size,
                             comp_bytes=len(stream), d_head_window=d_head.ptr if d_head else 0,
                             d_splits=d_sp.ptr if d_sp is not None else 0, parts=parts)
            return bytes(d_out.download(n_out))
        finally:
            d_comp.free(); d_off.free(); d_out.free()
            if d_sp is not None:
                d_sp.free()
            if d_head is not None:
                d_head.free()

    def find_sync(self, stream, start: int = 0, cap: int = 1 << 16):
        """zamd_find_sync_dev on host bytes: sorted offsets just behind every 00 00 FF FF at or after `start`."""
        n = len(stream)
        d = self.alloc(max(n, 16))
        try:
            d.upload(stream)
            lst = (C.c_uint64 * cap)()
            cnt = C.c_uint32(0)
            _chk(self.lib.zamd_find_sync_dev(self.handle, d.ptr, start, n, lst, cap, C.byref(cnt)), "zamd_find_sync_dev")
            if cnt.value > cap:
                raise ZamdError(-5, "zamd_find_sync_dev", "more markers than the list holds", -1)
            return sorted(lst[:cnt.value])
        finally:
            d.free()

    def find_blocks(self, stream, targets_bit, window_bits):
        """zamd_find_blocks_dev on host bytes: per target the first plausible dynamic-block start, or None."""
        d = self.alloc(len(stream) + 16)
        try:
            d.upload(stream)
            n = len(targets_bit)
            tg = (C.c_uint64 * n)(*targets_bit)
            out = (C.c_uint64 * n)()
            _chk(self.lib.zamd_find_blocks_dev(self.handle, d.ptr, len(stream), tg, n, window_bits, out), "zamd_find_blocks_dev")
            return [None if v == 0xffffffffffffffff else int(v) for v in out]
        finally:
            d.free()

    def inflate_segments(self, stream, segs, *, out_bytes: int = 0, head_window=None, mark_step: int = 0):
        """zamd_inflate_segments_dev on host bytes.  segs: list of dicts with the input fields of InflateSeg.
        Returns (list of result dicts, output bytes or None when out_bytes == 0).  mark_step (counting passes:
        out_bytes == 0) asks for block marks (zamd_inflate_block_marks); every result dict then has "marks", the
        segment's list of (bits from in_bit, bytes of output) pairs."""
        arr = (InflateSeg * len(segs))()
        marks = None
        if mark_step:
            marks = (C.c_ulonglong * (len(segs) * INF_MARKS))()
            _chk(self.lib.zamd_inflate_block_marks(self.handle, marks, mark_step), "zamd_inflate_block_marks")
        for a, sdict in zip(arr, segs):
            for k, v in sdict.items():
                setattr(a, k, v)
        d_comp = self.alloc(len(stream) + 16)
        d_out = self.alloc(out_bytes) if out_bytes else None
        d_head = self.alloc(32768) if head_window is not None else None
        try:
            d_comp.upload(stream)
            if d_head is not None:
                d_head.upload(head_window)
            _chk(self.lib.zamd_inflate_segments_dev(self.handle, d_comp.ptr, len(stream), arr, len(segs),
                                                    d_head.ptr if d_head else None, d_out.ptr if d_out else None, out_bytes),
                 "zamd_inflate_segments_dev")
            res = [{f[0]: getattr(a, f[0]) for f in InflateSeg._fields_} for a in arr]
            for i, r in enumerate(res):
                r["msg"] = self.lib.zamd_inflate_message(r["status"]).decode()
                if marks is not None:
                    row = marks[i * INF_MARKS:(i + 1) * INF_MARKS]
                    r["marks"] = [(int(v) >> 32, int(v) & 0xffffffff) for v in row if v]
            return res, (bytes(d_out.download(out_bytes)) if d_out else None)
        finally:
            d_comp.free()
            if d_out is not None:
                d_out.free()
            if d_head is not None:
                d_head.free()

    def debug_dump(self, what: int, chunk: int, max_elems: int, elem_bytes: int) -> bytearray:
        buf = bytearray(max_elems * elem_bytes)
        arr = (C.c_char * len(buf)).from_buffer(buf)
        cnt = self.lib.zamd_debug_dump(self.handle, what, chunk, C.addressof(arr), max_elems)
        if cnt < 0:
            raise ZamdError(int(cnt), "zamd_debug_dump")
        del arr
        return buf[: cnt * elem_bytes]
