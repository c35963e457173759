"""GPU tests: the gz* file layer of libzamd_z.so (gzlib.c, gzread.c, gzwrite.c of the reference) — files it
writes are read by Python's gzip module (system zlib), files Python writes are read by it, and the calls of
the reference's own test (test/example.c:88-170, test_gzio) give the reference's results."""
import ctypes as C
import gzip
import os
import subprocess

import pytest

import _oracle as O
import _zapi as Z

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _read_all(l, g, piece):
    buf = C.create_string_buffer(piece)
    out = bytearray()
    while True:
        n = l.gzread(g, buf, piece)
        assert n >= 0, l.gzerror(g, None)
        if n == 0:
            break
        out += buf.raw[:n]
    return bytes(out)


def test_the_calls_of_the_reference_test(tmp_path):
    """test/example.c:88-170: putc, puts, printf, a seek while writing, then read, seek back, getc, ungetc, gets."""
    l = Z.lib()
    name = str(tmp_path / "foo.gz").encode()
    g = l.gzopen(name, b"wb")
    assert g
    assert l.gzputc(g, ord("h")) == ord("h")
    assert l.gzputs(g, b"ello") == 4
    assert l.gzprintf(C.c_void_p(g), b", %s!", b"hello") == 8
    assert l.gzseek(g, 1, 1) == 14                      # one zero byte more
    assert l.gztell(g) == 14
    assert l.gzclose(g) == Z.Z_OK
    assert gzip.open(name.decode()).read() == b"hello, hello!\0"

    g = l.gzopen(name, b"rb")
    assert g
    buf = C.create_string_buffer(64)
    assert l.gzread(g, buf, 64) == 14
    assert buf.raw[:14] == b"hello, hello!\0"
    assert l.gzseek(g, -8, 1) == 6
    assert l.gztell(g) == 6
    assert l.gzgetc(g) == ord(" ")
    assert l.gzungetc(ord(" "), g) == ord(" ")
    line = C.create_string_buffer(64)
    assert l.gzgets(g, line, 64) == C.addressof(line)
    assert line.value == b" hello!"
    assert line.raw[:9] == b" hello!\0\0"                # the zero byte of the seek, then the terminator
    assert l.gzeof(g) == 1 and l.gzgetc(g) == -1          # gzgets ran into the end looking for a newline
    assert l.gzdirect(g) == 0
    assert l.gzclose(g) == Z.Z_OK


def test_files_written_here_are_gzip_files(tmp_path):
    l = Z.lib()
    data = O.corpus(5 * (1 << 20) + 4321, 4, 101)
    name = str(tmp_path / "w.gz").encode()
    g = l.gzopen(name, b"wb6")
    assert g
    assert l.gzbuffer(g, 1 << 20) == 0
    pos = 0
    for step in (1, 7, 100000, 3 << 20, 12345, 1 << 20):            # small writes are collected, large ones go straight through
        piece = data[pos:pos + step]
        assert l.gzwrite(g, piece, len(piece)) == len(piece)
        pos += len(piece)
        if step == 100000:
            assert l.gzflush(g, Z.Z_SYNC_FLUSH) == Z.Z_OK
        if step == 12345:
            assert l.gzsetparams(g, 9, 1) == Z.Z_OK
    rest = data[pos:]
    assert l.gzwrite(g, rest, len(rest)) == len(rest)
    assert l.gztell(g) == len(data)
    assert l.gzbuffer(g, 4096) == -1                               # too late
    assert l.gzclose(g) == Z.Z_OK
    assert gzip.open(name.decode()).read() == data

    # append: a second member, and gzip readers see the concatenation
    g = l.gzopen(name, b"ab1")
    assert l.gzwrite(g, b"tail", 4) == 4
    assert l.gzclose(g) == Z.Z_OK
    assert gzip.open(name.decode()).read() == data + b"tail"

    # and back through this library, in ragged reads
    g = l.gzopen(name, b"rb")
    assert l.gzbuffer(g, 1 << 20) == 0
    assert _read_all(l, g, 777777) == data + b"tail"
    assert l.gzeof(g) == 1
    err = C.c_int(99)
    assert l.gzerror(g, C.byref(err)) == b"" and err.value == Z.Z_OK
    assert l.gzclose(g) == Z.Z_OK


def test_files_written_by_the_system_zlib_are_read(tmp_path):
    l = Z.lib()
    lines = b"".join(b"line %d of the file\n" % i for i in range(200000))
    name = str(tmp_path / "r.gz")
    with gzip.open(name, "wb", compresslevel=6) as f:
        f.write(lines)
    with open(name, "ab") as f:                                     # a second member, then bytes that are no gzip stream
        f.write(gzip.compress(b"second member\n") + b"trailing garbage")
    g = l.gzopen(name.encode(), b"r")
    assert g
    buf = C.create_string_buffer(256)
    for i in (0, 1, 2):
        assert l.gzgets(g, buf, 256) and buf.value == b"line %d of the file\n" % i
    assert l.gzgets(g, buf, 8) and buf.value == b"line 3 "              # len - 1 bytes at most
    here = l.gztell(g)
    assert l.gzseek(g, len(lines) - 5, 0) == len(lines) - 5            # forward: by reading on
    rest = _read_all(l, g, 4096)
    assert rest == b"file\n" + b"second member\n"                       # garbage behind the last member is ignored (gzread.c:144-151)
    assert l.gzseek(g, here, 0) == here                                 # backward: rewind and read again
    assert l.gzgets(g, buf, 256) and buf.value == b"of the file\n"
    assert l.gzrewind(g) == 0 and l.gztell(g) == 0
    assert l.gzread(g, buf, 5) == 5 and buf.raw[:5] == b"line "
    assert l.gzoffset(g) > 0
    assert l.gzclose_r(g) == Z.Z_OK


def test_transparent_files(tmp_path):
    l = Z.lib()
    plain = O.corpus(300000, 0, 103)
    name = str(tmp_path / "plain.txt")
    open(name, "wb").write(plain)
    g = l.gzopen(name.encode(), b"rb")
    assert l.gzdirect(g) == 1
    assert _read_all(l, g, 65536) == plain
    assert l.gzseek(g, 1000, 0) == 1000                               # plain files are sought directly (gzlib.c:394-407)
    buf = C.create_string_buffer(10)
    assert l.gzread(g, buf, 10) == 10 and buf.raw == plain[1000:1010]
    assert l.gzclose(g) == Z.Z_OK
    # 'T' writes without compressing
    name2 = str(tmp_path / "t.out")
    g = l.gzopen(name2.encode(), b"wbT")
    assert l.gzwrite(g, plain, len(plain)) == len(plain)
    assert l.gzclose(g) == Z.Z_OK
    assert open(name2, "rb").read() == plain
    assert not l.gzopen(name.encode(), b"rT") and not l.gzopen(name.encode(), b"r+") and not l.gzopen(name.encode(), b"q")
    empty = str(tmp_path / "empty")
    open(empty, "wb").close()
    g = l.gzopen(empty.encode(), b"rb")
    assert l.gzread(g, buf, 10) == 0 and l.gzeof(g) == 1 and l.gzdirect(g) == 1
    assert l.gzclose(g) == Z.Z_OK


def test_damaged_files_report_the_reference_errors(tmp_path):
    l = Z.lib()
    data = O.corpus(400000, 0, 105)
    blob = gzip.compress(data, 6)
    cut = str(tmp_path / "cut.gz")
    open(cut, "wb").write(blob[:len(blob) // 2])
    g = l.gzopen(cut.encode(), b"rb")
    got = bytearray()
    buf = C.create_string_buffer(65536)
    while True:
        n = l.gzread(g, buf, 65536)
        if n <= 0:
            break
        got += buf.raw[:n]
    assert data.startswith(bytes(got)) and len(got) > 0
    err = C.c_int(0)
    msg = l.gzerror(g, C.byref(err))
    assert err.value == Z.Z_BUF_ERROR and msg == cut.encode() + b": unexpected end of file"      # gzread.c:186
    assert l.gzclose(g) == Z.Z_BUF_ERROR                                                        # gzread.c:588
    bad = bytearray(blob)
    bad[-6] ^= 0x10                                                                              # inside the stored crc
    badname = str(tmp_path / "bad.gz")
    open(badname, "wb").write(bytes(bad))
    g = l.gzopen(badname.encode(), b"rb")
    while True:
        n = l.gzread(g, buf, 65536)
        if n <= 0:
            break
    assert n == -1
    msg = l.gzerror(g, C.byref(err))
    assert err.value == Z.Z_DATA_ERROR and msg == badname.encode() + b": incorrect data check"
    assert l.gzread(g, buf, 10) == -1
    l.gzclearerr(g)
    assert l.gzerror(g, C.byref(err)) == b"" and err.value == Z.Z_OK
    assert l.gzclose(g) == Z.Z_OK
    # a write handle refuses reads and the other way round
    g = l.gzopen(str(tmp_path / "x.gz").encode(), b"wb")
    assert l.gzread(g, buf, 10) == -1 and l.gzgetc(g) == -1 and l.gzclose_r(g) == Z.Z_STREAM_ERROR
    assert l.gzflush(g, 9) == Z.Z_STREAM_ERROR
    assert l.gzseek(g, -1, 1) == -1                                                             # no way back while writing
    assert l.gzclose(g) == Z.Z_OK
    assert gzip.open(str(tmp_path / "x.gz")).read() == b""


def test_the_example_program_on_a_64_mib_file(tmp_path):
    """The reference's CPU-runnable configuration (minigzip -6 on a 64 MiB mixed file) through examples/zamd_gzip."""
    import zlib_amd
    exe = os.path.join(ROOT, "examples", "zamd_gzip")
    assert os.path.exists(exe), "run make"
    data = bytes(zlib_amd.corpus(64 << 20, 4, 5, piece_bytes=16 << 20, threads=8))
    name = str(tmp_path / "mixed.bin")
    open(name, "wb").write(data)
    subprocess.run([exe, "-6", name], check=True, timeout=300)
    packed = open(name + ".gz", "rb").read()
    assert len(packed) < len(data) * 0.5
    assert gzip.decompress(packed) == data
    os.remove(name)
    subprocess.run([exe, "-d", name + ".gz"], check=True, timeout=300)
    assert open(name, "rb").read() == data


def test_a_failing_engine_call_is_an_error_of_gzwrite(tmp_path):
    """deflate() of this library can fail where the reference's cannot (device memory, engine errors).  gzwrite must
    then report the failure (0 bytes written, gzerror Z_MEM_ERROR) and must not come back to the caller's buffer on the
    next call (gzwrite.c:74-131 only ever sees Z_STREAM_ERROR).  The failure is injected: ZAMD_Z_FAIL_DEFLATE_AFTER.
    Runs in a process of its own because the switch is counted per process."""
    script = r'''
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(%r, "tests"))
import _zapi as Z
l = Z.lib()
name = sys.argv[1].encode()
big = bytes(range(256)) * (40 << 10)            # 10 MiB: goes to deflate() from the caller's buffer
os.environ["ZAMD_Z_FAIL_DEFLATE_AFTER"] = "1"
g = l.gzopen(name, b"wb6")
assert l.gzbuffer(g, 1 << 20) == 0
assert l.gzwrite(g, big, len(big)) == len(big)  # buffered by deflate(): no engine call yet
assert l.gzflush(g, Z.Z_SYNC_FLUSH) == 0        # engine call 1 goes through
assert l.gzwrite(g, big, len(big)) == len(big)
rc = l.gzflush(g, Z.Z_SYNC_FLUSH)               # engine call 2 fails
err = C.c_int(0)
msg = l.gzerror(g, C.byref(err))
assert rc == Z.Z_MEM_ERROR and err.value == Z.Z_MEM_ERROR and msg == b"out of memory", (rc, err.value, msg)
assert l.gzwrite(g, b"abc", 3) == 0             # the error sticks (gzwrite.c:177-178), and no stale pointer is followed
assert l.gzclose(g) == Z.Z_MEM_ERROR
print("ok")
''' % ROOT
    r = subprocess.run([os.sys.executable, "-c", script, str(tmp_path / "f.gz")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip() == "ok", r.stdout + r.stderr


def test_writes_collected_behind_the_caller_give_the_same_file(tmp_path):
    """gzwrite() collects writes of some size in buffers of deflate()'s batch size and compresses a full one on a thread
    of the file's own.  What the file holds must not depend on it (deflate() does not care how its input is cut,
    zlib.h:252-262): the same calls with ZAMD_GZ_BEHIND=0 give the same bytes; putc, printf, a flush, a seek and a write
    larger than the buffers lie between the collected writes (gzwrite.c:165-243, 311-373, 478-510).  Then the thread's
    engine call fails: a later call reports it, and gzclose does.  (A child process: sizes come from the environment.)"""
    script = r'''
import ctypes as C, gzip, os, sys
sys.path.insert(0, os.path.join(%r, "tests"))
os.environ["ZAMD_Z_BATCH_MIB"] = "2"
os.environ["ZAMD_Z_CHUNK"] = "262144"
os.environ["ZAMD_GZ_BUF_MIB"] = "1"
import _zapi as Z
import zlib_amd
l = Z.lib()
data = bytes(zlib_amd.corpus(14 << 20, 4, 31))
def write(name, behind):
    os.environ["ZAMD_GZ_BEHIND"] = behind
    g = l.gzopen(name, b"wb6I")
    want = bytearray()
    at = 0
    def put(k):
        nonlocal at
        assert l.gzwrite(g, data[at:at + k], k) == k
        want.extend(data[at:at + k]); at += k
        assert l.gztell(g) == len(want)
    put(1000)                                   # small: the file's own buffer
    for _ in range(5): put(700000)              # collected; the buffer fills and goes to the thread
    assert l.gzputc(g, 65) == 65; want.append(65)
    put(600000)
    assert l.gzprintf(C.c_void_p(g), b"%%s-%%d", b"abc", 7) == 5; want.extend(b"abc-7")
    put(3 << 20)                                # larger than a buffer: from the caller's memory
    for _ in range(3): put(900001)
    assert l.gzflush(g, Z.Z_SYNC_FLUSH) == 0
    assert l.gzoffset(g) > 0
    assert l.gzseek(g, 5000, 1) == len(want) + 5000; want.extend(bytes(5000))
    for _ in range(4): put(524288)
    assert l.gzclose(g) == 0
    return bytes(want)
a, b = sys.argv[1].encode(), sys.argv[2].encode()
want = write(a, "1")
assert write(b, "0") == want
assert open(a, "rb").read() == open(b, "rb").read()
assert gzip.open(a.decode()).read() == want
g = l.gzopen(a, b"rb")
buf = C.create_string_buffer(len(want) + 1)
assert l.gzread(g, buf, len(want) + 1) == len(want) and buf.raw[:len(want)] == want
l.gzclose(g)
# the thread's engine call fails
os.environ["ZAMD_GZ_BEHIND"] = "1"
os.environ["ZAMD_Z_FAIL_DEFLATE_AFTER"] = "0"
g = l.gzopen(b, b"wb6")
sent = [l.gzwrite(g, data[:700000], 700000) for _ in range(8)]      # the third fills the buffer, the sixth waits for its job
assert sent[0] == 700000 and sent[-1] == 0, sent
err = C.c_int(0)
assert l.gzerror(g, C.byref(err)) == b"out of memory" and err.value == Z.Z_MEM_ERROR
assert l.gzclose(g) == Z.Z_MEM_ERROR
print("ok")
''' % ROOT
    r = subprocess.run([os.sys.executable, "-c", script, str(tmp_path / "a.gz"), str(tmp_path / "b.gz")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip() == "ok", r.stdout + r.stderr


def test_reading_ahead_of_the_caller_gives_the_same_bytes(tmp_path):
    """A reader that takes its data in small requests gets it inflated ahead, several buffers' worth at a time, by a
    thread of the file's own.  What gzread, gzgetc, gzungetc, gzgets, gzseek, gzrewind and gzeof hand out must not depend
    on it (gzread.c:288-544, gzlib.c:339-436): the same calls with ZAMD_GZ_AHEAD_MIB=0 give the same results, also on a
    file of two members, on one cut short (Z_BUF_ERROR behind everything that could be inflated, gzread.c:196-200) and on
    a damaged one (Z_DATA_ERROR in the end; the same bytes up to there either way).  (A child process: sizes come from the environment.)"""
    script = r'''
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(%r, "tests"))
os.environ["ZAMD_Z_CHUNK"] = "262144"
os.environ["ZAMD_GZ_BUF_MIB"] = "1"
os.environ["ZAMD_GZ_AHEAD_AFTER_KIB"] = "1024"
import _zapi as Z
import zlib_amd
l = Z.lib()
data = bytes(zlib_amd.corpus(30 << 20, 4, 32)).replace(b"\n", b" ")
data = data[:9 << 20] + b"\n" + data[(9 << 20) + 1:]
tmp = sys.argv[1]
def path(name): return os.path.join(tmp, name).encode()
g = l.gzopen(path("a.gz"), b"wb6I"); assert l.gzwrite(g, data, len(data)) == len(data); assert l.gzclose(g) == 0
g = l.gzopen(path("two.gz"), b"wb6"); assert l.gzwrite(g, data[:20 << 20], 20 << 20) == 20 << 20; assert l.gzclose(g) == 0
g = l.gzopen(path("two.gz"), b"ab1"); assert l.gzwrite(g, data[20 << 20:], 10 << 20) == 10 << 20; assert l.gzclose(g) == 0
raw = open(path("a.gz"), "rb").read()
open(path("cut.gz"), "wb").write(raw[:len(raw) * 6 // 10])
bad = bytearray(raw); bad[len(raw) // 2] ^= 0x55
open(path("bad.gz"), "wb").write(bytes(bad))
buf = C.create_string_buffer(1 << 20)
def session(name, ahead, whole):
    os.environ["ZAMD_GZ_AHEAD_MIB"] = ahead
    log = []
    g = l.gzopen(path(name), b"rb")
    out = bytearray()
    for k in ([300000] * 30 + [1, 70000, 999999] * 8) * (1 if whole else 4):     # (a file that is not whole: until it fails)
        r = l.gzread(g, buf, k); log.append(r)
        if r <= 0: break
        out += buf.raw[:r]
        if len(out) %% 7 == 0:
            assert l.gzungetc(out[-1], g) == out[-1]; del out[-1]
        if len(out) %% 5 == 0:
            c = l.gzgetc(g); log.append(c)
            if c >= 0: out.append(c)
        assert l.gztell(g) == len(out)
    err = C.c_int(0); l.gzerror(g, C.byref(err)); log.append(err.value)
    if whole:
        assert err.value == 0
        line = C.create_string_buffer(200)
        log.append(l.gzgets(g, line, 200)); out += line.value
        at = l.gzseek(g, 5 << 20, 1); log.append(at)              # forward, over what may have been inflated ahead
        r = l.gzread(g, buf, 1000); log.append(buf.raw[:r]); assert buf.raw[:r] == data[at:at + r]
        assert l.gzseek(g, 1000, 0) == 1000                        # back: from the start again
        r = l.gzread(g, buf, 100); assert buf.raw[:r] == data[1000:1100]
        assert l.gzseek(g, len(data) - 10, 0) == len(data) - 10
        r = l.gzread(g, buf, 100); log.append(r); log.append(l.gzeof(g)); assert buf.raw[:r] == data[-10:]
        assert l.gzrewind(g) == 0
        r = l.gzread(g, buf, 100); assert buf.raw[:r] == data[:100]
    log.append(l.gzclose(g))
    return bytes(out), log
for name in ("a.gz", "two.gz", "cut.gz", "bad.gz"):
    whole = name in ("a.gz", "two.gz")
    a, la = session(name, "4", whole)
    b, lb = session(name, "0", whole)
    if name != "bad.gz":
        assert data.startswith(a[:len(a) - 200]) and a == b and la == lb, (name, len(a), len(b), la[-8:], lb[-8:])
    else:       # (what a flipped byte does to the data behind it shows at the latest in the check value; both ways see the same bytes)
        m = min(len(a), len(b))
        assert m > 10 << 20 and a[:10 << 20] == data[:10 << 20] and a[:m] == b[:m], (len(a), len(b))
    if name == "cut.gz": assert la[-2] == Z.Z_BUF_ERROR and la[-1] == Z.Z_BUF_ERROR, la[-4:]
    if name == "bad.gz": assert la[-2] == Z.Z_DATA_ERROR and lb[-2] == Z.Z_DATA_ERROR, (la[-4:], lb[-4:])
print("ok")
''' % ROOT
    r = subprocess.run([os.sys.executable, "-c", script, str(tmp_path)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip() == "ok", r.stdout + r.stderr


def test_gzseek_starts_from_the_window_in_front_of_the_chunk(tmp_path):
    """A file written with 'W' stores the 32 KiB in front of every chunk beside the index (what examples/zran.c:67-72 keeps
    in memory): gzseek to any offset then inflates one chunk, not the file from its start.  256 MiB, 20 random offsets,
    forwards and backwards; other gzip readers still read the file, and a reader that ignores the index reads the same."""
    import random
    import zlib_amd
    l = Z.lib()
    n = 256 << 20
    data = bytes(zlib_amd.corpus(n, 4, 77, piece_bytes=16 << 20, threads=8))
    name = str(tmp_path / "big.gz").encode()
    g = l.gzopen(name, b"wb6W")
    assert g
    for at in range(0, n, 48 << 20):
        piece = data[at:at + (48 << 20)]
        assert l.gzwrite(g, piece, len(piece)) == len(piece)
    assert l.gzclose(g) == Z.Z_OK
    size = os.path.getsize(name)
    fd = os.open(name, os.O_RDONLY)
    ix = Z.GzIndex()
    assert l.zamd_gz_index_load(fd, C.byref(ix)) == 0
    assert ix.n_chunks == 256 and ix.window_every == 1 and ix.n_windows == 255 and 0 < ix.window_offset < size
    # the index carries the split records of its chunks too (block boundaries: inflate() puts a wavefront on every part)
    assert ix.splits_per_chunk == 7 and sum(1 for c in range(256) if ix.splits[7 * c]) > 200
    for c in (0, 100, 255):
        recs = [ix.splits[7 * c + k] for k in range(7) if ix.splits[7 * c + k]]
        assert recs == sorted(recs) and all(0 < (r & 0xffffffff) < (1 << 20) and 0 < (r >> 32) < 8 * ix.pairs[2 * c] for r in recs)
    l.zamd_gz_index_free(C.byref(ix))
    os.close(fd)
    assert size < 0.6 * n                                                  # windows cost 3 % of the data, not more

    g = l.gzopen(name, b"rb")
    assert g
    rng = random.Random(5)
    buf = C.create_string_buffer(70000)
    chunks, seeks = C.c_ulonglong(0), C.c_ulonglong(0)
    offsets = [rng.randrange(1 << 20, n - 70000) for _ in range(18)] + [n - 5000, (200 << 20) - 10]      # the last chunk; across a chunk boundary
    for k, off in enumerate(offsets):
        assert l.zamd_gz_seek_stats(g, C.byref(chunks), C.byref(seeks)) == 0
        before, seeks_before = chunks.value, seeks.value
        assert l.gzseek64(g, off, 0) == off
        want = 5000 if off == n - 5000 else 30
        assert l.gzread(g, buf, want) == want, l.gzerror(g, None)
        assert buf.raw[:want] == data[off:off + want], (k, off)
        assert l.gztell64(g) == off + want
        assert l.zamd_gz_seek_stats(g, C.byref(chunks), C.byref(seeks)) == 0
        assert seeks.value == seeks_before + 1, (k, off)
        assert chunks.value - before <= 2, (k, off, chunks.value - before)
    # reading on from where a seek ended, across chunks and to the end of the file
    off = n - (3 << 20) - 777
    assert l.gzseek64(g, off, 0) == off
    assert _read_all(l, g, 1 << 20) == data[off:]
    assert l.gzeof(g) == 1
    # a short seek forwards inside what is already inflated stays the ordinary kind
    assert l.gzseek64(g, 5 << 20, 0) == 5 << 20
    assert l.gzread(g, buf, 10) == 10 and buf.raw[:10] == data[5 << 20:(5 << 20) + 10]
    assert l.zamd_gz_seek_stats(g, C.byref(chunks), C.byref(seeks)) == 0
    s0 = seeks.value
    assert l.gzseek64(g, 100, 1) == (5 << 20) + 110
    assert l.gzread(g, buf, 10) == 10 and buf.raw[:10] == data[(5 << 20) + 110:(5 << 20) + 120]
    assert l.zamd_gz_seek_stats(g, C.byref(chunks), C.byref(seeks)) == 0 and seeks.value == s0
    assert l.gzclose(g) == Z.Z_OK

    # another reader (system zlib) and this one without the index
    with gzip.open(name.decode()) as f:
        assert f.read(1 << 20) == data[:1 << 20]
        f.seek(100 << 20)
        assert f.read(1000) == data[100 << 20:(100 << 20) + 1000]
    env = dict(os.environ, ZAMD_GZ_INDEX="0")
    code = ("import ctypes as C, sys\nsys.path.insert(0, %r)\nimport _zapi as Z\nl = Z.lib()\ng = l.gzopen(%r, b'rb')\n"
            "buf = C.create_string_buffer(1000)\nassert l.gzseek64(g, 150 << 20, 0) == 150 << 20\nassert l.gzread(g, buf, 1000) == 1000\n"
            "sys.stdout.buffer.write(buf.raw)\n" % (os.path.join(ROOT, "tests"), name))
    r = subprocess.run([os.sys.executable, "-c", code], env=env, capture_output=True, timeout=300)
    assert r.returncode == 0 and r.stdout == data[150 << 20:(150 << 20) + 1000], r.stderr[-500:]


def test_random_use_of_the_gz_layer():
    """tests/gpu_gz_fuzz.py: files written with random gzwrite() sizes, modes (plain, chunk index, windows) and chunk
    sizes are gzip files to Python's gzip, read back whole in random pieces and at random offsets.  (A child process: the
    tool sets sizes through the environment.)"""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "gpu_gz_fuzz.py"), "15", "21"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "bad 0" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
