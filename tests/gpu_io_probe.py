import os, time, tempfile, mmap
p = os.path.join(tempfile.gettempdir(), "io_probe.bin")
buf = bytearray(os.urandom(1 << 20)) * 1024
t0 = time.time(); fd = os.open(p, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
mv = memoryview(buf); at = 0
while at < len(buf): at += os.write(fd, mv[at:at + (64 << 20)])
os.close(fd); dt = time.time() - t0
print("write 1 GiB: %.0f ms = %.2f GB/s" % (dt * 1e3, len(buf) / dt / 1e9))
for it in range(2):
    t0 = time.time(); fd = os.open(p, os.O_RDONLY); got = 0
    dst = bytearray(len(buf)); dv = memoryview(dst)
    t1 = time.time()
    while got < len(buf):
        r = os.readv(fd, [dv[got:got + (64 << 20)]]); got += r
    os.close(fd); dt = time.time() - t1
    print("read 1 GiB: %.0f ms = %.2f GB/s (alloc %.0f ms)" % (dt * 1e3, len(buf) / dt / 1e9, (t1 - t0) * 1e3))
os.remove(p)
print(tempfile.gettempdir(), os.cpu_count())
