"""How fast does this box take 2.4 GB of text into a tmpfs file?  pwrite from N threads, each its own stretch (the dump of .junctions at config 4's size)."""
import os
import sys
import threading
import time

size = int(float(sys.argv[1])) if len(sys.argv) > 1 else 2_400_000_000
path = "/dev/shm/pwrite_rate.bin"
buf = bytearray(os.urandom(1 << 20)) * 64        # 64 MiB source
for nt in (1, 4, 8, 16):
    fd = os.open(path, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o666)
    per = size // nt

    def work(t):
        at, end = t * per, (t + 1) * per
        while at < end:
            n = min(len(buf), end - at)
            os.pwrite(fd, memoryview(buf)[:n], at)
            at += n

    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(t,)) for t in range(nt)]
    [t.start() for t in th]
    [t.join() for t in th]
    os.close(fd)
    dt = time.perf_counter() - t0
    print(f"{nt:2d} threads: {size / 1e9:.2f} GB in {1e3 * dt:.0f} ms = {size / dt / 1e9:.2f} GB/s", flush=True)
    os.unlink(path)
