"""What bounds a snapshot SAVE on the GPU box's tmpfs: the same 8 GB written by 16 threads (a) into ONE fresh file, (b) into 16 fresh
files, (c) again over the one file (its pages exist), (d) one fresh file after a single fallocate.  os.pwrite releases the GIL.
    python scripts/probe/tmpfs_write_probe.py [GB] [DIR] [THREADS]"""
import os
import sys
import threading
import time

gb = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0
d = sys.argv[2] if len(sys.argv) > 2 else "/dev/shm"
nt = int(sys.argv[3]) if len(sys.argv) > 3 else 16
chunk = 64 << 20
buf = bytes(bytearray(os.urandom(1 << 20)) * 64)
per = int(gb * 1e9 / nt) // chunk * chunk
total = per * nt


def run(label, fds, offs):
    def work(fd, off):
        for a in range(0, per, chunk):
            os.pwrite(fd, buf, off + a)
    ths = [threading.Thread(target=work, args=(fds[t], offs[t])) for t in range(nt)]
    t0 = time.perf_counter()
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    dt = time.perf_counter() - t0
    print("%-46s %6.2f GB/s" % (label, total / dt / 1e9), flush=True)


one = os.path.join(d, "probe_one_%d" % os.getpid())
parts = [os.path.join(d, "probe_part_%d_%d" % (os.getpid(), t)) for t in range(nt)]
try:
    fd = os.open(one, os.O_WRONLY | os.O_CREAT, 0o644)
    run("one fresh file, %d threads" % nt, [fd] * nt, [t * per for t in range(nt)])
    run("the same file again (pages exist)", [fd] * nt, [t * per for t in range(nt)])
    os.close(fd)
    os.remove(one)
    fds = [os.open(p, os.O_WRONLY | os.O_CREAT, 0o644) for p in parts]
    run("%d fresh files, one thread each" % nt, fds, [0] * nt)
    for f in fds:
        os.close(f)
    for p in parts:
        os.remove(p)
    fd = os.open(one, os.O_WRONLY | os.O_CREAT, 0o644)
    t0 = time.perf_counter()
    os.posix_fallocate(fd, 0, total)
    print("%-46s %6.2f GB/s" % ("posix_fallocate of the whole file (1 thread)", total / (time.perf_counter() - t0) / 1e9), flush=True)
    run("one file after fallocate, %d threads" % nt, [fd] * nt, [t * per for t in range(nt)])
    os.close(fd)
finally:
    for p in [one] + parts:
        if os.path.exists(p):
            os.remove(p)
