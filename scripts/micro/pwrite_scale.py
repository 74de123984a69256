import os, time, numpy as np, threading
n = 827_000_000
a = np.random.default_rng(0).random(n//8)
mv = memoryview(a.view(np.uint8))
def seq():
    with open("/dev/shm/_p.bin","w+b") as f:
        f.write(mv)
def par(T):
    with open("/dev/shm/_p.bin","w+b") as f:
        fd = f.fileno(); os.ftruncate(fd, len(mv))
        step = (len(mv)//T + 4095)//4096*4096
        def w(i):
            lo = i*step; hi = min(len(mv), lo+step); p = lo
            while p < hi:
                p += os.pwrite(fd, mv[p:min(hi, p+(64<<20))], p)
        th = [threading.Thread(target=w, args=(i,)) for i in range(T)]
        [t.start() for t in th]; [t.join() for t in th]
for name, fn in [("seq", seq), ("par2", lambda: par(2)), ("par4", lambda: par(4)), ("par8", lambda: par(8)), ("seq", seq)]:
    ts = []
    for r in range(3):
        t = time.perf_counter(); fn(); ts.append(time.perf_counter()-t)
    print(name, ["%.3f" % x for x in ts])
os.unlink("/dev/shm/_p.bin")
