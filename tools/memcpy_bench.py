import numpy as np, time, threading
n = 512 << 20
src = np.ones(n, dtype=np.uint8)
def run(T, fresh):
    dst = np.empty(n, dtype=np.uint8) if fresh else run.keep
    run.keep = dst
    parts = [(i * n // T, (i + 1) * n // T) for i in range(T)]
    t0 = time.perf_counter()
    th = [threading.Thread(target=lambda a=a, b=b: np.copyto(dst[a:b], src[a:b])) for a, b in parts]
    [t.start() for t in th]; [t.join() for t in th]
    return (time.perf_counter() - t0) * 1e3
run.keep = None
for T in (1, 2, 4, 8):
    print("threads %d: fresh destination %.1f ms, touched destination %.1f ms" % (T, run(T, True), run(T, False)))
