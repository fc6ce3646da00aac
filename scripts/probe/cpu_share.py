"""How much host CPU does the GPU box really give: wall time of P busy processes, P = 1, 2, 4, 8, 16."""
import multiprocessing as mp, os, time


def burn(_):
    t = time.perf_counter(); x = 0
    while time.perf_counter() - t < 2.0:
        for i in range(10000):
            x += i * i
    return x


def count(_):
    t = time.perf_counter(); c = 0
    while time.perf_counter() - t < 2.0:
        for i in range(10000):
            c += 1
    return c


if __name__ == "__main__":
    print("nproc", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
    for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
        try:
            print(f, open(f).read().strip())
        except OSError:
            pass
    for p in (1, 2, 4, 8, 16):
        with mp.Pool(p) as pool:
            r = pool.map(count, range(p))
        print("procs", p, "work per proc (M loops / 2 s)", [round(v / 1e6, 1) for v in r][:4], "total", round(sum(r) / 1e6, 1))
