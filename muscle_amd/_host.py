"""Host-side facts the Python layer needs (no GPU, no torch)."""
import os


def cpu_share() -> int:
    """CPUs this process may actually use: the affinity mask AND the cgroup's CPU quota.  The GPU boxes of this pool show
    256 CPUs with a quota of 16 (`cpu.max` = "1600000 100000"); torch then starts 128 intra-op threads and any CPU-side
    torch work (the test oracle, the CPU baseline of bench.py) runs 10-30x slower than on 8-16 threads
    (profiles/r05_cpu_probe.txt)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for quota_f, period_f in (("/sys/fs/cgroup/cpu.max", None),
                              ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us")):
        try:
            if period_f is None:
                quota, period = open(quota_f).read().split()
                if quota == "max":
                    continue
            else:
                quota, period = open(quota_f).read().strip(), open(period_f).read().strip()
            q, p = int(quota), int(period)
            if q > 0 and p > 0:
                n = min(n, max(1, q // p))
                break
        except (OSError, ValueError):
            continue
    return max(1, n)
