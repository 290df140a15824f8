"""CPUs this process may use: the smaller of the affinity mask and the cgroup CPU quota (mirrors bqc_cpu_limit of
bamqc_amd/host/parallel.h, which sizes the program's worker pools)."""
import math
import os


def cpu_limit():
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, math.ceil(int(q) / int(p)))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and p > 0:
                n = min(n, math.ceil(q / p))
        except Exception:
            pass
    return max(1, n)
