"""How many CPUs this process may really use: the smaller of its affinity mask and its cgroup's CPU quota.  TEST INFRASTRUCTURE (used by
the oracle's wrappers, tests/conftest.py and bench.py's cpu_baseline leg).  The GPU boxes of this pool show 256 logical CPUs and grant a
quota of 16 (cpu.max = "1600000 100000"): an OpenMP team of 256 -- or the 128 torch asks for -- is throttled by the scheduler's quota
the moment it spins at a barrier; 20 CG steps of the 1024 x 1024 system took 17 s on 256 threads and 7 ms on 64 (tools/
oracle_threads_probe.py, gpurun_out/r5c).  That, not the arithmetic, was round 4's "cpu_baseline is not reproducible" (14.8 ... 111
steps/s) and the 85 s per depth step of the first whole-solve test."""
import math
import os


def effective_cpus() -> int:
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max",):                       # cgroup v2
        try:
            quota, period = open(path).read().split()[:2]
            if quota != "max":
                n = min(n, max(1, math.ceil(int(quota) / int(period))))
        except Exception:
            pass
    try:                                                            # cgroup v1
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0 and p > 0:
            n = min(n, max(1, math.ceil(q / p)))
    except Exception:
        pass
    return max(1, n)


def quota_text() -> str:
    try:
        return open("/sys/fs/cgroup/cpu.max").read().strip()
    except Exception:
        return "unknown"
