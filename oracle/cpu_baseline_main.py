"""bench.py's cpu_baseline leg as a process of its own.  TEST INFRASTRUCTURE (the checker's C restatement timed on the host cores): started by
bench.py as a CHILD with OMP_PROC_BIND=spread OMP_PLACES=cores in its environment -- libgomp reads them when it is loaded, and a
bound main thread must not be the one that drives the GPU (its helper threads would inherit the one-core mask).  Prints one JSON line.

    python oracle/cpu_baseline_main.py H W SF [budget_seconds]        (full mask, like the bench's scene)"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main():
    cpus = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else None     # before libgomp binds this thread
    import cpu_budget
    eff = cpu_budget.effective_cpus()
    os.environ["SRPS_ORACLE_THREADS"] = str(eff)           # c_oracle would otherwise count the CPUs of the already bound main thread: one
    import numpy as np
    import c_oracle
    h, w, sf = (int(a) for a in sys.argv[1:4])
    budget = float(sys.argv[4]) if len(sys.argv) > 4 else 20.0
    out = c_oracle.bench_cpu_baseline(h, w, sf, np.ones(h * w, np.float32), budget_s=budget)
    out["nproc"] = os.cpu_count()
    out["affinity_cpus"] = cpus
    out["cgroup_cpu_max"] = cpu_budget.quota_text()
    out["cores_note"] = ("`cores` = OpenMP threads used = min(affinity, cgroup CPU quota): the box shows %d logical CPUs, the job may use %d"
                         % (os.cpu_count(), eff))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
