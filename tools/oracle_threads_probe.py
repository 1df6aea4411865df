"""Development aid: why the C oracle's depth step takes 85 s per call inside the GPU test process on the GPU boxes (0.9 s on 8 cores here).
Prints the CPU budget the process really has (cgroup quota, affinity) and times one CG solve of the 1024 x 1024 system per thread count,
bare and with torch imported first.   python tools/oracle_threads_probe.py [torch]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us", "/sys/fs/cgroup/cpuset.cpus.effective"):
    try:
        print(f, open(f).read().strip())
    except Exception as e:
        print(f, "-", type(e).__name__)
print({k: v for k, v in os.environ.items() if k.startswith(("OMP", "GOMP", "MKL", "OPENBLAS", "KMP"))})
if len(sys.argv) > 1 and sys.argv[1] == "torch":
    import torch
    print("torch imported, threads", torch.get_num_threads())
    if len(sys.argv) > 2 and sys.argv[2] == "cuda":
        torch.zeros(4, device="cuda"); torch.cuda.synchronize(); print("cuda initialised")
import numpy as np
import c_oracle as CO
h = w = 1024
st = CO.Structure(h, w, 4, np.ones(h * w, np.float32))
rng = np.random.default_rng(0)
M = np.abs(rng.normal(size=(6, st.P))).astype(np.float32); M[[1, 2, 4]] *= 0.1
t = time.perf_counter(); rp, ci, v = CO.assemble(st, M.reshape(-1)); print("assemble", round(time.perf_counter() - t, 3), "s on", CO.num_threads(), "threads")
b = rng.normal(size=st.P).astype(np.float32)
for thr in (0, 128, 64, 32, 16, 8, 1):
    if thr:
        CO.set_threads(thr)
    x = np.zeros(st.P, np.float32)
    t = time.perf_counter(); CO.cg_csr(rp, ci, v, x, b.copy(), fixed_iters=20); dt = time.perf_counter() - t
    print("threads", CO.num_threads(), "20 CG steps", round(dt, 3), "s")
    sys.stdout.flush()
