"""Development aid: ten set-ups of the metric's configuration in a row, each with the cgroup's throttling counters (cpu.stat) around it:
are the 6 - 10 ms outliers of srps_setup the scheduler's CPU quota (nr_throttled goes up) or something else?   python tools/setup_outliers.py [NAME=VALUE env knobs]"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for kv in sys.argv[1:]:
    if "=" in kv:
        os.environ[kv.split("=")[0]] = kv.split("=")[1]
import torch


def cpu_stat():
    try:
        return {l.split()[0]: int(l.split()[1]) for l in open("/sys/fs/cgroup/cpu.stat")}
    except Exception:
        return {}


pkg = importlib.import_module("srmeetsps-cuda_amd")
sc = pkg.synth.make_scene(2048, 2048, 4, 20, seed=1237, mask_kind="full")
dh = pkg.DataHandler.from_scene(sc)
ctx = pkg.Context(device_id=0)
ctx.set_option("exclusive_device", 1)
ctx.setup(dh); pkg.alternating_loop(ctx, None, max_outer=2)
for rep in range(12):
    a = cpu_stat()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ctx.setup(dh)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    b = cpu_stat()
    print(f"set-up {rep}: {1e3 * dt:.2f} ms; throttled periods +{b.get('nr_throttled', 0) - a.get('nr_throttled', 0)}, throttled {1e-3 * (b.get('throttled_usec', 0) - a.get('throttled_usec', 0)):.2f} ms, "
          f"cpu time {1e-3 * (b.get('usage_usec', 0) - a.get('usage_usec', 0)):.1f} ms", flush=True)
    pkg.alternating_loop(ctx, None, max_outer=1)
    time.sleep(0.05 * (rep % 3))
ctx.close()
