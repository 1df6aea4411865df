#!/usr/bin/env python3
"""bench.py -- CG iterations/sec of the SRPS hot path on synthetic data (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    N > 1 runs one process per GPU.  Started under torch.distributed.run (WORLD_SIZE in the environment) this process IS a rank;
    started bare (`python bench.py --gpus 2`) it starts `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ...
    bench.py <same arguments>` as a CHILD before anything here touches the GPU (no torch import in the parent), relays the
    ranks' output -- rank 0's JSON line -- and exits with the child's code.  The seam in the reference: Main.cpp:29 / SRPS.cu:88
    (`cudaSetDevice(Preferences::deviceId)`, its only device selection).

Workload (config.workload): synthetic full-mask HR grid 2048x2048, sf 4, 20 images per GPU, 3 channels,
seed 1234+3.  One "step" = one pass of the alternating loop of SRPS.cu:276-315:
lighting -> albedo -> depth (tensor assembly + 101 truncated CG steps + energy) -> normals, inputs
resident in HBM.  value = CG iterations of the whole job per second = 101*K / T (the CG is replicated
on every rank when the images are sharded, so it is counted once -- see DESIGN.md section 7).
Also reported: the isolated inner loop (cg_only_*), the full solve to the reference's stop rule
(total_solve_s), the roofline of the dominant kernel and the CPU baseline.
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


def cpu_baseline(sc, pkg, budget_s=20.0):
    """The oracle's C restatement of the depth CG (assembled CSR, exactly dc.cu:229-279) timed on the
    host cores on a bounded number of iterations of the SAME 2048^2 system.  Falls back to the numpy
    restatement when the C oracle is not built.  Checker code, imported here only."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    if os.path.exists(os.path.join(ROOT, "oracle", "libsrps_oracle.so")):
        # as a CHILD process with the OpenMP threads bound (libgomp reads the binding when it is loaded; this process's libgomp came in
        # with torch long ago, and a bound main thread here would hand its one-core mask to the GPU runtime's helper threads).  The child
        # never touches the GPU.  Full mask, like `sc`.
        import subprocess
        assert int(np.count_nonzero(sc.mask)) == sc.h * sc.w, "the cpu_baseline child takes a full mask"
        import cpu_budget                                   # checker-side helper: the CPUs the job may really use (affinity, cgroup quota)
        env = {k: v for k, v in os.environ.items() if k not in ("GOMP_CPU_AFFINITY", "KMP_AFFINITY")}
        env.update(OMP_PROC_BIND="spread", OMP_PLACES="cores", OMP_NUM_THREADS=str(cpu_budget.effective_cpus()))
        try:
            res = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "cpu_baseline_main.py"), str(sc.h), str(sc.w), str(sc.sf), str(budget_s)],
                                 env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
            line = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
            if res.returncode == 0 and line:
                return dict(json.loads(line[-1]), implementation="C/OpenMP (oracle/srps_oracle.c), timed in a child process with bound threads")
            print(f"bench.py: the cpu_baseline child failed (rc {res.returncode}): {res.stderr[-500:]}", file=sys.stderr)
        except Exception as exc:
            print(f"bench.py: the cpu_baseline child failed ({exc})", file=sys.stderr)
    else:
        print("bench.py: the C oracle is not built; timing the numpy restatement on one thread instead", file=sys.stderr)
    import srps_oracle as O
    geo = O.build_geometry(sc.h, sc.w, sc.sf, sc.mask)
    P = geo.npix
    rng = np.random.default_rng(0)
    M = np.abs(rng.normal(size=(6, P))).astype(np.float32)
    M[[1, 2, 4]] *= 0.1
    x = rng.normal(size=P).astype(np.float32)
    b = rng.normal(size=P).astype(np.float32)
    Dx = geo.Dx; Dy = geo.Dy; KT = geo.KT
    DxT = Dx.T.tocsr(); DyT = Dy.T.tocsr(); KTT = KT.T.tocsr()

    def mv(v):
        gx = Dx @ v; gy = Dy @ v
        u = M[0] * gx + M[1] * gy + M[2] * v
        w = M[1] * gx + M[3] * gy + M[4] * v
        t = M[2] * gx + M[4] * gy + M[5] * v
        return DxT @ u + DyT @ w + t + KTT @ (KT @ v)

    iters = 0
    t0 = time.perf_counter()
    p = b.copy(); r = b.copy(); r1 = float(r @ r)
    while time.perf_counter() - t0 < budget_s and iters < 101:
        w = mv(p); a = r1 / float(p @ w); x += a * p; r -= a * w
        r0 = r1; r1 = float(r @ r); p = (r1 / r0) * p + r
        iters += 1
    dt = time.perf_counter() - t0
    return {"value": iters / dt, "unit": "cg_iterations/s", "cores": 1, "kind": "port", "implementation": "numpy (oracle/libsrps_oracle.so was not built)",
            "sample": f"{iters} CG steps of the matrix-free 2048^2 system (numpy/scipy restatement, 1 thread)"}


def cpu_mitten_solve():
    """BASELINE.json config 1 ("mitten_sf2, CPU reference path on host cores"): the oracle's restatement of the whole alternating
    optimisation (assembled depth system, the reference's order of operations) on the Mitten frame of tests/golden, timed on the
    host.  The reference's MATLAB / .mat inputs are absent; this numpy / scipy port stands in.  Checker code, imported here only."""
    path = os.path.join(ROOT, "tests", "golden", "mitten_full.npz")
    if not os.path.exists(path):
        return None
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import srps_oracle as O
    M = np.load(path)
    h, w, sf = int(M["h"]), int(M["w"]), int(M["sf"])
    mask = np.unpackbits(M["mask_bits"])[: h * w].astype(np.float32)
    mi = np.flatnonzero(mask == 1)
    I = np.zeros((M["I_u8"].shape[0], 3, h * w), np.float32); I[:, :, mi] = M["I_u8"].astype(np.float32) / np.float32(255)
    zs = np.zeros((h // sf) * (w // sf), np.float32); zs[M["imasks"]] = M["zs_lr_masked"]
    zf = np.zeros(h * w, np.float32); zf[mi] = M["z_full_masked"]
    t0 = time.perf_counter()
    ref = O.execute(O.Problem(h, w, sf, mask, M["K"], I, zs, zf), depth="faithful")
    dt = time.perf_counter() - t0
    return {"total_solve_s": dt, "outer_iterations": ref.iterations, "final_energy": float(ref.energies[-1]), "kind": "port",
            "implementation": "numpy / scipy restatement (oracle/srps_oracle.py, assembled depth system)",
            "workload": "the reference's Mitten data set, whole 960x1280 frame, sf 2, 8 images: full alternating solve (compare legs.mitten_full_frame)"}


FP32_VALU_PEAK_TFLOPS = 157.3      # MI355X peak fp32 vector rate, /opt/skills/guides/MI355X_MICROARCH.md (256 CUs x 4 SIMDs x 64 flop/clk x 2.4 GHz)
# useful flops of one CG step per unknown in the executed (tensor-recompute, one-wait) form -- DESIGN.md section 4:
# P = sum_c g_c T_c 36, gradients 2, E'(.) 4, P(.) 15, E(.) 4, scatter 4, KT'KT + lambda 3, three dot products 6,
# p = beta p + r 2, x += alpha p and r -= alpha omega 4, r.r 2
FLOPS_PER_UNKNOWN_STEP = 82
FLOPS_PER_UNKNOWN_RESIDUAL_PASS = 71
# ... of which 36 rebuild the step-INVARIANT tensor P = sum_c g_c T_c every step because six values per unknown have nowhere to live on
# the chip: a CG step that could keep P would need 46 (the round-3 review's "minimal" count); both fractions are reported
FLOPS_PER_UNKNOWN_STEP_MINIMAL = 46
FLOPS_PER_UNKNOWN_RESIDUAL_PASS_MINIMAL = 35


def _profile(name):
    try:
        return json.load(open(os.path.join(ROOT, "profiles", name)))
    except Exception:
        return None


def _newest_profile(suffix):
    """(name, contents) of the newest round's profiles/rNN_<suffix> that parses"""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_" + suffix)), reverse=True):
        d = _profile(os.path.basename(path))
        if d:
            return os.path.basename(path), d
    return None, None


def live_traffic(H, sf, resident, timeout_s=150):
    """HBM-side bytes per launch of the dominant CG kernel, MEASURED BY THIS RUN: two child processes under `rocprofv3 --kernel-trace --pmc`
    (FETCH_SIZE, then WRITE_SIZE: separate passes, as /opt/skills/guides/MI355X_MICROARCH.md prescribes; never another trace domain beside
    them) run tools/cg_prof.py -- a few solves of the same grid with the same kernel -- and the counters of that kernel's launches are
    summed as the guide says for gfx950: 2 x FETCH_SIZE + WRITE_SIZE, in KB.  The children are started by this process and waited for
    (nothing is exec'ed); the profiler's command line has the program itself behind `--`.  Returns None when the profiler is not there
    or a pass fails: the line then replays the committed measurement and says so."""
    import csv
    import glob
    import shutil
    import statistics
    import subprocess
    import tempfile
    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(prof):
        return None
    kernel = "k_cg_resident" if resident else "k_apply_march"
    vals, durs = {}, []
    tmp = tempfile.mkdtemp(prefix="srps_bench_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, counter)
            cmd = [prof, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", out, "--",
                   sys.executable, os.path.join(ROOT, "tools", "cg_prof.py"), str(H), str(sf), "1", "0", "101", "1" if resident else "0"]
            res = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s, cwd="/tmp", env=env)
            files = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)
            # (the profiled child may die in its exit handlers AFTER the profiler has written its tables -- seen as rc -11 on these boxes;
            # what counts is that the table is there and holds the kernel's launches)
            if not files:
                print(f"bench.py: live traffic: the {counter} pass left no counter table (rc {res.returncode}): {res.stderr[-300:]}", file=sys.stderr)
                return None
            v = []
            for r in csv.DictReader(open(files[0])):
                if kernel in r["Kernel_Name"] and r["Counter_Name"] == counter:
                    v.append(float(r["Counter_Value"]))
                    if counter == "FETCH_SIZE":
                        durs.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
            if not v:
                return None
            vals[counter] = statistics.median(v)
        return {"bytes_per_launch": 2 * vals["FETCH_SIZE"] * 1024 + vals["WRITE_SIZE"] * 1024, "fetch_KB": vals["FETCH_SIZE"], "write_KB": vals["WRITE_SIZE"],
                "median_us_under_pmc": statistics.median(durs), "launches": len(durs), "kernel": kernel}
    except Exception as exc:
        print(f"bench.py: live traffic: {exc}", file=sys.stderr)
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def cg_legs(pkg, ctx, H, W, sf, resident_expected, solves=10, live=False):
    """isolated inner loop (srps_bench_cg: HIP events on the launch stream) and the roofline of its dominant kernel"""
    out = {}
    b = ctx.bench_cg(solves=solves, iters=101)
    us_iter = 1e6 * b["seconds"] / b["iterations"]
    out["cg_only_it_per_s"] = b["iterations"] / b["seconds"]
    out["cg_only_us_per_iteration"] = us_iter
    loop_bytes = b["apply_bytes"] + b["update_bytes"]
    P = ctx.dims()["npix"]
    resident = bool(ctx.get_option("cg_resident_active"))
    assert resident == resident_expected or not resident_expected, "the resident CG kernel was expected to run"
    tj_name, tj = _newest_profile("traffic.json")
    tj = tj or {}
    # `traffic` is NOT measured by this process: PMC counters need a profiler pass of their own (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE,
    # tools/profile_round.sh + tools/pmc_traffic.py); the line replays the committed summary of that pass and says so
    traffic_source = (f"replayed from profiles/{tj_name}: rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE (separate passes) of tools/cg_prof.py on another "
                      "run and box, 2 x FETCH_SIZE + WRITE_SIZE per launch, median; not measured by this process") if tj else None
    key = f"{H}x{W}_sf{sf}"
    # the headline leg measures its traffic itself (two profiler passes as child processes, ~10 s each); the side legs replay
    lt = live_traffic(H, sf, resident) if (live and H == W) else None
    live_value = lt["bytes_per_launch"] if lt else None
    if lt:
        traffic_source = (f"measured by this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes, child processes) of tools/cg_prof.py {H} {sf}, "
                          f"median over {lt['launches']} launches of {lt['kernel']}: 2 x {lt['fetch_KB']:.0f} KB + {lt['write_KB']:.0f} KB; {lt['median_us_under_pmc']:.1f} us per launch under the counters")
    if resident:
        # ONE launch runs the residual pass and all 101 steps with the CG state in registers + LDS (kernels_resident.hip).
        # HBM is not what bounds it (the state never leaves the chip): the binding resource is the vector ALU.  `achieved` =
        # useful flops of the launch / its duration against the fp32 vector peak; `issue` prices the instruction stream the
        # compiler actually emitted (tools/isa_loop_report.py) with the measured issue cost per instruction class
        # (tools/valu_issue_bench.hip: 2.3 clocks full rate, 4.5 packed fp32 / v_bfe / DPP, two waves per SIMD).
        launch_us = 1e6 * b["seconds"] / (b["iterations"] / 101)
        flops = float(P) * (101 * FLOPS_PER_UNKNOWN_STEP + FLOPS_PER_UNKNOWN_RESIDUAL_PASS)
        ach = flops / (launch_us * 1e6)                     # TFLOP/s
        isa_name, isa = _newest_profile("resident_isa.json")
        issue = None
        rect = bool(ctx.get_option("cg_resident_rect_active"))
        ikey = key if rect else key + "_general_kernel"
        if isa and ikey in isa:
            clk = isa[ikey]["valu_issue_clocks_per_wave_step"] * isa[ikey]["waves_per_simd"]
            floor_us = clk / (isa[ikey]["GHz"] * 1e3)
            issue = {"kind": "static_estimate: the emitted instruction stream priced with measured issue costs per class, both branches of the direct-sum path counted; not a counter",
                     "valu_issue_clocks_per_simd_step": clk, "issue_floor_us_per_step": floor_us, "kernel": isa[ikey]["kernel"],
                     "measured_us_per_step": launch_us / 101.0, "issue_frac": floor_us / (launch_us / 101.0), "source": isa[ikey]["source"], "report": "profiles/" + isa_name}
        out["roofline"] = {"bound": "valu", "kernel": "k_cg_resident (" + ("mask-free body: every tile lies inside the mask" if rect else "general body") +
                                                      "): residual pass + the whole truncated CG (101 steps) in one persistent launch",
                           "achieved": ach, "peak": FP32_VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / FP32_VALU_PEAK_TFLOPS,
                           "traffic": live_value if live_value is not None else (tj.get(key) or {}).get("resident"), "traffic_source": traffic_source, "avg_launch_us": launch_us, "steps_per_launch": 101,
                           "flops_per_launch": flops, "flops_per_unknown_and_step": FLOPS_PER_UNKNOWN_STEP, "issue": issue,
                           "minimal_flop_count": {"flops_per_unknown_and_step": FLOPS_PER_UNKNOWN_STEP_MINIMAL,
                                                  "frac": float(P) * (101 * FLOPS_PER_UNKNOWN_STEP_MINIMAL + FLOPS_PER_UNKNOWN_RESIDUAL_PASS_MINIMAL) / (launch_us * 1e6) / FP32_VALU_PEAK_TFLOPS,
                                                  "note": "the executed form rebuilds the step-invariant 3 x 3 tensor P = sum_c g_c T_c (36 flops per unknown and step); counted without it"},
                           # what a CG that streams its vectors would have to move for the same work, as a bandwidth -- NOT a roofline fraction
                           "hbm_equivalent": {"algorithmic_bytes_per_launch": loop_bytes * 101, "GBs": loop_bytes * 101 / (1e3 * launch_us),
                                              "x_hbm_peak": loop_bytes * 101 / (1e3 * launch_us) / HBM_PEAK_GBS}}
    else:
        out["cg_loop_roofline"] = {"bound": "hbm", "achieved": loop_bytes / (1e3 * us_iter), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                   "frac": loop_bytes / (1e3 * us_iter) / HBM_PEAK_GBS, "algorithmic_bytes_per_iteration": loop_bytes,
                                   "kernels_per_step": 2 if b["update_bytes"] > 0 else 1}
        # events around the whole 101-step loop give the time per step; an event after EVERY launch (apply_us, update_us) adds
        # ~3 us to each kernel, so it is used only to split the step between the kernels
        share = b["apply_us"] / (b["apply_us"] + b["update_us"]) if b["update_bytes"] > 0 else 1.0      # one launch per step: no update kernel
        apply_us = us_iter * share
        ach = b["apply_bytes"] / (1e3 * apply_us)
        # the CG's working set: g (3) + q-free vectors x, r (2), p (2), omega (2) planes + structure bytes -- at 2048^2 it lies INSIDE the
        # 256 MiB Infinity Cache (MALL), whose hits FETCH_SIZE counts: such a leg prices the cache, not HBM (MI355X_MICROARCH.md)
        ws_bytes = b["apply_bytes"] + b["update_bytes"]          # every byte of a step is touched once: its working set
        in_mall = ws_bytes < 256 * 2**20
        out["roofline"] = {"bound": "infinity_cache" if in_mall else "hbm",
                           "kernel": "k_apply_march: depth operator (p = beta p + r, omega = A_ p, partial dot products, deferred x / r updates)",
                           "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                           "traffic": live_value if live_value is not None else (tj.get(key) or {}).get("apply"), "traffic_source": traffic_source, "avg_launch_us": apply_us,
                           "algorithmic_bytes_per_launch": b["apply_bytes"]}
        if in_mall:
            out["roofline"]["note"] = (f"the step's {ws_bytes / 2**20:.0f} MiB fit the 256 MiB Infinity Cache: `frac` is against the 8 TB/s HBM peak by the contract of this "
                                       "line, but the bytes come from the cache; the HBM-side figure is the 4096^2 leg")
        if b["update_bytes"] > 0:
            out["roofline"]["update_kernel_us"] = us_iter - apply_us
            out["roofline"]["update_kernel_GBs"] = b["update_bytes"] / (1e3 * (us_iter - apply_us))
    return out


def launch_ranks(n_gpus):
    """`bench.py --gpus N` without a launcher around it: the N ranks as children of this process, which has made no GPU call
    (and never replaces itself: an exec from a process that has initialised the GPU takes these machines down)."""
    import socket
    import subprocess
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")            # dmabuf IPC: what RCCL needs on these hosts
    env["SRPS_BENCH_SELF_LAUNCHED"] = "1"
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env, cwd=ROOT)
    lines = 0
    for line in proc.stdout:
        if line.startswith("{"):
            lines += 1
        sys.stdout.write(line); sys.stdout.flush()
    rc = proc.wait()
    if rc == 0 and lines != 1:
        print(f"bench.py: the ranks printed {lines} JSON lines, expected one (rank 0's)", file=sys.stderr)
        rc = 1
    return rc


def select_workload(args):
    """Which of BASELINE.json's workloads a command line means (no GPU call, no torch: tests/test_bench_launcher.py runs it on CPU).
    1 GPU and nothing said: the configuration the metric is quoted on (2048^2, sf 4, 20 images).  N > 1 and nothing said:
    configs[3] -- 2048^2, sf 4, 40 images IN ALL, sharded over the N ranks (8 ranks hold 5 each): strong scaling of a workload
    BASELINE names, not N x 20 images.  --images keeps the weak-scaling form (that many images per GPU)."""
    world = args.gpus
    if args.images is not None and args.images_total is not None:
        sys.exit("bench.py: --images (per GPU) and --images-total exclude each other")
    cfg = args.config
    if cfg is None and world > 1 and args.images is None and args.images_total is None and args.size is None and args.sf is None:
        cfg = 4
    table = {3: (1024, 4, 20, "configs[2]"), 4: (2048, 4, 40, "configs[3]"), 5: (4096, 2, 64, "configs[4]")}
    named = None
    if cfg is not None:
        size, sf, total, named = table[cfg]
        args.size = args.size or size
        args.sf = args.sf or sf
        if args.images is None and args.images_total is None:
            args.images_total = total
        if cfg == 5 and args.partition is None and world > 1:
            args.partition = "strips"
    args.size = args.size or 2048
    args.sf = args.sf or 4
    args.partition = args.partition or "images"
    if args.images is not None:
        args.images_total = args.images * world
        args.scaling = "weak"
        how = f"{args.images} images/GPU x {world} GPU"
    else:
        if args.images_total is None:
            args.images_total = 20
        args.scaling = "strong" if world > 1 else "weak"          # one GPU: nothing is scaled; the field keeps the metric line's value
        args.images = None if world > 1 else args.images_total
        base, rem = divmod(args.images_total, world)                # api.shard_range: the first `rem` ranks hold one more
        how = f"{args.images_total} images" + (f" sharded over {world} GPUs ({', '.join(str(base + (r < rem)) for r in range(world))} per rank)" if world > 1 else "/GPU x 1 GPU")
    args.workload = f"synthetic full-mask HR grid {args.size}x{args.size}, sf {args.sf}, {how}, 3 channels"
    if named and (args.size, args.sf, args.images_total) == table[cfg][:3]:
        args.workload += f" [BASELINE.json {named}]"
    elif world == 1 and (args.size, args.sf, args.images_total) == (2048, 4, 20):
        args.workload += " [BASELINE.json metric configuration]"
    return args


def describe_parallelism(world, partition, comm_requested, comm_kind, resident_strips_active, strips_active, shared_gpu):
    """What ran, and what it fell back from (no GPU call: tests/test_bench_launcher.py runs it on CPU).  The chain of a multi-GPU job, every
    link decided by ALL ranks together inside the same run (nothing is exec'ed, no rank goes its own way):
        communicator of the library (RCCL)  --fails-->  torch.distributed collectives on the library's exchange buffers
        depth CG as the resident kernel on strips (cg_partition = 2)  --handshake fails / grid does not fit-->  streaming strips (4-double
        all-reduce + edge exchange per step)  --no transport / refused-->  replicated CG (every rank runs the 101 steps)
    Returns (text, degraded): `degraded` lists the links that were asked for and not taken."""
    if world == 1:
        return "1 GPU", []
    degraded = []
    if comm_requested == "library" and comm_kind == "torch":
        degraded.append("library communicator (RCCL inside libsrps_hip.so) -> torch.distributed collectives")
    if partition == "strips":
        if comm_kind == "torch":
            degraded.append("--partition strips needs the library's communicator -> replicated CG")
        elif not resident_strips_active:
            degraded.append("resident kernel on strips (cg_partition = 2) -> " + ("streaming strips" if strips_active else "replicated CG"))
    if resident_strips_active:
        cg = "depth CG as the resident kernel on strips of tile columns (sums and border edges through hipIpc-mapped buffers, no collective between the steps)"
    elif strips_active:
        cg = "depth CG partitioned into column strips (4-double all-reduce + edge-column exchange per step)"
    else:
        cg = "replicated CG"
    how = {"library": "RCCL all-reduce of partial sums", "hosted": "all-reduce of partial sums by host collectives under the library's loop",
           "torch": "torch.distributed all-reduce of partial sums"}.get(comm_kind, "all-reduce of partial sums")
    text = f"images sharded over {world} ranks, {how}, {cg}" + (" [dry run: ranks share one GPU, gloo]" if shared_gpu else "")
    return text, degraded


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--size", type=int, default=None, help="HR grid is size x size (default 2048; --config 5: 4096)")
    ap.add_argument("--sf", type=int, default=None, help="scale factor (default 4; --config 5: 2)")
    ap.add_argument("--images", type=int, default=None, help="images per GPU: weak scaling in images (N GPUs hold N x this many)")
    ap.add_argument("--images-total", type=int, default=None, help="images in all, sharded over the ranks: strong scaling")
    ap.add_argument("--config", type=int, choices=[3, 4, 5], default=None,
                    help="a workload of BASELINE.json by its number there (1-based): 3 = 1024^2 sf 4 x 20 images, 4 = 2048^2 sf 4 x 40 images sharded over "
                         "the ranks, 5 = 4096^2 sf 2 x 64 images sharded over the ranks (with --partition strips unless said otherwise).  Default: "
                         "1 GPU: the metric's configuration (2048^2 sf 4 x 20 images); N > 1: config 4")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-total-solve", action="store_true")
    ap.add_argument("--no-live-traffic", action="store_true", help="do not measure roofline.traffic with two rocprofv3 --pmc child runs (~20 s); replay the committed profile instead")
    ap.add_argument("--no-legs", action="store_true", help="skip the streaming-CG legs (2048^2 streaming, 4096^2 sf 2)")
    ap.add_argument("--apply-mode", type=int, default=0)
    ap.add_argument("--option", action="append", default=[], metavar="NAME=INT", help="srps_set_option before setup (A/B runs)")
    ap.add_argument("--partition", choices=["images", "strips"], default=None,
                    help="N > 1: 'images' shards the images and replicates the depth CG (default); 'strips' also partitions the depth CG into "
                         "column strips over the ranks (option cg_partition; needs --comm library) -- meant for --size 4096 --sf 2")
    ap.add_argument("--comm", choices=["library", "torch"], default="library",
                    help="N > 1: the all-reduces of a pass as ncclAllReduce inside libsrps_hip.so (srps_execute_sharded; default) or as "
                         "torch.distributed.all_reduce on views of the library's exchange buffers")
    args = ap.parse_args()
    assert args.gpus >= 1, "--gpus must be at least 1"
    select_workload(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus))

    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s) (WORLD_SIZE)")
    # Dry-run hook for a one-GPU box (tests/test_gpu_distributed.py::test_bench_two_ranks_on_one_gpu): all ranks share device 0 and
    # the process group is gloo, because RCCL refuses two ranks on one device.  Never set on a multi-GPU node.
    shared_gpu = os.environ.get("SRPS_BENCH_SHARED_GPU") == "1"
    if shared_gpu:
        local_rank = 0
    elif torch.cuda.device_count() < world:                       # counting devices initialises nothing
        sys.exit(f"bench.py: --gpus {world} but this node shows {torch.cuda.device_count()} GPU(s)")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if shared_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    pkg = importlib.import_module("srmeetsps-cuda_amd")
    pkg.load()                                       # no fallback: raises when the extension is missing
    H = W = args.size
    n_total = args.images_total
    lo, hi = pkg.shard_range(n_total, world, rank)
    sc = pkg.synth.make_scene(H, W, args.sf, n_total, seed=1234 + 3, mask_kind="full", img_begin=lo, img_end=hi)
    dh = pkg.DataHandler.from_scene(sc)
    ctx = pkg.Context(device_id=local_rank)
    # one explicit torch stream carries the library's kernels and the collectives' dependencies (N > 1)
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    ctx.set_stream(stream.cuda_stream)
    # the bench box is exclusive to this job (one rank per GPU), said explicitly: plain launches of the persistent kernels
    if not shared_gpu:
        ctx.set_option("exclusive_device", 1)
    if args.apply_mode:
        ctx.set_option("apply_mode", args.apply_mode)
    for kv in args.option:
        name, val = kv.split("=")
        ctx.set_option(name, int(val))
    # N > 1: RCCL behind the C ABI -- rank 0 makes the communicator id (ncclGetUniqueId through the library), torch.distributed only
    # carries its 128 bytes to the other ranks (and the timing barrier); every rank joins with its context.  Should that fail on
    # any rank (no librccl), all ranks fall back to torch.distributed's collectives together.
    comm_kind = "none"
    hosted = None
    if world > 1:
        comm_kind = "torch"
        if args.comm == "library" and shared_gpu:
            # the dry run on one GPU: the library's own sharded loop (srps_execute_sharded) with torch.distributed (gloo) UNDER it as the
            # context's collectives (srps_set_host_collectives) -- the code path of the RCCL run, with host functions for the collectives
            hosted = pkg.TorchCollectives(ctx, dist)
            ctx.set_option("spin_budget_ms", 2000)         # the ranks' persistent kernels share the device and start milliseconds apart
            comm_kind = "hosted"
        if args.comm == "library" and not shared_gpu:
            ok = 1
            try:
                uid = [pkg.Context.comm_unique_id() if rank == 0 else None]
                dist.broadcast_object_list(uid, src=0)
                ctx.comm_init_rank(uid[0], rank, world)
            except Exception as exc:
                print(f"bench.py rank {rank}: library communicator failed ({exc}); torch.distributed collectives instead", file=sys.stderr)
                ok = 0
            flag = torch.tensor([ok], device="cuda", dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 1:
                comm_kind = "library"
            elif ok:
                ctx.comm_release()
    if args.partition == "strips" and comm_kind in ("library", "hosted"):
        # 2: the resident kernel on every rank's strip of tile columns where the strips fit (<= one 256 x 64 tile per CU and rank), the ranks'
        # kernels talking through hipIpc-mapped exchange buffers; the library falls back to 1 (streaming strips, collectives per step)
        ctx.set_option("cg_partition", 2)
    ctx.setup(dh)
    dims = ctx.dims()

    def all_reduce(t):
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    ar = all_reduce if (world > 1 and comm_kind == "torch") else None

    def solve(max_outer=None):
        if comm_kind in ("library", "hosted"):
            return ctx.execute_sharded(max_outer or 0)
        return pkg.alternating_loop(ctx, ar, max_outer=max_outer)

    def step():
        return solve(1)[0]

    for _ in range(args.warmup):
        step()
    if dist: dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    energies = [step() for _ in range(args.steps)]
    torch.cuda.synchronize()
    if dist: dist.barrier()
    dt = time.perf_counter() - t0
    if dist:
        tt = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    # the rate is that of the steps the device executed: a CG that stopped early or a pass that fell back to the streaming kernels
    # would not be the workload named below
    depth_steps = ctx.last_cg_iterations()["depth"]
    fallbacks = ctx.get_option("persistent_fallbacks")
    assert depth_steps == 101, f"the depth CG ran {depth_steps} steps, not the 101 of devicecalls.cu:252"
    cg_iters = depth_steps * args.steps
    par_text, par_degraded = describe_parallelism(world, args.partition, args.comm, comm_kind, bool(ctx.get_option("cg_partition_resident_active")),
                                                  bool(ctx.get_option("cg_partition_active")), shared_gpu)
    for d in par_degraded:
        if rank == 0:
            print(f"bench.py: fell back: {d}", file=sys.stderr)
    out = {
        "metric": "cg_iterations_per_sec", "value": cg_iters / dt, "unit": "cg_iterations/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
        "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": args.workload,
                   "hr_grid": [H, W], "sf": args.sf, "images_per_gpu": args.images if args.images is not None else max(pkg.shard_range(n_total, world, r)[1] - pkg.shard_range(n_total, world, r)[0] for r in range(world)),
                   "images_total": n_total,
                   "unknowns": dims["npix"], "cg_steps_per_solve": depth_steps, "persistent_fallbacks": fallbacks,
                   "albedo_mode": {0: "SRPS_ALBEDO_CG", 1: "SRPS_ALBEDO_CLOSED_FORM", 2: "SRPS_ALBEDO_FUSED", 3: "SRPS_ALBEDO_AUTO (pipeline: the albedo CG's fixed point formed inside the sweep)"}[ctx.get_option("albedo_mode")],
                   "exclusive_device": ctx.get_option("exclusive_device"),       # 1: plain launches of the persistent kernels (the library's default is the cooperative launch: ~26 us per pass more)
                   "comm": {"none": "none (1 GPU)", "library": "ncclAllReduce inside libsrps_hip.so (srps_execute_sharded), communicator from srps_comm_init_rank",
                            "hosted": "srps_execute_sharded with torch.distributed (gloo) under it as the context's collectives (srps_set_host_collectives) [dry run]",
                            "torch": "torch.distributed.all_reduce on views of the library's exchange buffers"}[comm_kind],
                   # what the LIBRARY's communicator says of itself (ncclCommCount through srps_comm_info; 0: none bound) and who holds what
                   "ncclCommCount": ctx.comm_info()[1] if comm_kind == "library" else 0,
                   "ranks_seen_by_the_library": ctx.comm_info()[1] if comm_kind in ("library", "hosted") else 0,
                   "partition": args.partition if world > 1 else "none",
                   "images_per_rank": [pkg.shard_range(n_total, world, r)[1] - pkg.shard_range(n_total, world, r)[0] for r in range(world)],
                   "launched_by": "bench.py itself (child torch.distributed.run)" if os.environ.get("SRPS_BENCH_SELF_LAUNCHED") == "1" else ("torch.distributed.run" if "WORLD_SIZE" in os.environ else "single process"),
                   "parallelism": par_text, "degraded": par_degraded,
                   "forced_failures": os.environ.get("SRPS_FORCE_FAIL") or None},
        "energies": energies,
    }
    if not args.no_total_solve:
        # full solve to the reference's stop rule (SRPS.cu:297-302), from a fresh set-up; reported with and without srps_setup.  Timed HERE, right
        # behind the timed passes and before the side legs: behind them (a dozen contexts made and closed, gigabytes allocated and freed, the
        # host's CPU quota spent) the same set-up measured 1.0 - 1.6 ms longer (gpurun_out/r5g: 20.1 ms against 21.1 - 21.7)
        # (= SRPS.cu:100-270: the upload of the images from host memory, compaction, first normals)
        # Three runs, the MEDIAN reported (all three listed): the set-up is a 1 GB host-to-device transfer fed by host threads under a
        # CPU quota, and one run in five on these boxes is 6 - 10 ms off (28.0 against 20.4 ms in two back-to-back bench runs, gpurun_out/r5h)
        runs = []
        for _ in range(3):
            if dist: dist.barrier()
            torch.cuda.synchronize()
            t_setup0 = time.perf_counter()
            ctx.setup(dh)
            torch.cuda.synchronize()
            if dist: dist.barrier()
            t0 = time.perf_counter()
            en = solve()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            runs.append({"total_solve_s": t1 - t0, "total_solve_with_setup_s": t1 - t_setup0, "setup_s": t0 - t_setup0, "outer_iterations": len(en)})
        if rank == 0:
            mid = sorted(runs, key=lambda r: r["total_solve_with_setup_s"])[1]
            out["total_solve_s"] = mid["total_solve_s"]
            out["total_solve_with_setup_s"] = mid["total_solve_with_setup_s"]
            out["setup_s"] = mid["setup_s"]
            out["setup_host_bytes"] = int(dh.I.nbytes)
            out["total_solve_outer_iterations"] = mid["outer_iterations"]
            out["total_solve_runs"] = {"reported": "the median of three by total_solve_with_setup_s", "runs": runs}
    # the isolated CG loop: with the depth CG partitioned over the ranks (strips) a solve is a collective -- every rank takes part, rank 0 reports
    cg_collective = world > 1 and (ctx.get_option("cg_partition_active") == 1 or ctx.get_option("cg_partition_resident_active") == 1)
    if rank == 0 or cg_collective:
        # the resident kernel runs where the grid has at most one 256 x 64 tile per CU (2048 x 2048 on 256 CUs); it must then
        # really have run -- a persistent launch that gave up a wait would have switched the context to the streaming kernels
        tiles = -(-dims["grid_h"] // 256) * -(-dims["grid_w"] // 64)
        legs_cg = cg_legs(pkg, ctx, H, W, args.sf, resident_expected=(tiles <= ctx.get_option("num_cus")) and not cg_collective,
                          live=(world == 1 and not args.no_live_traffic))
        if rank == 0:
            out.update(legs_cg)
    if rank == 0:
        if not args.no_legs and world == 1:
            # the two HBM-bound legs of north_star, driver-timed with the headline: the streaming CG on the metric's grid
            # (">= 60 % of the HBM roofline on the CG SpMV + axpy loop at 2048 x 2048") and on the largest single-GPU grid
            # of BASELINE.json's configs (4096 x 4096, sf 2: four times the tiles the chip has CUs)
            legs = {}
            c2 = pkg.Context(device_id=local_rank)
            c2.set_stream(stream.cuda_stream)
            c2.set_option("cg_resident", 0)
            c2.set_option("exclusive_device", 1)
            c2.setup(dh)
            pkg.alternating_loop(c2, None, max_outer=1)
            legs["streaming_2048"] = dict(cg_legs(pkg, c2, H, W, args.sf, resident_expected=False, solves=5),
                                          workload=f"depth CG of the headline workload with the streaming kernels (cg_resident=0), {H}x{W}, sf {args.sf}")
            c2.close()
            sc4 = pkg.synth.make_scene(4096, 4096, 2, 2, seed=1234 + 5, mask_kind="full")
            c4 = pkg.Context(device_id=local_rank)
            c4.set_stream(stream.cuda_stream)
            c4.set_option("exclusive_device", 1)
            c4.setup(pkg.DataHandler.from_scene(sc4))
            pkg.alternating_loop(c4, None, max_outer=1)
            legs["largest_grid_4096_sf2"] = dict(cg_legs(pkg, c4, 4096, 4096, 2, resident_expected=False, solves=3),
                                                 workload="depth CG on a synthetic full-mask 4096x4096 HR grid, sf 2 (16.8 M unknowns; 2 images: the CG does not depend on their number)")
            c4.close()
            # The strip-partitioned CG (srps_strips.hip) with its ranks as 8 contexts on this one device (device copies instead of
            # RCCL): the group's time / 8 is what ONE rank computes per step on its eighth of the columns -- the compute side of the
            # per-step budget of an 8-GPU run; its communication side (a 32-byte all-reduce and a 48 KB neighbour exchange per step)
            # cannot be measured here.
            try:
                group = []
                for _ in range(8):
                    cs = pkg.Context(device_id=local_rank)
                    cs.set_stream(stream.cuda_stream)
                    cs.set_option("exclusive_device", 1)
                    cs.setup(pkg.DataHandler.from_scene(sc4))
                    cs.lighting(); cs.albedo(); cs.depth_partial()
                    group.append(cs)
                pkg.Context.strip_group_solve(group)
                for cs in group:
                    cs.depth_partial()
                torch.cuda.synchronize()
                tg = time.perf_counter()
                pkg.Context.strip_group_solve(group)
                torch.cuda.synchronize()
                dg = time.perf_counter() - tg
                legs["strip_cg_4096_sf2_8_ranks_on_one_device"] = {
                    "group_solve_s": dg, "us_per_step_and_rank": 1e6 * dg / (101 * 8), "ranks": 8, "columns_per_rank": 512,
                    "depth_steps": group[0].last_cg_iterations()["depth"],
                    "workload": "101 CG steps of the 4096x4096, sf 2 system as 8 column strips, all 8 ranks on this one device in lockstep (device copies for the collectives): time / 8 = one rank's compute per step"}
                for cs in group:
                    cs.close()
            except Exception as exc:                      # a side measurement: never fail the bench line because of it
                legs["strip_cg_4096_sf2_8_ranks_on_one_device"] = {"error": str(exc)}
            del sc4
            # The RESIDENT kernel on strips (round 4): the headline's grid as two launches of 128 blocks each, side by side on this one
            # device, exchanging their sums and border edges through each other's memory -- the multi-GPU form of the kernel with ordinary
            # device memory in the place of peer memory.  Host clock around the whole group solve (events, synchronisations and the gather
            # of the strips included): an upper bound of the step; the depth must be the single launch's, bit for bit.
            try:
                def prep():
                    cq = pkg.Context(device_id=local_rank)         # its OWN stream: the launches must run side by side
                    cq.set_option("exclusive_device", 1); cq.set_option("cg_resident_tile", 512); cq.set_option("spin_budget_ms", 1000)
                    cq.setup(dh); cq.lighting(); cq.albedo(); cq.depth_partial(); cq.synchronize()
                    return cq
                one = prep(); one.depth_solve(); one.synchronize(); z_one = one.get("z"); one.close()
                grp = [prep() for _ in range(2)]
                pkg.Context.strip_group_solve_resident(grp)
                same = all(np.array_equal(cq.get("z"), z_one) for cq in grp)
                tsg = []
                for _ in range(5):
                    for cq in grp:
                        cq.depth_partial(); cq.synchronize()
                    tg0 = time.perf_counter(); pkg.Context.strip_group_solve_resident(grp); tsg.append(time.perf_counter() - tg0)
                legs["resident_strips_2048_2_ranks_on_one_device"] = {
                    "group_solve_ms": 1e3 * sorted(tsg)[len(tsg) // 2], "us_per_step_upper_bound": 1e6 * sorted(tsg)[len(tsg) // 2] / 102,
                    "bit_identical_to_the_single_launch": bool(same), "ranks": 2, "blocks_per_rank": 128,
                    "workload": f"depth CG of the headline workload ({H}x{W}, sf {args.sf}) as two resident launches on two strips of tile columns, both on this device (srps_strip_group_solve_resident)"}
                for cq in grp:
                    cq.close()
            except Exception as exc:                      # a side measurement: never fail the bench line because of it
                legs["resident_strips_2048_2_ranks_on_one_device"] = {"error": str(exc)}
            # the headline's mask is the best case of the resident kernel (every tile inside the mask: the body without structure
            # bits); any other mask -- the reference's own data -- takes the general body.  An ellipse in the same frame:
            scg = pkg.synth.make_scene(H, W, args.sf, 2, seed=1234 + 6, mask_kind="ellipse")
            cg_ = pkg.Context(device_id=local_rank)
            cg_.set_stream(stream.cuda_stream)
            cg_.set_option("exclusive_device", 1)
            cg_.setup(pkg.DataHandler.from_scene(scg))
            pkg.alternating_loop(cg_, None, max_outer=1)
            gd = cg_.dims()
            legs["general_mask_2048"] = dict(cg_legs(pkg, cg_, H, W, args.sf, resident_expected=False, solves=5),
                                             resident=cg_.get_option("cg_resident_active"), rect_body=cg_.get_option("cg_resident_rect_active"),
                                             unknowns=gd["npix"], grid=[gd["grid_h"], gd["grid_w"]], persistent_fallbacks=cg_.get_option("persistent_fallbacks"),
                                             workload=f"depth CG on an elliptical mask (semi-axes 0.45 h x 0.45 w) in the {H}x{W} frame, sf {args.sf}: the resident kernel's general body")
            cg_.close()
            del scg
            # The headline workload with its images rounded to 8 bits, i.e. as the reference's image-folder loader delivers them
            # (k / 255.f, Utilities.cpp:343): the library then also keeps them as bytes and the two image sweeps of a pass read a
            # quarter of the bytes (option "image_store"; results identical bit for bit to the float store on the same input).
            # Reported next to the headline, which stays on the float-valued images of SURVEY section 8's recipe.
            I_keep = sc.I
            sc.I = (np.rint(np.clip(I_keep, 0, 1) * 255).astype(np.float32) / np.float32(255)).astype(np.float32)
            c8 = pkg.Context(device_id=local_rank)
            c8.set_stream(stream.cuda_stream)
            c8.set_option("exclusive_device", 1)
            c8.setup(pkg.DataHandler.from_scene(sc))
            for _ in range(max(args.warmup, 1)):
                pkg.alternating_loop(c8, None, max_outer=1)
            torch.cuda.synchronize()
            t8 = time.perf_counter()
            for _ in range(args.steps):
                pkg.alternating_loop(c8, None, max_outer=1)
            torch.cuda.synchronize()
            d8 = time.perf_counter() - t8
            it8 = c8.last_cg_iterations()["depth"]
            assert it8 == 101 and c8.get_option("persistent_fallbacks") == 0, (it8, c8.get_option("persistent_fallbacks"))
            legs["images_8bit"] = {"cg_iterations_per_sec": it8 * args.steps / d8, "ms_per_step": 1e3 * d8 / args.steps,
                                   "image_store_bytes_active": c8.get_option("image_store_bytes_active"),
                                   "workload": "the headline workload with 8-bit images (k / 255.f, the reference's image-folder input): image sweeps read bytes"}
            c8.close()
            sc.I = I_keep
            # The headline workload with the reference's albedo CG in the pipeline (albedo_mode = SRPS_ALBEDO_CG; the default since round 4
            # is the CG's fixed point formed inside the albedo sweep, SRPS_ALBEDO_AUTO -- include/srps.h has the measurements behind that):
            # num / den / image-sum planes, the persistent albedo CG, the depth assembly from the sums.
            def timed_passes(options):
                cf = pkg.Context(device_id=local_rank)
                cf.set_stream(stream.cuda_stream)
                cf.set_option("exclusive_device", 1)
                for name, val in options.items():
                    cf.set_option(name, val)
                cf.setup(dh)
                for _ in range(max(args.warmup, 1)):
                    pkg.alternating_loop(cf, None, max_outer=1)
                torch.cuda.synchronize()
                tf = time.perf_counter()
                for _ in range(args.steps):
                    pkg.alternating_loop(cf, None, max_outer=1)
                torch.cuda.synchronize()
                df = time.perf_counter() - tf
                itf = cf.last_cg_iterations()
                assert itf["depth"] == 101 and cf.get_option("persistent_fallbacks") == 0
                cgb = cf.bench_cg(solves=5, iters=101)
                res = {"cg_iterations_per_sec": itf["depth"] * args.steps / df, "ms_per_step": 1e3 * df / args.steps, "albedo_cg_steps": list(itf["albedo"][:3]),
                       "cg_only_us_per_iteration": 1e6 * cgb["seconds"] / cgb["iterations"]}
                cf.setup(dh)
                torch.cuda.synchronize()
                ts = time.perf_counter(); en_f = pkg.alternating_loop(cf, None); torch.cuda.synchronize()
                res["total_solve_s"] = time.perf_counter() - ts
                res["total_solve_outer_iterations"] = len(en_f)
                cf.close()
                return res
            legs["albedo_reference_cg"] = dict(timed_passes({"albedo_mode": 0}),
                                               workload="the headline workload with albedo_mode = SRPS_ALBEDO_CG (the reference's CG on the diagonal albedo system in the pipeline; not the default)")
            # The price of the reference's exact arithmetic: BOTH documented departures switched off -- the albedo by the reference's CG
            # (dc.cu:513-548) instead of its fixed point formed in the sweep, and r.r summed directly in every CG step (dc.cu:274)
            # instead of predicted from the step's three sums (a second grid-wide wait per step).  Same workload, same timing.
            legs["reference_arithmetic"] = dict(timed_passes({"albedo_mode": 0, "cg_one_sync": 0}),
                                                workload="the headline workload with albedo_mode = SRPS_ALBEDO_CG AND cg_one_sync = 0: every sum and every solve as devicecalls.cu has them (not the default)",
                                                headline_ms_per_step=out["ms_per_step"], price_of_exactness=None)
            legs["reference_arithmetic"]["price_of_exactness"] = legs["reference_arithmetic"]["ms_per_step"] / out["ms_per_step"]
            mitten = os.path.join(ROOT, "tests", "golden", "mitten_full.npz")
            mitten20 = os.path.join(ROOT, "tests", "golden", "mitten_full_20.npz")

            def mitten_leg(n_images):
                # BASELINE.json config 2 at its true size: the whole frame of the reference's bundled Mitten data set (960 x 1280,
                # sf 2, 148 600 masked pixels; tests/golden/mitten_full.npz holds the masked samples of the first 8 images,
                # mitten_full_20.npz those of the other 12), full alternating solve to the reference's stop rule -- parity against
                # the oracle: tests/test_mitten_full.py.  20 images is what `srps -t images -d dataset/Images/Mitten` solves
                # (cv::glob of RGB/, Utilities.cpp:349-352); 8 is BASELINE.json's "~8 images".
                M = np.load(mitten)
                X = np.load(mitten20) if n_images == 20 else M
                bytes_ = M["I_u8"] if n_images == 8 else np.concatenate([M["I_u8"], X["I_u8_9_to_20"]])
                mh, mw, msf = int(M["h"]), int(M["w"]), int(M["sf"])
                mmask = np.unpackbits(M["mask_bits"])[: mh * mw].astype(np.float32)
                mi = np.flatnonzero(mmask == 1)
                mI = np.zeros((n_images, 3, mh * mw), np.float32); mI[:, :, mi] = bytes_.astype(np.float32) / np.float32(255)
                mzs = np.zeros((mh // msf) * (mw // msf), np.float32); mzs[M["imasks"]] = M["zs_lr_masked"]
                mzf = np.zeros(mh * mw, np.float32); mzf[mi] = M["z_full_masked"]
                mdh = pkg.DataHandler(I=mI, mask=mmask, K=M["K"], sf=msf, z0=mzs.reshape(1, -1), I_h=mh, I_w=mw, I_c=3, I_n=n_images,
                                      I_n_total=n_images, zs_lr=mzs, z_full=mzf)
                cm = pkg.Context(device_id=local_rank)
                cm.set_stream(stream.cuda_stream)
                cm.set_option("exclusive_device", 1)
                cm.setup(mdh); pkg.alternating_loop(cm, None)              # warm-up (first launches, allocations)
                torch.cuda.synchronize()
                tm0 = time.perf_counter(); cm.setup(mdh); torch.cuda.synchronize()
                tm1 = time.perf_counter(); men = pkg.alternating_loop(cm, None); torch.cuda.synchronize()
                tm2 = time.perf_counter()
                res = {"total_solve_s": tm2 - tm1, "total_solve_with_setup_s": tm2 - tm0, "outer_iterations": len(men),
                       "final_energy": men[-1], "oracle_final_energy": float(X["energies"][-1]), "masked_pixels": int(mi.size),
                       "images": n_images, "image_store_bytes_active": cm.get_option("image_store_bytes_active"),
                       "workload": f"the reference's Mitten data set, whole 960x1280 frame, sf 2, {n_images} images: full alternating solve to its stop rule"}
                cm.close()
                return res
            if os.path.exists(mitten):
                legs["mitten_full_frame"] = mitten_leg(8)
                if os.path.exists(mitten20):
                    legs["mitten_full_frame_20_images"] = dict(mitten_leg(20), note="all 20 images of the folder: what the reference's CLI solves on its bundled data set (Utilities.cpp:349-352)")
            # what a bare stream of the CG step's shape reaches on THIS box (tools/hbm_ceiling_bench.hip, a child process; the pool's
            # boxes differ by 15 %): the streaming legs' fraction of that, next to their fraction of the 8 TB/s peak
            ceil_bin = os.path.join(ROOT, "tools", "hbm_ceiling_bench.bin")
            if os.path.exists(ceil_bin):
                try:
                    import subprocess
                    torch.cuda.synchronize()
                    res = subprocess.run([ceil_bin, "4096", "4096", "10", "1"], capture_output=True, text=True, timeout=300)
                    rows = [json.loads(ln) for ln in res.stdout.splitlines() if ln.startswith("{")]
                    best = {}
                    for r in rows:
                        v = r["variant"]
                        name = v.rsplit("_", 1)[0] if v.endswith("blk") else v         # best over the block counts tried
                        best[name] = max(best.get(name, 0.0), r["GBs"])
                    fam = lambda pre: max([g for n, g in best.items() if n.startswith(pre)] or [0.0])
                    legs["hbm_ceiling_this_box"] = {"GBs": best, "best_read_GBs": fam("guide_read"), "best_copy_GBs": fam("guide_copy"),
                                                    "workload": "tools/hbm_ceiling_bench.bin 4096 4096 10 1: float4 streams (grid-stride and block-contiguous, default and non-temporal policy, 1 - 8 loads in flight), the CG step's access shape (8 planes read, 4 written, marching) and the image sweeps' shape (60 planes / tile-major)"}
                    march = best.get("march_planar_8r_4w")
                    if march:
                        for key in ("largest_grid_4096_sf2",):
                            if key in legs and "roofline" in legs[key]:
                                legs[key]["roofline"]["frac_of_measured_ceiling"] = legs[key]["roofline"]["achieved"] / march
                                legs[key]["roofline"]["measured_ceiling_GBs"] = march
                except Exception as exc:
                    legs["hbm_ceiling_this_box"] = {"error": str(exc)}
            out["legs"] = legs
    if rank == 0:
        # measured device-copy ceiling: 1 GiB device-to-device copy, read + write bytes over the event time
        try:
            n = 256 * 1024 * 1024
            src = torch.empty(n, dtype=torch.float32, device="cuda"); dst = torch.empty_like(src)
            src.fill_(1.0); dst.copy_(src)
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                dst.copy_(src)
            e1.record(); e1.synchronize()
            out["device_copy_GBs"] = 5 * 2 * 4 * n / (e0.elapsed_time(e1) * 1e6)
            del src, dst
        except Exception as exc:                      # never fail the bench line because of the side measurement
            out["device_copy_GBs"] = None
            out["device_copy_error"] = str(exc)
    if rank == 0:
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(sc, pkg)
            mit = cpu_mitten_solve()
            if mit:
                out["cpu_baseline"]["mitten_full_frame"] = mit
        print(json.dumps(out))
    if hosted is not None:
        assert not hosted.errors, hosted.errors
        hosted.remove()
    ctx.close()
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
