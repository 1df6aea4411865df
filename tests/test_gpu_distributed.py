"""Two processes, one GPU: the real distributed driver (pkg.SRPS(distributed=True): shard_range, the
phase-split C-ABI entry points on the torch stream, torch.distributed.all_reduce on zero-copy views of
the library's exchange buffers) against the single-process result.  The process group uses gloo here
because RCCL refuses two ranks on one device; on a multi-GPU node bench.py runs the same code over nccl."""
import importlib
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, n_img, out_dir):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pkg = importlib.import_module("srmeetsps-cuda_amd")
    lo, hi = pkg.shard_range(n_img, world, rank)
    sc = pkg.synth.make_scene(48, 56, 2, n_img, seed=33, mask_kind="ragged", img_begin=lo, img_end=hi)
    srps = pkg.SRPS(pkg.DataHandler.from_scene(sc), distributed=True)
    en = srps.execute(max_outer=3)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), energies=np.array(en), z=srps.z(), rho=srps.rho(), s=srps.s())
    dist.barrier()
    srps.ctx.close()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_two_ranks_on_one_gpu_equal_single_process(tmp_path, pkg):
    import torch.multiprocessing as mp
    n_img = 5
    mp.spawn(_worker, args=(2, _free_port(), n_img, str(tmp_path)), nprocs=2, join=True)
    r0 = np.load(tmp_path / "rank0.npz"); r1 = np.load(tmp_path / "rank1.npz")
    np.testing.assert_array_equal(r0["z"], r1["z"]); np.testing.assert_array_equal(r0["s"], r1["s"])     # replicas stay bit-identical
    np.testing.assert_array_equal(r0["energies"], r1["energies"])
    sc = pkg.synth.make_scene(48, 56, 2, n_img, seed=33, mask_kind="ragged")
    ctx = pkg.Context(device_id=0)
    one = pkg.SRPS(pkg.DataHandler.from_scene(sc), ctx=ctx)
    e1 = one.execute(max_outer=3)
    np.testing.assert_allclose(r0["energies"], e1, rtol=5e-4)
    assert np.sqrt(np.mean((r0["z"] - one.z()) ** 2)) < 3e-5
    assert np.abs(r0["rho"] - one.rho()).max() < 5e-4
    ctx.close()


def _nccl_single(rank, port, out_dir):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    pkg = importlib.import_module("srmeetsps-cuda_amd")
    sc = pkg.synth.make_scene(48, 56, 2, 4, seed=35, mask_kind="ragged")
    srps = pkg.SRPS(pkg.DataHandler.from_scene(sc), distributed=True)
    en = srps.execute(max_outer=2)
    np.savez(os.path.join(out_dir, "nccl.npz"), energies=np.array(en), z=srps.z())
    srps.ctx.close()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_rccl_backend_accepts_the_exchange_buffers(tmp_path, pkg):
    """backend "nccl" (= RCCL) with one rank: the collectives run on zero-copy views of memory the library
    allocated with hipMalloc (not torch's caching allocator), on the explicit torch stream the kernels use"""
    import torch.multiprocessing as mp
    mp.spawn(_nccl_single, args=(_free_port(), str(tmp_path)), nprocs=1, join=True)
    r = np.load(tmp_path / "nccl.npz")
    sc = pkg.synth.make_scene(48, 56, 2, 4, seed=35, mask_kind="ragged")
    ctx = pkg.Context(device_id=0)
    one = pkg.SRPS(pkg.DataHandler.from_scene(sc), ctx=ctx)
    e1 = one.execute(max_outer=2)
    np.testing.assert_array_equal(r["energies"], np.array(e1))
    np.testing.assert_array_equal(r["z"], one.z())
    ctx.close()


@pytest.mark.timeout(900)
def test_bench_two_ranks_on_one_gpu():
    """bench.py exactly as the driver launches it for N > 1 (torch.distributed.run, one process per rank), dry run: both
    ranks share device 0 over gloo (SRPS_BENCH_SHARED_GPU).  A small grid, so that the persistent kernels of the two
    processes fit on the device side by side.  Checks the contract line and that sharding 2 x 3 images changes nothing."""
    import json
    import subprocess
    env = dict(os.environ, SRPS_BENCH_SHARED_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    common = ["--steps", "2", "--warmup", "1", "--size", "256", "--sf", "2", "--no-cpu-baseline", "--no-live-traffic"]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--images", "3"] + common
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=800, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]                      # rank 0 prints ONE line
    two = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline"):
        assert key in two, key
    assert two["n_gpus"] == 2 and two["steps"] == 2 and two["config"]["images_total"] == 6 and two["value"] > 0
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--images", "6"] + common, env=env,
                         capture_output=True, text=True, timeout=800, cwd=ROOT)
    assert one.returncode == 0, one.stderr[-2000:]
    ref = json.loads([ln for ln in one.stdout.splitlines() if ln.startswith("{")][0])
    np.testing.assert_allclose(two["energies"], ref["energies"], rtol=1e-4)       # same job, images sharded 3 + 3
    assert two["total_solve_outer_iterations"] == ref["total_solve_outer_iterations"]


@pytest.mark.timeout(900)
def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with NO launcher around it (WORLD_SIZE unset: the form the driver's N = 1 command line takes
    with another N): the parent starts the two ranks as children before it touches the GPU, relays rank 0's one line and exits
    with the children's code.  Dry run on one GPU over gloo, like the test above; the line must be the one torch.distributed.run
    around bench.py produces."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(SRPS_BENCH_SHARED_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    common = ["--steps", "2", "--warmup", "1", "--size", "256", "--sf", "2", "--no-cpu-baseline", "--no-live-traffic", "--images", "3"]
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + common, env=env, capture_output=True, text=True,
                         timeout=800, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    two = json.loads(lines[0])
    cfg = two["config"]
    assert two["n_gpus"] == 2 and cfg["images_total"] == 6 and cfg["images_per_rank"] == [3, 3] and cfg["partition"] == "images"
    assert cfg["launched_by"].startswith("bench.py itself") and "ncclCommCount" in cfg and two["value"] > 0
    launched = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                               "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2"] + common, env=env,
                              capture_output=True, text=True, timeout=800, cwd=ROOT)
    assert launched.returncode == 0, launched.stderr[-2000:]
    ref = json.loads([ln for ln in launched.stdout.splitlines() if ln.startswith("{")][0])
    assert two["energies"] == ref["energies"]                     # the same job either way, bit for bit
    # a launcher that started another number of ranks than --gpus says is an error with a message, not a traceback
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + common, env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"),
                         capture_output=True, text=True, timeout=800, cwd=ROOT)
    assert bad.returncode != 0 and "--gpus 2 but the launcher started 1" in bad.stderr


@pytest.mark.timeout(900)
def test_bench_default_line_for_two_ranks_is_a_baseline_workload():
    """`bench.py --gpus 2` with nothing else said (the driver's command line for N > 1) runs BASELINE.json's configs[3] -- 2048 x 2048,
    sf 4, 40 images IN ALL, sharded 20 + 20 -- and says so; not 2 x 20 images of a workload BASELINE does not name.  Dry run on one GPU."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(SRPS_BENCH_SHARED_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-total-solve"],
                         env=env, capture_output=True, text=True, timeout=800, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    two = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])
    cfg = two["config"]
    assert cfg["workload"] == ("synthetic full-mask HR grid 2048x2048, sf 4, 40 images sharded over 2 GPUs (20, 20 per rank), 3 channels "
                               "[BASELINE.json configs[3]]"), cfg["workload"]
    assert cfg["images_per_rank"] == [20, 20] and cfg["images_total"] == 40 and two["scaling"] == "strong" and two["n_gpus"] == 2
    assert cfg["cg_steps_per_solve"] == 101 and cfg["ranks_seen_by_the_library"] == 2


@pytest.mark.timeout(900)
def test_bench_dry_run_with_the_resident_kernel_on_strips():
    """`bench.py --gpus 2 --partition strips` on one GPU: the images sharded 3 + 3, the library's sharded loop over gloo host collectives,
    and the depth CG as the resident kernel on two strips of tile columns -- two PROCESSES whose kernels run side by side and talk
    through hipIpc-mapped exchange buffers (cg_partition = 2).  The line says which form ran; the energies are those of the one-GPU job."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(SRPS_BENCH_SHARED_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    common = ["--steps", "2", "--warmup", "1", "--size", "1024", "--sf", "4", "--no-cpu-baseline", "--no-legs", "--no-live-traffic", "--option", "cg_resident_tile=512"]
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--images", "3", "--partition", "strips"] + common, env=env,
                         capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    two = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])
    assert "resident kernel on strips" in two["config"]["parallelism"], two["config"]["parallelism"]
    assert two["config"]["persistent_fallbacks"] == 0 and two["config"]["cg_steps_per_solve"] == 101 and two["config"]["ranks_seen_by_the_library"] == 2
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--images", "6"] + common, env=env, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert one.returncode == 0, one.stderr[-3000:]
    ref = json.loads([ln for ln in one.stdout.splitlines() if ln.startswith("{")][0])
    np.testing.assert_allclose(two["energies"], ref["energies"], rtol=1e-4)


@pytest.mark.parametrize("forced,expect", [("resident_strips", "replicated CG"), ("resident_strips,strips", "replicated CG")])
def test_bench_dry_run_degrades_inside_the_same_run(forced, expect):
    """SRPS_FORCE_FAIL (round 6): the handshake of the resident strips is made to fail on every rank exactly where a real failure would
    be noticed -- the ranks leave that path TOGETHER, the run goes on in the same processes (no exec, no restart) with the next form of
    the depth CG, finishes its 101 steps with the one-GPU job's energies, and the line names what it fell back from.  (The gloo dry run
    has no neighbour transport for the streaming strips, so the next form here is the replicated CG; on RCCL it is the streaming strips.)"""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(SRPS_BENCH_SHARED_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0", SRPS_FORCE_FAIL=forced)
    common = ["--steps", "2", "--warmup", "1", "--size", "1024", "--sf", "4", "--no-cpu-baseline", "--no-legs", "--no-live-traffic", "--option", "cg_resident_tile=512"]
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--images", "3", "--partition", "strips"] + common, env=env,
                         capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    two = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])
    cfg = two["config"]
    assert expect in cfg["parallelism"] and "resident kernel on strips" not in cfg["parallelism"], cfg["parallelism"]
    assert cfg["forced_failures"] == forced and len(cfg["degraded"]) == 1 and cfg["degraded"][0].startswith("resident kernel on strips"), cfg["degraded"]
    assert cfg["cg_steps_per_solve"] == 101 and "fell back" in out.stderr
    env.pop("SRPS_FORCE_FAIL")
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--images", "6"] + common, env=env, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert one.returncode == 0, one.stderr[-3000:]
    ref = json.loads([ln for ln in one.stdout.splitlines() if ln.startswith("{")][0])
    np.testing.assert_allclose(two["energies"], ref["energies"], rtol=1e-4)


@pytest.mark.parametrize("bytes_store", [False, True])
def test_a_shard_forms_the_complete_albedo_denominator_itself(pkg, bytes_store):
    """srps_albedo_partial on a context that holds a shard of the images: den = sum_i (N . s_i)^2 does not involve the images, so it
    is formed over ALL images on every rank -- the same bits as on one GPU -- and only num (C P floats, not 2 C P) is exchanged;
    the shards' num add up to the single-GPU num.  One process, three contexts."""
    import torch
    from importlib import import_module
    api = import_module("srmeetsps-cuda_amd.api")
    n_img, n_ch = 7, 3
    # bytes_store: 8-bit images (k / 255.f), which the contexts then also hold as bytes -- the sweep's byte variant on a shard
    quant = (lambda sc: setattr(sc, "I", (np.rint(np.clip(sc.I, 0, 1) * 255).astype(np.float32) / np.float32(255))) or sc) if bytes_store else (lambda sc: sc)
    full = quant(pkg.synth.make_scene(64, 48, 2, n_img, seed=37, mask_kind="full" if bytes_store else "ragged"))
    ctx = pkg.Context(device_id=0)
    ctx.set_option("albedo_mode", 0)                       # num and den in the exchange buffer (the default forms the albedo inside the sweep)
    ctx.setup(pkg.DataHandler.from_scene(full))
    assert ctx.get_option("image_store_bytes_active") == (1 if bytes_store else 0)
    ctx.lighting()
    s_full = ctx.get("s")
    P = ctx.dims()["npix"]

    def num_den(c):
        c.albedo_partial()
        ptr, n = c.exchange_ptr("albedo")
        both = torch.as_tensor(api._DevView(ptr, 2 * n_ch * P), device="cuda:0").cpu().numpy().copy()      # den lies behind num
        return n, both[:n_ch * P], both[n_ch * P:]

    n_one, num_one, den_one = num_den(ctx)
    assert n_one == n_ch * P
    num_sum = np.zeros_like(num_one, dtype=np.float64)
    for lo, hi in ((0, 3), (3, 7)):
        sh = quant(pkg.synth.make_scene(64, 48, 2, n_img, seed=37, mask_kind="full" if bytes_store else "ragged", img_begin=lo, img_end=hi))
        c = pkg.Context(device_id=0)
        c.setup(pkg.DataHandler.from_scene(sh))
        assert c.get_option("image_store_bytes_active") == (1 if bytes_store else 0)
        c.set("s", s_full)                                            # what the all-reduce of s leaves on every rank
        n, num, den = num_den(c)
        assert n == n_ch * P
        assert np.array_equal(den.view(np.uint32), den_one.view(np.uint32))
        num_sum += num
        c.close()
    np.testing.assert_allclose(num_sum, num_one, rtol=2e-6, atol=1e-6)
    ctx.close()
