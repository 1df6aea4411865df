"""RCCL with MORE than one rank -- runs only where at least two devices exist (the GPU boxes of this build have one: there these
tests skip, and the multi-rank arithmetic is covered over gloo / host collectives by tests/test_gpu_comm.py,
tests/test_gpu_strips.py and tests/test_gpu_distributed.py).  On a multi-GPU node they exercise what no one-GPU box can:
srps_comm_init_rank with two processes, ncclAllReduce of s / num / q between devices, the overlapped exchange on a second stream,
the strip-partitioned CG's grouped ncclSend / ncclRecv and its gather by grouped broadcasts, srps_comm_init_all with one host thread
per device (`srps --gpus 2`), and bench.py --gpus 2 as the driver starts it."""
import importlib
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _n_devices():
    import torch
    return torch.cuda.device_count()


needs_two = pytest.mark.skipif("_n_devices() < 2", reason="needs two GPUs (RCCL takes one rank per device)")


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def rmse(a, b):
    return float(np.sqrt(np.mean((np.asarray(a, np.float64) - np.asarray(b, np.float64)) ** 2)))


def _rank(rank, world, port, h, w, sf, n_img, kind, seed, overlap, strips, out_dir):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(rank)
    dist.init_process_group("gloo", rank=rank, world_size=world)        # carries the 128-byte id only: the collectives are the library's
    pkg = importlib.import_module("srmeetsps-cuda_amd")
    lo, hi = pkg.shard_range(n_img, world, rank)
    sc = pkg.synth.make_scene(h, w, sf, n_img, seed=seed, mask_kind=kind, img_begin=lo, img_end=hi)
    ctx = pkg.Context(device_id=rank)
    ctx.set_option("overlap_exchange", overlap)
    if strips == 1:
        ctx.set_option("cg_resident", 0); ctx.set_option("cg_partition", 1)
    if strips == 2:
        ctx.set_option("cg_resident_tile", 512); ctx.set_option("cg_partition", 2)      # the resident kernel on strips, peer memory through hipIpc
    uid = [pkg.Context.comm_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(uid, src=0)
    ctx.comm_init_rank(uid[0], rank, world)
    assert ctx.comm_info() == (rank, world)                               # ncclCommCount / ncclCommUserRank through the library
    ctx.setup(pkg.DataHandler.from_scene(sc))
    en = ctx.execute_sharded(0)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), en=np.array(en), z=ctx.get("z"), rho=ctx.get("rho"), s=ctx.get("s"),
             strips=2 if ctx.get_option("cg_partition_resident_active") else ctx.get_option("cg_partition_active"), it=ctx.last_cg_iterations()["depth"])
    dist.barrier()
    ctx.close()
    dist.destroy_process_group()


@needs_two
@pytest.mark.timeout(900)
@pytest.mark.parametrize("overlap,strips", [(0, 0), (1, 0), (0, 1), (0, 2)])
def test_two_ranks_over_rccl_equal_one_gpu(pkg, tmp_path, overlap, strips):
    import torch.multiprocessing as mp
    h, w, sf, n_img, kind, seed = (512, 384, 2, 5, "ellipse", 71)
    mp.spawn(_rank, args=(2, _free_port(), h, w, sf, n_img, kind, seed, overlap, strips, str(tmp_path)), nprocs=2, join=True)
    r = [np.load(tmp_path / f"rank{q}.npz") for q in range(2)]
    np.testing.assert_array_equal(r[0]["z"], r[1]["z"]); np.testing.assert_array_equal(r[0]["s"], r[1]["s"])      # replicas, bit for bit
    np.testing.assert_array_equal(r[0]["en"], r[1]["en"])
    assert int(r[0]["strips"]) == strips
    sc = pkg.synth.make_scene(h, w, sf, n_img, seed=seed, mask_kind=kind)
    ctx = pkg.Context(device_id=0)
    if strips == 1:
        ctx.set_option("cg_resident", 0)
    if strips == 2:
        ctx.set_option("cg_resident_tile", 512)
    one = pkg.SRPS(pkg.DataHandler.from_scene(sc), ctx=ctx)
    e1 = one.execute()
    assert len(e1) == len(r[0]["en"])
    np.testing.assert_allclose(r[0]["en"], e1, rtol=5e-4)
    assert rmse(r[0]["z"], one.z()) < 3e-5 and np.abs(r[0]["rho"].reshape(-1) - one.rho().reshape(-1)).max() < 5e-4
    ctx.close()


@needs_two
@pytest.mark.timeout(900)
def test_two_ranks_that_both_hold_every_image_are_refused(tmp_path):
    """srps_execute_sharded all-reduces an image-coverage vector before its first pass: ranges that overlap are an error, not a
    doubled s / num / q"""
    code = (
        "import sys, os, importlib, numpy as np\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "import torch, torch.distributed as dist\n"
        "rank = int(os.environ['RANK']); torch.cuda.set_device(rank); dist.init_process_group('gloo')\n"
        "pkg = importlib.import_module('srmeetsps-cuda_amd')\n"
        "sc = pkg.synth.make_scene(64, 48, 2, 4, seed=5, mask_kind='ragged')\n"          # ALL images on both ranks
        "ctx = pkg.Context(device_id=rank)\n"
        "uid = [pkg.Context.comm_unique_id() if rank == 0 else None]; dist.broadcast_object_list(uid, src=0)\n"
        "ctx.comm_init_rank(uid[0], rank, 2); ctx.setup(pkg.DataHandler.from_scene(sc))\n"
        "try:\n    ctx.execute_sharded(1); print('ACCEPTED')\n"
        "except Exception as e:\n    print('REFUSED', e)\n"
        "ctx.close(); dist.destroy_process_group()\n")
    script = tmp_path / "overlap.py"; script.write_text(code)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(_free_port()), str(script)], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.count("REFUSED") == 2 and "is held by 2 ranks" in out.stdout, out.stdout[-2000:]


@needs_two
@pytest.mark.timeout(900)
@pytest.mark.parametrize("partition", ["images", "strips"])
def test_command_line_program_on_two_gpus(pkg, tmp_path, partition):
    """`srps --gpus 2` (one host thread + context per device, ncclCommInitAll) writes what `srps` writes on one GPU"""
    import scipy.io
    pkg.host.load()
    sc = pkg.synth.make_scene(256, 192, 2, 6, seed=43, mask_kind="ellipse")
    h, w = sc.h, sc.w
    I4 = np.transpose(sc.I.reshape(sc.n_img, sc.n_ch, w, h), (3, 2, 1, 0)).astype(np.float64)
    path = str(tmp_path / "scene.mat")
    scipy.io.savemat(path, {"I": I4, "K": sc.K.reshape(3, 3).T.astype(np.float64), "mask": sc.mask.reshape(w, h).T.astype(np.uint8),
                            "sf": float(sc.sf), "z0": sc.z0.reshape(w // sc.sf, h // sc.sf).T.astype(np.float64)}, do_compression=True)
    (tmp_path / "a").mkdir(); (tmp_path / "b").mkdir()
    one = subprocess.run([pkg.host.CLI, f"--dsloc={path}", "-o", str(tmp_path / "a")], capture_output=True, text=True, timeout=600)
    assert one.returncode == 0, one.stderr
    two = subprocess.run([pkg.host.CLI, f"--dsloc={path}", "-o", str(tmp_path / "b"), "--gpus", "2", "--partition", partition], capture_output=True, text=True, timeout=600)
    assert two.returncode == 0, two.stderr
    assert "Images sharded over 2 GPUs" in two.stdout and two.stdout.count("Iteration") == one.stdout.count("Iteration") and "Done!" in two.stdout
    if partition == "strips":
        # the ranks are threads of one process: the handshake takes the peer-pointer route (hipIpc handles open in OTHER processes only).
        # Whether the solve then ends on the resident kernel depends on the node: it needs peer access between the two devices and
        # fine-grained device memory for the exchange buffers -- a node without either falls back (streaming strips / replicated CG), which
        # is correct behaviour and is REPORTED here, not failed (round-5 advisor finding: this route has never run on two real devices; the
        # rmse checks below are what catch a coherence bug).  Where both are available the resident path is required.
        line = [ln for ln in two.stdout.splitlines() if ln.startswith("Depth CG:")]
        assert line, two.stdout[-1500:]
        import ctypes as C
        hip = C.CDLL("libamdhip64.so")
        can01, can10 = C.c_int(0), C.c_int(0)
        hip.hipDeviceCanAccessPeer(C.byref(can01), 0, 1); hip.hipDeviceCanAccessPeer(C.byref(can10), 1, 0)
        fine = C.c_void_p()
        fine_ok = hip.hipExtMallocWithFlags(C.byref(fine), C.c_size_t(1 << 20), C.c_uint(0x1)) == 0      # hipDeviceMallocFinegrained
        if fine_ok:
            hip.hipFree(fine)
        print(f"srps --gpus 2 --partition strips: {line[-1]} (peer access 0->1 {can01.value}, 1->0 {can10.value}, fine-grained memory {'yes' if fine_ok else 'no'})")
        if can01.value and can10.value and fine_ok:
            assert "the resident kernel on strips" in line[-1], two.stdout[-1500:]
    za = scipy.io.loadmat(str(tmp_path / "a" / "z.mat"))["x"][:, 0]; zb = scipy.io.loadmat(str(tmp_path / "b" / "z.mat"))["x"][:, 0]
    assert rmse(za, zb) < 3e-5
    ra = scipy.io.loadmat(str(tmp_path / "a" / "rho.mat"))["x"][:, 0]; rb = scipy.io.loadmat(str(tmp_path / "b" / "rho.mat"))["x"][:, 0]
    assert np.abs(ra - rb).max() < 5e-4


@needs_two
@pytest.mark.timeout(1200)
def test_bench_on_two_gpus_started_bare_and_by_the_launcher():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "SRPS_BENCH_SHARED_GPU")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    common = ["--steps", "3", "--warmup", "1", "--no-cpu-baseline"]
    bare = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + common, env=env, capture_output=True, text=True, timeout=1000, cwd=ROOT)
    assert bare.returncode == 0, bare.stderr[-2000:]
    line = json.loads([ln for ln in bare.stdout.splitlines() if ln.startswith("{")][0])
    cfg = line["config"]
    assert line["n_gpus"] == 2 and cfg["ncclCommCount"] == 2 and cfg["images_per_rank"] == [20, 20] and cfg["comm"].startswith("ncclAllReduce inside libsrps_hip.so")
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--images", "40", "--no-legs", "--no-total-solve", "--no-live-traffic"] + common, env=env,
                         capture_output=True, text=True, timeout=1000, cwd=ROOT)
    assert one.returncode == 0, one.stderr[-2000:]
    ref = json.loads([ln for ln in one.stdout.splitlines() if ln.startswith("{")][0])
    np.testing.assert_allclose(line["energies"], ref["energies"], rtol=1e-4)          # the same 40-image job, sharded 20 + 20
