"""RCCL behind the C ABI (include/srps.h "multi-GPU through the boundary"): communicators bound to contexts, the all-reduces of
the image-sharded pass inside the library, the C++ host's --gpus / --sharded path.  A one-GPU box can only form one-rank
communicators (RCCL takes one rank per device): they run the real ncclAllReduce calls on the real buffers and streams, and
the results must be those of srps_execute bit for bit.  The N > 1 arithmetic of the same phases is covered by the gloo tests
(tests/test_distributed_gloo.py, tests/test_gpu_distributed.py) and at volume by tests/test_gpu_full_size.py."""
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _scene(pkg, seed=51):
    return pkg.synth.make_scene(96, 80, 2, 5, seed=seed, mask_kind="ragged")


def _run(pkg, dh, how):
    ctx = pkg.Context(device_id=0)
    if how == "init_all":
        pkg.Context.comm_init_all([ctx])
    elif how == "init_rank":
        ctx.comm_init_rank(pkg.Context.comm_unique_id(), 0, 1)
    ctx.setup(dh)
    en = ctx.execute(0) if how == "plain" else ctx.execute_sharded(0)
    out = dict(en=en, z=ctx.get("z"), rho=ctx.get("rho"), s=ctx.get("s"), N=ctx.get("N"), it=ctx.last_cg_iterations(), comm=ctx.comm_info())
    ctx.close()
    return out


@pytest.mark.parametrize("how", ["init_all", "init_rank"])
def test_sharded_loop_on_a_one_rank_communicator_equals_execute(pkg, how):
    dh = pkg.DataHandler.from_scene(_scene(pkg))
    a = _run(pkg, dh, "plain")
    b = _run(pkg, dh, how)
    assert a["comm"] == (0, 0) and b["comm"] == (0, 1)
    assert a["en"] == b["en"] and len(a["en"]) >= 2
    for k in ("z", "rho", "s", "N"):
        np.testing.assert_array_equal(a[k], b[k])
    assert a["it"] == b["it"]


def test_all_reduce_entry_point_and_errors(pkg):
    import torch
    from importlib import import_module
    api = import_module("srmeetsps-cuda_amd.api")
    dh = pkg.DataHandler.from_scene(_scene(pkg, 52))
    ctx = pkg.Context(device_id=0)
    ctx.setup(dh)
    with pytest.raises(pkg.SRPSError) as ei:
        ctx.all_reduce("albedo")                         # no communicator bound
    assert ei.value.code == 3
    with pytest.raises(pkg.SRPSError) as ei:
        ctx.execute_sharded(1)
    assert ei.value.code == 3
    pkg.Context.comm_init_all([ctx])
    ctx.lighting_local(); ctx.all_reduce("s")
    ctx.albedo_partial()
    ptr, n = ctx.exchange_ptr("albedo")
    before = torch.as_tensor(api._DevView(ptr, n), device="cuda:0").clone()
    ctx.all_reduce("albedo"); ctx.synchronize()
    after = torch.as_tensor(api._DevView(ptr, n), device="cuda:0")
    assert torch.equal(before, after)                    # the sum over one rank
    with pytest.raises(pkg.SRPSError):
        ctx.all_reduce("nonsense")
    # two contexts on one device cannot form a communicator: refused by the library, not by a hang inside RCCL
    c2 = pkg.Context(device_id=0)
    with pytest.raises(pkg.SRPSError) as ei:
        pkg.Context.comm_init_all([ctx, c2])
    assert ei.value.code == 1 and "one rank per device" in str(ei.value)
    c2.close()
    ctx.comm_release()
    assert ctx.comm_info() == (0, 0)
    ctx.close()


def test_a_persistent_abort_in_the_sharded_loop_repeats_the_pass_and_matches_streaming(pkg):
    """the sharded loop defers the look at the abort flags to the end of the pass (they travel with the energy term): a depth
    CG whose waits cannot be served (a 40 ms budget against a co-tenant holding CUs for 1.5 s) gives up, the pass's tail is repeated by
    the streaming kernels from the iterate the launch started from -- the result is that of a context that streamed from the
    start, bit for bit"""
    import ctypes
    import os
    import torch
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so = os.path.join(root, "tools", "libcu_holder.so")
    if not os.path.exists(so):
        pytest.skip("tools/libcu_holder.so is not built")
    import time
    holder = ctypes.CDLL(so)
    sc = pkg.synth.make_scene(2048, 2048, 4, 2, seed=53, mask_kind="full")      # 256 tiles of 256 x 64, 157 KiB of LDS each: every CU, and none that a co-tenant holds
    dh = pkg.DataHandler.from_scene(sc)

    def run(resident, with_holder):
        ctx = pkg.Context(device_id=0)
        ctx.set_option("cg_resident", resident); ctx.set_option("albedo_persistent", 0)
        ctx.set_option("spin_budget_ms", 40)
        pkg.Context.comm_init_all([ctx])
        ctx.setup(dh)
        if with_holder:
            assert holder.cu_holder_launch(64, 1500, 16) == 0   # 64 blocks that pin LDS on a CU each for 1.5 s, on another stream
            time.sleep(0.2)
        en = ctx.execute_sharded(1)
        if with_holder:
            assert holder.cu_holder_wait() == 0
        out = dict(en=en, z=ctx.get("z"), rho=ctx.get("rho"), fb=ctx.get_option("persistent_fallbacks"), it=ctx.last_cg_iterations()["depth"])
        ctx.close()
        torch.cuda.synchronize()
        return out
    ref = run(0, False)
    got = run(1, True)
    assert got["fb"] == 1 and ref["fb"] == 0
    assert got["it"] == ref["it"] == 101
    assert got["en"] == ref["en"]
    np.testing.assert_array_equal(got["z"], ref["z"]); np.testing.assert_array_equal(got["rho"], ref["rho"])


def test_command_line_program_sharded_path_equals_one_gpu_path(pkg, tmp_path):
    """`srps --sharded` (C++ host: one context per device on its own thread, ncclCommInitAll, srps_execute_sharded) writes the
    same s / rho / z / N as `srps` on the same file; --gpus beyond the devices of the box is an error message, not a crash"""
    import scipy.io
    pkg.host.load()
    sc = pkg.synth.make_scene(40, 48, 2, 4, seed=41, mask_kind="ragged")
    h, w = sc.h, sc.w
    I4 = np.transpose(sc.I.reshape(sc.n_img, sc.n_ch, w, h), (3, 2, 1, 0)).astype(np.float64)
    path = str(tmp_path / "scene.mat")
    scipy.io.savemat(path, {"I": I4, "K": sc.K.reshape(3, 3).T.astype(np.float64), "mask": sc.mask.reshape(w, h).T.astype(np.uint8),
                            "sf": float(sc.sf), "z0": sc.z0.reshape(w // sc.sf, h // sc.sf).T.astype(np.float64)}, do_compression=True)
    (tmp_path / "a").mkdir(); (tmp_path / "b").mkdir()
    one = subprocess.run([pkg.host.CLI, f"--dsloc={path}", "-o", str(tmp_path / "a")], capture_output=True, text=True)
    assert one.returncode == 0, one.stderr
    sh = subprocess.run([pkg.host.CLI, f"--dsloc={path}", "-o", str(tmp_path / "b"), "--sharded", "--gpus", "1"], capture_output=True, text=True)
    assert sh.returncode == 0, sh.stderr
    assert "Images sharded over 1 GPU" in sh.stdout and sh.stdout.count("Iteration") == one.stdout.count("Iteration") and "Done!" in sh.stdout
    for name in ("s.mat", "rho.mat", "z.mat", "N.mat", "z_init.mat", "zs_init.mat"):
        np.testing.assert_array_equal(scipy.io.loadmat(str(tmp_path / "a" / name))["x"], scipy.io.loadmat(str(tmp_path / "b" / name))["x"])
    import torch
    n_dev = torch.cuda.device_count()
    bad = subprocess.run([pkg.host.CLI, f"--dsloc={path}", "--no-output", "--gpus", str(n_dev + 1)], capture_output=True, text=True)
    assert bad.returncode == 1 and f"this node shows {n_dev} HIP device(s)" in bad.stderr and "Usage" not in bad.stdout
    # with --partition strips (round 4) the one-rank path still equals the one-GPU path (a single strip is the whole grid)
    (tmp_path / "c").mkdir()
    st = subprocess.run([pkg.host.CLI, f"--dsloc={path}", "-o", str(tmp_path / "c"), "--sharded", "--gpus", "1", "--partition", "strips"], capture_output=True, text=True)
    assert st.returncode == 0, st.stderr
    assert "--partition strips has no effect on one device" in st.stdout
    np.testing.assert_array_equal(scipy.io.loadmat(str(tmp_path / "a" / "z.mat"))["x"], scipy.io.loadmat(str(tmp_path / "c" / "z.mat"))["x"])


def test_command_line_program_on_an_image_folder_hands_the_bytes_to_the_device(pkg, tmp_path):
    """`srps --dstype images` (reference: ImageDataHandler, Utilities.cpp:349-395): the C++ host keeps the bytes of the 8-bit PNGs
    and passes them as srps_problem.I_u8; the Python host on the same folder passes the floats byte / 255.f.  Same results, bit
    for bit, and the same again through --sharded."""
    import scipy.io
    from PIL import Image
    pkg.host.load()
    sc = pkg.synth.make_scene(48, 64, 2, 5, seed=61, mask_kind="ellipse")
    h, w = sc.h, sc.w
    root = tmp_path / "ds"; (root / "RGB").mkdir(parents=True); (root / "Depth").mkdir()
    I8 = np.rint(np.clip(sc.I, 0, 1) * 255).astype(np.uint8).reshape(sc.n_img, 3, w, h)            # [n][c][j][i]
    for n in range(sc.n_img):
        Image.fromarray(np.ascontiguousarray(np.transpose(I8[n], (2, 1, 0))), "RGB").save(root / "RGB" / f"I_{n + 1}.png")
    Image.fromarray((sc.mask.reshape(w, h).T * 255).astype(np.uint8), "L").save(root / "mask.png")
    z0 = sc.z0.reshape(w // sc.sf, h // sc.sf).T
    lo, hi = 0.5, 1.5
    Image.fromarray(np.rint((z0 - lo) / (hi - lo) * 65535).astype(np.uint16)).save(root / "Depth" / "D_0.png")
    K = sc.K.reshape(3, 3).T
    (root / "K.txt").write_text("\n".join(",".join(repr(float(v)) for v in K[r]) for r in range(3)) + f"\n{sc.sf},{lo},{hi}")
    (tmp_path / "a").mkdir(); (tmp_path / "b").mkdir()
    one = subprocess.run([pkg.host.CLI, "--dstype=images", f"--dsloc={root}", "-o", str(tmp_path / "a")], capture_output=True, text=True)
    assert one.returncode == 0, one.stderr
    sh = subprocess.run([pkg.host.CLI, "-t", "images", "-d", str(root), "-o", str(tmp_path / "b"), "--sharded"], capture_output=True, text=True)
    assert sh.returncode == 0, sh.stderr
    dh = pkg.host.load_dataset("images", str(root))
    ctx = pkg.Context(device_id=0)
    srps = pkg.SRPS(dh, ctx=ctx)
    en = srps.execute()
    assert one.stdout.count("Iteration") == len(en)
    for d in ("a", "b"):
        np.testing.assert_array_equal(scipy.io.loadmat(str(tmp_path / d / "z.mat"))["x"][:, 0], srps.z())
        np.testing.assert_array_equal(scipy.io.loadmat(str(tmp_path / d / "rho.mat"))["x"][:, 0], srps.rho().reshape(-1))
        np.testing.assert_array_equal(scipy.io.loadmat(str(tmp_path / d / "s.mat"))["x"][:, 0], srps.s().reshape(-1))
    ctx.close()


@pytest.mark.parametrize("h,w,sf,kind", [(96, 80, 2, "ragged"), (512, 640, 4, "full")])
def test_overlapped_exchange_gives_the_same_bits(pkg, h, w, sf, kind):
    """option overlap_exchange: the albedo sweep and the depth assembly cut into four pixel ranges, each range all-reduced on a
    second stream while the next is computed -- on a context that holds a SHARD (3 of 6 images; the compact q buffer and the SHARD
    form of the albedo sweep are in play) with a one-rank communicator: the same results bit for bit as the unchunked pass"""
    sc = pkg.synth.make_scene(h, w, sf, 6, seed=71, mask_kind=kind, img_begin=0, img_end=3)
    dh = pkg.DataHandler.from_scene(sc)
    out = []
    for overlap in (0, 1):
        ctx = pkg.Context(device_id=0)
        ctx.set_option("overlap_exchange", overlap)
        ctx.set_option("shard_range_check", 0)             # ONE shard alone on purpose (the SHARD kernels): no partition to verify
        pkg.Context.comm_init_all([ctx])
        ctx.setup(dh)
        en = ctx.execute_sharded(2)
        out.append((en, ctx.get("z"), ctx.get("rho"), ctx.get("s")))
        ctx.close()
    assert out[0][0] == out[1][0] and len(out[0][0]) == 2
    for a, b in zip(out[0][1:], out[1][1:]):
        np.testing.assert_array_equal(a, b)


def _hosted_sharded_worker(rank, world, port, h, w, sf, n_img, kind, seed, overlap, out_dir):
    import importlib
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pkg = importlib.import_module("srmeetsps-cuda_amd")
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _strip_protocol as strips
    lo, hi = pkg.shard_range(n_img, world, rank)
    sc = pkg.synth.make_scene(h, w, sf, n_img, seed=seed, mask_kind=kind, img_begin=lo, img_end=hi)
    ctx = pkg.Context(device_id=0)
    ctx.set_option("overlap_exchange", overlap)
    hc = strips.HostedCollectives(ctx, dist)
    assert ctx.comm_info() == (rank, world)
    ctx.setup(pkg.DataHandler.from_scene(sc))
    en = ctx.execute_sharded(0)                         # the loop of SRPS.cu:272-335 inside the library, to the reference's stop rule
    assert not hc.errors, hc.errors
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), en=np.array(en), z=ctx.get("z"), rho=ctx.get("rho"), s=ctx.get("s"), fb=ctx.get_option("persistent_fallbacks"))
    dist.barrier()
    hc.remove()
    ctx.close()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("h,w,sf,n_img,kind,world,overlap", [(96, 80, 2, 5, "ragged", 2, 0), (256, 320, 4, 7, "ellipse", 3, 1)])
def test_sharded_loop_inside_the_library_over_two_and_three_processes(pkg, tmp_path, h, w, sf, n_img, kind, world, overlap):
    """srps_execute_sharded with MORE than one rank: the library's own loop -- its all-reduces, the compact q buffer, the flags that
    travel with the energy term, the stop decision every rank takes for itself -- over collectives supplied by the caller
    (srps_set_host_collectives: torch.distributed / gloo between processes that share the GPU, which RCCL cannot do), against the
    one-context solve; also with the exchange overlapped"""
    import socket
    import torch.multiprocessing as mp
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    seed = h + 5 * w + world
    mp.spawn(_hosted_sharded_worker, args=(world, port, h, w, sf, n_img, kind, seed, overlap, str(tmp_path)), nprocs=world, join=True)
    r = [np.load(tmp_path / f"rank{q}.npz") for q in range(world)]
    for q in range(1, world):                                # replicas: the same bits on every rank
        for k in ("en", "z", "rho", "s"):
            np.testing.assert_array_equal(r[q][k], r[0][k])
    sc = pkg.synth.make_scene(h, w, sf, n_img, seed=seed, mask_kind=kind)
    ctx = pkg.Context(device_id=0)
    one = pkg.SRPS(pkg.DataHandler.from_scene(sc), ctx=ctx)
    e1 = one.execute()
    d_z = float(np.sqrt(np.mean((r[0]["z"] - one.z()) ** 2)))
    print(f"{world} ranks, library loop over gloo: {len(r[0]['en'])} passes (one context: {len(e1)}), depth RMSE {d_z:.3e}")
    assert len(r[0]["en"]) == len(e1)
    np.testing.assert_allclose(r[0]["en"], e1, rtol=5e-4)
    assert d_z < 3e-5 and np.abs(r[0]["rho"].reshape(one.rho().shape) - one.rho()).max() < 5e-4
    ctx.close()


def test_image_ranges_that_do_not_tile_the_image_set_are_refused(pkg):
    """srps_execute_sharded checks, with one small all-reduce before its first pass, that every image is held by exactly one rank
    (round-3 advisor finding: two ranks holding all images used to double s, num and q silently).  One rank, one-rank
    communicator, a context that holds images [0, 3) of 5: images 3 and 4 are on no rank."""
    sc = pkg.synth.make_scene(48, 40, 2, 5, seed=19, mask_kind="ragged", img_begin=0, img_end=3)
    ctx = pkg.Context(device_id=0)
    pkg.Context.comm_init_all([ctx])
    ctx.setup(pkg.DataHandler.from_scene(sc))
    with pytest.raises(Exception) as ei:
        ctx.execute_sharded(1)
    assert "image 3 of 5 is held by 0 ranks" in str(ei.value)
    ctx.close()
    # the whole set on the one rank is fine
    sc = pkg.synth.make_scene(48, 40, 2, 5, seed=19, mask_kind="ragged")
    ctx = pkg.Context(device_id=0)
    pkg.Context.comm_init_all([ctx])
    ctx.setup(pkg.DataHandler.from_scene(sc))
    assert len(ctx.execute_sharded(1)) == 1
    ctx.close()


def test_device_pointers_to_normals_stay_current(pkg):
    """srps_get_device_ptr("N" | "dz") hands out THE arrays: from then on the context stops double-buffering the normals
    (option fuse_normals swaps two sets per pass), so a zero-copy view keeps showing the current pass's normals (round-3 advisor
    finding: it showed the previous pass's)"""
    import torch
    from importlib import import_module
    api = import_module("srmeetsps-cuda_amd.api")
    sc = pkg.synth.make_scene(64, 48, 2, 4, seed=23, mask_kind="ellipse")
    ctx = pkg.Context(device_id=0)
    ctx.setup(pkg.DataHandler.from_scene(sc))
    pkg.alternating_loop(ctx, None, max_outer=1)
    ptr, n = ctx.device_ptr("N")
    view = torch.as_tensor(api._DevView(ptr, n), device="cuda:0")
    for _ in range(3):
        pkg.alternating_loop(ctx, None, max_outer=1)
        ctx.synchronize()
        assert ctx.device_ptr("N")[0] == ptr
        np.testing.assert_array_equal(view.cpu().numpy(), ctx.get("N"))
    ctx.close()
    # and the results are those of a context that was never asked
    a = pkg.Context(device_id=0); a.setup(pkg.DataHandler.from_scene(sc)); ea = pkg.alternating_loop(a, None, max_outer=3)
    b = pkg.Context(device_id=0); b.setup(pkg.DataHandler.from_scene(sc)); b.device_ptr("dz"); eb = pkg.alternating_loop(b, None, max_outer=3)
    assert ea == eb
    np.testing.assert_array_equal(a.get("z"), b.get("z")); np.testing.assert_array_equal(a.get("N"), b.get("N"))
    a.close(); b.close()
