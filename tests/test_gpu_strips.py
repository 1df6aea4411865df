"""The strip-partitioned depth CG on the GPU (srps_strips.hip): 2, 3 and 4 ranks as contexts of ONE process on one device and one
stream (srps_strip_group_solve: the collectives are device copies and a summing kernel; views, halo columns, totals and kernels
are those of the multi-GPU path) against the single-grid CG of the same library, and against the oracle.  The recurrence is
devicecalls.cu:252-275; what changes with the partition is only the grouping of the four dot-product sums (per rank, then over
ranks), so the results agree to rounding, not to the bit."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
f32 = np.float32


def rmse(a, b):
    return float(np.sqrt(np.mean((np.asarray(a, np.float64) - np.asarray(b, np.float64)) ** 2)))


def _pass_with_group(pkg, dh, world, stream):
    ctxs = []
    for r in range(world):
        c = pkg.Context(device_id=0)
        c.set_stream(stream)
        c.set_option("cg_resident", 0)
        c.setup(dh)
        ctxs.append(c)
    for c in ctxs:
        c.lighting(); c.albedo(); c.depth_partial()
    pkg.Context.strip_group_solve(ctxs)
    out = []
    for c in ctxs:
        c.energy_partial(); c.normals()
        e = c.energy_finish()
        out.append((e, c.get("z"), c.last_cg_iterations()["depth"]))
        c.close()
    return out


def _pass_single(pkg, dh, stream, fused=1):
    c = pkg.Context(device_id=0)
    c.set_stream(stream)
    c.set_option("cg_resident", 0); c.set_option("cg_fused_step", fused)
    c.setup(dh)
    c.lighting(); c.albedo()
    e = c.depth()
    out = (e, c.get("z"), c.last_cg_iterations()["depth"])
    c.close()
    return out


@pytest.mark.parametrize("h,w,sf,kind,world", [(96, 80, 2, "ragged", 2), (96, 80, 2, "ragged", 3), (300, 200, 1, "ragged", 4), (512, 384, 4, "ellipse", 2),
                                               (512, 384, 4, "ellipse", 4), (256, 640, 2, "full", 3)])
def test_strip_ranks_in_one_process_equal_the_single_grid_cg(pkg, oracle, h, w, sf, kind, world):
    import torch
    sc = pkg.synth.make_scene(h, w, sf, 3, seed=h + 7 * w + world, mask_kind=kind)
    dh = pkg.DataHandler.from_scene(sc)
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        e1, z1, it1 = _pass_single(pkg, dh, stream.cuda_stream)
        group = _pass_with_group(pkg, dh, world, stream.cuda_stream)
    for e, z, it in group[1:]:                                  # every rank ends with the same depth, bit for bit
        assert e == group[0][0] and it == group[0][2]
        np.testing.assert_array_equal(z, group[0][1])
    e, z, it = group[0]
    print(f"{h}x{w} sf {sf} {kind}, {world} strips: depth RMSE vs single grid {rmse(z, z1):.3e}, energy {e} vs {e1}, steps {it} / {it1}")
    assert abs(it - it1) <= 1 and it >= 10
    assert rmse(z, z1) < 2e-5
    assert abs(e - e1) <= 1e-3 * abs(e1)
    if h * w <= 96 * 80:
        ref = oracle.execute(oracle.Problem(sc.h, sc.w, sc.sf, sc.mask, sc.K, sc.I, sc.zs_lr, sc.z_init), depth="faithful", max_outer=1)
        assert rmse(z, ref.z) < 5e-5


@pytest.mark.timeout(1800)
@pytest.mark.parametrize("world", [2, 4])
def test_strips_at_the_largest_grid(pkg, world):
    """4096 x 4096, sf 2 (BASELINE.json configs[4]'s grid): 2 and 4 strips of 2048 / 1024 columns against the single grid,
    all 101 steps"""
    import torch
    sc = pkg.synth.make_scene(4096, 4096, 2, 2, seed=1238, mask_kind="full")
    dh = pkg.DataHandler.from_scene(sc)
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        e1, z1, it1 = _pass_single(pkg, dh, stream.cuda_stream)
        group = _pass_with_group(pkg, dh, world, stream.cuda_stream)
    e, z, it = group[0]
    print(f"4096^2, {world} strips: depth RMSE vs single grid {rmse(z, z1):.3e}, energy {e} vs {e1}")
    assert it == it1 == 101
    for e_r, z_r, it_r in group[1:]:
        np.testing.assert_array_equal(z_r, z)
    assert rmse(z, z1) < 2e-5
    assert abs(e - e1) <= 1e-4 * abs(e1)


def test_group_solve_refuses_contexts_that_do_not_belong_together(pkg):
    import torch
    sc = pkg.synth.make_scene(40, 32, 2, 2, seed=5, mask_kind="full")
    dh = pkg.DataHandler.from_scene(sc)
    a = pkg.Context(device_id=0); b = pkg.Context(device_id=0)
    a.setup(dh); b.setup(dh)
    for c in (a, b):
        c.lighting(); c.albedo(); c.depth_partial()
    with pytest.raises(pkg.SRPSError) as ei:
        pkg.Context.strip_group_solve([a, b])                   # two streams
    assert ei.value.code == 1 and "one stream" in str(ei.value)
    s = torch.cuda.Stream()
    a.set_stream(s.cuda_stream); b.set_stream(s.cuda_stream)
    c = pkg.Context(device_id=0); c.set_stream(s.cuda_stream)
    with pytest.raises(pkg.SRPSError) as ei:
        pkg.Context.strip_group_solve([a, c])                   # c has no system
    assert ei.value.code == 3
    with torch.cuda.stream(s):
        pkg.Context.strip_group_solve([a, b])                   # and this works
    np.testing.assert_array_equal(a.get("z"), b.get("z"))
    for x in (a, b, c):
        x.close()


def _hosted_worker(rank, world, port, h, w, sf, kind, seed, out_dir):
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
    sc = pkg.synth.make_scene(h, w, sf, 3, seed=seed, mask_kind=kind)
    ctx = pkg.Context(device_id=0)
    ctx.set_option("cg_resident", 0)
    ctx.set_option("cg_partition", 1)
    tr = strips.HostedTransport(ctx, dist)
    ctx.setup(pkg.DataHandler.from_scene(sc))
    ctx.lighting(); ctx.albedo()
    assert ctx.get_option("cg_partition_active") == 1
    e = ctx.depth()
    assert not tr.errors, tr.errors
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), e=e, z=ctx.get("z"), it=ctx.last_cg_iterations()["depth"])
    dist.barrier()
    tr.remove()
    ctx.close()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("h,w,sf,kind,world", [(96, 80, 2, "ragged", 2), (512, 384, 4, "ellipse", 3)])
def test_strips_over_two_processes_and_the_callers_transport(pkg, tmp_path, h, w, sf, kind, world):
    """the strips through srps_depth (option cg_partition) with the collectives supplied by the caller (srps_set_strip_transport):
    here torch.distributed over gloo between PROCESSES that share the GPU -- the multi-process protocol (views, halo columns,
    totals, final gather) with the real kernels, which RCCL cannot run on a one-GPU box"""
    import socket
    import torch
    import torch.multiprocessing as mp
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    seed = h + 3 * w
    mp.spawn(_hosted_worker, args=(world, port, h, w, sf, kind, seed, str(tmp_path)), nprocs=world, join=True)
    sc = pkg.synth.make_scene(h, w, sf, 3, seed=seed, mask_kind=kind)
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        e1, z1, it1 = _pass_single(pkg, pkg.DataHandler.from_scene(sc), stream.cuda_stream)
    r = [np.load(tmp_path / f"rank{q}.npz") for q in range(world)]
    for q in range(1, world):
        np.testing.assert_array_equal(r[q]["z"], r[0]["z"])
        assert float(r[q]["e"]) == float(r[0]["e"])
    print(f"{world} processes, gloo transport: depth RMSE vs single grid {rmse(r[0]['z'], z1):.3e}")
    assert int(r[0]["it"]) == it1
    assert rmse(r[0]["z"], z1) < 2e-5
    assert abs(float(r[0]["e"]) - e1) <= 1e-3 * abs(e1)


def _abort_worker(rank, world, port, inject, out_dir):
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
    sc = pkg.synth.make_scene(128, 96, 2, 3, seed=91, mask_kind="ellipse")
    ctx = pkg.Context(device_id=0)
    ctx.set_option("cg_resident", 0)
    ctx.set_option("cg_partition", 1)
    ctx.set_option("albedo_mode", 0)                       # the reference's albedo CG: the persistent kernel the injected abort belongs to
    ctx.set_option("albedo_persistent", 1 if inject else 0)
    tr = strips.HostedTransport(ctx, dist)
    ctx.setup(pkg.DataHandler.from_scene(sc))
    out = {}
    for it in range(2):                                   # the second pass starts from a depth that differs from the first solve's start
        ctx.lighting(); ctx.albedo()
        if inject and it == 1:
            assert ctx.get_option("albedo_persistent") == 1
            ctx.set_option("debug_inject_abort", 2)        # "another rank's persistent albedo CG gave up": found at the pass's one wait
        out[f"e{it}"] = ctx.depth()
        ctx.normals()
        out[f"z{it}"] = ctx.get("z"); out[f"rho{it}"] = ctx.get("rho")
    out["fb"] = ctx.get_option("persistent_fallbacks")
    assert not tr.errors, tr.errors
    np.savez(os.path.join(out_dir, f"inj{inject}_rank{rank}.npz"), **out)
    dist.barrier()
    tr.remove()
    ctx.close()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_albedo_abort_with_strips_repeats_from_the_pass_start_plane(pkg, tmp_path):
    """round-3 advisor finding: with cg_partition = 1 the strips update x in place, and an albedo abort found at the end of the pass
    (here injected, as if reported by another rank) must repeat the solve from the iterate the pass STARTED with -- the copy of
    that plane is now made before the strips run.  Two processes over gloo; the repeated pass must equal a pass that never used the
    persistent albedo kernel (the albedo CG converges to the same fixed point from the aborted launch's values: 1e-6), where the
    defect restarted the CG from the discarded solve's depth (1e-3 apart on this scene)."""
    import socket
    import torch.multiprocessing as mp
    for inject in (0, 1):
        sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
        mp.spawn(_abort_worker, args=(2, port, inject, str(tmp_path)), nprocs=2, join=True)
    ref = np.load(tmp_path / "inj0_rank0.npz"); got = [np.load(tmp_path / f"inj1_rank{q}.npz") for q in range(2)]
    assert int(got[0]["fb"]) == 1 and int(ref["fb"]) == 0
    np.testing.assert_array_equal(got[0]["z1"], got[1]["z1"])                  # the ranks stay replicas
    d_first = rmse(got[0]["z0"], ref["z0"]); d = rmse(got[0]["z1"], ref["z1"]); moved = rmse(ref["z1"], ref["z0"])
    print(f"strips + injected albedo abort: pass 1 depth RMSE vs never-persistent {d:.3e} (pass 0: {d_first:.3e}; the pass moved the depth by {moved:.3e})")
    assert d < 2e-6 and d < 0.05 * moved
    # (the albedo only to 5e-4: rounding differences of pass 0's depth reach pass 1's normals through the finite differences times the
    # focal length -- tests/test_mitten_full.py; measured 1.1e-4 on the worst pixel)
    assert np.abs(got[0]["rho1"] - ref["rho1"]).max() < 5e-4
    assert abs(float(got[0]["e1"]) - float(ref["e1"])) <= 1e-5 * abs(float(ref["e1"]))


# ------------------------------------------------------------------------------------------------
# round 4: the RESIDENT kernel on the strips (srps_strip_group_solve_resident)
# ------------------------------------------------------------------------------------------------
def _resident_single(pkg, dh, tile=512):
    c = pkg.Context(device_id=0)
    c.set_option("cg_resident_tile", tile)                 # the tile shape the strip group is made to run too (512: 256 x 64 tiles)
    c.setup(dh)
    c.lighting(); c.albedo()
    e = c.depth()
    assert c.get_option("cg_resident_active") == 1 and c.get_option("persistent_fallbacks") == 0
    out = (e, c.get("z"), c.last_cg_iterations()["depth"], c.get_option("cg_resident_rect_active"))
    c.close()
    return out


def _resident_group(pkg, dh, world, tile=512):
    ctxs = []
    for _ in range(world):
        c = pkg.Context(device_id=0)                       # every context keeps its OWN stream: the launches must run side by side
        c.set_option("cg_resident_tile", tile)
        c.set_option("spin_budget_ms", 1000)               # the launches start one after the other: the first waits for the last
        c.setup(dh)
        ctxs.append(c)
    for c in ctxs:
        c.lighting(); c.albedo(); c.depth_partial()
    for c in ctxs:
        c.synchronize()
    pkg.Context.strip_group_solve_resident(ctxs)
    out = []
    for c in ctxs:
        c.energy_partial(); c.normals()
        e = c.energy_finish()
        out.append((e, c.get("z"), c.last_cg_iterations()["depth"]))
        c.close()
    return out


@pytest.mark.timeout(900)
@pytest.mark.parametrize("h,w,sf,kind,world,tile", [(1024, 2048, 4, "full", 1, 512), (1024, 2048, 4, "full", 2, 512), (1024, 2048, 4, "full", 4, 512), (1024, 1536, 2, "ellipse", 3, 512),
                                                    (2048, 2048, 4, "full", 2, 512), (768, 1280, 1, "ragged", 2, 512),
                                                    # the other tile shapes: 256 x 32 with 512 and with 256 threads, 256 x 16 with 256 threads (sf 4) and with 512 (sf <= 2)
                                                    (1024, 1024, 4, "full", 2, 32), (1024, 1024, 2, "ellipse", 3, 256), (512, 1024, 4, "full", 2, 16), (512, 768, 2, "ellipse", 3, 2),
                                                    (768, 640, 1, "ragged", 2, 2)])
def test_resident_kernel_on_strips_equals_the_single_resident_launch_bit_for_bit(pkg, h, w, sf, kind, world, tile):
    """The resident depth CG as `world` launches, one per context, each on its own range of 256 x 64 tile columns, side by side on this
    one device and exchanging the three sums of a step and the border tiles' edge columns through each other's memory while they run
    (the multi-GPU form of the kernel, with ordinary device memory in the place of peer memory): every block computes what it
    computes in the single launch and the grid-wide sums are added in the single launch's order -- the depth, the energy and the
    step count are those of srps_depth_solve on one context, bit for bit, on every rank."""
    sc = pkg.synth.make_scene(h, w, sf, 3, seed=h + 3 * w + sf, mask_kind=kind)
    dh = pkg.DataHandler.from_scene(sc)
    e1, z1, it1, rect1 = _resident_single(pkg, dh, tile)
    group = _resident_group(pkg, dh, world, tile)
    for e, z, it in group:
        assert it == it1 == 101
        np.testing.assert_array_equal(z, z1)
        assert e == e1
    print(f"{h}x{w} sf {sf} {kind}, tile option {tile}: {world} resident strips == the single resident launch (rect body {rect1})")


def test_resident_strip_group_refuses_what_it_cannot_run(pkg):
    sc = pkg.synth.make_scene(256, 256, 2, 2, seed=3, mask_kind="full")
    dh = pkg.DataHandler.from_scene(sc)
    import torch
    stream = torch.cuda.Stream()
    ctxs = []
    for _ in range(2):
        c = pkg.Context(device_id=0); c.set_stream(stream.cuda_stream); c.setup(dh); c.lighting(); c.albedo(); c.depth_partial(); ctxs.append(c)
    with pytest.raises(Exception) as ei:                    # a shared stream: the two launches could never run side by side
        pkg.Context.strip_group_solve_resident(ctxs)
    assert "share a stream" in str(ei.value)
    for c in ctxs:
        c.close()
    ctxs = []
    for _ in range(5):                                      # five launches of one process: the runtime has four hardware queues for them
        c = pkg.Context(device_id=0); c.setup(dh); c.lighting(); c.albedo(); c.depth_partial(); ctxs.append(c)
    with pytest.raises(Exception) as ei:
        pkg.Context.strip_group_solve_resident(ctxs)
    assert "hardware queues" in str(ei.value)
    for c in ctxs:
        c.close()
    sc = pkg.synth.make_scene(256, 32, 2, 2, seed=3, mask_kind="full")       # two columns of 256 x 16 tiles cannot be dealt to three ranks
    dh = pkg.DataHandler.from_scene(sc)
    ctxs = []
    for _ in range(3):
        c = pkg.Context(device_id=0); c.setup(dh); c.lighting(); c.albedo(); c.depth_partial(); ctxs.append(c)
    with pytest.raises(Exception) as ei:
        pkg.Context.strip_group_solve_resident(ctxs)
    assert "columns of tiles" in str(ei.value)
    for c in ctxs:
        c.close()


def _ipc_worker(rank, world, port, h, w, sf, kind, seed, tile, out_dir):
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
    sc = pkg.synth.make_scene(h, w, sf, 3, seed=seed, mask_kind=kind)
    ctx = pkg.Context(device_id=0)
    ctx.set_option("cg_resident_tile", tile)
    ctx.set_option("cg_partition", 2)                      # the resident kernel on this rank's strip of tile columns
    ctx.set_option("spin_budget_ms", 2000)                 # two processes start their launches a few milliseconds apart
    hc = strips.HostedCollectives(ctx, dist)               # all-reduce / broadcast over gloo: carries the hipIpc handles, the barrier, the strips of x
    ctx.setup(pkg.DataHandler.from_scene(sc))
    out = {}
    for it in range(2):                                    # the second solve reuses the mapped buffers
        ctx.lighting(); ctx.albedo()
        out[f"e{it}"] = ctx.depth()
        ctx.normals()
        out[f"z{it}"] = ctx.get("z")
    out["resident"] = ctx.get_option("cg_partition_resident_active"); out["fb"] = ctx.get_option("persistent_fallbacks")
    out["it"] = ctx.last_cg_iterations()["depth"]
    out["fine"] = ctx.get_option("exchange_buffer_fine")
    assert not hc.errors, hc.errors
    np.savez(os.path.join(out_dir, f"ipc_rank{rank}.npz"), **out)
    dist.barrier()
    hc.remove()
    ctx.close()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("h,w,sf,kind,world,tile", [(1024, 2048, 4, "full", 2, 512), (1024, 1536, 2, "ellipse", 3, 512), (1024, 1024, 4, "full", 2, 32), (512, 768, 2, "ellipse", 2, 2)])
def test_resident_strips_between_processes_through_ipc_mapped_buffers(pkg, tmp_path, h, w, sf, kind, world, tile):
    """cg_partition = 2 as a multi-GPU job would run it -- one PROCESS per rank, every rank's exchange buffer exported with
    hipIpcGetMemHandle and mapped by the others (the handles travel through the context's all-reduce), the ranks' resident kernels side
    by side for the whole solve, the strips of x broadcast afterwards -- with the ranks sharing this one device: two passes, the results
    of the single resident launch bit for bit on every rank."""
    import socket
    import torch.multiprocessing as mp
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    seed = h + 3 * w + sf
    mp.spawn(_ipc_worker, args=(world, port, h, w, sf, kind, seed, tile, str(tmp_path)), nprocs=world, join=True)
    sc = pkg.synth.make_scene(h, w, sf, 3, seed=seed, mask_kind=kind)
    c = pkg.Context(device_id=0)
    c.set_option("cg_resident_tile", tile)
    c.setup(pkg.DataHandler.from_scene(sc))
    ref = {}
    for it in range(2):
        c.lighting(); c.albedo(); ref[f"e{it}"] = c.depth(); c.normals(); ref[f"z{it}"] = c.get("z")
    c.close()
    for q in range(world):
        r = np.load(tmp_path / f"ipc_rank{q}.npz")
        assert int(r["resident"]) == 1 and int(r["fb"]) == 0 and int(r["it"]) == 101, (int(r["resident"]), int(r["fb"]), int(r["it"]))
        # the memory kind a multi-GPU job needs (coherent across devices while kernels run) is what is exported and mapped here too
        assert int(r["fine"]) == 1, "the exchange buffer is not fine-grained memory on this box"
        for it in range(2):
            np.testing.assert_array_equal(r[f"z{it}"], ref[f"z{it}"])
            assert float(r[f"e{it}"]) == ref[f"e{it}"]
    print(f"{h}x{w} sf {sf} {kind}: {world} processes, resident strips through hipIpc-mapped fine-grained buffers == the single resident launch")


def _thread_ranks(pkg, sc, world, tile, options=None, passes=2):
    """cg_partition = 2 with the ranks as THREADS of this process, each with a context of its own on device 0 and the library's
    collectives served by tests/_strip_protocol.ThreadCollectives -- the shape of `srps --gpus N` and srps_comm_init_all (one process, a
    host thread and a context per device), on a one-GPU box"""
    import threading
    import _strip_protocol as strips
    tc = strips.ThreadCollectives(world)
    out = [dict() for _ in range(world)]
    errs = []

    # The contexts are made here, one after the other: a context creates three streams, the runtime deals streams to its (four) hardware
    # queues in the order of their creation, and two resident launches that share a hardware queue run one AFTER the other -- never
    # side by side (seen once with three contexts created concurrently by the rank threads: a fall-back, correct results, but not the
    # path under test).  On a multi-GPU node every rank has a device, and with it the queues, to itself.
    import torch
    torch.cuda.set_device(0)
    # Each rank's context runs on a stream that has been SEEN to run side by side with the other ranks' streams: the runtime deals the
    # streams of a process to a few hardware queues (four per priority level, by use count), and two resident launches in one queue
    # run one after the other, never side by side -- late in a long session two of three fresh streams did land in one queue, one run
    # in three.  tools/libcu_holder.so: cu_streams_concurrent spins a one-block kernel on one stream and looks whether a kernel on
    # the other gets through meanwhile.  Candidates of both priority levels are made until `world` mutually concurrent ones are found.
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    aid = C.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "libcu_holder.so"))
    aid.cu_streams_concurrent.argtypes = [C.c_void_p, C.c_void_p]
    streams, spare = [], []
    for cand in range(12):
        sh = C.c_void_p()
        rc = hip.hipStreamCreateWithPriority(C.byref(sh), C.c_uint(1), C.c_int(0 if cand % 2 == 0 else -1))      # 1 = hipStreamNonBlocking
        assert rc == 0 and sh.value, rc
        if all(aid.cu_streams_concurrent(t, sh) == 1 and aid.cu_streams_concurrent(sh, t) == 1 for t in streams):
            streams.append(sh)
            if len(streams) == world:
                break
        else:
            spare.append(sh)                                 # kept alive: it holds its queue's use count where it is
    assert len(streams) == world, f"no {world} streams of this process run side by side ({len(spare)} candidates shared a queue)"
    ctxs = [pkg.Context(device_id=0) for _ in range(world)]
    for rank, ctx in enumerate(ctxs):
        ctx.set_stream(streams[rank].value)
        ctx.set_option("cg_resident_tile", tile)
        ctx.set_option("cg_partition", 2)
        ctx.set_option("spin_budget_ms", 2000)
        for k, v in (options or {}).items():
            ctx.set_option(k, v)
        tc.bind(pkg, ctx, rank)

    def rank_main(rank):
        ctx = ctxs[rank]
        try:
            torch.cuda.set_device(0)
            ctx.setup(pkg.DataHandler.from_scene(sc))
            o = out[rank]
            for it in range(passes):
                ctx.lighting(); ctx.albedo()
                o[f"e{it}"] = ctx.depth()
                ctx.normals()
                o[f"z{it}"] = ctx.get("z")
            o["resident"] = ctx.get_option("cg_partition_resident_active"); o["strips"] = ctx.get_option("cg_partition_active")
            o["fb"] = ctx.get_option("persistent_fallbacks"); o["it"] = ctx.last_cg_iterations()["depth"]
            o["err"] = pkg.last_error() if hasattr(pkg, "last_error") else ""
        except Exception as exc:
            errs.append((rank, repr(exc)))
            tc.bar.abort()
        finally:
            try:
                tc.unbind(pkg, ctx)
                ctx.close()
            except Exception as exc:
                errs.append((rank, "close: " + repr(exc)))
    th = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(600)
    assert not any(t.is_alive() for t in th), "a rank thread hangs"
    for sh in streams + spare:
        hip.hipStreamDestroy(sh)
    return out, errs, tc.errors


def _single_reference(pkg, sc, tile, passes=2):
    c = pkg.Context(device_id=0)
    c.set_option("cg_resident_tile", tile)
    c.setup(pkg.DataHandler.from_scene(sc))
    ref = {}
    for it in range(passes):
        c.lighting(); c.albedo(); ref[f"e{it}"] = c.depth(); c.normals(); ref[f"z{it}"] = c.get("z")
    c.close()
    return ref


@pytest.mark.timeout(900)
@pytest.mark.parametrize("h,w,sf,kind,world,tile", [(1024, 2048, 4, "full", 2, 512), (1024, 1536, 2, "ellipse", 3, 512)])
def test_resident_strips_between_ranks_of_one_process(pkg, h, w, sf, kind, world, tile):
    """Ranks that share a PROCESS (srps_comm_init_all, the C++ host's thread per device) must not go through hipIpcOpenMemHandle -- HIP
    does not open a handle in the process that exported it: the handshake carries the process id and the buffer's address, and a
    same-process peer is reached through its pointer.  Two / three host threads with a context each, bound by host collectives: the
    solve must END on the resident path (cg_partition_resident_active == 1) with the single launch's bits."""
    seed = h + 3 * w + sf
    sc = pkg.synth.make_scene(h, w, sf, 3, seed=seed, mask_kind=kind)
    ref = _single_reference(pkg, sc, tile)
    # Ranks of one process on ONE device share the runtime's hardware queues (four per priority level, dealt by use count; the runtime of
    # this image has two levels): late in a long session two of three ranks' streams can be dealt the same queue, and two resident
    # launches in one queue run one after the other -- the bounded waits then end the group and the pass is repeated without the resident
    # strips (correct results, checked below, but not the path under test).  A test-bed artefact: on a multi-GPU node every rank has its
    # device's queues to itself.  So: up to four attempts, each with fresh streams; every attempt must be RIGHT, one must be resident.
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    ballast = []
    attempts = []
    for attempt in range(4):
        out, errs, cerrs = _thread_ranks(pkg, sc, world, tile)
        assert not errs and not cerrs, (errs, cerrs)
        attempts.append([{k: r[k] for k in ("resident", "strips", "fb", "it")} for r in out])
        for q in range(world):
            assert out[q]["it"] == 101
            for it in range(2):
                assert rmse(out[q][f"z{it}"], ref[f"z{it}"]) < 2e-5       # whichever CG path ran
        if all(r["resident"] == 1 and r["fb"] == 0 for r in out):
            break
        sh = C.c_void_p(); hip.hipStreamCreateWithFlags(C.byref(sh), C.c_uint(1)); ballast.append(sh)      # shifts the queues' use counts
    for sh in ballast:
        hip.hipStreamDestroy(sh)
    assert all(r["resident"] == 1 and r["fb"] == 0 for r in out), attempts
    for q in range(world):
        r = out[q]
        for it in range(2):
            np.testing.assert_array_equal(r[f"z{it}"], ref[f"z{it}"])
            assert r[f"e{it}"] == ref[f"e{it}"]
    print(f"{h}x{w} sf {sf} {kind}: {world} ranks as threads of one process, resident strips through each other's pointers == the single resident launch"
          f" (attempts: {attempts})")


def _ipc_refusal_scenario(option="debug_ipc_same_process"):
    """child process of the tests below: prints one JSON line"""
    import importlib
    import json
    pkg = importlib.import_module("srmeetsps-cuda_amd")
    sc = pkg.synth.make_scene(1024, 2048, 4, 3, seed=99, mask_kind="full")
    out, errs, cerrs = _thread_ranks(pkg, sc, 2, 512, options={option: 1}, passes=1)
    ref = _single_reference(pkg, sc, 512, passes=1)
    print("SCENARIO" + json.dumps({"errs": errs, "cerrs": cerrs, "it": [r.get("it") for r in out], "resident": [r.get("resident") for r in out],
                                   "strips": [r.get("strips") for r in out], "rmse": [rmse(r["z0"], ref["z0"]) for r in out],
                                   "ranks_equal": bool(np.array_equal(out[0]["z0"], out[1]["z0"]))}), flush=True)


@pytest.mark.timeout(900)
@pytest.mark.parametrize("option", ["debug_ipc_same_process", "debug_foreign_pid_twin"])
def test_in_process_ipc_mapping_failure_is_recognised(option):
    """With `debug_ipc_same_process` the ranks of one process map each other's buffers through the hipIpc handles, as round 4 did: HIP
    refuses (a handle is opened by OTHER processes only).  The failure must be recognised by all ranks together -- no hang, no wrong
    result -- and the solve goes on without the resident strips (the streaming strips where the ranks have a neighbour transport, else
    the replicated CG) with the same result to rounding.  Run in a child process: a runtime that has refused a mapping has been seen to
    crash when the process ends (after every result was delivered), which must not take the test session with it.
    `debug_foreign_pid_twin` (round 6, advisor finding): every rank's handshake record carries a process number of its own while the pids are
    equal -- what two ranks in separate PID namespaces look like.  A pid match alone must NOT put a peer's raw address into a kernel: the
    ranks take the handle route (refused here, because they really are one process), recognise it together and go on as above."""
    import json
    import subprocess
    import sys
    code = ("import sys, os; sys.path.insert(0, %r); sys.path.insert(0, %r); sys.path.insert(0, %r); import test_gpu_strips as T; T._ipc_refusal_scenario(%r)"
            % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)),
               os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"), option))
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=800)
    line = [ln for ln in res.stdout.splitlines() if ln.startswith("SCENARIO")]
    assert line, (res.returncode, res.stdout[-1500:], res.stderr[-1500:])
    r = json.loads(line[0][len("SCENARIO"):])
    assert not r["errs"] and not r["cerrs"], r
    assert r["it"] == [101, 101] and r["resident"][0] == r["resident"][1] and r["strips"][0] == r["strips"][1] and r["ranks_equal"], r
    if r["resident"][0] == 1:                                   # a runtime that does open its own handles: then it must simply be right
        assert max(r["rmse"]) == 0.0, r
    else:
        assert max(r["rmse"]) < 2e-5, r                         # another CG path took over, on every rank alike
    print("in-process hipIpcOpenMemHandle:", "opened (runtime allows it)" if r["resident"][0] == 1 else
          "refused, recognised by both ranks; " + ("streaming strips" if r["strips"][0] else "replicated CG") + " took over",
          "; child exit code", res.returncode)
