"""GPU parity at BASELINE.json's sizes against the C oracle (oracle/srps_oracle.c, OpenMP: it finishes a 2048 x 2048
depth step in seconds on the GPU box's host cores):

  * 2048 x 2048, sf 4 -- the metric's grid: the kernel the bench times (k_cg_resident, 256 tiles of 256 x 64, every CU),
    with one and with two grid-wide waits per CG step, against the streaming kernels and against the oracle's
    assembled-CSR CG (the reference's formulation, devicecalls.cu:229-279, 743-759);
  * 1024 x 1024, sf 4, 20 images (BASELINE.json configs[2]): one whole alternating pass against the oracle;
  * 4096 x 4096, sf 2 (configs[4]'s grid): the streaming kernels (the grid needs more tiles than the chip has CUs)
    against the oracle's matrix-free CG, plus the operator properties.

Tolerances: depth RMSE < 1e-4 against the oracle (north_star), < 2e-5 between two of our own kernel paths.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
f32 = np.float32


def rmse(a, b):
    return float(np.sqrt(np.mean((np.asarray(a, np.float64) - np.asarray(b, np.float64)) ** 2)))


@pytest.fixture(scope="module")
def coracle():
    import c_oracle
    return c_oracle


def _lr_compact(sc, st, v_lr):
    """masked LR vector in the order of KT's rows (SRPS.cu:237-239): block b -> LR linear index of its first pixel"""
    if st.Ps == 0:
        return np.zeros(0, f32)
    first = st.imask[st.blk_pix.reshape(st.Ps, -1)[:, 0]]
    j, i = first // sc.h, first % sc.h
    return np.ascontiguousarray(np.asarray(v_lr, f32)[(j // sc.sf) * (sc.h // sc.sf) + (i // sc.sf)])


def _oracle_start(sc, oracle, coracle):
    """SRPS.cu:151-270 with the C oracle's structure: compaction, initial values, first normals"""
    st = coracle.Structure(sc.h, sc.w, sc.sf, sc.mask)
    I = sc.I if st.P == sc.h * sc.w else np.ascontiguousarray(sc.I[:, :, st.imask])      # full mask: no second copy (12.9 GB at 4096^2 x 64)
    z = np.ascontiguousarray(sc.z_init[st.imask])
    K = np.asarray(sc.K, f32)
    xx = ((st.imask // sc.h).astype(f32) - K[6]).astype(f32)
    yy = ((st.imask % sc.h).astype(f32) - K[7]).astype(f32)
    zx, zy = coracle.gradient(st, z)
    N, dz = oracle.normal_init(z, zx, zy, xx, yy, K[0], K[4])
    return st, dict(I=I, z=z, xx=xx, yy=yy, N=N, dz=dz, z0s=_lr_compact(sc, st, sc.zs_lr), fx=float(K[0]), fy=float(K[4]))


def _depth_three_ways(pkg, sc, variants):
    """the state after lighting + albedo, and z / energy / iterations of one depth phase per variant (a dict of options)"""
    dh = pkg.DataHandler.from_scene(sc)
    # the state is read from a context of its own: handing out state arrays makes a context drop the image sums its albedo
    # sweep left for the depth assembly (the caller might write through the pointer), and the assembly then takes its other,
    # differently rounded route -- the variants below must all take the same one
    ctx = pkg.Context(device_id=0)
    ctx.setup(dh)
    ctx.lighting(); ctx.albedo()
    state = {k: ctx.get(k) for k in ("s", "rho", "dz", "z", "xx", "yy", "z0s")}
    ctx.close()
    out = {}
    for name, opts in variants.items():
        ctx = pkg.Context(device_id=0)
        for k, v in opts.items():
            ctx.set_option(k, v)
        ctx.setup(dh)
        ctx.lighting(); ctx.albedo()
        e = ctx.depth()
        out[name] = dict(e=e, z=ctx.get("z"), it=ctx.last_cg_iterations()["depth"], resident=ctx.get_option("cg_resident_active"))
        ctx.close()
    return state, out


def test_metric_size_depth_phase_resident_streaming_and_oracle(pkg, oracle, coracle):
    """2048 x 2048, sf 4, 4 images, 3 channels, full mask: exactly the launch the bench times"""
    sc = pkg.synth.make_scene(2048, 2048, 4, 4, seed=1237, mask_kind="full")
    variants = {"resident_one_wait": dict(cg_resident=1, cg_resident_tile=512, cg_one_sync=1),
                "resident_two_waits": dict(cg_resident=1, cg_resident_tile=512, cg_one_sync=0),
                "resident_general_body": dict(cg_resident=1, cg_resident_tile=512, cg_one_sync=1, cg_resident_rect=0),
                "streaming": dict(cg_resident=0)}
    state, out = _depth_three_ways(pkg, sc, variants)
    assert out["resident_one_wait"]["resident"] == 1 and out["resident_two_waits"]["resident"] == 1 and out["streaming"]["resident"] == 0
    # every tile of the full frame takes the body without structure bits (resident_body<.., RECT>); it performs the general
    # body's operations minus additions of zero: identical depth
    np.testing.assert_array_equal(out["resident_one_wait"]["z"], out["resident_general_body"]["z"])
    assert out["resident_one_wait"]["e"] == out["resident_general_body"]["e"]
    assert all(o["it"] == 101 for o in out.values()), {k: o["it"] for k, o in out.items()}
    # the oracle's depth step from the same state (assembled CSR + the reference's CG)
    st = coracle.Structure(sc.h, sc.w, sc.sf, sc.mask)
    P = st.P
    assert P == 2048 * 2048
    I = np.ascontiguousarray(sc.I[:, :, st.imask])
    z_ref = state["z"].copy()
    e_ref, it_ref = coracle.depth_estimation(st, state["s"].reshape(-1, 3, 4), state["rho"].reshape(3, P), I, state["xx"], state["yy"],
                                             state["dz"], state["z0s"], z_ref, float(sc.K[0]), float(sc.K[4]), assembled=True)
    assert it_ref == 101
    z_mf = state["z"].copy()
    e_mf, _ = coracle.depth_estimation(st, state["s"].reshape(-1, 3, 4), state["rho"].reshape(3, P), I, state["xx"], state["yy"],
                                       state["dz"], state["z0s"], z_mf, float(sc.K[0]), float(sc.K[4]), assembled=False)
    report = {k: (rmse(o["z"], z_ref), rmse(o["z"], z_mf), o["e"]) for k, o in out.items()}
    print("2048^2 depth phase: RMSE vs assembled oracle, vs matrix-free oracle, energy:", report, "oracle energies", e_ref, e_mf,
          "oracle assembled vs matrix-free", rmse(z_ref, z_mf))
    for k, o in out.items():
        assert rmse(o["z"], z_ref) < 1e-4, report
        assert abs(o["e"] - e_ref) <= 1e-3 * abs(e_ref), report
    # our own paths among each other: the drift of the predicted r.r at full size, and resident against streaming
    assert rmse(out["resident_one_wait"]["z"], out["resident_two_waits"]["z"]) < 2e-5, report
    assert rmse(out["resident_one_wait"]["z"], out["streaming"]["z"]) < 2e-5, report
    assert rmse(out["resident_two_waits"]["z"], out["streaming"]["z"]) < 2e-5, report


def _one_pass_vs_oracle(pkg, oracle, coracle, sc, expect_resident=True, e_tol=2e-3, keep=None, albedo_cg=False):
    """lighting -> albedo -> depth -> normals with default options against numpy (lighting dc.cu:376-444, albedo dc.cu:395-406 +
    513-548 on the diagonal system) and C (depth).  The library's default albedo step is the fixed point of the reference's CG formed
    inside the sweep (SRPS_ALBEDO_AUTO, include/srps.h); albedo_cg selects the CG itself, whose step counts are then compared too."""
    n_img, n_ch = sc.n_img, sc.n_ch
    ctx = pkg.Context(device_id=0)
    if albedo_cg:
        ctx.set_option("albedo_mode", 0)
    ctx.setup(pkg.DataHandler.from_scene(sc))
    en = pkg.alternating_loop(ctx, None, max_outer=1)
    z = ctx.get("z"); rho = ctx.get("rho").reshape(n_ch, -1); s = ctx.get("s").reshape(-1, n_ch, 4); Nrm = ctx.get("N").reshape(4, -1)
    iters = ctx.last_cg_iterations()
    assert ctx.get_option("cg_resident_active") == (1 if expect_resident else 0)
    assert ctx.get_option("persistent_fallbacks") == 0
    ctx.close()
    if keep is not None:
        keep.update(z=z, rho=rho, s=s, energy=en[0])
    st, o = _oracle_start(sc, oracle, coracle)
    P = st.P
    s_ref = np.zeros((n_img, n_ch, 4), f32); s_ref[:, :, 2] = -1
    rho_ref = np.full((n_ch, P), 0.5, f32)
    oracle.lighting_estimation(s_ref, rho_ref, o["N"], o["I"])
    num, den = oracle.albedo_numden(s_ref, o["N"], o["I"])
    alb_it = []
    oracle.albedo_solve_numden(rho_ref, num, den, alb_it)
    z_ref = o["z"].copy()
    e_ref, it_ref = coracle.depth_estimation(st, s_ref, rho_ref, o["I"], o["xx"], o["yy"], o["dz"], o["z0s"], z_ref, o["fx"], o["fy"], assembled=True)
    # normals (dc.cu:171-223) of the depth the GPU solved: N = (fx zx, fy zy, ...) / |.| multiplies depth differences by the
    # focal length (1229 at 1024^2), so they are compared on the same z rather than through the two solves
    zx, zy = coracle.gradient(st, z)
    N_ref, _ = oracle.normal_init(z, zx, zy, o["xx"], o["yy"], o["fx"], o["fy"])
    print(f"{sc.h}x{sc.w}, {n_img} images, one pass: depth RMSE", rmse(z, z_ref), "albedo max", np.abs(rho - rho_ref).max(), "energy", en[0], e_ref,
          "albedo CG", iters["albedo"][:n_ch], alb_it)
    assert iters["depth"] == it_ref == 101
    if albedo_cg:
        assert all(abs(a - b) <= 2 for a, b in zip(iters["albedo"][:n_ch], alb_it))
    else:
        assert list(iters["albedo"][:n_ch]) == [0] * n_ch                  # no CG ran: the fixed point came out of the sweep
    assert rmse(z, z_ref) < 1e-4
    assert np.abs(rho - rho_ref).max() < 1e-4
    # first pass.  Measured (round 3): 4.3e-6 at 1024 x 1024 x 20 images, 2.1e-4 at 2048 x 2048 x 40, 3.1e-4 at
    # 512 x 384 x 45 images, 2.4e-6 and 4.2e-5 at 4096 x 4096 x 64 on two boxes -- the GPU's energy was the same bits both times, the
    # ORACLE's moved (its dot products were float sums per host thread then; since they are sums over fixed blocks the oracle gives the same energy on 8 and on 256 threads); the callers allow about three times
    # the largest value seen
    print("first-pass energy, relative deviation", abs(en[0] - e_ref) / abs(e_ref), "allowed", e_tol)
    assert abs(en[0] - e_ref) <= e_tol * abs(e_ref)
    # lighting through the shading it predicts (the first pass's 4 x 4 systems have a flat direction)
    A = (rho_ref[:, None, :] * o["N"][None, :, :]).astype(np.float64)                 # [c][4][P]
    for c in range(n_ch):
        d = (s[:, c, :] - s_ref[:, c, :]).astype(np.float64) @ A[c]
        r = s_ref[:, c, :].astype(np.float64) @ A[c]
        assert np.linalg.norm(d) / np.linalg.norm(r) < 1e-4
    assert np.abs(Nrm - N_ref).max() < 2e-5


# one-context results of the two large configurations, kept for the two-rank tests below (same module, same process)
_KEEP = {"config4": {}, "config5": {}}


@pytest.mark.parametrize("albedo_cg", [False, True])
def test_config3_whole_pass_against_the_oracle(pkg, oracle, coracle, albedo_cg):
    """1024 x 1024, sf 4, 20 images (BASELINE.json configs[2]), full mask; with the default albedo step and with the reference's CG"""
    _one_pass_vs_oracle(pkg, oracle, coracle, pkg.synth.make_scene(1024, 1024, 4, 20, seed=1236, mask_kind="full"), e_tol=1e-4, albedo_cg=albedo_cg)


@pytest.mark.timeout(1200)
def test_config4_all_images_on_one_gpu_whole_pass_against_the_oracle(pkg, oracle, coracle):
    """2048 x 2048, sf 4, 40 images (BASELINE.json configs[3], whose 8 GPUs hold 5 images each): the same data volume as one
    job on one GPU -- two lighting batches of 20, the resident CG at every CU, 2 GB of images"""
    _one_pass_vs_oracle(pkg, oracle, coracle, pkg.synth.make_scene(2048, 2048, 4, 40, seed=1241, mask_kind="full"), e_tol=6.5e-4, keep=_KEEP["config4"])


def test_three_lighting_batches_at_mid_size(pkg, oracle, coracle):
    """45 images (the lighting sweep takes them in batches of 20: three batches, the last one partial) on a 512 x 384 ellipse,
    sf 2: the image loops of every sweep at a size where a pixel range spans several blocks"""
    _one_pass_vs_oracle(pkg, oracle, coracle, pkg.synth.make_scene(512, 384, 2, 45, seed=1240, mask_kind="ellipse"), e_tol=1e-3, albedo_cg=True)


def test_largest_grid_streaming_kernels_against_the_oracle(pkg, oracle, coracle):
    """4096 x 4096, sf 2 (BASELINE.json configs[4]'s grid; 2 images keep the host side of the test small): 4 x the tiles
    the chip has CUs, so the depth CG streams its vectors; against the oracle's matrix-free CG, and the operator is
    symmetric, positive and linear at this size"""
    import torch
    sc = pkg.synth.make_scene(4096, 4096, 2, 2, seed=1238, mask_kind="full")
    ctx = pkg.Context(device_id=0)
    ctx.setup(pkg.DataHandler.from_scene(sc))
    ctx.lighting(); ctx.albedo()
    state = {k: ctx.get(k) for k in ("s", "rho", "dz", "z", "xx", "yy", "z0s")}
    e = ctx.depth()
    assert ctx.get_option("cg_resident_active") == 0
    assert ctx.last_cg_iterations()["depth"] == 101
    z = ctx.get("z")
    P = ctx.dims()["npix"]
    assert P == 4096 * 4096
    g = torch.Generator(device="cuda").manual_seed(2)
    x = torch.randn(P, device="cuda", generator=g); y = torch.randn(P, device="cuda", generator=g)
    Ax = torch.empty_like(x); Ay = torch.empty_like(x); Az = torch.empty_like(x)
    ctx.depth_operator_apply(x, P, Ax); ctx.depth_operator_apply(y, P, Ay)
    ctx.depth_operator_apply(0.5 * x - 2.0 * y, P, Az)
    ctx.synchronize()
    xAx = torch.dot(x.double(), Ax.double()).item()
    assert xAx > 0
    assert abs(torch.dot(x.double(), Ay.double()).item() - torch.dot(y.double(), Ax.double()).item()) / xAx < 1e-5
    lin = 0.5 * Ax - 2.0 * Ay
    assert (torch.linalg.norm((Az - lin).double()) / torch.linalg.norm(lin.double())).item() < 1e-5
    ctx.close()
    del x, y, Ax, Ay, Az, lin
    st = coracle.Structure(sc.h, sc.w, sc.sf, sc.mask)
    I = np.ascontiguousarray(sc.I[:, :, st.imask])
    z_ref = state["z"].copy()
    e_ref, it_ref = coracle.depth_estimation(st, state["s"].reshape(-1, 3, 4), state["rho"].reshape(3, P), I, state["xx"], state["yy"],
                                             state["dz"], state["z0s"], z_ref, float(sc.K[0]), float(sc.K[4]), assembled=False)
    print("4096^2 depth phase: RMSE vs oracle", rmse(z, z_ref), "energy", e, e_ref)
    assert it_ref == 101
    assert rmse(z, z_ref) < 1e-4
    assert abs(e - e_ref) <= 1e-3 * abs(e_ref)


# ------------------------------------------------------------------------------------------------
# BASELINE.json configs[3] and configs[4] at their data volume, and sharded over two ranks at that volume
# ------------------------------------------------------------------------------------------------
@pytest.mark.timeout(3000)
def test_config5_full_volume_whole_pass_against_the_oracle(pkg, oracle, coracle):
    """4096 x 4096, sf 2, 64 images (BASELINE.json configs[4]: 12.9 GB of images, 16.8 M unknowns) as one job on one GPU: the
    per-image compaction of SRPS.cu:223-234 through the 2 GB staging ring of srps_setup, four lighting batches, the streaming
    depth CG -- one whole pass against the oracle (numpy lighting + albedo, C depth step in the reference's assembled-CSR form)"""
    sc = pkg.synth.make_scene(4096, 4096, 2, 64, seed=1242, mask_kind="full")
    _one_pass_vs_oracle(pkg, oracle, coracle, sc, expect_resident=False, e_tol=2e-4, keep=_KEEP["config5"])


def _two_rank_worker(rank, world, port, H, W, sf, n_img, seed, out_dir):
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
    lo, hi = pkg.shard_range(n_img, world, rank)
    sc = pkg.synth.make_scene(H, W, sf, n_img, seed=seed, mask_kind="full", img_begin=lo, img_end=hi)
    srps = pkg.SRPS(pkg.DataHandler.from_scene(sc), distributed=True)
    # two processes share the device: the persistent kernels of both must be resident side by side, or give up and stream --
    # either way the result has to agree; cooperative launches (the default) keep them from interleaving
    en = srps.execute(max_outer=1)
    ex = {k: srps.ctx.exchange_ptr(k)[1] for k in ("albedo", "depth", "energy", "s")}
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), energies=np.array(en), z=srps.z(), rho=srps.rho(), s=srps.s(),
             ex_albedo=ex["albedo"], ex_depth=ex["depth"], ex_s=ex["s"], fallbacks=srps.ctx.get_option("persistent_fallbacks"),
             iters=srps.ctx.last_cg_iterations()["depth"])
    dist.barrier()
    srps.ctx.close()
    dist.destroy_process_group()


def _two_ranks_equal_one_context(tmp_path, key, H, W, sf, n_img, seed):
    import socket
    import torch.multiprocessing as mp
    one = _KEEP[key]
    if not one:
        pytest.skip("the one-context pass of this configuration did not run in this session")
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    mp.spawn(_two_rank_worker, args=(2, port, H, W, sf, n_img, seed, str(tmp_path)), nprocs=2, join=True)
    r0 = np.load(tmp_path / "rank0.npz"); r1 = np.load(tmp_path / "rank1.npz")
    P = H * W
    # what travels: num [C][P] and the compact q [3][P], s [N][C][4]
    assert int(r0["ex_albedo"]) == 3 * P and int(r0["ex_depth"]) == 3 * P and int(r0["ex_s"]) == n_img * 3 * 4
    assert int(r0["iters"]) == int(r1["iters"]) == 101
    if int(r0["fallbacks"]) == int(r1["fallbacks"]):          # the replicas ran the same kernels: they stay bit-identical
        np.testing.assert_array_equal(r0["z"], r1["z"]); np.testing.assert_array_equal(r0["rho"], r1["rho"])
        np.testing.assert_array_equal(r0["energies"], r1["energies"])
    d_z = rmse(r0["z"], one["z"]); d_rho = float(np.abs(r0["rho"] - one["rho"]).max()); d_e = abs(float(r0["energies"][0]) - one["energy"]) / abs(one["energy"])
    print(f"{key}: two ranks ({n_img // 2} images each) against one context: depth RMSE {d_z:.3e}, albedo max {d_rho:.3e}, energy {d_e:.3e}, "
          f"fallbacks {int(r0['fallbacks'])} / {int(r1['fallbacks'])}")
    # measured: 2048 x 2048 x 40: 1.8e-6 / 3.6e-7 / 9.2e-5; 4096 x 4096 x 64: 2.1e-7 / 4.8e-7 / 2.5e-6 (another summation order of the images)
    assert d_z < 6e-6 and d_rho < 2e-6 and d_e < 3e-4


@pytest.mark.timeout(3000)
def test_config4_volume_two_ranks_on_one_gpu_equal_one_context(tmp_path):
    """2048 x 2048, sf 4, 40 images (configs[3]) as 2 ranks x 20 images on one GPU (gloo): shard_range, the *_partial / *_finish
    entry points, the all-reduce of num and of the compact q at their real size (50 MB each), against the one-context pass"""
    _two_ranks_equal_one_context(tmp_path, "config4", 2048, 2048, 4, 40, 1241)


@pytest.mark.timeout(3000)
def test_config5_volume_two_ranks_on_one_gpu_equal_one_context(tmp_path):
    """4096 x 4096, sf 2, 64 images (configs[4]) as 2 ranks x 32 images on one GPU (gloo): 201 MB of num and of q per exchange"""
    _two_ranks_equal_one_context(tmp_path, "config5", 4096, 4096, 2, 64, 1242)


@pytest.mark.parametrize("kind,n_img,seed", [("ellipse", 20, 4321), ("full", 40, 977)])
def test_one_wait_against_two_waits_and_streaming_on_further_scenes(pkg, kind, n_img, seed):
    """the drift bound of the one-wait form (r.r predicted from three products, anchored on direct sums) at the metric's size on
    scenes other than the four-image one above: an elliptical mask (the general body, 232 blocks) with 20 images and the full
    frame with 40 -- one wait, two waits and the streaming kernels among each other, all 101 steps"""
    sc = pkg.synth.make_scene(2048, 2048, 4, n_img, seed=seed, mask_kind=kind)
    variants = {"one": dict(cg_resident=1, cg_one_sync=1), "two": dict(cg_resident=1, cg_one_sync=0), "streaming": dict(cg_resident=0)}
    _, out = _depth_three_ways(pkg, sc, variants)
    assert out["one"]["resident"] == 1 and out["two"]["resident"] == 1 and out["streaming"]["resident"] == 0
    assert all(o["it"] == 101 for o in out.values())
    d = {f"{a}-{b}": rmse(out[a]["z"], out[b]["z"]) for a, b in (("one", "two"), ("one", "streaming"), ("two", "streaming"))}
    print(f"2048^2 {kind}, {n_img} images: depth RMSE between the paths", d, "energies", {k: o["e"] for k, o in out.items()})
    assert max(d.values()) < 2e-5
    assert abs(out["one"]["e"] - out["streaming"]["e"]) <= 1e-4 * abs(out["streaming"]["e"])
