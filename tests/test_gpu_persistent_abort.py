"""The persistent kernels (depth CG k_cg_resident, albedo CG k_dcg_persistent*) must FAIL, not hang, when their grid cannot
become resident: a co-tenant holds CUs at the metric's size (2048 x 2048 = 256 tiles = every CU).  Every wait inside them is
bounded by the spin budget (device_utils.h); the launch then ends with an abort flag, stores nothing, and the library repeats
the phase with the streaming kernels -- same results as a context that streamed from the start.

The co-tenant (tools/cu_holder.hip) pins 16 KiB of LDS on some CUs for a few seconds: a block of k_cg_resident needs 157 of a
CU's 160 KiB, so those CUs cannot take one.  In-process (another stream) the contention is certain; as a second process it
depends on how the driver shares the device between processes, so that test only demands "no hang, right result"."""
import ctypes
import os
import subprocess
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOLDER_SO = os.path.join(ROOT, "tools", "libcu_holder.so")
HOLDER_BIN = os.path.join(ROOT, "tools", "cu_holder.bin")


@pytest.fixture(scope="module")
def scene_and_streaming_result(pkg):
    sc = pkg.synth.make_scene(2048, 2048, 4, 2, seed=1239, mask_kind="full")
    dh = pkg.DataHandler.from_scene(sc)
    ctx = pkg.Context(device_id=0)
    ctx.set_option("cg_resident", 0)
    ctx.set_option("albedo_mode", 0)                       # the reference's albedo CG: its persistent kernel is one of the two under test
    ctx.setup(dh)
    ctx.lighting(); ctx.albedo()
    e = ctx.depth()
    z = ctx.get("z")
    ctx.close()
    return dh, e, z


def _prepared_context(pkg, dh, budget_ms):
    ctx = pkg.Context(device_id=0)
    ctx.set_option("spin_budget_ms", budget_ms)
    ctx.set_option("albedo_mode", 0)
    ctx.setup(dh)
    ctx.lighting(); ctx.albedo(); ctx.depth_partial()      # the assembly decides which operator kernels apply
    ctx.synchronize()
    assert ctx.get_option("cg_resident_active") == 1 and ctx.get_option("persistent_fallbacks") == 0
    return ctx


@pytest.mark.skipif(not os.path.exists(HOLDER_SO), reason="tools/libcu_holder.so not built (make -C tools)")
@pytest.mark.parametrize("exclusive", [0, 1])
def test_held_cus_in_process_abort_and_fall_back(pkg, scene_and_streaming_result, exclusive):
    """cooperative (default) and plain launch alike: the launch is accepted, cannot become resident, gives up after the
    budget, and the pass is repeated by the streaming kernels with bit-identical results"""
    dh, e_ref, z_ref = scene_and_streaming_result
    holder = ctypes.CDLL(HOLDER_SO)
    ctx = _prepared_context(pkg, dh, budget_ms=40)
    ctx.set_option("exclusive_device", exclusive)
    assert holder.cu_holder_launch(64, 3000, 16) == 0
    time.sleep(0.2)                                        # the holder's blocks are resident now
    t0 = time.perf_counter()
    e = ctx.depth()
    dt = time.perf_counter() - t0
    fallbacks = ctx.get_option("persistent_fallbacks")
    z = ctx.get("z")
    msg = pkg.last_error()
    print(f"in-process co-tenant (exclusive_device={exclusive}): depth phase returned after {dt * 1e3:.0f} ms, fallbacks {fallbacks}, {msg}")
    if fallbacks == 0 and not exclusive:
        # A cooperative launch is dispatched as a gang: on some boxes the dispatcher holds it back until the co-tenant's CUs are
        # free instead of starting it partially -- no wait inside the kernel, no abort, the phase simply ends when the co-tenant
        # does (its 3 s here).  Not a hang either; the result is then the persistent kernel's.
        assert dt < 3.5 and ctx.get_option("cg_resident_active") == 1
        assert abs(e - e_ref) <= 1e-4 * abs(e_ref) and float(np.sqrt(np.mean((z - z_ref) ** 2))) < 2e-5
        assert holder.cu_holder_wait() == 0
        ctx.close()
        return
    # (a cooperative launch may also be held back by the dispatcher for most of the co-tenant's 3 s and THEN start short of blocks and
    # give up -- seen once in round 4: 2.8 s, fallbacks 1 --: bounded by the co-tenant's stay plus the budget, never a hang)
    # -- and round 4's boxes did the same to the PLAIN launch (2.8 s, fallbacks 1, results bit-identical): the blocks that had not been
    # dispatched yet were given a CU only when the co-tenant left.  Either way the phase is bounded by the co-tenant's stay plus the
    # budget and never hangs, which is what this test is about.
    assert dt < 3.5, "the phase must not outlast the co-tenant"
    assert fallbacks == 1 and ctx.get_option("cg_resident_active") == 0
    assert ctx.last_cg_iterations()["depth"] == 101
    assert e == e_ref
    np.testing.assert_array_equal(z, z_ref)
    # the context keeps working (streaming) while the co-tenant is still there, and afterwards
    e2 = ctx.depth()
    assert np.isfinite(e2) and ctx.get_option("persistent_fallbacks") == 1
    assert holder.cu_holder_wait() == 0
    ctx.close()
    # a fresh context on the free device uses the persistent kernel again and agrees with the streaming result
    ctx = _prepared_context(pkg, dh, budget_ms=200)
    e3 = ctx.depth()
    assert ctx.get_option("persistent_fallbacks") == 0 and ctx.get_option("cg_resident_active") == 1
    assert abs(e3 - e_ref) <= 1e-4 * abs(e_ref)
    assert float(np.sqrt(np.mean((ctx.get("z") - z_ref) ** 2))) < 2e-5
    ctx.close()


@pytest.mark.skipif(not os.path.exists(HOLDER_BIN), reason="tools/cu_holder.bin not built (make -C tools)")
def test_second_process_holding_cus_never_wedges_the_solve(pkg, scene_and_streaming_result):
    dh, e_ref, z_ref = scene_and_streaming_result
    ctx = _prepared_context(pkg, dh, budget_ms=40)
    proc = subprocess.Popen([HOLDER_BIN, "64", "3000", "16"], stdout=subprocess.PIPE, text=True)
    try:
        assert proc.stdout.readline().strip() == "holding"
        time.sleep(0.2)
        t0 = time.perf_counter()
        en = pkg.alternating_loop(ctx, None, max_outer=1)          # a whole pass: albedo and depth persistent kernels
        dt = time.perf_counter() - t0
        fallbacks = ctx.get_option("persistent_fallbacks")
        print(f"second process holding 64 CUs: pass returned after {dt * 1e3:.0f} ms, fallbacks {fallbacks}")
        assert dt < 10.0
        assert np.isfinite(en[0]) and ctx.last_cg_iterations()["depth"] == 101
        assert np.all(np.isfinite(ctx.get("z")))
    finally:
        proc.wait(timeout=30)
        ctx.close()
    assert proc.returncode == 0


def test_spin_budget_option_range(pkg):
    ctx = pkg.Context(device_id=0)
    assert ctx.get_option("spin_budget_ms") == 200 and ctx.get_option("coop_launch") == 1 and ctx.get_option("exclusive_device") == 0
    with pytest.raises(Exception):
        ctx.set_option("spin_budget_ms", 0)
    ctx.set_option("exclusive_device", 1)
    assert ctx.get_option("coop_launch") == 0
    assert ctx.get_option("host_wait_spin") == 1
    ctx.set_option("host_wait_spin", 0)
    assert ctx.get_option("host_wait_spin") == 0
    ctx.close()


def test_host_wait_sleeping_or_polling_same_results(pkg):
    """option host_wait_spin: how the host waits for the pass's energy does not touch the results"""
    sc = pkg.synth.make_scene(256, 192, 2, 4, seed=77, mask_kind="ellipse")
    out = []
    for spin in (1, 0):
        ctx = pkg.Context(device_id=0)
        ctx.set_option("host_wait_spin", spin)
        srps = pkg.SRPS(pkg.DataHandler.from_scene(sc), ctx=ctx)
        en = srps.execute(max_outer=3)
        out.append((np.array(en, np.float32), srps.z()))
        ctx.close()
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])
