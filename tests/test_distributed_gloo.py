"""world_size-2 run of the image-sharded alternating loop under gloo (CPU): the package's host
logic (shard_range, alternating_loop, which exchange buffers are all-reduced) driven with the
oracle-backed stand-in engine of tests/_oracle_engine.py, against the single-process result."""
import importlib
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, n_img, out_dir):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    pkg_api = importlib.import_module("srmeetsps-cuda_amd.api")
    synth = importlib.import_module("srmeetsps-cuda_amd.synth")
    import srps_oracle as O
    from _oracle_engine import OracleEngine
    lo, hi = pkg_api.shard_range(n_img, world, rank)
    sc = synth.make_scene(24, 20, 2, n_img, seed=17, mask_kind="ragged", img_begin=lo, img_end=hi)
    eng = OracleEngine(O, sc)

    def all_reduce(t):
        dist.all_reduce(t, op=dist.ReduceOp.SUM)

    energies = pkg_api.alternating_loop(eng, all_reduce, max_outer=3)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), energies=np.array(energies), z=eng.st.z, rho=eng.st.rho, s=eng.s)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_rank_gloo_equals_single_process(tmp_path, pkg, oracle):
    import torch.multiprocessing as mp
    n_img = 5                                   # uneven shards: 3 + 2
    port = _free_port()
    mp.spawn(_worker, args=(2, port, n_img, str(tmp_path)), nprocs=2, join=True)
    r0 = np.load(tmp_path / "rank0.npz"); r1 = np.load(tmp_path / "rank1.npz")
    # replicas are identical after every all-reduce
    np.testing.assert_array_equal(r0["z"], r1["z"]); np.testing.assert_array_equal(r0["s"], r1["s"])
    np.testing.assert_array_equal(r0["energies"], r1["energies"])
    # single process, all images, same engine
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from _oracle_engine import OracleEngine
    sc = pkg.synth.make_scene(24, 20, 2, n_img, seed=17, mask_kind="ragged")
    eng = OracleEngine(oracle, sc)
    e1 = pkg.alternating_loop(eng, None, max_outer=3)
    np.testing.assert_allclose(r0["energies"], e1, rtol=1e-3)       # partial sums are added in a different order
    assert np.sqrt(np.mean((r0["z"] - eng.st.z) ** 2)) < 1e-4
    assert np.abs(r0["rho"] - eng.st.rho).max() < 1e-3
    # and the stand-in engine itself follows the faithful oracle loop
    ref = oracle.execute(oracle.Problem(sc.h, sc.w, sc.sf, sc.mask, sc.K, sc.I, sc.zs_lr, sc.z_init), depth="faithful", max_outer=3)
    np.testing.assert_allclose(e1, ref.energies, rtol=1e-2)
    assert np.sqrt(np.mean((eng.st.z - ref.z) ** 2)) < 1e-4
