"""bench.py's own launcher for N > 1, as far as a box without a GPU can check it: started bare with
--gpus 2 the parent must start the ranks as children (torch.distributed.run) WITHOUT importing torch itself -- a process that has
initialised the GPU must never start or replace another -- and hand their exit code on.  The ranks themselves fail here (no GPU)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_parent_starts_children_and_relays_their_failure(tmp_path):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    probe = tmp_path / "probe.py"
    # run bench.main() in-process with subprocess.Popen replaced by a recorder: which command, and was torch imported before it?
    probe.write_text(
        "import sys, json, subprocess, runpy\n"
        "rec = {}\n"
        "class P:\n"
        "    def __init__(self, cmd, **kw):\n"
        "        rec['cmd'] = cmd; rec['torch_imported'] = 'torch' in sys.modules; rec['env_flag'] = kw['env'].get('SRPS_BENCH_SELF_LAUNCHED')\n"
        "        self.stdout = iter(['{\"metric\": \"x\"}\\n'])\n"
        "    def wait(self): return 7\n"
        "subprocess.Popen = P\n"
        f"sys.argv = [{str(os.path.join(ROOT, 'bench.py'))!r}, '--gpus', '2', '--steps', '3']\n"
        "try:\n"
        f"    runpy.run_path({str(os.path.join(ROOT, 'bench.py'))!r}, run_name='__main__')\n"
        "except SystemExit as e:\n"
        "    rec['exit'] = e.code\n"
        "print('REC' + json.dumps(rec))\n")
    out = subprocess.run([sys.executable, str(probe)], env=env, capture_output=True, text=True, timeout=120, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    import json
    rec = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("REC")][0][3:])
    assert rec["torch_imported"] is False
    assert rec["exit"] == 7 and rec["env_flag"] == "1"
    cmd = rec["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "2"
    assert "127.0.0.1" in cmd and cmd[-4:] == ["--gpus", "2", "--steps", "3"]
    assert '{"metric": "x"}' in out.stdout                         # the ranks' line is relayed


def test_which_workload_a_command_line_means():
    """one GPU, nothing said: the metric's configuration; N > 1, nothing said: BASELINE.json configs[3] (40 images IN ALL, sharded --
    strong scaling); --config 5: configs[4] with the CG on strips; --images keeps the weak-scaling form"""
    import importlib.util
    import types
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)

    def pick(gpus, **kw):
        a = types.SimpleNamespace(gpus=gpus, images=None, images_total=None, config=None, size=None, sf=None, partition=None)
        for k, v in kw.items():
            setattr(a, k, v)
        return bench.select_workload(a)
    a = pick(1)
    assert (a.size, a.sf, a.images_total, a.scaling, a.partition) == (2048, 4, 20, "weak", "images") and "metric configuration" in a.workload
    a = pick(8)
    assert (a.size, a.sf, a.images_total, a.scaling) == (2048, 4, 40, "strong") and "(5, 5, 5, 5, 5, 5, 5, 5 per rank)" in a.workload and "configs[3]" in a.workload
    a = pick(2)
    assert a.images_total == 40 and "(20, 20 per rank)" in a.workload
    a = pick(4, config=5)
    assert (a.size, a.sf, a.images_total, a.partition) == (4096, 2, 64, "strips") and "configs[4]" in a.workload
    a = pick(4, config=5, partition="images")
    assert a.partition == "images"
    a = pick(2, images=20)
    assert (a.images_total, a.scaling) == (40, "weak") and "configs" not in a.workload
    a = pick(3, images_total=20)
    assert "(7, 7, 6 per rank)" in a.workload and a.scaling == "strong"
    a = pick(1, config=3)
    assert (a.size, a.sf, a.images_total) == (1024, 4, 20) and "configs[2]" in a.workload


def test_the_degrade_chain_of_a_multi_gpu_run_is_reported():
    """bench.py: describe_parallelism -- what `config.parallelism` and `config.degraded` say for every way a multi-GPU run can fall back
    inside the SAME run (round-5 review, next #5b).  The failures themselves are forced on a GPU box with SRPS_FORCE_FAIL
    (tests/test_gpu_distributed.py, tools/multi_gpu_first_contact.sh); this is the reporting logic, which needs no GPU."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_under_test2", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
    d = bench.describe_parallelism
    assert d(1, "images", "library", "none", False, False, False) == ("1 GPU", [])
    t, g = d(8, "images", "library", "library", False, False, False)
    assert "RCCL all-reduce" in t and "replicated CG" in t and g == []
    t, g = d(8, "images", "library", "torch", False, False, False)                       # the library's communicator failed on some rank
    assert "torch.distributed all-reduce" in t and len(g) == 1 and "library communicator" in g[0]
    t, g = d(8, "strips", "library", "library", True, False, False)                      # everything asked for ran
    assert "resident kernel on strips" in t and g == []
    t, g = d(8, "strips", "library", "library", False, True, False)                      # handshake failed / grid did not fit: streaming strips
    assert "column strips" in t and g == ["resident kernel on strips (cg_partition = 2) -> streaming strips"]
    t, g = d(8, "strips", "library", "library", False, False, False)                     # ... and those were not available either
    assert "replicated CG" in t and g == ["resident kernel on strips (cg_partition = 2) -> replicated CG"]
    t, g = d(8, "strips", "library", "torch", False, False, False)                       # no library communicator: no partitioned CG at all
    assert "replicated CG" in t and len(g) == 2 and "needs the library's communicator" in g[1]
    t, g = d(2, "strips", "library", "hosted", True, False, True)
    assert "[dry run" in t and "host collectives" in t and g == []
