import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
if os.path.join(ROOT, "oracle") not in sys.path:
    sys.path.insert(0, os.path.join(ROOT, "oracle"))

# Thread pools of the test process (OpenMP in the C oracle and torch, OpenBLAS under numpy) are sized to the CPUs the process may really
# use -- its cgroup quota, not the 256 logical CPUs a GPU box shows (oracle/cpu_budget.py) -- before any of those libraries is loaded.
import cpu_budget  # noqa: E402
for _var in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_var, str(cpu_budget.effective_cpus()))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    return importlib.import_module("srmeetsps-cuda_amd")


@pytest.fixture(scope="session")
def oracle():
    import srps_oracle
    return srps_oracle


@pytest.fixture(scope="session")
def gpu_ctx(pkg):
    """one library context on cuda:0; fails loudly when the HIP extension is missing"""
    import torch
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    pkg.load()
    ctx = pkg.Context(device_id=0)
    yield ctx
    ctx.close()
