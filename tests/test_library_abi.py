"""The C-ABI library builds for gfx950, loads without a GPU, exports every symbol include/srps.h
declares, and refuses to run (loudly) when no device is present -- there is no CPU fallback."""
import ctypes as C
import os
import re
import numpy as np
import pytest


def test_header_symbols_are_exported_and_bound(pkg):
    pkg.build()
    lib = pkg.load()
    declared = pkg.declared_symbols()
    assert len(declared) >= 40
    missing = [s for s in declared if not hasattr(lib, s)]
    assert not missing, missing
    assert set(lib._signatures) == set(declared)          # the ctypes binding covers the whole header


def test_every_entry_point_cites_the_reference_interface_it_replaces(pkg):
    txt = open(os.path.join(os.path.dirname(pkg.LIB_PATH), "..", "include", "srps.h")).read()
    assert txt.count("replaces:") >= 12
    for ref in ("devicecalls.cuh:26", "devicecalls.cuh:33", "devicecalls.cuh:34", "devicecalls.cuh:35", "devicecalls.cuh:36",
                "devicecalls.cuh:37", "SRPS.cu:88-98", "SRPS.cu:100-270"):
        assert ref in txt, ref


def test_no_device_is_a_loud_error_not_a_fallback(pkg):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    lib = pkg.load()
    assert lib.srps_version().startswith(b"srps-hip")
    with pytest.raises(pkg.SRPSError) as ei:
        pkg.Context(device_id=0)
    assert ei.value.code == 2 and "HIP error" in str(ei.value)
    # null-context calls return a status, they do not crash
    assert lib.srps_lighting(None) == 1
    assert b"null context" in lib.srps_last_error()


def test_product_never_imports_the_oracle():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    bad = []
    for base, _, files in os.walk(os.path.join(root, "srmeetsps-cuda_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h", ".hpp")):
                t = open(os.path.join(base, f), errors="replace").read()
                if re.search(r"srps_oracle|c_oracle|libsrps_oracle|oracle/", t):
                    bad.append(os.path.join(base, f))
    assert not bad, bad
