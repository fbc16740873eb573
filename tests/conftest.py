"""pytest configuration: the `gpu` marker, import paths, and the native checkers.

`-m "not gpu"` runs everywhere (oracle against the golden vectors, host logic, ABI export
check, two-process gloo test).  `-m gpu` needs an MI355X and goes through the C ABI only.
"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the CPU checkers are cheap to build (gcc, seconds); the HIP library is built by __graft_entry__.build()
    if not os.path.exists(os.path.join(ROOT, "oracle", "librt_oracle.so")):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "oracle"], check=True,
                       stdout=subprocess.DEVNULL)
    if not os.path.exists(os.path.join(ROOT, "raytracing_amd", "librt_host.so")):
        subprocess.run(["make", "-C", os.path.join(ROOT, "raytracing_amd", "csrc"), "../librt_host.so"],
                       check=True, stdout=subprocess.DEVNULL)


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    return load
