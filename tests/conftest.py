import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # a fresh checkout has no built artefacts (they are git-ignored): build them once (hipcc
    # cross-compiles gfx950 without a GPU), exactly what __graft_entry__.build() does
    pkg = os.path.join(ROOT, "voxel-cone-tracing_amd")
    needed = [os.path.join(pkg, "libvct_amd.so"), os.path.join(pkg, "libvct_host.so"),
              os.path.join(pkg, "vct_demo"), os.path.join(ROOT, "oracle", "libvct_oracle.so")]
    if not all(os.path.exists(f) for f in needed):
        import subprocess
        subprocess.check_call(["make", "-C", ROOT, "all", "-j4"])


@pytest.fixture(scope="session")
def oracle():
    from oracle import pyoracle
    pyoracle.lib()
    return pyoracle
