import os
import subprocess
import sys

import pytest

# torch bundles its own HIP runtime; it has to be the first one loaded into the process (libbasal_amd.so then binds
# to it). Loading libbasal_amd.so first and torch afterwards leaves torch without a visible device.
try:
    import torch  # noqa: F401
except Exception:  # pragma: no cover
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Make sure the oracle (checker) and the product library exist; both build without a GPU."""
    import oracle as orc
    if not (os.path.exists(orc.LIB) and os.path.exists(orc.CLI)):
        orc.build()
    import basal_amd
    if not (os.path.exists(basal_amd.lib_path()) and os.path.exists(os.path.join(ROOT, "basal_amd", "bin", "basal"))):
        basal_amd.build()
    yield
