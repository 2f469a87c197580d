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
    # make is incremental: always run it, so the tests never see artefacts older than the sources. (The GPU box has no
    # /root/reference and needs none: the prebuilt oracle/_ref/basal travels with the snapshot.)
    import oracle as orc
    orc.build()
    import basal_amd
    basal_amd.build()
    yield
