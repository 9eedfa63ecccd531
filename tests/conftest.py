import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# harness-side runtime default (not the library's business): see capi.harness_pinned_copy_default
os.environ.setdefault("GPU_PINNED_MIN_XFER_SIZE", "4095")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def emu_lib():
    """Host-emulation build of the sync-free HIP sources (tests/host_emu) -- test infrastructure."""
    from pair_allegro_amd import capi
    d = os.path.join(ROOT, "tests", "host_emu")
    subprocess.run(["make", "-C", d], check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    return capi.Library(os.path.join(d, "_build", "liballegro_emu.so"))


@pytest.fixture(scope="session")
def hip_lib():
    """The product library; the GPU tests must run through it (no fallback)."""
    from pair_allegro_amd import capi
    lib = capi.Library()
    assert lib.device_count() >= 1, "no HIP device visible"
    return lib


@pytest.fixture(scope="session")
def model_dir(tmp_path_factory):
    return str(tmp_path_factory.mktemp("models"))
