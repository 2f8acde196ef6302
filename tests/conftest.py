import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """A fresh checkout has no built libraries (they are git-ignored): build them once, as
    __graft_entry__.build() does.  Nothing is rebuilt when they are there (the GPU box receives them built)."""
    from molann_amd import _capi, script
    if not (os.path.exists(_capi.LIB_PATH) and os.path.exists(script.TORCH_LIB_PATH)):
        _capi.build_library()
    # The suite builds the same few dozen plan-specialised kernels hundreds of times (every test makes its own models): keep
    # their code objects for the length of the session.  Every distinct kernel is still compiled by hipRTC once.
    if not os.environ.get("MOLANN_JIT_CACHE_DIR"):
        import atexit
        import shutil
        import tempfile
        d = tempfile.mkdtemp(prefix="molann_jit_")
        os.environ["MOLANN_JIT_CACHE_DIR"] = d
        atexit.register(shutil.rmtree, d, True)


@pytest.fixture(scope="session")
def hip_device():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    return torch.device("cuda:0")
