import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # a fresh checkout has no built artefacts (they are git-ignored): compile the HIP library, the
    # oracle and the C++ host-mirror test once, exactly as the driver's build() does
    if shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc"):
        need = [os.path.join(ROOT, "ripcurrents_amd", "librcflow.so"), os.path.join(ROOT, "oracle", "liboracle.so")]
        if not all(os.path.exists(p) for p in need):
            import __graft_entry__
            __graft_entry__.build()


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (test infrastructure; see oracle/rc_oracle.h)."""
    from oracle import oracle
    oracle.lib()
    return oracle


@pytest.fixture(scope="session")
def ctx():
    """One librcflow context for the whole GPU session (4K-capable, 2 stream slots)."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from ripcurrents_amd.api import Context
    c = Context(3840, 2160, device=0, streams=2)
    yield c
    c.close()
