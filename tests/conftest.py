import os
import subprocess
import sys
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def emu_library():
    """Build (once) the CPU-emulator flavour of the C-ABI library: the SAME
    sources under larynx_amd/csrc compiled against tests/hipemu.  Test-only."""
    from tests.hipemu.build_emu import build_emu

    return build_emu()


@pytest.fixture(scope="session")
def emu_engine(emu_library):
    from larynx_amd.engine import Engine

    eng = Engine(device=0, library_path=emu_library)
    yield eng
    eng.close()


@pytest.fixture(scope="session")
def gpu_engine():
    from larynx_amd.engine import Engine

    eng = Engine(device=0)
    yield eng
    eng.close()
