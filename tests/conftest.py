import os
import subprocess
import sys
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_present() -> bool:
    return os.path.exists("/dev/kfd") and any(p.startswith("renderD") for p in os.listdir("/dev/dri")) if os.path.isdir("/dev/dri") else False


def pytest_collection_modifyitems(config, items):
    """Tests marked `gpu` need a real MI355X: without one they are skipped, not failed
    (plain `pytest tests` on a CPU-only box then equals `-m "not gpu"`)."""
    if _gpu_present():
        # torch wheels carry their own HIP runtime: in a process that uses both, torch's must come up before the
        # library's first HIP call (INTEGRATION.md "Sharing a process with PyTorch") — some gpu tests hand torch
        # tensors to the library, and any test may create the first engine
        if any("gpu" in item.keywords for item in items):
            try:
                import torch

                if torch.cuda.is_available():
                    torch.cuda.init()
            except ImportError:
                pass
        return
    skip = pytest.mark.skip(reason="no AMD GPU visible (/dev/kfd): gpu-marked tests need a real MI355X")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def emu_library_path():
    from tests.hipemu.build_emu import build_emu

    return build_emu()


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
    # torch wheels carry their own HIP runtime: when a process uses both, torch's must come up first
    # (INTEGRATION.md "Sharing a process with PyTorch"); some gpu tests hand torch tensors to the library
    try:
        import torch

        if torch.cuda.is_available():
            torch.cuda.init()
    except ImportError:
        pass
    from larynx_amd.engine import Engine

    eng = Engine(device=0)
    yield eng
    eng.close()
