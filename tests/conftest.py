import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

REFERENCE = "/root/reference"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def have_reference():
    return os.path.isdir(os.path.join(REFERENCE, "rl"))


if os.environ.get("LHW_EMU") == "1":   # debugging aid: replay the -m gpu stepper tests on the host-side SIMT emulator
    from tests import emu as _emu
    _emu.install_as_backend()
