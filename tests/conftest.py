import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def _ensure_hip_library():
    """Test infrastructure: (re)build libmst_hip.so when the sources are newer or it is missing.  The product itself
    never builds implicitly - it fails loudly without the library."""
    csrc = os.path.join(REPO, "music_mixing_style_transfer_amd", "csrc")
    if os.path.exists("/opt/rocm/bin/hipcc"):
        subprocess.run(["make", "-j4", "-C", csrc], check=False, capture_output=True)


def pytest_configure(config):
    _ensure_hip_library()
    # the oracle's torch-CPU convolutions run several times SLOWER with one thread per core of a 128-core host than with 16
    # (measured on the GPU box: 6.7 s per segment at 128 threads, 3.5 s at 8): cap the intra-op pool for the whole session
    import torch
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")
    config.addinivalue_line("markers", "slow: long CPU test")


@pytest.fixture(scope="session")
def emu():
    """The C ABI compiled for the host against the SIMT emulator (tests/emu) - same sources as libmst_hip.so."""
    sys.path.insert(0, os.path.join(REPO, "tests"))
    from emu_binding import bind_emulator
    return bind_emulator()


@pytest.fixture()
def emu_default(emu):
    """Route the module API (networks.*, mixing_manipulator.*) through the emulator for one test."""
    from music_mixing_style_transfer_amd import _lib
    prev = _lib._default
    _lib.set_default_binding(emu)
    yield emu
    _lib.set_default_binding(prev)


@pytest.fixture(scope="session")
def oracle_fx_lib():
    import ctypes
    d = os.path.join(REPO, "oracle")
    subprocess.run(["make", "-C", d], check=True, capture_output=True)
    return ctypes.CDLL(os.path.join(d, "libfx_ref.so"))
