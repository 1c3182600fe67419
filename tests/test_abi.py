"""The C-ABI library loads and exports every symbol include/mst_hip.h declares (no compute without a GPU)."""
import ctypes as C
import os
import re
import subprocess

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def hip_lib():
    subprocess.run(["make", "-j4", "-C", os.path.join(REPO, "music_mixing_style_transfer_amd", "csrc")], check=True, capture_output=True)
    from music_mixing_style_transfer_amd import _lib
    return _lib.bind(_lib.LIB_PATH)


def _declared():
    src = open(os.path.join(REPO, "include", "mst_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mst_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported_and_bound(hip_lib):
    from music_mixing_style_transfer_amd import _lib
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(hip_lib.cdll, n), f"{n} declared in include/mst_hip.h but not exported"
    assert sorted(_lib.SIGNATURES) == names, "python binding table out of sync with the header"
    assert hip_lib.mst_version() >= 100


def test_argument_errors_without_gpu(hip_lib):
    from music_mixing_style_transfer_amd import _lib
    h = C.c_void_p()
    assert hip_lib.mst_tcn_create(None, C.byref(h)) == -1
    d = _lib.MstTcnDesc()
    d.nblocks, d.ninputs, d.noutputs, d.channels, d.kernel_size, d.cond_dim = 4, 4, 2, 30, 3, 16
    for n in range(4):
        d.dilations[n] = 1
    assert hip_lib.mst_tcn_create(C.byref(d), C.byref(h)) == -2       # MST_ERR_UNSUPPORTED, with a message
    assert b"multiple of ninputs" in hip_lib.mst_last_error()
    with pytest.raises(NotImplementedError):
        hip_lib.check(-2, "x")
    assert hip_lib.mst_fx_gain(None, None, 1, 10, 2, 0.0, 0, None, None) == -1
    # b_0, lb_1 .. lb_32 and a pad per 32-sample chunk + one start value per chunk + the 256-entry log10 table + one carry value per sequence
    # + three energy partials per item and 64-sample tile
    assert hip_lib.mst_fx_compressor_scratch_bytes(64, 131072, 2) == 128 * (4096 * 34 * 8 + 4096 * 8) + 2048 + 128 * 8 + 2048 * 192 * 8
    assert hip_lib.mst_tcn_workspace_bytes(None, 32, 131072, 1) == 2 * 32 * 131072 * 128 * 2


def test_product_loader_fails_loudly_when_library_is_missing(tmp_path):
    from music_mixing_style_transfer_amd import _lib
    with pytest.raises(ImportError, match="no CPU fallback"):
        _lib.bind(str(tmp_path / "libmst_hip.so"))
