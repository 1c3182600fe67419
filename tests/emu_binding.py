"""TEST INFRASTRUCTURE: the C ABI compiled for the host against the SIMT emulator (tests/emu), bound like the product
library but accepting CPU tensors (the emulator's "device memory" is host memory, its stream is null)."""
import contextlib
import ctypes as C
import os
import subprocess

from music_mixing_style_transfer_amd import _lib

EMU_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu")
EMU_LIB = os.path.join(EMU_DIR, "libmst_emu.so")


class EmuBinding(_lib.Binding):
    def require_device(self, t, what):
        if t.is_cuda:
            raise RuntimeError(f"{what}: the emulator build takes CPU tensors")

    def to_device(self, t):
        return t

    def stream_ptr(self, t):
        return C.c_void_p(0)

    def device_ctx(self, t):
        return contextlib.nullcontext()


def bind_emulator(build=True):
    if build:
        subprocess.run(["make", "-j4", "-C", EMU_DIR], check=True, capture_output=True)
    return EmuBinding(EMU_LIB)
