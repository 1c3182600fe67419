"""ctypes binding of libmst_hip.so (the C ABI declared in include/mst_hip.h).

The product path has NO fallback: if the HIP library is missing, import of anything that needs it raises.
(tests/ may bind the CPU emulator build of the same sources through `bind(path)`; the product never does.)
"""
import ctypes as C
import os

import torch  # noqa: F401  - MUST precede loading libmst_hip.so: the library has to share torch's HIP runtime
#                            (same libamdhip64 instance => same streams / device pointers); loading it first binds
#                            /opt/rocm's copy and every hipMalloc then fails with "no ROCm-capable device".

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libmst_hip.so")

MST_OK = 0
MST_PREC_F32 = 0
MST_PREC_BF16 = 1
MST_PREC_BF16X3 = 2
MST_MAX_BLOCKS = 32
TCN_TUNING_DEFAULT = 245    # mst_tcn_set_tuning flags a fresh handle starts with (csrc/mst_tcn.hip: bit 0 x3_small_tiles, bits 1-2 bf16_form = 2, bit 4 bf16_reuse, bit 5 bf16_fuse0, bit 6 x3_half_cm, bit 7 bf16_cm128; bit 3 is refused)
PRECISIONS = {"fp32": MST_PREC_F32, "f32": MST_PREC_F32, "bf16": MST_PREC_BF16, "bf16x3": MST_PREC_BF16X3}

STATUS_NAMES = {0: "MST_OK", -1: "MST_ERR_ARG", -2: "MST_ERR_UNSUPPORTED", -3: "MST_ERR_HIP", -4: "MST_ERR_STATE",
                -5: "MST_ERR_WORKSPACE"}


class MstTcnDesc(C.Structure):
    _fields_ = [("nblocks", C.c_int), ("ninputs", C.c_int), ("noutputs", C.c_int), ("channels", C.c_int),
                ("kernel_size", C.c_int), ("cond_dim", C.c_int), ("dilations", C.c_int * MST_MAX_BLOCKS), ("causal", C.c_int)]


FX_FORM_EQ_LANE_APPLY, FX_FORM_EQ_VALU_ENDS, FX_FORM_COMP_SLICE_SMALL = 1, 2, 4          # MstFxFuse.forms (include/mst_hip.h)


class MstFxFuse(C.Structure):
    """include/mst_hip.h MstFxFuse; struct_size is filled in here (the library refuses a struct of another layout)."""
    _fields_ = [("struct_size", C.c_uint), ("forms", C.c_int), ("in_scale_dev", C.c_void_p), ("out_sumsq_dev", C.c_void_p),
                ("in_sumsq_dev", C.c_void_p), ("post_rms", C.c_int), ("post_gain", C.c_float), ("out_in_sumsq_dev", C.c_void_p),
                ("out_ms_dev", C.c_void_p), ("in_ms_dev", C.c_void_p)]

    def __init__(self, in_scale_dev=None, out_sumsq_dev=None, in_sumsq_dev=None, post_rms=0, post_gain=0.0, out_in_sumsq_dev=None,
                 out_ms_dev=None, in_ms_dev=None, forms=0):
        super().__init__(C.sizeof(MstFxFuse), forms, in_scale_dev, out_sumsq_dev, in_sumsq_dev, post_rms, post_gain, out_in_sumsq_dev,
                         out_ms_dev, in_ms_dev)


class MstEncDesc(C.Structure):
    _fields_ = [("nblocks", C.c_int), ("channels", C.c_int * (MST_MAX_BLOCKS + 1)), ("kernels", C.c_int * MST_MAX_BLOCKS),
                ("strides", C.c_int * MST_MAX_BLOCKS), ("dilations", C.c_int * MST_MAX_BLOCKS), ("valid_padding", C.c_int), ("act_slope", C.c_float)]


_P = C.c_void_p
_F = C.c_void_p   # float* passed as integer addresses (tensor.data_ptr())

# name -> (restype, argtypes); kept in sync with include/mst_hip.h (tests/test_abi.py checks the export list)
SIGNATURES = {
    "mst_version": (C.c_int, []),
    "mst_last_error": (C.c_char_p, []),
    "mst_tcn_create": (C.c_int, [C.POINTER(MstTcnDesc), C.POINTER(_P)]),
    "mst_tcn_destroy": (C.c_int, [_P]),
    "mst_tcn_load_block": (C.c_int, [_P, C.c_int, _F, _F, _F, _F, _F, C.c_float, _F, _F, _F, _P]),
    "mst_tcn_load_output": (C.c_int, [_P, _F, _F, _P]),
    "mst_tcn_set_cond": (C.c_int, [_P, _F, C.c_int, C.c_long, _P]),
    "mst_tcn_workspace_bytes": (C.c_size_t, [_P, C.c_int, C.c_int, C.c_int]),
    "mst_tcn_forward": (C.c_int, [_P, _F, _F, C.c_int, C.c_int, C.c_int, _P, C.c_size_t, _P]),
    "mst_tcn_forward_blocks": (C.c_int, [_P, _F, _F, C.c_int, C.c_int, C.c_int, C.c_int, _P, C.c_size_t, _P]),
    "mst_tcn_set_tuning": (C.c_int, [_P, C.c_int]),
    "mst_tcn_get_tuning": (C.c_int, [_P, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "mst_tcn_timing_begin": (C.c_int, [_P, C.c_int]),
    "mst_tcn_timing_end": (C.c_int, [_P, C.POINTER(C.c_float), C.POINTER(C.c_int)]),
    "mst_calib_mainloop": (C.c_int, [C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float), _P]),
    "mst_enc_create": (C.c_int, [C.POINTER(MstEncDesc), C.POINTER(_P)]),
    "mst_enc_destroy": (C.c_int, [_P]),
    "mst_enc_load_conv": (C.c_int, [_P, C.c_int, C.c_int, _F, _F, _F, _F, _F, _F, C.c_float, _P]),
    "mst_enc_set_tuning": (C.c_int, [_P, C.c_long]),
    "mst_enc_set_schedule": (C.c_int, [_P, C.c_int]),
    "mst_global_avgpool": (C.c_int, [_F, _F, C.c_long, C.c_int, _P]),
    "mst_enc_zero_stuff": (C.c_int, [_F, _F, C.c_long, C.c_long, C.c_int, C.c_long, C.c_long, _P]),
    "mst_enc_workspace_bytes": (C.c_size_t, [_P, C.c_int, C.c_int]),
    "mst_enc_forward": (C.c_int, [_P, _F, _F, C.c_int, C.c_int, C.c_int, _P, C.c_size_t, _P]),
    "mst_enc_forward_blocks": (C.c_int, [_P, _F, _F, C.c_int, C.c_int, C.c_int, C.c_int, _P, C.c_size_t, _P]),
    "mst_enc_block_length": (C.c_int, [_P, C.c_int, C.c_int]),
    "mst_enc_conv_length": (C.c_int, [_P, C.c_int, C.c_int, C.c_int]),
    "mst_enc_forward_conv": (C.c_int, [_P, C.c_int, C.c_int, _F, _F, C.c_int, C.c_int, _P]),
    "mst_film_forward": (C.c_int, [_F, _F, _F, C.c_int, C.c_int, C.c_int, _F, _F, C.c_int, C.c_long, _F, _P]),
    "mst_embedding_mean": (C.c_int, [_F, C.c_int, C.c_int, _F, _P]),
    "mst_fx_biquad_scratch_bytes": (C.c_size_t, [C.c_int, C.c_long, C.c_int, C.c_int]),
    "mst_fx_biquad_cascade": (C.c_int, [_F, _F, C.c_int, C.c_long, C.c_int, C.POINTER(C.c_double), C.c_int, _P, C.c_size_t, _P, _P]),
    "mst_fx_sumsq": (C.c_int, [_F, C.c_int, C.c_long, _P, _P]),
    "mst_fx_rms_pending": (C.c_int, [_P, _P, C.c_long, _P, C.c_long, _P, C.c_int, _P]),
    "mst_fx_scale_items": (C.c_int, [_F, _F, C.c_int, C.c_long, _P, _P]),
    "mst_fx_compressor_scratch_bytes": (C.c_size_t, [C.c_int, C.c_long, C.c_int]),
    "mst_fx_compressor": (C.c_int, [_F, _F, C.c_int, C.c_long, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double,
                                    C.c_double, _P, C.c_size_t, _P, _P]),
    "mst_fx_midside_imager": (C.c_int, [_F, _F, C.c_int, C.c_long, C.c_double, _P, _P, _P]),
    "mst_fx_gain": (C.c_int, [_F, _F, C.c_int, C.c_long, C.c_int, C.c_double, C.c_int, _P, _P]),
    "mst_fx_haas": (C.c_int, [_F, _F, C.c_int, C.c_long, C.c_int, C.c_long, C.c_double, C.c_int, _P]),
    "mst_fx_panner": (C.c_int, [_F, _F, C.c_int, C.c_long, C.c_int, C.c_float, C.c_float, _P]),
    "mst_fx_rms_normalize": (C.c_int, [_F, _F, C.c_int, C.c_long, C.c_long, _P, _P]),
    "mst_fx_convolver_create": (C.c_int, [C.c_long, C.c_long, C.c_int, C.c_int, C.POINTER(_P)]),
    "mst_fx_convolver_destroy": (None, [_P]),
    "mst_fx_convolver_workspace_bytes": (C.c_size_t, [_P]),
    "mst_fx_compressor_grid": (C.c_int, [_F, _F, C.c_int, C.c_long, C.c_int, _F, _F, C.c_double, C.c_double, C.c_double, _P, C.c_size_t,
                                         _P, _P]),
    "mst_fx_range_reduce": (C.c_int, [_F, C.c_long, C.c_int, C.c_int, _P, _P, _P, C.c_int, C.c_int, _P, _P]),
    "mst_fx_onset_hfc": (C.c_int, [_F, C.c_int, C.c_long, C.c_int, C.c_int, C.c_int, _F, _P]),
    "mst_fx_stft_create": (C.c_int, [C.c_long, C.c_long, C.POINTER(C.c_float), C.c_int, C.POINTER(_P)]),
    "mst_fx_stft_destroy": (None, [_P]),
    "mst_fx_stft_workspace_bytes": (C.c_size_t, [_P]),
    "mst_fx_stft_mean_magnitude": (C.c_int, [_P, _F, C.c_long, C.c_int, C.c_int, _F, _P, C.c_size_t, _P]),
    "mst_fx_algorithmic_reverb_scratch_bytes": (C.c_size_t, [C.c_int, C.c_long, C.c_int]),
    "mst_fx_algorithmic_reverb": (C.c_int, [_F, _F, C.c_int, C.c_long, C.c_int, C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_int), C.c_int,
                                            C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, _P, C.c_size_t, _P]),
    "mst_fx_stereo_moments": (C.c_int, [_F, C.c_int, C.c_long, _P, _P]),
    "mst_fx_stereo_mix": (C.c_int, [_F, _F, C.c_int, C.c_long, C.c_float, C.c_float, C.c_float, C.c_float, _P]),
    "mst_fx_convolve": (C.c_int, [_P, _F, _F, C.c_long, _F, C.c_long, C.c_double, C.c_double, _P, C.c_size_t, _P]),
}


class MstError(RuntimeError):
    pass


class Binding:
    """A loaded library with typed entry points and status checking."""

    def __init__(self, path):
        if not os.path.exists(path):
            raise ImportError(
                f"{path} not found: the HIP extension has not been built (python -c 'import __graft_entry__ as g; "
                f"g.build()' or make -C music_mixing_style_transfer_amd/csrc).  There is no CPU fallback.")
        self.path = path
        self.cdll = C.CDLL(path)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(self.cdll, name)   # AttributeError if the symbol is missing: fail loudly
            fn.restype = res
            fn.argtypes = args
            setattr(self, name, fn)

    def require_device(self, t, what):
        """The kernels run on the MI355X only: a tensor that is not in HBM is an error (there is no CPU path)."""
        if not t.is_cuda:
            raise RuntimeError(f"{what}: input must live on the MI355X (a CUDA/HIP tensor); there is no CPU path")

    def to_device(self, t):
        """Host arrays handed to a processor travel to the MI355X (numpy in -> numpy out at that boundary)."""
        if t.is_cuda:
            return t
        if not torch.cuda.is_available():
            raise RuntimeError("the kernels run on the MI355X only; no GPU is visible and there is no CPU path")
        return t.cuda()

    def stream_ptr(self, t):
        """The caller's current HIP stream for tensor t's device (what every entry point takes as `stream`)."""
        return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)

    def device_ctx(self, t):
        """Context that makes t's device the current HIP device: handles, weights and the FiLM table are allocated on,
        and kernels launched on, the device the data lives on (not whatever device happens to be current)."""
        return torch.cuda.device(t.device)

    def check(self, status, what=""):
        if status != MST_OK:
            msg = self.mst_last_error()
            msg = msg.decode() if msg else ""
            code = STATUS_NAMES.get(status, str(status))
            if status == -2:
                raise NotImplementedError(f"{what}: {code}: {msg}")
            if status == -1:
                raise ValueError(f"{what}: {code}: {msg}")
            raise MstError(f"{what}: {code}: {msg}")


def bind(path):
    return Binding(path)


_default = None


def lib():
    """The product binding (libmst_hip.so), loaded on first use."""
    global _default
    if _default is None:
        _default = Binding(LIB_PATH)
    return _default


def set_default_binding(binding):
    """TEST HOOK: route the module API through another build of the same C ABI (the CPU SIMT emulator)."""
    global _default
    _default = binding
