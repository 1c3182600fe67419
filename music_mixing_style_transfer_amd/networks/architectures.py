"""FXencoder and the TCN "MixFXcloner" behind the reference's module API
(networks/architectures.py of the reference: FXencoder :26-70, TCNModel :76-174, TCNBlock :177-234).

Same constructor signatures, attribute names and state_dict keys, so
    enc = FXencoder(cfg); enc.load_state_dict(ckpt)            # strict
    tcn = TCNModel(nparams=2048, ninputs=2, ...); tcn(x, emb)
work unchanged (inference/style_transfer.py:47-57,94-108,149,161).  forward() hands raw device pointers and
the current HIP stream to libmst_hip.so (include/mst_hip.h); weights are BN-folded and packed into MFMA
fragment order by the library when the module first runs (and again whenever parameters change).
pytorch_lightning is not a dependency: TCNModel is a plain nn.Module carrying `hparams`.
"""
import ctypes as C
import inspect
import os

import torch
import torch.nn as nn

from .. import _lib
from .network_utils import Conv1d_layer, ConvBlock, FiLM, Res_ConvBlock, _DeviceState  # noqa: F401  (star-export parity)


class _HParams(dict):
    """Attribute-style access like Lightning's save_hyperparameters() namespace (hasattr / copy.deepcopy / pickle work:
    a missing name is an AttributeError, not a KeyError)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k) from None

    def __setattr__(self, k, v):
        self[k] = v


def _stream_ptr(t):
    return _lib.lib().stream_ptr(t)


def _check_device(t, what):
    b = _lib.lib()
    b.require_device(t, what)
    if t.dtype != torch.float32:
        raise TypeError(f"{what}: float32 input expected, got {t.dtype}")
    return b


from .network_utils import _signature  # noqa: E402


class _Workspace:
    def __init__(self):
        self.buf = None

    def get(self, nbytes, device):
        if self.buf is None or self.buf.numel() < nbytes or self.buf.device != device:
            self.buf = None
            self.buf = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
        return self.buf


# ------------------------------------------------------------------------------------------------ encoder
class _EncoderRunner:
    """Owns one MstEnc handle for a list of Res_ConvBlocks."""

    def __init__(self, blocks):
        self.blocks = blocks
        self.handle = None
        self.sig = None
        self.ws = _Workspace()
        self.lib = None

    def _module_sig(self):
        return tuple(s for blk in self.blocks for s in _signature(blk))

    def _ensure(self, b):
        sig = self._module_sig()
        if self.handle is not None and sig == self.sig and self.lib is b:
            return
        self.close()
        slopes = {c.act_slope() for blk in self.blocks for c in (blk.conv1, blk.conv2)}
        for blk in self.blocks:
            if not (blk.conv1.hip_supported() and blk.conv2.hip_supported()):
                raise NotImplementedError("Res_ConvBlock stack: mode='conv' layers with padding='SAME' only on gfx950 (Conv1d_layer and ConvBlock "
                                          "run mode='deconv' / padding='VALID' layer by layer)")
        if len(slopes) != 1:
            raise NotImplementedError("Res_ConvBlock stack: one activation for all layers (activation == last_activation, as FXencoder "
                                      "builds them) on gfx950")
        d = _lib.MstEncDesc()
        d.act_slope = slopes.pop()
        d.nblocks = len(self.blocks)
        d.channels[0] = self.blocks[0].conv1.in_channels
        for i, blk in enumerate(self.blocks):
            d.channels[i + 1] = blk.conv2.out_channels
            d.kernels[i] = blk.conv2.kernel_size
            d.strides[i] = blk.conv2.stride
            d.dilations[i] = blk.conv2.dilation
        h = C.c_void_p()
        b.check(b.mst_enc_create(C.byref(d), C.byref(h)), "mst_enc_create")
        self.handle, self.lib = h, b
        for i, blk in enumerate(self.blocks):
            for which, conv in enumerate((blk.conv1, blk.conv2)):
                a = conv.export_arrays()
                ptr = lambda t: None if t is None else t.data_ptr()
                b.check(b.mst_enc_load_conv(h, i, which, ptr(a["w"]), ptr(a["bias"]), ptr(a["bn_w"]), ptr(a["bn_b"]),
                                            ptr(a["bn_mean"]), ptr(a["bn_var"]), a["eps"], None), "mst_enc_load_conv")
        self.sig = sig

    def close(self):
        if self.handle is not None:
            self.lib.mst_enc_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def run(self, x, pooled=True, n_run=None, precision="fp32"):
        b = _check_device(x, "FXencoder.forward")
        prec = _lib.PRECISIONS[precision]
        if x.dim() != 3 or x.shape[1] != self.blocks[0].conv1.in_channels:
            raise ValueError(f"FXencoder.forward: expected [B, {self.blocks[0].conv1.in_channels}, L], got {tuple(x.shape)}")
        with b.device_ctx(x):          # handle, weights, workspace and launches on the device the data lives on
            return self._run(b, x, pooled, n_run, prec)

    def _run(self, b, x, pooled, n_run, prec):
        self._ensure(b)
        x = x.contiguous()
        B, _, L = x.shape
        nbytes = b.mst_enc_workspace_bytes(self.handle, B, L)
        ws = self.ws.get(nbytes, x.device)
        st = _stream_ptr(x)
        if pooled:
            out = torch.empty(B, self.blocks[-1].conv2.out_channels, dtype=torch.float32, device=x.device)
            b.check(b.mst_enc_forward(self.handle, x.data_ptr(), out.data_ptr(), B, L, prec, ws.data_ptr(), nbytes, st),
                    "mst_enc_forward")
            return out
        n_run = len(self.blocks) if n_run is None else n_run
        lout = b.mst_enc_block_length(self.handle, n_run - 1, L)
        out = torch.empty(B, self.blocks[n_run - 1].conv2.out_channels, lout, dtype=torch.float32, device=x.device)
        b.check(b.mst_enc_forward_blocks(self.handle, x.data_ptr(), out.data_ptr(), B, L, prec, n_run, ws.data_ptr(),
                                         nbytes, st), "mst_enc_forward_blocks")
        return out


class FXencoder(_DeviceState, nn.Module):
    """Audio-effects encoder: stereo waveform [B, 2, L] -> FX embedding [B, channels[-1]]."""

    _DEVICE_STATE = (("_runner", None),)

    def __init__(self, config):
        super().__init__()
        # the reference prepends the stereo input to the caller's list in place (architectures.py:30);
        # the observable side effect on `config` is kept.
        config["channels"].insert(0, 2)
        ch = config["channels"]
        encoder = []
        for i in range(len(config["kernels"])):
            if config["conv_block"] == "res":
                encoder.append(Res_ConvBlock(dimension=1, in_channels=ch[i], out_channels=ch[i + 1],
                                             kernel_size=config["kernels"][i], stride=config["strides"][i],
                                             padding="SAME", dilation=config["dilation"][i], norm=config["norm"],
                                             activation=config["activation"], last_activation=config["activation"]))
            elif config["conv_block"] == "conv":
                encoder.append(ConvBlock(dimension=1, layer_num=1, in_channels=ch[i], out_channels=ch[i + 1],
                                         kernel_size=config["kernels"][i], stride=config["strides"][i], padding="VALID",
                                         dilation=config["dilation"][i], norm=config["norm"],
                                         activation=config["activation"], last_activation=config["activation"],
                                         mode="conv"))
        self.encoder = nn.Sequential(*encoder)
        self.glob_pool = nn.AdaptiveAvgPool1d(1)
        self.precision = os.environ.get("MST_ENC_PRECISION", "fp32")   # "fp32" (parity) | "bf16" (throughput)
        self._runner = None

    def _get_runner(self):
        if self._runner is None:
            blocks = list(self.encoder)
            if not all(isinstance(b, Res_ConvBlock) for b in blocks):
                raise NotImplementedError("FXencoder: the fused encoder handle exists for conv_block='res' stacks")
            self._runner = _EncoderRunner(blocks)
        return self._runner

    def forward(self, input):
        if len(self.encoder) and not isinstance(self.encoder[0], Res_ConvBlock):
            # conv_block='conv' (architectures.py:46-58): VALID-padded single convolutions, layer by layer through mst_enc_forward_conv
            # (exact-fp32 kernels whatever self.precision says), then the global average pool
            b = _check_device(input, "FXencoder.forward")
            x = self.encoder(input)
            with b.device_ctx(x):
                x = x.contiguous()
                out = torch.empty(x.shape[0], x.shape[1], dtype=torch.float32, device=x.device)
                b.check(b.mst_global_avgpool(x.data_ptr(), out.data_ptr(), x.shape[0] * x.shape[1], x.shape[2], _stream_ptr(x)),
                        "mst_global_avgpool")
            return out
        return self._get_runner().run(input, pooled=True, precision=self.precision)

    def forward_blocks(self, input, n_run):
        """Parity probe: output of the n_run-th Res_ConvBlock, [B, C, L_out]."""
        return self._get_runner().run(input, pooled=False, n_run=n_run, precision=self.precision)


# ------------------------------------------------------------------------------------------------ TCN
def _dense_conv_weight(conv):
    """conv.weight [out, in / groups, k] as the dense [out, in, k] weight of the same convolution (block diagonal for a grouped
    conv1 - the kernels contract over all input channels; the zeros add nothing)."""
    w = conv.weight.detach().to("cpu", torch.float32)
    if conv.groups == 1:
        return w.contiguous()
    out_ch, per, k = w.shape
    dense = torch.zeros(out_ch, conv.in_channels, k, dtype=torch.float32)
    opg = out_ch // conv.groups
    for o in range(out_ch):
        g = o // opg
        dense[o, g * per:(g + 1) * per] = w[o]
    return dense


def _load_tcn_block(b, h, n, blk):
    f = lambda t: t.detach().to("cpu", torch.float32).contiguous()
    arrs = [_dense_conv_weight(blk.conv1), f(blk.bn.weight), f(blk.bn.bias), f(blk.bn.running_mean), f(blk.bn.running_var),
            f(blk.film.film_fc.weight), f(blk.film.film_fc.bias), f(blk.res.weight)]
    b.check(b.mst_tcn_load_block(h, n, arrs[0].data_ptr(), arrs[1].data_ptr(), arrs[2].data_ptr(), arrs[3].data_ptr(),
                                 arrs[4].data_ptr(), float(blk.bn.eps), arrs[5].data_ptr(), arrs[6].data_ptr(), arrs[7].data_ptr(),
                                 None), "mst_tcn_load_block")


class TCNBlock(_DeviceState, nn.Module):
    """dilated conv -> BatchNorm -> LeakyReLU -> FiLM -> + grouped 1x1 residual.  Inside TCNModel.forward the block is one fused
    gfx950 kernel; called on its own (reference :222-234) it runs as a one-block net through the same library."""
    _DEVICE_STATE = (("_handle", None), ("_sig", None), ("_hlib", None), ("_ws", lambda: _Workspace()))

    def __init__(self, in_ch, out_ch, kernel_size=3, dilation=1, cond_dim=2048, grouped=False, causal=False,
                 conditional=False, **kwargs):
        super().__init__()
        self.in_ch, self.out_ch = in_ch, out_ch
        self.kernel_size, self.dilation = kernel_size, dilation
        self.grouped, self.causal, self.conditional = grouped, causal, conditional
        groups = out_ch if grouped and (in_ch % out_ch == 0) else 1
        span = (kernel_size - 1) * dilation
        self.pad_length = span if causal else span // 2
        self.conv1 = nn.Conv1d(in_ch, out_ch, kernel_size=kernel_size, padding=self.pad_length, dilation=dilation,
                               groups=groups, bias=False)
        if grouped:
            self.conv1b = nn.Conv1d(out_ch, out_ch, kernel_size=1)
        if conditional:
            self.film = FiLM(cond_dim, out_ch)
        self.bn = nn.BatchNorm1d(out_ch)
        self.relu = nn.LeakyReLU()
        self.res = nn.Conv1d(in_ch, out_ch, kernel_size=1, groups=in_ch, bias=False)
        self._handle = self._sig = self._hlib = None
        self._ws = _Workspace()

    def _ensure(self, b):
        sig = _signature(self)
        if self._handle is not None and sig == self._sig and self._hlib is b:
            return
        self._close()
        if not self.conditional:
            raise AttributeError("'TCNBlock' object has no attribute 'film'")       # what the reference's forward raises
        d = _lib.MstTcnDesc()
        d.nblocks, d.ninputs, d.noutputs, d.channels = 1, self.in_ch, 1, self.out_ch
        d.kernel_size, d.cond_dim, d.causal = self.kernel_size, self.film.film_fc.in_features, int(bool(self.causal))
        d.dilations[0] = self.dilation
        h = C.c_void_p()
        b.check(b.mst_tcn_create(C.byref(d), C.byref(h)), "mst_tcn_create")
        self._handle, self._hlib = h, b
        _load_tcn_block(b, h, 0, self)
        zero = torch.zeros(self.out_ch + 1, dtype=torch.float32)
        b.check(b.mst_tcn_load_output(h, zero.data_ptr(), zero[self.out_ch:].data_ptr(), None), "mst_tcn_load_output")   # unused head
        self._sig = sig

    def _close(self):
        if self._handle is not None:
            self._hlib.mst_tcn_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self._close()
        except Exception:
            pass

    def forward(self, x, p):
        b = _check_device(x, "TCNBlock.forward")
        if x.dim() != 3 or x.shape[1] != self.in_ch:
            raise ValueError(f"TCNBlock.forward: expected [B, {self.in_ch}, L], got {tuple(x.shape)}")
        with b.device_ctx(x):
            self._ensure(b)
            x = x.contiguous()
            B, _, L = x.shape
            cond = p.to(x.device, torch.float32).contiguous()
            if cond.shape[0] not in (1, B):
                raise RuntimeError(f"The size of tensor a ({cond.shape[0]}) must match the size of tensor b ({B}) at non-singleton dimension 0")
            st = _stream_ptr(x)
            b.check(b.mst_tcn_set_cond(self._handle, cond.data_ptr(), cond.shape[0], 0, st), "mst_tcn_set_cond")
            nbytes = b.mst_tcn_workspace_bytes(self._handle, B, L, _lib.MST_PREC_F32)
            ws = self._ws.get(nbytes, x.device)
            y = torch.empty(B, self.out_ch, L, dtype=torch.float32, device=x.device)
            b.check(b.mst_tcn_forward_blocks(self._handle, x.data_ptr(), y.data_ptr(), B, L, _lib.MST_PREC_F32, 1, ws.data_ptr(), nbytes, st),
                    "mst_tcn_forward_blocks")
        return y


class TCNModel(_DeviceState, nn.Module):
    """Temporal convolutional network with FiLM conditioning (the MixFXcloner).

    forward(x [B, ninputs, L], cond [1|B, cond_dim] or list of nblocks such tensors) -> [B, noutputs, L] in [-1, 1].
    `precision` selects the arithmetic of the dense dilated convolutions: "fp32" (exact fp32 on the matrix
    cores, the parity mode, default) or "bf16" (bf16 operands / fp32 accumulate, the throughput mode).
    """

    _DEVICE_STATE = (("_handle", None), ("_lib", None), ("_sig", None), ("_ws", lambda: _Workspace()))

    def __init__(self, nparams, ninputs=1, noutputs=1, nblocks=10, kernel_size=3, dilation_growth=1, channel_growth=1,
                 channel_width=32, stack_size=10, cond_dim=2048, grouped=False, causal=False, skip_connections=False,
                 num_examples=4, save_dir=None, **kwargs):
        super().__init__()
        frame = inspect.currentframe()
        names, _, _, values = inspect.getargvalues(frame)
        self.hparams = _HParams({n: values[n] for n in names if n not in ("self", "frame")})
        self.blocks = nn.ModuleList()
        out_ch = None
        for n in range(nblocks):
            in_ch = out_ch if n > 0 else ninputs
            out_ch = in_ch * channel_growth if channel_growth > 1 else channel_width
            dilation = dilation_growth ** (n % stack_size)
            self.blocks.append(TCNBlock(in_ch, out_ch, kernel_size=kernel_size, dilation=dilation,
                                        padding="same" if causal else "valid", causal=causal, cond_dim=cond_dim,
                                        grouped=grouped, conditional=True if nparams > 0 else False))
        self.output = nn.Conv1d(out_ch, noutputs, kernel_size=1)
        self.precision = os.environ.get("MST_TCN_PRECISION", "fp32")
        self._handle = None
        self._lib = None
        self._sig = None
        self._ws = _Workspace()

    # ---- reference API -------------------------------------------------------------------------
    def compute_receptive_field(self):
        """Receptive field in samples."""
        hp = self.hparams
        rf = hp.kernel_size
        for n in range(1, hp.nblocks):
            rf += (hp.kernel_size - 1) * hp.dilation_growth ** (n % hp.stack_size)
        return rf

    @staticmethod
    def add_model_specific_args(parent_parser):
        """The model's hyper-parameters as command-line flags (reference architectures.py:158-174: same names, types and defaults)."""
        from argparse import ArgumentParser
        parser = ArgumentParser(parents=[parent_parser], add_help=False)
        for name, default in (("ninputs", 1), ("noutputs", 1), ("nblocks", 4), ("kernel_size", 5), ("dilation_growth", 10),
                              ("channel_growth", 1), ("channel_width", 32), ("stack_size", 10)):
            parser.add_argument("--" + name, type=int, default=default)
        for name in ("grouped", "causal", "skip_connections"):
            parser.add_argument("--" + name, default=False, action="store_true")
        return parser

    # ---- gfx950 plumbing -----------------------------------------------------------------------
    def _ensure(self, b):
        sig = _signature(self)
        if self._handle is not None and sig == self._sig and self._lib is b:
            return
        self._close()
        hp = self.hparams
        if not hp.nparams > 0:
            raise AttributeError("'TCNBlock' object has no attribute 'film'")       # nparams = 0 builds blocks without FiLM; the
        d = _lib.MstTcnDesc()                                                          # reference's forward fails the same way
        # channel_growth > 1 (a different width per block, reference :112-115): the handle holds the LAST block + the output head, the
        # blocks in front of it run one by one through their own one-block handles (TCNBlock.forward) - all on the generic fp32 kernels
        held = list(self.blocks)[-1:] if hp.channel_growth > 1 else list(self.blocks)
        d.nblocks, d.ninputs, d.noutputs = len(held), held[0].in_ch, hp.noutputs
        d.channels, d.kernel_size, d.cond_dim, d.causal = held[-1].out_ch, hp.kernel_size, hp.cond_dim, int(bool(hp.causal))
        for n, blk in enumerate(held):
            d.dilations[n] = blk.dilation
        h = C.c_void_p()
        b.check(b.mst_tcn_create(C.byref(d), C.byref(h)), "mst_tcn_create")
        self._handle, self._lib = h, b
        f = lambda t: t.detach().to("cpu", torch.float32).contiguous()
        for n, blk in enumerate(held):
            _load_tcn_block(b, h, n, blk)
        ow, ob = f(self.output.weight), f(self.output.bias)
        b.check(b.mst_tcn_load_output(h, ow.data_ptr(), ob.data_ptr(), None), "mst_tcn_load_output")
        self._sig = sig

    def _close(self):
        if self._handle is not None:
            self._lib.mst_tcn_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self._close()
        except Exception:
            pass

    def _set_cond(self, b, x, cond):
        hp = self.hparams
        if isinstance(cond, (list, tuple)):     # one condition per block (reference's SeFa branch, :139-140)
            if len(cond) != hp.nblocks:
                raise ValueError("TCNModel.forward: a condition list needs one entry per block")
            if hp.channel_growth > 1:
                cond = cond[-1:]                # the handle holds the last block only
            c = torch.stack([ci.to(x.device, torch.float32) for ci in cond], 0).contiguous()   # [nblocks, rows, D]
            rows, stride = c.shape[1], c.shape[1] * c.shape[2]
        else:
            c = cond.to(x.device, torch.float32).contiguous()
            rows, stride = c.shape[0], 0
        if c.shape[-1] != hp.cond_dim:
            raise ValueError(f"TCNModel.forward: condition length {c.shape[-1]} != cond_dim {hp.cond_dim}")
        if rows not in (1, x.shape[0]):
            raise RuntimeError(f"The size of tensor a ({rows}) must match the size of tensor b ({x.shape[0]}) at "
                               f"non-singleton dimension 0")
        b.check(b.mst_tcn_set_cond(self._handle, c.data_ptr(), rows, stride, _stream_ptr(x)), "mst_tcn_set_cond")
        return c   # keep alive until the kernels are enqueued

    def forward(self, x, cond):
        return self._run(x, cond, None)

    def forward_blocks(self, x, cond, n_run):
        """Parity probe: activations after the n_run-th block, fp32 [B, channel_width, L]."""
        return self._run(x, cond, n_run)

    def _run(self, x, cond, n_run):
        b = _check_device(x, "TCNModel.forward")
        with b.device_ctx(x):          # handle, weights, FiLM table, workspace and launches on the device the data lives on
            return self._run_on(b, x, cond, n_run)

    def _run_on(self, b, x, cond, n_run):
        hp = self.hparams
        if x.dim() != 3 or x.shape[1] != hp.ninputs:
            raise ValueError(f"TCNModel.forward: expected [B, {hp.ninputs}, L], got {tuple(x.shape)}")
        self._ensure(b)
        prec = _lib.PRECISIONS[self.precision]
        if hp.channel_growth > 1:
            if n_run is not None:
                raise NotImplementedError("TCNModel.forward_blocks: the probe exists for channel_growth == 1 nets")
            prec = _lib.MST_PREC_F32
            per_block = isinstance(cond, (list, tuple))
            if per_block and len(cond) != hp.nblocks:
                raise ValueError("TCNModel.forward: a condition list needs one entry per block")
            for idx, blk in enumerate(list(self.blocks)[:-1]):
                x = blk(x, cond[idx] if per_block else cond)
        x = x.contiguous()
        B, _, L = x.shape
        keep = self._set_cond(b, x, cond)
        nbytes = b.mst_tcn_workspace_bytes(self._handle, B, L, prec)
        ws = self._ws.get(nbytes, x.device)
        st = _stream_ptr(x)
        if n_run is None:
            y = torch.empty(B, hp.noutputs, L, dtype=torch.float32, device=x.device)
            b.check(b.mst_tcn_forward(self._handle, x.data_ptr(), y.data_ptr(), B, L, prec, ws.data_ptr(), nbytes, st),
                    "mst_tcn_forward")
        else:
            y = torch.empty(B, hp.channel_width, L, dtype=torch.float32, device=x.device)
            b.check(b.mst_tcn_forward_blocks(self._handle, x.data_ptr(), y.data_ptr(), B, L, prec, n_run, ws.data_ptr(),
                                             nbytes, st), "mst_tcn_forward_blocks")
        del keep
        return y
