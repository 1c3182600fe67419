"""Building blocks of the networks module API - the drop-in boundary of the reference
(`from networks import FXencoder, TCNModel`, inference/style_transfer.py:22; networks/__init__.py star-exports
network_utils too).  Same class names, constructor arguments and state_dict key names/shapes as
networks/network_utils.py of the reference, so reference checkpoints load with strict=True:

    Conv1d_layer   :15-89   conv1d.{conv1d.weight, conv1d.bias, batch_norm.{weight,bias,running_mean,running_var,
                            num_batches_tracked}}
    Res_ConvBlock  :96-119  conv1.*, conv2.*
    ConvBlock      :126-151 conv_block.{i}.*
    FiLM           :156-182 film_fc.{weight,bias}

The torch.nn modules inside are parameter CONTAINERS only: arithmetic happens in the gfx950 library
(csrc/).  Every exported module runs on its own as well (Conv1d_layer / ConvBlock: mst_enc_forward_conv,
FiLM: mst_film_forward, Res_ConvBlock: a one-block encoder handle); FXencoder / TCNModel in architectures.py
drive whole stacks.  There is no torch fallback.
"""
import ctypes as C

import torch
import torch.nn as nn

from .. import _lib


def same_padding(kernel_size, dilation=1):
    """(left, right) reflection padding of mode 'conv' with padding='SAME' (reference network_utils.py:30-34):
    total (k-1)*d, left = total // 2 - even kernels pad one more sample on the right."""
    total = int((kernel_size - 1) * dilation)
    return total // 2, total - total // 2


class _DeviceState:
    """Mixin for modules that cache a library handle: the handle, its workspace and the binding are per-process device
    state and never travel through copy.deepcopy / pickle (the copy rebuilds them on first use)."""
    _DEVICE_STATE = ()

    def __getstate__(self):
        state = self.__dict__.copy()
        for k, v in self._DEVICE_STATE:
            state[k] = v() if callable(v) else v
        return state


def _signature(module):
    return tuple((p.data_ptr(), p._version, p.device) for p in list(module.parameters()) + list(module.buffers()))


def _device_input(t, what):
    b = _lib.lib()
    b.require_device(t, what)
    if t.dtype != torch.float32:
        raise TypeError(f"{what}: float32 input expected, got {t.dtype}")
    return b


class Conv1d_layer(_DeviceState, nn.Module):
    """mode "conv": ReflectionPad1d -> Conv1d -> BatchNorm1d -> ReLU (reference order conv -> norm -> activation);
    mode "deconv" (reference :24-26,38-42; not used by the inference path, provided for the module API): ConvTranspose1d with
    padding = dilation * (k - 1) / 2 and output_padding = (stride > 1) -> BatchNorm1d -> activation, computed as the stride-1 VALID
    convolution of the zero-stuffed input (mst_enc_zero_stuff) with the tap-reversed, channel-transposed kernel.
    The "alias_free_*" modes need torchaudio's resampler and stay unavailable."""
    _DEVICE_STATE = (("_handle", None), ("_sig", None), ("_hlib", None))

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding="SAME", dilation=1, bias=True,
                 norm="batch", activation="relu", mode="conv"):
        super().__init__()
        if mode not in ("conv", "deconv"):
            raise NotImplementedError(f"Conv1d_layer mode '{mode}' is not part of the inference hot path")
        if mode == "conv" and padding not in ("SAME", "VALID"):
            raise ValueError("padding must be 'SAME' or 'VALID'")
        self.mode = mode
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.dilation = kernel_size, stride, dilation
        self.norm, self.activation = norm, activation
        self._handle = self._sig = self._hlib = None
        self.conv1d = nn.Sequential()
        if mode == "deconv":
            self.deconv_padding = int(dilation * (kernel_size - 1) / 2)
            self.output_padding = 0 if stride == 1 else 1
            self.padding_area = (0, 0)
            self.conv1d.add_module("deconv1d", nn.ConvTranspose1d(in_channels, out_channels, kernel_size, stride=stride,
                                                                  padding=self.deconv_padding, output_padding=self.output_padding,
                                                                  dilation=dilation, bias=bias))
        else:
            self.padding_area = same_padding(kernel_size, dilation) if padding == "SAME" else (0, 0)
            self.conv1d.add_module("conv1d_pad", nn.ReflectionPad1d(self.padding_area))
            self.conv1d.add_module("conv1d", nn.Conv1d(in_channels, out_channels, kernel_size, stride=stride, padding=0,
                                                       dilation=dilation, bias=bias))
        if norm == "batch":
            self.conv1d.add_module("batch_norm", nn.BatchNorm1d(out_channels))
        if activation == "relu":
            self.conv1d.add_module("relu", nn.ReLU())
        elif activation == "lrelu":
            self.conv1d.add_module("lrelu", nn.LeakyReLU())

    def act_slope(self):
        """The activation as MstEncDesc.act_slope: what the layer does to negative values (ReLU 0, nn.LeakyReLU() 0.01, none 1)."""
        return {"relu": 0.0, "lrelu": 0.01}.get(self.activation, 1.0)

    def hip_supported(self):
        """Usable inside an encoder handle (Res_ConvBlock stack): SAME padding; any norm / activation the reference's layer accepts."""
        return self.mode == "conv" and self.padding_area == same_padding(self.kernel_size, self.dilation)

    # ---- the layer on its own (reference :86-89) ------------------------------------------------
    def _ensure(self, b):
        sig = _signature(self)
        if self._handle is not None and sig == self._sig and self._hlib is b:
            return
        self._close()
        d = _lib.MstEncDesc()
        d.act_slope = self.act_slope()
        d.nblocks = 1
        d.channels[0], d.channels[1] = self.in_channels, self.out_channels
        d.kernels[0], d.strides[0], d.dilations[0] = self.kernel_size, (1 if self.mode == "deconv" else self.stride), self.dilation
        d.valid_padding = 1 if self.padding_area == (0, 0) and self.kernel_size > 1 else 0
        h = C.c_void_p()
        b.check(b.mst_enc_create(C.byref(d), C.byref(h)), "mst_enc_create")
        self._handle, self._hlib = h, b
        a = self.export_arrays()
        ptr = lambda t: None if t is None else t.data_ptr()
        b.check(b.mst_enc_load_conv(h, 0, 1, ptr(a["w"]), ptr(a["bias"]), ptr(a["bn_w"]), ptr(a["bn_b"]), ptr(a["bn_mean"]),
                                    ptr(a["bn_var"]), a["eps"], None), "mst_enc_load_conv")
        self._sig = sig

    def _close(self):
        if self._handle is not None:
            self._hlib.mst_enc_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self._close()
        except Exception:
            pass

    def forward(self, input):
        b = _device_input(input, "Conv1d_layer.forward")
        if input.dim() != 3 or input.shape[1] != self.in_channels:
            raise ValueError(f"Conv1d_layer.forward: expected [B, {self.in_channels}, L], got {tuple(input.shape)}")
        with b.device_ctx(input):
            self._ensure(b)
            x = input.contiguous()
            B, _, L = x.shape
            if self.mode == "deconv":          # the zero-stuffed input of the equivalent stride-1 VALID convolution
                edge = self.dilation * (self.kernel_size - 1) - self.deconv_padding
                if edge < 0:
                    raise RuntimeError("Conv1d_layer(mode='deconv'): padding larger than the kernel's reach")
                Lu = edge + (L - 1) * self.stride + 1 + edge + self.output_padding
                xu = torch.empty(B, self.in_channels, Lu, dtype=torch.float32, device=x.device)
                b.check(b.mst_enc_zero_stuff(x.data_ptr(), xu.data_ptr(), B * self.in_channels, L, self.stride, edge, Lu, b.stream_ptr(x)),
                        "mst_enc_zero_stuff")
                x, L = xu, Lu
            lout = b.mst_enc_conv_length(self._handle, 0, 1, L)
            if lout < 1 or L <= max(self.padding_area):
                raise RuntimeError("Conv1d_layer.forward: the input is too short for this kernel / reflection padding")
            y = torch.empty(B, self.out_channels, lout, dtype=torch.float32, device=x.device)
            b.check(b.mst_enc_forward_conv(self._handle, 0, 1, x.data_ptr(), y.data_ptr(), B, L, b.stream_ptr(x)), "mst_enc_forward_conv")
        return y

    def export_arrays(self):
        """Host fp32 arrays in the reference's layouts for mst_enc_load_conv."""
        f = lambda t: None if t is None else t.detach().to("cpu", torch.float32).contiguous()
        if self.mode == "deconv":      # ConvTranspose1d weight [Cin, Cout, k] -> the equivalent convolution's [Cout, Cin, k], taps reversed
            import types
            dc = self.conv1d.deconv1d
            conv = types.SimpleNamespace(weight=dc.weight.detach().permute(1, 0, 2).flip(-1), bias=dc.bias)
        else:
            conv = self.conv1d.conv1d
        if self.norm != "batch":        # no normalisation layer (network_utils.py:70-73): the identity in BatchNorm form, folded exactly
            one, zero = torch.ones(self.out_channels), torch.zeros(self.out_channels)
            return dict(w=f(conv.weight), bias=f(conv.bias), bn_w=one, bn_b=zero, bn_mean=zero.clone(), bn_var=one.clone(), eps=0.0)
        bn = self.conv1d.batch_norm
        return dict(w=f(conv.weight), bias=f(conv.bias), bn_w=f(bn.weight), bn_b=f(bn.bias), bn_mean=f(bn.running_mean),
                    bn_var=f(bn.running_var), eps=float(bn.eps))


class Res_ConvBlock(_DeviceState, nn.Module):
    """conv2(conv1(x) + x): the skip is added after conv1's activation; only conv2 strides / changes channels."""

    _DEVICE_STATE = (("_runner", None),)

    def __init__(self, dimension, in_channels, out_channels, kernel_size, stride=1, padding="SAME", dilation=1,
                 bias=True, norm="batch", activation="relu", last_activation="relu", mode="conv"):
        super().__init__()
        if dimension != 1:
            raise NotImplementedError("only 1-d blocks exist on the inference path")
        self.conv1 = Conv1d_layer(in_channels, in_channels, kernel_size, padding=padding, dilation=dilation, bias=bias,
                                  norm=norm, activation=activation)
        self.conv2 = Conv1d_layer(in_channels, out_channels, kernel_size, stride=stride, padding=padding,
                                  dilation=dilation, bias=bias, norm=norm, activation=last_activation, mode=mode)
        self._runner = None

    def forward(self, input):
        from .architectures import _EncoderRunner   # single-block encoder handle, pooling skipped
        if self._runner is None:
            self._runner = _EncoderRunner([self])
        return self._runner.run(input, pooled=False)


class ConvBlock(nn.Module):
    """layer_num stacked Conv1d_layers; only the last one changes channels / strides (reference :126-151)."""

    def __init__(self, dimension, layer_num, in_channels, out_channels, kernel_size, stride=1, padding="SAME",
                 dilation=1, bias=True, norm="batch", activation="relu", last_activation="relu", mode="conv"):
        super().__init__()
        if dimension != 1:
            raise NotImplementedError("only 1-d blocks exist on the inference path")
        layers = [Conv1d_layer(in_channels, in_channels, kernel_size, padding=padding, dilation=dilation, bias=bias,
                               norm=norm, activation=activation) for _ in range(layer_num - 1)]
        layers.append(Conv1d_layer(in_channels, out_channels, kernel_size, stride=stride, padding=padding,
                                   dilation=dilation, bias=bias, norm=norm, activation=last_activation, mode=mode))
        self.conv_block = nn.Sequential(*layers)

    def forward(self, input):
        return self.conv_block(input)


class FiLM(nn.Module):
    """Feature-wise linear modulation: film_fc(condition) -> split (scale r = first half, shift b = second
    half) -> r * feature + b.  Inside TCNModel the factors for all blocks are produced by mst_tcn_set_cond and
    applied in the TCN block kernel's epilogue; on its own the module runs mst_film_forward."""

    def __init__(self, condition_len=2048, feature_len=1024):
        super().__init__()
        self.film_fc = nn.Linear(condition_len, feature_len * 2)
        self.feat_len = feature_len

    def _sefa_shift(self, condition, sefa):
        """SeFA edit of the condition (reference network_utils.py:164-178): sefa = (eigen index, scale).  The reference calls torch.eig,
        which current torch no longer has; torch.linalg.eig is the same LAPACK geev (eigenvalue order included).  A once-per-condition
        2048 x 2048 host-side decomposition - control plane, not the hot path; parity unpinned (the reference's call cannot run here).
        Like the reference, `condition` is modified IN PLACE and row `index` (not column) of the eigenvector matrix is used."""
        weight = self.film_fc.weight.detach().to("cpu", torch.float32).T
        weight = weight / torch.linalg.norm(weight + 1e-07, dim=0, keepdims=True)
        values, vectors = torch.linalg.eig(torch.matmul(weight, weight.T))
        chosen = sefa[0]
        alpha = values.real[chosen] * sefa[1]
        shift = (alpha * vectors.real[chosen]).to(condition.device, condition.dtype)
        condition += shift.repeat(condition.shape[0], 1)
        return condition

    def forward(self, feature, condition, sefa=None):
        """feature [B, feat_len, L] (a 1-d conv feature map) or [B, feat_len] (linear), condition [1|B, condition_len];
        sefa: None or (eigen index, scale) - see _sefa_shift."""
        b = _device_input(feature, "FiLM.forward")
        if sefa:
            condition = self._sefa_shift(condition, sefa)
        if feature.dim() not in (2, 3) or feature.shape[1] != self.feat_len:
            raise ValueError(f"FiLM.forward: expected a [B, {self.feat_len}(, L)] feature, got {tuple(feature.shape)}")
        x = feature.contiguous()
        cond = condition.to(x.device, torch.float32).contiguous()
        if cond.dim() != 2 or cond.shape[1] != self.film_fc.in_features:
            raise ValueError(f"FiLM.forward: condition must be [rows, {self.film_fc.in_features}]")
        B, L = x.shape[0], (x.shape[2] if x.dim() == 3 else 1)
        if cond.shape[0] not in (1, B):
            raise RuntimeError(f"The size of tensor a ({cond.shape[0]}) must match the size of tensor b ({B}) at non-singleton dimension 0")
        w, bias = self.film_fc.weight.detach().contiguous(), self.film_fc.bias.detach().contiguous()
        if w.device != x.device:
            raise RuntimeError("FiLM.forward: the module's parameters and the feature live on different devices")
        with b.device_ctx(x):
            y = torch.empty_like(x)
            table = torch.empty(cond.shape[0] * 2 * self.feat_len, dtype=torch.float32, device=x.device)
            b.check(b.mst_film_forward(w.data_ptr(), bias.data_ptr(), cond.data_ptr(), cond.shape[0], cond.shape[1], self.feat_len,
                                       x.data_ptr(), y.data_ptr(), B, L, table.data_ptr(), b.stream_ptr(x)), "mst_film_forward")
        return y
