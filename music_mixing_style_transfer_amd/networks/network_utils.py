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
(csrc/), driven by FXencoder / TCNModel in architectures.py.  There is no torch fallback.
"""
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


class _HipOnly(nn.Module):
    def forward(self, *args, **kwargs):  # pragma: no cover - guard
        raise NotImplementedError(
            f"{type(self).__name__} is a parameter container; it runs on MI355X inside FXencoder / TCNModel / "
            f"Res_ConvBlock.forward (libmst_hip.so).  There is no torch fallback.")


class Conv1d_layer(_HipOnly):
    """ReflectionPad1d -> Conv1d -> BatchNorm1d -> ReLU (reference order conv -> norm -> activation)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding="SAME", dilation=1, bias=True,
                 norm="batch", activation="relu", mode="conv"):
        super().__init__()
        if mode != "conv":
            raise NotImplementedError(f"Conv1d_layer mode '{mode}' is not part of the inference hot path")
        if padding not in ("SAME", "VALID"):
            raise ValueError("padding must be 'SAME' or 'VALID'")
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.dilation = kernel_size, stride, dilation
        self.padding_area = same_padding(kernel_size, dilation) if padding == "SAME" else (0, 0)
        self.norm, self.activation = norm, activation
        self.conv1d = nn.Sequential()
        self.conv1d.add_module("conv1d_pad", nn.ReflectionPad1d(self.padding_area))
        self.conv1d.add_module("conv1d", nn.Conv1d(in_channels, out_channels, kernel_size, stride=stride, padding=0,
                                                   dilation=dilation, bias=bias))
        if norm == "batch":
            self.conv1d.add_module("batch_norm", nn.BatchNorm1d(out_channels))
        if activation == "relu":
            self.conv1d.add_module("relu", nn.ReLU())
        elif activation == "lrelu":
            self.conv1d.add_module("lrelu", nn.LeakyReLU())

    def hip_supported(self):
        return self.norm == "batch" and self.activation == "relu" and self.padding_area == same_padding(
            self.kernel_size, self.dilation)

    def export_arrays(self):
        """Host fp32 arrays in the reference's layouts for mst_enc_load_conv."""
        conv, bn = self.conv1d.conv1d, self.conv1d.batch_norm
        f = lambda t: None if t is None else t.detach().to("cpu", torch.float32).contiguous()
        return dict(w=f(conv.weight), bias=f(conv.bias), bn_w=f(bn.weight), bn_b=f(bn.bias), bn_mean=f(bn.running_mean),
                    bn_var=f(bn.running_var), eps=float(bn.eps))


class Res_ConvBlock(_DeviceState, nn.Module):
    _DEVICE_STATE = (("_runner", None),)

    """conv2(conv1(x) + x): the skip is added after conv1's activation; only conv2 strides / changes channels."""

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


class ConvBlock(_HipOnly):
    """layer_num stacked Conv1d_layers; only the last one changes channels / strides (reference :126-151).
    Constructible for state_dict compatibility; the default configs use conv_block='res'."""

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


class FiLM(_HipOnly):
    """Feature-wise linear modulation: film_fc(condition) -> split (scale r = first half, shift b = second
    half) -> r * feature + b.  Inside TCNModel the factors for all blocks are produced by mst_tcn_set_cond and
    applied in the TCN block kernel's epilogue."""

    def __init__(self, condition_len=2048, feature_len=1024):
        super().__init__()
        self.film_fc = nn.Linear(condition_len, feature_len * 2)
        self.feat_len = feature_len
