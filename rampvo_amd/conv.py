"""Convolution front-end of the RAMP encoder towers.

All activations are channels-last ([N,C,H,W] shape, NHWC storage).  Two
back-ends implement the same three primitives:

  * ``hip``   -- the implicit-GEMM MFMA kernels of csrc/conv.hip (conv + bias with
                 fused per-channel InstanceNorm statistics, normalise + ReLU +
                 residual add epilogues);
  * ``torch`` -- ATen/MIOpen, kept as the bring-up path and as the numerics
                 reference the conv kernels are tested against.

Selected by ``set_backend`` / env RAMP_CONV_BACKEND (default: ``hip`` when the
library exports the conv entry points, else ``torch``).
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

_backend = os.environ.get("RAMP_CONV_BACKEND", "auto")


def set_backend(name):
    global _backend
    assert name in ("auto", "hip", "torch")
    _backend = name


def use_hip(x):
    if _backend == "torch" or not x.is_cuda:
        return False
    try:
        from . import conv_hip
        ok = conv_hip.available()
    except Exception:
        ok = False
    if _backend == "hip" and not ok:
        raise RuntimeError("RAMP_CONV_BACKEND=hip but libramp_hip.so has no conv kernels")
    return ok


def _is_instance(norm):
    return isinstance(norm, nn.InstanceNorm2d)


def _is_identity(norm):
    return norm is None or (isinstance(norm, nn.Sequential) and len(norm) == 0)


def conv_norm_relu(x, conv, norm, relu, out_scale=1.0):
    """y = [relu]([instance_norm](conv(x) + b)) * out_scale, channels-last in and out"""
    x = x.contiguous(memory_format=torch.channels_last)
    y = F.conv2d(x.to(conv.weight.dtype), conv.weight, conv.bias, conv.stride, conv.padding)
    if _is_instance(norm):
        y = F.instance_norm(y, eps=norm.eps)
    elif not _is_identity(norm):
        y = norm(y)
    if relu:
        y = F.relu(y, inplace=True)
    if out_scale != 1.0:
        y = y * out_scale
    return y


def add_relu(x, y):
    return F.relu(x + y, inplace=True)


def cat_channels(x, y):
    return torch.cat((x, y.to(x.dtype)), dim=1).contiguous(memory_format=torch.channels_last)
