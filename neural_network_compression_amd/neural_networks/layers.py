"""Keras-shaped layers on torch, weights resident on the GPU.

The reference's trainer talks to its layers through ``get_weights()`` / ``set_weights()``
(common/trainer.py:52,70,185,193,203,206) and relies on the Keras tensor layouts (Dense kernel
(in, out); Conv2D kernel (h, w, in, out)) -- the flattening order decides the order of NumPy's
float32 standard-deviation sum, hence the last bit of sigma.  These layers keep exactly those
layouts; ``get_weights()`` hands out the live device tensors (no host round trip), so pruning
and quantisation run in place in HBM.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F
from torch import nn


class KerasLikeLayer(nn.Module):
    def get_weights(self):
        return [p.data for p in self.parameters()]

    def set_weights(self, tensors):
        for p, t in zip(self.parameters(), tensors):
            if t.data_ptr() != p.data.data_ptr():
                p.data.copy_(torch.as_tensor(t, device=p.device, dtype=p.dtype).reshape(p.shape))


class Dense(KerasLikeLayer):
    """y = act(x @ kernel + bias), kernel stored (in_features, units) as Keras does."""

    def __init__(self, in_features: int, units: int, activation=None):
        super().__init__()
        limit = math.sqrt(6.0 / (in_features + units))  # Keras default: glorot_uniform
        self.kernel = nn.Parameter(torch.empty(in_features, units).uniform_(-limit, limit))
        self.bias = nn.Parameter(torch.zeros(units))
        self.activation = activation

    def forward(self, x):
        y = x @ self.kernel + self.bias
        return self.activation(y) if self.activation is not None else y


class Conv2D(KerasLikeLayer):
    """Convolution with Keras' padding "valid" or "same" (odd kernels), NHWC input, kernel stored
    (h, w, in, out) as Keras does."""

    def __init__(self, in_channels: int, filters: int, kernel_size: int, activation=None, padding: str = "valid"):
        super().__init__()
        if padding not in ("valid", "same"):
            raise ValueError("padding must be 'valid' or 'same'")
        self.pad = kernel_size // 2 if padding == "same" else 0
        fan_in, fan_out = kernel_size * kernel_size * in_channels, kernel_size * kernel_size * filters
        limit = math.sqrt(6.0 / (fan_in + fan_out))
        self.kernel = nn.Parameter(torch.empty(kernel_size, kernel_size, in_channels, filters).uniform_(-limit, limit))
        self.bias = nn.Parameter(torch.zeros(filters))
        self.activation = activation

    def forward(self, x):  # x: (N, H, W, C)
        y = F.conv2d(x.permute(0, 3, 1, 2), self.kernel.permute(3, 2, 0, 1), self.bias, padding=self.pad)
        y = y.permute(0, 2, 3, 1)
        return self.activation(y) if self.activation is not None else y


class Weightless(KerasLikeLayer):
    """Pooling / flatten / dropout: present in LeNet-5's layer dict, no weights (get_weights() == [])."""

    def __init__(self, fn):
        super().__init__()
        self.fn = fn

    def forward(self, x):
        return self.fn(x)
