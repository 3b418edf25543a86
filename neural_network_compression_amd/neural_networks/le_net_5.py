"""LeNet-5 as the reference defines it (conv 5x5x20 "same" -> pool 2/2 -> conv 5x5x50 "same" -> pool 2/2 ->
dense 256 -> dropout -> dense 10; 28x28x1 images give the 2450 inputs of the dense layer), the shape source of
BASELINE config 3.  Counterpart of neural_network_compression/neural_networks/le_net_5.py:6-55 (never wired
to a trainer upstream); layer names as in its get_config()."""
from __future__ import annotations

from typing import Any, Dict

import torch
import torch.nn.functional as F
from torch import nn

from .layers import Conv2D, Dense, Weightless


def _pool(x: torch.Tensor) -> torch.Tensor:   # MaxPooling2D(pool_size=2, strides=2) on NHWC
    return F.max_pool2d(x.permute(0, 3, 1, 2), 2, stride=2).permute(0, 2, 3, 1)


class LeNet5(nn.Module):
    def __init__(self) -> None:
        super().__init__()
        self.conv1 = Conv2D(1, 20, 5, activation=torch.relu, padding="same")    # (5,5,1,20) + 20
        self.pool1 = Weightless(_pool)
        self.conv2 = Conv2D(20, 50, 5, activation=torch.relu, padding="same")   # (5,5,20,50) + 50
        self.pool2 = Weightless(_pool)
        self.flatten = Weightless(lambda x: x.reshape(x.shape[0], -1))           # tf.reshape in the reference's call()
        self.dense = Dense(2450, 256, activation=torch.relu)                     # (2450,256) + 256
        self.dropout = Weightless(lambda x: x)                                   # identity at inference
        self.logits = Dense(256, 10)                                             # (256,10) + 10

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        for layer in (self.conv1, self.pool1, self.conv2, self.pool2, self.flatten, self.dense, self.dropout, self.logits):
            x = layer(x)
        return x

    def get_config(self) -> Dict[str, Any]:
        return {"conv1": self.conv1, "pool1": self.pool1, "conv2": self.conv2, "pool2": self.pool2,
                "dense": self.dense, "dropout": self.dropout, "logits": self.logits}
