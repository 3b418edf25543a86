"""LeNet-300-100 (784 -> 300 -> 100 -> 10), the shape source of BASELINE configs 1-2.
Counterpart of neural_network_compression/neural_networks/le_net_300_100.py:6-34."""
from __future__ import annotations

from typing import Any, Dict

import torch
from torch import nn

from .layers import Dense


class LeNet300100(nn.Module):
    def __init__(self, in_features: int = 28 * 28) -> None:
        super().__init__()
        self.dense1 = Dense(in_features, 300, activation=torch.relu)  # 235200 + 300
        self.dense2 = Dense(300, 100, activation=torch.relu)           # 30000 + 100
        self.out = Dense(100, 10)                                      # 1000 + 10

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.out(self.dense2(self.dense1(x)))

    def get_config(self) -> Dict[str, Any]:
        """name -> layer, in the order the reference's trainer walks them (le_net_300_100.py:29-34)."""
        return {"dense1": self.dense1, "dense2": self.dense2, "out": self.out}
