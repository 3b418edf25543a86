"""The LeNet-300-100 trainer: the reference's per-layer pruning thresholds, loss and optimiser
(neural_network_compression/le_net_300_100_trainer.py:9-39)."""
from __future__ import annotations

from typing import Dict, Tuple

import torch

from .common.trainer import Trainer
from .neural_networks import LeNet300100


class LeNet300100Trainer(Trainer):
    def __init__(self, device=None, in_features: int = 28 * 28) -> None:
        device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.neural_network = LeNet300100(in_features).to(device)
        self.optimizer = torch.optim.Adam(self.neural_network.parameters(), lr=0.001)

    @property
    def model_name(self) -> str:
        return "LeNet300100"

    @property
    def _layers_to_prune_with_threshold(self) -> Dict[torch.nn.Module, Tuple[float, float]]:
        # (weight q, bias q) in units of the tensor's standard deviation (le_net_300_100_trainer.py:21-27)
        net = self.neural_network
        return {net.dense1: (1, 0.1), net.dense2: (1, 0.1), net.out: (0.5, 0)}

    def _get_error(self, input_data: torch.Tensor, expected_output: torch.Tensor) -> torch.Tensor:
        logits = self.neural_network(input_data)
        cross_entropy = torch.nn.functional.binary_cross_entropy_with_logits(logits, expected_output)
        net = self.neural_network
        # tf.nn.l2_loss(w) = sum(w^2) / 2 over the three kernels, weighted 0.01
        l2 = sum((layer.kernel ** 2).sum() / 2 for layer in (net.dense1, net.dense2, net.out))
        return cross_entropy + 0.01 * l2
