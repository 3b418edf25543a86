"""The LeNet-5 trainer the reference lists as a TODO (README.md:140; the network itself is
neural_network_compression/neural_networks/le_net_5.py:6-55, never wired to a trainer upstream).

Same surface as LeNet300100Trainer (le_net_300_100_trainer.py:9-39).  Pruning thresholds follow the
reference's convention for LeNet-300-100 -- weights at 1 sigma, biases at 0.1 sigma for the hidden
layers, the output layer at (0.5, 0) -- applied in network order, convolutions before dense layers
(the order the reference's report discusses, report.tex:120,245).  BASELINE configs[2]
(conv + dense weights, 5-bit forgy k-means + Huffman length histogram) runs through this class."""
from __future__ import annotations

from typing import Dict, Tuple

import torch

from .common.trainer import Trainer
from .neural_networks import LeNet5


class LeNet5Trainer(Trainer):
    def __init__(self, device=None) -> None:
        device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.neural_network = LeNet5().to(device)
        self.optimizer = torch.optim.Adam(self.neural_network.parameters(), lr=0.001)

    @property
    def model_name(self) -> str:
        return "LeNet5"

    @property
    def _layers_to_prune_with_threshold(self) -> Dict[torch.nn.Module, Tuple[float, float]]:
        net = self.neural_network
        return {net.conv1: (1, 0.1), net.conv2: (1, 0.1), net.dense: (1, 0.1), net.logits: (0.5, 0)}

    def _get_error(self, input_data: torch.Tensor, expected_output: torch.Tensor) -> torch.Tensor:
        logits = self.neural_network(input_data)
        cross_entropy = torch.nn.functional.binary_cross_entropy_with_logits(logits, expected_output)
        net = self.neural_network
        l2 = sum((layer.kernel ** 2).sum() / 2 for layer in (net.conv1, net.conv2, net.dense, net.logits))
        return cross_entropy + 0.01 * l2
