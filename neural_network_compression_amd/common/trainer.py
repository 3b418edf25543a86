"""Trainer with the reference's call surface, weights and masks resident on the GPU.

Counterpart of neural_network_compression/common/trainer.py.  The compression steps are the
hot path and run in the HIP kernels:

    Trainer._prune_parameters        (trainer.py:177-193)  -> utility.prune_weigth per tensor, in place in HBM
    Trainer._reset_pruned_parameters (trainer.py:195-206)  -> ops.apply_mask_ with the stored device masks
    Trainer.quantize                 (trainer.py:42-72)    -> utility.get_weight_distribution (zeros skipped on
                                                              the device) + utility.get_quantized_weight

The reference round-trips every tensor through host NumPy on every batch
(layer.get_weights()/set_weights()); here nothing leaves the device.  The gradient / Adam / accuracy
parts (trainer.py:208-232, TensorFlow upstream) are a minimal torch loop: they are callers of the path,
not the path.
"""
from __future__ import annotations

import pathlib
from abc import ABC, abstractmethod
from typing import Dict, List, NamedTuple, Tuple

import numpy as np
import torch

from .. import ops
from . import utility


class LeNetDataset(NamedTuple):
    input_data: np.ndarray
    output_data: np.ndarray


def _batches(x: torch.Tensor, y: torch.Tensor, batch: int = 512, shuffle_buffer: int = 1000):
    """tf.data .shuffle(1000).batch(512, drop_remainder=True) in spirit: a windowed shuffle."""
    n = x.shape[0]
    order = torch.arange(n, device=x.device)
    for lo in range(0, n, shuffle_buffer):
        hi = min(n, lo + shuffle_buffer)
        order[lo:hi] = order[lo:hi][torch.randperm(hi - lo, device=x.device)]
    for lo in range(0, n - batch + 1, batch):
        idx = order[lo: lo + batch]
        yield x[idx], y[idx]


class Trainer(ABC):
    neural_network: torch.nn.Module
    optimizer: torch.optim.Optimizer

    # zero-weight masks per layer; class level, as in the reference (trainer.py:25)
    pruned_indexes_by_layer: Dict[torch.nn.Module, Tuple[torch.Tensor, torch.Tensor]] = {}

    @property
    @abstractmethod
    def model_name(self) -> str:
        """The model name."""

    @property
    @abstractmethod
    def _layers_to_prune_with_threshold(self) -> Dict[torch.nn.Module, Tuple[float, float]]:
        """layer -> (weight threshold, bias threshold)."""

    @abstractmethod
    def _get_error(self, input_data: torch.Tensor, expected_output: torch.Tensor) -> torch.Tensor:
        """The loss."""

    # ------------------------------------------------------------------ device plumbing
    @property
    def device(self) -> torch.device:
        return next(self.neural_network.parameters()).device

    def _to_device(self, a) -> torch.Tensor:
        return torch.as_tensor(np.asarray(a)).to(self.device)

    # ------------------------------------------------------------------ the hot path
    def quantize(self, test_dataset: LeNetDataset, with_cumulative_weight_distribution: bool,
                 maximum_centroid_bits: int, k_means_initialization_mode: str, *, arith: str = "auto", reloc: str = "auto") -> float:
        """The reference's signature (common/trainer.py:42-48) plus two keywords that select its own arithmetic where identity
        with it is wanted (utility.get_quantized_weight): ``arith="reference"`` = scikit-learn's float32 running sums in sample
        order on tensors of any length (centres, indices and n_iter_ then equal the reference's on one thread bit for bit; slower:
        a host round trip per Lloyd iteration beyond 4096 weights), ``reloc="reference"`` = numpy.argpartition's own choice of the
        far samples at an empty-cluster event.  The defaults are the order-independent exact sums and the on-device selection."""
        self.quantized_models_by_layer = {}   # layer -> [fitted model or None per tensor]: what fine_tune_centroids needs
        layers = [layer for _layer_name, layer in self.neural_network.get_config().items()]
        # The tensors of the network are independent: where the init draws nothing from NumPy's global generator (linear,
        # density) they go through the library's one-call layer, several side by side on streams of their own
        # (pipeline.compress_layers) -- the same results as the calls below one after the other
        # (tests/test_gpu_config5.py::test_layer_as_one_library_call_equals_the_step_by_step_path, test_layers_side_by_side_...).
        # Everything else -- forgy / kmeans++ (global generator, in layer order), tensors too short for the number of centroids
        # (the reference's "not enough bits" pass-through), the error cases -- takes the reference's own sequence of calls.
        mode, bits = k_means_initialization_mode, maximum_centroid_bits
        batched = {}
        if (arith, reloc) == ("auto", "auto") and mode in ("linear", "density") and (mode != "density" or with_cumulative_weight_distribution) and isinstance(bits, int) and 1 <= bits <= 10:
            from .. import pipeline

            todo = [(li, ti, params) for li, layer in enumerate(layers) for ti, params in enumerate(layer.get_weights())
                    if isinstance(params, torch.Tensor) and params.is_cuda and params.dtype == torch.float32 and params.is_contiguous()
                    and params.numel() >= 2 ** bits + 1]
            if todo:
                res = pipeline.compress_layers([p.reshape(-1) for _, _, p in todo], workers=8, q=None, bits=bits, mode=mode,
                                               with_cdf=(mode == "density"), huffman=True, want_values=True)
                for (li, ti, p), r in zip(todo, res):
                    batched[(li, ti)] = (r.values.view(p.shape), r.model)
        for li, layer in enumerate(layers):
            quantized_weights_and_bias = []
            models = []
            for ti, params in enumerate(layer.get_weights()):
                if (li, ti) in batched:
                    quantized, model = batched[(li, ti)]
                else:
                    cdfs = None
                    if with_cumulative_weight_distribution:
                        # the reference strips exact zeros with numpy.delete first (trainer.py:55-59);
                        # the device kernels skip them instead -- same histogram, no compaction
                        cdfs = utility.get_weight_distribution(params, skip_zeros=True)
                    quantized, model = utility.get_quantized_weight(params, bits=maximum_centroid_bits,
                                                                    mode=k_means_initialization_mode, cdfs=cdfs, arith=arith, reloc=reloc)
                quantized_weights_and_bias.append(quantized)
                models.append(model)
            layer.set_weights(quantized_weights_and_bias)
            if models:
                self.quantized_models_by_layer[layer] = models
        return self._get_accuracy(test_dataset)

    def fine_tune_centroids(self, train_dataset: LeNetDataset, test_dataset: LeNetDataset, epochs: int,
                            learning_rate: float = 1e-3) -> List[float]:
        """Deep Compression's trained quantization, which the reference describes and leaves out as too slow on the host
        (papers/lat/report.tex:149-158): after ``quantize`` the centroid indices stay fixed; per batch the gradient of
        every quantized tensor is summed per centroid (ops.centroid_gradient: dL/dC_k = sum of dL/dW over cluster k, on
        the device), the centroids take a plain gradient step and the tensor is re-decoded from its indices
        (ops.gather).  Tensors that passed through unquantized are left alone.  Returns the accuracy per epoch."""
        models = getattr(self, "quantized_models_by_layer", None)
        if not models:
            raise RuntimeError("fine_tune_centroids needs a quantized network: call quantize first")
        x = self._to_device(train_dataset.input_data).float()
        y = self._to_device(train_dataset.output_data).float()
        centers = {}
        for layer, ms in models.items():
            for ti, m in enumerate(ms):
                if m is not None:
                    centers[(layer, ti)] = torch.from_numpy(np.ascontiguousarray(m.cluster_centers_.ravel())).to(self.device)
        accuracies = []
        for _ in range(epochs):
            for xb, yb in _batches(x, y):
                self.optimizer.zero_grad(set_to_none=True)
                self._get_error(xb, yb).backward()
                with torch.no_grad():
                    for layer, ms in models.items():
                        tensors = layer.get_weights()
                        params = layer.trainable_tensors() if hasattr(layer, "trainable_tensors") else list(layer.parameters())
                        for ti, m in enumerate(ms):
                            if m is None or params[ti].grad is None:
                                continue
                            c = centers[(layer, ti)]
                            g = ops.centroid_gradient(params[ti].grad.contiguous(), m.labels_compact_, c.numel())
                            c.sub_((learning_rate * g).to(torch.float32))
                            tensors[ti] = ops.gather(c, m.labels_compact_).view(tensors[ti].shape)
                        layer.set_weights(tensors)
            accuracies.append(self._get_accuracy(test_dataset))
        for (layer, ti), c in centers.items():
            models[layer][ti].cluster_centers_ = c.cpu().numpy().reshape(-1, 1)
        return accuracies

    def _prune_parameters(self, with_standard_deviation_smoothing: bool) -> None:
        for layer, (weight_threshold, bias_threshold) in self._layers_to_prune_with_threshold.items():
            weights, biases = layer.get_weights()
            zero_weight = utility.prune_weigth(weights, threshold=weight_threshold,
                                               std_smooth=with_standard_deviation_smoothing)
            zero_bias = utility.prune_weigth(biases, threshold=bias_threshold,
                                             std_smooth=with_standard_deviation_smoothing)
            self.pruned_indexes_by_layer[layer] = (zero_weight, zero_bias)
            layer.set_weights([weights, biases])

    def _reset_pruned_parameters(self) -> None:
        for layer, (zero_weight, zero_bias) in self.pruned_indexes_by_layer.items():
            weights, biases = layer.get_weights()
            ops.apply_mask_(weights, zero_weight)
            ops.apply_mask_(biases, zero_bias)
            layer.set_weights([weights, biases])

    # ------------------------------------------------------------------ callers of the path
    def train(self, train_dataset: LeNetDataset, test_dataset: LeNetDataset, epochs: int) -> List[float]:
        return self._epochs(train_dataset, test_dataset, epochs, prune=False, reset=False)

    def semi_pruned_train(self, train_dataset: LeNetDataset, test_dataset: LeNetDataset, epochs: int) -> List[float]:
        return self._epochs(train_dataset, test_dataset, epochs, prune=False, reset=True)

    def pruned_train(self, train_dataset: LeNetDataset, test_dataset: LeNetDataset, epochs: int,
                     with_standard_deviation_smoothing: bool) -> List[float]:
        return self._epochs(train_dataset, test_dataset, epochs, prune=True, reset=True,
                            smoothing=with_standard_deviation_smoothing)

    def _epochs(self, train_dataset, test_dataset, epochs, prune, reset, smoothing=True) -> List[float]:
        x = self._to_device(train_dataset.input_data).float()
        y = self._to_device(train_dataset.output_data).float()
        accuracies = []
        for _ in range(epochs):
            for xb, yb in _batches(x, y):
                if prune:
                    self._prune_parameters(smoothing)
                self._apply_gradient(self._get_gradient(xb, yb))
                if reset:
                    self._reset_pruned_parameters()
            accuracies.append(self._get_accuracy(test_dataset))
        return accuracies

    def store_report(self, directory: str) -> None:
        """Zero counts per layer, as the reference's report.txt (common/trainer.py:154-175; its plots are out of scope).  After
        ``quantize`` the network is also written in its stored form (storage.save_compressed: codebook + Huffman-coded centroid
        indices, dense or relative-index sparse, whichever is smaller per tensor -> ``weights.nnc``) and the report gains what Deep
        Compression reports: bits per weight of every tensor and the compression ratio against 32-bit weights."""
        pathlib.Path(directory).mkdir(parents=True, exist_ok=True)
        report = ""
        for layer_name, layer in self.neural_network.get_config().items():
            tensors = layer.get_weights()
            if not tensors:
                continue
            weight_layer, bias_layer = tensors
            report += f"layer: {layer_name}\n"
            report += f"zeroed weights: {int((weight_layer == 0).sum())}\ntotal weights: {weight_layer.numel()}\n"
            report += f"zeroed biases: {int((bias_layer == 0).sum())}\ntotal weights: {bias_layer.numel()}\n\n"
        models = getattr(self, "quantized_models_by_layer", None)
        if models:
            from .. import storage

            stored = {}
            for layer_name, layer in self.neural_network.get_config().items():
                if layer not in models:
                    continue
                for kind, t, m in zip(("weights", "biases"), layer.get_weights(), models[layer]):
                    stored[f"{layer_name}.{kind}"] = (tuple(t.shape), m, t if m is None else None)
            rep = {}
            storage.save_compressed(f"{directory}/weights.nnc", stored, report=rep)
            self.compression_report = rep
            for name, r in rep.items():
                if name != "total":
                    report += f"stored {name}: {r['bytes']} bytes, {r['bits_per_weight']:.3f} bits per weight ({r['form']}, {r['k']} centroids)\n"
            t = rep["total"]
            report += (f"stored network: {t['bytes']} bytes for {t['n']} weights = {t['bits_per_weight']:.3f} bits per weight; "
                       f"compression ratio {t['compression_ratio']:.1f}x against float32\n")
        with open(f"{directory}/report.txt", "w") as f:
            f.write(report)

    def _get_gradient(self, input_data: torch.Tensor, expected_output: torch.Tensor):
        self.optimizer.zero_grad(set_to_none=True)
        loss = self._get_error(input_data, expected_output)
        loss.backward()
        return [p.grad for p in self.neural_network.parameters()]

    def _apply_gradient(self, gradient) -> None:
        for p, g in zip(self.neural_network.parameters(), gradient):
            p.grad = g
        self.optimizer.step()

    @torch.no_grad()
    def _get_accuracy(self, dataset: LeNetDataset) -> float:
        x = self._to_device(dataset.input_data).float()
        y = self._to_device(dataset.output_data)
        pred = torch.argmax(torch.softmax(self.neural_network(x), dim=1), dim=1)
        return float((pred == y.to(pred.dtype)).float().mean().item())
