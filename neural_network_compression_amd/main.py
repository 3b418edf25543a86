"""Experiment driver with the reference's entry point
(neural_network_compression/main.py:31-93): train -> prune while training -> train with the masks
held -> report -> quantize.  MNIST is fetched by the reference over the network
(common/utility.py:59); without the files this driver falls back to a synthetic 10-class problem of
the same shape so that the whole surface still runs."""
from __future__ import annotations

import gzip
import os
import pathlib
import struct
from typing import Tuple

import numpy as np
import torch

from .common.trainer import LeNetDataset
from .le_net_300_100_trainer import LeNet300100Trainer
from .le_net_5_trainer import LeNet5Trainer


def reset_seed() -> None:
    np.random.seed(0)
    torch.manual_seed(0)


def _read_idx(path: str) -> np.ndarray:
    opener = gzip.open if path.endswith(".gz") else open
    with opener(path, "rb") as f:
        _, _, dims = struct.unpack(">HBB", f.read(4))
        shape = tuple(struct.unpack(">I", f.read(4))[0] for _ in range(dims))
        return np.frombuffer(f.read(), dtype=np.uint8).reshape(shape)


def _synthetic(n_train: int = 8192, n_test: int = 2048) -> Tuple[LeNetDataset, LeNetDataset]:
    rng = np.random.RandomState(0)
    protos = rng.rand(10, 784).astype(np.float32)

    def make(n):
        y = rng.randint(0, 10, size=n)
        x = (protos[y] + 0.35 * rng.randn(n, 784)).astype(np.float32)
        return x, y

    xtr, ytr = make(n_train)
    xte, yte = make(n_test)
    return LeNetDataset(xtr, np.eye(10, dtype=np.float32)[ytr]), LeNetDataset(xte, yte)


def get_train_and_test_dataset(folder: str = "data/mnist") -> Tuple[LeNetDataset, LeNetDataset]:
    names = ["train-images-idx3-ubyte", "train-labels-idx1-ubyte", "t10k-images-idx3-ubyte", "t10k-labels-idx1-ubyte"]
    paths = [os.path.join(folder, n) for n in names]
    if not all(os.path.exists(p) for p in paths):
        return _synthetic()
    xtr, ytr, xte, yte = (_read_idx(p) for p in paths)
    xtr = xtr.reshape(len(xtr), -1).astype(np.float32) / 255.0
    xte = xte.reshape(len(xte), -1).astype(np.float32) / 255.0
    return LeNetDataset(xtr, np.eye(10, dtype=np.float32)[ytr]), LeNetDataset(xte, yte.astype(np.int64))


def run_experiment_with_lenet300100(train_epochs: int, prune_train_epochs: int, semi_prune_train_epochs: int,
                                    maximum_centroid_bits: int, k_means_initialization_mode: str,
                                    with_cumulative_weight_distribution: bool, experiment_name: str,
                                    *, arith: str = "auto", reloc: str = "auto") -> None:
    """The reference's entry point and arguments (main.py:31-39); ``arith`` / ``reloc``: Trainer.quantize."""
    reset_seed()
    train_dataset, test_dataset = get_train_and_test_dataset()
    _run_experiment(LeNet300100Trainer(), train_dataset, test_dataset, train_epochs, prune_train_epochs,
                    semi_prune_train_epochs, maximum_centroid_bits, k_means_initialization_mode,
                    with_cumulative_weight_distribution, experiment_name, arith=arith, reloc=reloc)


def run_experiment_with_lenet5(train_epochs: int, prune_train_epochs: int, semi_prune_train_epochs: int,
                               maximum_centroid_bits: int, k_means_initialization_mode: str,
                               with_cumulative_weight_distribution: bool, experiment_name: str,
                               *, arith: str = "auto", reloc: str = "auto") -> None:
    """The same experiment on LeNet-5 (the reference's README lists it as a TODO): images as (N, 28, 28, 1)."""
    reset_seed()
    train_dataset, test_dataset = get_train_and_test_dataset()
    as_images = lambda d: LeNetDataset(np.asarray(d.input_data).reshape(-1, 28, 28, 1), d.output_data)  # noqa: E731
    _run_experiment(LeNet5Trainer(), as_images(train_dataset), as_images(test_dataset), train_epochs, prune_train_epochs,
                    semi_prune_train_epochs, maximum_centroid_bits, k_means_initialization_mode,
                    with_cumulative_weight_distribution, experiment_name, arith=arith, reloc=reloc)


def _run_experiment(trainer, train_dataset, test_dataset, train_epochs, prune_train_epochs, semi_prune_train_epochs,
                    maximum_centroid_bits, k_means_initialization_mode, with_cumulative_weight_distribution,
                    experiment_name, arith: str = "auto", reloc: str = "auto") -> None:
    report_directory = f"{trainer.model_name}_{experiment_name}"

    train_accuracies = trainer.train(train_dataset=train_dataset, test_dataset=test_dataset, epochs=train_epochs)
    pruned_train_accuracies = trainer.pruned_train(train_dataset=train_dataset, test_dataset=test_dataset,
                                                   epochs=prune_train_epochs, with_standard_deviation_smoothing=True)
    semi_pruned_train_accuracies = trainer.semi_pruned_train(train_dataset=train_dataset, test_dataset=test_dataset,
                                                             epochs=semi_prune_train_epochs)
    trainer.store_report(report_directory)
    after_quantization_accuracy = trainer.quantize(
        test_dataset=test_dataset, with_cumulative_weight_distribution=with_cumulative_weight_distribution,
        maximum_centroid_bits=maximum_centroid_bits, k_means_initialization_mode=k_means_initialization_mode,
        arith=arith, reloc=reloc)

    pathlib.Path(report_directory).mkdir(parents=True, exist_ok=True)
    with open(f"{report_directory}/accuracies.txt", "w") as f:
        f.write(f"train {train_accuracies}\npruned train {pruned_train_accuracies}\n"
                f"semi pruned train {semi_pruned_train_accuracies}\nafter quantization {after_quantization_accuracy}\n")
    try:  # the reference's accuracy plot (main.py:68-93), if matplotlib is around
        import matplotlib
        matplotlib.use("Agg")
        import matplotlib.pyplot as plt

        fig, ax = plt.subplots()
        a, b, c = train_epochs, prune_train_epochs, semi_prune_train_epochs
        ax.plot(range(a), train_accuracies, color="r", label="train")
        ax.plot(range(a, a + b), pruned_train_accuracies, color="g", label="pruned train")
        ax.plot(range(a + b, a + b + c), semi_pruned_train_accuracies, color="b", label="semi pruned train")
        plt.axhline(y=after_quantization_accuracy, color="y", linestyle="-", label="after quantization")
        plt.legend(); plt.xlabel("epoch"); plt.ylabel("accuracy")
        plt.savefig(f"{report_directory}/accuracy_plot.png")
        plt.close(fig)
    except Exception:  # pragma: no cover - plotting is optional
        pass


if __name__ == "__main__":
    run_experiment_with_lenet300100(train_epochs=2, prune_train_epochs=2, semi_prune_train_epochs=2,
                                    maximum_centroid_bits=2, k_means_initialization_mode="density",
                                    with_cumulative_weight_distribution=True,
                                    experiment_name="2BitsDensityQuantization")
