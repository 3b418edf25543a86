"""MI355X-native prune -> k-means -> index/Huffman path of neural-network-compression.

Layout mirrors the reference package: ``common.utility`` (prune_weigth, get_quantized_weight,
get_weight_distribution), ``common.trainer`` (Trainer), ``le_net_300_100_trainer``, ``main``
(run_experiment_with_lenet300100), ``neural_networks``; plus ``ops`` / ``kmeans`` / ``pipeline``
(device-level operators) and ``csrc`` (HIP kernels behind the C ABI of include/nnc.h).
Importing the package does not need a GPU; calling into it does.
"""
__version__ = "0.1.0"
