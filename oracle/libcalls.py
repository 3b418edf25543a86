"""The reference's CPU path as the library calls it makes.  TEST / BENCH INFRASTRUCTURE ONLY.

``prune_weigth`` is the three NumPy calls of /root/reference/neural_network_compression/
common/utility.py:158-163; ``quantize`` is the scikit-learn call of utility.py:237-239
(``algorithm="full"`` is spelled "lloyd" in the installed scikit-learn 1.7.2; same algorithm).
bench.py times these on the GPU box's host cores as the ``cpu_baseline`` (kind "port": a
restatement, since the reference package itself never travels to the GPU box).
"""
from __future__ import annotations

import numpy as np


def prune_weigth(w, threshold=0.25, std_smooth=True):
    if std_smooth:
        threshold = np.std(w) * threshold
    idx = np.abs(w) < threshold
    w[idx] = 0
    return idx


def quantize(w, space, n_threads=None):
    from sklearn.cluster import KMeans
    from threadpoolctl import threadpool_limits

    space = np.asarray(space)
    km = KMeans(n_clusters=len(space), init=space.reshape(-1, 1), n_init=1, algorithm="lloyd")
    if n_threads is None:
        km.fit(w.reshape(-1, 1))
    else:
        with threadpool_limits(limits=n_threads):
            km.fit(w.reshape(-1, 1))
    ris = km.cluster_centers_[km.labels_].reshape(w.shape)
    return ris, km
