#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the reference itself.

Runs ONLY in the build container (it needs /root/reference, which never travels to
the GPU box).  It loads the reference's ``common/utility.py`` *by file path* (the
package import needs TensorFlow, which is absent) and calls its own functions

    prune_weigth            /root/reference/neural_network_compression/common/utility.py:134-163
    get_weight_distribution /root/reference/neural_network_compression/common/utility.py:334-392
    get_quantized_weight    /root/reference/neural_network_compression/common/utility.py:172-240

on inputs made by ``neural_network_compression_amd.synth`` (integer-only generator, so
the inputs are re-created bit-for-bit by the tests; only their SHA-256 is stored).

Two harness-side adaptations, neither touching a reference file:
  * the reference passes ``algorithm="full"`` to scikit-learn's KMeans (pinned 0.24);
    the installed 1.7.2 only knows the same algorithm under the name "lloyd", so the
    module's ``KMeans`` name is rebound to a factory that renames the argument and
    records the ``init`` array it was given (that is how the init "space" is captured);
  * everything runs under ``threadpool_limits(1)``: scikit-learn's Lloyd is run-to-run
    deterministic only on one thread.

Per-iteration traces and single-step known-answer tests call the same Cython routine
KMeans.fit loops over (``sklearn.cluster._k_means_lloyd.lloyd_iter_chunked_dense``).

Output: tests/golden/ref_goldens.npz (arrays) + tests/golden/ref_goldens.json (manifest).
"""
from __future__ import annotations

import hashlib
import importlib.util
import json
import os
import sys

sys.dont_write_bytecode = True
os.environ.setdefault("MPLBACKEND", "Agg")

import numpy as np  # noqa: E402
import sklearn  # noqa: E402
import sklearn.cluster  # noqa: E402
from sklearn.cluster._k_means_lloyd import lloyd_iter_chunked_dense  # noqa: E402
from threadpoolctl import threadpool_limits  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from neural_network_compression_amd import synth  # noqa: E402

REF_UTILITY = "/root/reference/neural_network_compression/common/utility.py"


def load_reference():
    spec = importlib.util.spec_from_file_location("ref_utility", REF_UTILITY)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    captured = {}

    def kmeans_factory(*args, **kwargs):
        if kwargs.get("algorithm") in ("full", "auto"):
            kwargs["algorithm"] = "lloyd"
        if "init" in kwargs and not isinstance(kwargs["init"], str):
            captured["init"] = np.array(kwargs["init"], copy=True)
        return sklearn.cluster.KMeans(*args, **kwargs)

    mod.KMeans = kmeans_factory
    return mod, captured


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def f32_bits(x) -> int:
    return int(np.array([x], dtype=np.float32).view(np.uint32)[0])


ARR = {}
MAN = {"versions": {"numpy": np.__version__, "sklearn": sklearn.__version__}, "cases": {}}


def put(name: str, a: np.ndarray) -> str:
    assert name not in ARR, name
    ARR[name] = a
    return name


# ----------------------------------------------------------------------------------
# the tensors of the five BASELINE configs at fixture-friendly sizes
# ----------------------------------------------------------------------------------
def lenet300_tensors():
    out = []
    for li, (name, wshape, bshape) in enumerate(synth.LENET_300_100):
        out.append((f"l300.{name}.w", wshape, 2000 + 2 * li))
        out.append((f"l300.{name}.b", bshape, 2000 + 2 * li + 1))
    return out


def lenet5_tensors():
    out = []
    for li, (name, wshape, bshape) in enumerate(synth.LENET_5):
        out.append((f"l5.{name}.w", wshape, 3000 + 2 * li))
        out.append((f"l5.{name}.b", bshape, 3000 + 2 * li + 1))
    return out


L300_Q = {"dense1": (1, 0.1), "dense2": (1, 0.1), "out": (0.5, 0)}  # le_net_300_100_trainer.py:21-27


def q_for(tname: str):
    net, layer, kind = tname.split(".")
    if net == "l300":
        qw, qb = L300_Q[layer]
    else:
        qw, qb = (1, 0.1)
    return qw if kind == "w" else qb


def gen_prune(ref):
    cases = {}
    for tname, shape, seed in lenet300_tensors() + lenet5_tensors():
        for q, smooth in [(1, True), (0.5, True), (0.1, True), (0, True), (0.25, False), (0.05, False)]:
            w = synth.weights(shape, seed)
            in_sha = sha(w)
            sigma = np.std(w)  # the call utility.py:159 makes
            mask = ref.prune_weigth(w, threshold=q, std_smooth=smooth)
            thr = sigma * q if smooth else q
            key = f"prune/{tname}/q{q}/{'std' if smooth else 'hard'}"
            cases[key] = {
                "shape": list(shape), "seed": seed, "q": q, "std_smooth": smooth,
                "input_sha256": in_sha,
                "sigma_bits": f32_bits(sigma), "sigma_dtype": str(np.asarray(sigma).dtype),
                "thr_dtype": str(np.asarray(thr).dtype),
                "thr_value": float(thr),
                "mask_dtype": str(mask.dtype),
                "mask_sha256": sha(np.packbits(mask.ravel())),
                "nzeroed": int(mask.sum()),
                "pruned_sha256": sha(w),
            }
    # sizes that exercise every branch of numpy's pairwise summation tree
    for n in [1, 2, 7, 8, 9, 15, 16, 17, 127, 128, 129, 130, 255, 1000, 8191, 8192, 8193, 8200,
              16383, 16384, 16385, 24577, 100003, 1 << 20, (1 << 20) + 4099]:
        w = synth.weights((n,), 7000 + (n % 997))
        in_sha = sha(w)
        sigma = np.std(w)
        mean = w.mean()
        ssum = np.sum(w)
        var = np.var(w)
        mask = ref.prune_weigth(w, threshold=1, std_smooth=True)
        cases[f"prune/flat/n{n}"] = {
            "shape": [n], "seed": 7000 + (n % 997), "q": 1, "std_smooth": True,
            "input_sha256": in_sha, "sigma_bits": f32_bits(sigma), "mean_bits": f32_bits(mean),
            "sum_bits": f32_bits(ssum), "var_bits": f32_bits(var),
            "mask_sha256": sha(np.packbits(mask.ravel())), "nzeroed": int(mask.sum()),
            "pruned_sha256": sha(w),
        }
    MAN["cases"].update(cases)


def pruned_tensor(ref, tname, shape, seed):
    w = synth.weights(shape, seed)
    ref.prune_weigth(w, threshold=q_for(tname), std_smooth=True)
    return w


def strip_zeros(w):
    """What Trainer.quantize does before the CDF (common/trainer.py:55-59)."""
    flat = w.flatten()
    (idx,) = np.nonzero(flat == 0)
    return np.delete(flat, idx, axis=0)


def gen_cdf(ref):
    for tname, shape, seed in lenet300_tensors() + lenet5_tensors():
        w = pruned_tensor(ref, tname, shape, seed)
        nz = strip_zeros(w)
        if nz.size < 2:
            continue
        xnew, cdf = ref.get_weight_distribution(nz)
        MAN["cases"][f"cdf/{tname}"] = {
            "shape": list(shape), "seed": seed, "q": q_for(tname), "n_nonzero": int(nz.size),
            "xnew": put(f"cdf/{tname}/xnew", np.asarray(xnew)),
            "cdf": put(f"cdf/{tname}/cdf", np.asarray(cdf)),
            "xnew_dtype": str(np.asarray(xnew).dtype), "cdf_dtype": str(np.asarray(cdf).dtype),
        }


def quantize_case(ref, captured, key, w, bits, mode, with_cdf, forgy_seed=None, store_labels=False):
    cdfs = None
    if with_cdf:
        nz = strip_zeros(w)
        cdfs = ref.get_weight_distribution(nz)
    if forgy_seed is not None:
        np.random.seed(forgy_seed)
    captured.pop("init", None)
    win = w.copy()
    q, km = ref.get_quantized_weight(win, bits=bits, mode=mode, cdfs=cdfs)
    entry = {"bits": bits, "mode": mode, "with_cdf": with_cdf, "forgy_seed": forgy_seed,
             "input_sha256": sha(w), "n": int(w.size)}
    if km is None:
        entry["passthrough"] = True
        assert q is win
    else:
        centers = km.cluster_centers_.ravel()
        labels = km.labels_
        entry.update({
            "passthrough": False,
            "K": int(centers.size),
            "n_iter": int(km.n_iter_),
            "init": put(f"{key}/init", captured["init"].ravel().astype(np.float32)),
            "init_dtype": str(captured["init"].dtype),
            "centers": put(f"{key}/centers", centers),
            "centers_dtype": str(centers.dtype),
            "labels_dtype": str(labels.dtype),
            "labels_sha256": sha(labels.astype(np.int32)),
            "bincount": put(f"{key}/bincount", np.bincount(labels, minlength=centers.size).astype(np.int64)),
            "quantized_sha256": sha(q),
            "quantized_dtype": str(q.dtype),
        })
        if store_labels:
            entry["labels"] = put(f"{key}/labels", labels.astype(np.uint16))
    MAN["cases"][key] = entry


def gen_quantize(ref, captured):
    # config 1: fc1, q=1 prune, density bits=2 (K=5)
    t = lenet300_tensors()
    name, shape, seed = t[0]
    w = pruned_tensor(ref, name, shape, seed)
    quantize_case(ref, captured, "quant/cfg1/l300.dense1.w/density2", w, 2, "density", True, store_labels=True)
    # config 2: LeNet-300-100, all tensors, linear bits=4  (+ the other modes / bit widths on the same tensors)
    for tname, shape, seed in t:
        w = pruned_tensor(ref, tname, shape, seed)
        small = w.size <= 30000
        for bits in (2, 4, 5):
            quantize_case(ref, captured, f"quant/cfg2/{tname}/linear{bits}", w, bits, "linear", False, store_labels=small)
            quantize_case(ref, captured, f"quant/cfg2/{tname}/density{bits}", w, bits, "density", True, store_labels=small)
            quantize_case(ref, captured, f"quant/cfg2/{tname}/forgy{bits}", w, bits, "forgy", False,
                          forgy_seed=100 + bits, store_labels=small)
    # config 3: LeNet-5 tensors, forgy bits=5
    for i, (tname, shape, seed) in enumerate(lenet5_tensors()):
        w = pruned_tensor(ref, tname, shape, seed)
        quantize_case(ref, captured, f"quant/cfg3/{tname}/forgy5", w, 5, "forgy", False, forgy_seed=300 + i,
                      store_labels=w.size <= 30000)
    # config 4 (reduced): unpruned flat vector, K=256 forgy / K=257 density
    w = synth.weights((200_000,), 4000)
    quantize_case(ref, captured, "quant/cfg4/flat200k/forgy8", w, 8, "forgy", False, forgy_seed=4)
    quantize_case(ref, captured, "quant/cfg4/flat200k/density8", w, 8, "density", True)
    # config 5 (reduced): one GPT-2-small-shaped layer (768x768), q=1 prune, linear bits=4
    w = synth.weights((768, 768), 5000)
    ref.prune_weigth(w, threshold=1, std_smooth=True)
    quantize_case(ref, captured, "quant/cfg5/attn_proj768/linear4", w, 4, "linear", False)
    # unpruned data, every mode, bits 2..6 on a mid-sized vector
    w = synth.weights((50_000,), 6000)
    for bits in (2, 3, 4, 6):
        for mode in ("linear", "density", "forgy"):
            quantize_case(ref, captured, f"quant/unpruned50k/{mode}{bits}", w, bits, mode, mode == "density",
                          forgy_seed=(600 + bits) if mode == "forgy" else None, store_labels=(bits == 4))


def lloyd_trace(X32, init32, max_iter=300, tol_rel=1e-4):
    """Re-run sklearn's _kmeans_single_lloyd loop (cluster/_kmeans.py:624-752) one
    iteration at a time with the same Cython routine, recording every iteration."""
    X = np.ascontiguousarray(X32.reshape(-1, 1))
    tol = np.mean(np.var(X, axis=0)) * tol_rel
    X_mean = X.mean(axis=0)
    X = X - X_mean
    centers = np.ascontiguousarray(init32.reshape(-1, 1).astype(np.float32) - X_mean)
    K = centers.shape[0]
    sw = np.ones(X.shape[0], dtype=np.float32)
    centers_new = np.zeros_like(centers)
    labels = np.full(X.shape[0], -1, dtype=np.int32)
    labels_old = labels.copy()
    wic = np.zeros(K, dtype=np.float32)
    shift = np.zeros(K, dtype=np.float32)
    trace_c, trace_cnt, trace_shift, trace_lsha = [], [], [], []
    strict = False
    for i in range(max_iter):
        lloyd_iter_chunked_dense(X, sw, centers, centers_new, wic, labels, shift, 1)
        centers, centers_new = centers_new, centers
        trace_c.append(centers.ravel().copy())
        trace_cnt.append(np.bincount(labels, minlength=K).astype(np.int64))
        tot = (shift ** 2).sum()
        trace_shift.append(np.float32(tot))
        trace_lsha.append(sha(labels))
        if np.array_equal(labels, labels_old):
            strict = True
            break
        elif tot <= tol:
            break
        labels_old[:] = labels
    if not strict:
        lloyd_iter_chunked_dense(X, sw, centers, centers, wic, labels, shift, 1, update_centers=False)
    return {
        "tol": np.float32(tol), "x_mean": np.float32(X_mean[0]), "n_iter": i + 1, "strict": strict,
        "centers_centred": np.array(trace_c, dtype=np.float32),
        "counts": np.array(trace_cnt), "shift_tot": np.array(trace_shift, dtype=np.float32),
        "labels_sha": trace_lsha, "final_centers": (centers + X_mean).ravel().astype(np.float32),
        "final_labels": labels.copy(),
    }


def gen_trace(ref, captured):
    name, shape, seed = lenet300_tensors()[0]
    w = pruned_tensor(ref, name, shape, seed)
    for key, bits, mode, with_cdf in [("cfg1.density2", 2, "density", True), ("cfg2.linear4", 4, "linear", False)]:
        cdfs = ref.get_weight_distribution(strip_zeros(w)) if with_cdf else None
        captured.pop("init", None)
        q, km = ref.get_quantized_weight(w.copy(), bits=bits, mode=mode, cdfs=cdfs)
        init = captured["init"].ravel().astype(np.float32)
        tr = lloyd_trace(w.ravel(), init)
        # the step-by-step replay must land exactly where KMeans.fit landed
        assert tr["n_iter"] == km.n_iter_, (tr["n_iter"], km.n_iter_)
        assert np.array_equal(tr["final_centers"], km.cluster_centers_.ravel())
        assert np.array_equal(tr["final_labels"], km.labels_)
        k = f"trace/{key}"
        MAN["cases"][k] = {
            "tensor": name, "bits": bits, "mode": mode, "n_iter": tr["n_iter"], "strict": bool(tr["strict"]),
            "tol_bits": f32_bits(tr["tol"]), "x_mean_bits": f32_bits(tr["x_mean"]),
            "init": put(f"{k}/init", init),
            "centers_centred": put(f"{k}/centers_centred", tr["centers_centred"]),
            "counts": put(f"{k}/counts", tr["counts"]),
            "shift_tot": put(f"{k}/shift_tot", tr["shift_tot"]),
            "labels_sha": tr["labels_sha"],
        }


def estep(X32, C32):
    """One E-step on already-centred data/centres (update_centers=False)."""
    X = np.ascontiguousarray(X32.reshape(-1, 1).astype(np.float32))
    C = np.ascontiguousarray(C32.reshape(-1, 1).astype(np.float32))
    labels = np.full(X.shape[0], -1, dtype=np.int32)
    sw = np.ones(X.shape[0], dtype=np.float32)
    wic = np.zeros(C.shape[0], dtype=np.float32)
    shift = np.zeros(C.shape[0], dtype=np.float32)
    lloyd_iter_chunked_dense(X, sw, C, C, wic, labels, shift, 1, update_centers=False)
    return labels


def gen_estep_kats():
    rng = np.random.RandomState(12345)
    kats = {}
    # (a) random data vs K=256 sorted/unsorted centres
    x = synth.weights((20000,), 8001)
    c = np.sort(synth.weights((256,), 8002, scale=0.06))
    kats["random256_sorted"] = (x, c)
    c2 = c.copy(); rng.shuffle(c2)
    kats["random256_shuffled"] = (x, c2)
    # (b) adversarial: samples within a few ulps of every midpoint of adjacent centres
    cs = np.sort(synth.weights((64,), 8003, scale=0.05))
    mids = ((cs[:-1].astype(np.float64) + cs[1:].astype(np.float64)) / 2).astype(np.float32)
    xs = []
    for d in range(-6, 7):
        v = mids.copy()
        for _ in range(abs(d)):
            v = np.nextafter(v, np.float32(np.inf if d > 0 else -np.inf), dtype=np.float32)
        xs.append(v)
    kats["midpoint_ulps64"] = (np.concatenate(xs), cs)
    # (c) duplicate centres (forgy draws with replacement) and near-duplicates 1 ulp apart
    cd = np.array([0.0, 0.01, 0.01, -0.02, 0.0, 0.03, np.nextafter(np.float32(0.03), np.float32(1)), -0.02],
                  dtype=np.float32)
    kats["duplicates"] = (synth.weights((5000,), 8004, scale=0.02), cd)
    # (d) many exact zeros (pruned tensor) against centres straddling zero
    xz = synth.weights((30000,), 8005)
    xz[np.abs(xz) < 0.05] = 0
    cz = np.linspace(xz.min(), xz.max(), 16).astype(np.float32)
    kats["pruned_linear16"] = (xz, cz)
    # (e) big dynamic range: a few huge outliers, tiny bulk
    xo = synth.weights((10000,), 8006, scale=1e-3)
    xo[::1000] *= 1e4
    co = np.linspace(xo.min(), xo.max(), 32).astype(np.float32)
    kats["outliers32"] = (xo, co)
    # (f) K=257 (density at 8 bits) with clustered centres
    c257 = np.sort(np.concatenate([synth.weights((200,), 8007, scale=0.01), synth.weights((57,), 8008, scale=0.2)]))
    kats["k257"] = (synth.weights((20000,), 8009, scale=0.08), c257.astype(np.float32))
    for name, (x, c) in kats.items():
        labels = estep(x, c)
        MAN["cases"][f"estep/{name}"] = {
            "x": put(f"estep/{name}/x", x.astype(np.float32)),
            "c": put(f"estep/{name}/c", c.astype(np.float32)),
            "labels": put(f"estep/{name}/labels", labels.astype(np.uint16)),
        }


def gen_step_kats():
    """Full single iterations (E+M, relocation, averaging, shift) on centred inputs."""
    cases = {}
    x = synth.weights((3000,), 9001)
    cases["plain16"] = (x, np.linspace(x.min(), x.max(), 16).astype(np.float32))
    # one empty cluster (duplicate centre): relocation with n_empty = 1
    c = np.linspace(x.min(), x.max(), 8).astype(np.float32)
    c[5] = c[2]
    cases["one_empty"] = (x, c)
    # several empty clusters: order decided by numpy's argpartition
    c = np.linspace(x.min(), x.max(), 12).astype(np.float32)
    c[7] = c[1]; c[9] = c[1]; c[10] = c[3]
    cases["three_empty"] = (x, c)
    # all samples identical: relocation bails out (max distance 0) and empty centres copy the biggest cluster
    xe = np.full(50, 0.125, dtype=np.float32)
    cases["all_equal"] = (xe, np.array([0.125, 0.5, -0.5, 0.125], dtype=np.float32))
    # pruned tensor, linear init: far-away centres in the gap go empty
    xz = synth.weights((4000,), 9002)
    xz[np.abs(xz) < 0.06] = 0
    cases["pruned_gap"] = (xz, np.linspace(xz.min(), xz.max(), 16).astype(np.float32))
    for name, (x, c) in cases.items():
        X = np.ascontiguousarray(x.reshape(-1, 1))
        C = np.ascontiguousarray(c.reshape(-1, 1))
        K = C.shape[0]
        Cn = np.zeros_like(C)
        labels = np.full(X.shape[0], -1, dtype=np.int32)
        sw = np.ones(X.shape[0], dtype=np.float32)
        wic = np.zeros(K, dtype=np.float32)
        shift = np.zeros(K, dtype=np.float32)
        lloyd_iter_chunked_dense(X, sw, C, Cn, wic, labels, shift, 1)
        MAN["cases"][f"step/{name}"] = {
            "x": put(f"step/{name}/x", x), "c": put(f"step/{name}/c", c),
            "labels": put(f"step/{name}/labels", labels.astype(np.uint16)),
            "centers_new": put(f"step/{name}/centers_new", Cn.ravel().copy()),
            "weight_in_clusters": put(f"step/{name}/wic", wic.copy()),
            "shift": put(f"step/{name}/shift", shift.copy()),
            "n_empty": int((np.bincount(labels, minlength=K) == 0).sum()),
        }


def main():
    ref, captured = load_reference()
    with threadpool_limits(1):
        gen_prune(ref)
        gen_cdf(ref)
        gen_quantize(ref, captured)
        gen_trace(ref, captured)
        gen_estep_kats()
        gen_step_kats()
    np.savez_compressed(os.path.join(HERE, "ref_goldens.npz"), **ARR)
    with open(os.path.join(HERE, "ref_goldens.json"), "w") as f:
        json.dump(MAN, f, indent=1, sort_keys=True)
    print(f"{len(MAN['cases'])} cases, {len(ARR)} arrays,",
          os.path.getsize(os.path.join(HERE, 'ref_goldens.npz')) // 1024, "KiB")


if __name__ == "__main__":
    main()
