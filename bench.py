#!/usr/bin/env python3
"""Headline benchmark: weights/s through prune + k-means (K = 256 class) on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Workload (BASELINE.json configs[3], "Synthetic 25 M-param fp32 weight vector, K=256"): every
GPU holds a 25 M-element float32 shard (weak scaling; --scaling strong shards ONE 25 M vector) of one synthetic weight vector
(neural_network_compression_amd.synth, seed 4000, 0.05 * bell-shaped).  One step = the whole
per-layer Deep-Compression pass on the data already resident in HBM:

    prune_weigth(q = 1 sigma)  ->  get_weight_distribution of the non-zeros  ->
    get_quantized_weight(bits = 8, mode = "density")  (K = 2^8 + 1 = 257 centroids: the
    reference's density init makes 2^bits + 1, utility.py:212), Lloyd to convergence ->
    labels + quantized values -> index histogram -> Huffman code lengths.

value = (weights processed by all ranks) / (time of the slowest rank) over exactly K steps.
prune_weigth works in place, so a step spends its input: every step (set-up, warm-up, timed) takes a batch of its own -- a copy of
the same synthetic vector -- that is resident in HBM before the timed region starts (config.input; --input-pool-gb bounds the pool,
beyond it the steps copy one resident vector at their head as they did until round 4).
The JSON line also carries
  roofline     : the k-means assignment pass over the vector, k_assign<labels> (4 B read + centroid index + 4 B decoded
                 value written per weight), against the 8 TB/s HBM peak; its duration is measured with HIP events
                 around the launch inside the timed region (in-library, on the launching stream);
  kernels      : the same for every data-touching kernel of the step (durations, launches per step, and -- for passes
                 over the vector -- algorithmic bytes and achieved GB/s);
  roofline_streaming_accumulate : the streaming form of the Lloyd pass (4 B read per weight per launch), which the
                 iterations on a sorted vector no longer need, timed on its own after the timed region;
  cpu_baseline : the same pipeline through NumPy / scikit-learn on the host cores (rank 0, N = 1
                 only) on a bounded sample of the same vector;
  parity       : the GPU fit of the timed region next to that CPU fit (n_iter, centres, index histogram, differing indices, the
                 k-means objective of both) and next to the committed full-size golden made by the reference itself
                 (tests/golden/ref_goldens_25m.json, read as data: device == oracle mode B bit for bit; the reference's n_iter).

    python bench.py --config 4 [--gpus N] ...

BASELINE.json configs[4] instead: the 122 tensors of a GPT-2-small-sized model (124.4 M weights), per tensor prune -> 4-bit
linear-init k-means (K = 16) -> labels + values -> index histogram -> Huffman lengths; the tensors are dealt out to the GPUs
(pipeline.compress_layers(group=...): replicas, no data-path collective; --shard-above shards the longest ones instead).  The total
work is fixed ("scaling": "strong").  The driver's default invocation (no --config) stays configs[3], the headline.
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
N_PER_GPU = 25_000_000
SEED = 4000


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", type=int, choices=(3, 4), default=3,
                    help="BASELINE.json configs[i]: 3 = the 25 M vector at K = 256 (the headline, default); 4 = the 122 tensors of a GPT-2-small-sized "
                         "model (124 M weights), K = 16 per layer, full prune -> quantize -> Huffman pipeline, the tensors dealt out to the GPUs")
    ap.add_argument("--shard-above", type=int, default=0, help="--config 4: tensors of this many weights or more are sharded over all GPUs (the sharded "
                                                                "fit) instead of dealt out whole; 0 = deal out everything")
    ap.add_argument("--workers", type=int, default=8, help="--config 4: tensors side by side per GPU (host threads, a stream each)")
    ap.add_argument("--n", type=int, default=N_PER_GPU, help="weights per GPU")
    ap.add_argument("--bits", type=int, default=8)
    ap.add_argument("--mode", default="density")
    ap.add_argument("--q", type=float, default=1.0)
    ap.add_argument("--cpu-sample", type=int, default=25_000_000, help="weights of the CPU baseline leg (default: the whole 25 M vector, about 20 s)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak: --n weights per GPU (the driver's scaling run); strong: --n weights in total, sharded over the GPUs "
                         "(BASELINE configs[3] read literally: one 25 M vector across 8 GPUs)")
    ap.add_argument("--python-exchange", action="store_true", help="N > 1: issue the per-iteration all-reduce from torch.distributed "
                                                                   "instead of inside the C library (nnc_kmeans_iterate_sharded)")
    ap.add_argument("--dump-durations", action="store_true", help="stderr: the per-launch durations (us) of the iteration kernels of the last step")
    ap.add_argument("--no-streaming-leg", action="store_true", help="skip the separate timing of the streaming Lloyd pass")
    ap.add_argument("--input-pool-gb", type=float, default=64.0,
                    help="configs[3]: room for the per-step input batches (100 MB each at 25 M weights); beyond it the steps copy one resident vector instead")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--step-by-step", action="store_true", help="drive the layer's steps from Python (pipeline.compress_layer(native=False)) instead of "
                                                                "the one-call form (nnc_compress_layer_f32): for comparison")
    ap.add_argument("--two-launch", action="store_true", help="Lloyd iterations launch by launch (k_bounds + k_finalize) instead of inside one resident "
                                                              "workgroup (k_lloyd): for comparison")
    ap.add_argument("--loop", action="store_true", help="Lloyd iterations inside the resident workgroup whatever the number of centres (the library "
                                                        "picks it by itself up to NNC_KM_LOOP_KMAX centres): for comparison")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to rehearse "
                                                      "the multi-rank path with all ranks on one GPU: --one-device)")
    ap.add_argument("--one-device", action="store_true", help="rehearsal: every rank uses cuda:0")
    return ap.parse_args()


def cpu_baseline(args, w_full: np.ndarray):
    """The reference's CPU path (NumPy + scikit-learn calls) on the first `cpu_sample` weights."""
    from oracle import libcalls
    from oracle import oracle as orc

    cores = len(os.sched_getaffinity(0))
    threads = min(cores, 16)  # the 1-GPU box's CPU share
    n = min(args.cpu_sample, w_full.size)
    w = w_full[:n].copy()
    try:
        import sklearn  # noqa: F401
        have_sklearn = True
    except Exception:
        have_sklearn = False
    t0 = time.perf_counter()
    mask_cpu = libcalls.prune_weigth(w, args.q, True)
    flat = w.ravel()
    cdfs = orc.get_weight_distribution(flat[flat != 0]) if args.mode == "density" else None
    space = orc.init_space(w, args.bits, args.mode, cdfs)
    if have_sklearn:
        _, km = libcalls.quantize(w, np.asarray(space, dtype=np.float32), n_threads=threads)
        n_iter = int(km.n_iter_)
        impl = f"numpy {np.__version__} + scikit-learn {sklearn.__version__} KMeans(lloyd), {threads} threads"
    else:
        km = orc.kmeans_lloyd(w.ravel(), space, accum="A")
        n_iter = km.n_iter_
        threads = 1
        impl = "C oracle (oracle/nnc_oracle.c), 1 thread"
    dt = time.perf_counter() - t0
    return {
        "value": n / dt, "unit": "weights/s", "cores": threads, "kind": "port",
        "sample": f"{'the whole vector' if n == w_full.size else f'prefix: first {n} weights of the same vector'}, same pipeline (prune q={args.q} sigma, CDF, "
                  f"{args.mode} init, bits={args.bits}, Lloyd to convergence: {n_iter} iterations) in {dt:.2f} s; {impl}",
    }, km, mask_cpu, threads


def parity_record(args, res, km_cpu, mask_cpu, n_cpu, threads, w_full=None):
    """The GPU step's result next to the CPU leg's (scikit-learn, float32 running sums, `threads` threads: the reference's arithmetic,
    run-to-run reproducible only on one thread) and next to the committed full-size golden record made by the reference itself on one
    thread (tests/golden/ref_goldens_25m.json, read as data)."""
    import hashlib

    m = res.model
    cg = m.cluster_centers_.ravel().astype(np.float64)
    out = {"n_iter_gpu": int(m.n_iter_)}
    lab_g = m.labels_
    hist_g = np.bincount(lab_g, minlength=cg.size)
    if km_cpu is not None and n_cpu == lab_g.size:
        cc = np.asarray(km_cpu.cluster_centers_, dtype=np.float64).ravel()
        lab_c = np.asarray(km_cpu.labels_)
        out.update({
            "n_iter_cpu": int(km_cpu.n_iter_), "cpu_threads": threads,
            "mask_equal": bool(np.array_equal(res.mask.cpu().numpy().astype(bool).ravel(), np.asarray(mask_cpu).ravel())) if res.mask is not None else None,
            "max_rel_centre_err": float(np.max(np.abs(cg - cc) / np.maximum(np.abs(cc), 1e-30))),
            "max_abs_centre_err": float(np.max(np.abs(cg - cc))),
            "max_abs_sorted_centre_err": float(np.max(np.abs(np.sort(cg) - np.sort(cc)))),
            "hist_l1": int(np.abs(hist_g - np.bincount(lab_c, minlength=cg.size)).sum()),
            "labels_differing": int(np.count_nonzero(lab_g != lab_c)),
        })
        if w_full is not None:
            # index-by-index numbers mean little once two trajectories have parted (another local optimum): the k-means objective, float64
            wp = np.where(np.asarray(mask_cpu).ravel(), 0.0, w_full.astype(np.float64))
            out["objective_f64"] = {"gpu": float(((wp - cg[lab_g]) ** 2).sum()), "cpu": float(((wp - cc[lab_c]) ** 2).sum())}
    gp = os.path.join(ROOT, "tests", "golden", "ref_goldens_25m.json")
    if os.path.exists(gp) and args.n == N_PER_GPU and (args.bits, args.mode, args.q) == (8, "density", 1.0):
        g = json.load(open(gp))
        lab_sha = hashlib.sha256(lab_g.astype(np.int32).tobytes()).hexdigest()
        cbits = [int(v) for v in m.cluster_centers_.ravel().view(np.uint32)]
        gold = {"source": "tests/golden/ref_goldens_25m.json"}
        if "oracle_B" in g:
            b = g["oracle_B"]
            gold["mode_b_bit_exact"] = bool(b["n_iter"] == m.n_iter_ and b["centers_bits"] == cbits and b["labels_sha256_int32"] == lab_sha)
            gold["mode_b_vs_reference"] = b.get("vs_reference")
        r = g["reference"]
        cr = np.array(r["centers_bits"], dtype=np.uint32).view(np.float32).astype(np.float64)
        gold.update({"n_iter_reference_one_thread": r["n_iter"], "reference_bit_exact": bool(r["n_iter"] == m.n_iter_ and r["centers_bits"] == cbits and r["labels_sha256_int32"] == lab_sha),
                     "max_abs_sorted_centre_err_vs_reference": float(np.max(np.abs(np.sort(cg) - np.sort(cr)))),
                     "objective_f64": {"reference": r.get("quality", {}).get("inertia_f64"), "mode_b": g.get("oracle_B", {}).get("quality", {}).get("inertia_f64")},
                     "note": "the reference at this size (float32 running sums over up to 17 M members, float32 member counts that stop at 2^24) takes 49 iterations on one "
                             "thread, 44-45 on 16 (not reproducible); the exact-sum fit takes 43 and ends in another local optimum with a lower objective; "
                             "arith='reference' reproduces the one-thread reference bit for bit (tests/test_gpu_parity.py)"})
        out["golden"] = gold
    return out


def _init_ranks(args):
    """(torch, rank, world, device, group, comm, comm_note): one process per GPU, RCCL through torch.distributed and -- for the data-path
    exchange of sharded fits -- the library's own communicator."""
    import torch

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = 0 if args.one_device else int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and rank == 0 and world == 1 and args.gpus > 1:
        print(f"bench.py: --gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks", file=sys.stderr)
        sys.exit(2)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    group = comm = comm_note = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=args.backend)
        group = dist.group.WORLD
    return torch, rank, world, dev, group, comm, comm_note


def _comm_preflight(torch, comm, rank, world, dev):
    """The library's communicator is believed only after it has added up what torch.distributed adds up: rank r contributes r + 1 in
    every word of a block the size of an iteration's exchange (2 K int64).  Returns (comm or None, note)."""
    if comm is None:
        return None, None
    try:
        probe = torch.full((2 * 1040,), rank + 1, dtype=torch.int64, device=dev)
        comm.allreduce_(probe)
        torch.cuda.synchronize(dev)
        if not bool((probe == world * (world + 1) // 2).all().item()):
            raise RuntimeError("the library's all-reduce returned a wrong sum")
        return comm, None
    except Exception as e:  # noqa: BLE001 - reported in the JSON line and on stderr
        note = f"pre-flight all-reduce: {type(e).__name__}: {e}"
        print(f"[bench] rank {rank}: {note}; falling back to torch.distributed", file=sys.stderr)
        try:
            comm.close()
        except Exception:  # noqa: BLE001
            pass
        return None, note


def _comm_checked(torch, dist, group, comm, comm_note, rank, world, dev):
    """Every rank has the communicator (or none has): try it, then agree on the verdict -- one rank's doubt drops it everywhere."""
    if comm is None:
        return None, comm_note
    comm, note = _comm_preflight(torch, comm, rank, world, dev)
    ok = torch.tensor([1 if comm is not None else 0], dtype=torch.int32, device=dev)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
    if int(ok.item()) == 0 and comm is not None:
        comm.close()
        comm = None
        note = note or "another rank's pre-flight all-reduce failed"
    return comm, (comm_note or note)


def cpu_baseline_config4(args, layers):
    """The reference's CPU path on a stated subset of configs[4]: the ten tensors of transformer block 0 (7.1 M of the 124.4 M
    weights: the four matrix shapes and the six 1-D tensors every block repeats), same seeds as the GPU run."""
    from oracle import libcalls
    from oracle import oracle as orc
    from neural_network_compression_amd import synth

    threads = min(len(os.sched_getaffinity(0)), 16)
    sub = [(i, name, shape) for i, (name, shape) in enumerate(layers) if name.startswith("h0.")]
    ws = [synth.weights(shape, 5000 + i) for i, _, shape in sub]
    n = sum(w.size for w in ws)
    iters = 0
    t0 = time.perf_counter()
    for w in ws:
        libcalls.prune_weigth(w, args.q, True)
        space = orc.init_space(w, 4, "linear")
        _, km = libcalls.quantize(w, np.asarray(space, dtype=np.float32), n_threads=threads)
        orc.huffman_lengths(np.bincount(km.labels_, minlength=16))
        iters += int(km.n_iter_)
    dt = time.perf_counter() - t0
    import sklearn

    return {"value": n / dt, "unit": "weights/s", "cores": threads, "kind": "port",
            "sample": f"block 0 of the layer list ({len(ws)} of 122 tensors, {n} of 124419840 weights): prune q={args.q} sigma, linear init, bits=4, Lloyd to "
                      f"convergence ({iters} iterations), index histogram, Huffman lengths, one tensor after the other as Trainer.quantize does, in {dt:.2f} s; "
                      f"numpy {np.__version__} + scikit-learn {sklearn.__version__} KMeans(lloyd), {threads} threads"}


def main_config4(args):
    """BASELINE configs[4]: synthetic 124 M-parameter layer list (GPT-2-small-sized), K = 16 per layer, full prune -> quantize -> Huffman
    pipeline; the tensors are dealt out to the GPUs (pipeline.compress_layers(group=...): replicas, no data-path collective; --shard-above
    adds the sharded fit for the longest ones).  The total work is fixed, so this is a STRONG-scaling line."""
    torch, rank, world, dev, group, comm, comm_note = _init_ranks(args)
    from neural_network_compression_amd import _native as nat
    from neural_network_compression_amd import pipeline, sharding, synth

    L = nat.load()
    layers = synth.gpt2_small_layers()
    sizes = [int(np.prod(shape)) for _, shape in layers]
    n_total = sum(sizes)
    shard_above = args.shard_above if args.shard_above > 0 else None
    owner = pipeline.partition_layers(sizes, world, shard_above)
    if group is not None and args.backend == "nccl" and any(o < 0 for o in owner) and not args.python_exchange:
        import torch.distributed as dist

        try:
            comm = sharding.RcclComm(group, dev)
        except Exception as e:  # noqa: BLE001
            comm_note = f"{type(e).__name__}: {e}"
        ok = torch.tensor([1 if comm is not None else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
        if int(ok.item()) == 0 and comm is not None:
            comm.close()
            comm = None
        comm, comm_note = _comm_checked(torch, dist, group, comm, comm_note, rank, world, dev)
    held = []
    for i, (name, shape) in enumerate(layers):
        if owner[i] == rank:
            held.append(torch.from_numpy(synth.weights(shape, 5000 + i)).to(dev).reshape(-1))
        elif owner[i] < 0:
            lo, hi = sharding.shard_bounds(sizes[i], world, rank)
            held.append(torch.from_numpy(synth.weights((hi - lo,), 5000 + i, start=lo)).to(dev))
        else:
            held.append(None)
    kw = dict(q=args.q, bits=4, mode="linear", huffman=True, want_values=True)

    def step(workers=args.workers):
        mine = [None if t is None else t.clone() for t in held]     # prune works in place; the copies are inside the timed region
        if group is None:
            return pipeline.compress_layers(mine, workers=workers, **kw)
        return pipeline.compress_layers(mine, workers=workers, group=group, sizes=sizes, shard_above=shard_above, comm=comm, **kw)

    def barrier():
        if group is not None:
            import torch.distributed as dist

            dist.barrier(group=group)
        torch.cuda.synchronize(dev)

    res = step()
    for _ in range(args.warmup):
        res = step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
    barrier()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if group is not None:
        import torch.distributed as dist

        dist.all_reduce(tmax, op=dist.ReduceOp.MAX, group=group)
    dt = float(tmax.item())
    # one more step, one tensor after the other, with HIP events around the assignment pass of every tensor (outside the timed region)
    nat.check(L.nnc_profile_tags(1 << 2))
    nat.check(L.nnc_profile_begin(256))
    step(workers=1)
    torch.cuda.synchronize(dev)
    ms_buf, tag_buf, cnt = (ctypes.c_float * 256)(), (ctypes.c_int32 * 256)(), ctypes.c_int64(0)
    nat.check(L.nnc_profile_end(ms_buf, tag_buf, 256, ctypes.byref(cnt)))
    nat.check(L.nnc_profile_tags(0xFFFFFFFF))
    lab_ms = np.array(ms_buf[:min(cnt.value, 256)], dtype=np.float64)
    if rank == 0:
        if group is None:
            n_iter = sum(r.model.n_iter_ for r in res if r.model is not None)
            bits = sum(int(r.total_bits or 0) for r in res)
            passed = sum(1 for r in res if r.model is None)
        else:
            n_iter = sum(r.n_iter for r in res)
            bits = sum(int(r.total_bits or 0) for r in res)
            passed = sum(1 for r in res if r.centers is None)
        per_rank = [sum(1 for o in owner if o == r) for r in range(world)]
        mine_long = [sizes[i] for i, o in enumerate(owner) if o in (0, -1) and sizes[i] > 4096]
        lab_bytes = sum((4 + 1 + 4) * (n if owner[i] >= 0 else -(-n // world)) for i, n in enumerate(sizes) if owner[i] in (0, -1) and n > 4096)
        roof = {"bound": "hbm", "kernel": "k_assign<labels> over rank 0's tensors of more than 4096 weights (final E-step: 4 B read + 1 B centroid index + 4 B decoded "
                                         "value per weight), one tensor after the other in one extra step",
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "traffic": None, "launches_timed": int(lab_ms.size), "achieved": None, "frac": None}
        if lab_ms.size == len(mine_long) and lab_ms.size:
            roof["achieved"] = lab_bytes / (lab_ms.sum() * 1e-3) / 1e9
            roof["frac"] = roof["achieved"] / HBM_PEAK_GBS
            roof["algorithmic_bytes"] = int(lab_bytes)
            roof["kernel_ms_total"] = float(lab_ms.sum())
        # the bytes no implementation could avoid (see the configs[3] line): per weight 8+8+9+8 + 4+1+4, per surviving weight 12; short tensors as one read + write
        step_bytes = n_total * (8 + 8 + 9 + 8 + 4 + 1 + 4) + int(0.32 * n_total) * 12
        out = {
            "metric": "weights/sec through prune+k-means (K=16 per layer) + Huffman, whole model",
            "value": n_total * args.steps / dt, "unit": "weights/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": f"configs[4]: synthetic GPT-2-small-sized layer list, 122 tensors, {n_total} fp32 weights in total (the same total at any number of GPUs), "
                            f"per tensor prune q={args.q} sigma -> linear-init k-means bits=4 (K=16) to convergence -> labels+values -> index histogram -> Huffman lengths",
                "tensors": len(sizes), "weights_total": n_total, "lloyd_iterations_total": int(n_iter), "huffman_bits_per_weight": bits / n_total,
                "passed_through": passed,
                "parallelism": (f"layers dealt out to {world} ranks (longest first), {args.workers} streams per rank" if world > 1 else f"single GPU, {args.workers} streams")
                               + (f"; tensors >= {shard_above} weights sharded over all ranks" if shard_above and world > 1 else ""),
                "tensors_per_rank": per_rank, "tensors_sharded": sum(1 for o in owner if o < 0),
                "exchange": None if not any(o < 0 for o in owner) else ("rccl-in-library" if comm is not None else "torch.distributed"),
                "rccl_world": (int(L.nnc_comm_world(comm.handle)) if comm is not None else None),
                "backend": (args.backend if world > 1 else None),
            },
            "roofline": roof,
            "roofline_step": {"bound": "hbm", "unavoidable_bytes_per_step": int(step_bytes), "achieved": step_bytes / (dt / args.steps) / 1e9, "peak": HBM_PEAK_GBS * world,
                              "unit": "GB/s", "frac": step_bytes / (dt / args.steps) / 1e9 / (HBM_PEAK_GBS * world),
                              "note": "122 chains of short dependent launches: latency, not bandwidth, bounds this configuration"},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline_config4(args, layers)
        print(json.dumps(out))
    if comm is not None:
        comm.close()
    if group is not None:
        import torch.distributed as dist

        dist.destroy_process_group()


def main():
    args = parse()
    if args.config == 4:
        return main_config4(args)
    import torch

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0 and world == 1 and args.gpus > 1:
            print(f"bench.py: --gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks", file=sys.stderr)
            sys.exit(2)
    if args.one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    group = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=args.backend)
        group = dist.group.WORLD

    from neural_network_compression_amd import _native as nat
    from neural_network_compression_amd import pipeline, sharding, synth

    L = nat.load()
    n_total = args.n * world if args.scaling == "weak" else args.n
    lo, hi = sharding.shard_bounds(n_total, world, rank)
    comm = None
    comm_note = None
    if group is not None and args.backend == "nccl" and not args.python_exchange:
        # the library's own RCCL communicator: the exchange is enqueued from C.  Every rank must end up on the same path, so the
        # ranks agree on whether it came up; if it did not anywhere, the exchange goes through torch.distributed and the line says so.
        import torch.distributed as dist

        try:
            comm = sharding.RcclComm(group, dev)
        except Exception as e:  # noqa: BLE001 - reported in the JSON line and on stderr
            comm_note = f"{type(e).__name__}: {e}"
            print(f"[bench] rank {rank}: library communicator unavailable ({comm_note}); falling back to torch.distributed", file=sys.stderr)
        ok = torch.tensor([1 if comm is not None else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
        if int(ok.item()) == 0 and comm is not None:
            comm.close()
            comm = None
            comm_note = comm_note or "another rank could not create it"
        comm, comm_note = _comm_checked(torch, dist, group, comm, comm_note, rank, world, dev)
    w_host = synth.weights((hi - lo,), SEED, start=lo)
    w0 = torch.from_numpy(w_host).to(dev)

    # Every step gets a batch of its own, resident in HBM before the timed region starts (prune_weigth works in place, so a batch
    # is spent once it has been through a step): 1 set-up + W warm-up + K timed + 1 step for the K-sized kernels' events, 100 MB
    # each.  Until round 4 a device-to-device copy of the one resident vector sat at the head of every timed step (2 x 22 us of copy
    # kernels that are not on the path; it also left the step's input warm in the Infinity Cache, which a layer coming from
    # elsewhere is not).  Beyond `--input-pool-gb` the old form is used and the line says so (config.input).
    n_batches = 1 + args.warmup + args.steps + 1
    pooled = n_batches * w0.numel() * 4 <= args.input_pool_gb * (1 << 30)
    pool = [w0.clone() for _ in range(n_batches)] if pooled else []
    torch.cuda.synchronize(dev)

    def step():
        x = pool.pop() if pool else w0.clone()
        return pipeline.compress_layer(x, q=args.q, bits=args.bits, mode=args.mode, group=group, comm=comm,
                                       huffman=True, want_values=True, native=not args.step_by_step, two_launch=args.two_launch, loop=args.loop)

    def barrier():
        if group is not None:
            import torch.distributed as dist

            dist.barrier(group=group)
        torch.cuda.synchronize(dev)

    res = step()  # set-up, not a warm-up step: first use loads the code objects, sizes the allocator pools, pins the host buffers
    for _ in range(args.warmup):
        res = step()
    def profiled(nsteps, mask):
        """nsteps steps with HIP events around the launches whose tag is in `mask`; (seconds, durations ms, tags)"""
        cap = 1000 * max(1, nsteps)
        nat.check(L.nnc_profile_tags(mask))
        nat.check(L.nnc_profile_begin(cap))
        barrier()
        t0 = time.perf_counter()
        r = None
        for _ in range(nsteps):
            r = step()
        barrier()
        dt = time.perf_counter() - t0
        ms_buf = (ctypes.c_float * cap)()
        tag_buf = (ctypes.c_int32 * cap)()
        cnt = ctypes.c_int64(0)
        nat.check(L.nnc_profile_end(ms_buf, tag_buf, cap, ctypes.byref(cnt)))
        nat.check(L.nnc_profile_tags(0xFFFFFFFF))
        m = min(cnt.value, cap)
        return dt, np.array(ms_buf[:m], dtype=np.float64), np.array(tag_buf[:m], dtype=np.int64), r

    # THE timed region: exactly K steps; events only around the passes over the vector (ten launches a step).  The K-sized
    # kernels of the Lloyd loop (some 110 launches a step) are timed in one more step afterwards, outside the timed region.
    STREAM_TAGS = (1 << 0) | (1 << 2) | (1 << 3) | (1 << 4) | (1 << 6) | (1 << 7)
    dt, all_ms, all_tags, res = profiled(args.steps, STREAM_TAGS)
    _, ms_k, tags_k, _ = profiled(1, (1 << 1) | (1 << 5) | (1 << 8) | (1 << 9))
    per_step_scale = {1: 1.0, 5: 1.0, 8: 1.0, 9: 1.0}

    # The streaming form of the Lloyd pass (k_assign<accumulate>: what an iteration costs on a vector that is NOT sorted, and
    # the fallback of the rank-boundary form) timed on its own, outside the timed region: 30 launches on the same sorted
    # 25 M vector with the converged centres.
    stream_ms = np.zeros(0)
    if rank == 0 and res.model is not None and not args.no_streaming_leg:
        from neural_network_compression_amd import kmeans as _km

        xq = w0.clone()
        pipeline.prune_sharded_(xq, args.q, True, None)
        km2 = _km.DeviceKMeans(xq, res.model.cluster_centers_.ravel(), rank_boundaries=False)
        for _ in range(3):
            nat.check(L.nnc_kmeans_accumulate(km2.x_iter.data_ptr(), km2.ws.data_ptr(), ctypes.byref(km2.p), km2.stream))
        nat.check(L.nnc_profile_begin(64))
        for _ in range(30):
            nat.check(L.nnc_kmeans_accumulate(km2.x_iter.data_ptr(), km2.ws.data_ptr(), ctypes.byref(km2.p), km2.stream))
        torch.cuda.synchronize(dev)
        ms2 = (ctypes.c_float * 64)()
        tg2 = (ctypes.c_int32 * 64)()
        c2 = ctypes.c_int64(0)
        nat.check(L.nnc_profile_end(ms2, tg2, 64, ctypes.byref(c2)))
        stream_ms = np.array([ms2[i] for i in range(min(c2.value, 64)) if tg2[i] == 0], dtype=np.float64)
        del km2, xq

    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if group is not None:
        import torch.distributed as dist

        dist.all_reduce(tmax, op=dist.ReduceOp.MAX, group=group)
    dt = float(tmax.item())
    # N > 1: every rank must have run the same fit (the status is replicated: same iterations, events, centres); checked, not assumed
    agreement = None
    if group is not None and res.model is not None:
        import hashlib
        import torch.distributed as dist

        h = int.from_bytes(hashlib.sha256(res.model.cluster_centers_.tobytes()).digest()[:7], "little")
        mine_t = torch.tensor([int(res.model.n_iter_), int(res.model.n_relocations_), h], dtype=torch.int64, device=dev)
        alls = [torch.zeros_like(mine_t) for _ in range(world)]
        dist.all_gather(alls, mine_t, group=group)
        alls = torch.stack(alls).cpu().numpy()
        agreement = {"n_iter": bool((alls[:, 0] == alls[0, 0]).all()), "relocations": bool((alls[:, 1] == alls[0, 1]).all()),
                     "centres_sha": bool((alls[:, 2] == alls[0, 2]).all()), "n_iter_per_rank": [int(v) for v in alls[:, 0]]}
        assert all(v for k, v in agreement.items() if k != "n_iter_per_rank"), f"ranks disagree on the fit: {agreement}"

    if rank == 0 and args.dump_durations:
        print("iteration kernels of one step, launch order (tag:us): " + " ".join(f"{int(t)}:{float(m) * 1e3:.0f}" for t, m in zip(tags_k, ms_k)), file=sys.stderr)
    if rank == 0:
        n_loc = hi - lo
        n_iter = res.model.n_iter_ if res.model is not None else 0
        k_fit = int(res.model.cluster_centers_.size) if res.model else 0
        label_bytes = 1 if k_fit <= 256 else 2
        # per-kernel durations over the timed region (HIP events on the launching stream, in-library)
        names = {0: "k_assign<accumulate>", 1: "k_bounds", 2: "k_assign<labels>", 3: "k_threshold", 4: "k_chunk_sums", 5: "k_finalize",
                 6: "k_prefix_blocks", 7: "k_minmax", 8: "k_lloyd", 9: "k_reloc_*"}
        # algorithmic bytes per weight and launch; None: not a pass over the vector
        bpw = {0: 4, 1: None, 2: 4 + label_bytes + 4, 3: 9, 4: 4, 5: None, 6: 4, 7: 4, 8: None, 9: None}
        kernels = {}
        for tag, name in names.items():
            d = all_ms[all_tags == tag]
            nst = args.steps
            if tag in per_step_scale:       # timed in the extra step
                d = ms_k[tags_k == tag]
                nst = 1
            if d.size == 0:
                continue
            live = d[d > 0.5 * np.median(d)] if tag in (0, 1, 5, 8, 9) else d   # launches enqueued behind a stop / pause return at once
            ent = {"launches_per_step": d.size / nst, "avg_ms": float(live.mean()), "median_ms": float(np.median(live)),
                   "ms_per_step": float(d.sum() / nst), "timed": "timed region" if nst == args.steps and tag not in per_step_scale else "one extra step"}
            if bpw[tag] is not None:
                ent["algorithmic_bytes_per_launch"] = bpw[tag] * n_loc
                ent["achieved_GBps"] = bpw[tag] * n_loc / (ent["avg_ms"] * 1e-3) / 1e9
                ent["frac_of_hbm_peak"] = ent["achieved_GBps"] / HBM_PEAK_GBS
            kernels[name] = ent
        # HBM bytes per launch of the roofline kernel from the PMC passes of tools/make_profiles.sh (rocprofv3 --pmc FETCH_SIZE /
        # WRITE_SIZE, corrected as MI355X_MICROARCH.md prescribes), if this round's summary is in the tree: a number measured in
        # another run of the same command, labelled as such -- bench.py itself cannot read the counters
        traffic, traffic_source = None, None
        for cand in ("r04_pmc_hbm_summary.json", "r03_pmc_hbm_summary.json"):
            tr_path = os.path.join(ROOT, "profiles", cand)
            if os.path.exists(tr_path):
                try:
                    traffic = json.load(open(tr_path)).get("k_assign_labels_hbm_bytes_per_launch")
                    traffic_source = f"profiles/{cand}: separate rocprofv3 --pmc passes over `python bench.py --steps 3 --warmup 1`, not this run"
                except Exception:
                    traffic = None
                if traffic is not None:
                    break
        lab = kernels.get("k_assign<labels>", {})
        out = {
            "metric": "weights/sec through prune+k-means (K=256)",
            "value": n_total * args.steps / dt,
            "unit": "weights/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": f"configs[3]: synthetic {(hi - lo)/1e6:g} M fp32 weights per GPU ({n_total/1e6:g} M total), "
                            f"prune q={args.q} sigma -> CDF -> {args.mode}-init k-means bits={args.bits} "
                            f"(K={k_fit}) to convergence -> labels+values -> Huffman lengths",
                "weights_per_gpu": hi - lo, "exchange": ("rccl-in-library" if comm is not None else (("torch.distributed" + (f" (library communicator unavailable: {comm_note})" if comm_note else "")) if world > 1 else None)),
                "k": k_fit, "lloyd_iterations": int(n_iter), "stop": res.model.stop_reason_ if res.model else None,
                "relocations": int(res.model.n_relocations_) if res.model else 0,
                "relocation_ties": int(getattr(res.model, "reloc_tie_", 0)) if res.model else 0,
                "parallelism": f"shard{world}" if world > 1 else "single",
                "rccl_world": (int(L.nnc_comm_world(comm.handle)) if comm is not None else None),
                "backend": (args.backend if world > 1 else None),
                "weight_iterations_per_s": n_total * n_iter * args.steps / dt,
                "input": ("a batch of its own per step, resident in HBM before the timed region (prune_weigth works in place)" if pooled
                          else "one resident vector, copied device-to-device at the head of every timed step (pool beyond --input-pool-gb)"),
            },
            # The k-means assignment pass over the whole vector (north_star's roofline kernel): E-step on the original order,
            # centroid index + decoded value written per weight.  The Lloyd iterations themselves no longer stream the vector
            # (rank-boundary form, k_bounds: O(K log N) reads per iteration), so they have no HBM roofline; their streaming
            # form is reported in `roofline_streaming_accumulate`.
            "roofline": {
                "bound": "hbm", "kernel": "k_assign<labels> (final E-step: 4 B read + centroid index + 4 B decoded value written per weight)",
                "achieved": lab.get("achieved_GBps"), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": lab.get("frac_of_hbm_peak"),
                "traffic": traffic, "traffic_source": traffic_source,
                "avg_kernel_ms": lab.get("avg_ms"), "launches_timed": int((all_tags == 2).sum()),
                "timing": "HIP events around the launch, inside the timed region (rocprofv3 --kernel-trace of the same command: profiles/r04_bench_kernel_stats.csv)",
                "algorithmic_bytes_per_launch": lab.get("algorithmic_bytes_per_launch"),
                # the same launch by SURVEY 8(d)'s literal accounting (the 4 B READ per weight only; the index and the decoded value it also
                # writes not counted): what "HBM-read roofline on the assignment" means if the writes are left out
                "frac_read_only_4B": (4.0 * n_loc / (lab["avg_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS) if lab.get("avg_ms") else None,
                "accounting": "achieved / frac count the 10 B a weight this launch really moves (4 B read + 2 B index + 4 B value; counter traffic 1.00 x that); "
                              "frac_read_only_4B counts the 4 B read alone.  In the step the vector comes from HBM (2 ms of other traffic lie between its last use "
                              "and this launch); launched back to back on data the Infinity Cache still holds the same kernel takes 39.5 us = 0.79 (tools/time_assign.py)",
            },
            "kernels": kernels,
        }
        # The step as a whole against the same peak: the bytes no implementation of this pipeline could avoid moving, over the step
        # time -- and how much of the step sits in K-sized kernels (the Lloyd iterations and the relocation events read a few MB
        # each: latency, not bandwidth, is what they cost).  Per weight: the in-place pass needs its own copy of the input (4 r + 4 w),
        # sigma = two passes (8 r), threshold (4 r + 4 w + 1 w mask), mean / variance of the pruned tensor (8 r), labels + values
        # (4 r + index + 4 w); per surviving weight: the value sort (one read, one write of the sorted copy: 8) and the prefix pass (4 r).
        if res.model is not None and res.nzeroed is not None:
            nnz = n_loc - int(res.nzeroed)
            step_bytes = n_loc * (8 + 8 + 9 + 8 + 4 + label_bytes + 4) + nnz * (8 + 4)
            ksz = sum(kernels[nm]["ms_per_step"] for nm in ("k_bounds", "k_finalize", "k_lloyd", "k_reloc_*") if nm in kernels)
            out["roofline_step"] = {
                "bound": "hbm", "unavoidable_bytes_per_step": int(step_bytes), "achieved": step_bytes / (dt / args.steps) / 1e9, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": step_bytes / (dt / args.steps) / 1e9 / HBM_PEAK_GBS,
                "k_sized_kernel_ms_per_step": ksz, "k_sized_share_of_step": ksz / (dt / args.steps * 1e3),
                "note": "roofline.frac describes one kernel (the assignment pass over the vector); this is the whole step",
            }
        if stream_ms.size:
            a = 4.0 * n_loc / (stream_ms.mean() * 1e-3) / 1e9
            out["roofline_streaming_accumulate"] = {
                "bound": "hbm", "kernel": "k_assign<accumulate> (streaming Lloyd pass, 4 B read per weight; off the timed path: 30 launches, converged centres)",
                "achieved": a, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": a / HBM_PEAK_GBS,
                "avg_kernel_ms": float(stream_ms.mean()), "median_kernel_ms": float(np.median(stream_ms)), "launches_timed": int(stream_ms.size),
                "algorithmic_bytes_per_launch": 4 * n_loc,
            }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"], km_cpu, w_cpu, thr_cpu = cpu_baseline(args, w_host)
            if res.model is not None:
                out["parity"] = parity_record(args, res, km_cpu, w_cpu, w_cpu.size, thr_cpu, w_host)
        elif world == 1 and res.model is not None:
            out["parity"] = parity_record(args, res, None, None, 0, 0)
        if agreement is not None:
            out["config"]["ranks_agree"] = agreement
        print(json.dumps(out))
    if comm is not None:
        comm.close()
    if group is not None:
        import torch.distributed as dist

        dist.destroy_process_group()


if __name__ == "__main__":
    main()
