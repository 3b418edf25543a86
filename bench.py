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
The JSON line also carries
  roofline     : the k-means assignment pass over the vector, k_assign<labels> (4 B read + centroid index + 4 B decoded
                 value written per weight), against the 8 TB/s HBM peak; its duration is measured with HIP events
                 around the launch inside the timed region (in-library, on the launching stream);
  kernels      : the same for every data-touching kernel of the step (durations, launches per step, and -- for passes
                 over the vector -- algorithmic bytes and achieved GB/s);
  roofline_streaming_accumulate : the streaming form of the Lloyd pass (4 B read per weight per launch), which the
                 iterations on a sorted vector no longer need, timed on its own after the timed region;
  cpu_baseline : the same pipeline through NumPy / scikit-learn on the host cores (rank 0, N = 1
                 only) on a bounded sample of the same vector.
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
    libcalls.prune_weigth(w, args.q, True)
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
    }


def main():
    args = parse()
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
    w_host = synth.weights((hi - lo,), SEED, start=lo)
    w0 = torch.from_numpy(w_host).to(dev)

    def step():
        x = w0.clone()  # prune works in place; the copy is device-to-device, inside the timed region
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
        for cand in ("r03_pmc_hbm_summary.json",):
            tr_path = os.path.join(ROOT, "profiles", cand)
            if os.path.exists(tr_path):
                try:
                    traffic = json.load(open(tr_path)).get("k_assign_labels_hbm_bytes_per_launch")
                    traffic_source = f"profiles/{cand}: separate rocprofv3 --pmc passes over `python bench.py --steps 3 --warmup 1`, not this run"
                except Exception:
                    traffic = None
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
                "weight_iterations_per_s": n_total * n_iter * args.steps / dt,
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
                "timing": "HIP events around the launch, inside the timed region (rocprofv3 --kernel-trace of the same command: profiles/r03_bench_kernel_stats.csv)",
                "algorithmic_bytes_per_launch": lab.get("algorithmic_bytes_per_launch"),
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
            out["cpu_baseline"] = cpu_baseline(args, w_host)
        print(json.dumps(out))
    if comm is not None:
        comm.close()
    if group is not None:
        import torch.distributed as dist

        dist.destroy_process_group()


if __name__ == "__main__":
    main()
