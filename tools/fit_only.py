"""The bench fit alone (prune + init outside), a few times: for rocprofv3 --kernel-trace timelines of the Lloyd loop."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from neural_network_compression_amd import kmeans as km, pipeline, synth
from neural_network_compression_amd.common import utility as U
n = int(sys.argv[1]) if len(sys.argv) > 1 else 25_000_000
two = len(sys.argv) > 2 and sys.argv[2] == "two"
x = torch.from_numpy(synth.weights((n,), 4000)).cuda()
pipeline.prune_sharded_(x, 1.0, True, None)
cdfs = U.get_weight_distribution(x, skip_zeros=True)
space = np.asarray(U._init_space(x, x.numel(), 8, "density", cdfs), dtype=np.float32)
for rep in range(4):
    d = km.DeviceKMeans(x, space, two_launch=two, loop=not two)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    m, _ = d.fit()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("rep", rep, "n_iter", m.n_iter_, "fit ms %.3f" % (dt * 1e3), d.loop_stats())
    if hasattr(d.L, "nnc_debug_kl_trace") and os.environ.get("NNC_DIAG"):
        import ctypes
        tr = (ctypes.c_uint64 * 24)()
        d.L.nnc_debug_kl_trace(d.ws.data_ptr(), tr)
        names = ["prologue", "own boundaries", "wait for others", "-", "epilogue", "combine+scatter", "votes", "shift+pairwise+state", "order", "distinct+tables", "zones", "thresholds+phi", "-", "-", "wave0 labelling", "wait for labelling"]
        print("   first wave inside the search (us): between sub-passes %.1f  set-up %.1f  lean probes %.1f  general rounds %.1f  results %.1f" % tuple(tr[16 + i] * 0.01 for i in range(5)))
        print("   search rounds of the first wave: %d" % tr[3])
        print("   shader clock inside k_lloyd: %.0f MHz over %.1f us" % (tr[12] / max(1, tr[13]) * 100.0, tr[13] * 0.01))
        print("   k_lloyd phases (us): " + "  ".join(f"{nm} {tr[i] * 0.01:.1f}" for i, nm in enumerate(names) if nm != "-"))
