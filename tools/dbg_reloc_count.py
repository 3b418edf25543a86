import numpy as np, torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neural_network_compression_amd import kmeans as km
xe = np.full(5000, -0.375, dtype=np.float32)
init = np.array([-0.375, 0.1, 0.2, -0.375], dtype=np.float32)
t = torch.from_numpy(xe).cuda()
for tl in (False, True):
    d = km.DeviceKMeans(t, init, two_launch=tl)
    orig = d._relocate_and_resume
    def wrap(st, orig=orig, d=d):
        print("   python relocate: paused", st.paused, "iter", st.iter, "same", st.same_counts, "n_empty", st.n_empty, "counts before", d.n_relocations, d.n_reloc_windowed, d.n_reloc_full)
        orig(st)
    d._relocate_and_resume = wrap
    m, _ = d.fit()
    print("two_launch", tl, "n_iter", m.n_iter_, "reloc", m.n_relocations_, "windowed", m.n_reloc_windowed_, "full", d.n_reloc_full, "stop", m.stop_reason_)
