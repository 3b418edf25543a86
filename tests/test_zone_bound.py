"""The error bound behind the E-step without K distance evaluations (csrc/nnc_hip.hip, km_pair_zone), checked on the CPU.

scikit-learn labels x with the first strict minimum over j of d_j = fl(fl(c_j^2) + fl(-2 * fl(x * c_j))) in float32
(_k_means_lloyd.pyx:196-213, reached from neural_network_compression/common/utility.py:237-238).  The device evaluates that
expression only for samples inside an interval around the crossing point of two centres and trusts the order of the centres
outside it.  This test restates the interval (same formulas, float64) and checks the claim by brute force with NumPy float32
arithmetic: for thousands of random pairs of centres, at every float32 value for hundreds of ulps beyond either end of the
interval and at random values further out, the float32 comparison is the one the interval promises."""
import numpy as np

U = 2.0 ** -24


def rcp_up(d):
    r = np.float64(np.float32(1.0) / np.float32(d))
    return r * (1.0 + 9.5367431640625e-07) if 1e-37 < d < 1e37 else 1.0 / d


def pair_zone(cp, cq, xb):
    """km_pair_zone: (lo, hi) for centres cp < cq (float64 values of float32 centres), |x| <= xb."""
    delta = cq - cp
    cm = max(abs(cp), abs(cq))
    mid = 0.5 * (cp + cq)
    E = 2.5 * U * (cm * cm + 2.0 * xb * cm) + 1e-42
    w0 = E * rcp_up(delta)
    lo, hi = mid - w0, mid + w0
    S = abs(cp) + abs(cq)
    den = 2.0 * delta - 4.5 * U * S
    if den > delta:
        Cp = np.float64(np.float32(cp) * np.float32(cp))
        Cq = np.float64(np.float32(cq) * np.float32(cq))
        d2 = 2.0 * delta
        if 1e-37 < d2 < 1e37:   # float32 reciprocal + one Newton step (the device avoids the double division)
            r0 = np.float64(np.float32(1.0) / np.float32(d2))
            xs = (Cq - Cp) * (r0 * (2.0 - d2 * r0))
        else:
            xs = (Cq - Cp) / d2
        Ds = max(abs(Cp - 2.0 * xs * cp), abs(Cq - 2.0 * xs * cq))
        num = U * (2.0 * Ds + 2.0000005 * abs(xs) * S) * 1.000001 + 1e-42
        w = (num * rcp_up(den)) * 1.000001 + abs(xs) * 1e-13
        lo, hi = max(lo, xs - w), min(hi, xs + w)
    return lo, hi


def dist32(x, c):
    """scikit-learn's float32 distance surrogate, every operation rounded to float32."""
    x = np.asarray(x, dtype=np.float32)
    c = np.float32(c)
    C = np.float32(c * c)
    t = (x * c).astype(np.float32)
    m = (np.float32(-2.0) * t).astype(np.float32)
    return (C + m).astype(np.float32)


def beyond(v, steps, up):
    """the `steps` float32 values right after v (up) or right before it (down)"""
    out = np.empty(steps, dtype=np.float32)
    cur = np.float32(v)
    tgt = np.float32(np.inf if up else -np.inf)
    for i in range(steps):
        cur = np.nextafter(cur, tgt, dtype=np.float32)
        out[i] = cur
    return out


def check_pair(cp, cq, xb, rng, steps=400):
    lo, hi = pair_zone(np.float64(cp), np.float64(cq), np.float64(xb))
    # first float32 strictly above hi / strictly below lo
    h32 = np.float32(hi)
    if np.float64(h32) <= hi:
        h32 = np.nextafter(h32, np.float32(np.inf), dtype=np.float32)
    l32 = np.float32(lo)
    if np.float64(l32) >= lo:
        l32 = np.nextafter(l32, np.float32(-np.inf), dtype=np.float32)
    above = np.concatenate([[h32], beyond(h32, steps, True), rng.uniform(h32, xb, 64).astype(np.float32)])
    below = np.concatenate([[l32], beyond(l32, steps, False), rng.uniform(-xb, l32, 64).astype(np.float32)])
    above = above[(np.abs(above) <= xb) & (above.astype(np.float64) > hi)]
    below = below[(np.abs(below) <= xb) & (below.astype(np.float64) < lo)]
    if above.size:
        bad = ~(dist32(above, cq) < dist32(above, cp))
        assert not bad.any(), ("above", cp, cq, xb, above[bad][:3], hi)
    if below.size:
        bad = ~(dist32(below, cp) < dist32(below, cq))
        assert not bad.any(), ("below", cp, cq, xb, below[bad][:3], lo)
    return hi - lo


def test_interval_is_safe_on_random_pairs():
    rng = np.random.RandomState(7)
    widths = []
    for trial in range(1500):
        scale = 10.0 ** rng.uniform(-6, 2)
        xb = np.float32(scale * rng.uniform(1.0, 8.0))
        c = np.float32(rng.uniform(-1.0, 1.0) * scale)
        gap = np.float32(abs(c) * 10.0 ** rng.uniform(-6.5, 0.5) + scale * 10.0 ** rng.uniform(-7, -1) * (trial % 3 == 0))
        cp, cq = np.float32(c), np.float32(c + gap)
        if not (cq > cp) or max(abs(cp), abs(cq)) > xb:
            continue
        widths.append(check_pair(cp, cq, xb, rng, steps=200))
    assert len(widths) > 1000


def test_interval_is_safe_on_the_bench_geometry():
    """0.05-sigma weights, 257 centres: the pairs the headline workload has (dense middle and sparse tails), every neighbour pair
    and the pairs two apart."""
    rng = np.random.RandomState(11)
    cs = np.sort((rng.standard_normal(257) * 0.05).astype(np.float32))
    xb = np.float32(0.27)
    for i in range(256):
        for d in (1, 2):
            if i + d < 257 and cs[i + d] > cs[i]:
                check_pair(cs[i], cs[i + d], xb, rng, steps=300)


def test_local_bound_is_tighter_where_it_matters():
    """In the dense middle (|c| well below the largest |x|) the interval is several times narrower than the global bound alone."""
    cp, cq, xb = np.float32(0.05), np.float32(0.0508), np.float32(0.25)
    lo, hi = pair_zone(np.float64(cp), np.float64(cq), np.float64(xb))
    cm = float(cq)
    w0 = (2.5 * U * (cm * cm + 2.0 * float(xb) * cm)) / float(np.float64(cq) - np.float64(cp))
    assert (hi - lo) < 0.25 * (2.0 * w0)


def test_centres_a_few_ulps_apart_fall_back_to_the_global_bound():
    cp = np.float32(0.1)
    cq = np.nextafter(cp, np.float32(1.0), dtype=np.float32)
    lo, hi = pair_zone(np.float64(cp), np.float64(cq), np.float64(0.3))
    assert hi - lo > 0.1  # float32 cannot tell them apart anywhere near
