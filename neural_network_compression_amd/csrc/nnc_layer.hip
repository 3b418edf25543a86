// nnc_layer.hip -- one layer tensor through the whole hot path as ONE host call: prune -> statistics -> sorted copy ->
// weight distribution -> initial centroids -> Lloyd fit -> centroid indices + decoded values -> index histogram -> Huffman
// code lengths.  What Trainer._prune_parameters (common/trainer.py:177-193) and Trainer.quantize (common/trainer.py:42-72)
// do to one tensor, with the K-sized host arithmetic of utility.py (np.linspace, the cumulative distribution and its linear
// interpolation, the density init) restated here in the reference's own order of float32 / float64 operations, so that a
// caller -- and several host threads at once, one stream each -- spends one foreign call per layer instead of forty.
// Everything on the device goes through the entry points of nnc_hip.hip / nnc_sort.hip; this file adds no kernel.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "nnc.h"

int nnc_set_error_(int code, const char *msg); // in nnc_hip.hip
int nnc_publish_bytes_(const void *src_dev, void *dst_host_mapped, int nbytes, void *host_ticket, uint64_t ticket, void *stream);
int nnc_wait_ticket_(void *host_ticket, uint64_t ticket, void *stream);
int nnc_sort_pruned_bounded_flagged_(const float *x, int64_t n, int64_t n_neg, int64_t n_zero, float vmin, float vmax, float thr,
                                     float *sorted_out, void *ws, size_t ws_bytes, int32_t *flag_dev, void *stream); // in nnc_sort.hip

// --------------------------------------------------------------------------------------
// host arithmetic, NumPy's way
// --------------------------------------------------------------------------------------
// np.linspace(start, stop, num) for float32 scalars (NumPy >= 2: the result type of two float32 is float32;
// numpy/_core/function_base.py): step = (stop - start) / (num - 1); y[i] = float32(i) * step + start, all in float32, and the
// last entry is `stop` itself.  A zero step (denormal range) takes numpy's other branch: y[i] = (i / div) * delta + start.
extern "C" int nnc_host_linspace_f32(float start, float stop, int32_t num, float *out)
{
    if (num < 0 || (num > 0 && !out)) return nnc_set_error_(NNC_EINVAL, "nnc_host_linspace_f32: bad argument");
    if (num == 0) return NNC_OK;
    const int div = num - 1;
    const volatile float delta = stop - start;
    if (div > 0) {
        const volatile float step = delta / (float)div;
        for (int i = 0; i < num; i++) {
            volatile float y = (float)i;
            if (step == 0.0f) { y = y / (float)div; y = y * delta; }
            else y = y * step;
            y = y + start;
            out[i] = y;
        }
        out[num - 1] = stop;
    } else {
        // num == 1: numpy multiplies by delta (step is NaN, retstep only) and adds start
        volatile float y = 0.0f;
        y = y * delta;
        y = y + start;
        out[0] = y;
    }
    return NNC_OK;
}

static void linspace_f64(double start, double stop, int num, double *out)
{
    const int div = num - 1;
    const volatile double delta = stop - start;
    if (div > 0) {
        const volatile double step = delta / (double)div;
        for (int i = 0; i < num; i++) {
            volatile double y = (double)i;
            if (step == 0.0) { y = y / (double)div; y = y * delta; }
            else y = y * step;
            y = y + start;
            out[i] = y;
        }
        out[num - 1] = stop;
    } else if (num == 1) {
        volatile double y = 0.0;
        y = y * delta;
        out[0] = y + start;
    }
}

// get_weight_distribution's host part (common/utility.py:374-392) from the 32 float32 steps and the 31 bin counts:
// normalised counts -> running float64 sum -> divided by its last entry -> scipy's linear interp1d at 300 points
// (xnew float32, cdf float64).  Same operations, same order, same types as the reference's calls.
extern "C" int nnc_host_cdf(const float *steps32, const int64_t *counts31, float *xnew300, double *cdf300)
{
    if (!steps32 || !counts31 || !xnew300 || !cdf300) return nnc_set_error_(NNC_EINVAL, "nnc_host_cdf: null pointer");
    const float *x = steps32; // x = steps[:-1]: 31 values
    int64_t tot = 0;
    for (int i = 0; i < 31; i++) tot += counts31[i];
    double cdf[31];
    volatile double acc = 0.0;
    for (int i = 0; i < 31; i++) {
        const volatile double t = (double)counts31[i] / (double)tot;
        acc = (i == 0) ? (double)t : acc + t;
        cdf[i] = acc;
    }
    const double last = cdf[30];
    for (int i = 0; i < 31; i++) { const volatile double q = cdf[i] / last; cdf[i] = q; }
    float xmin = x[0], xmax = x[0];
    for (int i = 1; i < 31; i++) { xmin = std::min(xmin, x[i]); xmax = std::max(xmax, x[i]); }
    int rc = nnc_host_linspace_f32(xmin, xmax, 300, xnew300);
    if (rc) return rc;
    for (int j = 0; j < 300; j++) {
        const float v = xnew300[j];
        int idx = (int)(std::lower_bound(x, x + 31, v) - x); // np.searchsorted(x, xnew), side="left"
        idx = std::min(std::max(idx, 1), 30);
        const int lo = idx - 1, hi = idx;
        const volatile float dx = x[hi] - x[lo];          // float32
        const volatile double dy = cdf[hi] - cdf[lo];
        const volatile double slope = dy / (double)dx;
        const volatile float xr = v - x[lo];              // float32
        const volatile double prod = slope * (double)xr;
        cdf300[j] = prod + cdf[lo];
    }
    return NNC_OK;
}

// utility.py:211-221: for each target t of np.linspace(0, 1, 2**bits + 1) the x of the FIRST cdf value closest to t.
extern "C" int nnc_host_density_init(const float *xnew300, const double *cdf300, int32_t bits, float *space_out)
{
    if (!xnew300 || !cdf300 || !space_out || bits < 0 || bits > 10) return nnc_set_error_(NNC_EINVAL, "nnc_host_density_init: bad argument");
    const int k = (1 << bits) + 1;
    std::vector<double> tmp((size_t)k);
    linspace_f64(0.0, 1.0, k, tmp.data());
    for (int t = 0; t < k; t++) {
        int best = 0;
        double bd = std::fabs(cdf300[0] - tmp[t]);
        for (int j = 1; j < 300; j++) {
            const double d = std::fabs(cdf300[j] - tmp[t]);
            if (d < bd) { bd = d; best = j; } // (NaN never wins: np.argmin would return the first NaN, which a cdf does not hold)
        }
        space_out[t] = xnew300[best];
    }
    return NNC_OK;
}

// --------------------------------------------------------------------------------------
// the layer
// --------------------------------------------------------------------------------------
static size_t al(size_t b) { return (b + 255) & ~(size_t)255; }

// A second stream (and two events) per host thread: the mean / variance passes of a long tensor run beside its sort.
struct SideStream { int dev = -1; hipStream_t stream = nullptr; hipEvent_t fork = nullptr, join = nullptr; };
static int side_stream(SideStream **out)
{
    static thread_local SideStream side;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nnc_set_error_(NNC_EHIP, "hipGetDevice failed");
    if (side.dev != dev) {
        if (side.stream) { (void)hipStreamDestroy(side.stream); (void)hipEventDestroy(side.fork); (void)hipEventDestroy(side.join); side = SideStream(); }
        if (hipStreamCreateWithFlags(&side.stream, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&side.fork, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&side.join, hipEventDisableTiming) != hipSuccess)
            return nnc_set_error_(NNC_EHIP, "could not create the second stream");
        side.dev = dev;
    }
    *out = &side;
    return NNC_OK;
}

struct LayerLayout {
    size_t prune_ws, stats_out, stats_ws, sorted, sort_ws, km_ws, prefix, reloc, back, small_out, total;
    size_t prune_ws_bytes, stats_ws_bytes, sort_ws_bytes, km_ws_bytes, prefix_bytes, reloc_bytes;
};

static LayerLayout layer_layout(int64_t n, int32_t k)
{
    LayerLayout L;
    size_t o = 0;
    auto take = [&](size_t bytes) { const size_t at = o; o += al(bytes); return at; };
    L.prune_ws_bytes = nnc_prune_stats_workspace_bytes(n);
    L.stats_ws_bytes = std::max(nnc_layer_stats_workspace_bytes(n), nnc_minmax_workspace_bytes(n));
    // the pruned sort is taken when at least a quarter of the weights are zero: its workspace is largest at exactly a quarter
    L.sort_ws_bytes = std::max(std::max(nnc_sort_workspace_bytes(n), nnc_sort_pruned_workspace_bytes(n, 0, (n + 3) / 4)),
                               nnc_sort_pruned_bounded_workspace_bytes(n - (n + 3) / 4));
    L.km_ws_bytes = nnc_kmeans_workspace_bytes(k);
    L.prefix_bytes = nnc_kmeans_prefix_bytes(n);
    L.reloc_bytes = n >= 512 ? nnc_kmeans_reloc_scratch_bytes(k, 256) : 0;
    L.prune_ws = take(std::max<size_t>(L.prune_ws_bytes, 16));
    L.stats_out = take(512); // out6 | signs[2] | prune stats[2] | nzeroed | ranks[33] (one read covers them)
    L.stats_ws = take(std::max<size_t>(L.stats_ws_bytes, 16));
    L.sorted = take((size_t)n * 4 + 16);
    L.sort_ws = take(std::max<size_t>(L.sort_ws_bytes, 16));
    L.km_ws = take(L.km_ws_bytes);
    L.prefix = take(std::max<size_t>(L.prefix_bytes, 16));
    L.reloc = take(std::max<size_t>(L.reloc_bytes, 16));
    L.back = take((size_t)k * 12);     // index histogram (k int64) | centres (k float32) ...
    L.small_out = take(64 + 256);      // ... and right behind it the one-launch fit's result block: one read covers both
    L.total = o + 256;
    return L;
}

extern "C" size_t nnc_compress_layer_workspace_bytes(int64_t n, int32_t k)
{
    if (n < 1 || k < 1 || k > NNC_KMAX - 8) return 0;
    return layer_layout(n, k).total;
}

// host block: [0, 512) the fit's two status slots | [512, 1024) scalars, ranks, the 32 steps | [1024, 16384) the K-sized read at
// the end | [16384, 24576) the initial centres (the kernels read them from here)
extern "C" size_t nnc_compress_layer_host_bytes(void) { return 24576; }

#define LCHK(call) do { int rc_ = (call); if (rc_) return rc_; } while (0)
#define LHIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return nnc_set_error_(NNC_EHIP, hipGetErrorString(e_)); } while (0)

extern "C" int nnc_compress_layer_f32(float *x, int64_t n, const nnc_layer_params *lp, uint8_t *mask_out, void *labels_out,
                                      float *values_out, void *ws_dev, size_t ws_bytes, void *host_pinned, size_t host_bytes,
                                      uint64_t *ticket_io, nnc_layer_result *res, void *stream)
{
    if (!x || n < 1 || !lp || !labels_out || !ws_dev || !host_pinned || !ticket_io || !res)
        return nnc_set_error_(NNC_EINVAL, "nnc_compress_layer_f32: null pointer or empty tensor");
    if (lp->bits < 1 || lp->bits > 10 || (lp->mode != NNC_INIT_LINEAR && lp->mode != NNC_INIT_DENSITY))
        return nnc_set_error_(NNC_EINVAL, "nnc_compress_layer_f32: bits in 1..10, mode linear or density");
    if (lp->prune && !mask_out) return nnc_set_error_(NNC_EINVAL, "nnc_compress_layer_f32: pruning needs mask_out");
    const int32_t k = (1 << lp->bits) + (lp->mode == NNC_INIT_DENSITY ? 1 : 0);
    if (n < (int64_t)(1 << lp->bits) + 1) return nnc_set_error_(NNC_EINVAL, "nnc_compress_layer_f32: not enough weights for this many centroids");
    if (host_bytes < nnc_compress_layer_host_bytes() || (reinterpret_cast<uintptr_t>(host_pinned) & 7) != 0)
        return nnc_set_error_(NNC_EINVAL, "nnc_compress_layer_f32: host block too small or unaligned");
    const LayerLayout L = layer_layout(n, k);
    if (ws_bytes < L.total) return nnc_set_error_(NNC_ENOSPACE, "nnc_compress_layer_f32: workspace too small");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    unsigned char *wb = reinterpret_cast<unsigned char *>((reinterpret_cast<uintptr_t>(ws_dev) + 255) & ~(uintptr_t)255);
    unsigned char *hb = reinterpret_cast<unsigned char *>(host_pinned);
    // host block: [0, 504) the fit's two status slots; 504 the ticket of this call's own reads; [512, 576) a mirror of the
    // device's scalar block (out6 | signs[2] | prune {sigma, threshold} | nzeroed), [576, 840) the ranks behind it; 864 the ticket and 872 the two floats of the read that
    // travels on the second stream (mean, variance); [896, 1024) the 32 histogram
    // steps (read from here by the rank kernel); [1024, ...) the K-sized read at the end.  A read = one small launch that copies into the block and
    // writes a ticket behind the bytes; the thread spins on the ticket (nnc_kmeans_status_publish's way).
    const float *h_f = reinterpret_cast<const float *>(hb + 512);                // out6
    const int64_t *h_signs = reinterpret_cast<const int64_t *>(hb + 512 + 32);   // #negative, #zero
    const float *h_prune = reinterpret_cast<const float *>(hb + 512 + 48);       // sigma, threshold
    const int64_t *h_nz = reinterpret_cast<const int64_t *>(hb + 512 + 56);
    int64_t *h_ranks = reinterpret_cast<int64_t *>(hb + 576);
    void *h_ticket = hb + 504;
    auto read_back = [&](const void *src, void *dst, int nbytes) -> int {
        const uint64_t t = ++(*ticket_io);
        int rc = nnc_publish_bytes_(src, dst, nbytes, h_ticket, t, stream);
        if (rc) return rc;
        return nnc_wait_ticket_(h_ticket, t, stream);
    };
    std::memset(res, 0, sizeof(*res));
    res->k = k;
    res->label_bytes = k <= 256 ? 1 : 2;

    float *out6 = reinterpret_cast<float *>(wb + L.stats_out);
    int64_t *signs = reinterpret_cast<int64_t *>(wb + L.stats_out + 32);
    float *pstats = reinterpret_cast<float *>(wb + L.stats_out + 48);
    int64_t *nzeroed = reinterpret_cast<int64_t *>(wb + L.stats_out + 56);
    // ---- prune_weigth (utility.py:131-169): in place
    // ... and, in the same pass, min / max (over all and over the non-zero weights) and the sign counts of what is left:
    // out6[2..5], signs.  Without pruning: the statistics pass on its own.
    if (lp->prune)
        LCHK(nnc_prune_stats_f32(x, n, lp->q, lp->std_smooth ? 1 : 0, mask_out, pstats, nzeroed, out6 + 2, signs, wb + L.prune_ws, L.prune_ws_bytes, stream));
    else
        LCHK(nnc_minmax_signs_f32(x, n, out6 + 2, signs, wb + L.stats_ws, L.stats_ws_bytes, stream));
    const bool short_tensor = n <= NNC_REF_NMAX && k <= NNC_REF_KMAX;
    // (not taken here: short tensors with the density init, tensors too short for the sorted form.  The tensor is pruned
    // already; the caller's own path goes on from there, see status)
    if ((short_tensor && lp->mode != NNC_INIT_LINEAR) || (!short_tensor && n < 512)) {
        if (lp->prune) {
            LCHK(read_back(out6, hb + 512, 64));
            res->sigma = h_prune[0]; res->threshold = h_prune[1]; res->n_zeroed = h_nz[0];
        }
        res->status = NNC_LAYER_HOST;
        return NNC_OK;
    }
    float space[NNC_KMAX];
    unsigned char *back = wb + L.back;
    int64_t *counts_d = reinterpret_cast<int64_t *>(back);
    float *centers_d = reinterpret_cast<float *>(back + (size_t)k * 8);
    int64_t *h_counts = reinterpret_cast<int64_t *>(hb + 1024);
    float *h_centers = reinterpret_cast<float *>(hb + 1024 + (size_t)k * 8);
    float *h_steps = reinterpret_cast<float *>(hb + 896), *h_space = reinterpret_cast<float *>(hb + 16384);

    if (short_tensor) {
        // ---- a short tensor: min / max -> linear init -> the whole fit in one launch, in the reference's own arithmetic
        LCHK(read_back(out6, hb + 512, 64));
        LCHK(nnc_host_linspace_f32(h_f[2], h_f[3], k, space));
        std::memcpy(h_space, space, (size_t)k * 4); // (the kernel reads the k floats straight from the pinned block: no copy command)
        void *result_d = wb + L.small_out;
        LCHK(nnc_kmeans_fit_reference_f32(x, (int32_t)n, h_space, k, 300, 1e-4f, reinterpret_cast<uint8_t *>(labels_out), lp->want_values ? values_out : nullptr,
                                          centers_d, counts_d, result_d, stream));
        // (the 32-byte result block sits right behind the K-sized block on the device, see layer_layout: one read)
        const int back_bytes = (int)al((size_t)k * 12) + 32;
        LCHK(read_back(back, hb + 1024, back_bytes));
        const int32_t *h_res = reinterpret_cast<const int32_t *>(hb + 1024 + al((size_t)k * 12));
        {
            const float *h_resf = reinterpret_cast<const float *>(h_res); // (RefOut: six int32, then x_mean, tol)
            if (!std::isfinite(h_resf[6]) || !std::isfinite(h_resf[7])) { // NaN / infinity in the tensor: the caller's own path raises, as KMeans.fit does
                if (lp->prune) { res->sigma = h_prune[0]; res->threshold = h_prune[1]; res->n_zeroed = h_nz[0]; }
                res->status = NNC_LAYER_HOST;
                return NNC_OK;
            }
        }
        res->n_iter = h_res[0]; res->stop = h_res[1]; res->n_relocations = h_res[2]; res->reloc_ties = h_res[3]; res->reloc_multi = h_res[4];
        res->arith = NNC_ARITH_REFERENCE;
    } else {
        // ---- min / max and the sign counts are read first (the sort needs the counts on the host); mean and variance -- two passes
        // and two single-wave folds, 90 us of mostly latency -- then run on a second stream beside the sort
        const int64_t nch = (n + 8191) / 8192; // NumPy's summation chunk
        float *c1 = reinterpret_cast<float *>(wb + L.stats_ws), *c2 = c1 + nch;
        LCHK(read_back(out6, hb + 512, 64));
        const float xmin = h_f[2], xmax = h_f[3], min_nz = h_f[4], max_nz = h_f[5];
        const int64_t n_neg = h_signs[0], n_zero = h_signs[1];
        SideStream *side = nullptr;
        LCHK(side_stream(&side));
        LHIP(hipEventRecord(side->fork, s));
        LHIP(hipStreamWaitEvent(side->stream, side->fork, 0));
        // From here on work is in flight on the second stream, writing into the caller's workspace: every way out of this scope --
        // an error return included -- first makes the caller's stream wait for it (and, on an error, waits itself), so that the
        // caller may drop the workspace as soon as the call is back.
        struct SideJoin {
            SideStream *side; hipStream_t s; bool joined = false;
            ~SideJoin() { if (!joined) { (void)hipEventRecord(side->join, side->stream); (void)hipStreamWaitEvent(s, side->join, 0); (void)hipStreamSynchronize(s); } }
        } side_join{side, s};
        LCHK(nnc_chunk_sums_f32(x, n, 0, nullptr, c1, side->stream));
        LCHK(nnc_fold_f32(c1, nch, n, NNC_FOLD_MEAN, nullptr, out6, side->stream));           // [0] = mean
        LCHK(nnc_chunk_sums_f32(x, n, 1, out6, c2, side->stream));
        LCHK(nnc_fold_f32(c2, nch, n, NNC_FOLD_MEAN, nullptr, out6 + 1, side->stream));       // [1] = variance
        // ... and go to the host from there, as soon as they exist (the sort is still running on the caller's stream): the fit's
        // parameters are then known and the prefix sums can be enqueued right behind the sort, in front of the round trip that
        // fetches the ranks for the initial centres -- the device builds the prefixes while the host works out the centres
        // (until round 4 both waited for one read behind the sort: 25-70 us of idle device per layer)
        void *h_ticket_side = hb + 864;
        const float *h_mv = reinterpret_cast<const float *>(hb + 872);
        const uint64_t t_mv = ++(*ticket_io);
        LCHK(nnc_publish_bytes_(out6, hb + 872, 8, h_ticket_side, t_mv, side->stream));
        LHIP(hipEventRecord(side->join, side->stream));
        float *xs = reinterpret_cast<float *>(wb + L.sorted);
        int32_t *oob_d = reinterpret_cast<int32_t *>(wb + L.stats_out + 336); // the bounded sort's verdict: a weight outside the bounds it was given
        const int32_t *h_oob = reinterpret_cast<const int32_t *>(hb + 512 + 336);
        bool bounded = false;
        // (a pruned tensor whose threshold is known: the surviving weights span few key bits -- three radix passes instead of four)
        if (4 * n_zero >= n && lp->prune && nnc_sort_pruned_bounded_bits(xmin, xmax, h_prune[1], n_neg, n - n_neg - n_zero) > 0) {
            LCHK(nnc_sort_pruned_bounded_flagged_(x, n, n_neg, n_zero, xmin, xmax, h_prune[1], xs, wb + L.sort_ws, L.sort_ws_bytes, oob_d, stream));
            bounded = true;
        } else if (4 * n_zero >= n) LCHK(nnc_sort_pruned_f32(x, n, n_neg, n_zero, xs, wb + L.sort_ws, L.sort_ws_bytes, stream));
        else LCHK(nnc_sort_f32(x, n, xs, wb + L.sort_ws, L.sort_ws_bytes, stream));
        const bool with_prefix = (reinterpret_cast<uintptr_t>(xs) & 15) == 0;
        int64_t *ranks_d = reinterpret_cast<int64_t *>(wb + L.stats_out + 64);
        h_ranks = reinterpret_cast<int64_t *>(hb + 512 + 64);
        float steps[32];
        if (lp->mode == NNC_INIT_DENSITY) {
            if (!std::isfinite(min_nz)) { // no non-zero weight: the reference's numpy call raises; so does the caller's own path
                LHIP(hipStreamWaitEvent(s, side->join, 0));
                LHIP(hipStreamSynchronize(s));
                side_join.joined = true;
                if (lp->prune) { res->sigma = h_prune[0]; res->threshold = h_prune[1]; res->n_zeroed = h_nz[0]; }
                res->status = NNC_LAYER_HOST;
                return NNC_OK;
            }
            LCHK(nnc_host_linspace_f32(min_nz, max_nz, 32, steps));
            std::memcpy(h_steps, steps, 128); // (read by the kernel straight from the pinned block)
            LCHK(nnc_rank_sorted_f32(xs, n, h_steps, 32, ranks_d, stream));
        }
        // ---- the second read (enqueued, not waited for yet): min / max, signs, the ranks, the sort's verdict
        const uint64_t t_rk = ++(*ticket_io);
        LCHK(nnc_publish_bytes_(out6 + 2, hb + 512 + 8, 56 + 33 * 8 + 16, h_ticket, t_rk, stream));
        LCHK(nnc_wait_ticket_(h_ticket_side, t_mv, side->stream));
        LHIP(hipStreamWaitEvent(s, side->join, 0));
        side_join.joined = true;
        const float mean = h_mv[0], var = h_mv[1];
        if (!std::isfinite(mean) || !std::isfinite(var) || !std::isfinite(xmin) || !std::isfinite(xmax)) {
            // a NaN or an infinity in the tensor: KMeans.fit raises on such input, and so does the caller's own path (from the pruned tensor)
            LCHK(nnc_wait_ticket_(h_ticket, t_rk, stream));
            if (lp->prune) { res->sigma = h_prune[0]; res->threshold = h_prune[1]; res->n_zeroed = h_nz[0]; }
            res->status = NNC_LAYER_HOST;
            return NNC_OK;
        }
        // ---- k-means parameters (KMeans.fit's tolerance and centring, _kmeans.py:279-287, 1479-1484; the fixed-point rule)
        nnc_kmeans_params p;
        std::memset(&p, 0, sizeof(p));
        const volatile float tol = var * 1e-4f;                 // np.mean(np.var(X, axis=0)) * tol, float32
        const volatile float lo = xmin - mean, hi = xmax - mean; // exact range of the centred data
        p.n = n; p.n_total = n; p.k = k; p.max_iter = 300;
        p.fix_shift = nnc_fix_shift(std::max(std::fabs((float)lo), std::fabs((float)hi)), n);
        p.grid_log2 = 0; p.replicas_log2 = -1; p.flags = lp->km_flags & (NNC_KM_TWO_LAUNCH | NNC_KM_LOOP);
        p.x_mean = mean; p.tol = tol; p.lo = lo; p.hi = hi;
        // the prefix sums of the sorted copy: behind the sort on the caller's stream, while the ranks travel
        // (the pointer is set before the fit is initialised: its first finalize step then knows that the iterations will not read the
        // cell table and leaves it to the labelling pass at the end -- one launch of k_cells less)
        if (with_prefix) {
            LCHK(nnc_kmeans_prefix_build(xs, &p, reinterpret_cast<int64_t *>(wb + L.prefix), stream));
            p.prefix_dev = reinterpret_cast<int64_t *>(wb + L.prefix);
        }
        LCHK(nnc_wait_ticket_(h_ticket, t_rk, stream));
        if (bounded && *h_oob) {
            // a weight outside [min, -threshold] u {0} u [threshold, max] -- a NaN, the statistics pass ignores those: the compact-key
            // sort clamped it, so the sorted copy is not the tensor's.  The step-by-step path (general sorts) goes on from here.
            if (lp->prune) { res->sigma = h_prune[0]; res->threshold = h_prune[1]; res->n_zeroed = h_nz[0]; }
            res->status = NNC_LAYER_HOST;
            return NNC_OK;
        }
        // ---- initial centroids (utility.py:206-226)
        if (lp->mode == NNC_INIT_LINEAR) {
            LCHK(nnc_host_linspace_f32(xmin, xmax, k, space));
        } else {
            int64_t counts31[31];
            for (int b = 0; b < 31; b++) counts31[b] = h_ranks[b + 1] - h_ranks[b];
            for (int b = 0; b < 31; b++)
                if (steps[b] <= 0.0f && 0.0f < steps[b + 1]) { counts31[b] -= n_zero; break; } // the zeros sit in that bin of the full vector
            float xnew[300];
            double cdf[300];
            LCHK(nnc_host_cdf(steps, counts31, xnew, cdf));
            LCHK(nnc_host_density_init(xnew, cdf, lp->bits, space));
        }
        std::memcpy(h_space, space, (size_t)k * 4);
        LCHK(nnc_kmeans_init(wb + L.km_ws, L.km_ws_bytes, &p, h_space, stream));
        // ---- the Lloyd loop
        nnc_kmeans_status st;
        std::memset(&st, 0, sizeof(st));
        int32_t nwin = 0;
        LCHK(nnc_kmeans_fit(xs, wb + L.km_ws, &p, 16, 1, L.reloc_bytes ? wb + L.reloc : nullptr, L.reloc_bytes, hb, ticket_io, &st, &nwin, stream));
        if (!st.done) { // full-pass relocation / strict-convergence check: the caller's own path (from the pruned tensor)
            if (lp->prune) { res->sigma = h_prune[0]; res->threshold = h_prune[1]; res->n_zeroed = h_nz[0]; }
            res->status = NNC_LAYER_HOST;
            return NNC_OK;
        }
        // ---- labels + decoded values from the original order, index histogram, centres: one host read
        // (one after the other: beside the labelling pass the index histogram's launches take bandwidth from it -- measured: the
        // labelling pass 43.6 -> 45.5 us for 15 us saved)
        LCHK(nnc_kmeans_get_centers(wb + L.km_ws, 0, 0, centers_d, stream));
        LCHK(nnc_kmeans_assign(x, wb + L.km_ws, &p, 0, labels_out, res->label_bytes, lp->want_values ? values_out : nullptr, nullptr, nullptr, stream));
        LCHK(nnc_kmeans_label_counts(xs, wb + L.km_ws, &p, 0, counts_d, stream));
        LCHK(read_back(back, hb + 1024, (int)al((size_t)k * 12)));
        res->n_iter = st.iter; res->stop = st.done; res->n_relocations = nwin; res->n_reloc_windowed = nwin;
        res->reloc_ties = st.reloc_ties; res->reloc_multi = st.reloc_multi;
        res->arith = NNC_ARITH_FIXED;
    }
    if (lp->prune) { res->sigma = h_prune[0]; res->threshold = h_prune[1]; res->n_zeroed = h_nz[0]; }
    std::memcpy(res->counts, h_counts, (size_t)k * 8);
    std::memcpy(res->centers, h_centers, (size_t)k * 4);
    int64_t hist[72]; // (nnc_huffman_lengths fills up to 65 entries: symbols per code length)
    LCHK(nnc_huffman_lengths(res->counts, k, res->code_lengths, hist, &res->total_bits));
    res->status = NNC_LAYER_DONE;
    return NNC_OK;
}
