/*
 * nnc_oracle.c -- CPU restatement of the reference's prune / k-means arithmetic.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under neural_network_compression_amd/ may import,
 * link or call this file; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, as the checker.
 *
 * What is restated, and from where (reference paths are relative to /root/reference):
 *   - prune_weigth                 neural_network_compression/common/utility.py:134-163
 *   - get_weight_distribution      neural_network_compression/common/utility.py:362-372  (31-bin counts)
 *   - get_quantized_weight         neural_network_compression/common/utility.py:237-239  (Lloyd fit + gather)
 * The arithmetic of those lines lives in third-party dependencies that are NOT in the
 * reference tree (pyproject.toml:10-13 pins numpy ~1.19.5, scikit-learn ~0.24.0; this
 * image has numpy 2.2.6, scikit-learn 1.7.2).  Their published algorithms are restated:
 *   - numpy float32 add.reduce: 8192-element buffered chunks folded left to right, each
 *     chunk by the pairwise routine (numpy/_core/src/umath/loops_utils.h.src, *_pairwise_sum);
 *   - numpy _mean/_var/_std (numpy/_core/_methods.py): sum, divide in double by the exact
 *     count, round to float32; var = sum((x-mean)^2)/n two-pass; std = sqrtf(var);
 *   - scikit-learn Lloyd E-step  (sklearn/cluster/_k_means_lloyd.pyx:168-218):
 *       d[j] = fl(|c_j|^2 + fl(-2 * fl(x*c_j))), label = first strict minimum;
 *     M-step (same file :215-218): per-cluster float32 running sums in sample order;
 *   - scikit-learn _relocate_empty_clusters_dense / _average_centers / _center_shift
 *     (sklearn/cluster/_k_means_common.pyx:167-311) are driven from oracle.py (they need
 *     numpy.argpartition's ordering), with the per-sample pieces here.
 * Parity is pinned against tests/golden/ref_goldens.* (outputs of the reference run in
 * the build container; see tests/golden/make_goldens.py).
 *
 * Build: make -C oracle   (gcc -O2 -ffp-contract=off: no FMA contraction, ever).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define NP_BUFSIZE 8192 /* numpy's default ufunc buffer, in elements */
#define PW_BLOCKSIZE 128

/* ---------------------------------------------------------------- numpy reductions */

static float pairwise_sum_f32(const float *a, int64_t n)
{
    if (n < 8) {
        float res = 0.0f;
        for (int64_t i = 0; i < n; i++) res += a[i];
        return res;
    } else if (n <= PW_BLOCKSIZE) {
        float r[8];
        int64_t i;
        for (int j = 0; j < 8; j++) r[j] = a[j];
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; j++) r[j] += a[i + j];
        float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += a[i];
        return res;
    } else {
        int64_t n2 = n / 2;
        n2 -= n2 % 8;
        return pairwise_sum_f32(a, n2) + pairwise_sum_f32(a + n2, n - n2);
    }
}

/* per-chunk sums (one float per 8192-element chunk), exposed so the sharded
 * (multi-rank) fold can be tested: chunk sums are gathered, then folded in order. */
void orc_np_chunk_sums_f32(const float *a, int64_t n, float *out)
{
    int64_t nchunks = (n + NP_BUFSIZE - 1) / NP_BUFSIZE;
    for (int64_t c = 0; c < nchunks; c++) {
        int64_t lo = c * NP_BUFSIZE;
        int64_t len = n - lo < NP_BUFSIZE ? n - lo : NP_BUFSIZE;
        out[c] = pairwise_sum_f32(a + lo, len);
    }
}

float orc_fold_f32(const float *chunk_sums, int64_t nchunks)
{
    float acc = 0.0f;
    for (int64_t c = 0; c < nchunks; c++) acc = acc + chunk_sums[c];
    return acc;
}

float orc_np_sum_f32(const float *a, int64_t n)
{
    float acc = 0.0f;
    for (int64_t lo = 0; lo < n; lo += NP_BUFSIZE) {
        int64_t len = n - lo < NP_BUFSIZE ? n - lo : NP_BUFSIZE;
        acc = acc + pairwise_sum_f32(a + lo, len);
    }
    return acc;
}

static float div_count_f32(float s, int64_t n) { return (float)((double)s / (double)n); }

float orc_np_mean_f32(const float *a, int64_t n) { return div_count_f32(orc_np_sum_f32(a, n), n); }

/* sum over (a[i]-mean)^2 with numpy's summation tree, without materialising the temp */
static float pairwise_sqdev_f32(const float *a, int64_t n, float mean)
{
    if (n < 8) {
        float res = 0.0f;
        for (int64_t i = 0; i < n; i++) { float d = a[i] - mean; res += d * d; }
        return res;
    } else if (n <= PW_BLOCKSIZE) {
        float r[8];
        int64_t i;
        for (int j = 0; j < 8; j++) { float d = a[j] - mean; r[j] = d * d; }
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; j++) { float d = a[i + j] - mean; r[j] += d * d; }
        float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) { float d = a[i] - mean; res += d * d; }
        return res;
    } else {
        int64_t n2 = n / 2;
        n2 -= n2 % 8;
        return pairwise_sqdev_f32(a, n2, mean) + pairwise_sqdev_f32(a + n2, n - n2, mean);
    }
}

void orc_np_chunk_sqdev_f32(const float *a, int64_t n, float mean, float *out)
{
    int64_t nchunks = (n + NP_BUFSIZE - 1) / NP_BUFSIZE;
    for (int64_t c = 0; c < nchunks; c++) {
        int64_t lo = c * NP_BUFSIZE;
        int64_t len = n - lo < NP_BUFSIZE ? n - lo : NP_BUFSIZE;
        out[c] = pairwise_sqdev_f32(a + lo, len, mean);
    }
}

float orc_np_var_f32(const float *a, int64_t n)
{
    float mean = orc_np_mean_f32(a, n);
    float acc = 0.0f;
    for (int64_t lo = 0; lo < n; lo += NP_BUFSIZE) {
        int64_t len = n - lo < NP_BUFSIZE ? n - lo : NP_BUFSIZE;
        acc = acc + pairwise_sqdev_f32(a + lo, len, mean);
    }
    return div_count_f32(acc, n);
}

float orc_np_std_f32(const float *a, int64_t n) { return sqrtf(orc_np_var_f32(a, n)); }

/* ---------------------------------------------------------------- prune (utility.py:158-163) */

/* thr = std*q (float32 * float32 under numpy 2 / NEP 50) when std_smooth, else q as given
 * (the comparison |w| < q then happens in float32 against the float32-rounded python float:
 * numpy compares float32 array with a weak python scalar in float32). */
int64_t orc_prune_f32(float *w, int64_t n, float q, int std_smooth, uint8_t *mask, float *sigma_out,
                      float *thr_out)
{
    float sigma = 0.0f, thr = q;
    if (std_smooth) {
        sigma = orc_np_std_f32(w, n);
        thr = sigma * q;
    }
    int64_t nz = 0;
    for (int64_t i = 0; i < n; i++) {
        uint8_t m = fabsf(w[i]) < thr;
        mask[i] = m;
        if (m) { w[i] = 0.0f; nz++; }
    }
    if (sigma_out) *sigma_out = sigma;
    if (thr_out) *thr_out = thr;
    return nz;
}

void orc_apply_mask_f32(float *w, const uint8_t *mask, int64_t n)
{
    for (int64_t i = 0; i < n; i++) if (mask[i]) w[i] = 0.0f;
}

/* ---------------------------------------------------------------- CDF pieces (utility.py:362-372) */

void orc_minmax_f32(const float *w, int64_t n, int skip_zeros, float *mn, float *mx, int64_t *count)
{
    float lo = INFINITY, hi = -INFINITY;
    int64_t c = 0;
    for (int64_t i = 0; i < n; i++) {
        if (skip_zeros && w[i] == 0.0f) continue;
        if (w[i] < lo) lo = w[i];
        if (w[i] > hi) hi = w[i];
        c++;
    }
    *mn = lo; *mx = hi; *count = c;
}

/* counts[i] = #{ w : steps[i] <= w < steps[i+1] }, i = 0..30, literally as the reference loops */
void orc_hist31_f32(const float *w, int64_t n, int skip_zeros, const float *steps, int64_t *counts)
{
    for (int b = 0; b < 31; b++) {
        float r1 = steps[b], r2 = steps[b + 1];
        int64_t c = 0;
        for (int64_t i = 0; i < n; i++) {
            if (skip_zeros && w[i] == 0.0f) continue;
            c += (w[i] < r2) & (w[i] >= r1);
        }
        counts[b] = c;
    }
}

/* ---------------------------------------------------------------- Lloyd pieces */

void orc_center_f32(const float *x, int64_t n, float mean, float *xc)
{
    for (int64_t i = 0; i < n; i++) xc[i] = x[i] - mean;
}

/* E-step, brute force over all K centres, exactly sklearn's float32 expression. */
void orc_estep_f32(const float *xc, int64_t n, const float *c, int K, int32_t *labels)
{
    float *csq = (float *)malloc(sizeof(float) * (size_t)K);
    for (int j = 0; j < K; j++) csq[j] = c[j] * c[j];
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; i++) {
        float x = xc[i];
        float best = csq[0] + (-2.0f * (x * c[0]));
        int32_t lab = 0;
        for (int j = 1; j < K; j++) {
            float d = csq[j] + (-2.0f * (x * c[j]));
            if (d < best) { best = d; lab = j; }
        }
        labels[i] = lab;
    }
    free(csq);
}

/* M-step, mode A: what scikit-learn does on ONE thread -- float32 running sums in sample order. */
void orc_mstep_a_f32(const float *xc, int64_t n, const int32_t *labels, int K, float *sums, float *wic)
{
    memset(sums, 0, sizeof(float) * (size_t)K);
    memset(wic, 0, sizeof(float) * (size_t)K);
    for (int64_t i = 0; i < n; i++) {
        int32_t l = labels[i];
        wic[l] += 1.0f;
        sums[l] += xc[i] * 1.0f;
    }
}

/* Fixed-point image of a float32 (shared, bit for bit, with the HIP kernels; see
 * include/nnc.h "fixed-point sums"): q = (int64) rint(v * 2^S), ties to even.  The widening to
 * double and the power-of-two scaling are exact, so this is one correctly rounded operation. */
int64_t orc_fix_f32(float v, int S)
{
    return (int64_t)rint(ldexp((double)v, S)); /* default rounding mode: to nearest even */
}

/* M-step, mode B: exact integer sums of the fixed-point images + integer counts.
 * Order independent, hence identical on any number of threads / GPUs. */
void orc_mstep_b_f32(const float *xc, int64_t n, const int32_t *labels, int K, int S, int64_t *sums,
                     int64_t *counts)
{
    memset(sums, 0, sizeof(int64_t) * (size_t)K);
    memset(counts, 0, sizeof(int64_t) * (size_t)K);
    for (int64_t i = 0; i < n; i++) {
        int32_t l = labels[i];
        counts[l] += 1;
        sums[l] += orc_fix_f32(xc[i], S);
    }
}

/* centre from a fixed-point sum: (float) ldexp((double)sum / (double)count, -S) */
float orc_center_from_fix(int64_t sum, int64_t count, int S)
{
    return (float)ldexp((double)sum / (double)count, -S);
}

/* squared distance of every sample to its own (old) centre, float32:
 * ((X - centers_old[labels])**2).sum(axis=1) with one feature (_k_means_common.pyx:187). */
void orc_dist_own_f32(const float *xc, int64_t n, const float *c, const int32_t *labels, float *d)
{
    for (int64_t i = 0; i < n; i++) {
        float t = xc[i] - c[labels[i]];
        d[i] = t * t;
    }
}

/* cluster_centers_[labels_] (utility.py:239) */
void orc_gather_f32(const float *centers, const int32_t *labels, int64_t n, float *out)
{
    for (int64_t i = 0; i < n; i++) out[i] = centers[labels[i]];
}

void orc_bincount_i32(const int32_t *labels, int64_t n, int K, int64_t *counts)
{
    memset(counts, 0, sizeof(int64_t) * (size_t)K);
    for (int64_t i = 0; i < n; i++) counts[labels[i]]++;
}
