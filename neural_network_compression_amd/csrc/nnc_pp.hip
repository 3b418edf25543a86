// nnc_pp.hip -- k-means++ seeding on the device: the reference's 4th initialisation mode,
// get_quantized_weight(mode="kmeans++") = KMeans(n_clusters=2**bits).fit(...)
// (neural_network_compression/common/utility.py:228-232), i.e. scikit-learn's _kmeans_plusplus
// (sklearn/cluster/_kmeans.py:163-253) on the mean-centred float32 weights with unit sample weights:
//
//   first seed: one uniform draw through the cdf of n equal probabilities (host, closed form);
//   every further seed: T = 2 + int(ln K) candidates drawn with probability proportional to the squared distance
//   to the nearest seed so far (searchsorted of T uniforms * potential in the running sum of those distances),
//   the candidate that lowers the potential most wins.
//
// The uniforms are drawn on the host from NumPy's global generator in scikit-learn's order and uploaded once; the
// device then runs all K - 1 rounds without a host round trip (4 launches per round).  Distances are scikit-learn's
// upcast form for float32 data (metrics/pairwise.py _euclidean_distances_upcast, one feature):
//     d = float32( ((-2 * (c * x)) + c * c) + x * x   in float64 ), clipped at 0.
//
// Summation orders.  scikit-learn takes the potential from a float32 BLAS GEMV (order = the BLAS kernel's) and the
// running sum from a sequential float64 cumsum; neither can be replayed by a parallel machine, and the first is not even
// defined by scikit-learn.  Both are float64 here, in an order fixed by this file (and restated by the oracle):
//   block   = 1024 consecutive samples; lane l of a wave adds elements {256 t + 4 l + u : t, u = 0..3} in that order,
//             the 64 lane sums are combined by an xor butterfly (32, 16, ..., 1);
//   group   = 256 consecutive blocks, block sums added left to right; group totals added left to right;
//   running sum at sample i = (G[group] + W[block]) + (left-to-right sum inside the block up to i).
// The potential differs from scikit-learn's float32 dot in its last bits, so a drawn candidate can come out as a
// neighbouring sample (measured: never below 30 000 samples, 2 of 6 seeds at 235 200; tests/test_oracle.py).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <string>

#include "nnc.h"

int nnc_set_error_(int code, const char *msg); // in nnc_hip.hip

#define PP_BLOCK 1024
#define PP_GROUP 256
#define PP_TMAX 8

struct PpState {
    double pot;          // potential of the seeds so far (float32 value, kept as double)
    int round;           // seeds chosen so far
    int best_t;
    long long cand_id[PP_TMAX];
    float cand_x[PP_TMAX]; // centred values of the candidates
    float seed_x;        // centred value of the seed chosen last (k_pp_update applies it)
};

__device__ __forceinline__ float pp_dist(double c, float x)
{
    const double xd = (double)x;
    double d = -2.0 * (c * xd);
    d += c * c;
    d += xd * xd;
    const float f = (float)d;
    return f > 0.0f ? f : 0.0f; // np.maximum(d, 0)
}

__device__ __forceinline__ double pp_butterfly(double v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = v + __shfl_xor(v, off);
    return v;
}

// element e of a block lives in lane (e % 256) / 4, slot 4 * (e / 256) + e % 4
#define PP_FOR_BLOCK(b, n, body)                                                                                           \
    for (int t_ = 0; t_ < 4; t_++) {                                                                                       \
        const long long i0_ = (long long)(b) * PP_BLOCK + 256 * t_ + 4 * lane;                                             \
        for (int u_ = 0; u_ < 4; u_++) { const long long i = i0_ + u_; const bool in = i < (n); body }                     \
    }

// first seed (round 0) or the seed chosen last: closest = (min of closest and) its distance; block sums
__global__ __launch_bounds__(256) void k_pp_update(const float *__restrict__ x, long long n, float mean, const PpState *__restrict__ st,
                                                   int first, float *__restrict__ closest, double *__restrict__ S)
{
    const int lane = threadIdx.x & 63;
    const long long nblk = (n + PP_BLOCK - 1) / PP_BLOCK;
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long long)gridDim.x * 4;
    const double c = (double)st->seed_x;
    for (long long b = wave; b < nblk; b += nwaves) {
        double acc = 0.0;
        PP_FOR_BLOCK(b, n, {
            if (in) {
                float d = pp_dist(c, x[i] - mean);
                if (!first) d = fminf(closest[i], d);
                closest[i] = d;
                acc = acc + (double)d;
            }
        })
        acc = pp_butterfly(acc);
        if (S && lane == 0) S[b] = acc;
    }
}

// block sums of min(closest, d(candidate t, .)) for the T candidates of this round
__global__ __launch_bounds__(256) void k_pp_pots(const float *__restrict__ x, long long n, float mean, const PpState *__restrict__ st, int T,
                                                 const float *__restrict__ closest, double *__restrict__ St, long long nblk_stride)
{
    const int lane = threadIdx.x & 63;
    const long long nblk = (n + PP_BLOCK - 1) / PP_BLOCK;
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long long)gridDim.x * 4;
    double c[PP_TMAX];
#pragma unroll
    for (int t = 0; t < PP_TMAX; t++) c[t] = t < T ? (double)st->cand_x[t] : 0.0;
    for (long long b = wave; b < nblk; b += nwaves) {
        double acc[PP_TMAX];
#pragma unroll
        for (int t = 0; t < PP_TMAX; t++) acc[t] = 0.0;
        PP_FOR_BLOCK(b, n, {
            if (in) {
                const float xc = x[i] - mean;
                const float cl = closest[i];
                _Pragma("unroll") for (int t = 0; t < PP_TMAX; t++)
                    if (t < T) acc[t] = acc[t] + (double)fminf(cl, pp_dist(c[t], xc));
            }
        })
#pragma unroll
        for (int t = 0; t < PP_TMAX; t++)
            if (t < T) {
                const double s = pp_butterfly(acc[t]);
                if (lane == 0) St[(long long)t * nblk_stride + b] = s;
            }
    }
}

// Left-to-right sums inside every group of PP_GROUP block sums: one wave per (candidate slot, group).  The lanes fetch the
// group's block sums together, lane 0 runs the chain out of LDS.  W[b] = sum of the group's blocks before b; gtot = group total.
__global__ __launch_bounds__(256) void k_pp_groups(const double *__restrict__ St, double *__restrict__ Wt, double *__restrict__ gtot,
                                                   long long nblk, long long stride, long long gstride, int T)
{
    __shared__ double buf[4][PP_GROUP + 1];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long long ngroups = (nblk + PP_GROUP - 1) / PP_GROUP;
    const long long item = (long long)blockIdx.x * 4 + wv;
    if (item >= ngroups * T) return;
    const int t = (int)(item / ngroups);
    const long long g = item % ngroups;
    const double *S = St + (long long)t * stride;
    double *W = Wt + (long long)t * stride;
    const long long b0 = g * PP_GROUP;
    const int cnt = (int)((b0 + PP_GROUP < nblk ? b0 + PP_GROUP : nblk) - b0);
    for (int q = lane; q < cnt; q += 64) buf[wv][q] = S[b0 + q];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (lane == 0) {
        double acc = 0.0;
        for (int q = 0; q < cnt; q++) { const double v = buf[wv][q]; buf[wv][q] = acc; acc = acc + v; }
        gtot[(long long)t * gstride + g] = acc;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (int q = lane; q < cnt; q += 64) W[b0 + q] = buf[wv][q];
}

// Left-to-right sums of the group totals for every slot (G[g] = sum of the groups before g, G[ngroups] = total), the
// potentials (float32), the winner (np.argmin: first minimum), the new state.  first != 0: slot 0 holds the first seed.
__global__ __launch_bounds__(64) void k_pp_pick(const double *__restrict__ gtot, double *__restrict__ Gt, long long ngroups, long long gstride,
                                                int T, int first, PpState *__restrict__ st, float *__restrict__ seeds, long long *__restrict__ seed_ids)
{
    __shared__ float pots[PP_TMAX];
    const int t = threadIdx.x;
    if (t < T) {
        const double *gt = gtot + (long long)t * gstride;
        double *G = Gt + (long long)t * gstride;
        double acc = 0.0;
        for (long long g = 0; g < ngroups; g++) { G[g] = acc; acc = acc + gt[g]; }
        G[ngroups] = acc;
        pots[t] = (float)acc;
    }
    __syncthreads();
    if (t == 0) {
        int best = 0;
        for (int q = 1; q < T; q++) if (pots[q] < pots[best]) best = q;
        st->pot = (double)pots[best];
        st->best_t = best;
        if (!first) {
            const int round = st->round;
            st->seed_x = st->cand_x[best];
            seeds[round] = st->cand_x[best];
            seed_ids[round] = st->cand_id[best];
            st->round = round + 1;
        }
    }
}

// the T candidates of a round: searchsorted(running sum of closest, u[t] * pot), clipped to n - 1.  One wave per candidate.
__global__ __launch_bounds__(64 * PP_TMAX) void k_pp_select(const float *__restrict__ x, long long n, float mean, const float *__restrict__ closest,
                                                           const double *__restrict__ St, const double *__restrict__ Wt, const double *__restrict__ Gt,
                                                           long long stride, long long gstride, const double *__restrict__ uniforms, int T,
                                                           PpState *__restrict__ st)
{
    __shared__ float blk[PP_TMAX][PP_BLOCK];
    const int lane = threadIdx.x & 63, t = threadIdx.x >> 6;
    if (t >= T) return;
    const long long nblk = (n + PP_BLOCK - 1) / PP_BLOCK, ngroups = (nblk + PP_GROUP - 1) / PP_GROUP;
    const int cur = st->best_t; // the slot whose sums describe `closest`
    const double *S = St + (long long)cur * stride, *W = Wt + (long long)cur * stride, *G = Gt + (long long)cur * gstride;
    const double r = uniforms[(long long)(st->round - 1) * T + t] * st->pot;
    // group: first g with G[g + 1] >= r, else the last
    long long g = ngroups - 1;
    for (long long q0 = 0; q0 < ngroups; q0 += 64) {
        const long long q = q0 + lane;
        const unsigned long long bal = __ballot(q < ngroups && G[q + 1] >= r);
        if (bal) { g = q0 + __ffsll((long long)bal) - 1; break; }
    }
    // block inside the group: first b with G[g] + (W[b] + S[b]) >= r, else the group's last
    const long long b0 = g * PP_GROUP, b1 = b0 + PP_GROUP < nblk ? b0 + PP_GROUP : nblk;
    const double Gg = G[g];
    long long b = b1 - 1;
    for (long long q0 = b0; q0 < b1; q0 += 64) {
        const long long q = q0 + lane;
        const unsigned long long bal = __ballot(q < b1 && Gg + (W[q] + S[q]) >= r);
        if (bal) { b = q0 + __ffsll((long long)bal) - 1; break; }
    }
    // sample inside the block: first i with (G[g] + W[b]) + (left-to-right sum up to i) >= r, else the block's last
    const double base = Gg + W[b];
    const long long i0 = b * PP_BLOCK;
    const int cnt = (int)((i0 + PP_BLOCK < n ? i0 + PP_BLOCK : n) - i0);
    for (int q = lane; q < cnt; q += 64) blk[t][q] = closest[i0 + q];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (lane == 0) {
        int id = cnt - 1;
        double run = 0.0;
        for (int q = 0; q < cnt; q++) { run = run + (double)blk[t][q]; if (base + run >= r) { id = q; break; } }
        st->cand_id[t] = i0 + id;
        st->cand_x[t] = x[i0 + id] - mean;
    }
}

__global__ void k_pp_first(const float *__restrict__ x, long long n, float mean, long long id0, PpState *__restrict__ st,
                           float *__restrict__ seeds, long long *__restrict__ seed_ids)
{
    const float v = x[id0] - mean;
    st->seed_x = v; st->round = 1; st->best_t = 0; st->pot = 0.0;
    seeds[0] = v; seed_ids[0] = id0;
}

static size_t pp_align(size_t b) { return (b + 255) & ~(size_t)255; }

extern "C" int32_t nnc_kmeanspp_trials(int32_t k) { return k >= 1 ? 2 + (int32_t)std::log((double)k) : 0; }

extern "C" size_t nnc_kmeanspp_workspace_bytes(int64_t n, int32_t k)
{
    if (n <= 0 || k < 1) return 0;
    const size_t nblk = (size_t)((n + PP_BLOCK - 1) / PP_BLOCK), ngroups = (nblk + PP_GROUP - 1) / PP_GROUP;
    return pp_align(sizeof(PpState)) + pp_align((size_t)n * 4) + 2 * pp_align(nblk * 8) * PP_TMAX + 2 * pp_align((ngroups + 1) * 8) * PP_TMAX;
}

// seeds_out_dev[k]: the centred seeds (x[id] - x_mean, float32) in the order chosen; seed_ids_out_dev[k]: their sample indices.
extern "C" int nnc_kmeanspp_seed_f32(const float *x, int64_t n, float x_mean, int32_t k, int64_t first_id, const double *uniforms_dev,
                                     float *seeds_out_dev, int64_t *seed_ids_out_dev, void *ws, size_t ws_bytes, void *stream)
{
    if (!x || n < 1 || k < 1 || k > NNC_KMAX || !seeds_out_dev || !seed_ids_out_dev || !ws || first_id < 0 || first_id >= n || (k > 1 && !uniforms_dev))
        return nnc_set_error_(NNC_EINVAL, "nnc_kmeanspp_seed_f32: bad argument");
    if (ws_bytes < nnc_kmeanspp_workspace_bytes(n, k)) return nnc_set_error_(NNC_ENOSPACE, "nnc_kmeanspp_seed_f32: workspace too small");
    if ((reinterpret_cast<uintptr_t>(ws) & 255) != 0) return nnc_set_error_(NNC_EINVAL, "nnc_kmeanspp_seed_f32: workspace must be 256-byte aligned");
    const long long nblk = (n + PP_BLOCK - 1) / PP_BLOCK, ngroups = (nblk + PP_GROUP - 1) / PP_GROUP;
    const int T = nnc_kmeanspp_trials(k);
    if (T > PP_TMAX) return nnc_set_error_(NNC_EINVAL, "nnc_kmeanspp_seed_f32: too many local trials");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    unsigned char *b = reinterpret_cast<unsigned char *>(ws);
    PpState *st = reinterpret_cast<PpState *>(b); b += pp_align(sizeof(PpState));
    float *closest = reinterpret_cast<float *>(b); b += pp_align((size_t)n * 4);
    const long long stride = (long long)(pp_align((size_t)nblk * 8) / 8), gstride = (long long)(pp_align((size_t)(ngroups + 1) * 8) / 8);
    double *St = reinterpret_cast<double *>(b); b += (size_t)stride * 8 * PP_TMAX;
    double *Wt = reinterpret_cast<double *>(b); b += (size_t)stride * 8 * PP_TMAX;
    double *gtot = reinterpret_cast<double *>(b); b += (size_t)gstride * 8 * PP_TMAX;
    double *Gt = reinterpret_cast<double *>(b);
    int cus = 256;
    nnc_device_info(nullptr, 0, &cus);
    const int grid = (int)std::max<long long>(1, std::min<long long>((nblk + 3) / 4, (long long)cus * 8));
#define PPCHK(name) do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return nnc_set_error_(NNC_EHIP, (std::string("launch ") + name + ": " + hipGetErrorString(e_)).c_str()); } while (0)
    hipLaunchKernelGGL(k_pp_first, dim3(1), dim3(1), 0, s, x, (long long)n, x_mean, (long long)first_id, st, seeds_out_dev, reinterpret_cast<long long *>(seed_ids_out_dev));
    PPCHK("k_pp_first");
    hipLaunchKernelGGL(k_pp_update, dim3(grid), dim3(256), 0, s, x, (long long)n, x_mean, st, 1, closest, St); // slot 0
    PPCHK("k_pp_update");
    hipLaunchKernelGGL(k_pp_groups, dim3((unsigned)((ngroups + 3) / 4)), dim3(256), 0, s, St, Wt, gtot, nblk, stride, gstride, 1);
    PPCHK("k_pp_groups");
    hipLaunchKernelGGL(k_pp_pick, dim3(1), dim3(64), 0, s, gtot, Gt, ngroups, gstride, 1, 1, st, seeds_out_dev, reinterpret_cast<long long *>(seed_ids_out_dev));
    PPCHK("k_pp_pick");
    for (int c = 1; c < k; c++) {
        hipLaunchKernelGGL(k_pp_select, dim3(1), dim3(64 * PP_TMAX), 0, s, x, (long long)n, x_mean, closest, St, Wt, Gt, stride, gstride, uniforms_dev, T, st);
        PPCHK("k_pp_select");
        hipLaunchKernelGGL(k_pp_pots, dim3(grid), dim3(256), 0, s, x, (long long)n, x_mean, st, T, closest, St, stride);
        PPCHK("k_pp_pots");
        hipLaunchKernelGGL(k_pp_groups, dim3((unsigned)((ngroups * T + 3) / 4)), dim3(256), 0, s, St, Wt, gtot, nblk, stride, gstride, T);
        PPCHK("k_pp_groups");
        hipLaunchKernelGGL(k_pp_pick, dim3(1), dim3(64), 0, s, gtot, Gt, ngroups, gstride, T, 0, st, seeds_out_dev, reinterpret_cast<long long *>(seed_ids_out_dev));
        PPCHK("k_pp_pick");
        if (c + 1 < k) { // the last seed's distances are nobody's business
            hipLaunchKernelGGL(k_pp_update, dim3(grid), dim3(256), 0, s, x, (long long)n, x_mean, st, 0, closest, (double *)nullptr);
            PPCHK("k_pp_update");
        }
    }
#undef PPCHK
    return NNC_OK;
}

// ======================================================================================
// Centroid fine-tuning: dL/dC_k = sum over the weights of cluster k of dL/dW  (Deep Compression's "trained quantization";
// the reference describes it and leaves it out because scanning all gradients for every batch on the host was too slow,
// papers/lat/report.tex:149-158).  A segmented reduction by centroid index: the same shape as the M-step, on the original
// (unsorted) order, so the sums go through LDS accumulators.  Gradients enter as fixed-point images rint(g * 2^S)
// (S = nnc_fix_shift(max |g|, n)), so the result is independent of summation order and of the number of GPUs.
// Also the decode step itself, cluster_centers_[labels_] (utility.py:239), for writing fine-tuned centroids back.
// ======================================================================================
template <typename LT>
__global__ __launch_bounds__(256) void k_centroid_grad(const float *__restrict__ g, const LT *__restrict__ labels, long long n, int k, int Sft,
                                                       int rlog2, unsigned long long *__restrict__ sums, unsigned long long *__restrict__ counts)
{
    extern __shared__ unsigned long long acc[]; // [k][R] sums, then [k][R] counts (as 32-bit)
    const int R = 1 << rlog2;
    unsigned *cnt = reinterpret_cast<unsigned *>(acc + ((size_t)k << rlog2));
    for (int i = threadIdx.x; i < (k << rlog2); i += 256) { acc[i] = 0ull; cnt[i] = 0u; }
    __syncthreads();
    const int rep = threadIdx.x & (R - 1);
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x, nthreads = (long long)gridDim.x * blockDim.x;
    for (long long i = tid; i < n; i += nthreads) {
        const int l = (int)labels[i];
        if (l < k) {
            const long long q = (long long)(int)rintf(ldexpf(g[i], Sft));
            atomicAdd(&acc[(l << rlog2) + rep], (unsigned long long)q);
            atomicAdd(&cnt[(l << rlog2) + rep], 1u);
        }
    }
    __syncthreads();
    for (int j = threadIdx.x; j < k; j += 256) {
        unsigned long long s = 0, c = 0;
        for (int r = 0; r < R; r++) { s += acc[(j << rlog2) + r]; c += cnt[(j << rlog2) + r]; }
        if (c) { atomicAdd(&sums[j], s); if (counts) atomicAdd(&counts[j], c); }
    }
}

extern "C" int nnc_centroid_grad_f32(const float *grad, const void *labels, int label_bytes, int64_t n, int32_t k, int32_t fix_shift,
                                     int64_t *sums_dev, int64_t *counts_dev, void *stream)
{
    if (n < 0 || k < 1 || k > NNC_KMAX || !sums_dev || (n > 0 && (!grad || !labels)) || (label_bytes != 1 && label_bytes != 2))
        return nnc_set_error_(NNC_EINVAL, "nnc_centroid_grad_f32: bad argument");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (hipMemsetAsync(sums_dev, 0, (size_t)k * 8, s) != hipSuccess) return nnc_set_error_(NNC_EHIP, "nnc_centroid_grad_f32: memset");
    if (counts_dev && hipMemsetAsync(counts_dev, 0, (size_t)k * 8, s) != hipSuccess) return nnc_set_error_(NNC_EHIP, "nnc_centroid_grad_f32: memset");
    if (n == 0) return NNC_OK;
    int rlog2 = k <= 64 ? 5 : (k <= 256 ? 3 : 1);
    int cus = 256;
    nnc_device_info(nullptr, 0, &cus);
    const int grid = (int)std::max<long long>(1, std::min<long long>((n + 2047) / 2048, (long long)cus * 4));
    const size_t lds = ((size_t)k << rlog2) * 12;
    if (label_bytes == 1)
        hipLaunchKernelGGL((k_centroid_grad<uint8_t>), dim3(grid), dim3(256), lds, s, grad, reinterpret_cast<const uint8_t *>(labels), (long long)n, (int)k,
                           (int)fix_shift, rlog2, reinterpret_cast<unsigned long long *>(sums_dev), reinterpret_cast<unsigned long long *>(counts_dev));
    else
        hipLaunchKernelGGL((k_centroid_grad<uint16_t>), dim3(grid), dim3(256), lds, s, grad, reinterpret_cast<const uint16_t *>(labels), (long long)n, (int)k,
                           (int)fix_shift, rlog2, reinterpret_cast<unsigned long long *>(sums_dev), reinterpret_cast<unsigned long long *>(counts_dev));
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return nnc_set_error_(NNC_EHIP, hipGetErrorString(e));
    return NNC_OK;
}

template <typename LT>
__global__ __launch_bounds__(256) void k_gather(const float *__restrict__ centers, int k, const LT *__restrict__ labels, long long n, float *__restrict__ out)
{
    extern __shared__ float cs[];
    for (int j = threadIdx.x; j < k; j += 256) cs[j] = centers[j];
    __syncthreads();
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x, nthreads = (long long)gridDim.x * blockDim.x;
    for (long long i = tid; i < n; i += nthreads) { const int l = (int)labels[i]; out[i] = l < k ? cs[l] : 0.0f; }
}

extern "C" int nnc_gather_f32(const float *centers_dev, int32_t k, const void *labels, int label_bytes, int64_t n, float *out, void *stream)
{
    if (n < 0 || k < 1 || k > NNC_KMAX || !centers_dev || (n > 0 && (!labels || !out)) || (label_bytes != 1 && label_bytes != 2))
        return nnc_set_error_(NNC_EINVAL, "nnc_gather_f32: bad argument");
    if (n == 0) return NNC_OK;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    int cus = 256;
    nnc_device_info(nullptr, 0, &cus);
    const int grid = (int)std::max<long long>(1, std::min<long long>((n + 1023) / 1024, (long long)cus * 8));
    if (label_bytes == 1) hipLaunchKernelGGL((k_gather<uint8_t>), dim3(grid), dim3(256), (size_t)k * 4, s, centers_dev, (int)k, reinterpret_cast<const uint8_t *>(labels), (long long)n, out);
    else hipLaunchKernelGGL((k_gather<uint16_t>), dim3(grid), dim3(256), (size_t)k * 4, s, centers_dev, (int)k, reinterpret_cast<const uint16_t *>(labels), (long long)n, out);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return nnc_set_error_(NNC_EHIP, hipGetErrorString(e));
    return NNC_OK;
}
