// nnc_hip.hip -- gfx950 (MI355X, CDNA4) kernels + C ABI (include/nnc.h) for the
// prune -> k-means -> index re-encode path of neural-network-compression.
//
// Everything here is HBM-bound byte/float streaming work (no dense contraction, hence no
// MFMA): wave64 shuffles for reductions, LDS for the centroid search table and the
// privatised per-cluster accumulators, 16-byte coalesced global loads, one workgroup per CU.
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared  (see build.py).
// -ffp-contract=off is REQUIRED: mask bits and centroid indices must reproduce the
// reference's unfused float32 arithmetic (NumPy / scikit-learn on x86).
#include "nnc_common.hpp"
#include "nnc_km_shared.hpp"

// ======================================================================================
// 4. Lloyd k-means
// ======================================================================================
//
// E-step.  scikit-learn labels a sample with the FIRST strict minimum over j of
//     d_j = fl( fl(c_j*c_j) + fl(-2 * fl(x*c_j)) )        (float32, unfused)
// on mean-centred x, c.  Evaluating all K of them is ~3 flop x K per weight: compute bound
// at K = 256 (ten times the HBM time).  In one dimension the winner is the nearest centre
// except within rounding distance of a midpoint, so each iteration a tiny kernel sorts the
// centres and builds a uniform grid over [lo, hi] whose cells hold the contiguous range of
// (sorted) centres that can possibly win for ANY x in the cell, using a rigorous bound
//     |d_j(float32) - d_j(exact)| <= E = 2.5 * 2^-24 * (c^2 + 2 |x|max |c|)
// Most cells hold one candidate (label known from the table alone); boundary cells hold two
// or more and those are evaluated with the exact float32 expression above, ties to the
// lowest original index.  Result: identical labels to the brute-force scan, ~1 LDS lookup
// per weight.
//
// M-step.  Each weight adds fix(x~) (64-bit fixed point, see nnc.h) and 1 to LDS accumulators
// of its (sorted) cluster; accumulators are replicated R times ([cluster][replica], replica =
// lane mod R, so the lanes of a 32-lane group hit 32 different banks and never the same
// address when R = 32).  Workgroups flush to NSHARD global shards with 64-bit atomics;
// integer addition makes the result independent of any ordering.
//
// (constants, KmTab, KmWs: nnc_km_shared.hpp)

extern "C" size_t nnc_kmeans_workspace_bytes(int32_t k)
{
    (void)k;
    return sizeof(KmWs) + 256;
}

extern "C" int32_t nnc_fix_shift(float absmax, int64_t n_total)
{
    int L = 1;
    while (((int64_t)1 << L) < n_total && L < 62) L++;
    if (!(absmax > 0.0f) || !std::isfinite(absmax)) return 0;
    int P = 0;
    (void)std::frexp((double)absmax, &P); // absmax = m * 2^P, m in [0.5, 1)
    // |fix| <= 2^min(28, 62-L): one image is a 32-bit integer (float32 scaling + v_cvt), four of
    // them add up inside int32, and sums of n_total images fit int64
    return std::min(28, 62 - L) - P;
}

// fix(v) = rint(v * 2^S), ties to even, as a 32-bit integer: the scaling by a power of two is
// exact in float32 (v_ldexp_f32) and S is chosen so that |v * 2^S| < 2^28.
__device__ __forceinline__ int fix_f32(float v, int Sft)
{
    return (int)rintf(ldexpf(v, Sft));
}

static void km_defaults(const nnc_kmeans_params *p, int *glog2, int *rlog2)
{
    int r = p->replicas_log2;
    if (r < 0) {
        // LDS accumulator replicas: with run accumulation on a sorted vector LDS atomics are rare,
        // so a few replicas suffice; small K (few clusters, long equal-index runs broken only by
        // unsorted small tensors) gets the full 32
        r = p->k <= 64 ? 5 : 2;
        while (r > 0 && (size_t)p->k * ((size_t)1 << r) * 12 > 40 * 1024) r--;
    }
    if (r > 5) r = 5;
    int g = p->grid_log2;
    if (g <= 0) g = p->k <= 32 ? KM_FUSE_GLOG2 - 1 : (p->k <= KM_FUSE_KMAX ? KM_FUSE_GLOG2 : 14); // few centres: a coarse grid leaves next to nothing to the general path
    if (g < 6) g = 6;
    if (g > 15) g = 15;
    // keep table + accumulators + candidates inside 156 KiB
    while (g > 6 && ((size_t)2 << g) + (size_t)p->k * ((size_t)1 << r) * 12 + (size_t)(p->k + 8) * 22 + KM_OVF_MAX * 4 + 64 > 156 * 1024) g--;
    *glog2 = g;
    *rlog2 = r;
}

static size_t km_lds_bytes(int k, int glog2, int rlog2, bool accumulate)
{
    size_t kp = (size_t)((k + 7) & ~7);
    size_t b = ((size_t)2 << glog2) + kp * 16 + kp * 4 + kp * 2;
    b = (b + 15) & ~(size_t)15;
    b += KM_OVF_MAX * 4;
    if (accumulate) b += (size_t)k * ((size_t)1 << rlog2) * 12;
    return b;
}

// ---- the streaming kernel ------------------------------------------------------------
// MODE 0: E-step + accumulate (Lloyd iteration).  MODE 1: E-step + write labels / values / distances.
// Diagnostics (phase timestamps, ablated kernels, tuning knobs read from the environment) exist only in a build with
// -DNNC_DIAG (NNC_DIAG=1 python -m neural_network_compression_amd.build -> libnnc_hip_diag.so, used by tools/).
#ifdef NNC_DIAG
__device__ unsigned long long *g_fin_trace = nullptr; // phase timestamps of k_finalize (thread 0)
__device__ unsigned long long *g_km_trace = nullptr; // per-workgroup {t_start, t_loop, t_epilogue, t_end} in 100 MHz ticks
#define NNC_KM_TRACE_PTR g_km_trace
#define NNC_FIN_TRACE_PTR g_fin_trace
#else
#define NNC_KM_TRACE_PTR ((unsigned long long *)nullptr)
#define NNC_FIN_TRACE_PTR ((unsigned long long *)nullptr)
#endif

struct KmCtx {
    const uint16_t *cell_s;  // u16[G]: p_lo | (min(cnt-1, 31) << 11)
    const float4 *pair_s;    // (c_p, c_p^2, c_{p+1}, c_{p+1}^2), sorted order
    const float *cval_s;     // c_p (sorted order)
    const uint16_t *orig_s;  // sorted position -> original index
    const unsigned *ovf_s;   // crowded cells: first | last << 16
    unsigned long long *sum_s; // [cluster][replica] fixed-point sums
    unsigned *cnt_s;           // [cluster][replica] counts
    float mean, lo, inv;
    int Sft, gmax, k, rlog2, rep;
};

// three or more candidates, or an exact float32 tie between two: the general scan
__device__ int km_find_slow(const KmCtx &c, float xc, int p, int cnt)
{
    if (cnt == KM_CNT_SAT) { // crowded cell: the entry holds an index into the side list, not a position
        if ((unsigned)p == KM_OVF_ALL) { p = 0; cnt = c.k - 1; }
        else { const unsigned u = c.ovf_s[p]; p = (int)(u & 0xFFFFu); cnt = (int)(u >> 16) - p; }
    }
    float4 cc = c.pair_s[p];
    float bestd = cc.y + (-2.0f * (xc * cc.x));
    int best = p;
    for (int i = 1; i <= cnt; i++) {
        float4 ci = c.pair_s[p + i];
        float d = ci.y + (-2.0f * (xc * ci.x));
        if (d < bestd || (d == bestd && c.orig_s[p + i] < c.orig_s[best])) { bestd = d; best = p + i; }
    }
    return best;
}

// B weights of one thread: all table reads, then all pair reads, then the arithmetic, so that B
// LDS round trips are in flight together.  No branches on the common path.
template <int B>
__device__ __forceinline__ void km_resolve(const KmCtx &c, const float (&xv)[B], float (&xc)[B], int (&p)[B])
{
    unsigned e[B];
    float4 pr[B];
#pragma unroll
    for (int i = 0; i < B; i++) {
        xc[i] = xv[i] - c.mean;
        float tt = (xc[i] - c.lo) * c.inv;
        int cell = (int)tt;
        cell = min(max(cell, 0), c.gmax);
        e[i] = c.cell_s[cell];
    }
    // the pair is needed only where a cell holds more than one candidate; every other lane reads
    // entry 0 (one address, broadcast), so the read costs no bank conflicts and no branch
#pragma unroll
    for (int i = 0; i < B; i++) pr[i] = c.pair_s[((e[i] >> KM_P_BITS) == 1) ? (e[i] & KM_P_MASK) : 0];
    bool slow_any = false;
#pragma unroll
    for (int i = 0; i < B; i++) {
        const unsigned cn = e[i] >> KM_P_BITS;
        const float d0 = pr[i].y + (-2.0f * (xc[i] * pr[i].x));
        const float d1 = pr[i].w + (-2.0f * (xc[i] * pr[i].z));
        p[i] = (int)(e[i] & KM_P_MASK) + (int)((cn == 1) & (d1 < d0));
        slow_any |= (cn >= 2) | ((cn == 1) & (d1 == d0));
    }
    if (slow_any) {
#pragma unroll
        for (int i = 0; i < B; i++) {
            const unsigned cn = e[i] >> KM_P_BITS;
            const float d0 = pr[i].y + (-2.0f * (xc[i] * pr[i].x));
            const float d1 = pr[i].w + (-2.0f * (xc[i] * pr[i].z));
            if (cn >= 2 || (cn == 1 && d1 == d0)) p[i] = km_find_slow(c, xc[i], (int)(e[i] & KM_P_MASK), (int)cn);
        }
    }
}

// A lane adds up runs of equal cluster index in registers and touches LDS only when the index
// changes: on a value-sorted vector (nnc_sort_f32) that is a handful of LDS atomics per lane per
// launch instead of one per weight (LDS atomics retire about one lane per clock on gfx950, which
// made them the whole cost of the unsorted form).  Any order gives the same sums.
struct KmRun { int p; unsigned cnt; long long sum; };

__device__ __forceinline__ void km_run_flush(const KmCtx &c, KmRun &run)
{
    if (run.cnt) {
        atomicAdd(&c.sum_s[(run.p << c.rlog2) + c.rep], (unsigned long long)run.sum);
        atomicAdd(&c.cnt_s[(run.p << c.rlog2) + c.rep], run.cnt);
    }
    run.cnt = 0;
    run.sum = 0;
}

__device__ __forceinline__ int km_cell(const KmCtx &c, float xc)
{
    const float tt = (xc - c.lo) * c.inv;
    return min(max((int)tt, 0), c.gmax);
}

// ABL (timing experiments only, results are wrong unless 0): 1 = no LDS atomics, 2 = no table
// lookups (cluster from the value bits), 3 = neither (pure stream + arithmetic)
template <int B, int ABL>
__device__ __forceinline__ void km_accumulate(const KmCtx &c, const float (&xv)[B], KmRun &run)
{
    float xc[B];
    int p[B];
    if (ABL == 2 || ABL == 3) {
#pragma unroll
        for (int i = 0; i < B; i++) { xc[i] = xv[i] - c.mean; p[i] = (int)(__float_as_uint(xv[i]) >> 5) & 255; if (p[i] >= c.k) p[i] = 0; }
    } else {
        km_resolve<B>(c, xv, xc, p);
    }
#pragma unroll
    for (int i = 0; i < B; i++) {
        const int q = fix_f32(xc[i], c.Sft);
        if (p[i] != run.p) {
            if (ABL != 1 && ABL != 3) km_run_flush(c, run);
            run.p = p[i];
        }
        run.cnt++;
        run.sum += q;
    }
}

// Four consecutive weights of one lane.  Fast path: if the cells of the smallest and the largest
// of the four hold the same single candidate, every cell between them does too (the candidate
// ranges are monotone in the cell index), so all four weights belong to that cluster: two table
// reads and no distance arithmetic.  On a value-sorted vector nearly every float4 takes it.
template <int ABL>
__device__ __forceinline__ void km_accumulate4(const KmCtx &c, const float4 v, KmRun &run)
{
    const float x0 = v.x - c.mean, x1 = v.y - c.mean, x2 = v.z - c.mean, x3 = v.w - c.mean;
    const float mn = fminf(fminf(x0, x1), fminf(x2, x3));
    const float mx = fmaxf(fmaxf(x0, x1), fmaxf(x2, x3));
    const unsigned el = c.cell_s[km_cell(c, mn)];
    const unsigned eh = c.cell_s[km_cell(c, mx)];
    if (ABL == 0 && el == eh && (el >> KM_P_BITS) == 0) {
        const int p = (int)el;
        // |fix| <= 2^28, so four of them add up inside int32
        const int q = (fix_f32(x0, c.Sft) + fix_f32(x1, c.Sft)) + (fix_f32(x2, c.Sft) + fix_f32(x3, c.Sft));
        if (p != run.p) { km_run_flush(c, run); run.p = p; }
        run.cnt += 4;
        run.sum += q;
    } else {
        const float xv[4] = {v.x, v.y, v.z, v.w};
        km_accumulate<4, ABL>(c, xv, run);
    }
}

// add this workgroup's LDS accumulators to its global shard
__device__ __forceinline__ void km_flush(const KmCtx &c, KmWs *ws)
{
    __syncthreads();
    const int R = 1 << c.rlog2;
    const int shard = blockIdx.x & (KM_NSHARD - 1);
    for (int p = threadIdx.x; p < c.k; p += KM_THREADS) {
        unsigned long long s = 0, n = 0;
        for (int r = 0; r < R; r++) { s += c.sum_s[(p << c.rlog2) + r]; n += c.cnt_s[(p << c.rlog2) + r]; }
        if (n) {
            atomicAdd(reinterpret_cast<unsigned long long *>(&ws->shard_sum[shard][p]), s);
            atomicAdd(&ws->shard_cnt[shard][p], n);
        }
    }
}

struct KmHistRun { unsigned bin, cnt; };

__device__ __forceinline__ void km_hist_add(unsigned *hist_s, KmHistRun &hr, float d)
{
    const unsigned b = (__float_as_uint(d) >> 19) & 4095u; // top 12 value bits of a non-negative float
    if (b != hr.bin) { if (hr.cnt) atomicAdd(&hist_s[hr.bin], hr.cnt); hr.bin = b; hr.cnt = 0; }
    hr.cnt++;
}

template <int B, typename LT>
__device__ __forceinline__ void km_emit(const KmCtx &c, const float (&xv)[B], int64_t i0, LT *__restrict__ labels_out,
                                        float *__restrict__ quant_out, float *__restrict__ dist_out,
                                        unsigned *hist_s, KmHistRun &hr)
{
    float xc[B];
    int p[B];
    km_resolve<B>(c, xv, xc, p);
    LT lab[B];
    float qv[B], dv[B];
#pragma unroll
    for (int i = 0; i < B; i++) {
        const float cv = c.cval_s[p[i]];
        lab[i] = (LT)c.orig_s[p[i]];
        qv[i] = cv + c.mean;
        const float dd = xc[i] - cv;
        dv[i] = dd * dd;
        if (hist_s) km_hist_add(hist_s, hr, dv[i]);
    }
    if (B == 4) {
        if (labels_out) {
            if (sizeof(LT) == 1) *reinterpret_cast<uchar4 *>(labels_out + i0) = make_uchar4(lab[0], lab[1], lab[2], lab[3]);
            else *reinterpret_cast<ushort4 *>(labels_out + i0) = make_ushort4(lab[0], lab[1], lab[2], lab[3]);
        }
        if (quant_out) *reinterpret_cast<float4 *>(quant_out + i0) = make_float4(qv[0], qv[1], qv[2], qv[3]);
        if (dist_out) *reinterpret_cast<float4 *>(dist_out + i0) = make_float4(dv[0], dv[1], dv[2], dv[3]);
    } else {
#pragma unroll
        for (int i = 0; i < B; i++) {
            if (labels_out) labels_out[i0 + i] = lab[i];
            if (quant_out) quant_out[i0 + i] = qv[i];
            if (dist_out) dist_out[i0 + i] = dv[i];
        }
    }
}

// MODE 0: E-step + accumulate (Lloyd iteration).  MODE 1: E-step + write labels / values / distances.
// Work split: tiles of 2 * KM_THREADS float4 (8192 weights); tile t belongs to workgroup t mod grid.
template <int MODE, bool VEC, typename LT, int ABL = 0, bool DEAL = false>
__global__ __launch_bounds__(KM_THREADS, 8) void k_assign(const float *__restrict__ x, int64_t n, KmWs *__restrict__ ws,
                                                       int which, LT *__restrict__ labels_out,
                                                       float *__restrict__ quant_out, float *__restrict__ dist_out,
                                                       unsigned long long *__restrict__ dist_hist = nullptr,
                                                       const int *__restrict__ n_dev = nullptr)
{
    extern __shared__ __align__(16) unsigned char smem[];
    if (MODE == 1 && n_dev) n = *n_dev; // label mode only: the real length lives on the device (the grid was sized for a bound)
    unsigned long long *trace = (MODE == 0) ? NNC_KM_TRACE_PTR : nullptr;
    unsigned long long tr0 = 0, tr1 = 0, tr2 = 0;
    if (trace) tr0 = __builtin_amdgcn_s_memrealtime();

    // work split: steps of KM_THREADS float4 (4096 weights); every workgroup takes a CONTIGUOUS
    // range of steps, so that on a sorted vector it stays inside a few clusters.  The first loads
    // depend on the kernel arguments only and go out before anything else (the state block and the
    // tables come from L2 while HBM is already streaming).
    const int64_t nvec = VEC ? (n >> 2) : 0;
    const int64_t nsteps = nvec / KM_THREADS;
    // DEAL (accumulate on a value-sorted vector): the outermost grid/2 steps at either end are dealt one to a workgroup,
    // the rest is split into contiguous ranges.  The tails are where clusters are narrowest -- right after a mass
    // relocation dozens of one-sample clusters sit there and every weight takes the general path -- so no single
    // workgroup should own a tail.  (The host picks DEAL only if nsteps >= 4 * grid and the grid is even.)
    const int64_t tails = DEAL ? (int64_t)(gridDim.x >> 1) : 0;
    const int64_t nmain = nsteps - 2 * tails;
    const int64_t per = (nmain + gridDim.x - 1) / gridDim.x;
    int64_t s0 = tails + (int64_t)blockIdx.x * per;
    const int64_t mend = nsteps - tails;
    if (s0 > mend) s0 = mend;
    const int64_t s1 = (s0 + per < mend) ? (s0 + per) : mend;
    const int64_t dealt = DEAL ? (((int64_t)blockIdx.x < tails) ? (int64_t)blockIdx.x : nsteps - 1 - ((int64_t)blockIdx.x - tails)) : 0;
    const int64_t count = (s1 - s0) + (DEAL ? 1 : 0); // steps of this workgroup: [dealt,] s0 .. s1-1
    auto step_of = [&](int64_t kk) -> int64_t { return DEAL ? (kk == 0 ? dealt : s0 + kk - 1) : s0 + kk; };
    const float4 *x4 = reinterpret_cast<const float4 *>(x);
    const float4 *base = x4 + threadIdx.x;
#ifdef KM_NT_LOADS // (tuning experiment: the vector is read once per launch)
    auto ld = [&](int64_t kk) -> float4 { return kk < count ? __builtin_nontemporal_load(&base[step_of(kk) * KM_THREADS]) : make_float4(0.f, 0.f, 0.f, 0.f); };
#else
    auto ld = [&](int64_t kk) -> float4 { return kk < count ? base[step_of(kk) * KM_THREADS] : make_float4(0.f, 0.f, 0.f, 0.f); };
#endif
    float4 r[KM_RING];
#pragma unroll
    for (int j = 0; j < KM_RING; j++) r[j] = ld(j);

    if (MODE == 0 && !(which & 2) && (ws->st.done | ws->st.paused)) return; // (which & 2: counting pass after the fit, see nnc_kmeans_label_counts)
    if (MODE == 1 && n_dev && s0 >= s1 && blockIdx.x != gridDim.x - 1) return; // nothing in this workgroup's range
    const int k = ws->p.k;
    const int glog2 = ws->glog2, rlog2 = ws->rlog2;
    const int G = 1 << glog2;
    const int kp = (k + 7) & ~7;
    const int t = ws->cur ^ (which & 1);
    const KmTab *__restrict__ tab = &ws->tab[t];
#ifdef KM_KU_FAST // (tuning experiment: one dependent round of loads less at the head; valid for the current table only)
    const int kt = (which & 1) ? tab->ku : ws->ku_cur;
#else
    const int kt = tab->ku; // distinct centres: the sorted tables hold only those
#endif

    uint16_t *cell_s = reinterpret_cast<uint16_t *>(smem);
    float4 *pair_s = reinterpret_cast<float4 *>(smem + ((size_t)2 << glog2));
    float *cval_s = reinterpret_cast<float *>(smem + ((size_t)2 << glog2) + (size_t)kp * 16);
    uint16_t *orig_s = reinterpret_cast<uint16_t *>(smem + ((size_t)2 << glog2) + (size_t)kp * 20);
    size_t off = (((size_t)2 << glog2) + (size_t)kp * 22 + 15) & ~(size_t)15;
    unsigned *ovf_s = reinterpret_cast<unsigned *>(smem + off);
    off += KM_OVF_MAX * 4;
    unsigned long long *sum_s = reinterpret_cast<unsigned long long *>(smem + off);
    unsigned *cnt_s = reinterpret_cast<unsigned *>(smem + off + ((size_t)k << rlog2) * 8);

    {
        const uint4 *src = reinterpret_cast<const uint4 *>(tab->cell);
        uint4 *dst = reinterpret_cast<uint4 *>(cell_s);
        for (int i = threadIdx.x; i < (G >> 3); i += KM_THREADS) dst[i] = src[i];
        for (int i = threadIdx.x; i < kt; i += KM_THREADS) {
            float2 a = tab->cand[i];
            float2 b = (i + 1 < kt) ? tab->cand[i + 1] : a;
            pair_s[i] = make_float4(a.x, a.y, b.x, b.y);
            cval_s[i] = a.x;
            orig_s[i] = tab->orig[i];
        }
        {
            const int novf = min(tab->n_ovf, KM_OVF_MAX);
            for (int i = threadIdx.x; i < novf; i += KM_THREADS) ovf_s[i] = tab->ovf[i];
        }
        if (MODE == 0) {
            const int tot = k << rlog2;
            for (int i = threadIdx.x; i < tot; i += KM_THREADS) { sum_s[i] = 0ull; cnt_s[i] = 0u; }
        }
        if (MODE == 1 && dist_hist) {
            unsigned *hz = reinterpret_cast<unsigned *>(sum_s); // label mode has no accumulators: the space holds the histogram
            for (int i = threadIdx.x; i < 4096; i += KM_THREADS) hz[i] = 0u;
        }
    }
    __syncthreads();
    unsigned *hist_s = (MODE == 1 && dist_hist) ? reinterpret_cast<unsigned *>(sum_s) : nullptr;
    KmHistRun hr;
    hr.bin = 0xFFFFFFFFu; hr.cnt = 0;

    KmCtx c;
    c.cell_s = cell_s; c.pair_s = pair_s; c.cval_s = cval_s; c.orig_s = orig_s; c.ovf_s = ovf_s; c.sum_s = sum_s; c.cnt_s = cnt_s;
    c.mean = ws->p.x_mean; c.lo = ws->p.lo; c.inv = ws->inv;
    c.Sft = ws->p.fix_shift; c.gmax = G - 1; c.k = kt; c.rlog2 = rlog2;
    c.rep = threadIdx.x & ((1 << rlog2) - 1);
    KmRun run;
    run.p = -1; run.cnt = 0; run.sum = 0;
    if (trace) tr1 = __builtin_amdgcn_s_memrealtime();

    for (int64_t kk = 0; kk < count; kk += KM_RING) {
#pragma unroll
        for (int j = 0; j < KM_RING; j++) {
            const int64_t kcur = kk + j;
            const float4 v = r[j];
            r[j] = ld(kcur + KM_RING);
            if (kcur < count) {
                if (MODE == 0) km_accumulate4<ABL>(c, v, run);
                else {
                    const float xa[4] = {v.x, v.y, v.z, v.w};
                    km_emit<4, LT>(c, xa, 4 * (step_of(kcur) * KM_THREADS + threadIdx.x), labels_out, quant_out, dist_out, hist_s, hr);
                }
            }
        }
    }
    // ragged end (less than one step of float4s, then the scalars): last workgroup
    if (blockIdx.x == gridDim.x - 1) {
        const int64_t vdone = nsteps * KM_THREADS;
        {
            for (int64_t v = vdone + threadIdx.x; v < nvec; v += KM_THREADS) {
                const float4 a4 = x4[v];
                const float xa[4] = {a4.x, a4.y, a4.z, a4.w};
                if (MODE == 0) km_accumulate4<ABL>(c, a4, run);
                else km_emit<4, LT>(c, xa, 4 * v, labels_out, quant_out, dist_out, hist_s, hr);
            }
            // scalars: fewer than 4 when the input is 16-byte aligned; the whole vector otherwise
            for (int64_t i = (nvec << 2) + threadIdx.x; i < n; i += KM_THREADS) {
                const float xs[1] = {x[i]};
                if (MODE == 0) km_accumulate<1, ABL>(c, xs, run);
                else km_emit<1, LT>(c, xs, i, labels_out, quant_out, dist_out, hist_s, hr);
            }
        }
    }
    if (trace) tr2 = __builtin_amdgcn_s_memrealtime();
    if (MODE == 0) {
        if (ABL != 1 && ABL != 3) km_run_flush(c, run);
        else if (run.sum == 0x7fffffffffffll) sum_s[0] = run.sum; // keep the ablated arithmetic alive
        km_flush(c, ws);
    }
    if (MODE == 1 && hist_s) {
        if (hr.cnt) atomicAdd(&hist_s[hr.bin], hr.cnt);
        __syncthreads();
        for (int i = threadIdx.x; i < 4096; i += KM_THREADS)
            if (hist_s[i]) atomicAdd(&dist_hist[i], (unsigned long long)hist_s[i]);
    }
    if (trace && threadIdx.x == 0) {
        trace[4 * blockIdx.x + 0] = tr0; trace[4 * blockIdx.x + 1] = tr1;
        trace[4 * blockIdx.x + 2] = tr2; trace[4 * blockIdx.x + 3] = __builtin_amdgcn_s_memrealtime();
    }
}




// ---- the rank-boundary iteration (value-sorted vector + block prefix sums) ---------------------
//
// Wave j owns the j-th distinct centre (value order) and the boundary above it.  From the zones k_finalize left
// (zl / zr: outside [zl[p], zr[p]] centre p cannot be scikit-learn's float32 arg-min) it derives
//     L_j = min over q > j of zl[q]      below L_j no centre above j can win
//     U_j = max over q <= j of zr[q]     above U_j no centre up to j can win
// and locates three ranks in the sorted vector by 64-ary search (64 probes per round, all three searches in flight together):
//     a_j = #{x~ < L_j},  b_j = #{x~ <= U_j},  b_{j-1}.
// Samples [b_{j-1}, a_j) are certainly centre j's: their count is a difference of ranks, their fixed-point sum a difference
// of prefix sums (block prefix + at most 255 images added by the wave).  Samples [max(a_j, b_{j-1}), b_j) cannot be decided
// from the zones; every one of them is evaluated with the exact float32 expression over the centres that can still win
// (j .. max{r : zl[r] <= U_j}; almost always j and j + 1), ties to the lowest original index.  The stretches of all waves
// partition the vector.  A long undecided stretch (two centres closer than float32 can tell apart make their whole
// neighbourhood undecided) is published as tiles that any wave of the launch may take.
#ifdef NNC_DIAG
#define KBSTAMP(slot, val) do { if (NNC_KM_TRACE_PTR && lane == 0) { NNC_KM_TRACE_PTR[(slot)] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#else
#define KBSTAMP(slot, val) do { } while (0)
#endif
#define KM_PB NNC_PREFIX_BLOCK
#define KM_PG 1024 // blocks per group of the two-level prefix
#define KM_TILE 2048
#ifndef KM_FIT_LAG
#define KM_FIT_LAG 2 // iterations a plain batch still has to run when its status goes to the host (nnc_kmeans_fit)
#endif
#define KM_ZONE_REACH 12 // neighbours either side whose pair with a centre is looked at for its zone (km_finalize_body)
#define KM_ACC_W 256 // candidates of a crowded stretch whose sums a wave gathers in LDS
// samples per tile of a long undecided stretch: the work of a tile is samples x candidates, so a stretch many centres compete for
// (a crowd of relocated centres side by side: 66 candidates for 11 000 samples in the bench fit's second iteration, 120 us in six
// tiles of 2048) is cut finer -- KM_TILE for two candidates, 64 samples for sixty-four and more
__host__ __device__ __forceinline__ int km_tile_len(int ncand) { int t = (2 * KM_TILE / (ncand < 2 ? 2 : ncand)) & ~63; return t < 64 ? 64 : t; }
#define KM_Q_VALID (1ull << 62)
#ifndef KM_PUBLISH_TILES
#define KM_PUBLISH_TILES 5 // an undecided stretch of more tiles than this goes to the queue; a shorter one is its wave's own work
#endif


// Wave-wide reductions whose result every lane needs, without the LDS crossbar: __shfl_xor is ds_bpermute_b32 here -- a trip through
// the LDS pipe and a wait per step, six dependent steps (twelve instructions for 64 bits) a reduction, and a k_bounds wave has four
// groups of them one behind the other on its critical path (zone ends, the candidate range, the two prefix sums, the sums of a short
// undecided stretch: 254 ds_bpermute in the kernel, 1.5-2 us of its 10).  Four DPP steps (two quad permutes, the half-row mirror, the
// row mirror: VALU operand modifiers, no memory pipe) leave every lane of a row of sixteen with the row's result; the four rows are
// read into scalar registers (v_readlane takes the lane by number, whatever EXEC is) and folded there.  Integer sums and max / min
// do not depend on the order, so the bits are those of the butterfly.  The whole wave must be active at the call (as for __shfl_xor).
#define KM_DPP(v, ctrl) __builtin_amdgcn_update_dpp(0, (v), (ctrl), 0xF, 0xF, true)
#define KM_DPP_XOR1 0xB1    // quad_perm [1,0,3,2]
#define KM_DPP_XOR2 0x4E    // quad_perm [2,3,0,1]
#define KM_DPP_HMIRROR 0x141 // row_half_mirror: lane i of a group of eight <-> 7 - i
#define KM_DPP_MIRROR 0x140  // row_mirror: lane i of a row of sixteen <-> 15 - i
template <int CTRL> __device__ __forceinline__ long long km_dpp_ll(long long v)
{
    const int lo = KM_DPP((int)v, CTRL), hi = KM_DPP((int)(v >> 32), CTRL);
    return (long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned long long)(unsigned)lo);
}
__device__ __forceinline__ long long km_readlane_ll(long long v, int l)
{
    const int lo = __builtin_amdgcn_readlane((int)v, l), hi = __builtin_amdgcn_readlane((int)(v >> 32), l);
    return (long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned long long)(unsigned)lo);
}
__device__ __forceinline__ long long wave_sum_ll(long long v)
{
    v += km_dpp_ll<KM_DPP_XOR1>(v); v += km_dpp_ll<KM_DPP_XOR2>(v); v += km_dpp_ll<KM_DPP_HMIRROR>(v); v += km_dpp_ll<KM_DPP_MIRROR>(v);
    return (km_readlane_ll(v, 0) + km_readlane_ll(v, 16)) + (km_readlane_ll(v, 32) + km_readlane_ll(v, 48));
}
__device__ __forceinline__ double wave_max_d(double v)
{
    v = fmax(v, __longlong_as_double(km_dpp_ll<KM_DPP_XOR1>(__double_as_longlong(v))));
    v = fmax(v, __longlong_as_double(km_dpp_ll<KM_DPP_XOR2>(__double_as_longlong(v))));
    v = fmax(v, __longlong_as_double(km_dpp_ll<KM_DPP_HMIRROR>(__double_as_longlong(v))));
    v = fmax(v, __longlong_as_double(km_dpp_ll<KM_DPP_MIRROR>(__double_as_longlong(v))));
    const long long b = __double_as_longlong(v);
    return fmax(fmax(__longlong_as_double(km_readlane_ll(b, 0)), __longlong_as_double(km_readlane_ll(b, 16))),
                fmax(__longlong_as_double(km_readlane_ll(b, 32)), __longlong_as_double(km_readlane_ll(b, 48))));
}
__device__ __forceinline__ double wave_min_d(double v)
{
    v = fmin(v, __longlong_as_double(km_dpp_ll<KM_DPP_XOR1>(__double_as_longlong(v))));
    v = fmin(v, __longlong_as_double(km_dpp_ll<KM_DPP_XOR2>(__double_as_longlong(v))));
    v = fmin(v, __longlong_as_double(km_dpp_ll<KM_DPP_HMIRROR>(__double_as_longlong(v))));
    v = fmin(v, __longlong_as_double(km_dpp_ll<KM_DPP_MIRROR>(__double_as_longlong(v))));
    const long long b = __double_as_longlong(v);
    return fmin(fmin(__longlong_as_double(km_readlane_ll(b, 0)), __longlong_as_double(km_readlane_ll(b, 16))),
                fmin(__longlong_as_double(km_readlane_ll(b, 32)), __longlong_as_double(km_readlane_ll(b, 48))));
}
__device__ __forceinline__ int wave_max_i(int v)
{
    v = max(v, KM_DPP(v, KM_DPP_XOR1)); v = max(v, KM_DPP(v, KM_DPP_XOR2)); v = max(v, KM_DPP(v, KM_DPP_HMIRROR)); v = max(v, KM_DPP(v, KM_DPP_MIRROR));
    return max(max(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)), max(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}

// the group prefixes live behind the block prefixes in the caller's buffer
__host__ __device__ __forceinline__ long long km_prefix_nblk(long long n) { return (n + NNC_PREFIX_BLOCK - 1) / NNC_PREFIX_BLOCK; }
__device__ __forceinline__ const long long *km_pgrp(const long long *pblk, long long n) { return pblk + km_prefix_nblk(n) + 2; }
// ... and behind those the FINE prefixes: entry i = sum of fix(x~) over the first 64 * i samples (i = 0 .. 4 * nblk; every entry from
// ceil(n / 64) on holds the total), what the one-workgroup loop (k_lloyd) reads: one 8-byte entry and the 256 bytes of the
// sorted vector it stands in front of give the prefix sum at any rank
#define KL_BLK 64
__host__ __device__ __forceinline__ long long km_prefix_ngroups(long long n) { return (km_prefix_nblk(n) + 1 + KM_PG - 1) / KM_PG; }
__host__ __device__ __forceinline__ long long km_pfine_off(long long n) { return km_prefix_nblk(n) + 2 + km_prefix_ngroups(n) + 2; }

// sum of fix(x~) over the first r samples: block prefix + the wave adds the rest of r's block
__device__ __forceinline__ long long km_prefix_at(const float *__restrict__ xs, const long long *__restrict__ pblk, long long r,
                                                  long long n, float mean, int Sft, int lane)
{
    const long long blk = r >> 8, base = blk << 8;
    const int rem = (int)(r - base);
    long long acc = 0;
    if (rem > 0) { // wave-uniform
        const long long i0 = base + 4 * lane;
        if (base + KM_PB <= n) {
            const float4 v = *reinterpret_cast<const float4 *>(xs + i0);
            const int left = rem - 4 * lane; // how many of the four are below r
            if (left > 0) acc += fix_f32(v.x - mean, Sft);
            if (left > 1) acc += fix_f32(v.y - mean, Sft);
            if (left > 2) acc += fix_f32(v.z - mean, Sft);
            if (left > 3) acc += fix_f32(v.w - mean, Sft);
        } else {
            for (int u = 0; u < 4; u++) if (i0 + u < r) acc += fix_f32(xs[i0 + u] - mean, Sft);
        }
        acc = wave_sum_ll(acc);
    }
    return pblk[blk] + km_pgrp(pblk, n)[blk / KM_PG] + acc;
}

// the same in two halves, so that a caller can have the loads of several prefixes (and more) in flight before adding anything up
struct KmPfx { float4 v; long long pb; int rem; };
__device__ __forceinline__ KmPfx km_prefix_load(const float *__restrict__ xs, const long long *__restrict__ pblk, long long r, long long n, int lane)
{
    KmPfx p;
    const long long blk = r >> 8, base = blk << 8;
    p.rem = (int)(r - base);
    p.pb = pblk[blk] + km_pgrp(pblk, n)[blk / KM_PG];
    p.v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (p.rem > 0) {
        const long long i0 = base + 4 * lane;
        if (base + KM_PB <= n) p.v = *reinterpret_cast<const float4 *>(xs + i0);
        else {
            if (i0 + 0 < r) p.v.x = xs[i0 + 0];
            if (i0 + 1 < r) p.v.y = xs[i0 + 1];
            if (i0 + 2 < r) p.v.z = xs[i0 + 2];
            if (i0 + 3 < r) p.v.w = xs[i0 + 3];
        }
    }
    return p;
}
__device__ __forceinline__ long long km_prefix_finish(const KmPfx &p, float mean, int Sft, int lane)
{
    long long acc = 0;
    if (p.rem > 0) { // wave-uniform
        const int left = p.rem - 4 * lane; // how many of the four are below r
        if (left > 0) acc += fix_f32(p.v.x - mean, Sft);
        if (left > 1) acc += fix_f32(p.v.y - mean, Sft);
        if (left > 2) acc += fix_f32(p.v.z - mean, Sft);
        if (left > 3) acc += fix_f32(p.v.w - mean, Sft);
        acc = wave_sum_ll(acc);
    }
    return p.pb + acc;
}

__device__ __forceinline__ void km_shard_add(KmWs *ws, int p, long long sum, unsigned long long cnt)
{
    if (cnt) {
        atomicAdd(reinterpret_cast<unsigned long long *>(&ws->shard_sum[p & (KM_NSHARD - 1)][p]), (unsigned long long)sum);
        atomicAdd(&ws->shard_cnt[p & (KM_NSHARD - 1)][p], cnt);
    }
}

// exact labels of the samples [s, e) whose candidates are the centres plo .. phi (value order), added to the sums
// (Inlined again: round 2 made this a real call because the inlined form hung in the tile loop of k_bounds.  The call only hid the
// cause -- a ticket produced under `if (lane == 0)` and read with v_readfirstlane, see km_claim below, which is where it is fixed;
// with that the claim loop compiles to a scalar loop around this body, 83 VGPRs and no scratch instead of 90 + 64 bytes.)
__device__ __forceinline__ void km_bounds_range(const float *__restrict__ xs, long long s, long long e, int plo, int phi, const KmTab *__restrict__ tab,
                                                        KmWs *ws, float mean, int Sft, int lane)
{
    s = uni_ll(s); e = uni_ll(e); plo = uni_i(plo); phi = uni_i(phi);
    // A run of EQUAL values -- the zeros of a pruned vector between two centres float32 cannot tell apart: 17 M samples of the
    // bench vector, twice per fit -- gets one label: the vector is sorted, so equal ends mean equal everything in between.
    // One evaluation (the general rule: first strict minimum, ties to the lowest original index), value x count.
    // (The two ends are asked for here and looked at below, behind the loads of the path taken: a helper that waited for them
    // first, then for the two centres, then for its samples was three round trips from its first sum.)
    const bool chk = e - s >= 128 && phi > plo;
    float x0 = 0.0f, x1 = 1.0f;
    if (chk) { x0 = xs[s]; x1 = xs[e - 1]; }
    auto flat_run = [&]() {
        const float xc = x0 - mean;
        float bestd = INFINITY;
        int best = plo, besto = 0x7fffffff;
        for (int c = plo; c <= phi; c++) { // (wave-uniform: scalar loads)
            const float2 cm = tab->cand[c];
            const int oc = (int)tab->orig[c];
            const float d = cm.y + (-2.0f * (xc * cm.x));
            if (d < bestd || (d == bestd && oc < besto)) { bestd = d; best = c; besto = oc; }
        }
        if (lane == 0) km_shard_add(ws, best, (long long)fix_f32(xc, Sft) * (e - s), (unsigned long long)(e - s));
    };
    if (phi <= plo) { // (cannot happen for an undecided stretch; kept total: everything is plo's)
        long long sum = 0;
        unsigned cnt = 0;
        for (long long i = s + lane; i < e; i += 64) { sum += fix_f32(xs[i] - mean, Sft); cnt++; }
        sum = wave_sum_ll(sum);
        const long long c = wave_sum_ll((long long)cnt);
        if (lane == 0) km_shard_add(ws, plo, sum, (unsigned long long)c);
        return;
    }
    if (phi == plo + 1) {
        const float2 c0 = tab->cand[plo], c1 = tab->cand[plo + 1];
        const bool tie1 = tab->orig[plo + 1] < tab->orig[plo];
        long long s0 = 0, s1 = 0;
        unsigned n0 = 0, n1 = 0;
        // sixteen loads in flight per lane: a wave that is left alone with the tiles of a long stretch (its helpers looked at the
        // queue before the record was out) is bound by the latency of these loads -- with four in flight a tile of 2048 samples took
        // 8 us and the 75 tiles of one record of the bench fit 600 us whenever nobody came
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; u++) { const long long i = s + lane + 64 * u; v[u] = i < e ? xs[i] : 0.0f; }
        if (chk && x0 == x1) { flat_run(); return; }
        for (long long i0 = s; i0 < e; i0 += 1024) {
            if (i0 > s) {
#pragma unroll
                for (int u = 0; u < 16; u++) { const long long i = i0 + lane + 64 * u; v[u] = i < e ? xs[i] : 0.0f; }
            }
#pragma unroll
            for (int u = 0; u < 16; u++) {
                const long long i = i0 + lane + 64 * u;
                if (i < e) {
                    const float xc = v[u] - mean;
                    const float d0 = c0.y + (-2.0f * (xc * c0.x));
                    const float d1 = c1.y + (-2.0f * (xc * c1.x));
                    const int q = fix_f32(xc, Sft);
                    if (d1 < d0 || (d1 == d0 && tie1)) { s1 += q; n1++; } else { s0 += q; n0++; }
                }
            }
        }
        s0 = wave_sum_ll(s0); s1 = wave_sum_ll(s1);
        const long long m0 = wave_sum_ll((long long)n0), m1 = wave_sum_ll((long long)n1);
        if (lane == 0) { km_shard_add(ws, plo, s0, (unsigned long long)m0); km_shard_add(ws, plo + 1, s1, (unsigned long long)m1); }
        return;
    }
    if (chk && x0 == x1) { flat_run(); return; }
    // three or more centres within rounding distance of each other: the general scan.  The candidates sit in the lanes'
    // registers (lane l holds centre g0 + l of the current group of 64) and go round by readlane; a lane takes four samples
    // per batch so that their loads are in flight together and every candidate is fetched once for the four.  Where float32
    // cannot tell two centres apart the winner changes from one sample to the next, so the sums are gathered per candidate in
    // LDS (one slot per candidate and wave) and go to the global sums once per call: thousands of samples of one stretch
    // would otherwise queue up on the same few addresses.
    // (KM_ACC_W slots a wave: beyond that -- never seen -- the sums go out as runs of equal labels, two global atomics a run, which
    // for a crowd is two atomics a sample on a handful of addresses: the 66 candidates of the bench fit's second iteration took
    // 100 us that way when the slots were 64)
    __shared__ unsigned long long acc_sum[KM_THREADS / 64][KM_ACC_W];
    __shared__ unsigned acc_cnt[KM_THREADS / 64][KM_ACC_W];
    const int wv = (int)(threadIdx.x >> 6);
    const int ncand = phi - plo + 1;
    const bool use_lds = ncand <= KM_ACC_W;
    if (use_lds) for (int c = lane; c < ncand; c += 64) { acc_sum[wv][c] = 0ull; acc_cnt[wv][c] = 0u; }
    int run_p = -1;
    unsigned run_n = 0;
    long long run_s = 0;
    float2 cm0 = make_float2(0.0f, 0.0f);
    int om0 = 0x7fffffff;
    const bool one_group = ncand <= 64; // the candidates stay in the lanes' registers over the whole call
    if (one_group && lane < ncand) { cm0 = tab->cand[plo + lane]; om0 = (int)tab->orig[plo + lane]; }
    wave_lds_fence();
    // (NU samples a lane and batch: four where the stretch is longer than a wave -- every candidate fetched once for the four --, one
    // for a tile of 64 samples, the tile length of a crowd: the unrolled four cost a 64-sample tile of 51 candidates 2200
    // instructions, three quarters of them for samples that are not there, and a lone wave issues one instruction in four clocks)
    auto scan = [&](auto nu_c) {
        constexpr int NU = decltype(nu_c)::value;
        for (long long i0 = s; i0 < e; i0 += 64 * NU) {
            float xc[NU], bestd[NU];
            int best[NU], besto[NU];
            bool have[NU];
#pragma unroll
            for (int u = 0; u < NU; u++) {
                const long long i = i0 + lane + 64 * u;
                have[u] = i < e;
                xc[u] = have[u] ? xs[i] - mean : 0.0f;
                bestd[u] = INFINITY; best[u] = plo; besto[u] = 0x7fffffff;
            }
            for (int g0 = plo; g0 <= phi; g0 += 64) {
                float2 cm = cm0;
                int om = om0;
                if (!one_group) {
                    const int mine = g0 + lane;
                    cm = mine <= phi ? tab->cand[mine] : make_float2(0.0f, 0.0f);
                    om = mine <= phi ? (int)tab->orig[mine] : 0x7fffffff;
                }
                const int cnt = min(64, phi - g0 + 1);
                for (int c = 0; c < cnt; c++) {
                    const float cx = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cm.x), c));
                    const float cy = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cm.y), c));
                    const int oc = __builtin_amdgcn_readlane(om, c);
#pragma unroll
                    for (int u = 0; u < NU; u++) {
                        const float d = cy + (-2.0f * (xc[u] * cx));
                        if (d < bestd[u] || (d == bestd[u] && oc < besto[u])) { bestd[u] = d; best[u] = g0 + c; besto[u] = oc; }
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < NU; u++) {
                if (have[u]) {
                    if (use_lds) {
                        atomicAdd(&acc_sum[wv][best[u] - plo], (unsigned long long)(long long)fix_f32(xc[u], Sft));
                        atomicAdd(&acc_cnt[wv][best[u] - plo], 1u);
                    } else {
                        if (best[u] != run_p) { if (run_n) km_shard_add(ws, run_p, run_s, run_n); run_p = best[u]; run_n = 0; run_s = 0; }
                        run_n++;
                        run_s += fix_f32(xc[u], Sft);
                    }
                }
            }
        }
    };
    if (e - s <= 64) scan(std::integral_constant<int, 1>{}); else scan(std::integral_constant<int, 4>{});
    if (use_lds) {
        wave_lds_fence();
        for (int c = lane; c < ncand; c += 64) if (acc_cnt[wv][c]) km_shard_add(ws, plo + c, (long long)acc_sum[wv][c], (unsigned long long)acc_cnt[wv][c]);
    } else if (run_n) km_shard_add(ws, run_p, run_s, run_n);
}

// A ticket for the whole wave.  EVERY lane runs the atomic (lane 0 adds 1, the others 0; the compiler folds the wave's adds into
// one), then lane 0's result is broadcast.  NOT "if (lane == 0) t = atomicAdd(...); t = readfirstlane(t)": v_readfirstlane reads the
// first ACTIVE lane, so that form is only right while lane 0 is active at the read -- and nothing makes the compiler keep it so.
// Round 2's hang was exactly that: with km_bounds_range inlined into the claim loop, the `if (lane == 0)` of its closing
// km_shard_add and the `if (lane == 0)` of the next claim were threaded into ONE lane-0-only path around the loop's back edge; the
// structurizer then sent lanes 1-63 round the loop ahead of lane 0 (ISA of fc02041^: the loop's continue mask is the lane-0 mask,
// `s_mov_b64 s[0:1], s[8:9]` ... `s_andn2_b64 exec, exec, s[56:57]`, and `v_readfirstlane_b32 s2, v15` runs with exec = ~1), they
// read their own zero as the ticket and ran tile 0 again, for ever.  A value one lane hands to the wave must be produced with
// the whole wave active.
__device__ __forceinline__ int km_claim(int *counter, int lane)
{
    const int t = atomicAdd(counter, lane == 0 ? 1 : 0);
    return uni_i(t);
}
// the same rule for a word the wave polls: every lane loads it (one address, one request), nobody loads it for the others
__device__ __forceinline__ int km_peek_i(const int *word) { return uni_i(__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)); }
__device__ __forceinline__ unsigned long long km_peek_ull(const unsigned long long *word)
{
    return (unsigned long long)uni_ll((long long)__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

// wave j's share of one rank-boundary pass (see above): centre j's certain stretch and the undecided stretch above it
// sources of one pass: the zones and centres of the table the pass runs against (the fixed-address copy of the current table,
// or -- counting pass against the previous centres -- that table itself)
struct KmBndSrc { const double *zr, *zl; const float2 *cand; const uint16_t *orig; const int *ku; };
#define KM_BND_R ((NNC_KMAX + 63) / 64)

// everything a wave needs to know before its first probe: the zone ends of all centres (lane by lane), its own two centres, where its
// boundaries were last time.  None of it depends on the state block, so a kernel issues these loads together with the state's
// (k_bounds), not a memory round trip behind them.
// (BR = rounds of 64 centres, a template parameter: with a run-time bound every round became a basic block of its own -- load, wait,
// branch -- and the five rounds of K = 257 five memory round trips one after the other; now all loads of a wave are in flight at once)
template <int BR> struct KmBndPre { double zrv[BR], zlv[BR]; float2 cj0, cj1r; int oj0, oj1r, ku; long long hint_a, hint_b, hint_bm; };
template <int BR>
__device__ __forceinline__ void km_bounds_preload(KmBndPre<BR> &p, const int j, const int lane, const KmWs *__restrict__ ws, const KmBndSrc src)
{
#pragma unroll
    for (int r = 0; r < BR; r++) {
        const int q = lane + 64 * r; // (< NNC_KMAX: BR * 64 <= NNC_KMAX; entries beyond ku are ignored by the reader)
        p.zrv[r] = src.zr[q]; p.zlv[r] = src.zl[q];
    }
    const int jq = j < NNC_KMAX - 1 ? j : NNC_KMAX - 2;
    p.cj0 = src.cand[jq]; p.cj1r = src.cand[jq + 1];
    p.oj0 = src.orig[jq]; p.oj1r = src.orig[jq + 1];
    p.hint_a = ws->hint_a[jq]; p.hint_b = ws->hint_b[jq]; p.hint_bm = jq > 0 ? ws->hint_b[jq - 1] : -1;
    p.ku = *src.ku;
}

template <int BR>
__device__ __forceinline__ bool km_bounds_wave(const int j, const int lane, const float *__restrict__ xs, const long long n, KmWs *__restrict__ ws,
                                               const KmTab *__restrict__ tab, const KmBndSrc src, const float mean, const int Sft,
                                               const long long *__restrict__ pblk, int *qn_seen, const KmBndPre<BR> &pre, const int announced = 0,
                                               int *wg_searched = nullptr)
{
    bool published = false;
    // ---- one round of loads: every zone end (a lane holds those of the centres lane, lane + 64, ...; bounded by the caller's k,
    // which is known before anything has arrived), the number of distinct centres, the two centres either side of this
    // wave's boundary, where the boundaries were last time
    // (by reference to ONE block the caller filled: through a pointer that could point to either of two blocks the arrays lived in
    // scratch memory -- 136 bytes a lane at K = 257, every access a trip to memory)
    const double (&zrv)[BR] = pre.zrv, (&zlv)[BR] = pre.zlv;
    const float2 cj0 = pre.cj0, cj1r = pre.cj1r;
    const int oj0 = pre.oj0, oj1r = pre.oj1r;
    const long long hint_a = pre.hint_a, hint_b = pre.hint_b, hint_bm = pre.hint_bm;
    const int ku = pre.ku;
    if (j < ku) {
        // ---- the zone ends that bound this wave's stretches
        double Uj = -INFINITY, Ujm1 = -INFINITY, Lj = INFINITY;
#pragma unroll
        for (int r = 0; r < BR; r++) {
            const int q = lane + 64 * r;
            if (q < ku) {
                if (q <= j) Uj = fmax(Uj, zrv[r]);
                if (q < j) Ujm1 = fmax(Ujm1, zrv[r]);
                if (q > j) Lj = fmin(Lj, zlv[r]);
            }
        }
        const float2 cj1 = j + 1 < ku ? cj1r : cj0;
        const bool tie1 = j + 1 < ku && oj1r < oj0;
        Uj = wave_max_d(Uj); Ujm1 = wave_max_d(Ujm1); Lj = wave_min_d(Lj);
        // the highest centre that can still win somewhere below U_j: max{q : zl[q] <= U_j}  (j + 1 unless centres crowd)
        int phi = j;
#pragma unroll
        for (int r = 0; r < BR; r++) {
            const int q = lane + 64 * r;
            if (q < ku && q > j && zlv[r] <= Uj) phi = q;
        }
        phi = wave_max_i(phi);
        KBSTAMP(16 * j + 1, 0);
        const bool top = j == ku - 1; // no boundary above the last centre
        // ---- three searches in lock step.  State k: the answer lies in [lo, hi]; hi is n or a position known to satisfy the test
        long long lo[3] = {0, 0, 0}, hi[3] = {n, n, n};
        const double T[3] = {Lj, Uj, Ujm1};
        if (top) { lo[0] = n; lo[1] = n; }   // a = b = n
        if (j == 0) hi[2] = 0;               // b_{-1} = 0
        // First round from where the boundary was found last time (centres move little from one iteration to the next): probes
        // at hint - 2^30 ... hint - 1, hint, hint + 1 ... hint + 2^30 and the last sample bracket the answer to within the
        // distance it moved; the 64-ary rounds below finish inside that bracket.
        {
            const long long hint[3] = {hint_a, hint_b, hint_bm};
            auto probe_at = [&](long long h, int t) -> long long {
                long long p = t < 31 ? h - ((long long)1 << (30 - t)) : (t == 31 ? h : (t < 63 ? h + ((long long)1 << (t - 32)) : n - 1));
                if (p < 0) p = 0;
                if (p > n - 1) p = n - 1;
                return p;
            };
            float v[3];
#pragma unroll
            for (int k = 0; k < 3; k++) {
                v[k] = 0.0f;
                if (lo[k] < hi[k] && hint[k] >= 0) v[k] = xs[probe_at(hint[k] < n ? hint[k] : n - 1, lane)];
            }
#pragma unroll
            for (int k = 0; k < 3; k++) {
                if (lo[k] < hi[k] && hint[k] >= 0) {
                    const long long h = hint[k] < n ? hint[k] : n - 1;
                    const double xc = (double)(v[k] - mean);
                    const bool pred = (k == 0) ? (xc >= T[k]) : (xc > T[k]);
                    const unsigned long long bal = __ballot(pred);
                    if (bal == 0ull) lo[k] = n; // not even the last sample passes: the answer is n  (hi is n)
                    else {
                        const int f = __ffsll((long long)bal) - 1;
                        hi[k] = probe_at(h, f);
                        lo[k] = f == 0 ? 0 : probe_at(h, f - 1) + 1;
                        if (lo[k] > hi[k]) lo[k] = hi[k]; // (clamped probes may coincide)
                    }
                }
            }
        }
        KBSTAMP(16 * j + 2, 0);
        while ((lo[0] < hi[0]) | (lo[1] < hi[1]) | (lo[2] < hi[2])) {
            float v[3];
            long long step[3];
#pragma unroll
            for (int k = 0; k < 3; k++) {
                v[k] = 0.0f; step[k] = 1;
                if (lo[k] < hi[k]) {
                    const long long len = hi[k] - lo[k];
                    step[k] = (len + 63) >> 6;
                    long long off = (long long)(lane + 1) * step[k];
                    if (off > len) off = len;
                    v[k] = xs[lo[k] + off - 1];
                }
            }
#pragma unroll
            for (int k = 0; k < 3; k++) {
                if (lo[k] < hi[k]) {
                    const long long len = hi[k] - lo[k];
                    const double xc = (double)(v[k] - mean);
                    const bool pred = (k == 0) ? (xc >= T[k]) : (xc > T[k]);
                    const unsigned long long bal = __ballot(pred);
                    if (bal == 0ull) lo[k] = hi[k];
                    else {
                        const int f = __ffsll((long long)bal) - 1;
                        long long offf = (long long)(f + 1) * step[k];
                        if (offf > len) offf = len;
                        long long offp = (long long)f * step[k];
                        if (offp > len) offp = len;
                        hi[k] = lo[k] + offf - 1;
                        lo[k] = lo[k] + offp; // one past the last probe that failed (lo itself if the first probe passed)
                    }
                }
            }
        }
        const long long a = uni_ll(lo[0]), b = uni_ll(lo[1]), bm = uni_ll(lo[2]);
        if (lane == 0) { ws->hint_a[j] = a; ws->hint_b[j] = b; ws->bnd_phi[j] = phi; }
        KBSTAMP(16 * j + 3, 0);
        // ---- this centre's certain stretch [bm, a) and the undecided stretch above it, [max(a, bm), b): every load of both
        // goes out before anything is added up (the block prefixes, the two partial blocks, up to four undecided samples a lane)
        const long long s = a > bm ? a : bm;
        const long long und = b - s;
        if (NNC_KM_TRACE_PTR && lane == 0) { NNC_KM_TRACE_PTR[16 * j + 8] = (unsigned long long)und; NNC_KM_TRACE_PTR[16 * j + 9] = (unsigned long long)(phi - j); NNC_KM_TRACE_PTR[16 * j + 10] = (unsigned long long)(a > bm ? a - bm : 0); }
        const bool quick = und > 0 && und <= 256 && phi == j + 1; // few samples, two candidates: settled right here
        // a long stretch goes out as tiles for everybody, and at once: the others look at the queue a round of loads from now
        // (a long stretch of one value -- the zero plateau -- is no work at all: km_bounds_range settles it with one evaluation)
        const int tile = km_tile_len(phi - j + 1);
        const bool flat = und > tile && xs[s] == xs[b - 1];
        // (a crowd's stretch from six tiles up: a helper is three dependent round trips away from its first sample -- the records, the claim, the samples --,
        // some 8 us in a pass where everybody is at the queue; the publisher is through five tiles of its own in less)
        // (... where its samples are few: a wave on its own has 256 samples in flight at a time, a round trip each)
        const int self_max = max(tile, min(1024, KM_PUBLISH_TILES * tile));
        if (und > self_max && !flat) {
            const int r = km_claim(&ws->q_n, lane);
            if (r < NNC_KMAX) { // (every wave publishes at most once per launch and there are at most NNC_KMAX waves)
                published = true;
                // (EVERY lane stores the record -- the same two words: the publisher sees its own tiles through by reading the queue
                // like any helper, lane r & 63 reads record r, and only a thread's own store is certain to be in front of its own load)
                __hip_atomic_store(&ws->q_w0[r], KM_Q_VALID | ((unsigned long long)j << 40) | (unsigned long long)s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&ws->q_w1[r], KM_Q_VALID | ((unsigned long long)phi << 40) | (unsigned long long)b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        // (an announced pass: the others wait at the queue until every wave has said whether it had something to publish.
        // Relaxed: a release at agent scope writes this XCD's L2 back -- measured 8 us a wave; the record above went out as atomic
        // stores in front of this one, and a helper that should still miss it costs nothing but its help)
        // (one atomic a WORKGROUP, by the last of its waves to get here -- they count in LDS first: 257 atomics on one word queued up at
        // the memory side, and memory operations return in order: every wave's next load waited behind its own increment, the own work
        // of a wave in an announced pass took 13 us against 8.5)
        if (announced) {
            const int mine = min(4, ku - 4 * (int)blockIdx.x); // waves of this workgroup that have a boundary (j < ku)
            int last = 0;
            if (lane == 0) last = atomicAdd(wg_searched, 1) + 1 == mine;
            if (uni_i(last) && lane == 0) __hip_atomic_fetch_add(&ws->q_searched, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        float uv[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        if (quick) {
#pragma unroll
            for (int u = 0; u < 4; u++) { const long long i = s + lane + 64 * u; if (i < b) uv[u] = xs[i]; }
        }
        if (a > bm) {
            const KmPfx pa = km_prefix_load(xs, pblk, a, n, lane), pm = km_prefix_load(xs, pblk, bm, n, lane);
            const long long sum = km_prefix_finish(pa, mean, Sft, lane) - km_prefix_finish(pm, mean, Sft, lane);
            if (lane == 0) km_shard_add(ws, j, sum, (unsigned long long)(a - bm));
        }
        // (how many long stretches are out by now: asked here, looked at by the caller when this wave's own work is through)
        *qn_seen = __hip_atomic_load(&ws->q_n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // (every lane: see km_claim)
        KBSTAMP(16 * j + 4, 0);
        if (quick) {
            long long s0 = 0, s1 = 0;
            unsigned n0 = 0, n1 = 0;
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if (s + lane + 64 * u < b) {
                    const float xc = uv[u] - mean;
                    const float d0 = cj0.y + (-2.0f * (xc * cj0.x));
                    const float d1 = cj1.y + (-2.0f * (xc * cj1.x));
                    const int q = fix_f32(xc, Sft);
                    if (d1 < d0 || (d1 == d0 && tie1)) { s1 += q; n1++; } else { s0 += q; n0++; }
                }
            }
            s0 = wave_sum_ll(s0); s1 = wave_sum_ll(s1);
            const long long m0 = wave_sum_ll((long long)n0), m1 = wave_sum_ll((long long)n1);
            if (lane == 0) { km_shard_add(ws, j, s0, (unsigned long long)m0); km_shard_add(ws, j + 1, s1, (unsigned long long)m1); }
        } else if (und > 0 && !published) km_bounds_range(xs, s, b, j, phi, tab, ws, mean, Sft, lane); // (also a long one the full queue refused)
    }
    return published;
}

// Tiles of long undecided stretches: every wave of the launch that comes through here takes tiles from the records that are out.
// A publisher comes through here after its own record is out and leaves only when all its tiles are taken, so no tile depends on
// anybody else (nobody waits for anybody: a record a wave does not see yet is finished by its publisher).
// (myj / nwaves: the caller's boundary and the number of waves of its pass that come through here, if it knows them.  A record with
// few tiles left is then approached by about that many waves, not by all of them: a claim is an atomic on ONE word at the memory
// side, and 260 of them for a record of five tiles -- every wave of an announced pass arrives at the same moment -- queued up for
// 10-20 us, most of what the passes behind a placement cost.  The publisher always sees to its own record.)
__device__ __forceinline__ void km_bounds_help(const int lane, const float *__restrict__ xs, KmWs *__restrict__ ws,
                                               const KmTab *__restrict__ tab, const float mean, const int Sft, const int myj = -1, const int nwaves = 0)
{
    // (relaxed on purpose: an acquire per look would drop the caches of a thousand waves)
    // Sixty-four records a look: lane r reads record r, so a walk through the queue is one round trip to the memory side, not one
    // per record (with a few dozen records out -- the iterations right after a mass relocation -- every wave of the launch used
    // to spend its hundred microseconds walking, whatever there was left to do).
    // (the first sixty-four records are fetched together with the counter, not behind it: a slot nobody has written holds no VALID
    // bit -- the finalize step clears what a pass has used -- so the counter only says whether there is a second round)
    int nrec = 64;
    for (int r0 = 0; r0 < nrec; r0 += 64) {
        const int rl = r0 + lane;
        unsigned long long w0 = 0ull, w1 = 0ull;
        int nx = 0x7fffffff;
        if (rl < NNC_KMAX) {
            w0 = __hip_atomic_load(&ws->q_w0[rl], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            w1 = __hip_atomic_load(&ws->q_w1[rl], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            nx = __hip_atomic_load(&ws->q_next[rl], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (r0 == 0) nrec = min(km_peek_i(&ws->q_n), (int)NNC_KMAX);
        if (rl >= nrec) { w0 = 0ull; w1 = 0ull; }
        bool live = false, solo = false;
        if ((w0 & KM_Q_VALID) && (w1 & KM_Q_VALID)) { // (else not out yet: its publisher will see to it)
            const long long s = (long long)(w0 & ((1ull << 40) - 1)), e = (long long)(w1 & ((1ull << 40) - 1));
            const int plo = (int)((w0 >> 40) & 0xFFFFF), phi = (int)((w1 >> 40) & 0xFFFFF);
            const int tile = km_tile_len(phi - plo + 1);
            const int ntl = (int)((e - s + tile - 1) / tile);
            live = nx < ntl; // (spent records: no need to bump their counters again)
            if (live && nwaves > 0 && plo != myj) {
                // (as of this look: with fewer tiles left than waves around, a tile a wave and one claim each -- asking for the next
                // tile while working on one doubles the atomics, and for most of them the answer is "none left")
                const int left = ntl - nx;
#ifdef KM_NO_SOLO
                if (false) {
#else
                if (left <= nwaves) {
#endif
                    solo = true;
                    const int stride = nwaves / left;
                    if (stride > 1 && (unsigned)(myj + 7 * rl) % (unsigned)stride != 0u) live = false;
                }
            }
        }
        unsigned long long todo = __ballot(live);
        const unsigned long long solo_mask = __ballot(live && solo);
        while (todo) {
            const int b = __ffsll((long long)todo) - 1;
            todo &= todo - 1ull;
            const unsigned long long v0 = (unsigned long long)__shfl((long long)w0, b), v1 = (unsigned long long)__shfl((long long)w1, b);
            const long long s = (long long)(v0 & ((1ull << 40) - 1)), e = (long long)(v1 & ((1ull << 40) - 1));
            const int plo = (int)((v0 >> 40) & 0xFFFFF), phi = (int)((v1 >> 40) & 0xFFFFF);
            const int tile = km_tile_len(phi - plo + 1);
            const int ntiles = (int)((e - s + tile - 1) / tile);
            int t = km_claim(&ws->q_next[r0 + b], lane);
            if ((solo_mask >> b) & 1ull) { // (wave-uniform)
                if (t < ntiles) {
                    const long long ts = s + (long long)t * tile;
                    km_bounds_range(xs, ts, ts + tile < e ? ts + tile : e, plo, phi, tab, ws, mean, Sft, lane);
                }
                continue;
            }
            while (t < ntiles) {
                const int tn = km_claim(&ws->q_next[r0 + b], lane); // (the next ticket is on its way while this tile is worked on: a
                                                                    // wave left alone with a record is otherwise bound by the round trips)
                const long long ts = s + (long long)t * tile;
                const long long te = ts + tile < e ? ts + tile : e;
                km_bounds_range(xs, ts, te, plo, phi, tab, ws, mean, Sft, lane);
                t = tn;
            }
        }
    }
}

template <int BR>
__global__ __launch_bounds__(256) void k_bounds(const float *__restrict__ xs, long long n, KmWs *__restrict__ ws, int which,
                                               const long long *__restrict__ pblk)
{
    const int lane = threadIdx.x & 63;
    const int j = uni_i(blockIdx.x * 4 + (threadIdx.x >> 6));
    const unsigned long long t_enter = __builtin_amdgcn_s_memrealtime(); // (100 MHz)
    __shared__ int wg_searched; // waves of this workgroup that are through their searches (an announced pass counts them here first)
    if (threadIdx.x == 0) wg_searched = 0;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); // (LDS only: the loads below do not wait for it; every wave passes here, the early returns are further down)
    KBSTAMP(16 * j + 0, 0);
    KmBndSrc src;
    const KmTab *tab;
    if (which & 1) { // counting pass against the previous centres: that table itself
        tab = &ws->tab[ws->cur ^ 1];
        src.zr = tab->zr; src.zl = tab->zl; src.cand = tab->cand; src.orig = tab->orig; src.ku = &tab->ku;
    } else {
        tab = &ws->tab[ws->cur]; // (needed only by long or crowded undecided stretches)
        src.zr = ws->bnd.zr; src.zl = ws->bnd.zl; src.cand = ws->bnd.cand; src.orig = ws->bnd.orig; src.ku = &ws->bnd.ku;
    }
    KmBndPre<BR> pre;
    const bool fixed_src = !(which & 1); // (the iteration's own tables live at a fixed address: their loads go out with the state's)
    if (fixed_src) km_bounds_preload<BR>(pre, j, lane, ws, src);
    const int stop = (which & 2) ? 0 : (ws->st.done | ws->st.paused); // (which & 2: counting pass after the fit)
    const int unasked = (which & 4) ? !ws->wide : 0; // (which & 4: enqueued behind k_lloyd in case it hands an iteration over)
    const int hint = ws->help_hint;
    const float mean = ws->p.x_mean;
    const int Sft = ws->p.fix_shift;
    if (stop | unasked) return;
    int qn_seen = 0; // the number of long stretches that were out when this wave's own loads went out
    if (!fixed_src) km_bounds_preload<BR>(pre, j, lane, ws, src);
    const bool published = km_bounds_wave<BR>(j, lane, xs, n, ws, tab, src, mean, Sft, pblk, &qn_seen, pre, hint == 2, &wg_searched);
    KBSTAMP(16 * j + 5, 0);
    // A look at the queue costs a round trip to the memory side (the counter is shared by all XCDs).  Worth it for the waves of a
    // pass whose predecessor published long stretches (centres stay crowded for a few iterations), and for a publisher: it has
    // to see its own tiles through.
    // A wave whose boundary sits in a wide undecided interval searches a wider bracket and publishes its stretch microseconds after
    // the others are through: its 75 tiles (bench fit, iteration 12: two centres 4e-8 apart at the edge of the pruned gap, 154 551
    // samples between them) were then left to itself and whoever came late -- 30-140 us for the launch, 510-630 us when nobody
    // came.  Where the finalize step has announced such an interval (help_hint), a wave therefore waits with its look at the queue
    // until every wave of the pass is through its searches (q_searched: whatever there was to publish is out by then).  Nobody's
    // PROGRESS depends on this -- a publisher sees its own tiles through alone if need be -- and the number of looks is bounded,
    // so the loop ends whatever the others do.  Cost: the pass lasts as long as its slowest search plus one look (+ 4 us).
    const bool waits = hint == 2 && !published && fixed_src;
    int looks = 0;
    if (waits) {
        // (the first look when the slowest searches of such a pass are about through, 9 us after this wave started, not at once: 260
        // waves asking one word a million times a second each were in the way of the waves still searching -- their own work took
        // 13.7 us against 8.5 in a pass nobody polls)
        const int nwaves = pre.ku;
        for (int nap = 0; nap < 64 && __builtin_amdgcn_s_memrealtime() - t_enter < 900ull; nap++) __builtin_amdgcn_s_sleep(16);
        for (int look = 0; look < 128; look++) {
            looks++;
            if (km_peek_i(&ws->q_searched) >= nwaves) break;
            __builtin_amdgcn_s_sleep(48);
        }
    }
    KBSTAMP(16 * j + 7, 0);
    if (NNC_KM_TRACE_PTR && lane == 0) NNC_KM_TRACE_PTR[16 * j + 12] = (unsigned long long)looks;
    if (published || hint || uni_i(qn_seen) > 0) km_bounds_help(lane, xs, ws, tab, mean, Sft, j, hint == 2 ? pre.ku : 0);
    KBSTAMP(16 * j + 6, 0);
}

// block sums of the fixed-point images of a sorted vector, then their exclusive scan (once per fit)
__global__ __launch_bounds__(256) void k_prefix_blocks(const float *__restrict__ xs, long long n, float mean, int Sft, long long *__restrict__ pblk)
{
    const int lane = threadIdx.x & 63;
    const long long nblk = (n + KM_PB - 1) / KM_PB;
    long long *__restrict__ pfine = pblk + km_pfine_off(n); // (here: the raw sums of the 64-sample quarters; k_prefix_fine turns them into prefixes)
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long long)gridDim.x * 4;
    // four blocks a turn, their loads in flight together (one a turn left every wave a chain of a dozen dependent round trips: 21-25 us
    // for 100 MB)
    for (long long b0 = wave; b0 < nblk; b0 += 4 * nwaves) {
        float4 v[4];
        bool full[4], have[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const long long b = b0 + u * nwaves;
            have[u] = b < nblk;
            full[u] = have[u] && (b + 1) * KM_PB <= n;
            v[u] = full[u] ? *reinterpret_cast<const float4 *>(xs + b * KM_PB + 4 * lane) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (!have[u]) continue; // (wave-uniform)
            const long long b = b0 + u * nwaves;
            const long long i0 = b * KM_PB + 4 * lane;
            long long acc = 0;
            if (full[u]) {
                acc = ((long long)fix_f32(v[u].x - mean, Sft) + fix_f32(v[u].y - mean, Sft)) + ((long long)fix_f32(v[u].z - mean, Sft) + fix_f32(v[u].w - mean, Sft));
            } else {
                for (int q = 0; q < 4; q++) if (i0 + q < n) acc += fix_f32(xs[i0 + q] - mean, Sft);
            }
#pragma unroll
            for (int off = 1; off <= 8; off <<= 1) acc += __shfl_xor(acc, off); // sixteen lanes = one quarter (64 samples)
            if ((lane & 15) == 0) pfine[4 * b + (lane >> 4)] = acc;
            acc += __shfl_xor(acc, 16);
            acc += __shfl_xor(acc, 32);
            if (lane == 0) pblk[b] = acc;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) pblk[nblk] = 0; // one virtual block behind the last: its prefix is the total
}

// fine prefixes from the scanned block / group prefixes and the quarters' raw sums (in place; one thread per 256-sample block)
__global__ __launch_bounds__(256) void k_prefix_fine(long long *__restrict__ pblk, long long n)
{
    const long long nblk = km_prefix_nblk(n);
    const long long *__restrict__ pgrp = km_pgrp(pblk, n);
    long long *__restrict__ pfine = pblk + km_pfine_off(n);
    const long long b = (long long)blockIdx.x * 256 + threadIdx.x;
    if (b > nblk) return;
    const long long base = pblk[b] + pgrp[b / KM_PG];
    if (b == nblk) { pfine[4 * b] = base; return; } // the total
    const long long q0 = pfine[4 * b], q1 = pfine[4 * b + 1], q2 = pfine[4 * b + 2];
    pfine[4 * b] = base; pfine[4 * b + 1] = base + q0; pfine[4 * b + 2] = base + q0 + q1; pfine[4 * b + 3] = base + q0 + q1 + q2;
}

// Two-level exclusive scan of the block sums: groups of KM_PG blocks, one workgroup per group (pblk[b] becomes the sum of the
// blocks of b's group before b, gtot[g] the group's total), then one workgroup turns the group totals into their exclusive
// prefixes pgrp[g].  The prefix at block b is pblk[b] + pgrp[b / KM_PG].
__global__ __launch_bounds__(KM_THREADS) void k_prefix_groups(long long *__restrict__ pblk, long long nblk, long long *__restrict__ pgrp)
{
    __shared__ long long wave_tot[KM_THREADS / 64];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const long long b = (long long)blockIdx.x * KM_PG + tid;
    const long long v = (b <= nblk) ? pblk[b] : 0; // (the sum of block b, as k_prefix_blocks left it; entry nblk: 0)
    long long s = v;
    for (int off = 1; off < 64; off <<= 1) { const long long t = __shfl_up(s, off); if (lane >= off) s += t; }
    if (lane == 63) wave_tot[wv] = s;
    __syncthreads();
    long long pre = 0, tot = 0;
    for (int w = 0; w < KM_THREADS / 64; w++) { if (w < wv) pre += wave_tot[w]; tot += wave_tot[w]; }
    if (b <= nblk) pblk[b] = pre + s - v; // exclusive, in place: a thread reads and writes its own entry only
    if (tid == 0) pgrp[blockIdx.x] = tot;
}

__global__ __launch_bounds__(KM_THREADS) void k_prefix_top(long long *__restrict__ pgrp, long long ngroups)
{
    __shared__ long long wave_tot[KM_THREADS / 64];
    __shared__ long long carry_s;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (long long base = 0; base < ngroups; base += KM_THREADS) {
        const long long g = base + tid;
        const long long v = g < ngroups ? pgrp[g] : 0;
        long long s = v;
        for (int off = 1; off < 64; off <<= 1) { const long long t = __shfl_up(s, off); if (lane >= off) s += t; }
        if (lane == 63) wave_tot[wv] = s;
        __syncthreads();
        long long pre = carry_s, tot = 0;
        for (int w = 0; w < KM_THREADS / 64; w++) { if (w < wv) pre += wave_tot[w]; tot += wave_tot[w]; }
        if (g < ngroups) pgrp[g] = pre + s - v;
        __syncthreads();
        if (tid == 0) carry_s += tot;
        __syncthreads();
    }
    if (tid == 0) pgrp[ngroups] = carry_s;
}

// ---- finalize / prepare kernel (one workgroup) -------------------------------------------
// (FIN_* modes of the finalize step: nnc_km_shared.hpp)

// 1 / d, a little too large rather than too small (float32 reciprocal widened by 2^-20): a zone end that comes out a hair
// further from its midpoint is still a valid zone, and the division is the slowest thing in the zone loop
__device__ __forceinline__ double km_rcp_up(double d)
{
    const float r = __frcp_rn((float)d);
    return (d > 1e-37 && d < 1e37) ? (double)r * (1.0 + 9.5367431640625e-07) : 1.0 / d;
}

// Between which x~ can scikit-learn's float32 comparison of the distances to two centres cp < cq go either way?  Outside the
// interval returned here it cannot: below it d_p < d_q, above it d_q < d_p, strictly, in float32 as computed by
//     d_j(x) = fl( C_j + fl(-2 * fl(x * c_j)) ),   C_j = fl(c_j * c_j)          (_k_means_lloyd.pyx:196-203)
// Two bounds, both rigorous; the interval is their intersection.
// (a) global: |d_j - (c_j^2 - 2 x c_j)| <= E = 2.5 * 2^-24 * (cm^2 + 2 xb cm) for every |x| <= xb (cm = max |c|), and the exact
//     difference of the two distances is 2 delta (x - mid): decided once |x - mid| > E / delta.
// (b) local, several times tighter where it matters (the dense middle of the data, where |x| is far below xb): the squares C_j are
//     known float32 numbers, so take them as they are: D_j(x) = C_j - 2 x c_j crosses at x* = (C_q - C_p) / (2 delta) (the
//     midpoint, moved by the rounding of the squares), D_p - D_q = 2 delta (x - x*), and with u = 2^-24
//         |d_j - D_j| <= u (1 + u) 2 |x c_j| + u |D_j|                      (one rounding in the product, one in the sum)
//         |D_j(x)| <= |D*| + 2 |x - x*| |c_j|,   D* = D_p(x*) = D_q(x*);    |x| <= |x*| + |x - x*|
//     so with S = |c_p| + |c_q| the comparison is decided once
//         |x - x*| (2 delta - (4 + 2u) u S) > u (2 |D*| + (2 + 2u) |x*| S)
//     -- linear in |x - x*| on both sides, hence for every x beyond that distance, not only near it.  (Both sides get a hair of
//     slack for the double arithmetic here and for products that underflow.)
struct KmZone { double lo, hi; };
__device__ __forceinline__ KmZone km_pair_zone(const double cp, const double cq, const double xb)
{
    const double U = 5.9604644775390625e-08; // 2^-24
    const double delta = cq - cp;            // > 0: the callers pass distinct centres in value order
    const double cm = fmax(fabs(cp), fabs(cq));
    const double mid = 0.5 * (cp + cq);
    const double E = 2.5 * U * (cm * cm + 2.0 * xb * cm) + 1e-42;
    const double w0 = E * km_rcp_up(delta);
    KmZone z;
    z.lo = mid - w0; z.hi = mid + w0;
    const double S = fabs(cp) + fabs(cq);
    const double den = 2.0 * delta - 4.5 * U * S;
    if (den > delta) { // (else the two centres are a few ulps apart: the global bound says all there is to say)
        const float cpf = (float)cp, cqf = (float)cq;
        const double Cp = (double)(cpf * cpf), Cq = (double)(cqf * cqf);
        // (no double division on this path: x* from a float32 reciprocal and one Newton step -- relative error below 3.4e-14, paid
        // for in w -- and w, which only has to be large enough, from a reciprocal rounded up)
        const double d2 = 2.0 * delta;
        double xs;
        if (d2 > 1e-37 && d2 < 1e37) {
            const double r0 = (double)__frcp_rn((float)d2);
            xs = (Cq - Cp) * (r0 * (2.0 - d2 * r0));
        } else xs = (Cq - Cp) / d2;
        const double Ds = fmax(fabs(Cp - 2.0 * xs * cp), fabs(Cq - 2.0 * xs * cq));
        const double num = U * (2.0 * Ds + 2.0000005 * fabs(xs) * S) * 1.000001 + 1e-42;
        const double w = (num * km_rcp_up(den)) * 1.000001 + fabs(xs) * 1e-13;
        z.lo = fmax(z.lo, xs - w); z.hi = fmin(z.hi, xs + w);
    }
    return z;
}
// how far the crossing point of a pair can lie from its midpoint at most (for the early exit of the loops over the pairs)
__device__ __forceinline__ double km_pair_slack(const double delta, const double xb) { return 1.25e-7 * xb * xb * km_rcp_up(delta); }

// workgroup barrier that orders LDS traffic only (see km_finalize_body)
__device__ __forceinline__ void km_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier\n\ts_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ unsigned f32_ordered_bits(float x)
{
    const unsigned u = __float_as_uint(x);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u); // unsigned order == float order
}
__device__ __forceinline__ float f32_from_ordered_bits(unsigned o)
{
    return __uint_as_float((o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o);
}

#include "nnc_lloyd.hpp"

// workgroup votes through an LDS word that starts at zero and is used once per launch
__device__ __forceinline__ int km_vote_or(int *word, const int x)
{
    if (__any(x) && (threadIdx.x & 63) == 0) *word = 1;
    km_lds_barrier();
    return *word;
}
__device__ __forceinline__ int km_vote_count(int *word, const int x) // threads with x != 0
{
    const int c = (int)__popcll(__ballot(x != 0));
    if (c && (threadIdx.x & 63) == 0) atomicAdd(word, c);
    km_lds_barrier();
    return *word;
}

// An empty-cluster event of a rank-boundary iteration settled by the finalize step itself -- kl_relocate, the selection the
// resident loop uses (nnc_lloyd.hpp), fed from what the k_bounds pass of this iteration left in the workspace: its ranks a_j / b_j
// (hint_a / hint_b), the candidate range of every undecided stretch (bnd_phi), the centres it labelled against (tab).  The chain
// of four launches behind the iteration (or, outside the batches that carry one, the host's look-in, its windowed relocation
// and the rest of a batch gone idle) is then not needed.  Returns 1 when the sums in sum_o / cnt_o have been edited and the step
// goes on as if resumed; 0 when the event is not one for this path (more than KL_RM_MAX empty clusters, long undecided
// stretches, no proof, ...): nothing has changed and the step pauses as before.
template <int NT, bool MASS>
__device__ int km_finalize_relocate(KmWs *__restrict__ ws, const float *__restrict__ xs, const long long n, const KmTab *__restrict__ tab, const int ku,
                                    const int k, const int cur, long long *sum_o, long long *cnt_o, const float mean, const int Sft)
{
    constexpr int KC = NNC_KMAX;
    __shared__ __align__(16) unsigned char buf[sizeof(KlHead) + (size_t)KC * (8 + 8 + 4 + 4 + 4 + 4 + 2 + 2) + 4096 * 2];
    KlHead *hd = reinterpret_cast<KlHead *>(buf);
    KlArr L = {};
    {
        unsigned char *q = buf + sizeof(KlHead);
        L.A = reinterpret_cast<long long *>(q); q += (size_t)KC * 8;
        L.B = reinterpret_cast<long long *>(q); q += (size_t)KC * 8;
        L.cs = reinterpret_cast<float *>(q); q += (size_t)KC * 4;
        L.csq = reinterpret_cast<float *>(q); q += (size_t)KC * 4;
        L.cold = reinterpret_cast<float *>(q); q += (size_t)KC * 4;
        L.call = reinterpret_cast<float *>(q); q += (size_t)KC * 4; // (the chunk list's first-chunk table)
        L.so = reinterpret_cast<uint16_t *>(q); q += (size_t)KC * 2;
        L.phi = reinterpret_cast<uint16_t *>(q); q += (size_t)KC * 2;
        L.qj = reinterpret_cast<uint16_t *>(q);
    }
    const int tid = threadIdx.x;
    if (tid == 0) { hd->n_empty = 0; hd->ku = ku; hd->nch = 0; hd->slow = 0; hd->r_flat = 0; }
    __syncthreads();
    int e = 0;
    for (int j = tid; j < k; j += NT) { e += cnt_o[j] == 0; L.cold[j] = ws->c[cur][j]; }
    if (e) atomicAdd(&hd->n_empty, e);
    __syncthreads();
    const bool mass = MASS && hd->n_empty > KL_RM_MAX; // (the first iterations behind a density / forgy init: kl_relocate_mass)
    if (!MASS && hd->n_empty > KL_RM_MAX) return 0;
    if (!mass && 2 * ku > KL_RPASS * (NT / 8)) return 0; // (not an event for this path: leave before the tables are fetched)
    for (int p = tid; p < ku; p += NT) {
        const float2 c = tab->cand[p];
        L.cs[p] = c.x; L.csq[p] = c.y; L.so[p] = tab->orig[p];
        L.A[p] = ws->hint_a[p]; L.B[p] = ws->hint_b[p];
        const int ph = ws->bnd_phi[p];
        L.phi[p] = (uint16_t)(ph < p + 1 ? p + 1 : (ph > ku - 1 ? ku - 1 : ph));
    }
    __syncthreads();
    // the chunks of the undecided stretches, as kl_chunks cuts them (a long stretch -- a run of equal values, two centres float32
    // cannot tell apart -- makes the event the chain's)
    int *qfirst = reinterpret_cast<int *>(L.call);
    for (int j = tid; j + 1 < ku; j += NT) {
        const long long bm = j > 0 ? L.B[j - 1] : 0, a = L.A[j], b = L.B[j];
        const long long s0 = a > bm ? a : bm;
        const long long len = b - s0;
        qfirst[j] = 0;
        if (len > 0) {
            const int cs_ = (int)L.phi[j] == j + 1 ? KL_CHUNK : KL_CHUNK_CROWD;
            const long long nc = (len + cs_ - 1) / cs_;
            if (nc > (mass ? 512 : 64)) hd->slow = 1; // (a mass event's keys live in the workspace: room for a long stretch as well)
            else {
                const int first = atomicAdd(&hd->nch, (int)nc);
                qfirst[j] = first;
                for (int i = 0; i < (int)nc; i++) if (first + i < KL_QMAX) L.qj[first + i] = (uint16_t)j;
            }
        }
    }
    __syncthreads();
    if (NNC_FIN_TRACE_PTR && tid == 0) { // diagnostics: what every event of the fit looked like (by iteration)
        unsigned long long *rt = NNC_FIN_TRACE_PTR;
        const int it = min(ws->st.iter, 63);
        rt[300 + 4 * it] = (unsigned long long)hd->n_empty; rt[301 + 4 * it] = (unsigned long long)hd->nch; rt[302 + 4 * it] = (unsigned long long)hd->slow; rt[303 + 4 * it] = (unsigned long long)ku;
    }
    if (hd->slow || hd->nch > 1024) return 0; // (beyond that the candidate list would not fit anyway: kl_relocate checks)
    if constexpr (MASS) {
        if (mass) {
            __shared__ KlMass mass_s;
            return kl_relocate_mass<NT>(xs, n, ws, hd, L, &mass_s, sum_o, cnt_o, k, hd->nch, mean, Sft);
        }
    }
    return kl_relocate<NT>(xs, n, ws, hd, L, sum_o, cnt_o, k, hd->nch, mean, Sft);
}

// NT threads: a fit with few centres runs it as a single wave (64) or four (256), for which the many barriers and
// wave-to-wave hand-overs of the scans cost next to nothing; NT >= k is all it needs (k > 1024 takes two rounds of 1024).
// WAVE (NT == 64 only): the body is run by ONE wave of a larger workgroup, so it may not use workgroup barriers; the
// wave's own lock step (plus a compiler fence) orders its LDS traffic.  Returns true if new zones were left
// (gcell / hcell / ku_out = {ku, cur} filled), i.e. the cell table has to be rebuilt.
template <int NT, bool ONEWAVE, bool MASS = false>
__device__ __forceinline__ bool km_finalize_body(KmWs *__restrict__ ws, int mode, int resume, int *gcell, int *hcell, int *ku_out, const bool lazy = false,
                                                 const float *__restrict__ reloc_xs = nullptr, const long long reloc_n = 0, const int k_hint = 0)
{
    static_assert(!ONEWAVE || NT == 64, "the barrier-free form is for a single wave");
    // The threads of this step talk to each other through LDS only; what they write to global memory is for later kernels.  A
    // __syncthreads() is a fence over ALL memory (s_waitcnt vmcnt(0): gfx9 counts stores there too), so every barrier behind a
    // batch of global stores would wait a memory round trip for them -- a microsecond each, and there are a dozen.  km_lds_barrier
    // orders the LDS traffic only; the stores drain while the step goes on.  The votes go through LDS words (one per site).
#define FIN_SYNC() do { if (ONEWAVE) wave_lds_fence(); else km_lds_barrier(); } while (0)
#define FIN_OR(slot, x) (ONEWAVE ? (int)__any(x) : km_vote_or(&fin_votes[slot], (x)))
#define FIN_AND(slot, x) (ONEWAVE ? (int)__all(x) : !km_vote_or(&fin_votes[slot], !(x)))
#define FIN_COUNT(slot, x) (ONEWAVE ? (int)__popcll(__ballot(x)) : km_vote_count(&fin_votes[slot], (x)))
    __shared__ int fin_votes[8];
    __shared__ long long sum_o[NNC_KMAX];
    __shared__ long long cnt_o[NNC_KMAX];
    __shared__ __align__(16) float cnew[NNC_KMAX];
    __shared__ float cs[NNC_KMAX];        // sorted centred centres
    __shared__ uint16_t so[NNC_KMAX];     // sorted -> original
    __shared__ float sq[NNC_KMAX];
    __shared__ unsigned long long sh_key;
    __shared__ int ovf_n;
    __shared__ PwHeap heap;
    __shared__ float cu[NNC_KMAX];     // distinct sorted centre values
    __shared__ uint16_t sou[NNC_KMAX]; // their (lowest) original indices
    __shared__ double wave_a[NT / 64], wave_b[NT / 64];

    const int tid = threadIdx.x;
    unsigned long long *ftr = NNC_FIN_TRACE_PTR;
#define FSTAMP(i) do { if (ftr && tid == 0) ftr[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
    FSTAMP(0);
    if (tid == 0) ws->cells_pending = 0;
    if (!ONEWAVE && tid < 8) fin_votes[tid] = 0; // (the barrier below is in front of the first vote)
    // ---- ONE round of loads for everything this step reads.  On an otherwise idle chip every DEPENDENT global load costs about a
    // microsecond (the lines were written by the kernel in front, on other compute units: each first touch goes to the memory side),
    // and until round 4 there were four of them one behind the other here -- the queue length, behind a barrier the header, behind
    // the header's `cur` the order and the previous centres, behind `ku` the shards: 3.6 of the step's 12.5 us.  Nothing below needs
    // a loaded value for its ADDRESS: both tables' orders and both centre arrays are fetched (the header says which one counts), the
    // shards of position `tid` whatever the number of distinct centres turns out to be (below the number of centres the LAUNCHER knows,
    // k_hint, a kernel argument: sixteen waves fetching shards for 1024 positions where 257 are in use took 1.4 us longer than the
    // four dependent rounds had).  Stores wait until the header has spoken.
    const int kb = k_hint > 0 ? min(k_hint, (int)NNC_KMAX) : (int)NNC_KMAX;
    const bool takes_a_pass = mode != FIN_FROM_PARTIALS; // (sharded: the packing call in front of the all-reduce has taken the queue)
    const bool from_shards = mode == FIN_FROM_SHARDS || mode == FIN_PACK_ONLY;
    const int qn_raw = takes_a_pass ? ws->q_n : 0;
    const int st_done = ws->st.done, st_paused = ws->st.paused, st_iter = ws->st.iter, reloc_fail = ws->reloc_fail;
    const int spec_go = ws->spec_go;
    const int k = ws->p.k, Sft = ws->p.fix_shift, max_iter = ws->p.max_iter, glog2 = ws->glog2;
    const float tol_v = ws->p.tol, p_lo = ws->p.lo, p_hi = ws->p.hi, inv_f = ws->inv;
    const long long n_tot = ws->p.n_total;
    const int ku0 = ws->tab[0].ku, ku1 = ws->tab[1].ku;
    int cur = ws->cur;
    // Could this iteration's labels equal the previous iteration's?  Only if every cluster kept its
    // count (prev_counts); the host runs the full label comparison (strict convergence) only then.
    const bool track = (mode == FIN_FROM_SHARDS) || (mode == FIN_FROM_PARTIALS && !resume);
    constexpr int RR = 2; // centres per thread: the launcher picks NT >= k / 2 (64 threads up to 64 centres, 256 up to 256, 1024 beyond)
    long long pc[RR];
    float c_both[RR][2];
    uint16_t pm_both[RR][2], pmn_both[RR][2];
#pragma unroll
    for (int r = 0; r < RR; r++) {
        pc[r] = 0; c_both[r][0] = c_both[r][1] = 0.0f; pm_both[r][0] = pm_both[r][1] = pmn_both[r][0] = pmn_both[r][1] = 0;
        const int j = tid + r * NT;
        if (j < kb) {
            if (track) pc[r] = ws->prev_counts[j];
            if (mode != FIN_INIT) {
                c_both[r][0] = ws->c[0][j]; c_both[r][1] = ws->c[1][j];
                pm_both[r][0] = ws->tab[0].perm[j]; pm_both[r][1] = ws->tab[1].perm[j];
                if (j + 1 < NNC_KMAX) { pmn_both[r][0] = ws->tab[0].perm[j + 1]; pmn_both[r][1] = ws->tab[1].perm[j + 1]; }
            }
        }
    }
    long long sh_s[KM_NSHARD];
    unsigned long long sh_c[KM_NSHARD];
    uint16_t og_both[2] = {0, 0};
    const bool sh_pre = from_shards && tid < kb;
#pragma unroll
    for (int sh = 0; sh < KM_NSHARD; sh++) { sh_s[sh] = 0; sh_c[sh] = 0; }
    if (sh_pre) {
        og_both[0] = ws->tab[0].orig[tid]; og_both[1] = ws->tab[1].orig[tid];
#pragma unroll
        for (int sh = 0; sh < KM_NSHARD; sh++) { sh_s[sh] = ws->shard_sum[sh][tid]; sh_c[sh] = ws->shard_cnt[sh][tid]; }
    }
    // the tile queue of the pass that produced these sums (k_bounds) is spent -- for the call that takes the sums of a pass; a
    // call that resumes behind a relocation (or finds it has nothing to do: the resume of a chain enqueued "in case") leaves
    // the queue and above all the announcement for the next pass alone (until round 3 the idle resume behind every iteration of
    // a batch with chains wiped it: the passes that most needed helpers at the queue went without).  The counter itself is reset
    // by thread 0 behind the first barrier of the path taken -- every thread has its copy of it by then (the loop below needs it).
    const int qn = min(qn_raw, (int)NNC_KMAX);
    for (int r = tid; r < qn; r += NT) { ws->q_w0[r] = 0ull; ws->q_w1[r] = 0ull; ws->q_next[r] = 0; }
    FSTAMP(44);
#define FIN_QUEUE_RESET() do { if (tid == 0 && takes_a_pass) { ws->q_n = 0; ws->q_searched = 0; ws->help_hint = qn > 0; } } while (0)
#define FIN_LEAVE() do { if (ONEWAVE) wave_lds_fence(); else __syncthreads(); FIN_QUEUE_RESET(); return false; } while (0)
    if (mode != FIN_INIT && mode != FIN_PACK_ONLY && st_done) FIN_LEAVE();
    if (mode == FIN_FROM_SHARDS && st_paused) FIN_LEAVE();
    if (mode == FIN_PACK_ONLY && (st_done | st_paused)) {
        // no new iteration was accumulated: hand the all-reduce this rank's own sums again,
        // so that reducing an idle iteration leaves `partials` unchanged
        const int k2 = 2 * k;
        for (int i = tid; i < k2; i += NT) ws->partials[i] = ws->partials_local[i];
        FIN_LEAVE();
    }
    if (mode == FIN_FROM_PARTIALS && st_paused && !resume) FIN_LEAVE();
    // the resume behind a relocation chain enqueued "in case": only if that chain had an event to settle (k_reloc_windows, the
    // head of every such chain, sets the flag either way; nobody clears it here, where waves still on their way would read it)
    if (resume == 2 && !spec_go) FIN_LEAVE();
    if (resume && reloc_fail) { // unproven windowed selection: stay paused, tell the host
        if (tid == 0) ws->st.paused = 2;
        FIN_LEAVE();
    }

    // the previous centres and the order the E-step of this iteration used
    float cold_r[RR];
    int spa[RR], spb[RR];
#pragma unroll
    for (int r = 0; r < RR; r++) {
        const int j = tid + r * NT;
        const bool in = mode != FIN_INIT && j < k;
        cold_r[r] = in ? (cur ? c_both[r][1] : c_both[r][0]) : 0.0f;
        spa[r] = in ? (int)(cur ? pm_both[r][1] : pm_both[r][0]) : 0;
        spb[r] = (in && j + 1 < k) ? (int)(cur ? pmn_both[r][1] : pmn_both[r][0]) : 0;
        if (!(track && j < k)) pc[r] = 0;
    }
    if (from_shards) {
        const KmTab *tab = &ws->tab[cur];
        const int ku_cur = cur ? ku1 : ku0;
        for (int j = tid; j < k; j += NT) { sum_o[j] = 0; cnt_o[j] = 0; } // duplicates of a centre own nothing
        FIN_SYNC();
        FIN_QUEUE_RESET();
        FSTAMP(45);
        for (int p = tid; p < ku_cur; p += NT) {
            long long s = 0;
            unsigned long long c = 0;
            int o;
            if (p == tid && sh_pre) { // (the round fetched above)
                o = cur ? og_both[1] : og_both[0];
#pragma unroll
                for (int sh = 0; sh < KM_NSHARD; sh++) { s += sh_s[sh]; c += sh_c[sh]; }
            } else {
                o = tab->orig[p];
                for (int sh = 0; sh < KM_NSHARD; sh++) { s += ws->shard_sum[sh][p]; c += ws->shard_cnt[sh][p]; }
            }
            for (int sh = 0; sh < KM_NSHARD; sh++) { ws->shard_sum[sh][p] = 0; ws->shard_cnt[sh][p] = 0; }
            sum_o[o] = s; cnt_o[o] = (long long)c;
        }
        FIN_SYNC();
        FSTAMP(46);
        for (int j = tid; j < k; j += NT) {
            ws->partials[j] = sum_o[j]; ws->partials[k + j] = cnt_o[j];
            ws->partials_local[j] = sum_o[j]; ws->partials_local[k + j] = cnt_o[j];
        }
        if (mode == FIN_PACK_ONLY) return false;
    } else if (mode == FIN_FROM_PARTIALS) {
        for (int j = tid; j < k; j += NT) { sum_o[j] = ws->partials[j]; cnt_o[j] = ws->partials[k + j]; }
    }
    FIN_SYNC();
    FSTAMP(47);
    if (!from_shards) FIN_QUEUE_RESET(); // (FIN_INIT: the first pass of a fit starts with an empty queue)
    int same_counts_now = 0;
    bool settled_event = false; // an empty-cluster event was settled in this very call (km_finalize_relocate)
    if (track) {
        int count_diff = 0;
#pragma unroll
        for (int r = 0; r < RR; r++) {
            const int j = tid + r * NT;
            if (j < k) {
                const long long c = cnt_o[j];
                count_diff |= (pc[r] != c);
                ws->prev_counts[j] = c;
            }
        }
        const int any_diff = FIN_OR(0, count_diff);
        same_counts_now = any_diff ? 0 : 1;
        if (tid == 0) ws->st.same_counts = any_diff ? 0 : 1;
    }
    FSTAMP(1);

    if (mode != FIN_INIT) {
        // ---- empty clusters?  (one barrier-with-count)
        int my_empty = 0;
        for (int j = tid; j < k; j += NT) my_empty += (cnt_o[j] == 0);
        if (tid == 0) sh_key = 0ull;
        int n_empty = FIN_COUNT(1, my_empty);
        bool settled = false;
        settled_event = false;
        if constexpr (!ONEWAVE) {
            // (the same conditions under which the chain enqueued "in case" goes ahead: km_spec_decide)
            // (n_empty counts THREADS with an empty cluster here: more of them than the selection takes clusters is a mass event)
            if (n_empty > 0 && (MASS || n_empty <= KL_RM_MAX) && !resume && reloc_xs && mode == FIN_FROM_SHARDS && lazy && !(st_iter >= 1 && same_counts_now)) {
                __syncthreads(); // (the global stores of this step so far are out before the selection reads the workspace)
                if (ftr && tid == 0) { ftr[20] = ftr[0]; ftr[21] = __builtin_amdgcn_s_memrealtime(); }
                const KmTab *tabc = &ws->tab[cur];
                settled = km_finalize_relocate<NT, MASS>(ws, reloc_xs, reloc_n, tabc, cur ? ku1 : ku0, k, cur, sum_o, cnt_o, ws->p.x_mean, Sft) != 0;
                if (settled) { n_empty = 0; settled_event = true; }
                if (settled && ftr && tid == 0) ftr[22] = __builtin_amdgcn_s_memrealtime();
            }
        }
        if (n_empty > 0 && !resume) {
            int tot_empty = 0;
            if (tid == 0) {
                for (int j = 0; j < k; j++) tot_empty += (cnt_o[j] == 0); // rare path, exact count
                ws->st.paused = 1; ws->st.n_empty = tot_empty;
                if (lazy) ws->tab[cur].n_ovf = 0; // the cell table of this (current) table is rebuilt on demand: start its side list empty
            }
            return false;
        }
        // ---- _average_centers
        for (int j = tid; j < k; j += NT)
            if (cnt_o[j] > 0) cnew[j] = (float)ldexp((double)sum_o[j] / (double)cnt_o[j], -Sft);
        FIN_SYNC();
        if (n_empty > 0) { // only after a relocation that bailed out (all samples on their centres)
            // first index of the largest count (key = count, then lowest index)
            for (int j = tid; j < k; j += NT)
                atomicMax(&sh_key, ((unsigned long long)cnt_o[j] << 11) | (unsigned long long)(2047 - j));
            FIN_SYNC();
            const int amax = 2047 - (int)(sh_key & 2047ull);
            for (int j = tid; j < k; j += NT)
                if (cnt_o[j] <= 0) {
                    // sklearn copies centers[argmax] as it stands: averaged if argmax < j, raw sum otherwise
                    cnew[j] = (amax < j) ? cnew[amax] : (float)ldexp((double)sum_o[amax], -Sft);
                }
            FIN_SYNC();
        }
        FSTAMP(2);
        // ---- _center_shift and the tolerance test
#pragma unroll
        for (int r = 0; r < RR; r++) {
            const int j = tid + r * NT;
            if (j < k) {
                float d = cnew[j] - cold_r[r];
                float s2 = d * d;
                float sft = (float)sqrt((double)s2);
                sq[j] = sft * sft;
            }
        }
        FIN_SYNC();
        // (NumPy's pairwise sum of the k squared shifts: by the first wave on its own, the others go on)
        float tot = 0.0f;
        if (ONEWAVE) tot = block_pairwise_sum<true>([&](int i) { return sq[i]; }, k, &heap);
        else if (tid < 64) tot = wave_pairwise_sum([&](int i) { return sq[i]; }, k, &heap);
        if (tid == 0) {
            int iter = st_iter + 1;
            int done = 0;
            if (tot <= tol_v) done = 1;
            else if (iter >= max_iter) done = 2;
            ws->st.iter = iter; ws->st.shift_tot = tot; ws->st.done = done;
            ws->st.paused = 0; ws->st.n_empty = 0;
            ws->cur = cur ^ 1;
        }
        cur ^= 1;
        for (int j = tid; j < k; j += NT) ws->c[cur][j] = cnew[j];
        FIN_SYNC();
    } else {
        for (int j = tid; j < k; j += NT) cnew[j] = ws->c[cur][j];
        FIN_SYNC();
    }

    FSTAMP(3);
    // ---- sort the centres (rank by counting; ties by original index)
    KmTab *tab = &ws->tab[cur];
    // the centres move little per iteration: first try the previous permutation
    int still_sorted = 0;
    if (mode != FIN_INIT) {
        int ok = 1;
#pragma unroll
        for (int r = 0; r < RR; r++) {
            const int p = tid + r * NT;
            if (p < k) {
                const int a = spa[r];
                so[p] = (uint16_t)a;
                cs[p] = cnew[a];
                if (p + 1 < k) {
                    const int b = spb[r];
                    const float va = cnew[a], vb = cnew[b];
                    ok &= (va < vb) || (va == vb && a < b);
                }
            }
        }
        FSTAMP(12);
        still_sorted = FIN_AND(2, ok);
        // nearly sorted (two neighbours changed places): a few odd-even transposition passes repair it
        // (not behind a relocation: a centre that was moved to a far sample is far from its old place, the passes would be for nothing)
        const bool moved_far = resume || settled_event;
        for (int pass = 0; pass < 3 && !still_sorted && !moved_far; pass++) {
            for (int parity = 0; parity < 2; parity++) {
                for (int p = 2 * tid + parity; p + 1 < k; p += 2 * NT) {
                    const float va = cs[p], vb = cs[p + 1];
                    const uint16_t a = so[p], b = so[p + 1];
                    if (!((va < vb) || (va == vb && a < b))) { cs[p] = vb; cs[p + 1] = va; so[p] = b; so[p + 1] = a; }
                }
                FIN_SYNC();
            }
            int ok2 = 1;
            for (int p = tid; p + 1 < k; p += NT) {
                const float va = cs[p], vb = cs[p + 1];
                ok2 &= (va < vb) || (va == vb && so[p] < so[p + 1]);
            }
            still_sorted = FIN_AND(3 + pass, ok2);
        }
    }
    FSTAMP(8);
    if (ftr && tid == 0) ftr[13] = (unsigned long long)still_sorted;
    if (!still_sorted) {
        // rank by counting; PARTS lanes share one element and split the comparisons
        int parts = 1;
        while (parts < 64 && k * parts * 2 <= NT) parts <<= 1;
        const int per_part = (((k + parts - 1) / parts) + 3) & ~3; // (a multiple of four: the values are read four at a time)
        const float4 *c4 = reinterpret_cast<const float4 *>(cnew);
        for (int t = tid; t < ((k * parts + NT - 1) / NT) * NT; t += NT) {
            const int j = t / parts, part = t % parts;
            int rank = 0;
            float v = 0.0f;
            if (j < k) {
                v = cnew[j];
                const int i0 = part * per_part, i1 = min(k, i0 + per_part);
#pragma unroll 4
                for (int i = i0; i < i1; i += 4) {
                    // (bit operations, not || and &&: the short-circuit form compiled to a branch per comparison, some twenty-five
                    // instructions an element -- 66 000 comparisons at K = 257 made this loop 7 us of every resumed step)
                    const float4 u = c4[i >> 2];
                    rank += (int)(u.x < v) | ((int)(u.x == v) & (int)(i < j));
                    rank += ((int)(u.y < v) | ((int)(u.y == v) & (int)(i + 1 < j))) & (int)(i + 1 < i1);
                    rank += ((int)(u.z < v) | ((int)(u.z == v) & (int)(i + 2 < j))) & (int)(i + 2 < i1);
                    rank += ((int)(u.w < v) | ((int)(u.w == v) & (int)(i + 3 < j))) & (int)(i + 3 < i1);
                }
            }
            for (int off = 1; off < parts; off <<= 1) rank += __shfl_xor(rank, off);
            if (j < k && part == 0) { cs[rank] = v; so[rank] = (uint16_t)j; }
        }
    }
    FIN_SYNC();
    FSTAMP(9);
    if (settled_event && ftr && tid == 0) { ftr[23] = ftr[3]; ftr[24] = ftr[9]; }
    // Equal centres: the first one (lowest original index; the sort breaks ties that way) takes every
    // tie, the others can never win.  Keep only distinct values in the search tables.
    int ku = 0;
    {
        int *wave_i = reinterpret_cast<int *>(wave_a);
        const int rounds_u = (k + NT - 1) / NT;
        int carry = 0;
        for (int rd = 0; rd < rounds_u; rd++) {
            const int p = rd * NT + tid;
            float v = 0.0f;
            int o = 0, first = 0;
            if (p < k) {
                v = cs[p]; o = so[p];
                tab->perm[p] = (uint16_t)o;
                first = (p == 0) || (cs[p - 1] != v);
            }
            const unsigned long long bal = __ballot(first);
            const int before = __popcll(bal & ((1ull << (tid & 63)) - 1ull));
            if ((tid & 63) == 0) wave_i[tid >> 6] = __popcll(bal);
            FIN_SYNC();
            int pre = carry, tot = carry;
            for (int w = 0; w < NT / 64; w++) { const int wv = wave_i[w]; if (w < (tid >> 6)) pre += wv; tot += wv; }
            if (first) { cu[pre + before] = v; sou[pre + before] = (uint16_t)o; }
            carry = tot;
            FIN_SYNC();
        }
        ku = min(carry, k); // (never more than there are centres)
    }
    FSTAMP(10);
    for (int p = tid; p < ku; p += NT) {
        const float v = cu[p];
        cs[p] = v;
        tab->cand[p] = make_float2(v, v * v);
        tab->orig[p] = sou[p];
        ws->bnd.cand[p] = make_float2(v, v * v);
        ws->bnd.orig[p] = sou[p];
    }
    if (tid == 0) { tab->ku = ku; ws->ku_cur = ku; ws->bnd.ku = ku; ovf_n = 0; }
    FIN_SYNC();
    FSTAMP(4);
    // ---- zone of every centre: the x-interval [left, right] on which it can be the float32 arg-min,
    // turned at once into cells: cell g covers x~ - lo in [g * ra, (g+1) * rb]; centre p can open
    // cell g iff right >= lo + g*ra  <=>  g <= G_p, and can close it iff left <= lo + (g+1)*rb  <=>
    // g >= H_p.  G_p, H_p are rounded outwards (that only ever widens a candidate range) and made
    // monotone (prefix max / suffix min), so that the candidate range of a cell is
    // [first p with G_p >= g, last p with H_p <= g].
    const double U = 5.9604644775390625e-08; // 2^-24
    const double xb = fmax(fabs((double)p_lo), fabs((double)p_hi));
    const int G = 1 << glog2;
    const double lo = (double)p_lo;
    const double inv = (double)inv_f;
    const double ra = inv > 0.0 ? (1.0 - 4.0 * U) / inv * (1.0 - 4.0 * U) : 0.0;
    const double rb = inv > 0.0 ? (1.0 + 4.0 * U) / inv * (1.0 + 4.0 * U) : 0.0;
    {
        const int rounds = (ku + NT - 1) / NT;
        int *wave_i = reinterpret_cast<int *>(wave_a);
        int carry_g = -2;
        int wide_zone = 0;
        // the interval of every pair of NEIGHBOURS once (its upper end bounds the lower centre's zone, its lower end the upper
        // centre's): the sums of the iteration are spent, their LDS holds the ends
        // ... and, while there are threads to spare, of the pairs two, three and four apart as well: where centres crowd (relocated
        // centres side by side in a sparse tail) a centre's zone is the intersection over several of them, and one thread working
        // through them one after the other holds up the whole step
        double *zhi = reinterpret_cast<double *>(sum_o), *zlo = reinterpret_cast<double *>(cnt_o);
        const int zd = ku > 0 ? max(1, min(min(4, NT / max(ku, 1)), NNC_KMAX / max(ku, 1))) : 1; // distances kept in LDS
        for (int t = tid; t < zd * ku; t += NT) {
            const int d = t / ku, p = t - d * ku, q = p + d + 1;
            if (q < ku) {
                const KmZone z = km_pair_zone((double)cs[p], (double)cs[q], xb); // (distinct values in order: delta > 0)
                zhi[d * ku + p] = z.hi; zlo[d * ku + q] = z.lo;
            }
        }
        FSTAMP(13);
        FIN_SYNC();
        FSTAMP(14);
        for (int rd = 0; rd < rounds; rd++) { // (the raw G_p / H_p go to gcell / hcell; the scans below run over them in place)
            const int p = rd * NT + tid;
            int gp = -2, hp_ = G + 1;
            if (p < ku) {
                const double cp = (double)cs[p];
                double right = p + 1 < ku ? zhi[p] : INFINITY, left = p > 0 ? zlo[p] : -INFINITY;
                // pairs further apart matter only where centres crowd; and there only the first few: a zone end is a minimum
                // over pairs, so stopping early leaves it too far out -- a wider zone, a few more samples evaluated exactly --
                // never wrong.  Without the cap a centre next to one float32 cannot tell it from goes through every pair up
                // to a millimetre away: eighty evaluations one after the other behind a mass relocation (47 us of one step).
                for (int q = p + 2; q < ku && q - p <= KM_ZONE_REACH; q++) {
                    const double cq = (double)cs[q];
                    const double mid = 0.5 * (cp + cq);
                    const double delta = cq - cp;
                    if (mid - km_pair_slack(delta, xb) >= right) break; // every later pair's interval ends further up still
                    const int d = q - p - 1;
                    if (delta > 0.0) right = fmin(right, d < zd ? zhi[d * ku + p] : km_pair_zone(cp, cq, xb).hi);
                }
                for (int q = p - 2; q >= 0 && p - q <= KM_ZONE_REACH; q--) {
                    const double cq = (double)cs[q];
                    const double mid = 0.5 * (cp + cq);
                    const double delta = cp - cq;
                    if (mid + km_pair_slack(delta, xb) <= left) break;
                    const int d = p - q - 1;
                    if (delta > 0.0) left = fmax(left, d < zd ? zlo[d * ku + p] : km_pair_zone(cq, cp, xb).lo);
                }
                tab->zl[p] = left; tab->zr[p] = right;
                ws->bnd.zl[p] = left; ws->bnd.zr[p] = right;
                // a zone wide enough to hold thousands of samples (two centres float32 can hardly tell apart): the coming pass will
                // publish a long undecided stretch, and every wave of it had better look at the tile queue when its own work is done
                // An UNDECIDED interval wide enough to hold thousands of samples (two centres float32 can hardly tell apart: what
                // neither neighbour can be ruled out on is the overlap of their zones, zr[p] - zl[p + 1], here by the pair's own
                // interval): the coming pass will publish a long stretch, and its waves had better look at the tile queue -- twice
                // (k_bounds).  (Until round 3 the test was on the centre's whole zone, its cell included: true for every centre,
                // so every wave of every pass paid the look.)
                // (announced from sixteen tiles' worth up, at four times the mean density: the announcement costs the pass some
                // 8 us of waiting at the queue; a publisher left alone with fewer tiles than that costs less)
                if (p + 1 < ku && (zhi[p] - zlo[p + 1]) * (double)n_tot * 4.0 > 16.0 * (double)KM_TILE * ((double)p_hi - (double)p_lo)) wide_zone = 1;
                gp = G - 1; hp_ = 0;
                if (inv > 0.0 && !lazy) {
                    const double qa = (right - lo) / ra; // may be +-inf
                    // largest g with g <= qa; the relative slack covers the rounding of the quotient
                    gp = (qa >= (double)(G - 1)) ? (G - 1) : (qa < 0.0 ? -1 : (int)(qa * (1.0 + 1e-12)));
                    if (gp > G - 1) gp = G - 1;
                    const double qb = ((left - lo) / rb - 1.0) * (1.0 - 1e-12);
                    // smallest g with g >= qb
                    if (qb <= 0.0) hp_ = 0;
                    else if (qb >= (double)G) hp_ = G;
                    else { hp_ = (int)qb; if ((double)hp_ < qb) hp_++; }
                }
            }
            if (p < ku) { gcell[p] = gp; hcell[p] = hp_; }
        }
        FSTAMP(15);
        // (2: the coming pass waits at the queue until every wave has searched; 1: one look.  A wide interval appears out of
        // nothing only where centres were just placed -- the first pass, the pass behind a relocation: two of them a few ulps
        // apart; one that grows as two centres drift together is seen a pass late through the records it published, and while it
        // is still a few tiles long that costs little)
        const bool placed = mode == FIN_INIT || resume || settled_event;
        if (FIN_OR(6, wide_zone) && tid == 0 && placed) ws->help_hint = 2;
        FIN_SYNC();
        FSTAMP(11);
        if (settled_event && ftr && tid == 0) ftr[25] = ftr[11];
        if (!lazy) { // (the rank-boundary iterations do not use the cell side of the zones: k_cells works it out on demand)
        // prefix max of G_p and suffix min of H_p, side by side (one pair of barriers for both)
        int *wave_g = reinterpret_cast<int *>(wave_a);
        int *wave_h = reinterpret_cast<int *>(wave_b);
        int carry_h = G + 1;
        const int lane = tid & 63, myw = tid >> 6;
        for (int it = 0; it < rounds; it++) {
            const int rg = it, rh = rounds - 1 - it; // the prefix runs up the rounds, the suffix down
            int a = (rg * NT + tid < ku) ? gcell[rg * NT + tid] : -2;
            int b = (rh * NT + tid < ku) ? hcell[rh * NT + tid] : G + 1;
            for (int off = 1; off < 64; off <<= 1) {
                const int oa = __shfl_up(a, off), ob = __shfl_down(b, off);
                if (lane >= off) a = max(a, oa);
                if (lane + off < 64) b = min(b, ob);
            }
            if (lane == 63) wave_g[myw] = a;
            if (lane == 0) wave_h[myw] = b;
            FIN_SYNC();
            int pre = carry_g, totg = carry_g, suf = carry_h, toth = carry_h;
            for (int w = 0; w < NT / 64; w++) {
                const int gv = wave_g[w], hv = wave_h[w];
                if (w < myw) pre = max(pre, gv);
                totg = max(totg, gv);
                if (w > myw) suf = min(suf, hv);
                toth = min(toth, hv);
            }
            a = max(a, pre);
            b = min(b, suf);
            const int pg = rg * NT + tid, ph = rh * NT + tid;
            if (pg < ku) gcell[pg] = a;
            if (ph < ku) hcell[ph] = b;
            carry_g = totg; carry_h = toth;
            FIN_SYNC();
        }
        }
    }
    FSTAMP(5);
    FSTAMP(6);
    // ---- the cell table itself is built by k_cells (many workgroups: one CU is VALU-bound on it)
    if (!lazy) for (int p = tid; p < ku; p += NT) { tab->gc[p] = gcell[p]; tab->hc[p] = hcell[p]; }
    if (tid == 0) { tab->n_ovf = 0; ws->cells_pending = (ONEWAVE || lazy) ? 0 : 1; ku_out[0] = ku; ku_out[1] = cur; }
    FSTAMP(7);
    if (settled_event && ftr && tid == 0) ftr[26] = ftr[7];
#undef FSTAMP
#undef FIN_QUEUE_RESET
#undef FIN_LEAVE
#undef FIN_SYNC
#undef FIN_OR
#undef FIN_AND
#undef FIN_COUNT
    return true;
}


// cell g -> candidate range [first p with gc[p] >= g, last p with hc[p] <= g] (both monotone in g), packed.  The first
// and last cells are open-ended: everything below lo / above hi is clamped into them.  Crowded cells go to the side
// list (*n_ovf, ovf[]; readers clamp the count to KM_OVF_MAX).
__device__ __forceinline__ uint16_t km_cell_entry(int g, int G, int ku, const int *gcell, const int *hcell, int *n_ovf, unsigned *ovf)
{
    int l = 0, h = ku - 1;
    while (l < h) { const int m = (l + h) >> 1; if (gcell[m] >= g) h = m; else l = m + 1; }
    const int plo = l;
    l = 0; h = ku - 1;
    while (l < h) { const int m = (l + h + 1) >> 1; if (hcell[m] <= g) l = m; else h = m - 1; }
    const int phi = l;
    int lo_p = (g == 0) ? 0 : plo;       // below lo nothing exists, but keep cell 0 / G-1 conservative
    int hi_p = (g == G - 1) ? ku - 1 : phi;
    if (hi_p < lo_p) { lo_p = 0; hi_p = ku - 1; }
    int c = hi_p - lo_p;
    int field = lo_p;
    if (c >= KM_CNT_SAT) {
        c = KM_CNT_SAT;
        const int idx = atomicAdd(n_ovf, 1);
        if (idx < KM_OVF_MAX) { ovf[idx] = (unsigned)lo_p | ((unsigned)hi_p << 16); field = idx; }
        else field = (int)KM_OVF_ALL;
    }
    return (uint16_t)(field | (c << KM_P_BITS));
}

// FUSED: NT = 64 busy threads inside a KM_THREADS workgroup.  The helpers sleep on an LDS flag while the first wave
// runs the body, then all of them turn the zones into the cell table (a few cells per thread) -- no k_cells launch.
// The last launch of a batch may carry the host's look-in (nnc_kmeans_iterate_publish): once the state is final the
// status block and the ticket go out here when no k_cells launch follows to carry them.
// MASS: the body may settle mass empty-cluster events as well (kl_relocate_mass; nnc_kmeans_params.flags & NNC_KM_MASS_IN_PLACE).  A
// variant of its own: the code is large, and code that runs once a launch arrives cold -- with it in the default kernel the small
// events the finalize step settles in place went from 41-55 to 59-65 us.
// (the kernel's body as a device function: k_finalize runs it, and so does the launch that runs the relocation chain's selection in
// front of it, k_reloc_select_finalize)
template <int NT, bool FUSED, bool MASS = false>
__device__ __forceinline__ void km_finalize_kernel(KmWs *__restrict__ ws, int mode, int resume,
                                                   nnc_kmeans_status *host_st, unsigned long long *host_ticket,
                                                   unsigned long long ticket, int lazy, const float *__restrict__ reloc_xs,
                                                   long long reloc_n)
{
    __shared__ int gcell[NNC_KMAX], hcell[NNC_KMAX]; // per centre: last cell it can open / first cell it can close
    __shared__ int fin_go, fin_kc[2], fin_novf; // fin_kc: {distinct centres, current table} from the body
    __shared__ int fin_asked;
    const int tid = threadIdx.x;
    const int cond = lazy & 2; // enqueued behind k_lloyd in case it hands an iteration over (ws->wide): nothing to do otherwise
    const int k_hint = lazy >> 8; // the number of centres as the launcher knows it (0: not told): bounds the first round of loads
    lazy &= 1;
    if (cond) {
        if (tid == 0) fin_asked = ws->wide;
        __syncthreads();
    }
    if (cond && !fin_asked) {
        // (fall through to the look-in)
    } else if (FUSED) {
        const int glog2 = ws->glog2; // (constant over the fit: the helpers may read it at once)
        if (tid == 0) { fin_go = 0; fin_novf = 0; }
        __syncthreads();
        if (tid < NT) {
            const bool built = km_finalize_body<NT, FUSED>(ws, mode, resume, gcell, hcell, fin_kc, lazy != 0, reloc_xs, reloc_n, k_hint);
            if (tid == 0) __hip_atomic_store(&fin_go, (built && !lazy) ? 1 : 2, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        int go; // every path of the body ends in the store above, so the wait is bounded by the body's run time
        while ((go = __hip_atomic_load(&fin_go, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)) == 0) __builtin_amdgcn_s_sleep(1);
        if (go == 1) {
            KmTab *tab = &ws->tab[fin_kc[1]];
            const int G = 1 << glog2, ku = fin_kc[0];
            for (int g = tid; g < G; g += KM_THREADS) tab->cell[g] = km_cell_entry(g, G, ku, gcell, hcell, &fin_novf, tab->ovf);
            __syncthreads();
            if (tid == 0) tab->n_ovf = fin_novf;
        }
    } else {
        km_finalize_body<NT, false, MASS>(ws, mode, resume, gcell, hcell, fin_kc, lazy != 0, reloc_xs, reloc_n, k_hint);
    }
    if (cond && fin_asked) { // the iteration k_lloyd handed over has been run (or has paused): the loop may go on
        __syncthreads();
        if (tid == 0) { ws->wide = 0; ws->kl_budget = ws->kl_budget - 1; ws->kl_stats[4] += 1; }
    }
    if (host_st) {
        __syncthreads();
        if (tid == 0) {
            *host_st = ws->st;
            __threadfence_system();
            *reinterpret_cast<volatile unsigned long long *>(host_ticket) = ticket;
        }
    }
}

template <int NT, bool FUSED, bool MASS = false>
__global__ __launch_bounds__(FUSED ? KM_THREADS : NT) void k_finalize(KmWs *__restrict__ ws, int mode, int resume,
                                                                      nnc_kmeans_status *host_st, unsigned long long *host_ticket,
                                                                      unsigned long long ticket, int lazy, const float *__restrict__ reloc_xs,
                                                                      long long reloc_n)
{
    km_finalize_kernel<NT, FUSED, MASS>(ws, mode, resume, host_st, host_ticket, ticket, lazy, reloc_xs, reloc_n);
}

// A whole batch of Lloyd iterations in ONE launch, for fits with few centres on a sorted vector (k <= KM_FUSE_KMAX, rank-boundary
// form): one workgroup; per iteration its sixteen waves locate the cluster boundaries and add up the sums (km_bounds_wave), then
// the first wave runs the finalize body; until the fit stops, pauses (an empty cluster: the host relocates) or `iters`
// iterations are through.  What the two-launch form pays per iteration in launches and kernel boundaries (a third of the time
// of a short layer's fit) is a workgroup barrier and a cache fence here.  The look-in rides on the launch.
__global__ __launch_bounds__(KM_THREADS) void k_fit_small(const float *__restrict__ xs, long long n, KmWs *__restrict__ ws,
                                                         const long long *__restrict__ pblk, int iters, nnc_kmeans_status *host_st,
                                                         unsigned long long *host_ticket, unsigned long long ticket)
{
    __shared__ int gcell[NNC_KMAX], hcell[NNC_KMAX];
    __shared__ int fin_kc[2];
    const int tid = threadIdx.x, lane = tid & 63, wv = uni_i(tid >> 6);
    const float mean = ws->p.x_mean;
    const int Sft = ws->p.fix_shift;
    for (int it = 0; it < iters; it++) {
        // (everything below reads what the first wave wrote in the previous round: the fence + barrier pair at the end of a
        // round writes its stores through and drops this CU's cached copies)
        if (ws->st.done | ws->st.paused) break; // the same for every thread
        const KmTab *tab = &ws->tab[ws->cur];
        const int ku = ws->bnd.ku;
        KmBndSrc src;
        src.zr = ws->bnd.zr; src.zl = ws->bnd.zl; src.cand = ws->bnd.cand; src.orig = ws->bnd.orig; src.ku = &ws->bnd.ku;
        bool published = false;
        int qn_seen = 0;
        for (int j = wv; j < ku; j += KM_THREADS / 64) {
            KmBndPre<(KM_FUSE_KMAX + 63) / 64> pre;
            km_bounds_preload<(KM_FUSE_KMAX + 63) / 64>(pre, j, lane, ws, src);
            published |= km_bounds_wave<(KM_FUSE_KMAX + 63) / 64>(j, lane, xs, n, ws, tab, src, mean, Sft, pblk, &qn_seen, pre);
        }
        if (published || ws->help_hint || uni_i(qn_seen) > 0) km_bounds_help(lane, xs, ws, tab, mean, Sft);
        // The sums went out as device-scope atomics (performed in L2); the first wave must not read them from a line its CU
        // still holds from the last round: drop the CU's cached copies (acquire), no write-back needed.
        __syncthreads();
        if (tid < 64) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            km_finalize_body<64, true>(ws, FIN_FROM_SHARDS, 0, gcell, hcell, fin_kc, true);
        }
        // What the first wave stored is read by the other waves of this workgroup only: same CU, same L1 -- a workgroup-scope
        // release / acquire (the stores have left the wave) is all it takes.
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
    if (host_st && tid == 0) {
        *host_st = ws->st;
        __threadfence_system();
        *reinterpret_cast<volatile unsigned long long *>(host_ticket) = ticket;
    }
}

// cell g -> candidate range [first p with gc[p] >= g, last p with hc[p] <= g] (both monotone in g);
// one cell per thread.  The first and last cells are open-ended: everything below lo / above hi
// is clamped into them.
// force: build the table of tab[cur ^ which] whatever the pending flag says (the rank-boundary iterations do not need the
// cell table, so they leave it stale; the kernels that look samples up -- labels, relocation candidates -- ask for it).
__global__ void k_cells_prepare(KmWs *__restrict__ ws, int which) { ws->tab[ws->cur ^ (which & 1)].n_ovf = 0; }

__device__ __forceinline__ void km_cells_body(KmWs *__restrict__ ws, nnc_kmeans_status *host_st, unsigned long long *host_ticket,
                                              unsigned long long ticket, int force, int which, int spec, const int bid)
{
    __shared__ int gcell[NNC_KMAX], hcell[NNC_KMAX];
    if (spec && !ws->spec_go) return; // (enqueued in case of an empty-cluster event: there is none to settle)
    // the last launch of a batch may carry the host's look-in (nnc_kmeans_iterate_publish): the state is final once
    // k_finalize is through, so the status block and the ticket go out here, without a launch of their own
    if (host_st && bid == 0 && threadIdx.x == 0) {
        *host_st = ws->st;
        __threadfence_system();
        *reinterpret_cast<volatile unsigned long long *>(host_ticket) = ticket;
    }
    if (!force && !ws->cells_pending) return;
    const int G = 1 << ws->glog2;
    const int g = bid * KM_THREADS + threadIdx.x;
    if ((int)(bid * KM_THREADS) >= G) return;
    KmTab *tab = &ws->tab[ws->cur ^ (which & 1)];
    const int ku = force ? tab->ku : ws->ku_cur;
    if (!force) {
        for (int p = threadIdx.x; p < ku; p += KM_THREADS) { gcell[p] = tab->gc[p]; hcell[p] = tab->hc[p]; }
        __syncthreads();
    } else {
        // on demand: the cell side of the zones (see km_finalize_body): centre p can open cells <= G_p and close cells >= H_p,
        // rounded outwards, then made monotone (prefix max / suffix min); every workgroup works it out for itself
        const double U = 5.9604644775390625e-08;
        const double lo = (double)ws->p.lo, inv = (double)ws->inv;
        const double ra = inv > 0.0 ? (1.0 - 4.0 * U) / inv * (1.0 - 4.0 * U) : 0.0;
        const double rb = inv > 0.0 ? (1.0 + 4.0 * U) / inv * (1.0 + 4.0 * U) : 0.0;
        for (int p = threadIdx.x; p < ku; p += KM_THREADS) {
            const double left = tab->zl[p], right = tab->zr[p];
            int gp = G - 1, hp_ = 0;
            if (inv > 0.0) {
                const double qa = (right - lo) / ra;
                gp = (qa >= (double)(G - 1)) ? (G - 1) : (qa < 0.0 ? -1 : (int)(qa * (1.0 + 1e-12)));
                if (gp > G - 1) gp = G - 1;
                const double qb = ((left - lo) / rb - 1.0) * (1.0 - 1e-12);
                if (qb <= 0.0) hp_ = 0;
                else if (qb >= (double)G) hp_ = G;
                else { hp_ = (int)qb; if ((double)hp_ < qb) hp_++; }
            }
            gcell[p] = gp; hcell[p] = hp_;
        }
        __syncthreads();
        for (int off = 1; off < ku; off <<= 1) { // Hillis-Steele, two entries a thread (ku <= NNC_KMAX < 2 * KM_THREADS)
            int a[2], b[2];
            for (int r = 0; r < 2; r++) {
                const int q = threadIdx.x + r * KM_THREADS;
                a[r] = (q < ku) ? ((q >= off) ? max(gcell[q], gcell[q - off]) : gcell[q]) : 0;
                b[r] = (q < ku) ? ((q + off < ku) ? min(hcell[q], hcell[q + off]) : hcell[q]) : 0;
            }
            __syncthreads();
            for (int r = 0; r < 2; r++) {
                const int q = threadIdx.x + r * KM_THREADS;
                if (q < ku) { gcell[q] = a[r]; hcell[q] = b[r]; }
            }
            __syncthreads();
        }
    }
    if (g >= G) return;
    tab->cell[g] = km_cell_entry(g, G, ku, gcell, hcell, &tab->n_ovf, tab->ovf);
}

__global__ __launch_bounds__(KM_THREADS) void k_cells(KmWs *__restrict__ ws, nnc_kmeans_status *host_st,
                                                      unsigned long long *host_ticket, unsigned long long ticket,
                                                      int force = 0, int which = 0, int spec = 0)
{
    km_cells_body(ws, host_st, host_ticket, ticket, force, which, spec, (int)blockIdx.x);
}


static bool km_fused(const nnc_kmeans_params *p)
{
    int glog2, rlog2;
    km_defaults(p, &glog2, &rlog2);
    return p->k <= KM_FUSE_KMAX && glog2 <= KM_FUSE_GLOG2;
}

// few centres on a sorted vector with prefix sums: whole batches of iterations run as one launch of one workgroup (k_fit_small)
static bool km_one_launch_fit(const nnc_kmeans_params *p, const float *x)
{
    return p->prefix_dev && p->n > 0 && p->n == p->n_total && km_fused(p) && (reinterpret_cast<uintptr_t>(x) & 15) == 0;
}

// The cell table of tab[cur ^ which], for the kernels that look samples up, when the iterations did not keep it current.
static int km_ensure_cells(KmWs *w, const nnc_kmeans_params *p, int which, void *stream)
{
    if (!p->prefix_dev) return NNC_OK; // streaming iterations keep it current
    hipLaunchKernelGGL(k_cells_prepare, dim3(1), dim3(1), 0, S(stream), w, which);
    LAUNCHCHK("k_cells_prepare");
    hipLaunchKernelGGL(k_cells, dim3(KM_GMAX / KM_THREADS), dim3(KM_THREADS), 0, S(stream), w, (nnc_kmeans_status *)nullptr,
                       (unsigned long long *)nullptr, 0ull, 1, which);
    LAUNCHCHK("k_cells");
    return NNC_OK;
}

// reloc_xs: the value-sorted vector, for callers that want the finalize step of a rank-boundary iteration to settle small
// empty-cluster events itself (km_finalize_relocate: nnc_kmeans_fit); nullptr: every event pauses (what nnc_kmeans_iterate shows)
int km_launch_finalize(KmWs *w, const nnc_kmeans_params *p, int mode, int resume, void *stream, void *host_mapped,
                       uint64_t ticket, bool cond, const float *reloc_xs)
{
    // p == nullptr: the caller does not know the fit's parameters (nnc_kmeans_finalize): full width, k_cells builds the table
    const int k = p ? p->k : 0;
    bool fused = false;
    if (p && mode != FIN_PACK_ONLY) fused = km_fused(p);
    const bool lazy = p && p->prefix_dev; // rank-boundary iterations: nobody reads the cell table between two of them
    const bool cells = mode != FIN_PACK_ONLY && !fused && !lazy;
    unsigned char *hb = reinterpret_cast<unsigned char *>(host_mapped);
    nnc_kmeans_status *hs = reinterpret_cast<nnc_kmeans_status *>(hb);
    unsigned long long *ht = reinterpret_cast<unsigned long long *>(hb ? hb + sizeof(nnc_kmeans_status) : nullptr);
    nnc_kmeans_status *fs = cells ? nullptr : hs; // the look-in rides on the last launch
#define KM_LAUNCH_FIN(NT, FUSED, THREADS) NNC_LAUNCH_PROF(NNC_PROF_FINALIZE, (k_finalize<NT, FUSED>), dim3(1), dim3(THREADS), 0, S(stream), w, mode, resume, fs, ht, (unsigned long long)ticket, (lazy ? 1 : 0) | (cond ? 2 : 0) | (k > 0 ? k << 8 : 0), reloc_xs, (long long)(p ? p->n : 0))
#define KM_LAUNCH_FIN_MASS(NT, THREADS) NNC_LAUNCH_PROF(NNC_PROF_FINALIZE, (k_finalize<NT, false, true>), dim3(1), dim3(THREADS), 0, S(stream), w, mode, resume, fs, ht, (unsigned long long)ticket, (lazy ? 1 : 0) | (cond ? 2 : 0) | (k > 0 ? k << 8 : 0), reloc_xs, (long long)(p ? p->n : 0))
    const bool mass = p && (p->flags & NNC_KM_MASS_IN_PLACE) && reloc_xs && !fused && k > 64;
    if (fused) KM_LAUNCH_FIN(64, true, KM_THREADS);
    else if (k > 0 && k <= 64) KM_LAUNCH_FIN(64, false, 64);
    else if (k > 0 && k <= 256) { if (mass) KM_LAUNCH_FIN_MASS(256, 256); else KM_LAUNCH_FIN(256, false, 256); } // (measured at K = 257: one wave 23.6 us, four waves 14.6 / 17.7 us median / mean, sixteen 14.4 / 15.9)
    else { if (mass) KM_LAUNCH_FIN_MASS(KM_THREADS, KM_THREADS); else KM_LAUNCH_FIN(KM_THREADS, false, KM_THREADS); }
#undef KM_LAUNCH_FIN_MASS
#undef KM_LAUNCH_FIN
    LAUNCHCHK("k_finalize");
    if (cells) {
        hipLaunchKernelGGL(k_cells, dim3(KM_GMAX / KM_THREADS), dim3(KM_THREADS), 0, S(stream), w, hs, ht, (unsigned long long)ticket, 0, 0);
        LAUNCHCHK("k_cells");
    }
    return NNC_OK;
}

int km_check(void *ws, const nnc_kmeans_params *p, const char *who)
{
    if (!ws || !p) return fail(NNC_EINVAL, std::string(who) + ": null workspace/params");
    if (p->k < 1 || p->k > NNC_KMAX - 8) return fail(NNC_EINVAL, std::string(who) + ": k out of range");
    if (p->n < 0 || p->n_total < p->n) return fail(NNC_EINVAL, std::string(who) + ": bad n / n_total");
    if ((reinterpret_cast<uintptr_t>(ws) & 15) != 0) return fail(NNC_EINVAL, std::string(who) + ": workspace must be 16-byte aligned");
    return NNC_OK;
}

__device__ __forceinline__ void km_init_body(KmWs *ws, const nnc_kmeans_params &p, int glog2, int rlog2, float inv,
                                             const float *__restrict__ centers_init)
{
    const int tid = threadIdx.x;
    if (tid == 0) {
        ws->st.iter = 0; ws->st.done = 0; ws->st.paused = 0; ws->st.n_empty = 0;
        ws->st.shift_tot = 0.0f; ws->st.tol = p.tol; ws->st.k = p.k; ws->st.same_counts = 0;
        ws->st.reloc_ties = 0; ws->st.reloc_multi = 0; ws->st.n_relocated = 0; ws->st.n_unproven = 0; ws->st.n_in_place = 0; ws->st.reserved = 0; ws->spec_go = 0;
        ws->p = p; ws->cur = 0; ws->glog2 = glog2; ws->rlog2 = rlog2; ws->inv = inv; ws->reloc_fail = 0; ws->cells_pending = 0;
    }
    for (int j = tid; j < p.k; j += KM_THREADS) {
        float c = centers_init[j] - p.x_mean; // init -= X_mean (float32)
        ws->c[0][j] = c; ws->c[1][j] = c;
    }
    for (int i = tid; i < KM_NSHARD * NNC_KMAX; i += KM_THREADS) {
        (&ws->shard_sum[0][0])[i] = 0; (&ws->shard_cnt[0][0])[i] = 0;
    }
    for (int i = tid; i < 2 * NNC_KMAX; i += KM_THREADS) { ws->partials[i] = 0; ws->partials_local[i] = 0; }
    for (int i = tid; i < NNC_KMAX; i += KM_THREADS) { ws->prev_counts[i] = -1; ws->q_w0[i] = 0ull; ws->q_w1[i] = 0ull; ws->q_next[i] = 0; ws->hint_a[i] = -1; ws->hint_b[i] = -1; }
    for (int i = tid; i < 2 * NNC_KMAX; i += KM_THREADS) { ws->kl_hL[i] = 0.0f; ws->kl_hR[i] = 0.0f; }
    if (tid == 0) { ws->q_n = 0; ws->q_searched = 0; ws->help_hint = 0; ws->help_pad = 0; ws->wide = 0; ws->kl_budget = 0; }
    if (tid < 8) ws->kl_stats[tid] = 0;
    if (tid < 24) ws->kl_trace[tid] = 0ull;
}

__global__ __launch_bounds__(KM_THREADS) void k_km_init(KmWs *ws, nnc_kmeans_params p, int glog2, int rlog2, float inv,
                                                        const float *__restrict__ centers_init)
{
    km_init_body(ws, p, glog2, rlog2, inv, centers_init);
}

// ... and the table of the initial centres in the same launch, where the finalize step runs with the same sixteen waves and leaves
// the cell table alone (more than 256 centres on a sorted vector with prefix sums): one launch boundary less at the head of a fit
__global__ __launch_bounds__(KM_THREADS) void k_km_init_finalize(KmWs *ws, nnc_kmeans_params p, int glog2, int rlog2, float inv,
                                                                 const float *__restrict__ centers_init)
{
    km_init_body(ws, p, glog2, rlog2, inv, centers_init);
    __syncthreads();                                    // the workspace is written ...
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); // ... and read past this compute unit's L1
    km_finalize_kernel<KM_THREADS, false, false>(ws, FIN_INIT, 0, nullptr, nullptr, 0ull, 1 | (p.k << 8), nullptr, 0);
}

extern "C" int nnc_kmeans_init(void *ws, size_t ws_bytes, const nnc_kmeans_params *p, const float *centers_init_dev,
                               void *stream)
{
    int rc = km_check(ws, p, "nnc_kmeans_init");
    if (rc) return rc;
    if (!centers_init_dev) return fail(NNC_EINVAL, "nnc_kmeans_init: null centers");
    if (ws_bytes < nnc_kmeans_workspace_bytes(p->k)) return fail(NNC_ENOSPACE, "nnc_kmeans_init: workspace too small");
    // KMeans.fit refuses such input (sklearn/utils/validation.py: "Input X contains NaN" / "infinity"), and the kernels' searches
    // and windows assume an ordered vector: a NaN anywhere in the data makes the NumPy mean NaN, an infinity the range infinite
    if (!std::isfinite(p->x_mean) || !std::isfinite(p->tol) || !std::isfinite(p->lo) || !std::isfinite(p->hi))
        return fail(NNC_EINVAL, "nnc_kmeans_init: Input X contains NaN or infinity (mean / variance / range of the data are not finite)");
    int glog2, rlog2;
    km_defaults(p, &glog2, &rlog2);
    // cells per unit, a hair under G / (hi - lo) so that x~ = hi still lands in the last cell
    float inv = 0.0f;
    double range = (double)p->hi - (double)p->lo;
    if (range > 0.0) inv = (float)(((double)(1 << glog2)) / range * (1.0 - 1.0 / 1048576.0));
    if (!std::isfinite(inv)) inv = 0.0f;
    KmWs *w = reinterpret_cast<KmWs *>(ws);
#ifndef KM_INIT_TWO
    if (p->k > 256 && !km_fused(p) && p->prefix_dev) {
        hipLaunchKernelGGL(k_km_init_finalize, dim3(1), dim3(KM_THREADS), 0, S(stream), w, *p, glog2, rlog2, inv, centers_init_dev);
        LAUNCHCHK("k_km_init_finalize");
        return NNC_OK;
    }
#endif
    hipLaunchKernelGGL(k_km_init, dim3(1), dim3(KM_THREADS), 0, S(stream), w, *p, glog2, rlog2, inv, centers_init_dev);
    LAUNCHCHK("k_km_init");
    return km_launch_finalize(w, p, FIN_INIT, 0, stream);
}

// Replace the current centres (warm start, stepping through a recorded trajectory, centroid fine-tuning): clears done /
// paused, keeps the iteration count, rebuilds the search tables.
__global__ __launch_bounds__(KM_THREADS) void k_km_set_centers(KmWs *ws, const float *__restrict__ centers, int centred)
{
    const int tid = threadIdx.x;
    const int k = ws->p.k, cur = ws->cur;
    const float mean = ws->p.x_mean;
    for (int j = tid; j < k; j += KM_THREADS) ws->c[cur][j] = centred ? centers[j] : (centers[j] - mean);
    for (int j = tid; j < NNC_KMAX; j += KM_THREADS) ws->prev_counts[j] = -1;
    if (tid == 0) { ws->st.done = 0; ws->st.paused = 0; ws->st.n_empty = 0; ws->st.same_counts = 0; ws->reloc_fail = 0; }
}

extern "C" int nnc_kmeans_set_centers(void *ws, const nnc_kmeans_params *p, const float *centers_dev, int centred, void *stream)
{
    int rc = km_check(ws, p, "nnc_kmeans_set_centers");
    if (rc) return rc;
    if (!centers_dev) return fail(NNC_EINVAL, "nnc_kmeans_set_centers: null centers");
    KmWs *w = reinterpret_cast<KmWs *>(ws);
    hipLaunchKernelGGL(k_km_set_centers, dim3(1), dim3(KM_THREADS), 0, S(stream), w, centers_dev, centred);
    LAUNCHCHK("k_km_set_centers");
    return km_launch_finalize(w, p, FIN_INIT, 0, stream);
}

static int km_grid(int64_t n, size_t lds_bytes)
{
    int64_t blocks = ((n + 7) / 8 + KM_THREADS - 1) / KM_THREADS;
    if (blocks < 1) blocks = 1;
    int per_cu = (lds_bytes + 1024 <= 80 * 1024) ? 2 : 1; // two 1024-thread workgroups fit a CU if LDS allows
    int mult_q = 4; // workgroups per resident slot, in quarters
#ifdef NNC_DIAG
    static int mult_env = -1; // tuning knob (NNC_KM_GRID_QUARTERS)
    if (mult_env < 0) { const char *e = getenv("NNC_KM_GRID_QUARTERS"); mult_env = e ? atoi(e) : 4; if (mult_env < 1) mult_env = 4; }
    mult_q = mult_env;
#endif
    return (int)std::min<int64_t>(blocks, (int64_t)cu_count() * per_cu * mult_q / 4);
}

#ifdef NNC_DIAG
// diagnostic: shader clock = d(s_memtime) / d(s_memrealtime) * 100 MHz over a VALU spin loop
__global__ void k_debug_clock(int iters, float *out)
{
    float a = threadIdx.x * 1e-3f, b = 1.0001f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; i++) { a = a * b + 0.5f; b = b * 0.99999f + 1e-6f; }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = (float)((double)(t1 - t0) / (double)(r1 - r0) * 0.1); // GHz
        out[2 * blockIdx.x + 1] = (float)((double)(r1 - r0) * 0.01);              // us
    }
    if (a + b == 123.456f) out[0] = a;
}

extern "C" int nnc_debug_kl_trace(void *ws, unsigned long long *out8)
{
    if (!ws || !out8) return fail(NNC_EINVAL, "nnc_debug_kl_trace: null pointer");
    HIPCHK(hipMemcpy(out8, reinterpret_cast<KmWs *>(ws)->kl_trace, 24 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return NNC_OK;
}

extern "C" int nnc_debug_set_trace(unsigned long long *buf_dev)
{
    HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_km_trace), &buf_dev, sizeof(buf_dev)));
    unsigned long long *fin = buf_dev ? buf_dev + 6 * 1024 : nullptr; // finalize stamps live behind the workgroup / wave records (buffer: 8192 entries)
    HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_fin_trace), &fin, sizeof(fin)));
    return NNC_OK;
}

extern "C" int nnc_debug_clock(int blocks, int iters, float *out_dev, void *stream)
{
    hipLaunchKernelGGL(k_debug_clock, dim3(blocks), dim3(256), 0, S(stream), iters, out_dev);
    LAUNCHCHK("k_debug_clock");
    return NNC_OK;
}

static int g_ablation = 0;
static int g_deal = []() { const char *e = getenv("NNC_KM_DEAL"); return e ? atoi(e) : 1; }(); // tuning knob: 0 = contiguous ranges only
extern "C" int nnc_debug_set_ablation(int a) { g_ablation = a; return NNC_OK; }
#else
static const int g_ablation = 0, g_deal = 1;
#endif

extern "C" size_t nnc_kmeans_prefix_bytes(int64_t n)
{
    if (n < 0) return 0;
    const long long nblk = (n + KM_PB - 1) / KM_PB;
    return (size_t)(nblk + 2 + (nblk + 1 + KM_PG - 1) / KM_PG + 2 + 4 * (nblk + 1) + 4) * sizeof(long long);
}

extern "C" int nnc_kmeans_prefix_build(const float *x_sorted, const nnc_kmeans_params *p, int64_t *prefix_dev, void *stream)
{
    if (!p || !prefix_dev || p->n < 0 || (p->n > 0 && !x_sorted)) return fail(NNC_EINVAL, "nnc_kmeans_prefix_build: bad argument");
    if ((reinterpret_cast<uintptr_t>(x_sorted) & 15) != 0 || (reinterpret_cast<uintptr_t>(prefix_dev) & 7) != 0)
        return fail(NNC_EINVAL, "nnc_kmeans_prefix_build: x_sorted must be 16-byte aligned, prefix_dev 8-byte aligned");
    if (p->n >= ((int64_t)1 << 40)) return fail(NNC_EINVAL, "nnc_kmeans_prefix_build: n >= 2^40");
    const long long nblk = (p->n + KM_PB - 1) / KM_PB;
    long long *pb = reinterpret_cast<long long *>(prefix_dev);
    const int grid = (int)std::max<long long>(1, std::min<long long>((nblk + 3) / 4, (long long)cu_count() * 8));
    NNC_LAUNCH_PROF(NNC_PROF_PREFIX, k_prefix_blocks, dim3(grid), dim3(256), 0, S(stream), x_sorted, (long long)p->n, p->x_mean, p->fix_shift, pb);
    LAUNCHCHK("k_prefix_blocks");
    const long long ngroups = (nblk + 1 + KM_PG - 1) / KM_PG; // (entries 0 .. nblk)
    long long *pg = pb + nblk + 2;
    if (ngroups > 0) {
        hipLaunchKernelGGL(k_prefix_groups, dim3((unsigned)ngroups), dim3(KM_THREADS), 0, S(stream), pb, nblk, pg);
        LAUNCHCHK("k_prefix_groups");
    }
    hipLaunchKernelGGL(k_prefix_top, dim3(1), dim3(KM_THREADS), 0, S(stream), pg, ngroups);
    LAUNCHCHK("k_prefix_top");
    hipLaunchKernelGGL(k_prefix_fine, dim3((unsigned)((nblk + 1 + 255) / 256)), dim3(256), 0, S(stream), pb, (long long)p->n);
    LAUNCHCHK("k_prefix_fine");
    return NNC_OK;
}

// one pass of per-cluster sums / counts over this rank's vector: by rank boundaries if the caller attached block prefix sums of a
// sorted vector (p->prefix_dev), else the streaming kernel
int km_launch_accumulate(const float *x, KmWs *w, const nnc_kmeans_params *p, void *stream, int which)
{
    if (p->prefix_dev && p->n > 0) {
        if ((reinterpret_cast<uintptr_t>(x) & 15) != 0) return fail(NNC_EINVAL, "rank-boundary iteration: the sorted vector must be 16-byte aligned");
        const int nb = (p->k + 3) / 4; // one wave per centre (distinct centres <= k); long undecided stretches are shared by the waves that are still running
        const long long *pb = reinterpret_cast<const long long *>(p->prefix_dev);
#define KM_LAUNCH_BOUNDS(BR) NNC_LAUNCH_PROF(NNC_PROF_BOUNDS, (k_bounds<BR>), dim3(nb), dim3(256), 0, S(stream), x, (long long)p->n, w, which, pb)
        if (p->k <= 64) KM_LAUNCH_BOUNDS(1);
        else if (p->k <= 128) KM_LAUNCH_BOUNDS(2);
        else if (p->k <= 256) KM_LAUNCH_BOUNDS(4);
        else if (p->k <= 512) KM_LAUNCH_BOUNDS(8);
        else if (p->k <= 1024) KM_LAUNCH_BOUNDS(16);
        else KM_LAUNCH_BOUNDS(KM_BND_R);
#undef KM_LAUNCH_BOUNDS
        LAUNCHCHK("k_bounds");
        return NNC_OK;
    }
    int glog2, rlog2;
    km_defaults(p, &glog2, &rlog2);
    size_t lds = km_lds_bytes(p->k, glog2, rlog2, true);
    const bool vec = (reinterpret_cast<uintptr_t>(x) & 15) == 0;
    int grid = km_grid(p->n, lds);
    if (p->n == 0) return NNC_OK;
    // hipExtLaunchKernelGGL stamps the events at the kernel's own begin and end (not at the
    // command processor's arrival), so the difference is the launch's execution time
    hipEvent_t ev_a, ev_b;
    prof_take(NNC_PROF_ASSIGN_ACCUMULATE, &ev_a, &ev_b);
#define KM_LAUNCH_ACC(...) hipExtLaunchKernelGGL((k_assign<0, __VA_ARGS__>), dim3(grid), dim3(KM_THREADS), lds, S(stream), ev_a, ev_b, 0, x, p->n, w, which, (uint8_t *)nullptr, (float *)nullptr, (float *)nullptr, (unsigned long long *)nullptr, (const int *)nullptr)
    if (false) {}
#ifdef NNC_DIAG
    else if (vec && g_ablation == 1) KM_LAUNCH_ACC(true, uint8_t, 1);
    else if (vec && g_ablation == 2) KM_LAUNCH_ACC(true, uint8_t, 2);
    else if (vec && g_ablation == 3) KM_LAUNCH_ACC(true, uint8_t, 3);
#endif
    else if (vec && g_deal && (grid & 1) == 0 && (p->n / (4 * KM_THREADS)) >= 4 * (int64_t)grid) KM_LAUNCH_ACC(true, uint8_t, 0, true);
    else if (vec) KM_LAUNCH_ACC(true, uint8_t);
    else KM_LAUNCH_ACC(false, uint8_t);
#undef KM_LAUNCH_ACC
    LAUNCHCHK("k_assign<accumulate>");
    return NNC_OK;
}

// ---- the one-workgroup loop (nnc_lloyd.hpp) and, behind it, the wide pair in case it hands an iteration over --------------------
#define KL_LDS_LIMIT ((size_t)158 * 1024) // dynamic LDS k_lloyd may ask for (160 KiB a workgroup on gfx950, minus its static part)
static bool km_lloyd_ok(const nnc_kmeans_params *p, const float *x)
{
    if (!(p->prefix_dev && p->n > 0 && p->n == p->n_total && (reinterpret_cast<uintptr_t>(x) & 15) == 0)) return false;
    if (kl_lds_bytes((p->k + 7) & ~7) > KL_LDS_LIMIT) return false; // (the launch-per-iteration pair takes it)
    if (p->flags & NNC_KM_TWO_LAUNCH) return false;
    return (p->flags & NNC_KM_LOOP) || p->k <= NNC_KM_LOOP_KMAX; // (beyond that one compute unit's instruction rate is the bound: include/nnc.h)
}

// reloc: the loop settles empty-cluster events itself where it can (kl_relocate) -- for callers that would otherwise enqueue the
// relocation chain "in case" (nnc_kmeans_fit); a caller that wants to see every pause (nnc_kmeans_iterate) passes 0
static int km_launch_lloyd(const float *xs, KmWs *w, const nnc_kmeans_params *p, int budget_set, void *stream, int reloc = 0)
{
    const int kc = (p->k + 7) & ~7;
    const size_t lds = kl_lds_bytes(kc);
    const long long *pb = reinterpret_cast<const long long *>(p->prefix_dev);
    if (p->k <= 128)
        NNC_LAUNCH_PROF(NNC_PROF_LLOYD, (k_lloyd<256>), dim3(1), dim3(256), lds, S(stream), xs, (long long)p->n, w, pb, budget_set, kc,
                        (nnc_kmeans_status *)nullptr, (unsigned long long *)nullptr, 0ull, reloc);
    else // (sixteen waves leave 128 registers a thread and the kernel spills some six hundred; eight waves -- 256 registers, 88 spilled --
         // were measured slower all the same: K = 129: 1.04-1.19 of the launch-per-iteration time against 0.87-0.99, K = 257: 4.30 ms per
         // bench step against 3.94: the searches want the waves more than the registers)
        NNC_LAUNCH_PROF(NNC_PROF_LLOYD, (k_lloyd<KM_THREADS>), dim3(1), dim3(KM_THREADS), lds, S(stream), xs, (long long)p->n, w, pb, budget_set, kc,
                        (nnc_kmeans_status *)nullptr, (unsigned long long *)nullptr, 0ull, reloc);
    LAUNCHCHK("k_lloyd");
    return NNC_OK;
}

// one round: the loop runs until the fit stops, pauses, or an iteration needs the wide pass; k_bounds / k_finalize then run that one
// iteration (they return at once otherwise).  The look-in, if any, rides on the last launch.
static int km_launch_lloyd_round(const float *xs, KmWs *w, const nnc_kmeans_params *p, int budget_set, void *stream, void *host_mapped = nullptr,
                                 uint64_t ticket = 0, int reloc = 0)
{
    int rc;
    if ((rc = km_launch_lloyd(xs, w, p, budget_set, stream, reloc))) return rc;
    if ((rc = km_launch_accumulate(xs, w, p, stream, 4))) return rc;
    return km_launch_finalize(w, p, FIN_FROM_SHARDS, 0, stream, host_mapped, ticket, true);
}
#define KM_LLOYD_ROUNDS_MAX 32 // rounds one nnc_kmeans_iterate call enqueues at most (a round runs at least one iteration)
#define KM_LLOYD_ROUNDS 6      // rounds nnc_kmeans_fit enqueues per look-in

static std::atomic<int> g_lds_attr_set[NNC_MAX_DEVICES]; // function attributes are per device
int km_set_lds_attr()
{
    const int dev = current_device();
    if (g_lds_attr_set[dev].load(std::memory_order_acquire)) return NNC_OK;
    const int maxlds = 160 * 1024;
#define SETATTR(fn) HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&fn), hipFuncAttributeMaxDynamicSharedMemorySize, maxlds))
    SETATTR((k_assign<0, true, uint8_t>));
    SETATTR((k_assign<0, true, uint8_t, 0, true>));
#ifdef NNC_DIAG
    SETATTR((k_assign<0, true, uint8_t, 1>));
    SETATTR((k_assign<0, true, uint8_t, 2>));
    SETATTR((k_assign<0, true, uint8_t, 3>));
#endif
    SETATTR((k_assign<0, false, uint8_t>));
    SETATTR((k_assign<1, true, uint8_t>));
    SETATTR((k_assign<1, false, uint8_t>));
    SETATTR((k_assign<1, true, uint16_t>));
    SETATTR((k_assign<1, false, uint16_t>));
#undef SETATTR
    // (k_lloyd has a little static LDS of its own -- the workgroup votes -- so it cannot ask for all 160 KiB: KL_LDS_LIMIT leaves it
    // 2 KiB; K = NNC_KMAX - 8 needs kl_lds_bytes(1024) = 153 KiB, and km_lloyd_ok refuses a K whose arrays would not fit)
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_lloyd<256>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)KL_LDS_LIMIT));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_lloyd<KM_THREADS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)KL_LDS_LIMIT));
    g_lds_attr_set[dev].store(1, std::memory_order_release);
    return NNC_OK;
}

extern "C" int nnc_kmeans_accumulate(const float *x, void *ws, const nnc_kmeans_params *pp, void *stream)
{
    int rc = km_check(ws, pp, "nnc_kmeans_accumulate");
    if (rc) return rc;
    const nnc_kmeans_params p = *pp;
    if (p.n > 0 && !x) return fail(NNC_EINVAL, "nnc_kmeans_accumulate: null x");
    if ((rc = km_set_lds_attr())) return rc;
    KmWs *w = reinterpret_cast<KmWs *>(ws);
    if ((rc = km_launch_accumulate(x, w, &p, stream))) return rc;
    return km_launch_finalize(w, &p, FIN_PACK_ONLY, 0, stream);
}

extern "C" int64_t *nnc_kmeans_partials(void *ws)
{
    if (!ws) return nullptr;
    return reinterpret_cast<int64_t *>(reinterpret_cast<KmWs *>(ws)->partials);
}

extern "C" int nnc_kmeans_finalize(void *ws, int resume, void *stream)
{
    if (!ws) return fail(NNC_EINVAL, "nnc_kmeans_finalize: null workspace");
    return km_launch_finalize(reinterpret_cast<KmWs *>(ws), nullptr, FIN_FROM_PARTIALS, resume ? 1 : 0, stream);
}

// Where the iterations of the fit ran so far (device counters, reset by nnc_kmeans_init): out[0] iterations run by the one-workgroup
// loop, out[1] launches of it that had work, out[2] iterations in which centres changed places, out[3] iterations it handed to the
// wide pair (a very long undecided stretch, a search that did not settle), out[4] iterations the wide pair ran.  Synchronises the stream.
extern "C" int nnc_kmeans_loop_stats(void *ws, int32_t *out8, void *stream)
{
    if (!ws || !out8) return fail(NNC_EINVAL, "nnc_kmeans_loop_stats: null pointer");
    HIPCHK(hipMemcpyAsync(out8, reinterpret_cast<KmWs *>(ws)->kl_stats, 8 * sizeof(int32_t), hipMemcpyDeviceToHost, S(stream)));
    HIPCHK(hipStreamSynchronize(S(stream)));
    return NNC_OK;
}

extern "C" int nnc_kmeans_iterate(const float *x, void *ws, const nnc_kmeans_params *pp, int32_t iters, void *stream)
{
    int rc = km_check(ws, pp, "nnc_kmeans_iterate");
    if (rc) return rc;
    const nnc_kmeans_params p = *pp;
    if (p.n > 0 && !x) return fail(NNC_EINVAL, "nnc_kmeans_iterate: null x");
    if (p.n != p.n_total) return fail(NNC_EINVAL, "nnc_kmeans_iterate: sharded vector (n != n_total) needs accumulate / all-reduce / finalize");
    if ((rc = km_set_lds_attr())) return rc;
    KmWs *w = reinterpret_cast<KmWs *>(ws);
    if (km_lloyd_ok(&p, x)) {
        // the one-workgroup loop; every round runs at least one iteration (its own, or the one it hands to the wide pair behind it)
        const int rounds = std::min<int>(iters, KM_LLOYD_ROUNDS_MAX);
        for (int r = 0; r < rounds; r++)
            if ((rc = km_launch_lloyd_round(x, w, &p, r == 0 ? (int)iters : -1, stream))) return rc;
        return NNC_OK;
    }
    if (km_one_launch_fit(&p, x)) {
        if (iters < 1) return NNC_OK;
        hipLaunchKernelGGL(k_fit_small, dim3(1), dim3(KM_THREADS), 0, S(stream), x, (long long)p.n, w, reinterpret_cast<const long long *>(p.prefix_dev),
                           (int)iters, (nnc_kmeans_status *)nullptr, (unsigned long long *)nullptr, 0ull);
        LAUNCHCHK("k_fit_small");
        return NNC_OK;
    }
    for (int i = 0; i < iters; i++) {
        if ((rc = km_launch_accumulate(x, w, &p, stream))) return rc;
        if ((rc = km_launch_finalize(w, &p, FIN_FROM_SHARDS, 0, stream))) return rc;
    }
    return NNC_OK;
}

// index histogram of the E-step on the current (which = 0) or previous (which = 1) centres, from the per-cluster
// counts of one more streaming pass (any order of the weights gives the same counts, so the value-sorted copy may
// be used: far cheaper than counting the labels of the original vector)
__global__ __launch_bounds__(KM_THREADS) void k_counts_from_shards(KmWs *__restrict__ ws, int which, long long *__restrict__ counts)
{
    const KmTab *tab = &ws->tab[ws->cur ^ (which & 1)];
    const int k = ws->p.k, ku = tab->ku;
    {   // the tile queue of the counting pass (k_bounds) is spent
        const int qn = min(ws->q_n, (int)NNC_KMAX);
        for (int r = threadIdx.x; r < qn; r += KM_THREADS) { ws->q_w0[r] = 0ull; ws->q_w1[r] = 0ull; ws->q_next[r] = 0; }
        __syncthreads();
        if (threadIdx.x == 0) ws->q_n = 0;
    }
    for (int j = threadIdx.x; j < k; j += KM_THREADS) counts[j] = 0; // duplicates of a centre own nothing
    __syncthreads();
    for (int p = threadIdx.x; p < ku; p += KM_THREADS) {
        unsigned long long c = 0;
        for (int sh = 0; sh < KM_NSHARD; sh++) {
            c += ws->shard_cnt[sh][p];
            ws->shard_sum[sh][p] = 0; ws->shard_cnt[sh][p] = 0;
        }
        counts[tab->orig[p]] = (long long)c;
    }
}

extern "C" int nnc_kmeans_label_counts(const float *x, void *ws, const nnc_kmeans_params *pp, int which, int64_t *counts_dev,
                                       void *stream)
{
    int rc = km_check(ws, pp, "nnc_kmeans_label_counts");
    if (rc) return rc;
    if (!counts_dev || (pp->n > 0 && !x)) return fail(NNC_EINVAL, "nnc_kmeans_label_counts: null pointer");
    if ((rc = km_set_lds_attr())) return rc;
    KmWs *w = reinterpret_cast<KmWs *>(ws);
    const nnc_kmeans_params p = *pp;
    if ((rc = km_launch_accumulate(x, w, &p, stream, 2 | (which & 1)))) return rc;
    hipLaunchKernelGGL(k_counts_from_shards, dim3(1), dim3(KM_THREADS), 0, S(stream), w, which, reinterpret_cast<long long *>(counts_dev));
    LAUNCHCHK("k_counts_from_shards");
    return NNC_OK;
}

static int km_iterate_publish_(const float *x, void *ws, const nnc_kmeans_params *pp, int32_t iters, void *host_mapped, uint64_t ticket, void *stream,
                               bool reloc_in_place, int lag = 0);
extern "C" int nnc_kmeans_iterate_publish(const float *x, void *ws, const nnc_kmeans_params *pp, int32_t iters, void *host_mapped,
                                          uint64_t ticket, void *stream)
{
    return km_iterate_publish_(x, ws, pp, iters, host_mapped, ticket, stream, false); // (a caller of this entry point sees every pause)
}

// reloc_in_place: x is the value-sorted vector and the finalize step may settle small empty-cluster events itself
// lag: the status goes to the host `lag` iterations BEFORE the end of the batch (launch-per-iteration form only), so that the caller
// can have the next batch enqueued while this one is still running (nnc_kmeans_fit)
static int km_iterate_publish_(const float *x, void *ws, const nnc_kmeans_params *pp, int32_t iters, void *host_mapped, uint64_t ticket, void *stream,
                               bool reloc_in_place, int lag)
{
    int rc = km_check(ws, pp, "nnc_kmeans_iterate_publish");
    if (rc) return rc;
    const nnc_kmeans_params p = *pp;
    if (p.n > 0 && !x) return fail(NNC_EINVAL, "nnc_kmeans_iterate_publish: null x");
    if (p.n != p.n_total) return fail(NNC_EINVAL, "nnc_kmeans_iterate_publish: sharded vector (n != n_total) needs accumulate / all-reduce / finalize");
    if (iters < 1 || !host_mapped || (reinterpret_cast<uintptr_t>(host_mapped) & 7) != 0)
        return fail(NNC_EINVAL, "nnc_kmeans_iterate_publish: iters < 1, or null / unaligned host pointer");
    if ((rc = km_set_lds_attr())) return rc;
    KmWs *w = reinterpret_cast<KmWs *>(ws);
    if (km_lloyd_ok(&p, x)) {
        const int rounds = std::min<int>(iters, KM_LLOYD_ROUNDS_MAX);
        for (int r = 0; r < rounds; r++)
            if ((rc = km_launch_lloyd_round(x, w, &p, r == 0 ? (int)iters : -1, stream, r == rounds - 1 ? host_mapped : nullptr, ticket))) return rc;
        return NNC_OK;
    }
    if (km_one_launch_fit(&p, x)) {
        unsigned char *hb = reinterpret_cast<unsigned char *>(host_mapped);
        hipLaunchKernelGGL(k_fit_small, dim3(1), dim3(KM_THREADS), 0, S(stream), x, (long long)p.n, w, reinterpret_cast<const long long *>(p.prefix_dev),
                           (int)iters, reinterpret_cast<nnc_kmeans_status *>(hb), reinterpret_cast<unsigned long long *>(hb + sizeof(nnc_kmeans_status)),
                           (unsigned long long)ticket);
        LAUNCHCHK("k_fit_small");
        return NNC_OK;
    }
    for (int i = 0; i < iters; i++) {
        if ((rc = km_launch_accumulate(x, w, &p, stream))) return rc;
        const bool last = i == std::max(0, (int)iters - 1 - lag);
        if ((rc = km_launch_finalize(w, &p, FIN_FROM_SHARDS, 0, stream, last ? host_mapped : nullptr, ticket, false,
                                     (reloc_in_place && p.prefix_dev) ? x : nullptr))) return rc;
    }
    return NNC_OK;
}

extern "C" int nnc_kmeans_status_async(void *ws, nnc_kmeans_status *host_out, void *stream)
{
    if (!ws || !host_out) return fail(NNC_EINVAL, "nnc_kmeans_status_async: null pointer");
    HIPCHK(hipMemcpyAsync(host_out, &reinterpret_cast<KmWs *>(ws)->st, sizeof(nnc_kmeans_status), hipMemcpyDeviceToHost, S(stream)));
    return NNC_OK;
}

// the status block straight into host memory the device can write (pinned / hipHostMalloc), then the ticket: the
// host spins on the ticket instead of paying a copy command plus a stream synchronisation
__global__ void k_publish_status(const KmWs *ws, nnc_kmeans_status *host_st, unsigned long long *host_ticket, unsigned long long ticket)
{
    *host_st = ws->st;
    __threadfence_system();
    *reinterpret_cast<volatile unsigned long long *>(host_ticket) = ticket;
}

extern "C" int nnc_kmeans_status_publish(void *ws, void *host_mapped, uint64_t ticket, void *stream)
{
    if (!ws || !host_mapped || (reinterpret_cast<uintptr_t>(host_mapped) & 7) != 0) return fail(NNC_EINVAL, "nnc_kmeans_status_publish: null or unaligned pointer");
    unsigned char *b = reinterpret_cast<unsigned char *>(host_mapped);
    hipLaunchKernelGGL(k_publish_status, dim3(1), dim3(1), 0, S(stream), reinterpret_cast<const KmWs *>(ws),
                       reinterpret_cast<nnc_kmeans_status *>(b), reinterpret_cast<unsigned long long *>(b + sizeof(nnc_kmeans_status)),
                       (unsigned long long)ticket);
    LAUNCHCHK("k_publish_status");
    return NNC_OK;
}

// A few bytes (scalars, a K-sized block) from device memory straight into host memory the device can write, then a ticket
// behind them: what a copy command plus a stream synchronisation would do, for the price of one small launch and a spin.
__global__ __launch_bounds__(256) void k_publish_bytes(const unsigned long long *__restrict__ src, unsigned long long *__restrict__ dst_host, int nwords,
                                                       unsigned long long *host_ticket, unsigned long long ticket)
{
    for (int i = threadIdx.x; i < nwords; i += 256) dst_host[i] = src[i];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) *reinterpret_cast<volatile unsigned long long *>(host_ticket) = ticket;
}

// (shared with nnc_layer.hip) nbytes a multiple of 8, both pointers 8-byte aligned
int nnc_publish_bytes_(const void *src_dev, void *dst_host_mapped, int nbytes, void *host_ticket, uint64_t ticket, void *stream)
{
    if (!src_dev || !dst_host_mapped || !host_ticket || nbytes < 0 || (nbytes & 7) || ((reinterpret_cast<uintptr_t>(src_dev) | reinterpret_cast<uintptr_t>(dst_host_mapped)) & 7))
        return fail(NNC_EINVAL, "publish: bad argument");
    hipLaunchKernelGGL(k_publish_bytes, dim3(1), dim3(256), 0, S(stream), reinterpret_cast<const unsigned long long *>(src_dev),
                       reinterpret_cast<unsigned long long *>(dst_host_mapped), nbytes / 8, reinterpret_cast<unsigned long long *>(host_ticket), (unsigned long long)ticket);
    LAUNCHCHK("k_publish_bytes");
    return NNC_OK;
}

__global__ void k_set_done(KmWs *ws, int code) { ws->st.done = code; }

extern "C" int nnc_kmeans_set_done(void *ws, int32_t done_code, void *stream)
{
    if (!ws) return fail(NNC_EINVAL, "nnc_kmeans_set_done: null workspace");
    hipLaunchKernelGGL(k_set_done, dim3(1), dim3(1), 0, S(stream), reinterpret_cast<KmWs *>(ws), done_code);
    LAUNCHCHK("k_set_done");
    return NNC_OK;
}

__global__ void k_get_centers(const KmWs *ws, int which, int centred, float *out)
{
    const int k = ws->p.k;
    const float *c = ws->c[ws->cur ^ (which & 1)];
    const float mean = ws->p.x_mean;
    for (int j = threadIdx.x; j < k; j += blockDim.x) out[j] = centred ? c[j] : (c[j] + mean);
}

extern "C" int nnc_kmeans_get_centers(void *ws, int which, int centred, float *out_dev, void *stream)
{
    if (!ws || !out_dev) return fail(NNC_EINVAL, "nnc_kmeans_get_centers: null pointer");
    hipLaunchKernelGGL(k_get_centers, dim3(1), dim3(256), 0, S(stream), reinterpret_cast<const KmWs *>(ws), which, centred, out_dev);
    LAUNCHCHK("k_get_centers");
    return NNC_OK;
}

static int km_assign(const float *x, void *ws, const nnc_kmeans_params *pp, int which, void *labels_out,
                     int label_bytes, float *quant_out, float *dist_out, int64_t *dist_hist4096_dev,
                     const int *n_dev, void *stream)
{
    int rc = km_check(ws, pp, "nnc_kmeans_assign");
    if (rc) return rc;
    const nnc_kmeans_params p = *pp;
    if (p.n > 0 && !x) return fail(NNC_EINVAL, "nnc_kmeans_assign: null x");
    if (labels_out && label_bytes != 1 && label_bytes != 2) return fail(NNC_EINVAL, "nnc_kmeans_assign: label_bytes must be 1 or 2");
    if (labels_out && label_bytes == 1 && p.k > 256) return fail(NNC_EINVAL, "nnc_kmeans_assign: k > 256 needs 2-byte labels");
    if (p.n == 0) return NNC_OK;
    if ((rc = km_set_lds_attr())) return rc;
    int glog2, rlog2;
    km_defaults(&p, &glog2, &rlog2);
    if (dist_hist4096_dev && !dist_out) return fail(NNC_EINVAL, "nnc_kmeans_assign: the distance histogram needs dist_out");
    size_t lds = km_lds_bytes(p.k, glog2, rlog2, false) + (dist_hist4096_dev ? 4096 * sizeof(unsigned) : 0);
    // the vector form stores labels 4 at a time (4 / 8 bytes) and values / distances as float4: every output counts
    const bool vec = ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(quant_out) | reinterpret_cast<uintptr_t>(dist_out)) & 15) == 0 &&
                     (reinterpret_cast<uintptr_t>(labels_out) & (size_t)(4 * label_bytes - 1)) == 0;
    int grid = km_grid(p.n, lds);
    KmWs *w = reinterpret_cast<KmWs *>(ws);
    if ((rc = km_ensure_cells(w, &p, which, stream))) return rc;
    unsigned long long *dh = reinterpret_cast<unsigned long long *>(dist_hist4096_dev);
    if (dh) HIPCHK(hipMemsetAsync(dh, 0, 4096 * sizeof(int64_t), S(stream)));
    if (label_bytes == 2) {
        if (vec) NNC_LAUNCH_PROF(NNC_PROF_ASSIGN_LABELS, (k_assign<1, true, uint16_t>), dim3(grid), dim3(KM_THREADS), lds, S(stream), x, p.n, w, which, reinterpret_cast<uint16_t *>(labels_out), quant_out, dist_out, dh, n_dev);
        else NNC_LAUNCH_PROF(NNC_PROF_ASSIGN_LABELS, (k_assign<1, false, uint16_t>), dim3(grid), dim3(KM_THREADS), lds, S(stream), x, p.n, w, which, reinterpret_cast<uint16_t *>(labels_out), quant_out, dist_out, dh, n_dev);
    } else {
        if (vec) NNC_LAUNCH_PROF(NNC_PROF_ASSIGN_LABELS, (k_assign<1, true, uint8_t>), dim3(grid), dim3(KM_THREADS), lds, S(stream), x, p.n, w, which, reinterpret_cast<uint8_t *>(labels_out), quant_out, dist_out, dh, n_dev);
        else NNC_LAUNCH_PROF(NNC_PROF_ASSIGN_LABELS, (k_assign<1, false, uint8_t>), dim3(grid), dim3(KM_THREADS), lds, S(stream), x, p.n, w, which, reinterpret_cast<uint8_t *>(labels_out), quant_out, dist_out, dh, n_dev);
    }
    LAUNCHCHK("k_assign<labels>");
    return NNC_OK;
}

extern "C" int nnc_kmeans_assign(const float *x, void *ws, const nnc_kmeans_params *pp, int which, void *labels_out,
                                 int label_bytes, float *quant_out, float *dist_out, int64_t *dist_hist4096_dev,
                                 void *stream)
{
    return km_assign(x, ws, pp, which, labels_out, label_bytes, quant_out, dist_out, dist_hist4096_dev, nullptr, stream);
}



// ======================================================================================
// 4b. farthest-sample selection for the empty-cluster relocation
//
// The n_empty samples with the largest squared distance to their own centre, out of millions:
// (1) 4096-bin histogram of 12 value bits of the (non-negative) float32 distances, starting with
// the top 12, refined by the next 12 / last 7 inside the cut bin while it is crowded, (2) the host
// picks the largest threshold that still leaves at least n_empty samples at or above it, (3) those
// samples are compacted as 64-bit keys (distance bits << 32 | global index) and the few
// survivors are sorted by the caller.  Keys are unique, so the outcome is
// deterministic: descending distance, equal distances by descending index.
// ======================================================================================
__global__ __launch_bounds__(256) void k_topm_hist(const float *__restrict__ d, int64_t n, int shift, unsigned mask, int pshift,
                                                   unsigned prefix, unsigned long long *__restrict__ hist)
{
    __shared__ unsigned h[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) h[i] = 0;
    __syncthreads();
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t nthreads = (int64_t)gridDim.x * blockDim.x;
    const int64_t nvec = ((reinterpret_cast<uintptr_t>(d) & 15) == 0) ? (n >> 2) : 0;
    const uint4 *d4 = reinterpret_cast<const uint4 *>(d);
    // most distances are tiny and share a handful of bins: count runs of equal bins in registers
    unsigned run_bin = 0xFFFFFFFFu, run_cnt = 0;
#define TH1(u) do { const unsigned u_ = (u); if (pshift < 0 || (u_ >> pshift) == prefix) { const unsigned b_ = (u_ >> shift) & mask; \
        if (b_ != run_bin) { if (run_cnt) atomicAdd(&h[run_bin], run_cnt); run_bin = b_; run_cnt = 0; } run_cnt++; } } while (0)
    for (int64_t v = tid; v < nvec; v += nthreads) {
        const uint4 a = d4[v];
        TH1(a.x); TH1(a.y); TH1(a.z); TH1(a.w);
    }
    for (int64_t i = (nvec << 2) + tid; i < n; i += nthreads) TH1(__float_as_uint(d[i]));
#undef TH1
    if (run_cnt) atomicAdd(&h[run_bin], run_cnt);
    __syncthreads();
    for (int i = threadIdx.x; i < 4096; i += 256)
        if (h[i]) atomicAdd(&hist[i], (unsigned long long)h[i]);
}

__global__ __launch_bounds__(256) void k_topm_compact(const float *__restrict__ d, const float *__restrict__ x, int64_t n,
                                                      unsigned thr_bits, long long *__restrict__ keys, long long cap,
                                                      unsigned long long *__restrict__ count)
{
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t nthreads = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = tid; i < n; i += nthreads) {
        const unsigned u = __float_as_uint(d[i]);
        if (u >= thr_bits) {
            const unsigned long long slot = atomicAdd(count, 1ull);
            if ((long long)slot < cap) keys[slot] = ((long long)u << 32) | (long long)f32_ordered_bits(x[i]);
        }
    }
}

extern "C" int nnc_topm_hist_f32(const float *d, int64_t n, int32_t shift, int32_t width, int32_t prefix_shift, uint32_t prefix,
                                 int64_t *hist4096_dev, void *stream)
{
    if (n < 0 || !hist4096_dev || (n > 0 && !d) || shift < 0 || shift > 31 || prefix_shift > 31 || width < 1 || width > 12)
        return fail(NNC_EINVAL, "nnc_topm_hist_f32: bad argument");
    HIPCHK(hipMemsetAsync(hist4096_dev, 0, 4096 * sizeof(int64_t), S(stream)));
    if (n == 0) return NNC_OK;
    int grid = stream_grid((n + 3) / 4, 256, 2);
    hipLaunchKernelGGL(k_topm_hist, dim3(grid), dim3(256), 0, S(stream), d, n, (int)shift, (1u << width) - 1u, (int)prefix_shift, (unsigned)prefix,
                       reinterpret_cast<unsigned long long *>(hist4096_dev));
    LAUNCHCHK("k_topm_hist");
    return NNC_OK;
}

extern "C" int nnc_topm_compact_f32(const float *d, const float *x, int64_t n, uint32_t thr_bits, int64_t *keys_dev,
                                    int64_t cap, int64_t *count_dev, void *stream)
{
    if (n < 0 || !keys_dev || !count_dev || cap <= 0 || (n > 0 && (!d || !x)))
        return fail(NNC_EINVAL, "nnc_topm_compact_f32: bad argument");
    HIPCHK(hipMemsetAsync(count_dev, 0, sizeof(int64_t), S(stream)));
    if (n == 0) return NNC_OK;
    int grid = stream_grid(n, 256 * 4, 8);
    hipLaunchKernelGGL(k_topm_compact, dim3(grid), dim3(256), 0, S(stream), d, x, n, (unsigned)thr_bits,
                       reinterpret_cast<long long *>(keys_dev), (long long)cap, reinterpret_cast<unsigned long long *>(count_dev));
    LAUNCHCHK("k_topm_compact");
    return NNC_OK;
}

// apply scikit-learn's relocation (_k_means_common.pyx:197-211) as additive edits to the
// (already global) per-cluster sums/counts: the i-th empty cluster (ascending index) takes the
// i-th key's sample; that sample's old cluster gives it up.  A key carries the sample's value
// (low 32 bits, order-preserving float bits); its cluster is re-derived with scikit-learn's exact
// float32 expression over all centres.  Every rank applies the same edits to its copy.
// Nothing happens when the largest distance is zero.
struct KmRelocLds { int empty_id[NNC_KMAX]; int wave_cnt[KM_THREADS / 64]; float cen[NNC_KMAX]; };

// first half: the current centres and the ordered list of the empty clusters into LDS (nothing here depends on the
// keys, so a caller that still has the selection to do runs it up front and hides the loads); returns n_empty
__device__ int km_relocate_prepare(const KmWs *__restrict__ ws, KmRelocLds *lds)
{
    int *empty_id = lds->empty_id, *wave_cnt = lds->wave_cnt;
    float *cen = lds->cen;
    const int tid = threadIdx.x;
    const int k = ws->p.k;
    const float *ccur = ws->c[ws->cur];
    for (int j = tid; j < k; j += KM_THREADS) cen[j] = ccur[j];
    // ordered list of the empty clusters
    const int rounds = (k + KM_THREADS - 1) / KM_THREADS;
    int carry = 0;
    for (int rd = 0; rd < rounds; rd++) {
        const int j = rd * KM_THREADS + tid;
        const int e = (j < k && ws->partials[k + j] == 0) ? 1 : 0;
        const unsigned long long bal = __ballot(e);
        const int before = __popcll(bal & ((1ull << (tid & 63)) - 1ull));
        if ((tid & 63) == 0) wave_cnt[tid >> 6] = __popcll(bal);
        __syncthreads();
        int pre = carry;
        for (int w = 0; w < (tid >> 6); w++) pre += wave_cnt[w];
        if (e) empty_id[pre + before] = j;
        int tot = carry;
        for (int w = 0; w < KM_THREADS / 64; w++) tot += wave_cnt[w];
        carry = tot;
        __syncthreads();
    }
    return carry;
}

// second half: the edits.  A sample's old cluster is scikit-learn's float32 arg-min over ALL centres (first minimum);
// with few samples a wave takes one (its lanes share the centres, the (distance, index) pairs are reduced
// lexicographically), with many a thread does.
__device__ void km_relocate_apply(KmWs *__restrict__ ws, const long long *__restrict__ keys, int nkeys, int n_empty, KmRelocLds *lds)
{
    const int *empty_id = lds->empty_id;
    const float *cen = lds->cen;
    const int tid = threadIdx.x, lane = tid & 63;
    const int k = ws->p.k;
    const int m = n_empty < nkeys ? n_empty : nkeys;
    if (m == 0 || (keys[0] >> 32) == 0) return;
    if (tid == 0) {
        // what scikit-learn leaves to numpy.argpartition: the pairing when several clusters are empty, and WHICH samples
        // are taken when two different ones tie at the cut (the runner-up key, if the caller passed it, tells)
        if (m > 1) ws->st.reloc_multi += 1;
        if (nkeys > m && keys[m] != 0 && (keys[m] >> 32) == (keys[m - 1] >> 32) && keys[m] != keys[m - 1]) ws->st.reloc_ties += 1; // (0: padding)
    }
    const float mean = ws->p.x_mean;
    const int Sft = ws->p.fix_shift;
    auto edit = [&](int i, float xc, int old) {
        const long long v = (long long)fix_f32(xc, Sft);
        const int nw = empty_id[i];
        atomicAdd(reinterpret_cast<unsigned long long *>(&ws->partials[old]), (unsigned long long)(-v));
        atomicAdd(reinterpret_cast<unsigned long long *>(&ws->partials[nw]), (unsigned long long)v);
        atomicAdd(reinterpret_cast<unsigned long long *>(&ws->partials[k + nw]), 1ull);
        atomicAdd(reinterpret_cast<unsigned long long *>(&ws->partials[k + old]), (unsigned long long)(-1ll));
    };
    if (m <= 4 * (KM_THREADS / 64)) {
        for (int i = tid >> 6; i < m; i += KM_THREADS / 64) { // wave-uniform
            const float xv = f32_from_ordered_bits((unsigned)(keys[i] & 0xFFFFFFFFll));
            const float xc = xv - mean;
            float best = INFINITY;
            int old = 0x7fffffff;
            for (int j = lane; j < k; j += 64) {
                const float cv = cen[j];
                const float dj = cv * cv + (-2.0f * (xc * cv));
                if (dj < best) { best = dj; old = j; }
            }
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
                const float ob = __shfl_xor(best, off);
                const int oo = __shfl_xor(old, off);
                if (ob < best || (ob == best && oo < old)) { best = ob; old = oo; }
            }
            if (old == 0x7fffffff) old = 0; // (every distance NaN: the scan below would have kept index 0)
            if (lane == 0) edit(i, xc, old);
        }
        return;
    }
    for (int i = tid; i < m; i += KM_THREADS) {
        const float xv = f32_from_ordered_bits((unsigned)(keys[i] & 0xFFFFFFFFll));
        const float xc = xv - mean;
        float best = cen[0] * cen[0] + (-2.0f * (xc * cen[0]));
        int old = 0;
        int j = 1;
        for (; j + 8 <= k; j += 8) { // eight LDS reads in flight
            float cv[8];
#pragma unroll
            for (int u = 0; u < 8; u++) cv[u] = cen[j + u];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const float dj = cv[u] * cv[u] + (-2.0f * (xc * cv[u]));
                if (dj < best) { best = dj; old = j + u; }
            }
        }
        for (; j < k; j++) {
            const float dj = cen[j] * cen[j] + (-2.0f * (xc * cen[j]));
            if (dj < best) { best = dj; old = j; }
        }
        edit(i, xc, old);
    }
}

__device__ void km_relocate_body(KmWs *__restrict__ ws, const long long *__restrict__ keys, int nkeys, KmRelocLds *lds)
{
    const int n_empty = km_relocate_prepare(ws, lds);
    km_relocate_apply(ws, keys, nkeys, n_empty, lds);
}

__global__ __launch_bounds__(KM_THREADS) void k_relocate(KmWs *__restrict__ ws, const long long *__restrict__ keys, int nkeys)
{
    __shared__ KmRelocLds lds;
    if (threadIdx.x == 0) ws->reloc_fail = 0;
    km_relocate_body(ws, keys, nkeys, &lds);
}

extern "C" int nnc_kmeans_relocate(void *ws, const int64_t *keys_sorted_dev, int32_t nkeys, void *stream)
{
    if (!ws || !keys_sorted_dev || nkeys < 0) return fail(NNC_EINVAL, "nnc_kmeans_relocate: bad argument");
    if (nkeys == 0) return NNC_OK;
    hipLaunchKernelGGL(k_relocate, dim3(1), dim3(KM_THREADS), 0, S(stream), reinterpret_cast<KmWs *>(ws),
                       reinterpret_cast<const long long *>(keys_sorted_dev), nkeys);
    LAUNCHCHK("k_relocate");
    return NNC_OK;
}

// --------------------------------------------------------------------------------------
// Windowed selection on a VALUE-SORTED vector.  Inside one cluster the squared distance to the
// centre falls monotonically towards the centre, so the farthest samples sit at the two ends of
// every cluster's stretch of the sorted vector.  The label counts of the paused iteration give
// those stretches (prefix sums over the centres in value order), so the candidates are the
// `window` samples either side of every boundary (and at the two ends of the vector):
// 2 * window * (k + 1) values instead of millions.  Candidates get their exact label / distance
// from the ordinary E-step kernel; k_reloc_verify then proves that no sample outside the windows
// can enter the top n_empty: every stretch between two windows must lie in single-candidate
// cells of one centre (so its distances are monotone either side of that centre) and the
// distances at both of its ends must be strictly below the n_empty-th best candidate's.
// Otherwise the workspace is flagged (reloc_fail), the checked relocation and the resumed
// finalize do nothing, the status reports paused = 2 and the host falls back to the full pass.
// --------------------------------------------------------------------------------------
struct KmWin { long long start; int len; int off; };

// inclusive scan (sum or running maximum) over up to 2 * KM_THREADS values, two consecutive ones per thread
template <bool MAXOP>
__device__ __forceinline__ void block_scan2(long long &a0, long long &a1, long long *wave_tot)
{
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    auto op = [](long long x, long long y) -> long long { return MAXOP ? (x > y ? x : y) : x + y; };
    a1 = op(a0, a1);
    long long s = a1;
    for (int off = 1; off < 64; off <<= 1) {
        const long long t = __shfl_up(s, off);
        if (lane >= off) s = op(s, t);
    }
    if (lane == 63) wave_tot[wv] = s;
    __syncthreads();
    long long pre = 0; // both uses scan non-negative values
    for (int w = 0; w < wv; w++) pre = op(pre, wave_tot[w]);
    __syncthreads();
    const long long prev = __shfl_up(s, 1);
    const long long before = (lane == 0) ? pre : op(pre, prev); // everything before this thread's pair
    a0 = op(before, a0);
    a1 = op(before, a1);
}

#define KM_RELOC_WMAX 8192 // a window side never grows beyond this many samples
#ifndef KM_RELOC_WMIN
#define KM_RELOC_WMIN 64  // smallest window side (16 was tried: the proof fails more often and the full pass it falls back to costs a millisecond)
#endif
#define KM_SURV_MAX 2048   // survivors of the histogram cut that are ranked exactly
#define KM_SURV_SMALL 384  // ... the cut is refined while there are more than this many

// one workgroup: the window table.  meta = {n_cand, n_windows, bad, window}
// spec_wmax > 0: the launch was enqueued behind an iteration in case it pauses for an empty cluster (the host does not look in
// between).  Then the kernel decides by itself whether there is an event it can settle -- paused for empty clusters, not due
// for the host's strict-convergence check, window (the host's rule, nnc_kmeans_reloc_window) within the scratch -- and says so
// in ws->spec_go for the launches behind it (k_cells, k_reloc_dist, k_reloc_select, the resumed finalize), which do nothing
// otherwise.
// (the decision of a chain enqueued "in case": every thread of every workgroup that asks gets the same answer)
__device__ __forceinline__ bool km_spec_decide(const KmWs *__restrict__ ws, long long n, int spec_wmax, int *W_out)
{
    const int done = ws->st.done, paused = ws->st.paused, it = ws->st.iter, same = ws->st.same_counts, ne = ws->st.n_empty;
    long long w = KM_RELOC_WMIN;
    while (w < ne) w *= 2;
    if (W_out) *W_out = (int)w;
    return !done && paused == 1 && !(it >= 1 && same) && ne >= 1 && w <= spec_wmax && 2 * w <= n;
}

__device__ __forceinline__ void km_reloc_windows_body(const float *__restrict__ xs, long long n, KmWs *__restrict__ ws, int W, long long cap,
                                                      KmWin *__restrict__ win, int *__restrict__ meta, unsigned *__restrict__ hist0, int spec_wmax)
{
    if (spec_wmax > 0) {
        const bool go = km_spec_decide(ws, n, spec_wmax, &W);
        // (the overflow list of the table k_cells rebuilds beside or behind this workgroup was emptied by the finalize step
        // that paused: km_finalize_body)
        if (threadIdx.x == 0) ws->spec_go = go ? 1 : 0;
        if (!go) return;
    }
    if (hist0) for (int i = threadIdx.x; i < 4096; i += KM_THREADS) hist0[i] = 0u; // k_reloc_dist adds to it
    __shared__ long long bnd[2 * KM_THREADS + 2];
    __shared__ long long wst[2 * KM_THREADS], wen[2 * KM_THREADS];
    __shared__ long long wave_tot[KM_THREADS / 64];
    const int tid = threadIdx.x;
    const KmTab *tab = &ws->tab[ws->cur];
    const int kt = tab->ku, k = ws->p.k;
    const float mean = ws->p.x_mean;
    const int i0 = 2 * tid, i1 = 2 * tid + 1;
    long long a0 = (i0 < kt) ? ws->partials_local[k + tab->orig[i0]] : 0;
    long long a1 = (i1 < kt) ? ws->partials_local[k + tab->orig[i1]] : 0;
    block_scan2<false>(a0, a1, wave_tot);
    if (tid == 0) bnd[0] = 0;
    bnd[i0 + 1] = a0; bnd[i1 + 1] = a1;   // bnd[j] = samples in the first j clusters (value order), j = 0 .. kt
    __syncthreads();
    const int nwin = kt + 1;
    const int bad = (bnd[kt] != n) ? 1 : 0; // the counts must be those of this very vector
    if (bad) { // (nothing below may trust the boundaries then: no window, no candidate, the selection reports the failure)
        for (int j = tid; j < nwin; j += KM_THREADS) { KmWin w; w.start = 0; w.len = 0; w.off = 0; win[j] = w; }
        if (tid == 0) { meta[0] = 0; meta[1] = nwin; meta[2] = 1; meta[3] = W; }
        return;
    }
    // Window j surrounds position bnd[j].  Each side starts at W samples and doubles until its
    // outermost sample has left the zone in which the float32 arg-min between the two
    // neighbouring centres is open (wide when two centres are close): the proof needs that.
    for (int j = tid; j < 2 * KM_THREADS; j += KM_THREADS) {
        long long st = 0, en = 0;
        if (j < nwin) {
            long long b = bnd[j];
            b = b < 0 ? 0 : (b > n ? n : b);
            long long wl = W, wr = W;
            if (j > 0 && j < kt && n > 0) {
                const double zr = tab->zr[j - 1], zl = tab->zl[j];
                bool more_r = true, more_l = true;
                while (more_r || more_l) { // both probes of a round are in flight together
                    const long long pr = b + wr - 1, pl = b - wl;
                    const float vr = (more_r && pr < n - 1) ? xs[pr] : 0.0f;
                    const float vl = (more_l && pl > 0) ? xs[pl] : 0.0f;
                    if (more_r) { if (pr >= n - 1 || (double)(vr - mean) > zr || wr >= KM_RELOC_WMAX) more_r = false; else wr *= 2; }
                    if (more_l) { if (pl <= 0 || (double)(vl - mean) < zl || wl >= KM_RELOC_WMAX) more_l = false; else wl *= 2; }
                }
            }
            st = b - wl < 0 ? 0 : b - wl;
            en = b + wr > n ? n : b + wr;
        }
        wst[j] = st; wen[j] = en;
    }
    __syncthreads();
    long long e0 = wen[i0], e1 = wen[i1];
    block_scan2<true>(e0, e1, wave_tot); // running maximum of the window ends
    if (tid == 0) bnd[0] = 0;
    bnd[i0 + 1] = e0; bnd[i1 + 1] = e1;  // bnd[j + 1] = end of windows 0 .. j   (all reads of bnd[] as boundaries are behind a barrier)
    __syncthreads();
    long long l0 = 0, l1 = 0, s0 = 0, s1 = 0;
    if (i0 < nwin) { s0 = wst[i0] > bnd[i0] ? wst[i0] : bnd[i0]; l0 = bnd[i0 + 1] - s0; if (l0 <= 0) { l0 = 0; s0 = bnd[i0 + 1]; } }
    if (i1 < nwin) { s1 = wst[i1] > bnd[i1] ? wst[i1] : bnd[i1]; l1 = bnd[i1 + 1] - s1; if (l1 <= 0) { l1 = 0; s1 = bnd[i1 + 1]; } }
    const long long len0 = l0, len1 = l1;
    block_scan2<false>(l0, l1, wave_tot);
    if (i0 < nwin) { KmWin w; w.start = s0; w.len = (int)len0; w.off = l0 <= cap ? (int)(l0 - len0) : 0; win[i0] = w; }
    if (i1 < nwin) { KmWin w; w.start = s1; w.len = (int)len1; w.off = l1 <= cap ? (int)(l1 - len1) : 0; win[i1] = w; }
    if (tid == (nwin - 1) / 2) {
        const long long total = ((nwin - 1) & 1) ? l1 : l0;
        const int over = total > cap ? 1 : 0;
        meta[0] = over ? 0 : (int)total; meta[1] = nwin; meta[2] = bad | over; meta[3] = W;
    }
}

__global__ __launch_bounds__(KM_THREADS) void k_reloc_windows(const float *__restrict__ xs, long long n, KmWs *__restrict__ ws,
                                                              int W, long long cap, KmWin *__restrict__ win, int *__restrict__ meta,
                                                              unsigned *__restrict__ hist0 = nullptr, int spec_wmax = 0)
{
    if (spec_wmax > 0 && threadIdx.x == 0 && km_spec_decide(ws, n, spec_wmax, nullptr)) ws->tab[ws->cur].n_ovf = 0; // (launched on its own: k_cells follows)
    km_reloc_windows_body(xs, n, ws, W, cap, win, meta, hist0, spec_wmax);
}

// Head of the relocation chain enqueued "in case": the window table (last workgroup) and the cell table of the current centres
// (the others) side by side -- neither needs the other, both read the zones.
__global__ __launch_bounds__(KM_THREADS) void k_reloc_head(const float *__restrict__ xs, long long n, KmWs *__restrict__ ws, long long cap,
                                                           KmWin *__restrict__ win, int *__restrict__ meta, unsigned *__restrict__ hist0, int spec_wmax)
{
    if (blockIdx.x == gridDim.x - 1) {
        km_reloc_windows_body(xs, n, ws, 0, cap, win, meta, hist0, spec_wmax);
    } else {
        if (!km_spec_decide(ws, n, spec_wmax, nullptr)) return;
        km_cells_body(ws, nullptr, nullptr, 0ull, 1, 0, 0, (int)blockIdx.x);
    }
}

// the candidates themselves (any grid): one wave per window
__global__ __launch_bounds__(256) void k_reloc_fill(const float *__restrict__ xs, const KmWin *__restrict__ win,
                                                    const int *__restrict__ meta, float *__restrict__ cand_x, long long cap)
{
    const int lane = threadIdx.x & 63;
    const int wave = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6), nwaves = (int)((gridDim.x * blockDim.x) >> 6);
    const int nwin = meta[1];
    const long long total = meta[0];
    if (!meta[2])
        for (int j = wave; j < nwin; j += nwaves) {
            const KmWin w = win[j];
            for (int i = lane; i < w.len; i += 64) cand_x[w.off + i] = xs[w.start + i];
        }
    // up to three scalars after the last candidate are read as part of a float4 by the distance pass
    if (blockIdx.x == 0 && threadIdx.x < 4 && total + threadIdx.x < cap) cand_x[total + threadIdx.x] = 0.0f;
}

// Candidates AND their squared distances in one launch (what k_reloc_fill + k_assign<labels> on the candidate array
// give, without the second launch): every workgroup stages the search tables in LDS as the streaming kernel does, then
// its waves take windows in turn; the label is the exact float32 arg-min (km_resolve), the distance (x~ - c)^2.
__global__ __launch_bounds__(KM_THREADS) void k_reloc_dist(const float *__restrict__ xs, const KmWin *__restrict__ win,
                                                           const int *__restrict__ meta, float *__restrict__ cand_x,
                                                           float *__restrict__ cand_d, long long cap, const KmWs *__restrict__ ws,
                                                           unsigned *__restrict__ hist0, int spec = 0)
{
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ unsigned h0_s[4096]; // first level of the selection's histogram (top 12 bits of the distance), for free here
    if (spec && !ws->spec_go) return;
    for (int i = threadIdx.x; i < 4096; i += KM_THREADS) h0_s[i] = 0u;
    if (meta[2]) {
        if (blockIdx.x == 0 && threadIdx.x < 4 && threadIdx.x < cap) { cand_x[threadIdx.x] = 0.0f; cand_d[threadIdx.x] = 0.0f; }
        return;
    }
    const int k = ws->p.k, glog2 = ws->glog2;
    const int G = 1 << glog2, kp = (k + 7) & ~7;
    const KmTab *__restrict__ tab = &ws->tab[ws->cur];
    const int kt = tab->ku;
    uint16_t *cell_s = reinterpret_cast<uint16_t *>(smem);
    float4 *pair_s = reinterpret_cast<float4 *>(smem + ((size_t)2 << glog2));
    float *cval_s = reinterpret_cast<float *>(smem + ((size_t)2 << glog2) + (size_t)kp * 16);
    uint16_t *orig_s = reinterpret_cast<uint16_t *>(smem + ((size_t)2 << glog2) + (size_t)kp * 20);
    const size_t off = (((size_t)2 << glog2) + (size_t)kp * 22 + 15) & ~(size_t)15;
    unsigned *ovf_s = reinterpret_cast<unsigned *>(smem + off);
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(tab->cell);
        uint4 *dst = reinterpret_cast<uint4 *>(cell_s);
        for (int i = threadIdx.x; i < (G >> 3); i += KM_THREADS) dst[i] = src[i];
        for (int i = threadIdx.x; i < kt; i += KM_THREADS) {
            const float2 a = tab->cand[i];
            const float2 b = (i + 1 < kt) ? tab->cand[i + 1] : a;
            pair_s[i] = make_float4(a.x, a.y, b.x, b.y);
            cval_s[i] = a.x;
            orig_s[i] = tab->orig[i];
        }
        const int novf = min(tab->n_ovf, KM_OVF_MAX);
        for (int i = threadIdx.x; i < novf; i += KM_THREADS) ovf_s[i] = tab->ovf[i];
    }
    __syncthreads();
    KmCtx c;
    c.cell_s = cell_s; c.pair_s = pair_s; c.cval_s = cval_s; c.orig_s = orig_s; c.ovf_s = ovf_s; c.sum_s = nullptr; c.cnt_s = nullptr;
    c.mean = ws->p.x_mean; c.lo = ws->p.lo; c.inv = ws->inv;
    c.Sft = ws->p.fix_shift; c.gmax = G - 1; c.k = kt; c.rlog2 = 0; c.rep = 0;
    // candidate i lives in the window whose offset range holds i (a tail window may be thousands of values long, so
    // the work is split by candidate, not by window): window offsets in LDS, one binary search per candidate
    const int nwin = meta[1];
    const long long total = meta[0];
    int *woff_s = reinterpret_cast<int *>(smem + off + KM_OVF_MAX * 4); // behind the side list: (NNC_KMAX + 2) ints
    for (int j = threadIdx.x; j < nwin; j += KM_THREADS) woff_s[j] = win[j].off;
    __syncthreads();
    for (long long i = (long long)blockIdx.x * KM_THREADS + threadIdx.x; i < total; i += (long long)gridDim.x * KM_THREADS) {
        int l = 0, h = nwin - 1; // last window with off <= i (empty windows share an offset with their successor: any of them will do
        while (l < h) { const int m = (l + h + 1) >> 1; if (woff_s[m] <= (int)i) l = m; else h = m - 1; }
        // ... as long as it is not empty: step back over empty ones to the window that really holds i)
        KmWin w = win[l];
        while (w.len == 0 || (int)i - w.off >= w.len) { l--; w = win[l]; }
        const float xv[1] = {xs[w.start + ((int)i - w.off)]};
        float xc[1];
        int p[1];
        km_resolve<1>(c, xv, xc, p);
        const float dd = xc[0] - cval_s[p[0]];
        const float dv = dd * dd;
        cand_x[i] = xv[0];
        cand_d[i] = dv;
        atomicAdd(&h0_s[(__float_as_uint(dv) >> 19) & 4095u], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 4096; i += KM_THREADS)
        if (h0_s[i]) atomicAdd(&hist0[i], h0_s[i]);
    // up to three values after the last candidate are read as part of a 16-byte load by the selection
    if (blockIdx.x == 0 && threadIdx.x < 4 && total + threadIdx.x < cap) { cand_x[total + threadIdx.x] = 0.0f; cand_d[total + threadIdx.x] = 0.0f; }
}

// One workgroup: the n_empty largest keys among the candidates (histogram cut on the distance
// bits, refined while crowded, exact ranking of the survivors), the proof that nothing outside
// the windows can beat them, and -- if it holds -- the relocation itself.
__device__ __forceinline__ void km_reloc_select_body(KmWs *__restrict__ ws, const float *__restrict__ cand_x,
                                                     const float *__restrict__ cand_d, const KmWin *__restrict__ win,
                                                     const int *__restrict__ meta, int n_empty, long long *__restrict__ keys_out,
                                                     int do_relocate, const unsigned *__restrict__ hist0, int spec)
{
    if (spec) { // enqueued in case of an event: nothing to do without one; the number of empty clusters is the device's
        if (!ws->spec_go) return;
        n_empty = ws->st.n_empty;
    }
    __shared__ __align__(8) unsigned hist[4096];
    __shared__ unsigned long long surv[KM_SURV_MAX];
    __shared__ int wave_i[KM_THREADS / 64];
    __shared__ int s_cut, s_above, s_ge, s_nsurv, s_bad;
    __shared__ KmRelocLds rl;
    __shared__ double zl_s[NNC_KMAX];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    unsigned long long *strc = NNC_KM_TRACE_PTR; // diagnostics: phase stamps behind the workgroup records
#define RSTAMP(i) do { if (strc && tid == 0) strc[7000 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
    RSTAMP(0);
    const int n_cand = meta[0];
    // (the relocation's own inputs -- centres, list of empty clusters -- do not depend on the selection: fetched now)
    const int n_empty_ws = do_relocate ? km_relocate_prepare(ws, &rl) : 0;
    {
        int bad0 = meta[2] ? 1 : 0;
        if (n_empty < 1 || n_cand < n_empty) bad0 |= 2;
        if (bad0) { if (tid == 0) { ws->reloc_fail = bad0; if (spec) ws->st.n_unproven += 1; } return; }
    }
    // Coalesced 16-byte reads; neighbours in position have similar distances, so most of the time a
    // whole wave lands in one bin: one LDS atomic per wave instead of 256 (LDS atomics retire about
    // one lane per clock).
    const int nvec = n_cand >> 2; // cand_d is the start of an allocation: 16-byte aligned
    const uint4 *d4 = reinterpret_cast<const uint4 *>(cand_d);
    unsigned prefix = 0, thr = 0;
    int pshift = -1, above = 0, total_ge = n_cand;
    const int shifts[3] = {19, 7, 0}, widths[3] = {12, 12, 7};
    for (int lvl = 0; lvl < 3; lvl++) {
        const int shift = shifts[lvl], width = widths[lvl];
        const unsigned mask = (1u << width) - 1u;
        for (int i = tid; i < 4096; i += KM_THREADS) hist[i] = (lvl == 0 && hist0) ? hist0[i] : 0u; // the first level may come with the candidates
        if (tid == 0) { s_cut = 0; s_above = above; s_ge = total_ge; }
        __syncthreads();
        const unsigned NOBIN = 0xFFFFu;
        auto bin_of = [&](unsigned u) -> unsigned { return (pshift < 0 || (u >> pshift) == prefix) ? ((u >> shift) & mask) : NOBIN; };
        for (int v0 = 0; v0 < ((lvl == 0 && hist0) ? 0 : nvec); v0 += 8 * KM_THREADS) { // wave-uniform trip count; eight loads in flight per thread
            uint4 qv[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int v = v0 + u * KM_THREADS + tid;
                qv[u] = v < nvec ? d4[v] : make_uint4(0u, 0u, 0u, 0u);
            }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int v = v0 + u * KM_THREADS + tid;
                const bool have = v < nvec;
                const uint4 q = qv[u];
                const unsigned b0 = have ? bin_of(q.x) : NOBIN, b1 = have ? bin_of(q.y) : NOBIN;
                const unsigned b2 = have ? bin_of(q.z) : NOBIN, b3 = have ? bin_of(q.w) : NOBIN;
                const unsigned long long act = __ballot(have);
                if (!act) continue;
                const unsigned first = (unsigned)__shfl((int)b0, __ffsll((long long)act) - 1);
                const bool same = !have || (b0 == first && b1 == first && b2 == first && b3 == first);
                if (__all(same)) {
                    if (first != NOBIN && lane == 0) atomicAdd(&hist[first], 4u * (unsigned)__popcll(act));
                } else if (have) {
                    unsigned rb = b0, rc = 1;
                    if (b1 == rb) rc++; else { if (rb != NOBIN) atomicAdd(&hist[rb], rc); rb = b1; rc = 1; }
                    if (b2 == rb) rc++; else { if (rb != NOBIN) atomicAdd(&hist[rb], rc); rb = b2; rc = 1; }
                    if (b3 == rb) rc++; else { if (rb != NOBIN) atomicAdd(&hist[rb], rc); rb = b3; rc = 1; }
                    if (rb != NOBIN) atomicAdd(&hist[rb], rc);
                }
            }
        }
        for (int i = ((lvl == 0 && hist0) ? n_cand : (nvec << 2)) + tid; i < n_cand; i += KM_THREADS) {
            const unsigned bn = bin_of(__float_as_uint(cand_d[i]));
            if (bn != NOBIN) atomicAdd(&hist[bn], 1u);
        }
        __syncthreads();
        // thread t owns bins 4t .. 4t+3; count of everything in higher bins via a block prefix sum
        const unsigned h0 = hist[4 * tid], h1 = hist[4 * tid + 1], h2 = hist[4 * tid + 2], h3 = hist[4 * tid + 3];
        int own = (int)(h0 + h1 + h2 + h3), sc = own;
        for (int off = 1; off < 64; off <<= 1) { const int t = __shfl_up(sc, off); if (lane >= off) sc += t; }
        if (lane == 63) wave_i[wv] = sc;
        __syncthreads();
        int pre = 0, tot = 0;
        for (int w = 0; w < KM_THREADS / 64; w++) { const int v = wave_i[w]; if (w < wv) pre += v; tot += v; }
        int run = above + (tot - (pre + sc)); // samples above this thread's bins (and above the prefix)
        const unsigned hb[4] = {h0, h1, h2, h3};
        for (int q = 3; q >= 0; q--) {
            const int before = run;
            run += (int)hb[q];
            if (before < n_empty && run >= n_empty) { s_cut = 4 * tid + q; s_above = before; s_ge = run; } // exactly one thread, one bin
        }
        __syncthreads();
        const int bcut = s_cut;
        prefix = (pshift < 0) ? (unsigned)bcut : ((prefix << width) | (unsigned)bcut);
        pshift = shift;
        above = s_above;
        total_ge = s_ge;
        thr = prefix << shift;
        __syncthreads();
        if (total_ge <= KM_SURV_SMALL) break;
    }
    RSTAMP(1);
    if (strc && tid == 0) { strc[7010] = (unsigned long long)n_cand; strc[7011] = (unsigned long long)total_ge; }
    int bad = 0;
    if (total_ge > KM_SURV_MAX) bad |= 128; // a crowd of exactly equal distances at the cut
    if (tid == 0) { s_nsurv = 0; s_bad = 0; }
    for (int i = tid; i < KM_SURV_MAX; i += KM_THREADS) surv[i] = 0ull; // padding sorts last
    __syncthreads();
    if (!bad) {
        auto take = [&](unsigned u, int i) {
            if (u >= thr) {
                const int slot = atomicAdd(&s_nsurv, 1);
                if (slot < KM_SURV_MAX) surv[slot] = ((unsigned long long)u << 32) | (unsigned long long)f32_ordered_bits(cand_x[i]);
            }
        };
        for (int v0 = 0; v0 < nvec; v0 += 8 * KM_THREADS) {
            uint4 qv[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int v = v0 + u * KM_THREADS + tid;
                qv[u] = v < nvec ? d4[v] : make_uint4(0u, 0u, 0u, 0u);
            }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int v = v0 + u * KM_THREADS + tid;
                if (v < nvec) { take(qv[u].x, 4 * v); take(qv[u].y, 4 * v + 1); take(qv[u].z, 4 * v + 2); take(qv[u].w, 4 * v + 3); }
            }
        }
        for (int i = (nvec << 2) + tid; i < n_cand; i += KM_THREADS) take(__float_as_uint(cand_d[i]), i);
    }
    __syncthreads();
    RSTAMP(2);
    const int m = min(s_nsurv, KM_SURV_MAX);
    if (!bad && m < n_empty) bad |= 2;
    if (!bad) {
        if (m <= KM_SURV_SMALL) {
            // few survivors: rank by counting; the m * m comparisons are spread over all threads (P of them share
            // a survivor, each takes a slice of the others; 64-bit compares are slow, so the VALU time matters)
            int *rank_s = reinterpret_cast<int *>(hist); // the histogram is done with
            const int P = max(1, KM_THREADS / max(m, 1));
            const int slice = (m + P - 1) / P;
            for (int i = tid; i < m; i += KM_THREADS) rank_s[i] = 0;
            __syncthreads();
            unsigned long long mine = 0ull;
            const int i = tid % max(m, 1), part = tid / max(m, 1);
            if (part < P && m > 0) {
                mine = surv[i];
                int r = 0;
                const int j1 = min(m, (part + 1) * slice);
                for (int j = part * slice; j < j1; j++) { const unsigned long long kj = surv[j]; r += (kj > mine) || (kj == mine && j < i); }
                if (r) atomicAdd(&rank_s[i], r);
            }
            __syncthreads();
            if (part == 0 && m > 0) surv[rank_s[i]] = mine; // every rank 0 .. m-1 is taken exactly once
        } else {
            // bitonic sort, descending, of the survivors (padded with zeros to a power of two)
            int M = 2;
            while (M < m) M <<= 1;
            for (int size = 2; size <= M; size <<= 1)
                for (int stride = size >> 1; stride > 0; stride >>= 1) {
                    __syncthreads();
                    for (int t = tid; t < (M >> 1); t += KM_THREADS) {
                        const int lo_i = 2 * t - (t & (stride - 1)), hi_i = lo_i + stride;
                        const bool up = (lo_i & size) == 0;
                        const unsigned long long ka = surv[lo_i], kb = surv[hi_i];
                        if ((ka < kb) == up) { surv[lo_i] = kb; surv[hi_i] = ka; }
                    }
                }
        }
        __syncthreads();
        // (one key more than needed, 0 if there is none: the runner-up shows a tie at the cut, see km_relocate_apply)
        for (int r = tid; r <= n_empty; r += KM_THREADS) keys_out[r] = (r < m) ? (long long)surv[r] : 0ll;
    }
    __syncthreads();
    RSTAMP(3);
    if (!bad) {
        const unsigned dT = (unsigned)(surv[n_empty - 1] >> 32);
        const KmTab *tab = &ws->tab[ws->cur];
        const float mean = ws->p.x_mean;
        const int ku = tab->ku, nwin = meta[1];
        // zone ends into LDS (the right ends over the histogram, which is free now)
        double *zr_s = reinterpret_cast<double *>(hist); // 4096 * 4 B = 2048 doubles
        for (int q = tid; q < ku; q += KM_THREADS) { zr_s[q] = tab->zr[q]; zl_s[q] = tab->zl[q]; }
        __syncthreads();
        // running maximum of the upper ends (from below) and running minimum of the lower ends (from above):
        // Hillis-Steele over at most NNC_KMAX values, two per thread -- unless the ends are in order already (no zone reaches
        // across a neighbour's: the usual case away from a mass relocation), which one vote tells
        int in_order = 1;
        for (int q = tid; q + 1 < ku; q += KM_THREADS) in_order &= (zr_s[q + 1] >= zr_s[q]) && (zl_s[q] <= zl_s[q + 1]);
        const int mono = __syncthreads_and(in_order);
        for (int off = 1; off < ku && !mono; off <<= 1) {
            double a[2], b[2];
            for (int r = 0; r < 2; r++) {
                const int q = tid + r * KM_THREADS;
                a[r] = (q < ku && q >= off) ? fmax(zr_s[q], zr_s[q - off]) : (q < ku ? zr_s[q] : 0.0);
                b[r] = (q < ku && q + off < ku) ? fmin(zl_s[q], zl_s[q + off]) : (q < ku ? zl_s[q] : 0.0);
            }
            __syncthreads();
            for (int r = 0; r < 2; r++) {
                const int q = tid + r * KM_THREADS;
                if (q < ku) { zr_s[q] = a[r]; zl_s[q] = b[r]; }
            }
            __syncthreads();
        }
        for (int j = tid; j + 1 < nwin; j += KM_THREADS) {
            const KmWin a = win[j], b = win[j + 1];
            if (b.start > a.start + a.len) { // samples between the two windows that are no candidates
                // the candidates either side of the stretch (candidates are stored in position order, so an
                // empty window -- an empty cluster's boundary coincides with its neighbour's -- is skipped over)
                const int iu = a.off + a.len - 1, iw = b.off;
                if (iu < 0 || iw >= n_cand) { bad |= 4; continue; }
                const float u = cand_x[iu], w = cand_x[iw];
                const unsigned du = __float_as_uint(cand_d[iu]), dw = __float_as_uint(cand_d[iw]);
                // every value in [u, w] must have centre j (value order) as its only candidate:
                // above the zones of all smaller centres, below the zones of all larger ones
                const double uc = (double)(u - mean), wc = (double)(w - mean);
                if (j > 0 && j - 1 < ku && !(uc > zr_s[j - 1])) bad |= 8;
                if (j + 1 < ku && !(wc < zl_s[j + 1])) bad |= 16;
                if (!(du < dT)) bad |= 32;
                if (!(dw < dT)) bad |= 64;
            }
        }
    }
    // reason bits (diagnostics): 1 counts/capacity, 2 too few candidates, 4 window table, 8 / 16 lower / upper end of a
    // stretch inside another centre's zone, 32 / 64 lower / upper end not strictly below the cut, 128 tie crowd
    if (bad) atomicOr(&s_bad, bad);
    __syncthreads();
    RSTAMP(4);
    const int any_bad = s_bad;
    if (tid == 0) { ws->reloc_fail = any_bad; if (spec && any_bad) ws->st.n_unproven += 1; }
    if (any_bad || !do_relocate) return; // (sharded vector: the ranks first exchange their keys and their verdicts)
    __threadfence_block();
    km_relocate_apply(ws, keys_out, min(m, n_empty + 1), n_empty_ws, &rl);
    if (spec && tid == 0) ws->st.n_relocated += 1; // (the host did not see this event: it counts them from here)
    RSTAMP(5);
#undef RSTAMP
}

__global__ __launch_bounds__(KM_THREADS) void k_reloc_select(KmWs *__restrict__ ws, const float *__restrict__ cand_x,
                                                             const float *__restrict__ cand_d, const KmWin *__restrict__ win,
                                                             const int *__restrict__ meta, int n_empty, long long *__restrict__ keys_out,
                                                             int do_relocate, const unsigned *__restrict__ hist0, int spec)
{
    km_reloc_select_body(ws, cand_x, cand_d, win, meta, n_empty, keys_out, do_relocate, hist0, spec);
}

// The last two launches of the relocation chain enqueued "in case" as ONE (round 4): the selection, then -- the same workgroup, behind a
// barrier -- the resumed finalize step that used to be a launch of its own (a boundary and the re-reading of what the selection has
// just written: 5 us an event, and a launch less for every chain that finds nothing to do).  Sixteen waves either way, so only where the
// finalize step runs with sixteen (more than 256 centres).
__global__ __launch_bounds__(KM_THREADS) void k_reloc_select_finalize(KmWs *__restrict__ ws, const float *__restrict__ cand_x,
                                                                      const float *__restrict__ cand_d, const KmWin *__restrict__ win,
                                                                      const int *__restrict__ meta, long long *__restrict__ keys_out,
                                                                      const unsigned *__restrict__ hist0, nnc_kmeans_status *host_st,
                                                                      unsigned long long *host_ticket, unsigned long long ticket, int lazy)
{
    km_reloc_select_body(ws, cand_x, cand_d, win, meta, 0, keys_out, 1, hist0, 1);
    __syncthreads();                                    // what the selection wrote (sums, counts, status) is out ...
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); // ... and is read past whatever this compute unit's L1 still holds of it
    km_finalize_kernel<KM_THREADS, false, false>(ws, FIN_FROM_PARTIALS, 2, host_st, host_ticket, ticket, lazy, nullptr, 0);
}

static std::atomic<int> g_reloc_dist_attr[NNC_MAX_DEVICES];
// windows, candidates and their distances (two launches)
static int km_reloc_windows_dist(const float *x_sorted, void *ws, const nnc_kmeans_params *p, int32_t window, float *cand_x,
                                 float *cand_d, int64_t cap, void *win_dev, int32_t *meta_dev, unsigned *hist0, void *stream)
{
    int glog2, rlog2;
    km_defaults(p, &glog2, &rlog2);
    const size_t lds = km_lds_bytes(p->k, glog2, rlog2, false) + (NNC_KMAX + 2) * sizeof(int);
    if (lds > 128 * 1024) return fail(NNC_EINVAL, "relocation: search tables too large for the candidate kernel");
    if (!g_reloc_dist_attr[current_device()].load(std::memory_order_acquire)) {
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_reloc_dist), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024)); // + 16 KB static
        g_reloc_dist_attr[current_device()].store(1, std::memory_order_release);
    }
    {
        int rc = km_ensure_cells(reinterpret_cast<KmWs *>(ws), p, 0, stream); // k_reloc_dist looks the candidates up in the cell table
        if (rc) return rc;
    }
    NNC_LAUNCH_PROF(NNC_PROF_RELOC, k_reloc_windows, dim3(1), dim3(KM_THREADS), 0, S(stream), x_sorted, (long long)p->n,
                       reinterpret_cast<KmWs *>(ws), (int)window, (long long)cap,
                       reinterpret_cast<KmWin *>(win_dev), reinterpret_cast<int *>(meta_dev), hist0, 0);
    LAUNCHCHK("k_reloc_windows");
    // a candidate costs its thread two dependent reads (window record, sample): one or two per thread, not a queue of them
    // (the windows double where centres are close, so there are about twice 2 * window * (k + 1) of them)
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(128, (2 * (int64_t)window * (p->k + 1) + KM_THREADS - 1) / KM_THREADS));
    NNC_LAUNCH_PROF(NNC_PROF_RELOC, k_reloc_dist, dim3(grid), dim3(KM_THREADS), lds, S(stream), x_sorted, reinterpret_cast<const KmWin *>(win_dev),
                       reinterpret_cast<const int *>(meta_dev), cand_x, cand_d, (long long)cap, reinterpret_cast<const KmWs *>(ws), hist0, 0);
    LAUNCHCHK("k_reloc_dist");
    return NNC_OK;
}

extern "C" int nnc_kmeans_reloc_candidates(const float *x_sorted, void *ws, const nnc_kmeans_params *p, int32_t window,
                                           float *cand_x_dev, int64_t cap, void *win_dev, int32_t *meta_dev, void *stream)
{
    int rc = km_check(ws, p, "nnc_kmeans_reloc_candidates");
    if (rc) return rc;
    if (!x_sorted || !cand_x_dev || !win_dev || !meta_dev || window < 1) return fail(NNC_EINVAL, "nnc_kmeans_reloc_candidates: bad argument");
    if (cap < 2 * (int64_t)window * (p->k + 1) || cap > 0x7FFFFFFF) return fail(NNC_ENOSPACE, "nnc_kmeans_reloc_candidates: cap must be in [2 * window * (k + 1), 2^31)");
    NNC_LAUNCH_PROF(NNC_PROF_RELOC, k_reloc_windows, dim3(1), dim3(KM_THREADS), 0, S(stream), x_sorted, (long long)p->n,
                       reinterpret_cast<KmWs *>(ws), (int)window, (long long)cap,
                       reinterpret_cast<KmWin *>(win_dev), reinterpret_cast<int *>(meta_dev), (unsigned *)nullptr, 0);
    LAUNCHCHK("k_reloc_windows");
    const int grid = (int)std::min<int64_t>(256, (cap + 16383) / 16384 + p->k / 4 + 1);
    hipLaunchKernelGGL(k_reloc_fill, dim3(grid), dim3(256), 0, S(stream), x_sorted, reinterpret_cast<const KmWin *>(win_dev),
                       reinterpret_cast<const int *>(meta_dev), cand_x_dev, (long long)cap);
    LAUNCHCHK("k_reloc_fill");
    return NNC_OK;
}

#ifdef NNC_DIAG
extern "C" int nnc_debug_reloc_fail(void *ws, int32_t *host_out)
{
    if (!ws || !host_out) return fail(NNC_EINVAL, "nnc_debug_reloc_fail: null pointer");
    HIPCHK(hipMemcpy(host_out, &reinterpret_cast<KmWs *>(ws)->reloc_fail, sizeof(int32_t), hipMemcpyDeviceToHost));
    return NNC_OK;
}
#endif

extern "C" int nnc_kmeans_relocate_checked(void *ws, const float *cand_x_dev, const float *cand_d_dev, const void *win_dev,
                                           const int32_t *meta_dev, int32_t n_empty, int64_t *keys_out_dev, void *stream)
{
    if (!ws || !cand_x_dev || !cand_d_dev || !win_dev || !meta_dev || !keys_out_dev || n_empty < 1 || n_empty > NNC_KMAX)
        return fail(NNC_EINVAL, "nnc_kmeans_relocate_checked: bad argument");
    NNC_LAUNCH_PROF(NNC_PROF_RELOC, k_reloc_select, dim3(1), dim3(KM_THREADS), 0, S(stream), reinterpret_cast<KmWs *>(ws), cand_x_dev, cand_d_dev,
                       reinterpret_cast<const KmWin *>(win_dev), reinterpret_cast<const int *>(meta_dev), (int)n_empty,
                       reinterpret_cast<long long *>(keys_out_dev), 1, (const unsigned *)nullptr, 0);
    LAUNCHCHK("k_reloc_select");
    return NNC_OK;
}

// The whole windowed relocation as one call: windows -> candidates -> exact distances -> selection
// + proof + relocation -> resumed finalize.  No host read; the outcome shows in the next status.

extern "C" int32_t nnc_kmeans_reloc_window(int64_t n, int32_t n_empty)
{
    int64_t w = KM_RELOC_WMIN;
    while (w < n_empty) w *= 2;
    if (n_empty < 1 || w > 1024 || 2 * w > n) return 0; // not applicable: use the full pass
    return (int32_t)w;
}

static int64_t reloc_cap(int32_t k, int32_t window) { return 8 * (int64_t)window * (k + 1); } // sides double where two centres are close

extern "C" size_t nnc_kmeans_reloc_scratch_bytes(int32_t k, int32_t window)
{
    if (k < 1 || window < 1) return 0;
    const size_t cap = (size_t)reloc_cap(k, window);
    return 2 * reloc_align(cap * 4) + reloc_align(16 * (size_t)(k + 2)) + reloc_align(16) + reloc_align(8 * (size_t)NNC_KMAX) + reloc_align(4096 * 4);
}

extern "C" int nnc_kmeans_relocate_windowed(const float *x_sorted, void *ws, const nnc_kmeans_params *p, int32_t n_empty,
                                            void *scratch_dev, size_t scratch_bytes, void *stream)
{
    int rc = km_check(ws, p, "nnc_kmeans_relocate_windowed");
    if (rc) return rc;
    const int32_t window = nnc_kmeans_reloc_window(p->n, n_empty);
    if (window == 0) return fail(NNC_EINVAL, "nnc_kmeans_relocate_windowed: not applicable (nnc_kmeans_reloc_window() == 0)");
    if (!x_sorted || !scratch_dev || (reinterpret_cast<uintptr_t>(scratch_dev) & 255) != 0)
        return fail(NNC_EINVAL, "nnc_kmeans_relocate_windowed: null or unaligned (256 B) pointer");
    if (scratch_bytes < nnc_kmeans_reloc_scratch_bytes(p->k, window)) return fail(NNC_ENOSPACE, "nnc_kmeans_relocate_windowed: scratch too small");
    const int64_t cap = reloc_cap(p->k, window);
    unsigned char *b = reinterpret_cast<unsigned char *>(scratch_dev);
    float *cand_x = reinterpret_cast<float *>(b); b += reloc_align((size_t)cap * 4);
    float *cand_d = reinterpret_cast<float *>(b); b += reloc_align((size_t)cap * 4);
    void *win = b; b += reloc_align(16 * (size_t)(p->k + 2));
    int32_t *meta = reinterpret_cast<int32_t *>(b); b += reloc_align(16);
    int64_t *keys = reinterpret_cast<int64_t *>(b); b += reloc_align(8 * (size_t)NNC_KMAX);
    unsigned *hist0 = reinterpret_cast<unsigned *>(b);
    if ((rc = km_reloc_windows_dist(x_sorted, ws, p, window, cand_x, cand_d, cap, win, meta, hist0, stream))) return rc;
    NNC_LAUNCH_PROF(NNC_PROF_RELOC, k_reloc_select, dim3(1), dim3(KM_THREADS), 0, S(stream), reinterpret_cast<KmWs *>(ws), cand_x, cand_d,
                       reinterpret_cast<const KmWin *>(win), reinterpret_cast<const int *>(meta), (int)n_empty,
                       reinterpret_cast<long long *>(keys), 1, (const unsigned *)hist0, 0);
    LAUNCHCHK("k_reloc_select");
    return km_launch_finalize(reinterpret_cast<KmWs *>(ws), p, FIN_FROM_PARTIALS, 1, stream);
}

// The same chain enqueued behind an iteration IN CASE it pauses for an empty cluster: the host is not looking, the kernels
// decide (k_reloc_windows) and do nothing when there is no event they can settle.  The scratch is laid out for windows of
// `wmax`; the look-in of a batch rides on the last launch (the resumed finalize).
#define KM_SPEC_WMAX 256
static int km_launch_spec_reloc(const float *x_sorted, KmWs *w, const nnc_kmeans_params *p, void *scratch_dev, void *stream,
                                void *host_mapped, uint64_t ticket)
{
    const int32_t wmax = KM_SPEC_WMAX;
    const int64_t cap = reloc_cap(p->k, wmax);
    unsigned char *b = reinterpret_cast<unsigned char *>(scratch_dev);
    float *cand_x = reinterpret_cast<float *>(b); b += reloc_align((size_t)cap * 4);
    float *cand_d = reinterpret_cast<float *>(b); b += reloc_align((size_t)cap * 4);
    KmWin *win = reinterpret_cast<KmWin *>(b); b += reloc_align(16 * (size_t)(p->k + 2));
    int *meta = reinterpret_cast<int *>(b); b += reloc_align(16);
    long long *keys = reinterpret_cast<long long *>(b); b += reloc_align(8 * (size_t)NNC_KMAX);
    unsigned *hist0 = reinterpret_cast<unsigned *>(b);
    int glog2, rlog2;
    km_defaults(p, &glog2, &rlog2);
    const size_t lds = km_lds_bytes(p->k, glog2, rlog2, false) + (NNC_KMAX + 2) * sizeof(int);
    if (lds > 128 * 1024) return fail(NNC_EINVAL, "relocation: search tables too large for the candidate kernel");
    if (!g_reloc_dist_attr[current_device()].load(std::memory_order_acquire)) {
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_reloc_dist), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
        g_reloc_dist_attr[current_device()].store(1, std::memory_order_release);
    }
    NNC_LAUNCH_PROF(NNC_PROF_RELOC, k_reloc_head, dim3(KM_GMAX / KM_THREADS + 1), dim3(KM_THREADS), 0, S(stream), x_sorted, (long long)p->n, w, (long long)cap, win, meta,
                       hist0, (int)wmax);
    LAUNCHCHK("k_reloc_head");
    // (sized for windows of 64, the common case: the kernel strides over whatever there is)
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(128, (2 * (int64_t)64 * (p->k + 1) + KM_THREADS - 1) / KM_THREADS));
    NNC_LAUNCH_PROF(NNC_PROF_RELOC, k_reloc_dist, dim3(grid), dim3(KM_THREADS), lds, S(stream), x_sorted, (const KmWin *)win, (const int *)meta, cand_x, cand_d, (long long)cap,
                       reinterpret_cast<const KmWs *>(w), hist0, 1);
    LAUNCHCHK("k_reloc_dist");
#ifndef KM_CHAIN_FIVE
    if (p->k > 256 && !km_fused(p) && p->prefix_dev) { // (where km_launch_finalize would take sixteen waves and the lazy form: see k_reloc_select_finalize)
        unsigned char *hb = reinterpret_cast<unsigned char *>(host_mapped);
        NNC_LAUNCH_PROF(NNC_PROF_RELOC, k_reloc_select_finalize, dim3(1), dim3(KM_THREADS), 0, S(stream), w, (const float *)cand_x, (const float *)cand_d, (const KmWin *)win,
                        (const int *)meta, keys, (const unsigned *)hist0, reinterpret_cast<nnc_kmeans_status *>(hb),
                        reinterpret_cast<unsigned long long *>(hb ? hb + sizeof(nnc_kmeans_status) : nullptr), (unsigned long long)ticket, 1 | (p->k << 8));
        LAUNCHCHK("k_reloc_select_finalize");
        return NNC_OK;
    }
#endif
    NNC_LAUNCH_PROF(NNC_PROF_RELOC, k_reloc_select, dim3(1), dim3(KM_THREADS), 0, S(stream), w, (const float *)cand_x, (const float *)cand_d, (const KmWin *)win, (const int *)meta, 0, keys, 1, (const unsigned *)hist0, 1);
    LAUNCHCHK("k_reloc_select");
    return km_launch_finalize(w, p, FIN_FROM_PARTIALS, 2, stream, host_mapped, ticket);
}


// --------------------------------------------------------------------------------------
// The Lloyd loop of one fit on one GPU as ONE call: batches of iterations, the look-ins (the status block arrives in pinned
// host memory, the calling thread polls the ticket), batch sizing from the decay of the centre shift, and the windowed
// relocation of empty clusters -- everything the host does between two launches, without leaving the library.  It comes
// back when the fit has stopped (status.done) or needs something only the caller can do: the full-pass relocation
// (status.paused == 2: the proof of a windowed selection failed; paused == 1: windows not applicable, the strict-convergence
// check is due, or the scratch is too small).  The caller handles that and calls again.
// --------------------------------------------------------------------------------------
int nnc_wait_ticket_(void *host_ticket, uint64_t ticket, void *stream) // (shared with nnc_layer.hip)
{
    return km_wait_ticket(reinterpret_cast<volatile unsigned long long *>(host_ticket), (unsigned long long)ticket, S(stream));
}
int km_wait_ticket(volatile unsigned long long *word, unsigned long long ticket, hipStream_t stream)
{
    unsigned long long spins = 0;
    bool synced = false;
    auto t0 = std::chrono::steady_clock::now();
    while (__atomic_load_n(word, __ATOMIC_ACQUIRE) != ticket) { // (acquire: the payload the device wrote in front of the ticket is read after it)
        if ((++spins & 0x3FFF) == 0) {
            const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (el > 0.02 && !synced) { // a non-coherent mapping, or a device error: make the write visible / surface the error
                hipError_t e = hipStreamSynchronize(stream);
                if (e != hipSuccess) return fail(NNC_EHIP, std::string("nnc_kmeans_fit: ") + hipGetErrorString(e));
                synced = true;
            } else if (el > 60.0) return fail(NNC_EHIP, "nnc_kmeans_fit: the status never arrived");
        }
    }
    return NNC_OK;
}

extern "C" int nnc_kmeans_fit(const float *x_iter, void *ws, const nnc_kmeans_params *pp, int32_t max_batch, int32_t sorted,
                              void *reloc_scratch_dev, size_t reloc_scratch_bytes, void *host_mapped, uint64_t *ticket_io,
                              nnc_kmeans_status *status_out, int32_t *n_windowed_out, void *stream)
{
    int rc = km_check(ws, pp, "nnc_kmeans_fit");
    if (rc) return rc;
    if (!host_mapped || !ticket_io || !status_out || (reinterpret_cast<uintptr_t>(host_mapped) & 7) != 0) return fail(NNC_EINVAL, "nnc_kmeans_fit: null / unaligned pointer");
    const nnc_kmeans_params p = *pp;
    if (p.n != p.n_total) return fail(NNC_EINVAL, "nnc_kmeans_fit: single GPU only (sharded fits: nnc_kmeans_iterate_sharded)");
    if (max_batch < 1) max_batch = 1;
    const bool lloyd = km_lloyd_ok(&p, x_iter);
    const bool one_launch = !lloyd && km_one_launch_fit(&p, x_iter);
    const size_t slot = sizeof(nnc_kmeans_status) + 8;
    unsigned char *hb = reinterpret_cast<unsigned char *>(host_mapped); // two slots, used alternately
    // Empty clusters come in runs (duplicate initial centres: the bench fit pauses in iterations 0-9, 11, 16, 19), and an event
    // the host has to see costs a round trip.  So, where the windowed relocation applies, the iterations of a batch are
    // followed by the relocation chain "in case" (km_launch_spec_reloc: five launches that do nothing without an event), until
    // a batch goes by without one.
    const bool spec_ok = sorted && !one_launch && p.prefix_dev && reloc_scratch_dev && (reinterpret_cast<uintptr_t>(reloc_scratch_dev) & 255) == 0 &&
                         reloc_scratch_bytes >= nnc_kmeans_reloc_scratch_bytes(p.k, KM_SPEC_WMAX) && p.n >= 2 * KM_SPEC_WMAX;
    // Round 4, opt-in (flags & NNC_KM_MASS_IN_PLACE): the finalize step of a launch-per-iteration pass settles mass events itself as well
    // (kl_relocate_mass) and that form then carries no chain.  Same trajectory; measured on the bench fit it does not pay (12 of 13
    // events in place, a mass event 92-100 us in the finalize launch against about 95 through the chain: 2.87 ms per step against 2.76),
    // so the default stays the chain.
    const bool mass_in_place = (p.flags & NNC_KM_MASS_IN_PLACE) && !lloyd && p.k > 64;
    bool spec = spec_ok && !mass_in_place;
    int batch = one_launch ? p.max_iter : (spec_ok ? 6 : 1); // the first iteration is where duplicate initial centres surface as empty clusters
                                                              // (six with chains: the mass events come first; small ones need no chain)
    if (lloyd) batch = KM_LLOYD_ROUNDS; // rounds per look-in: a round ends at an empty cluster (settled by the chain behind it, if there is one) or a handed-over iteration
    bool first_launch = true;
    int nwin = 0, nrel_seen = status_out->n_relocated, nunp_seen = status_out->n_unproven; // (a call after a full-pass relocation carries on from the last status)
    int nip_seen = status_out->n_in_place;
    if (!spec_ok) { nrel_seen = 0; nunp_seen = 0; nip_seen = 0; }
    double s_prev = -1.0, s_last = -1.0;
    int i_prev = 0, i_last = 0;
    KmWs *w = reinterpret_cast<KmWs *>(ws);
    for (;;) {
        const uint64_t ticket = ++(*ticket_io);
        unsigned char *sl = hb + (ticket & 1) * slot;
        if (lloyd) {
            if ((rc = km_set_lds_attr())) return rc;
            for (int i = 0; i < batch; i++) {
                const bool last = i == batch - 1;
                if ((rc = km_launch_lloyd_round(x_iter, w, &p, first_launch ? p.max_iter : -1, stream, (last && !spec) ? sl : nullptr, ticket, spec ? 1 : 0))) return rc;
                first_launch = false;
                if (spec && (rc = km_launch_spec_reloc(x_iter, w, &p, reloc_scratch_dev, stream, last ? sl : nullptr, ticket))) return rc;
            }
        } else if (spec) {
            if ((rc = km_set_lds_attr())) return rc;
            for (int i = 0; i < batch; i++) {
                if ((rc = km_launch_accumulate(x_iter, w, &p, stream))) return rc;
                if ((rc = km_launch_finalize(w, &p, FIN_FROM_SHARDS, 0, stream, nullptr, 0, false, x_iter))) return rc; // (small events: in place)
                if ((rc = km_launch_spec_reloc(x_iter, w, &p, reloc_scratch_dev, stream, i == batch - 1 ? sl : nullptr, ticket))) return rc;
            }
        } else {
            // The status of a plain batch leaves two iterations before its end: the round trip to the host and the enqueuing of the
            // next batch (8-13 us, seven to nine times a fit) then happen while those two are running, not on an idle GPU.  What the
            // host has not seen yet costs nothing but launches: every kernel of an iteration looks at the status itself and returns
            // at once when the fit is over or has paused, so a batch enqueued behind a convergence the host learns of a look-in
            // later is a row of no-ops; the trajectory is the device's alone.
            const int lag = (!one_launch && batch >= 4) ? KM_FIT_LAG : 0;
            if ((rc = km_iterate_publish_(x_iter, ws, &p, batch, sl, ticket, stream, spec_ok, lag))) return rc;
        }
        if ((rc = km_wait_ticket(reinterpret_cast<volatile unsigned long long *>(sl + sizeof(nnc_kmeans_status)), ticket, S(stream)))) return rc;
        const nnc_kmeans_status st = *reinterpret_cast<const nnc_kmeans_status *>(sl);
        *status_out = st;
        const int dev_events = spec_ok ? st.n_relocated - nrel_seen : 0; // events the device settled by itself in this batch
        const int chain_events = dev_events - (spec_ok ? st.n_in_place - nip_seen : 0); // ... of them by a chain enqueued in case
        nrel_seen = st.n_relocated;
        nip_seen = st.n_in_place;
        nwin += dev_events;
        // an unproven windowed selection comes back as paused == 2 and the caller takes one windowed event back before it redoes
        // it in full: the attempts of the chains enqueued "in case" are counted like the ones this loop asks for itself
        nwin += spec_ok ? st.n_unproven - nunp_seen : 0;
        nunp_seen = st.n_unproven;
        if (st.done) break;
        if (lloyd && !st.paused) continue; // every round met an event the device settled: the same number of rounds again
        if (spec && !st.paused) {
            // a chain that finds nothing to do still costs its five launches (about as much as the round trip it would have saved):
            // carry on only while most iterations pause
            // (events the iterations settle in their own launch do not need one)
            if (2 * chain_events >= batch) { batch = 4; s_prev = s_last = -1.0; continue; }
            spec = false; // back to plain iterations, sized by the decay of the shift
            s_prev = s_last = -1.0;
        }
        if (st.paused) {
            const bool strict_check = st.iter >= 1 && st.same_counts;
            const int32_t window = (sorted && st.paused == 1 && !strict_check) ? nnc_kmeans_reloc_window(p.n, st.n_empty) : 0;
            if (window == 0 || !reloc_scratch_dev || reloc_scratch_bytes < nnc_kmeans_reloc_scratch_bytes(p.k, window)) break; // the caller's turn
            if ((rc = nnc_kmeans_relocate_windowed(x_iter, ws, &p, st.n_empty, reloc_scratch_dev, reloc_scratch_bytes, stream))) return rc;
            nwin++;
            if (!lloyd) batch = one_launch ? p.max_iter : 1;
            s_prev = s_last = -1.0;
            continue;
        }
        if (one_launch) continue;
        // size the next batch so that it ends about where the shift crosses the tolerance (launches enqueued after convergence
        // are no-ops, but they still cost a dispatch)
        s_prev = s_last; i_prev = i_last;
        s_last = (double)st.shift_tot; i_last = st.iter;
        batch = std::min(max_batch, batch * 2);
        if (s_prev > 0.0 && s_last > 0.0 && s_prev > s_last && p.tol > 0.0f) {
            const double rate = std::log(s_prev / s_last) / std::max(1, i_last - i_prev);
            const double left = s_last > (double)p.tol ? std::log(s_last / (double)p.tol) / rate : 0.0;
            batch = (int)std::max(1.0, std::min((double)max_batch, std::floor(left * 0.9)));
        }
    }
    if (n_windowed_out) *n_windowed_out = nwin;
    return NNC_OK;
}

// Sharded vector: every rank selects (and proves) its own n_empty farthest samples from its shard's windows; the ranks
// then all-gather the keys and all-reduce (MAX) the verdict word nnc_kmeans_reloc_flag(); the merged top n_empty go to
// nnc_kmeans_relocate_if_proven, which does nothing if any rank's proof failed (the resumed finalize then reports
// paused = 2 on every rank and the full-pass form takes over).
extern "C" int nnc_kmeans_reloc_select_local(const float *x_sorted, void *ws, const nnc_kmeans_params *p, int32_t n_empty,
                                             void *scratch_dev, size_t scratch_bytes, int64_t *keys_out_dev, void *stream)
{
    int rc = km_check(ws, p, "nnc_kmeans_reloc_select_local");
    if (rc) return rc;
    const int32_t window = nnc_kmeans_reloc_window(p->n, n_empty);
    if (window == 0) return fail(NNC_EINVAL, "nnc_kmeans_reloc_select_local: not applicable (nnc_kmeans_reloc_window() == 0)");
    if (!x_sorted || !scratch_dev || !keys_out_dev || (reinterpret_cast<uintptr_t>(scratch_dev) & 255) != 0)
        return fail(NNC_EINVAL, "nnc_kmeans_reloc_select_local: null or unaligned (256 B) pointer");
    if (scratch_bytes < nnc_kmeans_reloc_scratch_bytes(p->k, window)) return fail(NNC_ENOSPACE, "nnc_kmeans_reloc_select_local: scratch too small");
    const int64_t cap = reloc_cap(p->k, window);
    unsigned char *b = reinterpret_cast<unsigned char *>(scratch_dev);
    float *cand_x = reinterpret_cast<float *>(b); b += reloc_align((size_t)cap * 4);
    float *cand_d = reinterpret_cast<float *>(b); b += reloc_align((size_t)cap * 4);
    void *win = b; b += reloc_align(16 * (size_t)(p->k + 2));
    int32_t *meta = reinterpret_cast<int32_t *>(b); b += reloc_align(16) + reloc_align(8 * (size_t)NNC_KMAX);
    unsigned *hist0 = reinterpret_cast<unsigned *>(b);
    if ((rc = km_reloc_windows_dist(x_sorted, ws, p, window, cand_x, cand_d, cap, win, meta, hist0, stream))) return rc;
    NNC_LAUNCH_PROF(NNC_PROF_RELOC, k_reloc_select, dim3(1), dim3(KM_THREADS), 0, S(stream), reinterpret_cast<KmWs *>(ws), cand_x, cand_d,
                       reinterpret_cast<const KmWin *>(win), reinterpret_cast<const int *>(meta), (int)n_empty,
                       reinterpret_cast<long long *>(keys_out_dev), 0, (const unsigned *)hist0, 0);
    LAUNCHCHK("k_reloc_select");
    return NNC_OK;
}

extern "C" int32_t *nnc_kmeans_reloc_flag(void *ws)
{
    if (!ws) return nullptr;
    return &reinterpret_cast<KmWs *>(ws)->reloc_fail;
}

__global__ __launch_bounds__(KM_THREADS) void k_relocate_if_proven(KmWs *__restrict__ ws, const long long *__restrict__ keys, int nkeys)
{
    __shared__ KmRelocLds lds;
    if (ws->reloc_fail) return;
    km_relocate_body(ws, keys, nkeys, &lds);
}

extern "C" int nnc_kmeans_relocate_if_proven(void *ws, const int64_t *keys_sorted_dev, int32_t nkeys, void *stream)
{
    if (!ws || !keys_sorted_dev || nkeys < 1) return fail(NNC_EINVAL, "nnc_kmeans_relocate_if_proven: bad argument");
    hipLaunchKernelGGL(k_relocate_if_proven, dim3(1), dim3(KM_THREADS), 0, S(stream), reinterpret_cast<KmWs *>(ws),
                       reinterpret_cast<const long long *>(keys_sorted_dev), nkeys);
    LAUNCHCHK("k_relocate_if_proven");
    return NNC_OK;
}

// flag = 1 if the two label vectors are identical, else 0
template <typename LT>
__global__ __launch_bounds__(256) void k_labels_equal(const LT *__restrict__ a, const LT *__restrict__ b, int64_t n, int *flag)
{
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t nthreads = (int64_t)gridDim.x * blockDim.x;
    int diff = 0;
    for (int64_t i = tid; i < n; i += nthreads) diff |= (a[i] != b[i]);
    if (__any(diff) && (threadIdx.x & 63) == 0) *flag = 0;
}
__global__ void k_set_i32(int *p, int v) { *p = v; }
__global__ void k_set_done_if(KmWs *ws, const int *flag, int code) { if (*flag) ws->st.done = code; }

extern "C" int nnc_labels_equal(const void *a, const void *b, int64_t n, int label_bytes, int32_t *flag_dev, void *stream)
{
    if (!flag_dev || n < 0 || (n > 0 && (!a || !b))) return fail(NNC_EINVAL, "nnc_labels_equal: bad argument");
    hipLaunchKernelGGL(k_set_i32, dim3(1), dim3(1), 0, S(stream), flag_dev, 1);
    LAUNCHCHK("k_set_i32");
    if (n == 0) return NNC_OK;
    // compare 8 bytes at a time when both are aligned and the byte count allows it
    const int64_t bytes = n * label_bytes;
    if (((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b)) & 7) == 0 && (bytes & 7) == 0) {
        int grid = stream_grid(bytes / 8, 256 * 4, 8);
        hipLaunchKernelGGL((k_labels_equal<unsigned long long>), dim3(grid), dim3(256), 0, S(stream), reinterpret_cast<const unsigned long long *>(a), reinterpret_cast<const unsigned long long *>(b), bytes / 8, flag_dev);
    } else if (label_bytes == 1) {
        int grid = stream_grid(n, 256 * 4, 8);
        hipLaunchKernelGGL((k_labels_equal<uint8_t>), dim3(grid), dim3(256), 0, S(stream), reinterpret_cast<const uint8_t *>(a), reinterpret_cast<const uint8_t *>(b), n, flag_dev);
    } else {
        int grid = stream_grid(n, 256 * 4, 8);
        hipLaunchKernelGGL((k_labels_equal<uint16_t>), dim3(grid), dim3(256), 0, S(stream), reinterpret_cast<const uint16_t *>(a), reinterpret_cast<const uint16_t *>(b), n, flag_dev);
    }
    LAUNCHCHK("k_labels_equal");
    return NNC_OK;
}

// --------------------------------------------------------------------------------------
// The M-step sums as scikit-learn runs them on one thread (_k_means_lloyd.pyx:215-218, _update_chunk_dense): for every cluster
// the float32 running sum of its members' centred values IN SAMPLE ORDER, starting from +0.0.  A sequential sum cannot be spread
// over lanes: one wave per cluster walks the label vector 256 samples at a time (loads well ahead of the chain), takes the members
// of each 64-sample chunk from a ballot and adds them one after the other (v_readlane + v_add_f32 per member).  The cost of a
// call is the size of the largest cluster times a few nanoseconds -- the opt-in "reference arithmetic" fit of tensors beyond
// NNC_REF_NMAX (kmeans.fit_reference_large), not the product's default path.
// --------------------------------------------------------------------------------------
template <typename LT>
__global__ __launch_bounds__(256) void k_ref_sums(const float *__restrict__ x, long long n, float mean, const LT *__restrict__ labels, int k,
                                                  float *__restrict__ sums, long long *__restrict__ counts)
{
    const int lane = threadIdx.x & 63;
    const int j = uni_i((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
    if (j >= k) return;
    float acc = 0.0f;
    long long cnt = 0;
    for (long long i0 = 0; i0 < n; i0 += 256) { // (wave-uniform trip count)
        bool mine[4];
        float xv[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const long long i = i0 + 64 * u + lane;
            mine[u] = i < n && (int)labels[i] == j;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) { // only the members' values are fetched
            const long long i = i0 + 64 * u + lane;
            xv[u] = mine[u] ? x[i] - mean : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            unsigned long long m = __ballot(mine[u]);
            cnt += __popcll(m);
            while (m) {
                const int b = __ffsll((long long)m) - 1;
                m &= m - 1ull;
                acc = acc + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(xv[u]), b));
            }
        }
    }
    if (lane == 0) { sums[j] = acc; counts[j] = cnt; }
}

extern "C" int nnc_ref_sums_f32(const float *x, int64_t n, float x_mean, const void *labels, int32_t label_bytes, int32_t k,
                                float *sums_out_dev, int64_t *counts_out_dev, void *stream)
{
    if (n < 0 || k < 1 || k > NNC_KMAX || (n > 0 && (!x || !labels)) || !sums_out_dev || !counts_out_dev || (label_bytes != 1 && label_bytes != 2))
        return fail(NNC_EINVAL, "nnc_ref_sums_f32: bad argument");
    const int grid = (k + 3) / 4;
    if (label_bytes == 1)
        hipLaunchKernelGGL((k_ref_sums<uint8_t>), dim3(grid), dim3(256), 0, S(stream), x, (long long)n, x_mean, reinterpret_cast<const uint8_t *>(labels), (int)k,
                           sums_out_dev, reinterpret_cast<long long *>(counts_out_dev));
    else
        hipLaunchKernelGGL((k_ref_sums<uint16_t>), dim3(grid), dim3(256), 0, S(stream), x, (long long)n, x_mean, reinterpret_cast<const uint16_t *>(labels), (int)k,
                           sums_out_dev, reinterpret_cast<long long *>(counts_out_dev));
    LAUNCHCHK("k_ref_sums");
    return NNC_OK;
}

extern "C" int nnc_kmeans_set_done_if(void *ws, const int32_t *flag_dev, int32_t done_code, void *stream)
{
    if (!ws || !flag_dev) return fail(NNC_EINVAL, "nnc_kmeans_set_done_if: null pointer");
    hipLaunchKernelGGL(k_set_done_if, dim3(1), dim3(1), 0, S(stream), reinterpret_cast<KmWs *>(ws), flag_dev, done_code);
    LAUNCHCHK("k_set_done_if");
    return NNC_OK;
}

