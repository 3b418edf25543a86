// nnc_reffit.hip -- the whole KMeans.fit of a SHORT tensor in one launch of one workgroup, in scikit-learn's own summation order.
#include "nnc_common.hpp"
#include "nnc_km_shared.hpp"

// ======================================================================================
// 4c. A whole fit of a SHORT tensor on chip, in the reference's own arithmetic
//
// For n <= NNC_REF_NMAX weights and k <= NNC_REF_KMAX centres one workgroup runs everything KMeans.fit does
// (utility.py:237-238 -> sklearn _kmeans.py:1427-1554, 624-752) without leaving the chip: NumPy's float32 mean and variance
// (the same pairwise tree as nnc_chunk_sums_f32), the centring, then per iteration the brute-force E-step with scikit-learn's
// float32 expression, the M-step as scikit-learn runs it on one thread -- float32 running sums IN SAMPLE ORDER
// (_k_means_lloyd.pyx:215-218) --, empty-cluster relocation, averaging with float32(1 / count), centre shift, the two
// stopping rules, the final E-step.  With the sums in the reference's order the result is the reference's, bit for bit
// (centres, indices, n_iter_), where the exact-integer sums of the long-vector path stay within its summation error; what
// remains open is what scikit-learn itself leaves to numpy.argpartition (pairing and ties at a relocation cut: reported).
// A sequential sum cannot be spread over lanes, so a wave takes a cluster and walks the label vector 64 samples at a time
// (ballot of the members, one add per member from the lanes' registers): the cost of an iteration is the size of the largest
// cluster times a few nanoseconds, which is why this is the path of short tensors only.
// ======================================================================================
#define REF_NMAX NNC_REF_NMAX
#define REF_KMAX NNC_REF_KMAX
#define REF_PER (REF_NMAX / KM_THREADS)

struct RefOut { int32_t n_iter, stop, n_relocations, reloc_ties, reloc_multi, pad; float x_mean, tol; };

// NumPy's pairwise sum of n <= 128 float32 (one leaf: eight strided accumulators, combined as a tree, then the ragged tail),
// by one wave on its own; every lane returns the sum.
template <typename F> __device__ __forceinline__ float wave_leaf_sum(F elem, int n)
{
    const int l8 = threadIdx.x & 7;
    float res = 0.0f;
    if (n < 8) {
        for (int i = 0; i < n; i++) res += elem(i);
    } else {
        float r = elem(l8);
        const int lim = n - (n % 8);
        for (int i = 8; i < lim; i += 8) r += elem(i + l8);
        r = r + __shfl_xor(r, 1);
        r = r + __shfl_xor(r, 2);
        r = r + __shfl_xor(r, 4);
        res = r;
        for (int i = lim; i < n; i++) res += elem(i);
    }
    return res;
}

// One link of sixteen running sums: on entry lane 15 of every row of 16 lanes holds that row's sum so far (s) and lane r the
// row's next value (v; +0.0 beyond the end of the row's list, which leaves a sum as it is: a float32 sum that started at
// +0.0 is never -0.0).  The sum moves to lane 0 (row rotate), lane 0 adds its value, then fifteen adds each take the sum from
// the lane below (row shift: lane 0 has no lane below and is left alone), so that lane r ends with s + v0 + ... + vr added in
// exactly that order.  One VALU instruction per member and row; the s_nop are the two wait states a DPP read of a register
// written by the instruction before needs, which the compiler does not insert inside an asm block.
#define REF_DPP_ADD "s_nop 1\n\tv_add_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
__device__ __forceinline__ float ref_chain16(float s, float v)
{
    asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_add_f32 %0, %0, %1\n\t" REF_DPP_ADD REF_DPP_ADD REF_DPP_ADD REF_DPP_ADD REF_DPP_ADD REF_DPP_ADD REF_DPP_ADD REF_DPP_ADD
                     REF_DPP_ADD REF_DPP_ADD REF_DPP_ADD REF_DPP_ADD REF_DPP_ADD REF_DPP_ADD REF_DPP_ADD "s_nop 1\n\t"
                 : "+v"(s)
                 : "v"(v));
    return s;
}

__global__ __launch_bounds__(KM_THREADS) void k_fit_reference(const float *__restrict__ x, int n, const float *__restrict__ init, int k, int max_iter,
                                                             float tol_rel, uint8_t *__restrict__ labels_out, float *__restrict__ values_out,
                                                             float *__restrict__ centers_out, long long *__restrict__ counts_out, RefOut *__restrict__ out)
{
    __shared__ float xc[REF_NMAX];
    __shared__ float dist[REF_NMAX]; // the members of every cluster in sample order, cluster after cluster (M-step); the squared distances (relocation)
    __shared__ uint8_t labs[2][REF_NMAX];
    __shared__ __align__(16) float cen[REF_KMAX], csq[REF_KMAX];
    __shared__ float cnew[REF_KMAX], sums[REF_KMAX], wic[REF_KMAX], sq[REF_KMAX];
    __shared__ unsigned hist[REF_KMAX];
    __shared__ int base[REF_KMAX];
    // tab[label][chunk of 64 samples]: members of the cluster in the chunk, then where the chunk's first member goes;
    // the start-up statistics use the same bytes for their tree
    __shared__ __align__(16) unsigned char scratch[REF_KMAX * (REF_NMAX / 64) * 2];
    static_assert(sizeof(PwHeap) <= sizeof(scratch), "scratch holds the pairwise-sum tree");
    PwHeap &heap = *reinterpret_cast<PwHeap *>(scratch);
    uint16_t(*tab)[REF_NMAX / 64] = reinterpret_cast<uint16_t(*)[REF_NMAX / 64]>(scratch);
    __shared__ unsigned long long wave_key[KM_THREADS / 64];
    __shared__ int wave_idx[KM_THREADS / 64];
    __shared__ int s_flag, s_empty[REF_KMAX], s_nempty, s_far[REF_KMAX], s_moved;
    __shared__ float s_tot;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int k4 = (k + 3) & ~3;
    const int nbits = k > 1 ? 32 - __builtin_clz((unsigned)(k - 1)) : 0; // bits of a centroid index
#ifdef NNC_DIAG
    long long tph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = wall_clock64(); // phase times of thread 0, 10 ns units, behind the result block
#define REFSTAMP(p) { const long long now_ = wall_clock64(); tph[p] += now_ - tlast; tlast = now_; }
#else
#define REFSTAMP(p)
#endif

    // ---- X_mean = X.mean(axis=0); tol = np.mean(np.var(X, axis=0)) * tol; X -= X_mean   (all float32, NumPy's pairwise sums)
    for (int i = tid; i < n; i += KM_THREADS) xc[i] = x[i];
    __syncthreads();
    const float s1 = block_pairwise_sum<false>([&](int i) { return xc[i]; }, n, &heap);
    const float mean = (float)((double)s1 / (double)n);
    __syncthreads();
    const float s2 = block_pairwise_sum<false>([&](int i) { const float a = xc[i] - mean; return a * a; }, n, &heap);
    const float var = (float)((double)s2 / (double)n);
    const float tol = var * tol_rel;
    __syncthreads();
    float xv[REF_PER];
#pragma unroll
    for (int u = 0; u < REF_PER; u++) {
        const int i = tid + u * KM_THREADS;
        xv[u] = i < n ? xc[i] - mean : 0.0f;
        if (i < n) { xc[i] = xv[u]; labs[1][i] = 255; }
    }
    // centres beyond k: never the arg-min (c^2 = +inf)
    if (tid < REF_KMAX) { const float c = tid < k ? init[tid] - mean : 0.0f; cen[tid] = c; csq[tid] = tid < k ? c * c : __builtin_inff(); hist[tid] = 0u; }
    for (int q = tid; q < k * (REF_NMAX / 64) / 2; q += KM_THREADS) reinterpret_cast<unsigned *>(scratch)[q] = 0u;
    if (tid == 0) s_nempty = 0;
    __syncthreads();

    // E-step: first strict minimum of fl(c^2) + fl(-2 * fl(x * c)) over the centres in index order; index histogram, and per
    // sample its rank among the members of its cluster inside its chunk
    int rank[REF_PER], lfin[REF_PER];
    auto estep = [&](uint8_t *__restrict__ lab, const uint8_t *__restrict__ lab_prev) -> int {
        float best[REF_PER];
        int l[REF_PER];
#pragma unroll
        for (int u = 0; u < REF_PER; u++) { best[u] = __builtin_inff(); l[u] = 0; }
        for (int j = 0; j < k4; j += 4) {
            const float4 c4 = *reinterpret_cast<const float4 *>(&cen[j]), q4 = *reinterpret_cast<const float4 *>(&csq[j]);
            const float cc[4] = {c4.x, c4.y, c4.z, c4.w}, qq[4] = {q4.x, q4.y, q4.z, q4.w};
#pragma unroll
            for (int t = 0; t < 4; t++) {
#pragma unroll
                for (int u = 0; u < REF_PER; u++) {
                    const float d = qq[t] + (-2.0f * (xv[u] * cc[t]));
                    if (d < best[u] || (j + t) == 0) { best[u] = d; l[u] = j + t; }
                }
            }
        }
        int same = 1;
#pragma unroll
        for (int u = 0; u < REF_PER; u++) {
            const int i = tid + u * KM_THREADS; // the wave's lanes hold 64 consecutive samples: chunk wv + 16 u
            const bool valid = i < n;
            // the lanes of the chunk with the same index (a ballot per index bit); rank[u] = how many of them come before
            unsigned long long peers = __ballot(valid);
            for (int b = 0; b < nbits; b++) {
                const bool bit = (l[u] >> b) & 1;
                const unsigned long long mb = __ballot(valid && bit);
                peers &= bit ? mb : ~mb;
            }
            rank[u] = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(peers >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)peers, 0u));
            lfin[u] = l[u];
            if (valid) {
                lab[i] = (uint8_t)l[u];
                if (lab_prev) same &= (lab_prev[i] == (uint8_t)l[u]);
                if (rank[u] == 0) {
                    const int c = __popcll(peers);
                    tab[l[u]][wv + u * (KM_THREADS / 64)] = (uint16_t)c;
                    atomicAdd(&hist[l[u]], (unsigned)c);
                }
            }
        }
        return same;
    };

    REFSTAMP(0)
    int n_iter = 0, stop = 2, n_reloc = 0, n_ties = 0, n_multi = 0;
    bool strict = false;
    int cur = 0;
    for (int it = 0; it < max_iter; it++) {
        uint8_t *lab = labs[cur];
        const int all_same = __syncthreads_and(estep(lab, labs[cur ^ 1]));
        REFSTAMP(1)
        // ---- M-step: float32 running sums in sample order (_k_means_lloyd.pyx:215-218).  The members of every cluster are first
        // put, in sample order, into a stretch of `dist` of their own (a stable counting sort: stretch start = members of
        // lower clusters, from the histogram; inside it, members in earlier chunks, from a wave scan of the table row; then the
        // rank inside the chunk).  Then a row of 16 lanes per cluster, four clusters per wave, walks its stretch 16 members
        // per link (ref_chain16): one add per member, in sample order.
        if (wv == KM_THREADS / 64 - 1) { // (this wave also lists the empty clusters)
            const bool e0 = lane < k && hist[lane] == 0u, e1 = lane + 64 < k && hist[lane + 64] == 0u;
            const unsigned long long m0 = __ballot(e0), m1 = __ballot(e1);
            if (e0) s_empty[__builtin_amdgcn_mbcnt_hi((unsigned)(m0 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m0, 0u))] = lane;
            if (e1) s_empty[__popcll(m0) + __builtin_amdgcn_mbcnt_hi((unsigned)(m1 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m1, 0u))] = lane + 64;
            if (lane == 0) s_nempty = __popcll(m0) + __popcll(m1);
        }
        for (int j = wv; j < k; j += KM_THREADS / 64) {
            const int cj = (int)hist[j];
            int below = 0;
            for (int q = lane; q < j; q += 64) below += (int)hist[q];
            for (int off = 32; off >= 1; off >>= 1) below += __shfl_xor(below, off);
            if (lane == 0) base[j] = below;
            if (cj == 0) continue;
            const int mine = (int)tab[j][lane];
            int incl = mine;
            for (int off = 1; off < 64; off <<= 1) { const int t = __shfl_up(incl, off); if (lane >= off) incl += t; }
            tab[j][lane] = (uint16_t)(below + incl - mine);
        }
        __syncthreads();
        REFSTAMP(2)
#pragma unroll
        for (int u = 0; u < REF_PER; u++) {
            const int i = tid + u * KM_THREADS;
            if (i < n) dist[(int)tab[lfin[u]][wv + u * (KM_THREADS / 64)] + rank[u]] = xv[u];
        }
        __syncthreads();
        for (int r0 = 0; r0 < k; r0 += KM_THREADS / 16) {
            const int j = r0 + wv * 4 + (lane >> 4), rl = lane & 15;
            const int cj = j < k ? (int)hist[j] : 0, bj = j < k ? base[j] : 0;
            int maxc = max(cj, __shfl_xor(cj, 16));
            maxc = uni_i(max(maxc, __shfl_xor(maxc, 32)));
            float s = 0.0f;
            float v = rl < cj ? dist[bj + rl] : 0.0f;
            for (int c0 = 0; c0 < maxc; c0 += 16) {
                const float vn = c0 + 16 + rl < cj ? dist[bj + c0 + 16 + rl] : 0.0f;
                s = ref_chain16(s, v);
                v = vn;
            }
            if (rl == 15 && j < k) { sums[j] = s; wic[j] = (float)cj; }
        }
        // (waves without a cluster get here at once) the table is cleared for the next iteration
        for (int q = tid; q < k * (REF_NMAX / 64) / 2; q += KM_THREADS) reinterpret_cast<unsigned *>(scratch)[q] = 0u;
        __syncthreads();
        REFSTAMP(3)
        // ---- _relocate_empty_clusters_dense (_k_means_common.pyx:167-211)
        const int n_empty = s_nempty;
        if (n_empty > 0) {
            for (int i = tid; i < n; i += KM_THREADS) { const float t = xc[i] - cen[lab[i]]; dist[i] = t * t; }
            if (tid == 0) s_flag = 0;
            __syncthreads();
            // the n_empty farthest samples: descending distance, equal distances by descending value, then by descending index
            // (numpy leaves the order to its introselect; this is the order it was observed to leave on 69 of 70 reference fits)
            for (int r = 0; r <= n_empty && r < n; r++) { // one more than needed: the runner-up shows a tie at the cut
                unsigned long long bk = 0ull;
                int bi = -1;
                for (int i = tid; i < n; i += KM_THREADS) {
                    const float d = dist[i];
                    if (d >= 0.0f) {
                        const unsigned xb = __float_as_uint(xc[i]);
                        const unsigned long long key = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned long long)((xb & 0x80000000u) ? ~xb : (xb | 0x80000000u));
                        if (bi < 0 || key > bk || (key == bk && i > bi)) { bk = key; bi = i; }
                    }
                }
                for (int off = 32; off >= 1; off >>= 1) {
                    const unsigned long long ok = __shfl_xor(bk, off);
                    const int oi = __shfl_xor(bi, off);
                    if (oi >= 0 && (bi < 0 || ok > bk || (ok == bk && oi > bi))) { bk = ok; bi = oi; }
                }
                if (lane == 0) { wave_key[wv] = bk; wave_idx[wv] = bi; }
                __syncthreads();
                if (tid == 0) {
                    unsigned long long k0 = 0ull;
                    int i0 = -1;
                    for (int w = 0; w < KM_THREADS / 64; w++)
                        if (wave_idx[w] >= 0 && (i0 < 0 || wave_key[w] > k0 || (wave_key[w] == k0 && wave_idx[w] > i0))) { k0 = wave_key[w]; i0 = wave_idx[w]; }
                    if (r < n_empty) { s_far[r] = i0; dist[i0] = -1.0f; } // taken (n_empty < k <= n: there is always one left)
                    else if (i0 >= 0) {
                        // the runner-up: a DIFFERENT sample value at the same non-zero distance as the last one taken is a tie
                        // at the cut (which of the two scikit-learn takes is numpy.argpartition's business)
                        const int last = s_far[n_empty - 1];
                        const float tl = xc[last] - cen[lab[last]], dl = tl * tl;
                        s_flag = (dl != 0.0f) && ((unsigned)(k0 >> 32) == __float_as_uint(dl)) && (xc[i0] != xc[last]);
                    }
                }
                __syncthreads();
            }
            if (tid == 0) {
                const int first = s_far[0];
                const float t0 = xc[first] - cen[lab[first]], dmax = t0 * t0;
                if (dmax != 0.0f) { // (np.max(distances) == 0: nothing is relocated)
                    for (int q = 0; q < n_empty; q++) {
                        const int nw = s_empty[q], fi = s_far[q], old = lab[fi];
                        const float v = xc[fi] * 1.0f;
                        sums[old] = sums[old] - v;
                        sums[nw] = v;
                        wic[nw] = 1.0f;
                        wic[old] = wic[old] - 1.0f;
                    }
                    s_moved = 1;
                } else s_moved = 0;
            }
            __syncthreads();
            if (s_moved) { n_reloc++; if (n_empty > 1) n_multi++; if (s_flag) n_ties++; }
        }
        REFSTAMP(4)
        // ---- _average_centers (_k_means_common.pyx:274-296): in place and in index order, so an empty cluster copies the biggest
        // one averaged if that comes before it and its raw sum otherwise; _center_shift; NumPy's sum of the squared shifts.
        // One wave, two clusters per lane.
        if (wv == 0) {
            unsigned key = 0u; // the first maximum of the counts: largest (count bits, 255 - index)
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int j = lane + 64 * h;
                if (j < k) key = max(key, (__float_as_uint(wic[j]) & 0xFFFFFF00u) | (unsigned)(255 - j)); // (counts <= 4096: the low 8 mantissa bits are 0)
            }
            for (int off = 32; off >= 1; off >>= 1) key = max(key, (unsigned)__shfl_xor((int)key, off));
            const int amax = 255 - (int)(key & 0xFFu);
            const float big_raw = sums[amax], big_avg = big_raw * (float)(1.0 / (double)wic[amax]);
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int j = lane + 64 * h;
                if (j < k) {
                    const float w = wic[j];
                    const float c = w > 0.0f ? sums[j] * (float)(1.0 / (double)w) : (amax < j ? big_avg : big_raw);
                    const float t = c - cen[j], sh = sqrtf(t * t);
                    cnew[j] = c;
                    sq[j] = sh * sh;
                }
            }
            wave_lds_fence();
            const float tot = wave_leaf_sum([&](int i) { return sq[i]; }, k);
            if (lane == 0) s_tot = tot;
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int j = lane + 64 * h;
                if (j < k) { const float c = cnew[j]; cen[j] = c; csq[j] = c * c; }
                hist[j] = 0u;
            }
        }
        __syncthreads();
        REFSTAMP(5)
        n_iter = it + 1;
        if (all_same) { strict = true; stop = 3; break; }
        if (s_tot <= tol) { stop = 1; break; }
        cur ^= 1;
    }
    // ---- labels of the final centres unless the label test stopped the loop (then cur still names that iteration's labels and
    // the histogram is rebuilt from them); centres += X_mean; cluster_centers_[labels_]
    uint8_t *lab = labs[cur];
    if (!strict) estep(lab, nullptr);
    else
        for (int i = tid; i < n; i += KM_THREADS) atomicAdd(&hist[lab[i]], 1u);
    if (tid < k) { cnew[tid] = cen[tid] + mean; centers_out[tid] = cnew[tid]; }
    __syncthreads();
    for (int i = tid; i < n; i += KM_THREADS) {
        labels_out[i] = lab[i];
        if (values_out) values_out[i] = cnew[lab[i]];
    }
    if (counts_out && tid < k) counts_out[tid] = (long long)hist[tid];
#ifdef NNC_DIAG
    REFSTAMP(6)
    if (tid == 0)
        for (int q = 0; q < 8; q++) reinterpret_cast<long long *>(out + 1)[q] = tph[q];
#endif
    if (tid == 0) { out->n_iter = n_iter; out->stop = stop; out->n_relocations = n_reloc; out->reloc_ties = n_ties; out->reloc_multi = n_multi; out->pad = 0; out->x_mean = mean; out->tol = tol; }
}

extern "C" int nnc_kmeans_fit_reference_f32(const float *x, int32_t n, const float *centers_init_dev, int32_t k, int32_t max_iter, float tol,
                                            uint8_t *labels_out, float *values_out, float *centers_out, int64_t *counts_out,
                                            void *result_dev, void *stream)
{
    if (!x || n < 1 || n > REF_NMAX || !centers_init_dev || k < 1 || k > REF_KMAX || n < k || max_iter < 1 || !labels_out || !centers_out || !result_dev)
        return fail(NNC_EINVAL, "nnc_kmeans_fit_reference_f32: bad argument (1 <= k <= NNC_REF_KMAX, k <= n <= NNC_REF_NMAX)");
    hipLaunchKernelGGL(k_fit_reference, dim3(1), dim3(KM_THREADS), 0, S(stream), x, (int)n, centers_init_dev, (int)k, (int)max_iter, tol, labels_out,
                       values_out, centers_out, reinterpret_cast<long long *>(counts_out), reinterpret_cast<RefOut *>(result_dev));
    LAUNCHCHK("k_fit_reference");
    return NNC_OK;
}

