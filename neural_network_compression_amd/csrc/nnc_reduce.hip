// nnc_reduce.hip -- the passes over a tensor that are not k-means: NumPy-exact float32 reductions (sigma, mean, variance), the
// threshold pass of prune_weigth (/root/reference/neural_network_compression/common/utility.py:158-163), min / max / sign counts,
// the 31-bin histogram of get_weight_distribution (utility.py:366-372), ranks in a sorted vector, the index histogram.
#include "nnc_common.hpp"

// ======================================================================================
// 1. NumPy-exact float32 reductions
//
// np.add.reduce over float32 walks the array in 8192-element buffered chunks; each chunk is
// summed by the pairwise routine: blocks of <=128 elements with 8 strided accumulators
// combined ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)), blocks merged as a binary tree (split at
// n/2 rounded down to a multiple of 8); the chunk sums are folded left to right.  A full
// 8192 chunk is a perfectly balanced tree of 64 leaves, which maps onto a wave:
//   * 8 leaves (1024 elements, 4 KiB) per step, loaded as coalesced float4 and transposed
//     through LDS (leaf stride padded to 136 floats: conflict-free column reads);
//   * lane (leaf b, accumulator j) adds its 16 elements sequentially;
//   * xor-shuffles 1,2,4 build the leaf, 8,16,32 the 1024-element node (IEEE addition is
//     commutative, so both partners of a butterfly hold the same bits);
//   * the 8 step nodes are merged in registers as a balanced tree.
// ======================================================================================
#define LEAF_PAD 136
#define STEP_ELEMS 1024

template <bool SQDEV>
__device__ __forceinline__ float4 xform4(float4 v, float mean)
{
    if (SQDEV) {
        float a = v.x - mean, b = v.y - mean, c = v.z - mean, d = v.w - mean;
        v.x = a * a; v.y = b * b; v.z = c * c; v.w = d * d;
    }
    return v;
}

template <bool SQDEV>
__device__ __forceinline__ float xform1(float v, float mean)
{
    if (SQDEV) { float a = v - mean; return a * a; }
    return v;
}

// The ragged last chunk (m < 8192 elements): the generic workgroup-parallel pairwise sum.
struct PwFrame { int start, len, stage; float left; };

// One WORKGROUP (four waves) per full chunk: wave w takes the 1024-element steps 2w and 2w + 1 (both loaded at once), the eight step
// nodes meet in LDS and one lane merges them as NumPy's tree does.  (Round 4: one wave used to walk all eight steps of its chunk with
// one step of loads ahead -- 3052 chunks gave every wave a chain of eight dependent round trips and left two thirds of the wave
// slots empty: 20.9-23.5 us for 100 MB.)  With a ragged last chunk the grid has one more workgroup, which sums it beside the others.
template <bool SQDEV, bool VEC>
__global__ __launch_bounds__(256) void k_chunk_sums(const float *__restrict__ x, int64_t nfull,
                                                    const float *__restrict__ mean_dev,
                                                    float *__restrict__ out, int tail)
{
    __shared__ __align__(16) float lds[4][8 * LEAF_PAD];
    __shared__ float nodes[8];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float *my = lds[wave];
    const float mean = SQDEV ? *mean_dev : 0.0f;
    const int64_t gmain = (int64_t)gridDim.x - (tail > 0 ? 1 : 0); // workgroups that take full chunks
    if (tail > 0 && blockIdx.x == gridDim.x - 1) {
        __shared__ PwHeap heap;
        const float *xt = x + nfull * NNC_CHUNK;
        const float r = block_pairwise_sum([&](int i) { return xform1<SQDEV>(xt[i], mean); }, tail, &heap);
        if (threadIdx.x == 0) out[nfull] = r;
        return;
    }
    for (int64_t chunk = (int64_t)blockIdx.x; chunk < nfull; chunk += gmain) {
        const float *base = x + chunk * NNC_CHUNK + (int64_t)wave * 2 * STEP_ELEMS;
        float4 cur[2][4];
#pragma unroll
        for (int st = 0; st < 2; st++) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                if (VEC) cur[st][r] = reinterpret_cast<const float4 *>(base + st * STEP_ELEMS)[lane + 64 * r];
                else {
                    const float *p = base + st * STEP_ELEMS + 4 * (lane + 64 * r);
                    cur[st][r] = make_float4(p[0], p[1], p[2], p[3]);
                }
            }
        }
#pragma unroll
        for (int st = 0; st < 2; st++) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                int e = 4 * (lane + 64 * r); // element index inside the 1024-element step
                int b = e >> 7, pos = e & 127;
                *reinterpret_cast<float4 *>(&my[b * LEAF_PAD + pos]) = xform4<SQDEV>(cur[st][r], mean);
            }
            wave_lds_fence();
            const float *lp = &my[(lane >> 3) * LEAF_PAD + (lane & 7)];
            float acc = lp[0];
#pragma unroll
            for (int t = 1; t < 16; t++) acc = acc + lp[8 * t];
            wave_lds_fence();
            acc = acc + __shfl_xor(acc, 1);
            acc = acc + __shfl_xor(acc, 2);
            acc = acc + __shfl_xor(acc, 4);
            acc = acc + __shfl_xor(acc, 8);
            acc = acc + __shfl_xor(acc, 16);
            acc = acc + __shfl_xor(acc, 32);
            if (lane == 0) nodes[2 * wave + st] = acc;
        }
        __syncthreads();
        if (threadIdx.x == 0) out[chunk] = ((nodes[0] + nodes[1]) + (nodes[2] + nodes[3])) + ((nodes[4] + nodes[5]) + (nodes[6] + nodes[7]));
        __syncthreads(); // (nodes[] is reused by the next chunk)
    }
}

extern "C" int nnc_chunk_sums_f32(const float *x, int64_t n, int sqdev, const float *mean_dev,
                                  float *chunk_out, void *stream)
{
    if (n < 0 || (n > 0 && (!x || !chunk_out))) return fail(NNC_EINVAL, "nnc_chunk_sums_f32: null pointer");
    if (sqdev && !mean_dev) return fail(NNC_EINVAL, "nnc_chunk_sums_f32: sqdev needs mean_dev");
    if (n == 0) return NNC_OK;
    const int64_t nfull = n / NNC_CHUNK;
    const int tail = (int)(n % NNC_CHUNK);
    const bool vec = (reinterpret_cast<uintptr_t>(x) & 15) == 0;
    {
        int64_t blocks = nfull;                  // a workgroup a chunk
        int64_t cap = (int64_t)cu_count() * 16;  // (LDS: 17 KiB a workgroup)
        int grid = (int)std::min<int64_t>(blocks, cap) + (tail > 0 ? 1 : 0);
        if (sqdev) {
            if (vec) NNC_LAUNCH_PROF(NNC_PROF_CHUNK_SUMS, (k_chunk_sums<true, true>), dim3(grid), dim3(256), 0, S(stream), x, nfull, mean_dev, chunk_out, tail);
            else NNC_LAUNCH_PROF(NNC_PROF_CHUNK_SUMS, (k_chunk_sums<true, false>), dim3(grid), dim3(256), 0, S(stream), x, nfull, mean_dev, chunk_out, tail);
        } else {
            if (vec) NNC_LAUNCH_PROF(NNC_PROF_CHUNK_SUMS, (k_chunk_sums<false, true>), dim3(grid), dim3(256), 0, S(stream), x, nfull, mean_dev, chunk_out, tail);
            else NNC_LAUNCH_PROF(NNC_PROF_CHUNK_SUMS, (k_chunk_sums<false, false>), dim3(grid), dim3(256), 0, S(stream), x, nfull, mean_dev, chunk_out, tail);
        }
        LAUNCHCHK("k_chunk_sums");
    }
    return NNC_OK;
}

// Sequential float32 fold of the chunk sums (NumPy's order).  The chain of dependent adds is the
// whole cost (one add per chunk sum, nothing to parallelise); one lane runs it out of LDS with
// 16-byte reads issued well ahead, the others stage the next tile.
#define FOLD_TILE 8192
// (scale_val / zero64: round 4 -- the factor as an argument and a counter of the NEXT pass zeroed on the way, where the caller used
// to spend a launch each on parking the factor in device memory and on a memset)
__global__ __launch_bounds__(1024) void k_fold(const float *__restrict__ chunks, int64_t nchunks, int64_t count,
                                               int op, const float *__restrict__ scale_dev,
                                               float *__restrict__ out, float scale_val = 0.0f, int has_scale_val = 0,
                                               unsigned long long *__restrict__ zero64 = nullptr)
{
    if (zero64 && threadIdx.x == 0) *zero64 = 0ull;
    __shared__ __align__(16) float buf[FOLD_TILE];
    float acc = 0.0f;
    for (int64_t base = 0; base < nchunks; base += FOLD_TILE) {
        int len = (int)((nchunks - base) < FOLD_TILE ? (nchunks - base) : FOLD_TILE);
        for (int i = threadIdx.x; i < len; i += 1024) buf[i] = chunks[base + i];
        __syncthreads();
        if (threadIdx.x == 0) {
            const float4 *b4 = reinterpret_cast<const float4 *>(buf);
            const int nq = len >> 2;
            int qd = 0;
            for (; qd + 8 <= nq; qd += 8) {
                float4 v[8];
#pragma unroll
                for (int u = 0; u < 8; u++) v[u] = b4[qd + u];
#pragma unroll
                for (int u = 0; u < 8; u++) { acc = acc + v[u].x; acc = acc + v[u].y; acc = acc + v[u].z; acc = acc + v[u].w; }
            }
            for (; qd < nq; qd++) { const float4 v = b4[qd]; acc = acc + v.x; acc = acc + v.y; acc = acc + v.z; acc = acc + v.w; }
            for (int i = nq << 2; i < len; i++) acc = acc + buf[i];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        float r = acc;
        if (op == NNC_FOLD_MEAN || op == NNC_FOLD_STD) r = (float)((double)acc / (double)count);
        if (op == NNC_FOLD_STD) r = (float)sqrt((double)r); // double sqrt then round == correctly rounded sqrtf
        out[0] = r;
        if (scale_dev) out[1] = r * (*scale_dev);
        else if (has_scale_val) out[1] = r * scale_val;
    }
}

extern "C" int nnc_fold_f32(const float *chunks, int64_t nchunks, int64_t count, int op, const float *scale_dev,
                            float *out_dev, void *stream)
{
    if (!out_dev || nchunks < 0 || (nchunks > 0 && !chunks)) return fail(NNC_EINVAL, "nnc_fold_f32: bad argument");
    if ((op == NNC_FOLD_MEAN || op == NNC_FOLD_STD) && count <= 0) return fail(NNC_EINVAL, "nnc_fold_f32: count <= 0");
    hipLaunchKernelGGL(k_fold, dim3(1), dim3(1024), 0, S(stream), chunks, nchunks, count, op, scale_dev, out_dev, 0.0f, 0, (unsigned long long *)nullptr);
    LAUNCHCHK("k_fold");
    return NNC_OK;
}

// ======================================================================================
// 2. threshold pass: mask = |x| < thr, zero in place, count
// ======================================================================================
template <bool VEC>
__global__ __launch_bounds__(256) void k_threshold(float *__restrict__ x, int64_t n,
                                                   const float *__restrict__ thr_dev,
                                                   uint8_t *__restrict__ mask,
                                                   unsigned long long *__restrict__ nzeroed)
{
    const float thr = *thr_dev;
    unsigned cnt = 0;
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t nthreads = (int64_t)gridDim.x * blockDim.x;
    int64_t done = 0;
    if (VEC) {
        const int64_t nvec = n >> 2;
        float4 *x4 = reinterpret_cast<float4 *>(x);
        uchar4 *m4 = reinterpret_cast<uchar4 *>(mask);
        for (int64_t v = tid; v < nvec; v += nthreads) {
            float4 a = x4[v];
            uchar4 m;
            m.x = fabsf(a.x) < thr; m.y = fabsf(a.y) < thr; m.z = fabsf(a.z) < thr; m.w = fabsf(a.w) < thr;
            cnt += m.x + m.y + m.z + m.w;
            a.x = m.x ? 0.0f : a.x; a.y = m.y ? 0.0f : a.y; a.z = m.z ? 0.0f : a.z; a.w = m.w ? 0.0f : a.w;
            x4[v] = a;
            m4[v] = m;
        }
        done = nvec << 2;
    }
    for (int64_t i = done + tid; i < n; i += nthreads) {
        float a = x[i];
        uint8_t m = fabsf(a) < thr;
        cnt += m;
        if (m) x[i] = 0.0f;
        mask[i] = m;
    }
    if (nzeroed) {
        // same-address global atomics retire one per ~12 ns: one per workgroup, few workgroups
        __shared__ unsigned wsum[4];
        for (int off = 32; off > 0; off >>= 1) cnt += __shfl_down(cnt, off);
        if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = cnt;
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned tot = wsum[0] + wsum[1] + wsum[2] + wsum[3];
            if (tot) atomicAdd(nzeroed, (unsigned long long)tot);
        }
    }
}


extern "C" int nnc_threshold_mask_f32(float *x, int64_t n, const float *thr_dev, uint8_t *mask,
                                      int64_t *nzeroed_dev, void *stream)
{
    if (n < 0 || !thr_dev || (n > 0 && (!x || !mask))) return fail(NNC_EINVAL, "nnc_threshold_mask_f32: bad argument");
    if (nzeroed_dev) HIPCHK(hipMemsetAsync(nzeroed_dev, 0, sizeof(int64_t), S(stream)));
    if (n == 0) return NNC_OK;
    const bool vec = ((reinterpret_cast<uintptr_t>(x) & 15) == 0) && ((reinterpret_cast<uintptr_t>(mask) & 3) == 0);
    int grid = stream_grid((n + 3) / 4, 256, 4);
    if (vec) NNC_LAUNCH_PROF(NNC_PROF_THRESHOLD, (k_threshold<true>), dim3(grid), dim3(256), 0, S(stream), x, n, thr_dev, mask, reinterpret_cast<unsigned long long *>(nzeroed_dev));
    else NNC_LAUNCH_PROF(NNC_PROF_THRESHOLD, (k_threshold<false>), dim3(grid), dim3(256), 0, S(stream), x, n, thr_dev, mask, reinterpret_cast<unsigned long long *>(nzeroed_dev));
    LAUNCHCHK("k_threshold");
    return NNC_OK;
}

extern "C" size_t nnc_prune_workspace_bytes(int64_t n)
{
    int64_t nchunks = (n + NNC_CHUNK - 1) / NNC_CHUNK;
    return (size_t)(nchunks + 64) * sizeof(float);
}

__global__ void k_set_thr(float q, float *stats)
{
    stats[0] = 0.0f;
    stats[1] = q;
}
__global__ void k_set_f32(float v, float *dst) { *dst = v; }

extern "C" int nnc_prune_f32(float *x, int64_t n, float q, int std_smooth, uint8_t *mask, float *stats_dev,
                             int64_t *nzeroed_dev, void *ws, size_t ws_bytes, void *stream)
{
    if (n < 0 || !stats_dev || (n > 0 && (!x || !mask))) return fail(NNC_EINVAL, "nnc_prune_f32: bad argument");
    if (std_smooth) {
        if (!ws || ws_bytes < nnc_prune_workspace_bytes(n)) return fail(NNC_ENOSPACE, "nnc_prune_f32: workspace too small");
        if (n == 0) return fail(NNC_EINVAL, "nnc_prune_f32: std of an empty tensor");
        float *wsf = reinterpret_cast<float *>(ws);
        float *scal = wsf;          // [0] mean, [1] q
        float *chunks = wsf + 16;
        const int64_t nchunks = (n + NNC_CHUNK - 1) / NNC_CHUNK;
        int rc;
        hipLaunchKernelGGL(k_set_f32, dim3(1), dim3(1), 0, S(stream), q, scal + 1);
        LAUNCHCHK("k_set_f32");
        if ((rc = nnc_chunk_sums_f32(x, n, 0, nullptr, chunks, stream))) return rc;
        if ((rc = nnc_fold_f32(chunks, nchunks, n, NNC_FOLD_MEAN, nullptr, scal, stream))) return rc;
        if ((rc = nnc_chunk_sums_f32(x, n, 1, scal, chunks, stream))) return rc;
        // stats = {sigma, sigma * q}
        if ((rc = nnc_fold_f32(chunks, nchunks, n, NNC_FOLD_STD, scal + 1, stats_dev, stream))) return rc;
    } else {
        hipLaunchKernelGGL(k_set_thr, dim3(1), dim3(1), 0, S(stream), q, stats_dev);
        LAUNCHCHK("k_set_thr");
    }
    return nnc_threshold_mask_f32(x, n, stats_dev + 1, mask, nzeroed_dev, stream);
}

template <bool VEC>
__global__ __launch_bounds__(256) void k_apply_mask(float *__restrict__ x, const uint8_t *__restrict__ mask, int64_t n)
{
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t nthreads = (int64_t)gridDim.x * blockDim.x;
    int64_t done = 0;
    if (VEC) {
        const int64_t nvec = n >> 2;
        float4 *x4 = reinterpret_cast<float4 *>(x);
        const uchar4 *m4 = reinterpret_cast<const uchar4 *>(mask);
        for (int64_t v = tid; v < nvec; v += nthreads) {
            uchar4 m = m4[v];
            if (m.x | m.y | m.z | m.w) {
                float4 a = x4[v];
                a.x = m.x ? 0.0f : a.x; a.y = m.y ? 0.0f : a.y; a.z = m.z ? 0.0f : a.z; a.w = m.w ? 0.0f : a.w;
                x4[v] = a;
            }
        }
        done = nvec << 2;
    }
    for (int64_t i = done + tid; i < n; i += nthreads)
        if (mask[i]) x[i] = 0.0f;
}

extern "C" int nnc_apply_mask_f32(float *x, const uint8_t *mask, int64_t n, void *stream)
{
    if (n < 0 || (n > 0 && (!x || !mask))) return fail(NNC_EINVAL, "nnc_apply_mask_f32: bad argument");
    if (n == 0) return NNC_OK;
    const bool vec = ((reinterpret_cast<uintptr_t>(x) & 15) == 0) && ((reinterpret_cast<uintptr_t>(mask) & 3) == 0);
    int grid = stream_grid((n + 3) / 4, 256, 16);
    if (vec) hipLaunchKernelGGL((k_apply_mask<true>), dim3(grid), dim3(256), 0, S(stream), x, mask, n);
    else hipLaunchKernelGGL((k_apply_mask<false>), dim3(grid), dim3(256), 0, S(stream), x, mask, n);
    LAUNCHCHK("k_apply_mask");
    return NNC_OK;
}

// ======================================================================================
// 3. min / max / count, 31-bin histogram, bincount
// ======================================================================================
struct MinMaxPartial { float mn, mx; unsigned long long cnt; unsigned long long neg, zer; float mn_nz, mx_nz; };

template <bool VEC>
__global__ __launch_bounds__(256) void k_minmax(const float *__restrict__ x, int64_t n, int skip_zeros,
                                                MinMaxPartial *__restrict__ part)
{
    float mn = INFINITY, mx = -INFINITY, mn_nz = INFINITY, mx_nz = -INFINITY;
    unsigned long long cnt = 0;
    unsigned neg = 0, zer = 0; // per thread: well below 2^32
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t nthreads = (int64_t)gridDim.x * blockDim.x;
    int64_t done = 0;
#define MM1(v) do { float v_ = (v); neg += (v_ < 0.0f); const bool z_ = (v_ == 0.0f); zer += z_; if (!z_) { mn_nz = fminf(mn_nz, v_); mx_nz = fmaxf(mx_nz, v_); } \
        bool use_ = !(skip_zeros && z_); if (use_) { mn = fminf(mn, v_); mx = fmaxf(mx, v_); cnt++; } } while (0)
    if (VEC) {
        const int64_t nvec = n >> 2;
        const float4 *x4 = reinterpret_cast<const float4 *>(x);
        for (int64_t v = tid; v < nvec; v += nthreads) {
            float4 a = x4[v];
            MM1(a.x); MM1(a.y); MM1(a.z); MM1(a.w);
        }
        done = nvec << 2;
    }
    for (int64_t i = done + tid; i < n; i += nthreads) MM1(x[i]);
#undef MM1
    unsigned long long negl = neg, zerl = zer;
    for (int off = 32; off > 0; off >>= 1) {
        mn = fminf(mn, __shfl_down(mn, off));
        mx = fmaxf(mx, __shfl_down(mx, off));
        mn_nz = fminf(mn_nz, __shfl_down(mn_nz, off));
        mx_nz = fmaxf(mx_nz, __shfl_down(mx_nz, off));
        cnt += __shfl_down(cnt, off);
        negl += __shfl_down(negl, off);
        zerl += __shfl_down(zerl, off);
    }
    __shared__ MinMaxPartial sh[4];
    if ((threadIdx.x & 63) == 0) { MinMaxPartial q; q.mn = mn; q.mx = mx; q.cnt = cnt; q.neg = negl; q.zer = zerl; q.mn_nz = mn_nz; q.mx_nz = mx_nz; sh[threadIdx.x >> 6] = q; }
    __syncthreads();
    if (threadIdx.x == 0) {
        MinMaxPartial p = sh[0];
        for (int w = 1; w < 4; w++) {
            p.mn = fminf(p.mn, sh[w].mn); p.mx = fmaxf(p.mx, sh[w].mx); p.cnt += sh[w].cnt; p.neg += sh[w].neg; p.zer += sh[w].zer;
            p.mn_nz = fminf(p.mn_nz, sh[w].mn_nz); p.mx_nz = fmaxf(p.mx_nz, sh[w].mx_nz);
        }
        part[blockIdx.x] = p;
    }
}

__global__ __launch_bounds__(256) void k_minmax_final(const MinMaxPartial *__restrict__ part, int nparts,
                                                      float *__restrict__ out, long long *__restrict__ count,
                                                      long long *__restrict__ signs)
{
    float mn = INFINITY, mx = -INFINITY, mn_nz = INFINITY, mx_nz = -INFINITY;
    unsigned long long cnt = 0, neg = 0, zer = 0;
    for (int i = threadIdx.x; i < nparts; i += 256) {
        mn = fminf(mn, part[i].mn); mx = fmaxf(mx, part[i].mx); cnt += part[i].cnt; neg += part[i].neg; zer += part[i].zer;
        mn_nz = fminf(mn_nz, part[i].mn_nz); mx_nz = fmaxf(mx_nz, part[i].mx_nz);
    }
    for (int off = 32; off > 0; off >>= 1) {
        mn = fminf(mn, __shfl_down(mn, off));
        mx = fmaxf(mx, __shfl_down(mx, off));
        mn_nz = fminf(mn_nz, __shfl_down(mn_nz, off));
        mx_nz = fmaxf(mx_nz, __shfl_down(mx_nz, off));
        cnt += __shfl_down(cnt, off);
        neg += __shfl_down(neg, off);
        zer += __shfl_down(zer, off);
    }
    __shared__ MinMaxPartial sh[4];
    if ((threadIdx.x & 63) == 0) { MinMaxPartial q; q.mn = mn; q.mx = mx; q.cnt = cnt; q.neg = neg; q.zer = zer; q.mn_nz = mn_nz; q.mx_nz = mx_nz; sh[threadIdx.x >> 6] = q; }
    __syncthreads();
    if (threadIdx.x == 0) {
        MinMaxPartial p = sh[0];
        for (int w = 1; w < 4; w++) {
            p.mn = fminf(p.mn, sh[w].mn); p.mx = fmaxf(p.mx, sh[w].mx); p.cnt += sh[w].cnt; p.neg += sh[w].neg; p.zer += sh[w].zer;
            p.mn_nz = fminf(p.mn_nz, sh[w].mn_nz); p.mx_nz = fmaxf(p.mx_nz, sh[w].mx_nz);
        }
        out[0] = p.mn; out[1] = p.mx;
        if (count) *count = (long long)p.cnt;
        if (signs) { signs[0] = (long long)p.neg; signs[1] = (long long)p.zer; out[2] = p.mn_nz; out[3] = p.mx_nz; }
    }
}

static int minmax_grid(int64_t n) { return stream_grid((n + 3) / 4, 256, 8); }

extern "C" size_t nnc_minmax_workspace_bytes(int64_t n)
{
    (void)n;
    return (size_t)(cu_count() * 8 + 8) * sizeof(MinMaxPartial);
}

static int minmax_impl(const float *x, int64_t n, int skip_zeros, float *out_dev, int64_t *count_dev, int64_t *signs_dev,
                       void *ws, size_t ws_bytes, void *stream)
{
    if (n <= 0 || !x || !out_dev || !ws) return fail(NNC_EINVAL, "nnc_minmax_f32: bad argument (n must be > 0)");
    if (ws_bytes < nnc_minmax_workspace_bytes(n)) return fail(NNC_ENOSPACE, "nnc_minmax_f32: workspace too small");
    int grid = minmax_grid(n);
    MinMaxPartial *part = reinterpret_cast<MinMaxPartial *>(ws);
    const bool vec = (reinterpret_cast<uintptr_t>(x) & 15) == 0;
    if (vec) NNC_LAUNCH_PROF(NNC_PROF_MINMAX, (k_minmax<true>), dim3(grid), dim3(256), 0, S(stream), x, n, skip_zeros, part);
    else NNC_LAUNCH_PROF(NNC_PROF_MINMAX, (k_minmax<false>), dim3(grid), dim3(256), 0, S(stream), x, n, skip_zeros, part);
    LAUNCHCHK("k_minmax");
    hipLaunchKernelGGL(k_minmax_final, dim3(1), dim3(256), 0, S(stream), part, grid, out_dev, reinterpret_cast<long long *>(count_dev),
                       reinterpret_cast<long long *>(signs_dev));
    LAUNCHCHK("k_minmax_final");
    return NNC_OK;
}

extern "C" int nnc_minmax_f32(const float *x, int64_t n, int skip_zeros, float *out_dev, int64_t *count_dev,
                              void *ws, size_t ws_bytes, void *stream)
{
    return minmax_impl(x, n, skip_zeros, out_dev, count_dev, nullptr, ws, ws_bytes, stream);
}

extern "C" int nnc_minmax_signs_f32(const float *x, int64_t n, float *out_dev, int64_t *signs_dev, void *ws, size_t ws_bytes,
                                    void *stream)
{
    if (!signs_dev) return fail(NNC_EINVAL, "nnc_minmax_signs_f32: null signs_dev");
    return minmax_impl(x, n, 0, out_dev, nullptr, signs_dev, ws, ws_bytes, stream);
}

// The threshold pass AND the min / max / sign statistics of what it leaves behind, in one pass over the vector: the pruned
// values are in registers anyway, and the statistics pass that the sort and the k-means set-up need next (nnc_minmax_signs_f32)
// would read them again.  Same mask, same zeroing, same count as k_threshold; same partials as k_minmax.
template <bool VEC>
__global__ __launch_bounds__(256) void k_threshold_stats(float *__restrict__ x, int64_t n, const float *__restrict__ thr_dev,
                                                         uint8_t *__restrict__ mask, unsigned long long *__restrict__ nzeroed,
                                                         MinMaxPartial *__restrict__ part)
{
    const float thr = *thr_dev;
    unsigned cnt = 0, neg = 0, zer = 0;
    float mn = INFINITY, mx = -INFINITY, mn_nz = INFINITY, mx_nz = -INFINITY;
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t nthreads = (int64_t)gridDim.x * blockDim.x;
    int64_t done = 0;
#define TS1(v) do { const float v_ = (v); neg += (v_ < 0.0f); const bool z_ = (v_ == 0.0f); zer += z_; \
        if (!z_) { mn_nz = fminf(mn_nz, v_); mx_nz = fmaxf(mx_nz, v_); } mn = fminf(mn, v_); mx = fmaxf(mx, v_); } while (0)
    if (VEC) {
        const int64_t nvec = n >> 2;
        float4 *x4 = reinterpret_cast<float4 *>(x);
        uchar4 *m4 = reinterpret_cast<uchar4 *>(mask);
        for (int64_t v = tid; v < nvec; v += nthreads) {
            float4 a = x4[v];
            uchar4 m;
            m.x = fabsf(a.x) < thr; m.y = fabsf(a.y) < thr; m.z = fabsf(a.z) < thr; m.w = fabsf(a.w) < thr;
            cnt += m.x + m.y + m.z + m.w;
            a.x = m.x ? 0.0f : a.x; a.y = m.y ? 0.0f : a.y; a.z = m.z ? 0.0f : a.z; a.w = m.w ? 0.0f : a.w;
            x4[v] = a;
            m4[v] = m;
            TS1(a.x); TS1(a.y); TS1(a.z); TS1(a.w);
        }
        done = nvec << 2;
    }
    for (int64_t i = done + tid; i < n; i += nthreads) {
        float a = x[i];
        const uint8_t m = fabsf(a) < thr;
        cnt += m;
        if (m) { a = 0.0f; x[i] = 0.0f; }
        mask[i] = m;
        TS1(a);
    }
#undef TS1
    unsigned long long negl = neg, zerl = zer, cntl = cnt;
    for (int off = 32; off > 0; off >>= 1) {
        mn = fminf(mn, __shfl_down(mn, off));
        mx = fmaxf(mx, __shfl_down(mx, off));
        mn_nz = fminf(mn_nz, __shfl_down(mn_nz, off));
        mx_nz = fmaxf(mx_nz, __shfl_down(mx_nz, off));
        negl += __shfl_down(negl, off);
        zerl += __shfl_down(zerl, off);
        cntl += __shfl_down(cntl, off);
    }
    __shared__ MinMaxPartial sh[4];
    __shared__ unsigned long long shc[4];
    if ((threadIdx.x & 63) == 0) {
        MinMaxPartial q; q.mn = mn; q.mx = mx; q.cnt = 0; q.neg = negl; q.zer = zerl; q.mn_nz = mn_nz; q.mx_nz = mx_nz;
        sh[threadIdx.x >> 6] = q; shc[threadIdx.x >> 6] = cntl;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        MinMaxPartial p = sh[0];
        unsigned long long tot = shc[0];
        for (int w = 1; w < 4; w++) {
            p.mn = fminf(p.mn, sh[w].mn); p.mx = fmaxf(p.mx, sh[w].mx); p.neg += sh[w].neg; p.zer += sh[w].zer;
            p.mn_nz = fminf(p.mn_nz, sh[w].mn_nz); p.mx_nz = fmaxf(p.mx_nz, sh[w].mx_nz);
            tot += shc[w];
        }
        part[blockIdx.x] = p;
        if (nzeroed && tot) atomicAdd(nzeroed, tot);
    }
}

extern "C" size_t nnc_prune_stats_workspace_bytes(int64_t n) { return ((nnc_prune_workspace_bytes(n) + 255) & ~(size_t)255) + nnc_minmax_workspace_bytes(n); }

extern "C" int nnc_prune_stats_f32(float *x, int64_t n, float q, int std_smooth, uint8_t *mask, float *stats_dev, int64_t *nzeroed_dev,
                                   float *minmax4_dev, int64_t *signs_dev, void *ws, size_t ws_bytes, void *stream)
{
    if (n <= 0 || !x || !mask || !stats_dev || !minmax4_dev || !signs_dev || !ws) return fail(NNC_EINVAL, "nnc_prune_stats_f32: bad argument (n must be > 0)");
    if (ws_bytes < nnc_prune_stats_workspace_bytes(n)) return fail(NNC_ENOSPACE, "nnc_prune_stats_f32: workspace too small");
    if (std_smooth) {
        float *wsf = reinterpret_cast<float *>(ws);
        float *scal = wsf;          // [0] mean, [1] q
        float *chunks = wsf + 16;
        const int64_t nchunks = (n + NNC_CHUNK - 1) / NNC_CHUNK;
        int rc;
        if ((rc = nnc_chunk_sums_f32(x, n, 0, nullptr, chunks, stream))) return rc;
        if ((rc = nnc_fold_f32(chunks, nchunks, n, NNC_FOLD_MEAN, nullptr, scal, stream))) return rc;
        if ((rc = nnc_chunk_sums_f32(x, n, 1, scal, chunks, stream))) return rc;
        // stats = {sigma, sigma * q}; the threshold pass's counter is zeroed on the way
        hipLaunchKernelGGL(k_fold, dim3(1), dim3(1024), 0, S(stream), chunks, nchunks, n, (int)NNC_FOLD_STD, (const float *)nullptr, stats_dev, q, 1,
                           reinterpret_cast<unsigned long long *>(nzeroed_dev));
        LAUNCHCHK("k_fold");
    } else {
        hipLaunchKernelGGL(k_set_thr, dim3(1), dim3(1), 0, S(stream), q, stats_dev);
        LAUNCHCHK("k_set_thr");
        if (nzeroed_dev) HIPCHK(hipMemsetAsync(nzeroed_dev, 0, sizeof(int64_t), S(stream)));
    }
    MinMaxPartial *part = reinterpret_cast<MinMaxPartial *>(reinterpret_cast<unsigned char *>(ws) + ((nnc_prune_workspace_bytes(n) + 255) & ~(size_t)255));
    const bool vec = ((reinterpret_cast<uintptr_t>(x) & 15) == 0) && ((reinterpret_cast<uintptr_t>(mask) & 3) == 0);
    const int grid = std::min(stream_grid((n + 3) / 4, 256, 4), cu_count() * 8);
    if (vec) NNC_LAUNCH_PROF(NNC_PROF_THRESHOLD, (k_threshold_stats<true>), dim3(grid), dim3(256), 0, S(stream), x, n, stats_dev + 1, mask, reinterpret_cast<unsigned long long *>(nzeroed_dev), part);
    else NNC_LAUNCH_PROF(NNC_PROF_THRESHOLD, (k_threshold_stats<false>), dim3(grid), dim3(256), 0, S(stream), x, n, stats_dev + 1, mask, reinterpret_cast<unsigned long long *>(nzeroed_dev), part);
    LAUNCHCHK("k_threshold_stats");
    hipLaunchKernelGGL(k_minmax_final, dim3(1), dim3(256), 0, S(stream), part, grid, minmax4_dev, (long long *)nullptr, reinterpret_cast<long long *>(signs_dev));
    LAUNCHCHK("k_minmax_final");
    return NNC_OK;
}

// Everything LayerStats wants of one (whole, single-GPU) vector, enqueued by one call: NumPy-exact mean and variance
// (the two chunk-sum passes and their folds) and the min / max / sign pass.
extern "C" size_t nnc_layer_stats_workspace_bytes(int64_t n)
{
    const size_t nch = (size_t)((n + NNC_CHUNK - 1) / NNC_CHUNK);
    return ((2 * nch * sizeof(float) + 255) & ~(size_t)255) + nnc_minmax_workspace_bytes(n);
}

extern "C" int nnc_layer_stats_f32(const float *x, int64_t n, float *out6_dev, int64_t *signs_dev, void *ws, size_t ws_bytes,
                                   void *stream)
{
    if (n <= 0 || !x || !out6_dev || !signs_dev || !ws) return fail(NNC_EINVAL, "nnc_layer_stats_f32: bad argument (n must be > 0)");
    if (ws_bytes < nnc_layer_stats_workspace_bytes(n)) return fail(NNC_ENOSPACE, "nnc_layer_stats_f32: workspace too small");
    const int64_t nch = (n + NNC_CHUNK - 1) / NNC_CHUNK;
    float *c1 = reinterpret_cast<float *>(ws), *c2 = c1 + nch;
    unsigned char *mm_ws = reinterpret_cast<unsigned char *>(ws) + ((2 * (size_t)nch * sizeof(float) + 255) & ~(size_t)255);
    int rc;
    if ((rc = nnc_chunk_sums_f32(x, n, 0, nullptr, c1, stream))) return rc;
    if ((rc = nnc_fold_f32(c1, nch, n, NNC_FOLD_MEAN, nullptr, out6_dev, stream))) return rc;           // [0] = mean
    if ((rc = nnc_chunk_sums_f32(x, n, 1, out6_dev, c2, stream))) return rc;
    if ((rc = nnc_fold_f32(c2, nch, n, NNC_FOLD_MEAN, nullptr, out6_dev + 1, stream))) return rc;       // [1] = variance
    return minmax_impl(x, n, 0, out6_dev + 2, nullptr, signs_dev, mm_ws, nnc_minmax_workspace_bytes(n), stream); // [2..5]
}

// bin(x) = #{ steps[i] <= x } - 1 for non-decreasing steps (np.linspace is monotone), which is
// exactly "steps[b] <= x < steps[b+1]"; x >= steps[31] (the maximum itself) falls in no bin.
template <bool VEC>
__global__ __launch_bounds__(256) void k_hist31(const float *__restrict__ x, int64_t n, int skip_zeros,
                                                const float *__restrict__ steps,
                                                unsigned long long *__restrict__ counts)
{
    __shared__ unsigned h[32][32]; // [bin][replica]; replica = lane & 31 -> bank = replica
    __shared__ float st[32];
    for (int i = threadIdx.x; i < 32 * 32; i += 256) (&h[0][0])[i] = 0;
    if (threadIdx.x < 32) st[threadIdx.x] = steps[threadIdx.x];
    __syncthreads();
    float s[32];
#pragma unroll
    for (int i = 0; i < 32; i++) s[i] = st[i];
    const int rep = threadIdx.x & 31;
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t nthreads = (int64_t)gridDim.x * blockDim.x;
#define H1(v) do { float v_ = (v); if (!(skip_zeros && v_ == 0.0f)) { int c_ = 0; _Pragma("unroll") for (int i = 0; i < 32; i++) c_ += (v_ >= s[i]); if (c_ >= 1 && c_ <= 31) atomicAdd(&h[c_ - 1][rep], 1u); } } while (0)
    int64_t done = 0;
    if (VEC) {
        const int64_t nvec = n >> 2;
        const float4 *x4 = reinterpret_cast<const float4 *>(x);
        for (int64_t v = tid; v < nvec; v += nthreads) {
            float4 a = x4[v];
            H1(a.x); H1(a.y); H1(a.z); H1(a.w);
        }
        done = nvec << 2;
    }
    for (int64_t i = done + tid; i < n; i += nthreads) H1(x[i]);
#undef H1
    __syncthreads();
    if (threadIdx.x < 31) {
        unsigned long long t = 0;
        for (int r = 0; r < 32; r++) t += h[threadIdx.x][r];
        if (t) atomicAdd(&counts[threadIdx.x], t);
    }
}

extern "C" int nnc_hist31_f32(const float *x, int64_t n, int skip_zeros, const float *steps32_dev,
                              int64_t *counts_dev, void *stream)
{
    if (n < 0 || !steps32_dev || !counts_dev || (n > 0 && !x)) return fail(NNC_EINVAL, "nnc_hist31_f32: bad argument");
    if (n == 0) return NNC_OK;
    int grid = stream_grid((n + 3) / 4, 256, 2);
    const bool vec = (reinterpret_cast<uintptr_t>(x) & 15) == 0;
    if (vec) hipLaunchKernelGGL((k_hist31<true>), dim3(grid), dim3(256), 0, S(stream), x, n, skip_zeros, steps32_dev, reinterpret_cast<unsigned long long *>(counts_dev));
    else hipLaunchKernelGGL((k_hist31<false>), dim3(grid), dim3(256), 0, S(stream), x, n, skip_zeros, steps32_dev, reinterpret_cast<unsigned long long *>(counts_dev));
    LAUNCHCHK("k_hist31");
    return NNC_OK;
}

// ranks_out[i] = #{ j : xs[j] < values[i] } in an ascending vector (lower bound; the same float32 comparison the histogram
// kernel makes): the 31 bin counts of get_weight_distribution (utility.py:366-372) are differences of 32 such ranks.
__global__ __launch_bounds__(64) void k_rank_sorted(const float *__restrict__ xs, long long n, const float *__restrict__ values, int m,
                                                    long long *__restrict__ ranks_out)
{
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= m) return;
    const float v = values[i];
    long long lo = 0, hi = n;
    while (lo < hi) { const long long mid = (lo + hi) >> 1; if (xs[mid] < v) lo = mid + 1; else hi = mid; }
    ranks_out[i] = lo;
}

extern "C" int nnc_rank_sorted_f32(const float *x_sorted, int64_t n, const float *values_dev, int32_t m, int64_t *ranks_out_dev, void *stream)
{
    if (n < 0 || m < 0 || (m > 0 && (!values_dev || !ranks_out_dev)) || (n > 0 && !x_sorted)) return fail(NNC_EINVAL, "nnc_rank_sorted_f32: bad argument");
    if (m == 0) return NNC_OK;
    hipLaunchKernelGGL(k_rank_sorted, dim3((m + 63) / 64), dim3(64), 0, S(stream), x_sorted, (long long)n, values_dev, (int)m, reinterpret_cast<long long *>(ranks_out_dev));
    LAUNCHCHK("k_rank_sorted");
    return NNC_OK;
}

template <typename LT>
__global__ __launch_bounds__(256) void k_bincount(const LT *__restrict__ labels, int64_t n, int k,
                                                  unsigned long long *__restrict__ counts)
{
    extern __shared__ unsigned hb[]; // [k][8] replicas
    for (int i = threadIdx.x; i < k * 8; i += 256) hb[i] = 0;
    __syncthreads();
    const int rep = threadIdx.x & 7;
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t nthreads = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = tid; i < n; i += nthreads) {
        int l = labels[i];
        if (l < k) atomicAdd(&hb[l * 8 + rep], 1u);
    }
    __syncthreads();
    for (int j = threadIdx.x; j < k; j += 256) {
        unsigned long long t = 0;
        for (int r = 0; r < 8; r++) t += hb[j * 8 + r];
        if (t) atomicAdd(&counts[j], t);
    }
}

extern "C" int nnc_bincount(const void *labels, int label_bytes, int64_t n, int32_t k, int64_t *counts_dev, void *stream)
{
    if (n < 0 || k <= 0 || k > NNC_KMAX || !counts_dev || (n > 0 && !labels)) return fail(NNC_EINVAL, "nnc_bincount: bad argument");
    if (label_bytes != 1 && label_bytes != 2) return fail(NNC_EINVAL, "nnc_bincount: label_bytes must be 1 or 2");
    if (n == 0) return NNC_OK;
    int grid = stream_grid(n, 256 * 8, 2);
    size_t lds = (size_t)k * 8 * sizeof(unsigned);
    if (label_bytes == 1) hipLaunchKernelGGL((k_bincount<uint8_t>), dim3(grid), dim3(256), lds, S(stream), reinterpret_cast<const uint8_t *>(labels), n, k, reinterpret_cast<unsigned long long *>(counts_dev));
    else hipLaunchKernelGGL((k_bincount<uint16_t>), dim3(grid), dim3(256), lds, S(stream), reinterpret_cast<const uint16_t *>(labels), n, k, reinterpret_cast<unsigned long long *>(counts_dev));
    LAUNCHCHK("k_bincount");
    return NNC_OK;
}

