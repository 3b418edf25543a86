// nnc_sort.hip -- one-time reordering of a weight vector by value (ascending) for the Lloyd
// iterations.  Per-cluster sums do not depend on the order of the weights (exact integer
// sums), so the iterations may stream a sorted copy: neighbouring weights then fall in the
// same cluster and each lane accumulates runs in registers instead of issuing one LDS atomic
// per weight.  The sort itself is a plain library call (rocPRIM device radix sort), outside
// the per-iteration path; labels / quantized values are produced from the ORIGINAL vector.
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

#include <algorithm>
#include <string>

#include "nnc.h"

extern "C" const char *nnc_last_error(void);
int nnc_set_error_(int code, const char *msg); // in nnc_hip.hip

extern "C" size_t nnc_sort_workspace_bytes(int64_t n)
{
    if (n <= 0) return 0;
    size_t bytes = 0;
    hipError_t e = rocprim::radix_sort_keys(nullptr, bytes, (const float *)nullptr, (float *)nullptr, (size_t)n);
    if (e != hipSuccess) return 0;
    return bytes + 256;
}

extern "C" int nnc_sort_f32(const float *x, int64_t n, float *sorted_out, void *ws, size_t ws_bytes, void *stream)
{
    if (n < 0 || (n > 0 && (!x || !sorted_out))) return nnc_set_error_(NNC_EINVAL, "nnc_sort_f32: bad argument");
    if (n == 0) return NNC_OK;
    size_t need = 0;
    hipError_t e = rocprim::radix_sort_keys(nullptr, need, x, sorted_out, (size_t)n);
    if (e != hipSuccess) return nnc_set_error_(NNC_EHIP, hipGetErrorString(e));
    if (!ws || ws_bytes < need) return nnc_set_error_(NNC_ENOSPACE, "nnc_sort_f32: workspace too small");
    e = rocprim::radix_sort_keys(ws, need, x, sorted_out, (size_t)n, 0, 32, reinterpret_cast<hipStream_t>(stream));
    if (e != hipSuccess) return nnc_set_error_(NNC_EHIP, hipGetErrorString(e));
    return NNC_OK;
}

// --------------------------------------------------------------------------------------
// The same for a PRUNED vector: most weights are exact zeros, which need no sorting.  One
// three-way split (negatives / positives / zeros dropped), one radix sort of the non-zeros, the
// two halves copied to their places with the zeros filled in between.  The counts come from
// nnc_minmax_signs_f32 (the fit set-up reads them together with min / max).  -0.0 counts as a
// zero and comes back as +0.0: equal as a value, which is all the iterations look at.
// --------------------------------------------------------------------------------------
// negatives -> t_neg[], positives (and NaN) -> t_pos[], zeros dropped; any order within a part.
// Tiles of 16384 weights; one packed 64-bit atomic per tile hands out both output ranges.
#define SPLIT_THREADS 1024
__global__ __launch_bounds__(SPLIT_THREADS) void k_split_signs(const float *__restrict__ x, long long n, float *__restrict__ t_neg,
                                                               float *__restrict__ t_pos, unsigned long long *__restrict__ counter,
                                                               long long cap_neg, long long cap_pos)
{
    __shared__ unsigned wave_tot[SPLIT_THREADS / 64];
    __shared__ unsigned long long base_s;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const bool vec = (reinterpret_cast<uintptr_t>(x) & 15) == 0;
    const long long nvec = vec ? (n >> 2) : 0;
    const float4 *x4 = reinterpret_cast<const float4 *>(x);
    const long long ntiles = (nvec + 4 * SPLIT_THREADS - 1) / (4 * SPLIT_THREADS);
    for (long long t = blockIdx.x; t < ntiles; t += gridDim.x) {
        float4 q[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const long long v = t * (4 * SPLIT_THREADS) + (long long)j * SPLIT_THREADS + tid;
            q[j] = v < nvec ? x4[v] : make_float4(0.f, 0.f, 0.f, 0.f); // padding zeros are dropped like any zero
        }
        unsigned c = 0; // negatives | positives << 16
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const float e[4] = {q[j].x, q[j].y, q[j].z, q[j].w};
#pragma unroll
            for (int i = 0; i < 4; i++) c += (e[i] < 0.0f) ? 1u : ((e[i] == 0.0f) ? 0u : 0x10000u);
        }
        unsigned sc = c;
        for (int off = 1; off < 64; off <<= 1) { const unsigned o = __shfl_up(sc, off); if (lane >= off) sc += o; }
        if (lane == 63) wave_tot[wv] = sc;
        __syncthreads();
        unsigned pre = 0, tot = 0;
        for (int w = 0; w < SPLIT_THREADS / 64; w++) { const unsigned o = wave_tot[w]; if (w < wv) pre += o; tot += o; }
        if (tid == 0) base_s = atomicAdd(counter, (unsigned long long)(tot & 0xFFFFu) | ((unsigned long long)(tot >> 16) << 32));
        __syncthreads();
        const unsigned long long base = base_s;
        const unsigned excl = pre + sc - c;
        long long on = (long long)(base & 0xFFFFFFFFull) + (excl & 0xFFFFu);
        long long op = (long long)(base >> 32) + (excl >> 16);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const float e[4] = {q[j].x, q[j].y, q[j].z, q[j].w};
#pragma unroll
            for (int i = 0; i < 4; i++) {
                // (bounds: the counts are the caller's; wrong ones must not fault)
                if (e[i] < 0.0f) { if (on < cap_neg) t_neg[on] = e[i]; on++; }
                else if (!(e[i] == 0.0f)) { if (op < cap_pos) t_pos[op] = e[i]; op++; }
            }
        }
        __syncthreads(); // base_s / wave_tot are reused by the next tile
    }
    // the scalars after the last float4 (the whole vector if it is not 16-byte aligned): first workgroup, one by one
    if (blockIdx.x == 0) {
        for (long long i0 = nvec << 2; i0 < n; i0 += SPLIT_THREADS) {
            const long long i = i0 + tid;
            const float e = i < n ? x[i] : 0.0f;
            const bool ng = e < 0.0f, ps = !(e < 0.0f) && !(e == 0.0f);
            const unsigned long long bn = __ballot(ng), bp = __ballot(ps);
            unsigned long long base = 0;
            if (lane == 0 && (bn | bp)) base = atomicAdd(counter, (unsigned long long)__popcll(bn) | ((unsigned long long)__popcll(bp) << 32));
            base = __shfl(base, 0);
            const unsigned long long below = (1ull << lane) - 1ull;
            const long long in_ = (long long)(base & 0xFFFFFFFFull) + __popcll(bn & below), ip_ = (long long)(base >> 32) + __popcll(bp & below);
            if (ng && in_ < cap_neg) t_neg[in_] = e;
            if (ps && ip_ < cap_pos) t_pos[ip_] = e;
        }
    }
}

static size_t al256(size_t b) { return (b + 255) & ~(size_t)255; }

static hipError_t pruned_sizes(int64_t n_nz, size_t *sort_bytes)
{
    size_t sa = 0;
    hipError_t e;
    if (n_nz > 0 && (e = rocprim::radix_sort_keys(nullptr, sa, (const float *)nullptr, (float *)nullptr, (size_t)n_nz)) != hipSuccess) return e;
    *sort_bytes = sa;
    return hipSuccess;
}

extern "C" size_t nnc_sort_pruned_workspace_bytes(int64_t n, int64_t n_neg, int64_t n_zero)
{
    if (n <= 0 || n_neg < 0 || n_zero < 0 || n_neg + n_zero > n) return 0;
    const int64_t n_nz = n - n_zero;
    size_t sb = 0;
    if (pruned_sizes(n_nz, &sb) != hipSuccess) return 0;
    return 2 * al256((size_t)n_nz * 4 + 16) + al256(sizeof(unsigned long long)) + al256(sb) + 256;
}

extern "C" int nnc_sort_pruned_f32(const float *x, int64_t n, int64_t n_neg, int64_t n_zero, float *sorted_out, void *ws,
                                   size_t ws_bytes, void *stream)
{
    if (n < 0 || n_neg < 0 || n_zero < 0 || n_neg + n_zero > n || n >= ((int64_t)1 << 32) || (n > 0 && (!x || !sorted_out)))
        return nnc_set_error_(NNC_EINVAL, "nnc_sort_pruned_f32: bad argument");
    if (n == 0) return NNC_OK;
    const int64_t n_pos = n - n_neg - n_zero, n_nz = n_neg + n_pos;
    size_t sb = 0;
    hipError_t e = pruned_sizes(n_nz, &sb);
    if (e != hipSuccess) return nnc_set_error_(NNC_EHIP, hipGetErrorString(e));
    if (!ws || ws_bytes < nnc_sort_pruned_workspace_bytes(n, n_neg, n_zero)) return nnc_set_error_(NNC_ENOSPACE, "nnc_sort_pruned_f32: workspace too small");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    unsigned char *b = reinterpret_cast<unsigned char *>((reinterpret_cast<uintptr_t>(ws) + 255) & ~(uintptr_t)255);
    float *t_all = reinterpret_cast<float *>(b);    b += al256((size_t)n_nz * 4 + 16); // the non-zeros: negatives, then positives, each in any order
    float *t_sorted = reinterpret_cast<float *>(b); b += al256((size_t)n_nz * 4 + 16);
    unsigned long long *counter = reinterpret_cast<unsigned long long *>(b); b += al256(sizeof(unsigned long long));
    void *stemp = b;
    e = hipMemsetAsync(counter, 0, sizeof(unsigned long long), s);
    if (e != hipSuccess) return nnc_set_error_(NNC_EHIP, hipGetErrorString(e));
    {
        const long long tiles = (n / 4 + 4 * SPLIT_THREADS - 1) / (4 * SPLIT_THREADS);
        int grid = (int)std::min<long long>(std::max<long long>(tiles, 1), 512);
        hipLaunchKernelGGL(k_split_signs, dim3(grid), dim3(SPLIT_THREADS), 0, s, x, (long long)n, t_all, t_all + n_neg, counter, (long long)n_neg, (long long)n_pos);
        e = hipGetLastError();
        if (e != hipSuccess) return nnc_set_error_(NNC_EHIP, hipGetErrorString(e));
    }
    // one sort of all the non-zeros (a radix sort of 8 M keys is well under twice the cost of one of 4 M), then the two
    // halves go to their places either side of the zeros
    if (n_nz > 0) {
        size_t need = sb;
        e = rocprim::radix_sort_keys(stemp, need, (const float *)t_all, t_sorted, (size_t)n_nz, 0, 32, s);
        if (e != hipSuccess) return nnc_set_error_(NNC_EHIP, hipGetErrorString(e));
    }
    if (n_neg > 0 && (e = hipMemcpyAsync(sorted_out, t_sorted, (size_t)n_neg * 4, hipMemcpyDeviceToDevice, s)) != hipSuccess)
        return nnc_set_error_(NNC_EHIP, hipGetErrorString(e));
    if (n_zero > 0 && (e = hipMemsetAsync(sorted_out + n_neg, 0, (size_t)n_zero * 4, s)) != hipSuccess)
        return nnc_set_error_(NNC_EHIP, hipGetErrorString(e));
    if (n_pos > 0 && (e = hipMemcpyAsync(sorted_out + n_neg + n_zero, t_sorted + n_neg, (size_t)n_pos * 4, hipMemcpyDeviceToDevice, s)) != hipSuccess)
        return nnc_set_error_(NNC_EHIP, hipGetErrorString(e));
    return NNC_OK;
}
