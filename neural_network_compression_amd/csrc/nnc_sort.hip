// nnc_sort.hip -- one-time reordering of a weight vector by value (ascending) for the Lloyd
// iterations.  Per-cluster sums do not depend on the order of the weights (exact integer
// sums), so the iterations may stream a sorted copy: neighbouring weights then fall in the
// same cluster and each lane accumulates runs in registers instead of issuing one LDS atomic
// per weight.  The sort itself is a plain library call (rocPRIM device radix sort), outside
// the per-iteration path; labels / quantized values are produced from the ORIGINAL vector.
#include <hip/hip_runtime.h>
#include <cmath>
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

// --------------------------------------------------------------------------------------
// The sort of a PRUNED vector whose bounds are known: the surviving weights lie in [vmin, -thr] and [thr, vmax], so their
// order-preserving integer images, taken relative to the two ends, fit in far fewer than 32 bits (26 at a 1-sigma threshold):
// three radix passes of at most 9 bits instead of the four 8-bit passes a general 32-bit sort takes.  Hand-written least-
// significant-digit radix sort of the compact keys: per pass a digit histogram per workgroup (every workgroup owns a contiguous
// range of the keys), an exclusive scan over (digit, workgroup), and a stable scatter in which a wave ranks its keys digit by
// digit with one ballot per digit bit and the workgroup writes them out digit run by digit run through LDS.
// --------------------------------------------------------------------------------------
#define RS_THREADS 256
#define RS_ITEMS 16
#define RS_TILE (RS_THREADS * RS_ITEMS)
#define RS_MAXBITS 9
#define RS_MAXR (1 << RS_MAXBITS)
#define RS_MAXBLOCKS 1024

__device__ __forceinline__ unsigned f32_ord(float v)
{
    const unsigned b = __float_as_uint(v);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float f32_unord(unsigned o)
{
    return __uint_as_float((o & 0x80000000u) ? (o ^ 0x80000000u) : ~o);
}

struct RsBounds { unsigned lo_neg, hi_neg, lo_pos, hi_pos, span_neg, total; };

// non-zeros -> compact keys, any order (one 64-bit counter hands out the output ranges, a tile at a time)
__global__ __launch_bounds__(SPLIT_THREADS) void k_split_keys(const float *__restrict__ x, long long n, unsigned *__restrict__ keys,
                                                              unsigned long long *__restrict__ counter, long long cap, RsBounds bd)
{
    __shared__ unsigned wave_tot[SPLIT_THREADS / 64];
    __shared__ unsigned long long base_s;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const bool vec = (reinterpret_cast<uintptr_t>(x) & 15) == 0;
    const long long nvec = vec ? (n >> 2) : 0;
    const float4 *x4 = reinterpret_cast<const float4 *>(x);
    // (a value outside the caller's bounds -- or a NaN, which is neither below zero nor inside the positive range -- is clamped so
    // that nothing is written out of range, and reported: the sorted vector then holds values that are not in the input)
    bool outside = false;
    auto key_of = [&](float e) -> unsigned {
        const unsigned o = f32_ord(e);
        if (e < 0.0f) { outside |= (o < bd.lo_neg) | (o > bd.hi_neg); const unsigned c = o < bd.lo_neg ? bd.lo_neg : (o > bd.hi_neg ? bd.hi_neg : o); return c - bd.lo_neg; }
        outside |= (o < bd.lo_pos) | (o > bd.hi_pos);
        const unsigned c = o < bd.lo_pos ? bd.lo_pos : (o > bd.hi_pos ? bd.hi_pos : o);
        return bd.span_neg + (c - bd.lo_pos);
    };
    const long long ntiles = (nvec + 4 * SPLIT_THREADS - 1) / (4 * SPLIT_THREADS);
    for (long long t = blockIdx.x; t < ntiles; t += gridDim.x) {
        float4 q[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const long long v = t * (4 * SPLIT_THREADS) + (long long)j * SPLIT_THREADS + tid;
            q[j] = v < nvec ? x4[v] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        unsigned c = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const float e[4] = {q[j].x, q[j].y, q[j].z, q[j].w};
#pragma unroll
            for (int i = 0; i < 4; i++) c += (e[i] == 0.0f) ? 0u : 1u;
        }
        unsigned sc = c;
        for (int off = 1; off < 64; off <<= 1) { const unsigned o = __shfl_up(sc, off); if (lane >= off) sc += o; }
        if (lane == 63) wave_tot[wv] = sc;
        __syncthreads();
        unsigned pre = 0, tot = 0;
        for (int w = 0; w < SPLIT_THREADS / 64; w++) { const unsigned o = wave_tot[w]; if (w < wv) pre += o; tot += o; }
        if (tid == 0) base_s = atomicAdd(counter, (unsigned long long)tot);
        __syncthreads();
        long long op = (long long)base_s + (pre + sc - c);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const float e[4] = {q[j].x, q[j].y, q[j].z, q[j].w};
#pragma unroll
            for (int i = 0; i < 4; i++)
                if (!(e[i] == 0.0f)) { if (op < cap) keys[op] = key_of(e[i]); op++; }
        }
        __syncthreads();
    }
    if (blockIdx.x == 0) {
        for (long long i0 = nvec << 2; i0 < n; i0 += SPLIT_THREADS) {
            const long long i = i0 + tid;
            const float e = i < n ? x[i] : 0.0f;
            const bool nzv = !(e == 0.0f);
            const unsigned long long bn = __ballot(nzv);
            unsigned long long base = 0;
            if (lane == 0 && bn) base = atomicAdd(counter, (unsigned long long)__popcll(bn));
            base = __shfl(base, 0);
            const long long at = (long long)base + __popcll(bn & ((1ull << lane) - 1ull));
            if (nzv && at < cap) keys[at] = key_of(e);
        }
    }
    if (outside) atomicOr(counter + 1, 1ull);
}

// every workgroup owns the contiguous range [b * chunk, (b + 1) * chunk) of the keys; table[d * nblk + b] = its keys with digit d
__global__ __launch_bounds__(RS_THREADS) void k_rs_hist(const unsigned *__restrict__ in, long long n, long long chunk, int shift, int rb,
                                                        unsigned *__restrict__ table, int nblk)
{
    __shared__ unsigned h[RS_MAXR];
    const int R = 1 << rb;
    for (int d = threadIdx.x; d < R; d += RS_THREADS) h[d] = 0u;
    __syncthreads();
    const long long lo = (long long)blockIdx.x * chunk, hi = lo + chunk < n ? lo + chunk : n;
    const unsigned mask = (unsigned)R - 1u;
    for (long long i = lo + threadIdx.x; i < hi; i += RS_THREADS) atomicAdd(&h[(in[i] >> shift) & mask], 1u);
    __syncthreads();
    for (int d = threadIdx.x; d < R; d += RS_THREADS) table[(size_t)d * nblk + blockIdx.x] = h[d];
}

// one workgroup per digit: exclusive scan of its nblk (<= 1024: four per thread) counts in place, total to totals[d]
__global__ __launch_bounds__(RS_THREADS) void k_rs_scan(unsigned *__restrict__ table, int nblk, unsigned *__restrict__ totals)
{
    __shared__ unsigned wsum[RS_THREADS / 64];
    const int d = blockIdx.x, t = threadIdx.x, lane = t & 63, wv = t >> 6;
    unsigned v[4], own = 0;
#pragma unroll
    for (int r = 0; r < 4; r++) { const int b = 4 * t + r; v[r] = b < nblk ? table[(size_t)d * nblk + b] : 0u; own += v[r]; }
    unsigned sc = own;
    for (int off = 1; off < 64; off <<= 1) { const unsigned o = __shfl_up(sc, off); if (lane >= off) sc += o; }
    if (lane == 63) wsum[wv] = sc;
    __syncthreads();
    unsigned pre = 0, tot = 0;
    for (int w = 0; w < RS_THREADS / 64; w++) { if (w < wv) pre += wsum[w]; tot += wsum[w]; }
    unsigned run = pre + sc - own;
#pragma unroll
    for (int r = 0; r < 4; r++) { const int b = 4 * t + r; if (b < nblk) table[(size_t)d * nblk + b] = run; run += v[r]; }
    if (t == 0) totals[d] = tot;
}

__global__ __launch_bounds__(RS_THREADS) void k_rs_scatter(const unsigned *__restrict__ in, unsigned *__restrict__ out, long long n, long long chunk,
                                                           int shift, int rb, const unsigned *__restrict__ table, const unsigned *__restrict__ totals, int nblk)
{
    __shared__ unsigned keys_s[RS_TILE];
    __shared__ unsigned wcnt[RS_THREADS / 64][RS_MAXR]; // per wave: keys with this digit so far in the tile; then the wave's offset inside the digit
    __shared__ unsigned tpre[RS_MAXR];                  // tile: keys with a smaller digit
    __shared__ unsigned gbase[RS_MAXR];                 // where this workgroup's next key with digit d goes
    __shared__ unsigned wsum[RS_THREADS / 64];
    const int R = 1 << rb, t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const unsigned mask = (unsigned)R - 1u;
    // digit bases: exclusive scan of the R totals (two per thread), plus this workgroup's offset inside the digit
    {
        const unsigned a = (2 * t < R) ? totals[2 * t] : 0u, b = (2 * t + 1 < R) ? totals[2 * t + 1] : 0u;
        unsigned sc = a + b;
        const unsigned own = sc;
        for (int off = 1; off < 64; off <<= 1) { const unsigned o = __shfl_up(sc, off); if (lane >= off) sc += o; }
        if (lane == 63) wsum[wv] = sc;
        __syncthreads();
        unsigned pre = 0;
        for (int w = 0; w < wv; w++) pre += wsum[w];
        const unsigned ex = pre + sc - own;
        if (2 * t < R) gbase[2 * t] = ex + table[(size_t)(2 * t) * nblk + blockIdx.x];
        if (2 * t + 1 < R) gbase[2 * t + 1] = ex + a + table[(size_t)(2 * t + 1) * nblk + blockIdx.x];
        __syncthreads();
    }
    const long long lo = (long long)blockIdx.x * chunk, hi = lo + chunk < n ? lo + chunk : n;
    for (long long tile0 = lo; tile0 < hi; tile0 += RS_TILE) {
        const int cnt_tile = (int)((hi - tile0) < RS_TILE ? (hi - tile0) : RS_TILE);
        for (int d = t; d < R; d += RS_THREADS) { wcnt[0][d] = 0u; wcnt[1][d] = 0u; wcnt[2][d] = 0u; wcnt[3][d] = 0u; }
        __syncthreads();
        unsigned key[RS_ITEMS];
        unsigned short rnk[RS_ITEMS];
        // order of the keys inside the tile: wave, item, lane
#pragma unroll
        for (int i = 0; i < RS_ITEMS; i++) {
            const int idx = wv * (64 * RS_ITEMS) + i * 64 + lane;
            const bool valid = idx < cnt_tile;
            key[i] = valid ? in[tile0 + idx] : 0xFFFFFFFFu;
        }
#pragma unroll
        for (int i = 0; i < RS_ITEMS; i++) {
            const int idx = wv * (64 * RS_ITEMS) + i * 64 + lane;
            const bool valid = idx < cnt_tile;
            const unsigned d = (key[i] >> shift) & mask;
            unsigned long long peers = __ballot(valid);
            for (int b = 0; b < rb; b++) {
                const bool bit = (d >> b) & 1u;
                const unsigned long long mb = __ballot(valid && bit);
                peers &= bit ? mb : ~mb;
            }
            const unsigned before = (unsigned)__builtin_amdgcn_mbcnt_hi((unsigned)(peers >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)peers, 0u));
            unsigned prev = 0;
            if (valid) {
                prev = wcnt[wv][d];                                            // (every lane of the group reads before its first lane writes)
                if (before == 0) wcnt[wv][d] = prev + (unsigned)__popcll(peers);
            }
            rnk[i] = (unsigned short)(prev + before);
            __builtin_amdgcn_wave_barrier(); // (the next item's reads of the wave's counters come after this item's writes)
        }
        __syncthreads();
        // per digit: the waves' offsets inside the digit and the tile's count; then the tile's exclusive prefix over the digits
        unsigned tc[2] = {0u, 0u};
        for (int r = 0; r < 2; r++) {
            const int d = 2 * t + r;
            if (d < R) {
                unsigned run = 0;
                for (int w = 0; w < RS_THREADS / 64; w++) { const unsigned c = wcnt[w][d]; wcnt[w][d] = run; run += c; }
                tc[r] = run;
            }
        }
        {
            unsigned sc = tc[0] + tc[1];
            const unsigned own = sc;
            for (int off = 1; off < 64; off <<= 1) { const unsigned o = __shfl_up(sc, off); if (lane >= off) sc += o; }
            if (lane == 63) wsum[wv] = sc;
            __syncthreads();
            unsigned pre = 0;
            for (int w = 0; w < wv; w++) pre += wsum[w];
            const unsigned ex = pre + sc - own;
            if (2 * t < R) tpre[2 * t] = ex;
            if (2 * t + 1 < R) tpre[2 * t + 1] = ex + tc[0];
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < RS_ITEMS; i++) {
            const int idx = wv * (64 * RS_ITEMS) + i * 64 + lane;
            if (idx < cnt_tile) {
                const unsigned d = (key[i] >> shift) & mask;
                keys_s[tpre[d] + wcnt[wv][d] + rnk[i]] = key[i];
            }
        }
        __syncthreads();
        for (int j = t; j < cnt_tile; j += RS_THREADS) {
            const unsigned kv = keys_s[j];
            const unsigned d = (kv >> shift) & mask;
            out[(size_t)gbase[d] + (unsigned)(j - (int)tpre[d])] = kv;
        }
        __syncthreads();
        for (int r = 0; r < 2; r++) {
            const int d = 2 * t + r;
            if (d < R) gbase[d] += tc[r];
        }
        __syncthreads();
    }
}

// sorted compact keys -> the sorted vector: negatives, the zeros, positives
__global__ __launch_bounds__(256) void k_rs_assemble(const unsigned *__restrict__ keys, long long n, long long n_neg, long long n_zero,
                                                     float *__restrict__ out, RsBounds bd, const unsigned long long *__restrict__ counter,
                                                     int *__restrict__ flag_out)
{
    if (flag_out && blockIdx.x == 0 && threadIdx.x == 0) *flag_out = counter[1] != 0ull; // (k_split_keys met a value outside the bounds)
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float v;
        if (i < n_neg) v = f32_unord(keys[i] + bd.lo_neg);
        else if (i < n_neg + n_zero) v = 0.0f;
        else v = f32_unord(keys[i - n_zero] - bd.span_neg + bd.lo_pos);
        out[i] = v;
    }
}

static unsigned host_ord(float v)
{
    unsigned b;
    std::memcpy(&b, &v, 4);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

// bits of the compact keys, 0 if the bounded form does not apply (no threshold, bounds out of order, more than 27 bits)
static int rs_bounds(float vmin, float vmax, float thr, int64_t n_neg, int64_t n_pos, RsBounds *bd)
{
    if (!(thr > 0.0f) || !std::isfinite(thr) || !std::isfinite(vmin) || !std::isfinite(vmax)) return 0;
    RsBounds b;
    std::memset(&b, 0, sizeof(b));
    unsigned long long total = 0;
    if (n_neg > 0) {
        if (!(vmin <= -thr)) return 0;
        b.lo_neg = host_ord(vmin); b.hi_neg = host_ord(-thr);
        b.span_neg = b.hi_neg - b.lo_neg + 1u;
        total += b.span_neg;
    }
    if (n_pos > 0) {
        if (!(vmax >= thr)) return 0;
        b.lo_pos = host_ord(thr); b.hi_pos = host_ord(vmax);
        total += (unsigned long long)(b.hi_pos - b.lo_pos) + 1ull;
    }
    if (total == 0 || total > (1ull << 27)) return 0;
    b.total = (unsigned)total;
    int bits = 1;
    while ((1ull << bits) < total) bits++;
    *bd = b;
    return bits;
}

extern "C" int32_t nnc_sort_pruned_bounded_bits(float vmin, float vmax, float thr, int64_t n_neg, int64_t n_pos)
{
    RsBounds b;
    return rs_bounds(vmin, vmax, thr, n_neg, n_pos, &b);
}

extern "C" size_t nnc_sort_pruned_bounded_workspace_bytes(int64_t n_nz)
{
    if (n_nz < 0) return 0;
    return 2 * al256((size_t)n_nz * 4 + 16) + al256(8) + al256((size_t)RS_MAXR * RS_MAXBLOCKS * 4) + al256(RS_MAXR * 4) + 256;
}

int nnc_sort_pruned_bounded_flagged_(const float *x, int64_t n, int64_t n_neg, int64_t n_zero, float vmin, float vmax, float thr,
                                     float *sorted_out, void *ws, size_t ws_bytes, int32_t *flag_dev, void *stream);
extern "C" int nnc_sort_pruned_bounded_f32(const float *x, int64_t n, int64_t n_neg, int64_t n_zero, float vmin, float vmax, float thr,
                                           float *sorted_out, void *ws, size_t ws_bytes, void *stream)
{
    return nnc_sort_pruned_bounded_flagged_(x, n, n_neg, n_zero, vmin, vmax, thr, sorted_out, ws, ws_bytes, nullptr, stream);
}

static unsigned char *rs_ws_base(void *ws) { return reinterpret_cast<unsigned char *>((reinterpret_cast<uintptr_t>(ws) + 255) & ~(uintptr_t)255); }

// device int32 inside the workspace: non-zero after the sort iff a weight lay outside [vmin, -thr] u {0} u [thr, vmax] (or was NaN)
extern "C" const int32_t *nnc_sort_pruned_bounded_flag(void *ws, int64_t n_nonzero)
{
    if (!ws || n_nonzero < 0) return nullptr;
    unsigned char *b = rs_ws_base(ws) + 2 * al256((size_t)n_nonzero * 4 + 16);
    return reinterpret_cast<const int32_t *>(b + 8); // (the low half of the 64-bit word behind the counter)
}

// (flag_dev: where the verdict goes as well -- the layer call keeps it next to the scalars it reads back anyway)
int nnc_sort_pruned_bounded_flagged_(const float *x, int64_t n, int64_t n_neg, int64_t n_zero, float vmin, float vmax, float thr,
                                     float *sorted_out, void *ws, size_t ws_bytes, int32_t *flag_dev, void *stream)
{
    if (n < 0 || n_neg < 0 || n_zero < 0 || n_neg + n_zero > n || n >= ((int64_t)1 << 31) || (n > 0 && (!x || !sorted_out)))
        return nnc_set_error_(NNC_EINVAL, "nnc_sort_pruned_bounded_f32: bad argument");
    if (n == 0) return NNC_OK;
    const int64_t n_pos = n - n_neg - n_zero, n_nz = n_neg + n_pos;
    RsBounds bd;
    std::memset(&bd, 0, sizeof(bd));
    const int bits = n_nz > 0 ? rs_bounds(vmin, vmax, thr, n_neg, n_pos, &bd) : 1;
    if (bits == 0) return nnc_set_error_(NNC_EINVAL, "nnc_sort_pruned_bounded_f32: bounds do not apply (see nnc_sort_pruned_bounded_bits)");
    if (!ws || ws_bytes < nnc_sort_pruned_bounded_workspace_bytes(n_nz)) return nnc_set_error_(NNC_ENOSPACE, "nnc_sort_pruned_bounded_f32: workspace too small");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    unsigned char *b = reinterpret_cast<unsigned char *>((reinterpret_cast<uintptr_t>(ws) + 255) & ~(uintptr_t)255);
    unsigned *ka = reinterpret_cast<unsigned *>(b); b += al256((size_t)n_nz * 4 + 16);
    unsigned *kb = reinterpret_cast<unsigned *>(b); b += al256((size_t)n_nz * 4 + 16);
    unsigned long long *counter = reinterpret_cast<unsigned long long *>(b); b += al256(16); // [0] keys written, [1] a value outside the bounds
    unsigned *table = reinterpret_cast<unsigned *>(b); b += al256((size_t)RS_MAXR * RS_MAXBLOCKS * 4);
    unsigned *totals = reinterpret_cast<unsigned *>(b);
    hipError_t e = hipMemsetAsync(counter, 0, 2 * sizeof(unsigned long long), s);
    if (e != hipSuccess) return nnc_set_error_(NNC_EHIP, hipGetErrorString(e));
    if (n_nz > 0) {
        const long long tiles = (n / 4 + 4 * SPLIT_THREADS - 1) / (4 * SPLIT_THREADS);
        const int grid = (int)std::min<long long>(std::max<long long>(tiles, 1), 512); // (one tile a workgroup, grid 2048, was measured: 324 us for the whole sorted copy against 315)
        hipLaunchKernelGGL(k_split_keys, dim3(grid), dim3(SPLIT_THREADS), 0, s, x, (long long)n, ka, counter, (long long)n_nz, bd);
        if ((e = hipGetLastError()) != hipSuccess) return nnc_set_error_(NNC_EHIP, hipGetErrorString(e));
        const int passes = (bits + RS_MAXBITS - 1) / RS_MAXBITS;
        const long long ntiles = (n_nz + RS_TILE - 1) / RS_TILE;
        const int nblk = (int)std::min<long long>(RS_MAXBLOCKS, ntiles);
        const long long chunk = ((ntiles + nblk - 1) / nblk) * RS_TILE;
        int shift = 0;
        for (int p = 0; p < passes; p++) {
            const int rb = (bits - shift + (passes - p) - 1) / (passes - p); // the remaining bits, spread evenly over the remaining passes
            hipLaunchKernelGGL(k_rs_hist, dim3(nblk), dim3(RS_THREADS), 0, s, ka, (long long)n_nz, chunk, shift, rb, table, nblk);
            hipLaunchKernelGGL(k_rs_scan, dim3(1 << rb), dim3(RS_THREADS), 0, s, table, nblk, totals);
            hipLaunchKernelGGL(k_rs_scatter, dim3(nblk), dim3(RS_THREADS), 0, s, ka, kb, (long long)n_nz, chunk, shift, rb, table, totals, nblk);
            if ((e = hipGetLastError()) != hipSuccess) return nnc_set_error_(NNC_EHIP, hipGetErrorString(e));
            std::swap(ka, kb);
            shift += rb;
        }
    }
    {
        const int grid = (int)std::min<long long>((n + 1023) / 1024, 2048);
        hipLaunchKernelGGL(k_rs_assemble, dim3(std::max(grid, 1)), dim3(256), 0, s, ka, (long long)n, (long long)n_neg, (long long)n_zero, sorted_out, bd, (const unsigned long long *)counter, (int *)flag_dev);
        if ((e = hipGetLastError()) != hipSuccess) return nnc_set_error_(NNC_EHIP, hipGetErrorString(e));
    }
    return NNC_OK;
}
