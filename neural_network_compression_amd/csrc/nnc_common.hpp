// nnc_common.hpp -- what the translation units of libnnc_hip.so share: error plumbing, launch macros (with the in-library profiler's
// event hooks), per-device caches, small device helpers and NumPy's pairwise float32 sum on a workgroup / a wave.  Internal: the C ABI is
// include/nnc.h.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <mutex>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <queue>
#include <string>
#include <vector>

#include "nnc.h"

int nnc_set_error_(int code, const char *msg);                 // nnc_core.hip: the thread's error text
int nnc_current_device_(void);                                 // nnc_core.hip
int nnc_cu_count_(void);                                       // nnc_core.hip: compute units of the current device (cached)
void nnc_prof_take_(int tag, hipEvent_t *a, hipEvent_t *b);    // nnc_core.hip: a pair of events if the profiler wants this launch timed

static inline int fail(int code, const std::string &msg) { return nnc_set_error_(code, msg.c_str()); }
static inline int current_device() { return nnc_current_device_(); }
static inline int cu_count() { return nnc_cu_count_(); }
static inline void prof_take(int tag, hipEvent_t *a, hipEvent_t *b) { nnc_prof_take_(tag, a, b); }
#define NNC_MAX_DEVICES 64

#define HIPCHK(expr)                                                                       \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess)                                                              \
            return fail(NNC_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_));      \
    } while (0)

#define LAUNCHCHK(name)                                                                    \
    do {                                                                                   \
        hipError_t e_ = hipGetLastError();                                                 \
        if (e_ != hipSuccess)                                                              \
            return fail(NNC_EHIP, std::string("launch ") + name + ": " + hipGetErrorString(e_)); \
    } while (0)

static inline hipStream_t S(void *stream) { return reinterpret_cast<hipStream_t>(stream); }

// workgroups of a streaming pass: enough for the work, at most per_cu per compute unit
static inline int stream_grid(int64_t work_items, int threads, int per_cu)
{
    int64_t blocks = (work_items + threads - 1) / threads;
    int64_t cap = (int64_t)cu_count() * per_cu;
    if (blocks < 1) blocks = 1;
    return (int)std::min<int64_t>(blocks, cap);
}

// hipExtLaunchKernelGGL stamps the events at the kernel's own begin and end (not at the command processor's arrival), so the
// difference is the launch's execution time; without a profiling session the events are null and the launch is a plain one
#define NNC_LAUNCH_PROF(tag, kernel, grid, block, lds, stream, ...)                                   \
    do {                                                                                               \
        hipEvent_t ea_, eb_;                                                                           \
        prof_take(tag, &ea_, &eb_);                                                                    \
        hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, ea_, eb_, 0, __VA_ARGS__);             \
    } while (0)

// small device helpers
// --------------------------------------------------------------------------------------
#define WAVE 64

__device__ __forceinline__ void wave_lds_fence()
{
    // LDS traffic of one wave executes in order; this only stops the compiler from moving
    // LDS reads across LDS writes of other lanes of the same wave.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

#define LEAF 128
struct PwHeap { int start[256]; int len[256]; float val[256]; };

// NumPy's pairwise float32 sum of n <= 8192 values by a whole workgroup (>= 256 threads).  The
// split tree (n/2 rounded down to a multiple of 8, leaves of <= 128) is laid out as a binary
// heap in LDS (node i -> children 2i, 2i+1; depth <= 7), leaves are summed by 8 lanes each, and
// the tree is folded level by level.  F(i) returns element i.  All threads get the result.

template <bool WAVE_ONLY = false, typename F>
__device__ float block_pairwise_sum(F elem, int n, PwHeap *hp)
{
    const int tid = threadIdx.x, nthr = blockDim.x;
    if (n <= LEAF) { // a single leaf: eight lanes of the first wave, no tree
        if (tid < 8) {
            float res = 0.0f;
            if (n < 8) {
                for (int i = 0; i < n; i++) res += elem(i);
            } else {
                float r = elem(tid);
                const int lim = n - (n % 8);
                for (int i = 8; i < lim; i += 8) r += elem(i + tid);
                r = r + __shfl_xor(r, 1);
                r = r + __shfl_xor(r, 2);
                r = r + __shfl_xor(r, 4);
                res = r;
                for (int i = lim; i < n; i++) res += elem(i);
            }
            if (tid == 0) hp->val[1] = res;
        }
        if (WAVE_ONLY) wave_lds_fence(); // (the caller is a single wave and n <= LEAF: no workgroup barrier anywhere)
        else __syncthreads();
        return hp->val[1];
    }
    for (int i = tid; i < 256; i += nthr) { hp->start[i] = 0; hp->len[i] = 0; hp->val[i] = 0.0f; }
    __syncthreads();
    if (tid == 0) { hp->start[1] = 0; hp->len[1] = n; }
    __syncthreads();
    int depth = 0; // levels with anything to split: a node of length l > 128 has children of about l/2
    while (depth < 7 && ((n + (1 << depth) - 1) >> depth) > LEAF) depth++;
    if (depth < 7) depth++; // rounding to multiples of 8 can push one child just over the leaf size
    for (int lev = 0; lev < depth; lev++) {
        const int i = (1 << lev) + tid;
        if (tid < (1 << lev)) {
            const int l = hp->len[i];
            if (l > LEAF) {
                int n2 = l / 2; n2 -= n2 % 8;
                hp->start[2 * i] = hp->start[i]; hp->len[2 * i] = n2;
                hp->start[2 * i + 1] = hp->start[i] + n2; hp->len[2 * i + 1] = l - n2;
            }
        }
        __syncthreads();
    }
    // leaves: 8 lanes per leaf (a group of 8 lanes stays together in the loop)
    const int j = tid & 7;
    for (int node = 1 + (tid >> 3); node < (2 << depth) && node < 256; node += (nthr >> 3)) {
        const int l = hp->len[node];
        if (l > 0 && l <= LEAF) {
            const int st = hp->start[node];
            float res;
            if (l < 8) {
                res = 0.0f;
                for (int i = 0; i < l; i++) res += elem(st + i);
            } else {
                float r = elem(st + j);
                const int lim = l - (l % 8);
                for (int i = 8; i < lim; i += 8) r += elem(st + i + j);
                r = r + __shfl_xor(r, 1);
                r = r + __shfl_xor(r, 2);
                r = r + __shfl_xor(r, 4);
                res = r;
                for (int i = lim; i < l; i++) res += elem(st + i);
            }
            if (j == 0) hp->val[node] = res;
        }
    }
    __syncthreads();
    for (int lev = depth - 1; lev >= 0; lev--) {
        const int i = (1 << lev) + tid;
        if (tid < (1 << lev) && hp->len[i] > LEAF) hp->val[i] = hp->val[2 * i] + hp->val[2 * i + 1];
        __syncthreads();
    }
    return hp->val[1];
}

// The same tree by ONE wave on its own (n <= 2048: at most 32 leaves), wave-level LDS fences instead of workgroup barriers:
// for the K-sized sums of the finalize step, where sixteen waves meeting at eight barriers cost more than the arithmetic.
// Every lane of the calling wave returns the sum; the other waves of the workgroup must not touch *hp meanwhile.
template <typename F>
__device__ float wave_pairwise_sum(F elem, int n, PwHeap *hp)
{
    const int lane = threadIdx.x & 63;
    if (n <= LEAF) {
        if (lane < 8) {
            float res = 0.0f;
            if (n < 8) {
                for (int i = 0; i < n; i++) res += elem(i);
            } else {
                float r = elem(lane);
                const int lim = n - (n % 8);
                for (int i = 8; i < lim; i += 8) r += elem(i + lane);
                r = r + __shfl_xor(r, 1);
                r = r + __shfl_xor(r, 2);
                r = r + __shfl_xor(r, 4);
                res = r;
                for (int i = lim; i < n; i++) res += elem(i);
            }
            if (lane == 0) hp->val[1] = res;
        }
        wave_lds_fence();
        return hp->val[1];
    }
    hp->start[lane] = 0; hp->len[lane] = 0; hp->val[lane] = 0.0f;
    wave_lds_fence();
    if (lane == 0) { hp->start[1] = 0; hp->len[1] = n; }
    wave_lds_fence();
    int depth = 0;
    while (depth < 5 && ((n + (1 << depth) - 1) >> depth) > LEAF) depth++;
    if (depth < 5) depth++; // rounding to multiples of 8 can push one child just over the leaf size
    for (int lev = 0; lev < depth; lev++) {
        const int i = (1 << lev) + lane;
        if (lane < (1 << lev)) {
            const int l = hp->len[i];
            if (l > LEAF) {
                int n2 = l / 2; n2 -= n2 % 8;
                hp->start[2 * i] = hp->start[i]; hp->len[2 * i] = n2;
                hp->start[2 * i + 1] = hp->start[i] + n2; hp->len[2 * i + 1] = l - n2;
            }
        }
        wave_lds_fence();
    }
    const int j8 = lane & 7;
    for (int node = 1 + (lane >> 3); node < (2 << depth) && node < 64; node += 8) {
        const int l = hp->len[node];
        if (l > 0 && l <= LEAF) {
            const int st = hp->start[node];
            float res;
            if (l < 8) {
                res = 0.0f;
                for (int i = 0; i < l; i++) res += elem(st + i);
            } else {
                float r = elem(st + j8);
                const int lim = l - (l % 8);
                for (int i = 8; i < lim; i += 8) r += elem(st + i + j8);
                r = r + __shfl_xor(r, 1);
                r = r + __shfl_xor(r, 2);
                r = r + __shfl_xor(r, 4);
                res = r;
                for (int i = lim; i < l; i++) res += elem(st + i);
            }
            if (j8 == 0) hp->val[node] = res;
        }
    }
    wave_lds_fence();
    for (int lev = depth - 1; lev >= 0; lev--) {
        const int i = (1 << lev) + lane;
        if (lane < (1 << lev) && hp->len[i] > LEAF) hp->val[i] = hp->val[2 * i] + hp->val[2 * i + 1];
        wave_lds_fence();
    }
    return hp->val[1];
}
