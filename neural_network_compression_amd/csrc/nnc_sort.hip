// nnc_sort.hip -- one-time reordering of a weight vector by value (ascending) for the Lloyd iterations.
// Per-cluster sums do not depend on the order of the weights (exact integer sums), so the iterations work on a sorted copy
// (rank boundaries, prefix sums: nnc_hip.hip section 4); labels / quantized values are produced from the ORIGINAL vector.
//
// Hand-written least-significant-digit radix sort of 32-bit keys (round 4; rocPRIM is gone), one kernel per digit pass:
//   k_os_prep   one read of the vector: order-preserving integer keys (of a pruned vector only the non-zeros, compacted through
//               LDS and written out whole lines at a time) and the digit histograms of ALL passes (LDS, then global atomics);
//   k_os_pass   per digit: a workgroup takes tiles of 8192 keys by ticket, ranks them (a wave ranks its keys digit by digit
//               with one ballot per digit bit), learns where the tile's keys of every digit go from the tiles before it by
//               decoupled look-back over one 64-bit status word per (tile, digit) -- {tag, count}, relaxed agent-scope atomics,
//               self-validating, so no fence anywhere --, reorders the tile by digit in LDS and writes every digit's run
//               contiguously.  Per pass the keys are read once and written once; the last pass writes floats, for a pruned
//               vector either side of the block of zeros, which is a memset.
// A pruned vector whose bounds are known has keys of 26 bits at a 1-sigma threshold (three passes of 9 / 9 / 8 bits); anything
// else takes four passes of 8 bits.  Tickets guarantee that every tile a look-back waits for belongs to a workgroup that is
// already running; the wait is bounded all the same (a flag says so and the successors are released).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdlib>
#include <cstring>

#include <algorithm>
#include <string>

#include "nnc.h"

extern "C" const char *nnc_last_error(void);
int nnc_set_error_(int code, const char *msg); // in nnc_hip.hip

#ifndef OS_THREADS
#define OS_THREADS 1024 // (measured on the bench vector, 8 M keys of 26 bits: 1024 x 16 x 1 workgroup a CU 217 us for the sorted copy, 512 x 16 x 2: 229,
#endif                  //  512 x 16 x 3: 277, 256 x 16 x 4: 347 -- the fewer tiles, the shorter the look-back chains of the first round)
#define OS_WAVES (OS_THREADS / 64)
#ifndef OS_ITEMS
#define OS_ITEMS 16
#endif
#ifndef OS_BLOCKS_PER_CU
#define OS_BLOCKS_PER_CU 1
#endif
#ifndef OS_RANK_LDS
#define OS_RANK_LDS 1 // 1: a wave finds the lanes that share a digit through a 64-bit lane mask per digit in LDS (one atomic OR, one read); 0: one ballot per digit bit
#endif
#define OS_TILE (OS_THREADS * OS_ITEMS)
#define OS_MAXBITS 9
#define OS_MAXR (1 << OS_MAXBITS)
#define OS_MAXPASS 4
#define OS_SPIN_LIMIT (1 << 22) // polls of one status word before a look-back gives up (seconds; never seen)
#define OSP_THREADS 1024
#define OSP_PER 16
#define OSP_TILE (OSP_THREADS * OSP_PER)

__device__ __forceinline__ unsigned f32_ord(float v)
{
    const unsigned b = __float_as_uint(v);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float f32_unord(unsigned o)
{
    return __uint_as_float((o & 0x80000000u) ? (o ^ 0x80000000u) : ~o);
}

// key <-> value.  bounded: the non-zero weights lie in [lo_neg, hi_neg] (negative) and [lo_pos, hi_pos] (as ordered images); key =
// distance from the lower end of the negative range, the positive range follows it directly.  Otherwise key = ordered image.
struct OsMap { unsigned lo_neg, hi_neg, lo_pos, hi_pos, span_neg; int bounded; };
struct OsPlan { int passes; int shift[OS_MAXPASS]; int rb[OS_MAXPASS]; };

__device__ __forceinline__ unsigned os_key(const OsMap &m, float e, bool &outside)
{
    const unsigned o = f32_ord(e);
    if (!m.bounded) return o;
    // (a value outside the caller's bounds -- or a NaN, which is neither below zero nor inside the positive range -- is clamped so
    // that nothing is written out of range, and reported: the sorted vector then holds values that are not in the input)
    if (e < 0.0f) { outside |= (o < m.lo_neg) | (o > m.hi_neg); const unsigned c = o < m.lo_neg ? m.lo_neg : (o > m.hi_neg ? m.hi_neg : o); return c - m.lo_neg; }
    outside |= (o < m.lo_pos) | (o > m.hi_pos);
    const unsigned c = o < m.lo_pos ? m.lo_pos : (o > m.hi_pos ? m.hi_pos : o);
    return m.span_neg + (c - m.lo_pos);
}
__device__ __forceinline__ float os_value(const OsMap &m, unsigned k)
{
    if (!m.bounded) return f32_unord(k);
    return f32_unord(k < m.span_neg ? k + m.lo_neg : k - m.span_neg + m.lo_pos);
}

// ctrl block of a sort (device): [0] keys written by k_os_prep, [1] flags (1 = a value outside the bounds, 2 = a look-back gave up),
// [2..3] tile tickets of the passes (32 bits each)
#define OS_CTRL_WORDS 4

// One read of the vector.  SKIPZ: exact zeros are dropped, the other weights' keys go to `keys` in any order (a 64-bit counter hands
// out the output ranges a tile at a time; the tile is compacted in LDS and leaves in whole lines).  Digit histogram of the FIRST pass
// (LDS atomics retire about one lane a clock on gfx950: one histogram costs a pass's worth of them, so every radix pass makes the
// histogram of the next one while it has the keys in registers, and this kernel only the first).
template <bool SKIPZ>
__global__ __launch_bounds__(OSP_THREADS) void k_os_prep(const float *__restrict__ x, long long n, unsigned *__restrict__ keys,
                                                         unsigned long long *__restrict__ ctrl, long long cap, OsMap km, int rb0, unsigned *__restrict__ ghist)
{
    __shared__ unsigned comp_s[SKIPZ ? OSP_TILE : 1];
    __shared__ unsigned hist_s[OS_MAXR];
    __shared__ unsigned wave_tot[OSP_THREADS / 64];
    __shared__ unsigned long long base_s;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (int i = tid; i < OS_MAXR; i += OSP_THREADS) hist_s[i] = 0u;
    __syncthreads();
    const bool vec = (reinterpret_cast<uintptr_t>(x) & 15) == 0;
    const float4 *x4 = reinterpret_cast<const float4 *>(x);
    const unsigned mask0 = (1u << rb0) - 1u;
    bool outside = false;
    const long long ntiles = (n + OSP_TILE - 1) / OSP_TILE;
    // the loads of a tile are issued a tile ahead: the compaction of one tile (five barriers, one global atomic) covers the next one's way from memory
    auto load_tile = [&](long long tt, float (&e)[OSP_PER]) {
        const long long tile0 = tt * OSP_TILE;
        if (tt >= ntiles) return;
        if (vec && tile0 + OSP_TILE <= n) {
#pragma unroll
            for (int j = 0; j < OSP_PER / 4; j++) {
                const float4 q = x4[(tile0 >> 2) + (long long)j * OSP_THREADS + tid];
                e[4 * j] = q.x; e[4 * j + 1] = q.y; e[4 * j + 2] = q.z; e[4 * j + 3] = q.w;
            }
        } else { // ragged last tile, or a vector that is not 16-byte aligned: one by one, zero beyond the end
#pragma unroll
            for (int j = 0; j < OSP_PER; j++) {
                const long long i = tile0 + (long long)j * OSP_THREADS + tid;
                e[j] = i < n ? x[i] : 0.0f;
            }
        }
    };
    float en[OSP_PER];
    load_tile(blockIdx.x, en);
    for (long long t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const long long tile0 = t * OSP_TILE;
        const bool fast = vec && tile0 + OSP_TILE <= n; // (how the elements were loaded decides which position element j is)
        float e[OSP_PER];
#pragma unroll
        for (int j = 0; j < OSP_PER; j++) e[j] = en[j];
        load_tile(t + gridDim.x, en);
        unsigned key[OSP_PER];
        unsigned take = 0; // bit j: element j is a key
#pragma unroll
        for (int j = 0; j < OSP_PER; j++) {
            bool ok = SKIPZ ? !(e[j] == 0.0f) : true;
            if (!SKIPZ && !fast) ok = tile0 + (long long)j * OSP_THREADS + tid < n; // (positions beyond the end read as zero above: not keys)
            key[j] = 0u;
            if (ok) { key[j] = os_key(km, e[j], outside); take |= 1u << j; }
        }
#pragma unroll
        for (int j = 0; j < OSP_PER; j++)
            if (take & (1u << j)) atomicAdd(&hist_s[key[j] & mask0], 1u);
        if (SKIPZ) {
            const unsigned c = (unsigned)__popc(take);
            unsigned sc = c;
            for (int off = 1; off < 64; off <<= 1) { const unsigned o = __shfl_up(sc, off); if (lane >= off) sc += o; }
            if (lane == 63) wave_tot[wv] = sc;
            __syncthreads();
            unsigned pre = 0, tot = 0;
            for (int w = 0; w < OSP_THREADS / 64; w++) { const unsigned o = wave_tot[w]; if (w < wv) pre += o; tot += o; }
            if (tid == 0) base_s = atomicAdd(ctrl, (unsigned long long)tot);
            unsigned at = pre + sc - c;
#pragma unroll
            for (int j = 0; j < OSP_PER; j++)
                if (take & (1u << j)) comp_s[at++] = key[j];
            __syncthreads();
            const long long base = (long long)base_s;
            for (unsigned i = tid; i < tot; i += OSP_THREADS)
                if (base + i < cap) keys[base + i] = comp_s[i]; // (bounds: the counts are the caller's; wrong ones must not fault)
            __syncthreads(); // comp_s / wave_tot / base_s are reused by the next tile
        }
    }
    if (outside) atomicOr(ctrl + 1, 1ull);
    __syncthreads();
    for (int i = tid; i < OS_MAXR; i += OSP_THREADS) {
        const unsigned v = hist_s[i];
        if (v) atomicAdd(&ghist[i], v);
    }
}

#ifdef NNC_DIAG
__device__ unsigned long long *g_os_trace = nullptr; // diagnostics: per tile of a pass 8 timestamps (10 ns ticks) by thread 0
extern "C" int nnc_debug_os_trace(void *buf)
{
    unsigned long long *p = reinterpret_cast<unsigned long long *>(buf);
    return hipMemcpyToSymbol(HIP_SYMBOL(g_os_trace), &p, sizeof(p)) == hipSuccess ? NNC_OK : NNC_EHIP;
}
#define OS_STAMP(i) do { if (g_os_trace && t == 0) g_os_trace[((size_t)pass * 8192 + (size_t)(tile < 8192 ? tile : 8191)) * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define OS_STAMP(i) do { } while (0)
#endif

// status word of (tile, digit): tag << 32 | count; tag 2 * pass + 1 = the tile's own count ("aggregate"), 2 * pass + 2 = the count of
// this tile and all tiles before it ("prefix").  Zeroed before the sort; tags only grow, so a word of an earlier pass reads as "not yet".
__device__ __forceinline__ void os_publish(unsigned long long *w, unsigned tag, unsigned v)
{
    __hip_atomic_store(w, ((unsigned long long)tag << 32) | v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long os_peek(const unsigned long long *w)
{
    return __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The keys of one wave's item (64 keys, one a lane) that share this lane's digit, as a lane mask: one ballot per digit bit.
// FULL: every lane holds a key.  (32-bit halves: v_bfe_i32, v_cmp, and two xnor + and per bit.)
template <int RB, bool FULL>
__device__ __forceinline__ void os_peers(unsigned d, bool valid, unsigned &plo, unsigned &phi)
{
    const unsigned long long all = FULL ? ~0ull : __ballot(valid);
    plo = (unsigned)all; phi = (unsigned)(all >> 32);
#pragma unroll
    for (int b = 0; b < RB; b++) {
        const int m = ((int)(d << (31 - b))) >> 31; // 0 or -1: bit b of the digit, spread over the word
        const unsigned long long mb = FULL ? __ballot(m != 0) : __ballot(valid && m != 0);
        plo &= ~((unsigned)mb ^ (unsigned)m);
        phi &= ~((unsigned)(mb >> 32) ^ (unsigned)m);
    }
}

#ifndef OS_LOOK
#define OS_LOOK 4 // status words a look-back round has in flight per digit
#endif

// One digit pass.  RB: bits of the digit.  SRC 0: unsigned keys; 1: floats (keys are their ordered images: the first pass of an unpruned
// vector reads it directly).  DST 0: unsigned keys; 1: floats (the last pass): key k of rank r goes to out[r + (r >= n_neg ? n_zero : 0)]
// as its value.  nrb > 0: the pass also makes the digit histogram of the next pass (digit = (key >> (shift + RB)) & (2^nrb - 1)).
template <int RB, int SRC, int DST>
__global__ __launch_bounds__(OS_THREADS, OS_BLOCKS_PER_CU * OS_THREADS / 256) void k_os_pass(const void *__restrict__ in_, void *__restrict__ out_, long long n, int shift, int pass, int nrb,
                                                           const unsigned *__restrict__ ghist, unsigned *__restrict__ ghist_next, unsigned long long *__restrict__ status,
                                                           unsigned long long *__restrict__ ctrl, OsMap km, long long n_neg, long long n_zero, int abl)
{
    constexpr int R = 1 << RB;
    constexpr unsigned MASK = (unsigned)R - 1u;
    constexpr int DPT = (R + OS_THREADS - 1) / OS_THREADS; // digits per thread: thread t owns digits t * DPT .. t * DPT + DPT - 1
    __shared__ unsigned keys_s[OS_TILE];
    __shared__ unsigned short wcnt[OS_WAVES][R]; // per wave: keys with this digit so far in the tile (at most 64 * OS_ITEMS); then the wave's offset inside the digit
    __shared__ unsigned short tpre[R];           // tile: keys with a smaller digit
    __shared__ unsigned gpos[R];                 // where the tile's first key of digit d goes, minus tpre[d] (mod 2^32)
    __shared__ unsigned gdig[R];                 // all keys with a smaller digit (from the histogram of the pass)
    __shared__ unsigned nhist[OS_MAXR];          // digit histogram of the next pass, this workgroup's tiles
    __shared__ unsigned wsum[OS_WAVES];
    __shared__ long long tile_s;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const unsigned *in_k = reinterpret_cast<const unsigned *>(in_);
    const float *in_f = reinterpret_cast<const float *>(in_);
    unsigned *out_k = reinterpret_cast<unsigned *>(out_);
    float *out_f = reinterpret_cast<float *>(out_);
    const unsigned tagA = 2u * (unsigned)pass + 1u, tagP = tagA + 1u;
    unsigned *ticket = reinterpret_cast<unsigned *>(ctrl + 2) + pass;
    const int nshift = shift + RB;
    const unsigned nmask = nrb > 0 ? (1u << nrb) - 1u : 0u;
    // exclusive scan of the R digit totals
    {
        unsigned v[DPT], own = 0;
#pragma unroll
        for (int r = 0; r < DPT; r++) { const int d = t * DPT + r; v[r] = d < R ? ghist[d] : 0u; own += v[r]; }
        unsigned sc = own;
        for (int off = 1; off < 64; off <<= 1) { const unsigned o = __shfl_up(sc, off); if (lane >= off) sc += o; }
        if (lane == 63) wsum[wv] = sc;
        for (int i = t; i < OS_MAXR; i += OS_THREADS) nhist[i] = 0u;
        __syncthreads();
        unsigned run = sc - own;
        for (int w = 0; w < wv; w++) run += wsum[w];
#pragma unroll
        for (int r = 0; r < DPT; r++) { const int d = t * DPT + r; if (d < R) gdig[d] = run; run += v[r]; }
        __syncthreads();
    }
    const long long ntiles = (n + OS_TILE - 1) / OS_TILE;
    bool gave_up = false;
    for (;;) {
        if (t == 0) tile_s = (long long)atomicAdd(ticket, 1u);
        for (int d = t; d < R; d += OS_THREADS) {
#pragma unroll
            for (int w = 0; w < OS_WAVES; w++) wcnt[w][d] = 0;
        }
#if OS_RANK_LDS
        // the lane masks of the ranking loop live where the reordered tile goes afterwards: all zero at the start of a tile
        static_assert((size_t)OS_WAVES * R * 8 <= (size_t)OS_TILE * 4, "lane masks do not fit the tile buffer");
        for (int i = t; i < OS_WAVES * R * 2; i += OS_THREADS) keys_s[i] = 0u;
#endif
        __syncthreads();
        const long long tile = tile_s;
        if (tile >= ntiles) break;
        OS_STAMP(0);
        const long long tile0 = tile * OS_TILE;
        const int cnt_tile = (int)((n - tile0) < OS_TILE ? (n - tile0) : OS_TILE);
        const bool full = cnt_tile == OS_TILE;
        unsigned key[OS_ITEMS];
        unsigned short rnk[OS_ITEMS];
        // order of the keys inside the tile: wave, item, lane (= their order in the input: the sort is stable)
#pragma unroll
        for (int i = 0; i < OS_ITEMS; i++) {
            const int idx = wv * (64 * OS_ITEMS) + i * 64 + lane;
            if (SRC == 0) key[i] = (full || idx < cnt_tile) ? in_k[tile0 + idx] : 0xFFFFFFFFu;
            else key[i] = (full || idx < cnt_tile) ? f32_ord(in_f[tile0 + idx]) : 0xFFFFFFFFu;
        }
#ifdef NNC_DIAG
        if (g_os_trace) { if (key[OS_ITEMS - 1] == 0x12345678u && key[0] == 0x87654321u) keys_s[0] = 1u; OS_STAMP(1); } // (the stamp waits for the loads)
#endif
        if (nrb > 0) { // the next pass's histogram (LDS atomics: about a lane a clock, beside the ballots below)
#pragma unroll
            for (int i = 0; i < OS_ITEMS; i++) {
                const int idx = wv * (64 * OS_ITEMS) + i * 64 + lane;
                if ((full || idx < cnt_tile) && !(abl & 8)) atomicAdd(&nhist[(key[i] >> nshift) & nmask], 1u);
            }
        }
#pragma unroll
        for (int i = 0; i < OS_ITEMS; i++) {
            const int idx = wv * (64 * OS_ITEMS) + i * 64 + lane;
            const bool valid = full || idx < cnt_tile;
            const unsigned d = (key[i] >> shift) & MASK;
            unsigned plo, phi;
#if OS_RANK_LDS
            // every lane sets its bit in the mask of its digit, then reads the mask: the lanes of this item with the same digit (the
            // LDS serves a wave's instructions in order: the read sees the whole OR); the first of them clears the mask again
            unsigned long long *mrow = reinterpret_cast<unsigned long long *>(keys_s) + wv * R;
            if (valid) atomicOr(&mrow[d], 1ull << lane);
            __builtin_amdgcn_wave_barrier();
            const unsigned long long pm = valid ? mrow[d] : 0ull;
            plo = (unsigned)pm; phi = (unsigned)(pm >> 32);
            if (abl & 2) { plo = lane < 32 ? 1u << lane : 0u; phi = lane >= 32 ? 1u << (lane - 32) : 0u; }
#else
            if (abl & 2) { plo = lane < 32 ? 1u << lane : 0u; phi = lane >= 32 ? 1u << (lane - 32) : 0u; } // (diagnostics build only: no ranking)
            else if (full) os_peers<RB, true>(d, true, plo, phi);
            else os_peers<RB, false>(d, valid, plo, phi);
#endif
            const unsigned before = (unsigned)__builtin_amdgcn_mbcnt_hi(phi, __builtin_amdgcn_mbcnt_lo(plo, 0u));
            unsigned prev = 0;
            if (valid) {
                prev = wcnt[wv][d];                                            // (every lane of the group reads before its first lane writes)
                if (before == 0) {
                    wcnt[wv][d] = (unsigned short)(prev + (unsigned)__popc(plo) + (unsigned)__popc(phi));
#if OS_RANK_LDS
                    mrow[d] = 0ull;
#endif
                }
            }
            rnk[i] = (unsigned short)(prev + before);
            __builtin_amdgcn_wave_barrier(); // (the next item's reads of the wave's counters come after this item's writes)
        }
        __syncthreads();
        OS_STAMP(2);
        // per digit: the waves' offsets inside the digit and the tile's count; the tile's exclusive prefix over the digits
        unsigned tc[DPT], own = 0;
#pragma unroll
        for (int r = 0; r < DPT; r++) {
            const int d = t * DPT + r;
            tc[r] = 0;
            if (d < R) {
                unsigned run = 0;
#pragma unroll
                for (int w = 0; w < OS_WAVES; w++) { const unsigned c = wcnt[w][d]; wcnt[w][d] = (unsigned short)run; run += c; }
                tc[r] = run;
                // this tile's count is out before anything else waits: tile 0 has nothing in front of it
                os_publish(&status[(size_t)tile * R + d], tile == 0 ? tagP : tagA, run);
            }
            own += tc[r];
        }
        {
            unsigned sc = own;
            for (int off = 1; off < 64; off <<= 1) { const unsigned o = __shfl_up(sc, off); if (lane >= off) sc += o; }
            if (lane == 63) wsum[wv] = sc;
            __syncthreads();
            unsigned run = sc - own;
            for (int w = 0; w < wv; w++) run += wsum[w];
#pragma unroll
            for (int r = 0; r < DPT; r++) { const int d = t * DPT + r; if (d < R) tpre[d] = (unsigned short)run; run += tc[r]; }
        }
        __syncthreads();
        // the tile ordered by digit, in LDS (while the tiles in front publish)
#pragma unroll
        for (int i = 0; i < OS_ITEMS; i++) {
            const int idx = wv * (64 * OS_ITEMS) + i * 64 + lane;
            if (full || idx < cnt_tile) {
                const unsigned d = (key[i] >> shift) & MASK;
                keys_s[(unsigned)tpre[d] + (unsigned)wcnt[wv][d] + (unsigned)rnk[i]] = key[i];
            }
        }
        OS_STAMP(3);
        if (DST == 1 && n_zero > 0) {
            // the block of zeros of a pruned vector, a slice per tile: stores that need no answer, issued where the tile would
            // otherwise only wait for the tiles in front of it
            const long long per = ((n_zero + ntiles - 1) / ntiles + 3) & ~3ll;
            const long long z0 = tile * per, z1 = (z0 + per < n_zero) ? z0 + per : n_zero;
            float *zp = out_f + n_neg;
            if ((reinterpret_cast<uintptr_t>(zp) & 15) == 0) {
                for (long long i = z0 + 4ll * t; i + 3 < z1; i += 4ll * OS_THREADS) *reinterpret_cast<float4 *>(zp + i) = make_float4(0.f, 0.f, 0.f, 0.f);
                if (t == 0) for (long long i = z0 + ((z1 - z0) & ~3ll); i < z1; i++) zp[i] = 0.0f;
            } else {
                for (long long i = z0 + t; i < z1; i += OS_THREADS) zp[i] = 0.0f;
            }
        }
        // look-back: keys of this thread's digits in the tiles before this one, OS_LOOK status words in flight at a time
#pragma unroll
        for (int r = 0; r < DPT; r++) {
            const int d = t * DPT + r;
            if (d >= R) continue;
            unsigned excl = 0;
            if (tile > 0 && !(abl & 1)) { // (abl: diagnostics build only -- no look-back)
                long long i = tile - 1;
                int polls = 0;
                bool done = false;
                while (!done) {
                    unsigned long long w[OS_LOOK];
#pragma unroll
                    for (int j = 0; j < OS_LOOK; j++) w[j] = (i - j >= 0) ? os_peek(&status[(size_t)(i - j) * R + d]) : 0ull;
                    int used = 0;
#pragma unroll
                    for (int j = 0; j < OS_LOOK; j++) {
                        if (!done && used == j) {
                            const unsigned tg = (unsigned)(w[j] >> 32);
                            if (tg == tagP) { excl += (unsigned)w[j]; done = true; }
                            else if (tg == tagA) { excl += (unsigned)w[j]; used = j + 1; } // (tile 0 publishes a prefix: the walk ends there at the latest)
                        }
                    }
                    if (!done) {
                        i -= used;
                        if (used == 0) {
                            if (++polls > OS_SPIN_LIMIT) { gave_up = true; done = true; }
                            __builtin_amdgcn_s_sleep(1);
                        } else polls = 0;
                    }
                }
                os_publish(&status[(size_t)tile * R + d], tagP, excl + tc[r]);
            }
            gpos[d] = gdig[d] + excl - (unsigned)tpre[d];
        }
        OS_STAMP(4);
        __syncthreads();
        OS_STAMP(5);
        for (int j = t; j < cnt_tile; j += OS_THREADS) {
            const unsigned kv = keys_s[j];
            const unsigned d = (kv >> shift) & MASK;
            const unsigned r = gpos[d] + (unsigned)j;
            if ((long long)r < n && !(abl & 4)) { // (always, unless the caller's counts were wrong: then garbage, never a fault)
                if (DST == 0) out_k[r] = kv;
                else out_f[(long long)r + ((long long)r >= n_neg ? n_zero : 0)] = os_value(km, kv);
            }
        }
        OS_STAMP(6);
        __syncthreads(); // LDS is reused by the next tile
        OS_STAMP(7);
    }
    if (gave_up) atomicOr(ctrl + 1, 2ull);
    if (nrb > 0) { // (every thread is past the loop's last barrier: the workgroup's histogram is complete)
        for (int i = t; i <= (int)nmask; i += OS_THREADS) {
            const unsigned v = nhist[i];
            if (v) atomicAdd(&ghist_next[i], v);
        }
    }
}

static unsigned host_ord(float v)
{
    unsigned b;
    std::memcpy(&b, &v, 4);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

static size_t al256(size_t b) { return (b + 255) & ~(size_t)255; }

static int os_cus()
{
    static int cus[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (!cus[dev]) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        cus[dev] = v;
    }
    return cus[dev];
}

// digits of 8 or 9 bits only (two instances of the pass kernel): high digits may reach past the keys' bits, which are zero there
static OsPlan os_plan(int bits)
{
    OsPlan pl;
    std::memset(&pl, 0, sizeof(pl));
    int rb;
    if (bits <= 8) { pl.passes = 1; rb = 8; }
    else if (bits <= 16) { pl.passes = 2; rb = 8; }
    else if (bits <= 18) { pl.passes = 2; rb = 9; }
    else if (bits <= 24) { pl.passes = 3; rb = 8; }
    else if (bits <= 27) { pl.passes = 3; rb = 9; }
    else { pl.passes = 4; rb = 8; }
    for (int p = 0; p < pl.passes; p++) { pl.shift[p] = p * rb; pl.rb[p] = rb; }
    return pl;
}

// workspace: [keys A][keys B][ctrl][digit histograms][status words]; n_keys = keys to sort
static size_t os_status_bytes(int64_t n_keys)
{
    const int64_t ntiles = (n_keys + OS_TILE - 1) / OS_TILE;
    return al256((size_t)std::max<int64_t>(ntiles, 1) * OS_MAXR * 8);
}
static size_t os_ws_bytes(int64_t n_keys)
{
    return 2 * al256((size_t)n_keys * 4 + 16) + al256(OS_CTRL_WORDS * 8) + al256((size_t)(OS_MAXPASS + 1) * OS_MAXR * 4) + os_status_bytes(n_keys) + 256;
}
static unsigned char *os_ws_base(void *ws) { return reinterpret_cast<unsigned char *>((reinterpret_cast<uintptr_t>(ws) + 255) & ~(uintptr_t)255); }

// The whole sort.  skip_zeros: only the non-zero weights are keys (n_neg negative ones, n_zero zeros: the caller's counts), the zeros
// come back as one block between the negative and the positive values.  bits: significant bits of the keys under `km`.
static int os_sort(const float *x, int64_t n, bool skip_zeros, int64_t n_neg, int64_t n_zero, const OsMap &km, int bits, float *sorted_out, void *ws,
                   size_t ws_bytes, int32_t *flag_dev, const char *who, void *stream)
{
    const int64_t n_keys = skip_zeros ? n - n_zero : n;
    if (!ws || ws_bytes < os_ws_bytes(n_keys)) return nnc_set_error_(NNC_ENOSPACE, (std::string(who) + ": workspace too small").c_str());
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    unsigned char *b = os_ws_base(ws);
    unsigned *ka = reinterpret_cast<unsigned *>(b); b += al256((size_t)n_keys * 4 + 16);
    unsigned *kb = reinterpret_cast<unsigned *>(b); b += al256((size_t)n_keys * 4 + 16);
    unsigned long long *ctrl = reinterpret_cast<unsigned long long *>(b); b += al256(OS_CTRL_WORDS * 8);
    unsigned *ghist = reinterpret_cast<unsigned *>(b); b += al256((size_t)(OS_MAXPASS + 1) * OS_MAXR * 4);
    unsigned long long *status = reinterpret_cast<unsigned long long *>(b);
    hipError_t e;
    // ctrl + histograms + status words lie side by side: one memset
    const size_t zero_bytes = al256(OS_CTRL_WORDS * 8) + al256((size_t)(OS_MAXPASS + 1) * OS_MAXR * 4) + os_status_bytes(n_keys);
    if ((e = hipMemsetAsync(ctrl, 0, zero_bytes, s)) != hipSuccess) return nnc_set_error_(NNC_EHIP, hipGetErrorString(e));
    // (the zeros of a pruned vector are written by the tiles of the last pass; with no key at all there is no pass)
    if (skip_zeros && n_zero > 0 && n_keys == 0 && (e = hipMemsetAsync(sorted_out + n_neg, 0, (size_t)n_zero * 4, s)) != hipSuccess) return nnc_set_error_(NNC_EHIP, hipGetErrorString(e));
    const OsPlan pl = os_plan(bits);
    int abl = 0;
#ifdef NNC_DIAG
    { const char *ev = getenv("NNC_OS_ABLATE"); abl = ev ? atoi(ev) : 0; } // timing experiments (tools/time_sort.py): 1 no look-back, 2 no ranking, 4 no write-out
#endif
    if (n_keys > 0) {
        const long long ptiles = (n + OSP_TILE - 1) / OSP_TILE;
        const int pgrid = (int)std::min<long long>(std::max<long long>(ptiles, 1), 2LL * os_cus());
        if (skip_zeros) hipLaunchKernelGGL((k_os_prep<true>), dim3(pgrid), dim3(OSP_THREADS), 0, s, x, (long long)n, ka, ctrl, (long long)n_keys, km, pl.rb[0], ghist);
        else hipLaunchKernelGGL((k_os_prep<false>), dim3(pgrid), dim3(OSP_THREADS), 0, s, x, (long long)n, (unsigned *)nullptr, ctrl, (long long)n_keys, km, pl.rb[0], ghist);
        if ((e = hipGetLastError()) != hipSuccess) return nnc_set_error_(NNC_EHIP, hipGetErrorString(e));
        const long long ntiles = (n_keys + OS_TILE - 1) / OS_TILE;
        const int grid = (int)std::min<long long>(ntiles, (long long)OS_BLOCKS_PER_CU * os_cus());
        const void *src = skip_zeros ? (const void *)ka : (const void *)x;
        unsigned *dst = skip_zeros ? kb : ka;
        for (int p = 0; p < pl.passes; p++) {
            const bool first_f = (p == 0 && !skip_zeros), last = (p == pl.passes - 1);
            void *o = last ? (void *)sorted_out : (void *)dst;
#define OS_LAUNCH(SRC, DST) do { \
                if (pl.rb[p] == 9) hipLaunchKernelGGL((k_os_pass<9, SRC, DST>), dim3(grid), dim3(OS_THREADS), 0, s, src, o, (long long)n_keys, pl.shift[p], p, last ? 0 : pl.rb[p + 1], \
                                                      ghist + (size_t)p * OS_MAXR, ghist + (size_t)(p + 1) * OS_MAXR, status, ctrl, km, (long long)(skip_zeros ? n_neg : n_keys), (long long)(skip_zeros ? n_zero : 0), abl); \
                else hipLaunchKernelGGL((k_os_pass<8, SRC, DST>), dim3(grid), dim3(OS_THREADS), 0, s, src, o, (long long)n_keys, pl.shift[p], p, last ? 0 : pl.rb[p + 1], \
                                        ghist + (size_t)p * OS_MAXR, ghist + (size_t)(p + 1) * OS_MAXR, status, ctrl, km, (long long)(skip_zeros ? n_neg : n_keys), (long long)(skip_zeros ? n_zero : 0), abl); \
            } while (0)
            if (first_f && last) OS_LAUNCH(1, 1);
            else if (first_f) OS_LAUNCH(1, 0);
            else if (last) OS_LAUNCH(0, 1);
            else OS_LAUNCH(0, 0);
#undef OS_LAUNCH
            if ((e = hipGetLastError()) != hipSuccess) return nnc_set_error_(NNC_EHIP, hipGetErrorString(e));
            src = dst;
            dst = (dst == ka) ? kb : ka;
        }
    }
    if (flag_dev && (e = hipMemcpyAsync(flag_dev, reinterpret_cast<unsigned char *>(ctrl) + 8, 4, hipMemcpyDeviceToDevice, s)) != hipSuccess)
        return nnc_set_error_(NNC_EHIP, hipGetErrorString(e));
    return NNC_OK;
}

static OsMap os_identity()
{
    OsMap m;
    std::memset(&m, 0, sizeof(m));
    return m;
}

extern "C" size_t nnc_sort_workspace_bytes(int64_t n) { return n <= 0 ? 0 : os_ws_bytes(n); }

extern "C" int nnc_sort_f32(const float *x, int64_t n, float *sorted_out, void *ws, size_t ws_bytes, void *stream)
{
    if (n < 0 || n >= ((int64_t)1 << 31) || (n > 0 && (!x || !sorted_out))) return nnc_set_error_(NNC_EINVAL, "nnc_sort_f32: bad argument");
    if (n == 0) return NNC_OK;
    return os_sort(x, n, false, 0, 0, os_identity(), 32, sorted_out, ws, ws_bytes, nullptr, "nnc_sort_f32", stream);
}

// --------------------------------------------------------------------------------------
// The same for a PRUNED vector: most weights are exact zeros, which need no sorting: only the non-zeros are keys, the zeros come
// back as a memset between the negative and the positive values.  The counts come from nnc_minmax_signs_f32 (the fit set-up reads
// them together with min / max).  -0.0 counts as a zero and comes back as +0.0: equal as a value, which is all the iterations look at.
// --------------------------------------------------------------------------------------
extern "C" size_t nnc_sort_pruned_workspace_bytes(int64_t n, int64_t n_neg, int64_t n_zero)
{
    if (n <= 0 || n_neg < 0 || n_zero < 0 || n_neg + n_zero > n) return 0;
    return os_ws_bytes(n - n_zero);
}

extern "C" int nnc_sort_pruned_f32(const float *x, int64_t n, int64_t n_neg, int64_t n_zero, float *sorted_out, void *ws,
                                   size_t ws_bytes, void *stream)
{
    if (n < 0 || n_neg < 0 || n_zero < 0 || n_neg + n_zero > n || n >= ((int64_t)1 << 31) || (n > 0 && (!x || !sorted_out)))
        return nnc_set_error_(NNC_EINVAL, "nnc_sort_pruned_f32: bad argument");
    if (n == 0) return NNC_OK;
    return os_sort(x, n, true, n_neg, n_zero, os_identity(), 32, sorted_out, ws, ws_bytes, nullptr, "nnc_sort_pruned_f32", stream);
}

// --------------------------------------------------------------------------------------
// The sort of a PRUNED vector whose bounds are known: the surviving weights lie in [vmin, -thr] and [thr, vmax], so their
// order-preserving integer images, taken relative to the two ends, fit in far fewer than 32 bits (26 at a 1-sigma threshold):
// three radix passes of at most 9 bits instead of four.
// --------------------------------------------------------------------------------------
// bits of the compact keys, 0 if the bounded form does not apply (no threshold, bounds out of order, more than 27 bits)
static int rs_bounds(float vmin, float vmax, float thr, int64_t n_neg, int64_t n_pos, OsMap *bd)
{
    if (!(thr > 0.0f) || !std::isfinite(thr) || !std::isfinite(vmin) || !std::isfinite(vmax)) return 0;
    OsMap b;
    std::memset(&b, 0, sizeof(b));
    b.bounded = 1;
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
    int bits = 1;
    while ((1ull << bits) < total) bits++;
    *bd = b;
    return bits;
}

extern "C" int32_t nnc_sort_pruned_bounded_bits(float vmin, float vmax, float thr, int64_t n_neg, int64_t n_pos)
{
    OsMap b;
    return rs_bounds(vmin, vmax, thr, n_neg, n_pos, &b);
}

extern "C" size_t nnc_sort_pruned_bounded_workspace_bytes(int64_t n_nz)
{
    if (n_nz < 0) return 0;
    return os_ws_bytes(n_nz);
}

int nnc_sort_pruned_bounded_flagged_(const float *x, int64_t n, int64_t n_neg, int64_t n_zero, float vmin, float vmax, float thr,
                                     float *sorted_out, void *ws, size_t ws_bytes, int32_t *flag_dev, void *stream);
extern "C" int nnc_sort_pruned_bounded_f32(const float *x, int64_t n, int64_t n_neg, int64_t n_zero, float vmin, float vmax, float thr,
                                           float *sorted_out, void *ws, size_t ws_bytes, void *stream)
{
    return nnc_sort_pruned_bounded_flagged_(x, n, n_neg, n_zero, vmin, vmax, thr, sorted_out, ws, ws_bytes, nullptr, stream);
}

// device int32 inside the workspace: non-zero after the sort iff a weight lay outside [vmin, -thr] u {0} u [thr, vmax] (or was NaN)
// (bit 0), or a look-back of the sort gave up (bit 1: the sorted vector is then not to be used)
extern "C" const int32_t *nnc_sort_pruned_bounded_flag(void *ws, int64_t n_nonzero)
{
    if (!ws || n_nonzero < 0) return nullptr;
    unsigned char *b = os_ws_base(ws) + 2 * al256((size_t)n_nonzero * 4 + 16);
    return reinterpret_cast<const int32_t *>(b + 8); // (the low half of the 64-bit flag word behind the counter)
}

// (flag_dev: where the verdict goes as well -- the layer call keeps it next to the scalars it reads back anyway)
int nnc_sort_pruned_bounded_flagged_(const float *x, int64_t n, int64_t n_neg, int64_t n_zero, float vmin, float vmax, float thr,
                                     float *sorted_out, void *ws, size_t ws_bytes, int32_t *flag_dev, void *stream)
{
    if (n < 0 || n_neg < 0 || n_zero < 0 || n_neg + n_zero > n || n >= ((int64_t)1 << 31) || (n > 0 && (!x || !sorted_out)))
        return nnc_set_error_(NNC_EINVAL, "nnc_sort_pruned_bounded_f32: bad argument");
    if (n == 0) return NNC_OK;
    const int64_t n_pos = n - n_neg - n_zero, n_nz = n_neg + n_pos;
    OsMap bd;
    std::memset(&bd, 0, sizeof(bd));
    bd.bounded = 1;
    const int bits = n_nz > 0 ? rs_bounds(vmin, vmax, thr, n_neg, n_pos, &bd) : 1;
    if (bits == 0) return nnc_set_error_(NNC_EINVAL, "nnc_sort_pruned_bounded_f32: bounds do not apply (see nnc_sort_pruned_bounded_bits)");
    return os_sort(x, n, true, n_neg, n_zero, bd, bits, sorted_out, ws, ws_bytes, flag_dev, "nnc_sort_pruned_bounded_f32", stream);
}
