// nnc_km_shared.hpp -- what the k-means translation units share (internal).
#pragma once
#define KM_THREADS 1024
#define KL_RKEYS 16384 // candidate keys an empty-cluster event inside the one-workgroup loop may have (16 per thread)
#define KM_NSHARD 8
#define KM_GMAX 32768
#define KM_CNT_SAT 31
#define KM_P_BITS 11 // cell entry: first candidate (sorted position, < 2048) | min(count-1, 31) << 11
#define KM_P_MASK 2047u
#define KM_OVF_MAX 1024     // cells with a saturated count keep their exact candidate range in a side list
#define KM_OVF_ALL 2047u    // ... or, if even that list is full, scan every centre
// Few centres and a small grid: k_finalize builds the cell table itself (its one busy wave plus fifteen helper waves
// that sleep until the zones are known), which saves the k_cells launch where a launch is a third of the iteration.
#define KM_FUSE_GLOG2 11
#define KM_FUSE_KMAX 64
#ifndef KM_RING
#define KM_RING 4 // float4 loads kept in flight per thread
#endif

struct KmTab {
    float2 cand[NNC_KMAX];   // sorted: (c~, fl(c~*c~))
    uint16_t orig[NNC_KMAX]; // sorted position -> original centroid index (lowest index among equal centres)
    uint16_t perm[NNC_KMAX]; // the full sorted permutation of all k centres (duplicates included)
    uint32_t ovf[KM_OVF_MAX]; // crowded cells (more than 31 candidates): first | last << 16, indexed by the cell entry
    int32_t n_ovf;
    int32_t ku;              // number of DISTINCT centre values = entries of cand/orig; equal centres never win (ties go to the lowest index)
    int32_t pad_[2];
    uint16_t cell[KM_GMAX];  // p_lo | (min(cnt-1, 31) << 11)
    double zl[NNC_KMAX], zr[NNC_KMAX]; // zone of every distinct centre: outside [zl, zr] (centred x) it cannot be the float32 arg-min
    int32_t gc[NNC_KMAX], hc[NNC_KMAX]; // the same in cells (monotone): centre p can open cells <= gc[p], close cells >= hc[p]; k_cells turns them into cell[]
};

struct KmWs {
    nnc_kmeans_status st;
    nnc_kmeans_params p;
    int32_t cur;       // which KmTab / centre set is current
    int32_t glog2, rlog2;
    int32_t reloc_fail; // the windowed farthest-sample selection could not prove its result: redo it the long way
    int32_t spec_go;    // the relocation chain enqueued behind an iteration "in case" has an event to settle (k_reloc_windows decides)
    float inv;         // cells per unit: cell = (int)((x~ - lo) * inv)
    int32_t cells_pending; // k_finalize left new zones: k_cells has to rebuild tab[cur].cell
    int32_t ku_cur;    // = tab[cur].ku, here so that k_cells learns it in its first round of loads
    float pad1;
    float c[2][NNC_KMAX];        // centred centres in ORIGINAL index order; [cur] current, [cur^1] previous
    long long partials[2 * NNC_KMAX]; // sums then counts, original index order (all-reduced across ranks)
    long long partials_local[2 * NNC_KMAX]; // this rank's own sums/counts of the last accumulated iteration
    long long prev_counts[NNC_KMAX];        // label counts of the previous iteration (before any relocation edit)
    long long shard_sum[KM_NSHARD][NNC_KMAX]; // sorted index order of tab[cur]
    unsigned long long shard_cnt[KM_NSHARD][NNC_KMAX];
    // k_bounds: long stretches of samples whose cluster float32 cannot tell from the zones alone, cut into tiles any wave
    // may take (two self-validating words per record, see k_bounds); emptied by the kernel that consumes the sums
    // what k_bounds needs of the CURRENT table, at an address that does not depend on which of the two tables is current
    // (one round of loads less at the head of every iteration); k_finalize writes it next to tab[cur]
    struct Bnd { int32_t ku, pad; double zr[NNC_KMAX], zl[NNC_KMAX]; float2 cand[NNC_KMAX]; uint16_t orig[NNC_KMAX]; } bnd;
    int32_t q_n, q_searched; // records published; waves of an announced pass (help_hint) that are through their searches -- whatever they had to publish is out
    int32_t help_hint, help_pad; // the previous pass published long stretches: this one had better look at the queue (k_finalize sets it)
    long long hint_a[NNC_KMAX], hint_b[NNC_KMAX]; // where k_bounds found boundary j last time: the next search starts there
    unsigned long long q_w0[NNC_KMAX], q_w1[NNC_KMAX];
    int32_t q_next[NNC_KMAX];
    // k_lloyd (the one-workgroup loop, nnc_lloyd.hpp): `wide` = the next iteration needs the multi-workgroup pass (centres closer
    // than float32 can tell apart, a search that did not settle): the k_bounds / k_finalize pair enqueued behind the loop "in case"
    // runs only then and clears it; kl_budget = iterations the launches of the current host call may still run
    int32_t wide, kl_budget;
    unsigned long long kl_trace[24]; // diagnostics build: time per phase of k_lloyd (10 ns ticks), summed over the fit
    int32_t kl_stats[8]; // [0] iterations the loop ran, [1] its launches, [2] of them with centres changing places, [3] iterations handed over, [4] iterations the wide pair ran, [5] empty-cluster events the loop settled itself
    unsigned long long kl_keys[KL_RKEYS]; // candidate keys of such an event (kl_relocate)
    int32_t bnd_phi[NNC_KMAX]; // per boundary j of the last k_bounds pass: the highest centre that could still win below U_j (the finalize step labels the undecided samples of an empty-cluster event with it, km_finalize_relocate)
    unsigned long long kl_mkeys[40960]; // candidate keys of a MASS empty-cluster event settled by the finalize step (kl_relocate_mass, KL_MKEYS)
    float kl_hL[2 * NNC_KMAX], kl_hR[2 * NNC_KMAX]; // per search (two a boundary: its ranks hint_a / hint_b): the threshold the rank was found for, the local density there (samples per unit)
    KmTab tab[2];
};

// modes of the finalize step (km_launch_finalize)
#define FIN_INIT 0          // build the table for the initial centres
#define FIN_FROM_SHARDS 1   // single GPU: reduce shards -> partials -> finalize
#define FIN_FROM_PARTIALS 2 // multi GPU / resume: partials already hold the global sums
#define FIN_PACK_ONLY 3     // reduce shards -> partials, nothing else

// a value every lane holds alike, moved to scalar registers (so that the control flow that depends on it is scalar)
__device__ __forceinline__ int uni_i(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ long long uni_ll(long long v)
{
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v & 0xFFFFFFFFll));
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((unsigned long long)v >> 32));
    return (long long)(((unsigned long long)hi << 32) | lo);
}

// host-side pieces of the k-means path that more than one translation unit uses (defined in nnc_hip.hip)
int km_check(void *ws, const nnc_kmeans_params *p, const char *who);
int km_set_lds_attr();
int km_launch_accumulate(const float *x, KmWs *w, const nnc_kmeans_params *p, void *stream, int which = 0);
// reloc_xs: the value-sorted vector, for callers that want the finalize step of a rank-boundary iteration to settle small
// empty-cluster events itself (km_finalize_relocate: nnc_kmeans_fit); nullptr: every event pauses (what nnc_kmeans_iterate shows)
int km_launch_finalize(KmWs *w, const nnc_kmeans_params *p, int mode, int resume, void *stream, void *host_mapped = nullptr,
                       uint64_t ticket = 0, bool cond = false, const float *reloc_xs = nullptr);
int km_wait_ticket(volatile unsigned long long *word, unsigned long long ticket, hipStream_t stream);
static inline size_t reloc_align(size_t b) { return (b + 255) & ~(size_t)255; }
