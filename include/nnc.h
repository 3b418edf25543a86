/*
 * nnc.h -- C ABI of libnnc_hip.so: the MI355X (gfx950) implementation of the per-layer
 * Deep-Compression hot path of angelocatalani/neural-network-compression:
 *
 *     magnitude-threshold pruning  ->  Lloyd k-means weight quantisation  ->
 *     centroid-index re-encode (+ index histogram / Huffman code lengths)
 *
 * The reference has no FFI of its own (it is ~870 lines of Python on NumPy /
 * scikit-learn); the boundary it offers is three module-level functions and the Trainer
 * methods that call them.  Each entry point below names the reference lines it replaces
 * (paths relative to the reference repo root); INTEGRATION.md shows the ctypes binding a
 * maintainer of the reference would add.
 *
 * Conventions
 *   - one calling thread per (device, stream) at a time: all state lives in caller-owned workspaces; the few process-wide
 *     caches (CU count, function attributes) are per device and thread safe, the bench profiler's event pool is locked;
 *   - every function returns 0 on success, a negative NNC_E* code on failure;
 *     nnc_last_error() gives the message of the calling thread's last failure;
 *   - pointers named *_dev / x / mask / labels are DEVICE pointers (hipMalloc'ed or
 *     torch.Tensor.data_ptr()); nothing is allocated behind the caller's back: scratch is
 *     a caller-owned device workspace sized by the *_workspace_bytes() functions;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); all work is
 *     enqueued asynchronously on it, no entry point synchronises unless it says so;
 *   - float32 arithmetic that decides a mask bit or a centroid index reproduces the
 *     reference's NumPy / scikit-learn float32 arithmetic exactly (no FMA contraction).
 *
 * Fixed-point sums ("mode B").  Per-cluster sums are exact 64-bit integer sums of
 *     fix(v) = rint(v * 2^S)   (nearest integer, ties to even; v = float32 centred weight)
 * so that they do not depend on summation order, block count or GPU count; the new
 * centre is (float) ldexp((double)sum / (double)count, -S).  S = min(28, 62 - L) - P with
 * L = ceil(log2(n_total)) and 2^P > max|v| (nnc_fix_shift()).
 */
#ifndef NNC_H
#define NNC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NNC_VERSION 100

#define NNC_OK 0
#define NNC_EINVAL (-1)   /* bad argument (null pointer, size, k out of range ...) */
#define NNC_ENOSPACE (-2) /* workspace too small */
#define NNC_EHIP (-3)     /* a HIP runtime call failed; see nnc_last_error() */
#define NNC_ENODEV (-4)   /* no gfx950 device / wrong architecture */

#define NNC_KMAX 1040     /* largest supported number of centroids (2^10 + 1 for density init) */
#define NNC_CHUNK 8192    /* NumPy's reduction buffer: float32 sums are folded per 8192-element chunk */

int nnc_version(void);
const char *nnc_last_error(void);
/* Fills name (e.g. "gfx950") and the CU count of the current device. */
int nnc_device_info(char *arch_out, size_t arch_len, int *cu_count_out);

/* ------------------------------------------------------------------------------------
 * NumPy-exact float32 reductions (np.sum / np.mean / np.var / np.std on float32):
 * replaces the np.std call in prune_weigth (neural_network_compression/common/utility.py:159)
 * and the X.mean / np.var calls scikit-learn's KMeans.fit makes on the weights
 * (utility.py:237-238 -> sklearn/cluster/_kmeans.py:279-287,1479-1484).
 * ---------------------------------------------------------------------------------- */

/* chunk_out[c] = NumPy pairwise sum of x[c*8192 : (c+1)*8192]           (sqdev == 0)
 *              = same tree over (x[i] - *mean_dev)^2 in float32          (sqdev != 0)
 * nchunks = ceil(n / 8192).  Shards of a longer vector must start on a multiple of 8192. */
int nnc_chunk_sums_f32(const float *x, int64_t n, int sqdev, const float *mean_dev, float *chunk_out,
                       void *stream);

#define NNC_FOLD_SUM 0  /* out = fold                                   */
#define NNC_FOLD_MEAN 1 /* out = (float)((double)fold / count)           */
#define NNC_FOLD_STD 2  /* out = sqrtf((float)((double)fold / count))    */
/* Left-to-right float32 fold of nchunks chunk sums (NumPy's order), then `op`.
 * out_dev[0] = result; if scale_dev != NULL additionally out_dev[1] = result * *scale_dev. */
int nnc_fold_f32(const float *chunks, int64_t nchunks, int64_t count, int op, const float *scale_dev,
                 float *out_dev, void *stream);

/* ------------------------------------------------------------------------------------
 * Pruning: utility.prune_weigth (common/utility.py:134-163), and the re-application of
 * stored masks, Trainer._reset_pruned_parameters (common/trainer.py:195-206).
 * ---------------------------------------------------------------------------------- */

size_t nnc_prune_workspace_bytes(int64_t n);

/* mask[i] = |x[i]| < thr ; x[i] = 0 where mask ; thr = std(x) * q if std_smooth else q.
 * stats_dev (device float[2]) receives {sigma, thr}; nzeroed_dev (device int64) the number of
 * mask bits set.  q is the float32 the NumPy comparison would use (host resolves NEP 50). */
int nnc_prune_f32(float *x, int64_t n, float q, int std_smooth, uint8_t *mask, float *stats_dev,
                  int64_t *nzeroed_dev, void *ws, size_t ws_bytes, void *stream);

/* nnc_prune_f32 and, in the same pass over the vector, what nnc_minmax_signs_f32 would report of the PRUNED tensor
 * (minmax4_dev = {min, max, min over the non-zeros, max over the non-zeros}, signs_dev = {#negative, #zero}): the statistics the
 * sort and the k-means set-up need next, without reading the tensor again.  ws: nnc_prune_stats_workspace_bytes(n). */
size_t nnc_prune_stats_workspace_bytes(int64_t n);
int nnc_prune_stats_f32(float *x, int64_t n, float q, int std_smooth, uint8_t *mask, float *stats_dev, int64_t *nzeroed_dev,
                        float *minmax4_dev, int64_t *signs_dev, void *ws, size_t ws_bytes, void *stream);

/* Same elementwise pass with the threshold already on the device (sharded pruning: the
 * ranks all-gather their chunk sums, fold, and each thresholds its own shard). */
int nnc_threshold_mask_f32(float *x, int64_t n, const float *thr_dev, uint8_t *mask, int64_t *nzeroed_dev,
                           void *stream);

/* x[i] = 0 where mask[i]  (trainer.py:203-205). */
int nnc_apply_mask_f32(float *x, const uint8_t *mask, int64_t n, void *stream);

/* ------------------------------------------------------------------------------------
 * Weight distribution: the data passes of utility.get_weight_distribution
 * (common/utility.py:362-372) and of the linear init (utility.py:207-208).
 * ---------------------------------------------------------------------------------- */

size_t nnc_minmax_workspace_bytes(int64_t n);
/* out_dev[0] = min, out_dev[1] = max over the elements (over the non-zero ones if
 * skip_zeros: Trainer.quantize strips exact zeros first, trainer.py:55-59);
 * count_dev = number of elements considered. */
int nnc_minmax_f32(const float *x, int64_t n, int skip_zeros, float *out_dev, int64_t *count_dev, void *ws,
                   size_t ws_bytes, void *stream);

/* One pass for everything the pipeline wants to know about a (pruned) vector: out_dev[0..1] =
 * min / max over all elements, out_dev[2..3] = min / max over the non-zero ones (+inf / -inf if
 * there are none), signs_dev[0] = #{x < 0}, signs_dev[1] = #{x == 0} (what nnc_sort_pruned_f32
 * needs).  out_dev holds 4 floats. */
int nnc_minmax_signs_f32(const float *x, int64_t n, float *out_dev, int64_t *signs_dev, void *ws, size_t ws_bytes,
                         void *stream);

/* The pruned sort when the bounds of the surviving weights are known (vmin <= x <= -thr or thr <= x <= vmax for every non-zero
 * x, as after nnc_prune_f32 with threshold thr; vmin / vmax from nnc_minmax_signs_f32): their order-preserving integer images
 * relative to the two ends fit in nnc_sort_pruned_bounded_bits(...) bits (0: the form does not apply -- no threshold, or more
 * than 27 bits), and a hand-written radix sort of those compact keys takes three passes of at most 9 bits instead of the four
 * 8-bit passes of the general 32-bit sort.  Same result as nnc_sort_pruned_f32.  Every non-zero weight must lie in [vmin, -thr] or
 * [thr, vmax]: a weight outside (or a NaN) is clamped to the nearest end -- nothing is written out of range, but the sorted
 * vector then holds a value the input does not -- and the int32 at nnc_sort_pruned_bounded_flag(ws, n_nonzero) (device memory inside
 * the workspace) is non-zero once the sort has run; nnc_compress_layer_f32 checks it and hands such a tensor to its caller's own
 * path (status NNC_LAYER_HOST). */
int32_t nnc_sort_pruned_bounded_bits(float vmin, float vmax, float thr, int64_t n_neg, int64_t n_pos);
size_t nnc_sort_pruned_bounded_workspace_bytes(int64_t n_nonzero);
const int32_t *nnc_sort_pruned_bounded_flag(void *ws, int64_t n_nonzero);
int nnc_sort_pruned_bounded_f32(const float *x, int64_t n, int64_t n_neg, int64_t n_zero, float vmin, float vmax, float thr,
                                float *sorted_out, void *ws, size_t ws_bytes, void *stream);

/* The statistics of one whole vector by one call (what np.mean / np.var of the reference's KMeans set-up,
 * sklearn _kmeans.py:1011 / 286, and the pass above deliver one by one): out6_dev = {mean, variance (both
 * NumPy-exact float32, as nnc_chunk_sums_f32 + nnc_fold_f32), min, max, min over the non-zeros, max over the
 * non-zeros}, signs_dev = {#negative, #zero}.  n > 0; ws: nnc_layer_stats_workspace_bytes(n) bytes, 256-byte aligned. */
size_t nnc_layer_stats_workspace_bytes(int64_t n);
int nnc_layer_stats_f32(const float *x, int64_t n, float *out6_dev, int64_t *signs_dev, void *ws, size_t ws_bytes,
                        void *stream);

/* ranks_out_dev[i] = #{ j : x_sorted[j] < values_dev[i] } for an ascending x_sorted: the histogram of
 * get_weight_distribution (utility.py:366-372) from the value-sorted copy, as differences of the ranks of the 32 steps. */
int nnc_rank_sorted_f32(const float *x_sorted, int64_t n, const float *values_dev, int32_t m, int64_t *ranks_out_dev, void *stream);
/* counts_dev[b] += #{ i : steps[b] <= x[i] < steps[b+1] }, b = 0..30 (caller zeroes counts_dev). */
int nnc_hist31_f32(const float *x, int64_t n, int skip_zeros, const float *steps32_dev, int64_t *counts_dev,
                   void *stream);

/* ------------------------------------------------------------------------------------
 * Lloyd k-means on the flattened weights: KMeans(n_clusters=K, init=space, n_init=1,
 * algorithm="full").fit(w.reshape(-1,1)) and the gather cluster_centers_[labels_]
 * (common/utility.py:237-239; scikit-learn's _kmeans.py:624-752, _k_means_lloyd.pyx:23-218,
 * _k_means_common.pyx:167-311).
 *
 * One iteration = nnc_kmeans_accumulate (E-step + per-cluster fixed-point sums over this
 * rank's shard) -> [all-reduce of nnc_kmeans_partials() across ranks] -> nnc_kmeans_finalize
 * (average, centre shift, tolerance test, search table for the next E-step).
 * nnc_kmeans_iterate enqueues `iters` such iterations back to back for the single-GPU case.
 * All state lives in the caller's workspace; once the state says done (or paused) further
 * iterations are no-ops, so the host may enqueue batches and look at the status afterwards.
 * The entry points that size a launch take the same (host) nnc_kmeans_params the workspace
 * was initialised with.
 * ---------------------------------------------------------------------------------- */

/* One-time ascending reorder of the vector for the iterations (nnc_kmeans_accumulate /
 * nnc_kmeans_iterate accept any order: integer sums are order independent; on a sorted copy a
 * lane sees runs of equal cluster index and adds them up in registers).  Hand-written radix sort
 * (csrc/nnc_sort.hip: four passes of 8 bits over the ordered images of the floats, decoupled look-back; no library sort);
 * not part of the per-iteration path.  nnc_kmeans_assign must get the original.  n < 2^31. */
size_t nnc_sort_workspace_bytes(int64_t n);
int nnc_sort_f32(const float *x, int64_t n, float *sorted_out, void *ws, size_t ws_bytes, void *stream);
/* The same for a pruned vector with n_neg negative and n_zero zero elements (host counts, e.g.
 * from nnc_minmax_signs_f32): the zeros are not sorted, only partitioned out and filled back
 * (-0.0 comes back as +0.0).  Any other count gives an unspecified order (never a fault). */
size_t nnc_sort_pruned_workspace_bytes(int64_t n, int64_t n_neg, int64_t n_zero);
int nnc_sort_pruned_f32(const float *x, int64_t n, int64_t n_neg, int64_t n_zero, float *sorted_out, void *ws,
                        size_t ws_bytes, void *stream);

typedef struct nnc_kmeans_params {
    int64_t n;         /* length of this rank's shard */
    int64_t n_total;   /* length of the whole vector (all ranks) */
    int32_t k;         /* number of centroids, 1..NNC_KMAX */
    int32_t max_iter;  /* scikit-learn default 300 */
    int32_t fix_shift; /* S of the fixed-point sums, from nnc_fix_shift() */
    int32_t grid_log2; /* log2 of the number of cells of the search grid; 0 = library default */
    int32_t replicas_log2; /* log2 of LDS accumulator replicas; -1 = library default */
    int32_t flags;     /* 0 (the library chooses), NNC_KM_TWO_LAUNCH or NNC_KM_LOOP, | NNC_KM_MASS_IN_PLACE */
    float x_mean;      /* NumPy float32 mean of the whole vector */
    float tol;         /* float32(np.var(x)) * float32(1e-4) */
    float lo, hi;      /* min and max of the centred data x - x_mean (float32) */
    const int64_t *prefix_dev; /* NULL: the vector handed to the iterations is in any order (streaming pass over all of it).
                          Else it is sorted ascending (nnc_sort_f32) and prefix_dev the block prefix sums built from it by
                          nnc_kmeans_prefix_build: an iteration then only looks up the K - 1 cluster boundaries (below) */
} nnc_kmeans_params;

/* flags: iterate launch by launch (k_bounds + k_finalize per iteration, or k_fit_small for few centres) even where the
 * one-workgroup loop (below, "The Lloyd loop in one workgroup") applies: for comparison and for the tests that pin the two forms
 * to each other; same results. */
#define NNC_KM_TWO_LAUNCH 1
/* ... and the other way round: the one-workgroup loop whatever the number of centres (by itself the library takes it up to
 * NNC_KM_LOOP_KMAX centres, where one compute unit's instruction rate is not yet the bound: measured on 0.1 M - 25 M weights it
 * takes 0.44 - 0.64 of the launch-per-iteration time up to K = 32, about the same at K = 65, 1.03 - 1.09 of it at K = 129 and about
 * 1.4 at K = 257). */
#define NNC_KM_LOOP 2
#define NNC_KM_MASS_IN_PLACE 4 /* (with the launch-per-iteration form, more than 64 centres) nnc_kmeans_fit lets the finalize step settle MASS
                               * empty-cluster events itself as well instead of enqueuing the relocation chain "in case": same trajectory;
                               * measured slower on the bench fit, kept for experiments and covered by tests */
#define NNC_KM_LOOP_KMAX 64

typedef struct nnc_kmeans_status {
    int32_t iter;      /* completed Lloyd iterations (scikit-learn's n_iter_ when done) */
    int32_t done;      /* 1: centre shift <= tol, 2: max_iter reached, 3: set by host (strict) */
    int32_t paused;    /* 1: an empty cluster was found; host must relocate, then resume.  2: a windowed
                          relocation could not be proven (nnc_kmeans_relocate_checked): redo it in full */
    int32_t n_empty;   /* number of empty clusters when paused */
    float shift_tot;   /* last sum of squared centre shifts (float32, NumPy order) */
    float tol;
    int32_t k;
    int32_t same_counts; /* 1: every cluster has as many members as in the previous iteration (labels MAY be equal) */
    int32_t reloc_ties;  /* relocation events in which two DIFFERENT samples tied (equal float32 distance) at the selection
                            cut: scikit-learn keeps whichever numpy.argpartition's introselect leaves there
                            (_k_means_common.pyx:186-187), this library the larger value -- the fits may part ways */
    int32_t reloc_multi; /* relocation events with more than one empty cluster: which far sample goes to which empty cluster
                            is the order numpy.argpartition leaves (implementation defined); here: descending distance */
    int32_t n_relocated; /* relocation events the device settled without the host (nnc_kmeans_fit enqueues the windowed relocation
                            behind the iterations of a batch in case they pause; it does nothing when they do not) */
    int32_t n_unproven;  /* ... and the events such a chain met but could not prove (the fit then pauses with paused == 2) */
    int32_t n_in_place;  /* of n_relocated: the events settled inside an iteration's own launch (the resident loop, or the finalize
                            step of a rank-boundary pass: small events, no relocation chain needed) */
    int32_t reserved;
} nnc_kmeans_status;

int32_t nnc_fix_shift(float absmax, int64_t n_total);
size_t nnc_kmeans_workspace_bytes(int32_t k);

/* Rank-boundary form of the iteration on a value-sorted vector.  In one dimension a cluster is a stretch of the sorted
 * vector; scikit-learn's float32 arg-min (_k_means_lloyd.pyx:196-213) can deviate from "nearest centre" only inside a
 * narrow zone around each midpoint (the bound of include/nnc.h's E-step note).  So per iteration every boundary is located by
 * a 64-ary search (one wave per boundary), the sums of the stretches that are certain come from block prefix sums of the
 * fixed-point images (differences of exact integers: the same sums as adding the members one by one), and only the samples
 * inside a zone are evaluated with the exact float32 expression.  Same labels, same sums, O(K log N) instead of O(N) reads.
 * prefix_dev: nnc_kmeans_prefix_bytes(n) bytes (per-block prefixes inside groups of 1024 blocks, then the group prefixes); build once per fit, after nnc_kmeans_init's x_mean / fix_shift are known.
 * x_sorted must be 16-byte aligned. */
#define NNC_PREFIX_BLOCK 256
/* The Lloyd loop in one workgroup (csrc/nnc_lloyd.hpp, k_lloyd): with prefix_dev set, the whole vector on one GPU
 * (n == n_total) and at most NNC_KM_LOOP_KMAX centres (or flags = NNC_KM_LOOP), nnc_kmeans_iterate / nnc_kmeans_iterate_publish /
 * nnc_kmeans_fit run the iterations inside ONE resident workgroup -- centres, zones, sums and counts in LDS; a rank = a Newton
 * step on the rank function from where it was last time, read as 128 samples + two 8-byte fine prefixes -- instead of two launches
 * per iteration; an empty cluster ends the launch (status.paused, as always) and an iteration with more than a million undecided
 * samples is run by the k_bounds / k_finalize pair enqueued behind every launch of the loop.  Bit-identical trajectory.  The buffer
 * therefore also holds the fine prefixes (one int64 per 64 samples) behind the block / group prefixes.
 * nnc_kmeans_iterate(iters) on this path enqueues min(iters, 32) rounds, each of which runs at least one iteration and all of
 * which together run at most `iters`: a call with iters > 32 may come back with FEWER than `iters` iterations run (every round that
 * ends in a hand-over to the wide pair, or in an event, uses up one of the 32) -- the contract is "at most iters, at least
 * min(iters, 32) unless the fit stops or pauses"; read status.iter and call again for the rest. */
size_t nnc_kmeans_prefix_bytes(int64_t n);
/* Where the iterations of the fit ran so far (device counters, reset by nnc_kmeans_init): out8[0] iterations run by the
 * one-workgroup loop, [1] launches of it that had work, [2] iterations in which centres changed places, [3] iterations it handed
 * to the wide pair (a very long undecided stretch, a search that did not settle), [4] iterations the wide pair ran, [5..7]
 * diagnostics.  Synchronises the stream. */
int nnc_kmeans_loop_stats(void *ws, int32_t *out8, void *stream);
int nnc_kmeans_prefix_build(const float *x_sorted, const nnc_kmeans_params *p, int64_t *prefix_dev, void *stream);

/* centers_init_dev: k float32, un-centred (the reference's `space`).  Resets the state. */
int nnc_kmeans_init(void *ws, size_t ws_bytes, const nnc_kmeans_params *p, const float *centers_init_dev,
                    void *stream);
/* Replaces the current centres by centers_dev[k] (centred != 0: values with x_mean already subtracted, taken as they
 * are; else un-centred, x_mean is subtracted in float32 as KMeans.fit does with `init`, _kmeans.py:1484), clears done /
 * paused and rebuilds the search tables; the iteration count stays.  For warm starts, for stepping through a recorded
 * trajectory one iteration at a time (tests against the reference's per-iteration fixtures) and for writing back
 * fine-tuned centroids (nnc_centroid_grad_f32). */
int nnc_kmeans_set_centers(void *ws, const nnc_kmeans_params *p, const float *centers_dev, int centred, void *stream);
int nnc_kmeans_accumulate(const float *x, void *ws, const nnc_kmeans_params *p, void *stream);
/* device int64[2*k]: sums (fixed point) then counts, indexed by centroid; valid after
 * nnc_kmeans_accumulate; the caller may all-reduce (SUM) or edit it before finalize. */
int64_t *nnc_kmeans_partials(void *ws);
/* resume = 0: normal; if a cluster is empty, set paused and change nothing else.
 * resume = 1: clear paused and finalize with the (host-edited) partials as they are. */
int nnc_kmeans_finalize(void *ws, int resume, void *stream);
int nnc_kmeans_iterate(const float *x, void *ws, const nnc_kmeans_params *p, int32_t iters, void *stream);
/* Asynchronous copy of the status block to host_out (pinned or pageable host memory). */
int nnc_kmeans_status_async(void *ws, nnc_kmeans_status *host_out, void *stream);
/* The same without a copy command and a stream synchronisation: a one-thread kernel writes the
 * status block to host_mapped (host memory the device can write: hipHostMalloc / pinned, 8-byte
 * aligned, sizeof(nnc_kmeans_status) + 8 bytes) and then `ticket` into the 8 bytes behind it;
 * the host polls that word until it reads its ticket, then reads the status. */
int nnc_kmeans_status_publish(void *ws, void *host_mapped, uint64_t ticket, void *stream);
/* nnc_kmeans_iterate with the look-in attached to the last launch of the batch (no launch of its
 * own): status block and ticket are written to host_mapped as by nnc_kmeans_status_publish.
 * With few centres (k <= 64) on a sorted vector with prefix sums (p->prefix_dev) both run the whole batch as ONE launch of
 * one workgroup that iterates until the fit stops, pauses or `iters` iterations are through: pass max_iter and look once. */
int nnc_kmeans_iterate_publish(const float *x, void *ws, const nnc_kmeans_params *p, int32_t iters, void *host_mapped,
                               uint64_t ticket, void *stream);
/* The whole Lloyd loop of one fit on one GPU as one call (what the host would do between two launches: batches of
 * iterations sized from the decay of the centre shift, look-ins, windowed relocation of empty clusters), returning when the
 * fit has stopped (status_out->done) or needs the caller: status_out->paused == 2 (a windowed selection could not be proven:
 * full-pass relocation) or == 1 (windows not applicable, strict-convergence check due, scratch too small).  The caller
 * relocates, resumes (nnc_kmeans_finalize(ws, 1)) and calls again.  host_mapped: 2 * (sizeof(nnc_kmeans_status) + 8) bytes
 * of host memory the device can write (see nnc_kmeans_status_publish); *ticket_io: a counter that never repeats for this
 * buffer; sorted: x_iter is in ascending order (windowed relocation allowed); reloc_scratch_dev: as for
 * nnc_kmeans_relocate_windowed (may be NULL); *n_windowed_out: relocation events settled inside.  The calling thread polls.
 * Empty clusters come in runs (duplicate initial centres): when the scratch holds windows of 256 samples
 * (nnc_kmeans_reloc_scratch_bytes(k, 256)) the iterations of the first batches are followed by the windowed relocation "in
 * case" -- its launches read the status themselves and do nothing without an event they can settle -- so that such events
 * cost no look-in; status_out->n_relocated counts them (included in *n_windowed_out).  status_out is read on entry (pass a
 * zeroed block for a new fit, the block of the previous call when calling again).
 * The status of a batch of plain iterations is written two iterations before the batch ends, so that the next batch is on the
 * stream before the device runs dry; when the call returns with done != 0, launches of iterations the host had enqueued in the
 * meantime may still be on the stream: they read the status themselves and return at once (work enqueued behind the call on the
 * same stream sees the finished fit, as before). */
int nnc_kmeans_fit(const float *x_iter, void *ws, const nnc_kmeans_params *p, int32_t max_batch, int32_t sorted,
                   void *reloc_scratch_dev, size_t reloc_scratch_bytes, void *host_mapped, uint64_t *ticket_io,
                   nnc_kmeans_status *status_out, int32_t *n_windowed_out, void *stream);
int nnc_kmeans_set_done(void *ws, int32_t done_code, void *stream);
/* counts_dev[j] (int64, k entries) = number of weights of x whose nearest centre is j, for the
 * current centres (which = 0) or the previous ones (which = 1): the index histogram the Huffman
 * step needs, from one more streaming pass over any permutation of the vector (e.g. the sorted
 * copy) instead of a count over the label vector.  Call between iterations or after the fit; it
 * ignores done / paused and leaves the state as it was. */
int nnc_kmeans_label_counts(const float *x, void *ws, const nnc_kmeans_params *p, int which, int64_t *counts_dev,
                            void *stream);

/* which: 0 = current centres (used by the next E-step), 1 = centres of the previous E-step.
 * centred != 0: as stored (x_mean subtracted); else un-centred (+ x_mean, float32 add). */
int nnc_kmeans_get_centers(void *ws, int which, int centred, float *out_dev, void *stream);

/* E-step only, against the CURRENT centres (which = 0) or the PREVIOUS ones (which = 1):
 * any of the outputs may be NULL.
 *   labels_out : centroid index per element, uint8 if label_bytes == 1 (k <= 256) else uint16
 *   quant_out  : cluster_centers_[labels_] (un-centred float32 centre values), utility.py:239
 *   dist_out   : (x~ - c~[label])^2 in float32, the distances _relocate_empty_clusters needs
 *   dist_hist4096_dev : with dist_out, also the 4096-bin histogram of (bits(dist) >> 19) & 4095
 *                (zeroed here): the first level of nnc_topm_hist_f32, for free in the same pass
 * Any alignment works; the 16-byte-per-lane form is taken when x, quant_out and dist_out are 16-byte aligned and
 * labels_out is aligned to 4 * label_bytes. */
int nnc_kmeans_assign(const float *x, void *ws, const nnc_kmeans_params *p, int which, void *labels_out,
                      int label_bytes, float *quant_out, float *dist_out, int64_t *dist_hist4096_dev, void *stream);

/* Selection of the farthest samples for scikit-learn's empty-cluster relocation
 * (_k_means_common.pyx:167-211: np.argpartition(distances, -n_empty)), on the device (d >= 0, so the
 * float32 bits order like the values):
 *   hist4096_dev[b] = #{ i : (bits(d[i]) >> shift) & (2^width - 1) == b  and, if prefix_shift >= 0,
 *                            bits(d[i]) >> prefix_shift == prefix }        (zeroed here)
 *   compact: every sample with bits(d[i]) >= thr_bits is written as the 64-bit key
 *   (bits(d[i]) << 32 | order-preserving bits of x[i]) to keys_dev[0..cap), arbitrary order;
 *   count_dev = how many there were (if it exceeds cap the buffer holds only the first cap of
 *   them).  Sorting the keys in descending order ranks the samples by distance, equal distances
 *   by value; samples equal in both are interchangeable.  d and x are in the same order (any). */
int nnc_topm_hist_f32(const float *d, int64_t n, int32_t shift, int32_t width, int32_t prefix_shift, uint32_t prefix,
                      int64_t *hist4096_dev, void *stream);
int nnc_topm_compact_f32(const float *d, const float *x, int64_t n, uint32_t thr_bits, int64_t *keys_dev,
                         int64_t cap, int64_t *count_dev, void *stream);

/* The relocation edits themselves (_k_means_common.pyx:197-211), as additive changes to the
 * per-cluster sums/counts in nnc_kmeans_partials(): keys_sorted_dev = the selected samples' keys
 * in descending order (nkeys of them, the same on every rank); the i-th empty cluster takes the
 * i-th sample, whose value travels in the key and whose current cluster is re-derived exactly.
 * Pass one key more than there are empty clusters when there is one: the runner-up shows whether
 * the cut fell between two different samples of equal distance (status.reloc_ties). */
int nnc_kmeans_relocate(void *ws, const int64_t *keys_sorted_dev, int32_t nkeys, void *stream);

/* The same selection without a pass over the whole vector, for a VALUE-SORTED x (nnc_sort_f32),
 * single rank.  Inside a cluster the distance falls monotonically towards the centre, so the
 * farthest samples sit at the ends of each cluster's stretch of the sorted vector; the label
 * counts of the paused iteration locate those stretches.
 *   nnc_kmeans_reloc_candidates: gathers the `window` samples either side of every cluster
 *     boundary (and at both ends of x) into cand_x_dev[cap] (cap >= 2 * window * (k + 1); the
 *     tail beyond n_cand is left as it was), their positions into win_dev (16 * (k + 2) bytes) and
 *     {n_cand, n_windows, bad, window} into meta_dev[4].
 *   The caller runs nnc_kmeans_assign on cand_x_dev (n = cap, dist_out) for the exact distances.
 *   nnc_kmeans_relocate_checked: picks the n_empty largest keys (distance bits << 32 | ordered
 *     value bits) among the n_cand candidates into keys_out_dev[n_empty + 1] (descending; the last
 *     entry is the runner-up, 0 if there is none), proves
 *     that no sample outside the windows can be among the n_empty farthest (every stretch
 *     between windows lies outside the zones of all centres but one and ends strictly below the
 *     n_empty-th key's distance) and then relocates as nnc_kmeans_relocate.  If the proof fails
 *     nothing is changed, the following nnc_kmeans_finalize(resume = 1) leaves the state paused
 *     with status.paused = 2, and the caller repeats the relocation with the full-pass
 *     selection above. */
int nnc_kmeans_reloc_candidates(const float *x_sorted, void *ws, const nnc_kmeans_params *p, int32_t window,
                                float *cand_x_dev, int64_t cap, void *win_dev, int32_t *meta_dev, void *stream);
int nnc_kmeans_relocate_checked(void *ws, const float *cand_x_dev, const float *cand_d_dev, const void *win_dev,
                                const int32_t *meta_dev, int32_t n_empty, int64_t *keys_out_dev, void *stream);
/* All of the above in one call (windows, candidates, distances, selection + proof + relocation,
 * resumed finalize), no host read.  nnc_kmeans_reloc_window: the window size used for n_empty
 * empty clusters on a vector of n samples, 0 if the windowed form does not apply (then use the
 * full pass); scratch: nnc_kmeans_reloc_scratch_bytes(k, window) bytes of device memory, 256-byte
 * aligned.  Afterwards status.paused is 0 (done) or 2 (not proven: repeat with the full pass). */
int32_t nnc_kmeans_reloc_window(int64_t n, int32_t n_empty);
size_t nnc_kmeans_reloc_scratch_bytes(int32_t k, int32_t window);
int nnc_kmeans_relocate_windowed(const float *x_sorted, void *ws, const nnc_kmeans_params *p, int32_t n_empty,
                                 void *scratch_dev, size_t scratch_bytes, void *stream);
/* The same for a sharded vector (one rank per GPU): nnc_kmeans_reloc_select_local leaves this
 * rank's n_empty farthest keys (descending) in keys_out_dev and its verdict (0 = proven) in the
 * int32 nnc_kmeans_reloc_flag(ws) points to, without touching the sums.  The caller all-gathers the
 * keys, all-reduces (MAX) the verdict word in place, sorts the gathered keys and hands the first
 * n_empty to nnc_kmeans_relocate_if_proven, which edits the (already all-reduced) sums on every
 * rank alike -- or does nothing if any verdict was non-zero; nnc_kmeans_finalize(resume = 1)
 * then reports paused = 2 on every rank. */
int nnc_kmeans_reloc_select_local(const float *x_sorted, void *ws, const nnc_kmeans_params *p, int32_t n_empty,
                                  void *scratch_dev, size_t scratch_bytes, int64_t *keys_out_dev, void *stream);
int32_t *nnc_kmeans_reloc_flag(void *ws);
int nnc_kmeans_relocate_if_proven(void *ws, const int64_t *keys_sorted_dev, int32_t nkeys, void *stream);
/* flag_dev = 1 if the two label vectors are identical else 0 (scikit-learn's strict convergence
 * test, _kmeans.py:717); nnc_kmeans_set_done_if sets done = done_code when *flag_dev != 0. */
int nnc_labels_equal(const void *a, const void *b, int64_t n, int label_bytes, int32_t *flag_dev, void *stream);
/* The M-step sums as scikit-learn runs them on one thread (sklearn/cluster/_k_means_lloyd.pyx:215-218, reached from
 * neural_network_compression/common/utility.py:237-238): sums_out[j] = float32 running sum, IN SAMPLE ORDER and from +0.0, of
 * (x[i] - x_mean) over the samples with labels[i] == j; counts_out[j] = their number.  labels: uint8 (label_bytes 1) or uint16 (2).
 * One wave per cluster; the cost is the size of the largest cluster times a few nanoseconds.  For the opt-in fit in the
 * reference's own arithmetic of tensors beyond NNC_REF_NMAX weights (kmeans.fit_reference_large). */
int nnc_ref_sums_f32(const float *x, int64_t n, float x_mean, const void *labels, int32_t label_bytes, int32_t k,
                     float *sums_out_dev, int64_t *counts_out_dev, void *stream);
int nnc_kmeans_set_done_if(void *ws, const int32_t *flag_dev, int32_t done_code, void *stream);

/* counts_dev[j] += #{ i : labels[i] == j }  (caller zeroes counts_dev; int64[k]). */
int nnc_bincount(const void *labels, int label_bytes, int64_t n, int32_t k, int64_t *counts_dev, void *stream);

/* ------------------------------------------------------------------------------------
 * A whole fit of a short tensor in one launch, in the reference's own arithmetic: KMeans(n_clusters=k, init=space, n_init=1,
 * algorithm="full").fit (common/utility.py:237-238) with the M-step as scikit-learn runs it on one thread -- float32 running
 * sums in sample order (_k_means_lloyd.pyx:215-218), float32(1 / count) averaging (_k_means_common.pyx:274-296) -- so that
 * centres, indices and n_iter_ are the reference's bit for bit (what stays open is what scikit-learn leaves to
 * numpy.argpartition: pairing and ties at a relocation cut, reported as in nnc_kmeans_status).  One workgroup, everything
 * on chip: NumPy-exact mean / variance, centring, E-step (brute force), M-step, relocation, stopping rules, final E-step,
 * decode.  n <= NNC_REF_NMAX, k <= NNC_REF_KMAX.
 *   x                : n float32 (un-centred, original order);  centers_init_dev: k float32 (un-centred)
 *   tol              : scikit-learn's relative tolerance (1e-4)
 *   labels_out[n] uint8, values_out[n] float32 (may be NULL) = cluster_centers_[labels_], centers_out[k] (un-centred),
 *   counts_out[k] int64 (may be NULL) = index histogram
 *   result_dev       : 32 bytes {int32 n_iter, stop (1 tol, 2 max_iter, 3 labels unchanged), n_relocations, reloc_ties,
 *                      reloc_multi, pad; float x_mean, tol_abs}
 * ---------------------------------------------------------------------------------- */
#define NNC_REF_NMAX 4096
#define NNC_REF_KMAX 128
int nnc_kmeans_fit_reference_f32(const float *x, int32_t n, const float *centers_init_dev, int32_t k, int32_t max_iter, float tol,
                                 uint8_t *labels_out, float *values_out, float *centers_out, int64_t *counts_out,
                                 void *result_dev, void *stream);

/* ------------------------------------------------------------------------------------
 * One layer tensor through the whole path as ONE host call: what Trainer._prune_parameters (common/trainer.py:177-193) and
 * Trainer.quantize (common/trainer.py:42-72) do to one tensor -- prune_weigth, get_weight_distribution of the non-zeros,
 * get_quantized_weight (modes "linear" and "density") -- plus the index histogram and Huffman code lengths.  Everything
 * above stays available step by step; this call only strings the steps together, with the K-sized host arithmetic of
 * utility.py (np.linspace, the cumulative distribution and its interp1d, the density init) restated in the reference's
 * order of float32 / float64 operations (the nnc_host_* functions, callable on their own and without a device).
 * The calling thread waits three times (statistics, bin counts, the K-sized results) and polls the fit's look-ins; several
 * host threads may run layers side by side, each with its own stream, workspace and host block.
 *   x           : n float32 on the device, pruned IN PLACE when p->prune
 *   mask_out    : n bytes (p->prune), labels_out: n uint8 (k <= 256) or uint16, values_out: n float32 (p->want_values)
 *   ws_dev      : nnc_compress_layer_workspace_bytes(n, k) device bytes; host_pinned: nnc_compress_layer_host_bytes() bytes of
 *                 host memory the device can write (hipHostMalloc / pinned), 8-byte aligned; *ticket_io: a counter kept
 *                 with that host block (never reset)
 *   result      : host struct; status NNC_LAYER_DONE, or NNC_LAYER_HOST: the tensor has been pruned (mask, sigma, threshold,
 *                 n_zeroed are valid) but the fit needs the caller's step-by-step path (short tensor with density init,
 *                 fewer than 512 weights with more than 128 centroids, full-pass relocation, strict-convergence check).
 * ---------------------------------------------------------------------------------- */
#define NNC_INIT_LINEAR 0
#define NNC_INIT_DENSITY 1
#define NNC_LAYER_DONE 0
#define NNC_LAYER_HOST 1
#define NNC_ARITH_FIXED 0
#define NNC_ARITH_REFERENCE 1
typedef struct nnc_layer_params {
    float q;              /* prune_weigth's q (float32) */
    int32_t prune;        /* 0: leave x as it is */
    int32_t std_smooth;   /* threshold = std(x) * q, else q */
    int32_t bits;         /* 2**bits centroids (+ 1 for density) */
    int32_t mode;         /* NNC_INIT_LINEAR / NNC_INIT_DENSITY */
    int32_t want_values;  /* write cluster_centers_[labels_] to values_out */
    int32_t km_flags;     /* nnc_kmeans_params.flags of the fit (0, NNC_KM_TWO_LAUNCH or NNC_KM_LOOP) */
    int32_t reserved;
} nnc_layer_params;
typedef struct nnc_layer_result {
    int32_t status, k, label_bytes, arith;
    int32_t n_iter, stop, n_relocations, n_reloc_windowed, reloc_ties, reloc_multi;
    float sigma, threshold;
    int64_t n_zeroed, total_bits;
    float centers[NNC_KMAX];        /* cluster_centers_ */
    int64_t counts[NNC_KMAX];       /* index histogram */
    uint8_t code_lengths[NNC_KMAX]; /* Huffman code length per index */
} nnc_layer_result;
size_t nnc_compress_layer_workspace_bytes(int64_t n, int32_t k);
size_t nnc_compress_layer_host_bytes(void);
int nnc_compress_layer_f32(float *x, int64_t n, const nnc_layer_params *p, uint8_t *mask_out, void *labels_out, float *values_out,
                           void *ws_dev, size_t ws_bytes, void *host_pinned, size_t host_bytes, uint64_t *ticket_io,
                           nnc_layer_result *result, void *stream);
/* np.linspace(start, stop, num) for float32 scalars (NumPy >= 2), bit for bit (common/utility.py:208, 365) */
int nnc_host_linspace_f32(float start, float stop, int32_t num, float *out);
/* (xnew[300] float32, cdf[300] float64) = get_weight_distribution's host part (common/utility.py:374-392) from the 32 steps
 * and the 31 bin counts */
int nnc_host_cdf(const float *steps32, const int64_t *counts31, float *xnew300, double *cdf300);
/* the density init (common/utility.py:211-221): 2**bits + 1 float32 centroids from that curve */
int nnc_host_density_init(const float *xnew300, const double *cdf300, int32_t bits, float *space_out);

/* ------------------------------------------------------------------------------------
 * k-means++ seeding: the reference's 4th initialisation mode, get_quantized_weight(mode="kmeans++") =
 * KMeans(n_clusters=2**bits).fit(...) (common/utility.py:228-232) -> scikit-learn's _kmeans_plusplus
 * (cluster/_kmeans.py:163-253) on the mean-centred float32 weights.  The host draws the random numbers from NumPy's
 * global generator in scikit-learn's order (one for the first seed, nnc_kmeanspp_trials(k) uniforms per further seed)
 * and resolves the first seed's index; the device runs the k - 1 rounds of D^2 sampling without a host round trip:
 * distances in scikit-learn's upcast form ((-2 * (c * x)) + c * c) + x * x in float64 -> float32, clipped at 0;
 * potentials and running sums in float64 in a fixed order (csrc/nnc_pp.hip; scikit-learn's own potential is a float32
 * BLAS dot whose order is the BLAS kernel's, so its last bits -- and with them, on long vectors, a candidate now and
 * then -- are not reproducible by anybody).  Single GPU (the whole vector).
 *   uniforms_dev : (k - 1) * nnc_kmeanspp_trials(k) float64 in [0, 1), round-major
 *   seeds_out_dev[k] : the seeds, centred (x[id] - x_mean in float32), in the order chosen; seed_ids_out_dev[k]: their indices
 *   ws: nnc_kmeanspp_workspace_bytes(n, k) bytes, 256-byte aligned.
 * ---------------------------------------------------------------------------------- */
int32_t nnc_kmeanspp_trials(int32_t k);
size_t nnc_kmeanspp_workspace_bytes(int64_t n, int32_t k);
int nnc_kmeanspp_seed_f32(const float *x, int64_t n, float x_mean, int32_t k, int64_t first_id, const double *uniforms_dev,
                          float *seeds_out_dev, int64_t *seed_ids_out_dev, void *ws, size_t ws_bytes, void *stream);

/* ------------------------------------------------------------------------------------
 * Centroid fine-tuning (Deep Compression's trained quantization).  The reference describes it and leaves it out "because the
 * latency is excessive: for each single batch in every epoch, I should have scanned all the gradients"
 * (papers/lat/report.tex:149-158):   dL/dC_k = sum over the weights with centroid index k of dL/dW.
 * nnc_centroid_grad_f32: sums_dev[k] (int64, zeroed here) = sum of rint(grad * 2^fix_shift) per centroid index
 *   (fix_shift = nnc_fix_shift(max |grad|, n): exact integer sums, independent of order and of GPU count -- a sharded
 *   caller all-reduces them); counts_dev[k] (may be NULL) = members per index.  dL/dC_k = ldexp(sums[k], -fix_shift).
 * nnc_gather_f32: out[i] = centers_dev[labels[i]], the decode step cluster_centers_[labels_] (common/utility.py:239), for
 *   writing updated centroids back into the layer.
 * ---------------------------------------------------------------------------------- */
int nnc_centroid_grad_f32(const float *grad, const void *labels, int label_bytes, int64_t n, int32_t k, int32_t fix_shift,
                          int64_t *sums_dev, int64_t *counts_dev, void *stream);
int nnc_gather_f32(const float *centers_dev, int32_t k, const void *labels, int label_bytes, int64_t n, float *out, void *stream);

/* ------------------------------------------------------------------------------------
 * Multi-GPU: the vector is sharded across one process per GPU (contiguous shards starting on multiples of
 * NNC_CHUNK elements); the exchange per Lloyd iteration is one all-reduce (SUM) of the 2K int64 sums / counts over
 * RCCL / xGMI, enqueued by the library on the caller's stream between its own kernels.  The reference has no
 * counterpart (single process; scikit-learn's threads share memory, _k_means_lloyd.pyx:118-152): these entry points
 * are what BASELINE.json's north_star adds.  RCCL is bound at run time (dlopen), sharing the copy a host framework
 * has already loaded.
 * ---------------------------------------------------------------------------------- */
#define NNC_COMM_ID_BYTES 128
#define NNC_I64 0
#define NNC_I32 1
#define NNC_F32 2
#define NNC_SUM 0
#define NNC_MAX 1
#define NNC_MIN 2
/* Rank 0 makes the id, the caller carries the NNC_COMM_ID_BYTES bytes to the other ranks (any channel), every rank calls
 * nnc_comm_init on the thread whose current device is its GPU (blocking rendezvous). */
/* NNC_OK if librccl can be bound in this process (creates nothing).  nnc_comm_init blocks until every rank has entered it, so the
 * ranks should agree (over the caller's own group) that all of them can, before any of them does. */
int nnc_comm_available(void);
int nnc_comm_unique_id(void *id_out, size_t len);
int nnc_comm_init(void **comm_out, const void *id, size_t len, int32_t rank, int32_t world);
int nnc_comm_destroy(void *comm);
int nnc_comm_rank(void *comm);
int nnc_comm_world(void *comm);
/* In place on buf_dev; dtype NNC_I64 / NNC_I32 / NNC_F32, op NNC_SUM / NNC_MAX / NNC_MIN. */
int nnc_comm_allreduce(void *comm, void *buf_dev, int64_t count, int32_t dtype, int32_t op, void *stream);
/* recv_dev holds world * bytes_per_rank bytes, rank r's block at r * bytes_per_rank. */
int nnc_comm_allgather(void *comm, const void *send_dev, void *recv_dev, int64_t bytes_per_rank, void *stream);
/* nnc_kmeans_iterate for a sharded vector: per iteration nnc_kmeans_accumulate on this rank's shard, the all-reduce of
 * nnc_kmeans_partials(), nnc_kmeans_finalize -- `iters` of them enqueued back to back, no host round trip in between.
 * host_mapped / ticket as in nnc_kmeans_iterate_publish (host_mapped may be NULL: no look-in). */
int nnc_kmeans_iterate_sharded(void *comm, const float *x, void *ws, const nnc_kmeans_params *p, int32_t iters,
                               void *host_mapped, uint64_t ticket, void *stream);
/* nnc_kmeans_relocate_windowed for a sharded vector, exchanges included (verdict word: all-reduce MAX; keys: all-gather,
 * merged on the device); n_empty and the applicability (nnc_kmeans_reloc_window on the SHORTEST shard) must be the same
 * on every rank.  scratch: nnc_kmeans_reloc_scratch_bytes_sharded(k, window, world) bytes, 256-byte aligned. */
size_t nnc_kmeans_reloc_scratch_bytes_sharded(int32_t k, int32_t window, int32_t world);
/* The merge step on its own: `nlists` descending lists of `per` positive int64 keys each (padded with 0 or -1), laid end
 * to end in lists_dev -> the m largest keys overall, descending, in out_dev[m] (0 where there are fewer). */
int nnc_merge_keys(const int64_t *lists_dev, int32_t nlists, int32_t per, int64_t *out_dev, int32_t m, void *stream);
int nnc_kmeans_relocate_windowed_sharded(void *comm, const float *x_sorted, void *ws, const nnc_kmeans_params *p, int32_t n_empty,
                                         void *scratch_dev, size_t scratch_bytes, void *stream);
/* nnc_kmeans_fit for a sharded vector: the whole Lloyd loop of KMeans.fit (sklearn/cluster/_kmeans.py:624-752, reached from
 * neural_network_compression/common/utility.py:237-238) as ONE call per rank -- batches of nnc_kmeans_iterate_sharded, the look-ins,
 * batch sizing from the decay of the centre shift, and nnc_kmeans_relocate_windowed_sharded for the empty-cluster events it
 * applies to; no host language between two launches, so N ranks do not pay an interpreter round trip per batch or per event.
 * Every rank makes the same call; the status is the same on every rank after each iteration, so every rank takes the same
 * decisions.  n_min = the length of the SHORTEST shard (the applicability of the windowed relocation must come out alike
 * everywhere).  Other arguments as in nnc_kmeans_fit; reloc_scratch: nnc_kmeans_reloc_scratch_bytes_sharded(k, window, world)
 * bytes.  Returns with status_out->done != 0, or with status_out->paused != 0 when the caller has to run the full-pass
 * relocation / strict-convergence check (then call again).  *n_windowed_out = windowed events this call asked for. */
int nnc_kmeans_fit_sharded(void *comm, const float *x_iter, void *ws, const nnc_kmeans_params *p, int64_t n_min, int32_t max_batch,
                           int32_t sorted, void *reloc_scratch_dev, size_t reloc_scratch_bytes, void *host_mapped,
                           uint64_t *ticket_io, nnc_kmeans_status *status_out, int32_t *n_windowed_out, void *stream);

/* ------------------------------------------------------------------------------------
 * Measurement aid (used by bench.py): HIP events around the launches of the data-touching kernels, recorded on the
 * stream each kernel is launched on.
 * ---------------------------------------------------------------------------------- */
#define NNC_PROF_ASSIGN_ACCUMULATE 0 /* k_assign<accumulate>: streaming Lloyd pass (vector in any order) */
#define NNC_PROF_BOUNDS 1            /* k_bounds: rank-boundary Lloyd pass (sorted vector) */
#define NNC_PROF_ASSIGN_LABELS 2     /* k_assign<labels>: final E-step, labels + values */
#define NNC_PROF_THRESHOLD 3         /* k_threshold: mask + zero in place */
#define NNC_PROF_CHUNK_SUMS 4        /* k_chunk_sums: NumPy-exact chunk sums (two passes per sigma) */
#define NNC_PROF_FINALIZE 5          /* k_finalize (K-sized) */
#define NNC_PROF_PREFIX 6            /* k_prefix_blocks */
#define NNC_PROF_MINMAX 7            /* k_minmax */
#define NNC_PROF_LLOYD 8             /* k_lloyd: the one-workgroup Lloyd loop (any number of iterations per launch) */
#define NNC_PROF_RELOC 9             /* the windowed empty-cluster relocation: k_reloc_head / windows / dist / select (K-sized) */
/* Which tags get events from now on (bit t = NNC_PROF_* tag t; default all): an event pair costs its launch a little, so a
 * timed run may want the passes over the vector only. */
int nnc_profile_tags(uint32_t mask);
int nnc_profile_begin(int32_t max_launches);
/* Waits for the recorded events; ms_out[i] / tags_out[i] = duration in ms and NNC_PROF_* tag of the i-th timed launch
 * (launch order), up to cap entries; count_out = launches timed.  Launches enqueued after the state machine had stopped
 * return at once and show up as very short entries. */
int nnc_profile_end(float *ms_out, int32_t *tags_out, int64_t cap, int64_t *count_out);

#ifdef NNC_DIAG
/* Diagnostics: exported only by a library built with -DNNC_DIAG (libnnc_hip_diag.so, tools/); the product library has
 * no process-global switches.
 * nnc_debug_set_ablation: a != 0 selects an ablated build of the Lloyd streaming kernel whose RESULTS ARE WRONG (1: no
 *   LDS atomics, 2: no table lookups, 3: neither).
 * nnc_debug_set_trace: if buf_dev != NULL every workgroup of the Lloyd streaming kernel stores four 100 MHz timestamps
 *   {start, loop start, loop end, end} at buf_dev[4*blockIdx].
 * nnc_debug_clock: out_dev[2b] = shader clock in GHz seen by workgroup b over a spin loop, out_dev[2b+1] = its length in us.
 * nnc_debug_reloc_fail: why the last windowed-relocation proof failed (0 = it held); synchronous. */
int nnc_debug_set_ablation(int a);
int nnc_debug_set_trace(unsigned long long *buf_dev);
int nnc_debug_clock(int blocks, int iters, float *out_dev, void *stream);
int nnc_debug_reloc_fail(void *ws, int32_t *host_out);
#endif

/* Huffman code length per centroid index from the index histogram (HOST function, host
 * pointers).  The reference names Huffman coding (README.md:9) but never implements it; the
 * definition is in DESIGN.md.  lengths_out[k]; hist_out[65] (hist_out[l] = symbols of length l). */
int nnc_huffman_lengths(const int64_t *counts, int32_t k, uint8_t *lengths_out, int64_t *hist_out,
                        int64_t *total_bits_out);

/* ------------------------------------------------------------------------------------
 * Compressed form of a quantized layer: the centroid indices as a canonical-Huffman bit stream (Deep Compression's third
 * stage; named by the reference, README.md:9, never implemented).  Built from what the path already has: indices
 * (nnc_kmeans_assign), histogram (nnc_kmeans_label_counts / nnc_bincount), code lengths (nnc_huffman_lengths).
 *   nnc_huffman_codes        (host)  canonical codes from code lengths <= 32 bits (symbols ordered by (length, symbol))
 *   nnc_huffman_chunk_offsets        bit offset of every chunk of NNC_CODEC_CHUNK indices: chunk_off_dev[nchunks + 1] uint64,
 *                                    chunk_off_dev[nchunks] = total bits   (nchunks = nnc_codec_chunks(n))
 *   nnc_huffman_encode               words_dev[nwords] (uint32, zeroed here; nwords >= total_bits / 32 + 2), MSB first
 *   nnc_huffman_decode_tables (host) first-code / count / first-index per length + symbols by (length, symbol), packed into
 *                                    nnc_huffman_decode_tables_bytes() bytes; copy them to the device for
 *   nnc_huffman_decode               one thread per chunk; *bad_dev = 1 if a chunk does not parse to its recorded length
 * The pruned zeros share one centroid, so their index is the most frequent symbol and costs one bit in the dense stream.
 *
 * Relative-index sparse form (Deep Compression section 3; the format the reference's report cites, papers/lat/report.tex:327,
 * README.md:9): only the indices that are NOT zero_symbol are stored, each with the distance to the previous stored position in
 * delta_bits bits (stored value = distance - 1); a longer gap takes filler entries (distance 2^delta_bits, index zero_symbol).
 * Distances restart at every chunk of NNC_CODEC_CHUNK positions.  The two entry streams are Huffman coded with the functions above.
 *   nnc_sparse_entry_offsets         entries_off_dev[nchunks + 1] uint64: entries in front of every chunk, [nchunks] = all entries
 *   nnc_sparse_emit                  delta_out_dev[entries] uint8, sym_out_dev[entries] (width of the labels)
 *   nnc_sparse_expand                the inverse; *bad_dev = 1 if an entry points outside its chunk
 * storage.py stores whichever of the dense stream and the sparse form (delta_bits 4 or 8) is smaller, per tensor.
 * ---------------------------------------------------------------------------------- */
#define NNC_CODEC_CHUNK 1024
int nnc_huffman_codes(const uint8_t *lengths, int32_t k, uint32_t *codes_out);
size_t nnc_codec_chunks(int64_t n);
int nnc_huffman_chunk_offsets(const void *labels, int label_bytes, int64_t n, const uint8_t *lengths_dev, int32_t k,
                              uint64_t *chunk_off_dev, void *stream);
int nnc_huffman_encode(const void *labels, int label_bytes, int64_t n, const uint32_t *codes_dev, const uint8_t *lengths_dev, int32_t k,
                       const uint64_t *chunk_off_dev, uint32_t *words_dev, int64_t nwords, void *stream);
size_t nnc_huffman_decode_tables_bytes(void);
int nnc_huffman_decode_tables(const uint8_t *lengths, int32_t k, void *tables_out, size_t tables_bytes);
int nnc_huffman_decode(const uint32_t *words_dev, const uint64_t *chunk_off_dev, int64_t n, const void *tables_dev, int32_t k,
                       void *labels_out, int label_bytes, int32_t *bad_dev, void *stream);
int nnc_sparse_entry_offsets(const void *labels, int label_bytes, int64_t n, int32_t zero_symbol, int32_t delta_bits, uint64_t *entries_off_dev, void *stream);
int nnc_sparse_emit(const void *labels, int label_bytes, int64_t n, int32_t zero_symbol, int32_t delta_bits, const uint64_t *entries_off_dev,
                    uint8_t *delta_out_dev, void *sym_out_dev, void *stream);
int nnc_sparse_expand(const uint8_t *delta_dev, const void *sym_dev, int label_bytes, const uint64_t *entries_off_dev, int64_t n, int32_t zero_symbol,
                      void *labels_out, int32_t *bad_dev, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* NNC_H */
