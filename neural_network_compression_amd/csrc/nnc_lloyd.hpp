// nnc_lloyd.hpp -- the Lloyd loop of one fit inside ONE workgroup (textually included by nnc_hip.hip, which owns KmWs and the helpers).
//
// The two-launch iteration (k_bounds: one wave per cluster boundary, sums by global atomics; k_finalize: one workgroup) spends
// its time in launch boundaries and dependent round trips to memory, not in work.  Here one resident workgroup runs iteration after
// iteration until the fit stops, pauses for an empty cluster or meets something it hands to the wide path:
//   * everything K-sized lives in LDS: centres, zone ends (as float32 thresholds), per-cluster sums and counts, the previous label
//     counts, and for each of the 2 (K - 1) rank searches where it ended last time (rank, threshold, density there);
//   * pass 1, kl_search: the ranks a_j = #{x~ < L_j} and b_j = #{x~ <= U_j} of every boundary, each by a group of EIGHT lanes.  A probe
//     = the 128 samples around the predicted rank + the two 8-byte fine prefixes in front of them (nnc_kmeans_prefix_build); the
//     prediction is a Newton step on the rank function (rank moves by density x threshold shift, the density measured over the
//     last move).  A second miss goes to a bracketed search (Newton from the block just read, else a nine-way split of the
//     bracket: eight single-sample probes), logarithmic whatever the data look like.  The block that holds the rank also gives the
//     prefix sum there (fine prefix + the images in front of the rank inside the block);
//   * pass 2, kl_label: the samples float32 cannot place from the zones alone, [max(a_j, b_{j-1}), b_j), cut into chunks dealt round
//     robin to the groups and labelled with scikit-learn's exact float32 expression -- between two centres, or against all of
//     j .. phi_j where three or more centres sit within rounding distance of each other;
//   * the finish step (sums from prefix differences, average, shift, NumPy's pairwise sum, tolerance test, centre order, zones) on the
//     same LDS arrays: six barriers when nothing unusual happens (kl_finish_fast), the general step otherwise (kl_finish).
// What it does not do itself: empty clusters (status.paused, as k_finalize reports them: the relocation chain takes over) and
// iterations with more than a million undecided samples (two centres float32 can hardly tell apart) or a search that does not
// settle (ws->wide: the k_bounds / k_finalize pair enqueued behind every launch of the loop runs exactly then).
// Same integers, same float32 operations as the two-launch form: the trajectory is bit-identical (tests/test_gpu_lloyd.py).
// Measured (tools/sweep_loop_vs_two.py, 0.1 M - 25 M weights; fit time against the launch-per-iteration form): 0.44 - 0.64 up to
// K = 32, about the same at K = 65, 1.03 - 1.09 at K = 129, about 1.4 at K = 257 -- sixteen waves on one compute unit issue an
// instruction every eight cycles each, and 512 searches of a few hundred instructions are then the bound -- so the library takes
// the loop up to NNC_KM_LOOP_KMAX centres.
//
// Reference: the loop of sklearn/cluster/_kmeans.py:_kmeans_single_lloyd (624-752), reached from utility.py:237-238.

struct KlDiag { unsigned long long ph[16], last, sp[8], slast; };
#ifdef NNC_DIAG
#define KLSTAMP(slot) do { if (threadIdx.x == 0) { const unsigned long long now_ = __builtin_amdgcn_s_memrealtime(); dg->ph[slot] += now_ - dg->last; dg->last = now_; } } while (0)
#define KLSUB(slot) do { if (threadIdx.x == 0) { const unsigned long long now_ = __builtin_amdgcn_s_memrealtime(); dg->sp[slot] += now_ - dg->slast; dg->slast = now_; } } while (0)
#else
#define KLSTAMP(slot) do { } while (0)
#define KLSUB(slot) do { } while (0)
#endif
#define KL_MAXR 24      // probes per boundary before it is handed to the wide path (never reached on monotone data: <= 20)
#define KL_TAIL_MAX 2048 // samples between the ranks a and b a group of eight lanes goes through itself

struct KlHeap { int start[64]; int len[64]; float val[64]; };
struct KlHead {
    int ku, slow, nch, n_empty, depth, pad0, pad1, pad2;
    float tot;
    int f_empty, f_diff, f_reorder;   // votes of the fast finish step (set by any wave, read behind a barrier, cleared by thread 0)
    int wave_i[16];
    double wmax[16], wmin[16];        // per wave: the largest zone end from below, the smallest from above
    KlHeap heap;
    // relocation of empty clusters inside the loop (kl_relocate)
    int r_cnt, r_bad, r_flat, r_pad1; // (r_flat: a run of equal values was settled as a whole in this iteration, kl_chunks)
    unsigned long long r_keys[8];     // the selected keys, descending (one more than there are empty clusters)
    unsigned long long r_wkey[2][16]; // per wave: its largest remaining key ...
    int r_widx[2][16];                // ... and the slot that holds it
    int r_empty[8], r_old[8];         // the empty clusters (ascending index); the clusters the selected samples leave
};
static_assert(sizeof(KlHead) % 16 == 0, "the arrays behind the header are 16-byte aligned");

struct KlArr {
    double *Lb, *Ub;                                                   // per boundary j: L_j, U_j (per centre while the zones are being made: zl, zr)
    long long *sum_s, *cnt_s, *A, *B, *PA, *PB, *prevc, *hint;          // PA/PB double as the per-cluster sums / counts in original index order
    float *hL, *hR, *cs, *csq, *cnew, *cold, *sq, *call, *Lf, *Uf;         // Lf / Uf: L_j rounded up, U_j rounded down to float32
    uint16_t *so, *perm, *phi, *qj;                                      // phi[j]: the highest centre that can still win below U_j; qj: the labelling chunks' boundaries
};

static size_t kl_lds_bytes(int kc) { return sizeof(KlHead) + (size_t)kc * (2 * 8 + 9 * 8 + 12 * 4 + 3 * 2) + 4096 * 2; }

__device__ __forceinline__ void kl_carve(unsigned char *smem, int kc, KlHead **hd, KlArr *L)
{
    *hd = reinterpret_cast<KlHead *>(smem);
    unsigned char *p = smem + sizeof(KlHead);
    auto take = [&](size_t bytes) { unsigned char *q = p; p += bytes; return q; };
    L->Lb = reinterpret_cast<double *>(take((size_t)kc * 8)); L->Ub = reinterpret_cast<double *>(take((size_t)kc * 8));
    L->sum_s = reinterpret_cast<long long *>(take((size_t)kc * 8)); L->cnt_s = reinterpret_cast<long long *>(take((size_t)kc * 8));
    L->A = reinterpret_cast<long long *>(take((size_t)kc * 8)); L->B = reinterpret_cast<long long *>(take((size_t)kc * 8));
    L->PA = reinterpret_cast<long long *>(take((size_t)kc * 8)); L->PB = reinterpret_cast<long long *>(take((size_t)kc * 8));
    L->prevc = reinterpret_cast<long long *>(take((size_t)kc * 8)); L->hint = reinterpret_cast<long long *>(take((size_t)kc * 16)); // (two searches per boundary)
    L->hL = reinterpret_cast<float *>(take((size_t)kc * 8)); L->hR = reinterpret_cast<float *>(take((size_t)kc * 8));
    L->cs = reinterpret_cast<float *>(take((size_t)kc * 4)); L->csq = reinterpret_cast<float *>(take((size_t)kc * 4));
    L->cnew = reinterpret_cast<float *>(take((size_t)kc * 4)); L->cold = reinterpret_cast<float *>(take((size_t)kc * 4));
    L->sq = reinterpret_cast<float *>(take((size_t)kc * 4)); L->call = reinterpret_cast<float *>(take((size_t)kc * 4));
    L->Lf = reinterpret_cast<float *>(take((size_t)kc * 4)); L->Uf = reinterpret_cast<float *>(take((size_t)kc * 4));
    L->so = reinterpret_cast<uint16_t *>(take((size_t)kc * 2)); L->perm = reinterpret_cast<uint16_t *>(take((size_t)kc * 2));
    L->phi = reinterpret_cast<uint16_t *>(take((size_t)kc * 2));
    L->qj = reinterpret_cast<uint16_t *>(take((size_t)4096 * 2));
}

// Sums / minimum over a group of eight lanes (every lane of the group gets the result) by DPP: two quad permutes and a mirror
// inside the half row -- no LDS crossbar, one VALU instruction each.  Call with the whole wave active.
#define KL_DPP_I(v, ctrl) __builtin_amdgcn_update_dpp(0, (v), (ctrl), 0xF, 0xF, true)
#define KL_QP_XOR1 0xB1 // quad_perm [1,0,3,2]
#define KL_QP_XOR2 0x4E // quad_perm [2,3,0,1]
#define KL_HALF_MIRROR 0x141
__device__ __forceinline__ int grp8_sum_i(int v)
{
    v += KL_DPP_I(v, KL_QP_XOR1); v += KL_DPP_I(v, KL_QP_XOR2); v += KL_DPP_I(v, KL_HALF_MIRROR);
    return v;
}
__device__ __forceinline__ int grp8_min_i(int v)
{
    v = min(v, KL_DPP_I(v, KL_QP_XOR1)); v = min(v, KL_DPP_I(v, KL_QP_XOR2)); v = min(v, KL_DPP_I(v, KL_HALF_MIRROR));
    return v;
}
__device__ __forceinline__ long long grp8_sum_ll(long long v)
{
#define KL_STEP_LL(ctrl) do { const int lo_ = KL_DPP_I((int)(unsigned)(v & 0xFFFFFFFFll), ctrl), hi_ = KL_DPP_I((int)(v >> 32), ctrl); \
        v += (long long)(((unsigned long long)(unsigned)hi_ << 32) | (unsigned)lo_); } while (0)
    KL_STEP_LL(KL_QP_XOR1); KL_STEP_LL(KL_QP_XOR2); KL_STEP_LL(KL_HALF_MIRROR);
#undef KL_STEP_LL
    return v;
}

// ---- NumPy's pairwise float32 sum of e[0 .. n) (n <= 2048), the split tree made once per launch -------------------------------
__device__ void kl_heap_build(KlHead *hd, int n) // by the first wave; the caller puts a workgroup barrier behind it
{
    KlHeap *hp = &hd->heap;
    const int lane = threadIdx.x & 63;
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
    if (lane == 0) hd->depth = depth;
}

// one leaf (l <= 128 values from e[st]) by eight lanes: eight strided accumulators, combined as a tree, then the ragged tail
__device__ __forceinline__ float kl_leaf_sum(const float *e, int st, int l, int j8)
{
    float res;
    if (l < 8) {
        res = 0.0f;
        for (int i = 0; i < l; i++) res += e[st + i];
    } else {
        float r = e[st + j8];
        const int lim = l - (l % 8);
        for (int i = 8; i < lim; i += 8) r += e[st + i + j8];
        r = r + __shfl_xor(r, 1);
        r = r + __shfl_xor(r, 2);
        r = r + __shfl_xor(r, 4);
        res = r;
        for (int i = lim; i < l; i++) res += e[st + i];
    }
    return res;
}

template <int NT>
__device__ __forceinline__ void kl_pairwise(KlHead *hd, const float *e, int n) // hd->tot, valid behind the trailing barrier
{
    KlHeap *hp = &hd->heap;
    const int tid = threadIdx.x, lane = tid & 63, j8 = tid & 7;
    if (n <= LEAF) {
        if (tid < 8) { const float r = kl_leaf_sum(e, 0, n, j8); if (tid == 0) hd->tot = r; }
    } else {
        const int depth = hd->depth;
        for (int node = 1 + (tid >> 3); node < (2 << depth) && node < 64; node += NT / 8) { // (a group of eight lanes stays together)
            const int l = hp->len[node];
            if (l > 0 && l <= LEAF) { const float r = kl_leaf_sum(e, hp->start[node], l, j8); if (j8 == 0) hp->val[node] = r; }
        }
        km_lds_barrier();
        if (tid < 64) {
            for (int lev = depth - 1; lev >= 0; lev--) {
                const int i = (1 << lev) + lane;
                if (lane < (1 << lev) && hp->len[i] > LEAF) hp->val[i] = hp->val[2 * i] + hp->val[2 * i + 1];
                wave_lds_fence();
            }
            if (lane == 0) hd->tot = hp->val[1];
        }
    }
    km_lds_barrier();
}

// ---- zone ends per centre (zl in Lb[], zr in Ub[]) -> thresholds per boundary, and whether any three centres crowd --------------
//   U_j = max over q <= j of zr[q] (above it no centre up to j can win), L_j = min over q > j of zl[q] (below it none above j can).
template <int NT>
__device__ __forceinline__ void kl_derive(KlHead *hd, const KlArr &L, int ku)
{
    const int tid = threadIdx.x;
    int ok = 1;
    for (int p = tid; p < ku; p += NT) {
        if (p > 0 && !(L.Ub[p] >= L.Ub[p - 1])) ok = 0;
        if (p + 1 < ku && !(L.Lb[p] <= L.Lb[p + 1])) ok = 0;
    }
    const int mono = __syncthreads_and(ok);
    if (!mono) { // zones that reach across a neighbour's (crowded centres): running maximum from below, running minimum from above
        for (int off = 1; off < ku; off <<= 1) {
            double a[2], b[2];
#pragma unroll
            for (int r = 0; r < 2; r++) {
                const int q = tid + r * NT;
                a[r] = (q < ku && q >= off) ? fmax(L.Ub[q], L.Ub[q - off]) : (q < ku ? L.Ub[q] : 0.0);
                b[r] = (q < ku && q + off < ku) ? fmin(L.Lb[q], L.Lb[q + off]) : (q < ku ? L.Lb[q] : 0.0);
            }
            km_lds_barrier();
#pragma unroll
            for (int r = 0; r < 2; r++) {
                const int q = tid + r * NT;
                if (q < ku) { L.Ub[q] = a[r]; L.Lb[q] = b[r]; }
            }
            km_lds_barrier();
        }
    }
    double t[2];
#pragma unroll
    for (int r = 0; r < 2; r++) { const int q = tid + r * NT; t[r] = (q + 1 < ku) ? L.Lb[q + 1] : INFINITY; }
    km_lds_barrier();
#pragma unroll
    for (int r = 0; r < 2; r++) { const int q = tid + r * NT; if (q < ku) L.Lb[q] = t[r]; }
    km_lds_barrier();
    // phi_j = max{q : zl[q] <= U_j}: the candidates of the samples boundary j cannot decide from the zones are the centres j .. phi_j
    // (j + 1 unless centres crowd).  With the running minimum in place: the largest q whose L_{q-1} is still <= U_j.
    for (int j = tid; j + 1 < ku; j += NT) {
        int q = j + 1;
        const double u = L.Ub[j], l = L.Lb[j];
        while (q + 1 < ku && L.Lb[q] <= u) q++;
        L.phi[j] = (uint16_t)q;
        // the thresholds as float32: the smallest one >= L_j, the largest one <= U_j
        float lf = (float)l, uf = (float)u;
        if ((double)lf < l) lf = nextafterf(lf, INFINITY);
        if ((double)uf > u) uf = nextafterf(uf, -INFINITY);
        L.Lf[j] = lf; L.Uf[j] = uf;
    }
    km_lds_barrier();
}

// ---- order of the centres, distinct values, zones: what km_finalize_body leaves for the next E-step, from cnew[] in LDS --------
// (perm[] holds the order of the previous iteration; `fresh`: it holds nothing yet)
template <int NT>
__device__ __forceinline__ void kl_tables(KmWs *__restrict__ ws, KlHead *hd, const KlArr &L, const int k, const int cur, const bool fresh,
                                          const float p_lo, const float p_hi, KlDiag *dg)
{
    const int tid = threadIdx.x;
    KmTab *tab = &ws->tab[cur];
    int still_sorted = 0;
    if (!fresh) {
        for (int p = tid; p < k; p += NT) L.call[p] = L.cnew[L.perm[p]];
        km_lds_barrier();
        int ok = 1;
        for (int p = tid; p + 1 < k; p += NT) {
            const float va = L.call[p], vb = L.call[p + 1];
            const int a = L.perm[p], b = L.perm[p + 1];
            ok &= (va < vb) || (va == vb && a < b);
        }
        still_sorted = __syncthreads_and(ok);
        // nearly sorted (two neighbours changed places): a few odd-even transposition passes repair it
        for (int pass = 0; pass < 3 && !still_sorted; pass++) {
            for (int parity = 0; parity < 2; parity++) {
                for (int p = 2 * tid + parity; p + 1 < k; p += 2 * NT) {
                    const float va = L.call[p], vb = L.call[p + 1];
                    const uint16_t a = L.perm[p], b = L.perm[p + 1];
                    if (!((va < vb) || (va == vb && a < b))) { L.call[p] = vb; L.call[p + 1] = va; L.perm[p] = b; L.perm[p + 1] = a; }
                }
                km_lds_barrier();
            }
            int ok2 = 1;
            for (int p = tid; p + 1 < k; p += NT) {
                const float va = L.call[p], vb = L.call[p + 1];
                ok2 &= (va < vb) || (va == vb && L.perm[p] < L.perm[p + 1]);
            }
            still_sorted = __syncthreads_and(ok2);
        }
    }
    if (!still_sorted) { // rank by counting; ties by original index
        for (int j = tid; j < k; j += NT) {
            const float v = L.cnew[j];
            int rank = 0;
#pragma unroll 8
            for (int i = 0; i < k; i++) { const float u = L.cnew[i]; rank += (int)(u < v) | ((int)(u == v) & (int)(i < j)); } // (no short circuit: no branches)
            L.call[rank] = v; L.perm[rank] = (uint16_t)j;
        }
        km_lds_barrier();
    }
    KLSTAMP(8); // order
    // equal centres: the first one (lowest original index) takes every tie, the others never win: distinct values only
    int ku = 0;
    {
        const int rounds_u = (k + NT - 1) / NT;
        int carry = 0;
        for (int rd = 0; rd < rounds_u; rd++) {
            const int p = rd * NT + tid;
            float v = 0.0f;
            int o = 0, first = 0;
            if (p < k) {
                v = L.call[p]; o = L.perm[p];
                tab->perm[p] = (uint16_t)o;
                first = (p == 0) || (L.call[p - 1] != v);
            }
            const unsigned long long bal = __ballot(first);
            const int before = __popcll(bal & ((1ull << (tid & 63)) - 1ull));
            if ((tid & 63) == 0) hd->wave_i[tid >> 6] = __popcll(bal);
            km_lds_barrier();
            int pre = carry, tot = carry;
            for (int w = 0; w < NT / 64; w++) { const int wv = hd->wave_i[w]; if (w < (tid >> 6)) pre += wv; tot += wv; }
            if (first) { L.cs[pre + before] = v; L.so[pre + before] = (uint16_t)o; }
            carry = tot;
            km_lds_barrier();
        }
        ku = carry;
    }
    for (int p = tid; p < ku; p += NT) {
        const float v = L.cs[p];
        const float v2 = v * v;
        L.csq[p] = v2;
        tab->cand[p] = make_float2(v, v2);
        tab->orig[p] = L.so[p];
        ws->bnd.cand[p] = make_float2(v, v2);
        ws->bnd.orig[p] = L.so[p];
    }
    if (tid == 0) { tab->ku = ku; ws->ku_cur = ku; ws->bnd.ku = ku; tab->n_ovf = 0; ws->cells_pending = 0; hd->ku = ku; }
    KLSTAMP(9); // distinct + tables
    // zone of every centre: the x-interval on which it can be the float32 arg-min (km_finalize_body's rule and error bound)
    const double xb = fmax(fabs((double)p_lo), fabs((double)p_hi));
    for (int p = tid; p + 1 < ku; p += NT) { // the interval of every pair of neighbours once: both centres' zones end there
        const KmZone z = km_pair_zone((double)L.cs[p], (double)L.cs[p + 1], xb);
        L.Ub[p] = z.hi; L.Lb[p + 1] = z.lo;
    }
    km_lds_barrier();
    for (int p = tid; p < ku; p += NT) {
        const double cp = (double)L.cs[p];
        double right = p + 1 < ku ? L.Ub[p] : INFINITY, left = p > 0 ? L.Lb[p] : -INFINITY;
        for (int q = p + 2; q < ku; q++) {
            const double cq = (double)L.cs[q];
            const double mid = 0.5 * (cp + cq);
            const double delta = cq - cp;
            if (mid - km_pair_slack(delta, xb) >= right) break; // every later pair's interval ends further up still
            if (delta > 0.0) right = fmin(right, km_pair_zone(cp, cq, xb).hi);
        }
        for (int q = p - 2; q >= 0; q--) {
            const double cq = (double)L.cs[q];
            const double mid = 0.5 * (cp + cq);
            const double delta = cp - cq;
            if (mid + km_pair_slack(delta, xb) <= left) break;
            if (delta > 0.0) left = fmax(left, km_pair_zone(cq, cp, xb).lo);
        }
        L.Lb[p] = left; L.Ub[p] = right;
        tab->zl[p] = left; tab->zr[p] = right;
        ws->bnd.zl[p] = left; ws->bnd.zr[p] = right;
    }
    km_lds_barrier();
    KLSTAMP(10); // zones
    kl_derive<NT>(hd, L, ku);
    KLSTAMP(11); // thresholds, candidate ranges
}

// ---- the boundaries of one iteration ---------------------------------------------------------------------------------------------
// Pass 1 (kl_search).  Group g (eight lanes) takes the boundaries g, g + NT/8, ...  For boundary j (between the distinct centres j
// and j + 1, value order) it finds the two ranks a = #{x~ < L_j} and b = #{x~ <= U_j} and the prefix sums at both (fine prefix of
// the block that holds the rank + the images in front of it inside the block): A, PA, B, PB.  Usually one probe gives all four.
// Pass 2 (kl_label).  The samples float32 cannot place from the zones alone are [max(a_j, b_{j-1}), b_j) for every j: cut into chunks of
// KL_CHUNK samples, dealt round robin to the groups (so one long stretch does not hold up the workgroup and nothing is looked at
// twice), labelled with scikit-learn's exact float32 expression -- between the centres j and j + 1 (sums in registers), or, where
// three or more centres sit within rounding distance of each other, against all of j .. phi_j (first minimum, ties to the lowest
// original index) -- and added to sum_s / cnt_s by LDS atomics.
// The thresholds are float32 here (L_j rounded up, U_j rounded down): for a float32 x~, x~ < L_j <=> x~ < ceil32(L_j) and
// x~ <= U_j <=> x~ <= floor32(U_j), so the comparisons are the double ones at a quarter of the instructions.
template <int NT, typename IDX>
__device__ __forceinline__ void kl_search(const float *__restrict__ xs, const long long n_, const long long *__restrict__ pf, const long long total,
                                          const float mean, const int Sft, const int ku, KlHead *hd, const KlArr &L, KlDiag *dg)
{
    // 2 (ku - 1) independent searches "how many samples are at or below T": search 2j for a_j (T = the float32 below L_j, since
    // x~ < L_j <=> x~ <= pred32(L_j)), search 2j + 1 for b_j (T = U_j).  Each has its own hint (rank, threshold, density there).
    const int tid = threadIdx.x, g = tid >> 3, gl = tid & 7, lane = tid & 63, g0 = lane & ~7;
    const IDX n = (IDX)n_; // (IDX = int while the ranks fit)
    const int ns = 2 * (ku - 1);
    const float4 *__restrict__ x4 = reinterpret_cast<const float4 *>(xs);
    for (int s0 = 0; s0 < ns; s0 += NT / 8) {
        KLSUB(0); // (between the sub-passes)
        const int sx = s0 + g;
        const bool have = sx < ns;
        const int si = have ? sx : 0, jj = si >> 1;
        const float T = (si & 1) ? L.Uf[jj] : nextafterf(L.Lf[jj], -INFINITY);
        IDX lo = 0, hi = n;  // the rank lies in [lo, hi]: samples below lo are <= T, samples from hi on are not
        int st = have ? 0 : 2;     // 0 searching, 2 found, 3 handed over
        int mode = 1;              // 0: read the block that holds `pred`; 1: nine-way split of the bracket
        IDX pred = 0;
        {
            const long long h = L.hint[si];
            if (h >= 0 && h <= n) {
                const float r = L.hR[si];
                float shift = 0.0f;
                if (r > 0.0f && r < 3.0e38f) shift = (T - L.hL[si]) * r; // Newton step on the rank function
                if (!(shift > -1.0e9f && shift < 1.0e9f)) shift = 0.0f;
                long long pr = h + (long long)shift;
                if (pr < 0) pr = 0;
                if (pr > n_ - 1) pr = n_ - 1;
                pred = (IDX)pr; mode = 0;
            }
        }
        IDX rank = 0;
        long long pfr = 0, part = 0;
        int misses = 0;
        float rho_found = 0.0f;
        KLSUB(1); // set-up of the searches
        // ---- the steady state first: the block the hint points at (two tries, the second a Newton step from the first block).  No
        // bracket bookkeeping, a third of the instructions of the general loop below, which only the searches that miss twice
        // (and the first iteration, without hints) go through.
        for (int fr = 0; fr < 2; fr++) {
            // two blocks: the one that holds rank pred - 32 and the next (the prediction sits at least 32 samples from either end)
            const IDX b0 = (pred > 32 ? pred - 32 : 0) >> 6;
            const IDX base = b0 << 6;
            const bool go = st == 0 && mode == 0 && base + 2 * KL_BLK <= n; // (two full blocks)
            if (!__any(go)) break;
            float x[16];
            long long pfx0 = 0, pfx1 = 0;
            if (go) {
                const float4 v0 = x4[(base >> 2) + gl], v1 = x4[(base >> 2) + 8 + gl], v2 = x4[(base >> 2) + 16 + gl], v3 = x4[(base >> 2) + 24 + gl];
                x[0] = v0.x - mean; x[1] = v0.y - mean; x[2] = v0.z - mean; x[3] = v0.w - mean;
                x[4] = v1.x - mean; x[5] = v1.y - mean; x[6] = v1.z - mean; x[7] = v1.w - mean;
                x[8] = v2.x - mean; x[9] = v2.y - mean; x[10] = v2.z - mean; x[11] = v2.w - mean;
                x[12] = v3.x - mean; x[13] = v3.y - mean; x[14] = v3.z - mean; x[15] = v3.w - mean;
                pfx0 = pf[b0]; pfx1 = pf[b0 + 1];
            } else {
#pragma unroll
                for (int t = 0; t < 16; t++) x[t] = 0.0f;
            }
            int cnt = 0;
#pragma unroll
            for (int t = 0; t < 16; t++) cnt += (x[t] <= T) ? 1 : 0;
            const int cT = grp8_sum_i(cnt); // 0 .. 128
            const bool hit = go && ((cT > 0 || base == 0) && cT < 2 * KL_BLK);
            if (hit) { // the rank lies inside these two blocks: the fine prefix of its own block + the images in front of it there
                rank = base + cT; st = 2;
                const int first = cT >= KL_BLK ? KL_BLK : 0;
                pfr = cT >= KL_BLK ? pfx1 : pfx0;
#pragma unroll
                for (int t = 0; t < 16; t++) { // sample t of this lane is number 32 (t / 4) + 4 gl + t % 4 of the 128
                    const int idx = 32 * (t >> 2) + 4 * gl + (t & 3);
                    if (idx >= first && idx < cT) part += fix_f32(x[t], Sft);
                }
            }
            const bool miss = go && !hit;
            if (__any(hit | miss)) {
                const float fa = __shfl(x[0], g0), fb = __shfl(x[15], g0 + 7);
                const float rho = (fb > fa) ? 127.0f / (fb - fa) : 0.0f;
                if (hit) rho_found = rho;
                if (miss) {
                    if (cT == 0) { if (base < hi) hi = base; } else { if (base + 2 * KL_BLK > lo) lo = base + 2 * KL_BLK; }
                    misses++;
                    mode = 1;
                    if (rho > 0.0f) {
                        const float sh = (cT == 0) ? (T - fa) * rho : 127.0f + (T - fb) * rho;
                        if (sh > -1.0e9f && sh < 1.0e9f) {
                            const long long pr = (long long)base + (long long)sh;
                            if (pr >= (long long)lo && pr < (long long)hi) { mode = 0; pred = (IDX)pr; }
                        }
                    }
                    if (hi - lo <= KL_BLK) { mode = 0; pred = lo; }
                }
            }
        }
        KLSUB(2); // the lean probes
        for (int round = 0; round < KL_MAXR; round++) {
            if (!__any(st == 0)) break;
            if (st == 0 && lo == hi && lo >= n) { rank = n; pfr = total; st = 2; } // every sample is at or below T
            const bool blockmode = (st == 0 && mode == 0), scanmode = (st == 0 && mode == 1);
            float x[8];
#pragma unroll
            for (int t = 0; t < 8; t++) x[t] = 0.0f;
            IDX base = 0, w = 0;
            long long pfx = 0;
            int len = 0;
            float sv = 0.0f;
            if (blockmode) {
                if (lo == hi) pred = lo; // the bracket is closed: the block that holds the rank
                if (pred > n - 1) pred = n - 1;
                if (pred < 0) pred = 0;
                const IDX blk = pred >> 6;
                base = blk << 6;
                len = (int)((n - base) < KL_BLK ? (n - base) : KL_BLK);
                if (len == KL_BLK) {
                    const float4 v0 = x4[(base >> 2) + gl], v1 = x4[(base >> 2) + 8 + gl];
                    x[0] = v0.x; x[1] = v0.y; x[2] = v0.z; x[3] = v0.w; x[4] = v1.x; x[5] = v1.y; x[6] = v1.z; x[7] = v1.w;
                } else {
#pragma unroll
                    for (int t = 0; t < 8; t++) { const int idx = (t < 4) ? 4 * gl + t : 32 + 4 * gl + (t - 4); if (idx < len) x[t] = xs[base + idx]; }
                }
                pfx = pf[blk];
            } else if (scanmode) {
                w = hi - lo; // >= 1
                sv = xs[lo + (IDX)(((long long)w * (gl + 1)) / 9)]; // probes lo <= r_1 <= ... <= r_8 < hi
            }
            int cnt = 0;
            if (blockmode) {
#pragma unroll
                for (int t = 0; t < 8; t++) {
                    const int idx = (t < 4) ? 4 * gl + t : 32 + 4 * gl + (t - 4);
                    x[t] = x[t] - mean;
                    cnt += (idx < len && x[t] <= T) ? 1 : 0;
                }
            } else if (scanmode) { sv = sv - mean; cnt = (sv <= T) ? 1 : 0; }
            const int cT = grp8_sum_i(cnt);
            // first / last sample of the block, or the two probes either side of the cut
            const int srcA = blockmode ? g0 : g0 + (cT > 0 ? cT - 1 : 0);
            const int srcB = blockmode ? g0 + 7 : g0 + (cT < 8 ? cT : 7);
            const float fa = __shfl(blockmode ? x[0] : sv, srcA), fb = __shfl(blockmode ? x[7] : sv, srcB);
            if (blockmode) {
                bool found = false;
                if (lo == hi) found = true;
                else if (cT == 0 && base > lo) hi = base;
                else if (cT == len && base + len < hi) lo = base + len;
                else { found = true; lo = hi = base + cT; }
                const float rho = (len == KL_BLK && fb > fa) ? 63.0f / (fb - fa) : 0.0f;
                if (found) {
                    const int ct = (int)(lo - base);
                    if (ct < len) { // the block holds the rank: the prefix sum there = the block's fine prefix + the images in front of it
                        rank = lo; pfr = pfx; rho_found = rho; st = 2;
#pragma unroll
                        for (int t = 0; t < 8; t++) { const int idx = (t < 4) ? 4 * gl + t : 32 + 4 * gl + (t - 4); if (idx < ct) part += fix_f32(x[t], Sft); }
                    } // else: the rank sits right behind this block; the next round reads that one (lo == hi)
                } else {
                    misses++;
                    mode = 1;
                    if (rho > 0.0f && misses <= 2) {
                        const float sh = (cT == 0) ? (T - fa) * rho : 63.0f + (T - fb) * rho;
                        if (sh > -1.0e9f && sh < 1.0e9f) {
                            const long long pr = (long long)base + (long long)sh;
                            if (pr >= (long long)lo && pr < (long long)hi) { mode = 0; pred = (IDX)pr; }
                        }
                    }
                    if (hi - lo <= KL_BLK) { mode = 0; pred = lo; }
                }
            } else if (scanmode) {
                const IDX ra = lo + (IDX)(((long long)w * cT) / 9), rb = lo + (IDX)(((long long)w * (cT + 1)) / 9); // the last probe that passes, the first that does not
                if (cT > 0) lo = ra + 1;
                if (cT < 8) hi = rb;
                misses = 0;
                mode = 1;
                if (hi - lo <= KL_BLK) { mode = 0; pred = lo; }
                else if (cT > 0 && cT < 8 && fb > fa) {
                    double pr = (double)ra + ((double)T - (double)fa) * ((double)(rb - ra) / ((double)fb - (double)fa));
                    if (!(pr >= (double)lo)) pr = (double)lo;
                    if (!(pr <= (double)(hi - 1))) pr = (double)(hi - 1);
                    mode = 0; pred = (IDX)pr;
                }
            }
#ifdef NNC_DIAG
            if (threadIdx.x == 0) dg->ph[3] += 1; // rounds of the first wave
#endif
            if (!__any(st == 0)) break;
        }
        if (st == 0) st = 3;
        KLSUB(3); // the general rounds
        const long long PS = grp8_sum_ll(part);
        if (have && gl == 0) {
            if (st == 2) {
                if (si & 1) { L.B[jj] = (long long)rank; L.PB[jj] = pfr + PS; } else { L.A[jj] = (long long)rank; L.PA[jj] = pfr + PS; }
                // the density for the next Newton step: over the move just made where that was long enough to measure one (the 64 or
                // 128 samples of one probe give the local density to some ten per cent only), else the probe's
                float rnew = rho_found;
                {
                    const float dT = T - L.hL[si];
                    const long long dr = (long long)rank - L.hint[si];
                    if (L.hint[si] >= 0 && (dr >= 256 || dr <= -256) && dT != 0.0f) {
                        const float sec = (float)dr / dT;
                        if (sec > 0.0f && sec < 3.0e38f) rnew = sec;
                    }
                }
                L.hint[si] = (long long)rank; L.hL[si] = T; L.hR[si] = rnew;
            } else {
                hd->slow = 1;
#ifdef NNC_DIAG
                hd->pad0 = si; hd->pad1 = st; hd->pad2 = (int)(hi - lo);
#endif
            }
        }
        KLSUB(4); // results written
    }
}

#define KL_CHUNK 256  // samples of one labelling chunk (two candidates per sample)
#define KL_CHUNK_CROWD 32 // ... where three or more centres are candidates (each sample is a loop over them): spread over more groups
#define KL_QMAX 4096  // chunks one iteration may label inside the loop (a million samples); beyond that the wide pass takes the iteration
#define KL_QONE 512   // ... and so many of them from one boundary

// between the passes: the chunk list (one thread per boundary)
template <int NT>
__device__ __forceinline__ void kl_chunks(const float *__restrict__ xs, const float mean, const int Sft, const int ku, KlHead *hd, const KlArr &L)
{
    const int tid = threadIdx.x, nb = ku - 1;
    int *qfirst = reinterpret_cast<int *>(L.call); // (free between two finish steps)
    for (int j = tid; j < nb; j += NT) {
        const long long bm = j > 0 ? L.B[j - 1] : 0, a = L.A[j], e = L.B[j];
        const long long s = a > bm ? a : bm;
        const long long len = e - s;
        if (len > 4 * KL_CHUNK && xs[s] == xs[e - 1]) {
            // a long run of EQUAL values (the zeros of a pruned vector between two centres float32 cannot tell apart): one label
            // for all of them -- one evaluation, value x count (km_bounds_range has the same short cut)
            const float xc = xs[s] - mean;
            float bestd = INFINITY;
            int best = j, besto = 0x7fffffff;
            for (int cc = j; cc <= (int)L.phi[j]; cc++) {
                const float d = L.csq[cc] + (-2.0f * (xc * L.cs[cc]));
                const int oc = (int)L.so[cc];
                if (d < bestd || (d == bestd && oc < besto)) { bestd = d; best = cc; besto = oc; }
            }
            atomicAdd(reinterpret_cast<unsigned long long *>(&L.sum_s[best]), (unsigned long long)((long long)fix_f32(xc, Sft) * len));
            atomicAdd(reinterpret_cast<unsigned long long *>(&L.cnt_s[best]), (unsigned long long)len);
            qfirst[j] = -1;
            hd->r_flat = 1; // (its samples are in no chunk: an empty-cluster event of this iteration goes to the relocation chain)
        } else if (len > 0) {
            const int cs_ = (int)L.phi[j] == j + 1 ? KL_CHUNK : KL_CHUNK_CROWD;
            const long long nc = (len + cs_ - 1) / cs_;
            if (nc > KL_QONE) hd->slow = 1;
            else {
                const int first = atomicAdd(&hd->nch, (int)nc);
                qfirst[j] = first;
                for (int i = 0; i < (int)nc; i++) if (first + i < KL_QMAX) L.qj[first + i] = (uint16_t)j;
            }
        }
    }
}

template <int NT>
__device__ __forceinline__ void kl_label(const float *__restrict__ xs, const float mean, const int Sft, KlHead *hd, const KlArr &L)
{
    const int tid = threadIdx.x, g = tid >> 3, gl = tid & 7;
    const int nch = hd->nch;
    const int *qfirst = reinterpret_cast<const int *>(L.call);
    for (int c = g; c < nch; c += NT / 8) {
        const int j = (int)L.qj[c];
        const long long bm = j > 0 ? L.B[j - 1] : 0, a = L.A[j], e = L.B[j];
        const long long s = a > bm ? a : bm;
        const int phi = L.phi[j];
        const bool two = phi == j + 1;
        const int cs_ = two ? KL_CHUNK : KL_CHUNK_CROWD;
        const long long start = s + (long long)cs_ * (c - qfirst[j]);
        const long long end = start + cs_ < e ? start + cs_ : e;
        const float c0 = L.cs[j], q0 = L.csq[j], c1 = L.cs[j + 1], q1 = L.csq[j + 1];
        const bool tie1 = L.so[j + 1] < L.so[j];
        long long s0 = 0, s1 = 0;
        int n0 = 0, n1 = 0;
        for (long long r0 = start + gl; r0 < end; r0 += 64) { // eight loads a lane in flight
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) { const long long r = r0 + 8 * u; v[u] = r < end ? xs[r] : 0.0f; }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                if (r0 + 8 * u < end) {
                    const float xc = v[u] - mean;
                    const int q = fix_f32(xc, Sft);
                    if (two) {
                        const float d0 = q0 + (-2.0f * (xc * c0));
                        const float d1 = q1 + (-2.0f * (xc * c1));
                        if (d1 < d0 || (d1 == d0 && tie1)) { s1 += q; n1++; } else { s0 += q; n0++; }
                    } else {
                        float bestd = q0 + (-2.0f * (xc * c0));
                        int best = j, besto = (int)L.so[j];
                        for (int cc = j + 1; cc <= phi; cc++) {
                            const float d = L.csq[cc] + (-2.0f * (xc * L.cs[cc]));
                            const int oc = (int)L.so[cc];
                            if (d < bestd || (d == bestd && oc < besto)) { bestd = d; best = cc; besto = oc; }
                        }
                        atomicAdd(reinterpret_cast<unsigned long long *>(&L.sum_s[best]), (unsigned long long)(long long)q);
                        atomicAdd(reinterpret_cast<unsigned long long *>(&L.cnt_s[best]), 1ull);
                    }
                }
            }
        }
        const long long S0 = grp8_sum_ll(s0), S1 = grp8_sum_ll(s1);
        const int N01 = grp8_sum_i(n0 | (n1 << 16));
        if (gl == 0) {
            const int N0 = N01 & 0xFFFF, N1 = N01 >> 16;
            if (N0) { atomicAdd(reinterpret_cast<unsigned long long *>(&L.sum_s[j]), (unsigned long long)S0); atomicAdd(reinterpret_cast<unsigned long long *>(&L.cnt_s[j]), (unsigned long long)N0); }
            if (N1) { atomicAdd(reinterpret_cast<unsigned long long *>(&L.sum_s[j + 1]), (unsigned long long)S1); atomicAdd(reinterpret_cast<unsigned long long *>(&L.cnt_s[j + 1]), (unsigned long long)N1); }
        }
    }
}

// ---- empty clusters, settled without leaving the loop ------------------------------------------------------------------------------
// scikit-learn's _relocate_empty_clusters_dense (_k_means_common.pyx:167-211) for the iteration whose sums lie in sumo / cnto (original
// index order, in LDS): the i-th empty cluster (ascending index) takes the sample with the i-th largest key
// (float32 squared distance to its own centre, then value -- the keys of k_reloc_select), which leaves the cluster it was labelled
// with.  On the value-sorted vector the loop knows every sample's label from the ranks of this iteration: [b_{p-1}, a_p) is centre
// p's, the stretches in between were labelled sample by sample.  Inside a certain stretch the distance to the centre is monotone
// either side of it, so the candidates are the KL_RW samples at each end of every certain stretch and every sample of the undecided
// stretches; the selection (one more key than there are empty clusters, so that a tie at the cut shows) is right if the innermost
// candidate of every end lies STRICTLY below the cut -- the proof k_reloc_select runs on its windows, with the loop's exact ranks in
// place of windows that must be wide enough.  Returns 1 when the event is settled (sums, counts and ws->partials edited as
// km_relocate_apply edits them, status counters updated: the caller goes on with the averages), 0 when it is not (more than
// KL_RM_MAX empty clusters, too many candidates, no proof, every sample on its centre, a cluster that would be left empty): nothing
// has changed then and the caller pauses as before -- the relocation chain behind the launch, or the host, takes the event.
#define KL_RW 8      // candidates per end of a certain stretch: the lanes of a group
#define KL_RM_MAX 7  // empty clusters one event may have here (KL_RM_MAX + 1 keys are selected)
#define KL_RKPT 16   // candidate keys one thread holds during the selection
#define KL_RPASS 10  // ends of certain stretches (groups of eight lanes) one thread takes: 2 ku <= KL_RPASS * NT / 8
template <int NT>
__device__ __forceinline__ int kl_relocate(const float *__restrict__ xs, const long long n, KmWs *__restrict__ ws, KlHead *hd, const KlArr &L,
                                           long long *sumo, long long *cnto, const int k, const int nch, const float mean, const int Sft)
{
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, g = tid >> 3, gl = tid & 7;
    const int ku = hd->ku, m = hd->n_empty;
    if (tid == 0 && m > ws->kl_stats[7]) ws->kl_stats[7] = m; // (diagnostics: the largest event seen, why events were passed on)
    if (m < 1 || m > KL_RM_MAX || 2 * ku > KL_RPASS * (NT / 8) || hd->r_flat) { if (tid == 0) ws->kl_stats[6] |= 1; return 0; }
    unsigned long long *keys = ws->kl_keys;
    const int base = 2 * KL_RW * ku;
    unsigned long long *rtr = NNC_FIN_TRACE_PTR; // diagnostics: phase stamps
#define KRSTAMP(i) do { if (rtr && tid == 0) rtr[30 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
    KRSTAMP(0);
    if (tid == 0) { hd->r_cnt = 0; hd->r_bad = 0; }
    __syncthreads();
    // ---- the ends of the certain stretches: group q = (centre p, lower / upper end), one sample a lane, the outermost first
    unsigned inner = 0u; // the largest distance of an innermost candidate that has samples behind it
    {   // (at most KL_RPASS groups a thread; all their samples are fetched before any is looked at)
        long long rr[KL_RPASS];
        float xq[KL_RPASS];
        bool deep[KL_RPASS];
#pragma unroll
        for (int u = 0; u < KL_RPASS; u++) {
            const int q = g + u * (NT / 8);
            rr[u] = -1; deep[u] = false;
            if (q < 2 * ku) {
                const int p = q >> 1, side = q & 1;
                const long long lo_r = p > 0 ? L.B[p - 1] : 0, hi_r = p == ku - 1 ? n : L.A[p];
                long long r = -1;
                if (side == 0) { r = lo_r + gl; if (r >= hi_r) r = -1; }
                else { r = hi_r - 1 - gl; if (r < lo_r + KL_RW) r = -1; } // (a short stretch: its first KL_RW samples are the lower end's)
                rr[u] = r;
                deep[u] = gl == KL_RW - 1 && hi_r - lo_r > 2 * KL_RW;
            }
        }
#pragma unroll
        for (int u = 0; u < KL_RPASS; u++) xq[u] = rr[u] >= 0 ? xs[rr[u]] : 0.0f;
#pragma unroll
        for (int u = 0; u < KL_RPASS; u++) {
            const int q = g + u * (NT / 8);
            if (q < 2 * ku) {
                unsigned long long key = 0ull;
                if (rr[u] >= 0) {
                    const float dd = (xq[u] - mean) - L.cs[q >> 1];
                    const float dv = dd * dd;
                    key = ((unsigned long long)__float_as_uint(dv) << 32) | (unsigned long long)f32_ordered_bits(xq[u]);
                    if (deep[u]) inner = max(inner, __float_as_uint(dv));
                }
                keys[q * KL_RW + gl] = key;
            }
        }
    }
    KRSTAMP(1); // certain ends
    // ---- the undecided stretches, chunk by chunk as kl_label went through them: exact label, then the distance to that centre
    const int *qfirst = reinterpret_cast<const int *>(L.call);
    for (int c = g; c < nch; c += NT / 8) {
        const int j = (int)L.qj[c];
        const long long bm = j > 0 ? L.B[j - 1] : 0, a = L.A[j], e = L.B[j];
        const long long s = a > bm ? a : bm;
        const int phi = L.phi[j];
        const int cs_ = phi == j + 1 ? KL_CHUNK : KL_CHUNK_CROWD;
        const long long start = s + (long long)cs_ * (c - qfirst[j]);
        const long long end = start + cs_ < e ? start + cs_ : e;
        // (one slot range a chunk: a counter bumped once per SAMPLE -- six thousand LDS atomics on one word -- was most of this phase)
        int cbase = 0;
        if (gl == 0) cbase = atomicAdd(&hd->r_cnt, (int)(end - start));
        cbase = __shfl(cbase, (int)(threadIdx.x & 63 & ~7));
        for (long long r0 = start + gl; r0 < end; r0 += 64) { // eight loads a lane in flight
            float xq[8];
#pragma unroll
            for (int u = 0; u < 8; u++) { const long long r = r0 + 8 * u; xq[u] = r < end ? xs[r] : 0.0f; }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                if (r0 + 8 * u < end) {
                    const float xv = xq[u];
                    const float xc = xv - mean;
                    float bestd = L.csq[j] + (-2.0f * (xc * L.cs[j]));
                    int best = j, besto = (int)L.so[j];
                    for (int cc = j + 1; cc <= phi; cc++) {
                        const float d = L.csq[cc] + (-2.0f * (xc * L.cs[cc]));
                        const int oc = (int)L.so[cc];
                        if (d < bestd || (d == bestd && oc < besto)) { bestd = d; best = cc; besto = oc; }
                    }
                    const float dd = xc - L.cs[best];
                    const float dv = dd * dd;
                    const int slot = base + cbase + (int)(r0 + 8 * u - start);
                    if (slot < KL_RKEYS) keys[slot] = ((unsigned long long)__float_as_uint(dv) << 32) | (unsigned long long)f32_ordered_bits(xv);
                }
            }
        }
    }
    KRSTAMP(2); // stretch samples (this wave)
    __syncthreads();
    KRSTAMP(3);
    const int N = base + hd->r_cnt;
    if (N > KL_RKPT * NT || N > KL_RKEYS) { if (tid == 0) ws->kl_stats[6] |= 2; return 0; }
    // ---- the m + 1 largest keys: every thread holds its share in registers, one round of workgroup maximum per key
    unsigned long long kr[KL_RKPT];
#pragma unroll
    for (int i = 0; i < KL_RKPT; i++) { const int slot = tid + i * NT; kr[i] = slot < N ? keys[slot] : 0ull; }
    KRSTAMP(4); // keys back in registers
    if (tid <= KL_RM_MAX) hd->r_keys[tid] = 0ull;
#pragma unroll 1
    for (int r = 0; r <= m; r++) { // (one copy of the round's code, not eight: it runs once or twice a call and every copy arrives cold)
        {
            unsigned long long bk = 0ull;
            int bi = -1;
#pragma unroll
            for (int i = 0; i < KL_RKPT; i++) if (kr[i] > bk) { bk = kr[i]; bi = tid + i * NT; }
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) { // equal keys are interchangeable samples: the lower slot goes first
                const unsigned long long ok = (unsigned long long)__shfl_xor((long long)bk, off);
                const int oi = __shfl_xor(bi, off);
                if (ok > bk || (ok == bk && (unsigned)oi < (unsigned)bi)) { bk = ok; bi = oi; }
            }
            if (lane == 0) { hd->r_wkey[r & 1][wv] = bk; hd->r_widx[r & 1][wv] = bi; }
            __syncthreads();
            bk = 0ull; bi = -1;
#pragma unroll
            for (int w = 0; w < NT / 64; w++) {
                const unsigned long long ok = hd->r_wkey[r & 1][w];
                const int oi = hd->r_widx[r & 1][w];
                if (ok > bk || (ok == bk && (unsigned)oi < (unsigned)bi)) { bk = ok; bi = oi; }
            }
            if (tid == 0) hd->r_keys[r] = bk;
            if (bi >= 0 && (bi % NT) == tid) {
                const int mine = bi / NT;
#pragma unroll
                for (int i = 0; i < KL_RKPT; i++) if (i == mine) kr[i] = 0ull;
            }
        }
    }
    __syncthreads();
    KRSTAMP(5); // rounds
    // ---- is the selection right, and is there anything to move?
    const unsigned long long kcut = hd->r_keys[m - 1], ktop = hd->r_keys[0];
    int bad = 0;
    if (kcut == 0ull || (ktop >> 32) == 0ull) bad = 1;     // fewer candidates than empty clusters; every sample on its centre
    if (inner >= (unsigned)(kcut >> 32)) bad = 1;          // samples behind an innermost candidate might reach the cut
    if (__syncthreads_or(bad)) { if (tid == 0) ws->kl_stats[6] |= (kcut == 0ull ? 4 : 0) | ((ktop >> 32) == 0ull ? 8 : 0) | 16; return 0; }
    // ---- the empty clusters in ascending order; the cluster each selected sample leaves (the float32 arg-min over ALL centres,
    // first minimum -- km_relocate_apply)
    {
        const int rounds = (k + NT - 1) / NT;
        int carry = 0;
        for (int rd = 0; rd < rounds; rd++) {
            const int j = rd * NT + tid;
            const int e = (j < k && cnto[j] == 0) ? 1 : 0;
            const unsigned long long bal = __ballot(e);
            const int before = __popcll(bal & ((1ull << lane) - 1ull));
            if (lane == 0) hd->wave_i[wv] = __popcll(bal);
            __syncthreads();
            int pre = carry, tot = carry;
            for (int w = 0; w < NT / 64; w++) { const int c = hd->wave_i[w]; if (w < wv) pre += c; tot += c; }
            if (e && pre + before < 8) hd->r_empty[pre + before] = j;
            carry = tot;
            __syncthreads();
        }
    }
    KRSTAMP(6); // proof vote + list of the empty clusters
    for (int i = wv; i < m; i += NT / 64) { // wave-uniform
        const float xv = f32_from_ordered_bits((unsigned)(hd->r_keys[i] & 0xFFFFFFFFull));
        const float xc = xv - mean;
        float best = INFINITY;
        int old = 0x7fffffff;
        for (int j = lane; j < k; j += 64) {
            const float cv = L.cold[j];
            const float dj = cv * cv + (-2.0f * (xc * cv));
            if (dj < best) { best = dj; old = j; }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const float ob = __shfl_xor(best, off);
            const int oo = __shfl_xor(old, off);
            if (ob < best || (ob == best && oo < old)) { best = ob; old = oo; }
        }
        if (old == 0x7fffffff) old = 0;
        if (lane == 0) hd->r_old[i] = old;
    }
    __syncthreads();
    if (tid == 0) {
        int nb = 0;
        for (int i = 0; i < m; i++) { // no cluster may be left without a sample (scikit-learn does not look again; neither path here tries)
            int leave = 0;
            for (int q = 0; q < m; q++) leave += hd->r_old[q] == hd->r_old[i];
            if (cnto[hd->r_old[i]] - leave < 1) nb = 1;
        }
        hd->r_bad = nb;
        if (nb) ws->kl_stats[6] |= 32;
        if (!nb) {
            for (int i = 0; i < m; i++) {
                const float xv = f32_from_ordered_bits((unsigned)(hd->r_keys[i] & 0xFFFFFFFFull));
                const long long v = (long long)fix_f32(xv - mean, Sft);
                const int old = hd->r_old[i], nw = hd->r_empty[i];
                sumo[old] -= v; cnto[old] -= 1;
                sumo[nw] += v; cnto[nw] += 1;
            }
            for (int i = 0; i < m; i++) { // (ws->partials: what the relocation kernels edit; partials_local stays as labelled, as there)
                const int old = hd->r_old[i], nw = hd->r_empty[i];
                ws->partials[old] = sumo[old]; ws->partials[k + old] = cnto[old];
                ws->partials[nw] = sumo[nw]; ws->partials[k + nw] = cnto[nw];
            }
            // what scikit-learn leaves to numpy.argpartition (km_relocate_apply): the pairing of several, a tie at the cut
            if (m > 1) ws->st.reloc_multi += 1;
            const unsigned long long kn = hd->r_keys[m];
            if (kn != 0ull && (kn >> 32) == (kcut >> 32) && kn != kcut) ws->st.reloc_ties += 1;
            ws->st.n_relocated += 1; // (the host does not see this event: it counts them from here)
            ws->st.n_in_place += 1;
            ws->reloc_fail = 0;
            ws->kl_stats[5] += 1;
        }
    }
    __syncthreads();
    KRSTAMP(7); // old clusters + edits
#undef KRSTAMP
    return hd->r_bad ? 0 : 1;
}

// ---- the same for a MASS event (more than KL_RM_MAX empty clusters: the first iterations behind a density / forgy init, whose duplicate
// centres own nothing) ------------------------------------------------------------------------------------------------------------------
// kl_relocate finds its m + 1 keys by m + 1 rounds of a workgroup-wide maximum: fine for a handful, hopeless for the 164 / 98 / 54 ...
// empty clusters of the bench fit's first iterations, which therefore went through the chain of four launches behind the iteration
// (window table, candidates, selection, resumed finalize: about 85 us with its launch boundaries, seven times a step).  Same candidates,
// same keys, same proof here, with what a mass event needs:
//   * the keys go to the workspace (ws->kl_mkeys), not to registers;
//   * the m largest come from a cut on the distance bits (12 + 12 + 7 bits, refined while more than KL_MS_SMALL keys lie at or above
//     it) and an exact ranking of the few that survive it -- k_reloc_select's selection, on 64-bit keys;
//   * windows in two turns: KL_MW1 samples at each end of every certain stretch first (an end may have to supply all m, but the far
//     samples of a mass event sit at a few ends in the tails); the ends whose innermost candidate is not strictly below the cut --
//     and only those -- are then read to a depth above m and the selection is repeated over everything.  The cut can only rise when
//     keys are added, so ends that passed in the first turn pass in the second, and an end read deeper than m cannot hold m + 1 keys
//     at or above the m-th largest: two turns settle it (equal keys in a crowd at the cut aside, which the ranking refuses);
//   * the clusters the selected samples leave by one thread per sample; the edits of m clusters by LDS atomics (integers: any order).
// Returns 1 when the event is settled, 0 when it is not (nothing has changed then: the step pauses and the host's look-in takes it).
#define KL_MKEYS 40960   // keys of a mass event in the workspace (ws->kl_mkeys): the first turn's 2 * ku * 16 (ku <= 1032) and what joins them
#define KL_MW1L 4        // log2 of the first turn's window
#define KL_MS_MAX 2048   // keys at or above the cut that are ranked exactly
#define KL_MS_SMALL 384  // ... the cut is refined while there are more
struct KlMass {
    unsigned hist[4096];
    unsigned long long surv[KL_MS_MAX];
    unsigned long long sel[NNC_KMAX + 8];  // the selected keys, descending (one more than there are empty clusters)
    unsigned innerq[2 * NNC_KMAX];         // per end: distance bits of its innermost candidate if samples lie behind it, else 0
    uint16_t flist[2 * NNC_KMAX];          // the ends that go into the second turn
    uint8_t failq[2 * NNC_KMAX];
    uint16_t empty[NNC_KMAX], old[NNC_KMAX];
    int leave[NNC_KMAX];
    int wave_i[16];
    int cut, above, ge, nsurv, nf, cnt, over;
    unsigned thr;
};

// The cut: a threshold thr on the distance bits such that at least m and at most KL_MS_MAX of keys[0 .. N) (zero = no key) lie at or
// above it; those keys go to M->surv (any order), their number to M->nsurv, thr to M->thr.  0 if there are fewer than m keys or a
// crowd of more than KL_MS_MAX equal distances sits at the finest cut.
template <int NT>
__device__ int kl_mass_cut(const unsigned long long *__restrict__ keys, const int N, const int m, KlMass *M)
{
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    unsigned prefix = 0, thr = 0;
    int pshift = -1, above = 0, total_ge = N;
    const int shifts[3] = {19, 7, 0}, widths[3] = {12, 12, 7};
    for (int lvl = 0; lvl < 3; lvl++) {
        const int shift = shifts[lvl], width = widths[lvl];
        const unsigned mask = (1u << width) - 1u;
        for (int i = tid; i < 4096; i += NT) M->hist[i] = 0u;
        if (tid == 0) { M->cut = 0; M->above = above; M->ge = total_ge; M->nsurv = 0; }
        __syncthreads();
        for (int i0 = 0; i0 < N; i0 += 8 * NT) { // (neighbouring keys have similar distances: a wave that lands in one bin adds once)
            unsigned long long kq[8];
#pragma unroll
            for (int u = 0; u < 8; u++) { const int i = i0 + u * NT + tid; kq[u] = i < N ? keys[i] : 0ull; }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const unsigned uu = (unsigned)(kq[u] >> 32);
                const bool have = kq[u] != 0ull && (pshift < 0 || (uu >> pshift) == prefix);
                const unsigned bn = have ? (uu >> shift) & mask : 0xFFFFu;
                const unsigned long long act = __ballot(have);
                if (act) {
                    const unsigned first = (unsigned)__shfl((int)bn, __ffsll((long long)act) - 1);
                    if (__all(!have || bn == first)) { if (lane == 0) atomicAdd(&M->hist[first], (unsigned)__popcll(act)); }
                    else if (have) atomicAdd(&M->hist[bn], 1u);
                }
            }
        }
        __syncthreads();
        // thread t owns bins [t * per, (t + 1) * per); everything in higher bins via a block prefix sum
        constexpr int per = 4096 / NT > 0 ? 4096 / NT : 1;
        unsigned hb[per];
        int own = 0;
#pragma unroll
        for (int q = 0; q < per; q++) { hb[q] = (tid * per + q < 4096) ? M->hist[tid * per + q] : 0u; own += (int)hb[q]; }
        int sc = own;
        for (int off = 1; off < 64; off <<= 1) { const int t_ = __shfl_up(sc, off); if (lane >= off) sc += t_; }
        if (lane == 63) M->wave_i[wv] = sc;
        __syncthreads();
        int pre = 0, tot = 0;
        for (int w = 0; w < NT / 64; w++) { const int v = M->wave_i[w]; if (w < wv) pre += v; tot += v; }
        int run = above + (tot - (pre + sc)); // keys above this thread's bins (and above the prefix)
        for (int q = per - 1; q >= 0; q--) {
            const int before = run;
            run += (int)hb[q];
            if (before < m && run >= m) { M->cut = tid * per + q; M->above = before; M->ge = run; } // exactly one thread, one bin
        }
        __syncthreads();
        const int bcut = M->cut;
        prefix = (pshift < 0) ? (unsigned)bcut : ((prefix << width) | (unsigned)bcut);
        pshift = shift;
        above = M->above;
        total_ge = M->ge;
        thr = prefix << shift;
        __syncthreads();
        if (total_ge <= KL_MS_SMALL) break;
    }
    if (total_ge > KL_MS_MAX || total_ge < m) return 0; // a crowd of equal distances at the cut / fewer keys than empty clusters
    for (int i0 = 0; i0 < N; i0 += 8 * NT) {
        unsigned long long kq[8];
#pragma unroll
        for (int u = 0; u < 8; u++) { const int i = i0 + u * NT + tid; kq[u] = i < N ? keys[i] : 0ull; }
#pragma unroll
        for (int u = 0; u < 8; u++)
            if (kq[u] != 0ull && (unsigned)(kq[u] >> 32) >= thr) {
                const int slot = atomicAdd(&M->nsurv, 1);
                if (slot < KL_MS_MAX) M->surv[slot] = kq[u];
            }
    }
    if (tid == 0) M->thr = thr;
    __syncthreads();
    return M->nsurv >= m && M->nsurv <= KL_MS_MAX;
}

// M->surv[0 .. ms) into descending order (every thread walks the list once: the reads are LDS broadcasts), then M->sel[0 .. m] = the
// first m + 1 of them (0 where there is none: the runner-up shows a tie at the cut)
template <int NT>
__device__ void kl_mass_rank(const int ms, const int m, KlMass *M)
{
    const int tid = threadIdx.x;
    constexpr int KPT = (KL_MS_MAX + NT - 1) / NT;
    unsigned long long mine[KPT];
    int rk[KPT];
#pragma unroll
    for (int q = 0; q < KPT; q++) { const int i = tid + q * NT; mine[q] = i < ms ? M->surv[i] : 0ull; rk[q] = 0; }
    const int mine_n = (ms - tid + NT - 1) / NT; // keys this thread holds (<= 0: none)
    if (mine_n > 0) {
#pragma unroll 4
        for (int j = 0; j < ms; j++) {
            const unsigned long long kj = M->surv[j];
#pragma unroll
            for (int q = 0; q < KPT; q++) rk[q] += (kj > mine[q]) || (kj == mine[q] && j < tid + q * NT);
        }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < KPT; q++) if (tid + q * NT < ms) M->surv[rk[q]] = mine[q]; // every rank 0 .. ms-1 is taken exactly once
    __syncthreads();
    for (int r = tid; r <= m; r += NT) M->sel[r] = (r < ms) ? M->surv[r] : 0ull;
    __syncthreads();
}

template <int NT>
__device__ int kl_relocate_mass(const float *__restrict__ xs, const long long n, KmWs *__restrict__ ws, KlHead *hd, const KlArr &L, KlMass *M,
                                long long *sumo, long long *cnto, const int k, const int nch, const float mean, const int Sft)
{
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, g = tid >> 3, gl = tid & 7;
    const int ku = hd->ku, m = hd->n_empty;
    if (tid == 0 && m > ws->kl_stats[7]) ws->kl_stats[7] = m;
    unsigned long long *rtr = NNC_FIN_TRACE_PTR; // diagnostics: phase stamps, ten slots per event of the fit (by the number of events before it)
    const int ord_ = min(ws->st.n_relocated, 15);
#define KMSTAMP(i) do { if (rtr && tid == 0) rtr[100 + 10 * ord_ + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
    KMSTAMP(0);
    if (rtr && tid == 0) rtr[100 + 10 * ord_ + 9] = (unsigned long long)m;
    if (m < 1 || m >= k || hd->r_flat) return 0;
    int Wl2 = KL_MW1L;
    while ((1 << Wl2) <= m) Wl2++;                 // second turn: W2 > m
    const int W1 = 1 << KL_MW1L, W2 = 1 << Wl2;
    const int n_ends = 2 * ku * W1;
    unsigned long long *keys = ws->kl_mkeys;
    for (int i = tid; i < k; i += NT) M->leave[i] = 0;
    for (int q = tid; q < 2 * ku; q += NT) { M->innerq[q] = 0u; M->failq[q] = 0; }
    if (tid == 0) { M->over = 0; M->nf = 0; M->cnt = 0; }
    __syncthreads();
    // rank of the sample at depth i of end q (-1 if the end has no such sample); the lower end's window reaches W_lo samples up from
    // the bottom of its stretch, the upper end owns what lies above that; top_held: samples at the top of the stretch that are the
    // upper end's already (none while the lower end's first window is made, W1 when it is deepened: no sample gets two keys)
    auto end_rank = [&](int q, int i, int W_lo, int top_held, long long &lo_r, long long &hi_r) -> long long {
        const int p_ = q >> 1, side = q & 1;
        lo_r = p_ > 0 ? L.B[p_ - 1] : 0; hi_r = p_ == ku - 1 ? n : L.A[p_];
        long long r;
        if (side == 0) { r = lo_r + i; if (r >= hi_r - top_held) r = -1; }
        else { r = hi_r - 1 - i; if (r < lo_r + W_lo) r = -1; } // (a short stretch: its first W_lo samples are the lower end's)
        return r;
    };
    auto key_of = [&](float xv, float cen) -> unsigned long long {
        const float dd = (xv - mean) - cen;
        const float dv = dd * dd;
        return ((unsigned long long)__float_as_uint(dv) << 32) | (unsigned long long)f32_ordered_bits(xv);
    };
    // ---- first turn: W1 samples at each end of every certain stretch, the outermost first
    for (int s0 = 0; s0 < n_ends; s0 += 8 * NT) { // (eight loads a thread in flight)
        long long rr[8];
        float xq[8];
        bool deep[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int sl = s0 + u * NT + tid;
            rr[u] = -1; deep[u] = false;
            if (sl < n_ends) {
                const int q = sl >> KL_MW1L, i = sl & (W1 - 1);
                long long lo_r, hi_r;
                rr[u] = end_rank(q, i, W1, 0, lo_r, hi_r);
                deep[u] = i == W1 - 1 && hi_r - lo_r > 2ll * W1;
            }
        }
#pragma unroll
        for (int u = 0; u < 8; u++) xq[u] = rr[u] >= 0 ? xs[rr[u]] : 0.0f;
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int sl = s0 + u * NT + tid;
            unsigned long long key = 0ull;
            if (rr[u] >= 0) {
                const int q = sl >> KL_MW1L;
                key = key_of(xq[u], L.cs[q >> 1]);
                if (deep[u]) M->innerq[q] = (unsigned)(key >> 32);
            }
            if (sl < n_ends) keys[sl] = key;
        }
    }
    __syncthreads(); // (also a fence over the keys in the workspace: every thread reads other threads' keys below)
    KMSTAMP(1); // ends of the certain stretches
    // ---- the cut over the ends alone: a lower bound of the final cut (more keys can only raise the m-th largest)
    if (n_ends > KL_MKEYS) { if (tid == 0) ws->kl_stats[6] |= 2; return 0; }
    unsigned thr = 0u;
    int Ncut = n_ends;
  for (int turn = 0;; turn++) { // (ONE call site of the cut: with two the compiler (ROCm 7.2) dies on the LDS pointer it shares between them)
    if (!kl_mass_cut<NT>(keys, Ncut, m, M)) { if (tid == 0) ws->kl_stats[6] |= 64; return 0; }
    if (turn == 1) break; // (the cut over everything, made again because many keys joined the survivors of the first one)
    thr = M->thr;
    kl_mass_rank<NT>(M->nsurv, m, M);
    KMSTAMP(2); // first selection
    // ---- which ends are not deep enough?  (their innermost candidate has samples behind it and is not strictly below the cut)
    {
        const unsigned cutb = (unsigned)(M->sel[m - 1] >> 32);
        for (int q = tid; q < 2 * ku; q += NT)
            if (M->innerq[q] != 0u && M->innerq[q] >= cutb) { M->failq[q] = 1; M->flist[atomicAdd(&M->nf, 1)] = (uint16_t)q; }
    }
    __syncthreads();
    const int nf = M->nf;
    if (rtr && tid == 0) rtr[100 + 10 * ord_ + 8] = ((unsigned long long)n_ends << 32) | (unsigned long long)nf;
    // whatever is found from here on joins the survivors if it lies at or above the first threshold (nothing below it can be among
    // the m largest of the larger set)
    // (they are also appended to the keys in the workspace: should the survivors become many -- a coarse first threshold --, the cut
    // is made again over everything instead of ranking a crowd)
    auto add_surv = [&](unsigned long long key) {
        if (key != 0ull && (unsigned)(key >> 32) >= thr) {
            const int at = n_ends + atomicAdd(&M->cnt, 1);
            if (at < KL_MKEYS) keys[at] = key;
            if (at >= KL_MKEYS) atomicOr(&M->over, 1); // (an atomic: the compiler folds a plain LDS store and the global store above into one flat store and trips over it)
            const int slot = atomicAdd(&M->nsurv, 1);
            if (slot < KL_MS_MAX) M->surv[slot] = key;
        }
    };
    if (nf > 0) {
        // ---- second turn: those ends to a depth above m
        if (W2 <= W1) { if (tid == 0) ws->kl_stats[6] |= 16; return 0; } // (W1 > m already: more than m keys at or above the m-th largest -- a crowd of equal ones)
        const int extra = W2 - W1;
        const int n2 = nf * extra;
        for (int s0 = 0; s0 < n2; s0 += 8 * NT) {
            long long rr[8];
            float xq[8];
            bool deep[8];
            int qq[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int sl = s0 + u * NT + tid;
                rr[u] = -1; deep[u] = false; qq[u] = 0;
                if (sl < n2) {
                    const int f = sl / extra, i = W1 + sl % extra;
                    const int q = (int)M->flist[f];
                    qq[u] = q;
                    const int W_lo = M->failq[q & ~1] ? W2 : W1; // how far the lower end of this stretch reaches now
                    const int W_up = M->failq[q | 1] ? W2 : W1;
                    long long lo_r, hi_r;
                    rr[u] = end_rank(q, i, W_lo, W1, lo_r, hi_r);
                    deep[u] = i == W2 - 1 && hi_r - lo_r > (long long)W_lo + W_up;
                }
            }
#pragma unroll
            for (int u = 0; u < 8; u++) xq[u] = rr[u] >= 0 ? xs[rr[u]] : 0.0f;
#pragma unroll
            for (int u = 0; u < 8; u++) {
                if (rr[u] >= 0) {
                    const unsigned long long key = key_of(xq[u], L.cs[qq[u] >> 1]);
                    if (deep[u]) M->innerq[qq[u]] = (unsigned)(key >> 32); // (the end's innermost candidate is this one now)
                    add_surv(key);
                }
            }
        }
        // (an end of the second turn with nothing left behind its window is read to its end)
        for (int f = tid; f < nf; f += NT) {
            const int q = (int)M->flist[f];
            long long lo_r, hi_r;
            const int W_lo = M->failq[q & ~1] ? W2 : W1, W_up = M->failq[q | 1] ? W2 : W1;
            (void)end_rank(q, 0, W_lo, 0, lo_r, hi_r);
            if (!(hi_r - lo_r > (long long)W_lo + W_up)) M->innerq[q] = 0u;
        }
        if (tid == 0) ws->kl_stats[6] |= 128; // (diagnostics: an event that took the second turn)
    }
    KMSTAMP(3); // second turn
    // ---- the undecided stretches, chunk by chunk as kl_label went through them.  A sample's key is the distance to ITS centre, one of
    // the candidates j .. phi of its stretch, so it is at most the distance to the farther of the two outermost candidates: only
    // samples for which that bound reaches the threshold get the exact label (after a mass relocation thousands of samples sit
    // between crowded centres -- a loop over dozens of candidates each -- and next to none of them is far from its centre)
    const int *qfirst = reinterpret_cast<const int *>(L.call);
    for (int c = g; c < nch; c += NT / 8) {
        const int j = (int)L.qj[c];
        const long long bm = j > 0 ? L.B[j - 1] : 0, a = L.A[j], e = L.B[j];
        const long long s = a > bm ? a : bm;
        const int phi = L.phi[j];
        const int cs_ = phi == j + 1 ? KL_CHUNK : KL_CHUNK_CROWD;
        const long long start = s + (long long)cs_ * (c - qfirst[j]);
        const long long end = start + cs_ < e ? start + cs_ : e;
        const float cj = L.cs[j], cp = L.cs[phi];
        for (long long r0 = start + gl; r0 < end; r0 += 64) { // eight loads a lane in flight
            float xq[8];
#pragma unroll
            for (int u = 0; u < 8; u++) { const long long r = r0 + 8 * u; xq[u] = r < end ? xs[r] : 0.0f; }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                if (r0 + 8 * u < end) {
                    const float xv = xq[u];
                    const float xc = xv - mean;
                    const float da = xc - cj, db = xc - cp;
                    const float ub = fmaxf(da * da, db * db);
                    if (__float_as_uint(ub) >= thr) {
                        float bestd = L.csq[j] + (-2.0f * (xc * L.cs[j]));
                        int best = j, besto = (int)L.so[j];
                        for (int cc = j + 1; cc <= phi; cc++) {
                            const float d = L.csq[cc] + (-2.0f * (xc * L.cs[cc]));
                            const int oc = (int)L.so[cc];
                            if (d < bestd || (d == bestd && oc < besto)) { bestd = d; best = cc; besto = oc; }
                        }
                        add_surv(key_of(xv, L.cs[best]));
                    }
                }
            }
        }
    }
    __syncthreads();
    KMSTAMP(4); // undecided stretches
    if (M->over) { if (tid == 0) ws->kl_stats[6] |= 2; return 0; }
    if (M->nsurv <= KL_MS_SMALL) break;
    Ncut = n_ends + M->cnt; // many joined: a finer threshold over all keys rather than the ranking of a crowd
    __syncthreads();
  }
    kl_mass_rank<NT>(M->nsurv, m, M);
    {
        const unsigned cutb = (unsigned)(M->sel[m - 1] >> 32);
        int deep_bad = 0;
        for (int q = tid; q < 2 * ku; q += NT) if (M->innerq[q] != 0u && M->innerq[q] >= cutb) deep_bad = 1;
        if (__syncthreads_or(deep_bad)) { if (tid == 0) ws->kl_stats[6] |= 16; return 0; }
    }
    const unsigned long long kcut = M->sel[m - 1], ktop = M->sel[0];
    if (kcut == 0ull || (ktop >> 32) == 0ull) { if (tid == 0) ws->kl_stats[6] |= (kcut == 0ull ? 4 : 0) | ((ktop >> 32) == 0ull ? 8 : 0); return 0; } // fewer candidates than empty clusters; every sample on its centre
    // ---- the empty clusters in ascending order
    {
        const int rounds = (k + NT - 1) / NT;
        int carry = 0;
        for (int rd = 0; rd < rounds; rd++) {
            const int j = rd * NT + tid;
            const int e = (j < k && cnto[j] == 0) ? 1 : 0;
            const unsigned long long bal = __ballot(e);
            const int before = __popcll(bal & ((1ull << lane) - 1ull));
            if (lane == 0) M->wave_i[wv] = __popcll(bal);
            __syncthreads();
            int pre = carry, tot = carry;
            for (int w = 0; w < NT / 64; w++) { const int c = M->wave_i[w]; if (w < wv) pre += c; tot += c; }
            if (e && pre + before < NNC_KMAX) M->empty[pre + before] = (uint16_t)j;
            carry = tot;
            __syncthreads();
        }
    }
    KMSTAMP(5); // final ranking, proof, list of the empty clusters
    // ---- the cluster each selected sample leaves: the float32 arg-min over ALL centres, first minimum (km_relocate_apply); a thread a
    // sample (every thread reads the same centre at the same time: LDS broadcasts, eight in flight)
    for (int i = tid; i < m; i += NT) {
        const float xv = f32_from_ordered_bits((unsigned)(M->sel[i] & 0xFFFFFFFFull));
        const float xc = xv - mean;
        float best = INFINITY;
        int old = 0;
        int j = 0;
        for (; j + 8 <= k; j += 8) {
            float cv[8];
#pragma unroll
            for (int u = 0; u < 8; u++) cv[u] = L.cold[j + u];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const float dj = cv[u] * cv[u] + (-2.0f * (xc * cv[u]));
                if (dj < best) { best = dj; old = j + u; }
            }
        }
        for (; j < k; j++) {
            const float cv = L.cold[j];
            const float dj = cv * cv + (-2.0f * (xc * cv));
            if (dj < best) { best = dj; old = j; }
        }
        M->old[i] = (uint16_t)old;
        atomicAdd(&M->leave[old], 1);
    }
    __syncthreads();
    KMSTAMP(6); // old clusters
    // no cluster may be left without a sample (scikit-learn does not look again; neither path here tries)
    int nb = 0;
    for (int i = tid; i < m; i += NT) { const int o = M->old[i]; if (cnto[o] - M->leave[o] < 1) nb = 1; }
    if (__syncthreads_or(nb)) { if (tid == 0) ws->kl_stats[6] |= 32; return 0; }
    // ---- the edits: integers, any order
    for (int i = tid; i < m; i += NT) {
        const float xv = f32_from_ordered_bits((unsigned)(M->sel[i] & 0xFFFFFFFFull));
        const long long v = (long long)fix_f32(xv - mean, Sft);
        const int old = M->old[i], nw = M->empty[i];
        atomicAdd(reinterpret_cast<unsigned long long *>(&sumo[old]), (unsigned long long)(-v));
        atomicAdd(reinterpret_cast<unsigned long long *>(&cnto[old]), (unsigned long long)(-1ll));
        atomicAdd(reinterpret_cast<unsigned long long *>(&sumo[nw]), (unsigned long long)v);
        atomicAdd(reinterpret_cast<unsigned long long *>(&cnto[nw]), 1ull);
    }
    __syncthreads();
    for (int i = tid; i < m; i += NT) { // (ws->partials: what the relocation kernels edit; partials_local stays as labelled, as there)
        const int old = M->old[i], nw = M->empty[i];
        ws->partials[old] = sumo[old]; ws->partials[k + old] = cnto[old];
        ws->partials[nw] = sumo[nw]; ws->partials[k + nw] = cnto[nw];
    }
    if (tid == 0) {
        // what scikit-learn leaves to numpy.argpartition (km_relocate_apply): the pairing of several, a tie at the cut
        if (m > 1) ws->st.reloc_multi += 1;
        const unsigned long long kn = M->sel[m];
        if (kn != 0ull && (kn >> 32) == (kcut >> 32) && kn != kcut) ws->st.reloc_ties += 1;
        ws->st.n_relocated += 1; // (the host does not see this event: it counts them from here)
        ws->st.n_in_place += 1;
        ws->reloc_fail = 0;
        ws->kl_stats[5] += 1;
    }
    __syncthreads();
    KMSTAMP(7); // edits
#undef KMSTAMP
    return 1;
}

// ---- the M-step and everything behind it: 0 = go on, 1 = stopped (done), 2 = paused for empty clusters --------------------------
// second half (from sums and counts in original index order, no cluster empty): _average_centers, _center_shift, the tolerance
// test, the tables of the new centres
template <int NT>
__device__ __forceinline__ int kl_finish_tail(KmWs *__restrict__ ws, KlHead *hd, const KlArr &L, const long long *sumo, const long long *cnto, const int k,
                                              int &cur, int &iter, const int Sft, const int max_iter, const float tol_v, const float p_lo,
                                              const float p_hi, KlDiag *dg)
{
    const int tid = threadIdx.x;
    for (int j = tid; j < k; j += NT) {
        const float c = (float)ldexp((double)sumo[j] / (double)cnto[j], -Sft);
        L.cnew[j] = c;
        const float d = c - L.cold[j];
        const float s2 = d * d;
        const float sft = (float)sqrt((double)s2);
        L.sq[j] = sft * sft;
    }
    km_lds_barrier();
    kl_pairwise<NT>(hd, L.sq, k);
    const float tot = hd->tot;
    iter += 1;
    int done = 0;
    if (tot <= tol_v) done = 1;
    else if (iter >= max_iter) done = 2;
    cur ^= 1;
    if (tid == 0) {
        ws->st.iter = iter; ws->st.shift_tot = tot; ws->st.done = done;
        ws->st.paused = 0; ws->st.n_empty = 0;
        ws->cur = cur;
    }
    for (int j = tid; j < k; j += NT) { const float c = L.cnew[j]; ws->c[cur][j] = c; L.cold[j] = c; }
    km_lds_barrier();
    KLSTAMP(7); // shift, pairwise sum, state
    kl_tables<NT>(ws, hd, L, k, cur, false, p_lo, p_hi, dg);
    return done ? 1 : 0;
}

template <int NT>
__device__ __forceinline__ int kl_finish(const float *__restrict__ xs, KmWs *__restrict__ ws, KlHead *hd, const KlArr &L, const long long n, const long long total,
                                         const int k, const int kc, int &cur, int &iter, const float mean, const int Sft, const int max_iter,
                                         const float tol_v, const float p_lo, const float p_hi, const int nch, const int reloc, KlDiag *dg)
{
    const int tid = threadIdx.x;
    const int ku = hd->ku;
    long long *sumo = L.PA, *cnto = L.PB; // original index order (the prefix sums are spent once they are in registers; the ranks A / B stay for kl_relocate)
    // ---- certain stretches: [b_{p-1}, a_p) is centre p's; its sum is a difference of prefix sums
    long long rs[2], rc[2];
    int ro[2];
#pragma unroll
    for (int r = 0; r < 2; r++) {
        const int p = tid + r * NT;
        rs[r] = 0; rc[r] = 0; ro[r] = -1;
        if (p < ku) {
            const long long lo_r = p > 0 ? L.B[p - 1] : 0, plo = p > 0 ? L.PB[p - 1] : 0;
            const long long hi_r = p == ku - 1 ? n : L.A[p], phi = p == ku - 1 ? total : L.PA[p];
            long long s = L.sum_s[p], c = L.cnt_s[p];
            if (hi_r > lo_r) { s += phi - plo; c += hi_r - lo_r; }
            rs[r] = s; rc[r] = c; ro[r] = (int)L.so[p];
        }
    }
    __syncthreads();
    for (int j = tid; j < kc; j += NT) { sumo[j] = 0; cnto[j] = 0; L.sum_s[j] = 0; L.cnt_s[j] = 0; } // duplicates of a centre own nothing
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 2; r++) if (ro[r] >= 0) { sumo[ro[r]] = rs[r]; cnto[ro[r]] = rc[r]; }
    __syncthreads();
    // ---- could the labels equal the previous iteration's?  (only if every cluster kept its count)  Empty clusters?
    int diff = 0, my_empty = 0;
    for (int j = tid; j < k; j += NT) {
        const long long c = cnto[j], sm = sumo[j];
        diff |= (L.prevc[j] != c);
        L.prevc[j] = c;
        my_empty += (c == 0);
        // (what the relocation kernels edit, and what nnc_kmeans_partials shows: the sums / counts of the iteration, as k_finalize leaves them)
        ws->partials[j] = sm; ws->partials[k + j] = c;
        ws->partials_local[j] = sm; ws->partials_local[k + j] = c;
    }
    KLSTAMP(5); // combine + scatter
    const int any_diff = __syncthreads_or(diff);
    const int any_empty = __syncthreads_or(my_empty);
    KLSTAMP(6); // votes
    if (tid == 0) ws->st.same_counts = any_diff ? 0 : 1;
    if (any_empty) {
        if (tid == 0) hd->n_empty = 0;
        __syncthreads();
        if (my_empty) atomicAdd(&hd->n_empty, my_empty);
        __syncthreads();
        // (with every count unchanged the host's strict-convergence check is due first: km_spec_decide)
        if (!(reloc && !(iter >= 1 && !any_diff) && kl_relocate<NT>(xs, n, ws, hd, L, sumo, cnto, k, nch, mean, Sft))) {
            if (tid == 0) { ws->st.paused = 1; ws->st.n_empty = hd->n_empty; ws->tab[cur].n_ovf = 0; }
            return 2;
        }
    }
    return kl_finish_tail<NT>(ws, hd, L, sumo, cnto, k, cur, iter, Sft, max_iter, tol_v, p_lo, p_hi, dg);
}

// ---- the same step when nothing unusual happens: every centre distinct (one thread per cluster, k <= NT), no empty cluster, the
// order of the centres unchanged.  Then nothing has to change places: thread p owns the p-th centre in value order from the sums
// to the zones, and the step needs seven workgroup barriers instead of some forty -- barriers that order the LDS traffic only
// (km_lds_barrier): a __syncthreads() would also wait for the global stores of the step (state for the other kernels, write-only here).  Returns -1 (before it has changed anything
// that the general step would not change in the same way) when that does not hold; else 0 = go on, 1 = stopped, 2 = paused.
template <int NT>
__device__ __forceinline__ int kl_finish_fast(const float *__restrict__ xs, KmWs *__restrict__ ws, KlHead *hd, const KlArr &L, const long long n, const long long total,
                                              const int k, int &cur, int &iter, const float mean, const int Sft, const int max_iter, const float tol_v,
                                              const float p_lo, const float p_hi, const int nch, const int reloc, KlDiag *dg)
{
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int p = tid;
    const bool mine = p < k;
    // ---- this cluster's sum and count: the certain stretch [b_{p-1}, a_p) from the prefix sums + what the labelling added
    long long sm = 0, c = 0;
    int o = 0;
    float cn = 0.0f;
    int empty = 0, diff = 0;
    if (mine) {
        const long long lo_r = p > 0 ? L.B[p - 1] : 0, plo = p > 0 ? L.PB[p - 1] : 0;
        const long long hi_r = p == k - 1 ? n : L.A[p], phi_ = p == k - 1 ? total : L.PA[p];
        sm = L.sum_s[p]; c = L.cnt_s[p];
        if (hi_r > lo_r) { sm += phi_ - plo; c += hi_r - lo_r; }
        L.sum_s[p] = 0; L.cnt_s[p] = 0;
        o = (int)L.so[p];
        empty = c == 0;
        diff = L.prevc[o] != c;
        L.prevc[o] = c;
        ws->partials[o] = sm; ws->partials[k + o] = c;
        ws->partials_local[o] = sm; ws->partials_local[k + o] = c;
        if (!empty) {
            cn = (float)ldexp((double)sm / (double)c, -Sft);
            const float d = cn - L.cold[o];
            const float s2 = d * d;
            const float sft = (float)sqrt((double)s2);
            L.sq[o] = sft * sft;
            L.cnew[o] = cn;
        }
    }
    if (__any(empty) && lane == 0) hd->f_empty = 1;
    if (__any(diff) && lane == 0) hd->f_diff = 1;
    km_lds_barrier();                                                                                   // ---- 1
    KLSTAMP(5);
    const int any_empty = hd->f_empty, any_diff = hd->f_diff;
    if (tid == 0) ws->st.same_counts = any_diff ? 0 : 1;
    if (any_empty) {
        // the relocation here (kl_relocate), then the general second half; or pause: the relocation chain (or the host) takes over,
        // nothing else of the state has changed
        long long *sumo = L.PA, *cnto = L.PB; // (spent: their values are in sm / c)
        if (tid == 0) hd->n_empty = 0;
        __syncthreads();
        if (empty) atomicAdd(&hd->n_empty, 1);
        if (mine) { sumo[o] = sm; cnto[o] = c; } // k distinct centres: every index is some thread's
        __syncthreads();
        if (tid == 0) { hd->f_empty = 0; hd->f_diff = 0; }
        if (!(reloc && !(iter >= 1 && !any_diff) && kl_relocate<NT>(xs, n, ws, hd, L, sumo, cnto, k, nch, mean, Sft))) {
            if (tid == 0) { ws->st.paused = 1; ws->st.n_empty = hd->n_empty; ws->tab[cur].n_ovf = 0; }
            return 2;
        }
        return kl_finish_tail<NT>(ws, hd, L, sumo, cnto, k, cur, iter, Sft, max_iter, tol_v, p_lo, p_hi, dg);
    }
    // ---- NumPy's pairwise sum of the squared shifts (leaves now, eight lanes each; the first wave folds them behind the barrier),
    // and: is the order of the centres still the same, all of them distinct?
    KlHeap *hp = &hd->heap;
    if (k <= LEAF) {
        if (tid < 8) { const float r = kl_leaf_sum(L.sq, 0, k, tid & 7); if (tid == 0) hd->tot = r; }
    } else {
        const int depth = hd->depth;
        for (int node = 1 + (tid >> 3); node < (2 << depth) && node < 64; node += NT / 8) {
            const int l = hp->len[node];
            if (l > 0 && l <= LEAF) { const float r = kl_leaf_sum(L.sq, hp->start[node], l, tid & 7); if ((tid & 7) == 0) hp->val[node] = r; }
        }
    }
    int ok = 1;
    if (mine && p + 1 < k) ok = cn < L.cnew[L.so[p + 1]]; // strictly: equal centres would have to be merged
    if (!__all(ok) && lane == 0) hd->f_reorder = 1;

    km_lds_barrier();                                                                                   // ---- 2
    KLSTAMP(6);
    if (tid < 64 && k > LEAF) {
        for (int lev = hd->depth - 1; lev >= 0; lev--) {
            const int i = (1 << lev) + lane;
            if (lane < (1 << lev) && hp->len[i] > LEAF) hp->val[i] = hp->val[2 * i] + hp->val[2 * i + 1];
            wave_lds_fence();
        }
        if (lane == 0) hd->tot = hp->val[1];
    }
    const int reorder = hd->f_reorder;
    cur ^= 1;
    iter += 1;
    if (mine) { ws->c[cur][o] = cn; L.cold[o] = cn; }
    if (reorder && tid == 0) ws->kl_stats[2] += 1; // (iterations in which centres changed places)
    if (reorder) { // the general way for the tables (and only for them: the sums, the shift and the new centres are settled)
        __syncthreads();
        const float tot = hd->tot;
        int done = 0;
        if (tot <= tol_v) done = 1;
        else if (iter >= max_iter) done = 2;
        if (tid == 0) {
            ws->st.iter = iter; ws->st.shift_tot = tot; ws->st.done = done; ws->st.paused = 0; ws->st.n_empty = 0; ws->cur = cur;
            hd->f_reorder = 0; hd->f_diff = 0;
        }
        __syncthreads();
        kl_tables<NT>(ws, hd, L, k, cur, false, p_lo, p_hi, dg);
        return done ? 1 : 0;
    }
    // ---- the tables of the new centres: same order, same indices
    KmTab *tab = &ws->tab[cur];
    if (mine) {
        const float v2 = cn * cn;
        L.cs[p] = cn; L.csq[p] = v2;
        tab->perm[p] = (uint16_t)o; tab->cand[p] = make_float2(cn, v2); tab->orig[p] = (uint16_t)o;
        ws->bnd.cand[p] = make_float2(cn, v2); ws->bnd.orig[p] = (uint16_t)o;
    }
    if (tid == 0) { tab->ku = k; ws->ku_cur = k; ws->bnd.ku = k; tab->n_ovf = 0; ws->cells_pending = 0; hd->f_diff = 0; }
    km_lds_barrier();                                                                                   // ---- 3
    KLSTAMP(7);
    const float tot = hd->tot;
    int done = 0;
    if (tot <= tol_v) done = 1;
    else if (iter >= max_iter) done = 2;
    if (tid == 0) { ws->st.iter = iter; ws->st.shift_tot = tot; ws->st.done = done; ws->st.paused = 0; ws->st.n_empty = 0; ws->cur = cur; }
    // ---- zones (km_pair_zone), then U_p = the largest zone end of the centres up to p, S_p = the smallest one from p on: scans inside
    // the wave by shuffles, across the waves through sixteen LDS words
    const double xb = fmax(fabs((double)p_lo), fabs((double)p_hi));
    double left = -INFINITY, right = INFINITY;
    if (mine && p + 1 < k) { // the interval of every pair of neighbours once: both centres' zones end there
        const KmZone z = km_pair_zone((double)cn, (double)L.cs[p + 1], xb);
        L.Ub[p] = z.hi; L.Lb[p + 1] = z.lo;
    }
    km_lds_barrier();
    if (mine) {
        const double cp = (double)cn;
        if (p + 1 < k) right = L.Ub[p];
        if (p > 0) left = L.Lb[p];
        for (int q = p + 2; q < k; q++) {
            const double cq = (double)L.cs[q];
            const double mid = 0.5 * (cp + cq);
            const double delta = cq - cp;
            if (mid - km_pair_slack(delta, xb) >= right) break;
            if (delta > 0.0) right = fmin(right, km_pair_zone(cp, cq, xb).hi);
        }
        for (int q = p - 2; q >= 0; q--) {
            const double cq = (double)L.cs[q];
            const double mid = 0.5 * (cp + cq);
            const double delta = cp - cq;
            if (mid + km_pair_slack(delta, xb) <= left) break;
            if (delta > 0.0) left = fmax(left, km_pair_zone(cq, cp, xb).lo);
        }
        tab->zl[p] = left; tab->zr[p] = right;
        ws->bnd.zl[p] = left; ws->bnd.zr[p] = right;
    }
    double um = mine ? right : -INFINITY, sn = mine ? left : INFINITY; // (neutral where there is no centre)
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const double a = __shfl_up(um, off), b = __shfl_down(sn, off);
        if (lane >= off) um = fmax(um, a);
        if (lane + off < 64) sn = fmin(sn, b);
    }
    if (lane == 63) hd->wmax[wv] = um;
    if (lane == 0) hd->wmin[wv] = sn;
    km_lds_barrier();                                                                                   // ---- 4
    KLSTAMP(10);
    for (int w = 0; w < NT / 64; w++) {
        if (w < wv) um = fmax(um, hd->wmax[w]);
        if (w > wv) sn = fmin(sn, hd->wmin[w]);
    }
    if (mine) {
        // thresholds of the boundaries either side of this centre: U_p (boundary p, above it) and L_{p-1} = S_p (boundary p - 1, below it)
        L.Ub[p] = um;
        float uf = (float)um;
        if ((double)uf > um) uf = nextafterf(uf, -INFINITY);
        L.Uf[p] = uf;
        if (p > 0) {
            L.Lb[p - 1] = sn;
            float lf = (float)sn;
            if ((double)lf < sn) lf = nextafterf(lf, INFINITY);
            L.Lf[p - 1] = lf;
        }
        if (p == k - 1) { L.Lb[p] = INFINITY; L.Lf[p] = INFINITY; }
    }
    if (tid == 0) hd->ku = k;
    km_lds_barrier();                                                                                   // ---- 5
    if (mine && p + 1 < k) { // phi_p: the highest centre that can still win below U_p
        int q = p + 1;
        while (q + 1 < k && L.Lb[q] <= um) q++;
        L.phi[p] = (uint16_t)q;
    }
    km_lds_barrier();                                                                                   // ---- 6
    KLSTAMP(11);
    return done ? 1 : 0;
}

// budget_set >= 0: this launch opens a host call that may run that many iterations; < 0: it carries on with what is left
template <int NT>
__global__ __launch_bounds__(NT) void k_lloyd(const float *__restrict__ xs, long long n, KmWs *__restrict__ ws, const long long *__restrict__ pblk,
                                              int budget_set, int kc, nnc_kmeans_status *host_st, unsigned long long *host_ticket,
                                              unsigned long long ticket, int reloc)
{
    extern __shared__ __align__(16) unsigned char kl_smem[];
#ifdef NNC_DIAG
    KlDiag dgs, *dg = &dgs;
    for (int q = 0; q < 16; q++) dgs.ph[q] = 0;
    for (int q = 0; q < 8; q++) dgs.sp[q] = 0;
    dgs.slast = 0;
    dgs.last = __builtin_amdgcn_s_memrealtime();
    const unsigned long long clk0 = __builtin_amdgcn_s_memtime(), rt0 = dgs.last;
#else
    KlDiag *dg = nullptr;
#endif
    KlHead *hd;
    KlArr L;
    kl_carve(kl_smem, kc, &hd, &L);
    const int tid = threadIdx.x;
    // ---- one round of loads: the state block
    const int st_done = ws->st.done, st_paused = ws->st.paused;
    int iter = ws->st.iter, cur = ws->cur;
    const int k = ws->p.k, Sft = ws->p.fix_shift, max_iter = ws->p.max_iter;
    const float tol_v = ws->p.tol, mean = ws->p.x_mean, p_lo = ws->p.lo, p_hi = ws->p.hi;
    const int ku0 = ws->bnd.ku, wide = ws->wide;
    int budget = budget_set >= 0 ? budget_set : ws->kl_budget;
    const long long *__restrict__ pf = pblk + km_pfine_off(n);
    const bool run = !(st_done | st_paused) && !wide && budget > 0 && k <= kc && k <= 2 * NT;
    if (run) {
        const long long total = pf[4 * km_prefix_nblk(n)];
        // ---- second round: the current centres and their tables
        const KmTab *tab = &ws->tab[cur];
        for (int j = tid; j < kc; j += NT) { L.sum_s[j] = 0; L.cnt_s[j] = 0; }
        for (int j = tid; j < k; j += NT) {
            L.cold[j] = ws->c[cur][j];
            L.perm[j] = tab->perm[j];
            L.prevc[j] = ws->prev_counts[j];
            L.hint[2 * j] = ws->hint_a[j]; L.hint[2 * j + 1] = ws->hint_b[j];
            L.hL[2 * j] = ws->kl_hL[2 * j]; L.hL[2 * j + 1] = ws->kl_hL[2 * j + 1];
            L.hR[2 * j] = ws->kl_hR[2 * j]; L.hR[2 * j + 1] = ws->kl_hR[2 * j + 1];
        }
        for (int p = tid; p < ku0; p += NT) {
            const float2 c = ws->bnd.cand[p];
            L.cs[p] = c.x; L.csq[p] = c.y; L.so[p] = ws->bnd.orig[p];
            L.Lb[p] = ws->bnd.zl[p]; L.Ub[p] = ws->bnd.zr[p];
        }
        if (tid == 0) { hd->ku = ku0; hd->slow = 0; hd->nch = 0; hd->depth = 0; hd->f_empty = 0; hd->f_diff = 0; hd->f_reorder = 0; }
        if (tid < 64 && k > LEAF) kl_heap_build(hd, k);
        __syncthreads();
        kl_derive<NT>(hd, L, ku0);
        int ran = 0;
        KLSTAMP(0); // prologue
        for (;;) {
            if (budget <= 0) break;
#ifdef NNC_DIAG
            if (tid == 0) dg->slast = __builtin_amdgcn_s_memrealtime();
#endif
            if (tid == 0) hd->r_flat = 0; // (set by kl_chunks behind the next barrier, read by the finish step of this iteration)
            if (n < (1ll << 31) - 256) kl_search<NT, int>(xs, n, pf, total, mean, Sft, hd->ku, hd, L, dg);
            else kl_search<NT, long long>(xs, n, pf, total, mean, Sft, hd->ku, hd, L, dg);
            KLSTAMP(1); // this wave's searches
            __syncthreads();
            KLSTAMP(2); // ... until the last group is through
            if (!hd->slow) kl_chunks<NT>(xs, mean, Sft, hd->ku, hd, L);
            __syncthreads();
            if (hd->slow || hd->nch > KL_QMAX) { if (tid == 0) { ws->wide = 1; ws->help_hint = 1; hd->slow = 1; } break; } // (a long stretch: every wave of the wide pass had better look at the tile queue)
            kl_label<NT>(xs, mean, Sft, hd, L);
            KLSTAMP(14); // this wave's labelling
            const int nch = hd->nch; // (the chunk list stays as it is until the next kl_chunks: kl_relocate goes through it again)
            __syncthreads();
            if (tid == 0) hd->nch = 0;
            KLSTAMP(15); // ... until the last group is through
            int r = -1;
            if (hd->ku == k && k <= NT) r = kl_finish_fast<NT>(xs, ws, hd, L, n, total, k, cur, iter, mean, Sft, max_iter, tol_v, p_lo, p_hi, nch, reloc, dg);
            else r = kl_finish<NT>(xs, ws, hd, L, n, total, k, kc, cur, iter, mean, Sft, max_iter, tol_v, p_lo, p_hi, nch, reloc, dg);
            budget--; ran++;
            if (r) break;
        }
        // ---- what the other kernels (and the next launch of this one) pick up from the workspace
        __syncthreads();
        if (tid == 0) { // where the iterations of this fit ran (nnc_kmeans_loop_stats)
            ws->kl_stats[0] += ran;                                     // iterations (or attempts that paused) run by the loop
            ws->kl_stats[1] += 1;                                       // launches that had something to do
            if (ran == 0 || !(ws->st.done | ws->st.paused)) {
                if (hd->slow) ws->kl_stats[3] += 1;                     // handed over: a long undecided stretch / a search that did not settle
#ifdef NNC_DIAG
                (void)0;
#endif
            }
        }
        for (int j = tid; j < k; j += NT) {
            ws->hint_a[j] = L.hint[2 * j]; ws->hint_b[j] = L.hint[2 * j + 1];
            ws->kl_hL[2 * j] = L.hL[2 * j]; ws->kl_hL[2 * j + 1] = L.hL[2 * j + 1];
            ws->kl_hR[2 * j] = L.hR[2 * j]; ws->kl_hR[2 * j + 1] = L.hR[2 * j + 1];
            if (ran) ws->prev_counts[j] = L.prevc[j];
        }
    }
    if (tid == 0) ws->kl_budget = budget;
#ifdef NNC_DIAG
    KLSTAMP(4); // epilogue
    dg->ph[12] = __builtin_amdgcn_s_memtime() - clk0; dg->ph[13] = __builtin_amdgcn_s_memrealtime() - rt0; // shader cycles, 10 ns ticks
    if (tid == 0) { for (int q = 0; q < 16; q++) ws->kl_trace[q] += dg->ph[q]; for (int q = 0; q < 8; q++) ws->kl_trace[16 + q] += dg->sp[q]; }
#endif
    if (host_st) {
        __syncthreads();
        if (tid == 0) {
            *host_st = ws->st;
            __threadfence_system();
            *reinterpret_cast<volatile unsigned long long *>(host_ticket) = ticket;
        }
    }
}
