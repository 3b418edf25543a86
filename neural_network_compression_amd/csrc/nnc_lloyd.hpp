// nnc_lloyd.hpp -- the Lloyd loop of one fit inside ONE workgroup (textually included by nnc_hip.hip, which owns KmWs and the helpers).
//
// The two-launch iteration (k_bounds: one wave per cluster boundary, sums by global atomics; k_finalize: one workgroup) spends
// its time in launch boundaries and dependent round trips to memory, not in work: at K = 257 on 25 M weights an iteration reads
// about 3 MB and takes 24 us.  Here one resident workgroup runs iteration after iteration until the fit stops, pauses for an
// empty cluster or meets something it hands to the wide path:
//   * everything K-sized lives in LDS: centres, zone ends, per-cluster sums and counts, the previous label counts, and for every
//     boundary where it was found last time (rank, threshold, local density);
//   * a boundary is looked up by a group of EIGHT lanes.  One probe = the 64 samples (256 B, two lines) around the predicted rank
//     plus the 8-byte fine prefix in front of them (nnc_kmeans_prefix_build).  The prediction is a Newton step on the rank
//     function (rank moves by density x threshold shift); from the second Lloyd iteration on it lands in the right block almost
//     always, so an iteration costs about three cache lines per boundary -- which matters more than latency here: one CU takes in
//     a few hundred lines per microsecond.  A miss is followed by another Newton step from the block just read, or by a nine-way
//     split of the bracket (eight single-sample probes), so the search is logarithmic whatever the data look like;
//   * the block that holds the boundary also gives the prefix sum at the rank (fine prefix + the images below the cut) and the
//     few undecided samples above it, which are labelled with scikit-learn's exact float32 expression between the two centres;
//   * the finalize step (average, shift, NumPy's pairwise sum, tolerance test, centre order, zones) runs on the same LDS arrays.
// What it does not do itself: empty clusters (status.paused, as k_finalize reports them: the relocation chain takes over) and
// iterations in which three or more centres sit within float32 rounding of each other or an undecided stretch is long
// (ws->wide: the k_bounds / k_finalize pair enqueued behind every launch of the loop runs exactly then).
// Same integers, same float32 operations as the two-launch form: the trajectory is bit-identical (tests/test_gpu_lloyd.py).
//
// Reference: the loop of sklearn/cluster/_kmeans.py:_kmeans_single_lloyd (624-752), reached from utility.py:237-238.

#define KL_MAXR 24      // probes per boundary before it is handed to the wide path (never reached on monotone data: <= 20)
#define KL_TAIL_MAX 256 // undecided samples per boundary the group labels itself

struct KlHeap { int start[64]; int len[64]; float val[64]; };
struct KlHead {
    int ku, slow, crowd, n_empty, depth, pad0, pad1, pad2;
    float tot;
    int wave_i[16];
    int pad3[3];
    KlHeap heap;
};
static_assert(sizeof(KlHead) % 16 == 0, "the arrays behind the header are 16-byte aligned");

struct KlArr {
    double *Lb, *Ub;                                                   // per boundary j: L_j, U_j (per centre while the zones are being made: zl, zr)
    long long *sum_s, *cnt_s, *A, *B, *PA, *PB, *prevc, *hint;          // A/B double as the per-cluster sums / counts in original index order
    float *hL, *hR, *cs, *csq, *cnew, *cold, *sq, *call;
    uint16_t *so, *perm;
};

static size_t kl_lds_bytes(int kc) { return sizeof(KlHead) + (size_t)kc * (2 * 8 + 8 * 8 + 8 * 4 + 2 * 2); }

__device__ __forceinline__ void kl_carve(unsigned char *smem, int kc, KlHead **hd, KlArr *L)
{
    *hd = reinterpret_cast<KlHead *>(smem);
    unsigned char *p = smem + sizeof(KlHead);
    auto take = [&](size_t bytes) { unsigned char *q = p; p += bytes; return q; };
    L->Lb = reinterpret_cast<double *>(take((size_t)kc * 8)); L->Ub = reinterpret_cast<double *>(take((size_t)kc * 8));
    L->sum_s = reinterpret_cast<long long *>(take((size_t)kc * 8)); L->cnt_s = reinterpret_cast<long long *>(take((size_t)kc * 8));
    L->A = reinterpret_cast<long long *>(take((size_t)kc * 8)); L->B = reinterpret_cast<long long *>(take((size_t)kc * 8));
    L->PA = reinterpret_cast<long long *>(take((size_t)kc * 8)); L->PB = reinterpret_cast<long long *>(take((size_t)kc * 8));
    L->prevc = reinterpret_cast<long long *>(take((size_t)kc * 8)); L->hint = reinterpret_cast<long long *>(take((size_t)kc * 8));
    L->hL = reinterpret_cast<float *>(take((size_t)kc * 4)); L->hR = reinterpret_cast<float *>(take((size_t)kc * 4));
    L->cs = reinterpret_cast<float *>(take((size_t)kc * 4)); L->csq = reinterpret_cast<float *>(take((size_t)kc * 4));
    L->cnew = reinterpret_cast<float *>(take((size_t)kc * 4)); L->cold = reinterpret_cast<float *>(take((size_t)kc * 4));
    L->sq = reinterpret_cast<float *>(take((size_t)kc * 4)); L->call = reinterpret_cast<float *>(take((size_t)kc * 4));
    L->so = reinterpret_cast<uint16_t *>(take((size_t)kc * 2)); L->perm = reinterpret_cast<uint16_t *>(take((size_t)kc * 2));
}

// sums over a group of eight lanes (every lane of the group gets the total)
__device__ __forceinline__ int grp8_sum_i(int v) { v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); return v; }
__device__ __forceinline__ long long grp8_sum_ll(long long v) { v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); return v; }

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
        __syncthreads();
        if (tid < 64) {
            for (int lev = depth - 1; lev >= 0; lev--) {
                const int i = (1 << lev) + lane;
                if (lane < (1 << lev) && hp->len[i] > LEAF) hp->val[i] = hp->val[2 * i] + hp->val[2 * i + 1];
                wave_lds_fence();
            }
            if (lane == 0) hd->tot = hp->val[1];
        }
    }
    __syncthreads();
}

// ---- zone ends per centre (zl in Lb[], zr in Ub[]) -> thresholds per boundary, and whether any three centres crowd --------------
//   U_j = max over q <= j of zr[q] (above it no centre up to j can win), L_j = min over q > j of zl[q] (below it none above j can).
//   Boundary j is plain when only the centres j and j + 1 can win between L_j and U_j:  L_{j+1} > U_j.
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
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 2; r++) {
                const int q = tid + r * NT;
                if (q < ku) { L.Ub[q] = a[r]; L.Lb[q] = b[r]; }
            }
            __syncthreads();
        }
    }
    double t[2];
#pragma unroll
    for (int r = 0; r < 2; r++) { const int q = tid + r * NT; t[r] = (q + 1 < ku) ? L.Lb[q + 1] : INFINITY; }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 2; r++) { const int q = tid + r * NT; if (q < ku) L.Lb[q] = t[r]; }
    __syncthreads();
    int crowd = 0;
    for (int j = tid; j + 2 < ku; j += NT) crowd |= (L.Lb[j + 1] <= L.Ub[j]) ? 1 : 0;
    const int any = __syncthreads_or(crowd);
    if (tid == 0) hd->crowd = any;
    __syncthreads();
}

// ---- order of the centres, distinct values, zones: what km_finalize_body leaves for the next E-step, from cnew[] in LDS --------
// (perm[] holds the order of the previous iteration; `fresh`: it holds nothing yet)
template <int NT>
__device__ __forceinline__ void kl_tables(KmWs *__restrict__ ws, KlHead *hd, const KlArr &L, const int k, const int cur, const bool fresh,
                                          const float p_lo, const float p_hi)
{
    const int tid = threadIdx.x;
    KmTab *tab = &ws->tab[cur];
    int still_sorted = 0;
    if (!fresh) {
        for (int p = tid; p < k; p += NT) L.call[p] = L.cnew[L.perm[p]];
        __syncthreads();
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
                __syncthreads();
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
            for (int i = 0; i < k; i++) { const float u = L.cnew[i]; rank += (u < v) || (u == v && i < j); }
            L.call[rank] = v; L.perm[rank] = (uint16_t)j;
        }
        __syncthreads();
    }
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
            __syncthreads();
            int pre = carry, tot = carry;
            for (int w = 0; w < NT / 64; w++) { const int wv = hd->wave_i[w]; if (w < (tid >> 6)) pre += wv; tot += wv; }
            if (first) { L.cs[pre + before] = v; L.so[pre + before] = (uint16_t)o; }
            carry = tot;
            __syncthreads();
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
    // zone of every centre: the x-interval on which it can be the float32 arg-min (km_finalize_body's rule and error bound)
    const double U = 5.9604644775390625e-08; // 2^-24
    const double xb = fmax(fabs((double)p_lo), fabs((double)p_hi));
    for (int p = tid; p < ku; p += NT) {
        const double cp = (double)L.cs[p];
        double right = INFINITY, left = -INFINITY;
        for (int q = p + 1; q < ku; q++) {
            const double cq = (double)L.cs[q];
            const double mid = 0.5 * (cp + cq);
            if (mid >= right) break; // every later midpoint is larger still
            const double delta = cq - cp;
            if (delta > 0.0) {
                const double cm = fmax(fabs(cp), fabs(cq));
                const double E = 2.5 * U * (cm * cm + 2.0 * xb * cm) + 1e-42;
                right = fmin(right, mid + E * km_rcp_up(delta));
            }
        }
        for (int q = p - 1; q >= 0; q--) {
            const double cq = (double)L.cs[q];
            const double mid = 0.5 * (cp + cq);
            if (mid <= left) break;
            const double delta = cp - cq;
            if (delta > 0.0) {
                const double cm = fmax(fabs(cp), fabs(cq));
                const double E = 2.5 * U * (cm * cm + 2.0 * xb * cm) + 1e-42;
                left = fmax(left, mid - E * km_rcp_up(delta));
            }
        }
        L.Lb[p] = left; L.Ub[p] = right;
        tab->zl[p] = left; tab->zr[p] = right;
        ws->bnd.zl[p] = left; ws->bnd.zr[p] = right;
    }
    __syncthreads();
    kl_derive<NT>(hd, L, ku);
}

// ---- the boundaries of one iteration ---------------------------------------------------------------------------------------------
// Group g (eight lanes) takes the boundaries g, g + NT/8, ...  For boundary j (between the distinct centres j and j + 1, value order):
//   a = #{x~ < L_j}, b = #{x~ <= U_j}, the prefix sums at both ranks, and the samples [a, b) labelled and added to the sums.
// Results: A, PA, B, PB; the sums of the undecided samples go to sum_s / cnt_s by LDS atomics.
template <int NT>
__device__ __forceinline__ void kl_boundaries(const float *__restrict__ xs, const long long n, const long long *__restrict__ pf, const long long total,
                                               const float mean, const int Sft, const int ku, KlHead *hd, const KlArr &L)
{
    const int tid = threadIdx.x, g = tid >> 3, gl = tid & 7, lane = tid & 63, g0 = lane & ~7;
    const int nb = ku - 1;
    const float4 *__restrict__ x4 = reinterpret_cast<const float4 *>(xs);
    for (int j0 = 0; j0 < nb; j0 += NT / 8) {
        const int j = j0 + g;
        const bool have = j < nb;
        const int jj = have ? j : 0;
        const double Lt = L.Lb[jj], Ut = L.Ub[jj];
        const float c0 = L.cs[jj], q0 = L.csq[jj], c1 = L.cs[jj + 1], q1 = L.csq[jj + 1];
        const bool tie1 = L.so[jj + 1] < L.so[jj];
        long long lo = 0, hi = n;  // the rank a lies in [lo, hi]: samples below lo are < L_j, samples from hi on are not
        int st = have ? 0 : 2;     // 0 searching, 1 labelling on into the next block, 2 done, 3 handed over
        int mode = 1;              // 0: read the block that holds `pred`; 1: nine-way split of the bracket
        long long pred = 0;
        {
            const long long h = L.hint[jj];
            if (h >= 0 && h <= n) {
                const float r = L.hR[jj];
                double pr = (double)h;
                if (r > 0.0f && r < 3.0e38f) pr += ((double)(float)Lt - (double)L.hL[jj]) * (double)r; // Newton step on the rank function
                if (!(pr >= 0.0)) pr = 0.0;
                if (!(pr <= (double)(n - 1))) pr = (double)(n - 1);
                pred = (long long)pr; mode = 0;
            }
        }
        long long a = 0, b = 0, pfa = 0;           // results (group-uniform)
        long long s0 = 0, s1 = 0, pa_part = 0;     // this lane's share
        int n0 = 0, n1 = 0, und = 0, misses = 0;
        float rho_found = 0.0f;
        for (int round = 0; round < KL_MAXR; round++) {
            if (st == 0 && lo == hi && lo >= n) { a = n; b = n; pfa = total; st = 2; } // every sample is below L_j
            const bool blockmode = (st == 1) || (st == 0 && mode == 0);
            const bool scanmode = (st == 0 && mode == 1);
            float x[8];
#pragma unroll
            for (int t = 0; t < 8; t++) x[t] = 0.0f;
            long long base = 0, pfx = 0, r_i = 0;
            int len = 0;
            float sv = 0.0f;
            if (blockmode) {
                if (st == 0 && lo == hi) pred = lo; // the bracket is closed: the block that holds rank a
                if (pred > n - 1) pred = n - 1;
                if (pred < 0) pred = 0;
                const long long blk = pred >> 6;
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
                const long long w = hi - lo; // >= 1
                r_i = lo + (w * (gl + 1)) / 9; // lo <= r_i < hi, non-decreasing over the lanes
                sv = xs[r_i];
            }
            // ---- counts of the block (below L_j: low byte; at or below U_j: next byte) or of the eight probes
            int cnt = 0;
            if (blockmode) {
#pragma unroll
                for (int t = 0; t < 8; t++) {
                    const int idx = (t < 4) ? 4 * gl + t : 32 + 4 * gl + (t - 4);
                    const double xd = (double)(x[t] - mean);
                    const bool valid = idx < len;
                    cnt += (valid && xd < Lt) ? 1 : 0;
                    cnt += (valid && xd <= Ut) ? 256 : 0;
                }
            } else if (scanmode) cnt = ((double)(sv - mean) < Lt) ? 1 : 0;
            cnt = grp8_sum_i(cnt);
            const int cL = cnt & 0xFF, cE = cnt >> 8;
            // first / last sample of the block, or the two probes either side of the cut
            const int srcA = blockmode ? g0 : g0 + (cL > 0 ? cL - 1 : 0);
            const int srcB = blockmode ? g0 + 7 : g0 + (cL < 8 ? cL : 7);
            const float mine_a = blockmode ? (x[0] - mean) : (sv - mean), mine_b = blockmode ? (x[7] - mean) : (sv - mean);
            const float fa = __shfl(mine_a, srcA), fb = __shfl(mine_b, srcB);
            const long long ra = __shfl(r_i, srcA), rb = __shfl(r_i, srcB);
            if (blockmode && st < 2) {
                int ca = 0;        // samples of this block that are certainly below the boundary
                bool label = false;
                if (st == 1) label = true;
                else {
                    bool found = false;
                    if (lo == hi) found = true;
                    else if (cL == 0 && base > lo) hi = base;
                    else if (cL == len && base + len < hi) lo = base + len;
                    else { found = true; lo = hi = base + cL; }
                    if (found) {
                        ca = (int)(lo - base);
                        if (ca < len) { // the block holds rank a: prefix sum there, and the labelling starts
                            a = lo; pfa = pfx; label = true;
                            rho_found = (len == KL_BLK && fb > fa) ? 63.0f / (fb - fa) : 0.0f;
#pragma unroll
                            for (int t = 0; t < 8; t++) {
                                const int idx = (t < 4) ? 4 * gl + t : 32 + 4 * gl + (t - 4);
                                if (idx < ca) pa_part += fix_f32(x[t] - mean, Sft);
                            }
                        } // else: a sits right behind this block; the next round reads that one (lo == hi)
                    } else {
                        misses++;
                        const float rho = (len == KL_BLK && fb > fa) ? 63.0f / (fb - fa) : 0.0f;
                        mode = 1;
                        if (rho > 0.0f && misses <= 2) {
                            const double pr = (cL == 0) ? (double)base + (Lt - (double)fa) * (double)rho
                                                        : (double)(base + 63) + (Lt - (double)fb) * (double)rho;
                            if (pr >= (double)lo && pr < (double)hi) { mode = 0; pred = (long long)pr; }
                        }
                        if (hi - lo <= KL_BLK) { mode = 0; pred = lo; }
                    }
                }
                if (label) {
                    const int ce = cE > ca ? cE : ca;
#pragma unroll
                    for (int t = 0; t < 8; t++) {
                        const int idx = (t < 4) ? 4 * gl + t : 32 + 4 * gl + (t - 4);
                        if (idx >= ca && idx < ce) {
                            const float xc = x[t] - mean;
                            const float d0 = q0 + (-2.0f * (xc * c0));
                            const float d1 = q1 + (-2.0f * (xc * c1));
                            const int q = fix_f32(xc, Sft);
                            if (d1 < d0 || (d1 == d0 && tie1)) { s1 += q; n1++; } else { s0 += q; n0++; }
                        }
                    }
                    und += ce - ca;
                    if (ce >= len && base + len < n) { st = 1; pred = base + len; if (und > KL_TAIL_MAX) st = 3; }
                    else { b = base + ce; st = 2; }
                }
            } else if (scanmode) {
                if (cL > 0) lo = ra + 1;
                if (cL < 8) hi = rb;
                misses = 0;
                mode = 1;
                if (hi - lo <= KL_BLK) { mode = 0; pred = lo; }
                else if (cL > 0 && cL < 8 && fb > fa) {
                    double pr = (double)ra + (Lt - (double)fa) * ((double)(rb - ra) / ((double)fb - (double)fa));
                    if (!(pr >= (double)lo)) pr = (double)lo;
                    if (!(pr <= (double)(hi - 1))) pr = (double)(hi - 1);
                    mode = 0; pred = (long long)pr;
                }
            }
            if (!__any(st < 2)) break;
        }
        if (st < 2) st = 3;
        // ---- the group's totals; its first lane writes them down
        const long long PAs = grp8_sum_ll(pa_part), S0 = grp8_sum_ll(s0), S1 = grp8_sum_ll(s1);
        const int N01 = grp8_sum_i(n0 | (n1 << 16));
        if (have && gl == 0) {
            if (st == 2) {
                const long long pa = pfa + PAs;
                L.A[j] = a; L.PA[j] = pa; L.B[j] = b; L.PB[j] = pa + S0 + S1;
                const int N0 = N01 & 0xFFFF, N1 = N01 >> 16;
                if (N0) { atomicAdd(reinterpret_cast<unsigned long long *>(&L.sum_s[j]), (unsigned long long)S0); atomicAdd(reinterpret_cast<unsigned long long *>(&L.cnt_s[j]), (unsigned long long)N0); }
                if (N1) { atomicAdd(reinterpret_cast<unsigned long long *>(&L.sum_s[j + 1]), (unsigned long long)S1); atomicAdd(reinterpret_cast<unsigned long long *>(&L.cnt_s[j + 1]), (unsigned long long)N1); }
                L.hint[j] = a; L.hL[j] = (float)Lt; L.hR[j] = rho_found;
            } else hd->slow = 1;
        }
    }
}

// ---- the M-step and everything behind it: 0 = go on, 1 = stopped (done), 2 = paused for empty clusters --------------------------
template <int NT>
__device__ __forceinline__ int kl_finish(KmWs *__restrict__ ws, KlHead *hd, const KlArr &L, const long long n, const long long total, const int k,
                                         const int kc, int &cur, int &iter, const int Sft, const int max_iter, const float tol_v, const float p_lo,
                                         const float p_hi)
{
    const int tid = threadIdx.x;
    const int ku = hd->ku;
    long long *sumo = L.A, *cnto = L.B; // original index order (the boundary results are spent once they are in registers)
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
        const long long c = cnto[j];
        diff |= (L.prevc[j] != c);
        L.prevc[j] = c;
        my_empty += (c == 0);
    }
    const int any_diff = __syncthreads_or(diff);
    const int any_empty = __syncthreads_or(my_empty);
    if (tid == 0) ws->st.same_counts = any_diff ? 0 : 1;
    if (any_empty) {
        if (tid == 0) hd->n_empty = 0;
        __syncthreads();
        if (my_empty) atomicAdd(&hd->n_empty, my_empty);
        __syncthreads();
        for (int j = tid; j < k; j += NT) {
            ws->partials[j] = sumo[j]; ws->partials[k + j] = cnto[j];
            ws->partials_local[j] = sumo[j]; ws->partials_local[k + j] = cnto[j];
        }
        if (tid == 0) { ws->st.paused = 1; ws->st.n_empty = hd->n_empty; ws->tab[cur].n_ovf = 0; }
        return 2;
    }
    // ---- _average_centers, _center_shift, the tolerance test
    for (int j = tid; j < k; j += NT) {
        const float c = (float)ldexp((double)sumo[j] / (double)cnto[j], -Sft);
        L.cnew[j] = c;
        const float d = c - L.cold[j];
        const float s2 = d * d;
        const float sft = (float)sqrt((double)s2);
        L.sq[j] = sft * sft;
    }
    __syncthreads();
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
    __syncthreads();
    kl_tables<NT>(ws, hd, L, k, cur, false, p_lo, p_hi);
    return done ? 1 : 0;
}

// budget_set >= 0: this launch opens a host call that may run that many iterations; < 0: it carries on with what is left
template <int NT>
__global__ __launch_bounds__(NT) void k_lloyd(const float *__restrict__ xs, long long n, KmWs *__restrict__ ws, const long long *__restrict__ pblk,
                                              int budget_set, int kc, nnc_kmeans_status *host_st, unsigned long long *host_ticket,
                                              unsigned long long ticket)
{
    extern __shared__ __align__(16) unsigned char kl_smem[];
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
            L.hint[j] = ws->hint_a[j]; L.hL[j] = ws->kl_hL[j]; L.hR[j] = ws->kl_hR[j];
        }
        for (int p = tid; p < ku0; p += NT) {
            const float2 c = ws->bnd.cand[p];
            L.cs[p] = c.x; L.csq[p] = c.y; L.so[p] = ws->bnd.orig[p];
            L.Lb[p] = ws->bnd.zl[p]; L.Ub[p] = ws->bnd.zr[p];
        }
        if (tid == 0) { hd->ku = ku0; hd->slow = 0; hd->crowd = 0; hd->depth = 0; }
        if (tid < 64 && k > LEAF) kl_heap_build(hd, k);
        __syncthreads();
        kl_derive<NT>(hd, L, ku0);
        int ran = 0;
        for (;;) {
            if (budget <= 0) break;
            if (hd->crowd) { if (tid == 0) ws->wide = 1; break; }
            kl_boundaries<NT>(xs, n, pf, total, mean, Sft, hd->ku, hd, L);
            __syncthreads();
            if (hd->slow) { if (tid == 0) ws->wide = 1; break; }
            const int r = kl_finish<NT>(ws, hd, L, n, total, k, kc, cur, iter, Sft, max_iter, tol_v, p_lo, p_hi);
            budget--; ran++;
            if (r) break;
        }
        // ---- what the other kernels (and the next launch of this one) pick up from the workspace
        __syncthreads();
        for (int j = tid; j < k; j += NT) {
            const long long h = L.hint[j];
            ws->hint_a[j] = h; ws->hint_b[j] = h; ws->kl_hL[j] = L.hL[j]; ws->kl_hR[j] = L.hR[j];
            if (ran) ws->prev_counts[j] = L.prevc[j];
        }
    }
    if (tid == 0) ws->kl_budget = budget;
    if (host_st) {
        __syncthreads();
        if (tid == 0) {
            *host_st = ws->st;
            __threadfence_system();
            *reinterpret_cast<volatile unsigned long long *>(host_ticket) = ticket;
        }
    }
}
