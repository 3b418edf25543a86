// nnc_comm.hip -- the multi-GPU exchange inside the library: RCCL over xGMI (bound by dlopen), the sharded iteration and the sharded fit.
#include "nnc_common.hpp"
#include "nnc_km_shared.hpp"

// ======================================================================================
// 6. Multi-GPU exchange inside the library: RCCL over xGMI
//
// One process per GPU.  The only data-path exchange of a sharded fit is the all-reduce (SUM) of the
// 2K int64 per-cluster sums / counts between the streaming pass and the finalize step of every Lloyd
// iteration (about 4 KB at K = 256: latency bound), plus, per empty-cluster event, the verdict word and
// the ranks' farthest-sample keys.  All of it is enqueued here on the caller's stream, back to back with
// the kernels: no host round trip per iteration, no second stream, no event.
//
// RCCL is bound at run time (dlopen): a process that never shards a vector never loads it, and one that
// runs beside PyTorch shares the librccl PyTorch has already mapped instead of a second copy.
// ======================================================================================
#include <dlfcn.h>
#include <rccl/rccl.h>

struct RcclApi {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
static RcclApi g_rccl;
static std::mutex g_rccl_mu;

static int rccl_load()
{
    std::lock_guard<std::mutex> lock(g_rccl_mu);
    if (g_rccl.handle) return NNC_OK;
    void *h = nullptr;
    // the copy that is already mapped (PyTorch's), else the system's
    for (const char *name : {"librccl.so", "librccl.so.1"}) { if ((h = dlopen(name, RTLD_NOW | RTLD_NOLOAD))) break; }
    if (!h) for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) { if ((h = dlopen(name, RTLD_NOW | RTLD_GLOBAL))) break; }
    if (!h) return fail(NNC_ENODEV, std::string("cannot load librccl: ") + dlerror());
    RcclApi a;
    a.handle = h;
#define RSYM(field, sym) do { a.field = reinterpret_cast<decltype(a.field)>(dlsym(h, sym)); if (!a.field) return fail(NNC_ENODEV, std::string("librccl lacks ") + sym); } while (0)
    RSYM(GetUniqueId, "ncclGetUniqueId"); RSYM(CommInitRank, "ncclCommInitRank"); RSYM(CommDestroy, "ncclCommDestroy");
    RSYM(AllReduce, "ncclAllReduce"); RSYM(AllGather, "ncclAllGather"); RSYM(GetErrorString, "ncclGetErrorString");
#undef RSYM
    g_rccl = a;
    return NNC_OK;
}

#define RCCLCHK(expr)                                                                                  \
    do {                                                                                               \
        ncclResult_t r_ = (expr);                                                                      \
        if (r_ != ncclSuccess) return fail(NNC_EHIP, std::string(#expr) + ": " + g_rccl.GetErrorString(r_)); \
    } while (0)

struct NncComm { ncclComm_t comm; int rank, world; };

// NNC_OK if librccl can be bound in this process (nothing is created): every rank asks before any of them enters
// nnc_comm_init, which blocks until all ranks have entered it.
extern "C" int nnc_comm_available(void) { return rccl_load(); }

extern "C" int nnc_comm_unique_id(void *id_out, size_t len)
{
    if (!id_out || len < NNC_COMM_ID_BYTES) return fail(NNC_EINVAL, "nnc_comm_unique_id: buffer shorter than NNC_COMM_ID_BYTES");
    static_assert(NNC_COMM_ID_BYTES == sizeof(ncclUniqueId), "NNC_COMM_ID_BYTES");
    int rc = rccl_load();
    if (rc) return rc;
    ncclUniqueId id;
    RCCLCHK(g_rccl.GetUniqueId(&id));
    std::memcpy(id_out, &id, sizeof(id));
    return NNC_OK;
}

extern "C" int nnc_comm_init(void **comm_out, const void *id, size_t len, int32_t rank, int32_t world)
{
    if (!comm_out || !id || len < NNC_COMM_ID_BYTES || world < 1 || rank < 0 || rank >= world) return fail(NNC_EINVAL, "nnc_comm_init: bad argument");
    int rc = rccl_load();
    if (rc) return rc;
    ncclUniqueId uid;
    std::memcpy(&uid, id, sizeof(uid));
    NncComm *c = new NncComm{nullptr, rank, world};
    ncclResult_t r = g_rccl.CommInitRank(&c->comm, world, uid, rank);   // on the calling thread's current device
    if (r != ncclSuccess) { delete c; return fail(NNC_EHIP, std::string("ncclCommInitRank: ") + g_rccl.GetErrorString(r)); }
    *comm_out = c;
    return NNC_OK;
}

extern "C" int nnc_comm_destroy(void *comm)
{
    if (!comm) return NNC_OK;
    NncComm *c = reinterpret_cast<NncComm *>(comm);
    if (c->comm) RCCLCHK(g_rccl.CommDestroy(c->comm));
    delete c;
    return NNC_OK;
}

extern "C" int nnc_comm_rank(void *comm) { return comm ? reinterpret_cast<NncComm *>(comm)->rank : -1; }
extern "C" int nnc_comm_world(void *comm) { return comm ? reinterpret_cast<NncComm *>(comm)->world : -1; }

extern "C" int nnc_comm_allreduce(void *comm, void *buf_dev, int64_t count, int32_t dtype, int32_t op, void *stream)
{
    if (!comm || count < 0 || (count > 0 && !buf_dev)) return fail(NNC_EINVAL, "nnc_comm_allreduce: bad argument");
    if (count == 0) return NNC_OK;
    ncclDataType_t dt;
    switch (dtype) { case NNC_I64: dt = ncclInt64; break; case NNC_I32: dt = ncclInt32; break; case NNC_F32: dt = ncclFloat32; break;
    default: return fail(NNC_EINVAL, "nnc_comm_allreduce: dtype"); }
    ncclRedOp_t ro;
    switch (op) { case NNC_SUM: ro = ncclSum; break; case NNC_MAX: ro = ncclMax; break; case NNC_MIN: ro = ncclMin; break;
    default: return fail(NNC_EINVAL, "nnc_comm_allreduce: op"); }
    RCCLCHK(g_rccl.AllReduce(buf_dev, buf_dev, (size_t)count, dt, ro, reinterpret_cast<NncComm *>(comm)->comm, S(stream)));
    return NNC_OK;
}

extern "C" int nnc_comm_allgather(void *comm, const void *send_dev, void *recv_dev, int64_t bytes_per_rank, void *stream)
{
    if (!comm || bytes_per_rank < 0 || (bytes_per_rank > 0 && (!send_dev || !recv_dev))) return fail(NNC_EINVAL, "nnc_comm_allgather: bad argument");
    if (bytes_per_rank == 0) return NNC_OK;
    RCCLCHK(g_rccl.AllGather(send_dev, recv_dev, (size_t)bytes_per_rank, ncclInt8, reinterpret_cast<NncComm *>(comm)->comm, S(stream)));
    return NNC_OK;
}

// `iters` Lloyd iterations of a sharded fit, enqueued back to back: streaming pass over this rank's shard -> pack ->
// all-reduce of the 2K int64 sums / counts -> finalize (identical on every rank); the look-in rides on the last launch.
extern "C" int nnc_kmeans_iterate_sharded(void *comm, const float *x, void *ws, const nnc_kmeans_params *pp, int32_t iters,
                                          void *host_mapped, uint64_t ticket, void *stream)
{
    int rc = km_check(ws, pp, "nnc_kmeans_iterate_sharded");
    if (rc) return rc;
    if (!comm) return fail(NNC_EINVAL, "nnc_kmeans_iterate_sharded: null communicator");
    const nnc_kmeans_params p = *pp;
    if (p.n > 0 && !x) return fail(NNC_EINVAL, "nnc_kmeans_iterate_sharded: null x");
    if (iters < 1 || (host_mapped && (reinterpret_cast<uintptr_t>(host_mapped) & 7) != 0))
        return fail(NNC_EINVAL, "nnc_kmeans_iterate_sharded: iters < 1 or unaligned host pointer");
    if ((rc = km_set_lds_attr())) return rc;
    KmWs *w = reinterpret_cast<KmWs *>(ws);
    NncComm *c = reinterpret_cast<NncComm *>(comm);
    for (int i = 0; i < iters; i++) {
        if ((rc = km_launch_accumulate(x, w, &p, stream))) return rc;
        if ((rc = km_launch_finalize(w, &p, FIN_PACK_ONLY, 0, stream))) return rc;
        RCCLCHK(g_rccl.AllReduce(w->partials, w->partials, (size_t)(2 * p.k), ncclInt64, ncclSum, c->comm, S(stream)));
        const bool last = i == iters - 1;
        if ((rc = km_launch_finalize(w, &p, FIN_FROM_PARTIALS, 0, stream, last ? host_mapped : nullptr, ticket))) return rc;
    }
    return NNC_OK;
}

// The ranks' farthest-sample lists (each descending, `per` keys, padded with 0 or -1) merged into the `m` largest,
// descending.  A key's place is the number of keys that sort before it: for every list a binary search, ties between
// lists by list number (equal keys are interchangeable samples).  One workgroup.
__global__ __launch_bounds__(KM_THREADS) void k_merge_keys(const long long *__restrict__ lists, int nlists, int per, long long *__restrict__ out, int m)
{
    const int total = nlists * per;
    for (int i = threadIdx.x; i < m; i += KM_THREADS) out[i] = 0ll;
    __syncthreads();
    for (int t = threadIdx.x; t < total; t += KM_THREADS) {
        const int li = t / per, pos = t % per;
        const long long key = lists[t];
        if (key <= 0) continue; // padding
        int place = pos; // the keys before it in its own list
        for (int l = 0; l < nlists; l++) {
            if (l == li) continue;
            const long long *a = lists + (size_t)l * per;
            // number of keys in list l that sort before `key`: strictly greater, or equal and l < li
            int lo = 0, hi = per;
            while (lo < hi) { const int mid = (lo + hi) >> 1; const long long v = a[mid]; if (v > key || (v == key && l < li)) lo = mid + 1; else hi = mid; }
            place += lo;
        }
        if (place < m) out[place] = key;
    }
}

extern "C" int nnc_merge_keys(const int64_t *lists_dev, int32_t nlists, int32_t per, int64_t *out_dev, int32_t m, void *stream)
{
    if (!lists_dev || !out_dev || nlists < 1 || per < 1 || m < 1) return fail(NNC_EINVAL, "nnc_merge_keys: bad argument");
    hipLaunchKernelGGL(k_merge_keys, dim3(1), dim3(KM_THREADS), 0, S(stream), reinterpret_cast<const long long *>(lists_dev), (int)nlists, (int)per,
                       reinterpret_cast<long long *>(out_dev), (int)m);
    LAUNCHCHK("k_merge_keys");
    return NNC_OK;
}

extern "C" size_t nnc_kmeans_reloc_scratch_bytes_sharded(int32_t k, int32_t window, int32_t world)
{
    if (world < 1) return 0;
    const size_t base = nnc_kmeans_reloc_scratch_bytes(k, window);
    return base ? base + reloc_align(8 * (size_t)(NNC_KMAX + 8)) * (size_t)(world + 2) : 0;
}

// nnc_kmeans_relocate_windowed for a sharded vector, collectives included: every rank selects and proves from the windows of
// its own shard -> all-reduce (MAX) of the verdict word -> all-gather of the keys -> merge -> the same edits on every rank
// (or none, if any proof failed) -> resumed finalize.
extern "C" int nnc_kmeans_relocate_windowed_sharded(void *comm, const float *x_sorted, void *ws, const nnc_kmeans_params *p, int32_t n_empty,
                                                    void *scratch_dev, size_t scratch_bytes, void *stream)
{
    int rc = km_check(ws, p, "nnc_kmeans_relocate_windowed_sharded");
    if (rc) return rc;
    if (!comm) return fail(NNC_EINVAL, "nnc_kmeans_relocate_windowed_sharded: null communicator");
    NncComm *c = reinterpret_cast<NncComm *>(comm);
    const int32_t window = nnc_kmeans_reloc_window(p->n, n_empty); // (the caller made sure that every rank's shard allows it)
    if (window == 0 || n_empty > NNC_KMAX) return fail(NNC_EINVAL, "nnc_kmeans_relocate_windowed_sharded: not applicable");
    if (!scratch_dev || (reinterpret_cast<uintptr_t>(scratch_dev) & 255) != 0) return fail(NNC_EINVAL, "nnc_kmeans_relocate_windowed_sharded: null or unaligned (256 B) scratch");
    if (scratch_bytes < nnc_kmeans_reloc_scratch_bytes_sharded(p->k, window, c->world)) return fail(NNC_ENOSPACE, "nnc_kmeans_relocate_windowed_sharded: scratch too small");
    const size_t base = nnc_kmeans_reloc_scratch_bytes(p->k, window), slot = reloc_align(8 * (size_t)(NNC_KMAX + 8));
    unsigned char *b = reinterpret_cast<unsigned char *>(scratch_dev) + base;
    long long *mine = reinterpret_cast<long long *>(b);
    long long *merged = reinterpret_cast<long long *>(b + slot);
    long long *all = reinterpret_cast<long long *>(b + 2 * slot);
    KmWs *w = reinterpret_cast<KmWs *>(ws);
    const int per = n_empty + 1;
    if ((rc = nnc_kmeans_reloc_select_local(x_sorted, ws, p, n_empty, scratch_dev, base, reinterpret_cast<int64_t *>(mine), stream))) return rc;
    RCCLCHK(g_rccl.AllReduce(&w->reloc_fail, &w->reloc_fail, 1, ncclInt32, ncclMax, c->comm, S(stream)));
    RCCLCHK(g_rccl.AllGather(mine, all, (size_t)per, ncclInt64, c->comm, S(stream)));
    if ((rc = nnc_merge_keys(reinterpret_cast<const int64_t *>(all), c->world, per, reinterpret_cast<int64_t *>(merged), per, stream))) return rc;
    if ((rc = nnc_kmeans_relocate_if_proven(ws, reinterpret_cast<const int64_t *>(merged), per, stream))) return rc;
    return km_launch_finalize(w, p, FIN_FROM_PARTIALS, 1, stream);
}

// The Lloyd loop of a SHARDED fit as one call (what nnc_kmeans_fit is to the single GPU): batches of iterations with the all-reduce
// inside (nnc_kmeans_iterate_sharded), the look-ins, batch sizing from the decay of the centre shift, and the windowed relocation of
// empty clusters with its collectives (nnc_kmeans_relocate_windowed_sharded) -- no host language between two launches.  Every rank
// makes the same call; the state machine is replicated (the status is the same on every rank after each finalize), so every rank
// takes the same decisions.  n_min = the shortest shard: whether the windowed relocation applies must come out alike everywhere
// (the window itself depends on n_empty only).  Comes back when the fit has stopped or needs what only the caller can do
// (full-pass relocation, strict-convergence check: status.paused != 0).
// No relocation chain "in case" behind the iterations here: it would put two more collectives behind every one of them.
extern "C" int nnc_kmeans_fit_sharded(void *comm, const float *x_iter, void *ws, const nnc_kmeans_params *pp, int64_t n_min, int32_t max_batch,
                                      int32_t sorted, void *reloc_scratch_dev, size_t reloc_scratch_bytes, void *host_mapped,
                                      uint64_t *ticket_io, nnc_kmeans_status *status_out, int32_t *n_windowed_out, void *stream)
{
    int rc = km_check(ws, pp, "nnc_kmeans_fit_sharded");
    if (rc) return rc;
    if (!comm) return fail(NNC_EINVAL, "nnc_kmeans_fit_sharded: null communicator");
    if (!host_mapped || !ticket_io || !status_out || (reinterpret_cast<uintptr_t>(host_mapped) & 7) != 0) return fail(NNC_EINVAL, "nnc_kmeans_fit_sharded: null / unaligned pointer");
    const nnc_kmeans_params p = *pp;
    if (n_min < 0 || n_min > p.n) return fail(NNC_EINVAL, "nnc_kmeans_fit_sharded: n_min is the shortest shard (0 <= n_min <= n)");
    if (max_batch < 1) max_batch = 1;
    const int world = reinterpret_cast<NncComm *>(comm)->world;
    const size_t slot = sizeof(nnc_kmeans_status) + 8;
    unsigned char *hb = reinterpret_cast<unsigned char *>(host_mapped); // two slots, used alternately
    int batch = 1; // the first iteration is where duplicate initial centres surface as empty clusters
    int nwin = 0;
    double s_prev = -1.0, s_last = -1.0;
    int i_prev = 0, i_last = 0;
    for (;;) {
        const uint64_t ticket = ++(*ticket_io);
        unsigned char *sl = hb + (ticket & 1) * slot;
        if ((rc = nnc_kmeans_iterate_sharded(comm, x_iter, ws, &p, batch, sl, ticket, stream))) return rc;
        if ((rc = km_wait_ticket(reinterpret_cast<volatile unsigned long long *>(sl + sizeof(nnc_kmeans_status)), ticket, S(stream)))) return rc;
        const nnc_kmeans_status st = *reinterpret_cast<const nnc_kmeans_status *>(sl);
        *status_out = st;
        if (st.done) break;
        if (st.paused) {
            const bool strict_check = st.iter >= 1 && st.same_counts;
            const int32_t window = (sorted && st.paused == 1 && !strict_check) ? nnc_kmeans_reloc_window(n_min, st.n_empty) : 0;
            if (window == 0 || !reloc_scratch_dev || reloc_scratch_bytes < nnc_kmeans_reloc_scratch_bytes_sharded(p.k, window, world)) break; // the caller's turn
            if ((rc = nnc_kmeans_relocate_windowed_sharded(comm, x_iter, ws, &p, st.n_empty, reloc_scratch_dev, reloc_scratch_bytes, stream))) return rc;
            nwin++; // (an unproven selection comes back as paused == 2 with the next look-in, and the caller takes this one back)
            batch = 1;
            s_prev = s_last = -1.0;
            continue;
        }
        // size the next batch so that it ends about where the shift crosses the tolerance (as nnc_kmeans_fit does)
        s_prev = s_last; i_prev = i_last;
        s_last = (double)st.shift_tot; i_last = st.iter;
        batch = std::min(max_batch, batch * 2);
        if (s_prev > 0.0 && s_last > 0.0 && s_prev > s_last && p.tol > 0.0f) {
            const double rate = std::log(s_prev / s_last) / std::max(1, i_last - i_prev);
            const double left = s_last > (double)p.tol ? std::log(s_last / (double)p.tol) / rate : 0.0;
            batch = (int)std::max(1.0, std::min((double)max_batch, std::floor(left * 0.9)));
        }
    }
    if (n_windowed_out) *n_windowed_out = nwin;
    return NNC_OK;
}


