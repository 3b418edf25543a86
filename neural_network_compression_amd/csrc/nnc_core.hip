// nnc_core.hip -- error text, version, device info, per-device caches and the optional in-library profiler of libnnc_hip.so.
#include "nnc_common.hpp"

// --------------------------------------------------------------------------------------
// error plumbing
// --------------------------------------------------------------------------------------
static thread_local std::string g_err;

int nnc_set_error_(int code, const char *msg) { g_err = msg ? msg : ""; return code; } // (every translation unit reports through this)

extern "C" int nnc_version(void) { return NNC_VERSION; }
extern "C" const char *nnc_last_error(void) { return g_err.c_str(); }

// Per-device caches (a process may drive several GPUs, from several host threads): indexed by the current device, the
// entries only ever go from "unknown" to the one value every thread would compute, so plain atomics suffice.
int nnc_current_device_(void)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= NNC_MAX_DEVICES) dev = 0;
    return dev;
}
static std::atomic<int> g_cu_count[NNC_MAX_DEVICES];
int nnc_cu_count_(void)
{
    const int dev = current_device();
    int c = g_cu_count[dev].load(std::memory_order_relaxed);
    if (c == 0) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) == hipSuccess) c = prop.multiProcessorCount;
        if (c <= 0) c = 256;
        g_cu_count[dev].store(c, std::memory_order_relaxed);
    }
    return c;
}

extern "C" int nnc_device_info(char *arch_out, size_t arch_len, int *cu_count_out)
{
    int dev = 0;
    hipDeviceProp_t prop;
    HIPCHK(hipGetDevice(&dev));
    HIPCHK(hipGetDeviceProperties(&prop, dev));
    if (arch_out && arch_len) {
        std::snprintf(arch_out, arch_len, "%s", prop.gcnArchName);
    }
    if (cu_count_out) *cu_count_out = prop.multiProcessorCount;
    return NNC_OK;
}


// --------------------------------------------------------------------------------------
// optional in-library profiler: HIP events around the launches of the data-touching kernels, on the
// stream each kernel is launched on (hipExtLaunchKernelGGL stamps the events at the kernel's own begin
// and end, not at the command processor's arrival, so the difference is the launch's execution time).
// bench.py turns it on to report per-kernel durations over the timed region.
// --------------------------------------------------------------------------------------
struct ProfRec { hipEvent_t a, b; int tag; };
static std::mutex g_prof_mu;            // the pool is shared by every calling thread
static std::vector<ProfRec> g_prof_pool;
static size_t g_prof_used = 0;
static std::atomic<bool> g_prof_on{false};
static std::atomic<uint32_t> g_prof_mask{0xFFFFFFFFu}; // which NNC_PROF_* tags get events (an event pair costs its launch a little)
static int64_t g_prof_skipped = 0;

extern "C" int nnc_profile_tags(uint32_t mask) { g_prof_mask = mask; return NNC_OK; }

void nnc_prof_take_(int tag, hipEvent_t *a, hipEvent_t *b)
{
    *a = nullptr; *b = nullptr;
    if (!g_prof_on.load(std::memory_order_relaxed) || !((g_prof_mask.load(std::memory_order_relaxed) >> tag) & 1u)) return;
    std::lock_guard<std::mutex> lock(g_prof_mu);
    if (g_prof_on && g_prof_used < g_prof_pool.size()) { ProfRec &r = g_prof_pool[g_prof_used++]; r.tag = tag; *a = r.a; *b = r.b; }
    else g_prof_skipped++;
}
extern "C" int nnc_profile_begin(int32_t max_launches)
{
    if (max_launches < 1) return fail(NNC_EINVAL, "nnc_profile_begin: max_launches < 1");
    std::lock_guard<std::mutex> lock(g_prof_mu);
    while ((int64_t)g_prof_pool.size() < max_launches) {
        ProfRec pr;
        pr.tag = -1;
        HIPCHK(hipEventCreate(&pr.a));
        HIPCHK(hipEventCreate(&pr.b));
        g_prof_pool.push_back(pr);
    }
    g_prof_used = 0;
    g_prof_skipped = 0;
    g_prof_on = true;
    return NNC_OK;
}

// Waits for the recorded events and copies the duration (ms) and the NNC_PROF_* tag of each timed launch, in launch
// order, into ms_out / tags_out[0..cap); count_out = number of launches timed.
extern "C" int nnc_profile_end(float *ms_out, int32_t *tags_out, int64_t cap, int64_t *count_out)
{
    g_prof_on = false;
    std::lock_guard<std::mutex> lock(g_prof_mu);
    int64_t cnt = 0;
    for (size_t i = 0; i < g_prof_used; i++) {
        HIPCHK(hipEventSynchronize(g_prof_pool[i].b));
        float ms = 0.0f;
        HIPCHK(hipEventElapsedTime(&ms, g_prof_pool[i].a, g_prof_pool[i].b));
        if (ms_out && cnt < cap) ms_out[cnt] = ms;
        if (tags_out && cnt < cap) tags_out[cnt] = g_prof_pool[i].tag;
        cnt++;
    }
    if (count_out) *count_out = cnt;
    g_prof_used = 0;
    return NNC_OK;
}


// --------------------------------------------------------------------------------------
