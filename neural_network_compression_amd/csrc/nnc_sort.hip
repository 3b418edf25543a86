// nnc_sort.hip -- one-time reordering of a weight vector by value (ascending) for the Lloyd
// iterations.  Per-cluster sums do not depend on the order of the weights (exact integer
// sums), so the iterations may stream a sorted copy: neighbouring weights then fall in the
// same cluster and each lane accumulates runs in registers instead of issuing one LDS atomic
// per weight.  The sort itself is a plain library call (rocPRIM device radix sort), outside
// the per-iteration path; labels / quantized values are produced from the ORIGINAL vector.
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

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
