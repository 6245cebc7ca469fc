// qh_internal.hpp -- declarations shared by the translation units of libquiskhip.so.
#pragma once
#include <cstdlib>
#include <hip/hip_runtime.h>
#include <string>
#include "../../include/quiskhip.h"

namespace qh {

extern thread_local std::string g_last_error;

// Records the message for qh_last_error() and returns `code`.
int set_error(int code, const char *fmt, ...);

#define QH_HIP(expr)                                                                                  \
    do {                                                                                              \
        hipError_t _e = (expr);                                                                       \
        if (_e != hipSuccess)                                                                         \
            return qh::set_error(QH_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

// QH_POISON_ALLOC=1 (diagnostics): every allocation starts as NaNs / huge negative integers instead of what the allocator hands out,
// so that a buffer read before it is written shows in the results of any run, not only where the pool returns used memory
template <typename T> static inline hipError_t dev_alloc(T **p, size_t n)
{
    const hipError_t e = hipMalloc(reinterpret_cast<void **>(p), n * sizeof(T));
    static const bool poison = [] { const char *v = std::getenv("QH_POISON_ALLOC"); return v && v[0] == '1'; }();
    if (e == hipSuccess && poison && n) { (void)hipMemset(*p, 0xFF, n * sizeof(T)); (void)hipDeviceSynchronize(); }
    return e;
}

// Zeros for a buffer that kernels on a NON-BLOCKING stream will use next: hipMemset runs on the null stream, which such streams do not wait
// for, and the API lets it return before the fill has run (it does not on this ROCm, but nothing promises that)
static inline hipError_t dev_zero(void *p, size_t bytes)
{
    hipError_t e = hipMemset(p, 0, bytes);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    return e;
}

}  // namespace qh
