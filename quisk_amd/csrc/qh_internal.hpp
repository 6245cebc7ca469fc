// qh_internal.hpp -- declarations shared by the translation units of libquiskhip.so.
#pragma once
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

template <typename T> static inline hipError_t dev_alloc(T **p, size_t n)
{
    return hipMalloc(reinterpret_cast<void **>(p), n * sizeof(T));
}

}  // namespace qh
