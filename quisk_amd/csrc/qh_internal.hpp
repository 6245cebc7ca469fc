// qh_internal.hpp -- declarations shared by the translation units of libquiskhip.so.
#pragma once
#include <string>

namespace qh {

extern thread_local std::string g_last_error;

// Records the message for qh_last_error() and returns `code`.
int set_error(int code, const char *fmt, ...);

}  // namespace qh
