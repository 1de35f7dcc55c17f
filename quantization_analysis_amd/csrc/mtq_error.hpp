// mtq_error.hpp — thread-local error reporting shared by the translation units of libmtq_hip.so.
#pragma once
#include "../../include/mtq.h"

namespace mtq {
// Records `msg` as the calling thread's last error and returns `code`.
int fail(int code, const char *msg);
int failf(int code, const char *fmt, ...);
// MTQ_OK when a HIP device is usable by this process, MTQ_ERR_NO_DEVICE otherwise (cached).
int require_device();
// hipGetLastError() → MTQ_OK / MTQ_ERR_HIP with the kernel name in the message.
int check_launch(const char *what);
} // namespace mtq
