// common.hpp - error reporting and small host helpers shared by the translation units of
// libdcvc_amd.so.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <string>

#include "dcvc_amd.h"

namespace dcvc {

enum { OK = 0, E_ARG = -1, E_HIP = -2, E_MEM = -3, E_STREAM = -4 };

void set_error(const char* fmt, ...);

inline size_t elem_size(int dtype) { return dtype == DCVC_F16 ? 2 : 4; }
inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

}  // namespace dcvc

#define DCVC_HIP(expr)                                                                      \
    do {                                                                                    \
        hipError_t _e = (expr);                                                             \
        if (_e != hipSuccess) {                                                             \
            dcvc::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, \
                            __LINE__);                                                      \
            return dcvc::E_HIP;                                                             \
        }                                                                                   \
    } while (0)

#define DCVC_REQUIRE(cond, ...)           \
    do {                                  \
        if (!(cond)) {                    \
            dcvc::set_error(__VA_ARGS__); \
            return dcvc::E_ARG;           \
        }                                 \
    } while (0)

#define DCVC_LAUNCH_CHECK()                                                               \
    do {                                                                                  \
        hipError_t _e = hipGetLastError();                                                \
        if (_e != hipSuccess) {                                                           \
            dcvc::set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(_e),    \
                            __FILE__, __LINE__);                                          \
            return dcvc::E_HIP;                                                           \
        }                                                                                 \
    } while (0)
