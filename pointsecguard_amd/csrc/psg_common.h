// Internal helpers shared by the libpsg translation units (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

#include "../../include/psg.h"

namespace psg {

void set_error(const char *fmt, ...);

#define PSG_CHECK_HIP(expr)                                                                      \
    do {                                                                                         \
        hipError_t _e = (expr);                                                                  \
        if (_e != hipSuccess) {                                                                  \
            psg::set_error("%s:%d: %s failed: %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
            return PSG_ERR_HIP;                                                                  \
        }                                                                                        \
    } while (0)

#define PSG_REQUIRE(cond, ...)                                                                   \
    do {                                                                                         \
        if (!(cond)) {                                                                           \
            psg::set_error(__VA_ARGS__);                                                         \
            return PSG_ERR_ARG;                                                                  \
        }                                                                                        \
    } while (0)

#define PSG_LAUNCH_CHECK() PSG_CHECK_HIP(hipGetLastError())

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline int round_up(int a, int b) { return ceil_div(a, b) * b; }

}  // namespace psg

struct psg_ctx {
    int device;
    int num_cu;
};
