// Shared helpers for the gfx950 hot-path library (see include/dua_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/dua_hip.h"

extern "C" void dua_set_error(const char* fmt, ...);

#define DUA_REQUIRE(cond, ...)            \
    do {                                  \
        if (!(cond)) {                    \
            dua_set_error(__VA_ARGS__);   \
            return DUA_EINVAL;            \
        }                                 \
    } while (0)

static inline int dua_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        dua_set_error("%s: %s", what, hipGetErrorString(e));
        return DUA_EHIP;
    }
    return DUA_OK;
}

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// Memory-bound grids: cap at 256 CUs x 8 blocks and grid-stride the rest.
static inline unsigned stream_grid(size_t work_items, unsigned block) {
    size_t g = (work_items + block - 1) / block;
    if (g > 2048) g = 2048;
    if (g < 1) g = 1;
    return (unsigned)g;
}

__device__ __forceinline__ float silu_f(float v) { return v / (1.0f + __expf(-v)); }
