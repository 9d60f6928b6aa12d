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

// ---- bench facility: per-entry-point launch timing (dua_kernel_profile, csrc/blocks.hip).  An entry point that wants to appear in
// bench.py's `hbm_kernels` / per-leg roofline brackets its launches with a DuaKTimer: HIP events on the LAUNCH stream, tagged with a
// kind and the launch's ALGORITHMIC bytes / FLOPs.  Off (the default): one load and one branch per entry point.
extern "C" void* dua_kprof_begin_(void* stream);
extern "C" void dua_kprof_end_(void* h, void* stream, const char* kind, double bytes, double flops);
struct DuaKTimer {
    void* h;
    void* st;
    explicit DuaKTimer(void* stream) : h(dua_kprof_begin_(stream)), st(stream) {}
    void done(const char* kind, double bytes, double flops = 0.0) { if (h) dua_kprof_end_(h, st, kind, bytes, flops); }
};

// Memory-bound grids: cap at 256 CUs x 8 blocks and grid-stride the rest.
static inline unsigned stream_grid(size_t work_items, unsigned block) {
    size_t g = (work_items + block - 1) / block;
    if (g > 2048) g = 2048;
    if (g < 1) g = 1;
    return (unsigned)g;
}

__device__ __forceinline__ float silu_f(float v) { return v / (1.0f + __expf(-v)); }

// A lane id formed where it is used (volatile: neither hoisted nor shared) and a lane ^ mask exchange built on it.  __shfl_xor's own
// __lane_id() is hoisted to the start of the kernel, lives across the K loop, is spilled by the register-full kernels and comes back by
// a scratch load behind s_waitcnt vmcnt(0) - which, retiring in order, drains every output store in flight: in k_wconv4 once per tile,
// right after its 64 stores, in front of the column-sum fold (round 5).
__device__ __forceinline__ int dua_lane_fresh() {
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}
__device__ __forceinline__ float dua_xor_fresh(float v, int lane, int mask) {
    return __int_as_float(__builtin_amdgcn_ds_bpermute(4 * (lane ^ mask), __float_as_int(v)));
}

// ---- f16x3 range check (debug facility, off by default): with dua_range_check(1) every kernel that WRITES f16x3 operands (the GroupNorm /
// LayerNorm / GELU / attention plane writers, the Winograd V writer, the on-the-fly split of raw operands) counts the operands whose scaled
// magnitude leaves fp16's range (|s| > 65504: hi saturates and the pair carries fewer than 22 bits - counter 0) and those below fp16's
// smallest normal (0 < |s| < 2^-14: hi is subnormal; the pair is still exact to 2^-25 / scale absolutely - counter 1).  Each translation
// unit holds its own copy of the device pointer (static: no relocatable device code needed); dua_range_check sets them all.
// __constant__: a scalar load the compiler may hoist and share.  As a plain __device__ variable the pointer was RE-LOADED in front of every
// group of operands that followed a store (it might alias): 64 - 128 vector loads per tile in the epilogues of k_hgemm and in the plane writers
// of k_flash_attn_x - each behind an s_waitcnt vmcnt that, retiring in order, drained every output store in flight (U-ViT fc1 -9 %).
static __constant__ unsigned long long* dua_range_dev = nullptr;
// (kernels load the pointer ONCE - dua_range_ctr() - and test it per group of operands: one uniform branch when the check is off)
__device__ __forceinline__ unsigned long long* dua_range_ctr() {
#ifdef DUA_NO_RANGE_CHECK                                             // (A/B build: prices the check in the plane writers)
    return nullptr;
#else
    return dua_range_dev;
#endif
}
__device__ __forceinline__ void dua_range_note(unsigned long long* c, float s) {            // s = operand * plane scale, BEFORE the clamp
    const float a = fabsf(s);
    if (a > 65504.f) atomicAdd(c, 1ull);
    else if (a > 0.f && a < 6.103515625e-05f) atomicAdd(c + 1, 1ull);
}
#define DUA_RANGE_TU(NAME)                                                                                      \
    int dua_range_set_##NAME(unsigned long long* p) { return hipMemcpyToSymbol(HIP_SYMBOL(dua_range_dev), &p, sizeof(p)) == hipSuccess ? 0 : 1; }

// Four consecutive columns of one row as f16x3 planes: plane 0 fp16(x * scale) (saturating), plane 1 the fp16 remainder.
__device__ __forceinline__ void dua_store_f16x3_quad(void* planes, long plane_elems, long o, float4 w, float scale) {
    typedef _Float16 f16x4_t __attribute__((ext_vector_type(4)));
    const float f[4] = {w.x * scale, w.y * scale, w.z * scale, w.w * scale};
    f16x4_t h, l;
    if (unsigned long long* rc_ = dua_range_ctr()) {
#pragma unroll
        for (int j = 0; j < 4; ++j) dua_range_note(rc_, f[j]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float sv = fminf(fmaxf(f[j], -65504.f), 65504.f);
        const _Float16 t = (_Float16)sv;
        h[j] = t;
        l[j] = (_Float16)(sv - (float)t);
    }
    *(f16x4_t*)((_Float16*)planes + o) = h;
    *(f16x4_t*)((_Float16*)planes + plane_elems + o) = l;
}

