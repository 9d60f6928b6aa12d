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

// Four consecutive columns of one row as f16f8 planes (the operand format of the presplit LDS-DMA GEMM, see dua_hip.h): plane 0
// fp16(x) at element offset o; plane 1 (starting plane_elems 2-byte elements later) per 32-column block
// [32 x e4m3((x - fp16 x) * 2^12) | 32 x e4m3(x)].  The row length must be a multiple of 32 and o a multiple of 4.
__device__ __forceinline__ void dua_store_f16f8_quad(void* planes, long plane_elems, long o, float4 w) {
    typedef _Float16 f16x4_t __attribute__((ext_vector_type(4)));
    const float f[4] = {w.x, w.y, w.z, w.w};
    float l[4], g[4];
    f16x4_t h;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const _Float16 t = (_Float16)f[j];
        h[j] = t;
        l[j] = fminf(fmaxf((f[j] - (float)t) * 4096.0f, -448.f), 448.f);
        g[j] = fminf(fmaxf(f[j], -448.f), 448.f);
    }
    *(f16x4_t*)((_Float16*)planes + o) = h;
    unsigned char* x8 = (unsigned char*)((_Float16*)planes + plane_elems) + (o >> 5) * 64 + (o & 31);
    int lo8 = __builtin_amdgcn_cvt_pk_fp8_f32(l[0], l[1], 0, false), hi8 = __builtin_amdgcn_cvt_pk_fp8_f32(g[0], g[1], 0, false);
    lo8 = __builtin_amdgcn_cvt_pk_fp8_f32(l[2], l[3], lo8, true);
    hi8 = __builtin_amdgcn_cvt_pk_fp8_f32(g[2], g[3], hi8, true);
    *(int*)x8 = lo8;
    *(int*)(x8 + 32) = hi8;
}


// ---- f16x3 range check (debug facility, off by default): with dua_range_check(1) every kernel that WRITES f16x3 operands (the GroupNorm /
// LayerNorm / GELU / attention plane writers, the Winograd V writer, the on-the-fly split of raw operands) counts the operands whose scaled
// magnitude leaves fp16's range (|s| > 65504: hi saturates and the pair carries fewer than 22 bits - counter 0) and those below fp16's
// smallest normal (0 < |s| < 2^-14: hi is subnormal; the pair is still exact to 2^-25 / scale absolutely - counter 1).  Each translation
// unit holds its own copy of the device pointer (static: no relocatable device code needed); dua_range_check sets them all.
static __device__ unsigned long long* dua_range_dev = nullptr;
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
    void dua_range_set_##NAME(unsigned long long* p) { (void)hipMemcpyToSymbol(HIP_SYMBOL(dua_range_dev), &p, sizeof(p)); }

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

// e2m3 (MXFP6) code of v * 2^(127 - sbyte): sign<<5 | eemmm, round to nearest even; magnitudes < 1 use the subnormal grid (step 1/8).
// sbyte is the E8M0 block exponent (biased exponent of the block maximum minus two: the maximum lands in [4, 8); codes above 7.5 clamp).
__device__ __forceinline__ unsigned dua_e2m3_code(float v, int sbyte) {
    const float u = v * __uint_as_float((unsigned)(254 - sbyte) << 23);
    const float a = fabsf(u);
    const float t = a < 1.0f ? a + 1.0f : a;
    const unsigned b = __float_as_uint(t);
    const unsigned rb = (b + 0x7FFFFu + ((b >> 20) & 1u)) >> 20;                // RNE to 3 mantissa bits: exponent | mantissa
    unsigned c = rb - 1016u + (a < 1.0f ? 0u : 8u);
    c = c > 31u ? 31u : c;
    return c | ((__float_as_uint(u) >> 31) << 5);
}
