// Diagnostic (not part of the library): wall-clock rate of the f16f6 product mix of k_xconv_dma's K loop on the two MFMA shape
// families, operands in registers, 2 waves per SIMD, random data, >= 2 s per variant (the chip is power-limited on these loops:
// MI355X_MICROARCH.md, DVFS give-back item 7).  Per 32-deep K step and 64 x 128 wave tile:
//   shape 32: 16 x v_mfma_f32_32x32x16_f16 + 8 x v_mfma_scale_f32_32x32x64_f8f6f4 (fp6 e2m3)
//   shape 16: 32 x v_mfma_f32_16x16x32_f16 + 16 x v_mfma_scale_f32_16x16x128_f8f6f4 per TWO steps (K = 128 = two steps' cross terms)
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_shape_bench scripts/diag/mfma_shape_bench.hip ; run: ./mfma_shape_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

template <int SHAPE>
__global__ __launch_bounds__(512, 1) void k_mix(const int* __restrict__ src, float* __restrict__ out, int iters) {
    const int t = blockIdx.x * 512 + threadIdx.x;
    const int* s = src + (size_t)(t % 4096) * 64;
    f16x8 ah[4], bh[8];
    i32x8 ax[4], bx[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) { ah[i] = *(const f16x8*)(s + 4 * i); ax[i] = *(const i32x8*)(s + 16 + 8 * (i & 1)); ax[i][6] = 100 + (i & 3); }
#pragma unroll
    for (int i = 0; i < 8; ++i) { bh[i] = *(const f16x8*)(s + 32 + 4 * (i & 3)); bx[i] = *(const i32x8*)(s + 48 + 8 * (i & 1)); bx[i][6] = 101 + (i & 3); }
    if (SHAPE == 32) {
        f32x16 acc[2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int st = 0; st < 2; ++st) {            // two K steps per iteration
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[2 * h + i], bh[4 * h + j], acc[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(ax[2 * st + i], bx[4 * st + j], acc[i][j], 2, 2, 0, ax[2 * st + i][6], 0, bx[4 * st + j][6]);
            }
        }
        float v = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) v += acc[i][j][r];
        out[t] = v;
    } else {
        f32x4 acc[4][8];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int st = 0; st < 2; ++st)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[(i + st) & 3], bh[(j + st) & 7], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(ax[i], bx[j], acc[i][j], 2, 2, 0, ax[i][6], 0, bx[j][6]);
        }
        float v = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) v += acc[i][j][r];
        out[t] = v;
    }
}

template <int SHAPE>
static double run(const int* src, float* out, int iters, int launches) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k_mix<SHAPE>, 256, 512, 0, 0, src, out, iters);
    hipEventRecord(e0, 0);
    for (int i = 0; i < launches; ++i) hipLaunchKernelGGL(k_mix<SHAPE>, 256, 512, 0, 0, src, out, iters);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    // algorithmic products per iteration and wave: two K steps of a 64 x 128 x 32 tile
    const double flop = 2.0 * 64 * 128 * 32 * 2 * (double)iters * 8 * 256 * launches;
    return flop / (ms * 1e-3) / 1e12;
}

int main() {
    const size_t n = 4096 * 64;
    std::vector<int> h(n);
    srand(1);
    for (size_t i = 0; i < n; ++i) {
        const size_t w = i % 64;
        if ((w % 32) < 16) {          // f16 pairs, |v| <~ 1
            _Float16 a = (_Float16)((rand() % 2001 - 1000) * 1e-3f), b = (_Float16)((rand() % 2001 - 1000) * 1e-3f);
            unsigned short ua, ub; __builtin_memcpy(&ua, &a, 2); __builtin_memcpy(&ub, &b, 2);
            h[i] = (int)(ua | ((unsigned)ub << 16));
        } else h[i] = (int)(((unsigned)rand() << 16) ^ (unsigned)rand());      // fp6 codes: random bits
    }
    int* src; float* out;
    hipMalloc(&src, n * 4); hipMalloc(&out, 256 * 512 * 4);
    hipMemcpy(src, h.data(), n * 4, hipMemcpyHostToDevice);
    const int iters = 4000;
    for (int rep = 0; rep < 3; ++rep) {
        const double t32 = run<32>(src, out, iters, 150), t16 = run<16>(src, out, iters, 150);
        printf("rep %d: 32x32 shapes %.1f TF algorithmic, 16x16 shapes %.1f TF algorithmic, ratio %.3f\n", rep, t32, t16, t16 / t32);
    }
    return 0;
}
