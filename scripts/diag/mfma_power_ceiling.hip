// Diagnostic (not part of the library): the matrix pipe's SUSTAINED rate for the product mix of the f16x3 kernels under the part's power cap.
// bench.py's roofline divides by the datasheet peak (2.5 PFLOP/s dense fp16 = 2.4 GHz x 256 CUs x 4 SIMDs x 1024 FLOP per clock); the headline
// job runs at 1 359 W of a 1 400 W cap with in-kernel clocks of 1.46 - 1.66 GHz (profiles/r06_power_trace_headline_job.txt, r06d_pmc_*), so no
// kernel can reach that peak.  This program measures what BARE v_mfma_f32_16x16x32_f16 instructions reach - operands in registers, no LDS, no
// memory, >= 2 s per variant - for
//   hi/lo   the f16x3 mix: per (a, b) pair three products a_hi b_lo + a_hi b_hi + a_lo b_hi, a_lo / b_lo at 2^-11 of a_hi / b_hi (random mantissas)
//   hi      one product per pair on full-magnitude random operands (the f16w mix)
//   zero    all-zero operands (no toggling: the data-independent part of the power)
// each with one wave per SIMD (k_wconv4s occupancy) and two (k_hgemm / k_wconv), 256 / 128 accumulators per lane (all in AGPRs: no copies, no scratch).
// build: hipcc --offload-arch=gfx950 -O3 -o scripts/diag/mfma_power_ceiling scripts/diag/mfma_power_ceiling.hip ; run: ./mfma_power_ceiling
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// (inline assembly with explicit operand classes, as the product kernels: the compiler's own AGPR handling inserts ~1.5 register copies per MFMA)
template <int WPS>
__device__ __forceinline__ void mma(f32x4& c, const f16x8& a, const f16x8& b) {
    if constexpr (WPS == 1) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
    else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}

// NB x NB accumulator tiles of 16 x 16 per wave (NB = 4 x 6 groups at one wave per SIMD = 384 accumulators, as k_wconv4's 6 frequencies)
template <int G, int THREE, int WPS>
__global__ __launch_bounds__(256 * WPS, 1) void k_bare(const int* __restrict__ src, float* __restrict__ out, int iters) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int* s = src + (size_t)(t % 4096) * 64;
    f16x8 ah[4], al[4], bh[4], bl[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        ah[i] = *(const f16x8*)(s + 4 * i);
        al[i] = *(const f16x8*)(s + 16 + 4 * i);
        bh[i] = *(const f16x8*)(s + 32 + 4 * i);
        bl[i] = *(const f16x8*)(s + 48 + 4 * i);
    }
    f32x4 acc[G][4][4];
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[g][i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < G; ++g) {                       // one "K step" per group: 16 tiles x (3 | 1) products, every accumulator touched 16 MFMAs apart
            if (THREE) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) mma<WPS>(acc[g][i][j], ah[i], bl[j]);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) mma<WPS>(acc[g][i][j], ah[i], bh[j]);
            if (THREE) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) mma<WPS>(acc[g][i][j], al[i], bh[j]);
            }
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");          // (wait states between the last matrix-core writes and the reads below)
    float v = 0.f;
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) v += acc[g][i][j][0] + acc[g][i][j][1] + acc[g][i][j][2] + acc[g][i][j][3];
    out[t] = v;
}

template <int G, int THREE, int WPS>
static double run(const int* src, float* out, int iters, double seconds) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 256, block = 256 * WPS;
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k_bare<G, THREE, WPS>), grid, block, 0, 0, src, out, iters);
    hipDeviceSynchronize();
    // size the measurement: one launch timed, then enough launches for `seconds`
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k_bare<G, THREE, WPS>), grid, block, 0, 0, src, out, iters);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms1 = 0; hipEventElapsedTime(&ms1, e0, e1);
    int launches = (int)(seconds * 1e3 / ms1) + 1;
    if (launches > 4000) launches = 4000;
    hipEventRecord(e0, 0);
    for (int i = 0; i < launches; ++i) hipLaunchKernelGGL((k_bare<G, THREE, WPS>), grid, block, 0, 0, src, out, iters);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    // executed matrix-core FLOPs: per wave and iteration G groups x 16 tiles x (3 | 1) MFMAs x 2 * 16 * 16 * 32
    const double mfma = (double)G * 16 * (THREE ? 3 : 1) * (double)iters * (4.0 * WPS) * grid * launches;
    return mfma * 2.0 * 16 * 16 * 32 / (ms * 1e-3) / 1e12;
}

static void fill(std::vector<int>& h, int mode) {
    srand(1);
    for (size_t i = 0; i < h.size(); ++i) {
        const size_t w = i % 64;
        const bool lo = (w % 32) >= 16;                     // words 16..31 / 48..63 of a lane's 64: the lo operands
        float s = mode == 2 ? 0.f : (lo ? 1.0f / 2048.0f : 1.0f);
        _Float16 a = (_Float16)((rand() % 2001 - 1000) * 1e-3f * s), b = (_Float16)((rand() % 2001 - 1000) * 1e-3f * s);
        unsigned short ua, ub; __builtin_memcpy(&ua, &a, 2); __builtin_memcpy(&ub, &b, 2);
        h[i] = (int)(ua | ((unsigned)ub << 16));
    }
}

int main() {
    const size_t n = 4096 * 64;
    std::vector<int> h(n);
    int* src; float* out;
    hipMalloc(&src, n * 4); hipMalloc(&out, 256 * 512 * 4);
    const int iters = 2000;
    const char* names[3] = {"hi/lo f16x3 mix (3 products per pair)", "hi only (1 product per pair)", "all-zero operands (3 products per pair)"};
    for (int rep = 0; rep < 2; ++rep)
        for (int mode = 0; mode < 3; ++mode) {
            fill(h, mode);
            hipMemcpy(src, h.data(), n * 4, hipMemcpyHostToDevice);
            double a, b;
            if (mode == 1) { a = run<4, 0, 1>(src, out, iters, 2.0); b = run<2, 0, 2>(src, out, iters, 2.0); }
            else { a = run<4, 1, 1>(src, out, iters, 2.0); b = run<2, 1, 2>(src, out, iters, 2.0); }
            printf("rep %d  %-42s  1 wave/SIMD (256 acc): %7.1f TF executed = %.3f of 2500   2 waves/SIMD (128 acc): %7.1f TF executed = %.3f\n",
                   rep, names[mode], a, a / 2500.0, b, b / 2500.0);
            fflush(stdout);
        }
    return 0;
}
