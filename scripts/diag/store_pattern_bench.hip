// Micro-benchmark: how fast can 256 persistent workgroups of 4 waves (one per SIMD, like k_wconv4) write / read the output tile of a
// Winograd F(4,3) conv - 128 V positions x 4 output rows x 128 channels of an [pixels][C] fp32 tensor - depending on how a wave's 16-byte
// lanes are laid over the pixel rows?  Patterns (bytes of one pixel row covered by ONE store instruction of a wave):
//   0: 16 B per lane, adjacent lanes 1 KB apart (accumulator layout)      1: 64 B  (4 adjacent lanes)
//   2: 128 B (8 adjacent lanes)      3: 256 B (16 adjacent lanes = the wave's 64 channels of a row)
// build: hipcc --offload-arch=gfx950 -O3 -o scripts/diag/store_pattern_bench scripts/diag/store_pattern_bench.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

template <int PAT, bool LOADS>
__global__ __launch_bounds__(256, 1) void k_pat(float* C, const float* R, int Mv, int W, int N, int ntn, int sleeps) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave >> 1, wn = wave & 1;
    const int ntm = Mv / 128, ntiles = ntm * ntn;
    float4 accv = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int tn = t % ntn, tm = t / ntn;
        const int pos0 = tm * 128 + wm * 64, c0 = tn * 128 + wn * 64;
        for (int i = 0; i < sleeps; ++i) __builtin_amdgcn_s_sleep(127);
        // 64 instructions: each covers 1 KB of the wave's 64 positions x 4 rows x 64 channels
        constexpr int LPR = PAT == 0 ? 1 : (PAT == 1 ? 4 : (PAT == 2 ? 8 : 16));      // lanes per pixel row
        constexpr int RPI = 64 / LPR;                                                   // pixel rows per instruction
#pragma unroll 4
        for (int k = 0; k < 64; ++k) {
            // enumerate (r, position group, channel group) so that every (pos, r, 16-byte chunk) is covered once
            const int chunks_per_row = 16;                                              // 64 channels = 16 chunks of 16 B
            const int cg_n = chunks_per_row / LPR;                                      // channel groups
            const int r = k & 3, kk = k >> 2;                                           // 16 (pos group, channel group) pairs
            const int cg = kk % cg_n, pg = kk / cg_n;                                   // pg < 64 / RPI ... (16 / cg_n)
            const int pos = pos0 + pg * RPI + (PAT == 0 ? (lane & 15) + 16 * 0 : lane / LPR);
            int chunk = cg * LPR + (PAT == 0 ? (lane >> 4) : lane % LPR);
            int p2 = pos;
            if (PAT == 0) { p2 = pos0 + (kk & 3) * 16 + (lane & 15); chunk = (kk >> 2) * 4 + (lane >> 4); }
            const long row = (long)p2 + 3L * (p2 & ~(W - 1)) + (long)r * W;
            float* o = C + row * N + c0 + 4 * chunk;
            float4 v = accv;
            if (LOADS) { const float4 q = *(const float4*)(R + row * N + c0 + 4 * chunk); v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w; }
            *(float4*)o = v;
        }
        accv.x += 1.0f;
    }
}

static int g_grid = 256;
template <int PAT, bool LOADS>
static float run(float* C, const float* R, int Mv, int W, int N, int sleeps, int reps) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k_pat<PAT, LOADS>), dim3(g_grid), dim3(256), 0, 0, C, R, Mv, W, N, N / 128, sleeps);
    hipEventRecord(a);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k_pat<PAT, LOADS>), dim3(g_grid), dim3(256), 0, 0, C, R, Mv, W, N, N / 128, sleeps);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms / reps;
}

int main(int argc, char** argv) {
    const int B = 128, H = 128, W = 128, N = argc > 1 ? atoi(argv[1]) : 256;
    const int Mv = B * (H / 4) * W;
    const size_t bytes = (size_t)B * H * W * N * 4;
    float *C, *R;
    hipMalloc(&C, bytes); hipMalloc(&R, bytes);
    hipMemset(R, 0, bytes);
    printf("output %.2f GB (+ residual the same), 256 workgroups x 4 waves, %d tiles per workgroup\n", bytes / 1e9, (Mv / 128) * (N / 128) / 256);
    // fewer workgroups, same tiles per workgroup: the per-CU rate without the other CUs' traffic
    for (int g = 256; g >= 8; g /= 4) {
        g_grid = g;
        const int mv = Mv / (256 / g);
        float ms = run<3, false>(C, R, mv, W, N, 0, 3), ml = run<3, true>(C, R, mv, W, N, 0, 3), m0 = run<0, true>(C, R, mv, W, N, 0, 3);
        const double kb = 256.0 * 32;                                         // KB written per workgroup
        printf("%3d workgroups: store only %6.1f GB/s per CU | load+store %6.1f GB/s per CU (256-byte rows), %6.1f (16-byte lanes)\n", g, kb * 1024 / ms / 1e6, 2 * kb * 1024 / ml / 1e6,
               2 * kb * 1024 / m0 / 1e6);
    }
    g_grid = 256;
    for (int sleeps = 0; sleeps <= 20; sleeps += 20) {
        printf("-- %d x s_sleep(127) of pseudo-compute per tile\n", sleeps);
        float base = sleeps ? run<0, false>(C, R, 0, W, N, sleeps, 1) : 0.f; (void)base;
#define RUN(P, L) { float ms = run<P, L>(C, R, Mv, W, N, sleeps, 3); printf("pattern %d %s: %7.3f ms  -> %6.2f TB/s\n", P, L ? "load+store" : "store only", ms, bytes * (L ? 2 : 1) / ms / 1e9); }
        RUN(0, false) RUN(1, false) RUN(2, false) RUN(3, false)
        RUN(0, true) RUN(1, true) RUN(2, true) RUN(3, true)
    }
    return 0;
}
