// Parameter block shared by the LDS-DMA conv / GEMM kernels (mfma_bf16x3.hip: k_xconv_dma; conv_f16x3.hip: k_hconv).
#pragma once
#include <hip/hip_runtime.h>

struct DConvP {
    const __bf16* A;         // hi plane; lo plane follows at +a_plane elements
    const __bf16* B;         // pre-split weights, hi plane then lo plane (+b_plane elements)
    const float *bias, *res;
    float* C;
    long a_plane, b_plane;
    int M, N, K, H, W, Cin, nchw_out, ntm, ntn;
    float* colsum;           // optional [ceil(M/64)][2][N] column sums / sums of squares per 64-row block
    int dbg_norec;           // diagnostic: zero-record descriptors (every DMA returns zeros at once) - prices the memory path
    int ntaps;               // 9 = 3x3 convolution, 1 = plain GEMM over rows (1x1 convolution / linear layer)
    int act_out;             // 2 = GELU on the output (U-ViT fc1)
    __bf16* Csplit;          // optional: write the output as split bf16 planes (hi | lo, [M][N] each) instead of fp32 C
    int stagger_phases;      // > 1: workgroup start-up stagger (see k_xconv_dma); stagger_sleeps = s_sleep(127) per phase step
    int stagger_sleeps;
    const float* Cscale;     // FMT 3 (f16x3): per weight row 2^-s (the planes hold w * 2^s), padded to the tile grid
    float out_scale;         // FMT 3: 1 / activation plane scale; the epilogue multiplies the accumulator by Cscale[col] * out_scale
    float split_scale;       // FMT 3 with Csplit: the output planes hold fp16 hi | lo of C * split_scale
    // Split-K (k_xconv_dma, per-tap form of the bf16x3 / f16x3 loops only): ksplit > 1 makes every (tile, part) pair a work item of
    // its own; part s runs K steps [s*nk/ksplit, (s+1)*nk/ksplit) and stores its SCALED accumulators (no bias / residual) to
    // kpart[s][M][N]; k_splitk_reduce then folds the parts in index order (deterministic) and applies the epilogue.
    int ksplit;
    float* kpart;
    int tail_pass;           // host side only: this launch is one half of a head | tail split of the rows (launch_dconv)
};

// conv_f16x3.hip: the f16x3 kernel on 16x16x32 MFMA shapes (row-shift form where the image width divides 256)
int dua_launch_hconv(DConvP& d, hipStream_t st, const char* what);
