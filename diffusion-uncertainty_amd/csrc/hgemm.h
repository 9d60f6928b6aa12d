// Parameter block of the f16x3 one-tap GEMM kernel (gemm_f16x3.hip: k_hgemm, k_hgemm_reduce).
#pragma once
#include "common.h"

struct HGemmP {
    const _Float16 *A1, *A2;     // activation planes [hi | lo][M][K1] / [M][K2] (the lo plane a*_plane elements behind the hi plane); A2 optional (K concat)
    long a1_plane, a2_plane;
    const _Float16* B;           // dua_pack_weight_f16x3: [N / 128 row tiles][K / 32 steps][hi | lo][128][32], then the per-row 2^-s
    long b_bytes;                // bytes of the packed planes (hi and lo)
    const float *Cscale, *bias, *res;
    float* C;                    // EPI 0: fp32 [M][N] = act(acc * Cscale * out_scale + bias) + res
    _Float16* Cplanes;           // EPI 1: f16x3 planes [hi | lo][M][N] of act(acc * Cscale * out_scale + bias) * split_scale
    _Float16* Cextra;            // EPI 0, optional: additionally the f16x3 planes of the fp32 result * split_scale (operand of a later presplit GEMM)
    int extra_from, extra_only;  // ... for the columns >= extra_from (a multiple of 32) only; extra_only: those columns are NOT written to C
    float* colsum;               // optional [ceil(M / 64)][2][N] column sums / sums of squares per 64-row block
    int M, N, K1, K2, ntm, ntn;
    int act_out;                 // 2 = GELU
    float out_scale, split_scale;
    // Work list of the persistent grid: items [0, n_full) are whole tiles (tile id = tm * ntn + tn).  The K-step units of the remaining
    // tail_tiles tiles are cut into tail_p contiguous ranges (stream-K), range c = [c U / tail_p, (c + 1) U / tail_p) owned by workgroup c:
    // a partial first tile (result to kpart[2 c]), whole tiles (real epilogue), a partial last tile (kpart[2 c + 1]), all in accumulator
    // layout; k_hgemm_reduce folds the parts of a tile in workgroup order.
    int n_full, tail_tiles, tail_p;
    float* kpart;
};

int dua_launch_hgemm(HGemmP& p, hipStream_t st, const char* what);
void dua_hgemm_set_workspace(void* ws, size_t bytes);
int dua_hgemm_enabled();

// Parameter block of the raw-operand f16x3 GEMM kernel (gemm_skip_f16x3.hip: k_sgemm): fp32 A (two sources concatenated along K), split on the fly.
struct SGemmP {
    const float *A1, *A2;        // [M][lda1] / [M][lda2] fp32; A2 optional
    long lda1, lda2;             // floats, multiples of 4
    const _Float16* B;           // dua_pack_weight_f16x3
    long b_bytes;
    const float *Cscale, *bias;
    float* C;                    // fp32 [M][N] = acc * Cscale * out_scale + bias
    int M, N, K1, K2, ntm, ntn;
    float a_scale, out_scale;    // power of two applied to A before the split; alpha / a_scale
    unsigned long long* range_ctr;   // f16x3 range check counters (NULL: off)
};
int dua_launch_sgemm(SGemmP& p, hipStream_t st, const char* what);
int dua_sgemm_enabled();
