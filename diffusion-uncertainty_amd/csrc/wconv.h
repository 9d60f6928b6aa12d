// Shared by the Winograd F(2,3)-along-y conv kernels (conv_wino_f16x3.hip: fp16 hi | lo operands; conv_wino_f32.hip: exact fp32
// operands): the parameter block and the epilogue (inverse transform across partner waves, scale / bias / residual, row stores,
// column sums).  Both kernels use the same LDS map in bytes: weight ring 3 x 32 KB at 0, activation ring 2 x 32 KB behind it.
#pragma once
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

// The store loops of wconv_epilogue and the tile-start wait of the kernels share these bounds: the wait leaves exactly the epilogue's
// always-issued output stores in flight (vmcnt retires in order), so the count is DERIVED from the loops, not a second literal.
constexpr int WCONV_ST_PAIRS = 2;    // channel-block pairs (32 channels each) of a wave
constexpr int WCONV_ST_BLOCKS = 4;   // 16-position blocks
constexpr int WCONV_ST_HALVES = 2;   // 8-row halves of a block (one 128-byte row store instruction each)
constexpr int WCONV_EPILOGUE_STORES = WCONV_ST_PAIRS * WCONV_ST_BLOCKS * WCONV_ST_HALVES;
static_assert(WCONV_EPILOGUE_STORES <= 63, "vmcnt is a 6-bit counter");

struct WConvP {
    const void* V;           // [4 f][ntm tiles of 128 positions][Cin / 32] stage blocks of 16 KB: [hi | lo][128][32] halves or [128][32] floats
    const void* B;           // packed weights [ntn][K / 32 steps] stage blocks of 32 KB ([hi | lo][256][32] halves or [256][32] floats), then the row scales
    const float *bias, *res, *Cscale;
    float* C;
    float* colsum;
    long b_plane;            // ntn * 256 * K
    int Mv, N, Cin, K, H, W, ntm, ntn;
    float out_scale;
    int res_mode;            // 0: res is [N][H][W][Cout]; +1: res is [N][H/2][W/2][Cout], read through nearest-neighbour 2x upsampling;
                             // -1: res is [N][2H][2W][Cout], read through 2x2 average pooling (the resampled identity skip of an ADM
                             // up / down block, unet_openai.py:238-243, without materialising it)
};


// Inverse transform across the partner waves (wave ^ 4: same positions and channels, other frequencies): the f = 1, 2 wave sends
// P = M1 + M2 and finishes the ODD output row (M1 - M2) - M3, the f = 0, 3 wave sends M3 and finishes the EVEN row M0 + P.  Two rounds
// of 8 KB per wave through activation stage 1 / weight stage 2 (the slots the next tile's prefetch leaves free), then channel-block
// pairs (32 channels) of one 16-position block through a 16 x 144-byte patch (in the partner's slot, which only this wave still
// reads), so that residual reads and output stores move whole 128-byte rows.  acc0 / acc1: [channel block][position block] of
// fi = 0 / 1, lane = channels 4 (lane >> 4) .. + 3 of position lane & 15.
__device__ __forceinline__ void wconv_epilogue(const WConvP& p, f32x4 (&acc0)[4][4], f32x4 (&acc1)[4][4], char* ldsw, int wave, int lane,
                                               int m0, int n0) {
    constexpr int A1_BYTES = 3 * 32768 + 32768, B2_BYTES = 2 * 32768;       // activation stage 1, weight stage 2
    const int wm = wave >> 1, wn = wave & 1, grp_w = wave >> 2, ph = wm & 1;
    const int lane_e = dua_lane_fresh();         // formed here: the kernel's own lane id, kept across the K loop, was spilled and came back behind a vmcnt(0)
    auto slot_of = [&](int w) -> char* { return ldsw + (w < 4 ? A1_BYTES + w * 8192 : B2_BYTES + (w - 4) * 8192); };
    char* mine = slot_of(wave);
    char* theirs = slot_of(wave ^ 4);
    const bool odd = grp_w == 0;
    // the epilogue's first global operands (column scales, bias, the first residual rows of both channel-block pairs) are requested
    // BEFORE the exchange: its two barriers would otherwise stand between the tile's last MFMA and the first of three dependent
    // load latencies per pair
    const int pr_e = lane_e >> 3, pc_e = lane_e & 7;
    const int pos0 = m0 + ph * 64;
    const long oddoff = odd ? p.W : 0;
    auto opix = [&](int i, int h) -> long {
        const int pos = pos0 + 16 * i + 8 * h + pr_e;
        return pos < p.Mv ? (long)pos + (pos & ~(p.W - 1)) + oddoff : -1L;
    };
    float4 sq_[2], bq_[2], rn_[2][4][2];
    auto res_ld = [&](int jp, int i, float4 (&dst)[2]) {
        const int c = n0 + wn * 64 + 32 * jp + 4 * pc_e;
        const bool cok = c < p.N;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const long row = opix(i, h);
            dst[h] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (!(p.res && cok && row >= 0)) continue;
            if (p.res_mode == 0) {
                dst[h] = *(const float4*)(p.res + row * p.N + c);
                continue;
            }
            const unsigned t = (unsigned)row >> (31 - __clz(p.W)), xo = (unsigned)row & (p.W - 1), n = t / (unsigned)p.H, yo = t - n * p.H;
            if (p.res_mode > 0) {
                const long src = ((long)n * (p.H >> 1) + (yo >> 1)) * (p.W >> 1) + (xo >> 1);
                dst[h] = *(const float4*)(p.res + src * p.N + c);
            } else {                                                  // same order of additions as k_resample2x (denoiser.hip)
                const long W2 = 2 * p.W, src = ((long)n * 2 * p.H + 2 * yo) * W2 + 2 * xo;
                const float4 a = *(const float4*)(p.res + src * p.N + c), b = *(const float4*)(p.res + (src + 1) * p.N + c);
                const float4 d = *(const float4*)(p.res + (src + W2) * p.N + c), e = *(const float4*)(p.res + (src + W2 + 1) * p.N + c);
                dst[h] = make_float4((a.x + b.x + d.x + e.x) * 0.25f, (a.y + b.y + d.y + e.y) * 0.25f, (a.z + b.z + d.z + e.z) * 0.25f,
                                     (a.w + b.w + d.w + e.w) * 0.25f);
            }
        }
    };
#pragma unroll
    for (int jp = 0; jp < 2; ++jp) {
        const int c = n0 + wn * 64 + 32 * jp + 4 * pc_e;
        sq_[jp] = *(const float4*)(p.Cscale + c);                        // (padded to the tile grid)
        bq_[jp] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.bias && c < p.N) bq_[jp] = *(const float4*)(p.bias + c);
        if (jp == 0) {                                                   // (the second pair's rows after the exchange, into the registers it frees)
#pragma unroll
            for (int i = 0; i < 4; ++i) res_ld(0, i, rn_[0][i]);
        }
    }
    // The output stores are buffer stores that ALWAYS issue (lanes outside the tensor get an out-of-range offset, which the buffer unit
    // drops): the kernels count on WCONV_EPILOGUE_STORES vector-memory operations behind the next tile's prologue DMA, so that their
    // tile-start wait can leave that many operations - these stores - in flight instead of draining the queue (vmcnt retires in order)
    const long cpix0 = 2L * m0;                                          // the tile's 128 positions are output pixels 2 m0 .. 2 m0 + 255
    const long crem = (2L * p.Mv - cpix0) * p.N * 4;
    const auto rc = __builtin_amdgcn_make_buffer_rsrc((void*)(p.C + cpix0 * p.N), 0, (int)(crem < 0x7FFFFFF0L ? crem : 0x7FFFFFF0L), 0x00020000);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int jj = 2 * h + (k >> 2), i = k & 3;
            const f32x4 v = odd ? acc0[jj][i] + acc1[jj][i] : acc1[jj][i];
            *(f32x4*)(mine + k * 1024 + lane_e * 16) = v;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int jj = 2 * h + (k >> 2), i = k & 3;
            const f32x4 r = *(const f32x4*)(theirs + k * 1024 + lane_e * 16);
            acc0[jj][i] = odd ? (acc0[jj][i] - acc1[jj][i]) - r : acc0[jj][i] + r;
        }
        if (h == 0) __syncthreads();
    }
    // ---- stores: channel-block pairs (32 channels) of one 16-position block through a 16 x 144-byte patch (in the partner's
    // slot, which only this wave still reads), so that residual reads and output stores move whole 128-byte rows
    const int lr_e = lane_e & 15, lc_e = lane_e >> 4;
    char* pw = theirs;
#pragma unroll
    for (int i = 0; i < 4; ++i) res_ld(1, i, rn_[1][i]);
#pragma unroll
    for (int jp = 0; jp < WCONV_ST_PAIRS; ++jp) {
        const int c = n0 + wn * 64 + 32 * jp + 4 * pc_e;
        const bool cok = c < p.N;
        float4 sq = sq_[jp];
        sq.x *= p.out_scale; sq.y *= p.out_scale; sq.z *= p.out_scale; sq.w *= p.out_scale;
        const float4 bq = bq_[jp];
        float4 cs = make_float4(0.f, 0.f, 0.f, 0.f), cq2 = cs;
#pragma unroll
        for (int i = 0; i < WCONV_ST_BLOCKS; ++i) {
            const float4 (&rq)[2] = rn_[jp][i];
            *(f32x4*)(pw + lr_e * 144 + 16 * lc_e) = acc0[2 * jp][i];
            *(f32x4*)(pw + lr_e * 144 + 64 + 16 * lc_e) = acc0[2 * jp + 1][i];
#pragma unroll
            for (int h = 0; h < WCONV_ST_HALVES; ++h) {
                const f32x4 a = *(const f32x4*)(pw + (8 * h + pr_e) * 144 + 16 * pc_e);
                const long row = opix(i, h);
                float4 v;
                v.x = a[0] * sq.x + bq.x + rq[h].x;
                v.y = a[1] * sq.y + bq.y + rq[h].y;
                v.z = a[2] * sq.z + bq.z + rq[h].z;
                v.w = a[3] * sq.w + bq.w + rq[h].w;
                const bool ok = cok && row >= 0;
                typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
                const u32x4_ vb = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
                __builtin_amdgcn_raw_buffer_store_b128(vb, rc, ok ? (unsigned)(((row - cpix0) * p.N + c) * 4) : 0xFFFFFFF0u, 0, 0);
                if (ok) {
                    cs.x += v.x; cs.y += v.y; cs.z += v.z; cs.w += v.w;
                    cq2.x += v.x * v.x; cq2.y += v.y * v.y; cq2.z += v.z * v.z; cq2.w += v.w * v.w;
                }
            }
        }
        if (p.colsum) {
            // The wave's 64 positions are 64 output pixels of ONE image (H*W % 128 == 0), every other row of a block of 128
            // consecutive output pixels = two 64-pixel slots of the column-sum array; this wave fills the slot of its parity.
            // The consumer (dua_groupnorm_stats_from_colsums) folds all slots of an image, so only the per-image totals matter.
            float t[8] = {cs.x, cs.y, cs.z, cs.w, cq2.x, cq2.y, cq2.z, cq2.w};
            const int lf_ = dua_lane_fresh();            // (not __shfl_xor: its hoisted lane id is spilled and reloaded behind a vmcnt(0), common.h)
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                t[u] += dua_xor_fresh(t[u], lf_, 8);
                t[u] += dua_xor_fresh(t[u], lf_, 16);
                t[u] += dua_xor_fresh(t[u], lf_, 32);
            }
            if (pr_e == 0 && cok && pos0 < p.Mv) {
                float* o = p.colsum + ((long)(pos0 >> 5) + (odd ? 1 : 0)) * 2 * p.N + c;
                *(float4*)o = make_float4(t[0], t[1], t[2], t[3]);
                *(float4*)(o + p.N) = make_float4(t[4], t[5], t[6], t[7]);
            }
        }
    }
}
