// Block-level entry points of the C ABI: one ADM ResBlock / AttentionBlock forward per call (SURVEY.md 8(b):
// adm_resblock_fwd, adm_attnblock_fwd).  Pure host-side sequencing of the per-op entry points of this library on the caller's
// stream over a caller-provided workspace: the same kernels, in the same order, with the same operands as the Python
// composition in guided_diffusion/unet_openai.py (ResBlock.run / AttentionBlock.run), so the results are bit-identical to it.
// Precisions: DUA_PREC_FP32 (exact fp32 MFMA) and DUA_PREC_F16X3 (the fp32-equivalent split) - the two parity modes.
#include "common.h"
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

namespace {

// Optional per-launch timing of the conv3x3 launches made inside the block functions (bench.py's roofline leg): HIP events on
// the launch stream, collected and summed by dua_conv_profile_collect.  Off by default; not thread-safe (a bench facility).
struct ConvRec { hipEvent_t a, b; double flops, mfma; };
bool g_prof = false;
std::vector<ConvRec> g_recs;
struct ConvTimer {
    hipEvent_t a = nullptr;
    hipStream_t st;
    explicit ConvTimer(void* stream) : st(as_stream(stream)) {
        if (g_prof && hipEventCreate(&a) == hipSuccess) (void)hipEventRecord(a, st); else a = nullptr;
    }
    // flops: algorithmic (2*M*9*Cin*Cout); mfma_mult: matrix-core product FLOPs executed per algorithmic FLOP (direct f16x3: 3 fp16
    // products; Winograd form: 6 of 9 taps x 3 = 2; exact fp32 MFMA: 1)
    void done(double flops, double mfma_mult) {
        if (!a) return;
        hipEvent_t b;
        if (hipEventCreate(&b) != hipSuccess) { (void)hipEventDestroy(a); return; }
        (void)hipEventRecord(b, st);
        g_recs.push_back({a, b, flops, flops * mfma_mult});
    }
};

struct Carver {                       // bump allocator over the caller's workspace, 256-byte aligned pieces
    char* base;
    size_t used, cap;
    bool dry;                         // dry run: only add up the sizes
    void* take(size_t bytes) {
        size_t at = (used + 255) & ~(size_t)255;
        used = at + bytes;
        return (dry || used > cap) ? nullptr : base + at;
    }
};

inline size_t colsum_floats(long M, int cout) { return (size_t)((M + 63) / 64) * 2 * cout; }

// GroupNorm statistics of [x1 | x2]: from the producers' column sums when both sources carry them (and H*W % 64 == 0), else
// by the stand-alone kernel.  `gnws` must hold dua_groupnorm_ws_floats(...) floats in the second case.
int gn_stats(const float* x1, int C1, const float* cs1, const float* x2, int C2, const float* cs2, int N, int HW, int G, float* gnws,
             float* stats, void* stream) {
    if (cs1 && (C2 == 0 || cs2) && HW % 64 == 0)
        return dua_groupnorm_stats_from_colsums(cs1, C1, C2 ? cs2 : nullptr, C2, N, HW, G, 1e-5f, stats, stream);
    return dua_groupnorm_stats(x1, C1, C2 ? x2 : nullptr, C2, N, HW, G, 1e-5f, gnws, stats, stream);
}

int resblock_run(const dua_resblock_args* a, Carver& w, void* stream) {
    const int C = a->C1 + a->C2, G = 32;
    const int Ho = a->updown < 0 ? a->H / 2 : (a->updown > 0 ? 2 * a->H : a->H), Wo = a->updown < 0 ? a->W / 2 : (a->updown > 0 ? 2 * a->W : a->W);
    const long M = (long)a->N * Ho * Wo;
    const bool single = a->precision == DUA_PREC_F16W;                 // one fp16 product per tap in the Winograd-form convs
    const bool f16 = a->precision == DUA_PREC_F16X3 || single, colsum2 = a->Cout % 4 == 0;
    float* stats1 = (float*)w.take((size_t)a->N * G * 2 * 4);
    float* gnws = (float*)w.take(dua_groupnorm_ws_floats(a->N, a->H * a->W > Ho * Wo ? a->H * a->W : Ho * Wo, C > a->Cout ? C : a->Cout, G) * 4);
    // two activation-sized regions, reused like the per-op composition's freed tensors (a smaller live footprint keeps more of
    // the block in the Infinity Cache): region A = conv1's input, later conv2's input; region B = conv1's output, later the skip
    // convolution's output (conv1's output is dead once the second GroupNorm has consumed it)
    // Winograd form of a 3x3 conv (csrc/conv_wino_f16x3.hip) where its packed weights are given and the shape allows: the GroupNorm
    // in front writes the four transformed images (8 bytes per element instead of 4)
    const bool four1 = a->conv1_wino_m == 4, four2 = a->conv2_wino_m == 4;          // F(4,3) packing (csrc/conv_wino4_f16x3.hip / conv_wino4_f32.hip): the f16x3 three-product mode and fp32
    const int wsup1 = (a->conv1_wino && a->C1 % 32 == 0) ? (four1 ? dua_conv3x3_f16x3_wino4_supported : dua_conv3x3_f16x3_wino_supported)(a->N, Ho, Wo, C, a->Cout) : 0;
    const int wsup2 = a->conv2_wino ? (four2 ? dua_conv3x3_f16x3_wino4_supported : dua_conv3x3_f16x3_wino_supported)(a->N, Ho, Wo, a->Cout, a->Cout) : 0;
    const bool wino1 = wsup1 != 0, wino2 = wsup2 != 0;          // (== 2: no column sums from that conv; the statistics kernel runs instead)
    const size_t szA1 = wino1 ? (four1 ? dua_f16x3_wino4_act_bytes : dua_f16x3_wino_act_bytes)(a->N, Ho, Wo, C) : (size_t)M * C * 4;             // (whole tiles of 128 positions)
    const size_t szA2 = wino2 ? (four2 ? dua_f16x3_wino4_act_bytes : dua_f16x3_wino_act_bytes)(a->N, Ho, Wo, a->Cout) : (size_t)M * a->Cout * 4;
    void* regA = w.take(szA1 > szA2 ? szA1 : szA2);
    // the resampled identity skip x_upd (:243): the Winograd conv2 reads x through the resampling in its epilogue; the other forms take a copy
    const bool fuse_xr = a->updown && wino2;
    float* xr = (a->updown && !fuse_xr) ? (float*)w.take((size_t)M * a->C1 * 4) : nullptr;
    float* regB = (float*)w.take((size_t)M * a->Cout * 4);
    float* cs2 = (colsum2 && wsup1 != 2) ? (float*)w.take(colsum_floats(M, a->Cout) * 4) : nullptr;
    float* stats2 = (float*)w.take((size_t)a->N * G * 2 * 4);
    void* h1 = regA;                                                 // fp32 [N,Ho,Wo,C] or two 2-byte planes of it
    float* h2 = regB;
    void* h3 = regA;
    float* res = a->skip_packed ? regB : nullptr;
    if (w.dry) return DUA_OK;
    DUA_REQUIRE(w.used <= w.cap && a->workspace, "dua_adm_resblock_fwd: workspace of %zu bytes needed, %zu given", w.used, w.cap);
    int rc;
    // in_layers: GroupNorm32 -> SiLU (-> 2x resampling of h, :238-243) -> conv3x3                       unet_openai.py:183-187
    if ((rc = gn_stats(a->x1, a->C1, a->x1_colsums, a->x2, a->C2, a->x2_colsums, a->N, a->H * a->W, G, gnws, stats1, stream))) return rc;
    if (wino1 && !f16 && four1) rc = dua_groupnorm_apply_wino4_f32(a->x1, a->C1, a->x2, a->C2, a->N, a->H, a->W, G, stats1, a->gn1_w, a->gn1_b, nullptr, 0, 1,
                                                                    a->updown, nullptr, h1, stream);
    else if (wino1 && !f16) rc = dua_groupnorm_apply_wino_f32(a->x1, a->C1, a->x2, a->C2, a->N, a->H, a->W, G, stats1, a->gn1_w, a->gn1_b, nullptr, 0, 1,
                                                          a->updown, nullptr, h1, stream);
    else if (wino1 && four1) rc = dua_groupnorm_apply_wino4_f16x3(a->x1, a->C1, a->x2, a->C2, a->N, a->H, a->W, G, stats1, a->gn1_w, a->gn1_b, nullptr, 0, 1, a->updown,
                                                                  nullptr, 0.125f * a->act_scale, h1, stream);
    else if (wino1) rc = (single ? dua_groupnorm_apply_wino_f16w : dua_groupnorm_apply_wino_f16x3)(a->x1, a->C1, a->x2, a->C2, a->N, a->H, a->W, G, stats1, a->gn1_w,
                                                   a->gn1_b, nullptr, 0, 1, a->updown, nullptr, 0.5f * a->act_scale, h1, stream);
    else if (f16) rc = dua_groupnorm_apply_split_f16x3(a->x1, a->C1, a->x2, a->C2, a->N, a->H, a->W, G, stats1, a->gn1_w, a->gn1_b, nullptr, 0, 1, a->updown,
                                                  nullptr, a->act_scale, h1, stream);
    else rc = dua_groupnorm_apply(a->x1, a->C1, a->x2, a->C2, a->N, a->H, a->W, G, stats1, a->gn1_w, a->gn1_b, nullptr, 0, 1, a->updown, nullptr,
                                  (float*)h1, stream);
    if (rc) return rc;
    if (xr && (rc = dua_resample2x(a->x1, xr, a->N, a->H, a->W, a->C1, a->updown, stream))) return rc;          // x_upd :243
    {
        ConvTimer tm(stream);
        if (wino1 && !f16 && four1) rc = dua_conv3x3_f32_wino4(h1, a->conv1_wino, a->conv1_b, nullptr, 0, h2, a->N, Ho, Wo, C, a->Cout, cs2, stream);
        else if (wino1 && !f16) rc = dua_conv3x3_f32_wino(h1, a->conv1_wino, a->conv1_b, nullptr, 0, h2, a->N, Ho, Wo, C, a->Cout, cs2, stream);
        else if (wino1 && four1) rc = dua_conv3x3_f16x3_wino4(h1, a->conv1_wino, a->conv1_b, nullptr, 0, h2, a->N, Ho, Wo, C, a->Cout, cs2, 0.125f * a->act_scale, stream);
        else if (wino1) rc = (single ? dua_conv3x3_f16w_wino : dua_conv3x3_f16x3_wino)(h1, a->conv1_wino, a->conv1_b, nullptr, 0, h2, a->N, Ho, Wo, C, a->Cout, cs2,
                                                                                       0.5f * a->act_scale, stream);
        else if (f16) rc = dua_conv3x3_f16x3_presplit(h1, a->conv1_packed, a->conv1_b, nullptr, h2, a->N, Ho, Wo, C, a->Cout, 0, cs2, a->act_scale, stream);
        else rc = dua_conv3x3((const float*)h1, (const float*)a->conv1_packed, a->conv1_b, nullptr, h2, a->N, Ho, Wo, C, a->Cout, 0, cs2, stream);
        tm.done(2.0 * (double)M * 9.0 * C * a->Cout, (f16 ? ((single && wino1) ? 1.0 : 3.0) : 1.0) * (wino1 ? (four1 ? 0.5 : 2.0 / 3.0) : 1.0));
    }
    if (rc) return rc;
    // out_layers: GroupNorm32 * (1 + scale) + shift -> SiLU -> dropout -> conv3x3 (+ skip)               :207-214,245-257
    if ((rc = gn_stats(h2, a->Cout, cs2, nullptr, 0, nullptr, a->N, Ho * Wo, G, gnws, stats2, stream))) return rc;
    if (wino2 && !f16 && four2) rc = dua_groupnorm_apply_wino4_f32(h2, a->Cout, nullptr, 0, a->N, Ho, Wo, G, stats2, a->gn2_w, a->gn2_b, a->film, a->film_ld, 1, 0,
                                                                    a->dropmask, h3, stream);
    else if (wino2 && !f16) rc = dua_groupnorm_apply_wino_f32(h2, a->Cout, nullptr, 0, a->N, Ho, Wo, G, stats2, a->gn2_w, a->gn2_b, a->film, a->film_ld, 1, 0,
                                                          a->dropmask, h3, stream);
    else if (wino2 && four2) rc = dua_groupnorm_apply_wino4_f16x3(h2, a->Cout, nullptr, 0, a->N, Ho, Wo, G, stats2, a->gn2_w, a->gn2_b, a->film, a->film_ld, 1, 0, a->dropmask,
                                                                  0.125f * a->act_scale, h3, stream);
    else if (wino2) rc = (single ? dua_groupnorm_apply_wino_f16w : dua_groupnorm_apply_wino_f16x3)(h2, a->Cout, nullptr, 0, a->N, Ho, Wo, G, stats2, a->gn2_w, a->gn2_b,
                                                   a->film, a->film_ld, 1, 0, a->dropmask, 0.5f * a->act_scale, h3, stream);
    else if (f16) rc = dua_groupnorm_apply_split_f16x3(h2, a->Cout, nullptr, 0, a->N, Ho, Wo, G, stats2, a->gn2_w, a->gn2_b, a->film, a->film_ld, 1, 0,
                                                  a->dropmask, a->act_scale, h3, stream);
    else rc = dua_groupnorm_apply(h2, a->Cout, nullptr, 0, a->N, Ho, Wo, G, stats2, a->gn2_w, a->gn2_b, a->film, a->film_ld, 1, 0, a->dropmask,
                                  (float*)h3, stream);
    if (rc) return rc;
    const float* xin = xr ? xr : a->x1;
    const int rmode = fuse_xr ? a->updown : 0;
    const float* resid = xin;                                                                                  // identity skip :223-226
    if (a->skip_packed) {                                                                                      // 1x1 convolution over [x | skip]
        if (f16) {
            rc = dua_gemm_f16x3(xin, a->C1, a->C1, a->C2 ? a->x2 : nullptr, a->C2, a->C2 > 4 ? a->C2 : 4, a->skip_packed, a->skip_b, nullptr, a->Cout,
                                res, a->Cout, (int)M, a->Cout, 0, 0, 1.0f, nullptr, a->raw_scale, stream);
        } else {
            dua_gemm_args g{};
            g.A1 = xin; g.A2 = a->C2 ? a->x2 : nullptr; g.B = (const float*)a->skip_packed; g.bias = a->skip_b; g.C = res;
            g.M = (int)M; g.N = a->Cout; g.K1 = a->C1; g.K2 = a->C2;
            g.lda1 = a->C1; g.lda2 = a->C2 > 4 ? a->C2 : 4; g.ldb = C; g.ldc = a->Cout; g.ldr = a->Cout;
            g.batch = 1; g.batch_inner = 1; g.transB = 1; g.alpha = 1.0f;
            rc = dua_gemm(&g, stream);
        }
        if (rc) return rc;
        resid = res;
    }
    float* ycs = (a->y_colsums && colsum2) ? a->y_colsums : nullptr;
    DUA_REQUIRE(!(ycs && wsup2 == 2), "dua_adm_resblock_fwd: y_colsums with the Winograd conv needs H*W %% 128 == 0 (F(4,3): %% 256)");
    ConvTimer tm(stream);
    if (wino2 && !f16 && four2) rc = dua_conv3x3_f32_wino4(h3, a->conv2_wino, a->conv2_b, resid, rmode, a->y, a->N, Ho, Wo, a->Cout, a->Cout, ycs, stream);
    else if (wino2 && !f16) rc = dua_conv3x3_f32_wino(h3, a->conv2_wino, a->conv2_b, resid, rmode, a->y, a->N, Ho, Wo, a->Cout, a->Cout, ycs, stream);
    else if (wino2 && four2) rc = dua_conv3x3_f16x3_wino4(h3, a->conv2_wino, a->conv2_b, resid, rmode, a->y, a->N, Ho, Wo, a->Cout, a->Cout, ycs, 0.125f * a->act_scale, stream);
    else if (wino2) rc = (single ? dua_conv3x3_f16w_wino : dua_conv3x3_f16x3_wino)(h3, a->conv2_wino, a->conv2_b, resid, rmode, a->y, a->N, Ho, Wo, a->Cout, a->Cout, ycs,
                                                                                   0.5f * a->act_scale, stream);
    else if (f16) rc = dua_conv3x3_f16x3_presplit(h3, a->conv2_packed, a->conv2_b, resid, a->y, a->N, Ho, Wo, a->Cout, a->Cout, 0, ycs, a->act_scale, stream);
    else rc = dua_conv3x3((const float*)h3, (const float*)a->conv2_packed, a->conv2_b, resid, a->y, a->N, Ho, Wo, a->Cout, a->Cout, 0, ycs, stream);
    tm.done(2.0 * (double)M * 9.0 * a->Cout * a->Cout, (f16 ? ((single && wino2) ? 1.0 : 3.0) : 1.0) * (wino2 ? (four2 ? 0.5 : 2.0 / 3.0) : 1.0));
    return rc;
}

int attnblock_run(const dua_attnblock_args* a, Carver& w, void* stream) {
    const int C = a->C, G = 32, T = a->H * a->W, ch = C / a->heads;
    const long M = (long)a->N * T;
    const bool f16 = a->precision == DUA_PREC_F16X3;
    float* stats = (float*)w.take((size_t)a->N * G * 2 * 4);
    float* gnws = (float*)w.take(dua_groupnorm_ws_floats(a->N, T, C, G) * 4);
    void* xn = w.take((size_t)M * C * 4);
    float* qkv = (float*)w.take((size_t)M * 3 * C * 4);
    void* att = w.take((size_t)M * C * 4);
    if (w.dry) return DUA_OK;
    DUA_REQUIRE(w.used <= w.cap && a->workspace, "dua_adm_attnblock_fwd: workspace of %zu bytes needed, %zu given", w.used, w.cap);
    int rc;
    if ((rc = gn_stats(a->x, C, a->x_colsums, nullptr, 0, nullptr, a->N, T, G, gnws, stats, stream))) return rc;                       // norm :300
    const float scale = (float)(1.0 / sqrt((double)ch));                  // the two ch^-1/4 operand scales of :349 / :382
    float* ycs = (a->y_colsums && C % 4 == 0) ? a->y_colsums : nullptr;
    if (f16) {
        if ((rc = dua_groupnorm_apply_split_f16x3(a->x, C, nullptr, 0, a->N, a->H, a->W, G, stats, a->gn_w, a->gn_b, nullptr, 0, 0, 0, nullptr,
                                                  a->act_scale, xn, stream))) return rc;
        if ((rc = dua_gemm_f16x3_presplit(xn, a->qkv_packed, a->qkv_b, nullptr, qkv, (int)M, 3 * C, C, 0, nullptr, a->act_scale, stream))) return rc;   // qkv :301
        if ((rc = dua_attention_f16x3(qkv, nullptr, att, a->raw_scale, a->N, T, a->heads, ch, scale, stream))) return rc;                               // :302
        return dua_gemm_f16x3_presplit(att, a->proj_packed, a->proj_b, a->x, a->y, (int)M, C, C, 0, ycs, a->raw_scale, stream);                         // proj_out + x :303-306
    }
    if ((rc = dua_groupnorm_apply(a->x, C, nullptr, 0, a->N, a->H, a->W, G, stats, a->gn_w, a->gn_b, nullptr, 0, 0, 0, nullptr, (float*)xn, stream))) return rc;
    dua_gemm_args g{};
    g.A1 = (const float*)xn; g.B = (const float*)a->qkv_packed; g.bias = a->qkv_b; g.C = qkv;
    g.M = (int)M; g.N = 3 * C; g.K1 = C; g.lda1 = C; g.lda2 = 4; g.ldb = C; g.ldc = 3 * C; g.ldr = 3 * C;
    g.batch = 1; g.batch_inner = 1; g.transB = 1; g.alpha = 1.0f;
    if ((rc = dua_gemm(&g, stream))) return rc;
    if ((rc = dua_attention(qkv, (float*)att, a->N, T, a->heads, ch, scale, stream))) return rc;
    dua_gemm_args p{};
    p.A1 = (const float*)att; p.B = (const float*)a->proj_packed; p.bias = a->proj_b; p.residual = a->x; p.C = a->y;
    p.M = (int)M; p.N = C; p.K1 = C; p.lda1 = C; p.lda2 = 4; p.ldb = C; p.ldc = C; p.ldr = C;
    p.batch = 1; p.batch_inner = 1; p.transB = 1; p.alpha = 1.0f; p.colsum = ycs;
    return dua_gemm(&p, stream);
}

int uvitblock_run(const dua_uvitblock_args* a, Carver& w, void* stream) {
    const int D = a->D, ch = D / a->heads, Hd = a->hidden;
    const long M = (long)a->B * a->L;
    const bool f16 = a->precision == DUA_PREC_F16X3;
    float* x1 = a->skip_packed ? (float*)w.take((size_t)M * D * 4) : nullptr;          // skip_linear(cat([x, skip]))
    void* hn = w.take((size_t)M * D * 4);                                               // LayerNorm output (fp32 or planes), both norms
    float* qkv = (float*)w.take((size_t)M * 3 * D * 4);
    void* att = w.take((size_t)M * D * 4);
    float* x2 = (float*)w.take((size_t)M * D * 4);                                      // x + proj(attention)
    void* hid = w.take((size_t)M * Hd * 4);                                             // gelu(fc1) (fp32 or planes)
    if (w.dry) return DUA_OK;
    DUA_REQUIRE(w.used <= w.cap && a->workspace, "dua_uvit_block_fwd: workspace of %zu bytes needed, %zu given", w.used, w.cap);
    int rc;
    const float scale = (float)(1.0 / sqrt((double)ch));
    auto gemm32 = [&](const float* A1, int K1, const float* A2, int K2, const void* W, const float* bias, const float* res, float* C, int N,
                      int act_out) {
        dua_gemm_args g{};
        g.A1 = A1; g.A2 = K2 ? A2 : nullptr; g.B = (const float*)W; g.bias = bias; g.residual = res; g.C = C;
        g.M = (int)M; g.N = N; g.K1 = K1; g.K2 = K2; g.lda1 = K1; g.lda2 = K2 > 4 ? K2 : 4; g.ldb = K1 + K2; g.ldc = N; g.ldr = N;
        g.batch = 1; g.batch_inner = 1; g.transB = 1; g.act_out = act_out; g.alpha = 1.0f;
        return dua_gemm(&g, stream);
    };
    const float* x = a->x;
    if (a->skip_packed) {                                                                               // uvit.py:116-117
        if (f16 && a->x_planes && a->skip_planes)                                                       // both operands exist as planes: presplit two-source GEMM
            rc = dua_gemm_f16x3_presplit2(a->x_planes, D, a->skip_planes, D, a->skip_packed, a->skip_b, nullptr, x1, nullptr, 0, 0, (int)M, D, 0, nullptr,
                                          a->raw_scale, a->raw_scale, stream);
        else if (f16) rc = dua_gemm_f16x3(a->x, D, D, a->skip, D, D, a->skip_packed, a->skip_b, nullptr, D, x1, D, (int)M, D, 0, 0, 1.0f, nullptr,
                                          a->raw_scale, stream);
        else rc = gemm32(a->x, D, a->skip, D, a->skip_packed, a->skip_b, nullptr, x1, D, 0);
        if (rc) return rc;
        x = x1;
    }
    if (f16) {
        if ((rc = dua_layernorm_split_f16x3(x, a->norm1_w, a->norm1_b, hn, M, D, a->eps, a->act_scale, stream))) return rc;        // :118 norm1
        if ((rc = dua_gemm_f16x3_presplit(hn, a->qkv_packed, a->qkv_b, nullptr, qkv, (int)M, 3 * D, D, 0, nullptr, a->act_scale, stream))) return rc;
        if ((rc = dua_attention_f16x3(qkv, nullptr, att, a->raw_scale, a->B, a->L, a->heads, ch, scale, stream))) return rc;        // :69-73
        if ((rc = dua_gemm_f16x3_presplit(att, a->proj_packed, a->proj_b, x, x2, (int)M, D, D, 0, nullptr, a->raw_scale, stream))) return rc;
        if ((rc = dua_layernorm_split_f16x3(x2, a->norm2_w, a->norm2_b, hn, M, D, a->eps, a->act_scale, stream))) return rc;       // :119 norm2
        if ((rc = dua_gemm_f16x3_presplit_planes(hn, a->fc1_packed, a->fc1_b, hid, (int)M, Hd, D, 2, a->act_scale, a->raw_scale, stream))) return rc;
        if (a->y_planes)
            return dua_gemm_f16x3_presplit2(hid, Hd, nullptr, 0, a->fc2_packed, a->fc2_b, x2, a->y, a->y_planes, 0, 0, (int)M, D, 0, nullptr, a->raw_scale,
                                            a->raw_scale, stream);
        return dua_gemm_f16x3_presplit(hid, a->fc2_packed, a->fc2_b, x2, a->y, (int)M, D, Hd, 0, nullptr, a->raw_scale, stream);
    }
    if ((rc = dua_layernorm(x, a->norm1_w, a->norm1_b, (float*)hn, M, D, a->eps, stream))) return rc;
    if ((rc = gemm32((const float*)hn, D, nullptr, 0, a->qkv_packed, a->qkv_b, nullptr, qkv, 3 * D, 0))) return rc;
    if ((rc = dua_attention(qkv, (float*)att, a->B, a->L, a->heads, ch, scale, stream))) return rc;
    if ((rc = gemm32((const float*)att, D, nullptr, 0, a->proj_packed, a->proj_b, x, x2, D, 0))) return rc;
    if ((rc = dua_layernorm(x2, a->norm2_w, a->norm2_b, (float*)hn, M, D, a->eps, stream))) return rc;
    if ((rc = gemm32((const float*)hn, D, nullptr, 0, a->fc1_packed, a->fc1_b, nullptr, (float*)hid, Hd, 2))) return rc;
    {
        dua_gemm_args g{};
        g.A1 = (const float*)hid; g.B = (const float*)a->fc2_packed; g.bias = a->fc2_b; g.residual = x2; g.C = a->y;
        g.M = (int)M; g.N = D; g.K1 = Hd; g.lda1 = Hd; g.lda2 = 4; g.ldb = Hd; g.ldc = D; g.ldr = D;
        g.batch = 1; g.batch_inner = 1; g.transB = 1; g.alpha = 1.0f;
        return dua_gemm(&g, stream);
    }
}

int check_uvitblock(const dua_uvitblock_args* a) {
    DUA_REQUIRE(a && a->x && a->y && a->norm1_w && a->norm1_b && a->norm2_w && a->norm2_b && a->qkv_packed && a->proj_packed && a->fc1_packed &&
                a->fc2_packed, "dua_uvit_block_fwd: null pointer");
    DUA_REQUIRE(a->precision == DUA_PREC_FP32 || a->precision == DUA_PREC_F16X3, "dua_uvit_block_fwd: precision must be DUA_PREC_FP32 or DUA_PREC_F16X3");
    DUA_REQUIRE(a->B > 0 && a->L > 0 && a->D > 0 && a->D % 32 == 0 && a->hidden > 0 && a->hidden % 32 == 0 && a->heads > 0 && a->D % a->heads == 0,
                "dua_uvit_block_fwd: D and hidden must be positive multiples of 32");
    const int ch = a->D / a->heads;
    DUA_REQUIRE(ch % 8 == 0 && ch <= 128, "dua_uvit_block_fwd: head dim %d is outside the fused attention kernels", ch);
    DUA_REQUIRE((a->skip_packed != nullptr) == (a->skip != nullptr), "dua_uvit_block_fwd: skip tokens and skip_linear weights go together");
    DUA_REQUIRE(a->precision != DUA_PREC_F16X3 || (a->act_scale > 0.f && a->raw_scale > 0.f), "dua_uvit_block_fwd: f16x3 needs positive scales");
    return DUA_OK;
}

int check_resblock(const dua_resblock_args* a) {
    DUA_REQUIRE(a && a->x1 && a->y && a->gn1_w && a->gn1_b && a->gn2_w && a->gn2_b && a->film, "dua_adm_resblock_fwd: null pointer");
    DUA_REQUIRE(a->precision == DUA_PREC_FP32 || a->precision == DUA_PREC_F16X3 || a->precision == DUA_PREC_F16W,
                "dua_adm_resblock_fwd: precision must be DUA_PREC_FP32, DUA_PREC_F16X3 or DUA_PREC_F16W");
    DUA_REQUIRE(a->N > 0 && a->H > 0 && a->W > 0 && a->C1 > 0 && a->C2 >= 0 && a->Cout > 0 && (a->C1 + a->C2) % 32 == 0 && a->Cout % 32 == 0,
                "dua_adm_resblock_fwd: channel counts must be positive multiples of 32 (GroupNorm32)");
    DUA_REQUIRE((a->C2 == 0) == (a->x2 == nullptr), "dua_adm_resblock_fwd: x2 / C2 mismatch");
    DUA_REQUIRE(a->updown >= -1 && a->updown <= 1 && (a->updown == 0 || (a->C2 == 0 && !a->skip_packed && !a->dropmask)),
                "dua_adm_resblock_fwd: a resampling block takes one source, an identity skip and no dropout mask");
    DUA_REQUIRE(a->skip_packed || a->C1 + a->C2 == a->Cout, "dua_adm_resblock_fwd: a channel-changing block needs the 1x1 skip weights");
    DUA_REQUIRE(a->precision == DUA_PREC_FP32 || (a->act_scale > 0.f && a->raw_scale > 0.f && a->C1 % 32 == 0), "dua_adm_resblock_fwd: f16x3 needs positive scales and C1 %% 32 == 0");
    {   // the direct-form weights of a conv may be NULL when its Winograd-form weights are given AND the kernel takes the shape
        const int Ho = a->updown < 0 ? a->H / 2 : (a->updown > 0 ? 2 * a->H : a->H), Wo = a->updown < 0 ? a->W / 2 : (a->updown > 0 ? 2 * a->W : a->W);
        DUA_REQUIRE((a->conv1_wino_m == 0 || a->conv1_wino_m == 2 || a->conv1_wino_m == 4) && (a->conv2_wino_m == 0 || a->conv2_wino_m == 2 || a->conv2_wino_m == 4),
                    "dua_adm_resblock_fwd: conv*_wino_m must be 0, 2 or 4");
        DUA_REQUIRE((a->conv1_wino_m != 4 && a->conv2_wino_m != 4) || a->precision == DUA_PREC_F16X3 || a->precision == DUA_PREC_FP32,
                    "dua_adm_resblock_fwd: the F(4,3) packing is a DUA_PREC_F16X3 / DUA_PREC_FP32 option");
        const bool w1 = a->conv1_wino && a->C1 % 32 == 0 && (a->conv1_wino_m == 4 ? dua_conv3x3_f16x3_wino4_supported : dua_conv3x3_f16x3_wino_supported)(a->N, Ho, Wo, a->C1 + a->C2, a->Cout) != 0;
        const bool w2 = a->conv2_wino && (a->conv2_wino_m == 4 ? dua_conv3x3_f16x3_wino4_supported : dua_conv3x3_f16x3_wino_supported)(a->N, Ho, Wo, a->Cout, a->Cout) != 0;
        DUA_REQUIRE((a->conv1_packed || w1) && (a->conv2_packed || w2),
                    "dua_adm_resblock_fwd: conv*_packed may only be NULL for a conv that runs in the Winograd form (conv*_wino given, shape supported)");
    }
    return DUA_OK;
}

int check_attnblock(const dua_attnblock_args* a) {
    DUA_REQUIRE(a && a->x && a->y && a->gn_w && a->gn_b && a->qkv_packed && a->qkv_b && a->proj_packed && a->proj_b, "dua_adm_attnblock_fwd: null pointer");
    DUA_REQUIRE(a->precision == DUA_PREC_FP32 || a->precision == DUA_PREC_F16X3, "dua_adm_attnblock_fwd: precision must be DUA_PREC_FP32 or DUA_PREC_F16X3");
    DUA_REQUIRE(a->N > 0 && a->H > 0 && a->W > 0 && a->C > 0 && a->C % 32 == 0 && a->heads > 0 && a->C % a->heads == 0, "dua_adm_attnblock_fwd: bad shape");
    const int ch = a->C / a->heads;
    DUA_REQUIRE(a->precision == DUA_PREC_F16X3 ? (ch % 4 == 0 && ch >= 8 && ch <= 256) : (ch % 8 == 0 && ch <= 128),
                "dua_adm_attnblock_fwd: head dim %d is outside the fused attention kernels of this precision", ch);
    DUA_REQUIRE(a->precision != DUA_PREC_F16X3 || (a->act_scale > 0.f && a->raw_scale > 0.f), "dua_adm_attnblock_fwd: f16x3 needs positive scales");
    return DUA_OK;
}

}  // namespace

// ---- per-entry-point launch timing (DuaKTimer of common.h): bench facility, off by default, not thread-safe
namespace {
struct KRec { hipEvent_t a, b; const char* kind; double bytes, flops; };
bool g_kprof = false;
std::vector<KRec> g_krecs;
}  // namespace
extern "C" void* dua_kprof_begin_(void* stream) {
    if (!g_kprof) return nullptr;
    hipEvent_t a;
    if (hipEventCreate(&a) != hipSuccess) return nullptr;
    (void)hipEventRecord(a, as_stream(stream));
    return (void*)a;
}
extern "C" void dua_kprof_end_(void* h, void* stream, const char* kind, double bytes, double flops) {
    hipEvent_t b;
    if (hipEventCreate(&b) != hipSuccess) { (void)hipEventDestroy((hipEvent_t)h); return; }
    (void)hipEventRecord(b, as_stream(stream));
    g_krecs.push_back({(hipEvent_t)h, b, kind, bytes, flops});
}
extern "C" int dua_kernel_profile(int on) {
    const int was = g_kprof ? 1 : 0;
    g_kprof = on != 0;
    return was;
}
extern "C" long dua_kernel_profile_collect(char* buf, size_t cap) {
    struct Sum { long n = 0; double ms = 0, bytes = 0, flops = 0, min_ms = 1e30; };
    std::map<std::string, Sum> sums;
    {   // the buffer is checked BEFORE the records are consumed: a too-small buffer loses nothing
        size_t need = 3;
        std::map<std::string, int> kinds;
        for (KRec& r : g_krecs) kinds[r.kind] = 1;
        for (auto& kv : kinds) need += kv.first.size() + 192;
        if (!buf || cap < need) { dua_set_error("dua_kernel_profile_collect: buffer of %zu bytes, %zu needed", cap, need); return DUA_EINVAL; }
    }
    for (KRec& r : g_krecs) {
        float t = 0.f;
        if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&t, r.a, r.b) == hipSuccess) {
            Sum& s = sums[r.kind];
            s.n += 1; s.ms += t; s.bytes += r.bytes; s.flops += r.flops;
            if (t < s.min_ms) s.min_ms = t;
        }
        (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
    }
    g_krecs.clear();
    std::string out = "{";
    char line[512];
    for (auto& kv : sums) {
        snprintf(line, sizeof(line), "%s\"%s\": {\"launches\": %ld, \"ms\": %.6f, \"min_ms\": %.6f, \"bytes\": %.9e, \"flops\": %.9e}", out.size() > 1 ? ", " : "",
                 kv.first.c_str(), kv.second.n, kv.second.ms, kv.second.min_ms, kv.second.bytes, kv.second.flops);
        out += line;
    }
    out += "}";
    std::memcpy(buf, out.c_str(), out.size() + 1);
    return (long)out.size();
}

extern "C" int dua_conv_profile(int on) {
    const int was = g_prof ? 1 : 0;
    g_prof = on != 0;
    return was;
}
extern "C" int dua_conv_profile_collect(double* total_ms, double* total_flops, long* launches, double* total_mfma_flops) {
    double ms = 0.0, fl = 0.0, mf = 0.0;
    long n = 0;
    for (ConvRec& r : g_recs) {
        float t = 0.f;
        if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&t, r.a, r.b) == hipSuccess) { ms += t; fl += r.flops; mf += r.mfma; ++n; }
        (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
    }
    g_recs.clear();
    if (total_ms) *total_ms = ms;
    if (total_flops) *total_flops = fl;
    if (launches) *launches = n;
    if (total_mfma_flops) *total_mfma_flops = mf;
    return DUA_OK;
}

extern "C" size_t dua_adm_resblock_workspace_bytes(const dua_resblock_args* a) {
    if (check_resblock(a) != DUA_OK) return 0;
    Carver w{nullptr, 0, 0, true};
    resblock_run(a, w, nullptr);
    return w.used + 256;
}
extern "C" int dua_adm_resblock_fwd(const dua_resblock_args* a, void* stream) {
    int rc = check_resblock(a);
    if (rc) return rc;
    Carver w{(char*)a->workspace, 0, a->workspace_bytes, false};
    return resblock_run(a, w, stream);
}
extern "C" size_t dua_adm_attnblock_workspace_bytes(const dua_attnblock_args* a) {
    if (check_attnblock(a) != DUA_OK) return 0;
    Carver w{nullptr, 0, 0, true};
    attnblock_run(a, w, nullptr);
    return w.used + 256;
}
extern "C" int dua_adm_attnblock_fwd(const dua_attnblock_args* a, void* stream) {
    int rc = check_attnblock(a);
    if (rc) return rc;
    Carver w{(char*)a->workspace, 0, a->workspace_bytes, false};
    return attnblock_run(a, w, stream);
}
extern "C" size_t dua_uvit_block_workspace_bytes(const dua_uvitblock_args* a) {
    if (check_uvitblock(a) != DUA_OK) return 0;
    Carver w{nullptr, 0, 0, true};
    uvitblock_run(a, w, nullptr);
    return w.used + 256;
}
extern "C" int dua_uvit_block_fwd(const dua_uvitblock_args* a, void* stream) {
    int rc = check_uvitblock(a);
    if (rc) return rc;
    Carver w{(char*)a->workspace, 0, a->workspace_bytes, false};
    return uvitblock_run(a, w, stream);
}
