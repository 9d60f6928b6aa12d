// f16x3 one-tap GEMM  C[M][N] = A[M][K] * W[N][K]^T  on v_mfma_f32_16x16x32_f16 (gfx950): the linear layers of U-ViT
// (uvit/uvit.py:54-120, timm.py:96-111), the qkv / proj_out 1x1 convolutions of the ADM attention blocks (unet_openai.py:287,295) and
// the 1x1 skip convolutions (unet_openai.py:219-223) once their operand exists as planes.
//
// Operands: activation planes [hi | lo][M][K] of x * act_scale (fp16 hi + fp16 remainder, written by the LayerNorm / GroupNorm /
// attention / GELU-epilogue kernels), optionally two sources concatenated along K; weights packed by dua_pack_weight_f16x3
// ([row tile of 128][K step of 32][hi | lo][128][32] + per-row 2^-s).  Product = w_hi*x_lo + w_hi*x_hi + w_lo*x_hi, fp32 accumulate.
//
// Why a kernel of its own (k_xconv_dma<NT, 3> / k_hconv<NT, false> ran these at 230 - 340 TFLOP/s algorithmic; the 3x3 kernels of the
// same arithmetic reach 430 - 450):
//   * a one-tap GEMM has 16 - 144 K steps per tile, not 72 - 576: everything outside the steady K loop is exposed.  The loop keeps the
//     discipline of k_wconv (conv_wino_f16x3.hip): ring slots are COMPILE-TIME constants (six-step body, rings of 3 and 2 or 3), fragment
//     addresses are loop-invariant registers plus immediates, one s_barrier per step, the DMA pieces of a step sit one per MFMA pair;
//     the first stages of the workgroup's NEXT work item are requested before the epilogue, and the item-start wait leaves the
//     epilogue's last stores in flight;
//   * tile widths of 256, 192 and 128 columns (N = 1152 / 3456 / 4608 are multiples of 192, not of 256);
//   * stream-K tail: the tiles of the last, partly filled round are cut into K-step units and spread evenly over the 256 workgroups
//     (585 tiles on 256 CUs were three rounds for 2.3 rounds of work); the parts go through a workspace in accumulator layout and
//     k_hgemm_reduce folds them in a fixed order (deterministic) through the same epilogue.
// Epilogues: fp32 output with bias / GELU / residual, the next GroupNorm's column sums and optionally the f16x3 planes of the result
// (EPI 0); or GELU + f16x3 planes only (EPI 1: U-ViT fc1 -> fc2).
// LDS image as in k_hconv / k_wconv: rows of 64 B, 16-B chunk c of row r at chunk position c ^ (2 * ((r >> 2) & 1)); the DMA writes
// lane-linear 1-KB pieces (16 rows), so the swizzle is applied to each lane's SOURCE chunk.
#include "common.h"
#include "hgemm.h"
#include <type_traits>

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define HG_ROWS 256
// Timing-only diagnostic builds (scripts/hgemm_diag.py; WRONG results, never in the product library): HG_DIAG bit 0 drops the epilogue,
// bit 1 its output stores (dropped by the buffer unit), bit 2 the prefetch of the next item's first stages, bit 3 the GELU.
#ifndef HG_DIAG
#define HG_DIAG 0
#endif
#ifndef HG_COND_LOADS
#define HG_COND_LOADS 0              // 1: the round-5 epilogue loads (conditional global loads) - A/B builds only (scripts/diag/hgemm_condloads_build.py)
#endif
#define HG_TAIL_WAIT 8               // vector-memory operations an epilogue ALWAYS issues after the next item's prologue DMA (stores)

// Exact GELU v * Phi(v) (uvit/timm.py:96-111, nn.GELU) without the two-branch erff: with x = |v| / sqrt(2), erfc(x) = exp(-x g(x)) and g a
// degree-8 polynomial on [0, 4] (least-squares fit weighted by the sensitivity of erfc; beyond 4 erfc < 2e-8), so Phi(v) = erfc(x) / 2 for
// v < 0 - no cancellation on the negative side, unlike 0.5 (1 + erf) - and 1 - erfc(x) / 2 otherwise.
// Two elements at a time on the packed fp32 pipe (v_pk_mul / v_pk_fma_f32: the polynomial, the exponent and the final combination are one
// instruction per pair; as 18 scalar operations per element the GELU was an eighth of fc1's time, HG_DIAG 8).  Folded for that: -log2(e) into
// the coefficients, the 1/2 and a sqrt(2) into the exponent (exp2(x g' - 1/2) = sqrt(2) erfc(x) / 2 =: hs), and
// v Phi(v) = max(v, 0) - |v| he = (v + |v|) / 2 - x hs on both sides of zero (v >= 0: v - v he; v < 0: v he = -|v| he): no select, no 1 - he.
// The product uses the CLAMPED x (not |v|): beyond |v| = 4 sqrt(2) the tail term stays below 4.5e-8 instead of growing with |v|
// (|v| he reached -7.9e21 at v = -1e30 and inf - inf = NaN at v = +inf), and (v + |v|) / 2 - formed as v/2 + |v|/2, exact, no
// overflow - propagates NaN and gives +inf -> +inf, -inf -> NaN like torch's v Phi(v).  Against fp64 over [-12, 12]: max abs error 2.4e-7
// (the fp32 erf formula: 4.5e-7), no divergence.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 hg_gelu2(f32x2 v) {
    if (HG_DIAG & 8) return v;
    const f32x2 a = {fabsf(v[0]), fabsf(v[1])};
    f32x2 x = a * 0.70710678118654752f;
    x[0] = fminf(x[0], 4.0f); x[1] = fminf(x[1], 4.0f);
    constexpr float L = -1.4426950408889634f;
    f32x2 g = {-8.056373400e-06f * L, -8.056373400e-06f * L};
    g = __builtin_elementwise_fma(g, x, (f32x2){1.061473860e-04f * L, 1.061473860e-04f * L});
    g = __builtin_elementwise_fma(g, x, (f32x2){-5.884277713e-04f * L, -5.884277713e-04f * L});
    g = __builtin_elementwise_fma(g, x, (f32x2){1.577756018e-03f * L, 1.577756018e-03f * L});
    g = __builtin_elementwise_fma(g, x, (f32x2){-5.995723508e-05f * L, -5.995723508e-05f * L});
    g = __builtin_elementwise_fma(g, x, (f32x2){-1.921636361e-02f * L, -1.921636361e-02f * L});
    g = __builtin_elementwise_fma(g, x, (f32x2){1.027989408e-01f * L, 1.027989408e-01f * L});
    g = __builtin_elementwise_fma(g, x, (f32x2){6.366161475e-01f * L, 6.366161475e-01f * L});
    g = __builtin_elementwise_fma(g, x, (f32x2){1.128379331e+00f * L, 1.128379331e+00f * L});
    const f32x2 e = __builtin_elementwise_fma(x, g, (f32x2){-0.5f, -0.5f});
    const f32x2 hs = {__builtin_amdgcn_exp2f(e[0]), __builtin_amdgcn_exp2f(e[1])};
    const f32x2 r = __builtin_elementwise_fma(v, (f32x2){0.5f, 0.5f}, a * 0.5f);
    return __builtin_elementwise_fma(-x, hs, r);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Epilogues of one 256 x (64 NT) tile from the accumulator layout acc[channel block][row block]: lane = channels 4 (lane >> 4) .. + 3 of row
// lane & 15.  Every output store is a buffer store that ALWAYS issues (lanes outside the tensor get an out-of-range offset, which the
// buffer unit drops): at least HG_TAIL_WAIT of them follow the next item's prologue DMA in the queue.
//
// Patch form: channel-block pairs (32 channels) of one 16-row block go through a 16 x 128-byte LDS patch of the wave (two patches,
// alternating, 16-B chunk c of row r at chunk position c ^ ((r >> 1) & 7): conflict-free for the accumulator-layout writes and for both
// row-major read shapes), so that residual reads and output stores move whole 128-byte rows; the write of slot t + 2 and the read of slot
// t + 1 are in the LDS queue while slot t is computed and stored.
template <int NT, int EPI>
__device__ __forceinline__ void hgemm_epilogue(const HGemmP& p, f32x4 (&acc)[2 * NT][4], char* pw, int wave, int lane, int m0, int n0) {
    static_assert(8 * NT >= HG_TAIL_WAIT, "the item-start wait counts on this many stores");
    const int lane_e = dua_lane_fresh();         // formed here: the kernel's own lane id, kept across the K loop, was spilled and came back behind a vmcnt(0)
    const int wm = wave >> 1, wn = wave & 1;
    const int lr_e = lane_e & 15, lc_e = lane_e >> 4;
    const int row0 = m0 + wm * 64;
    const unsigned OOB = 0xFFFFFFF0u;
    constexpr int T = 4 * NT;                                    // slots (pair jp = t >> 2, row block i = t & 3)
    const int wr_off = lr_e * 128, wr_sw = (lr_e >> 1) & 7;
    auto patch_put = [&](int t) {
        char* b = pw + (t & 1) * 2048 + wr_off;
        *(f32x4*)(b + 16 * (lc_e ^ wr_sw)) = acc[2 * (t >> 2)][t & 3];
        *(f32x4*)(b + 16 * ((4 + lc_e) ^ wr_sw)) = acc[2 * (t >> 2) + 1][t & 3];
    };
    if constexpr (EPI == 0) {
        const long crem = ((long)p.M - m0) * p.N * 4;
        const auto rc = __builtin_amdgcn_make_buffer_rsrc((void*)(p.C + (long)m0 * p.N), 0, (int)(crem < 0x7FFFFFF0L ? crem : 0x7FFFFFF0L), 0x00020000);
        const long xrem = ((long)p.M - m0) * p.N * 2;
        const int xrec = p.Cextra ? (int)(xrem < 0x7FFFFFF0L ? xrem : 0x7FFFFFF0L) : 0;
        const auto rxh = __builtin_amdgcn_make_buffer_rsrc((void*)(p.Cextra + (long)m0 * p.N), 0, xrec, 0x00020000);
        const auto rxl = __builtin_amdgcn_make_buffer_rsrc((void*)(p.Cextra + (long)p.M * p.N + (long)m0 * p.N), 0, xrec, 0x00020000);
        const int pr_e = lane_e >> 3, pc_e = lane_e & 7;         // row-major side: row of the half block, 16-B chunk of the 128-B row
        // every global operand of the epilogue is requested up front (column scales, bias, the first residual rows): one exposed
        // latency instead of two per pair; the residual rows of slot t + 2 are requested while slot t is stored
        constexpr int RR = NT == 4 ? 2 : 3;                      // residual ring: slots ahead + 1 (the 256-column kernel has no registers for three)
        float4 rres[RR][2];
        // Residual rows, column scales and biases through buffer loads that ALWAYS issue (a lane outside the tensor - or every lane when
        // there is no residual / bias: a descriptor of zero records - reads zeros): the epilogue is then straight-line code between its
        // stores and the compiler's vector-memory waits count the loads exactly.  Behind a conditional load it has to assume the load
        // did not issue and waits vmcnt(0) at the next use - draining, in order, every output store in flight: four to seven times per
        // tile in the round-5 code (the fix that paid in k_wconv4)
        const long rrem = ((long)p.M - m0) * p.N * 4;
        const auto rrs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.res ? p.res + (long)m0 * p.N : nullptr), 0,
                                                           p.res ? (int)(rrem < 0x7FFFFFF0L ? rrem : 0x7FFFFFF0L) : 0, 0x00020000);
        const auto rsc = __builtin_amdgcn_make_buffer_rsrc((void*)p.Cscale, 0, p.N * 4, 0x00020000);
        const auto rbi = __builtin_amdgcn_make_buffer_rsrc((void*)p.bias, 0, p.bias ? p.N * 4 : 0, 0x00020000);
        auto ldq = [&](const auto& rs, unsigned off) {
            const u32x4 q = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
            return make_float4(__uint_as_float(q[0]), __uint_as_float(q[1]), __uint_as_float(q[2]), __uint_as_float(q[3]));
        };
        auto res_ld = [&](int t, float4 (&dst)[2]) {
            const int c = n0 + wn * 32 * NT + 32 * (t >> 2) + 4 * pc_e;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int row = row0 + 16 * (t & 3) + 8 * h + pr_e;
#if HG_COND_LOADS
                dst[h] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (p.res && c < p.N && row < p.M) dst[h] = *(const float4*)(p.res + (long)row * p.N + c);
#else
                dst[h] = ldq(rrs, (c < p.N && row < p.M) ? (unsigned)(((long)(row - m0) * p.N + c) * 4) : OOB);
#endif
            }
        };
        // (column scale and bias of a pair are requested one pair ahead: all NT pairs up front cost 32 registers at NT = 4)
        float4 sqn, bqn;
        auto sb_ld = [&](int jp) {
            const int c = n0 + wn * 32 * NT + 32 * jp + 4 * pc_e;
#if HG_COND_LOADS
            sqn = make_float4(0.f, 0.f, 0.f, 0.f);
            bqn = sqn;
            if (c < p.N) {
                sqn = *(const float4*)(p.Cscale + c);
                if (p.bias) bqn = *(const float4*)(p.bias + c);
            }
#else
            sqn = ldq(rsc, c < p.N ? (unsigned)c * 4u : OOB);
            bqn = ldq(rbi, c < p.N ? (unsigned)c * 4u : OOB);
#endif
        };
        sb_ld(0);
        float4 sq_cur = sqn, bq_cur = bqn;
        res_ld(0, rres[0]);
        if (RR == 3) res_ld(1, rres[1]);
        patch_put(0);
        patch_put(1);
        f32x4 pa[2][2];
        auto patch_get = [&](int t, f32x4 (&d)[2]) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int r = 8 * h + pr_e;
                d[h] = *(const f32x4*)(pw + (t & 1) * 2048 + r * 128 + 16 * (pc_e ^ ((r >> 1) & 7)));
            }
        };
        patch_get(0, pa[0]);
        float4 cs = make_float4(0.f, 0.f, 0.f, 0.f), cq2 = cs;
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const int jp = t >> 2, i = t & 3;
            const int c = n0 + wn * 32 * NT + 32 * jp + 4 * pc_e;
            const bool cok = c < p.N;
            if (t + 1 < T) patch_get(t + 1, pa[(t + 1) & 1]);
            if (t + RR - 1 < T) res_ld(t + RR - 1, rres[(t + RR - 1) % RR]);
            if (i == 0) {
                sq_cur = sqn; bq_cur = bqn;
                if (jp + 1 < NT) sb_ld(jp + 1);
            }
            float4 sq = sq_cur;
            sq.x *= p.out_scale; sq.y *= p.out_scale; sq.z *= p.out_scale; sq.w *= p.out_scale;
            const float4 bq = bq_cur;
            const float4 (&rq)[2] = rres[t % RR];
            if (i == 0) { cs = make_float4(0.f, 0.f, 0.f, 0.f); cq2 = cs; }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const f32x4 a = pa[t & 1][h];
                const int row = row0 + 16 * i + 8 * h + pr_e;
                float4 v;
                v.x = a[0] * sq.x + bq.x; v.y = a[1] * sq.y + bq.y; v.z = a[2] * sq.z + bq.z; v.w = a[3] * sq.w + bq.w;
                if (p.act_out == 2) {
                    const f32x2 g0 = hg_gelu2((f32x2){v.x, v.y}), g1 = hg_gelu2((f32x2){v.z, v.w});
                    v.x = g0[0]; v.y = g0[1]; v.z = g1[0]; v.w = g1[1];
                }
                v.x += rq[h].x; v.y += rq[h].y; v.z += rq[h].z; v.w += rq[h].w;
                const bool ok = cok && row < p.M;
                const bool xcol = p.Cextra && c >= p.extra_from;              // (uniform per 32-column pair: extra_from % 32 == 0)
                const u32x4 vb = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
                if (!(xcol && p.extra_only))
                    __builtin_amdgcn_raw_buffer_store_b128(vb, rc, (ok && !(HG_DIAG & 2)) ? (unsigned)(((long)(row - m0) * p.N + c) * 4) : OOB, 0, 0);
                if (xcol) {                   // the same values as f16x3 planes (operand of a later presplit GEMM / the attention kernel)
                    typedef _Float16 f16x4_t __attribute__((ext_vector_type(4)));
                    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                    const float f[4] = {v.x * p.split_scale, v.y * p.split_scale, v.z * p.split_scale, v.w * p.split_scale};
                    f16x4_t hh, ll;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        if (unsigned long long* rc_ = dua_range_ctr()) dua_range_note(rc_, f[q]);
                        const float sv = fminf(fmaxf(f[q], -65504.f), 65504.f);
                        const _Float16 t16 = (_Float16)sv;
                        hh[q] = t16;
                        ll[q] = (_Float16)(sv - (float)t16);
                    }
                    const unsigned xo = ok ? (unsigned)(((long)(row - m0) * p.N + c) * 2) : OOB;
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, hh), rxh, xo, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, ll), rxl, xo, 0, 0);
                }
                if (ok) {
                    cs.x += v.x; cs.y += v.y; cs.z += v.z; cs.w += v.w;
                    cq2.x += v.x * v.x; cq2.y += v.y * v.y; cq2.z += v.z * v.z; cq2.w += v.w * v.w;
                }
            }
            if (t + 2 < T) patch_put(t + 2);     // (its patch was read for slot t: the LDS queue is in order)
            if (i == 3 && p.colsum) {            // GroupNorm statistics of the next layer: fold the 8 row lanes that share these channels
                float u_[8] = {cs.x, cs.y, cs.z, cs.w, cq2.x, cq2.y, cq2.z, cq2.w};
                const int lf_ = dua_lane_fresh();            // (not __shfl_xor: its hoisted lane id is spilled and reloaded behind a vmcnt(0), common.h)
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    u_[u] += dua_xor_fresh(u_[u], lf_, 8);
                    u_[u] += dua_xor_fresh(u_[u], lf_, 16);
                    u_[u] += dua_xor_fresh(u_[u], lf_, 32);
                }
                if (pr_e == 0 && cok && row0 < p.M) {
                    float* o = p.colsum + (long)(row0 >> 6) * 2 * p.N + c;
                    *(float4*)o = make_float4(u_[0], u_[1], u_[2], u_[3]);
                    *(float4*)(o + p.N) = make_float4(u_[4], u_[5], u_[6], u_[7]);
                }
            }
        }
    } else {
        // GELU (or identity) -> f16x3 planes: lane = row lane >> 2 of the 16-row block, channels 8 (lane & 3) .. + 7 of the 32-channel pair
        const long crem = ((long)p.M - m0) * p.N * 2;
        const int crec = (int)(crem < 0x7FFFFFF0L ? crem : 0x7FFFFFF0L);
        const auto rch = __builtin_amdgcn_make_buffer_rsrc((void*)(p.Cplanes + (long)m0 * p.N), 0, crec, 0x00020000);
        const auto rcl = __builtin_amdgcn_make_buffer_rsrc((void*)(p.Cplanes + (long)p.M * p.N + (long)m0 * p.N), 0, crec, 0x00020000);
        const int r16 = lane_e >> 2, ch8 = lane_e & 3;
        // column scales and biases of a pair (this lane's eight channels), requested ONE pair ahead (all NT pairs up front are 64 registers
        // at NT = 4: the K loop of that instantiation then reloaded a spilled LDS address behind an s_waitcnt vmcnt(0) - scripts/isa_lint.py)
        float4 s0n, s1n, b0n, b1n;
        // (always-issued buffer loads, as in the fp32 epilogue above: no conditional load between the plane stores)
        const auto rsc = __builtin_amdgcn_make_buffer_rsrc((void*)p.Cscale, 0, p.N * 4, 0x00020000);
        const auto rbi = __builtin_amdgcn_make_buffer_rsrc((void*)p.bias, 0, p.bias ? p.N * 4 : 0, 0x00020000);
        auto ldq = [&](const auto& rs, unsigned off) {
            const u32x4 q = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
            return make_float4(__uint_as_float(q[0]), __uint_as_float(q[1]), __uint_as_float(q[2]), __uint_as_float(q[3]));
        };
        auto sb_ld8 = [&](int jp) {
            const int c = n0 + wn * 32 * NT + 32 * jp + 8 * ch8;
#if HG_COND_LOADS
            s0n = make_float4(0.f, 0.f, 0.f, 0.f);
            s1n = s0n; b0n = s0n; b1n = s0n;
            if (c < p.N) {
                s0n = *(const float4*)(p.Cscale + c); s1n = *(const float4*)(p.Cscale + c + 4);
                if (p.bias) { b0n = *(const float4*)(p.bias + c); b1n = *(const float4*)(p.bias + c + 4); }
            }
#else
            const unsigned o = c < p.N ? (unsigned)c * 4u : OOB;             // (N % 8 == 0 on this path: the second chunk is in range with the first)
            s0n = ldq(rsc, o); s1n = ldq(rsc, o == OOB ? OOB : o + 16u);
            b0n = ldq(rbi, o); b1n = ldq(rbi, o == OOB ? OOB : o + 16u);
#endif
        };
        sb_ld8(0);
        float4 s0c = s0n, s1c = s1n, b0c = b0n, b1c = b1n;
        patch_put(0);
        patch_put(1);
        f32x4 pa[2][2];
        const int rd_sw = (r16 >> 1) & 7;
        auto patch_get = [&](int t, f32x4 (&d)[2]) {
            const char* b = pw + (t & 1) * 2048 + r16 * 128;
            d[0] = *(const f32x4*)(b + 16 * ((2 * ch8) ^ rd_sw));
            d[1] = *(const f32x4*)(b + 16 * ((2 * ch8 + 1) ^ rd_sw));
        };
        patch_get(0, pa[0]);
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const int jp = t >> 2, i = t & 3;
            const int c = n0 + wn * 32 * NT + 32 * jp + 8 * ch8;
            const bool cok = c < p.N;
            if (t + 1 < T) patch_get(t + 1, pa[(t + 1) & 1]);
            if (i == 0) {
                s0c = s0n; s1c = s1n; b0c = b0n; b1c = b1n;
                if (jp + 1 < NT) sb_ld8(jp + 1);
            }
            const float4 s0 = s0c, s1 = s1c, b0 = b0c, b1 = b1c;
            const float sq[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
            const float bq[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
            const f32x4 a0 = pa[t & 1][0], a1 = pa[t & 1][1];
            const float a[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
            const int row = row0 + 16 * i + r16;
            h16x8 hh, ll;
            float vv[8];
#pragma unroll
            for (int q = 0; q < 8; q += 2) {
                f32x2 v2 = {a[q] * (sq[q] * p.out_scale) + bq[q], a[q + 1] * (sq[q + 1] * p.out_scale) + bq[q + 1]};
                if (p.act_out == 2) v2 = hg_gelu2(v2);
                vv[q] = v2[0]; vv[q + 1] = v2[1];
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float v = vv[q];
                if (unsigned long long* rc_ = dua_range_ctr()) dua_range_note(rc_, v * p.split_scale);
                const float sv = fminf(fmaxf(v * p.split_scale, -65504.f), 65504.f);
                const _Float16 t16 = (_Float16)sv;
                hh[q] = t16;
                ll[q] = (_Float16)(sv - (float)t16);
            }
            const unsigned o = (cok && row < p.M && !(HG_DIAG & 2)) ? (unsigned)(((long)(row - m0) * p.N + c) * 2) : OOB;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, hh), rch, o, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, ll), rcl, o, 0, 0);
            if (t + 2 < T) patch_put(t + 2);
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------------------------
template <int NT, int EPI>
__global__ __launch_bounds__(512, 2) void k_hgemm(HGemmP p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int NB = 2 * NT;                               // 16-channel blocks of a wave (wave tile = 64 rows x 32 NT channels)
    constexpr int H2 = NT >= 3 ? 2 : 1;                      // the weight fragments are held per half of the wave's channel blocks
    constexpr int HB = NB / H2;
    constexpr int RA = 3, RB = NT == 2 ? 3 : 2;              // ring depths (stages)
    constexpr int APL = HG_ROWS * 64, BPL = 64 * NT * 64;    // bytes per plane of a stage
    constexpr int ABY = 2 * APL, BBY = 2 * BPL;              // bytes per stage
    constexpr int BOFF = RA * ABY;
    static_assert(RA * ABY + RB * BBY <= 160 * 1024, "LDS of a gfx950 CU");
    __shared__ __attribute__((aligned(16))) char lds[RA * ABY + RB * BBY];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int nk = (p.K1 + p.K2) / 32, nk1 = p.K1 / 32;
    // persistent workgroups; whole tiles chunked per XCD (blockIdx % 8) so that co-running workgroups of an XCD share operand panels in its L2
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslot = (gridDim.x + 7 - xcd) >> 3;
    const int cq = p.n_full >> 3, cr = p.n_full & 7;
    const int chunk0 = xcd < cr ? xcd * (cq + 1) : cr * (cq + 1) + (xcd - cr) * cq, chunk_n = cq + (xcd < cr ? 1 : 0);
    const int n_head = slot < chunk_n ? (chunk_n - slot + nslot - 1) / nslot : 0;
    // stream-K part of the work list: the K-step units of the tiles [n_full, n_full + tail_tiles) are cut into tail_p contiguous ranges,
    // workgroup cu < tail_p owns [cu * U / tail_p, (cu + 1) * U / tail_p): a partial first tile, whole tiles, a partial last tile
    const int cu = xcd * (gridDim.x >> 3) + slot;
    const long U = (long)p.tail_tiles * nk;
    long u0 = 0, u1 = 0;
    if (p.tail_tiles > 0 && cu < p.tail_p) { u0 = ((long)cu * U) / p.tail_p; u1 = ((long)(cu + 1) * U) / p.tail_p; }
    const int seg_first = (int)(u0 / nk), seg_k0 = (int)(u0 - (long)seg_first * nk);
    const int seg_last = u1 > u0 ? (int)((u1 - 1) / nk) : seg_first - 1, seg_k1 = (int)(u1 - (long)seg_last * nk);
    const int n_seg = seg_last - seg_first + 1;
    const int n_items = n_head + n_seg;
    // item idx -> (tile id, K range, workspace slot or -1 for a whole tile with the real epilogue)
    auto get_item = [&](int idx, int& tile, int& k0, int& k1, int& part) {
        if (idx < n_head) { tile = __builtin_amdgcn_readfirstlane(chunk0 + slot + idx * nslot); k0 = 0; k1 = nk; part = -1; return; }
        const int j = idx - n_head;
        tile = p.n_full + seg_first + j;
        k0 = j == 0 ? seg_k0 : 0;
        k1 = j == n_seg - 1 ? seg_k1 : nk;
        part = (k0 == 0 && k1 == nk) ? -1 : 2 * cu + (j == 0 ? 0 : 1);
        // (wave-uniform, and SAID so: `part` otherwise lives in a vector register, is spilled by the 256-column instantiations and comes
        // back from scratch behind an s_waitcnt vmcnt(0) in front of the epilogue - a wait for the next item's prologue DMA, once per tile)
        tile = __builtin_amdgcn_readfirstlane(tile); k0 = __builtin_amdgcn_readfirstlane(k0);
        k1 = __builtin_amdgcn_readfirstlane(k1); part = __builtin_amdgcn_readfirstlane(part);
    };

    const int dl_row = lane >> 2;
    const int dl_src = (lane & 3) ^ (2 * ((lane >> 4) & 1));
    const int lr = lane & 15, lc = lane >> 4;

    // ---- DMA state (lives across items: the next item's first stages are requested before this item's epilogue)
    // Activation pieces: per-lane offset = (row of the tile, clamped to the last row of A) * K * 2 + source chunk, constant per item; the K
    // step and the hi -> lo plane distance ride in the scalar offset (not bounds-checked, so the descriptors need no tight record).
    // Weight pieces: the per-lane part (row of the 16-row group, source chunk) is the same for every piece; (128-row block, group, plane,
    // K step) ride in the scalar offset.  Column tiles that reach beyond the packed rows read the last packed block instead: an output
    // column depends on its own weight row only, and columns >= N are never stored.
    unsigned aoff1[2] = {0, 0}, aoff2[2] = {0, 0};
    const unsigned blane = (unsigned)(dl_row * 64 + 16 * dl_src);
    unsigned sb_off[NT];                                     // scalar: byte offset of this wave's piece j of the next weight stage
    int bdst[NT];
    auto ra1 = __builtin_amdgcn_make_buffer_rsrc((void*)p.A1, 0, 0x7FFFFFF0, 0x00020000);
    auto ra2 = ra1;
    const auto rb = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, (int)p.b_bytes, 0x00020000);
    const unsigned lo1 = (unsigned)(p.a1_plane * 2), lo2 = (unsigned)(p.a2_plane * 2);
    const int nblk = (int)(p.b_bytes / ((long)nk * 16384));  // packed 128-row blocks
    int ka = 0;                                              // K step of the next activation stage to request
    bool a_src2 = false;
    unsigned a_soff = 0;
#pragma unroll
    for (int j = 0; j < NT; ++j) { sb_off[j] = 0; bdst[j] = 0; }

    auto item_geom = [&](int m0_, int n0_, int k0_) {
        const int rows = p.M - m0_;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            int m = 16 * (2 * wave + j) + dl_row;
            m = m < rows ? m : rows - 1;                     // rows beyond M re-read the last row (their results are never stored)
            aoff1[j] = (unsigned)(((long)m * p.K1 + 8 * dl_src) * 2);
            aoff2[j] = (unsigned)(((long)m * p.K2 + 8 * dl_src) * 2);
        }
        ra1 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A1 + (long)m0_ * p.K1), 0, 0x7FFFFFF0, 0x00020000);
        if (p.K2) ra2 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A2 + (long)m0_ * p.K2), 0, 0x7FFFFFF0, 0x00020000);
        ka = k0_;
#pragma unroll
        for (int j = 0; j < NT; ++j) {                       // weight pieces of this wave: piece q = wave + 8 j of the stage's 8 NT (plane, 16-row group) pieces
            const int q = wave + 8 * j, pl = q / (4 * NT), g = q - pl * 4 * NT;
            const int r_ = n0_ + 16 * g;
            int blk = r_ >> 7;
            blk = blk < nblk ? blk : nblk - 1;
            sb_off[j] = (unsigned)(((long)blk * nk + k0_) * 16384 + pl * 8192 + (r_ & 127) * 64);
            bdst[j] = pl * BPL + g * 1024;
        }
    };
    auto a_setup = [&]() {                                   // source and scalar offset of activation stage ka
        a_src2 = ka >= nk1;
        a_soff = (unsigned)(a_src2 ? ka - nk1 : ka) * 64u;
        ++ka;
    };
    auto dma_a = [&](int sa, int i) {                        // piece i = 0..3 of the activation stage a_setup() prepared
        char* st = lds + sa * ABY;
        const auto r = a_src2 ? ra2 : ra1;
        const unsigned lo = a_src2 ? lo2 : lo1;
        const unsigned v0 = a_src2 ? aoff2[0] : aoff1[0], v1 = a_src2 ? aoff2[1] : aoff1[1];
        if (i == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, st + (16 * (2 * wave)) * 64, 16, v0, a_soff, 0, 0);
        if (i == 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, st + APL + (16 * (2 * wave)) * 64, 16, v0, a_soff + lo, 0, 0);
        if (i == 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, st + (16 * (2 * wave + 1)) * 64, 16, v1, a_soff, 0, 0);
        if (i == 3) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, st + APL + (16 * (2 * wave + 1)) * 64, 16, v1, a_soff + lo, 0, 0);
    };
    auto dma_b = [&](int sb, int j) {                        // piece j = 0..NT-1 of the next weight stage (sequential in K)
        char* st = lds + BOFF + sb * BBY;
#pragma unroll
        for (int jj = 0; jj < NT; ++jj)
            if (jj == j) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, st + bdst[jj], 16, blane, sb_off[jj], 0, 0);
                sb_off[jj] += 16384;
            }
    };
    auto issue_prologue = [&](int n_) {                      // A(0) B(0) [A(1) B(1)]
        a_setup();
#pragma unroll
        for (int i = 0; i < 4; ++i) dma_a(0, i);
#pragma unroll
        for (int j = 0; j < NT; ++j) dma_b(0, j);
        if (n_ > 1) {
            a_setup();
#pragma unroll
            for (int i = 0; i < 4; ++i) dma_a(1, i);
#pragma unroll
            for (int j = 0; j < NT; ++j) dma_b(1, j);
        }
    };

    const char* ldsb = reinterpret_cast<const char*>(lds);
    const int off_l = lr * 64 + 16 * (lc ^ (2 * ((lr >> 2) & 1)));        // this lane's fragment chunk inside a 16-row block
    int axs[RA], bws[RB];                                                  // loop-invariant fragment bases per ring slot (bytes)
#pragma unroll
    for (int s_ = 0; s_ < RA; ++s_) axs[s_] = s_ * ABY + wm * 64 * 64 + off_l;
#pragma unroll
    for (int s_ = 0; s_ < RB; ++s_) bws[s_] = BOFF + s_ * BBY + wn * 32 * NT * 64 + off_l;

    int tile = 0, k0 = 0, k1 = 0, part = -1;
    bool prefetched = false;
    for (int it = 0; it < n_items; ++it) {
        if (!prefetched) {
            get_item(it, tile, k0, k1, part);
            item_geom((tile / p.ntn) * HG_ROWS, (tile % p.ntn) * 64 * NT, k0);
            issue_prologue(k1 - k0);
        }
        const int m0 = (tile / p.ntn) * HG_ROWS, n0 = (tile % p.ntn) * 64 * NT;
        const int n = k1 - k0, part_now = part;
        const bool after_epilogue = prefetched;
        prefetched = false;

        f32x4 acc[NB][4];
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        h16x8 xh[4], xl[4], wh[HB], wl[HB];
        auto rd_x = [&](int sa, int pl, int i) -> h16x8 { return *(const h16x8*)(ldsb + axs[sa] + (pl * APL + i * 1024)); };
        auto rd_w = [&](int sb, int pl, int jj) -> h16x8 { return *(const h16x8*)(ldsb + bws[sb] + (pl * BPL + jj * 1024)); };
#define GSB() __builtin_amdgcn_sched_barrier(0)

        // ---- item start: the first stages have landed; behind them in the (in-order) queue are only the previous epilogue's
        // operations, of which the last HG_TAIL_WAIT - stores - may stay in flight
        if (after_epilogue) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(HG_TAIL_WAIT) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int i = 0; i < 4; ++i) { xl[i] = rd_x(0, 1, i); xh[i] = rd_x(0, 0, i); }
#pragma unroll
        for (int jj = 0; jj < HB; ++jj) { wh[jj] = rd_w(0, 0, jj); wl[jj] = rd_w(0, 1, jj); }
        GSB();

        // One K step: 12 NB MFMAs.  Stage s lives in activation slot SA_ = s % 3 and weight slot SB_ = s % RB (compile-time: the loop
        // body is six steps).  `more`: a stage s + 1 exists; `pre`: a stage s + 2 exists and is requested in this step - its activation
        // pieces (and, ring of three, its weight pieces) under the first MFMAs; with a weight ring of two the slot of stage s + 2 is the
        // one being read in this step, so those pieces go out after the barrier.
        auto step = [&](auto sa_c, auto sb_c, auto tail_c, bool more_rt, bool pre_rt) __attribute__((always_inline)) {
            constexpr int SA_ = decltype(sa_c)::value, SB_ = decltype(sb_c)::value;
            constexpr bool TAIL = decltype(tail_c)::value;
            constexpr int sa_n = (SA_ + 2) % RA, sb_n = (SB_ + 2) % RB, sa_r = (SA_ + 1) % RA, sb_r = (SB_ + 1) % RB;
            const bool more = TAIL ? more_rt : true, pre = TAIL ? pre_rt : true;
            if (pre) a_setup();
            auto early_piece = [&](int q) {
                if (q < 4) { if (pre) dma_a(sa_n, q); }
                else if (RB == 3 && q < 4 + NT) { if (pre) dma_b(sb_n, q - 4); }
            };
            GSB();
#pragma unroll
            for (int h = 0; h < H2; ++h) {
                const bool lasth = h == H2 - 1;
                // P1: w_hi * x_lo | (first half) the early DMA pieces, one per MFMA pair
#pragma unroll
                for (int jj = 0; jj < HB; ++jj)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        acc[h * HB + jj][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[jj], xl[i], acc[h * HB + jj][i], 0, 0, 0);
                        GSB();
                        const int m_ = jj * 4 + i;
                        if (h == 0 && (m_ & 1) == 0) { early_piece(m_ >> 1); GSB(); }
                    }
                if (lasth) {
                    // every read of stage s has been issued: publish stage s + 1
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    if (pre) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(RB == 3 ? 4 + NT : 4) : "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    GSB();
                }
                // P2: w_hi * x_hi | the hi weight fragments of the other half (same stage) or of the next stage; last half: x_lo of the next stage
#pragma unroll
                for (int jj = 0; jj < HB; ++jj) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        acc[h * HB + jj][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[jj], xh[i], acc[h * HB + jj][i], 0, 0, 0);
                        GSB();
                        if (lasth && jj == 0 && more) { xl[i] = rd_x(sa_r, 1, i); GSB(); }
                    }
                    if (!lasth) wh[jj] = rd_w(SB_, 0, (h + 1) * HB + jj);
                    else if (more) wh[jj] = rd_w(sb_r, 0, jj);
                    GSB();
                    if (lasth && RB == 2 && jj < NT && pre) { dma_b(sb_n, jj); GSB(); }
                }
                // P3: w_lo * x_hi | the lo weight fragments likewise; last half: x_hi of the next stage
#pragma unroll
                for (int jj = 0; jj < HB; ++jj) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        acc[h * HB + jj][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[jj], xh[i], acc[h * HB + jj][i], 0, 0, 0);
                        GSB();
                        if (lasth && jj == HB - 1 && more) { xh[i] = rd_x(sa_r, 0, i); GSB(); }
                    }
                    if (!lasth) wl[jj] = rd_w(SB_, 1, (h + 1) * HB + jj);
                    else if (more) wl[jj] = rd_w(sb_r, 1, jj);
                    GSB();
                }
            }
        };
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        using I2 = std::integral_constant<int, 2>;
        using BF = std::integral_constant<bool, false>;
        using BT = std::integral_constant<bool, true>;
        using B1 = std::integral_constant<int, 1 % RB>;
        using B2 = std::integral_constant<int, 2 % RB>;
        using B3 = std::integral_constant<int, 3 % RB>;
        using B4 = std::integral_constant<int, 4 % RB>;
        using B5 = std::integral_constant<int, 5 % RB>;
        int s = 0;
#pragma nounroll
        for (; s + 8 <= n; s += 6) {                         // steady state: every one of the six steps has a stage s + 2 to request
            step(I0{}, I0{}, BF{}, true, true); step(I1{}, B1{}, BF{}, true, true); step(I2{}, B2{}, BF{}, true, true);
            step(I0{}, B3{}, BF{}, true, true); step(I1{}, B4{}, BF{}, true, true); step(I2{}, B5{}, BF{}, true, true);
        }
        {
            const int r = n - s;                              // 1 .. 7 steps left (slots continue from 0: s is a multiple of six)
            if (r > 0) step(I0{}, I0{}, BT{}, 1 < r, 2 < r);
            if (r > 1) step(I1{}, B1{}, BT{}, 2 < r, 3 < r);
            if (r > 2) step(I2{}, B2{}, BT{}, 3 < r, 4 < r);
            if (r > 3) step(I0{}, B3{}, BT{}, 4 < r, 5 < r);
            if (r > 4) step(I1{}, B4{}, BT{}, 5 < r, 6 < r);
            if (r > 5) step(I2{}, B5{}, BT{}, 6 < r, false);
            if (r > 6) step(I0{}, I0{}, BT{}, false, false);
        }
#undef GSB
        // the last barrier of the item has retired every LDS read: request the first stages of the next item now (activation slots
        // 0, 1 and weight slots 0, 1); their latency runs under the epilogue, whose patches live in activation slot 2
        if (it + 1 < n_items && !(HG_DIAG & 4)) {
            get_item(it + 1, tile, k0, k1, part);
            item_geom((tile / p.ntn) * HG_ROWS, (tile % p.ntn) * 64 * NT, k0);
            issue_prologue(k1 - k0);
            prefetched = true;
        }
        if (part_now >= 0) {
            // stream-K part: the raw accumulators in accumulator layout, 1 KB per store instruction (k_hgemm_reduce folds the parts)
            float* dst = p.kpart + (long)part_now * (HG_ROWS * 64 * NT) + (long)wave * (NB * 4 * 256) + lane * 4;
#pragma unroll
            for (int j = 0; j < NB; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) *(f32x4*)(dst + (j * 4 + i) * 256) = acc[j][i];
        } else if (HG_DIAG & 1) {                            // (diagnostic: no epilogue - the accumulators are kept alive by a store that never happens)
            float t_ = 0.f;
#pragma unroll
            for (int j = 0; j < NB; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) t_ += acc[j][i][0] + acc[j][i][1] + acc[j][i][2] + acc[j][i][3];
            if (t_ == 1.2345e-30f) p.C[0] = t_;
        } else {
            hgemm_epilogue<NT, EPI>(p, acc, lds + 2 * ABY + wave * 4096, wave, lane, m0, n0);
        }
        // the patches must be done before another wave's DMA of the next item's third stage lands in activation slot 2: that piece is
        // requested after the next item's opening barrier
    }
#endif
}

// One workgroup per tail tile: sum the stream-K parts of the tile in workgroup order, then the epilogue of k_hgemm.
template <int NT, int EPI>
__global__ __launch_bounds__(512) void k_hgemm_reduce(HGemmP p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int NB = 2 * NT;
    __shared__ __attribute__((aligned(16))) char patch[8 * 4096];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int t = blockIdx.x, nk = (p.K1 + p.K2) / 32;
    const long U = (long)p.tail_tiles * nk, P = p.tail_p;
    const long ulo = (long)t * nk, uhi = ulo + nk;
    // range c holds unit x  <=>  floor(c U / P) <= x < floor((c + 1) U / P)  <=>  c = floor(((x + 1) P - 1) / U)
    const int c_lo = (int)(((ulo + 1) * P - 1) / U), c_hi = (int)((uhi * P - 1) / U);
    if (c_lo == c_hi) return;                                // one workgroup ran the whole tile (and its epilogue)
    f32x4 acc[NB][4];
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int c = c_lo; c <= c_hi; ++c) {
        const int first = (int)((((long)c * U) / P) / nk);      // the first tile of range c: its part sits in slot 2 c, any other in 2 c + 1
        const int part = 2 * c + (t != first ? 1 : 0);
        const float* src = p.kpart + (long)part * (HG_ROWS * 64 * NT) + (long)wave * (NB * 4 * 256) + lane * 4;
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[j][i] += *(const f32x4*)(src + (j * 4 + i) * 256);
    }
    const int tile = p.n_full + t;
    const int m0 = (tile / p.ntn) * HG_ROWS, n0 = (tile % p.ntn) * 64 * NT;
    hgemm_epilogue<NT, EPI>(p, acc, patch + wave * 4096, wave, lane, m0, n0);
#endif
}

// ---------------------------------------------------------------------------------------------------------------------------------
static float* hg_ws = nullptr;
static size_t hg_ws_bytes = 0;
static int hg_sk_whole = -1;              // stream-K over the whole launch: -1 = per shape, 0 = tail only, 1 = always
static int hg_force_nt = 0, hg_streamk = 1;
void dua_hgemm_set_workspace(void* ws, size_t bytes) { hg_ws = (float*)ws; hg_ws_bytes = bytes; }
// A/B switches (process-global, correct kernels only): nt = 0 picks the tile width per shape, 2 / 3 / 4 forces 128 / 192 / 256 columns,
// -1 sends the presplit GEMMs back to the previous engine (k_xconv_dma<NT, 3> / k_hconv<NT, false>); streamk bit 0 = 0 runs whole tiles
// only (bits 1-2 are ignored: they selected the since-removed epilogue straight from the accumulators, measured no better)
int dua_hgemm_enabled() { return hg_force_nt >= 0; }
extern "C" int dua_set_hgemm_mode(int nt, int streamk) {
    DUA_REQUIRE(nt == -1 || nt == 0 || (nt >= 2 && nt <= 4), "dua_set_hgemm_mode: nt must be -1 (previous engine), 0, 2, 3 or 4");
    hg_force_nt = nt;
    hg_streamk = (streamk & 1) != 0;
    hg_sk_whole = (streamk & 24) == 8 ? 0 : ((streamk & 24) == 16 ? 1 : -1);      // bits 3-4: 8 = stream-K on the tail round only, 16 = on the whole launch
    return DUA_OK;
}

// The caller runs several launches side by side on HIP streams of their own (UViT / UNetModel._forward_streams: the batch in chunks):
// the CUs a launch's last, partly filled round leaves idle are taken by the other stream's workgroups, so the tile width is chosen by
// work per column (width / loop efficiency, columns beyond N included) and not by whole rounds of tiles - U-ViT's fc1 at 32 images
// (594 tiles of 256 columns) picked 128-column tiles for their five full rounds.  Returns the previous setting.  Results do not depend on
// the tile width.
static int hg_concurrent = 0;
extern "C" int dua_set_hgemm_concurrent(int on) {
    const int was = hg_concurrent;
    hg_concurrent = on != 0;
    return was;
}

template <int NT>
static void hg_launch(const HGemmP& p, int grid, bool planes, hipStream_t st) {
    if (planes) {
        hipLaunchKernelGGL((k_hgemm<NT, 1>), dim3(grid), 512, 0, st, p);
        if (p.tail_tiles > 0 && p.tail_p > 1) hipLaunchKernelGGL((k_hgemm_reduce<NT, 1>), dim3(p.tail_tiles), 512, 0, st, p);
    } else {
        hipLaunchKernelGGL((k_hgemm<NT, 0>), dim3(grid), 512, 0, st, p);
        if (p.tail_tiles > 0 && p.tail_p > 1) hipLaunchKernelGGL((k_hgemm_reduce<NT, 0>), dim3(p.tail_tiles), 512, 0, st, p);
    }
}

int dua_launch_hgemm(HGemmP& p, hipStream_t st, const char* what) {
    const int nk = (p.K1 + p.K2) / 32;
    p.ntm = (p.M + HG_ROWS - 1) / HG_ROWS;
    const bool ws_ok = hg_streamk && hg_ws != nullptr;
    // Tile width: estimated launch time = rounds of tiles on 256 CUs x (K steps + ~8 steps of per-tile overhead) x tile width / loop
    // efficiency of that width (the widest tile moves the least L2 -> LDS traffic per MFMA; measured steady rates 256 : 192 : 128 columns
    // = 1 : 0.95 : 0.85, scripts/diag/hgemm_sweep.py); columns beyond N are wasted work of the last column tile.
    int nt = hg_force_nt;
    if (!nt) {
        double best = 1e300;
        for (int c = 4; c >= 2; --c) {
            const int w = 64 * c, tn = (p.N + w - 1) / w;
            const long tiles = (long)p.ntm * tn;
            const double rounds = hg_concurrent ? (double)tiles / 256.0 : (double)((tiles + 255) / 256);
            const double t = rounds * (nk + 8) * c / (c == 4 ? 1.0 : (c == 3 ? 0.95 : 0.85));
            if (t < best) { best = t; nt = c; }
        }
    }
    p.ntn = (p.N + 64 * nt - 1) / (64 * nt);
    const int ntiles = p.ntm * p.ntn;
    p.n_full = ntiles; p.tail_tiles = 0; p.tail_p = 1; p.kpart = nullptr;
    int grid = ntiles < 256 ? ntiles : 256;
    const size_t need = (size_t)2 * 256 * HG_ROWS * 64 * nt * sizeof(float);
    if (ws_ok && nk >= 16 && need <= hg_ws_bytes) {
        // Stream-K.  Default: launches that fill at most 3/8 of the chip (up to 96 tiles: small batches - SD's CFG batch of 2, ADM-64
        // at batch 4); measured on U-ViT-H/2's shapes (scripts/diag/hgemm_sweep.py), neither the tail round alone (585 or 390 tiles on
        // 256 CUs: the part dumps and the fold kernel cost more than the idle half round) nor the whole launch (workgroups out of phase
        // spread their HBM-bound epilogue bursts, but stop sharing the weight stream in their XCD's L2: -22 % in the K loop) pays above
        // that; both stay selectable (dua_set_hgemm_mode) for A/B runs.
        const int rem = ntiles % 256;
        const int tail = hg_sk_whole == 1 ? ntiles : (hg_sk_whole == 0 ? (rem <= 216 ? rem : 0) : (ntiles <= 96 ? ntiles : 0));
        if (tail > 0) {
            const long U = (long)tail * nk;
            long P = U / 12;
            if (P > 256) P = 256;
            if (P < 1) P = 1;
            p.n_full = ntiles - tail; p.tail_tiles = tail; p.tail_p = (int)P; p.kpart = hg_ws;
            grid = 256;
        }
    }
    const bool planes = p.Cplanes != nullptr;
    DuaKTimer kt((void*)st);
    if (nt == 4) hg_launch<4>(p, grid, planes, st);
    else if (nt == 3) hg_launch<3>(p, grid, planes, st);
    else hg_launch<2>(p, grid, planes, st);
    {   // algorithmic: 2 M N K FLOPs; bytes = the operand planes (4 B per element, A once per launch, the packed weights once) + the outputs
        const double K = (double)(p.K1 + p.K2), M = (double)p.M, N = (double)p.N;
        kt.done("k_hgemm", 4.0 * M * K + 4.0 * N * K + (p.C ? 4.0 * M * N : 0.0) + ((p.Cplanes || p.Cextra) ? 4.0 * M * N : 0.0) + (p.res ? 4.0 * M * N : 0.0), 2.0 * M * N * K);
    }
    return dua_check_launch(what);
}

DUA_RANGE_TU(hgemm)
