// Fused (flash-style) multi-head attention on the fp32 matrix cores, exact fp32 arithmetic.
//
// out[b, t, h*ch + c] = sum_s softmax_s(scale * q[b,t,h,:] . k[b,s,h,:]) v[b,s,h,c]
// for qkv rows [B*T][3C] laid out (q | k | v), head-major inside each third (the layout the qkv projections write).
// Replaces the materialised-logits path (batched GEMM -> dua_softmax_rows -> transpose -> batched GEMM) of
// QKVAttentionLegacy / QKVAttention (guided_diffusion/unet_openai.py:338-355, :371-390) and of the fp32 SDPA call of
// U-ViT (uvit/uvit.py:69-73): the T x T logits never leave the CU.
//
// One workgroup = QW waves; every wave owns 32 queries.  Per 32-key tile, with K and V staged once in LDS for the
// whole workgroup:
//   S^T = K . Q^T      v_mfma_f32_32x32x2_f32, A = K rows (keys), B = Q rows: the accumulator then has the QUERY on the
//                      lane (column) and 16 keys in registers, so the softmax running max / sum are lane-local plus
//                      one exchange between the two half-waves;
//   O^T += V^T . P^T   A = V read column-wise from the LDS tile (ds_read_b32, lanes on consecutive channels), B = the
//                      exponentiated accumulator registers as they stand (no LDS round trip for P): O^T again has the
//                      query on the lane, so the online-softmax rescale and the final 1/l are lane-local too.
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct AttnP {
    const float* qkv;
    float* out;
    int T, heads, ch;
    float scale;
    void* out_planes;        // f16x3 kernel only: write the result as f16x3 planes [B*T][C] (operand of the presplit proj GEMM)
    long plane_elems;
    float planes_scale;      // f16x3 kernel: out_planes holds fp16 hi | fp16 lo of out * planes_scale
    // split kernels only: keys / values from their own tensors (cross-attention).  k == NULL: the packed (q | k | v) rows above.
    // q rows at qkv + (b*T + t)*ldq, keys at k + (b*Tk + s)*ldkv, values at v + (b*Tk + s)*ldkv; head h at column h*ch.
    const float *k, *v;
    long ldq, ldkv;
    int Tk;
};

template <int CT /* 32-channel tiles: ceil(ch/32) */, int QW /* waves (32 queries each) per workgroup */>
__global__ __launch_bounds__(64 * QW) void k_flash_attn(AttnP p) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int ch = p.ch, C = p.heads * ch, T = p.T;
    const int chp = (CT * 32 > ch ? CT * 32 : ch) + 4;     // LDS row pitch: covers the padded channel tiles, +4 floats vs bank conflicts
    float* Qs = sm;                                          // [QW*32][chp]
    float* Ks = Qs + QW * 32 * chp;                          // [32][chp]
    float* Vs = Ks + 32 * chp;                               // [32][chp]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 31, kh = lane >> 5;
    const int b = blockIdx.y / p.heads, h = blockIdx.y % p.heads;
    const int q0 = blockIdx.x * (QW * 32);
    const float* base = p.qkv + (long)b * T * 3 * C + h * ch;
    const int nthr = 64 * QW, c4n = chp / 4;

    // ---- stage Q (rows past T are clamped: their results are never stored); zero the padded columns of all three tiles
    for (int i = tid; i < QW * 32 * c4n; i += nthr) {
        const int r = i / c4n, c = 4 * (i % c4n);
        const int t = min(q0 + r, T - 1);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c < ch) v = *(const float4*)(base + (long)t * 3 * C + c);
        *(float4*)(Qs + r * chp + c) = v;
    }
    const int c4 = ch / 4, npad4 = c4n - c4;
    for (int i = tid; i < 32 * npad4; i += nthr) {
        const int r = i / npad4, c = ch + 4 * (i % npad4);
        *(float4*)(Ks + r * chp + c) = make_float4(0.f, 0.f, 0.f, 0.f);
        *(float4*)(Vs + r * chp + c) = make_float4(0.f, 0.f, 0.f, 0.f);
    }

    f32x16 o[CT];
#pragma unroll
    for (int t = 0; t < CT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;                    // running max (full row) and this half-wave's partial sum

    // K/V staging plan: thread-owned 16-byte chunks of the 32 x ch tile, fetched one key tile AHEAD into registers so
    // that the HBM/L2 latency runs under the previous tile's MFMAs (the LDS tile itself is single-buffered).
    constexpr int NL = (32 * CT * 8 + 64 * QW - 1) / (64 * QW);
    int goff[NL], loff[NL], krow[NL];
#pragma unroll
    for (int j = 0; j < NL; ++j) {
        const int i = tid + j * nthr;
        const bool live = i < 32 * c4;
        const int r = live ? i / c4 : 0, c = live ? 4 * (i % c4) : 0;
        goff[j] = r * 3 * C + C + c;
        loff[j] = r * chp + c;
        krow[j] = live ? r : (1 << 30);                      // dead slots never pass the t < T test
    }
    float4 kr[NL], vr[NL];
    constexpr bool PF = CT <= 4;                             // wider heads have no registers to spare: fetch in place
#define DUA_ATTN_FETCH(KT)                                                                                           \
    {                                                                                                                \
        const long adv = (long)(KT) * 32 * 3 * C;                                                                    \
        _Pragma("unroll") for (int j = 0; j < NL; ++j) {                                                             \
            const bool ok = (KT) * 32 + krow[j] < T;                                                                 \
            kr[j] = make_float4(0.f, 0.f, 0.f, 0.f);                                                                 \
            vr[j] = kr[j];                                                                                           \
            if (ok) {                                                                                                \
                kr[j] = *(const float4*)(base + adv + goff[j]);                                                      \
                vr[j] = *(const float4*)(base + adv + goff[j] + C);                                                  \
            }                                                                                                        \
        }                                                                                                            \
    }
    if (PF) DUA_ATTN_FETCH(0)

    const int nkt = (T + 31) / 32;
    for (int kt = 0; kt < nkt; ++kt) {
        if (!PF) DUA_ATTN_FETCH(kt)
        __syncthreads();                                     // previous tile's readers are done (also orders the Q staging)
#pragma unroll
        for (int j = 0; j < NL; ++j)
            if (krow[j] < 32) { *(float4*)(Ks + loff[j]) = kr[j]; *(float4*)(Vs + loff[j]) = vr[j]; }
        __syncthreads();
        if (PF) DUA_ATTN_FETCH(kt + 1)

        // ---- S^T tile: keys on rows, this wave's 32 queries on columns
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
        const float* ka = Ks + lr * chp + 4 * kh;
        const float* qb = Qs + (wave * 32 + lr) * chp + 4 * kh;
        for (int g = 0; g < ch / 8; ++g) {
            const float4 a = *(const float4*)(ka + 8 * g), q = *(const float4*)(qb + 8 * g);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, q.x, s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, q.y, s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, q.z, s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, q.w, s, 0, 0, 0);
        }
        // ---- online softmax (query = lane; this lane holds keys (r&3) + 8(r>>2) + 4kh of the tile)
        float mloc = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
            s[r] = key < T ? s[r] * p.scale : -INFINITY;
            mloc = fmaxf(mloc, s[r]);
        }
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32));
        const float m_new = fmaxf(m_run, mloc);              // finite: every tile holds at least one valid key
        const float alpha = __expf(m_run - m_new);           // 0 on the first tile (m_run = -inf)
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s[r] = __expf(s[r] - m_new);                     // masked keys: exp(-inf) = 0
            psum += s[r];
        }
        l_run = l_run * alpha + psum;
        m_run = m_new;
#pragma unroll
        for (int t = 0; t < CT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[t][r] *= alpha;
        // ---- O^T += V^T . P^T
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float* vrow = Vs + ((r & 3) + 8 * (r >> 2) + 4 * kh) * chp + lr;
#pragma unroll
            for (int t = 0; t < CT; ++t) o[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(vrow[32 * t], s[r], o[t], 0, 0, 0);
        }
    }

#undef DUA_ATTN_FETCH
    // ---- normalise and store: lane = query, registers = channels (4 consecutive per register quad)
    const float l_full = l_run + __shfl_xor(l_run, 32);
    const float inv = 1.0f / l_full;
    const int q = q0 + wave * 32 + lr;
    if (q < T) {
        float* orow = p.out + ((long)b * T + q) * C + h * ch;
#pragma unroll
        for (int t = 0; t < CT; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c = 32 * t + 8 * g + 4 * kh;
                if (c < ch) *(float4*)(orow + c) = make_float4(o[t][4 * g] * inv, o[t][4 * g + 1] * inv, o[t][4 * g + 2] * inv, o[t][4 * g + 3] * inv);
            }
    }
}

// ================================================================================================================
// Split-bf16 form of the same kernel (precision mode bf16x3 of the models): both products run as
// hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_bf16 (2^-17 per product), 16 channels / 16 keys per instruction instead
// of 2, i.e. 96 instead of 512 matrix-core cycles per 32x32x16 block of the exact-fp32 form.
//   S^T = K . Q^T : A = K tile (LDS, bf16 hi | lo planes, key rows, channel-contiguous), B = this wave's 32 queries, kept in
//                   REGISTERS for the whole kernel (pre-scaled, split once): query on the lane as before.
//   O^T += V^T.P^T: B = the exponentiated accumulator registers, split to bf16 where they stand; a 32x32x16 step sums over
//                   accumulator ROWS 16s + 8(j>>2) + 4h + (j&3) (element j of lane half h), so V is staged TRANSPOSED with its
//                   32 keys stored in exactly that order: the A fragment is one 16-byte LDS read, no lane exchange anywhere.
// Head dims up to 128 (queries in registers); wider heads keep the GEMM path.
typedef __bf16 abf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 abf16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void asplit(float v, __bf16& h, __bf16& l) {
    h = (__bf16)v;
    l = (__bf16)(v - (float)h);
}

// fp16 split of the f16x3 format: hi = fp16(v) saturating at +-65504, lo = fp16(v - hi) (saturating): 22 significant bits
__device__ __forceinline__ void asplit(float v, _Float16& h, _Float16& l) {
    h = (_Float16)fminf(fmaxf(v, -65504.f), 65504.f);
    l = (_Float16)fminf(fmaxf(v - (float)h, -65504.f), 65504.f);
}
typedef _Float16 af16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 af16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x16 amfma(abf16x8 a, abf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x16 amfma(af16x8 a, af16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
template <bool F16> struct AT { typedef __bf16 e; typedef abf16x8 v8; typedef abf16x4 v4; };
template <> struct AT<true> { typedef _Float16 e; typedef af16x8 v8; typedef af16x4 v4; };
// F16 (the f16x3 mode): the same kernel on fp16 hi / lo halves with power-of-two operand scales that keep the lo halves in
// fp16's normal range - q * 2^8 (after the ch^-1/2 scale), k * 2^4 (logits rescaled by 2^-12 before the softmax), P * 2^12
// (P in (0, 1]), v * 2^4 (the 2^-16 is folded into the final 1/l).  Every product is hi*hi + hi*lo + lo*hi: 2^-22 relative.
#define DUA_AX_SQ 256.0f
#define DUA_AX_SK 16.0f
#define DUA_AX_SP 4096.0f
#define DUA_AX_SV 16.0f

// Two ds_read_b64_tr_b16 (gfx950 transposed LDS read; EXEC must be full): 4 + 4 sixteen-bit elements -> one 8-element MFMA operand
template <typename V8, typename E>
__device__ __forceinline__ V8 tr_pair(const E* a, const E* b) {
    typedef short s4_t __attribute__((__vector_size__(4 * sizeof(short))));
    const s4_t x = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4_t*)a);
    const s4_t y = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4_t*)b);
    struct { s4_t lo, hi; } pr{x, y};
    V8 r;
    __builtin_memcpy(&r, &pr, 16);
    return r;
}

// Heads wider than 128 channels (CT16 > 8: ADM-128's 16^2 / 8^2 levels, 192 / 256 channels per head): the query halves (up to 128
// registers) and the output tile (up to 128) leave no room for two waves per SIMD, so the kernel is built for ONE workgroup of four
// waves per CU with the 512-register budget (the K / V tiles then live in dynamic LDS: 55 - 71 KB), and the logit chain runs on two
// accumulators so that a wave alone on its SIMD does not wait for its own previous MFMA.
template <int CT16 /* 16-channel steps: ceil(ch/16) */, int QW, bool F16 = false>
__global__ __launch_bounds__(64 * QW, CT16 > 8 ? 1 : 2) void k_flash_attn_x(AttnP p) {
    typedef typename AT<F16>::e el;
    typedef typename AT<F16>::v8 abf16x8;
    typedef typename AT<F16>::v4 abf16x4;
    constexpr bool WIDE = CT16 > 8;
    constexpr int CHP = CT16 * 16, CT32 = (CHP + 31) / 32;
    constexpr int KP = CHP + 8;          // K tile row pitch (bf16): 16-byte reads of 16 consecutive rows hit distinct banks
    // V tile row-major [key][channel] like K, read TRANSPOSED by ds_read_b64_tr_b16 (a 16-lane group fetches a 4-key x 16-channel block
    // and every lane receives its channel's 4 keys): the staging writes are 8-byte row writes instead of 2-byte column writes, which
    // were 16-way bank-conflicted (80 % of the LDS cycles of this kernel were conflict cycles).  Row pitch = 32 or 96 (mod 128)
    // elements: the 4 key rows of a block then start 16 dwords apart (mod 64) and a 32-lane half covers all 64 banks once.
    constexpr int VPR = CT32 * 32 <= 32 ? 32 : (CT32 * 32 <= 96 ? 96 : (CT32 * 32 <= 160 ? 160 : (CT32 * 32 <= 224 ? 224 : 288)));
    __shared__ __attribute__((aligned(16))) el Kh_s[WIDE ? 8 : 32 * KP], Kl_s[WIDE ? 8 : 32 * KP], Vh_s[WIDE ? 8 : 32 * VPR], Vl_s[WIDE ? 8 : 32 * VPR];
    // Heads of 112 - 128 channels: the lo halves of the wave's queries live in LDS (lane-private 16-byte slots, re-read once per
    // key tile and step) instead of 32 registers - with them the kernel spilled (40 - 300 bytes of scratch per lane)
    // (256-channel heads too: with them in registers the 512-register kernel spilled 232 bytes per lane)
    constexpr bool QL_LDS = (CT16 >= 7 && !WIDE) || CT16 > 12;
    extern __shared__ __attribute__((aligned(16))) unsigned char qlds_raw_[];
    el* const dyn = reinterpret_cast<el*>(qlds_raw_);
    el* const Kh = WIDE ? dyn : Kh_s;
    el* const Kl = WIDE ? dyn + 32 * KP : Kl_s;
    el* const Vh = WIDE ? dyn + 64 * KP : Vh_s;
    el* const Vl = WIDE ? dyn + 64 * KP + 32 * VPR : Vl_s;
    el* const qlds = WIDE ? dyn + 64 * KP + 64 * VPR : dyn;
    const int ch = p.ch, C = p.heads * ch, T = p.T;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 31, kh = lane >> 5;
    const int b = blockIdx.y / p.heads, h = blockIdx.y % p.heads;
    const int q0 = blockIdx.x * (QW * 32);
    const int Tk = p.k ? p.Tk : T;                                       // keys (cross-attention: its own count and tensors)
    const long ldq = p.k ? p.ldq : 3L * C, ldk = p.k ? p.ldkv : 3L * C;
    const float* base = p.qkv + (long)b * T * ldq + h * ch;
    const float* kbase = p.k ? p.k + (long)b * Tk * ldk + h * ch : base + C;
    const float* vbase = p.k ? p.v + (long)b * Tk * ldk + h * ch : base + 2 * C;
    constexpr int nthr = 64 * QW;

    // ---- this lane's query row, scaled and split once: element j of step c = channel 16c + 8kh + j
    abf16x8 qh[CT16], ql[QL_LDS ? 1 : CT16];
    {
        const float* qrow = base + (long)min(q0 + wave * 32 + lr, T - 1) * ldq;
        // The row is requested in groups of four 16-channel steps (eight float4 in flight) from addresses that are ALWAYS valid (a channel
        // beyond the head is clamped and zeroed afterwards): behind `if (c0 + j < ch)` every load was followed by its own vmcnt(0), CT16
        // memory latencies in a row at the start of a workgroup that lives for 9 key tiles at U-ViT's 258 tokens.
        constexpr int QG = 4;
#pragma unroll
        for (int cg = 0; cg < CT16; cg += QG) {
            float4 tq[QG][2];
#pragma unroll
            for (int u = 0; u < QG; ++u)
#pragma unroll
                for (int jh = 0; jh < 2; ++jh) {
                    const int cc = 16 * (cg + u) + 8 * kh + 4 * jh;
                    tq[u][jh] = (cg + u < CT16) ? *(const float4*)(qrow + min(cc, ch - 4)) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
            for (int u = 0; u < QG; ++u) {
                const int c = cg + u;
                if (c >= CT16) continue;
                const int c0 = 16 * c + 8 * kh;
                float v[8];
#pragma unroll
                for (int jh = 0; jh < 2; ++jh) {
                    const bool in = c0 + 4 * jh < ch;                        // ch % 4 == 0: a float4 is all in or all out
                    const float4 t = tq[u][jh];
                    const float qs = F16 ? p.scale * DUA_AX_SQ : p.scale;
                    v[4 * jh] = in ? t.x * qs : 0.f; v[4 * jh + 1] = in ? t.y * qs : 0.f; v[4 * jh + 2] = in ? t.z * qs : 0.f; v[4 * jh + 3] = in ? t.w * qs : 0.f;
                }
                abf16x8 lo_;
#pragma unroll
                for (int j = 0; j < 8; ++j) { el a_, b_; asplit(v[j], a_, b_); qh[c][j] = a_; lo_[j] = b_; }
                if (QL_LDS) *(abf16x8*)(qlds + ((wave * CT16 + c) * 64 + lane) * 8) = lo_;
                else ql[QL_LDS ? 0 : c] = lo_;
            }
        }
    }
    // ---- zero the padded channel columns of the K tile (they meet zero Q channels: must not be NaN patterns) and all of V^T
    for (int i = tid; i < 32 * KP; i += nthr) { Kh[i] = (el)0.f; Kl[i] = (el)0.f; }
    for (int i = tid; i < 32 * VPR; i += nthr) { Vh[i] = (el)0.f; Vl[i] = (el)0.f; }

    f32x16 o[CT32];
#pragma unroll
    for (int t = 0; t < CT32; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    // K/V staging plan: thread-owned float4 chunks of the 32 x ch tile, fetched one key tile ahead into registers
    const int c4 = ch / 4;
    constexpr int NL = (32 * (CHP / 4) + nthr - 1) / nthr;
    int goff[NL], kcol[NL], krow[NL];
#pragma unroll
    for (int j = 0; j < NL; ++j) {
        const int i = tid + j * nthr;
        const bool live = i < 32 * c4;
        const int r = live ? i / c4 : 0, c = live ? 4 * (i % c4) : 0;
        goff[j] = r * (int)ldk + c;
        kcol[j] = c;
        krow[j] = live ? r : (1 << 30);
    }
    float4 kr[NL], vr[NL];
#define DUA_ATTNX_FETCH(KT)                                                                                          \
    {                                                                                                                \
        const long adv = (long)(KT) * 32 * ldk;                                                                      \
        _Pragma("unroll") for (int j = 0; j < NL; ++j) {                                                             \
            const bool ok = (KT) * 32 + krow[j] < Tk;                                                                \
            kr[j] = make_float4(0.f, 0.f, 0.f, 0.f);                                                                 \
            vr[j] = kr[j];                                                                                           \
            if (ok) {                                                                                                \
                kr[j] = *(const float4*)(kbase + adv + goff[j]);                                                     \
                vr[j] = *(const float4*)(vbase + adv + goff[j]);                                                     \
            }                                                                                                        \
        }                                                                                                            \
    }
    DUA_ATTNX_FETCH(0)

    const int nkt = (Tk + 31) / 32;
    for (int kt = 0; kt < nkt; ++kt) {
        __syncthreads();                                     // previous tile's readers are done
#pragma unroll
        for (int j = 0; j < NL; ++j)
            if (krow[j] < 32) {
                const int r = krow[j], c = kcol[j];
                const float kf[4] = {kr[j].x, kr[j].y, kr[j].z, kr[j].w}, vf[4] = {vr[j].x, vr[j].y, vr[j].z, vr[j].w};
                abf16x4 a_, b_;
#pragma unroll
                for (int u = 0; u < 4; ++u) { el x_, y_; asplit(F16 ? kf[u] * DUA_AX_SK : kf[u], x_, y_); a_[u] = x_; b_[u] = y_; }
                *(abf16x4*)(Kh + r * KP + c) = a_;
                *(abf16x4*)(Kl + r * KP + c) = b_;
                abf16x4 c_, d_;
#pragma unroll
                for (int u = 0; u < 4; ++u) { el x_, y_; asplit(F16 ? vf[u] * DUA_AX_SV : vf[u], x_, y_); c_[u] = x_; d_[u] = y_; }
                *(abf16x4*)(Vh + r * VPR + c) = c_;
                *(abf16x4*)(Vl + r * VPR + c) = d_;
            }
        __syncthreads();
        DUA_ATTNX_FETCH(kt + 1)

        // ---- S^T tile: keys on rows, this wave's 32 queries on columns (scale folded into Q)
        f32x16 s, s_b;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; s_b[r] = 0.f; }
#pragma unroll
        for (int c = 0; c < CT16; ++c) {
            const abf16x8 ah = *(const abf16x8*)(Kh + lr * KP + 16 * c + 8 * kh), al = *(const abf16x8*)(Kl + lr * KP + 16 * c + 8 * kh);
            const abf16x8 qlc = QL_LDS ? *(const abf16x8*)(qlds + ((wave * CT16 + c) * 64 + lane) * 8) : ql[QL_LDS ? 0 : c];
            if (WIDE && (c & 1)) {
                s_b = amfma(al, qh[c], s_b);
                s_b = amfma(ah, qlc, s_b);
                s_b = amfma(ah, qh[c], s_b);
            } else {
                s = amfma(al, qh[c], s);
                s = amfma(ah, qlc, s);
                s = amfma(ah, qh[c], s);
            }
        }
        if (WIDE) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] += s_b[r];
        }
        // ---- online softmax (query = lane; this lane holds keys (r&3) + 8(r>>2) + 4kh of the tile)
        float mloc = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
            s[r] = key < Tk ? (F16 ? s[r] * (1.0f / (DUA_AX_SQ * DUA_AX_SK)) : s[r]) : -INFINITY;
            mloc = fmaxf(mloc, s[r]);
        }
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32));
        const float m_new = fmaxf(m_run, mloc);
        const float alpha = __expf(m_run - m_new);
        float psum = 0.f;
        abf16x8 ph[2], pl[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float e = __expf(s[r] - m_new);
            psum += e;
            el x_, y_;
            asplit(F16 ? e * DUA_AX_SP : e, x_, y_);
            ph[r >> 3][r & 7] = x_;
            pl[r >> 3][r & 7] = y_;
        }
        l_run = l_run * alpha + psum;
        m_run = m_new;
#pragma unroll
        for (int t = 0; t < CT32; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[t][r] *= alpha;
        // ---- O^T += V^T . P^T: rows = channels 32t + lr, k = the 16 keys of step s2 in accumulator-row order
#pragma unroll
        for (int t = 0; t < CT32; ++t)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                // A operand of O^T += V^T P^T: lane (lr, kh) = channel 32t + lr, k index j = key 16 s2 + 8 (j >> 2) + 4 kh + (j & 3) of the
                // tile (the order of the accumulator rows that hold P): two transposed 4-key blocks, keys r0 .. r0+3 and r0+8 .. r0+11
                const int vo = (16 * s2 + 4 * kh + ((lr >> 2) & 3)) * VPR + 32 * t + 16 * (lr >> 4) + 4 * (lr & 3);
                const abf16x8 vh = tr_pair<abf16x8>(Vh + vo, Vh + vo + 8 * VPR), vl = tr_pair<abf16x8>(Vl + vo, Vl + vo + 8 * VPR);
                o[t] = amfma(vl, ph[s2], o[t]);
                o[t] = amfma(vh, pl[s2], o[t]);
                o[t] = amfma(vh, ph[s2], o[t]);
            }
    }
#undef DUA_ATTNX_FETCH
    const float l_full = l_run + __shfl_xor(l_run, 32);
    const float inv = (F16 ? 1.0f / (DUA_AX_SP * DUA_AX_SV) : 1.0f) / l_full;
    const int q = q0 + wave * 32 + lr;
    if (q < T) {
        const long ro = ((long)b * T + q) * C + h * ch;
#pragma unroll
        for (int t = 0; t < CT32; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c = 32 * t + 8 * g + 4 * kh;
                if (c < ch) {
                    const float4 w = make_float4(o[t][4 * g] * inv, o[t][4 * g + 1] * inv, o[t][4 * g + 2] * inv, o[t][4 * g + 3] * inv);
                    if (p.out_planes && F16) {           // f16x3 planes: fp16 hi | fp16 lo of out * planes_scale
                        const float f[4] = {w.x * p.planes_scale, w.y * p.planes_scale, w.z * p.planes_scale, w.w * p.planes_scale};
                        af16x4 hh, ll;
#pragma unroll
                        for (int u = 0; u < 4; ++u) { _Float16 x_, y_; if (unsigned long long* rc_ = dua_range_ctr()) dua_range_note(rc_, f[u]); asplit(f[u], x_, y_); hh[u] = x_; ll[u] = y_; }
                        *(af16x4*)((_Float16*)p.out_planes + ro + c) = hh;
                        *(af16x4*)((_Float16*)p.out_planes + p.plane_elems + ro + c) = ll;
                    } else *(float4*)(p.out + ro + c) = w;
                }
            }
    }
}

template <int CT16, int QW, bool F16 = false>
static int launch_attn_x(const AttnP& p, int B, hipStream_t s) {
    dim3 grid((p.T + QW * 32 - 1) / (QW * 32), B * p.heads);
    constexpr int CHP_ = CT16 * 16, CT32_ = (CHP_ + 31) / 32, VPR_ = CT32_ * 32 <= 160 ? 160 : (CT32_ * 32 <= 224 ? 224 : 288);
    // heads of 112 - 128 channels: the lo halves of the queries; wider heads: the K / V tiles (hi and lo)
    constexpr size_t qlds = CT16 > 8 ? (size_t)(64 * (CHP_ + 8) + 64 * VPR_) * 2 + (CT16 > 12 ? (size_t)QW * CT16 * 64 * 16 : 0)
                                     : (CT16 >= 7 ? (size_t)QW * CT16 * 64 * 16 : 0);
    if (qlds) {
        static bool raised = false;            // (per instantiation; static + dynamic LDS exceeds the 64 KB default)
        if (!raised) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_flash_attn_x<CT16, QW, F16>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)qlds);
            if (e != hipSuccess) { dua_set_error("dua_attention (split): cannot reserve %zu bytes of LDS: %s", qlds, hipGetErrorString(e)); return DUA_EHIP; }
            raised = true;
        }
    }
    DuaKTimer kt((void*)s);
    hipLaunchKernelGGL((k_flash_attn_x<CT16, QW, F16>), grid, 64 * QW, qlds, s, p);
    {   // algorithmic: 4 B H Tq Tk ch FLOPs; bytes: q, k, v rows read once + the output (fp32 and / or planes)
        const double Tk = p.Tk > 0 ? (double)p.Tk : (double)p.T, BH = (double)B * p.heads;
        kt.done(F16 ? "k_flash_attn_x (f16x3)" : "k_flash_attn_x (bf16x3)", 4.0 * BH * p.ch * ((double)p.T * 2.0 + 2.0 * Tk), 4.0 * BH * (double)p.T * Tk * p.ch);
    }
    return dua_check_launch("dua_attention_bf16x3");
}

static int attention_x_launch(const AttnP& p, int B, void* stream, bool f16 = false);

/* The fused attention on the fp32-equivalent split (f16x3): fp32 output, or f16x3 planes of out * planes_scale (operand of
 * dua_gemm_f16x3_presplit: proj_out) when out_planes != NULL. */
extern "C" int dua_attention_f16x3(const float* qkv, float* out, void* out_planes, float planes_scale, int B, int T, int heads, int ch,
                                   float scale, void* stream) {
    DUA_REQUIRE(qkv && (out || out_planes) && B > 0 && T > 0 && heads > 0, "dua_attention_f16x3: bad arguments");
    DUA_REQUIRE(ch >= 8 && ch % 4 == 0 && ch <= 256, "dua_attention_f16x3: head dim must be a multiple of 4 in [8, 256] (got %d)", ch);
    DUA_REQUIRE((long)B * heads <= 65535, "dua_attention_f16x3: B*heads must be <= 65535");
    DUA_REQUIRE(!out_planes || planes_scale > 0.f, "dua_attention_f16x3: planes_scale must be positive");
    AttnP p{qkv, out, T, heads, ch, scale, out_planes, (long)B * T * heads * ch, planes_scale};
    return attention_x_launch(p, B, stream, true);
}

/* Cross-attention on the same kernel: queries [B*Tq][ldq], keys / values [B*Tk][ldkv] from their own tensors (Tk: e.g. 77 text tokens). */
extern "C" int dua_cross_attention_f16x3(const float* q, long ldq, const float* k, const float* v, long ldkv, float* out, void* out_planes,
                                         float planes_scale, int B, int Tq, int Tk, int heads, int ch, float scale, void* stream) {
    DUA_REQUIRE(q && k && v && (out || out_planes) && B > 0 && Tq > 0 && Tk > 0 && heads > 0, "dua_cross_attention_f16x3: bad arguments");
    DUA_REQUIRE(ch >= 8 && ch % 4 == 0 && ch <= 256, "dua_cross_attention_f16x3: head dim must be a multiple of 4 in [8, 256] (got %d)", ch);
    DUA_REQUIRE(ldq >= (long)heads * ch && ldkv >= (long)heads * ch && ldq % 4 == 0 && ldkv % 4 == 0 && ldkv * 32 < (1L << 31),
                "dua_cross_attention_f16x3: row strides must cover heads*ch and be multiples of 4");
    DUA_REQUIRE(((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v) & 15) == 0), "dua_cross_attention_f16x3: operands must be 16-byte aligned");
    DUA_REQUIRE((long)B * heads <= 65535, "dua_cross_attention_f16x3: B*heads must be <= 65535");
    DUA_REQUIRE(!out_planes || planes_scale > 0.f, "dua_cross_attention_f16x3: planes_scale must be positive");
    AttnP p{q, out, Tq, heads, ch, scale, out_planes, (long)B * Tq * heads * ch, planes_scale, k, v, ldq, ldkv, Tk};
    return attention_x_launch(p, B, stream, true);
}

extern "C" int dua_attention_bf16x3(const float* qkv, float* out, int B, int T, int heads, int ch, float scale, void* stream) {
    DUA_REQUIRE(qkv && out && B > 0 && T > 0 && heads > 0, "dua_attention_bf16x3: bad arguments");
    DUA_REQUIRE(ch >= 8 && ch % 4 == 0 && ch <= 128, "dua_attention_bf16x3: head dim must be a multiple of 4 in [8, 128] (got %d)", ch);
    DUA_REQUIRE((long)B * heads <= 65535, "dua_attention_bf16x3: B*heads must be <= 65535");
    AttnP p{qkv, out, T, heads, ch, scale, nullptr, 0};
    return attention_x_launch(p, B, stream);
}

static int attention_x_launch(const AttnP& p, int B, void* stream, bool f16) {
    const int T = p.T, ch = p.ch;
    hipStream_t s = as_stream(stream);
    const bool small = T <= 512;
    // short sequences: 64 queries per workgroup, or 96 when that pads fewer (U-ViT-H/2: 258 tokens = 5 x 64 with 62 idle rows, 3 x 96 with 30)
    const bool three = small && (T + 95) / 96 * 96 < (T + 63) / 64 * 64;
    // long sequences in the f16x3 form: 256 queries per workgroup (8 waves) - every workgroup converts the whole K / V of its head, so
    // twice the queries halve that work per query (ADM-128 32^2 x 128: 1.18 -> 1.13 ms, SD-1.5 64^2 x 40: 0.29 -> 0.24 ms)
    const bool big = f16 && T >= 1024;
#define ATTNX_CASE(CTV) \
    case CTV: return f16 ? (big ? launch_attn_x<CTV, 8, true>(p, B, s) : three ? launch_attn_x<CTV, 3, true>(p, B, s) : small ? launch_attn_x<CTV, 2, true>(p, B, s) : launch_attn_x<CTV, 4, true>(p, B, s)) \
                         : (three ? launch_attn_x<CTV, 3>(p, B, s) : small ? launch_attn_x<CTV, 2>(p, B, s) : launch_attn_x<CTV, 4>(p, B, s));
    if (f16 && ch > 128 && ch <= 256)            // wide heads (f16x3 only): 128 queries per workgroup, one workgroup per CU
        return ch <= 192 ? launch_attn_x<12, 4, true>(p, B, s) : launch_attn_x<16, 4, true>(p, B, s);
    switch ((ch + 15) / 16) {
        ATTNX_CASE(1) ATTNX_CASE(2) ATTNX_CASE(3) ATTNX_CASE(4) ATTNX_CASE(5) ATTNX_CASE(6) ATTNX_CASE(7) ATTNX_CASE(8)
    }
    dua_set_error("dua_attention_bf16x3: unsupported head dim %d", ch);
    return DUA_EINVAL;
}

template <int CT, int QW>
static int launch_attn(const AttnP& p, int B, hipStream_t s) {
    const int chp = (CT * 32 > p.ch ? CT * 32 : p.ch) + 4;
    const size_t lds = (size_t)(QW * 32 + 64) * chp * sizeof(float);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_flash_attn<CT, QW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { dua_set_error("dua_attention: cannot reserve %zu bytes of LDS: %s", lds, hipGetErrorString(e)); return DUA_EHIP; }
    }
    dim3 grid((p.T + QW * 32 - 1) / (QW * 32), B * p.heads);
    hipLaunchKernelGGL((k_flash_attn<CT, QW>), grid, 64 * QW, lds, s, p);
    return dua_check_launch("dua_attention");
}

extern "C" int dua_attention(const float* qkv, float* out, int B, int T, int heads, int ch, float scale, void* stream) {
    DUA_REQUIRE(qkv && out && B > 0 && T > 0 && heads > 0, "dua_attention: bad arguments");
    DUA_REQUIRE(ch >= 8 && ch % 8 == 0 && ch <= 256, "dua_attention: head dim must be a multiple of 8 in [8, 256] (got %d)", ch);
    DUA_REQUIRE((long)B * heads <= 65535, "dua_attention: B*heads must be <= 65535");
    AttnP p{qkv, out, T, heads, ch, scale, nullptr, 0};
    hipStream_t s = as_stream(stream);
    const int ct = (ch + 31) / 32;
    const bool small = T <= 512 || ch > 128;             // fewer queries per workgroup: less tail waste / LDS for wide heads
#define ATTN_CASE(CTV)                                                                    \
    case CTV: return small ? launch_attn<CTV, 2>(p, B, s) : launch_attn<CTV, 4>(p, B, s);
    switch (ct) {
        ATTN_CASE(1) ATTN_CASE(2) ATTN_CASE(3) ATTN_CASE(4) ATTN_CASE(5) ATTN_CASE(6) ATTN_CASE(7) ATTN_CASE(8)
    }
    dua_set_error("dua_attention: unsupported head dim %d", ch);
    return DUA_EINVAL;
}

DUA_RANGE_TU(attention)
