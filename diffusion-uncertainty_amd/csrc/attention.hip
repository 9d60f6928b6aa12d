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
    AttnP p{qkv, out, T, heads, ch, scale};
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
