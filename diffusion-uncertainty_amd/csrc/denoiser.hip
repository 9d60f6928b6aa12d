// Memory-bound denoiser kernels: GroupNorm32 (+SiLU, FiLM, dropout mask, 2x resampling), row softmax,
// LayerNorm, timestep / label embeddings.  NHWC fp32, 16 B per lane, wave64 shuffles for reductions.
#include "common.h"

#define BLK 256

// =================================== GroupNorm statistics ======================================
// Pass 1: one workgroup per (image, pixel chunk) streams its contiguous NHWC slab once with float4 loads;
// every thread owns fixed channel quads, so per-channel sums stay in registers.  Per-group partials are
// combined in double and written to ws[N][chunks][G][2].  Pass 2 (k_gn_finalize) folds the chunks.
__global__ __launch_bounds__(BLK) void k_gn_partial(const float* __restrict__ x1, int C1, const float* __restrict__ x2,
                                                    int C2, int HW, int G, int chunk_px, int nchunks,
                                                    double* __restrict__ ws) {
    extern __shared__ float sm[];   // [rowsPerIter][C] sums, then [rowsPerIter][C] sumsq
    const int C = C1 + C2, cpg = C / G;
    const int n = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x;
    const int p0 = chunk * chunk_px, p1 = min(HW, p0 + chunk_px);
    int rows_per_iter = 1;
    for (int s = 0; s < 2; ++s) {
        const float* x = s == 0 ? x1 : x2;
        const int Cs = s == 0 ? C1 : C2, coff = s == 0 ? 0 : C1;
        if (Cs == 0) continue;
        const int Q = Cs / 4;
        const float* base = x + ((size_t)n * HW) * Cs;
        if (Q <= BLK) {
            const int PP = BLK / Q, pl = tid / Q, q = tid % Q;
            float4 su = make_float4(0, 0, 0, 0), sq = make_float4(0, 0, 0, 0);
            if (pl < PP)
                for (int p = p0 + pl; p < p1; p += PP) {
                    float4 v = *(const float4*)(base + (size_t)p * Cs + 4 * q);
                    su.x += v.x; su.y += v.y; su.z += v.z; su.w += v.w;
                    sq.x += v.x * v.x; sq.y += v.y * v.y; sq.z += v.z * v.z; sq.w += v.w * v.w;
                }
            // fold the PP pixel lanes through LDS (deterministic order)
            float* S = sm;              // [PP][Cs]
            float* SS = sm + BLK * 4;   // PP*Cs <= 1024
            if (pl < PP) {
                *(float4*)(S + pl * Cs + 4 * q) = su;
                *(float4*)(SS + pl * Cs + 4 * q) = sq;
            }
            __syncthreads();
            // one thread per group whose channels live in this source
            for (int g = tid; g < G; g += BLK) {
                int c_lo = max(g * cpg, coff) - coff, c_hi = min((g + 1) * cpg, coff + Cs) - coff;
                if (c_lo < c_hi) {
                    double a = 0, b = 0;
                    for (int pp = 0; pp < PP; ++pp)
                        for (int c = c_lo; c < c_hi; ++c) { a += S[pp * Cs + c]; b += SS[pp * Cs + c]; }
                    double* o = ws + (((size_t)n * nchunks + chunk) * G + g) * 2;
                    if (s == 0 || g * cpg >= coff) { o[0] = a; o[1] = b; } else { o[0] += a; o[1] += b; }
                }
            }
            __syncthreads();
        } else {
            // wide sources (Q > 256, at most 512): each thread owns quads tid and tid+256
            float4 su[2], sq[2];
            for (int j = 0; j < 2; ++j) { su[j] = make_float4(0, 0, 0, 0); sq[j] = make_float4(0, 0, 0, 0); }
            for (int p = p0; p < p1; ++p)
                for (int j = 0; j < 2; ++j) {
                    int q = tid + j * BLK;
                    if (q < Q) {
                        float4 v = *(const float4*)(base + (size_t)p * Cs + 4 * q);
                        su[j].x += v.x; su[j].y += v.y; su[j].z += v.z; su[j].w += v.w;
                        sq[j].x += v.x * v.x; sq[j].y += v.y * v.y; sq[j].z += v.z * v.z; sq[j].w += v.w * v.w;
                    }
                }
            float* S = sm;
            float* SS = sm + 2048;
            for (int j = 0; j < 2; ++j) {
                int q = tid + j * BLK;
                if (q < Q) { *(float4*)(S + 4 * q) = su[j]; *(float4*)(SS + 4 * q) = sq[j]; }
            }
            __syncthreads();
            for (int g = tid; g < G; g += BLK) {
                int c_lo = max(g * cpg, coff) - coff, c_hi = min((g + 1) * cpg, coff + Cs) - coff;
                if (c_lo < c_hi) {
                    double a = 0, b = 0;
                    for (int c = c_lo; c < c_hi; ++c) { a += S[c]; b += SS[c]; }
                    double* o = ws + (((size_t)n * nchunks + chunk) * G + g) * 2;
                    if (s == 0 || g * cpg >= coff) { o[0] = a; o[1] = b; } else { o[0] += a; o[1] += b; }
                }
            }
            __syncthreads();
        }
    }
    (void)rows_per_iter;
}

// one wave per (image, group): the lanes stride over the chunks and fold their partial sums with a butterfly, in double
__global__ __launch_bounds__(256) void k_gn_finalize(const double* __restrict__ ws, int NG, int G, int nchunks, double count, float eps,
                                                      float* __restrict__ stats) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= NG) return;
    const int n = i / G, g = i % G;
    double a = 0, b = 0;
    for (int c = lane; c < nchunks; c += 64) {
        const double* o = ws + (((size_t)n * nchunks + c) * G + g) * 2;
        a += o[0];
        b += o[1];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        a += __shfl_xor(a, off);
        b += __shfl_xor(b, off);
    }
    if (lane == 0) {
        double mean = a / count, var = b / count - mean * mean;
        if (var < 0) var = 0;
        stats[2 * i] = (float)mean;
        stats[2 * i + 1] = (float)(1.0 / sqrt(var + (double)eps));
    }
}

static void gn_chunking(int N, int HW, int* chunk_px, int* nchunks) {
    // aim for >= ~2048 workgroups, at least 16 pixels per chunk
    int want = (2048 + N - 1) / N;
    int cp = (HW + want - 1) / want;
    // small maps (SD's CFG batch of 2, row views of a few hundred tokens): a 16-pixel chunk is 16 dependent iterations of one workgroup
    // (11 - 20 us for under a megabyte of data); 4 - 8 pixel chunks put the same work on more workgroups
    const int floor_px = (long)N * HW > 65536 ? 16 : (HW >= 4096 ? 8 : 4);
    if (cp < floor_px) cp = floor_px;
    if (cp > HW) cp = HW;
    *chunk_px = cp;
    *nchunks = (HW + cp - 1) / cp;
}

extern "C" size_t dua_groupnorm_ws_floats(int N, int HW, int C, int G) {
    int cp, nc;
    gn_chunking(N, HW, &cp, &nc);
    (void)C;
    return (size_t)N * nc * G * 2 * 2 + 4;   // doubles, expressed in floats
}

extern "C" int dua_groupnorm_stats(const float* x1, int C1, const float* x2, int C2, int N, int HW, int G, float eps,
                                   float* ws, float* stats, void* stream) {
    int C = C1 + C2;
    DUA_REQUIRE(x1 && ws && stats && N > 0 && HW > 0 && G > 0, "dua_groupnorm_stats: bad arguments");
    DUA_REQUIRE(C % G == 0 && C1 % 4 == 0 && C2 % 4 == 0 && C1 <= 2048 && C2 <= 2048, "dua_groupnorm_stats: unsupported channels C1=%d C2=%d G=%d", C1, C2, G);
    DUA_REQUIRE((C2 == 0) == (x2 == nullptr), "dua_groupnorm_stats: x2/C2 mismatch");
    DUA_REQUIRE(((uintptr_t)ws & 7) == 0, "dua_groupnorm_stats: ws must be 8-byte aligned");
    int cp, nc;
    gn_chunking(N, HW, &cp, &nc);
    DuaKTimer kt(stream);
    hipLaunchKernelGGL(k_gn_partial, dim3(nc, N), BLK, 4096 * sizeof(float), as_stream(stream), x1, C1, x2, C2, HW, G, cp, nc, (double*)ws);
    int NG = N * G;
    hipLaunchKernelGGL(k_gn_finalize, (NG + 3) / 4, 256, 0, as_stream(stream), (const double*)ws, NG, G, nc,
                       (double)HW * (double)(C / G), eps, stats);
    kt.done("k_gn_partial + k_gn_finalize", 4.0 * N * HW * C);             // the activation read once
    return dua_check_launch("dua_groupnorm_stats");
}

// GroupNorm statistics from the per-64-row column sums a producing conv / GEMM epilogue left behind (`colsum`,
// [ceil(M/64)][2][C]): one workgroup per (group, image) folds HW/64 blocks x C/G channels in double.  The activation
// itself is not read again.  ws2/C2: second half of a virtual channel concat (may be NULL / 0).
__global__ __launch_bounds__(BLK) void k_gn_from_colsums(const float* __restrict__ ws1, int C1, const float* __restrict__ ws2, int C2,
                                                         int HW, int G, float eps, float* __restrict__ stats) {
    const int g = blockIdx.x, n = blockIdx.y, tid = threadIdx.x;
    const int C = C1 + C2, cpg = C / G, nb = HW >> 6;
    double a = 0, b = 0;
    for (int i = tid; i < nb * cpg; i += BLK) {
        const int blk = i / cpg, c = g * cpg + i % cpg;
        const long row = (long)n * nb + blk;
        if (c < C1) { a += ws1[row * 2 * C1 + c]; b += ws1[row * 2 * C1 + C1 + c]; }
        else { a += ws2[row * 2 * C2 + (c - C1)]; b += ws2[row * 2 * C2 + C2 + (c - C1)]; }
    }
    __shared__ double sa[BLK], sb[BLK];
    sa[tid] = a;
    sb[tid] = b;
    __syncthreads();
    for (int o = BLK / 2; o > 0; o >>= 1) {
        if (tid < o) { sa[tid] += sa[tid + o]; sb[tid] += sb[tid + o]; }
        __syncthreads();
    }
    if (tid == 0) {
        const double cnt = (double)HW * cpg, mean = sa[0] / cnt;
        double var = sb[0] / cnt - mean * mean;
        if (var < 0) var = 0;
        stats[2 * ((size_t)n * G + g)] = (float)mean;
        stats[2 * ((size_t)n * G + g) + 1] = (float)(1.0 / sqrt(var + (double)eps));
    }
}
// ... of x + t for a per-(sample, channel) bias t: per 64-row block and channel, sum(x + t) = S1 + 64 t, sum((x + t)^2) = S2 + 2 t S1 + 64 t^2
__global__ __launch_bounds__(BLK) void k_gn_from_colsums_bias(const float* __restrict__ ws, int C, const float* __restrict__ tb, long tld,
                                                              int HW, int G, float eps, float* __restrict__ stats) {
    const int g = blockIdx.x, n = blockIdx.y, tid = threadIdx.x;
    const int cpg = C / G, nb = HW >> 6;
    double a = 0, b = 0;
    for (int i = tid; i < nb * cpg; i += BLK) {
        const int blk = i / cpg, c = g * cpg + i % cpg;
        const long row = (long)n * nb + blk;
        const double s1 = ws[row * 2 * C + c], s2 = ws[row * 2 * C + C + c], t = tb[(size_t)n * tld + c];
        a += s1 + 64.0 * t;
        b += s2 + 2.0 * t * s1 + 64.0 * t * t;
    }
    __shared__ double sa[BLK], sb[BLK];
    sa[tid] = a;
    sb[tid] = b;
    __syncthreads();
    for (int o = BLK / 2; o > 0; o >>= 1) {
        if (tid < o) { sa[tid] += sa[tid + o]; sb[tid] += sb[tid + o]; }
        __syncthreads();
    }
    if (tid == 0) {
        const double cnt = (double)HW * cpg, mean = sa[0] / cnt;
        double var = sb[0] / cnt - mean * mean;
        if (var < 0) var = 0;
        stats[2 * ((size_t)n * G + g)] = (float)mean;
        stats[2 * ((size_t)n * G + g) + 1] = (float)(1.0 / sqrt(var + (double)eps));
    }
}
extern "C" int dua_groupnorm_stats_from_colsums_bias(const float* ws, int C, int N, int HW, int G, float eps, const float* tbias, long tbias_ld,
                                                     float* stats, void* stream) {
    DUA_REQUIRE(ws && tbias && stats && N > 0 && G > 0 && HW > 0 && HW % 64 == 0 && C % G == 0 && tbias_ld >= C,
                "dua_groupnorm_stats_from_colsums_bias: HW must be a positive multiple of 64, C a multiple of G");
    hipLaunchKernelGGL(k_gn_from_colsums_bias, dim3(G, N), BLK, 0, as_stream(stream), ws, C, tbias, tbias_ld, HW, G, eps, stats);
    return dua_check_launch("dua_groupnorm_stats_from_colsums_bias");
}
extern "C" int dua_groupnorm_stats_from_colsums(const float* ws1, int C1, const float* ws2, int C2, int N, int HW, int G, float eps,
                                                float* stats, void* stream) {
    DUA_REQUIRE(ws1 && stats && N > 0 && G > 0 && HW > 0 && HW % 64 == 0, "dua_groupnorm_stats_from_colsums: HW must be a positive multiple of 64 (got %d)", HW);
    DUA_REQUIRE((C1 + C2) % G == 0 && (C2 == 0) == (ws2 == nullptr), "dua_groupnorm_stats_from_colsums: bad channel split");
    hipLaunchKernelGGL(k_gn_from_colsums, dim3(G, N), BLK, 0, as_stream(stream), ws1, C1, ws2, C2, HW, G, eps, stats);
    return dua_check_launch("dua_groupnorm_stats_from_colsums");
}

// =================================== GroupNorm apply ===========================================
// One float4 of the (virtually concatenated) input per lane-iteration.  resample: 0 none, +1 nearest x2
// (each value written 4x), -1 2x2 average of the ACTIVATED values (unet_openai.py:238-243: h_upd after in_rest).
struct GnApplyP {
    const float *x1, *x2, *stats, *gamma, *beta, *film, *drop;
    float* y;
    int C1, C2, N, H, W, G, act, resample;
    int split;               // 1: y is two bf16 planes (hi | lo) of the output shape, for the bf16x3 conv; 4: f16x3 planes (below)
    long plane;              // elements per plane
    long film_ld;            // row stride of film in floats (>= 2*C)
    float oscale;            // split == 4 (f16x3): planes hold fp16 hi | fp16 lo of y * oscale (a power of two)
    int film_pre;            // 1: `film` is a per-(sample, channel) bias added BEFORE the normalisation (rows of film_ld floats):
                             //    y = act(((x + t) - mean) * rstd * gamma + beta), the statistics being those of x + t
};

typedef __bf16 gn_bf16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void gn_store(const GnApplyP& p, size_t off, float4 v) {
    if (!p.split) {
        *(float4*)(p.y + off) = v;
    } else if (p.split == 4) {
        // f16x3 planes: hi = fp16(s), lo = fp16(s - hi) of s = y * oscale (saturated to the fp16 range): 22 significant bits
        typedef _Float16 gn_f16x4 __attribute__((ext_vector_type(4)));
        const float f[4] = {v.x, v.y, v.z, v.w};
        gn_f16x4 h, l;
        if (unsigned long long* rc_ = dua_range_ctr()) {
#pragma unroll
            for (int j = 0; j < 4; ++j) dua_range_note(rc_, f[j] * p.oscale);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float sv = fminf(fmaxf(f[j] * p.oscale, -65504.f), 65504.f);
            const _Float16 t = (_Float16)sv;
            h[j] = t;
            l[j] = (_Float16)(sv - (float)t);
        }
        _Float16* hi = (_Float16*)p.y + off;
        *(gn_f16x4*)hi = h;
        *(gn_f16x4*)(hi + p.plane) = l;
    } else {
        __bf16* hi = (__bf16*)p.y + off;
        float f[4] = {v.x, v.y, v.z, v.w};
        gn_bf16x4 h, l;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            __bf16 t = (__bf16)f[j];
            h[j] = t;
            l[j] = (__bf16)(f[j] - (float)t);
        }
        *(gn_bf16x4*)hi = h;
        *(gn_bf16x4*)(hi + p.plane) = l;
    }
}

// One workgroup per (image, chunk of iteration-space pixels).  Every thread owns fixed channel quads, so the
// per-channel affine  y = x*a + b  (a = rstd*gamma*(1+scale), b = (beta - mean*rstd*gamma)*(1+scale) + shift) is
// folded once into registers and the pixel loop is load -> fma -> SiLU -> store, 16 B per lane, with 32-bit index
// arithmetic only.  NQ = channel quads per thread (1 when C/4 <= 256, else 2).
template <int NQ>
__global__ __launch_bounds__(BLK) void k_gn_apply(GnApplyP p, int chunk_px) {
    const int C = p.C1 + p.C2, Q = C / 4, cpg = C / p.G;
    const int n = blockIdx.y, tid = threadIdx.x;
    const int Ho = p.resample < 0 ? p.H / 2 : p.H, Wo = p.resample < 0 ? p.W / 2 : p.W;   // iteration space
    const int npx = Ho * Wo;
    const int p0 = blockIdx.x * chunk_px, p1 = min(npx, p0 + chunk_px);
    const int PP = NQ == 1 ? BLK / Q : 1;
    const int pl = NQ == 1 ? tid / Q : 0;
    if (NQ == 1 && pl >= PP) return;

    float ca[NQ][4], cb[NQ][4];
    const float* src[NQ];
    int cs[NQ], cq[NQ], cc[NQ];
    bool live[NQ];
#pragma unroll
    for (int u = 0; u < NQ; ++u) {
        const int q = NQ == 1 ? tid % Q : tid + u * BLK;
        live[u] = q < Q;
        const int c = 4 * (live[u] ? q : 0);
        cc[u] = c;
        const bool second = c >= p.C1;
        src[u] = second ? p.x2 : p.x1;
        cs[u] = second ? p.C2 : p.C1;
        cq[u] = second ? c - p.C1 : c;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int ch = c + j;
            const float2 st = *(const float2*)(p.stats + 2 * ((size_t)n * p.G + ch / cpg));
            float a = st.y * p.gamma[ch];
            float b = p.beta[ch] - st.x * a;
            if (p.film && p.film_pre) {
                b += p.film[(size_t)n * p.film_ld + ch] * a;
            } else if (p.film) {
                const float sc = 1.0f + p.film[(size_t)n * p.film_ld + ch];
                a *= sc;
                b = b * sc + p.film[(size_t)n * p.film_ld + C + ch];
            }
            ca[u][j] = a;
            cb[u][j] = b;
        }
    }
    auto xf = [&](int u, size_t pix) -> float4 {
        const float4 v = *(const float4*)(src[u] + pix * cs[u] + cq[u]);
        float r[4] = {v.x * ca[u][0] + cb[u][0], v.y * ca[u][1] + cb[u][1], v.z * ca[u][2] + cb[u][2], v.w * ca[u][3] + cb[u][3]};
        if (p.act == 1) {
#pragma unroll
            for (int j = 0; j < 4; ++j) r[j] = silu_f(r[j]);
        }
        if (p.drop) {
            const float4 d = *(const float4*)(p.drop + pix * C + cc[u]);
            r[0] *= d.x; r[1] *= d.y; r[2] *= d.z; r[3] *= d.w;
        }
        return make_float4(r[0], r[1], r[2], r[3]);
    };
    const size_t img_in = (size_t)n * p.H * p.W;
    for (int px = p0 + pl; px < p1; px += PP) {
#pragma unroll
        for (int u = 0; u < NQ; ++u) {
            if (!live[u]) continue;
            if (p.resample == 0) {
                const size_t pix = img_in + px;
                gn_store(p, pix * C + cc[u], xf(u, pix));
            } else {
                const int yo = px / Wo, xo = px - yo * Wo;
                if (p.resample > 0) {
                    const float4 v = xf(u, img_in + px);
                    const int W2 = 2 * p.W;
                    const size_t o = (((size_t)n * 2 * p.H + 2 * yo) * W2 + 2 * xo) * C + cc[u];
                    gn_store(p, o, v);
                    gn_store(p, o + C, v);
                    gn_store(p, o + (size_t)W2 * C, v);
                    gn_store(p, o + (size_t)W2 * C + C, v);
                } else {
                    const size_t pix = img_in + (size_t)(2 * yo) * p.W + 2 * xo;
                    const float4 a = xf(u, pix), b = xf(u, pix + 1), d = xf(u, pix + p.W), e = xf(u, pix + p.W + 1);
                    gn_store(p, (((size_t)n * Ho + yo) * Wo + xo) * C + cc[u],
                             make_float4((a.x + b.x + d.x + e.x) * 0.25f, (a.y + b.y + d.y + e.y) * 0.25f,
                                         (a.z + b.z + d.z + e.z) * 0.25f, (a.w + b.w + d.w + e.w) * 0.25f));
                }
            }
        }
    }
}

// ---- 8 channels per lane: every plane is written 16 B per lane (two float4 for the fp32 output).  Used when both
// sources have a multiple of 8 channels.
typedef __bf16 gn_bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 gn_f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void gn_store8(const GnApplyP& p, size_t off, const float (&f)[8], int lane_odd) {
    if (!p.split) {
        *(float4*)(p.y + off) = make_float4(f[0], f[1], f[2], f[3]);
        *(float4*)(p.y + off + 4) = make_float4(f[4], f[5], f[6], f[7]);
    } else if (p.split == 4) {
        gn_f16x8 h, l;
        if (unsigned long long* rc_ = dua_range_ctr()) {
#pragma unroll
            for (int j = 0; j < 8; ++j) dua_range_note(rc_, f[j] * p.oscale);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float sv = fminf(fmaxf(f[j] * p.oscale, -65504.f), 65504.f);
            const _Float16 t = (_Float16)sv;
            h[j] = t;
            l[j] = (_Float16)(sv - (float)t);
        }
        _Float16* hi = (_Float16*)p.y + off;
        *(gn_f16x8*)hi = h;
        *(gn_f16x8*)(hi + p.plane) = l;
    } else if (p.split == 1) {
        gn_bf16x8 h, l;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const __bf16 t = (__bf16)f[j];
            h[j] = t;
            l[j] = (__bf16)(f[j] - (float)t);
        }
        __bf16* hi = (__bf16*)p.y + off;
        *(gn_bf16x8*)hi = h;
        *(gn_bf16x8*)(hi + p.plane) = l;
    }
}

__global__ __launch_bounds__(BLK) void k_gn_apply8(GnApplyP p, int chunk_px) {
    const int C = p.C1 + p.C2, O = C / 8, cpg = C / p.G;
    const int n = blockIdx.y, tid = threadIdx.x;
    const int Ho = p.resample < 0 ? p.H / 2 : p.H, Wo = p.resample < 0 ? p.W / 2 : p.W;   // iteration space
    const int npx = Ho * Wo;
    const int p0 = blockIdx.x * chunk_px, p1 = min(npx, p0 + chunk_px);
    const int PP = BLK / O, pl = tid / O;
    if (pl >= PP) return;                          // (O even: a lane pair is never split by this exit)
    const int c = 8 * (tid % O);
    const bool second = c >= p.C1;
    const float* src = second ? p.x2 : p.x1;
    const int cs = second ? p.C2 : p.C1, cq = second ? c - p.C1 : c;
    float ca[8], cb[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int ch = c + j;
        const float2 st = *(const float2*)(p.stats + 2 * ((size_t)n * p.G + ch / cpg));
        float a = st.y * p.gamma[ch];
        float b = p.beta[ch] - st.x * a;
        if (p.film && p.film_pre) {
            b += p.film[(size_t)n * p.film_ld + ch] * a;
        } else if (p.film) {
            const float sc = 1.0f + p.film[(size_t)n * p.film_ld + ch];
            a *= sc;
            b = b * sc + p.film[(size_t)n * p.film_ld + C + ch];
        }
        ca[j] = a;
        cb[j] = b;
    }
    auto xf = [&](size_t pix, float (&r)[8]) {
        const float4 v0 = *(const float4*)(src + pix * cs + cq), v1 = *(const float4*)(src + pix * cs + cq + 4);
        const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            r[j] = v[j] * ca[j] + cb[j];
            if (p.act == 1) r[j] = silu_f(r[j]);
        }
        if (p.drop) {
            const float4 d0 = *(const float4*)(p.drop + pix * C + c), d1 = *(const float4*)(p.drop + pix * C + c + 4);
            r[0] *= d0.x; r[1] *= d0.y; r[2] *= d0.z; r[3] *= d0.w; r[4] *= d1.x; r[5] *= d1.y; r[6] *= d1.z; r[7] *= d1.w;
        }
    };
    const int odd = tid & 1;
    const size_t img_in = (size_t)n * p.H * p.W;
    // every lane of a pair runs the same number of iterations (same pl), so the pair exchange in gn_store8 is safe
    for (int px = p0 + pl; px < p1; px += PP) {
        float r[8];
        if (p.resample == 0) {
            const size_t pix = img_in + px;
            xf(pix, r);
            gn_store8(p, pix * C + c, r, odd);
        } else {
            const int yo = px / Wo, xo = px - yo * Wo;
            if (p.resample > 0) {
                xf(img_in + px, r);
                const int W2 = 2 * p.W;
                const size_t o = (((size_t)n * 2 * p.H + 2 * yo) * W2 + 2 * xo) * C + c;
                gn_store8(p, o, r, odd);
                gn_store8(p, o + C, r, odd);
                gn_store8(p, o + (size_t)W2 * C, r, odd);
                gn_store8(p, o + (size_t)W2 * C + C, r, odd);
            } else {
                const size_t pix = img_in + (size_t)(2 * yo) * p.W + 2 * xo;
                float a[8], b[8], d[8], e[8];
                xf(pix, a); xf(pix + 1, b); xf(pix + p.W, d); xf(pix + p.W + 1, e);
#pragma unroll
                for (int j = 0; j < 8; ++j) r[j] = (a[j] + b[j] + d[j] + e[j]) * 0.25f;
                gn_store8(p, (((size_t)n * Ho + yo) * Wo + xo) * C + c, r, odd);
            }
        }
    }
}

static int gn_apply_wide = 1;          // 0: always the 4-channel-per-lane kernel (A/B switch)
extern "C" int dua_set_gn_apply_wide(int on) { int o = gn_apply_wide; gn_apply_wide = on; return o; }
static int gn_apply_impl(const float* x1, int C1, const float* x2, int C2, int N, int H, int W, int G,
                         const float* stats, const float* gamma, const float* beta, const float* film, long film_ld, int act,
                         int resample, const float* dropmask, float* y, int split, void* stream, float oscale = 1.0f, int film_pre = 0);

extern "C" int dua_groupnorm_apply(const float* x1, int C1, const float* x2, int C2, int N, int H, int W, int G,
                                   const float* stats, const float* gamma, const float* beta, const float* film, long film_ld, int act,
                                   int resample, const float* dropmask, float* y, void* stream) {
    return gn_apply_impl(x1, C1, x2, C2, N, H, W, G, stats, gamma, beta, film, film_ld, act, resample, dropmask, y, 0, stream);
}

extern "C" int dua_groupnorm_apply_split(const float* x1, int C1, const float* x2, int C2, int N, int H, int W, int G,
                                         const float* stats, const float* gamma, const float* beta, const float* film, long film_ld, int act,
                                         int resample, const float* dropmask, void* yplanes, void* stream) {
    return gn_apply_impl(x1, C1, x2, C2, N, H, W, G, stats, gamma, beta, film, film_ld, act, resample, dropmask, (float*)yplanes, 1, stream);
}



// "f16x3" planes (fp16 hi | fp16 lo of y * act_scale, act_scale a power of two): operand of dua_conv3x3_f16x3_presplit /
// dua_gemm_f16x3_presplit, which undo the scale in their epilogue.
extern "C" int dua_groupnorm_apply_split_f16x3(const float* x1, int C1, const float* x2, int C2, int N, int H, int W, int G,
                                               const float* stats, const float* gamma, const float* beta, const float* film, long film_ld, int act,
                                               int resample, const float* dropmask, float act_scale, void* yplanes, void* stream) {
    DUA_REQUIRE(act_scale > 0.f && (__builtin_bit_cast(unsigned, act_scale) & 0x7FFFFFu) == 0, "dua_groupnorm_apply_split_f16x3: act_scale must be a power of two");
    return gn_apply_impl(x1, C1, x2, C2, N, H, W, G, stats, gamma, beta, film, film_ld, act, resample, dropmask, (float*)yplanes, 4, stream, act_scale);
}

// ... of x + t, t a per-(sample, channel) bias [N][tbias_ld] (diffusers ResnetBlock2D: hidden + time_emb_proj(..)[:, :, None, None] in front
// of norm2): the add never makes its own pass over the tensor.  Statistics: dua_groupnorm_stats_from_colsums_bias.
extern "C" int dua_groupnorm_apply_prebias_split_f16x3(const float* x, int C, int N, int H, int W, int G, const float* stats, const float* gamma,
                                                       const float* beta, const float* tbias, long tbias_ld, int act, float act_scale,
                                                       void* yplanes, void* stream) {
    DUA_REQUIRE(tbias && tbias_ld >= C, "dua_groupnorm_apply_prebias_split_f16x3: tbias rows of at least C floats");
    DUA_REQUIRE(act_scale > 0.f && (__builtin_bit_cast(unsigned, act_scale) & 0x7FFFFFu) == 0, "dua_groupnorm_apply_prebias_split_f16x3: act_scale must be a power of two");
    return gn_apply_impl(x, C, nullptr, 0, N, H, W, G, stats, gamma, beta, tbias, tbias_ld, act, 0, nullptr, (float*)yplanes, 4, stream, act_scale, 1);
}

static int gn_apply_impl(const float* x1, int C1, const float* x2, int C2, int N, int H, int W, int G,
                         const float* stats, const float* gamma, const float* beta, const float* film, long film_ld, int act,
                         int resample, const float* dropmask, float* y, int split, void* stream, float oscale, int film_pre) {
    int C = C1 + C2;
    DUA_REQUIRE(x1 && stats && gamma && beta && y, "dua_groupnorm_apply: null pointer");
    DUA_REQUIRE(C % G == 0 && C1 % 4 == 0 && C2 % 4 == 0, "dua_groupnorm_apply: unsupported channels");
    DUA_REQUIRE((C2 == 0) == (x2 == nullptr), "dua_groupnorm_apply: x2/C2 mismatch");
    DUA_REQUIRE(resample >= 0 || (H % 2 == 0 && W % 2 == 0), "dua_groupnorm_apply: pooling needs even H, W");
    DUA_REQUIRE(!(dropmask && resample != 0), "dua_groupnorm_apply: dropout mask with resampling unsupported");
    GnApplyP p{x1, x2, stats, gamma, beta, film, dropmask, y, C1, C2, N, H, W, G, act, resample, split, 0, film_ld ? film_ld : 2L * (C1 + C2), oscale,
               film_pre};
    DUA_REQUIRE(p.film_ld >= (film_pre ? 1L : 2L) * (C1 + C2), "dua_groupnorm_apply: film_ld too small");
    {
        long Hy = resample < 0 ? H / 2 : (resample > 0 ? 2 * H : H), Wy = resample < 0 ? W / 2 : (resample > 0 ? 2 * W : W);
        p.plane = (long)N * Hy * Wy * C;
    }
    DUA_REQUIRE(C <= 3072, "dua_groupnorm_apply: at most 3072 channels");
    int Ho = resample < 0 ? H / 2 : H, Wo = resample < 0 ? W / 2 : W;
    if ((size_t)N * Ho * Wo == 0) return DUA_OK;
    int cp, nc;
    gn_chunking(N, Ho * Wo, &cp, &nc);
    // the 16-byte-store kernel wins on the large maps (+11 % at 128^2, +5 % at 64^2) and loses on the small ones, where its
    // fewer, fatter lanes leave part of a workgroup idle
    const bool wide = C1 % 8 == 0 && C2 % 8 == 0 && (C / 8) % 2 == 0 && C / 8 <= BLK && 
                      (long)Ho * Wo >= 4096 && BLK % (C / 8) == 0;
    DuaKTimer kt(stream);
    if (wide && gn_apply_wide) hipLaunchKernelGGL(k_gn_apply8, dim3(nc, N), BLK, 0, as_stream(stream), p, cp);
    else if (C / 4 <= BLK) hipLaunchKernelGGL(k_gn_apply<1>, dim3(nc, N), BLK, 0, as_stream(stream), p, cp);
    else if (C / 4 <= 2 * BLK) hipLaunchKernelGGL(k_gn_apply<2>, dim3(nc, N), BLK, 0, as_stream(stream), p, cp);
    else hipLaunchKernelGGL(k_gn_apply<3>, dim3(nc, N), BLK, 0, as_stream(stream), p, cp);    // SD-1.5 up blocks: 1280 + 1280
    // algorithmic bytes: the fp32 input once + 4 B per output element (fp32, or hi + lo halves of a split format)
    kt.done(split ? "k_gn_apply (operand planes)" : "k_gn_apply (fp32)", 4.0 * N * H * W * C + 4.0 * (double)p.plane);
    return dua_check_launch("dua_groupnorm_apply");
}

// =================================== 3x3 conv with a handful of output channels ==================
// conv3x3(x)[p][co] = sum over the 9 taps of T[p + offset(tap)][tap * cout + co], where T = x W_t^T is a plain one-tap GEMM over the
// pixels with the 9 * cout tap columns as its N (the UNet's `out` conv, 256 -> 3 kept channels: the implicit-GEMM conv kernels would
// compute a 128-column tile for it).  This kernel is the second half: it gathers the nine tap columns of the neighbour pixels (zero
// outside the image), adds the bias and writes NCHW.  T rows of ldt floats; HBM-light (T is M x 28 floats for cout = 3).
__global__ __launch_bounds__(BLK) void k_conv3x3_tap_gather(const float* __restrict__ T, int ldt, const float* __restrict__ bias,
                                                            float* __restrict__ out, int N, int H, int W, int cout) {
    const size_t total = (size_t)N * cout * H * W;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % W), y = (int)((i / W) % H), co = (int)((i / ((size_t)W * H)) % cout), n = (int)(i / ((size_t)W * H * cout));
        float acc = bias ? bias[co] : 0.f;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
            if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) acc += T[(((size_t)n * H + yy) * W + xx) * ldt + t * cout + co];
        }
        out[i] = acc;
    }
}
extern "C" int dua_conv3x3_tap_gather(const float* T, int ldt, const float* bias, float* out_nchw, int N, int H, int W, int cout, void* stream) {
    DUA_REQUIRE(T && out_nchw && N > 0 && H > 0 && W > 0 && cout > 0 && ldt >= 9 * cout, "dua_conv3x3_tap_gather: T rows must hold 9 * cout columns");
    hipLaunchKernelGGL(k_conv3x3_tap_gather, stream_grid((size_t)N * cout * H * W, BLK), BLK, 0, as_stream(stream), T, ldt, bias, out_nchw, N, H, W, cout);
    return dua_check_launch("dua_conv3x3_tap_gather");
}

// =================================== latent-UNet (SD-1.5) helpers ===============================
// x[n][p][c] += bias[n][c]  (ResnetBlock2D: hidden + time_emb_proj(silu(emb))[:, :, None, None]), in place
__global__ __launch_bounds__(BLK) void k_add_channel_bias(float* __restrict__ x, const float* __restrict__ b, long bld, int N, int HW, int C) {
    const int Q = C / 4;
    size_t total = (size_t)N * HW * Q, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int q = (int)(i % Q);
        const int n = (int)(i / ((size_t)HW * Q));
        float4 v = ((float4*)x)[i];
        const float4 bb = *(const float4*)(b + (size_t)n * bld + 4 * q);
        v.x += bb.x; v.y += bb.y; v.z += bb.z; v.w += bb.w;
        ((float4*)x)[i] = v;
    }
}
extern "C" int dua_add_channel_bias(float* x, const float* bias, long bias_ld, int N, int HW, int C, void* stream) {
    DUA_REQUIRE(x && bias && N > 0 && HW > 0 && C > 0 && C % 4 == 0 && bias_ld % 4 == 0 && bias_ld >= C, "dua_add_channel_bias: bad arguments");
    hipLaunchKernelGGL(k_add_channel_bias, stream_grid((size_t)N * HW * (C / 4), BLK), BLK, 0, as_stream(stream), x, bias, bias_ld, N, HW, C);
    return dua_check_launch("dua_add_channel_bias");
}

// GEGLU (diffusers activations.GEGLU): y[m][f] = x[m][f] * gelu(x[m][F + f]), exact erf GELU
__global__ __launch_bounds__(BLK) void k_geglu(const float* __restrict__ x, float* __restrict__ y, long M, int F) {
    const int Q = F / 4;
    size_t total = (size_t)M * Q, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const size_t m = i / Q;
        const int q = (int)(i % Q);
        const float4 a = *(const float4*)(x + m * 2 * F + 4 * q), g = *(const float4*)(x + m * 2 * F + F + 4 * q);
        auto ge = [](float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752f)); };
        ((float4*)y)[i] = make_float4(a.x * ge(g.x), a.y * ge(g.y), a.z * ge(g.z), a.w * ge(g.w));
    }
}
// ... with the result written as f16x3 operand planes (fp16 hi | lo of y * scale) for the presplit GEMM that consumes it
__global__ __launch_bounds__(BLK) void k_geglu_split(const float* __restrict__ x, void* __restrict__ planes, long M, int F, float scale) {
    const int Q = F / 4;
    size_t total = (size_t)M * Q, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const size_t m = i / Q;
        const int q = (int)(i % Q);
        const float4 a = *(const float4*)(x + m * 2 * F + 4 * q), g = *(const float4*)(x + m * 2 * F + F + 4 * q);
        auto ge = [](float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752f)); };
        dua_store_f16x3_quad(planes, (long)M * F, (long)(4 * i), make_float4(a.x * ge(g.x), a.y * ge(g.y), a.z * ge(g.z), a.w * ge(g.w)), scale);
    }
}
extern "C" int dua_geglu_split_f16x3(const float* x, void* planes, long M, int F, float scale, void* stream) {
    DUA_REQUIRE(x && planes && M > 0 && F > 0 && F % 4 == 0, "dua_geglu_split_f16x3: F must be a positive multiple of 4");
    DUA_REQUIRE(scale > 0.f && (__builtin_bit_cast(unsigned, scale) & 0x7FFFFFu) == 0, "dua_geglu_split_f16x3: scale must be a power of two");
    hipLaunchKernelGGL(k_geglu_split, stream_grid((size_t)M * (F / 4), BLK), BLK, 0, as_stream(stream), x, planes, M, F, scale);
    return dua_check_launch("dua_geglu_split_f16x3");
}
extern "C" int dua_geglu(const float* x, float* y, long M, int F, void* stream) {
    DUA_REQUIRE(x && y && M > 0 && F > 0 && F % 4 == 0, "dua_geglu: F must be a positive multiple of 4");
    hipLaunchKernelGGL(k_geglu, stream_grid((size_t)M * (F / 4), BLK), BLK, 0, as_stream(stream), x, y, M, F);
    return dua_check_launch("dua_geglu");
}

// y[n][i][j][c] = x[n][2i][2j][c]: the stride-2 / pad-1 3x3 convolution of diffusers' Downsample2D is the stride-1 result
// sampled at the even pixels
__global__ __launch_bounds__(BLK) void k_pick_even(const float* __restrict__ x, float* __restrict__ y, int N, int H, int W, int C) {
    const int Q = C / 4, Ho = H / 2, Wo = W / 2;
    size_t total = (size_t)N * Ho * Wo * Q, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int q = (int)(i % Q);
        size_t pp = i / Q;
        const int xo = (int)(pp % Wo);
        pp /= Wo;
        const int yo = (int)(pp % Ho), n = (int)(pp / Ho);
        ((float4*)y)[i] = *(const float4*)(x + ((((size_t)n * H + 2 * yo) * W + 2 * xo) * C + 4 * q));
    }
}
extern "C" int dua_pick_even(const float* x, float* y, int N, int H, int W, int C, void* stream) {
    DUA_REQUIRE(x && y && N > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && C % 4 == 0, "dua_pick_even: even H, W and C %% 4 == 0");
    hipLaunchKernelGGL(k_pick_even, stream_grid((size_t)N * (H / 2) * (W / 2) * (C / 4), BLK), BLK, 0, as_stream(stream), x, y, N, H, W, C);
    return dua_check_launch("dua_pick_even");
}

// =================================== 2x resampling of the skip path ============================
__global__ __launch_bounds__(BLK) void k_resample2x(const float* __restrict__ x, float* __restrict__ y, int N, int H, int W, int C, int mode) {
    const int Q = C / 4;
    const int Ho = mode < 0 ? H / 2 : H, Wo = mode < 0 ? W / 2 : W;
    size_t total = (size_t)N * Ho * Wo * Q, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        int c = (int)(i % Q) * 4;
        size_t pp = i / Q;
        int xo = (int)(pp % Wo);
        size_t t2 = pp / Wo;
        int yo = (int)(t2 % Ho), n = (int)(t2 / Ho);
        if (mode > 0) {
            float4 v = *(const float4*)(x + (((size_t)n * H + yo) * W + xo) * C + c);
            int W2 = 2 * W;
            size_t o = (((size_t)n * 2 * H + 2 * yo) * W2 + 2 * xo) * C + c;
            *(float4*)(y + o) = v;
            *(float4*)(y + o + C) = v;
            *(float4*)(y + o + (size_t)W2 * C) = v;
            *(float4*)(y + o + (size_t)W2 * C + C) = v;
        } else {
            size_t pix = ((size_t)n * H + 2 * yo) * W + 2 * xo;
            float4 a = *(const float4*)(x + pix * C + c), b = *(const float4*)(x + (pix + 1) * C + c);
            float4 d = *(const float4*)(x + (pix + W) * C + c), e = *(const float4*)(x + (pix + W + 1) * C + c);
            *(float4*)(y + (((size_t)n * Ho + yo) * Wo + xo) * C + c) =
                make_float4((a.x + b.x + d.x + e.x) * 0.25f, (a.y + b.y + d.y + e.y) * 0.25f,
                            (a.z + b.z + d.z + e.z) * 0.25f, (a.w + b.w + d.w + e.w) * 0.25f);
        }
    }
}
extern "C" int dua_resample2x(const float* x, float* y, int N, int H, int W, int C, int resample, void* stream) {
    DUA_REQUIRE(x && y && C % 4 == 0 && resample != 0, "dua_resample2x: bad arguments");
    DUA_REQUIRE(resample > 0 || (H % 2 == 0 && W % 2 == 0), "dua_resample2x: pooling needs even H, W");
    int Ho = resample < 0 ? H / 2 : H, Wo = resample < 0 ? W / 2 : W;
    size_t total = (size_t)N * Ho * Wo * (C / 4);
    if (total == 0) return DUA_OK;
    hipLaunchKernelGGL(k_resample2x, stream_grid(total, BLK), BLK, 0, as_stream(stream), x, y, N, H, W, C, resample);
    return dua_check_launch("dua_resample2x");
}

// =================================== row softmax ===============================================
// One wave per row; the row lives in registers (cols <= 64*32).
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
template <int PER>
__global__ __launch_bounds__(BLK) void k_softmax_rows(float* __restrict__ S, long rows, int cols, long ld) {
    long row = (long)blockIdx.x * (BLK / 64) + (threadIdx.x >> 6);
    if (row >= rows) return;
    int lane = threadIdx.x & 63;
    float* r = S + row * ld;
    float v[PER];
    float m = -INFINITY;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        int c = lane + 64 * j;
        v[j] = c < cols ? r[c] : -INFINITY;
        m = fmaxf(m, v[j]);
    }
    m = wave_max(m);
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        v[j] = expf(v[j] - m);
        s += (lane + 64 * j < cols) ? v[j] : 0.f;
    }
    s = wave_sum(s);
    float inv = 1.0f / s;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        int c = lane + 64 * j;
        if (c < cols) r[c] = v[j] * inv;
    }
}
extern "C" int dua_softmax_rows(float* S, long rows, int cols, long ld, void* stream) {
    DUA_REQUIRE(S && rows >= 0 && cols >= 1 && cols <= 4096 && ld >= cols, "dua_softmax_rows: cols must be in [1, 4096], ld >= cols");
    if (rows == 0) return DUA_OK;
    unsigned grid = (unsigned)((rows + 3) / 4);
    if (cols <= 64) hipLaunchKernelGGL(k_softmax_rows<1>, grid, BLK, 0, as_stream(stream), S, rows, cols, ld);
    else if (cols <= 256) hipLaunchKernelGGL(k_softmax_rows<4>, grid, BLK, 0, as_stream(stream), S, rows, cols, ld);
    else if (cols <= 512) hipLaunchKernelGGL(k_softmax_rows<8>, grid, BLK, 0, as_stream(stream), S, rows, cols, ld);
    else if (cols <= 1024) hipLaunchKernelGGL(k_softmax_rows<16>, grid, BLK, 0, as_stream(stream), S, rows, cols, ld);
    else if (cols <= 2048) hipLaunchKernelGGL(k_softmax_rows<32>, grid, BLK, 0, as_stream(stream), S, rows, cols, ld);
    else hipLaunchKernelGGL(k_softmax_rows<64>, grid, BLK, 0, as_stream(stream), S, rows, cols, ld);
    return dua_check_launch("dua_softmax_rows");
}

// =================================== LayerNorm =================================================
template <int PER>
__global__ __launch_bounds__(BLK) void k_layernorm(const float* __restrict__ x, const float* __restrict__ ga,
                                                   const float* __restrict__ be, float* __restrict__ y, long rows, int D, float eps,
                                                   int split, float oscale) {
    long row = (long)blockIdx.x * (BLK / 64) + (threadIdx.x >> 6);
    if (row >= rows) return;
    int lane = threadIdx.x & 63;
    const float* r = x + row * (long)D;
    float v[PER];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        int c = lane + 64 * j;
        v[j] = c < D ? r[c] : 0.f;
        s += v[j];
    }
    float mean = wave_sum(s) / (float)D, q = 0.f;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        float d = (lane + 64 * j < D) ? v[j] - mean : 0.f;
        q += d * d;
    }
    float rstd = rsqrtf(wave_sum(q) / (float)D + eps);
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        int c = lane + 64 * j;
        if (c < D) {
            const float o = (v[j] - mean) * rstd * ga[c] + be[c];
            if (!split) y[row * (long)D + c] = o;
            else if (split == 4) {           // f16x3 planes: fp16 hi | fp16 lo of o * oscale
                if (unsigned long long* rc_ = dua_range_ctr()) dua_range_note(rc_, o * oscale);
                const float sv = fminf(fmaxf(o * oscale, -65504.f), 65504.f);
                const _Float16 h = (_Float16)sv;
                ((_Float16*)y)[row * (long)D + c] = h;
                ((_Float16*)y)[rows * (long)D + row * (long)D + c] = (_Float16)(sv - (float)h);
            } else {
                __bf16* hi = (__bf16*)y;
                const __bf16 h = (__bf16)o;
                hi[row * (long)D + c] = h;
                hi[rows * (long)D + row * (long)D + c] = (__bf16)(o - (float)h);
            }
        }
    }
}
// D % 4 == 0: four consecutive channels per lane and iteration (16-byte loads; 16 / 8 + 8 / 8 + 4 + 4 byte stores)
template <int PER4>
__global__ __launch_bounds__(BLK) void k_layernorm4(const float* __restrict__ x, const float* __restrict__ ga,
                                                    const float* __restrict__ be, float* __restrict__ y, long rows, int D, float eps,
                                                    int split, float oscale) {
    long row = (long)blockIdx.x * (BLK / 64) + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const float* r = x + row * (long)D;
    float4 v[PER4];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < PER4; ++j) {
        const int c = 4 * (lane + 64 * j);
        v[j] = c < D ? *(const float4*)(r + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        s += (v[j].x + v[j].y) + (v[j].z + v[j].w);
    }
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < PER4; ++j)
        if (4 * (lane + 64 * j) < D) {
            const float a = v[j].x - mean, b = v[j].y - mean, c_ = v[j].z - mean, d = v[j].w - mean;
            q += (a * a + b * b) + (c_ * c_ + d * d);
        }
    const float rstd = rsqrtf(wave_sum(q) / (float)D + eps);
#pragma unroll
    for (int j = 0; j < PER4; ++j) {
        const int c = 4 * (lane + 64 * j);
        if (c < D) {
            const float4 g = *(const float4*)(ga + c), b = *(const float4*)(be + c);
            const float4 o = make_float4((v[j].x - mean) * rstd * g.x + b.x, (v[j].y - mean) * rstd * g.y + b.y,
                                         (v[j].z - mean) * rstd * g.z + b.z, (v[j].w - mean) * rstd * g.w + b.w);
            const long off = row * (long)D + c;
            if (!split) *(float4*)(y + off) = o;
            else if (split == 4) dua_store_f16x3_quad(y, rows * (long)D, off, o, oscale);
            else {
                typedef __bf16 ln_bf16x4 __attribute__((ext_vector_type(4)));
                const float f[4] = {o.x, o.y, o.z, o.w};
                ln_bf16x4 h, l;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const __bf16 t = (__bf16)f[u];
                    h[u] = t;
                    l[u] = (__bf16)(f[u] - (float)t);
                }
                *(ln_bf16x4*)((__bf16*)y + off) = h;
                *(ln_bf16x4*)((__bf16*)y + rows * (long)D + off) = l;
            }
        }
    }
}

static int layernorm_impl(const float* x, const float* gamma, const float* beta, float* y, long rows, int D, float eps, int split, void* stream,
                          float oscale = 1.0f) {
    DUA_REQUIRE(x && gamma && beta && y && D >= 1 && D <= 2048, "dua_layernorm: D must be in [1, 2048]");
    if (rows == 0) return DUA_OK;
    unsigned grid = (unsigned)((rows + 3) / 4);
    DuaKTimer kt(stream);
    if (D % 4 == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 15) == 0 && ((uintptr_t)gamma & 15) == 0 && ((uintptr_t)beta & 15) == 0) {
        if (D <= 256) hipLaunchKernelGGL(k_layernorm4<1>, grid, BLK, 0, as_stream(stream), x, gamma, beta, y, rows, D, eps, split, oscale);
        else if (D <= 1024) hipLaunchKernelGGL(k_layernorm4<4>, grid, BLK, 0, as_stream(stream), x, gamma, beta, y, rows, D, eps, split, oscale);
        else hipLaunchKernelGGL(k_layernorm4<8>, grid, BLK, 0, as_stream(stream), x, gamma, beta, y, rows, D, eps, split, oscale);
    } else if (D <= 256) hipLaunchKernelGGL(k_layernorm<4>, grid, BLK, 0, as_stream(stream), x, gamma, beta, y, rows, D, eps, split, oscale);
    else if (D <= 1024) hipLaunchKernelGGL(k_layernorm<16>, grid, BLK, 0, as_stream(stream), x, gamma, beta, y, rows, D, eps, split, oscale);
    else hipLaunchKernelGGL(k_layernorm<32>, grid, BLK, 0, as_stream(stream), x, gamma, beta, y, rows, D, eps, split, oscale);
    kt.done(split ? "k_layernorm (operand planes)" : "k_layernorm (fp32)", 8.0 * (double)rows * D);      // 4 B read + 4 B written per element
    return dua_check_launch("dua_layernorm");
}
extern "C" int dua_layernorm(const float* x, const float* gamma, const float* beta, float* y, long rows, int D, float eps, void* stream) {
    return layernorm_impl(x, gamma, beta, y, rows, D, eps, 0, stream);
}
extern "C" int dua_layernorm_split(const float* x, const float* gamma, const float* beta, void* yplanes, long rows, int D, float eps,
                                   void* stream) {
    return layernorm_impl(x, gamma, beta, (float*)yplanes, rows, D, eps, 1, stream);
}

extern "C" int dua_layernorm_split_f16x3(const float* x, const float* gamma, const float* beta, void* yplanes, long rows, int D, float eps,
                                         float act_scale, void* stream) {
    DUA_REQUIRE(D % 32 == 0 && act_scale > 0.f, "dua_layernorm_split_f16x3: D must be a multiple of 32, act_scale positive");
    return layernorm_impl(x, gamma, beta, (float*)yplanes, rows, D, eps, 4, stream, act_scale);
}

// =================================== embeddings ================================================
__global__ void k_timestep_embedding(const int64_t* __restrict__ t, float* __restrict__ emb, int N, int dim) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * dim) return;
    int n = i / dim, d = i % dim, half = dim / 2;
    float out = 0.f;
    if (d < 2 * half) {
        int j = d < half ? d : d - half;
        // nn.py:112-116: freqs = exp(-log(10000) * j / half) in fp32, args = t.float() * freqs
        float freq = expf(-9.210340371976184f * (float)j / (float)half);
        float a = (float)t[n] * freq;
        out = d < half ? cosf(a) : sinf(a);
    }
    emb[i] = out;
}
extern "C" int dua_timestep_embedding(const int64_t* t, float* emb, int N, int dim, void* stream) {
    DUA_REQUIRE(t && emb && N >= 0 && dim >= 2, "dua_timestep_embedding: bad arguments");
    if (N == 0) return DUA_OK;
    hipLaunchKernelGGL(k_timestep_embedding, (N * dim + 255) / 256, 256, 0, as_stream(stream), t, emb, N, dim);
    return dua_check_launch("dua_timestep_embedding");
}

__global__ void k_embedding_add(const float* __restrict__ a, const float* __restrict__ table, const int64_t* __restrict__ idx,
                                float* __restrict__ out, int N, int D) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * D) return;
    int n = i / D, d = i % D;
    float v = table[(size_t)idx[n] * D + d];
    out[i] = a ? a[i] + v : v;
}
extern "C" int dua_embedding_add(const float* a, const float* table, const int64_t* idx, float* out, int N, int D, void* stream) {
    DUA_REQUIRE(table && idx && out && N >= 0 && D >= 1, "dua_embedding_add: bad arguments");
    if (N == 0) return DUA_OK;
    hipLaunchKernelGGL(k_embedding_add, (N * D + 255) / 256, 256, 0, as_stream(stream), a, table, idx, out, N, D);
    return dua_check_launch("dua_embedding_add");
}

// =================================== U-ViT token plumbing ======================================
__global__ void k_patchify(const float* __restrict__ x, float* __restrict__ rows, int B, int C, int H, int W, int P) {
    int gw = W / P, gh = H / P, K = C * P * P;
    size_t total = (size_t)B * gh * gw * K;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        int k = (int)(i % K);
        size_t r = i / K;
        int pw = (int)(r % gw), ph = (int)((r / gw) % gh), b = (int)(r / ((size_t)gw * gh));
        int c = k / (P * P), p1 = (k / P) % P, p2 = k % P;
        rows[i] = x[(((size_t)b * C + c) * H + ph * P + p1) * W + pw * P + p2];
    }
}
extern "C" int dua_patchify(const float* x, float* rows, int B, int C, int H, int W, int P, void* stream) {
    DUA_REQUIRE(x && rows && P > 0 && H % P == 0 && W % P == 0, "dua_patchify: bad arguments");
    size_t total = (size_t)B * C * H * W;
    if (total == 0) return DUA_OK;
    hipLaunchKernelGGL(k_patchify, stream_grid(total, 256), 256, 0, as_stream(stream), x, rows, B, C, H, W, P);
    return dua_check_launch("dua_patchify");
}

__global__ void k_uvit_assemble(const float* __restrict__ patches, const float* __restrict__ tt, const float* __restrict__ table,
                                const int64_t* __restrict__ y, const float* __restrict__ pos, float* __restrict__ out, int B, int L, int D) {
    int extras = y ? 2 : 1, T = L + extras;
    size_t total = (size_t)B * T * D;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        int d = (int)(i % D);
        size_t r = i / D;
        int tok = (int)(r % T), b = (int)(r / T);
        float v;
        if (y && tok == 0) v = table[(size_t)y[b] * D + d];
        else if (tok == extras - 1) v = tt[(size_t)b * D + d];
        else v = patches[((size_t)b * L + (tok - extras)) * D + d];
        out[i] = v + pos[(size_t)tok * D + d];
    }
}
extern "C" int dua_uvit_assemble(const float* patches, const float* time_tok, const float* label_table, const int64_t* y,
                                 const float* pos_embed, float* out, int B, int L, int D, void* stream) {
    DUA_REQUIRE(patches && time_tok && pos_embed && out && (!y || label_table), "dua_uvit_assemble: bad arguments");
    size_t total = (size_t)B * (L + 2) * D;
    if (B == 0) return DUA_OK;
    hipLaunchKernelGGL(k_uvit_assemble, stream_grid(total, 256), 256, 0, as_stream(stream), patches, time_tok, label_table, y, pos_embed, out, B, L, D);
    return dua_check_launch("dua_uvit_assemble");
}

__global__ void k_unpatchify(const float* __restrict__ tok, float* __restrict__ img, int B, int L, int extras, int C, int P, int g) {
    int H = g * P;
    size_t total = (size_t)B * C * H * H;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        int xw = (int)(i % H), yh = (int)((i / H) % H), c = (int)((i / ((size_t)H * H)) % C), b = (int)(i / ((size_t)H * H * C));
        int ph = yh / P, p1 = yh % P, pw = xw / P, p2 = xw % P;
        img[i] = tok[((size_t)b * (L + extras) + extras + ph * g + pw) * (P * P * C) + (p1 * P + p2) * C + c];
    }
}
extern "C" int dua_unpatchify(const float* tok, float* img, int B, int L, int extras, int C, int P, void* stream) {
    int g = 1;
    while (g * g < L) ++g;
    DUA_REQUIRE(tok && img && g * g == L && extras >= 0, "dua_unpatchify: L must be a square");
    size_t total = (size_t)B * C * g * P * g * P;
    if (total == 0) return DUA_OK;
    hipLaunchKernelGGL(k_unpatchify, stream_grid(total, 256), 256, 0, as_stream(stream), tok, img, B, L, extras, C, P, g);
    return dua_check_launch("dua_unpatchify");
}

// =================================== batched strided transpose =================================
// out[b][c][r] = in[b][r][c] for r < R, c < Cc (32x32 LDS tiles; batch -> (outer, inner) strides).  Used to lay V out
// K-major ([head channel][token]) so that P.V runs through the fast K-major ("NT") GEMM path.
__global__ __launch_bounds__(BLK) void k_transpose(const float* __restrict__ in, float* __restrict__ out, int R, int Cc, long ld_in,
                                                   long ld_out, long sIn, long sIni, long sOut, long sOuti, int batch_inner) {
    __shared__ float tile[32][33];
    const int z = blockIdx.z, zo = z / batch_inner, zi = z % batch_inner;
    in += zo * sIn + zi * sIni;
    out += zo * sOut + zi * sOuti;
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;        // 32 x 8
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = r0 + ty + 8 * j, c = c0 + tx;
        tile[ty + 8 * j][tx] = (r < R && c < Cc) ? in[(long)r * ld_in + c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = c0 + ty + 8 * j, r = r0 + tx;
        if (c < Cc && r < R) out[(long)c * ld_out + r] = tile[tx][ty + 8 * j];
    }
}
extern "C" int dua_transpose_batched(const float* in, float* out, int R, int Cc, long ld_in, long ld_out, long sIn, long sIni,
                                     long sOut, long sOuti, int batch, int batch_inner, void* stream) {
    DUA_REQUIRE(in && out && R > 0 && Cc > 0 && batch >= 1 && batch_inner >= 1 && batch % batch_inner == 0 && batch <= 65535, "dua_transpose_batched: bad arguments");
    hipLaunchKernelGGL(k_transpose, dim3((Cc + 31) / 32, (R + 31) / 32, batch), BLK, 0, as_stream(stream), in, out, R, Cc, ld_in, ld_out,
                       sIn, sIni, sOut, sOuti, batch_inner);
    return dua_check_launch("dua_transpose_batched");
}

DUA_RANGE_TU(denoiser)

// ---- f16x3 range check: host side (see common.h).  Process-global debug facility; dua_range_check_read synchronises the device.
int dua_range_set_conv_wino(unsigned long long*);
int dua_range_set_conv_wino4(unsigned long long*);
int dua_range_set_xtile(unsigned long long*);
int dua_range_set_hgemm(unsigned long long*);
int dua_range_set_sgemm(unsigned long long*);
int dua_range_set_attention(unsigned long long*);
static unsigned long long* g_range_buf = nullptr;
static int g_range_on = 0;
static int range_set_all(unsigned long long* p) {              // every translation unit's copy of the pointer; non-zero: one of the copies failed
    return dua_range_set_denoiser(p) | dua_range_set_conv_wino(p) | dua_range_set_conv_wino4(p) | dua_range_set_xtile(p) | dua_range_set_hgemm(p) |
           dua_range_set_sgemm(p) | dua_range_set_attention(p);
}
extern "C" int dua_range_check(int on) {
    const int was = g_range_on;
    if (on && !g_range_buf) {
        if (hipMalloc((void**)&g_range_buf, 2 * sizeof(unsigned long long)) != hipSuccess) { dua_set_error("dua_range_check: hipMalloc failed"); return DUA_EHIP; }
    }
    if (on && hipMemset(g_range_buf, 0, 2 * sizeof(unsigned long long)) != hipSuccess) { dua_set_error("dua_range_check: hipMemset failed"); return DUA_EHIP; }
    if (hipDeviceSynchronize() != hipSuccess) { dua_set_error("dua_range_check: device error"); return DUA_EHIP; }
    if (range_set_all(on ? g_range_buf : nullptr)) { dua_set_error("dua_range_check: hipMemcpyToSymbol failed"); return DUA_EHIP; }
    g_range_on = on != 0;
    return was;
}
extern "C" int dua_range_check_read(unsigned long long* saturated, unsigned long long* subnormal, int reset) {
    DUA_REQUIRE(g_range_buf != nullptr, "dua_range_check_read: dua_range_check(1) first");
    unsigned long long h[2] = {0, 0};
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(h, g_range_buf, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) {
        dua_set_error("dua_range_check_read: device error");
        return DUA_EHIP;
    }
    if (saturated) *saturated = h[0];
    if (subnormal) *subnormal = h[1];
    if (reset) (void)hipMemset(g_range_buf, 0, sizeof(h));
    return DUA_OK;
}
