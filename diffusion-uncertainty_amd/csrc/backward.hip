// Input-gradient kernels (vector-Jacobian products w.r.t. ACTIVATIONS, never weights) of the latent UNet.
//
// Gradient-mode guidance of the reference, uncertainty_guidance.py:83-108,121-129 (use_posterior = False, the default of
// scripts/generate_with_uncertainty_threshold_stable_diffusion.py:16-17), differentiates sum(var_m eps_hat_m) through the M
// denoiser calls w.r.t. the predicted epsilon.  The linear layers of that backward pass reuse the forward engines with
// transposed / flipped weights (dua_gemm, dua_conv3x3: exact fp32 MFMA); this file holds what has no forward counterpart:
// GroupNorm(+SiLU), LayerNorm, GEGLU and softmax input gradients, the adjoint of the even-pixel pick of the stride-2
// convolutions and the seed gradient of the variance.  All HBM-bound, fp32.
#include "common.h"

#define BLK 256

namespace {

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// sums of two values over the workgroup (BLK threads); every thread gets the totals
__device__ __forceinline__ void block_sum2(float& a, float& b, float* sh) {
    a = wsum(a);
    b = wsum(b);
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    __syncthreads();
    if (l == 0) {
        sh[2 * w] = a;
        sh[2 * w + 1] = b;
    }
    __syncthreads();
    a = 0.f;
    b = 0.f;
#pragma unroll
    for (int i = 0; i < BLK / 64; ++i) {
        a += sh[2 * i];
        b += sh[2 * i + 1];
    }
}

struct GnBwdP {
    const float *x1, *x2, *stats, *gamma, *beta, *dy, *add1, *add2;
    float *dx1, *dx2;
    int C1, C2, N, G, act;
    long HW;
};

// One workgroup per (group, sample).  y = act(xh * gamma + beta), xh = (x - mean) * rstd over the group:
//   dxh = dy * act'(pre) * gamma;  dx = rstd * (dxh - mean(dxh) - xh * mean(dxh * xh))   (+ add: the other branch of a fan-out)
__global__ __launch_bounds__(BLK) void k_gn_bwd(GnBwdP p) {
    __shared__ float sh[2 * (BLK / 64)];
    const int g = blockIdx.x, n = blockIdx.y, C = p.C1 + p.C2, cpg = C / p.G, c0 = g * cpg;
    const float mean = p.stats[2 * ((size_t)n * p.G + g)], rstd = p.stats[2 * ((size_t)n * p.G + g) + 1];
    const long cnt = p.HW * cpg;
    auto term = [&](long i, float& xh, int& c, long& px) -> float {
        px = i / cpg;
        c = c0 + (int)(i - px * cpg);
        const size_t pix = (size_t)n * p.HW + px;
        const float x = c < p.C1 ? p.x1[pix * p.C1 + c] : p.x2[pix * p.C2 + (c - p.C1)];
        xh = (x - mean) * rstd;
        const float ga = p.gamma[c];
        float d = p.dy[pix * C + c];
        if (p.act == 1) {
            const float pre = xh * ga + p.beta[c];
            const float sg = 1.0f / (1.0f + __expf(-pre));
            d *= sg * (1.0f + pre * (1.0f - sg));
        }
        return d * ga;
    };
    float s1 = 0.f, s2 = 0.f;
    for (long i = threadIdx.x; i < cnt; i += BLK) {
        float xh;
        int c;
        long px;
        const float d = term(i, xh, c, px);
        s1 += d;
        s2 += d * xh;
    }
    block_sum2(s1, s2, sh);
    const float m1 = s1 / (float)cnt, m2 = s2 / (float)cnt;
    for (long i = threadIdx.x; i < cnt; i += BLK) {
        float xh;
        int c;
        long px;
        const float d = term(i, xh, c, px);
        const size_t pix = (size_t)n * p.HW + px;
        float o = rstd * (d - m1 - xh * m2);
        if (c < p.C1) {
            if (p.add1) o += p.add1[pix * p.C1 + c];
            p.dx1[pix * p.C1 + c] = o;
        } else {
            const size_t k = pix * p.C2 + (c - p.C1);
            if (p.add2) o += p.add2[k];
            p.dx2[k] = o;
        }
    }
}

// One wave per row.  y = xh * gamma + beta;  dx = rstd * (dy*gamma - mean(dy*gamma) - xh * mean(dy*gamma*xh)) (+ add)
template <int PER>
__global__ __launch_bounds__(BLK) void k_ln_bwd(const float* __restrict__ x, const float* __restrict__ ga, const float* __restrict__ dy,
                                                const float* __restrict__ add, float* __restrict__ dx, long rows, int D, float eps) {
    const long row = (long)blockIdx.x * (BLK / 64) + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const float* r = x + row * (long)D;
    float v[PER], d[PER];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int c = lane + 64 * j;
        v[j] = c < D ? r[c] : 0.f;
        s += v[j];
    }
    const float mean = wsum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const float t = (lane + 64 * j < D) ? v[j] - mean : 0.f;
        q += t * t;
    }
    const float rstd = rsqrtf(wsum(q) / (float)D + eps);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int c = lane + 64 * j;
        v[j] = (v[j] - mean) * rstd;
        d[j] = c < D ? dy[row * (long)D + c] * ga[c] : 0.f;
        s1 += d[j];
        s2 += c < D ? d[j] * v[j] : 0.f;
    }
    const float m1 = wsum(s1) / (float)D, m2 = wsum(s2) / (float)D;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int c = lane + 64 * j;
        if (c < D) {
            float o = rstd * (d[j] - m1 - v[j] * m2);
            if (add) o += add[row * (long)D + c];
            dx[row * (long)D + c] = o;
        }
    }
}

// y = a * gelu(g) with x = [a | g]:  da = dy * gelu(g);  dg = dy * a * gelu'(g)   (exact erf GELU)
__global__ __launch_bounds__(BLK) void k_geglu_bwd(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx, long M, int F) {
    const size_t total = (size_t)M * F, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const size_t m = i / F;
        const int f = (int)(i - m * F);
        const float a = x[m * 2 * F + f], g = x[m * 2 * F + F + f], d = dy[i];
        const float cdf = 0.5f * (1.0f + erff(g * 0.70710678118654752f));
        const float pdf = 0.3989422804014327f * __expf(-0.5f * g * g);
        dx[m * 2 * F + f] = d * (g * cdf);
        dx[m * 2 * F + F + f] = d * a * (cdf + g * pdf);
    }
}

// One wave per row, in place on dP:  dS = P * (dP - sum_j dP_j * P_j)
template <int PER>
__global__ __launch_bounds__(BLK) void k_softmax_bwd(const float* __restrict__ P, float* __restrict__ dP, long rows, int cols, long ld) {
    const long row = (long)blockIdx.x * (BLK / 64) + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const float* pr = P + row * ld;
    float* dr = dP + row * ld;
    float pv[PER], dv[PER];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int c = lane + 64 * j;
        pv[j] = c < cols ? pr[c] : 0.f;
        dv[j] = c < cols ? dr[c] : 0.f;
        s += pv[j] * dv[j];
    }
    s = wsum(s);
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int c = lane + 64 * j;
        if (c < cols) dr[c] = pv[j] * (dv[j] - s);
    }
}

// One wave per row: S holds logits on entry; on exit S = P = softmax(S) and dP = dS = P * (dP - sum_j dP_j P_j).  One pass over
// the two matrices instead of a softmax pass followed by the gradient pass.
template <int PER>
__global__ __launch_bounds__(BLK) void k_softmax_fwd_bwd(float* __restrict__ S, float* __restrict__ dP, long rows, int cols, long ld) {
    const long row = (long)blockIdx.x * (BLK / 64) + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    float* pr = S + row * ld;
    float* dr = dP + row * ld;
    float pv[PER], dv[PER];
    float m = -INFINITY;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int c = lane + 64 * j;
        pv[j] = c < cols ? pr[c] : -INFINITY;
        dv[j] = c < cols ? dr[c] : 0.f;
        m = fmaxf(m, pv[j]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    float z = 0.f;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        pv[j] = expf(pv[j] - m);                  // exp(-inf) = 0 on the padding lanes
        z += pv[j];
    }
    const float inv = 1.0f / wsum(z);
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        pv[j] *= inv;
        s += pv[j] * dv[j];
    }
    s = wsum(s);
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int c = lane + 64 * j;
        if (c < cols) {
            pr[c] = pv[j];
            dr[c] = pv[j] * (dv[j] - s);
        }
    }
}

// adjoint of y[n][i][j] = x[n][2i][2j]: the gradient lands on the even pixels, zero elsewhere
__global__ __launch_bounds__(BLK) void k_place_even(const float* __restrict__ g, float* __restrict__ out, int N, int H, int W, int C) {
    const int Q = C / 4, Ho = H / 2, Wo = W / 2;
    const size_t total = (size_t)N * H * W * Q, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int q = (int)(i % Q);
        size_t pp = i / Q;
        const int xo = (int)(pp % W);
        pp /= W;
        const int yo = (int)(pp % H), n = (int)(pp / H);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (!(xo & 1) && !(yo & 1) && (yo >> 1) < Ho && (xo >> 1) < Wo)
            v = *(const float4*)(g + (((size_t)n * Ho + (yo >> 1)) * Wo + (xo >> 1)) * C + 4 * q);
        ((float4*)out)[i] = v;
    }
}

// G[m] = scale * (e[m] - mean_m e): gradient of scale/2 * sum_m (e_m - mean)^2 w.r.t. e_m (scale = 2 / ((M-1) * B) for the
// unbiased variance averaged over the batch axis and summed over pixels, uncertainty_guidance.py:106-107)
__global__ __launch_bounds__(BLK) void k_mc_var_grad(const float* __restrict__ stack, float* __restrict__ G, int M, size_t n, float scale) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float s = 0.f;
        for (int m = 0; m < M; ++m) s += stack[(size_t)m * n + i];
        const float mean = s / (float)M;
        for (int m = 0; m < M; ++m) G[(size_t)m * n + i] = scale * (stack[(size_t)m * n + i] - mean);
    }
}

}  // namespace

extern "C" int dua_groupnorm_bwd(const float* x1, int C1, const float* x2, int C2, int N, long HW, int G, const float* stats,
                                 const float* gamma, const float* beta, int act, const float* dy, const float* add1, const float* add2,
                                 float* dx1, float* dx2, void* stream) {
    DUA_REQUIRE(x1 && stats && gamma && beta && dy && dx1 && C1 > 0 && C2 >= 0 && N > 0 && HW > 0 && G > 0, "dua_groupnorm_bwd: null / empty operand");
    DUA_REQUIRE((C2 == 0) == (x2 == nullptr) && (C2 == 0 || dx2), "dua_groupnorm_bwd: x2 / dx2 must be given exactly when C2 > 0");
    DUA_REQUIRE((C1 + C2) % G == 0 && N <= 65535 && (act == 0 || act == 1), "dua_groupnorm_bwd: channels must divide into G groups, act in {0, 1}");
    GnBwdP p{x1, x2, stats, gamma, beta, dy, add1, add2, dx1, dx2, C1, C2, N, G, act, HW};
    hipLaunchKernelGGL(k_gn_bwd, dim3(G, N), BLK, 0, as_stream(stream), p);
    return dua_check_launch("dua_groupnorm_bwd");
}

extern "C" int dua_layernorm_bwd(const float* x, const float* gamma, const float* dy, const float* add, float* dx, long rows, int D,
                                 float eps, void* stream) {
    DUA_REQUIRE(x && gamma && dy && dx && rows >= 0 && D >= 1 && D <= 2048, "dua_layernorm_bwd: D must be in [1, 2048]");
    if (rows == 0) return DUA_OK;
    const unsigned grid = (unsigned)((rows + 3) / 4);
    if (D <= 256) hipLaunchKernelGGL(k_ln_bwd<4>, grid, BLK, 0, as_stream(stream), x, gamma, dy, add, dx, rows, D, eps);
    else if (D <= 1024) hipLaunchKernelGGL(k_ln_bwd<16>, grid, BLK, 0, as_stream(stream), x, gamma, dy, add, dx, rows, D, eps);
    else hipLaunchKernelGGL(k_ln_bwd<32>, grid, BLK, 0, as_stream(stream), x, gamma, dy, add, dx, rows, D, eps);
    return dua_check_launch("dua_layernorm_bwd");
}

extern "C" int dua_geglu_bwd(const float* x, const float* dy, float* dx, long M, int F, void* stream) {
    DUA_REQUIRE(x && dy && dx && M > 0 && F > 0, "dua_geglu_bwd: null / empty operand");
    hipLaunchKernelGGL(k_geglu_bwd, stream_grid((size_t)M * F, BLK), BLK, 0, as_stream(stream), x, dy, dx, M, F);
    return dua_check_launch("dua_geglu_bwd");
}

extern "C" int dua_softmax_bwd_rows(const float* P, float* dP, long rows, int cols, long ld, void* stream) {
    DUA_REQUIRE(P && dP && rows >= 0 && cols >= 1 && cols <= 4096 && ld >= cols, "dua_softmax_bwd_rows: cols must be in [1, 4096], ld >= cols");
    if (rows == 0) return DUA_OK;
    const unsigned grid = (unsigned)((rows + 3) / 4);
    if (cols <= 64) hipLaunchKernelGGL(k_softmax_bwd<1>, grid, BLK, 0, as_stream(stream), P, dP, rows, cols, ld);
    else if (cols <= 256) hipLaunchKernelGGL(k_softmax_bwd<4>, grid, BLK, 0, as_stream(stream), P, dP, rows, cols, ld);
    else if (cols <= 1024) hipLaunchKernelGGL(k_softmax_bwd<16>, grid, BLK, 0, as_stream(stream), P, dP, rows, cols, ld);
    else if (cols <= 2048) hipLaunchKernelGGL(k_softmax_bwd<32>, grid, BLK, 0, as_stream(stream), P, dP, rows, cols, ld);
    else hipLaunchKernelGGL(k_softmax_bwd<64>, grid, BLK, 0, as_stream(stream), P, dP, rows, cols, ld);
    return dua_check_launch("dua_softmax_bwd_rows");
}

extern "C" int dua_softmax_fwd_bwd_rows(float* S, float* dP, long rows, int cols, long ld, void* stream) {
    DUA_REQUIRE(S && dP && rows >= 0 && cols >= 1 && cols <= 4096 && ld >= cols, "dua_softmax_fwd_bwd_rows: cols must be in [1, 4096], ld >= cols");
    if (rows == 0) return DUA_OK;
    const unsigned grid = (unsigned)((rows + 3) / 4);
    if (cols <= 64) hipLaunchKernelGGL(k_softmax_fwd_bwd<1>, grid, BLK, 0, as_stream(stream), S, dP, rows, cols, ld);
    else if (cols <= 256) hipLaunchKernelGGL(k_softmax_fwd_bwd<4>, grid, BLK, 0, as_stream(stream), S, dP, rows, cols, ld);
    else if (cols <= 1024) hipLaunchKernelGGL(k_softmax_fwd_bwd<16>, grid, BLK, 0, as_stream(stream), S, dP, rows, cols, ld);
    else if (cols <= 2048) hipLaunchKernelGGL(k_softmax_fwd_bwd<32>, grid, BLK, 0, as_stream(stream), S, dP, rows, cols, ld);
    else hipLaunchKernelGGL(k_softmax_fwd_bwd<64>, grid, BLK, 0, as_stream(stream), S, dP, rows, cols, ld);
    return dua_check_launch("dua_softmax_fwd_bwd_rows");
}

extern "C" int dua_place_even(const float* g, float* out, int N, int H, int W, int C, void* stream) {
    DUA_REQUIRE(g && out && N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && H % 2 == 0 && W % 2 == 0,
                "dua_place_even: H, W must be even and C a multiple of 4");
    hipLaunchKernelGGL(k_place_even, stream_grid((size_t)N * H * W * (C / 4), BLK), BLK, 0, as_stream(stream), g, out, N, H, W, C);
    return dua_check_launch("dua_place_even");
}

extern "C" int dua_mc_var_grad(const float* stack, float* grad, int M, size_t n, float scale, void* stream) {
    DUA_REQUIRE(stack && grad && M >= 2 && n > 0, "dua_mc_var_grad: needs M >= 2 samples");
    hipLaunchKernelGGL(k_mc_var_grad, stream_grid(n, BLK), BLK, 0, as_stream(stream), stack, grad, M, n, scale);
    return dua_check_launch("dua_mc_var_grad");
}
