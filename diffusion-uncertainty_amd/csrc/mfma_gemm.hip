// fp32 MFMA tile engine: batched GEMM and implicit-GEMM 3x3 convolution on v_mfma_f32_32x32x2_f32.
//
// Workgroup = 256 threads = 4 waves (2x2), block tile 128(M) x 128(N) x 32(K); each wave owns a 64x64
// sub-tile = 2x2 MFMA tiles of 32x32 (64 accumulator VGPRs).  Operand tiles are staged
// global -> registers -> LDS (so the im2col gather, zero padding, K-concat of two sources and the SiLU
// on the embedding input are applied in flight) into double-buffered [row][k] images with a 36-float
// row stride: the 16-lane groups of ds_read_b128 then hit 16 distinct 4-bank slots (conflict-free) and
// the 8-lane groups of ds_write_b128 cover all 32 banks.  One barrier per K step; the next tile's
// global loads are issued before the MFMAs of the current one.
//
// K is consumed in a permuted order inside each 32-deep step: a lane with k-half kh = lane>>5 reads the
// 4 contiguous floats k = 8g + 4kh .. +3 (one ds_read_b128) and feeds element e to MFMA (g, e), whose two
// k-slots are therefore {8g+e, 8g+4+e} for A and B alike.
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define BM 128
#define BN 128
#define BK 32
#define LDT 36                    // LDS row stride (floats) of [row][k] images
#define LDN 132                   // LDS row stride of the [k][n] image used when B is K-major
#define TILE_F (BM * LDT)         // 4608 floats per operand tile

enum { MODE_GEMM = 0, MODE_CONV = 1, MODE_CONV_SMALLC = 2 };

struct TileP {
    const float *A1, *A2, *B, *bias, *res;
    float* C;
    int M, N, K1, K2;
    long lda1, lda2, ldb, ldc, ldr;
    long sA1, sA2, sB, sC, sR, sA1i, sBi, sCi;
    int batch_inner;
    int act_in, act_out;
    float alpha;
    // conv
    int H, W, Cin, nchw_out;
    int ntm, ntn;                  // tile counts
};

__device__ __forceinline__ float gelu_f(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752f)); }

template <int MODE, bool TRANSB>
__global__ __launch_bounds__(256, 2) void k_tile(TileP p) {
    __shared__ __attribute__((aligned(16))) float lds[2 * 2 * TILE_F];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    // XCD-aware tile order: workgroups that share an XCD (equal id % 8) get a contiguous run of tiles, so the
    // A rows / halo rows and the weight panel they share stay in that XCD's L2.
    const int nwg = gridDim.x;
    int lid = blockIdx.x;
    {
        int q = nwg >> 3, r = nwg & 7, xcd = lid & 7, idx = lid >> 3;
        lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tn = lid % p.ntn, tm = lid / p.ntn;
    const int m0 = tm * BM, n0 = tn * BN;

    const float *A1 = p.A1, *A2 = p.A2, *Bp = p.B, *res = p.res;
    float* Cp = p.C;
    if (MODE == MODE_GEMM) {
        int z = blockIdx.y, zo = z / p.batch_inner, zi = z % p.batch_inner;
        A1 += zo * p.sA1 + zi * p.sA1i;
        if (A2) A2 += zo * p.sA2;
        Bp += zo * p.sB + zi * p.sBi;
        Cp += zo * p.sC + zi * p.sCi;
        if (res) res += zo * p.sR;
    }

    // ---- staging geometry: thread -> (row r0 + 32 i, float4 column c4) of a [128][32] tile
    const int c4 = tid & 7, r0 = tid >> 3;
    const int K = (MODE == MODE_GEMM) ? p.K1 + p.K2 : (MODE == MODE_CONV ? 9 * p.Cin : 64);
    const int nk = (K + BK - 1) / BK;

    // conv: decode the 4 output pixels this thread stages
    int py[4], px[4];
    long pbase[4];
    if (MODE != MODE_GEMM) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int m = m0 + r0 + 32 * i;
            if (m < p.M) {
                int hw = p.H * p.W, n = m / hw, rem = m - n * hw;
                py[i] = rem / p.W;
                px[i] = rem - py[i] * p.W;
                pbase[i] = (long)m * p.Cin;
            } else {
                py[i] = -100000;
                px[i] = 0;
                pbase[i] = 0;
            }
        }
    }

    float4 ra[4], rb[4];
    auto load_tiles = [&](int kt) {
        const int k0 = kt * BK;
        // ---------------- A ----------------
        if (MODE == MODE_GEMM) {
            const int k = k0 + 4 * c4;
            const float* src;
            long ld;
            int kk;
            if (k < p.K1) { src = A1; ld = p.lda1; kk = k; } else { src = A2; ld = p.lda2; kk = k - p.K1; }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int m = m0 + r0 + 32 * i;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (m < p.M) {
                    const float* q = src + (long)m * ld + kk;
                    if (k + 3 < K) v = *(const float4*)q;
                    else if (k < K) { v.x = q[0]; if (k + 1 < K) v.y = q[1]; if (k + 2 < K) v.z = q[2]; }
                }
                if (p.act_in == 1) { v.x = silu_f(v.x); v.y = silu_f(v.y); v.z = silu_f(v.z); v.w = silu_f(v.w); }
                ra[i] = v;
            }
        } else if (MODE == MODE_CONV) {
            const int cpb = p.Cin / BK;
            const int tap = kt / cpb, cb = kt - tap * cpb;
            const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
            const long off = ((long)dy * p.W + dx) * p.Cin + cb * BK + 4 * c4;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int yy = py[i] + dy, xx = px[i] + dx;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (yy >= 0 && yy < p.H && xx >= 0 && xx < p.W) v = *(const float4*)(A1 + pbase[i] + off);
                ra[i] = v;
            }
        } else {   // small-Cin: float4 column = one tap, 4 channels
            const int tap = kt * 8 + c4;
            const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
            const long off = ((long)dy * p.W + dx) * 4;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int yy = py[i] + dy, xx = px[i] + dx;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (tap < 9 && yy >= 0 && yy < p.H && xx >= 0 && xx < p.W) v = *(const float4*)(A1 + pbase[i] + off);
                ra[i] = v;
            }
        }
        // ---------------- B ----------------
        if (TRANSB) {   // B is [N][K]: same geometry as A
            const int k = k0 + 4 * c4;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int n = n0 + r0 + 32 * i;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (MODE != MODE_GEMM) v = *(const float4*)(Bp + (long)n * K + k);          // packed + padded weights
                else if (n < p.N) {
                    const float* q = Bp + (long)n * p.ldb + k;
                    if (k + 3 < K) v = *(const float4*)q;
                    else if (k < K) { v.x = q[0]; if (k + 1 < K) v.y = q[1]; if (k + 2 < K) v.z = q[2]; }
                }
                rb[i] = v;
            }
        } else {        // B is [K][N]: thread -> (k row kr + 8 i, float4 column n4)
            const int n4 = tid & 31, kr = tid >> 5;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int k = k0 + kr + 8 * i, n = n0 + 4 * n4;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (k < K && n < p.N) v = *(const float4*)(Bp + (long)k * p.ldb + n);       // N % 4 == 0 checked on host
                rb[i] = v;
            }
        }
    };
    auto store_tiles = [&](int buf) {
        float* la = lds + buf * 2 * TILE_F;
        float* lb = la + TILE_F;
#pragma unroll
        for (int i = 0; i < 4; ++i) *(float4*)(la + (r0 + 32 * i) * LDT + 4 * c4) = ra[i];
        if (TRANSB) {
#pragma unroll
            for (int i = 0; i < 4; ++i) *(float4*)(lb + (r0 + 32 * i) * LDT + 4 * c4) = rb[i];
        } else {
            const int n4 = tid & 31, kr = tid >> 5;
#pragma unroll
            for (int i = 0; i < 4; ++i) *(float4*)(lb + (kr + 8 * i) * LDN + 4 * n4) = rb[i];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int lr = lane & 31, kh = lane >> 5;
    load_tiles(0);
    store_tiles(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) load_tiles(kt + 1);
        const float* la = lds + buf * 2 * TILE_F + (wm * 64 + lr) * LDT + 4 * kh;
        const float* lb = lds + buf * 2 * TILE_F + TILE_F;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float4 a0 = *(const float4*)(la + 8 * g), a1 = *(const float4*)(la + 32 * LDT + 8 * g);
            float4 b0, b1;
            if (TRANSB) {
                const float* q = lb + (wn * 64 + lr) * LDT + 4 * kh + 8 * g;
                b0 = *(const float4*)q;
                b1 = *(const float4*)(q + 32 * LDT);
            } else {
                const float* q = lb + (8 * g + 4 * kh) * LDN + wn * 64 + lr;
                b0 = make_float4(q[0], q[LDN], q[2 * LDN], q[3 * LDN]);
                b1 = make_float4(q[32], q[LDN + 32], q[2 * LDN + 32], q[3 * LDN + 32]);
            }
            const float av0[4] = {a0.x, a0.y, a0.z, a0.w}, av1[4] = {a1.x, a1.y, a1.z, a1.w};
            const float bv0[4] = {b0.x, b0.y, b0.z, b0.w}, bv1[4] = {b1.x, b1.y, b1.z, b1.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0[e], bv0[e], acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0[e], bv1[e], acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1[e], bv0[e], acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1[e], bv1[e], acc[1][1], 0, 0, 0);
            }
        }
        if (kt + 1 < nk) store_tiles(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: C/D layout col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const int col = n0 + wn * 64 + ni * 32 + lr;
            if (col >= p.N) continue;
            const float bv = p.bias ? p.bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (row >= p.M) continue;
                float v = acc[mi][ni][r] * p.alpha + bv;
                if (p.act_out == 2) v = gelu_f(v);
                if (res) v += res[(long)row * p.ldr + col];
                if (MODE != MODE_GEMM && p.nchw_out) {
                    int hw = p.H * p.W, n = row / hw, rem = row - n * hw;
                    Cp[((long)n * p.N + col) * hw + rem] = v;
                } else {
                    Cp[(long)row * p.ldc + col] = v;
                }
            }
        }
}

template <int MODE, bool TRANSB>
static int launch_tile(TileP& p, int batch, hipStream_t s) {
    p.ntm = (p.M + BM - 1) / BM;
    p.ntn = (p.N + BN - 1) / BN;
    dim3 grid(p.ntm * p.ntn, batch);
    hipLaunchKernelGGL((k_tile<MODE, TRANSB>), grid, 256, 0, s, p);
    return dua_check_launch("dua tile kernel");
}

extern "C" int dua_gemm(const dua_gemm_args* a, void* stream) {
    DUA_REQUIRE(a && a->A1 && a->B && a->C, "dua_gemm: null pointer");
    DUA_REQUIRE(a->M > 0 && a->N > 0 && a->K1 > 0 && a->K2 >= 0 && a->batch >= 1 && a->batch_inner >= 1, "dua_gemm: bad sizes");
    DUA_REQUIRE(a->lda1 % 4 == 0 && a->ldb % 4 == 0, "dua_gemm: leading dims must be multiples of 4 floats (K itself may be ragged)");
    DUA_REQUIRE((a->K2 == 0) == (a->A2 == nullptr), "dua_gemm: A2/K2 mismatch");
    DUA_REQUIRE(a->K2 == 0 || (a->K1 % 4 == 0 && a->lda2 % 4 == 0), "dua_gemm: K1 must be a multiple of 4 for a two-source A");
    DUA_REQUIRE(a->transB || a->N % 4 == 0, "dua_gemm: N must be a multiple of 4 when B is [K][N]");
    DUA_REQUIRE(a->batch % a->batch_inner == 0, "dua_gemm: batch must be a multiple of batch_inner");
    DUA_REQUIRE((((uintptr_t)a->A1 | (uintptr_t)a->B | (uintptr_t)a->A2) & 15) == 0, "dua_gemm: operands must be 16-byte aligned");
    TileP p{};
    p.A1 = a->A1; p.A2 = a->A2; p.B = a->B; p.bias = a->bias; p.res = a->residual; p.C = a->C;
    p.M = a->M; p.N = a->N; p.K1 = a->K1; p.K2 = a->K2;
    p.lda1 = a->lda1; p.lda2 = a->lda2; p.ldb = a->ldb; p.ldc = a->ldc; p.ldr = a->ldr;
    p.sA1 = a->sA1; p.sA2 = a->sA2; p.sB = a->sB; p.sC = a->sC; p.sR = a->sR;
    p.sA1i = a->sA1i; p.sBi = a->sBi; p.sCi = a->sCi;
    p.batch_inner = a->batch_inner; p.act_in = a->act_in; p.act_out = a->act_out; p.alpha = a->alpha;
    return a->transB ? launch_tile<MODE_GEMM, true>(p, a->batch, as_stream(stream))
                     : launch_tile<MODE_GEMM, false>(p, a->batch, as_stream(stream));
}

static inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

extern "C" size_t dua_conv3x3_packed_floats(int Cout, int Cin) {
    int coutp = round_up(Cout, BN);
    return Cin <= 4 ? (size_t)coutp * 64 : (size_t)coutp * 9 * Cin;
}

__global__ void k_pack_conv(const float* __restrict__ w, float* __restrict__ wp, int Cout, int CinReal, int CoutPad, int small) {
    size_t Kp = small ? 64 : (size_t)9 * CinReal;
    size_t total = (size_t)CoutPad * Kp;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        int co = (int)(i / Kp);
        int k = (int)(i % Kp);
        int tap, ci;
        if (small) { tap = k / 4; ci = k % 4; } else { tap = k / CinReal; ci = k % CinReal; }
        float v = 0.f;
        if (co < Cout && tap < 9 && ci < CinReal) v = w[((size_t)co * CinReal + ci) * 9 + tap];
        wp[i] = v;
    }
}
extern "C" int dua_pack_conv3x3_weight(const float* w, float* wp, int Cout, int CinReal, void* stream) {
    DUA_REQUIRE(w && wp && Cout > 0 && CinReal > 0, "dua_pack_conv3x3_weight: bad arguments");
    DUA_REQUIRE(CinReal <= 4 || CinReal % BK == 0, "dua_pack_conv3x3_weight: Cin must be <= 4 or a multiple of 32 (got %d)", CinReal);
    int coutp = round_up(Cout, BN), small = CinReal <= 4;
    size_t total = dua_conv3x3_packed_floats(Cout, CinReal);
    hipLaunchKernelGGL(k_pack_conv, stream_grid(total, 256), 256, 0, as_stream(stream), w, wp, Cout, CinReal, coutp, small);
    return dua_check_launch("dua_pack_conv3x3_weight");
}

extern "C" int dua_conv3x3(const float* x, const float* wp, const float* bias, const float* residual, float* y, int N, int H,
                           int W, int Cin, int Cout, int nchw_out, void* stream) {
    DUA_REQUIRE(x && wp && y, "dua_conv3x3: null pointer");
    DUA_REQUIRE(N > 0 && H > 0 && W > 0 && Cout > 0, "dua_conv3x3: bad sizes");
    DUA_REQUIRE(Cin == 4 || (Cin > 4 && Cin % BK == 0), "dua_conv3x3: Cin must be 4 (padded small-C form) or a multiple of 32 (got %d)", Cin);
    DUA_REQUIRE((long)N * H * W < (1L << 31), "dua_conv3x3: too many pixels");
    TileP p{};
    p.A1 = x; p.B = wp; p.bias = bias; p.res = residual; p.C = y;
    p.M = N * H * W; p.N = Cout; p.ldc = Cout; p.ldr = Cout; p.alpha = 1.0f;
    p.H = H; p.W = W; p.Cin = Cin; p.nchw_out = nchw_out; p.batch_inner = 1;
    return Cin == 4 ? launch_tile<MODE_CONV_SMALLC, true>(p, 1, as_stream(stream))
                    : launch_tile<MODE_CONV, true>(p, 1, as_stream(stream));
}
