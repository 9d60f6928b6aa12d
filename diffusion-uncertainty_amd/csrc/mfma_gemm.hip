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
    float* colsum;                 // optional [ceil(M/64)][2][N]: per 64-row block column sums / sums of squares of C
};

__device__ __forceinline__ float gelu_f(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752f)); }

template <int MODE, bool TRANSB, bool SILU_IN, bool RAGGED>
__global__ __launch_bounds__(256, 2) void k_tile(TileP p) {
    __shared__ __attribute__((aligned(16))) float lds[2 * 2 * TILE_F];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    // XCD-aware tile order: workgroups that share an XCD (equal id % 8) get a contiguous run of tiles, so the
    // A rows / halo rows and the weight panel they share stay in that XCD's L2.
    const int nwg = gridDim.x;
    int lid = blockIdx.x;
    // Stagger: two workgroups share a CU and would otherwise run their barrier / staging sections in lockstep, leaving
    // the matrix pipe idle during both.  The second resident generation of the launch starts half a K step late, which
    // the slots then keep (speed only; any placement gives the same result).
    if (blockIdx.x >= 256 && blockIdx.x < 512) __builtin_amdgcn_s_sleep(64);
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
            const int m = min(m0 + r0 + 32 * i, p.M - 1);          // rows past M: clamped, never stored
            const int hw = p.H * p.W, n = m / hw, rem = m - n * hw;
            py[i] = rem / p.W;
            px[i] = rem - py[i] * p.W;
            pbase[i] = (long)m * p.Cin;
        }
    }

    // All loads are unconditional (clamped addresses + select) so the compiler never branches around a load:
    // rows past M / N are clamped to the last valid row (their results are never stored), conv padding taps
    // read the centre pixel and are zeroed by a select.  Only a ragged K tail takes the predicated slow path.
    // The loaders return values (named registers in the caller): arrays captured by reference end up in scratch.
    auto ragged4 = [&](const float* q, int k) -> float4 {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k + 3 < K) v = *(const float4*)q;
        else if (k < K) { v.x = q[0]; if (k + 1 < K) v.y = q[1]; if (k + 2 < K) v.z = q[2]; }
        return v;
    };
    // Staging pointers are carried across K steps and advanced by 32 floats per step; they are recomputed only at a
    // wave-uniform boundary (new 3x3 tap / switch to the second K-concat source), which keeps the per-step address
    // VALU work to a handful of 64-bit adds.  LOAD_TILES must be called with kt = 0, 1, 2, ... in order.
    unsigned amask = 0;       // conv: bit i = tap of staged row i is inside the image (applied at ds_write time)
    const float* pa[4];
    const float* pb[4];
    int a_left = 0;           // K steps left before the A pointers must be recomputed
    int tap_next = 0;         // conv: next tap to set up
    auto setup_a = [&](int kt) {
        if (MODE == MODE_GEMM) {
            const int k = kt * BK + 4 * c4;
            const bool second = p.K2 != 0 && kt * BK >= p.K1;          // host guarantees K1 % 32 == 0 when K2 != 0
            const float* src = second ? A2 : A1;
            const long ld = second ? p.lda2 : p.lda1;
            const int kk = second ? k - p.K1 : k;
#pragma unroll
            for (int i = 0; i < 4; ++i) pa[i] = src + (long)min(m0 + r0 + 32 * i, p.M - 1) * ld + kk;
            a_left = (p.K2 != 0 && !second) ? p.K1 / BK - kt : 0x7fffffff;
        } else if (MODE == MODE_CONV) {
            const int tap = tap_next++;
            const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
            const long off = ((long)dy * p.W + dx) * p.Cin;
            amask = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const bool ok = (unsigned)(py[i] + dy) < (unsigned)p.H && (unsigned)(px[i] + dx) < (unsigned)p.W;
                amask |= (unsigned)ok << i;
                pa[i] = A1 + pbase[i] + (ok ? off : 0L) + 4 * c4;
            }
            a_left = p.Cin / BK;
        } else {   // small-Cin: float4 column = one tap, 4 channels; two K steps, both set up from scratch
            const int tap = kt * 8 + c4;
            const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
            const long off = ((long)dy * p.W + dx) * 4;
            amask = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const bool ok = tap < 9 && (unsigned)(py[i] + dy) < (unsigned)p.H && (unsigned)(px[i] + dx) < (unsigned)p.W;
                amask |= (unsigned)ok << i;
                pa[i] = A1 + pbase[i] + (ok ? off : 0L);
            }
            a_left = 1;
        }
    };
    {   // B pointers never need a re-setup
        if (TRANSB) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (MODE != MODE_GEMM) pb[i] = Bp + (long)(n0 + r0 + 32 * i) * K + 4 * c4;          // packed + padded weights
                else pb[i] = Bp + (long)min(n0 + r0 + 32 * i, p.N - 1) * p.ldb + 4 * c4;
            }
        } else {   // B is [K][N]: thread -> (k row kr + 8 i, float4 column n4); N % 4 == 0 checked on host
#pragma unroll
            for (int i = 0; i < 4; ++i)
                pb[i] = Bp + (long)min((tid >> 5) + 8 * i, K - 1) * p.ldb + min(n0 + 4 * (tid & 31), p.N - 4);
        }
    }
    auto load_a = [&](int kt, int i, bool ragged) -> float4 {
        const float* q = pa[i];
        pa[i] = q + BK;
        if (MODE == MODE_GEMM && ragged) return ragged4(q, kt * BK + 4 * c4);
        return *(const float4*)q;
    };
    auto load_b = [&](int kt, int i, bool ragged) -> float4 {
        const float* q = pb[i];
        if (TRANSB) {
            pb[i] = q + BK;
            if (MODE == MODE_GEMM && ragged) return ragged4(q, kt * BK + 4 * c4);
            return *(const float4*)q;
        } else {
            // rows past K (ragged last step) re-read row K-1 and are zeroed at ds_write time
            const int knext = (kt + 1) * BK + (tid >> 5) + 8 * i;
            pb[i] = (!ragged || knext < K) ? q + (long)BK * p.ldb : q;
            return *(const float4*)q;
        }
    };
    // Fix-ups (zero padding, SiLU on the embedding input, K-tail rows) are applied here, at ds_write time, so that
    // no instruction depends on a global load until its data has had most of a K step to land.
    auto store_a = [&](int buf, int i, float4 v) {
        if (MODE != MODE_GEMM) {
            const bool ok = (amask >> i) & 1u;
            v = make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
        } else if (SILU_IN) {
            v = make_float4(silu_f(v.x), silu_f(v.y), silu_f(v.z), silu_f(v.w));
        }
        *(float4*)(lds + buf * 2 * TILE_F + (r0 + 32 * i) * LDT + 4 * c4) = v;
    };
    auto store_b = [&](int buf, int i, float4 v, int kt) {
        float* lb = lds + buf * 2 * TILE_F + TILE_F;
        if (TRANSB) *(float4*)(lb + (r0 + 32 * i) * LDT + 4 * c4) = v;
        else {
            const bool ok = kt * BK + (tid >> 5) + 8 * i < K;
            v = make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
            *(float4*)(lb + ((tid >> 5) + 8 * i) * LDN + 4 * (tid & 31)) = v;
        }
    };
#define LOAD_TILES_(kt, T)                                                                                  \
    if (a_left == 0) setup_a(kt);                                                                           \
    --a_left;                                                                                               \
    ra0 = load_a(kt, 0, T); ra1 = load_a(kt, 1, T); ra2 = load_a(kt, 2, T); ra3 = load_a(kt, 3, T);         \
    rb0 = load_b(kt, 0, T); rb1 = load_b(kt, 1, T); rb2 = load_b(kt, 2, T); rb3 = load_b(kt, 3, T);
    // RAGGED (K % 32 != 0, U-ViT attention / patch embedding only) is a separate kernel variant whose loads are all
    // predicated; the common variant has straight-line unconditional loads and no branch in the K loop.
#define LOAD_TILES(kt) LOAD_TILES_(kt, RAGGED)
#define STORE_TILES(buf, kt)                                                                        \
    store_a(buf, 0, ra0); store_a(buf, 1, ra1); store_a(buf, 2, ra2); store_a(buf, 3, ra3);         \
    store_b(buf, 0, rb0, kt); store_b(buf, 1, rb1, kt); store_b(buf, 2, rb2, kt); store_b(buf, 3, rb3, kt);
    float4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int lr = lane & 31, kh = lane >> 5;

    // Fragment reads for k-group g of the tile in LDS buffer `buf` (see header comment for the k permutation).
    struct Frag { float4 a0, a1, b0, b1; };
    auto read_frag = [&](int buf, int g) -> Frag {
        Frag f;
        const float* la = lds + buf * 2 * TILE_F + (wm * 64 + lr) * LDT + 4 * kh + 8 * g;
        const float* lb = lds + buf * 2 * TILE_F + TILE_F;
        f.a0 = *(const float4*)la;
        f.a1 = *(const float4*)(la + 32 * LDT);
        if (TRANSB) {
            const float* q = lb + (wn * 64 + lr) * LDT + 4 * kh + 8 * g;
            f.b0 = *(const float4*)q;
            f.b1 = *(const float4*)(q + 32 * LDT);
        } else {
            const float* q = lb + (8 * g + 4 * kh) * LDN + wn * 64 + lr;
            f.b0 = make_float4(q[0], q[LDN], q[2 * LDN], q[3 * LDN]);
            f.b1 = make_float4(q[32], q[LDN + 32], q[2 * LDN + 32], q[3 * LDN + 32]);
        }
        return f;
    };
#define MFMA4(AV0, AV1, BV0, BV1)                                                          \
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(AV0, BV0, acc[0][0], 0, 0, 0);        \
    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(AV0, BV1, acc[0][1], 0, 0, 0);        \
    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(AV1, BV0, acc[1][0], 0, 0, 0);        \
    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(AV1, BV1, acc[1][1], 0, 0, 0);
#define MFMA16(F)                                   \
    MFMA4(F.a0.x, F.a1.x, F.b0.x, F.b1.x)           \
    MFMA4(F.a0.y, F.a1.y, F.b0.y, F.b1.y)           \
    MFMA4(F.a0.z, F.a1.z, F.b0.z, F.b1.z)           \
    MFMA4(F.a0.w, F.a1.w, F.b0.w, F.b1.w)

    // Software pipeline, one barrier per 32-deep K step, everything else interleaved with the MFMA stream:
    //   [MFMA g0 | read g1] [MFMA g1 | read g2] [write tile kt+1 -> other buffer]
    //   [MFMA g2 | read g3] lgkmcnt(0) + barrier [MFMA g3 | read g0 of tile kt+1 | issue global loads of tile kt+2]
    // The barrier orders (a) every wave's reads of this tile's buffer before the next step's writes into it and
    // (b) every wave's writes of tile kt+1 before anyone reads it.
    LOAD_TILES(0)
    STORE_TILES(0, 0)
    if (nk > 1) { LOAD_TILES(1) }
    __syncthreads();
    Frag f0 = read_frag(0, 0), f1;
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        f1 = read_frag(buf, 1);
        MFMA16(f0)
        f0 = read_frag(buf, 2);
        MFMA16(f1)
        if (kt + 1 < nk) { STORE_TILES(buf ^ 1, kt + 1) }
        f1 = read_frag(buf, 3);
        MFMA16(f0)
        __syncthreads();
        if (kt + 1 < nk) f0 = read_frag(buf ^ 1, 0);
        if (kt + 2 < nk) { LOAD_TILES(kt + 2) }      // staging registers are free again: the writes above have retired
        MFMA16(f1)
    }

    // ---- epilogue.  Fast path: each wave re-lays its 32 x 64 half tile as rows in a private LDS patch (the K loop's
    // last barrier has retired every read of the staging buffers) and moves whole rows, 16 B per lane, 256 contiguous
    // bytes per row, for the residual read and the output store.
    if (!(MODE != MODE_GEMM && p.nchw_out) && (p.N & 3) == 0 && (p.ldc & 3) == 0 && (!res || (p.ldr & 3) == 0)) {
        float* patch = reinterpret_cast<float*>(lds) + wave * (32 * 64);
        const int c4e = lane & 15, rre = lane >> 4;
        const int colq = n0 + wn * 64 + 4 * c4e;
        float4 bq = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.bias && colq < p.N) bq = *(const float4*)(p.bias + colq);
        float4 cs = make_float4(0.f, 0.f, 0.f, 0.f), cq = cs;      // this lane's column sums / sums of squares
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    patch[((r & 3) + 8 * (r >> 2) + 4 * kh) * 64 + ni * 32 + lr] = acc[mi][ni][r];
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int rl = it * 4 + rre;
                const int row = m0 + wm * 64 + mi * 32 + rl;
                float4 v = *(const float4*)(patch + rl * 64 + 4 * c4e);
                if (row < p.M && colq < p.N) {
                    v.x = v.x * p.alpha + bq.x; v.y = v.y * p.alpha + bq.y; v.z = v.z * p.alpha + bq.z; v.w = v.w * p.alpha + bq.w;
                    if (p.act_out == 2) { v.x = gelu_f(v.x); v.y = gelu_f(v.y); v.z = gelu_f(v.z); v.w = gelu_f(v.w); }
                    if (res) {
                        const float4 q = *(const float4*)(res + (long)row * p.ldr + colq);
                        v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
                    }
                    *(float4*)(Cp + (long)row * p.ldc + colq) = v;
                    cs.x += v.x; cs.y += v.y; cs.z += v.z; cs.w += v.w;
                    cq.x += v.x * v.x; cq.y += v.y * v.y; cq.z += v.z * v.z; cq.w += v.w * v.w;
                }
            }
        }
        if (p.colsum) {   // GroupNorm statistics of the NEXT layer: fold the 4 row lanes, one float4 store per column quad
            float t[8] = {cs.x, cs.y, cs.z, cs.w, cq.x, cq.y, cq.z, cq.w};
#pragma unroll
            for (int j = 0; j < 8; ++j) { t[j] += __shfl_xor(t[j], 16); t[j] += __shfl_xor(t[j], 32); }
            if (rre == 0 && colq < p.N && m0 + wm * 64 < p.M) {
                float* o = p.colsum + (long)((m0 + wm * 64) >> 6) * 2 * p.N + colq;
                *(float4*)o = make_float4(t[0], t[1], t[2], t[3]);
                *(float4*)(o + p.N) = make_float4(t[4], t[5], t[6], t[7]);
            }
        }
        return;
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

template <int MODE, bool TRANSB, bool SILU_IN = false, bool RAGGED = false>
static int launch_tile(TileP& p, int batch, hipStream_t s) {
    p.ntm = (p.M + BM - 1) / BM;
    p.ntn = (p.N + BN - 1) / BN;
    dim3 grid(p.ntm * p.ntn, batch);
    hipLaunchKernelGGL((k_tile<MODE, TRANSB, SILU_IN, RAGGED>), grid, 256, 0, s, p);
    return dua_check_launch("dua tile kernel");
}

extern "C" int dua_gemm(const dua_gemm_args* a, void* stream) {
    DUA_REQUIRE(a && a->A1 && a->B && a->C, "dua_gemm: null pointer");
    DUA_REQUIRE(a->M > 0 && a->N > 0 && a->K1 > 0 && a->K2 >= 0 && a->batch >= 1 && a->batch_inner >= 1, "dua_gemm: bad sizes");
    DUA_REQUIRE(a->lda1 % 4 == 0 && a->ldb % 4 == 0, "dua_gemm: leading dims must be multiples of 4 floats (K itself may be ragged)");
    DUA_REQUIRE((a->K2 == 0) == (a->A2 == nullptr), "dua_gemm: A2/K2 mismatch");
    DUA_REQUIRE(a->K2 == 0 || (a->K1 % BK == 0 && a->lda2 % 4 == 0), "dua_gemm: K1 must be a multiple of 32 for a two-source A");
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
    p.colsum = a->colsum;
    DUA_REQUIRE(!a->colsum || (a->batch == 1 && a->N % 4 == 0 && a->ldc == a->N && (!a->residual || a->ldr % 4 == 0)),
                "dua_gemm: colsum needs batch 1, N %% 4 == 0 and a dense row-major C");
    DUA_REQUIRE(a->act_in == 0 || (a->act_in == 1 && a->transB), "dua_gemm: act_in must be 0, or 1 (SiLU) with transB=1");
    const bool ragged = (a->K1 + a->K2) % BK != 0;
    hipStream_t st = as_stream(stream);
    if (a->act_in == 1) return ragged ? launch_tile<MODE_GEMM, true, true, true>(p, a->batch, st) : launch_tile<MODE_GEMM, true, true, false>(p, a->batch, st);
    if (a->transB) return ragged ? launch_tile<MODE_GEMM, true, false, true>(p, a->batch, st) : launch_tile<MODE_GEMM, true, false, false>(p, a->batch, st);
    return ragged ? launch_tile<MODE_GEMM, false, false, true>(p, a->batch, st) : launch_tile<MODE_GEMM, false, false, false>(p, a->batch, st);
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
                           int W, int Cin, int Cout, int nchw_out, float* colsum, void* stream) {
    DUA_REQUIRE(!colsum || (!nchw_out && Cout % 4 == 0), "dua_conv3x3: colsum needs an NHWC output with Cout %% 4 == 0");
    DUA_REQUIRE(x && wp && y, "dua_conv3x3: null pointer");
    DUA_REQUIRE(N > 0 && H > 0 && W > 0 && Cout > 0, "dua_conv3x3: bad sizes");
    DUA_REQUIRE(Cin == 4 || (Cin > 4 && Cin % BK == 0), "dua_conv3x3: Cin must be 4 (padded small-C form) or a multiple of 32 (got %d)", Cin);
    DUA_REQUIRE((long)N * H * W < (1L << 31), "dua_conv3x3: too many pixels");
    TileP p{};
    p.A1 = x; p.B = wp; p.bias = bias; p.res = residual; p.C = y;
    p.M = N * H * W; p.N = Cout; p.ldc = Cout; p.ldr = Cout; p.alpha = 1.0f;
    p.H = H; p.W = W; p.Cin = Cin; p.nchw_out = nchw_out; p.batch_inner = 1; p.colsum = colsum;
    return Cin == 4 ? launch_tile<MODE_CONV_SMALLC, true>(p, 1, as_stream(stream))
                    : launch_tile<MODE_CONV, true>(p, 1, as_stream(stream));
}
