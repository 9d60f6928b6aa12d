// Split-bf16 ("bf16x3") tile engine: fp32 GEMM / 3x3 convolution evaluated on the bf16 matrix cores.
//
// Every fp32 operand is split a = hi + lo with hi = bf16(a) (RNE) and lo = bf16(a - hi); the product is
// accumulated in fp32 as  hi*hi + hi*lo + lo*hi  with three v_mfma_f32_32x32x16_bf16 per tile and 16-deep K step
// (the dropped lo*lo term is 2^-18 relative).  Per-product relative error is about 2^-17 (1e-5), i.e. between
// fp16 autocast (what the reference's GPU scripts run) and exact fp32; tests hold the model output
// to the north star's 1e-3 per pixel against the fp32 CPU oracle.
//
// Same structure as mfma_gemm.hip: 128x128 block tile, 4 waves x (2x2) 32x32 MFMA tiles, 32-deep K step, staging
// global -> registers -> LDS with the im2col gather / zero padding / SiLU / split applied at ds_write time, one
// barrier per K step, software-pipelined.  Activations (A) are fp32 in HBM and split on the fly; weights (B) are
// pre-split once into two bf16 planes [Npad128][K] by dua_pack_*_bf16x3.  LDS rows are 32 bf16 = four 16-byte chunks;
// chunk c of row r is stored at chunk position c ^ ((r >> 2) & 3): the 16-lane groups of ds_read_b128 (16 rows, one
// chunk column) then land on 16 distinct 16-byte slots and the 8-lane groups of ds_write_b128 (2 rows x 4 chunks) on 8.
#include "common.h"
#include "dconv.h"
#include "hgemm.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
// 16-deep 32x32 MFMA on either 2-byte operand type (the split planes are bf16 in the bf16x3 format, fp16 in f16x3)
__device__ __forceinline__ f32x16 mfma16(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x16 mfma16(h16x8 a, h16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
template <bool F16> struct frag8 { typedef bf16x8 type; };
template <> struct frag8<true> { typedef h16x8 type; };

#define BM 128
#define BN 128
#define BK 32
#define LDH 32                       // LDS row pitch in bf16 elements (64 bytes, no padding; XOR-swizzled chunks)
#define PLANE (BM * LDH)             // bf16 elements per plane tile (5120 -> 10 KB)
#define BUF (4 * PLANE)              // A_hi, A_lo, B_hi, B_lo

enum { XMODE_GEMM = 0, XMODE_CONV = 1 };

struct XTileP {
    const float *A1, *A2;
    const __bf16 *Ahi, *Alo;         // PRESPLIT conv input: two bf16 NHWC planes written by dua_groupnorm_apply_split
    const __bf16 *Bhi, *Blo;         // pre-split weights [Npad][K]
    const float* Bf;                 // BSPLIT: fp32 B [N][K] (row-major, ldb), split on the fly like A
    long ldb;
    long sA, sAi, sB, sBi, sC, sCi;  // batch strides (outer / inner), floats
    int batch_inner;
    const float *bias, *res;
    float* C;
    int M, N, K1, K2;
    long lda1, lda2, ldc, ldr;
    int act_out;
    float alpha;
    int H, W, Cin, nchw_out;
    int ntm, ntn;
    float* colsum;                   // optional [ceil(M/64)][2][N] column sums / sums of squares of C per 64-row block
    const float* Cscale;             // F16 (f16x3): per weight row 2^-s of the packed planes (padded to the tile grid)
    float a_scale;                   // F16: power of two applied to A before the fp16 split (alpha carries its inverse)
};

__device__ __forceinline__ float gelu_x(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752f)); }

__device__ __forceinline__ void split4(float4 v, bf16x4& hi, bf16x4& lo) {
    float f[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        __bf16 h = (__bf16)f[j];
        hi[j] = h;
        lo[j] = (__bf16)(f[j] - (float)h);
    }
}

// f16x3 split of four values scaled by sc: hi = fp16(s), lo = fp16(s - hi).  hi saturates at +-65504 and lo then takes the
// remainder (also saturating): out-of-range operands degrade gracefully instead of turning into inf.
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
// The f16x3 range check counts in REGISTERS (n_sat / n_sub, reported once at the end of the kernel): an atomic inside the K loop - even
// one that never executes - is a vector-memory operation behind a branch, and the compiler then waits for vmcnt(0) in front of every
// staged operand, i.e. for the prefetch it has just issued.
// Four operations per element (the K loop of the skip GEMMs is VALU-issue bound: the plain C form compiled to 7.5): clamp, packed RNE
// conversion of two hi halves, clamp of the remainder's minuend, and ONE v_fma_mix that forms s - hi and rounds it to fp16 (s - hi is
// exact in fp32, so the single rounding equals the two-step form bit for bit).
__device__ __forceinline__ void split4h(float4 v, float sc, h16x4& hi, h16x4& lo, bool rc_on, unsigned& n_sat, unsigned& n_sub) {
    const float f[4] = {v.x * sc, v.y * sc, v.z * sc, v.w * sc};
    if (rc_on) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float af = fabsf(f[j]);
            n_sat += af > 65504.f ? 1u : 0u;
            n_sub += (af > 0.f && af < 6.103515625e-05f) ? 1u : 0u;
        }
    }
    typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
    unsigned hp[2], lp[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const float c0 = __builtin_amdgcn_fmed3f(f[2 * j], -65504.f, 65504.f), c1 = __builtin_amdgcn_fmed3f(f[2 * j + 1], -65504.f, 65504.f);
        const h16x2 h2 = {(_Float16)c0, (_Float16)c1};                      // (one v_cvt_pk_f16_f32)
        hp[j] = __builtin_bit_cast(unsigned, h2);
        // the remainder stays finite up to twice the range (hi saturated, lo carries the rest), as before
        const float g0 = __builtin_amdgcn_fmed3f(f[2 * j], -131008.f, 131008.f), g1 = __builtin_amdgcn_fmed3f(f[2 * j + 1], -131008.f, 131008.f);
        unsigned d;
        asm("v_fma_mixlo_f16 %0, %1, 1.0, -%2 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(d) : "v"(g0), "v"(hp[j]));
        asm("v_fma_mixhi_f16 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(d) : "v"(g1), "v"(hp[j]));
        lp[j] = d;
    }
    typedef unsigned u32x2_ __attribute__((ext_vector_type(2)));
    hi = __builtin_bit_cast(h16x4, (u32x2_){hp[0], hp[1]});
    lo = __builtin_bit_cast(h16x4, (u32x2_){lp[0], lp[1]});
}

template <int MODE, bool SILU_IN, bool PRESPLIT, bool BSPLIT, bool F16 = false>
__global__ __launch_bounds__(256, 2) void k_xtile(XTileP p) {
    __shared__ __attribute__((aligned(16))) __bf16 lds[2 * BUF];      // 80 KB: two workgroups fill a CU's 160 KB
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int nwg = gridDim.x;
    int lid = blockIdx.x;
    if (blockIdx.x >= 256 && blockIdx.x < 512) __builtin_amdgcn_s_sleep(16);
    {
        int q = nwg >> 3, r = nwg & 7, xcd = lid & 7, idx = lid >> 3;
        lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tn = lid % p.ntn, tm = lid / p.ntn;
    const int m0 = tm * BM, n0 = tn * BN;
    const float *A1 = p.A1, *A2 = p.A2, *Bf = p.Bf;
    float* Cp = p.C;
    if (BSPLIT) {   // batched activation x activation products (attention): blockIdx.y = outer*batch_inner + inner
        const int z = blockIdx.y, zo = z / p.batch_inner, zi = z % p.batch_inner;
        A1 += zo * p.sA + zi * p.sAi;
        Bf += zo * p.sB + zi * p.sBi;
        Cp += zo * p.sC + zi * p.sCi;
    }

    const int c4 = tid & 7, r0 = tid >> 3;
    const int K = (MODE == XMODE_GEMM) ? p.K1 + p.K2 : 9 * p.Cin;
    const int nk = K / BK;

    int py[4], px[4];
    long pbase[4];
    if (MODE == XMODE_CONV) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = min(m0 + r0 + 32 * i, p.M - 1);
            const int hw = p.H * p.W, n = m / hw, rem = m - n * hw;
            py[i] = rem / p.W;
            px[i] = rem - py[i] * p.W;
            pbase[i] = (long)m * p.Cin;
        }
    }

    // ---- PRESPLIT: A tile per plane = 128 rows x 32 bf16 = 512 chunks of 16 B; thread -> rows brow, brow + 64
    int qy[2], qx[2];
    long qbase[2];
    if (PRESPLIT) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int m = min(m0 + (tid >> 2) + 64 * j, p.M - 1);
            const int hw = p.H * p.W, n = m / hw, rem = m - n * hw;
            qy[j] = rem / p.W;
            qx[j] = rem - qy[j] * p.W;
            qbase[j] = (long)m * p.Cin + 8 * (tid & 3);
        }
    }
    const __bf16* qh[2];
    const __bf16* ql[2];
    unsigned amask = 0xf;
    const float* pa[4];
    int a_left = 0, tap_next = 0, cb_next = 0;
    unsigned okbits[4] = {0, 0, 0, 0}, qbits[2] = {0, 0};     // bit t = tap t of the staged row lies inside the image
    if (MODE == XMODE_CONV) {
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int dy = t / 3 - 1, dx = t % 3 - 1;
            if (PRESPLIT) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    qbits[j] |= (unsigned)((unsigned)(qy[j] + dy) < (unsigned)p.H && (unsigned)(qx[j] + dx) < (unsigned)p.W) << t;
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    okbits[i] |= (unsigned)((unsigned)(py[i] + dy) < (unsigned)p.H && (unsigned)(px[i] + dx) < (unsigned)p.W) << t;
            }
        }
    }
    auto setup_a = [&](int kt) {
        if (MODE == XMODE_GEMM) {
            const int k = kt * BK + 4 * c4;
            const bool second = p.K2 != 0 && kt * BK >= p.K1;
            const float* src = second ? A2 : A1;
            const long ld = second ? p.lda2 : p.lda1;
            const int kk = second ? k - p.K1 : k;
#pragma unroll
            for (int i = 0; i < 4; ++i) pa[i] = src + (long)min(m0 + r0 + 32 * i, p.M - 1) * ld + kk;
            a_left = (p.K2 != 0 && !second) ? p.K1 / BK - kt : 0x7fffffff;
        } else {
            // K order: channel-block major, tap minor (k = cb*288 + tap*32 + ci) - see mfma_gemm.hip
            const int tap = tap_next, cb = cb_next;
            if (++tap_next == 9) { tap_next = 0; ++cb_next; }
            const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
            const long off = ((long)dy * p.W + dx) * p.Cin;
            amask = 0;
            if (PRESPLIT) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const bool ok = (qbits[j] >> tap) & 1u;
                    amask |= (unsigned)ok << j;
                    const long o = qbase[j] + (ok ? off : 0L) + cb * BK;
                    qh[j] = p.Ahi + o;
                    ql[j] = p.Alo + o;
                }
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const bool ok = (okbits[i] >> tap) & 1u;
                    amask |= (unsigned)ok << i;
                    pa[i] = A1 + pbase[i] + (ok ? off : 0L) + cb * BK + 4 * c4;
                }
            }
            a_left = 1;
        }
    };
    // B: each plane tile is 128 rows x 32 bf16 = 512 chunks of 16 B; thread -> chunks tid and tid + 256
    const int brow = tid >> 2, bch = tid & 3;
    const int bsw = bch ^ ((brow >> 2) & 3);          // swizzled chunk position (rows brow and brow + 64 share it)
    const __bf16* pbh[2];
    const __bf16* pbl[2];
    const float* pbf[4];
    if (!BSPLIT) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (F16) {         // f16x3 weights: stage-contiguous packing (n0 is a multiple of 128: one row tile)
                const long o = (long)(n0 >> 7) * (K / BK) * 8192 + (brow + 64 * j) * 32 + 8 * bch;
                pbh[j] = p.Bhi + o;
                pbl[j] = p.Bhi + o + 4096;
            } else {
                const long o = (long)(n0 + brow + 64 * j) * K + 8 * bch;          // planes are padded to 128 rows
                pbh[j] = p.Bhi + o;
                pbl[j] = p.Blo + o;
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) pbf[i] = Bf + (long)min(n0 + r0 + 32 * i, p.N - 1) * p.ldb + 4 * c4;
    }

    // Two register stages: with ~2k-cycle K steps a global load needs more than one step to land, so tile kt+3 is
    // requested while tile kt+1 is written to LDS (stage (kt+1)&1 is free again right after that write).
    // A registers: 4 x float4 (fp32 rows, split at ds_write time) or, PRESPLIT, 4 x uint4 (hi, hi, lo, lo chunks)
    struct Stage {
        uint4 a0, a1, a2, a3;
        uint4 bh0, bh1, bl0, bl1;        // pre-split weight chunks
        unsigned mask;                   // conv: in-image bits of the staged rows, captured at load time
    };
    // `live` = false (past the last K step): the loads STILL issue - from the last valid tile - and nothing advances.  Unconditional
    // loads keep the K loop straight-line for the compiler's s_waitcnt bookkeeping: behind `if (kt + 3 < nk)` it had to assume that the
    // other stage's prefetch was not issued and waited for vmcnt(0) in front of every staged operand, i.e. for the loads requested
    // one step - not two - earlier.
    auto xload = [&](Stage& S, int kt, bool live) {
        if (live) {
            if (a_left == 0) setup_a(kt);
            --a_left;
        }
        const int back = live ? 0 : -BK;                       // (elements; the pointers stand one tile past the last loaded one)
        S.mask = amask;
        if (PRESPLIT) {
            S.a0 = *(const uint4*)(qh[0] + back); S.a1 = *(const uint4*)(qh[1] + back); S.a2 = *(const uint4*)(ql[0] + back); S.a3 = *(const uint4*)(ql[1] + back);
            if (live) { qh[0] += BK; qh[1] += BK; ql[0] += BK; ql[1] += BK; }
        } else {
            S.a0 = *(const uint4*)(pa[0] + back); S.a1 = *(const uint4*)(pa[1] + back); S.a2 = *(const uint4*)(pa[2] + back); S.a3 = *(const uint4*)(pa[3] + back);
            if (live) { pa[0] += BK; pa[1] += BK; pa[2] += BK; pa[3] += BK; }
        }
        if (!BSPLIT) {
            constexpr int BSTEP = F16 ? 8192 : BK;                                // halves to the next K step
            const int bback = live ? 0 : -BSTEP;
            S.bh0 = *(const uint4*)(pbh[0] + bback); S.bh1 = *(const uint4*)(pbh[1] + bback); S.bl0 = *(const uint4*)(pbl[0] + bback); S.bl1 = *(const uint4*)(pbl[1] + bback);
            if (live) { pbh[0] += BSTEP; pbh[1] += BSTEP; pbl[0] += BSTEP; pbl[1] += BSTEP; }
        } else {
            S.bh0 = *(const uint4*)(pbf[0] + back); S.bh1 = *(const uint4*)(pbf[1] + back); S.bl0 = *(const uint4*)(pbf[2] + back); S.bl1 = *(const uint4*)(pbf[3] + back);
            if (live) { pbf[0] += BK; pbf[1] += BK; pbf[2] += BK; pbf[3] += BK; }
        }
    };
    unsigned n_sat = 0, n_sub = 0;                     // f16x3 range check (debug facility): per-thread counts, reported after the K loop
    const bool rc_on = F16 && dua_range_ctr() != nullptr;
    auto store_a = [&](int buf, int i, uint4 u, unsigned mask) {
        float4 v = make_float4(__uint_as_float(u.x), __uint_as_float(u.y), __uint_as_float(u.z), __uint_as_float(u.w));
        if (MODE == XMODE_CONV) {
            const bool ok = (mask >> i) & 1u;
            v = make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
        } else if (SILU_IN) {
            v = make_float4(silu_f(v.x), silu_f(v.y), silu_f(v.z), silu_f(v.w));
        }
        const int row = r0 + 32 * i;
        __bf16* base = lds + buf * BUF + row * LDH + 8 * ((c4 >> 1) ^ ((row >> 2) & 3)) + 4 * (c4 & 1);
        if (F16) {
            h16x4 hi, lo;
            split4h(v, p.a_scale, hi, lo, rc_on, n_sat, n_sub);
            *(h16x4*)base = hi;
            *(h16x4*)(base + PLANE) = lo;
        } else {
            bf16x4 hi, lo;
            split4(v, hi, lo);
            *(bf16x4*)base = hi;
            *(bf16x4*)(base + PLANE) = lo;
        }
    };
    auto store_b = [&](int buf, int i, uint4 u) {       // BSPLIT: fp32 row chunk -> hi / lo planes of the B tile
        bf16x4 hi, lo;
        split4(make_float4(__uint_as_float(u.x), __uint_as_float(u.y), __uint_as_float(u.z), __uint_as_float(u.w)), hi, lo);
        const int row = r0 + 32 * i;
        __bf16* base = lds + buf * BUF + 2 * PLANE + row * LDH + 8 * ((c4 >> 1) ^ ((row >> 2) & 3)) + 4 * (c4 & 1);
        *(bf16x4*)base = hi;
        *(bf16x4*)(base + PLANE) = lo;
    };
    auto store_pre = [&](int buf, int j, uint4 h, uint4 l, unsigned mask) {
        const bool ok = (mask >> j) & 1u;
        h = make_uint4(ok ? h.x : 0u, ok ? h.y : 0u, ok ? h.z : 0u, ok ? h.w : 0u);
        l = make_uint4(ok ? l.x : 0u, ok ? l.y : 0u, ok ? l.z : 0u, ok ? l.w : 0u);
        __bf16* base = lds + buf * BUF + (brow + 64 * j) * LDH + 8 * bsw;
        *(uint4*)base = h;
        *(uint4*)(base + PLANE) = l;
    };
    auto xstore = [&](const Stage& S, int buf) {
        if (PRESPLIT) { store_pre(buf, 0, S.a0, S.a2, S.mask); store_pre(buf, 1, S.a1, S.a3, S.mask); }
        else { store_a(buf, 0, S.a0, S.mask); store_a(buf, 1, S.a1, S.mask); store_a(buf, 2, S.a2, S.mask); store_a(buf, 3, S.a3, S.mask); }
        if (!BSPLIT) {
            *(uint4*)(lds + buf * BUF + 2 * PLANE + brow * LDH + 8 * bsw) = S.bh0;
            *(uint4*)(lds + buf * BUF + 2 * PLANE + (brow + 64) * LDH + 8 * bsw) = S.bh1;
            *(uint4*)(lds + buf * BUF + 3 * PLANE + brow * LDH + 8 * bsw) = S.bl0;
            *(uint4*)(lds + buf * BUF + 3 * PLANE + (brow + 64) * LDH + 8 * bsw) = S.bl1;
        } else {
            store_b(buf, 0, S.bh0); store_b(buf, 1, S.bh1); store_b(buf, 2, S.bl0); store_b(buf, 3, S.bl1);
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
    typedef typename frag8<F16>::type fr8;
    struct XFrag { fr8 ah0, ah1, al0, al1, bh0, bh1, bl0, bl1; };
    auto read_frag = [&](int buf, int s) -> XFrag {       // s = 16-deep sub-step 0 / 1 of the 32-deep tile
        XFrag f;
        const int cs = 8 * ((2 * s + kh) ^ ((lr >> 2) & 3));     // rows lr, lr + 32 (+64) share the swizzle
        const __bf16* a = lds + buf * BUF + (wm * 64 + lr) * LDH + cs;
        const __bf16* b = lds + buf * BUF + 2 * PLANE + (wn * 64 + lr) * LDH + cs;
        f.ah0 = *(const fr8*)a;
        f.ah1 = *(const fr8*)(a + 32 * LDH);
        f.al0 = *(const fr8*)(a + PLANE);
        f.al1 = *(const fr8*)(a + PLANE + 32 * LDH);
        f.bh0 = *(const fr8*)b;
        f.bh1 = *(const fr8*)(b + 32 * LDH);
        f.bl0 = *(const fr8*)(b + PLANE);
        f.bl1 = *(const fr8*)(b + PLANE + 32 * LDH);
        return f;
    };
#define XM(A, B, ACC) ACC = mfma16(A, B, ACC);
#define XMFMA12(F)                                                                        \
    XM(F.al0, F.bh0, acc[0][0]) XM(F.al0, F.bh1, acc[0][1]) XM(F.al1, F.bh0, acc[1][0]) XM(F.al1, F.bh1, acc[1][1]) \
    XM(F.ah0, F.bl0, acc[0][0]) XM(F.ah0, F.bl1, acc[0][1]) XM(F.ah1, F.bl0, acc[1][0]) XM(F.ah1, F.bl1, acc[1][1]) \
    XM(F.ah0, F.bh0, acc[0][0]) XM(F.ah0, F.bh1, acc[0][1]) XM(F.ah1, F.bh0, acc[1][0]) XM(F.ah1, F.bh1, acc[1][1])

    // pipeline per K step: [MFMA s0 | read s1] [write tile kt+1 from its stage] barrier
    //                      [MFMA s1 | read s0 of tile kt+1 | request tile kt+3 into the stage just written]
    Stage st0, st1;
    xload(st0, 0, true);
    xstore(st0, 0);
    xload(st1, 1, nk > 1);
    xload(st0, 2, nk > 2);
    __syncthreads();
    XFrag f0 = read_frag(0, 0), f1;
#define XSTEP(kt, S)                                                     \
    {                                                                    \
        const int buf = (kt) & 1;                                        \
        f1 = read_frag(buf, 1);                                          \
        XMFMA12(f0)                                                      \
        if ((kt) + 1 < nk) xstore(S, buf ^ 1);                           \
        __syncthreads();                                                 \
        if ((kt) + 1 < nk) f0 = read_frag(buf ^ 1, 0);                   \
        xload(S, (kt) + 3, (kt) + 3 < nk);                               \
        XMFMA12(f1)                                                      \
    }
    int kt2 = 0;
    for (; kt2 + 1 < nk; kt2 += 2) {         // whole pairs of steps, nothing conditional inside (see xload)
        XSTEP(kt2, st1)                      // tile kt+1 lives in st1 (odd tiles), kt+2 in st0
        XSTEP(kt2 + 1, st0)
    }
    if (kt2 < nk) XSTEP(kt2, st1)            // an odd number of K steps

    if (rc_on && (n_sat | n_sub)) {
        unsigned long long* rc_ = dua_range_ctr();
        if (n_sat) atomicAdd(rc_, (unsigned long long)n_sat);
        if (n_sub) atomicAdd(rc_ + 1, (unsigned long long)n_sub);
    }

    // ---- epilogue.  Fast path: each wave re-lays its 32 x 64 half tile as rows in a private LDS patch (the K loop's
    // last barrier has retired every read of the staging buffers) and moves whole rows, 16 B per lane, 256 contiguous
    // bytes per row, for the residual read and the output store.
    if (!(MODE == XMODE_CONV && p.nchw_out) && (p.N & 3) == 0 && (p.ldc & 3) == 0 && (!p.res || (p.ldr & 3) == 0)) {
        float* patch = reinterpret_cast<float*>(lds) + wave * (32 * 64);
        const int c4e = lane & 15, rre = lane >> 4;
        const int colq = n0 + wn * 64 + 4 * c4e;
        float4 bq = make_float4(0.f, 0.f, 0.f, 0.f), sq = make_float4(p.alpha, p.alpha, p.alpha, p.alpha);
        if (p.bias && colq < p.N) bq = *(const float4*)(p.bias + colq);
        if (F16) {                 // (Cscale is padded to the tile grid)
            const float4 w4 = *(const float4*)(p.Cscale + colq);
            sq.x *= w4.x; sq.y *= w4.y; sq.z *= w4.z; sq.w *= w4.w;
        }
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
                    v.x = v.x * sq.x + bq.x; v.y = v.y * sq.y + bq.y; v.z = v.z * sq.z + bq.z; v.w = v.w * sq.w + bq.w;
                    if (p.act_out == 2) { v.x = gelu_x(v.x); v.y = gelu_x(v.y); v.z = gelu_x(v.z); v.w = gelu_x(v.w); }
                    if (p.res) {
                        const float4 q = *(const float4*)(p.res + (long)row * p.ldr + colq);
                        v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
                    }
                    *(float4*)(Cp + (long)row * p.ldc + colq) = v;
                    cs.x += v.x; cs.y += v.y; cs.z += v.z; cs.w += v.w;
                    cq.x += v.x * v.x; cq.y += v.y * v.y; cq.z += v.z * v.z; cq.w += v.w * v.w;
                }
            }
        }
        if (p.colsum) {
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
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const int col = n0 + wn * 64 + ni * 32 + lr;
            if (col >= p.N) continue;
            const float bv = p.bias ? p.bias[col] : 0.f;
            const float sv = F16 ? p.alpha * p.Cscale[col] : p.alpha;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (row >= p.M) continue;
                float v = acc[mi][ni][r] * sv + bv;
                if (p.act_out == 2) v = gelu_x(v);
                if (p.res) v += p.res[(long)row * p.ldr + col];
                if (MODE == XMODE_CONV && p.nchw_out) {
                    int hw = p.H * p.W, n = row / hw, rem = row - n * hw;
                    Cp[((long)n * p.N + col) * hw + rem] = v;
                } else {
                    Cp[(long)row * p.ldc + col] = v;
                }
            }
        }
}

template <int MODE, bool SILU_IN = false, bool PRESPLIT = false, bool BSPLIT = false, bool F16 = false>
static int launch_x(XTileP& p, hipStream_t s, int batch = 1) {
    p.ntm = (p.M + BM - 1) / BM;
    p.ntn = (p.N + BN - 1) / BN;
    if (p.batch_inner < 1) p.batch_inner = 1;
    hipLaunchKernelGGL((k_xtile<MODE, SILU_IN, PRESPLIT, BSPLIT, F16>), dim3(p.ntm * p.ntn, batch), 256, 0, s, p);
    return dua_check_launch("dua bf16x3 tile kernel");
}

// ================================================================================================================
// LDS-DMA variant of the pre-split 3x3 convolution: 256 x 128 block tile, 8 waves (4 x 2, each 64 x 64), 32-deep K
// steps, THREE LDS stages of 48 KB filled by `buffer_load_dwordx4 ... lds` straight from HBM/L2 (no staging
// registers, no ds_write, no padding selects: taps outside the image use an out-of-range buffer offset, which the
// buffer unit returns as zeros).  The XOR swizzle of the LDS image is applied on the per-lane SOURCE chunk (the DMA
// writes lane-linear 1 KB pieces = 16 rows x 64 B).  Tile kt+2 is requested while tile kt is multiplied; a counted
// `s_waitcnt vmcnt(6)` + one raw s_barrier per step publishes tile kt+1.
#define DBM 256

// (DConvP: dconv.h)

// NT = 32-column MFMA tiles per wave: block tile 256 x (64*NT), waves 4(M) x 2(N), wave tile 64 x (32*NT).
//   NT = 2: 256 x 128; activation ring 3 x 32 KB + weight ring 3 x 16 KB (tile kt+2 of both in flight during step kt);
//   NT = 4: 256 x 256 (half the L2->LDS bytes per MFMA); activation ring 3 x 32 KB + weight ring 2 x 32 KB = the whole
//           160 KB of LDS: the activation operand (unique per workgroup, the one that misses L2) is requested two
//           tiles ahead, the L2-resident weight panel one tile ahead.
// Row epilogue of a FULL 256 x (64*NT) tile, specialised at compile time (no per-row predicates, no run-time branches in
// the row loop).  Each wave re-lays one 32 x (32*NT) half of its accumulators as rows in a private LDS patch (pitch +8
// floats: the two half-waves write rows 4 apart and must land on different banks), then moves whole rows, 16 B per lane.
// All residual rows of the half are requested BEFORE the transposition so that their latency runs under it (read one by
// one they cost a full memory round trip per row group: 55k of the 87k epilogue cycles of a 256x256 tile).
#ifndef DUA_A_PREFETCH
#define DUA_A_PREFETCH 1
#endif
#ifndef DUA_DIAG        // timing diagnostics only (wrong results): 1 = no in-loop barrier, 2 = no in-loop DMA, 4 = no in-loop fragment reads
#define DUA_DIAG 0
#endif
// pitch of the epilogue patch rows: 256-wide tiles use the bare 128 floats (8 patches = 128 KB: activation stage 1 stays free)
#define DUA_PATCH_PAD (NT == 4 ? 0 : 8)
template <int ACT, bool RES, bool SPLIT, int FMT = 0>
__device__ __forceinline__ void dconv_row_out(const DConvP& p, float4 w, const float4 q, const float4 bq, const float4 sq, long o, float4& cs, float4& cq) {
    if (FMT == 3) { w.x = w.x * sq.x + bq.x; w.y = w.y * sq.y + bq.y; w.z = w.z * sq.z + bq.z; w.w = w.w * sq.w + bq.w; }
    else { w.x += bq.x; w.y += bq.y; w.z += bq.z; w.w += bq.w; }
    if (ACT == 2) { w.x = gelu_x(w.x); w.y = gelu_x(w.y); w.z = gelu_x(w.z); w.w = gelu_x(w.w); }
    if (RES) { w.x += q.x; w.y += q.y; w.z += q.z; w.w += q.w; }
    if (SPLIT) {               // the consumer is another split-operand GEMM: hand it the operand format directly
        if (FMT == 3) {
            dua_store_f16x3_quad(p.Csplit, (long)p.M * p.N, o, w, p.split_scale);
        } else {
            bf16x4 hi, lo;
            split4(w, hi, lo);
            *(bf16x4*)(p.Csplit + o) = hi;
            *(bf16x4*)(p.Csplit + o + (long)p.M * p.N) = lo;
        }
    } else {
        *(float4*)(p.C + o) = w;
    }
    cs.x += w.x; cs.y += w.y; cs.z += w.z; cs.w += w.w;
    cq.x += w.x * w.x; cq.y += w.y * w.y; cq.z += w.z * w.z; cq.w += w.w * w.w;
}

template <int NT, bool RES, int ACT, bool SPLIT, int FMT = 0>
__device__ __forceinline__ void dconv_rows_full(const DConvP& p, f32x16 (&acc)[2][NT], float* patch, int lane, int row0, int colq,
                                                float4 bq, float4 sq, float4& cs, float4& cq) {
    constexpr int WC = 32 * NT, WCP = WC + DUA_PATCH_PAD, RPI = 64 / (WC / 4), NIT = 32 / RPI;
    const int c4 = lane % (WC / 4), rr = lane / (WC / 4), lr = lane & 31, kh = lane >> 5;
    const long rstep = (long)RPI * p.N;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const float* rd = patch + rr * WCP + 4 * c4;
    float* wr = patch + (4 * kh) * WCP + lr;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        const long rb = (long)(row0 + mi * 32 + rr) * p.N + colq;
        // the residual rows of this half are requested before the transposition: their latency runs under it
        float4 q[RES ? NIT : 1];
        if (RES) {
#pragma unroll
            for (int it = 0; it < NIT; ++it) q[it] = *(const float4*)(p.res + rb + it * rstep);
        }
#pragma unroll
        for (int ni = 0; ni < NT; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) wr[((r & 3) + 8 * (r >> 2)) * WCP + ni * 32] = acc[mi][ni][r];
        // same-wave LDS write -> read: in-order LDS queue, no barrier needed
#pragma unroll
        for (int g = 0; g < NIT; g += 4) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *(const float4*)(rd + (g + u) * RPI * WCP);
#pragma unroll
            for (int u = 0; u < 4; ++u) dconv_row_out<ACT, RES, SPLIT, FMT>(p, v[u], RES ? q[g + u] : z4, bq, sq, rb + (g + u) * rstep, cs, cq);
        }
    }
}

#ifdef DUA_CONV_TIMING   // diagnostic build only (scratch/): shader-clock stamps of the tile phases of every workgroup
__device__ unsigned long long dua_tt[256 * 8 * 4];
extern "C" int dua_debug_tile_times(unsigned long long* host) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(dua_tt), sizeof(unsigned long long) * 256 * 8 * 4) == hipSuccess ? 0 : 1;
}
#define DUA_STAMP(k) if (threadIdx.x == 0 && blockIdx.x < 256 && (ti - slot) / nslot < 8) dua_tt[(blockIdx.x * 8 + (ti - slot) / nslot) * 4 + (k)] = clock64();
#else
#define DUA_STAMP(k)
#endif

// FMT selects the operand format of the two planes (same bytes, same DMA):
//   0 "bf16x3": plane 0 = bf16 hi, plane 1 = bf16 lo; product = hi*hi + hi*lo + lo*hi, three v_mfma_f32_32x32x16_bf16 per 16 k.
//   3 "f16x3":  fp16 hi | lo of the power-of-two-scaled operand, the same loop on v_mfma_f32_32x32x16_f16 (see dua_hip.h).
// (Round 4 removed formats 1 "f16f8" and 2 "f16f6" - fp16 hi + block-scaled fp8 / MXFP6 cross terms - together with their K loops and
// the row-shift form RS they alone used: narrower than the parity modes, and since the Winograd form of the 3x3 convs no longer
// faster than f16x3.  RS is kept as a template parameter that must be false.)
template <int NT, int FMT, bool RS = false, bool KS = false /* split-K work items (DConvP::ksplit): its own instantiation, so that the
                                                               single-pass kernels compile exactly as they did without it */>
__global__ __launch_bounds__(512, 2) void k_xconv_dma(DConvP p) {
#if defined(__HIP_DEVICE_COMPILE__)   // the buffer-resource builtins have no host-side type: the host pass only needs the stub
    constexpr int DBN = 64 * NT;
    // Ring depths (stages) of the activation / weight operand.  With the 256x256 tile the 160 KB of LDS hold five 32 KB
    // stages: the loop (matrix-core bound) keeps the activations two tiles ahead.
    constexpr bool BDEEP = RS;
    constexpr int RA = BDEEP ? 2 : 3, RB = NT == 2 || BDEEP ? 3 : 2;
    constexpr int SA = DBM * LDH;                    // bf16 elements per A plane of a stage
    constexpr int SB = DBN * LDH;
    // RS puts the weight ring first: a fragment row -1 of activation stage 0 then reads (masked) weight bytes, not a negative address
    constexpr int BOFF = RS ? 0 : RA * 2 * SA;       // start of the weight ring
    constexpr int AOFF = RS ? RB * 2 * SB : 0;       // start of the activation ring
    __shared__ __attribute__((aligned(16))) __bf16 lds[RA * 2 * SA + RB * 2 * SB];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    // Persistent workgroups: gridDim.x <= #CUs workgroups walk the tile list.  Workgroups that share an XCD (equal
    // blockIdx % 8) own one contiguous chunk of tiles and sweep it side by side, so concurrently running workgroups
    // of an XCD work on neighbouring tiles (shared halo rows / weight panel in that XCD's L2).  A workgroup moves to
    // its next tile right after issuing its output stores: they drain while the next tile's first DMA is in flight.
    constexpr bool KSPLIT_OK = KS;
    static_assert(!KS || ((FMT == 0 || FMT == 3) && !RS), "split-K is built for the per-tap bf16x3 / f16x3 loops");
    const int S = KSPLIT_OK && p.ksplit > 1 ? p.ksplit : 1;          // split-K parts per tile (see DConvP::ksplit)
    const int ntiles = p.ntm * p.ntn * S;
    const int nk = p.K / BK;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslot = (gridDim.x + 7 - xcd) >> 3;
    const int cq = ntiles >> 3, cr = ntiles & 7;
    const int chunk0 = xcd < cr ? xcd * (cq + 1) : cr * (cq + 1) + (xcd - cr) * cq, chunk_n = cq + (xcd < cr ? 1 : 0);
    // Start-up stagger.  All workgroups run equally long tiles, so left alone they reach their epilogues together: the
    // output / residual traffic of the whole chip (0.5 MB per tile and CU) then hits HBM in one burst while the matrix
    // cores idle, and HBM idles during the K loops.  Delaying the workgroups of an XCD by phase/P of a tile period spreads
    // the epilogues over the period; the phase persists because every tile takes the same time.
    if (p.stagger_phases > 1) {
        const int ph = slot % p.stagger_phases;
        for (int i = 0; i < ph * p.stagger_sleeps; ++i) __builtin_amdgcn_s_sleep(127);
    }
  // ---- activation-side DMA geometry of a tile (its state lives across tiles: the row-shift loop requests the first
  // activation stage of the NEXT tile before the epilogue of the current one).  One wave instruction = 16 rows x 4 chunks
  // of 16 B; wave w stages A row groups 2w, 2w+1 (both planes) and B row groups (NT/2)*w .. (both planes).
  // The buffer descriptors are rebased per tile at the first pixel its taps can touch (tile start minus one image row and
  // one pixel), so the 32-bit buffer offsets stay tiny whatever the size of the activation planes.
  const unsigned OOB = 0xFFFFFFF0u;
  unsigned abase[2] = {0, 0}, abits[2] = {0, 0}, rsy = 0;
  auto ra = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, 0, 0x00020000);
  auto ral = ra;
  int tap = 0, cb = 0;
  bool a_prefetched = false;
  auto a_geom = [&](int m0_) {
    int lane_g = lane;
    asm volatile("" : "+v"(lane_g));
    const int lrow = lane_g >> 2;
    const int csrc = (lane_g & 3) ^ ((lane_g >> 4) & 3);                 // source chunk for this lane's LDS slot
    const int mbase = p.ntaps == 9 ? max(0, m0_ - (p.W + 1)) : m0_;
    rsy = 0;
    const int lw = __builtin_ctz(p.W);
    if (RS && ((p.M >> lw) >> 21) == 0) {
        // row-shift tiles: W is a power of two (it divides 256) and only the y-edge flags are needed: shifts, 24-bit
        // multiplies and one float quotient (exact for image-row indices below 2^21) instead of four 32-bit integer
        // divisions (~3k cycles per tile)
        const float rh = 1.0f / (float)p.H;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int m = min(m0_ + 16 * (2 * wave + j) + lrow, p.M - 1);
            abase[j] = (__umul24((unsigned)(m - mbase), (unsigned)p.Cin) + 8u * csrc) * 2u;
            const int rowq = m >> lw;                                            // image * H + y  (< 2^22: exact in float)
            const int y = rowq - (int)__umul24((unsigned)(int)(((float)rowq + 0.5f) * rh), (unsigned)p.H);
            abits[j] = 0;
            rsy |= (y == 0 ? 1u : 0u) << j | (y == p.H - 1 ? 1u : 0u) << (2 + j);
        }
    } else
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int m = min(m0_ + 16 * (2 * wave + j) + lrow, p.M - 1);
        abase[j] = (unsigned)(((long)(m - mbase) * p.Cin + 8 * csrc) * 2);
        unsigned bits = 1;
        if (p.ntaps == 9) {
            const int hw = p.H * p.W, n = m / hw, rem = m - n * hw;
            const int y = rem / p.W, x = rem - y * p.W;
            bits = 0;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int dy = t / 3 - 1, dx = t % 3 - 1;
                bits |= (unsigned)((unsigned)(y + dy) < (unsigned)p.H && (unsigned)(x + dx) < (unsigned)p.W) << t;
            }
        }
        abits[j] = bits;
        if (RS) rsy |= ((bits >> 1) & 1u ? 0u : 1u) << j | ((bits >> 7) & 1u ? 0u : 1u) << (2 + j);   // taps (-1, 0) / (+1, 0) outside
    }
    const long arem = ((long)p.M - mbase) * p.Cin * 2;                       // bytes from the rebased origin to the plane's end
    const int arec = (p.dbg_norec & 1) ? 0 : (int)(arem < 0x7FFFFFF0L ? arem : 0x7FFFFFF0L);
    ra = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A + (long)mbase * p.Cin), 0, arec, 0x00020000);
    ral = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A + p.a_plane + (long)mbase * p.Cin), 0, arec, 0x00020000);
    tap = 0;
    cb = 0;
  };
  for (int ti = slot; ti < chunk_n; ti += nslot) {
    const int lid = chunk0 + ti;
    const int part = KSPLIT_OK ? lid % S : 0, tl = KSPLIT_OK ? lid / S : lid;       // parts of a tile are neighbours: they share its operands in L2
    const int tn = tl % p.ntn, tm = tl / p.ntn;
    const int m0 = tm * DBM, n0 = tn * DBN;
    DUA_STAMP(0)
    const int lrow = lane >> 2;
    const int csrc = (lane & 3) ^ ((lane >> 4) & 3);
    if (!a_prefetched) a_geom(m0);
    const int kb = KSPLIT_OK && S > 1 ? (int)((long)part * nk / S) : 0;            // first K step of this work item
    const int nkt = KSPLIT_OK && S > 1 ? (int)((long)(part + 1) * nk / S) - kb : nk;
    if (KSPLIT_OK && S > 1) {
        tap = kb % p.ntaps;
        cb = kb / p.ntaps;
    }
    unsigned boff[NT / 2];
    // Every workgroup of the chip streams the SAME weight tile at about the same time: rotate which 16-row group a wave
    // fetches by the workgroup's slot, so that at a given moment the workgroups of an XCD ask different L2 channels.
#ifndef DUA_B_ROT
#define DUA_B_ROT 1
#endif
    const int brot = DUA_B_ROT ? (slot * 5) % (4 * NT) : 0;
    int bgrp[NT / 2];
#pragma unroll
    for (int j = 0; j < NT / 2; ++j) {
        bgrp[j] = ((NT / 2) * wave + j + brot) % (4 * NT);
        if (FMT == 3) {        // f16x3: stage-contiguous packing [row tile of 128][K step][hi | lo][128][32]
            const int r_ = n0 + 16 * bgrp[j] + lrow;
            boff[j] = (unsigned)((((long)(r_ >> 7) * (p.K / BK) + kb) * 16384) + (r_ & 127) * 64 + 16 * csrc);
        } else {
            boff[j] = (unsigned)(((long)(n0 + 16 * bgrp[j] + lrow) * p.K + 8 * csrc + (long)kb * BK) * 2);
        }
    }
    const auto rb = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, (p.dbg_norec & 2) ? 0 : (int)(2 * p.b_plane * 2), 0x00020000);
    const int b_lo = FMT == 3 ? 8192 : (int)(p.b_plane * 2);                 // byte distance hi -> lo weight plane / image
    // The DMA pieces of a tile are issued one at a time so that the caller can spread them between MFMA groups
    // (an LDS-DMA issue holds the wave for ~60-180 cycles).
    unsigned vo0 = 0, vo1 = 0;
    auto dma_setup = [&]() {
        if (RS) {      // one stage per (channel block, dy): the dx = 0 tap's rows; `tap` counts dy + 1 here
            const int toff = (((tap - 1) * p.W) * p.Cin + cb * BK) * 2;
            // rows whose dy neighbour lies outside the image: rsy bit j = row group j at y == 0, bit 2 + j = at y == H - 1
            const unsigned out_ = tap == 0 ? rsy : (tap == 2 ? rsy >> 2 : 0u);
            vo0 = (out_ & 1u) ? OOB : abase[0] + (unsigned)toff;
            vo1 = (out_ & 2u) ? OOB : abase[1] + (unsigned)toff;
            if (++tap == 3) { tap = 0; ++cb; }
            return;
        }
        const int dy = p.ntaps == 9 ? tap / 3 - 1 : 0, dx = p.ntaps == 9 ? tap - (tap / 3) * 3 - 1 : 0;
        const int toff = ((dy * p.W + dx) * p.Cin + cb * BK) * 2;           // byte offset of this (tap, channel block)
        vo0 = ((abits[0] >> tap) & 1u) ? abase[0] + (unsigned)toff : OOB;
        vo1 = ((abits[1] >> tap) & 1u) ? abase[1] + (unsigned)toff : OOB;
        if (++tap == p.ntaps) { tap = 0; ++cb; }
    };
    auto dma_a = [&](int sa, int i) {          // piece i = 0..3 of the activation tile whose offsets dma_setup() prepared
        __bf16* st = lds + AOFF + sa * 2 * SA;
#ifndef DUA_A_AUX
#define DUA_A_AUX 0
#endif
        if (i == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, st + (16 * (2 * wave)) * LDH, 16, vo0, 0, 0, DUA_A_AUX);
        if (i == 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(ral, st + (16 * (2 * wave)) * LDH + SA, 16, vo0, 0, 0, DUA_A_AUX);
        if (i == 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, st + (16 * (2 * wave + 1)) * LDH, 16, vo1, 0, 0, DUA_A_AUX);
        if (i == 3) __builtin_amdgcn_raw_ptr_buffer_load_lds(ral, st + (16 * (2 * wave + 1)) * LDH + SA, 16, vo1, 0, 0, DUA_A_AUX);
    };
    auto dma_b = [&](int sb, int i) {          // piece i = 0..NT-1 of the next weight tile (sequential in K)
        __bf16* st = lds + BOFF + sb * 2 * SB;
#pragma unroll
        for (int j = 0; j < NT / 2; ++j) {
            __bf16* dst = st + (16 * bgrp[j]) * LDH;
#ifndef DUA_B_AUX
#define DUA_B_AUX 0
#endif
            if (i == 2 * j) __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, dst, 16, boff[j], 0, 0, DUA_B_AUX);
            if (i == 2 * j + 1) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, dst + SB, 16, boff[j], b_lo, 0, DUA_B_AUX);
                boff[j] += FMT == 3 ? 16384 : BK * 2;
            }
        }
    };
    auto issue_a = [&](int sa) {
        dma_setup();
#pragma unroll
        for (int i = 0; i < 4; ++i) dma_a(sa, i);
    };
    auto issue_b = [&](int sb) {
#pragma unroll
        for (int i = 0; i < NT; ++i) dma_b(sb, i);
    };

    f32x16 acc[2][NT];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int lr = lane & 31, kh = lane >> 5;
  {
    static_assert(FMT == 0 || FMT == 3, "operand formats: 0 = split bf16 planes, 3 = f16x3 planes");
    static_assert(!RS, "the row-shift loop of the removed f16f8 / f16f6 formats");
    typedef typename frag8<FMT == 3>::type bf16x8;      // FMT 3: the same loop on fp16 planes (v_mfma_f32_32x32x16_f16)
    // Fragment registers of ONE half step (16 of the 32 K columns of a stage); they are refilled in place while the
    // MFMA group that no longer needs them is executing, so the matrix pipe always has a full group queued when the
    // next one's operands arrive (no extra registers: the accumulators already take 128 of the 256):
    //   reload ah           | T1 lo*hi (al, bh)
    //   reload bl           | T2 hi*hi (ah, bh)
    //   reload al, bh (NEXT half step; free since T1 / T2 were issued) | T3 hi*lo (ah, bl)
    // Each reload is issued just before the MFMA group that precedes its consumer, so its LDS latency runs under that
    // group.  The barrier that publishes stage kt+1 therefore sits before T3 of the second half step of step kt.
    bf16x8 ah[2], al[2], bh[NT], bl[NT];
    const int csw0 = 8 * ((0 + kh) ^ ((lr >> 2) & 3)), csw1 = 8 * ((2 + kh) ^ ((lr >> 2) & 3));
    const __bf16* abase_l = lds + (wm * 64 + lr) * LDH;
    const __bf16* bbase_l = lds + BOFF + (wn * 32 * NT + lr) * LDH;
#define FR_A(st, s) (abase_l + (st) * 2 * SA + ((s) ? csw1 : csw0))
#define FR_B(st, s) (bbase_l + (st) * 2 * SB + ((s) ? csw1 : csw0))
#define LD_AH(st, s) { const __bf16* a_ = FR_A(st, s); ah[0] = *(const bf16x8*)a_; ah[1] = *(const bf16x8*)(a_ + 32 * LDH); }
#define LD_AL(st, s) { const __bf16* a_ = FR_A(st, s) + SA; al[0] = *(const bf16x8*)a_; al[1] = *(const bf16x8*)(a_ + 32 * LDH); }
#define LD_BH(st, s) { const __bf16* b_ = FR_B(st, s); _Pragma("unroll") for (int j_ = 0; j_ < NT; ++j_) bh[j_] = *(const bf16x8*)(b_ + 32 * j_ * LDH); }
#define LD_BL(st, s) { const __bf16* b_ = FR_B(st, s) + SB; _Pragma("unroll") for (int j_ = 0; j_ < NT; ++j_) bl[j_] = *(const bf16x8*)(b_ + 32 * j_ * LDH); }
#define MM(AV, BV)                                                                                              \
    _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) _Pragma("unroll") for (int j_ = 0; j_ < NT; ++j_)          \
        acc[i_][j_] = mfma16(AV[i_], BV[j_], acc[i_][j_]);
    // DMA pieces of the step, a few per MFMA group (an LDS-DMA issue holds the wave ~60-180 cycles), front-loaded:
    // NT = 4: [B0..B3 | A0..A3 | - ...]; NT = 2: [A0 A1 | A2 A3 | B0 B1 | - ...]
#define DMA_GROUP()                                                                   \
    _Pragma("unroll") for (int u_ = 0; u_ < (NT == 4 ? 4 : 2); ++u_) {                \
        if (NT == 4) {                                                                \
            if (piece < 4) { if (preB) dma_b(sbn, piece); }                           \
            else if (piece < 8) { if (preA) dma_a(sa2, piece - 4); }                  \
        } else {                                                                      \
            if (piece < 4) { if (preA) dma_a(sa2, piece); }                           \
            else if (piece < 6) { if (preB) dma_b(sbn, piece - 4); }                  \
        }                                                                             \
        ++piece;                                                                      \
    }
#define SBAR() __builtin_amdgcn_sched_barrier(0)

    // In-order vmcnt: when stage kt+1 is published everything older than the pieces that may stay in flight must have landed.
    //   NT = 4: order per step  B(kt+1) x4, A(kt+2) x4  -> wait vmcnt(4) keeps only A(kt+2) outstanding;
    //   NT = 2: order per step  A(kt+2) x4, B(kt+2) x2  -> wait vmcnt(6).
    constexpr int KEEP = NT == 4 ? 4 : 6;
#define DWAIT_TILE(more) \
    if (more) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(KEEP) : "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    issue_a(0);
    issue_b(0);
    if (nkt > 1) {
        issue_a(1);
        if (NT == 2) issue_b(1);
    }
    DWAIT_TILE(nkt > 1)
    __builtin_amdgcn_s_barrier();
    DUA_STAMP(1)
    int sa = 0, sb = 0;
    LD_AL(0, 0)
    LD_BH(0, 0)
    for (int kt = 0; kt < nkt; ++kt) {
        int sa2 = sa + 2;                                // activation stage of tile kt+2
        if (sa2 >= RA) sa2 -= RA;
        int sbn = sb + (RB - 1);                         // weight stage of tile kt+2 (NT = 2) / kt+1 (NT = 4)
        if (sbn >= RB) sbn -= RB;
        const bool preA = kt + 2 < nkt;
        const bool preB = NT == 4 ? kt + 1 < nkt : kt + 2 < nkt;
        if (preA) dma_setup();
        int piece = 0;
        SBAR();
        // ---- K columns 0..15 of the stage.  Every reload is issued one MFMA group AHEAD of its consumer.
        LD_AH(sa, 0) SBAR(); MM(al, bh) SBAR(); DMA_GROUP() SBAR();
        LD_BL(sb, 0) SBAR(); MM(ah, bh) SBAR(); DMA_GROUP() SBAR();
        LD_AL(sa, 1) LD_BH(sb, 1) SBAR(); MM(ah, bl) SBAR(); DMA_GROUP() SBAR();
        // ---- K columns 16..31
        LD_AH(sa, 1) SBAR(); MM(al, bh) SBAR(); DMA_GROUP() SBAR();
        LD_BL(sb, 1) SBAR(); MM(ah, bh) SBAR(); DMA_GROUP() SBAR();
        // every LDS read of stage kt has been issued; let them land, then publish stage kt+1
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        DWAIT_TILE(preA)
        __builtin_amdgcn_s_barrier();
        if (++sa == RA) sa = 0;
        if (++sb == RB) sb = 0;
        SBAR();
        LD_AL(sa, 0) LD_BH(sb, 0) SBAR(); MM(ah, bl) SBAR();   // (after the last step: a harmless read of a stale stage)
    }
#undef FR_A
#undef FR_B
#undef LD_AH
#undef LD_AL
#undef LD_BH
#undef LD_BL
#undef MM
#undef DMA_GROUP
#undef SBAR
#undef DWAIT_TILE

  }
    DUA_STAMP(2)
    // The K loop's last barrier has retired every read of the stages: the row-shift loop requests the first activation stage
    // of this workgroup's NEXT tile now, into stage 1 (outside the epilogue patches), so that its HBM latency runs under the
    // epilogue.  (Issued between the two halves of the epilogue instead: K loop and epilogue both slower, 208k -> 224k per tile.)
    auto prefetch_next = [&]() {
        if constexpr (RS) {
            if (DUA_A_PREFETCH && ti + nslot < chunk_n) {
                a_geom(((chunk0 + ti + nslot) / p.ntn) * DBM);
                dma_setup();
#pragma unroll
                for (int i = 0; i < 4; ++i) dma_a(1, i);
                a_prefetched = true;
            }
        }
    };
    // The epilogue's lane-derived addresses must not be hoisted above the K loop (the loop owns every VGPR: hoisted
    // values end up in scratch, and their reload waits on vmcnt(0), i.e. on the whole DMA pipeline).
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));
    const int lr_e = lane_e & 31, kh_e = lane_e >> 5;
    const bool full_tile = m0 + DBM <= p.M && n0 + DBN <= p.N && !p.nchw_out && !(p.res && (p.act_out == 2 || p.Csplit));
    if (KSPLIT_OK && S > 1) {
        // ---- split-K work item: the scaled accumulators go to this part's slab of the workspace, rows re-laid through a private
        // LDS patch (16 B per lane, whole rows); bias / residual / statistics belong to k_splitk_reduce
        constexpr int WC = 32 * NT;
        float* patch = reinterpret_cast<float*>(lds) + wave * (32 * WC);
        const int c4 = lane_e % (WC / 4), rr = lane_e / (WC / 4);
        constexpr int RPI = 64 / (WC / 4);
        const int colq = n0 + wn * WC + 4 * c4;
        float4 sq = make_float4(1.f, 1.f, 1.f, 1.f);
        if (FMT == 3) {
            sq = *(const float4*)(p.Cscale + colq);
            sq.x *= p.out_scale; sq.y *= p.out_scale; sq.z *= p.out_scale; sq.w *= p.out_scale;
        }
        float* slab = p.kpart + (long)part * p.M * p.N;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
            for (int ni = 0; ni < NT; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    patch[((r & 3) + 8 * (r >> 2) + 4 * kh_e) * WC + ni * 32 + lr_e] = acc[mi][ni][r];
#pragma unroll
            for (int it = 0; it < 32 / RPI; ++it) {
                const int rl = it * RPI + rr;
                const int row = m0 + wm * 64 + mi * 32 + rl;
                float4 v = *(const float4*)(patch + rl * WC + 4 * c4);
                if (row < p.M && colq < p.N) {
                    v.x *= sq.x; v.y *= sq.y; v.z *= sq.z; v.w *= sq.w;
                    *(float4*)(slab + (long)row * p.N + colq) = v;
                }
            }
        }
    } else if (full_tile) {
        constexpr int WC = 32 * NT, WCP = WC + DUA_PATCH_PAD;
        float* patch = reinterpret_cast<float*>(lds) + wave * (32 * WCP);
        const int c4 = lane_e % (WC / 4), rr = lane_e / (WC / 4);
        const int colq = n0 + wn * WC + 4 * c4;
        float4 bq = make_float4(0.f, 0.f, 0.f, 0.f), sq = make_float4(1.f, 1.f, 1.f, 1.f);
        if (p.bias) bq = *(const float4*)(p.bias + colq);
        if (FMT == 3) {
            sq = *(const float4*)(p.Cscale + colq);
            sq.x *= p.out_scale; sq.y *= p.out_scale; sq.z *= p.out_scale; sq.w *= p.out_scale;
        }
        if constexpr (RS) {        // the bias must not queue behind the prefetch in the in-order vmcnt counter
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            prefetch_next();
        }
        float4 cs = make_float4(0.f, 0.f, 0.f, 0.f), cq = cs;
        const int row0 = m0 + wm * 64;
        if (p.Csplit) {
            if (p.act_out == 2) dconv_rows_full<NT, false, 2, true, FMT>(p, acc, patch, lane_e, row0, colq, bq, sq, cs, cq);
            else dconv_rows_full<NT, false, 0, true, FMT>(p, acc, patch, lane_e, row0, colq, bq, sq, cs, cq);
        } else if (p.res) {
            dconv_rows_full<NT, true, 0, false, FMT>(p, acc, patch, lane_e, row0, colq, bq, sq, cs, cq);
        } else if (p.act_out == 2) {
            dconv_rows_full<NT, false, 2, false, FMT>(p, acc, patch, lane_e, row0, colq, bq, sq, cs, cq);
        } else {
            dconv_rows_full<NT, false, 0, false, FMT>(p, acc, patch, lane_e, row0, colq, bq, sq, cs, cq);
        }
        if (p.colsum) {   // GroupNorm statistics of the NEXT layer: fold the row lanes, one float4 store per column quad
            float t[8] = {cs.x, cs.y, cs.z, cs.w, cq.x, cq.y, cq.z, cq.w};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (NT == 2) t[j] += __shfl_xor(t[j], 16);
                t[j] += __shfl_xor(t[j], 32);
            }
            if (rr == 0) {
                float* o = p.colsum + (long)(row0 >> 6) * 2 * p.N + colq;
                *(float4*)o = make_float4(t[0], t[1], t[2], t[3]);
                *(float4*)(o + p.N) = make_float4(t[4], t[5], t[6], t[7]);
            }
        }
    } else if (!p.nchw_out && (p.N & 3) == 0) {
        prefetch_next();
        // ---- edge tiles: same scheme with per-row / per-column predicates.  Epilogue through LDS: each wave re-lays its 32 x (32*NT) half tile as rows in a private LDS patch
        // (the K loop's last barrier has retired every read of the stages), then moves whole rows: 16 B per lane,
        // 128*NT contiguous bytes per row for the residual read and the output store (the accumulator layout itself
        // would give 4-byte accesses in 128 B runs, which made the residual read cost as much as a third of the kernel).
        constexpr int WC = 32 * NT;                       // columns of a wave tile
        float* patch = reinterpret_cast<float*>(lds) + wave * (32 * WC);
        const int c4 = lane_e % (WC / 4), rr = lane_e / (WC / 4);          // float4 column, row within a group
        constexpr int RPI = 64 / (WC / 4);                // rows per wave instruction (4 for NT=2, 2 for NT=4)
        const int colq = n0 + wn * WC + 4 * c4;
        float4 bq = make_float4(0.f, 0.f, 0.f, 0.f), sq = make_float4(1.f, 1.f, 1.f, 1.f);
        if (p.bias && colq < p.N) bq = *(const float4*)(p.bias + colq);
        if (FMT == 3) {            // (Cscale is padded to the tile grid: no column predicate needed)
            sq = *(const float4*)(p.Cscale + colq);
            sq.x *= p.out_scale; sq.y *= p.out_scale; sq.z *= p.out_scale; sq.w *= p.out_scale;
        }
        float4 cs = make_float4(0.f, 0.f, 0.f, 0.f), cq = cs;      // this lane's column sums / sums of squares
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
            for (int ni = 0; ni < NT; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    patch[((r & 3) + 8 * (r >> 2) + 4 * kh_e) * WC + ni * 32 + lr_e] = acc[mi][ni][r];
            // same-wave LDS write -> read: in-order LDS queue, no barrier needed
#pragma unroll
            for (int it = 0; it < 32 / RPI; ++it) {
                const int rl = it * RPI + rr;
                const int row = m0 + wm * 64 + mi * 32 + rl;
                float4 v = *(const float4*)(patch + rl * WC + 4 * c4);
                if (row < p.M && colq < p.N) {
                    if (FMT == 3) { v.x = v.x * sq.x + bq.x; v.y = v.y * sq.y + bq.y; v.z = v.z * sq.z + bq.z; v.w = v.w * sq.w + bq.w; }
                    else { v.x += bq.x; v.y += bq.y; v.z += bq.z; v.w += bq.w; }
                    if (p.act_out == 2) { v.x = gelu_x(v.x); v.y = gelu_x(v.y); v.z = gelu_x(v.z); v.w = gelu_x(v.w); }
                    if (p.res) {
                        const float4 q = *(const float4*)(p.res + (long)row * p.N + colq);
                        v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
                    }
                    if (p.Csplit) {        // the consumer is another split-operand GEMM: hand it the operand format directly
                        if (FMT == 3) {
                            dua_store_f16x3_quad(p.Csplit, (long)p.M * p.N, (long)row * p.N + colq, v, p.split_scale);
                        } else {
                            bf16x4 hi, lo;
                            split4(v, hi, lo);
                            __bf16* o = p.Csplit + (long)row * p.N + colq;
                            *(bf16x4*)o = hi;
                            *(bf16x4*)(o + (long)p.M * p.N) = lo;
                        }
                    } else {
                        *(float4*)(p.C + (long)row * p.N + colq) = v;
                    }
                    cs.x += v.x; cs.y += v.y; cs.z += v.z; cs.w += v.w;
                    cq.x += v.x * v.x; cq.y += v.y * v.y; cq.z += v.z * v.z; cq.w += v.w * v.w;
                }
            }
        }
        if (p.colsum) {   // GroupNorm statistics of the NEXT layer: fold the row lanes, one float4 store per column quad
            float t[8] = {cs.x, cs.y, cs.z, cs.w, cq.x, cq.y, cq.z, cq.w};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (NT == 2) t[j] += __shfl_xor(t[j], 16);
                t[j] += __shfl_xor(t[j], 32);
            }
            if (rr == 0 && colq < p.N && m0 + wm * 64 < p.M) {
                float* o = p.colsum + (long)((m0 + wm * 64) >> 6) * 2 * p.N + colq;
                *(float4*)o = make_float4(t[0], t[1], t[2], t[3]);
                *(float4*)(o + p.N) = make_float4(t[4], t[5], t[6], t[7]);
            }
        }
    } else {
        prefetch_next();
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
    #pragma unroll
            for (int ni = 0; ni < NT; ++ni) {
                const int col = n0 + wn * 32 * NT + ni * 32 + lr_e;
                if (col >= p.N) continue;
                const float bv = p.bias ? p.bias[col] : 0.f;
                const float sv = FMT == 3 ? p.Cscale[col] * p.out_scale : 1.f;
    #pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh_e;
                    if (row >= p.M) continue;
                    float v = acc[mi][ni][r] * sv + bv;
                    if (p.res) v += p.res[(long)row * p.N + col];
                    if (p.nchw_out) {
                        int hw = p.H * p.W, n = row / hw, rem = row - n * hw;
                        p.C[((long)n * p.N + col) * hw + rem] = v;
                    } else {
                        p.C[(long)row * p.N + col] = v;
                    }
                }
            }
        }
    __builtin_amdgcn_s_barrier();      // every wave is done with this tile's LDS stages before the next tile's DMA
    DUA_STAMP(3)
  }
#endif
}

static inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

// ---- weight splitting: fp32 [N][K] (or OIHW conv weights) -> two bf16 planes [Npad128][K] -------------------
__device__ __forceinline__ float split_weight_src(const float* __restrict__ w, int n, int k, int N, int K, int conv_cin) {
    if (n >= N) return 0.f;
    if (conv_cin) {                       // OIHW -> k = cb*288 + tap*32 + c  (channel-block major, tap minor)
        int tap = (k % 288) / 32, ci = (k / 288) * 32 + k % 32;
        return w[((size_t)n * conv_cin + ci) * 9 + tap];
    }
    return w[(size_t)n * K + k];
}
__global__ void k_split_weight(const float* __restrict__ w, __bf16* __restrict__ hi, __bf16* __restrict__ lo, int N, int K,
                               int Npad, int conv_cin) {
    size_t total = (size_t)Npad * K;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        int n = (int)(i / K), k = (int)(i % K);
        float v = split_weight_src(w, n, k, N, K, conv_cin);
        __bf16 h = (__bf16)v;
        hi[i] = h;
        lo[i] = (__bf16)(v - (float)h);
    }
}

extern "C" size_t dua_bf16x3_plane_elems(int N, int K) { return (size_t)round_up(N, BN) * K; }

extern "C" int dua_pack_weight_bf16x3(const float* w, void* planes, int N, int K, int conv_cin, void* stream) {
    DUA_REQUIRE(w && planes && N > 0 && K > 0 && K % BK == 0, "dua_pack_weight_bf16x3: K must be a positive multiple of 32 (got %d)", K);
    DUA_REQUIRE(conv_cin == 0 || K == 9 * conv_cin, "dua_pack_weight_bf16x3: conv K must be 9*Cin");
    int Npad = round_up(N, BN);
    size_t plane = (size_t)Npad * K;
    __bf16* hi = (__bf16*)planes;
    hipLaunchKernelGGL(k_split_weight, stream_grid(plane, 256), 256, 0, as_stream(stream), w, hi, hi + plane, N, K, Npad, conv_cin);
    return dua_check_launch("dua_pack_weight_bf16x3");
}


extern "C" int dua_conv3x3_bf16x3(const float* x, const void* planes, const float* bias, const float* residual, float* y, int N,
                                  int H, int W, int Cin, int Cout, int nchw_out, float* colsum, void* stream) {
    DUA_REQUIRE(!colsum || (!nchw_out && Cout % 4 == 0), "dua_conv3x3_bf16x3: colsum needs an NHWC output with Cout %% 4 == 0");
    DUA_REQUIRE(x && planes && y, "dua_conv3x3_bf16x3: null pointer");
    DUA_REQUIRE(N > 0 && H > 0 && W > 0 && Cout > 0 && Cin >= BK && Cin % BK == 0, "dua_conv3x3_bf16x3: Cin must be a multiple of 32 (got %d)", Cin);
    DUA_REQUIRE((long)N * H * W < (1L << 31), "dua_conv3x3_bf16x3: too many pixels");
    XTileP p{};
    size_t plane = dua_bf16x3_plane_elems(Cout, 9 * Cin);
    p.A1 = x; p.Bhi = (const __bf16*)planes; p.Blo = p.Bhi + plane; p.bias = bias; p.res = residual; p.C = y;
    p.M = N * H * W; p.N = Cout; p.ldc = Cout; p.ldr = Cout; p.alpha = 1.0f;
    p.H = H; p.W = W; p.Cin = Cin; p.nchw_out = nchw_out; p.colsum = colsum;
    return launch_x<XMODE_CONV>(p, as_stream(stream));
}

// Engine selector for the pre-split conv: 1 = LDS-DMA kernels (default; 256x256 tile where Cout % 256 == 0),
// 2 = LDS-DMA 256x128 only, 0 = register-staged kernel (debug / A-B).
// Process-global A/B switch (set it before any work is queued; not synchronised with running streams).  Modes 3 - 5 build
// zero-record buffer descriptors - timing-only runs with WRONG results - and exist only in the diagnostic library
// (-DDUA_CONV_TIMING, scripts/conv_tile_timing.py --build); the product library rejects them.
static int dua_dma_conv_enabled = 1;
extern "C" int dua_set_dma_conv(int on) {
#ifndef DUA_CONV_TIMING
    DUA_REQUIRE(on == 0 || on == 1 || on == 2 || on == 6 || on == 7, "dua_set_dma_conv: mode %d exists only in the diagnostic build (wrong results by design)", on);
#endif
    int o = dua_dma_conv_enabled;
    dua_dma_conv_enabled = on;
    return o;
}

// Start-up stagger of the persistent LDS-DMA workgroups: `phases` phases per XCD (0 / 1 = off), each `percent` % of 1/phases
// of the estimated tile period.  Applied only when every workgroup owns at least `min_tiles` tiles.
static int dua_stagger_phases = 0, dua_stagger_percent = 100, dua_stagger_min_tiles = 6;
extern "C" int dua_set_conv_stagger(int phases, int percent, int min_tiles) {
#ifndef DUA_CONV_TIMING
    DUA_REQUIRE(phases <= 1, "dua_set_conv_stagger: the start-up stagger (measured: -1 .. -4 %%) is compiled into the diagnostic build only");
#endif
    dua_stagger_phases = phases; dua_stagger_percent = percent; dua_stagger_min_tiles = min_tiles;
    return 0;
}
static void dconv_set_stagger(DConvP& d, bool wide, int ntiles, int grid) {
    d.stagger_phases = 0; d.stagger_sleeps = 0;
    if (dua_stagger_phases <= 1 || ntiles < (long)dua_stagger_min_tiles * grid) return;
    // tile period in shader cycles: K steps of ~3840 (256x256) / ~1920 (256x128) cycles plus the epilogue
    const long period = (long)(d.K / BK) * (wide ? 3840 : 1920) + (wide ? 68000 : 34000);
    const long step = period * dua_stagger_percent / 100 / dua_stagger_phases;
    d.stagger_phases = dua_stagger_phases;
    d.stagger_sleeps = (int)((step + 4064) / 8128);
}

// ---- split-K for launches that cannot fill the chip (small batches: SD's CFG batch of 2 gives the 32^2 / 16^2 / 8^2 levels 10 - 40
// tiles for 256 CUs, and a launch lasts one whole tile whatever the tile count).  Opt-in: the caller registers a workspace.
static float* dua_splitk_ws = nullptr;
static size_t dua_splitk_ws_bytes = 0;
extern "C" int dua_set_splitk_workspace(void* ws, size_t bytes) {
    DUA_REQUIRE((ws == nullptr) == (bytes == 0) && ((uintptr_t)ws & 15) == 0, "dua_set_splitk_workspace: a 16-byte aligned buffer, or (NULL, 0) to switch split-K off");
    dua_splitk_ws = (float*)ws;
    dua_splitk_ws_bytes = bytes;
    dua_hgemm_set_workspace(ws, bytes);          // the stream-K tail of the one-tap GEMM kernel (gemm_f16x3.hip) shares the buffer
    return DUA_OK;
}
// One workgroup per (64-row block, 64-column block): C = sum_s part[s] (index order) + bias (+ residual), and the block's column
// sums / sums of squares in the layout of the conv epilogue.  Thread = (column quad cq, row lane rl): rows rl, rl + 16, rl + 32, rl + 48.
__global__ __launch_bounds__(256) void k_splitk_reduce(const float* __restrict__ part, int S, const float* __restrict__ bias,
                                                       const float* __restrict__ res, float* __restrict__ C, float* __restrict__ colsum,
                                                       int M, int N) {
    __shared__ float4 sh[2][16][16];
    const int cq = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int r0 = blockIdx.x * 64, col = blockIdx.y * 64 + 4 * cq;
    const long slab = (long)M * N;
    const bool live = col < N;
    float4 b = make_float4(0.f, 0.f, 0.f, 0.f), cs = b, cq2 = b;
    if (live && bias) b = *(const float4*)(bias + col);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = r0 + rl + 16 * i;
        if (!live || r >= M) continue;
        const long o = (long)r * N + col;
        float4 v = *(const float4*)(part + o);
        for (int s_ = 1; s_ < S; ++s_) {
            const float4 w = *(const float4*)(part + s_ * slab + o);
            v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
        }
        v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
        if (res) {
            const float4 w = *(const float4*)(res + o);
            v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
        }
        *(float4*)(C + o) = v;
        cs.x += v.x; cs.y += v.y; cs.z += v.z; cs.w += v.w;
        cq2.x += v.x * v.x; cq2.y += v.y * v.y; cq2.z += v.z * v.z; cq2.w += v.w * v.w;
    }
    if (!colsum) return;
    sh[0][rl][cq] = cs;
    sh[1][rl][cq] = cq2;
    __syncthreads();
    if (rl < 2 && live) {                  // rl 0: sums, rl 1: sums of squares; fixed order over the 16 row lanes
        float4 t = sh[rl][0][cq];
#pragma unroll
        for (int j = 1; j < 16; ++j) {
            const float4 w = sh[rl][j][cq];
            t.x += w.x; t.y += w.y; t.z += w.z; t.w += w.w;
        }
        *(float4*)(colsum + (long)blockIdx.x * 2 * N + (long)rl * N + col) = t;
    }
}
// parts per tile for a launch of `ntiles` tiles and nk K steps: fill the 256 CUs, keep >= 12 K steps per part, fit the workspace
static int splitk_parts(const DConvP& d, int fmt, int ntiles, int nk) {
    if (!dua_splitk_ws || !(fmt == 0 || fmt == 3) || d.nchw_out || d.Csplit || d.act_out != 0 || (d.N & 3) || ntiles > 128) return 1;
    int S = 256 / ntiles;
    if (S > nk / 12) S = nk / 12;
    if (S > 16) S = 16;
    while (S > 1 && (size_t)S * d.M * d.N * sizeof(float) > dua_splitk_ws_bytes) --S;
    return S < 2 ? 1 : S;
}

static int launch_dconv(DConvP& d, int fmt, hipStream_t st, const char* what) {
    d.ntm = (d.M + DBM - 1) / DBM;
    // 256-wide column tiles (half the L2->LDS bytes per MFMA) when they tile N exactly and still give every CU
    // at least two workgroups; otherwise 128-wide tiles keep the 256 CUs busy.
    const bool wide = dua_dma_conv_enabled != 2 && d.N % 256 == 0 && (long)d.ntm * (d.N / 256) >= 512;
    d.ntn = wide ? d.N / 256 : (d.N + BN - 1) / BN;
    const int ntiles = d.ntm * d.ntn, grid = ntiles < 256 ? ntiles : 256;      // one persistent workgroup per CU
    // Tail split (plain GEMMs, split-K workspace registered): when the last round of tiles would be at most ~60 % full, the rows are
    // cut into a head whose tiles fill whole rounds and a tail of <= 128 tiles that runs split along K on all CUs (U-ViT-H/2 at batch
    // 64: the N = 1152 projections are 585 tiles = 2.29 rounds -> 504 tiles in two rounds + 81 tiles x 3 K parts in a third of one).
    if (fmt == 3 && dua_splitk_ws && !d.tail_pass && d.ntaps == 1 && !d.Csplit && d.act_out == 0 && !d.nchw_out && (d.N & 3) == 0 &&
        ntiles > 256 && dua_dma_conv_enabled != 6 && dua_dma_conv_enabled != 7) {
        const int r_full = ntiles / 256, rem = ntiles - 256 * r_full;
        const int head_rt = (256 * r_full) / d.ntn, tail_rt = d.ntm - head_rt;
        const int tail_tiles = tail_rt * ((d.N + BN - 1) / BN);
        DConvP t = d;
        t.M = d.M - head_rt * DBM;
        t.tail_pass = 1;
        if (rem > 0 && rem <= 160 && head_rt > 0 && tail_tiles <= 128 && splitk_parts(t, fmt, tail_tiles, d.K / BK) >= 2) {
            DConvP h = d;
            h.M = head_rt * DBM;
            h.tail_pass = 1;
            const long r0 = (long)head_rt * DBM;
            t.A = d.A + r0 * d.K;                                   // (the lo plane stays a_plane elements behind)
            t.C = d.C + r0 * d.N;
            if (d.res) t.res = d.res + r0 * d.N;
            if (d.colsum) t.colsum = d.colsum + (r0 / 64) * 2 * d.N;
            const int rc = launch_dconv(h, fmt, st, what);
            return rc != DUA_OK ? rc : launch_dconv(t, fmt, st, what);
        }
    }
#ifdef DUA_CONV_TIMING
    d.dbg_norec = dua_dma_conv_enabled == 3 ? 3 : (dua_dma_conv_enabled == 4 ? 1 : (dua_dma_conv_enabled == 5 ? 2 : 0));   // 4: A only, 5: B only
#else
    d.dbg_norec = 0;
#endif
    dconv_set_stagger(d, wide, ntiles, grid);
    if (fmt == 0) {
        if (wide) hipLaunchKernelGGL((k_xconv_dma<4, 0>), dim3(grid), 512, 0, st, d);
        else hipLaunchKernelGGL((k_xconv_dma<2, 0>), dim3(grid), 512, 0, st, d);
    } else if (fmt == 3 && dua_dma_conv_enabled != 6 && !d.Csplit && d.act_out == 0 &&
               ((wide && (long)d.H * d.W <= 4096) || dua_dma_conv_enabled == 7)) {
        // f16x3, 256-channel tiles on maps up to 64^2: the 16x16x32 kernel of conv_f16x3.hip (+2..5 % there; the 128^2 layers and
        // the 128-channel tiles of the 16^2 / 8^2 levels run 2 - 15 % faster on the 32x32x16 loop below: measured per layer with
        // scripts/bench_conv_mix.py --fmt f16x3 --engine 7 / 6).  Mode 7 forces the new kernel, 6 the old one.
        return dua_launch_hconv(d, st, what);
    } else if (fmt == 3) {                                                        // (mode 6, A/B: the bf16x3 loop on fp16 planes, 32x32x16 shape)
        const int S = splitk_parts(d, fmt, ntiles, d.K / BK);
        if (S > 1) {
            d.ksplit = S;
            d.kpart = dua_splitk_ws;
            const int items = ntiles * S;
            if (wide) hipLaunchKernelGGL((k_xconv_dma<4, 3, false, true>), dim3(items < 256 ? items : 256), 512, 0, st, d);
            else hipLaunchKernelGGL((k_xconv_dma<2, 3, false, true>), dim3(items < 256 ? items : 256), 512, 0, st, d);
            hipLaunchKernelGGL(k_splitk_reduce, dim3((d.M + 63) / 64, (d.N + 63) / 64), 256, 0, st, d.kpart, S, d.bias, d.res, d.C, d.colsum, d.M, d.N);
            return dua_check_launch(what);
        }
        if (wide) hipLaunchKernelGGL((k_xconv_dma<4, 3>), dim3(grid), 512, 0, st, d);
        else hipLaunchKernelGGL((k_xconv_dma<2, 3>), dim3(grid), 512, 0, st, d);
    } else {
        dua_set_error("%s: unknown operand format %d", what, fmt);
        return DUA_EINVAL;
    }
    return dua_check_launch(what);
}

static int conv3x3_presplit_impl(const void* xplanes, const void* planes, const float* bias, const float* residual, float* y, int N,
                                 int H, int W, int Cin, int Cout, int nchw_out, float* colsum, int fmt, void* stream, float act_scale = 1.0f) {
    DUA_REQUIRE(!colsum || (!nchw_out && Cout % 4 == 0), "dua_conv3x3 presplit: colsum needs an NHWC output with Cout %% 4 == 0");
    DUA_REQUIRE(xplanes && planes && y, "dua_conv3x3 presplit: null pointer");
    DUA_REQUIRE(N > 0 && H > 0 && W > 0 && Cout > 0 && Cin >= BK && Cin % BK == 0, "dua_conv3x3 presplit: Cin must be a multiple of 32 (got %d)", Cin);
    DUA_REQUIRE((long)N * H * W < (1L << 31), "dua_conv3x3 presplit: too many pixels");
    size_t plane = dua_bf16x3_plane_elems(Cout, 9 * Cin);
    const size_t a_plane = (size_t)N * H * W * Cin;
    const bool dma_ok = 4 * plane < 0x7FFFFFF0ull && (size_t)(2 * W + 260) * Cin * 2 < 0x7FFFFFF0ull;
    if (fmt >= 1) DUA_REQUIRE(dma_ok, "dua_conv3x3 presplit: operand too large for the LDS-DMA kernel");
    if ((dua_dma_conv_enabled != 0 || fmt >= 1) && dma_ok) {
        DConvP d{};
        d.A = (const __bf16*)xplanes; d.B = (const __bf16*)planes; d.bias = bias; d.res = residual; d.C = y;
        d.a_plane = (long)a_plane; d.b_plane = (long)plane;
        d.M = N * H * W; d.N = Cout; d.K = 9 * Cin; d.H = H; d.W = W; d.Cin = Cin; d.nchw_out = nchw_out; d.colsum = colsum;
        d.ntaps = 9;
        d.Cscale = (const float*)((const __bf16*)planes + 2 * plane);
        d.out_scale = 1.0f / act_scale;
        return launch_dconv(d, fmt, as_stream(stream), "dua_conv3x3 presplit (dma)");
    }
    XTileP p{};
    p.Ahi = (const __bf16*)xplanes; p.Alo = p.Ahi + a_plane;
    p.Bhi = (const __bf16*)planes; p.Blo = p.Bhi + plane; p.bias = bias; p.res = residual; p.C = y;
    p.M = N * H * W; p.N = Cout; p.ldc = Cout; p.ldr = Cout; p.alpha = 1.0f;
    p.H = H; p.W = W; p.Cin = Cin; p.nchw_out = nchw_out; p.colsum = colsum;
    return launch_x<XMODE_CONV, false, true>(p, as_stream(stream));
}

extern "C" int dua_conv3x3_bf16x3_presplit(const void* xplanes, const void* planes, const float* bias, const float* residual,
                                           float* y, int N, int H, int W, int Cin, int Cout, int nchw_out, float* colsum, void* stream) {
    return conv3x3_presplit_impl(xplanes, planes, bias, residual, y, N, H, W, Cin, Cout, nchw_out, colsum, 0, stream);
}

// ---- f16x3 weight format (fp32-equivalent split): plane 0 = fp16(w * 2^s) [Npad128][K], plane 1 = fp16(w * 2^s - plane 0), then
// Npad floats 2^-s.  s per output row such that the row maximum lands in [2^14, 2^15): both planes stay in fp16's normal range
// for every element within 2^-18 of the row maximum, so hi + lo carries 22 significant bits of w; the product
// hi*hi + hi*lo + lo*hi (three v_mfma_f32_32x32x16_f16, fp32 accumulate) drops only the lo*lo term (2^-22 relative).
__global__ void k_weight_row_scale_f16x3(const float* __restrict__ w, float* __restrict__ inv_scale, int N, int K, int Npad) {
    const int n = blockIdx.x;
    float m = 0.f;
    if (n < N)
        for (int k = threadIdx.x; k < K; k += blockDim.x) m = fmaxf(m, fabsf(w[(size_t)n * K + k]));
    __shared__ float red[256];
    red[threadIdx.x] = m;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + o]);
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        int e = 0;
        if (red[0] > 0.f) e = 14 - (int)((__float_as_uint(red[0]) >> 23) & 0xff) + 127;      // 14 - floor(log2 max) (normal floats)
        e = max(-100, min(100, e));
        inv_scale[n] = __uint_as_float((unsigned)(127 - e) << 23);                           // 2^-e
    }
}
// Stage-contiguous layout (round 3): [row tile of 128][K step of 32][hi | lo][128 rows][32] - the 16 KB a K step of a 128-row tile
// stages are one block in LDS-image order, so every 1-KB LDS-DMA piece (16 rows x 64 B) reads eight whole 128-byte lines instead
// of sixteen half lines of a row-major [N][K] plane.  Element (n, k) of plane pl:
//   ((n / 128 * (K / 32) + k / 32) * 2 + pl) * 4096 + (n % 128) * 32 + k % 32     (halves)
__device__ __forceinline__ size_t f16x3_staged_index(int n, int k, int K) {
    return ((size_t)(n >> 7) * (K >> 5) + (k >> 5)) * 8192 + (size_t)(n & 127) * 32 + (k & 31);
}
__global__ void k_split_weight_f16x3(const float* __restrict__ w, _Float16* __restrict__ packed,
                                     const float* __restrict__ inv_scale, int N, int K, int Npad, int conv_cin) {
    size_t total = (size_t)Npad * K;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        int n = (int)(i / K), k = (int)(i % K);
        const float sc = 1.0f / inv_scale[n];                                                // exact: a power of two
        const float v = split_weight_src(w, n, k, N, K, conv_cin) * sc;
        const _Float16 h = (_Float16)v;
        const size_t o = f16x3_staged_index(n, k, K);
        packed[o] = h;
        packed[o + 4096] = (_Float16)(v - (float)h);
    }
}
extern "C" size_t dua_f16x3_weight_bytes(int N, int K) { return 4 * dua_bf16x3_plane_elems(N, K) + 4 * (size_t)round_up(N, 256); }
extern "C" int dua_pack_weight_f16x3(const float* w, void* packed, int N, int K, int conv_cin, void* stream) {
    DUA_REQUIRE(w && packed && N > 0 && K > 0 && K % BK == 0, "dua_pack_weight_f16x3: K must be a positive multiple of 32 (got %d)", K);
    DUA_REQUIRE(conv_cin == 0 || (K == 9 * conv_cin && conv_cin % BK == 0), "dua_pack_weight_f16x3: conv K must be 9*Cin, Cin %% 32 == 0");
    const int Npad = round_up(N, BN), Nsc = round_up(N, 256);
    size_t plane = (size_t)Npad * K;
    _Float16* hi = (_Float16*)packed;
    float* inv_scale = (float*)(hi + 2 * plane);
    hipLaunchKernelGGL(k_weight_row_scale_f16x3, dim3(Nsc), 256, 0, as_stream(stream), w, inv_scale, N, K, Npad);
    hipLaunchKernelGGL(k_split_weight_f16x3, stream_grid(plane, 256), 256, 0, as_stream(stream), w, hi, inv_scale, N, K, Npad, conv_cin);
    return dua_check_launch("dua_pack_weight_f16x3");
}
extern "C" int dua_conv3x3_f16x3_presplit(const void* xplanes, const void* packed, const float* bias, const float* residual,
                                          float* y, int N, int H, int W, int Cin, int Cout, int nchw_out, float* colsum,
                                          float act_scale, void* stream) {
    DUA_REQUIRE(act_scale > 0.f, "dua_conv3x3_f16x3_presplit: act_scale must be positive");
    return conv3x3_presplit_impl(xplanes, packed, bias, residual, y, N, H, W, Cin, Cout, nchw_out, colsum, 3, stream, act_scale);
}

// C[M][N] = A * W^T (+ bias) (+ residual) with A given as split planes (written by dua_groupnorm_apply_split / _split_f16x3) and W
// packed by dua_pack_weight_bf16x3 / _f16x3: the LDS-DMA kernel with a single tap (f16x3: k_hgemm of gemm_f16x3.hip by default).
static int gemm_presplit_impl(const void* aplanes, const void* planes, const float* bias, const float* residual, float* C,
                              void* Csplit, int M, int N, int K, int act_out, float* colsum, int fmt, void* stream, float act_scale = 1.0f,
                              float split_scale = 1.0f) {
    DUA_REQUIRE(aplanes && planes && (C || Csplit) && M > 0 && N > 0 && K >= BK && K % BK == 0, "dua_gemm presplit: K must be a multiple of 32 (got %d)", K);
    DUA_REQUIRE(N % 4 == 0, "dua_gemm presplit: N must be a multiple of 4");
    DUA_REQUIRE(!Csplit || fmt == 0 || N % 32 == 0, "dua_gemm presplit: a split output needs N %% 32 == 0");
    size_t plane = dua_bf16x3_plane_elems(N, K);
    DUA_REQUIRE(4 * plane < 0x7FFFFFF0ull && (size_t)260 * K * 2 < 0x7FFFFFF0ull, "dua_gemm presplit: operand too large");
    if (fmt == 3 && dua_hgemm_enabled() && (!Csplit || N % 32 == 0) && (size_t)M * K * 2 < 0xFFFFFFF0ull) {
        // the one-tap GEMM kernel of gemm_f16x3.hip (16x16x32 MFMA shape, static rings, 256 / 192 / 128-column tiles, stream-K tail)
        HGemmP h{};
        h.A1 = (const _Float16*)aplanes; h.a1_plane = (long)M * K; h.K1 = K; h.K2 = 0;
        h.B = (const _Float16*)planes; h.b_bytes = (long)(4 * plane);
        h.Cscale = (const float*)((const __bf16*)planes + 2 * plane); h.bias = bias; h.res = residual; h.C = C;
        h.Cplanes = (_Float16*)Csplit; h.colsum = colsum; h.M = M; h.N = N; h.act_out = act_out;
        h.out_scale = 1.0f / act_scale; h.split_scale = split_scale;
        return dua_launch_hgemm(h, as_stream(stream), "dua_gemm presplit (hgemm)");
    }
    DConvP d{};
    d.A = (const __bf16*)aplanes; d.B = (const __bf16*)planes; d.bias = bias; d.res = residual; d.C = C;
    d.a_plane = (long)M * K; d.b_plane = (long)plane;
    d.M = M; d.N = N; d.K = K; d.H = 1; d.W = 1; d.Cin = K; d.nchw_out = 0; d.colsum = colsum; d.ntaps = 1;
    d.act_out = act_out; d.Csplit = (__bf16*)Csplit;
    d.Cscale = (const float*)((const __bf16*)planes + 2 * plane);
    d.out_scale = 1.0f / act_scale;
    d.split_scale = split_scale;
    return launch_dconv(d, fmt, as_stream(stream), "dua_gemm presplit");
}
extern "C" int dua_gemm_bf16x3_presplit(const void* aplanes, const void* planes, const float* bias, const float* residual, float* C,
                                        void* Csplit, int M, int N, int K, int act_out, float* colsum, void* stream) {
    return gemm_presplit_impl(aplanes, planes, bias, residual, C, Csplit, M, N, K, act_out, colsum, 0, stream);
}

extern "C" int dua_gemm_f16x3_presplit(const void* aplanes, const void* packed, const float* bias, const float* residual, float* C,
                                       int M, int N, int K, int act_out, float* colsum, float act_scale, void* stream) {
    DUA_REQUIRE(act_scale > 0.f, "dua_gemm_f16x3_presplit: act_scale must be positive");
    return gemm_presplit_impl(aplanes, packed, bias, residual, C, nullptr, M, N, K, act_out, colsum, 3, stream, act_scale);
}
// ... with the A operand concatenated along K from two plane sets, and optionally the f16x3 planes of the result beside it
extern "C" int dua_gemm_f16x3_presplit2(const void* a1planes, int K1, const void* a2planes, int K2, const void* packed, const float* bias,
                                        const float* residual, float* C, void* extra_planes, int extra_from, int extra_only, int M, int N,
                                        int act_out, float* colsum, float act_scale, float split_scale, void* stream) {
    DUA_REQUIRE(a1planes && packed && C && M > 0 && N > 0 && K1 >= BK && K1 % BK == 0 && K2 >= 0 && K2 % BK == 0,
                "dua_gemm_f16x3_presplit2: K1 and K2 must be multiples of 32 (got %d, %d)", K1, K2);
    DUA_REQUIRE((K2 == 0) == (a2planes == nullptr), "dua_gemm_f16x3_presplit2: a2planes / K2 mismatch");
    DUA_REQUIRE(N % 4 == 0 && act_scale > 0.f && (!extra_planes || split_scale > 0.f), "dua_gemm_f16x3_presplit2: N %% 4 == 0 and positive scales are required");
    DUA_REQUIRE(!extra_planes || (extra_from >= 0 && extra_from % 32 == 0 && (!extra_only || (!colsum && !residual))),
                "dua_gemm_f16x3_presplit2: extra_from must be a multiple of 32; extra_only excludes colsum / residual");
    const int K = K1 + K2;
    const size_t plane = dua_bf16x3_plane_elems(N, K);
    DUA_REQUIRE(4 * plane < 0x7FFFFFF0ull && (size_t)M * (K1 > K2 ? K1 : K2) * 2 < 0xFFFFFFF0ull, "dua_gemm_f16x3_presplit2: operand too large");
    HGemmP h{};
    h.A1 = (const _Float16*)a1planes; h.a1_plane = (long)M * K1; h.K1 = K1;
    h.A2 = (const _Float16*)a2planes; h.a2_plane = (long)M * K2; h.K2 = K2;
    h.B = (const _Float16*)packed; h.b_bytes = (long)(4 * plane);
    h.Cscale = (const float*)((const __bf16*)packed + 2 * plane); h.bias = bias; h.res = residual; h.C = C;
    h.Cextra = (_Float16*)extra_planes; h.extra_from = extra_from; h.extra_only = extra_only; h.colsum = colsum; h.M = M; h.N = N; h.act_out = act_out;
    h.out_scale = 1.0f / act_scale; h.split_scale = split_scale;
    return dua_launch_hgemm(h, as_stream(stream), "dua_gemm_f16x3_presplit2");
}
// ... with the result written as f16x3 planes of C * split_scale ([M][N] hi | lo) for the next presplit GEMM (U-ViT fc1 -> GELU -> fc2)
extern "C" int dua_gemm_f16x3_presplit_planes(const void* aplanes, const void* packed, const float* bias, void* Cplanes, int M, int N, int K,
                                              int act_out, float act_scale, float split_scale, void* stream) {
    DUA_REQUIRE(act_scale > 0.f && split_scale > 0.f && Cplanes, "dua_gemm_f16x3_presplit_planes: bad arguments");
    return gemm_presplit_impl(aplanes, packed, bias, nullptr, nullptr, Cplanes, M, N, K, act_out, nullptr, 3, stream, act_scale, split_scale);
}

extern "C" int dua_gemm_bf16x3(const float* A1, int K1, long lda1, const float* A2, int K2, long lda2, const void* planes,
                               const float* bias, const float* residual, long ldr, float* C, long ldc, int M, int N, int act_in,
                               int act_out, float alpha, float* colsum, void* stream) {
    DUA_REQUIRE(!colsum || (N % 4 == 0 && ldc == N && (!residual || ldr % 4 == 0)), "dua_gemm_bf16x3: colsum needs N %% 4 == 0 and a dense row-major C");
    DUA_REQUIRE(A1 && planes && C && M > 0 && N > 0 && K1 > 0 && K2 >= 0, "dua_gemm_bf16x3: bad arguments");
    DUA_REQUIRE((K1 + K2) % BK == 0 && (K2 == 0 || K1 % BK == 0), "dua_gemm_bf16x3: K (and K1 for a two-source A) must be multiples of 32");
    DUA_REQUIRE(lda1 % 4 == 0 && (K2 == 0 || lda2 % 4 == 0) && (K2 == 0) == (A2 == nullptr), "dua_gemm_bf16x3: bad A operands");
    DUA_REQUIRE(act_in == 0 || act_in == 1, "dua_gemm_bf16x3: act_in must be 0 or 1");
    XTileP p{};
    size_t plane = dua_bf16x3_plane_elems(N, K1 + K2);
    p.A1 = A1; p.A2 = A2; p.Bhi = (const __bf16*)planes; p.Blo = p.Bhi + plane; p.bias = bias; p.res = residual; p.C = C;
    p.M = M; p.N = N; p.K1 = K1; p.K2 = K2; p.lda1 = lda1; p.lda2 = lda2; p.ldc = ldc; p.ldr = ldr;
    p.act_out = act_out; p.alpha = alpha; p.colsum = colsum;
    return act_in ? launch_x<XMODE_GEMM, true>(p, as_stream(stream)) : launch_x<XMODE_GEMM, false>(p, as_stream(stream));
}

// dua_gemm_bf16x3 on the fp32-equivalent split: fp32 A scaled by act_scale and split into fp16 hi + lo on the fly (saturating),
// W packed by dua_pack_weight_f16x3; the epilogue applies alpha / act_scale and the per-row weight exponents.
extern "C" int dua_gemm_f16x3(const float* A1, int K1, long lda1, const float* A2, int K2, long lda2, const void* packed,
                              const float* bias, const float* residual, long ldr, float* C, long ldc, int M, int N, int act_in,
                              int act_out, float alpha, float* colsum, float act_scale, void* stream) {
    DUA_REQUIRE(!colsum || (N % 4 == 0 && ldc == N && (!residual || ldr % 4 == 0)), "dua_gemm_f16x3: colsum needs N %% 4 == 0 and a dense row-major C");
    DUA_REQUIRE(A1 && packed && C && M > 0 && N > 0 && K1 > 0 && K2 >= 0, "dua_gemm_f16x3: bad arguments");
    DUA_REQUIRE((K1 + K2) % BK == 0 && (K2 == 0 || K1 % BK == 0), "dua_gemm_f16x3: K (and K1 for a two-source A) must be multiples of 32");
    DUA_REQUIRE(lda1 % 4 == 0 && (K2 == 0 || lda2 % 4 == 0) && (K2 == 0) == (A2 == nullptr), "dua_gemm_f16x3: bad A operands");
    DUA_REQUIRE(act_in == 0 || act_in == 1, "dua_gemm_f16x3: act_in must be 0 or 1");
    DUA_REQUIRE(act_scale > 0.f, "dua_gemm_f16x3: act_scale must be positive");
    const size_t wplane = dua_bf16x3_plane_elems(N, K1 + K2);
    // The plain form (bias only: the ADM 1x1 skip convolutions) runs on the persistent raw-operand kernel of gemm_skip_f16x3.hip.  The
    // choice depends on the operation's FORM only, never on M: a row's result does not depend on the batch it runs in.
    if (dua_sgemm_enabled() && !act_in && !act_out && !residual && !colsum && ldc == N && N % 4 == 0 && K1 + K2 >= 2 * BK && lda1 < (1L << 20) &&
        lda2 < (1L << 20) && 4 * wplane < 0x7FFFFFF0ull && ((((uintptr_t)A1 | (uintptr_t)A2 | (uintptr_t)C) & 15) == 0)) {
        SGemmP g{};
        g.A1 = A1; g.A2 = A2; g.lda1 = lda1; g.lda2 = K2 ? lda2 : 0; g.B = (const _Float16*)packed; g.b_bytes = (long)(4 * wplane);
        g.Cscale = (const float*)((const __bf16*)packed + 2 * wplane); g.bias = bias; g.C = C; g.M = M; g.N = N; g.K1 = K1; g.K2 = K2;
        g.a_scale = act_scale; g.out_scale = alpha / act_scale;
        return dua_launch_sgemm(g, as_stream(stream), "dua_gemm_f16x3 (sgemm)");
    }
    XTileP p{};
    size_t plane = dua_bf16x3_plane_elems(N, K1 + K2);
    p.A1 = A1; p.A2 = A2; p.Bhi = (const __bf16*)packed; p.Blo = p.Bhi + plane; p.bias = bias; p.res = residual; p.C = C;
    p.M = M; p.N = N; p.K1 = K1; p.K2 = K2; p.lda1 = lda1; p.lda2 = lda2; p.ldc = ldc; p.ldr = ldr;
    p.act_out = act_out; p.alpha = alpha / act_scale; p.colsum = colsum;
    p.Cscale = (const float*)(p.Bhi + 2 * plane); p.a_scale = act_scale;
    return act_in ? launch_x<XMODE_GEMM, true, false, false, true>(p, as_stream(stream))
                  : launch_x<XMODE_GEMM, false, false, false, true>(p, as_stream(stream));
}

// Batched  C[b] = alpha * A[b] * B[b]^T  with BOTH operands fp32 activations split on the fly (attention products
// q.k^T and P.v^T, unet_openai.py:350-354).  A: [M][K] lda, B: [N][K] ldb, K % 32 == 0; batch index
// b -> (b / batch_inner, b % batch_inner) with outer / inner strides as in dua_gemm_args.
extern "C" int dua_gemm_nt_bf16x3_batched(const float* A, long lda, long sA, long sAi, const float* B, long ldb, long sB, long sBi,
                                          float* C, long ldc, long sC, long sCi, int M, int N, int K, int batch, int batch_inner,
                                          float alpha, void* stream) {
    DUA_REQUIRE(A && B && C && M > 0 && N > 0 && K > 0 && K % BK == 0, "dua_gemm_nt_bf16x3_batched: K must be a positive multiple of 32 (got %d)", K);
    DUA_REQUIRE(lda % 4 == 0 && ldb % 4 == 0 && batch >= 1 && batch_inner >= 1 && batch % batch_inner == 0, "dua_gemm_nt_bf16x3_batched: bad strides / batch");
    DUA_REQUIRE((((uintptr_t)A | (uintptr_t)B) & 15) == 0, "dua_gemm_nt_bf16x3_batched: operands must be 16-byte aligned");
    XTileP p{};
    p.A1 = A; p.Bf = B; p.C = C; p.M = M; p.N = N; p.K1 = K; p.lda1 = lda; p.ldb = ldb; p.ldc = ldc; p.ldr = ldc;
    p.sA = sA; p.sAi = sAi; p.sB = sB; p.sBi = sBi; p.sC = sC; p.sCi = sCi; p.batch_inner = batch_inner; p.alpha = alpha;
    return launch_x<XMODE_GEMM, false, false, true>(p, as_stream(stream), batch);
}

DUA_RANGE_TU(xtile)
