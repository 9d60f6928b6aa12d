// f16x3 3x3 convolution (ResBlock in_layers / out_layers convs, guided_diffusion/unet_openai.py:183-187,207-214 of the reference) with a
// Winograd F(4,3) transform ALONG Y (gfx950): four output rows cost six 1x3 row convolutions instead of
// twelve, i.e. 4.5 instead of 9 matrix-core taps per output pixel (x 3 fp16 products each; the F(2,3) form of conv_wino_f16x3.hip
// spends 6), and the transformed input is 1.5 x the input instead of 2 x.  Operand precision is unchanged (22 significant bits per
// operand, fp32 accumulation); the transform's constants (points 0, +-1, +-2, inf) amplify rounding by ~2.5 x against the direct fp32
// convolution (scripts/diag/wino43_precision.py: rms 5.1e-7 vs 2.3e-7 of the output scale on a 256-channel layer).
//
//   d0..d5 = input rows 4j-1 .. 4j+4 (zero outside the image)                 g0..g2 = kernel rows (dy = -1, 0, +1)
//   V0 = 4 d0 - 5 d2 + d4            V1 = (d3 + d4) - 4 (d1 + d2)             U0 = g0 / 4        U1 = -(g0 + g1 + g2) / 6
//   V2 = (d4 - d3) + 4 (d1 - d2)     V3 = (d4 - d2) + 2 (d3 - d1)             U2 = -(g0 - g1 + g2) / 6
//   V4 = (d4 - d2) - 2 (d3 - d1)     V5 = 4 d1 - 5 d3 + d5                    U3 = g0 / 24 + g1 / 12 + g2 / 6   U4 = g0 / 24 - g1 / 12 + g2 / 6   U5 = g2
//   M_f = conv1x3_x(V_f, U_f)  (zero padding in x)
//   out row 4j = M0 + M1 + M2 + M3 + M4          4j+1 = (M1 - M2) + 2 (M3 - M4)
//   out row 4j+2 = (M1 + M2) + 4 (M3 + M4)       4j+3 = (M1 - M2) + 8 (M3 - M4) + M5
//
// Pieces (the layouts are those of conv_wino_f16x3.hip with six frequencies and a 128-row weight stage):
//   k_gn_apply_wino4     GroupNorm * gamma + beta (+FiLM, SiLU, dropout mask, 2x resampling, virtual concat) -> the six V_f images as
//                        f16x3 images [f][tile of 128 positions][C/32][hi | lo][128][32]; a position = (image, row group j, column x)
//   k_pack_wino4         OIHW fp32 weights -> U_f, per-output-row power-of-two scale, fp16 hi | lo images, one contiguous 16 KB block
//                        [hi | lo][128 channels][32] per (column tile, K step = (channel block, f, dx))
//   k_wconv4             LDS-DMA implicit GEMM on v_mfma_f32_16x16x32_f16: a workgroup tile = 128 positions x 128 output channels x
//                        6 frequencies = 98 304 accumulators on FOUR waves (one per SIMD, 512 registers each: 384 accumulators): wave =
//                        64 positions x 64 channels x all six frequencies, so the inverse transform is wave-local (no partner
//                        exchange, no barrier in the epilogue).  K steps (channel block, f, dx): one 128-row activation stage per
//                        (channel block, f) serves the three dx taps through row-shifted fragment addresses, one 128-row weight stage
//                        per step.  Per wave and step the DMA pieces, fragment reads and MFMAs are those of k_wconv.
#include "common.h"
#include "wconv.h"
#include <type_traits>

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));

#ifndef WCONV4_DIAG
#define WCONV4_DIAG 0             // timing-only diagnostic builds (scripts/wconv_diag.py --kernel w4; WRONG results): bit 0 no in-loop barrier,
                                  // bit 1 no epilogue, bit 2 no in-loop DMA, bit 3 no in-loop fragment reads, bit 4 output stores dropped by the buffer unit,
                                  // bit 5 no residual loads, bit 6 wave 0 writes s_memrealtime at the start and end of every epilogue into the colsum buffer, bit 7 no scale / bias loads
#endif

// (A/B build -DW4_EPI_NOFENCE=1: the epilogue's sub-blocks without scheduling fences - the compiler may interleave their dependent chains)
#if defined(W4_EPI_NOFENCE) && W4_EPI_NOFENCE
#define W4_EPI_FENCE()
#else
#define W4_EPI_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif
constexpr int W4_STORES = 2 * 4 * 2 * 4;          // channel-block pairs x position blocks x halves x output rows: always-issued stores of a wave
static_assert(W4_STORES >= 63, "the tile-start wait leaves the 63 youngest vector-memory operations in flight: all must be behind the prologue DMA");

// One matrix-core product into an accumulator that LIVES in the accumulation registers (AG: frequencies 0-3, 256 AGPRs) or in the
// architectural registers (frequencies 4, 5: 128 VGPRs).  The compiler picks one form of MFMA per function (with more than 256
// accumulators: the AGPR form, parking a third of them in VGPRs through ~340 v_accvgpr moves and 58 scratch accesses per channel block);
// the explicit operand classes give the mixed form.  Every accumulator is next touched 16 MFMAs later (no back-to-back dependence); the
// first non-MFMA read of an accumulator comes after the s_nop block that follows the K loop.
template <int N, class F>
__device__ __forceinline__ void w4_static_for(F&& f) {
    if constexpr (N > 0) {
        w4_static_for<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}

// (the -DDUA_RS_ZERO_SELECT=1 test build zeroes edge fragments with VALU selects right in front of the products: it pays the wait states
// between a VALU write and a matrix-core read that the compiler would have inserted for an instruction it knows)
#if defined(DUA_RS_ZERO_SELECT) && DUA_RS_ZERO_SELECT
#define W4_MMA_PRE "s_nop 3\n\t"
#else
#define W4_MMA_PRE
#endif
template <bool AG>
__device__ __forceinline__ void w4_mma(f32x4& c, const h16x8& a, const h16x8& b) {
    if constexpr (AG) asm volatile(W4_MMA_PRE "v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
    else asm volatile(W4_MMA_PRE "v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}

// The compiler does not know that w4_mma is a matrix-core instruction, so it would not keep the wait states between one and a copy /
// arithmetic instruction that touches the same accumulator.  w4_fence ties all 16 quads of a frequency to a block of wait states: any
// compiler-generated instruction that reads or writes them is ordered behind (zero initialisation: in front of) the block.
template <bool AG>
__device__ __forceinline__ void w4_fence(f32x4 (&A)[4][4]) {
    if constexpr (AG)
        asm volatile("s_nop 15\n\ts_nop 3" : "+a"(A[0][0]), "+a"(A[0][1]), "+a"(A[0][2]), "+a"(A[0][3]), "+a"(A[1][0]), "+a"(A[1][1]), "+a"(A[1][2]), "+a"(A[1][3]),
                                            "+a"(A[2][0]), "+a"(A[2][1]), "+a"(A[2][2]), "+a"(A[2][3]), "+a"(A[3][0]), "+a"(A[3][1]), "+a"(A[3][2]), "+a"(A[3][3]));
    else
        asm volatile("s_nop 15\n\ts_nop 3" : "+v"(A[0][0]), "+v"(A[0][1]), "+v"(A[0][2]), "+v"(A[0][3]), "+v"(A[1][0]), "+v"(A[1][1]), "+v"(A[1][2]), "+v"(A[1][3]),
                                            "+v"(A[2][0]), "+v"(A[2][1]), "+v"(A[2][2]), "+v"(A[2][3]), "+v"(A[3][0]), "+v"(A[3][1]), "+v"(A[3][2]), "+v"(A[3][3]));
}

// RMODE: 0 no residual, 1 residual [N][H][W][Cout], 2 / 3 residual read through nearest 2x upsampling / 2x2 average pooling - four kernels, so that each
// holds only its own epilogue (the variants' different register needs would otherwise spill accumulators around the dispatch)
template <int RMODE>
__global__ __launch_bounds__(256, 1) void k_wconv4(WConvP p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int RA = 3, RB = 3;                          // activation ring: the stage of group g + 2 is requested while group g runs
    constexpr int SP = 128 * 32;                           // halves per plane of a stage (8 KB)
    constexpr int STG = 2 * SP;                            // halves per stage: [hi | lo][128][32]
    constexpr int BOFF = 0, AOFF = RB * STG;               // weight ring first: a shifted row -1 of activation stage 0 stays inside LDS
    __shared__ __attribute__((aligned(16))) _Float16 lds[(RA + RB) * STG];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;               // position half, channel half
    const int ntiles = p.ntm * p.ntn;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslot = (gridDim.x + 7 - xcd) >> 3;
    const int cq = ntiles >> 3, cr = ntiles & 7;
    const int chunk0 = xcd < cr ? xcd * (cq + 1) : cr * (cq + 1) + (xcd - cr) * cq, chunk_n = cq + (xcd < cr ? 1 : 0);

    constexpr int OOBL = 0x40000;                          // LDS byte address beyond the allocation: DS reads there return zeros
    static_assert(OOBL >= 160 * 1024 + 64 * 64 + 64, "the out-of-range LDS address must lie beyond the 160 KB of a gfx950 CU");
    const int lr = lane & 15, lc = lane >> 4;

    // ---- DMA state: one per-lane byte offset (this lane's 16 bytes of a 16-row piece; the chunk swizzle of k_wconv applied to the
    // source), everything else - piece row group, channel block, K step, hi / lo image - is wave-uniform and travels as the scalar offset
    const unsigned lanebase = (unsigned)((lane >> 2) * 64 + 16 * ((lane & 3) ^ (2 * ((lane >> 4) & 1))));
    const int wrow0 = (16 * (2 * wave)) * 64, wrow1 = (16 * (2 * wave + 1)) * 64;      // bytes: this wave's two 16-row pieces of a 128-row image
    const int ncb = p.Cin / 32;
    const long fstride = (long)p.ntm * ncb * 16384;        // bytes between the images of consecutive frequencies
    const char* vtile = (const char*)p.V;
    auto ra = __builtin_amdgcn_make_buffer_rsrc((void*)p.V, 0, 0, 0x00020000);
    const auto rb = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, (int)(2 * p.b_plane * 2), 0x00020000);
    constexpr int b_lo = SP * 2, a_lo = SP * 2;            // byte distance hi -> lo image inside a stage
    int toffA = 0, toffReq = 0, sboff = 0;                 // scalar: channel block of the running group / of the activation group being requested; weight stage of the next request
    bool prefetched = false;

    auto tile_geom = [&](int m0_, int tn_) {
        sboff = tn_ * (p.K / 32) * 16384;                                  // stage 0 of the column tile
        vtile = (const char*)p.V + (long)(m0_ >> 7) * ncb * 16384;       // the tile's stage blocks of frequency 0
        toffA = 0;
        toffReq = 0;
    };
    // the activation stage of (channel block at toffA, frequency f): descriptor over the tile's ncb stage blocks of that frequency
    auto a_setup = [&](int f) { ra = __builtin_amdgcn_make_buffer_rsrc((void*)(vtile + f * fstride), 0, ncb * 16384, 0x00020000); };
    auto dma_a = [&](int sa, int i) {
        _Float16* st = lds + AOFF + sa * STG;
        if (i == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, st + (16 * (2 * wave)) * 32, 16, lanebase, toffReq + wrow0, 0, 0);
        if (i == 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, st + (16 * (2 * wave)) * 32 + SP, 16, lanebase, toffReq + wrow0 + a_lo, 0, 0);
        if (i == 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, st + (16 * (2 * wave + 1)) * 32, 16, lanebase, toffReq + wrow1, 0, 0);
        if (i == 3) __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, st + (16 * (2 * wave + 1)) * 32 + SP, 16, lanebase, toffReq + wrow1 + a_lo, 0, 0);
    };
    auto dma_b = [&](int sb, int i) {
        _Float16* st = lds + BOFF + sb * STG;
        if (i == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, st + (16 * (2 * wave)) * 32, 16, lanebase, sboff + wrow0, 0, 0);
        if (i == 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, st + (16 * (2 * wave)) * 32 + SP, 16, lanebase, sboff + wrow0 + b_lo, 0, 0);
        if (i == 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, st + (16 * (2 * wave + 1)) * 32, 16, lanebase, sboff + wrow1, 0, 0);
        if (i == 3) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, st + (16 * (2 * wave + 1)) * 32 + SP, 16, lanebase, sboff + wrow1 + b_lo, 0, 0);
            sboff += 16384;                                // the next K step's stage
        }
    };
    auto issue_prologue = [&]() {          // A(channel block 0, f = 0) A(channel block 0, f = 1) B(0) B(1)
        toffReq = 0;
        a_setup(0);
#pragma unroll
        for (int i = 0; i < 4; ++i) dma_a(0, i);
        a_setup(1);
#pragma unroll
        for (int i = 0; i < 4; ++i) dma_a(1, i);
#pragma unroll
        for (int i = 0; i < 4; ++i) dma_b(0, i);
#pragma unroll
        for (int i = 0; i < 4; ++i) dma_b(1, i);
    };

    const char* ldsb = reinterpret_cast<const char*>(lds);
    const int off_l = lr * 64 + 16 * (lc ^ (2 * ((lr >> 2) & 1)));
    const int a_row0 = AOFF * 2 + wm * 64 * 64, b_row0 = BOFF * 2 + wn * 64 * 64;       // bytes, wave-uniform
    auto x_off = [&](int dx) -> int {
        const int r_ = lr + dx;                                           // -1 .. 16
        return r_ * 64 + 16 * (lc ^ (2 * (((r_ + 64) >> 2) & 1)));
    };
#if defined(DUA_RS_ZERO_SELECT) && DUA_RS_ZERO_SELECT
    constexpr bool ZSEL = true;
#else
    constexpr bool ZSEL = false;
#endif
    const int zero_addr = OOBL + (lane & 1) * 16;
    const int ax0 = a_row0 + x_off(0);
    int axm[4], axp[4];
    unsigned xedge = 0;                                    // bit i: position block i at x == 0; bit 4 + i: at x == W - 1
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int x_ = (wm * 64 + 16 * i + lr) & (p.W - 1);
        xedge |= (x_ == 0 ? 1u : 0u) << i | (x_ == p.W - 1 ? 1u : 0u) << (4 + i);
        axm[i] = (!ZSEL && x_ == 0) ? zero_addr : a_row0 + x_off(-1);
        axp[i] = (!ZSEL && x_ == p.W - 1) ? zero_addr : a_row0 + x_off(1);
    }
    const int aw0 = b_row0 + off_l;

    for (int ti = slot; ti < chunk_n; ti += nslot) {
        const int lid = chunk0 + ti;
        const int tn = lid % p.ntn, tm = lid / p.ntn;
        const int m0 = tm * 128, n0 = tn * 128;
        if (!prefetched) {
            tile_geom(m0, tn);
            issue_prologue();
        }
        const bool after_epilogue = prefetched;       // this tile's first stages were requested before the previous tile's epilogue
        prefetched = false;

        f32x4 acc[6][4][4];                           // [frequency][channel block][position block]
        {
            // (an opaque zero: a literal would be hoisted out of the tile loop as 96 live zero quads, spilled, and reloaded from scratch
            // memory at every tile start behind a full drain of the vector-memory queue)
            float z = 0.f;
            asm volatile("" : "+v"(z));
#pragma unroll
            for (int f = 0; f < 6; ++f)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[f][j][i] = (f32x4){z, z, z, z};
        }
        w4_fence<true>(acc[0]); w4_fence<true>(acc[1]); w4_fence<true>(acc[2]); w4_fence<true>(acc[3]); w4_fence<false>(acc[4]); w4_fence<false>(acc[5]);

        h16x8 xh[4], xl[4], wh[4], wl[4];
        auto rd_x = [&](int sa, int pl, int i, int dx) -> h16x8 {
            const int a = dx < 0 ? axm[i] : (dx > 0 ? axp[i] : ax0);
            const h16x8 v = *(const h16x8*)(ldsb + a + (sa * STG * 2 + pl * SP * 2 + 16 * i * 64));
            if (ZSEL && dx != 0) {
                const h16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
                return ((xedge >> (dx < 0 ? i : 4 + i)) & 1u) ? z : v;
            }
            return v;
        };
        auto rd_w = [&](int sb, int pl, int jj) -> h16x8 { return *(const h16x8*)(ldsb + aw0 + (sb * STG * 2 + pl * SP * 2 + 16 * jj * 64)); };
#define WSB() __builtin_amdgcn_sched_barrier(0)

        // the first stages have landed; behind them in the (in-order) queue are only the previous epilogue's operations, of which the
        // youngest 63 - output stores, W4_STORES of them always issue - may stay in flight
        if (after_epilogue) asm volatile("s_waitcnt vmcnt(63)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int i = 0; i < 4; ++i) { xl[i] = rd_x(0, 1, i, -1); xh[i] = rd_x(0, 0, i, -1); }
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) { wh[jj] = rd_w(0, 0, jj); wl[jj] = rd_w(0, 1, jj); }
        WSB();

        // One K step = (channel block, f, dx): 48 MFMAs into the accumulators of f.  Ring slots are static: the activation stage of
        // (channel block, f) lives in slot f % 3, the weight stage of step dxi in slot dxi.  ONE copy of the 18 steps serves every channel
        // block: in the tile's final group (last channel block, f = 5) the requests that would run past the tile are skipped by uniform
        // branches around single instructions, so that no two regions of code hold the accumulators in different registers (the
        // copies between them would sit next to matrix-core instructions the compiler cannot see).
        auto step = [&](auto fi_c, auto dxi_c, const bool last_cb) __attribute__((always_inline)) {
            constexpr int FI = decltype(fi_c)::value, DXI = decltype(dxi_c)::value;
            constexpr bool AG = FI < 4;
            const bool lastg = FI == 5 && last_cb;                              // (compile-time false for f < 5): no weight stage beyond the tile
            const bool lastA = FI >= 4 && last_cb;                              // group g + 2 lies beyond the tile: no activation request
            constexpr int FN = (FI + 1) % 6, F2 = (FI + 2) % 6;
            constexpr int sa_n = F2 % 3, sb_n = (DXI + 2) % 3;
            constexpr int sa_r = DXI == 2 ? (FN % 3) : (FI % 3), sb_r = (DXI + 1) % 3, dxn = DXI == 2 ? -1 : DXI;   // the NEXT step's fragments
            f32x4 (&A)[4][4] = acc[FI];
            if (DXI == 0 && !lastA) {                                           // set up the activation stage of group g + 2 ...
                toffReq = toffA + (FI >= 4 ? 16384 : 0);                        // (f = 4, 5: it opens the next channel block)
                a_setup(F2);
            }
            auto early_piece = [&](int q) {
                if (WCONV4_DIAG & 4) return;
                if (q < 4) { if (!lastg || DXI == 0) dma_b(sb_n, q); }          // the weight stage of step s + 2
                else if (q < 6) { if (DXI < 2 && !lastA) dma_a(sa_n, q - 4 + 2 * DXI); }    // ... requested half in this group's first, half in its second step
            };
            WSB();
            // P1: w_hi * x_lo | the DMA pieces of this step
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    w4_mma<AG>(A[jj][i], wh[jj], xl[i]);
                    WSB();
                    const int m_ = jj * 4 + i;
                    if ((m_ & 1) == 0) early_piece(m_ >> 1);
                    WSB();
                }
            // every read of stage s was issued during the previous step: publish stage s + 1
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            // in issue order: [B(s+1) x4, A x2] of the previous step, [B(s+2) x4, A x2] of this one (A pieces in a group's first two steps
            // only).  B(s+1) must have landed; the activation stage the next group reads was requested a whole group earlier.  In the
            // last two groups of the tile no activation pieces are requested, in the final one no weight stage beyond its last step.
            if (lastA) {
                if (FI == 5 && DXI > 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            } else {
                if (DXI == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            }
            if (!(WCONV4_DIAG & 1)) __builtin_amdgcn_s_barrier();
            WSB();
            // P2: w_hi * x_hi | x_lo and w_hi of the next step (after the tile's last step: reads whose values nobody uses)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    w4_mma<AG>(A[jj][i], wh[jj], xh[i]);
                    WSB();
                    if (jj == 0 && !(WCONV4_DIAG & 8)) { xl[i] = rd_x(sa_r, 1, i, dxn); WSB(); }
                }
                if (!(WCONV4_DIAG & 8)) wh[jj] = rd_w(sb_r, 0, jj);
                WSB();
            }
            // P3: w_lo * x_hi | w_lo and x_hi of the next step
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    w4_mma<AG>(A[jj][i], wl[jj], xh[i]);
                    WSB();
                    if (jj == 3 && !(WCONV4_DIAG & 8)) { xh[i] = rd_x(sa_r, 0, i, dxn); WSB(); }
                }
                if (!(WCONV4_DIAG & 8)) wl[jj] = rd_w(sb_r, 1, jj);
                WSB();
            }
        };
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        using I2 = std::integral_constant<int, 2>;
        using I3 = std::integral_constant<int, 3>;
        using I4 = std::integral_constant<int, 4>;
        using I5 = std::integral_constant<int, 5>;
#define W4_GROUP(F) step(F{}, I0{}, last_cb); step(F{}, I1{}, last_cb); step(F{}, I2{}, last_cb);
#pragma nounroll
        for (int cbi = 0; cbi < ncb; ++cbi) {
            toffA = cbi * 16384;
            int last_i = __builtin_amdgcn_readfirstlane(cbi == ncb - 1 ? 1 : 0);
            asm volatile("" : "+s"(last_i));                                    // (opaque: the loop is neither peeled nor versioned on it)
            const bool last_cb = last_i != 0;
            W4_GROUP(I0) W4_GROUP(I1) W4_GROUP(I2) W4_GROUP(I3) W4_GROUP(I4) W4_GROUP(I5)
        }
#undef W4_GROUP
        // the last barrier of the tile has retired every LDS read of the rings: request the first stages of the next tile now
        if (ti + nslot < chunk_n) {
            const int nl = chunk0 + ti + nslot;
            tile_geom((nl / p.ntn) * 128, nl % p.ntn);
            issue_prologue();
            prefetched = true;
        }
#undef WSB
        unsigned long long t_epi0 = 0;
        if (WCONV4_DIAG & 64) t_epi0 = __builtin_amdgcn_s_memrealtime();
        // the last MFMAs' results before any other instruction reads them
        w4_fence<true>(acc[0]); w4_fence<true>(acc[1]); w4_fence<true>(acc[2]); w4_fence<true>(acc[3]); w4_fence<false>(acc[4]); w4_fence<false>(acc[5]);
        if (WCONV4_DIAG & 2) {                       // (diagnostic: no epilogue - the accumulators are kept alive by a store that never happens)
            float t_ = 0.f;
#pragma unroll
            for (int f = 0; f < 6; ++f)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int i = 0; i < 4; ++i) t_ += acc[f][j][i][0] + acc[f][j][i][1] + acc[f][j][i][2] + acc[f][j][i][3];
            if (t_ == 1.2345e-30f) p.C[0] = t_;
            continue;
        }

        // ---- epilogue.  The accumulator layout (lane = position lane & 15, channels 4 (lane >> 4) .. + 3 of a 16-channel block) puts the
        // 16 bytes of ADJACENT lanes 1 KB apart (one pixel row each).  Stored as is - or as 64-byte pieces - the output costs the
        // memory side a fill per partial 128-byte line: 3.4 TB/s against 5.4 TB/s for whole lines with all CUs writing
        // (scripts/diag/store_pattern_bench.hip, profiles/r04y_store_pattern.txt).  So per (channel-block pair jp, position block i):
        // inverse transform of the 2 x 6 quads in registers, then a lane transpose through ds_bpermute (the LDS crossbar: no LDS
        // memory, no barrier) into the row layout - lane l' = position (l' >> 3) + 8 h, channels 4 (l' & 7) .. + 3 of the pair's 32:
        // eight adjacent lanes hold one whole 128-byte line of a pixel row, and an instruction writes eight of them.  The residual is
        // read in that layout too.  vmcnt retires in order, so a wait for residual rows is also a wait for every OLDER output store:
        // the rows of the next half block are requested before the stores of this one go out.
        // (lane-derived values are recomputed per block from a fresh lane id instead of living in registers across the K loop: the
        // register file is full there, and a reload from scratch memory is a vector-memory wait that drains every store in flight)
        auto lane_now = [&]() -> int {                                      // (volatile: neither hoisted out of the tile loop nor shared between uses)
            int l;
            asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
            return l;
        };
        const int pos0 = m0 + wm * 64;
        const long cpix0 = 4L * m0;                                          // the tile's 128 positions are output pixels 4 m0 .. 4 m0 + 511
        const long crem = (4L * p.Mv - cpix0) * p.N * 4;
        const auto rc = __builtin_amdgcn_make_buffer_rsrc((void*)(p.C + cpix0 * p.N), 0, (int)(crem < 0x7FFFFFF0L ? crem : 0x7FFFFFF0L), 0x00020000);
        // output pixel (row of C) of position pl of block i, output row 0; -1 beyond the tensor
        const bool tile_full = n0 + 128 <= p.N && m0 + 128 <= p.Mv;
        // ... RELATIVE to the tile's first output pixel cpix0 = 4 m0 (m0 is a multiple of 128 >= W): d + 3 (d & ~(W - 1)) for the position d of
        // the tile - small 32-bit numbers.  (As 64-bit pixel indices every sub-block paid a v_mad_u64_u32 and every store a v_mul_lo_u32 by N:
        // quarter-rate instructions, ~1 300 of the epilogue's ~9 000 issue cycles per wave and tile.)
        const int w_mask = ~(p.W - 1);
        auto opix = [&](int i, int pl) -> int {
            const int d = wm * 64 + 16 * i + pl;
            return (tile_full || m0 + d < p.Mv) ? d + 3 * (d & w_mask) : -1;
        };
        const unsigned row_bytes = (unsigned)p.W * (unsigned)p.N * 4u;         // one output row further (wave-uniform)
        // the residual descriptor: RMODE 1 mirrors the output's (same offsets); RMODE 2 / 3 start at the first image the tile touches in the
        // half / double resolution tensor (offsets stay far below 2^32)
        const unsigned res_n0 = RMODE >= 2 ? (unsigned)(cpix0 / ((long)p.H * p.W)) : 0u;
        const long res_img = RMODE == 2 ? (long)(p.H >> 1) * (p.W >> 1) : (RMODE == 3 ? 4L * p.H * p.W : 0L);       // source pixels per image
        const long res_tot = RMODE >= 2 ? (long)(4L * p.Mv / ((long)p.H * p.W)) * res_img : 0L;                      // ... in the whole tensor
        const long rrem = RMODE >= 2 ? (res_tot - (long)res_n0 * res_img) * p.N * 4 : crem;
        const auto rres = __builtin_amdgcn_make_buffer_rsrc((void*)(RMODE == 0 ? (const float*)p.C : (RMODE == 1 ? p.res + cpix0 * p.N : p.res + (long)res_n0 * res_img * p.N)), 0,
                                                            RMODE == 0 ? 0 : (int)(rrem < 0x7FFFFFF0L ? rrem : 0x7FFFFFF0L), 0x00020000);
        // FULL: the tile lies inside the tensor (all 128 positions, all 128 channels): no lane is masked, so the stores, the residual loads and
        // the column sums are straight-line code without selects - which is what lets the compiler count the vector-memory queue exactly.
        auto epilogue = [&](auto mode_c, auto full_c) __attribute__((always_inline)) {
        constexpr int MODE = decltype(mode_c)::value;
        constexpr bool FULL = decltype(full_c)::value;
        // sub-block sb = (jp, i, h): dst[r] = the lane's 16 bytes of output row r (position half h of block i, channel-block pair jp)
        auto res_ld = [&](int sb, float4 (&dst)[4]) {
            const int jp = sb >> 3, i = (sb >> 1) & 3, h = sb & 1;
            if constexpr (MODE == 0) {                                     // no residual: nothing to request, nothing to wait for
#pragma unroll
                for (int r = 0; r < 4; ++r) dst[r] = make_float4(0.f, 0.f, 0.f, 0.f);
                return;
            }
            const int le = lane_now();
            const int c = n0 + wn * 64 + 32 * jp + 4 * (le & 7);
            const int rel0 = opix(i, (le >> 3) + 8 * h);
            const long row0 = rel0 < 0 ? -1L : cpix0 + rel0;                 // (the absolute pixel: the resampled modes only)
            const bool okl = FULL ? !(WCONV4_DIAG & 32) : (c < p.N && rel0 >= 0 && !(WCONV4_DIAG & 32));
            if constexpr (MODE == 1) {
                // plain residual: buffer loads that ALWAYS issue (masked lanes read beyond the descriptor: zeros), so the code is straight
                // and the compiler's vector-memory waits count the instructions exactly instead of assuming the worst at every join
                const unsigned o0 = (__umul24((unsigned)rel0, (unsigned)p.N) + (unsigned)c) * 4u;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
                    const u32x4_ q4 = __builtin_amdgcn_raw_buffer_load_b128(rres, okl ? o0 + (unsigned)r * row_bytes : 0xFFFFFFF0u, 0, 0);
                    dst[r] = make_float4(__uint_as_float(q4[0]), __uint_as_float(q4[1]), __uint_as_float(q4[2]), __uint_as_float(q4[3]));
                }
                return;
            }
            // resampled identity skip of an up (MODE 2: nearest 2x upsampling of the half-resolution x) / down (MODE 3: 2x2 average of the
            // double-resolution x) block: the same always-issued buffer loads, relative to the first image the tile touches
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
                const long row = row0 + (long)r * p.W;
                const unsigned t = (unsigned)row >> (31 - __clz(p.W)), xo = (unsigned)row & (p.W - 1), n = t / (unsigned)p.H, yo = t - n * p.H;
                auto ld = [&](long src) -> float4 {
                    const u32x4_ q4 = __builtin_amdgcn_raw_buffer_load_b128(rres, okl ? (unsigned)((src * p.N + c) * 4) : 0xFFFFFFF0u, 0, 0);
                    return make_float4(__uint_as_float(q4[0]), __uint_as_float(q4[1]), __uint_as_float(q4[2]), __uint_as_float(q4[3]));
                };
                if constexpr (MODE == 2) {
                    dst[r] = ld(((long)(n - res_n0) * (p.H >> 1) + (yo >> 1)) * (p.W >> 1) + (xo >> 1));
                } else {
                    // MODE 3: four loads per row would need a 128-register ring to be requested ahead: they stay conditional global loads
                    // (conservative waits), in the order of additions of k_resample2x (denoiser.hip)
                    dst[r] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (okl) {
                        const long W2 = 2 * p.W, src = ((long)n * 2 * p.H + 2 * yo) * W2 + 2 * xo;
                        const float4 a = *(const float4*)(p.res + src * p.N + c), b2 = *(const float4*)(p.res + (src + 1) * p.N + c);
                        const float4 d = *(const float4*)(p.res + (src + W2) * p.N + c), e = *(const float4*)(p.res + (src + W2 + 1) * p.N + c);
                        dst[r] = make_float4((a.x + b2.x + d.x + e.x) * 0.25f, (a.y + b2.y + d.y + e.y) * 0.25f, (a.z + b2.z + d.z + e.z) * 0.25f,
                                             (a.w + b2.w + d.w + e.w) * 0.25f);
                    }
                }
            }
        };
        // column scale * 1 / v_scale and bias of both channel-block pairs (this lane's four channels of each), before any store goes out
        float4 sq_[2], bq_[2];
#pragma unroll
        for (int jp = 0; jp < 2; ++jp) {
            const int c = n0 + wn * 64 + 32 * jp + 4 * (lane_now() & 7);
            if (WCONV4_DIAG & 128) { sq_[jp] = make_float4(1.f, 1.f, 1.f, 1.f); bq_[jp] = make_float4(0.f, 0.f, 0.f, 0.f); continue; }    // (timing only: no scale / bias loads)
            sq_[jp] = *(const float4*)(p.Cscale + c);                        // (padded to the tile grid)
            bq_[jp] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (p.bias && c < p.N) bq_[jp] = *(const float4*)(p.bias + c);
        }
        float4 rn_[2][4];
        res_ld(0, rn_[0]);
        w4_static_for<2>([&](auto jp_c) __attribute__((always_inline)) {
            constexpr int jp = decltype(jp_c)::value;
            float4 sq = sq_[jp];
            const float4 bq = bq_[jp];
            sq.x *= p.out_scale; sq.y *= p.out_scale; sq.z *= p.out_scale; sq.w *= p.out_scale;
            float4 cs = make_float4(0.f, 0.f, 0.f, 0.f), cq2 = cs;
            w4_static_for<4>([&](auto i_c) __attribute__((always_inline)) {
                constexpr int i = decltype(i_c)::value;
                f32x4 R[2][4];                                               // [channel block of the pair][output row]
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int jj = 2 * jp + u;
                    const f32x4 s12 = acc[1][jj][i] + acc[2][jj][i], d12 = acc[1][jj][i] - acc[2][jj][i];
                    const f32x4 s34 = acc[3][jj][i] + acc[4][jj][i], d34 = acc[3][jj][i] - acc[4][jj][i];
                    R[u][0] = (acc[0][jj][i] + s12) + s34;
                    R[u][1] = d12 + 2.0f * d34;
                    R[u][2] = s12 + 4.0f * s34;
                    R[u][3] = (acc[5][jj][i] + d12) + 8.0f * d34;
                }
                w4_static_for<2>([&](auto h_c) __attribute__((always_inline)) {
                    constexpr int h = decltype(h_c)::value;
                    constexpr int sb = (jp * 4 + i) * 2 + h;
                    W4_EPI_FENCE();
                    if constexpr (sb + 1 < 16) res_ld(sb + 1, rn_[(sb + 1) & 1]);
                    W4_EPI_FENCE();
                    const int le = lane_now();
                    const int q = le & 7, c = n0 + wn * 64 + 32 * jp + 4 * q, pl = (le >> 3) + 8 * h;
                    const int rel0 = opix(i, pl);
                    const bool ok = FULL ? !(WCONV4_DIAG & 16) : (c < p.N && rel0 >= 0 && !(WCONV4_DIAG & 16));
                    const unsigned o0 = (__umul24((unsigned)rel0, (unsigned)p.N) + (unsigned)c) * 4u;
                    // The 32 lanes whose accumulator-layout position (lane & 15) lies in half h hold the first block's quads of the 8 rows
                    // this instruction writes; the other 32 lanes are free to carry the second block's (a rotation by 8 inside the 16-lane
                    // row, one DPP move): then ONE ds_bpermute per register gathers a row's eight 16-byte pieces into adjacent lanes.
                    const int src4 = 4 * ((q < 4 ? pl : (pl ^ 8)) + 16 * (q & 3));
                    // all sixteen transposes of the half block are requested before the first is used (one crossbar latency, not four)
                    float t4[4][4];
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            // row_ror:8 with a bank mask: the two banks (8 lanes of every 16) outside half h take the second block's value
                            // from 8 lanes away, the others keep the first block's - one instruction, no select
                            const int y = __builtin_amdgcn_update_dpp(__float_as_int(R[0][r][e]), __float_as_int(R[1][r][e]), 0x128, 0xf, h == 0 ? 0xc : 0x3, false);
                            t4[r][e] = __int_as_float(__builtin_amdgcn_ds_bpermute(src4, y));
                        }
                    W4_EPI_FENCE();
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float (&t)[4] = t4[r];
                        const float4 rq = rn_[sb & 1][r];
                        float4 v;
                        v.x = __builtin_fmaf(t[0], sq.x, bq.x);                   // (one fixed rounding order in every variant of the epilogue)
                        v.y = __builtin_fmaf(t[1], sq.y, bq.y);
                        v.z = __builtin_fmaf(t[2], sq.z, bq.z);
                        v.w = __builtin_fmaf(t[3], sq.w, bq.w);
                        // (no residual: no `+ 0` either - IEEE keeps x + 0.0 as an instruction (it turns -0 into +0): two packed adds and two
                        // wait states per row, 256 instructions per tile of the kernel that is 31 % of the headline job)
                        if constexpr (MODE != 0) { v.x += rq.x; v.y += rq.y; v.z += rq.z; v.w += rq.w; }
                        typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
                        const u32x4_ vb = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
                        __builtin_amdgcn_raw_buffer_store_b128(vb, rc, ok ? o0 + (unsigned)r * row_bytes : 0xFFFFFFF0u, 0, 0);
                        if (FULL || ok) {                                       // (explicit fma: the same rounding in every variant)
                            cs.x += v.x; cs.y += v.y; cs.z += v.z; cs.w += v.w;
                            cq2.x = __builtin_fmaf(v.x, v.x, cq2.x); cq2.y = __builtin_fmaf(v.y, v.y, cq2.y);
                            cq2.z = __builtin_fmaf(v.z, v.z, cq2.z); cq2.w = __builtin_fmaf(v.w, v.w, cq2.w);
                        }
                    }
                });
            });
            if (p.colsum && !(WCONV4_DIAG & 64)) {
                // The wave's 64 positions x 4 output rows are 256 consecutive output pixels of ONE image (H*W % 256 == 0) = four 64-pixel
                // slots of the column-sum array: their sums go to the first, zeros to the other three.  The consumer
                // (dua_groupnorm_stats_from_colsums) folds all slots of an image, so only the per-image totals matter.
                float t[8] = {cs.x, cs.y, cs.z, cs.w, cq2.x, cq2.y, cq2.z, cq2.w};
                const int le = lane_now();                                   // (not __shfl_xor: its hoisted lane id came back from scratch behind a vmcnt(0))
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    t[u] += dua_xor_fresh(t[u], le, 8);
                    t[u] += dua_xor_fresh(t[u], le, 16);
                    t[u] += dua_xor_fresh(t[u], le, 32);
                }
                const int c = n0 + wn * 64 + 32 * jp + 4 * (le & 7), lp_e = le >> 3;
                if (lp_e < 4 && c < p.N && pos0 < p.Mv) {
                    const float z = lp_e == 0 ? 1.0f : 0.0f;
                    float* o = p.colsum + ((long)(pos0 >> 4) + lp_e) * 2 * p.N + c;
                    *(float4*)o = make_float4(t[0] * z, t[1] * z, t[2] * z, t[3] * z);
                    *(float4*)(o + p.N) = make_float4(t[4] * z, t[5] * z, t[6] * z, t[7] * z);
                }
            }
        });
        };
        using EF = std::integral_constant<bool, false>;
        using ET = std::integral_constant<bool, true>;
        if (RMODE == 3 || !tile_full) epilogue(std::integral_constant<int, RMODE>{}, EF{});
        else epilogue(std::integral_constant<int, RMODE>{}, ET{});
        if (WCONV4_DIAG & 64) {                      // (diagnostic: [workgroup][tile of the workgroup][start, all issued, drained] in s_memrealtime ticks of 10 ns)
            const unsigned long long tm_ = __builtin_amdgcn_s_memrealtime();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
            if (tid == 0 && p.colsum) {
                unsigned long long* o = (unsigned long long*)p.colsum + ((size_t)blockIdx.x * 64 + (size_t)(ti / nslot)) * 4;
                if (ti / nslot < 64) { o[0] = t_epi0; o[1] = t1; o[2] = tm_; }
            }
        }
    }
#endif
}

// =================================== weight packer ===========================================
__device__ __forceinline__ double wino4_u(const float* __restrict__ w, int cout, int ci, int dx, int f, int Cin) {
    const float* g = w + ((size_t)cout * Cin + ci) * 9 + dx;             // OIHW: [ky][kx], kx = dx
    const double g0 = g[0], g1 = g[3], g2 = g[6];
    switch (f) {
        case 0: return g0 / 4.0;
        case 1: return -(g0 + g1 + g2) / 6.0;
        case 2: return -(g0 - g1 + g2) / 6.0;
        case 3: return g0 / 24.0 + g1 / 12.0 + g2 / 6.0;
        case 4: return g0 / 24.0 - g1 / 12.0 + g2 / 6.0;
        default: return g2;
    }
}
__global__ __launch_bounds__(256) void k_wino4_row_scale(const float* __restrict__ w, float* __restrict__ inv_scale, int Cout, int Cin) {
    const int n = blockIdx.x;
    float m = 0.f;
    if (n < Cout)
        for (int k = threadIdx.x; k < Cin * 3; k += blockDim.x) {
            const int ci = k / 3, dx = k % 3;
#pragma unroll
            for (int f = 0; f < 6; ++f) m = fmaxf(m, fabsf((float)wino4_u(w, n, ci, dx, f, Cin)));
        }
    __shared__ float red[256];
    red[threadIdx.x] = m;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + o]);
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        int e = 0;
        if (red[0] > 0.f) e = 14 - (int)((__float_as_uint(red[0]) >> 23) & 0xff) + 127;
        e = max(-100, min(100, e));
        inv_scale[n] = __uint_as_float((unsigned)(127 - e) << 23);
    }
}
// Packed layout: [column tile][K step = (channel block, f, dx)][hi | lo][128 channels][32 input channels]: the 16 KB weight stage of a K
// step is one contiguous block in the order of its LDS image.
__global__ void k_pack_wino4(const float* __restrict__ w, _Float16* __restrict__ packed, const float* __restrict__ inv_scale, int Cout,
                             int Cin, int ntn) {
    const int nk = 18 * (Cin / 32);
    const size_t total = (size_t)ntn * nk * 128 * 32;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i & 31), R = (int)((i >> 5) & 127);
        const size_t st = i >> 12;                                       // (column tile, step)
        const int tn = (int)(st / nk), s = (int)(st % nk);
        const int cout = tn * 128 + R;
        const int cb = s / 18, t = s % 18;
        const int f = t / 3, dx = t % 3, ci = cb * 32 + c;
        float v = 0.f;
        if (cout < Cout) v = (float)(wino4_u(w, cout, ci, dx, f, Cin) * (double)(1.0f / inv_scale[cout]));
        const _Float16 h = (_Float16)v;
        _Float16* o = packed + st * (2 * 128 * 32) + R * 32 + c;
        o[0] = h;
        o[128 * 32] = (_Float16)(v - (float)h);
    }
}

static inline int wino4_ntn(int Cout) { return (Cout + 127) / 128; }
extern "C" size_t dua_f16x3_wino4_weight_bytes(int Cout, int Cin) {
    return 4 * (size_t)wino4_ntn(Cout) * 128 * 18 * Cin + 4 * (size_t)((Cout + 255) / 256 * 256);
}
extern "C" int dua_pack_weight_f16x3_wino4(const float* w, void* packed, int Cout, int Cin, void* stream) {
    DUA_REQUIRE(w && packed && Cout > 0 && Cin >= 32 && Cin % 32 == 0, "dua_pack_weight_f16x3_wino4: Cin must be a positive multiple of 32 (got %d)", Cin);
    const int ntn = wino4_ntn(Cout), Nsc = (Cout + 255) / 256 * 256;
    const size_t plane = (size_t)ntn * 128 * 18 * Cin;
    _Float16* hi = (_Float16*)packed;
    float* inv_scale = (float*)(hi + 2 * plane);
    hipLaunchKernelGGL(k_wino4_row_scale, dim3(Nsc), 256, 0, as_stream(stream), w, inv_scale, Cout, Cin);
    hipLaunchKernelGGL(k_pack_wino4, stream_grid(plane, 256), 256, 0, as_stream(stream), w, hi, inv_scale, Cout, Cin, ntn);
    return dua_check_launch("dua_pack_weight_f16x3_wino4");
}

// 0 = not supported, 1 = supported, 2 = supported without the column-sum output (maps smaller than 256 pixels: a wave's 64 positions
// then span several images)
extern "C" int dua_conv3x3_f16x3_wino4_supported(int N, int H, int W, int Cin, int Cout) {
    if (N <= 0 || H <= 0 || W <= 0 || Cin < 32 || Cin % 32 || Cout <= 0 || Cout % 4) return 0;
    if (H % 4 || W < 2 || W > 128 || (W & (W - 1))) return 0;
    const long Mv = (long)N * (H / 4) * W;
    if (Mv >= (1L << 29) || ((long)H * W) % 64) return 0;                           // (16-position groups of the writer stay inside one image)
    if ((size_t)wino4_ntn(Cout) * 128 * 18 * Cin * 4 >= 0x7FFFFFF0ull) return 0;
    return ((long)H * W) % 256 == 0 ? 1 : 2;
}

// Which Winograd form a 3x3 conv of this shape should run in: 0 none, 2 = F(2,3) (dua_conv3x3_f16x3_wino), 4 = F(4,3).  Both kernels are
// persistent over 256 workgroups, so a launch costs whole rounds of tiles; F(4,3) has half as many (each with 1.5 x the products of an
// F(2,3) tile) and runs ~0.82 of F(2,3)'s time per output at equal occupancy (profiles/r04s_convmix_*.txt): it is taken where that
// still wins after the rounding to whole rounds (not the 8^2 level and some 16^2 layers of ADM-128 at batch 128).
extern "C" int dua_conv3x3_f16x3_wino_pick(int N, int H, int W, int Cin, int Cout) {
    const int s2 = dua_conv3x3_f16x3_wino_supported(N, H, W, Cin, Cout), s4 = dua_conv3x3_f16x3_wino4_supported(N, H, W, Cin, Cout);
    if (!s4) return s2 ? 2 : 0;
    if (!s2) return 4;
    if ((s2 == 1) != (s4 == 1)) return 2;                                 // (keep the column-sum output where only F(2,3) has it)
    const long ntn = (Cout + 127) / 128;
    const long t2 = (((long)N * (H / 2) * W + 127) / 128) * ntn, t4 = (((long)N * (H / 4) * W + 127) / 128) * ntn;
    const double c2 = (double)((t2 + 255) / 256) * 1.0, c4 = (double)((t4 + 255) / 256) * 2.0 * 0.82;      // rounds x time per tile
    return c4 < c2 ? 4 : 2;
}

extern "C" int dua_conv3x3_f16x3_wino4(const void* vplanes, const void* packed, const float* bias, const float* residual, int res_resample, float* y,
                                       int N, int H, int W, int Cin, int Cout, float* colsum, float v_scale, void* stream) {
    DUA_REQUIRE(vplanes && packed && y, "dua_conv3x3_f16x3_wino4: null pointer");
    DUA_REQUIRE(res_resample >= -1 && res_resample <= 1 && (res_resample <= 0 || W >= 2), "dua_conv3x3_f16x3_wino4: res_resample must be -1, 0 or +1");
    DUA_REQUIRE(v_scale > 0.f, "dua_conv3x3_f16x3_wino4: v_scale must be positive");
    const int sup = dua_conv3x3_f16x3_wino4_supported(N, H, W, Cin, Cout);
    DUA_REQUIRE(sup, "dua_conv3x3_f16x3_wino4: unsupported shape N=%d H=%d W=%d Cin=%d Cout=%d (H %% 4 == 0, W a power of two <= 128, "
                "Cin %% 32 == 0, Cout %% 4 == 0)", N, H, W, Cin, Cout);
    DUA_REQUIRE(!colsum || sup == 1, "dua_conv3x3_f16x3_wino4: the column-sum output needs H*W %% 256 == 0 (got %d x %d)", H, W);
    WConvP p{};
    p.V = vplanes; p.B = packed; p.bias = bias; p.res = residual; p.res_mode = residual ? res_resample : 0; p.C = y; p.colsum = colsum;
    p.Mv = N * (H / 4) * W; p.N = Cout; p.Cin = Cin; p.K = 18 * Cin; p.H = H; p.W = W;
    p.ntm = (p.Mv + 127) / 128; p.ntn = wino4_ntn(Cout);
    p.b_plane = (long)p.ntn * 128 * p.K;
    p.Cscale = (const float*)((const _Float16*)packed + 2 * p.b_plane);
    p.out_scale = 1.0f / v_scale;
    const int ntiles = p.ntm * p.ntn, grid = ntiles < 256 ? ntiles : 256;
    DuaKTimer kt(stream);
    if (!residual) hipLaunchKernelGGL(k_wconv4<0>, dim3(grid), 256, 0, as_stream(stream), p);
    else if (p.res_mode == 0) hipLaunchKernelGGL(k_wconv4<1>, dim3(grid), 256, 0, as_stream(stream), p);
    else if (p.res_mode > 0) hipLaunchKernelGGL(k_wconv4<2>, dim3(grid), 256, 0, as_stream(stream), p);
    else hipLaunchKernelGGL(k_wconv4<3>, dim3(grid), 256, 0, as_stream(stream), p);
    {   // algorithmic: 2 N H W 9 Cin Cout FLOPs; bytes: the V images once (6 B per input element) + the packed weights + the fp32 output (+ residual)
        const double px = (double)N * H * W;
        kt.done(residual ? "k_wconv4 (residual)" : "k_wconv4", 6.0 * px * Cin + 4.0 * 18.0 * Cin * Cout + 4.0 * px * Cout * (residual ? 2.0 : 1.0), 2.0 * px * 9.0 * Cin * Cout);
    }
    return dua_check_launch("dua_conv3x3_f16x3_wino4");
}

// =================================== GroupNorm apply -> V planes ===========================================
struct GnWino4P {
    const float *x1, *x2, *stats, *gamma, *beta, *film, *drop;
    _Float16* v;
    int C1, C2, N, H, W, G, act, resample;
    int ntm;                 // position tiles of 128 per frequency image
    long film_ld;
    float oscale;
    int JC;                  // row groups per chunk
};

// One wave = 16 consecutive positions of a row group x one 32-channel block: lane -> (position lane >> 2, channels 8 (lane & 3) ..+7),
// so a wave reads 16 whole 128-byte lines per row and writes, per frequency, the 1-KB piece [16 positions][32 channels] of the hi image
// and of the lo image of the conv's stage block (tile, channel block).  Maps narrower than 16 put 16 / W consecutive row groups in a wave.
// STREAM (maps at least 16 wide, no pooling, no dropout mask): the four new rows of the NEXT row group are requested before this group's
// twelve stores go out.  The sliding window over y keeps every input element read once (plus two halo rows per chunk).
// F32: the six images as fp32 stage blocks [128 positions][32 channels] floats (the operand of the exact-fp32 kernel, conv_wino4_f32.hip):
// the same 16 KB per (frequency, tile, channel block) as the [hi | lo] fp16 pair, no scale.
template <bool STREAM, bool F32 = false>
__global__ __launch_bounds__(256) void k_gn_apply_wino4(GnWino4P p) {
    const int C = p.C1 + p.C2, ncb = C / 32, cpg = C / p.G;
    const int Ho = p.resample < 0 ? p.H / 2 : (p.resample > 0 ? 2 * p.H : p.H), Wo = p.resample < 0 ? p.W / 2 : (p.resample > 0 ? 2 * p.W : p.W);
    const int P = Ho / 4;
    const int JS = Wo >= 16 ? 1 : 16 / Wo;                 // row groups covered by one 16-position group
    const int npg = Wo >= 16 ? Wo / 16 : 1;                // position groups per row-group row
    const int nitems = npg * ncb, nxb = (nitems + 3) / 4;
    const int n = blockIdx.y, jc = blockIdx.x / nxb, item = (blockIdx.x % nxb) * 4 + (threadIdx.x >> 6);
    if (item >= nitems) return;
    const int lane = threadIdx.x & 63, p16 = lane >> 2;
    const int cbk = item % ncb, pg = item / ncb;
    const int c = cbk * 32 + 8 * (lane & 3);
    const int xo = Wo >= 16 ? pg * 16 + p16 : p16 % Wo, jo = Wo >= 16 ? 0 : p16 / Wo;
    const bool second = c >= p.C1;
    const float* src = second ? p.x2 : p.x1;
    const int cs = second ? p.C2 : p.C1, cq = second ? c - p.C1 : c;
    float ca[8], cb[8];
    {
        const float4 g0 = *(const float4*)(p.gamma + c), g1 = *(const float4*)(p.gamma + c + 4);
        const float4 b0 = *(const float4*)(p.beta + c), b1 = *(const float4*)(p.beta + c + 4);
        float2 st[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) st[j] = *(const float2*)(p.stats + 2 * ((size_t)n * p.G + (c + j) / cpg));
        float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0, h0 = s0, h1 = s0;
        if (p.film) {
            const float* fr = p.film + (size_t)n * p.film_ld + c;
            s0 = *(const float4*)fr; s1 = *(const float4*)(fr + 4);
            h0 = *(const float4*)(fr + C); h1 = *(const float4*)(fr + C + 4);
        }
        const float g[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w}, bt[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
        const float fs[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w}, fh[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float a = st[j].y * g[j];
            float b = bt[j] - st[j].x * a;
            if (p.film) {
                const float sc = 1.0f + fs[j];
                a *= sc;
                b = b * sc + fh[j];
            }
            ca[j] = a;
            cb[j] = b;
        }
    }
    const size_t img_in = (size_t)n * p.H * p.W;
    auto xf = [&](size_t pix, float (&r)[8]) {
        const float4 v0 = *(const float4*)(src + pix * cs + cq), v1 = *(const float4*)(src + pix * cs + cq + 4);
        const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            r[j] = __builtin_fmaf(v[j], ca[j], cb[j]);
            if (p.act == 1) r[j] = silu_f(r[j]);
        }
        if (p.drop) {
            const float4 d0 = *(const float4*)(p.drop + pix * C + c), d1 = *(const float4*)(p.drop + pix * C + c + 4);
            r[0] *= d0.x; r[1] *= d0.y; r[2] *= d0.z; r[3] *= d0.w; r[4] *= d1.x; r[5] *= d1.y; r[6] *= d1.z; r[7] *= d1.w;
        }
    };
    // value of the conv's input image (after the optional 2x resampling) at row yo of this thread's column, zero outside
    auto val = [&](int yo, float (&r)[8]) {
        if (yo < 0 || yo >= Ho) {
#pragma unroll
            for (int j = 0; j < 8; ++j) r[j] = 0.f;
            return;
        }
        if (p.resample == 0) {
            xf(img_in + (size_t)yo * p.W + xo, r);
        } else if (p.resample > 0) {
            xf(img_in + (size_t)(yo >> 1) * p.W + (xo >> 1), r);
        } else {
            const size_t pix = img_in + (size_t)(2 * yo) * p.W + 2 * xo;
            float a[8], b[8], d[8], e[8];
            xf(pix, a); xf(pix + 1, b); xf(pix + p.W, d); xf(pix + p.W + 1, e);
#pragma unroll
            for (int j = 0; j < 8; ++j) r[j] = (a[j] + b[j] + d[j] + e[j]) * 0.25f;
        }
    };
    unsigned long long* const rctr = dua_range_ctr();          // f16x3 range check (debug facility; nullptr when off)
    auto put = [&](int f, size_t m, const float (&v)[8]) {
        if constexpr (F32) {
            float* o = (float*)p.v + (((size_t)f * p.ntm + (m >> 7)) * ncb + cbk) * 4096 + (m & 127) * 32 + (c & 31);
            __builtin_nontemporal_store((f32x4){v[0], v[1], v[2], v[3]}, (f32x4*)o);
            __builtin_nontemporal_store((f32x4){v[4], v[5], v[6], v[7]}, (f32x4*)(o + 4));
            return;
        }
        h16x8 h, l;
        if (rctr) {
#pragma unroll
            for (int j = 0; j < 8; ++j) dua_range_note(rctr, v[j] * p.oscale);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float sv = fminf(fmaxf(v[j] * p.oscale, -65504.f), 65504.f);
            const _Float16 t = (_Float16)sv;
            h[j] = t;
            l[j] = (_Float16)(sv - (float)t);
        }
        // stage block (f, tile m / 128, channel block): [hi | lo][128 positions][32 channels]
        _Float16* o = p.v + ((((size_t)f * p.ntm + (m >> 7)) * ncb + cbk) * 2) * 4096 + (m & 127) * 32 + (c & 31);
        __builtin_nontemporal_store(h, (h16x8*)o);
        __builtin_nontemporal_store(l, (h16x8*)(o + 4096));
    };
    // the six transformed rows of one row group from its six input rows
    auto emit = [&](size_t m, const float (&d0)[8], const float (&d1)[8], const float (&d2)[8], const float (&d3)[8], const float (&d4)[8], const float (&d5)[8]) {
        // (explicit fma: the row groups of a chunk are emitted by two inlined copies of this code - the first group and the loop body - and
        // a sample's result must not depend on which copy, i.e. on the chunking, i.e. on the batch size)
        float v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = __builtin_fmaf(4.0f, d0[q], __builtin_fmaf(-5.0f, d2[q], d4[q]));
        put(0, m, v);
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = __builtin_fmaf(-4.0f, d1[q] + d2[q], d3[q] + d4[q]);
        put(1, m, v);
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = __builtin_fmaf(4.0f, d1[q] - d2[q], d4[q] - d3[q]);
        put(2, m, v);
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = __builtin_fmaf(2.0f, d3[q] - d1[q], d4[q] - d2[q]);
        put(3, m, v);
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = __builtin_fmaf(-2.0f, d3[q] - d1[q], d4[q] - d2[q]);
        put(4, m, v);
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = __builtin_fmaf(4.0f, d1[q], __builtin_fmaf(-5.0f, d3[q], d5[q]));
        put(5, m, v);
    };
    const int j0 = jc * p.JC + jo, j1 = min(P, (jc + 1) * p.JC);
    float dA[8], dB[8], dC[8], dD[8], dE[8], dF[8];
    if constexpr (STREAM) {
        if (j0 >= j1) return;
        struct Raw { float4 a, b; };
        auto ld = [&](int yo) -> Raw {                                   // (rows outside the image: any valid address, zeroed in `fin`)
            const int yc = min(max(yo, 0), Ho - 1);
            const size_t pix = img_in + (p.resample > 0 ? (size_t)(yc >> 1) * p.W + (xo >> 1) : (size_t)yc * p.W + xo);
            return Raw{*(const float4*)(src + pix * cs + cq), *(const float4*)(src + pix * cs + cq + 4)};
        };
        auto fin = [&](const Raw& w, int yo, float (&r)[8]) {
            const float v[8] = {w.a.x, w.a.y, w.a.z, w.a.w, w.b.x, w.b.y, w.b.z, w.b.w};
            const bool in = yo >= 0 && yo < Ho;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                float t = __builtin_fmaf(v[q], ca[q], cb[q]);
                if (p.act == 1) t = silu_f(t);
                r[q] = in ? t : 0.f;
            }
        };
        Raw rA = ld(4 * j0 - 1), rB = ld(4 * j0), rC = ld(4 * j0 + 1), rD = ld(4 * j0 + 2), rE = ld(4 * j0 + 3), rF = ld(4 * j0 + 4);
        fin(rA, 4 * j0 - 1, dA);
        fin(rB, 4 * j0, dB);
        auto group = [&](int j) {
            fin(rC, 4 * j + 1, dC);
            fin(rD, 4 * j + 2, dD);
            fin(rE, 4 * j + 3, dE);
            fin(rF, 4 * j + 4, dF);
            if (j + 1 < j1) { rC = ld(4 * j + 5); rD = ld(4 * j + 6); rE = ld(4 * j + 7); rF = ld(4 * j + 8); }
            emit(((size_t)n * P + j) * Wo + xo, dA, dB, dC, dD, dE, dF);
#pragma unroll
            for (int q = 0; q < 8; ++q) { dA[q] = dE[q]; dB[q] = dF[q]; }
        };
        group(j0);
        for (int j = j0 + 1; j < j1; ++j) group(j);
        return;
    }
    if (j0 < j1) { val(4 * j0 - 1, dA); val(4 * j0, dB); }
    for (int j = j0; j < j1; j += JS) {
        val(4 * j + 1, dC);
        val(4 * j + 2, dD);
        val(4 * j + 3, dE);
        val(4 * j + 4, dF);
        emit(((size_t)n * P + j) * Wo + xo, dA, dB, dC, dD, dE, dF);
        if (JS == 1) {
#pragma unroll
            for (int q = 0; q < 8; ++q) { dA[q] = dE[q]; dB[q] = dF[q]; }
        } else if (j + JS < j1) {
            val(4 * (j + JS) - 1, dA);
            val(4 * (j + JS), dB);
        }
    }
}

// bytes of the V planes of a conv input [N][Ho][Wo][C] (Ho, Wo after the resampling): 6 frequencies x 2 planes x N*(Ho/4)*Wo*C halves
extern "C" size_t dua_f16x3_wino4_act_bytes(int N, int Ho, int Wo, int C) {
    const size_t Mv = (size_t)N * (Ho / 4) * Wo;
    return (size_t)24 * ((Mv + 127) / 128 * 128) * C;                 // whole tiles of 128 positions
}

static int gn_apply_wino4_impl(const float* x1, int C1, const float* x2, int C2, int N, int H, int W, int G,
                               const float* stats, const float* gamma, const float* beta, const float* film, long film_ld,
                               int act, int resample, const float* dropmask, float v_scale, void* vplanes, void* stream, bool f32_planes) {
    const int C = C1 + C2;
    DUA_REQUIRE(x1 && stats && gamma && beta && vplanes, "dua_groupnorm_apply_wino4_f16x3: null pointer");
    DUA_REQUIRE(C % G == 0 && C1 % 8 == 0 && C2 % 8 == 0 && C % 32 == 0, "dua_groupnorm_apply_wino4_f16x3: channel counts must be multiples of 8 (sum of 32)");
    DUA_REQUIRE((C2 == 0) == (x2 == nullptr), "dua_groupnorm_apply_wino4_f16x3: x2/C2 mismatch");
    DUA_REQUIRE(resample >= -1 && resample <= 1 && (resample >= 0 || (H % 2 == 0 && W % 2 == 0)), "dua_groupnorm_apply_wino4_f16x3: pooling needs even H, W");
    DUA_REQUIRE(!(dropmask && resample != 0), "dua_groupnorm_apply_wino4_f16x3: dropout mask with resampling unsupported");
    DUA_REQUIRE(v_scale > 0.f && (__builtin_bit_cast(unsigned, v_scale) & 0x7FFFFFu) == 0, "dua_groupnorm_apply_wino4_f16x3: v_scale must be a power of two");
    const int Ho = resample < 0 ? H / 2 : (resample > 0 ? 2 * H : H), Wo = resample < 0 ? W / 2 : (resample > 0 ? 2 * W : W);
    DUA_REQUIRE(Ho >= 4 && Ho % 4 == 0, "dua_groupnorm_apply_wino4_f16x3: the conv input needs a multiple of four rows (got %d)", Ho);
    if ((size_t)N * Ho * Wo == 0) return DUA_OK;
    DUA_REQUIRE(C1 % 32 == 0 && Wo >= 2 && Wo <= 128 && (Wo & (Wo - 1)) == 0 && ((long)Ho * Wo) % 64 == 0,
                "dua_groupnorm_apply_wino4_f16x3: C1 %% 32 == 0, the conv input width a power of two <= 128 and H*W %% 64 == 0 are required");
    GnWino4P p{x1, x2, stats, gamma, beta, film, dropmask, (_Float16*)vplanes, C1, C2, N, H, W, G, act, resample, 0, film_ld ? film_ld : 2L * C, v_scale, 0};
    DUA_REQUIRE(p.film_ld >= 2L * C, "dua_groupnorm_apply_wino4_f16x3: film_ld too small");
    DUA_REQUIRE((((uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)film) & 15) == 0 && (!film || p.film_ld % 4 == 0),
                "dua_groupnorm_apply_wino4_f16x3: gamma, beta and the FiLM rows must be 16-byte aligned");
    const int P = Ho / 4;
    p.ntm = (int)(((long)N * P * Wo + 127) / 128);
    const int nxb = ((Wo >= 16 ? Wo / 16 : 1) * (C / 32) + 3) / 4;
    // Row groups per workgroup chunk: 4 (two halo rows re-read per 16 = +12 % reads), fewer while the grid would be small
    int JC = P < 4 ? P : 4;
    while ((long)N * nxb * ((P + JC - 1) / JC) < 1024 && JC > 2) JC = (JC + 1) / 2;
    if (Wo < 16 && JC % (16 / Wo)) JC = (JC / (16 / Wo) + 1) * (16 / Wo);       // chunks hold whole 16-position groups
    p.JC = JC;
    const dim3 grid(nxb * ((P + JC - 1) / JC), N);
    const bool strm = resample >= 0 && !dropmask && Wo >= 16;
    DuaKTimer kt(stream);
    if (f32_planes) {
        if (strm) hipLaunchKernelGGL((k_gn_apply_wino4<true, true>), grid, 256, 0, as_stream(stream), p);
        else hipLaunchKernelGGL((k_gn_apply_wino4<false, true>), grid, 256, 0, as_stream(stream), p);
    } else {
        if (strm) hipLaunchKernelGGL((k_gn_apply_wino4<true>), grid, 256, 0, as_stream(stream), p);
        else hipLaunchKernelGGL((k_gn_apply_wino4<false>), grid, 256, 0, as_stream(stream), p);
    }
    // algorithmic bytes: the fp32 input once + the six V images of the conv's input positions (24 B per group of four rows = 6 B per element)
    kt.done(f32_planes ? "k_gn_apply_wino4 (fp32 V images)" : "k_gn_apply_wino4", 4.0 * N * H * W * C + 6.0 * N * Ho * Wo * C);
    return dua_check_launch(f32_planes ? "dua_groupnorm_apply_wino4_f32" : "dua_groupnorm_apply_wino4_f16x3");
}

extern "C" int dua_groupnorm_apply_wino4_f16x3(const float* x1, int C1, const float* x2, int C2, int N, int H, int W, int G,
                                               const float* stats, const float* gamma, const float* beta, const float* film, long film_ld,
                                               int act, int resample, const float* dropmask, float v_scale, void* vplanes, void* stream) {
    return gn_apply_wino4_impl(x1, C1, x2, C2, N, H, W, G, stats, gamma, beta, film, film_ld, act, resample, dropmask, v_scale, vplanes, stream, false);
}
// ... with the six images as fp32 planes (same bytes: dua_f16x3_wino4_act_bytes): the operand of dua_conv3x3_f32_wino4
extern "C" int dua_groupnorm_apply_wino4_f32(const float* x1, int C1, const float* x2, int C2, int N, int H, int W, int G,
                                             const float* stats, const float* gamma, const float* beta, const float* film, long film_ld,
                                             int act, int resample, const float* dropmask, void* vplanes, void* stream) {
    return gn_apply_wino4_impl(x1, C1, x2, C2, N, H, W, G, stats, gamma, beta, film, film_ld, act, resample, dropmask, 1.0f, vplanes, stream, true);
}

DUA_RANGE_TU(conv_wino4)
