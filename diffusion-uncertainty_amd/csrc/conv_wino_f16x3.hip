// f16x3 3x3 convolution with a Winograd F(2,3) transform ALONG Y (gfx950): two output rows cost four 1x3 row convolutions
// instead of six, i.e. 6 instead of 9 matrix-core taps per output pixel (x 3 fp16 products each) at unchanged operand precision
// (22 significant bits per operand, fp32 accumulation; the transform adds / halves are exact or one fp32 rounding).
//
//   d0..d3 = input rows 2j-1, 2j, 2j+1, 2j+2 (zero outside the image)        g0..g2 = kernel rows (dy = -1, 0, +1)
//   V0 = d0 - d2   V1 = d1 + d2   V2 = d2 - d1   V3 = d1 - d3                U0 = g0  U1 = (g0+g1+g2)/2  U2 = (g0-g1+g2)/2  U3 = g2
//   M_f = conv1x3_x(V_f, U_f)  (zero padding in x)                           out row 2j = M0 + M1 + M2,  out row 2j+1 = M1 - M2 - M3
//
// Three pieces:
//   k_gn_apply_wino      GroupNorm * gamma + beta (+FiLM, SiLU, dropout mask, 2x resampling, virtual concat) -> the four V_f images
//                        as f16x3 images [f][tile of 128 positions][C/32][hi | lo][128][32]: the 16 KB the conv stages per
//                        (tile, channel block, frequency) are contiguous in its LDS-image order (the sliding window over y keeps
//                        every input element read once)
//   k_pack_wino          OIHW fp32 weights -> U_f, per-output-row power-of-two scale, fp16 hi | lo images, one contiguous 32 KB
//                        block per (column tile, K step) in the order of the kernel's LDS weight stage
//   k_wconv              LDS-DMA implicit GEMM on v_mfma_f32_16x16x32_f16 (the loop of conv_f16x3.hip's row-shift kernel): a
//                        workgroup tile = 128 positions of the V images (whole image rows of W | 128 positions) x 128 output channels
//                        x 4 frequencies = 65 536 accumulators.  Waves 0-3 hold f = 1, 2 and waves 4-7 f = 0, 3 of 64 positions x 64
//                        channels each; K steps (channel block, fi, dx): ONE 256-row activation stage [group][128 positions] per
//                        (channel block, fi) serves the three dx taps through row-shifted fragment addresses (edge lanes read zeros
//                        through an LDS address beyond the allocation), one 256-row weight stage [group][128 channels] per step.
//                        The epilogue exchanges P = M1 + M2 / M3 between partner waves through LDS, so the f = 1, 2 wave stores the
//                        odd output row (M1 - M2) - M3 and the f = 0, 3 wave the even row M0 + P, with scale, bias, residual and the
//                        next GroupNorm's column sums as in the direct kernels.
// LDS image, DMA piece shape and ring discipline are those of k_hconv<4, true> (conv_f16x3.hip); the chunk swizzle is its own.
#include "common.h"
#include "wconv.h"
#include <type_traits>

#define BLK 256

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));

#define WROWS 256
#define WLD 32
// Timing-only diagnostic builds (scripts/wconv_diag.py; WRONG results, never in the product library): WCONV_DIAG bit 0 drops the
// per-step barrier, bit 1 the in-loop weight DMA, bit 2 the in-loop activation DMA, bit 3 the in-loop fragment reads; bit 4 reads the
// activations of every tile from the first 32 tiles (prices the L2-miss path of the activation stream).
#ifndef WCONV_DIAG
#define WCONV_DIAG 0
#endif

__global__ __launch_bounds__(512, 2) void k_wconv(WConvP p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int RA = 2, RB = 3;
    constexpr int SA = WROWS * WLD, SB = WROWS * WLD;      // halves per plane of a stage
    constexpr int BOFF = 0, AOFF = RB * 2 * SB;            // weight ring first: a shifted row -1 of activation stage 0 stays inside LDS
    constexpr int NBP = 4;                                 // weight DMA pieces per wave and stage
    __shared__ __attribute__((aligned(16))) _Float16 lds[RA * 2 * SA + RB * 2 * SB];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;               // wm = 2 * group + position half
    const int grp_w = wave >> 2, ph = wm & 1;
    const int ntiles = p.ntm * p.ntn;
    const int nk = p.K / 32;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslot = (gridDim.x + 7 - xcd) >> 3;
    const int cq = ntiles >> 3, cr = ntiles & 7;
    const int chunk0 = xcd < cr ? xcd * (cq + 1) : cr * (cq + 1) + (xcd - cr) * cq, chunk_n = cq + (xcd < cr ? 1 : 0);

    constexpr int OOBL = 0x40000;                          // LDS byte address beyond the allocation: DS reads there return zeros
    static_assert(OOBL >= 160 * 1024 + 64 * 64 + 64, "the out-of-range LDS address must lie beyond the 160 KB of a gfx950 CU");
    const int dl_row = lane >> 2;
    // LDS image: rows of 64 B, 16-byte chunk c of row r at chunk position c ^ (2 * ((r >> 2) & 1)).  For the lane groups of a
    // ds_read_b128 (lane -> row (l & 15) + dx, chunk l >> 4) this is conflict-free for dx = -1, 0 AND +1 (exhaustive search over
    // the 4-entry swizzle tables; the k_hconv swizzle c ^ ((-(r >> 2)) & 3) is conflict-free for dx = 0 only: its shifted reads
    // were 23 % of this kernel's LDS cycles).  The DMA writes lane-linear pieces, so the swizzle is applied to the SOURCE chunk.
    const int dl_src = (lane & 3) ^ (2 * ((lane >> 4) & 1));
    const int lr = lane & 15, lc = lane >> 4;

    // ---- DMA state
    unsigned abase[2] = {0, 0}, boff[2] = {0, 0};
    auto ra0 = __builtin_amdgcn_make_buffer_rsrc((void*)p.V, 0, 0, 0x00020000);
    auto ra1 = ra0;
    const auto rb = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, (int)(2 * p.b_plane * 2), 0x00020000);
    const int b_lo = 256 * 32 * 2;                         // byte distance hi -> lo image inside a weight stage
    const int a_lo = 128 * 32 * 2;                         // byte distance hi -> lo image inside an activation stage block
    int grpA = 0;                                          // next activation group to request: channel block grpA >> 1, fi = grpA & 1
    unsigned vo0 = 0, vo1 = 0;
    bool prefetched = false;
    const int f_a = grp_w == 0 ? 1 : 0, f_b = grp_w == 0 ? 2 : 3;        // this wave's frequencies for fi = 0 / 1

    const int ncb = p.Cin / 32;
    auto tile_geom = [&](int m0_, int tn_) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            abase[j] = (unsigned)((16 * ((2 * wave + j) & 7) + dl_row) * 64 + 16 * dl_src);     // this lane's 16 bytes of the 128-row image
            boff[j] = (unsigned)((long)tn_ * (p.K / 32) * 32768 + (16 * (2 * wave + j) + dl_row) * 64 + 16 * dl_src);   // stage 0 of the column tile
        }
        // one tile of one frequency = ncb stage blocks of 16 KB ([hi | lo][128 positions][32 channels]), contiguous
        const int rec = ncb * 16384;
        const long tmi = (WCONV_DIAG & 16) ? (m0_ >> 7) & 31 : m0_ >> 7;   // (diagnostic: activations from a cache-resident set of tiles)
        ra0 = __builtin_amdgcn_make_buffer_rsrc((void*)((const _Float16*)p.V + (((long)f_a * p.ntm + tmi) * ncb) * 8192), 0, rec, 0x00020000);
        ra1 = __builtin_amdgcn_make_buffer_rsrc((void*)((const _Float16*)p.V + (((long)f_b * p.ntm + tmi) * ncb) * 8192), 0, rec, 0x00020000);
        grpA = 0;
    };
    bool fiA = false;                                     // frequency index of the group being requested (selects the descriptor)
    auto a_setup = [&]() {
        const unsigned toff = (unsigned)((grpA >> 1) * 16384);            // channel block cb: the cb-th stage block of the tile
        vo0 = abase[0] + toff;
        vo1 = abase[1] + toff;
        fiA = (grpA & 1) != 0;
        ++grpA;
    };
    auto dma_a = [&](int sa, int i) {
        _Float16* st = lds + AOFF + sa * 2 * SA;
        const auto r = fiA ? ra1 : ra0;
        if (i == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, st + (16 * (2 * wave)) * WLD, 16, vo0, 0, 0, 0);
        if (i == 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, st + (16 * (2 * wave)) * WLD + SA, 16, vo0, a_lo, 0, 0);
        if (i == 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, st + (16 * (2 * wave + 1)) * WLD, 16, vo1, 0, 0, 0);
        if (i == 3) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, st + (16 * (2 * wave + 1)) * WLD + SA, 16, vo1, a_lo, 0, 0);
    };
    auto dma_b = [&](int sb, int i) {
        _Float16* st = lds + BOFF + sb * 2 * SB;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            _Float16* dst = st + (16 * (2 * wave + j)) * WLD;
            if (i == 2 * j) __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, dst, 16, boff[j], 0, 0, 0);
            if (i == 2 * j + 1) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, dst + SB, 16, boff[j], b_lo, 0, 0);
                boff[j] += 32768;                          // the next K step's stage
            }
        }
    };
    auto issue_a = [&](int sa) {
        a_setup();
#pragma unroll
        for (int i = 0; i < 4; ++i) dma_a(sa, i);
    };
    auto issue_b = [&](int sb) {
#pragma unroll
        for (int i = 0; i < NBP; ++i) dma_b(sb, i);
    };
    auto issue_prologue = [&]() {          // A(group 0) B(0) B(1)
        issue_a(0);
        issue_b(0);
        issue_b(1);
    };

    const char* ldsb = reinterpret_cast<const char*>(lds);
    const int off_l = lr * 64 + 16 * (lc ^ (2 * ((lr >> 2) & 1)));
    const int a_row0 = AOFF * 2 + wm * 64 * 64, b_row0 = BOFF * 2 + (grp_w * 128 + wn * 64) * 64;     // bytes, wave-uniform
    // Loop-invariant fragment addresses.  A tile is 128 consecutive positions starting at a multiple of 128 and W divides 128, so the x
    // coordinate of (position half, block i, lane row) - hence which lanes sit at the image's left / right edge - is the same for
    // every tile.  Row r + dx of the wave's 64 rows: the swizzle term depends on (lr + dx) only (16 i is a multiple of 4 rows * 4).
    auto x_off = [&](int dx) -> int {
        const int r_ = lr + dx;                                           // -1 .. 16
        return r_ * 64 + 16 * (lc ^ (2 * (((r_ + 64) >> 2) & 1)));
    };
    // The image-edge lanes take their dx = -1 / +1 fragments from an LDS address beyond the allocation (DS reads there return
    // zeros); -DDUA_RS_ZERO_SELECT=1 builds the data-select form instead (the fragment is read from its in-range address and
    // replaced by zeros in registers), compared bit for bit by tests/test_gpu_wino.py.
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
        const int x_ = (ph * 64 + 16 * i + lr) & (p.W - 1);
        xedge |= (x_ == 0 ? 1u : 0u) << i | (x_ == p.W - 1 ? 1u : 0u) << (4 + i);
        axm[i] = (!ZSEL && x_ == 0) ? zero_addr : a_row0 + x_off(-1);
        axp[i] = (!ZSEL && x_ == p.W - 1) ? zero_addr : a_row0 + x_off(1);
    }
    const int aw0 = b_row0 + off_l, aw2 = aw0 + 2 * 2 * SB * 2;

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

        f32x4 acc0[4][4], acc1[4][4];                // [channel block][position block] of fi = 0 / 1
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) { acc0[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc1[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f}; }

        h16x8 xh[4], xl[4], wh[4], wl[4];
        // fragment reads: one ds_read_b128 each, the address a loop-invariant per-lane register (ax*: activation rows shifted by dx,
        // with the image-edge lanes already pointed at the zero region; aw*: weight rows) plus an immediate (stage, plane, block)
        auto rd_x = [&](int sa, int pl, int i, int dx) -> h16x8 {
            const int a = dx < 0 ? axm[i] : (dx > 0 ? axp[i] : ax0);
            const h16x8 v = *(const h16x8*)(ldsb + a + (sa * 2 * SA * 2 + pl * SA * 2 + 16 * i * 64));
            if (ZSEL && dx != 0) {
                const h16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
                return ((xedge >> (dx < 0 ? i : 4 + i)) & 1u) ? z : v;
            }
            return v;
        };
        auto rd_w = [&](int sb, int pl, int jj) -> h16x8 {
            return *(const h16x8*)(ldsb + (sb == 2 ? aw2 : aw0) + ((sb == 2 ? 0 : sb * 2 * SB * 2) + pl * SB * 2 + 16 * jj * 64));
        };
#define WSB() __builtin_amdgcn_sched_barrier(0)

        // the first stages have landed; behind them in the (in-order) queue are only the previous epilogue's operations, of which the
        // last WCONV_EPILOGUE_STORES - output stores - may stay in flight
        if (after_epilogue) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WCONV_EPILOGUE_STORES) : "memory");      // (derived from the store loops: wconv.h)
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int i = 0; i < 4; ++i) { xl[i] = rd_x(0, 1, i, -1); xh[i] = rd_x(0, 0, i, -1); }
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) { wh[jj] = rd_w(0, 0, jj); wl[jj] = rd_w(0, 1, jj); }
        WSB();

        // One K step = (channel block, fi, dx): 48 MFMAs into the accumulators of fi.  Ring slots are static: the activation stage of
        // (channel block, fi) lives in slot fi, the weight stage of step dxi in slot dxi (three steps per group, ring of three).
        // LAST = the tile's final group (nothing further to request after its first step, nothing to read after its last).
        auto step = [&](auto fi_c, auto dxi_c, auto last_c, f32x4 (&A)[4][4]) __attribute__((always_inline)) {
            constexpr int FI = decltype(fi_c)::value, DXI = decltype(dxi_c)::value;
            constexpr bool LAST = decltype(last_c)::value;
            constexpr bool preA = !LAST && DXI == 0;                            // set up the next group's activation stage ...
            constexpr bool pieceA = !LAST && DXI < 2;                           // ... requested half in this group's first, half in its second step
            constexpr bool preB = !LAST || DXI == 0;                            // request the weight stage of step s + 2
            constexpr bool more = (!LAST || DXI < 2) && !(WCONV_DIAG & 8);
            constexpr int sa_n = FI ^ 1, sb_n = (DXI + 2) % 3;
            constexpr int sa_r = DXI == 2 ? (FI ^ 1) : FI, sb_r = (DXI + 1) % 3, dxn = DXI == 2 ? -1 : DXI;   // the NEXT step's fragments
            const bool idle_ = (WCONV_DIAG & 32) && FI == 1 && wave < 4;       // (diagnostic: the f = 2 products skipped, as if V2 == 0)
            if (preA) { a_setup(); fiA = FI == 0; }
            auto early_piece = [&](int q) {
                if (q < NBP) { if (preB && !(WCONV_DIAG & 2)) dma_b(sb_n, q); }
                else if (q < NBP + 2) { if (pieceA && !(WCONV_DIAG & 4)) dma_a(sa_n, q - NBP + 2 * DXI); }
            };
            WSB();
            // P1: w_hi * x_lo | the DMA pieces of this step
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (!idle_) A[jj][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[jj], xl[i], A[jj][i], 0, 0, 0);
                    WSB();
                    const int m_ = jj * 4 + i;
                    if ((m_ & 1) == 0) early_piece(m_ >> 1);
                    WSB();
                }
            // every read of stage s was issued during the previous step: publish stage s + 1
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            {
                // in issue order: [B(s+1) x4, A x2] of the previous step, [B(s+2) x4, A x2] of this one.  B(s+1) must have landed;
                // the group's last step also needs all four activation pieces
                constexpr int keep = LAST ? (preB ? NBP : 0) : (DXI == 0 ? NBP + 2 : (DXI == 1 ? NBP + 4 : NBP));
                if (keep == NBP + 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                else if (keep == NBP + 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                else if (keep == NBP) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            if (!(WCONV_DIAG & 1)) __builtin_amdgcn_s_barrier();
            WSB();
            // P2: w_hi * x_hi | x_lo and w_hi of the next step
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (!idle_) A[jj][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[jj], xh[i], A[jj][i], 0, 0, 0);
                    WSB();
                    if (jj == 0 && more) { xl[i] = rd_x(sa_r, 1, i, dxn); WSB(); }
                }
                if (more) wh[jj] = rd_w(sb_r, 0, jj);
                WSB();
            }
            // P3: w_lo * x_hi | w_lo and x_hi of the next step
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (!idle_) A[jj][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[jj], xh[i], A[jj][i], 0, 0, 0);
                    WSB();
                    if (jj == 3 && more) { xh[i] = rd_x(sa_r, 0, i, dxn); WSB(); }
                }
                if (more) wl[jj] = rd_w(sb_r, 1, jj);
                WSB();
            }
        };
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        using I2 = std::integral_constant<int, 2>;
        using BF = std::integral_constant<bool, false>;
        using BT = std::integral_constant<bool, true>;
#pragma nounroll
        for (int cbi = 0; cbi < ncb - 1; ++cbi) {
            step(I0{}, I0{}, BF{}, acc0); step(I0{}, I1{}, BF{}, acc0); step(I0{}, I2{}, BF{}, acc0);
            step(I1{}, I0{}, BF{}, acc1); step(I1{}, I1{}, BF{}, acc1); step(I1{}, I2{}, BF{}, acc1);
        }
        step(I0{}, I0{}, BF{}, acc0); step(I0{}, I1{}, BF{}, acc0); step(I0{}, I2{}, BF{}, acc0);
        step(I1{}, I0{}, BT{}, acc1); step(I1{}, I1{}, BT{}, acc1); step(I1{}, I2{}, BT{}, acc1);
        // the last barrier of the tile has retired every LDS read: request the first stages of the next tile now (they go to
        // activation stage 0 and weight stages 0, 1; the epilogue uses activation stage 1 and weight stage 2)
        if (ti + nslot < chunk_n) {
            const int nl = chunk0 + ti + nslot;
            tile_geom((nl / p.ntn) * 128, nl % p.ntn);
            issue_prologue();
            prefetched = true;
        }
#undef WSB

        if (WCONV_DIAG & 64) {                       // (diagnostic: no epilogue - the accumulators are kept alive by a store that never happens)
            float t_ = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) t_ += acc0[j][i][0] + acc0[j][i][1] + acc0[j][i][2] + acc0[j][i][3] + acc1[j][i][0] + acc1[j][i][1] + acc1[j][i][2] + acc1[j][i][3];
            if (t_ == 1.2345e-30f) p.C[0] = t_;
        } else
        wconv_epilogue(p, acc0, acc1, reinterpret_cast<char*>(lds), wave, lane, m0, n0);
        // the next tile's fragment reads wait for its opening barrier; the patches above must be done before another wave's
        // exchange writes of the next tile land in them: that barrier orders them too (the writes come after the K loop)
    }
#endif
}

// ---------------------------------------------------------------------------------------------------------------------------------
// k_wconv1: the same convolution with ONE fp16 product per tap (precision mode "f16w": fp16 operands, fp32 accumulation - the arithmetic
// of the reference's own GPU path, torch.autocast fp16, scripts/generate_dataset_score_uncertainty_imagenet.py:83; GroupNorm and
// softmax stay fp32 as nn.py:17-19).  Same tile, rings, stage blocks and epilogue as k_wconv; only the hi images of the activation and
// weight stages are fetched and multiplied (half the DMA pieces, a third of the MFMAs).  A step is then 16 MFMAs per wave, too short to
// hide a fragment read behind a later phase, so the fragments are double-buffered in registers: right after the barrier that publishes
// stage s + 1 the wave requests that stage's fragments into the other set and then issues the 16 MFMAs of stage s from this one.
// NOT a parity mode (11-bit operands); reported by bench.py under other_precision_modes only.
__global__ __launch_bounds__(512, 2) void k_wconv1(WConvP p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int RA = 2, RB = 3;
    constexpr int SA = WROWS * WLD, SB = WROWS * WLD;
    constexpr int BOFF = 0, AOFF = RB * 2 * SB;
    __shared__ __attribute__((aligned(16))) _Float16 lds[RA * 2 * SA + RB * 2 * SB];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int grp_w = wave >> 2, ph = wm & 1;
    const int ntiles = p.ntm * p.ntn;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslot = (gridDim.x + 7 - xcd) >> 3;
    const int cq = ntiles >> 3, cr = ntiles & 7;
    const int chunk0 = xcd < cr ? xcd * (cq + 1) : cr * (cq + 1) + (xcd - cr) * cq, chunk_n = cq + (xcd < cr ? 1 : 0);
    constexpr int OOBL = 0x40000;
    const int dl_row = lane >> 2;
    const int dl_src = (lane & 3) ^ (2 * ((lane >> 4) & 1));
    const int lr = lane & 15, lc = lane >> 4;

    unsigned abase[2] = {0, 0}, boff[2] = {0, 0};
    auto ra0 = __builtin_amdgcn_make_buffer_rsrc((void*)p.V, 0, 0, 0x00020000);
    auto ra1 = ra0;
    const auto rb = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, (int)(2 * p.b_plane * 2), 0x00020000);
    int grpA = 0;
    unsigned vo0 = 0, vo1 = 0;
    bool prefetched = false;
    const int f_a = grp_w == 0 ? 1 : 0, f_b = grp_w == 0 ? 2 : 3;
    const int ncb = p.Cin / 32;
    auto tile_geom = [&](int m0_, int tn_) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            abase[j] = (unsigned)((16 * ((2 * wave + j) & 7) + dl_row) * 64 + 16 * dl_src);
            boff[j] = (unsigned)((long)tn_ * (p.K / 32) * 32768 + (16 * (2 * wave + j) + dl_row) * 64 + 16 * dl_src);
        }
        const int rec = ncb * 16384;
        const long tmi = m0_ >> 7;
        ra0 = __builtin_amdgcn_make_buffer_rsrc((void*)((const _Float16*)p.V + (((long)f_a * p.ntm + tmi) * ncb) * 8192), 0, rec, 0x00020000);
        ra1 = __builtin_amdgcn_make_buffer_rsrc((void*)((const _Float16*)p.V + (((long)f_b * p.ntm + tmi) * ncb) * 8192), 0, rec, 0x00020000);
        grpA = 0;
    };
    bool fiA = false;
    auto a_setup = [&]() {
        const unsigned toff = (unsigned)((grpA >> 1) * 16384);
        vo0 = abase[0] + toff;
        vo1 = abase[1] + toff;
        fiA = (grpA & 1) != 0;
        ++grpA;
    };
    auto dma_a = [&](int sa, int j) {                      // hi piece of this wave's row group j = 0, 1
        _Float16* st = lds + AOFF + sa * 2 * SA;
        const auto r = fiA ? ra1 : ra0;
        if (j == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, st + (16 * (2 * wave)) * WLD, 16, vo0, 0, 0, 0);
        if (j == 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, st + (16 * (2 * wave + 1)) * WLD, 16, vo1, 0, 0, 0);
    };
    auto dma_b = [&](int sb, int j) {                      // hi piece of this wave's weight row group j = 0, 1; advances to the next K step
        _Float16* st = lds + BOFF + sb * 2 * SB;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
            if (jj == j) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, st + (16 * (2 * wave + jj)) * WLD, 16, boff[jj], 0, 0, 0);
                boff[jj] += 32768;
            }
    };
    auto issue_prologue = [&]() {          // A(group 0) B(0) B(1)
        a_setup();
        dma_a(0, 0); dma_a(0, 1);
        dma_b(0, 0); dma_b(0, 1);
        dma_b(1, 0); dma_b(1, 1);
    };

    const char* ldsb = reinterpret_cast<const char*>(lds);
    const int off_l = lr * 64 + 16 * (lc ^ (2 * ((lr >> 2) & 1)));
    const int a_row0 = AOFF * 2 + wm * 64 * 64, b_row0 = BOFF * 2 + (grp_w * 128 + wn * 64) * 64;
    auto x_off = [&](int dx) -> int {
        const int r_ = lr + dx;
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
    unsigned xedge = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int x_ = (ph * 64 + 16 * i + lr) & (p.W - 1);
        xedge |= (x_ == 0 ? 1u : 0u) << i | (x_ == p.W - 1 ? 1u : 0u) << (4 + i);
        axm[i] = (!ZSEL && x_ == 0) ? zero_addr : a_row0 + x_off(-1);
        axp[i] = (!ZSEL && x_ == p.W - 1) ? zero_addr : a_row0 + x_off(1);
    }
    const int aw0 = b_row0 + off_l, aw2 = aw0 + 2 * 2 * SB * 2;

    for (int ti = slot; ti < chunk_n; ti += nslot) {
        const int lid = chunk0 + ti;
        const int tn = lid % p.ntn, tm = lid / p.ntn;
        const int m0 = tm * 128, n0 = tn * 128;
        if (!prefetched) {
            tile_geom(m0, tn);
            issue_prologue();
        }
        const bool after_epilogue = prefetched;
        prefetched = false;

        f32x4 acc0[4][4], acc1[4][4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) { acc0[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc1[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
        h16x8 xA[4], wA[4], xB[4], wB[4];            // two fragment sets, alternating between steps
        auto rd_x = [&](int sa, int i, int dx) -> h16x8 {
            const int a = dx < 0 ? axm[i] : (dx > 0 ? axp[i] : ax0);
            const h16x8 v = *(const h16x8*)(ldsb + a + (sa * 2 * SA * 2 + 16 * i * 64));
            if (ZSEL && dx != 0) {
                const h16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
                return ((xedge >> (dx < 0 ? i : 4 + i)) & 1u) ? z : v;
            }
            return v;
        };
        auto rd_w = [&](int sb, int jj) -> h16x8 {
            return *(const h16x8*)(ldsb + (sb == 2 ? aw2 : aw0) + ((sb == 2 ? 0 : sb * 2 * SB * 2) + 16 * jj * 64));
        };
#define WSB() __builtin_amdgcn_sched_barrier(0)
        if (after_epilogue) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WCONV_EPILOGUE_STORES) : "memory");      // (derived from the store loops: wconv.h)
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int i = 0; i < 4; ++i) xA[i] = rd_x(0, i, -1);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) wA[jj] = rd_w(0, jj);
        WSB();

        // One K step = (channel block, fi, dx): the DMA pieces of later stages, the barrier that publishes the next stage, that stage's
        // fragments into the OTHER register set, then this step's 16 MFMAs (they also cover the barrier of the next step)
        auto step = [&](auto fi_c, auto dxi_c, auto last_c, f32x4 (&A)[4][4], h16x8 (&XC)[4], h16x8 (&WC)[4], h16x8 (&XN)[4], h16x8 (&WN)[4])
                        __attribute__((always_inline)) {
            constexpr int FI = decltype(fi_c)::value, DXI = decltype(dxi_c)::value;
            constexpr bool LAST = decltype(last_c)::value;
            constexpr bool preA = !LAST && DXI == 0;
            constexpr bool pieceA = !LAST && DXI < 2;                           // one activation piece in each of the group's first two steps
            constexpr bool preB = !LAST || DXI == 0;
            constexpr bool more = !LAST || DXI < 2;
            constexpr int sa_n = FI ^ 1, sb_n = (DXI + 2) % 3;
            constexpr int sa_r = DXI == 2 ? (FI ^ 1) : FI, sb_r = (DXI + 1) % 3, dxn = DXI == 2 ? -1 : DXI;
            if (preA) { a_setup(); fiA = FI == 0; }
            WSB();
            if (preB) { dma_b(sb_n, 0); dma_b(sb_n, 1); }
            if (pieceA) dma_a(sa_n, DXI);
            WSB();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            {
                // in issue order: [B(s+1) x2, A x1] of the previous step, [B(s+2) x2, A x1] of this one.  B(s+1) must have landed; the
                // group's last step also needs both activation pieces
                constexpr int keep = LAST ? (preB ? 2 : 0) : (DXI == 0 ? 3 : (DXI == 1 ? 4 : 2));
                if (keep == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else if (keep == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
                else if (keep == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();
            WSB();
            if (more) {
#pragma unroll
                for (int i = 0; i < 4; ++i) XN[i] = rd_x(sa_r, i, dxn);
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) WN[jj] = rd_w(sb_r, jj);
            }
            WSB();
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    A[jj][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(WC[jj], XC[i], A[jj][i], 0, 0, 0);
                    WSB();
                }
        };
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        using I2 = std::integral_constant<int, 2>;
        using BF = std::integral_constant<bool, false>;
        using BT = std::integral_constant<bool, true>;
        // six steps per channel block: the fragment sets alternate A B A B A B, so every channel block starts on set A
#pragma nounroll
        for (int cbi = 0; cbi < ncb - 1; ++cbi) {
            step(I0{}, I0{}, BF{}, acc0, xA, wA, xB, wB); step(I0{}, I1{}, BF{}, acc0, xB, wB, xA, wA); step(I0{}, I2{}, BF{}, acc0, xA, wA, xB, wB);
            step(I1{}, I0{}, BF{}, acc1, xB, wB, xA, wA); step(I1{}, I1{}, BF{}, acc1, xA, wA, xB, wB); step(I1{}, I2{}, BF{}, acc1, xB, wB, xA, wA);
        }
        step(I0{}, I0{}, BF{}, acc0, xA, wA, xB, wB); step(I0{}, I1{}, BF{}, acc0, xB, wB, xA, wA); step(I0{}, I2{}, BF{}, acc0, xA, wA, xB, wB);
        step(I1{}, I0{}, BT{}, acc1, xB, wB, xA, wA); step(I1{}, I1{}, BT{}, acc1, xA, wA, xB, wB); step(I1{}, I2{}, BT{}, acc1, xB, wB, xA, wA);
        if (ti + nslot < chunk_n) {
            const int nl = chunk0 + ti + nslot;
            tile_geom((nl / p.ntn) * 128, nl % p.ntn);
            issue_prologue();
            prefetched = true;
        }
#undef WSB
        wconv_epilogue(p, acc0, acc1, reinterpret_cast<char*>(lds), wave, lane, m0, n0);
    }
#endif
}

// =================================== weight packer ===========================================
// frequency of (group, fi): group 0 (waves 0-3) holds f = 1, 2; group 1 holds f = 0, 3
__device__ __forceinline__ double wino_u(const float* __restrict__ w, int cout, int ci, int dx, int f, int Cin) {
    const float* g = w + ((size_t)cout * Cin + ci) * 9 + dx;             // OIHW: [ky][kx], kx = dx
    const double g0 = g[0], g1 = g[3], g2 = g[6];
    return f == 0 ? g0 : (f == 1 ? 0.5 * (g0 + g1 + g2) : (f == 2 ? 0.5 * (g0 - g1 + g2) : g2));
}
__global__ __launch_bounds__(256) void k_wino_row_scale(const float* __restrict__ w, float* __restrict__ inv_scale, int Cout, int Cin) {
    const int n = blockIdx.x;
    float m = 0.f;
    if (n < Cout)
        for (int k = threadIdx.x; k < Cin * 3; k += blockDim.x) {
            const int ci = k / 3, dx = k % 3;
#pragma unroll
            for (int f = 0; f < 4; ++f) m = fmaxf(m, fabsf((float)wino_u(w, n, ci, dx, f, Cin)));
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
// Packed layout: [column tile][K step = (channel block, fi, dx)][hi | lo][256 rows = group * 128 + channel][32 channels] - the
// 32 KB weight stage of a K step is one contiguous block in the order of its LDS image, so every 1-KB DMA piece reads eight whole
// 128-byte lines.
__global__ void k_pack_wino(const float* __restrict__ w, _Float16* __restrict__ packed, const float* __restrict__ inv_scale, int Cout,
                            int Cin, int ntn) {
    const int nk = 6 * (Cin / 32);
    const size_t total = (size_t)ntn * nk * 256 * 32;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i & 31), R = (int)((i >> 5) & 255);
        const size_t st = i >> 13;                                       // (column tile, step)
        const int tn = (int)(st / nk), s = (int)(st % nk);
        const int g = R >> 7, cout = tn * 128 + (R & 127);
        const int cb = s / 6, t = s % 6;
        const int fi = t / 3, dx = t % 3, ci = cb * 32 + c;
        const int f = g == 0 ? (fi == 0 ? 1 : 2) : (fi == 0 ? 0 : 3);
        float v = 0.f;
        if (cout < Cout) v = (float)wino_u(w, cout, ci, dx, f, Cin) * (1.0f / inv_scale[cout]);
        const _Float16 h = (_Float16)v;
        _Float16* o = packed + st * (2 * 256 * 32) + R * 32 + c;
        o[0] = h;
        o[256 * 32] = (_Float16)(v - (float)h);
    }
}

static inline int wino_ntn(int Cout) { return (Cout + 127) / 128; }
extern "C" size_t dua_f16x3_wino_weight_bytes(int Cout, int Cin) {
    return 4 * (size_t)wino_ntn(Cout) * 256 * 6 * Cin + 4 * (size_t)((Cout + 255) / 256 * 256);
}
extern "C" int dua_pack_weight_f16x3_wino(const float* w, void* packed, int Cout, int Cin, void* stream) {
    DUA_REQUIRE(w && packed && Cout > 0 && Cin >= 32 && Cin % 32 == 0, "dua_pack_weight_f16x3_wino: Cin must be a positive multiple of 32 (got %d)", Cin);
    const int ntn = wino_ntn(Cout), Nsc = (Cout + 255) / 256 * 256;
    const size_t plane = (size_t)ntn * 256 * 6 * Cin;
    _Float16* hi = (_Float16*)packed;
    float* inv_scale = (float*)(hi + 2 * plane);
    hipLaunchKernelGGL(k_wino_row_scale, dim3(Nsc), 256, 0, as_stream(stream), w, inv_scale, Cout, Cin);
    hipLaunchKernelGGL(k_pack_wino, stream_grid(plane, 256), 256, 0, as_stream(stream), w, hi, inv_scale, Cout, Cin, ntn);
    return dua_check_launch("dua_pack_weight_f16x3_wino");
}

// 0 = not supported, 1 = supported, 2 = supported without the column-sum output (maps smaller than 128 pixels: a wave's 64
// positions then span several images)
extern "C" int dua_conv3x3_f16x3_wino_supported(int N, int H, int W, int Cin, int Cout) {
    if (N <= 0 || H <= 0 || W <= 0 || Cin < 32 || Cin % 32 || Cout <= 0 || Cout % 4) return 0;
    if (H % 2 || W < 2 || W > 128 || (W & (W - 1))) return 0;
    const long Mv = (long)N * (H / 2) * W;
    if (Mv >= (1L << 30) || ((long)H * W) % 32) return 0;                           // (16-position groups of the writer stay inside one image)
    if ((size_t)wino_ntn(Cout) * 256 * 6 * Cin * 4 >= 0x7FFFFFF0ull) return 0;
    return ((long)H * W) % 128 == 0 ? 1 : 2;
}

static int conv3x3_wino_impl(const void* vplanes, const void* packed, const float* bias, const float* residual, int res_resample, float* y,
                             int N, int H, int W, int Cin, int Cout, float* colsum, float v_scale, void* stream, bool single) {
    DUA_REQUIRE(vplanes && packed && y, "dua_conv3x3_f16x3_wino: null pointer");
    DUA_REQUIRE(res_resample >= -1 && res_resample <= 1 && (res_resample <= 0 || W >= 2), "dua_conv3x3_f16x3_wino: res_resample must be -1, 0 or +1");
    DUA_REQUIRE(v_scale > 0.f, "dua_conv3x3_f16x3_wino: v_scale must be positive");
    const int sup = dua_conv3x3_f16x3_wino_supported(N, H, W, Cin, Cout);
    DUA_REQUIRE(sup, "dua_conv3x3_f16x3_wino: unsupported shape N=%d H=%d W=%d Cin=%d Cout=%d (H even, W a power of two <= 128, "
                "Cin %% 32 == 0, Cout %% 4 == 0)", N, H, W, Cin, Cout);
    DUA_REQUIRE(!colsum || sup == 1, "dua_conv3x3_f16x3_wino: the column-sum output needs H*W %% 128 == 0 (got %d x %d)", H, W);
    WConvP p{};
    p.V = (const _Float16*)vplanes; p.B = (const _Float16*)packed; p.bias = bias; p.res = residual; p.res_mode = residual ? res_resample : 0; p.C = y; p.colsum = colsum;
    p.Mv = N * (H / 2) * W; p.N = Cout; p.Cin = Cin; p.K = 6 * Cin; p.H = H; p.W = W;
    p.ntm = (p.Mv + 127) / 128; p.ntn = wino_ntn(Cout);
    p.b_plane = (long)p.ntn * 256 * p.K;
    p.Cscale = (const float*)((const _Float16*)packed + 2 * p.b_plane);
    p.out_scale = 1.0f / v_scale;
    const int ntiles = p.ntm * p.ntn, grid = ntiles < 256 ? ntiles : 256;
    DuaKTimer kt(stream);
    if (single) hipLaunchKernelGGL(k_wconv1, dim3(grid), 512, 0, as_stream(stream), p);
    else hipLaunchKernelGGL(k_wconv, dim3(grid), 512, 0, as_stream(stream), p);
    {   // algorithmic: 2 N H W 9 Cin Cout FLOPs; bytes: the V images once (8 B per input element, hi images: 4) + the packed weights + the output (+ residual)
        const double px = (double)N * H * W;
        kt.done(single ? "k_wconv1" : "k_wconv", (single ? 4.0 : 8.0) * px * Cin + (single ? 2.0 : 4.0) * 12.0 * Cin * Cout + 4.0 * px * Cout * (residual ? 2.0 : 1.0),
                2.0 * px * 9.0 * Cin * Cout);
    }
    return dua_check_launch("dua_conv3x3_f16x3_wino");
}
extern "C" int dua_conv3x3_f16x3_wino(const void* vplanes, const void* packed, const float* bias, const float* residual, int res_resample, float* y,
                                      int N, int H, int W, int Cin, int Cout, float* colsum, float v_scale, void* stream) {
    return conv3x3_wino_impl(vplanes, packed, bias, residual, res_resample, y, N, H, W, Cin, Cout, colsum, v_scale, stream, false);
}
// ... with one fp16 product per tap (mode "f16w"): reads only the hi images of the planes and of the packed weights
extern "C" int dua_conv3x3_f16w_wino(const void* vplanes, const void* packed, const float* bias, const float* residual, int res_resample, float* y,
                                     int N, int H, int W, int Cin, int Cout, float* colsum, float v_scale, void* stream) {
    return conv3x3_wino_impl(vplanes, packed, bias, residual, res_resample, y, N, H, W, Cin, Cout, colsum, v_scale, stream, true);
}

// =================================== GroupNorm apply -> V planes ===========================================
struct GnWinoP {
    const float *x1, *x2, *stats, *gamma, *beta, *film, *drop;
    _Float16* v;
    int C1, C2, N, H, W, G, act, resample;
    int ntm;                 // position tiles of 128 per frequency image
    long film_ld;
    float oscale;
    int JC;                  // row pairs per chunk
};

// One wave = 16 consecutive positions of a row pair x one 32-channel block: lane -> (position lane >> 2, channels 8 (lane & 3) ..+7),
// so a wave reads 16 whole 128-byte lines per row and writes, per frequency, the 1-KB piece [16 positions][32 channels] of the hi
// image and of the lo image of the conv's stage block (tile, channel block) - exactly the pieces its LDS-DMA fetches.  Maps narrower
// than 16 put 16 / W consecutive row pairs in a wave (then no row is shared between iterations and all four rows are re-read).
// STREAM (maps at least 16 wide, no pooling, no dropout mask - every conv of the sampling jobs but the down blocks' first): the two new
// rows of the NEXT row pair are requested before this pair's eight stores go out, so a wave always has 64 bytes per lane of loads in
// flight behind its stores instead of waiting for one row at a time.  Same arithmetic, same bits.
template <bool F32, bool STREAM, bool HI_ONLY = false>
__global__ __launch_bounds__(BLK) void k_gn_apply_wino(GnWinoP p) {
    const int C = p.C1 + p.C2, ncb = C / 32, cpg = C / p.G;
    const int Ho = p.resample < 0 ? p.H / 2 : (p.resample > 0 ? 2 * p.H : p.H), Wo = p.resample < 0 ? p.W / 2 : (p.resample > 0 ? 2 * p.W : p.W);
    const int P = Ho / 2;
    const int JS = Wo >= 16 ? 1 : 16 / Wo;                 // row pairs covered by one 16-position group
    const int npg = Wo >= 16 ? Wo / 16 : 1;                // position groups per row-pair row
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
    // per-channel scale and offset of the thread's 8 channels; every operand is requested before the first use (one latency, not eight)
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
            r[j] = v[j] * ca[j] + cb[j];
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
        if (F32) {         // exact-fp32 operand of conv_wino_f32.hip: stage block (f, tile, channel block) = [128 positions][32 floats]
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
        // written once, read by the next kernel: non-temporal stores (the small maps gain 25 - 35 % in this kernel, the job +0.6 %)
        __builtin_nontemporal_store(h, (h16x8*)o);
        if (!HI_ONLY) __builtin_nontemporal_store(l, (h16x8*)(o + 4096));     // (mode "f16w": the conv reads the hi images only)
    };
    const int j0 = jc * p.JC + jo, j1 = min(P, (jc + 1) * p.JC);
    float dA[8], dB[8], dC[8], dD[8];
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
                float t = v[q] * ca[q] + cb[q];
                if (p.act == 1) t = silu_f(t);
                r[q] = in ? t : 0.f;
            }
        };
        Raw rA = ld(2 * j0 - 1), rB = ld(2 * j0), rC = ld(2 * j0 + 1), rD = ld(2 * j0 + 2);
        fin(rA, 2 * j0 - 1, dA);
        fin(rB, 2 * j0, dB);
        auto pair = [&](int j) {
            fin(rC, 2 * j + 1, dC);
            fin(rD, 2 * j + 2, dD);
            if (j + 1 < j1) { rC = ld(2 * j + 3); rD = ld(2 * j + 4); }
            const size_t m = ((size_t)n * P + j) * Wo + xo;
            float v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = dA[q] - dC[q];
            put(0, m, v);
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = dB[q] + dC[q];
            put(1, m, v);
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = dC[q] - dB[q];
            put(2, m, v);
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = dB[q] - dD[q];
            put(3, m, v);
#pragma unroll
            for (int q = 0; q < 8; ++q) { dA[q] = dC[q]; dB[q] = dD[q]; }
        };
        // the first pair stands outside the loop so that both ways into the loop body carry the same queue (four loads, then eight
        // stores): the wait in front of `fin` then leaves the stores in flight
        pair(j0);
        for (int j = j0 + 1; j < j1; ++j) pair(j);
        return;
    }
    if (j0 < j1) { val(2 * j0 - 1, dA); val(2 * j0, dB); }
    for (int j = j0; j < j1; j += JS) {
        val(2 * j + 1, dC);
        val(2 * j + 2, dD);
        const size_t m = ((size_t)n * P + j) * Wo + xo;
        float v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = dA[q] - dC[q];
        put(0, m, v);
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = dB[q] + dC[q];
        put(1, m, v);
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = dC[q] - dB[q];
        put(2, m, v);
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = dB[q] - dD[q];
        put(3, m, v);
        if (JS == 1) {
#pragma unroll
            for (int q = 0; q < 8; ++q) { dA[q] = dC[q]; dB[q] = dD[q]; }
        } else if (j + JS < j1) {
            val(2 * (j + JS) - 1, dA);
            val(2 * (j + JS), dB);
        }
    }
}

// bytes of the V planes of a conv input [N][Ho][Wo][C] (Ho, Wo after the resampling): 4 frequencies x 2 planes x N*(Ho/2)*Wo*C halves
extern "C" size_t dua_f16x3_wino_act_bytes(int N, int Ho, int Wo, int C) {
    const size_t Mv = (size_t)N * (Ho / 2) * Wo;
    return (size_t)16 * ((Mv + 127) / 128 * 128) * C;                 // whole tiles of 128 positions
}

static int gn_apply_wino_impl(const float* x1, int C1, const float* x2, int C2, int N, int H, int W, int G,
                              const float* stats, const float* gamma, const float* beta, const float* film, long film_ld,
                              int act, int resample, const float* dropmask, float v_scale, void* vplanes, void* stream, bool f32, bool hi_only = false) {
    const int C = C1 + C2;
    DUA_REQUIRE(x1 && stats && gamma && beta && vplanes, "dua_groupnorm_apply_wino_f16x3: null pointer");
    DUA_REQUIRE(C % G == 0 && C1 % 8 == 0 && C2 % 8 == 0 && C % 32 == 0, "dua_groupnorm_apply_wino_f16x3: channel counts must be multiples of 8 (sum of 32)");
    DUA_REQUIRE((C2 == 0) == (x2 == nullptr), "dua_groupnorm_apply_wino_f16x3: x2/C2 mismatch");
    DUA_REQUIRE(resample >= -1 && resample <= 1 && (resample >= 0 || (H % 2 == 0 && W % 2 == 0)), "dua_groupnorm_apply_wino_f16x3: pooling needs even H, W");
    DUA_REQUIRE(!(dropmask && resample != 0), "dua_groupnorm_apply_wino_f16x3: dropout mask with resampling unsupported");
    DUA_REQUIRE(v_scale > 0.f && (__builtin_bit_cast(unsigned, v_scale) & 0x7FFFFFu) == 0, "dua_groupnorm_apply_wino_f16x3: v_scale must be a power of two");
    const int Ho = resample < 0 ? H / 2 : (resample > 0 ? 2 * H : H), Wo = resample < 0 ? W / 2 : (resample > 0 ? 2 * W : W);
    DUA_REQUIRE(Ho >= 2 && Ho % 2 == 0, "dua_groupnorm_apply_wino_f16x3: the conv input needs an even number of rows (got %d)", Ho);
    if ((size_t)N * Ho * Wo == 0) return DUA_OK;
    DUA_REQUIRE(C1 % 32 == 0 && Wo >= 2 && Wo <= 128 && (Wo & (Wo - 1)) == 0 && ((long)Ho * Wo) % 32 == 0,
                "dua_groupnorm_apply_wino_f16x3: C1 %% 32 == 0, the conv input width a power of two <= 128 and H*W %% 32 == 0 are required");
    GnWinoP p{x1, x2, stats, gamma, beta, film, dropmask, (_Float16*)vplanes, C1, C2, N, H, W, G, act, resample, 0, film_ld ? film_ld : 2L * C, v_scale, 0};
    DUA_REQUIRE(p.film_ld >= 2L * C, "dua_groupnorm_apply_wino_f16x3: film_ld too small");
    DUA_REQUIRE((((uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)film) & 15) == 0 && (!film || p.film_ld % 4 == 0),
                "dua_groupnorm_apply_wino_f16x3: gamma, beta and the FiLM rows must be 16-byte aligned");
    const int P = Ho / 2;
    p.ntm = (int)(((long)N * P * Wo + 127) / 128);
    const int nxb = ((Wo >= 16 ? Wo / 16 : 1) * (C / 32) + 3) / 4;
    // Row pairs per workgroup chunk: 8 (two halo rows re-read per 16 = +12 % reads) keeps enough workgroups in flight on the large
    // maps (128^2 x 256 channels: 1.26 ms against 1.60 ms with one chunk per column; scripts/bench_gn_apply.py f16x3w)
    int JC = P < 8 ? P : 8;
    while ((long)N * nxb * ((P + JC - 1) / JC) < 1024 && JC > 4) JC = (JC + 1) / 2;
    if (Wo < 16 && JC % (16 / Wo)) JC = (JC / (16 / Wo) + 1) * (16 / Wo);       // chunks hold whole 16-position groups
    p.JC = JC;
    const dim3 grid(nxb * ((P + JC - 1) / JC), N);
    const bool strm = resample >= 0 && !dropmask && Wo >= 16;
    DuaKTimer kt(stream);
    if (hi_only && strm) hipLaunchKernelGGL((k_gn_apply_wino<false, true, true>), grid, BLK, 0, as_stream(stream), p);
    else if (hi_only) hipLaunchKernelGGL((k_gn_apply_wino<false, false, true>), grid, BLK, 0, as_stream(stream), p);
    else if (f32 && strm) hipLaunchKernelGGL((k_gn_apply_wino<true, true>), grid, BLK, 0, as_stream(stream), p);
    else if (f32) hipLaunchKernelGGL((k_gn_apply_wino<true, false>), grid, BLK, 0, as_stream(stream), p);
    else if (strm) hipLaunchKernelGGL((k_gn_apply_wino<false, true>), grid, BLK, 0, as_stream(stream), p);
    else hipLaunchKernelGGL((k_gn_apply_wino<false, false>), grid, BLK, 0, as_stream(stream), p);
    // algorithmic bytes: the fp32 input once + the four V images of the conv's input positions (8 B per element; hi images only: 4 B)
    kt.done(hi_only ? "k_gn_apply_wino (hi images)" : (f32 ? "k_gn_apply_wino (fp32 V images)" : "k_gn_apply_wino"), 4.0 * N * H * W * C + (hi_only ? 4.0 : 8.0) * N * Ho * Wo * C);
    return dua_check_launch("dua_groupnorm_apply_wino");
}
extern "C" int dua_groupnorm_apply_wino_f16x3(const float* x1, int C1, const float* x2, int C2, int N, int H, int W, int G,
                                              const float* stats, const float* gamma, const float* beta, const float* film, long film_ld,
                                              int act, int resample, const float* dropmask, float v_scale, void* vplanes, void* stream) {
    return gn_apply_wino_impl(x1, C1, x2, C2, N, H, W, G, stats, gamma, beta, film, film_ld, act, resample, dropmask, v_scale, vplanes, stream, false);
}
// ... the hi images only (mode "f16w", operand of dua_conv3x3_f16w_wino): same layout, the lo halves of the stage blocks are not written
extern "C" int dua_groupnorm_apply_wino_f16w(const float* x1, int C1, const float* x2, int C2, int N, int H, int W, int G,
                                             const float* stats, const float* gamma, const float* beta, const float* film, long film_ld,
                                             int act, int resample, const float* dropmask, float v_scale, void* vplanes, void* stream) {
    return gn_apply_wino_impl(x1, C1, x2, C2, N, H, W, G, stats, gamma, beta, film, film_ld, act, resample, dropmask, v_scale, vplanes, stream, false, true);
}
// ... the four transformed images in fp32 (same stage-block layout with [128][32] floats per block, same byte count): the operand of
// dua_conv3x3_f32_wino (conv_wino_f32.hip), the exact-fp32 mode
extern "C" int dua_groupnorm_apply_wino_f32(const float* x1, int C1, const float* x2, int C2, int N, int H, int W, int G,
                                            const float* stats, const float* gamma, const float* beta, const float* film, long film_ld,
                                            int act, int resample, const float* dropmask, void* vplanes, void* stream) {
    return gn_apply_wino_impl(x1, C1, x2, C2, N, H, W, G, stats, gamma, beta, film, film_ld, act, resample, dropmask, 1.0f, vplanes, stream, true);
}

DUA_RANGE_TU(conv_wino)
