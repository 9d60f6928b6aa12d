// Exact-fp32 3x3 convolution in the Winograd F(2,3)-along-y form (gfx950): the algebra, tile, ring discipline and epilogue of
// conv_wino_f16x3.hip with fp32 operands on v_mfma_f32_16x16x4_f32 (products and sums in fp32: the `precision = "fp32"` mode of
// the UNet).  6 instead of 9 taps per output pixel on a matrix pipe that runs fp32 at 1/16 of its fp16 rate: this kernel is
// purely MFMA-bound (128 MFMAs of 32 cycles per wave and K step against 4 - 6 LDS-DMA pieces and 32 fragment reads).
//
//   stage blocks: activations [f][tile of 128 positions][Cin / 32][128][32] floats (dua_groupnorm_apply_wino_f32), weights
//   [column tile][K step = (channel block, fi, dx)][group * 128 + channel][32] floats (dua_pack_weight_f32_wino); a 32-deep K slice
//   of a row is 128 B = eight 16-byte chunks, chunk c of row r at chunk position c ^ T[(r >> 1) & 7], T = (0, 1, 4, 0, 5, 1, 4, 5):
//   conflict-free for the two ds_read_b128 of a fragment (lane -> row (l & 15) + dx, floats 8 (l >> 4) .. + 7) at dx = -1, 0, +1
//   (exhaustive search over the 8-entry tables).  The K order inside a 32-deep stage is free as long as both operands agree: MFMA m
//   of a stage multiplies float m of every lane's 8-float fragments, i.e. k = 8 (l >> 4) + m.
#include "common.h"
#include "wconv.h"
#include <type_traits>

typedef float f32x8v __attribute__((ext_vector_type(8)));

__device__ __forceinline__ int wswz32(int r) { return (int)((0x54150410u >> (4 * ((r >> 1) & 7))) & 7u); }

__global__ __launch_bounds__(512, 2) void k_wconv32(WConvP p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int RA = 2, RB = 3;
    constexpr int ST = 256 * 32;                           // floats per stage (32 KB)
    constexpr int BOFF = 0, AOFF = RB * ST;                // weight ring first (bytes: 0 / 96 KB, as in the f16x3 kernel)
    constexpr int NBP = 4;
    __shared__ __attribute__((aligned(16))) float lds[(RA + RB) * ST];
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
    static_assert(OOBL >= 160 * 1024 + 64 * 128 + 128, "the out-of-range LDS address must lie beyond the 160 KB of a gfx950 CU");
    // DMA: a 1-KB piece = 8 rows x 128 B, lane -> row lane >> 3, chunk position lane & 7; the swizzle goes on the SOURCE chunk.
    // Wave w fills rows 32 w + 8 i .. + 7 of a stage with its piece i: (row >> 1) & 7 = (4 i + (lane >> 4)) & 7.
    const int dl_row = lane >> 3;
    const int lr = lane & 15, lc = lane >> 4;
    // byte offset of this lane's 16 bytes inside a 128-row stage block for piece i: rows (32 wave + 8 i) & 127 .. + 7; the swizzle
    // term depends on the parity of i only, so two registers serve the four pieces (+ 1024 i); the weight block has 256 rows:
    // waves 4 - 7 fill its second half (+ 16 KB, wave-uniform)
    unsigned src2[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int r = 32 * wave + 8 * e + dl_row;
        src2[e] = (unsigned)(((r & 127) - 8 * e) * 128 + 16 * ((lane & 7) ^ wswz32(r)));
    }
    const unsigned bhalf = wave >= 4 ? 16384u : 0u;
    auto asrc = [&](int i) -> unsigned { return src2[i & 1] + 1024u * i; };
    auto ra0 = __builtin_amdgcn_make_buffer_rsrc((void*)p.V, 0, 0, 0x00020000);
    auto ra1 = ra0;
    const auto rb = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, (int)(p.b_plane * 4), 0x00020000);
    int grpA = 0;
    unsigned boff = 0, toffA = 0;
    bool prefetched = false;
    const int f_a = grp_w == 0 ? 1 : 0, f_b = grp_w == 0 ? 2 : 3;
    const int ncb = p.Cin / 32;
    auto tile_geom = [&](int m0_, int tn_) {
        const int rec = ncb * 16384;                                       // one tile of one frequency: ncb stage blocks of 16 KB
        const long tmi = m0_ >> 7;
        ra0 = __builtin_amdgcn_make_buffer_rsrc((void*)((const float*)p.V + (((long)f_a * p.ntm + tmi) * ncb) * 4096), 0, rec, 0x00020000);
        ra1 = __builtin_amdgcn_make_buffer_rsrc((void*)((const float*)p.V + (((long)f_b * p.ntm + tmi) * ncb) * 4096), 0, rec, 0x00020000);
        boff = (unsigned)((long)tn_ * nk * 32768);
        grpA = 0;
    };
    bool fiA = false;
    auto a_setup = [&]() {
        toffA = (unsigned)((grpA >> 1) * 16384);
        fiA = (grpA & 1) != 0;
        ++grpA;
    };
    auto dma_a = [&](int sa, int i) {
        float* dst = lds + AOFF + sa * ST + (32 * wave + 8 * i) * 32;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(fiA ? ra1 : ra0, dst, 16, asrc(i) + toffA, 0, 0, 0);
    };
    auto dma_b = [&](int sb, int i) {
        float* dst = lds + BOFF + sb * ST + (32 * wave + 8 * i) * 32;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, dst, 16, asrc(i) + bhalf + boff, 0, 0, 0);
        if (i == 3) boff += 32768;
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
    auto issue_prologue = [&]() {
        issue_a(0);
        issue_b(0);
        issue_b(1);
    };

    // Loop-invariant fragment addresses (bytes): two 16-byte chunks (floats 8 lc .. + 3 and + 4 .. + 7) of row lr + dx; the image-edge
    // lanes of the shifted reads point beyond the LDS allocation (zeros), as in the f16x3 kernel.
    const char* ldsb = reinterpret_cast<const char*>(lds);
    const int a_row0 = AOFF * 4 + wm * 64 * 128, b_row0 = BOFF * 4 + (grp_w * 128 + wn * 64) * 128;
    auto x_off = [&](int dx, int h) -> int {
        const int r_ = lr + dx;                                            // -1 .. 16; (row >> 1) & 7 depends on lr + dx only
        return r_ * 128 + 16 * ((2 * lc + h) ^ wswz32(r_ + 64));
    };
    const int zero_addr = OOBL + (lane & 1) * 16;
    int ax[3][2], aw0[2];                                  // [dx + 1][chunk half]
    unsigned xedge = 0;                                    // bit i: position block i at x == 0; bit 4 + i: at x == W - 1
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        aw0[h] = b_row0 + x_off(0, h);
#pragma unroll
        for (int d = 0; d < 3; ++d) ax[d][h] = a_row0 + x_off(d - 1, h);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int x_ = (ph * 64 + 16 * i + lr) & (p.W - 1);
        xedge |= (x_ == 0 ? 1u : 0u) << i | (x_ == p.W - 1 ? 1u : 0u) << (4 + i);
    }

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

        f32x4 acc0[4][4], acc1[4][4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) { acc0[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc1[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f}; }

        f32x8v xf[4], wf[4];
        auto rd8 = [&](int a0, int a1, int off) -> f32x8v {
            const f32x4 u = *(const f32x4*)(ldsb + a0 + off), v = *(const f32x4*)(ldsb + a1 + off);
            return (f32x8v){u[0], u[1], u[2], u[3], v[0], v[1], v[2], v[3]};
        };
        auto rd_x = [&](int sa, int i, int dx) -> f32x8v {
            const int off = sa * ST * 4 + 16 * i * 128;
            if (dx == 0) return rd8(ax[1][0], ax[1][1], off);
            // image-edge lanes of a shifted read: the address beyond the allocation (zeros); one select per chunk - this kernel has
            // VALU slots to spare (128 MFMAs of 32 cycles per step), so the twelve addresses need not live in registers per block
            const bool edge = (xedge >> (dx < 0 ? i : 4 + i)) & 1u;
            const int d = dx + 1;
#if defined(DUA_RS_ZERO_SELECT) && DUA_RS_ZERO_SELECT
            const f32x8v v = rd8(ax[d][0], ax[d][1], off), z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};      // data-select form (test build)
            return edge ? z : v;
#else
            return rd8(edge ? zero_addr - off : ax[d][0], edge ? zero_addr - off : ax[d][1], off);
#endif
        };
        auto rd_w = [&](int sb, int jj) -> f32x8v {
            const int off = (sb == 2 ? 0 : sb * ST * 4) + 16 * jj * 128;
            return sb == 2 ? rd8(aw0[0] + 2 * ST * 4, aw0[1] + 2 * ST * 4, off) : rd8(aw0[0], aw0[1], off);
        };
#define WSB() __builtin_amdgcn_sched_barrier(0)

        // the first stages have landed; behind them in the (in-order) queue are only the previous epilogue's operations, of which the
        // last WCONV_EPILOGUE_STORES - output stores - may stay in flight
        if (after_epilogue) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WCONV_EPILOGUE_STORES) : "memory");      // (derived from the store loops: wconv.h)
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int i = 0; i < 4; ++i) xf[i] = rd_x(0, i, -1);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) wf[jj] = rd_w(0, jj);
        WSB();

        // One K step = (channel block, fi, dx): 128 MFMAs (16 tiles x 8 k-slices) into the accumulators of fi; slots static as in
        // the f16x3 kernel (activation stage of (channel block, fi) in slot fi, weight stage of step dxi in slot dxi).
        auto step = [&](auto fi_c, auto dxi_c, auto last_c, f32x4 (&A)[4][4]) __attribute__((always_inline)) {
            constexpr int FI = decltype(fi_c)::value, DXI = decltype(dxi_c)::value;
            constexpr bool LAST = decltype(last_c)::value;
            constexpr bool preA = !LAST && DXI == 0;
            constexpr bool pieceA = !LAST && DXI < 2;
            constexpr bool preB = !LAST || DXI == 0;
            constexpr bool more = !LAST || DXI < 2;
            constexpr int sa_n = FI ^ 1, sb_n = (DXI + 2) % 3;
            constexpr int sa_r = DXI == 2 ? (FI ^ 1) : FI, sb_r = (DXI + 1) % 3, dxn = DXI == 2 ? -1 : DXI;
            if (preA) { a_setup(); fiA = FI == 0; }
            auto tile8 = [&](int jj, int i) {
#pragma unroll
                for (int m = 0; m < 8; ++m) A[jj][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[jj][m], xf[i][m], A[jj][i], 0, 0, 0);
            };
            WSB();
            // first half (channel blocks 0, 1): the DMA pieces of this step between the tiles
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    tile8(jj, i);
                    WSB();
                    const int q = jj * 4 + i;
                    if (q < NBP) { if (preB) dma_b(sb_n, q); }
                    else if (q < NBP + 2) { if (pieceA) dma_a(sa_n, q - NBP + 2 * DXI); }
                    WSB();
                }
            // every read of stage s was issued during the previous step: publish stage s + 1
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            {
                constexpr int keep = LAST ? (preB ? NBP : 0) : (DXI == 0 ? NBP + 2 : (DXI == 1 ? NBP + 4 : NBP));
                if (keep == NBP + 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                else if (keep == NBP + 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                else if (keep == NBP) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();
            WSB();
            // second half: each fragment is replaced by the next step's right after its last use
            if (more) { wf[0] = rd_w(sb_r, 0); wf[1] = rd_w(sb_r, 1); }
            WSB();
#pragma unroll
            for (int i = 0; i < 4; ++i) { tile8(2, i); WSB(); }
            if (more) wf[2] = rd_w(sb_r, 2);
            WSB();
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                tile8(3, i);
                WSB();
                if (more) { xf[i] = rd_x(sa_r, i, dxn); WSB(); }
            }
            if (more) wf[3] = rd_w(sb_r, 3);
            WSB();
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
        if (ti + nslot < chunk_n) {                  // the next tile's first stages under the epilogue (activation slot 0, weight slots 0, 1)
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
__global__ void k_pack_wino32(const float* __restrict__ w, float* __restrict__ packed, int Cout, int Cin, int ntn) {
    const int nk = 6 * (Cin / 32);
    const size_t total = (size_t)ntn * nk * 256 * 32;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i & 31), R = (int)((i >> 5) & 255);
        const size_t st = i >> 13;
        const int tn = (int)(st / nk), s = (int)(st % nk);
        const int g = R >> 7, cout = tn * 128 + (R & 127);
        const int cb = s / 6, t = s % 6;
        const int fi = t / 3, dx = t % 3, ci = cb * 32 + c;
        const int f = g == 0 ? (fi == 0 ? 1 : 2) : (fi == 0 ? 0 : 3);
        float v = 0.f;
        if (cout < Cout) {
            const float* gp = w + ((size_t)cout * Cin + ci) * 9 + dx;
            const double g0 = gp[0], g1 = gp[3], g2 = gp[6];
            v = (float)(f == 0 ? g0 : (f == 1 ? 0.5 * (g0 + g1 + g2) : (f == 2 ? 0.5 * (g0 - g1 + g2) : g2)));
        }
        packed[i] = v;
    }
}
__global__ void k_fill_ones(float* __restrict__ x, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] = 1.0f;
}
static inline int wino_ntn(int Cout) { return (Cout + 127) / 128; }
extern "C" size_t dua_f32_wino_weight_bytes(int Cout, int Cin) {
    return 4 * (size_t)wino_ntn(Cout) * 256 * 6 * Cin + 4 * (size_t)((Cout + 255) / 256 * 256);
}
extern "C" int dua_pack_weight_f32_wino(const float* w, void* packed, int Cout, int Cin, void* stream) {
    DUA_REQUIRE(w && packed && Cout > 0 && Cin >= 32 && Cin % 32 == 0, "dua_pack_weight_f32_wino: Cin must be a positive multiple of 32 (got %d)", Cin);
    const int ntn = wino_ntn(Cout), Nsc = (Cout + 255) / 256 * 256;
    const size_t plane = (size_t)ntn * 256 * 6 * Cin;
    float* pk = (float*)packed;
    hipLaunchKernelGGL(k_pack_wino32, stream_grid(plane, 256), 256, 0, as_stream(stream), w, pk, Cout, Cin, ntn);
    hipLaunchKernelGGL(k_fill_ones, dim3((Nsc + 255) / 256), 256, 0, as_stream(stream), pk + plane, Nsc);     // (the shared epilogue's per-channel scale)
    return dua_check_launch("dua_pack_weight_f32_wino");
}

extern "C" int dua_conv3x3_f32_wino(const void* vplanes, const void* packed, const float* bias, const float* residual, int res_resample, float* y,
                                    int N, int H, int W, int Cin, int Cout, float* colsum, void* stream) {
    DUA_REQUIRE(vplanes && packed && y, "dua_conv3x3_f32_wino: null pointer");
    DUA_REQUIRE(res_resample >= -1 && res_resample <= 1 && (res_resample <= 0 || W >= 2), "dua_conv3x3_f32_wino: res_resample must be -1, 0 or +1");
    const int sup = dua_conv3x3_f16x3_wino_supported(N, H, W, Cin, Cout);
    DUA_REQUIRE(sup, "dua_conv3x3_f32_wino: unsupported shape N=%d H=%d W=%d Cin=%d Cout=%d (H even, W a power of two <= 128, "
                "H*W %% 32 == 0, Cin %% 32 == 0, Cout %% 4 == 0)", N, H, W, Cin, Cout);
    DUA_REQUIRE(!colsum || sup == 1, "dua_conv3x3_f32_wino: the column-sum output needs H*W %% 128 == 0 (got %d x %d)", H, W);
    WConvP p{};
    p.V = vplanes; p.B = packed; p.bias = bias; p.res = residual; p.res_mode = residual ? res_resample : 0; p.C = y; p.colsum = colsum;
    p.Mv = N * (H / 2) * W; p.N = Cout; p.Cin = Cin; p.K = 6 * Cin; p.H = H; p.W = W;
    p.ntm = (p.Mv + 127) / 128; p.ntn = wino_ntn(Cout);
    p.b_plane = (long)p.ntn * 256 * p.K;
    p.Cscale = (const float*)packed + p.b_plane;
    p.out_scale = 1.0f;
    const int ntiles = p.ntm * p.ntn, grid = ntiles < 256 ? ntiles : 256;
    hipLaunchKernelGGL(k_wconv32, dim3(grid), 512, 0, as_stream(stream), p);
    return dua_check_launch("dua_conv3x3_f32_wino");
}
