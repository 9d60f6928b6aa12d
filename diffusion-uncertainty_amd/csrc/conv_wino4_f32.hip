// Exact-fp32 3x3 convolution in the Winograd F(4,3)-along-y form (gfx950): the algebra of conv_wino4_f16x3.hip (four output rows from six
// transformed rows: 4.5 instead of 9 - F(2,3): 6 - matrix-core taps per output pixel) with fp32 operands on v_mfma_f32_16x16x4_f32, the
// ring discipline of conv_wino_f32.hip.  `precision = "fp32"` of the UNet (unet_openai.py:186,212): products and sums in fp32, the one
// mode whose arithmetic class is literally the reference's.  The fp32 matrix pipe runs at 1/16 of its fp16 rate, so the kernel is purely
// MFMA-bound and buys exactly the tap ratio: 6 / 4.5 against k_wconv32.
//
//   Tile = 128 positions (image, group of four rows, x) x 64 channels x 6 frequencies = 49 152 accumulators on eight waves: two frequency
//   groups ({1, 2, 0} and {3, 4, 5}) x two position halves x two channel halves; a wave holds 3 frequencies x 2 channel blocks x 4 position
//   blocks = 96 registers.  (128 channels would be 192 accumulator registers per lane at two waves per SIMD: no room for fragments.)
//   K step = (channel block, fi, dx): 64 MFMAs per wave (8 tiles x 8 k-slices of 4).
//   Stage blocks: activations [f][tile of 128 positions][Cin / 32][128][32] floats (dua_groupnorm_apply_wino4_f32: the V writer of the f16x3
//   form with fp32 stores - the same 6 B per input element); an activation stage = both groups' images of (channel block, fi): 256 rows.
//   Weights [column tile of 64][K step][group * 64 + channel][32] floats (dua_pack_weight_f32_wino4): a weight stage = 128 rows = 16 KB.
//   Rings: activations 3 x 32 KB (slot = fi: the stage of the next group is requested while this one runs), weights 3 x 16 KB (slot = dx).
//   LDS image and chunk swizzle as in k_wconv32 (conflict-free for the shifted fragment reads).
//   Inverse transform across partner waves (wave ^ 4: same positions and channels, the other three frequencies): the {1, 2, 0} wave sends
//   M1 - M2 and finishes output rows 0 and 2, the {3, 4, 5} wave sends M3 + M4 and finishes rows 1 and 3:
//     Y0 = (M0 + (M1 + M2)) + (M3 + M4)        Y1 = (M1 - M2) + 2 (M3 - M4)
//     Y2 = (M1 + M2) + 4 (M3 + M4)             Y3 = (M5 + (M1 - M2)) + 8 (M3 - M4)          (the sums of conv_wino4_f16x3.hip's epilogue)
#include "common.h"
#include "wconv.h"
#include <type_traits>

typedef float f32x8v __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int w4swz32(int r) { return (int)((0x54150410u >> (4 * ((r >> 1) & 7))) & 7u); }      // (the table of conv_wino_f32.hip)

constexpr int W432_STORES = 4 * 2 * 2;      // position blocks x the wave's two output rows x halves: always-issued stores of a wave per tile
static_assert(W432_STORES <= 63, "vmcnt is a 6-bit counter");

// ---------------------------------------------------------------------------------------------------------------------------------
// acc[fi][channel block][position block]: fi = 0, 1, 2 <-> frequencies {1, 2, 0} (group 0) / {3, 4, 5} (group 1); lane = channels
// 4 (lane >> 4) .. + 3 of position lane & 15.
__device__ __forceinline__ void w432_epilogue(const WConvP& p, f32x4 (&acc)[3][2][4], char* ldsw, int slot0_bytes, int wave, int lane, int m0, int n0) {
    const int grp = wave >> 2, ph = (wave >> 1) & 1, wn = wave & 1;
    const int lane_e = dua_lane_fresh();         // formed here: the kernel's own lane id, kept across the K loop, was spilled and came back behind a vmcnt(0)
    char* mine = ldsw + slot0_bytes + wave * 8192;
    char* theirs = ldsw + slot0_bytes + (wave ^ 4) * 8192;
    const bool second = grp == 1;                                          // rows 1, 3 (first group: rows 0, 2)
    const int pr_e = lane_e >> 3, pc_e = lane_e & 7;
    const int pos0 = m0 + ph * 64;
    const int c = n0 + wn * 32 + 4 * pc_e;
    const bool cok = c < p.N;
    // output pixel (row of C) of position (block i, half h, this lane's row of the half), output row 2 k + second; -1 beyond the tensor
    auto opix = [&](int i, int h, int k) -> long {
        const int pos = pos0 + 16 * i + 8 * h + pr_e;
        return pos < p.Mv ? (long)pos + 3L * (pos & ~(p.W - 1)) + (long)(2 * k + (second ? 1 : 0)) * p.W : -1L;
    };
    auto res_ld = [&](int k, int i, float4 (&dst)[2]) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const long row = opix(i, h, k);
            dst[h] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (!(p.res && cok && row >= 0)) continue;
            if (p.res_mode == 0) {
                dst[h] = *(const float4*)(p.res + row * p.N + c);
                continue;
            }
            const unsigned t = (unsigned)row >> (31 - __clz(p.W)), xo = (unsigned)row & (p.W - 1), n = t / (unsigned)p.H, yo = t - n * p.H;
            if (p.res_mode > 0) {
                const long src = ((long)n * (p.H >> 1) + (yo >> 1)) * (p.W >> 1) + (xo >> 1);
                dst[h] = *(const float4*)(p.res + src * p.N + c);
            } else {                                                  // same order of additions as k_resample2x (denoiser.hip)
                const long W2 = 2 * p.W, src = ((long)n * 2 * p.H + 2 * yo) * W2 + 2 * xo;
                const float4 a = *(const float4*)(p.res + src * p.N + c), b = *(const float4*)(p.res + (src + 1) * p.N + c);
                const float4 d = *(const float4*)(p.res + (src + W2) * p.N + c), e = *(const float4*)(p.res + (src + W2 + 1) * p.N + c);
                dst[h] = make_float4((a.x + b.x + d.x + e.x) * 0.25f, (a.y + b.y + d.y + e.y) * 0.25f, (a.z + b.z + d.z + e.z) * 0.25f,
                                     (a.w + b.w + d.w + e.w) * 0.25f);
            }
        }
    };
    // the epilogue's first global operands (bias, the residual rows of the wave's first output row) are requested BEFORE the exchange
    float4 bq = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.bias && cok) bq = *(const float4*)(p.bias + c);
    float4 rn_[2][4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i) res_ld(0, i, rn_[0][i]);
    // The output stores are buffer stores that ALWAYS issue (lanes outside the tensor get an out-of-range offset, which the buffer unit
    // drops): the kernel counts on W432_STORES vector-memory operations behind the next tile's prologue DMA (vmcnt retires in order)
    const long cpix0 = 4L * m0;
    const long crem = (4L * p.Mv - cpix0) * p.N * 4;
    const auto rc = __builtin_amdgcn_make_buffer_rsrc((void*)(p.C + cpix0 * p.N), 0, (int)(crem < 0x7FFFFFF0L ? crem : 0x7FFFFFF0L), 0x00020000);

    // ---- exchange: eight quads per wave (k = channel block * 4 + position block)
    f32x4 Y[2][2][4];                                                      // [the wave's output row 0 / 1][channel block][position block]
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f32x4 send = second ? acc[0][jj][i] + acc[1][jj][i] : acc[0][jj][i] - acc[1][jj][i];       // M3 + M4 | M1 - M2
            *(f32x4*)(mine + (jj * 4 + i) * 1024 + lane_e * 16) = send;
        }
    __syncthreads();
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f32x4 r = *(const f32x4*)(theirs + (jj * 4 + i) * 1024 + lane_e * 16);
            if (!second) {
                const f32x4 s12 = acc[0][jj][i] + acc[1][jj][i];
                Y[0][jj][i] = (acc[2][jj][i] + s12) + r;                   // Y0 = (M0 + s12) + s34
                Y[1][jj][i] = s12 + 4.0f * r;                              // Y2 = s12 + 4 s34
            } else {
                const f32x4 d34 = acc[0][jj][i] - acc[1][jj][i];
                Y[0][jj][i] = r + 2.0f * d34;                              // Y1 = d12 + 2 d34
                Y[1][jj][i] = (acc[2][jj][i] + r) + 8.0f * d34;            // Y3 = (M5 + d12) + 8 d34
            }
        }
    __syncthreads();                                                       // (the partner has read this wave's slot: it becomes the partner's patch)
    // ---- stores: the channel-block pair (32 channels) of one 16-position block through a 16 x 144-byte patch (in the partner's slot,
    // which only this wave still touches), so that residual reads and output stores move whole 128-byte rows
    const int lr_e = lane_e & 15, lc_e = lane_e >> 4;
    char* pw = theirs;
#pragma unroll
    for (int i = 0; i < 4; ++i) res_ld(1, i, rn_[1][i]);
    float4 cs = make_float4(0.f, 0.f, 0.f, 0.f), cq2 = cs;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float4 (&rq)[2] = rn_[k][i];
            *(f32x4*)(pw + lr_e * 144 + 16 * lc_e) = Y[k][0][i];
            *(f32x4*)(pw + lr_e * 144 + 64 + 16 * lc_e) = Y[k][1][i];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const f32x4 a = *(const f32x4*)(pw + (8 * h + pr_e) * 144 + 16 * pc_e);
                const long row = opix(i, h, k);
                float4 v;
                v.x = (a[0] + bq.x) + rq[h].x;
                v.y = (a[1] + bq.y) + rq[h].y;
                v.z = (a[2] + bq.z) + rq[h].z;
                v.w = (a[3] + bq.w) + rq[h].w;
                const bool ok = cok && row >= 0;
                const u32x4 vb = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
                __builtin_amdgcn_raw_buffer_store_b128(vb, rc, ok ? (unsigned)(((row - cpix0) * p.N + c) * 4) : 0xFFFFFFF0u, 0, 0);
                if (ok) {
                    cs.x += v.x; cs.y += v.y; cs.z += v.z; cs.w += v.w;
                    cq2.x += v.x * v.x; cq2.y += v.y * v.y; cq2.z += v.z * v.z; cq2.w += v.w * v.w;
                }
            }
        }
    }
    if (p.colsum) {
        // The partner waves' 64 positions x 4 output rows are 256 output pixels of ONE image (H*W % 256 == 0) = four 64-pixel slots of the
        // column-sum array: the first wave's sums (rows 0, 2) go to slot 0, the second's (rows 1, 3) to slot 1, zeros to slots 2 and 3.
        // The consumer (dua_groupnorm_stats_from_colsums) folds all slots of an image, so only the per-image totals matter.
        float t[8] = {cs.x, cs.y, cs.z, cs.w, cq2.x, cq2.y, cq2.z, cq2.w};
        const int lf_ = dua_lane_fresh();            // (not __shfl_xor: its hoisted lane id is spilled and reloaded behind a vmcnt(0), common.h)
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            t[u] += dua_xor_fresh(t[u], lf_, 8);
            t[u] += dua_xor_fresh(t[u], lf_, 16);
            t[u] += dua_xor_fresh(t[u], lf_, 32);
        }
        if (pr_e < 2 && cok && pos0 < p.Mv) {
            const float z = pr_e == 0 ? 1.0f : 0.0f;
            float* o = p.colsum + ((long)(pos0 >> 4) + (second ? 1 : 0) + 2 * pr_e) * 2 * p.N + c;
            *(float4*)o = make_float4(t[0] * z, t[1] * z, t[2] * z, t[3] * z);
            *(float4*)(o + p.N) = make_float4(t[4] * z, t[5] * z, t[6] * z, t[7] * z);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512, 2) void k_wconv432(WConvP p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int RA = 3, RB = 3;
    constexpr int STA = 256 * 32;                          // floats per activation stage (32 KB): rows 0 .. 127 group 0's image, 128 .. 255 group 1's
    constexpr int STB = 128 * 32;                          // floats per weight stage (16 KB): rows 0 .. 63 group 0's channels, 64 .. 127 group 1's
    constexpr int BOFF = 0, AOFF = RB * STB;               // weight ring first: a shifted row -1 of activation stage 0 stays inside LDS
    constexpr int NBP = 2;                                 // weight pieces (1 KB) per wave and stage
    __shared__ __attribute__((aligned(16))) float lds[RB * STB + RA * STA];      // 48 KB + 96 KB
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp_w = wave >> 2, ph = (wave >> 1) & 1, wn = wave & 1;
    const int ntiles = p.ntm * p.ntn;
    const int ncb = p.Cin / 32;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslot = (gridDim.x + 7 - xcd) >> 3;
    const int cq = ntiles >> 3, cr = ntiles & 7;
    const int chunk0 = xcd < cr ? xcd * (cq + 1) : cr * (cq + 1) + (xcd - cr) * cq, chunk_n = cq + (xcd < cr ? 1 : 0);

    constexpr int OOBL = 0x40000;                          // LDS byte address beyond the allocation: DS reads there return zeros
    static_assert(OOBL >= 160 * 1024 + 64 * 128 + 128, "the out-of-range LDS address must lie beyond the 160 KB of a gfx950 CU");
    // DMA: a 1-KB piece = 8 rows x 128 B, lane -> row lane >> 3, chunk position lane & 7; the swizzle goes on the SOURCE chunk.
    // Activations: wave w fills rows 32 w + 8 i .. + 7 (i = 0 .. 3) of a stage from row (r & 127) of ITS group's image (waves 0 - 3: group 0).
    // Weights: wave w fills rows 16 w + 8 i .. + 7 (i = 0, 1) of a stage from the same rows of the packed 128-row block.
    // (row >> 1) & 7 = (4 i + (lane >> 4)) & 7 in both: the swizzle term depends on the parity of i only.
    const int dl_row = lane >> 3;
    const int lr = lane & 15, lc = lane >> 4;
    unsigned asrc2[2], bsrc2[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int r = 32 * wave + 8 * e + dl_row;
        asrc2[e] = (unsigned)(((r & 127) - 8 * e) * 128 + 16 * ((lane & 7) ^ w4swz32(r)));
        const int rb_ = 16 * wave + 8 * e + dl_row;
        bsrc2[e] = (unsigned)(rb_ * 128 + 16 * ((lane & 7) ^ w4swz32(rb_)));
    }
    auto asrc = [&](int i) -> unsigned { return asrc2[i & 1] + 1024u * i; };
    const long fstride = (long)p.ntm * ncb * 16384;        // bytes between the images of consecutive frequencies
    const char* vtile = (const char*)p.V;
    auto ra = __builtin_amdgcn_make_buffer_rsrc((void*)p.V, 0, 0, 0x00020000);
    const auto rb = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, (int)(p.b_plane * 4), 0x00020000);
    unsigned boff = 0, toffReq = 0;
    int cbReq = 0;                                         // channel block of the activation stage being requested
    bool prefetched = false;
    auto tile_geom = [&](int m0_, int tn_) {
        vtile = (const char*)p.V + (long)(m0_ >> 7) * ncb * 16384;
        boff = (unsigned)((long)tn_ * (9 * ncb) * 16384);
        cbReq = 0;
    };
    // the activation stage of (channel block cbReq, fi): this wave's group reads frequency {1, 2, 0}[fi] / {3, 4, 5}[fi]
    auto a_setup = [&](int fi) {
        const int f = grp_w == 0 ? (fi == 2 ? 0 : fi + 1) : 3 + fi;
        ra = __builtin_amdgcn_make_buffer_rsrc((void*)(vtile + f * fstride), 0, ncb * 16384, 0x00020000);
        toffReq = (unsigned)(cbReq * 16384);
    };
    auto dma_a = [&](int sa, int i) {
        float* dst = lds + AOFF + sa * STA + (32 * wave + 8 * i) * 32;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, dst, 16, asrc(i), toffReq, 0, 0);
    };
    auto dma_b = [&](int sb, int i) {
        float* dst = lds + BOFF + sb * STB + (16 * wave + 8 * i) * 32;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, dst, 16, bsrc2[i & 1], boff, 0, 0);
        if (i == NBP - 1) boff += 16384;
    };
    auto issue_prologue = [&]() {          // A(channel block 0, fi = 0) B(0) B(1)
        a_setup(0);
#pragma unroll
        for (int i = 0; i < 4; ++i) dma_a(0, i);
#pragma unroll
        for (int i = 0; i < NBP; ++i) dma_b(0, i);
#pragma unroll
        for (int i = 0; i < NBP; ++i) dma_b(1, i);
    };

    // Loop-invariant fragment addresses (bytes): two 16-byte chunks (floats 8 lc .. + 3 and + 4 .. + 7) of row lr + dx; the image-edge
    // lanes of the shifted reads point beyond the LDS allocation (zeros), as in k_wconv32.
    const char* ldsb = reinterpret_cast<const char*>(lds);
    const int a_row0 = AOFF * 4 + (grp_w * 128 + ph * 64) * 128, b_row0 = BOFF * 4 + (grp_w * 64 + wn * 32) * 128;
    auto x_off = [&](int dx, int h) -> int {
        const int r_ = lr + dx;                                            // -1 .. 16; (row >> 1) & 7 depends on lr + dx only
        return r_ * 128 + 16 * ((2 * lc + h) ^ w4swz32(r_ + 64));
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
        const int m0 = tm * 128, n0 = tn * 64;
        if (!prefetched) {
            tile_geom(m0, tn);
            issue_prologue();
        }
        const bool after_epilogue = prefetched;       // this tile's first stages were requested before the previous tile's epilogue
        prefetched = false;

        f32x4 acc[3][2][4];
#pragma unroll
        for (int f = 0; f < 3; ++f)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[f][j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};

        f32x8v xf[4], wf[2];
        auto rd8 = [&](int a0, int a1, int off) -> f32x8v {
            const f32x4 u = *(const f32x4*)(ldsb + a0 + off), v = *(const f32x4*)(ldsb + a1 + off);
            return (f32x8v){u[0], u[1], u[2], u[3], v[0], v[1], v[2], v[3]};
        };
        auto rd_x = [&](int sa, int i, int dx) -> f32x8v {
            const int off = sa * STA * 4 + 16 * i * 128;
            if (dx == 0) return rd8(ax[1][0], ax[1][1], off);
            const bool edge = (xedge >> (dx < 0 ? i : 4 + i)) & 1u;
            const int d = dx + 1;
            return rd8(edge ? zero_addr - off : ax[d][0], edge ? zero_addr - off : ax[d][1], off);
        };
        auto rd_w = [&](int sb, int jj) -> f32x8v { return rd8(aw0[0], aw0[1], sb * STB * 4 + 16 * jj * 128); };
#define WSB() __builtin_amdgcn_sched_barrier(0)

        // the first stages have landed; behind them in the (in-order) queue are only the previous epilogue's operations, of which the
        // last W432_STORES - output stores - may stay in flight
        if (after_epilogue) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(W432_STORES) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int i = 0; i < 4; ++i) xf[i] = rd_x(0, i, -1);
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) wf[jj] = rd_w(0, jj);
        WSB();

        // One K step = (channel block, fi, dx): 64 MFMAs (8 tiles x 8 k-slices) into the accumulators of fi.  Ring slots are static: the
        // activation stage of (channel block, fi) lives in slot fi, the weight stage of step dx in slot dx.  LASTCB: the tile's last
        // channel block (its own copy of the nine steps: in its last group nothing beyond the tile is requested).
        auto step = [&](auto fi_c, auto dxi_c, auto last_c) __attribute__((always_inline)) {
            constexpr int FI = decltype(fi_c)::value, DXI = decltype(dxi_c)::value;
            constexpr bool LASTCB = decltype(last_c)::value;
            constexpr bool LG = LASTCB && FI == 2;                 // the tile's last group
            constexpr bool pieceA = !LG && DXI < 2;                // the next group's activation stage: two pieces in each of this group's first two steps
            constexpr bool preB = !LG || DXI == 0;                 // the weight stage two steps ahead
            constexpr bool more = !LG || DXI < 2;                  // a next step exists
            constexpr int FN = (FI + 1) % 3;
            constexpr int sb_n = (DXI + 2) % 3, sa_r = DXI == 2 ? FN : FI, sb_r = (DXI + 1) % 3, dxn = DXI == 2 ? -1 : DXI;
            f32x4 (&A)[2][4] = acc[FI];
            if (pieceA && DXI == 0) {
                if (FI == 2) ++cbReq;                              // (the next group opens the next channel block)
                a_setup(FN);
            }
            auto tile8 = [&](int jj, int i) {
#pragma unroll
                for (int m = 0; m < 8; ++m) A[jj][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[jj][m], xf[i][m], A[jj][i], 0, 0, 0);
            };
            WSB();
            // first half (channel block 0): the DMA pieces of this step between the tiles
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                tile8(0, i);
                WSB();
                if (i < NBP) { if (preB) dma_b(sb_n, i); }
                else { if (pieceA) dma_a(FN, i - NBP + 2 * DXI); }
                WSB();
            }
            // every read of stage s was issued during the previous step: publish stage s + 1.  In issue order: [B(s+1) x2, A x2] of the
            // previous step, [B(s+2) x2, A x2] of this one (A pieces in a group's first two steps only).  B(s+1) must have landed; the
            // activation stage the next group reads (requested in this group's steps 0 and 1) at the group's last step.
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            {
                constexpr int keep = (preB ? NBP : 0) + (pieceA ? (DXI == 1 ? 4 : 2) : 0);
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(keep) : "memory");
            }
            __builtin_amdgcn_s_barrier();
            WSB();
            // second half (channel block 1): each fragment is replaced by the next step's right after its last use
            if (more) wf[0] = rd_w(sb_r, 0);
            WSB();
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                tile8(1, i);
                WSB();
                if (more) { xf[i] = rd_x(sa_r, i, dxn); WSB(); }
            }
            if (more) wf[1] = rd_w(sb_r, 1);
            WSB();
        };
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        using I2 = std::integral_constant<int, 2>;
        using BF = std::integral_constant<bool, false>;
        using BT = std::integral_constant<bool, true>;
#define W432_CB(L) step(I0{}, I0{}, L{}); step(I0{}, I1{}, L{}); step(I0{}, I2{}, L{}); step(I1{}, I0{}, L{}); step(I1{}, I1{}, L{}); step(I1{}, I2{}, L{}); \
                   step(I2{}, I0{}, L{}); step(I2{}, I1{}, L{}); step(I2{}, I2{}, L{});
#pragma nounroll
        for (int cbi = 0; cbi < ncb - 1; ++cbi) { W432_CB(BF) }
        W432_CB(BT)
#undef W432_CB
        if (ti + nslot < chunk_n) {                  // the next tile's first stages under the epilogue (activation slot 0, weight slots 0, 1)
            const int nl = chunk0 + ti + nslot;
            tile_geom((nl / p.ntn) * 128, nl % p.ntn);
            issue_prologue();
            prefetched = true;
        }
#undef WSB
        // exchange slots / patches: activation slots 1 and 2 (64 KB; slot 0 and the weight slots 0, 1 hold the next tile's first stages)
        w432_epilogue(p, acc, reinterpret_cast<char*>(lds), (AOFF + STA) * 4, wave, lane, m0, n0);
    }
#endif
}

// =================================== weight packer ===========================================
// Packed layout: [column tile of 64][K step = (channel block, fi, dx)][group * 64 + channel][32 input channels] floats; the kernel
// transform G of F(4,3) (wino4_u of conv_wino4_f16x3.hip), evaluated in double, rounded once to fp32.
__global__ void k_pack_wino432(const float* __restrict__ w, float* __restrict__ packed, int Cout, int Cin, int ntn) {
    const int nk = 9 * (Cin / 32);
    const size_t total = (size_t)ntn * nk * 128 * 32;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i & 31), R = (int)((i >> 5) & 127);
        const size_t st = i >> 12;
        const int tn = (int)(st / nk), s = (int)(st % nk);
        const int g = R >> 6, cout = tn * 64 + (R & 63);
        const int cb = s / 9, t = s % 9;
        const int fi = t / 3, dx = t % 3, ci = cb * 32 + c;
        const int f = g == 0 ? (fi == 2 ? 0 : fi + 1) : 3 + fi;
        float v = 0.f;
        if (cout < Cout) {
            const float* gp = w + ((size_t)cout * Cin + ci) * 9 + dx;             // OIHW: [ky][kx], kx = dx
            const double g0 = gp[0], g1 = gp[3], g2 = gp[6];
            double u;
            switch (f) {
                case 0: u = g0 / 4.0; break;
                case 1: u = -(g0 + g1 + g2) / 6.0; break;
                case 2: u = -(g0 - g1 + g2) / 6.0; break;
                case 3: u = g0 / 24.0 + g1 / 12.0 + g2 / 6.0; break;
                case 4: u = g0 / 24.0 - g1 / 12.0 + g2 / 6.0; break;
                default: u = g2;
            }
            v = (float)u;
        }
        packed[i] = v;
    }
}
static inline int wino432_ntn(int Cout) { return (Cout + 63) / 64; }
extern "C" size_t dua_f32_wino4_weight_bytes(int Cout, int Cin) { return 4 * (size_t)wino432_ntn(Cout) * 128 * 9 * Cin; }
extern "C" int dua_pack_weight_f32_wino4(const float* w, void* packed, int Cout, int Cin, void* stream) {
    DUA_REQUIRE(w && packed && Cout > 0 && Cin >= 32 && Cin % 32 == 0, "dua_pack_weight_f32_wino4: Cin must be a positive multiple of 32 (got %d)", Cin);
    const int ntn = wino432_ntn(Cout);
    const size_t plane = (size_t)ntn * 128 * 9 * Cin;
    hipLaunchKernelGGL(k_pack_wino432, stream_grid(plane, 256), 256, 0, as_stream(stream), w, (float*)packed, Cout, Cin, ntn);
    return dua_check_launch("dua_pack_weight_f32_wino4");
}

extern "C" int dua_conv3x3_f32_wino4(const void* vplanes, const void* packed, const float* bias, const float* residual, int res_resample, float* y,
                                     int N, int H, int W, int Cin, int Cout, float* colsum, void* stream) {
    DUA_REQUIRE(vplanes && packed && y, "dua_conv3x3_f32_wino4: null pointer");
    DUA_REQUIRE(res_resample >= -1 && res_resample <= 1 && (res_resample <= 0 || W >= 2), "dua_conv3x3_f32_wino4: res_resample must be -1, 0 or +1");
    const int sup = dua_conv3x3_f16x3_wino4_supported(N, H, W, Cin, Cout);
    DUA_REQUIRE(sup, "dua_conv3x3_f32_wino4: unsupported shape N=%d H=%d W=%d Cin=%d Cout=%d (H %% 4 == 0, W a power of two <= 128, "
                "Cin %% 32 == 0, Cout %% 4 == 0)", N, H, W, Cin, Cout);
    DUA_REQUIRE(!colsum || sup == 1, "dua_conv3x3_f32_wino4: the column-sum output needs H*W %% 256 == 0 (got %d x %d)", H, W);
    WConvP p{};
    p.V = vplanes; p.B = packed; p.bias = bias; p.res = residual; p.res_mode = residual ? res_resample : 0; p.C = y; p.colsum = colsum;
    p.Mv = N * (H / 4) * W; p.N = Cout; p.Cin = Cin; p.K = 9 * Cin; p.H = H; p.W = W;
    p.ntm = (p.Mv + 127) / 128; p.ntn = wino432_ntn(Cout);
    p.b_plane = (long)p.ntn * 128 * p.K;
    DUA_REQUIRE(p.b_plane * 4 < 0x7FFFFFF0L, "dua_conv3x3_f32_wino4: packed weights too large");
    p.Cscale = nullptr;
    p.out_scale = 1.0f;
    const int ntiles = p.ntm * p.ntn, grid = ntiles < 256 ? ntiles : 256;
    hipLaunchKernelGGL(k_wconv432, dim3(grid), 512, 0, as_stream(stream), p);
    return dua_check_launch("dua_conv3x3_f32_wino4");
}
