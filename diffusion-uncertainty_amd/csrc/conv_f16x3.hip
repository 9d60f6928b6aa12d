// f16x3 implicit-GEMM 3x3 convolution / one-tap GEMM on the 16x16x32 fp16 matrix-core shape (gfx950).
//
// Operands are the f16x3 planes of include/dua_hip.h: activations [rows][Cin] as fp16 hi | fp16 lo of x * act_scale (written
// by dua_groupnorm_apply_split_f16x3 / dua_attention_f16x3), weights hi | lo of w * 2^s with per-row 2^-s, stage-contiguous
// [row tile of 128][K step of 32][hi | lo][128][32] (dua_pack_weight_f16x3; K ordered channel-block major, tap minor).  Product = lo*hi + hi*lo + hi*hi, fp32 accumulate, the
// epilogue multiplies by 2^-s / act_scale, adds bias / residual and leaves the next GroupNorm's column sums.
//
// Why a kernel of its own (k_xconv_dma<NT, 3> in mfma_bf16x3.hip is the same arithmetic on the 32x32x16 shape):
//   * v_mfma_f32_16x16x32_f16: the chip is power-limited on this loop and holds a higher clock on the 16x16 shape
//     (MI355X_MICROARCH.md, DVFS give-back item 7); one instruction consumes the whole 32-deep K stage of a 16-row block, so a
//     fragment is read once per stage;
//   * the WEIGHTS are the MFMA's A operand (rows) and the pixels its B operand (columns): an accumulator tile then holds, per
//     lane, FOUR CONSECUTIVE OUTPUT CHANNELS OF ONE PIXEL - a float4 of the NHWC output row.  The epilogue stores straight from
//     the accumulators (16 B per lane, 64-B runs per pixel) with no LDS transposition: LDS is free as soon as the K loop's last
//     barrier has passed, so the first TWO stages of the workgroup's next tile are requested before the epilogue and their
//     latency runs under it;
//   * row-shift form (RS, image width divides 256 so a 256-row tile is whole image rows): ONE activation stage per (channel
//     block, dy) serves the three dx taps; the taps read it at rows r-1, r, r+1 and the lanes at the image's left / right edge
//     read zeros through an LDS address beyond the allocation.  L2 -> LDS activation traffic drops to a third.
// LDS image: rows of 64 B (32 halves), 16-B chunk c of row r stored at chunk position c ^ (2 * ((r >> 2) & 1)) (round 3; conflict-free
// for rows shifted by -1, 0 and +1 - the earlier c ^ ((-(r >> 2)) & 3) was for unshifted rows only: DESIGN.md section 6.8): the four
// 16-lane groups of a ds_read_b128 (lane -> row l & 15, chunk l >> 4) land on 16 distinct 16-B slots.  The LDS-DMA writes
// lane-linear 1-KB pieces (16 rows), so the swizzle is applied to each lane's SOURCE chunk.
#include "common.h"
#include "dconv.h"

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define HROWS 256                    // pixel rows of a block tile
#define HLD 32                       // halves per LDS row

template <int NT, bool RS>
__global__ __launch_bounds__(512, 2) void k_hconv(DConvP p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int DBN = 64 * NT;                     // channels of a block tile; wave tile = 64 pixels x 32*NT channels
    constexpr int NB = 2 * NT;                       // 16-channel blocks of a wave
    constexpr int HB = NT;                           // ... of one half of the wave's channels (fragments are held per half)
    constexpr int RA = RS ? 2 : 3;                   // activation / weight ring depths (stages)
    constexpr int RB = RS ? 3 : (NT == 4 ? 2 : 3);
    constexpr int SA = HROWS * HLD, SB = DBN * HLD;  // halves per plane of a stage
    constexpr int BOFF = RS ? 0 : RA * 2 * SA;       // RS: weight ring first (a shifted row -1 of activation stage 0 stays inside LDS)
    constexpr int AOFF = RS ? RB * 2 * SB : 0;
    constexpr int NBP = NT;                          // weight DMA pieces per wave and stage (16-row groups x 2 planes)
    __shared__ __attribute__((aligned(16))) __bf16 lds[RA * 2 * SA + RB * 2 * SB];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int ntiles = p.ntm * p.ntn;
    const int nk = p.K / 32;
    // persistent workgroups, tiles chunked per XCD (blockIdx % 8) so that co-running workgroups of an XCD share weight panels / halos in its L2
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslot = (gridDim.x + 7 - xcd) >> 3;
    const int cq = ntiles >> 3, cr = ntiles & 7;
    const int chunk0 = xcd < cr ? xcd * (cq + 1) : cr * (cq + 1) + (xcd - cr) * cq, chunk_n = cq + (xcd < cr ? 1 : 0);

    const unsigned OOB = 0xFFFFFFF0u;                // buffer offset beyond every record: the DMA writes zeros
    // LDS byte address beyond the allocation: DS reads there return zeros (relied upon by the row-shift form for the image-edge
    // lanes; -DDUA_RS_ZERO_SELECT=1 builds the data-select form instead, compared bit for bit by tests/test_gpu_ops.py)
    constexpr int OOBL = 0x40000;
    static_assert(OOBL >= 160 * 1024 + 64 * 64 + 64, "the out-of-range LDS address must lie beyond the 160 KB of a gfx950 CU for every fragment offset");
    const int dl_row = lane >> 2;                                        // DMA: row of the 16-row piece this lane fills
    const int dl_src = (lane & 3) ^ (2 * ((lane >> 4) & 1));             // DMA: source chunk for this lane's LDS slot
    const int lr = lane & 15, lc = lane >> 4;                            // fragment: row of the 16-row block, K chunk

    // ---- DMA state (lives across tiles: the next tile's first stages are requested before this tile's epilogue)
    unsigned abase[2] = {0, 0}, abits[2] = {0, 0}, rsy = 0, boff[NT / 2];
    auto ra = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, 0, 0x00020000);
    auto ral = ra;
    const auto rb = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, (int)(2 * p.b_plane * 2), 0x00020000);
    const int b_lo = 8192;                                               // byte distance hi -> lo image inside a weight stage block
    int tap = 0, cb = 0;
    unsigned vo0 = 0, vo1 = 0;
    bool prefetched = false;

    auto tile_geom = [&](int m0_, int n0_) {
        const int mbase = p.ntaps == 9 ? max(0, m0_ - (p.W + 1)) : m0_;
        rsy = 0;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int m = min(m0_ + 16 * (2 * wave + j) + dl_row, p.M - 1);
            abase[j] = (unsigned)(((long)(m - mbase) * p.Cin + 8 * dl_src) * 2);
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
        const long arem = ((long)p.M - mbase) * p.Cin * 2;               // bytes from the rebased origin to the plane's end
        const int arec = (int)(arem < 0x7FFFFFF0L ? arem : 0x7FFFFFF0L);
        ra = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A + (long)mbase * p.Cin), 0, arec, 0x00020000);
        ral = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A + p.a_plane + (long)mbase * p.Cin), 0, arec, 0x00020000);
        tap = 0;
        cb = 0;
#pragma unroll
        for (int j = 0; j < NT / 2; ++j)
        {   // stage-contiguous packing [row tile of 128][K step][hi | lo][128][32] (dua_pack_weight_f16x3)
            const int r_ = n0_ + 16 * ((NT / 2) * wave + j) + dl_row;
            boff[j] = (unsigned)(((long)(r_ >> 7) * (p.K / 32)) * 16384 + (r_ & 127) * 64 + 16 * dl_src);
        }
    };
    // offsets of the next activation stage: per-tap form = the next (channel block, tap); RS = the next (channel block, dy)
    auto a_setup = [&]() {
        if (RS) {
            const int toff = (((tap - 1) * p.W) * p.Cin + cb * 32) * 2;
            const unsigned out_ = tap == 0 ? rsy : (tap == 2 ? rsy >> 2 : 0u);
            vo0 = (out_ & 1u) ? OOB : abase[0] + (unsigned)toff;
            vo1 = (out_ & 2u) ? OOB : abase[1] + (unsigned)toff;
            if (++tap == 3) { tap = 0; ++cb; }
            return;
        }
        const int dy = p.ntaps == 9 ? tap / 3 - 1 : 0, dx = p.ntaps == 9 ? tap - (tap / 3) * 3 - 1 : 0;
        const int toff = ((dy * p.W + dx) * p.Cin + cb * 32) * 2;
        vo0 = ((abits[0] >> tap) & 1u) ? abase[0] + (unsigned)toff : OOB;
        vo1 = ((abits[1] >> tap) & 1u) ? abase[1] + (unsigned)toff : OOB;
        if (++tap == p.ntaps) { tap = 0; ++cb; }
    };
    auto dma_a = [&](int sa, int i) {                // piece i = 0..3 of the activation stage a_setup() prepared
        __bf16* st = lds + AOFF + sa * 2 * SA;
        if (i == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, st + (16 * (2 * wave)) * HLD, 16, vo0, 0, 0, 0);
        if (i == 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(ral, st + (16 * (2 * wave)) * HLD + SA, 16, vo0, 0, 0, 0);
        if (i == 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, st + (16 * (2 * wave + 1)) * HLD, 16, vo1, 0, 0, 0);
        if (i == 3) __builtin_amdgcn_raw_ptr_buffer_load_lds(ral, st + (16 * (2 * wave + 1)) * HLD + SA, 16, vo1, 0, 0, 0);
    };
    auto dma_b = [&](int sb, int i) {                // piece i = 0..NBP-1 of the next weight stage (sequential in K)
        __bf16* st = lds + BOFF + sb * 2 * SB;
#pragma unroll
        for (int j = 0; j < NT / 2; ++j) {
            __bf16* dst = st + (16 * ((NT / 2) * wave + j)) * HLD;
            if (i == 2 * j) __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, dst, 16, boff[j], 0, 0, 0);
            if (i == 2 * j + 1) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, dst + SB, 16, boff[j], b_lo, 0, 0);
                boff[j] += 16384;
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
    // the first stages of a tile: per-tap form A(0) B(0) A(1) B(1); RS form A(group 0) B(0) B(1)
    auto issue_prologue = [&]() {
        issue_a(0);
        issue_b(0);
        if (nk > 1) {
            if (!RS) issue_a(1);
            issue_b(1);
        }
    };

    const char* ldsb = reinterpret_cast<const char*>(lds);
    const int off_l = lr * 64 + 16 * (lc ^ (2 * ((lr >> 2) & 1)));        // this lane's fragment chunk inside a 16-row block
    const int a_row0 = AOFF * 2 + wm * 64 * 64, b_row0 = BOFF * 2 + wn * 32 * NT * 64;    // bytes, wave-uniform

    for (int ti = slot; ti < chunk_n; ti += nslot) {
        const int lid = chunk0 + ti;
        const int tn = lid % p.ntn, tm = lid / p.ntn;
        const int m0 = tm * HROWS, n0 = tn * DBN;
        if (!prefetched) {
            tile_geom(m0, n0);
            issue_prologue();
        }
        prefetched = false;

        f32x4 acc[NB][4];                            // [channel block][pixel block]: lane holds channels 4*lc .. +3 of pixel lr
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};

        // image-edge lanes of the row-shift form: bit i = pixel block i sits at x == 0, bit 4 + i = at x == W - 1
        unsigned xedge = 0;
        if (RS) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int x_ = (m0 + wm * 64 + 16 * i + lr) & (p.W - 1);          // W divides 256: a power of two
                xedge |= (x_ == 0 ? 1u : 0u) << i | (x_ == p.W - 1 ? 1u : 0u) << (4 + i);
            }
        }
        h16x8 xh[4], xl[4], wh[HB], wl[HB];
        // fragment reads.  x: plane pl of activation stage sa, pixel block i, rows shifted by dx (RS); w: weight stage sb, block jj
        auto rd_x = [&](int sa, int pl, int i, int dx) -> h16x8 {
            int o = off_l;
            if (RS) {
                const int r_ = 16 * i + lr + dx;                                  // row inside the wave's 64 (may be -1 / 64: neighbour rows)
                o = r_ * 64 + 16 * (lc ^ (2 * (((r_ + 64) >> 2) & 1))) - 16 * i * 64;
                const bool edge = (dx < 0 && ((xedge >> i) & 1u)) || (dx > 0 && ((xedge >> (4 + i)) & 1u));
#if defined(DUA_RS_ZERO_SELECT) && DUA_RS_ZERO_SELECT
                const h16x8 v_ = *(const h16x8*)(ldsb + (a_row0 + sa * 2 * SA * 2 + pl * SA * 2 + 16 * i * 64) + o);
                const h16x8 z_ = {0, 0, 0, 0, 0, 0, 0, 0};
                return edge ? z_ : v_;                                              // data-select form (eight v_cndmask per fragment)
#else
                if (edge) return *(const h16x8*)(ldsb + OOBL + (lane & 1) * 16);   // zeros (address beyond the allocation)
#endif
            }
            return *(const h16x8*)(ldsb + (a_row0 + sa * 2 * SA * 2 + pl * SA * 2 + 16 * i * 64) + o);
        };
        auto rd_w = [&](int sb, int pl, int jj) -> h16x8 {
            return *(const h16x8*)(ldsb + (b_row0 + sb * 2 * SB * 2 + pl * SB * 2 + 16 * jj * 64) + off_l);
        };
#define HSB() __builtin_amdgcn_sched_barrier(0)
#define HMFMA(W_, X_, JB)                                                                              \
        acc[(JB) + jj_][i_] = __builtin_amdgcn_mfma_f32_16x16x32_f16(W_[jj_], X_[i_], acc[(JB) + jj_][i_], 0, 0, 0);

        // ---- tile start: the first stages have landed (this wave's pieces), everybody's after the barrier
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        int sa = 0, sb = 0, dxi = 0, grp = 0;
        const int ngroups = nk / 3;
        bool a_inflight = false;
#pragma unroll
        for (int i = 0; i < 4; ++i) { xl[i] = rd_x(0, 1, i, RS ? -1 : 0); xh[i] = rd_x(0, 0, i, RS ? -1 : 0); }
#pragma unroll
        for (int jj = 0; jj < HB; ++jj) { wh[jj] = rd_w(0, 0, jj); wl[jj] = rd_w(0, 1, jj); }
        HSB();

        for (int s = 0; s < nk; ++s) {
            // what this step requests: per-tap A(s+2) (ring 3) early, B(s+2) after the barrier (ring 2) or early (ring 3);
            // RS: B(s+2) early (ring 3), A(group + 1) early in the group's first step (ring 2)
            const bool preA = RS ? (dxi == 0 && grp + 1 < ngroups) : (s + 2 < nk);
            const bool preB = s + 2 < nk;
            constexpr bool B_EARLY = RB == 3;
            int sa_n = RS ? (sa ^ 1) : (sa + 2 >= RA ? sa + 2 - RA : sa + 2);
            int sb_n = sb + 2 >= RB ? sb + 2 - RB : sb + 2;
            if (preA) a_setup();
            auto early_piece = [&](int q) {          // DMA pieces issued under H0.P1: weights first (RS), then activations
                if (B_EARLY) {
                    if (q < NBP) { if (preB) dma_b(sb_n, q); }
                    else if (q < NBP + 4) { if (preA) dma_a(sa_n, q - NBP); }
                } else if (q < 4) { if (preA) dma_a(sa_n, q); }
            };
            const int dxc = RS ? dxi - 1 : 0;                     // dx of this step (RS)
            int dxn = dxc + 1, sa_r = sa;                          // dx / activation stage of the NEXT step's fragments
            if (RS) { if (dxi == 2) { dxn = -1; sa_r = sa ^ 1; } }
            else sa_r = sa + 1 == RA ? 0 : sa + 1;
            const int sb_r = sb + 1 == RB ? 0 : sb + 1;
            const bool more = s + 1 < nk;
            HSB();
            // ======== channel half 0
            // P1: w_hi * x_lo | the early DMA pieces, one per few MFMAs
#pragma unroll
            for (int jj_ = 0; jj_ < HB; ++jj_)
#pragma unroll
                for (int i_ = 0; i_ < 4; ++i_) {
                    HMFMA(wh, xl, 0) HSB();
                    const int m_ = jj_ * 4 + i_;
                    if (HB == 4) { if ((m_ & 1) == 0) early_piece(m_ >> 1); }
                    else early_piece(m_);
                    HSB();
                }
            // P2: w_hi * x_hi | w_hi of half 1 replaces each fragment after its last use
#pragma unroll
            for (int jj_ = 0; jj_ < HB; ++jj_) {
#pragma unroll
                for (int i_ = 0; i_ < 4; ++i_) { HMFMA(wh, xh, 0) HSB(); }
                wh[jj_] = rd_w(sb, 0, HB + jj_);
                HSB();
            }
            // P3: w_lo * x_hi | w_lo of half 1
#pragma unroll
            for (int jj_ = 0; jj_ < HB; ++jj_) {
#pragma unroll
                for (int i_ = 0; i_ < 4; ++i_) { HMFMA(wl, xh, 0) HSB(); }
                wl[jj_] = rd_w(sb, 1, HB + jj_);
                HSB();
            }
            // ======== channel half 1
            // P1': w_hi' * x_lo, then every read of stage s has been issued: publish stage s + 1
#pragma unroll
            for (int jj_ = 0; jj_ < HB; ++jj_)
#pragma unroll
                for (int i_ = 0; i_ < 4; ++i_) { HMFMA(wh, xl, HB) HSB(); }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (RS) {
                if (preA) a_inflight = true;
                const int keep = (preB ? NBP : 0) + ((a_inflight && dxi != 2) ? 4 : 0);
                if (keep >= NBP + 4) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NBP + 4) : "memory");
                else if (keep >= 4 && NBP < 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else if (keep >= NBP) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NBP) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else if (B_EARLY) {                    // per-tap, weight ring 3: A(s+2) and B(s+2) may stay in flight
                if (preA) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NBP + 4) : "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else {                                 // per-tap, weight ring 2: only A(s+2) may stay in flight
                if (preA) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();
            HSB();
            // P2': w_hi' * x_hi | x_lo of the next step, then w_hi (half 0) of the next stage; late weight DMA (ring 2)
#pragma unroll
            for (int jj_ = 0; jj_ < HB; ++jj_) {
#pragma unroll
                for (int i_ = 0; i_ < 4; ++i_) {
                    HMFMA(wh, xh, HB) HSB();
                    if (jj_ == 0 && more) { xl[i_] = rd_x(sa_r, 1, i_, dxn); HSB(); }
                    if (!B_EARLY && jj_ >= 1 && i_ == 0 && jj_ - 1 < NBP && preB) { dma_b(sb_n, jj_ - 1); HSB(); }
                }
                if (more) wh[jj_] = rd_w(sb_r, 0, jj_);
                HSB();
            }
            // P3': w_lo' * x_hi | w_lo (half 0) and x_hi of the next step
            if (!B_EARLY && HB - 1 < NBP && preB) { dma_b(sb_n, NBP - 1); HSB(); }
#pragma unroll
            for (int jj_ = 0; jj_ < HB; ++jj_) {
#pragma unroll
                for (int i_ = 0; i_ < 4; ++i_) {
                    HMFMA(wl, xh, HB) HSB();
                    if (jj_ == HB - 1 && more) { xh[i_] = rd_x(sa_r, 0, i_, dxn); HSB(); }
                }
                if (more) wl[jj_] = rd_w(sb_r, 1, jj_);
                HSB();
            }
            // advance the rings
            sb = sb_r;
            if (RS) {
                if (dxi == 2) { sa ^= 1; dxi = 0; ++grp; a_inflight = false; } else ++dxi;
            } else {
                sa = sa_r;
            }
            // the last barrier of the tile has retired every LDS read: request the first stages of the next tile now,
            // their latency runs under the epilogue
            if (!more && ti + nslot < chunk_n) {
                const int nl = chunk0 + ti + nslot;
                tile_geom((nl / p.ntn) * HROWS, (nl % p.ntn) * DBN);
                issue_prologue();
                prefetched = true;
            }
        }
#undef HMFMA
#undef HSB

        // ---- epilogue, straight from the accumulators: lane = pixel (row0 + 16 i + lr), channels c0 + 16 j + 4 lc .. + 3
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));             // keep the epilogue's addresses out of the K loop's register budget
        const int lr_e = lane_e & 15, lc_e = lane_e >> 4;
        const int row0 = m0 + wm * 64, c0 = n0 + wn * 32 * NT + 4 * lc_e;
        const bool vec = !p.nchw_out && (p.N & 3) == 0;
        if (vec) {
            // Channel-block pairs (32 channels) of one 16-pixel block go through a 16 x 144-byte LDS patch of the wave (the
            // activation ring's last slot: the next tile's prologue DMA targets the other slots) so that the residual read and
            // the output store move whole 128-byte rows, 8 pixels per instruction; straight from the accumulator layout they
            // would be 64-byte runs on 16 different rows per instruction.
            char* pw = reinterpret_cast<char*>(lds) + (AOFF + (RA - 1) * 2 * SA) * 2 + wave * (16 * 144);
            const int pr_e = lane_e >> 3, pc_e = lane_e & 7;                 // row-major side: pixel of the half block, 16-B chunk of the row
#pragma unroll
            for (int jp = 0; jp < NB / 2; ++jp) {
                const int c = n0 + wn * 32 * NT + 32 * jp + 4 * pc_e;
                const bool cok = c < p.N;
                float4 sq = *(const float4*)(p.Cscale + c);                  // (padded to the tile grid)
                sq.x *= p.out_scale; sq.y *= p.out_scale; sq.z *= p.out_scale; sq.w *= p.out_scale;
                float4 bq = make_float4(0.f, 0.f, 0.f, 0.f);
                if (p.bias && cok) bq = *(const float4*)(p.bias + c);
                // residual rows are requested one pixel block ahead of their use
                float4 rq[2], rn[2];
                auto res_ld = [&](int i, float4 (&dst)[2]) {
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int row = row0 + 16 * i + 8 * h + pr_e;
                        dst[h] = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (p.res && cok && row < p.M) dst[h] = *(const float4*)(p.res + (long)row * p.N + c);
                    }
                };
                res_ld(0, rn);
                float4 cs = make_float4(0.f, 0.f, 0.f, 0.f), cq2 = cs;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    rq[0] = rn[0]; rq[1] = rn[1];
                    if (i < 3) res_ld(i + 1, rn);
                    *(f32x4*)(pw + lr_e * 144 + 16 * lc_e) = acc[2 * jp][i];
                    *(f32x4*)(pw + lr_e * 144 + 64 + 16 * lc_e) = acc[2 * jp + 1][i];
                    // (same-wave LDS write -> read: in-order LDS queue, no barrier)
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const f32x4 a = *(const f32x4*)(pw + (8 * h + pr_e) * 144 + 16 * pc_e);
                        const int row = row0 + 16 * i + 8 * h + pr_e;
                        float4 v;
                        v.x = a[0] * sq.x + bq.x + rq[h].x;
                        v.y = a[1] * sq.y + bq.y + rq[h].y;
                        v.z = a[2] * sq.z + bq.z + rq[h].z;
                        v.w = a[3] * sq.w + bq.w + rq[h].w;
                        if (cok && row < p.M) {
                            *(float4*)(p.C + (long)row * p.N + c) = v;
                            cs.x += v.x; cs.y += v.y; cs.z += v.z; cs.w += v.w;
                            cq2.x += v.x * v.x; cq2.y += v.y * v.y; cq2.z += v.z * v.z; cq2.w += v.w * v.w;
                        }
                    }
                }
                if (p.colsum) {      // GroupNorm statistics of the next layer: fold the 8 pixel lanes that share these channels
                    float t[8] = {cs.x, cs.y, cs.z, cs.w, cq2.x, cq2.y, cq2.z, cq2.w};
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        t[u] += __shfl_xor(t[u], 8);
                        t[u] += __shfl_xor(t[u], 16);
                        t[u] += __shfl_xor(t[u], 32);
                    }
                    if (pr_e == 0 && cok && row0 < p.M) {
                        float* o = p.colsum + (long)(row0 >> 6) * 2 * p.N + c;
                        *(float4*)o = make_float4(t[0], t[1], t[2], t[3]);
                        *(float4*)(o + p.N) = make_float4(t[4], t[5], t[6], t[7]);
                    }
                }
            }
        } else {
            const int hw = p.H * p.W;
#pragma unroll
            for (int j = 0; j < NB; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int c = c0 + 16 * j + r;
                    if (c >= p.N) continue;
                    const float sv = p.Cscale[c] * p.out_scale, bv = p.bias ? p.bias[c] : 0.f;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int row = row0 + 16 * i + lr_e;
                        if (row >= p.M) continue;
                        float v = acc[j][i][r] * sv + bv;
                        if (p.res) v += p.res[(long)row * p.N + c];
                        if (p.nchw_out) {
                            const int n = row / hw, rem = row - n * hw;
                            p.C[((long)n * p.N + c) * hw + rem] = v;
                        } else {
                            p.C[(long)row * p.N + c] = v;
                        }
                    }
                }
        }
    }
#endif
}

int dua_launch_hconv(DConvP& d, hipStream_t st, const char* what) {
    d.ntm = (d.M + HROWS - 1) / HROWS;
    // 256-channel tiles (half the L2 -> LDS bytes per MFMA) when they tile N exactly and still give every CU two tiles
    const bool wide = d.N % 256 == 0 && (long)d.ntm * (d.N / 256) >= 512;
    d.ntn = wide ? d.N / 256 : (d.N + 127) / 128;
    const int ntiles = d.ntm * d.ntn, grid = ntiles < 256 ? ntiles : 256;       // one persistent workgroup per CU
    const bool rs = d.ntaps == 9 && 256 % d.W == 0 && d.K / 32 >= 3;
    if (rs) {
        if (wide) hipLaunchKernelGGL((k_hconv<4, true>), dim3(grid), 512, 0, st, d);
        else hipLaunchKernelGGL((k_hconv<2, true>), dim3(grid), 512, 0, st, d);
    } else {
        if (wide) hipLaunchKernelGGL((k_hconv<4, false>), dim3(grid), 512, 0, st, d);
        else hipLaunchKernelGGL((k_hconv<2, false>), dim3(grid), 512, 0, st, d);
    }
    return dua_check_launch(what);
}
