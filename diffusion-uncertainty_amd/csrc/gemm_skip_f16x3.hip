// f16x3 GEMM over a RAW fp32 operand:  C[M][N] = [A1 | A2][M][K1 + K2] * W[N][K]^T + bias  on v_mfma_f32_16x16x32_f16 (gfx950) - the 1x1
// skip convolutions of the ADM ResBlocks (unet_openai.py:219-223,257: `self.skip_connection(x)` over the never-materialised
// torch.cat([h, hs.pop()], dim=1) of :662), whose operand is the fp32 residual stream itself and exists nowhere as f16x3 planes.
//
// The previous engine for these launches (k_xtile<..., F16> of mfma_bf16x3.hip: a non-persistent 128 x 128 tile, global -> registers ->
// split -> ds_write staging, one barrier per 24 MFMAs) ran them at 0.10 of the fp16 matrix peak and 0.21 of HBM - bound by neither
// (VERDICT r04 weak 2).  This kernel is k_hgemm's loop (gemm_f16x3.hip: persistent 256-row tiles, static LDS rings, weight stages and
// activation stages by `buffer_load ... lds`, one s_barrier per K step of 48 / 96 MFMAs per wave, the next tile's first stages requested
// before the epilogue) with the activation ring holding the RAW fp32 rows:
//   * an activation stage is 256 rows x 32 floats = 32 KB (the bytes of k_hgemm's hi + lo planes), written by the LDS-DMA in 8-row x
//     128-byte pieces - every piece reads eight whole 128-byte lines of A, once;
//   * the split a * S = hi + lo (fp16 hi, fp16 remainder; S a power of two) happens in REGISTERS after the fragment read, four VALU
//     operations per element between the MFMAs of the same step - no staging registers across the barrier, no ds_write, no second LDS
//     image.  The two waves that share a row block (wn = 0, 1) both split it: with all N = 256 columns of the 128^2 / 64^2 levels in ONE
//     tile an operand element is split twice per launch instead of 2 x ceil(N / 128) times;
//   * 256 x 256 or 256 x 128 tiles chosen per shape by whole rounds on 256 CUs; both widths run the same per-element sequence of
//     products, so the result does not depend on the width (or on M: per-image results are batch-invariant).
// LDS image of an activation stage: row r = 128 bytes, 16-byte chunk c at chunk position c ^ g(r), g(r) = (((r & 15) >> 1) + 6) & 7.
// A fragment read is two ds_read_b128 per 16-row block (lane = row lane & 15, floats 8 (lane >> 4) .. + 7 = chunks 2 lc, 2 lc + 1); its
// four 16-lane groups ({0-3, 12-15, 20-27}, ... MI355X_MICROARCH.md section LDS) hold all 16 rows with lc differing by one between the row
// sets {0-3, 12-15} and {4-11}: g maps the row pairs of the second set onto {0..3} and the others onto {4..7}, both closed under the
// XOR with 2 that the lc difference applies, so the 16 lanes of a group land on 16 distinct 16-byte slots of the 256-byte bank row.
// The DMA writes lane-linear pieces, so the swizzle is applied to each lane's SOURCE chunk.
#include "common.h"
#include "hgemm.h"
#include <type_traits>

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define SG_ROWS 256
// Timing-only diagnostic builds (scripts/diag/sgemm_diag.py; WRONG results, never in the product library): bit 0 drops the epilogue,
// bit 1 the operand split (the raw bits are used as fragments), bit 2 the prefetch of the next item's first stages.
#ifndef SG_DIAG
#define SG_DIAG 0
#endif
#define SG_TAIL_WAIT 8               // vector-memory operations an epilogue ALWAYS issues after the next item's prologue DMA (stores)

// hi = fp16(x S) (RNE, saturating), lo = fp16(x S - hi): the arithmetic of split4h (mfma_bf16x3.hip) on the eight floats of a fragment.
__device__ __forceinline__ void sg_split8(const f32x4 r0, const f32x4 r1, float sc, h16x8& hi, h16x8& lo, bool rc_on, unsigned& n_sat, unsigned& n_sub) {
    const float f[8] = {r0[0] * sc, r0[1] * sc, r0[2] * sc, r0[3] * sc, r1[0] * sc, r1[1] * sc, r1[2] * sc, r1[3] * sc};
    if (rc_on) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float af = fabsf(f[j]);
            n_sat += af > 65504.f ? 1u : 0u;
            n_sub += (af > 0.f && af < 6.103515625e-05f) ? 1u : 0u;
        }
    }
    unsigned hp[4], lp[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float c0 = __builtin_amdgcn_fmed3f(f[2 * j], -65504.f, 65504.f), c1 = __builtin_amdgcn_fmed3f(f[2 * j + 1], -65504.f, 65504.f);
        const h16x2 h2 = {(_Float16)c0, (_Float16)c1};
        hp[j] = __builtin_bit_cast(unsigned, h2);
        const float g0 = __builtin_amdgcn_fmed3f(f[2 * j], -131008.f, 131008.f), g1 = __builtin_amdgcn_fmed3f(f[2 * j + 1], -131008.f, 131008.f);
        unsigned d;
        asm("v_fma_mixlo_f16 %0, %1, 1.0, -%2 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(d) : "v"(g0), "v"(hp[j]));
        asm("v_fma_mixhi_f16 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(d) : "v"(g1), "v"(hp[j]));
        lp[j] = d;
    }
    hi = __builtin_bit_cast(h16x8, (u32x4){hp[0], hp[1], hp[2], hp[3]});
    lo = __builtin_bit_cast(h16x8, (u32x4){lp[0], lp[1], lp[2], lp[3]});
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Epilogue of one 256 x (64 NT) tile: k_hgemm's patch form (channel-block pairs of one 16-row block through a 16 x 128-byte LDS patch
// of the wave, two patches alternating, whole 128-byte rows out) without residual / extra planes / column sums.  sq / bq: the column
// scales and biases of the wave's NT pairs (channels 4 (lane & 7) .. + 3 of pair jp), loaded by the caller BEFORE the next item's
// prologue DMA was issued, so that waiting for them does not wait for that DMA (vmcnt retires in order).  Every output store is a
// buffer store that ALWAYS issues (lanes outside the tensor get an out-of-range offset, which the buffer unit drops).
template <int NT>
__device__ __forceinline__ void sgemm_epilogue(const SGemmP& p, f32x4 (&acc)[2 * NT][4], const float4 (&sq)[NT], const float4 (&bq)[NT], char* pw,
                                               int wave, int lane, int m0, int n0) {
    static_assert(8 * NT >= SG_TAIL_WAIT, "the item-start wait counts on this many stores");
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));             // keep the epilogue's addresses out of the K loop's register budget
    const int wm = wave >> 1, wn = wave & 1;
    const int lr_e = lane_e & 15, lc_e = lane_e >> 4;
    const int row0 = m0 + wm * 64;
    const unsigned OOB = 0xFFFFFFF0u;
    constexpr int T = 4 * NT;                                    // slots (pair jp = t >> 2, row block i = t & 3)
    const int wr_off = lr_e * 128, wr_sw = (lr_e >> 1) & 7;
    auto patch_put = [&](int t) {
        char* b = pw + (t & 1) * 2048 + wr_off;
        *(f32x4*)(b + 16 * (lc_e ^ wr_sw)) = acc[2 * (t >> 2)][t & 3];
        *(f32x4*)(b + 16 * ((4 + lc_e) ^ wr_sw)) = acc[2 * (t >> 2) + 1][t & 3];
    };
    const long crem = ((long)p.M - m0) * p.N * 4;
    const auto rc = __builtin_amdgcn_make_buffer_rsrc((void*)(p.C + (long)m0 * p.N), 0, (int)(crem < 0x7FFFFFF0L ? crem : 0x7FFFFFF0L), 0x00020000);
    const int pr_e = lane_e >> 3, pc_e = lane_e & 7;             // row-major side: row of the half block, 16-B chunk of the 128-B row
    patch_put(0);
    patch_put(1);
    f32x4 pa[2][2];
    auto patch_get = [&](int t, f32x4 (&d)[2]) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int r = 8 * h + pr_e;
            d[h] = *(const f32x4*)(pw + (t & 1) * 2048 + r * 128 + 16 * (pc_e ^ ((r >> 1) & 7)));
        }
    };
    patch_get(0, pa[0]);
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const int jp = t >> 2, i = t & 3;
        const int c = n0 + wn * 32 * NT + 32 * jp + 4 * pc_e;
        const bool cok = c < p.N;
        if (t + 1 < T) patch_get(t + 1, pa[(t + 1) & 1]);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const f32x4 a = pa[t & 1][h];
            const int row = row0 + 16 * i + 8 * h + pr_e;
            const u32x4 vb = {__float_as_uint(fmaf(a[0], sq[jp].x, bq[jp].x)), __float_as_uint(fmaf(a[1], sq[jp].y, bq[jp].y)),
                              __float_as_uint(fmaf(a[2], sq[jp].z, bq[jp].z)), __float_as_uint(fmaf(a[3], sq[jp].w, bq[jp].w))};
            __builtin_amdgcn_raw_buffer_store_b128(vb, rc, (cok && row < p.M) ? (unsigned)(((long)(row - m0) * p.N + c) * 4) : OOB, 0, 0);
        }
        if (t + 2 < T) patch_put(t + 2);         // (its patch was read for slot t: the LDS queue is in order)
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
template <int NT, bool RC>
__global__ __launch_bounds__(512, 2) void k_sgemm(SGemmP p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int NB = 2 * NT;                               // 16-channel blocks of a wave (wave tile = 64 rows x 32 NT channels)
    constexpr int HB = 2, H2 = NB / HB;                      // the weight fragments are held two channel blocks at a time
    constexpr int RA = 3, RB = NT == 2 ? 3 : 2;              // ring depths (stages)
    constexpr int ABY = SG_ROWS * 128;                       // bytes per activation stage (raw fp32 rows)
    constexpr int BPL = 64 * NT * 64, BBY = 2 * BPL;         // weight stage: hi | lo planes of 64 NT rows x 64 B
    constexpr int BOFF = RA * ABY;
    static_assert(RA * ABY + RB * BBY <= 160 * 1024, "LDS of a gfx950 CU");
    __shared__ __attribute__((aligned(16))) char lds[RA * ABY + RB * BBY];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int nk = (p.K1 + p.K2) / 32, nk1 = p.K1 / 32;
    // persistent workgroups; whole tiles chunked per XCD (blockIdx % 8): neighbouring row tiles of the same operand panel share an L2
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslot = (gridDim.x + 7 - xcd) >> 3;
    const int ntiles = p.ntm * p.ntn;
    const int cq = ntiles >> 3, cr = ntiles & 7;
    const int chunk0 = xcd < cr ? xcd * (cq + 1) : cr * (cq + 1) + (xcd - cr) * cq, chunk_n = cq + (xcd < cr ? 1 : 0);
    const int n_items = slot < chunk_n ? (chunk_n - slot + nslot - 1) / nslot : 0;

    const int lr = lane & 15, lc = lane >> 4;

    // ---- DMA state (lives across items: the next item's first stages are requested before this item's epilogue)
    // Activation pieces: piece q = 0..3 of a stage covers rows 32 wave + 8 q .. + 7, lane = (row lane >> 3, chunk position lane & 7); the
    // per-lane offset (row * lda * 4 + 16 * source chunk) is bounds-checked against the rows left below the tile (rows >= M read zeros),
    // the K step rides in the scalar offset.  The swizzle depends on (row & 15) = 8 (q & 1) + (lane >> 3): two per-lane offsets per source.
    // Weight pieces: as in k_hgemm.
    const int dr = lane >> 3, dc = lane & 7;
    const unsigned a_row = (unsigned)(32 * wave + dr);                                  // this lane's row of piece 0
    const unsigned a_cg = 16u * (unsigned)(dc ^ (((dr >> 1) + 6) & 7));                 // source chunk (bytes) of even pieces; odd pieces: ^ 64 (g + 4 = g ^ 4 mod 8)
    unsigned lda4_1 = 0, lda4_2 = 0;                         // bytes per row
    const int dl_row = lane >> 2;
    const int dl_src = (lane & 3) ^ (2 * ((lane >> 4) & 1));
    const unsigned blane = (unsigned)(dl_row * 64 + 16 * dl_src);
    unsigned sb_off[NT];
    int bdst[NT];
    auto ra1 = __builtin_amdgcn_make_buffer_rsrc((void*)p.A1, 0, 0, 0x00020000);
    auto ra2 = ra1;
    const auto rb = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, (int)p.b_bytes, 0x00020000);
    const int nblk = (int)(p.b_bytes / ((long)nk * 16384));  // packed 128-row blocks
    int ka = 0;
    bool a_src2 = false;
    unsigned a_soff = 0;
#pragma unroll
    for (int j = 0; j < NT; ++j) { sb_off[j] = 0; bdst[j] = 0; }

    auto item_geom = [&](int m0_, int n0_) {
        const long rows = (long)p.M - m0_;
        const long rec1 = rows * p.lda1 * 4, rec2 = rows * p.lda2 * 4;
        ra1 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A1 + (long)m0_ * p.lda1), 0, (int)(rec1 < 0x7FFFFFF0L ? rec1 : 0x7FFFFFF0L), 0x00020000);
        if (p.K2) ra2 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A2 + (long)m0_ * p.lda2), 0, (int)(rec2 < 0x7FFFFFF0L ? rec2 : 0x7FFFFFF0L), 0x00020000);
        lda4_1 = (unsigned)p.lda1 * 4u;
        lda4_2 = (unsigned)p.lda2 * 4u;
        ka = 0;
#pragma unroll
        for (int j = 0; j < NT; ++j) {                       // weight pieces of this wave: piece q = wave + 8 j of the stage's 8 NT (plane, 16-row group) pieces
            const int q = wave + 8 * j, pl = q / (4 * NT), g = q - pl * 4 * NT;
            const int rr = n0_ + 16 * g;
            int blk = rr >> 7;
            blk = blk < nblk ? blk : nblk - 1;               // column tiles beyond the packed rows re-read the last block (never stored)
            sb_off[j] = (unsigned)(((long)blk * nk) * 16384 + pl * 8192 + (rr & 127) * 64);
            bdst[j] = pl * BPL + g * 1024;
        }
    };
    auto a_setup = [&]() {                                   // source and scalar offset of activation stage ka
        a_src2 = ka >= nk1;
        a_soff = (unsigned)(a_src2 ? ka - nk1 : ka) * 128u;
        ++ka;
    };
    auto dma_a = [&](int sa, int q) {                        // piece q = 0..3 of the activation stage a_setup() prepared
        char* st = lds + sa * ABY + (32 * wave + 8 * q) * 128;
        const auto r = a_src2 ? ra2 : ra1;
        // The per-lane offset is formed HERE, by instructions the compiler cannot hoist (row * pitch + chunk, + 8 q rows): the eight sums
        // are kernel invariants and would otherwise live - spilled - across the K loop, and a reload in front of a DMA is a vmcnt wait.
        const unsigned pitch = a_src2 ? lda4_2 : lda4_1;
        unsigned v;
        asm volatile("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(v) : "v"(a_row), "s"(pitch), "v"((q & 1) ? (a_cg ^ 64u) : a_cg));
        if (q) asm volatile("v_add_u32 %0, %1, %0" : "+v"(v) : "s"(8u * (unsigned)q * pitch));
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, st, 16, v, a_soff, 0, 0);
    };
    auto dma_b = [&](int sb, int j) {                        // piece j = 0..NT-1 of the next weight stage (sequential in K)
        char* st = lds + BOFF + sb * BBY;
#pragma unroll
        for (int jj = 0; jj < NT; ++jj)
            if (jj == j) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, st + bdst[jj], 16, blane, sb_off[jj], 0, 0);
                sb_off[jj] += 16384;
            }
    };
    auto issue_prologue = [&]() {                            // A(0) B(0) A(1) B(1)  (nk >= 2: checked on the host)
#pragma unroll
        for (int s_ = 0; s_ < 2; ++s_) {
            a_setup();
#pragma unroll
            for (int q = 0; q < 4; ++q) dma_a(s_, q);
#pragma unroll
            for (int j = 0; j < NT; ++j) dma_b(s_, j);
        }
    };

    const char* ldsb = reinterpret_cast<const char*>(lds);
    const int g_l = ((lr >> 1) + 6) & 7;
    const int offa_l = lr * 128 + 16 * ((2 * lc) ^ g_l);                   // first chunk of this lane's raw fragment (the second: ^ 16)
    const int offb_l = lr * 64 + 16 * (lc ^ (2 * ((lr >> 2) & 1)));        // this lane's weight fragment chunk inside a 16-row block
    int axs[RA], bws[RB];
#pragma unroll
    for (int s_ = 0; s_ < RA; ++s_) axs[s_] = s_ * ABY + wm * 64 * 128 + offa_l;
#pragma unroll
    for (int s_ = 0; s_ < RB; ++s_) bws[s_] = BOFF + s_ * BBY + wn * 32 * NT * 64 + offb_l;

    unsigned n_sat = 0, n_sub = 0;                           // f16x3 range check (debug facility, its own instantiation): counted once per operand element

    int tile = 0;
    bool prefetched = false;
    for (int it = 0; it < n_items; ++it) {
        if (!prefetched) {
            tile = chunk0 + slot + it * nslot;
            item_geom((tile / p.ntn) * SG_ROWS, (tile % p.ntn) * 64 * NT);
            issue_prologue();
        }
        const int m0 = (tile / p.ntn) * SG_ROWS, n0 = (tile % p.ntn) * 64 * NT;
        const bool after_epilogue = prefetched;
        const bool rc_on = RC && wn == 0 && n0 == 0;
        prefetched = false;

        f32x4 acc[NB][4];
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        h16x8 xh[4], xl[4], wh[HB], wl[HB];
        f32x4 raw[4][2];
        auto rd_raw = [&](int sa, int i) {
            raw[i][0] = *(const f32x4*)(ldsb + axs[sa] + i * 2048);
            raw[i][1] = *(const f32x4*)(ldsb + (axs[sa] ^ 16) + i * 2048);
        };
        auto split = [&](int i) {
            if (SG_DIAG & 2) {
                xh[i] = __builtin_bit_cast(h16x8, raw[i][0]);
                xl[i] = __builtin_bit_cast(h16x8, raw[i][1]);
            } else {
                sg_split8(raw[i][0], raw[i][1], p.a_scale, xh[i], xl[i], rc_on, n_sat, n_sub);
            }
        };
        auto rd_w = [&](int sb, int pl, int jj) -> h16x8 { return *(const h16x8*)(ldsb + bws[sb] + (pl * BPL + jj * 1024)); };
#define GSB() __builtin_amdgcn_sched_barrier(0)

        // ---- item start: the first stages have landed; behind them in the (in-order) queue are only the previous epilogue's
        // operations, of which the last SG_TAIL_WAIT - stores - may stay in flight
        if (after_epilogue) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(SG_TAIL_WAIT) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int i = 0; i < 4; ++i) rd_raw(0, i);
#pragma unroll
        for (int jj = 0; jj < HB; ++jj) { wh[jj] = rd_w(0, 0, jj); wl[jj] = rd_w(0, 1, jj); }
#pragma unroll
        for (int i = 0; i < 4; ++i) split(i);
        GSB();

        // One K step: 12 NB MFMAs per wave.  Stage s lives in activation slot SA_ = s % 3 and weight slot SB_ = s % RB (compile-time: the
        // loop body is six steps).  `more`: a stage s + 1 exists; `pre`: a stage s + 2 exists and is requested in this step.  The last
        // half's third product runs row block by row block: once block i's hi fragment has had its last use, the raw fragment of the
        // NEXT stage's block i (read one block earlier) is split into xh[i] / xl[i] - no second set of fragment registers.
        auto step = [&](auto sa_c, auto sb_c, auto tail_c, bool more_rt, bool pre_rt) __attribute__((always_inline)) {
            constexpr int SA_ = decltype(sa_c)::value, SB_ = decltype(sb_c)::value;
            constexpr bool TAIL = decltype(tail_c)::value;
            constexpr int sa_n = (SA_ + 2) % RA, sb_n = (SB_ + 2) % RB, sa_r = (SA_ + 1) % RA, sb_r = (SB_ + 1) % RB;
            const bool more = TAIL ? more_rt : true, pre = TAIL ? pre_rt : true;
            if (pre) a_setup();
            auto early_piece = [&](int q) {
                if (q < 4) { if (pre) dma_a(sa_n, q); }
                else if (RB == 3 && q < 4 + NT) { if (pre) dma_b(sb_n, q - 4); }
            };
            GSB();
#pragma unroll
            for (int h = 0; h < H2; ++h) {
                const bool lasth = h == H2 - 1;
                // P1: w_hi * x_lo | the early DMA pieces (before the step's barrier), one per MFMA pair
#pragma unroll
                for (int jj = 0; jj < HB; ++jj)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        acc[h * HB + jj][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[jj], xl[i], acc[h * HB + jj][i], 0, 0, 0);
                        GSB();
                        const int m_ = jj * 4 + i;
                        if ((m_ & 1) == 0) { early_piece(h * 2 * HB + (m_ >> 1)); GSB(); }
                    }
                if (lasth) {
                    // every read of stage s has been issued: publish stage s + 1
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    if (pre) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(RB == 3 ? 4 + NT : 4) : "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    GSB();
                }
                // P2: w_hi * x_hi | the hi weight fragments of the next pair of channel blocks (same stage) or of the next stage
#pragma unroll
                for (int jj = 0; jj < HB; ++jj) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        acc[h * HB + jj][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[jj], xh[i], acc[h * HB + jj][i], 0, 0, 0);
                        GSB();
                    }
                    if (!lasth) wh[jj] = rd_w(SB_, 0, (h + 1) * HB + jj);
                    else if (more) wh[jj] = rd_w(sb_r, 0, jj);
                    GSB();
                    if (lasth && RB == 2 && pre) {           // (weight ring of two: the slot of stage s + 2 is the one this step has just finished reading)
#pragma unroll
                        for (int q = jj * NT / HB; q < (jj + 1) * NT / HB; ++q) { dma_b(sb_n, q); GSB(); }
                    }
                    if (lasth && jj == HB - 1 && more) { rd_raw(sa_r, 0); GSB(); }
                }
                // P3: w_lo * x_hi | the lo weight fragments likewise
                if (!lasth) {
#pragma unroll
                    for (int jj = 0; jj < HB; ++jj) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            acc[h * HB + jj][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[jj], xh[i], acc[h * HB + jj][i], 0, 0, 0);
                            GSB();
                        }
                        wl[jj] = rd_w(SB_, 1, (h + 1) * HB + jj);
                        GSB();
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
#pragma unroll
                        for (int jj = 0; jj < HB; ++jj) {
                            acc[h * HB + jj][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[jj], xh[i], acc[h * HB + jj][i], 0, 0, 0);
                            GSB();
                            if (jj == 0 && i + 1 < 4 && more) { rd_raw(sa_r, i + 1); GSB(); }
                            if (i == 3 && more) { wl[jj] = rd_w(sb_r, 1, jj); GSB(); }
                        }
                        if (more) { split(i); GSB(); }
                    }
                }
            }
        };
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        using I2 = std::integral_constant<int, 2>;
        using BF = std::integral_constant<bool, false>;
        using BT = std::integral_constant<bool, true>;
        using B1 = std::integral_constant<int, 1 % RB>;
        using B2 = std::integral_constant<int, 2 % RB>;
        using B3 = std::integral_constant<int, 3 % RB>;
        using B4 = std::integral_constant<int, 4 % RB>;
        using B5 = std::integral_constant<int, 5 % RB>;
        const int n = nk;
        int s = 0;
#pragma nounroll
        for (; s + 8 <= n; s += 6) {                         // steady state: every one of the six steps has a stage s + 2 to request
            step(I0{}, I0{}, BF{}, true, true); step(I1{}, B1{}, BF{}, true, true); step(I2{}, B2{}, BF{}, true, true);
            step(I0{}, B3{}, BF{}, true, true); step(I1{}, B4{}, BF{}, true, true); step(I2{}, B5{}, BF{}, true, true);
        }
        {
            const int r = n - s;                              // 1 .. 7 steps left (slots continue from 0: s is a multiple of six)
            if (r > 0) step(I0{}, I0{}, BT{}, 1 < r, 2 < r);
            if (r > 1) step(I1{}, B1{}, BT{}, 2 < r, 3 < r);
            if (r > 2) step(I2{}, B2{}, BT{}, 3 < r, 4 < r);
            if (r > 3) step(I0{}, B3{}, BT{}, 4 < r, 5 < r);
            if (r > 4) step(I1{}, B4{}, BT{}, 5 < r, 6 < r);
            if (r > 5) step(I2{}, B5{}, BT{}, 6 < r, false);
            if (r > 6) step(I0{}, I0{}, BT{}, false, false);
        }
#undef GSB
        // Column scales and biases of this tile: loaded, waited for and scaled BEFORE the next item's first stages are requested (one L2-hit
        // latency per tile).  Behind the - conditional - prologue issue the compiler would wait vmcnt(0) at their first use, i.e. for the
        // DMA it has just queued (vmcnt retires in order); the empty asm pins the finished values in front of the issue.
        float4 sq[NT], bq[NT];
        {
            const int pc_e = lane & 7;
            // (buffer loads that always issue: columns >= N of the bias - or all of them when there is none - read zeros)
            const auto rbi = __builtin_amdgcn_make_buffer_rsrc((void*)p.bias, 0, p.bias ? p.N * 4 : 0, 0x00020000);
#pragma unroll
            for (int jp = 0; jp < NT; ++jp) {
                const int c = n0 + wn * 32 * NT + 32 * jp + 4 * pc_e;
                sq[jp] = *(const float4*)(p.Cscale + c);             // (padded to a multiple of 256 columns)
                const u32x4 b_ = __builtin_amdgcn_raw_buffer_load_b128(rbi, (unsigned)c * 4u, 0, 0);
                bq[jp] = make_float4(__uint_as_float(b_[0]), __uint_as_float(b_[1]), __uint_as_float(b_[2]), __uint_as_float(b_[3]));
            }
#pragma unroll
            for (int jp = 0; jp < NT; ++jp) {
                sq[jp].x *= p.out_scale; sq[jp].y *= p.out_scale; sq[jp].z *= p.out_scale; sq[jp].w *= p.out_scale;
                asm volatile("" : "+v"(sq[jp].x), "+v"(sq[jp].y), "+v"(sq[jp].z), "+v"(sq[jp].w), "+v"(bq[jp].x), "+v"(bq[jp].y), "+v"(bq[jp].z), "+v"(bq[jp].w)::"memory");
            }
        }
        // the last barrier of the item has retired every LDS read: request the first stages of the next item now (activation slots
        // 0, 1 and weight slots 0, 1); their latency runs under the epilogue, whose patches live in activation slot 2
        if (it + 1 < n_items && !(SG_DIAG & 4)) {
            tile = chunk0 + slot + (it + 1) * nslot;
            item_geom((tile / p.ntn) * SG_ROWS, (tile % p.ntn) * 64 * NT);
            issue_prologue();
            prefetched = true;
        }
        if (SG_DIAG & 1) {                                   // (diagnostic: no epilogue - the accumulators are kept alive by a store that never happens)
            float t_ = sq[0].x + bq[0].x;
#pragma unroll
            for (int j = 0; j < NB; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) t_ += acc[j][i][0] + acc[j][i][1] + acc[j][i][2] + acc[j][i][3];
            if (t_ == 1.2345e-30f) p.C[0] = t_;
        } else {
            sgemm_epilogue<NT>(p, acc, sq, bq, lds + 2 * ABY + wave * 4096, wave, lane, m0, n0);
        }
        // the patches must be done before another wave's DMA of the next item's third stage lands in activation slot 2: that piece is
        // requested after the next item's opening barrier
    }
    if (RC && (n_sat | n_sub)) {
        unsigned long long* rc_ = p.range_ctr;
        if (n_sat) atomicAdd(rc_, (unsigned long long)n_sat);
        if (n_sub) atomicAdd(rc_ + 1, (unsigned long long)n_sub);
    }
#endif
}

// ---------------------------------------------------------------------------------------------------------------------------------
static unsigned long long* sg_range_host = nullptr;
static int sg_force_nt = 0;               // A/B switch (process-global, correct kernels only): 0 = per shape, 2 / 4 = 128 / 256 columns, -1 = previous engine
int dua_sgemm_enabled() { return sg_force_nt >= 0; }
extern "C" int dua_set_sgemm_mode(int nt) {
    DUA_REQUIRE(nt == -1 || nt == 0 || nt == 2 || nt == 4, "dua_set_sgemm_mode: nt must be -1 (previous engine), 0, 2 or 4");
    sg_force_nt = nt;
    return DUA_OK;
}

int dua_launch_sgemm(SGemmP& p, hipStream_t st, const char* what) {
    const int nk = (p.K1 + p.K2) / 32;
    p.ntm = (p.M + SG_ROWS - 1) / SG_ROWS;
    // Tile width: rounds of tiles on 256 CUs x (K steps + the per-tile overhead in K steps) x width / loop efficiency of the width (the
    // wider tile moves half the L2 -> LDS weight bytes per MFMA and splits each operand element for twice the columns)
    int nt = sg_force_nt > 0 ? sg_force_nt : 0;
    if (!nt) {
        double best = 1e300;
        for (int c = 4; c >= 2; c -= 2) {
            const int w = 64 * c, tn = (p.N + w - 1) / w;
            const long tiles = (long)p.ntm * tn;
            const double t = (double)((tiles + 255) / 256) * (nk + 8) * c / (c == 4 ? 1.0 : 0.85);
            if (t < best) { best = t; nt = c; }
        }
    }
    p.ntn = (p.N + 64 * nt - 1) / (64 * nt);
    const int ntiles = p.ntm * p.ntn;
    const int grid = ntiles < 256 ? ntiles : 256;
    p.range_ctr = sg_range_host;
    if (p.range_ctr) {
        if (nt == 4) hipLaunchKernelGGL((k_sgemm<4, true>), dim3(grid), 512, 0, st, p);
        else hipLaunchKernelGGL((k_sgemm<2, true>), dim3(grid), 512, 0, st, p);
    } else {
        if (nt == 4) hipLaunchKernelGGL((k_sgemm<4, false>), dim3(grid), 512, 0, st, p);
        else hipLaunchKernelGGL((k_sgemm<2, false>), dim3(grid), 512, 0, st, p);
    }
    return dua_check_launch(what);
}

// (f16x3 range check, common.h: this translation unit keeps the counter pointer on the host and hands it to the kernel's checking instantiation)
void dua_range_set_sgemm(unsigned long long* p) { sg_range_host = p; }
