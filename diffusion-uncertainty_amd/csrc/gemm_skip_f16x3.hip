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
//     image.  The eight waves tile the 256 rows (wave w = rows 32 w .. + 31, ALL columns of the tile): every operand element is split by
//     exactly one wave, once per column tile - with all N = 256 columns of the 128^2 / 64^2 levels in ONE tile, once per launch (the
//     first version of this kernel used k_hgemm's 4 x 2 wave grid, where the two waves of a row block both split it: the split was 19 %
//     of the launch time, profiles/r05b_sgemm_diag.txt).  The weight fragments stream through two register sets, one pair of channel
//     blocks ahead of the MFMAs that use them;
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
#ifndef SG_STORE_AUX                  // cache policy bits of the output stores (A/B builds: 2 = non-temporal)
#define SG_STORE_AUX 0
#endif
#ifndef SG_SPLIT_SPREAD
#define SG_SPLIT_SPREAD 0
#endif
#define SG_TAIL_WAIT 8               // vector-memory operations an epilogue ALWAYS issues after the next item's prologue DMA (stores)

// hi = fp16(x S) (RNE, saturating), lo = fp16(x S - hi): the arithmetic of split4h (mfma_bf16x3.hip) on two neighbouring floats of a
// fragment -> one packed register of hi halves, one of lo halves (nine VALU operations; a step's 16 floats per lane are eight such chunks,
// placed one by one between the MFMAs of the step's last pairs).
__device__ __forceinline__ void sg_split2(float a, float b, float sc, unsigned& hp, unsigned& lp, bool rc_on, unsigned& n_sat, unsigned& n_sub) {
    const float f0 = a * sc, f1 = b * sc;
    if (rc_on) {
        const float a0 = fabsf(f0), a1 = fabsf(f1);
        n_sat += (a0 > 65504.f ? 1u : 0u) + (a1 > 65504.f ? 1u : 0u);
        n_sub += ((a0 > 0.f && a0 < 6.103515625e-05f) ? 1u : 0u) + ((a1 > 0.f && a1 < 6.103515625e-05f) ? 1u : 0u);
    }
    const float c0 = __builtin_amdgcn_fmed3f(f0, -65504.f, 65504.f), c1 = __builtin_amdgcn_fmed3f(f1, -65504.f, 65504.f);
    const h16x2 h2 = {(_Float16)c0, (_Float16)c1};
    hp = __builtin_bit_cast(unsigned, h2);
    const float g0 = __builtin_amdgcn_fmed3f(f0, -131008.f, 131008.f), g1 = __builtin_amdgcn_fmed3f(f1, -131008.f, 131008.f);
    unsigned d;
    asm("v_fma_mixlo_f16 %0, %1, 1.0, -%2 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(d) : "v"(g0), "v"(hp));
    asm("v_fma_mixhi_f16 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(d) : "v"(g1), "v"(hp));
    lp = d;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Epilogue of one 256 x (64 NT) tile: k_hgemm's patch form (channel-block pairs of one 16-row block through a 16 x 128-byte LDS patch
// of the wave, two patches alternating, whole 128-byte rows out) without residual / extra planes / column sums.  sqv / bqv: the column
// scales (times out_scale) and biases of columns n0 + 4 lane .. + 3 - the tile's 64 NT columns spread over the wave, 8 registers - loaded
// by the caller BEFORE the next item's prologue DMA was issued, so that waiting for them does not wait for that DMA (vmcnt retires in
// order); the lane that stores columns 32 jp + 4 (lane & 7) .. + 3 fetches them from lane 8 jp + (lane & 7) by ds_bpermute, once per
// pair.  Every output store is a buffer store that ALWAYS issues (lanes outside the tensor get an out-of-range offset: dropped).
template <int NT>
__device__ __forceinline__ void sgemm_epilogue(const SGemmP& p, f32x4 (&acc)[4 * NT][2], const float4 sqv, const float4 bqv, char* pw,
                                               int wave, int lane, int m0, int n0) {
    static_assert(8 * NT >= SG_TAIL_WAIT, "the item-start wait counts on this many stores");
    // the lane id is recomputed here (volatile: not shared with the K loop's copy, which would live - spilled - across the loop; its reload
    // in front of the first patch write would be an s_waitcnt vmcnt(0), i.e. a wait for the prologue DMA just queued)
    int lane_e;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_e));
    (void)lane;
    const int lr_e = lane_e & 15, lc_e = lane_e >> 4;
    const int row0 = m0 + wave * 32;
    const unsigned OOB = 0xFFFFFFF0u;
    constexpr int T = 4 * NT;                                    // slots (pair jp = t >> 1, row block i = t & 1)
    const int wr_off = lr_e * 128, wr_sw = (lr_e >> 1) & 7;
    auto patch_put = [&](int t) {
        char* b = pw + (t & 1) * 2048 + wr_off;
        *(f32x4*)(b + 16 * (lc_e ^ wr_sw)) = acc[2 * (t >> 1)][t & 1];
        *(f32x4*)(b + 16 * ((4 + lc_e) ^ wr_sw)) = acc[2 * (t >> 1) + 1][t & 1];
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
    float sq[4], bq[4];
    auto gather = [&](int jp) {
        const int src = 4 * (8 * jp + pc_e);
        const float s_[4] = {sqv.x, sqv.y, sqv.z, sqv.w}, b_[4] = {bqv.x, bqv.y, bqv.z, bqv.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            sq[k] = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(s_[k])));
            bq[k] = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(b_[k])));
        }
    };
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const int jp = t >> 1, i = t & 1;
        const int c = n0 + 32 * jp + 4 * pc_e;
        const bool cok = c < p.N;
        if (i == 0) gather(jp);
        if (t + 1 < T) patch_get(t + 1, pa[(t + 1) & 1]);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const f32x4 a = pa[t & 1][h];
            const int row = row0 + 16 * i + 8 * h + pr_e;
            const u32x4 vb = {__float_as_uint(fmaf(a[0], sq[0], bq[0])), __float_as_uint(fmaf(a[1], sq[1], bq[1])),
                              __float_as_uint(fmaf(a[2], sq[2], bq[2])), __float_as_uint(fmaf(a[3], sq[3], bq[3]))};
            __builtin_amdgcn_raw_buffer_store_b128(vb, rc, (cok && row < p.M) ? (unsigned)(((long)(row - m0) * p.N + c) * 4) : OOB, 0, SG_STORE_AUX);
        }
        if (t + 2 < T) patch_put(t + 2);         // (its patch was read for slot t: the LDS queue is in order)
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
template <int NT, bool RC>
__global__ __launch_bounds__(512, 2) void k_sgemm(SGemmP p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int NB = 4 * NT;                               // 16-channel blocks of a wave (wave tile = 32 rows x 64 NT channels: all columns of the tile)
    constexpr int G = NB / 2;                                // ... taken in pairs (even: pair g's weight fragments live in register set g & 1)
    constexpr int RA = 3, RB = NT == 2 ? 3 : 2;              // ring depths (stages)
    constexpr int ABY = SG_ROWS * 128;                       // bytes per activation stage (raw fp32 rows)
    constexpr int BPL = 64 * NT * 64, BBY = 2 * BPL;         // weight stage: hi | lo planes of 64 NT rows x 64 B
    constexpr int BOFF = RA * ABY;
    static_assert(RA * ABY + RB * BBY <= 160 * 1024, "LDS of a gfx950 CU");
    __shared__ __attribute__((aligned(16))) char lds[RA * ABY + RB * BBY];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
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
        ra2 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A2 + (long)m0_ * p.lda2), 0, (int)(rec2 < 0x7FFFFFF0L ? rec2 : 0x7FFFFFF0L), 0x00020000);
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
    // (ring slots are run-time scalars: a fragment address is this per-lane base + slot * stage bytes, one v_add per base and step)
    const int axb = wave * 32 * 128 + offa_l, bwb = BOFF + offb_l;

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
        const bool rc_on = RC && n0 == 0;
        prefetched = false;

        f32x4 acc[NB][2];
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        h16x8 xh[2], xl[2];                                  // [row block]: this step's fragments
        unsigned nhu[2][4], nlu[2][4];                       // the next step's, chunk by chunk (parked: xh / xl are in use until the step's last MFMAs)
        h16x8 wh[2][2], wl[2][2];                            // [register set][block of the pair]
        f32x4 raw[2];                                        // (one row block at a time)
        int ax0 = 0, ax1 = 0, bw_cur = 0, bw_nxt = 0;       // fragment address registers of the step (set at its start)
        auto rd_raw = [&](int i) {                           // row block i of the NEXT stage's raw rows (item start: of stage 0)
            raw[0] = *(const f32x4*)(ldsb + ax0 + i * 2048);
            raw[1] = *(const f32x4*)(ldsb + ax1 + i * 2048);
        };
        auto chunk = [&](int c) {                            // chunk c & 3 of the raw fragment in registers -> packed hi / lo halves of row block c >> 2
            const int i = c >> 2, j = c & 3;
            const f32x4 r = raw[j >> 1];
            if (SG_DIAG & 2) { nhu[i][j] = __float_as_uint(r[2 * (j & 1)]); nlu[i][j] = __float_as_uint(r[2 * (j & 1) + 1]); }
            else sg_split2(r[2 * (j & 1)], r[2 * (j & 1) + 1], p.a_scale, nhu[i][j], nlu[i][j], rc_on, n_sat, n_sub);
        };
        auto adopt = [&]() {                                 // the parked fragments become the current ones
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                xh[i] = __builtin_bit_cast(h16x8, (u32x4){nhu[i][0], nhu[i][1], nhu[i][2], nhu[i][3]});
                xl[i] = __builtin_bit_cast(h16x8, (u32x4){nlu[i][0], nlu[i][1], nlu[i][2], nlu[i][3]});
            }
        };
        auto rd_wq = [&](int base, int g, int set, int q) {  // q-th of the four fragments of pair g: hi a, hi b, lo a, lo b
            const h16x8 v = *(const h16x8*)(ldsb + base + ((q >> 1) * BPL + (2 * g + (q & 1)) * 1024));
            if (q < 2) wh[set][q] = v;
            else wl[set][q - 2] = v;
        };
#define GSB() __builtin_amdgcn_sched_barrier(0)

        // ---- item start: the first stages have landed; behind them in the (in-order) queue are only the previous epilogue's
        // operations, of which the last SG_TAIL_WAIT - stores - may stay in flight
        if (after_epilogue) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(SG_TAIL_WAIT) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        ax0 = axb; ax1 = axb ^ 16; bw_cur = bwb;
        rd_raw(0);
#pragma unroll
        for (int q = 0; q < 4; ++q) rd_wq(bw_cur, 0, 0, q);
#pragma unroll
        for (int c = 0; c < 4; ++c) chunk(c);
        rd_raw(1);
#pragma unroll
        for (int c = 4; c < 8; ++c) chunk(c);
        adopt();
        int sa = 0, sb = 0;                                  // ring slots of the current stage
        GSB();

        // One K step: 6 NB MFMAs per wave, pair by pair of channel blocks (a, b) x the two row blocks: w_hi x_lo | w_hi x_hi | w_lo x_hi,
        // so an accumulator is touched every fourth MFMA.  Stage s lives in activation slot sa = s % 3 and weight slot sb = s % RB
        // (run-time scalars), its fragments in xh / xl.  `more`: a stage s + 1 exists; `pre`: a stage s + 2 exists and is requested in
        // this step - both compile-time: the item's last two steps are their own straight-line instances behind the loop, so no step
        // carries a run-time condition.  While pair g runs, the weight fragments of pair g + 1 are read into the other register set and
        // (first pair) the DMA pieces go out, one per MFMA pair.
        // The operand split of stage s + 1 (eight nine-operation chunks per lane) runs under the last pair's MFMAs, behind the barrier.
        // SG_SPLIT_SPREAD = 1 (A/B build) spreads it over the step's last three pairs instead, one chunk per three MFMAs - a wave's rows of a
        // stage are written by its OWN DMA pieces, so it needs no barrier for them, only its own counted vmcnt (SPLIT_WAIT: everything
        // queued behind those four pieces stays in flight).  Measured no better (mix 9.66 against 9.54 ms per forward, 128^2 1.97 against
        // 1.92 ms: the earlier wait shortens the prefetch distance of the HBM-streamed operand; profiles/r05e_sgemm_diag.txt): the
        // split's cost (13 % of the launch against a build without it) is not one of placement.
        // Before the LAST pair every read of stage s's weights has been issued: lgkmcnt(0), the counted vmcnt, the barrier that publishes
        // stage s + 1's weights; under the last pair's MFMAs the wave reads the first weight pair of stage s + 1 and (weight ring of two)
        // requests the weight pieces of stage s + 2 into the slot this step has just finished reading.
        auto step = [&](auto more_c, auto pre_c) __attribute__((always_inline)) {
            constexpr bool more = decltype(more_c)::value, pre = decltype(pre_c)::value;
            constexpr int GS = G - 3;                        // first pair of the split (>= 1: the early pieces all go out in pair 0)
            static_assert(GS >= 1 && 4 + (RB == 3 ? NT : 0) <= 6, "schedule of a step");
            constexpr int SPLIT_WAIT = pre ? 8 : NT;         // queue behind stage s + 1's activation pieces: its weight pieces (NT) [+ stage s + 2's pieces]
            static_assert(!pre || NT + 4 + (RB == 3 ? NT : 0) == 8, "SPLIT_WAIT");
            const int sa_r = sa + 1 == RA ? 0 : sa + 1, sa_n = sa_r + 1 == RA ? 0 : sa_r + 1;
            const int sb_r = sb + 1 == RB ? 0 : sb + 1, sb_n = RB == 2 ? sb : (sb_r + 1 == RB ? 0 : sb_r + 1);
            ax0 = axb + sa_r * ABY; ax1 = (axb ^ 16) + sa_r * ABY;
            bw_cur = bwb + sb * BBY; bw_nxt = bwb + sb_r * BBY;
            if (pre) a_setup();
            auto early_piece = [&](int q) {
                if (q < 4) { if (pre) dma_a(sa_n, q); }
                else if (RB == 3 && q < 4 + NT) { if (pre) dma_b(sb_n, q - 4); }
            };
            GSB();
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const int set = g & 1;
                const bool last = g == G - 1;
                const int r = g - GS;                        // pair of the split window (0, 1, 2 = last)
                if (last) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    if (pre) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(RB == 3 ? 4 + NT : 4) : "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    GSB();
                }
                if (SG_SPLIT_SPREAD && more && r == 0) {     // this wave's rows of stage s + 1 have landed
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(SPLIT_WAIT) : "memory");
                    rd_raw(0);
                    GSB();
                }
#pragma unroll
                for (int m = 0; m < 12; ++m) {
                    const int ph = m >> 2, b_ = (m >> 1) & 1, i = m & 1, blk = 2 * g + b_;
                    const h16x8 wv = ph < 2 ? wh[set][b_] : wl[set][b_];
                    const h16x8 xv = ph == 0 ? xl[i] : xh[i];
                    acc[blk][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wv, xv, acc[blk][i], 0, 0, 0);
                    GSB();
                    if (!last) {
                        if ((m & 1) == 0) { early_piece(g * 6 + (m >> 1)); GSB(); }
                        if ((m & 1) == 1 && m < 8) { rd_wq(bw_cur, g + 1, set ^ 1, m >> 1); GSB(); }
                    } else {
                        const int wq_ = (m >= 1 && m <= 3) ? m - 1 : (m == 6 ? 3 : -1);
                        if (more && wq_ >= 0) { rd_wq(bw_nxt, 0, 0, wq_); GSB(); }
                        const int bq_ = m == 2 ? 0 : (m == 7 ? 1 : (m == 8 ? 2 : (m == 10 ? 3 : -1)));
                        if (RB == 2 && pre && bq_ >= 0 && bq_ < NT) { dma_b(sb_n, bq_); GSB(); }
                    }
                    if (more && SG_SPLIT_SPREAD) {
                        // chunks 0..3 = row block 0 (read at the window's start), 4..7 = row block 1 (read once chunk 3 has consumed the registers)
                        const int c = r == 0 ? (m == 3 ? 0 : (m == 6 ? 1 : (m == 9 ? 2 : -1)))
                                    : (r == 1 ? (m == 0 ? 3 : (m == 5 ? 4 : (m == 8 ? 5 : (m == 10 ? 6 : -1)))) : (r == 2 ? (m == 4 ? 7 : -1) : -1));
                        if (c >= 0) { chunk(c); GSB(); }
                        if (r == 1 && m == 2) { rd_raw(1); GSB(); }
                    }
                    if (more && !SG_SPLIT_SPREAD && last) {  // behind the barrier, in two bursts: row block 0 read at m = 0 and split at m = 4, row block 1 at m = 5 / 9
                        if (m == 0 || m == 5) { rd_raw(m == 0 ? 0 : 1); GSB(); }
                        if (m == 4 || m == 9) {
#pragma unroll
                            for (int c = 0; c < 4; ++c) chunk((m == 4 ? 0 : 4) + c);
                            GSB();
                        }
                    }
                }
            }
            if (more) adopt();
            sa = sa_r;
            sb = sb_r;
        };
        using BF = std::integral_constant<bool, false>;
        using BT = std::integral_constant<bool, true>;
        // every step but the item's last two requests a stage two ahead; those two are straight-line instances of their own behind the loop
#pragma nounroll
        for (int s_ = 2; s_ < nk; ++s_) step(BT{}, BT{});
        step(BT{}, BF{});
        step(BF{}, BF{});
        GSB();                                               // (nothing of the epilogue is hoisted into the last step: its registers are full)
#undef GSB
        // Column scales and biases of this tile: loaded, waited for and scaled BEFORE the next item's first stages are requested (one L2-hit
        // latency per tile).  Behind the - conditional - prologue issue the compiler would wait vmcnt(0) at their first use, i.e. for the
        // DMA it has just queued (vmcnt retires in order); the empty asm pins the finished values in front of the issue.
        float4 sqv, bqv;
        {
            // (buffer loads that always issue: columns >= N of the bias - or all of them when there is none - read zeros)
            const auto rbi = __builtin_amdgcn_make_buffer_rsrc((void*)p.bias, 0, p.bias ? p.N * 4 : 0, 0x00020000);
            const int c = n0 + 4 * lane;                             // (NT = 2: the upper half of the wave holds columns the tile does not have)
            sqv = *(const float4*)(p.Cscale + (c & 255) + (n0 & ~255));      // (Cscale is padded to a multiple of 256 columns)
            const u32x4 b_ = __builtin_amdgcn_raw_buffer_load_b128(rbi, (unsigned)c * 4u, 0, 0);
            bqv = make_float4(__uint_as_float(b_[0]), __uint_as_float(b_[1]), __uint_as_float(b_[2]), __uint_as_float(b_[3]));
            sqv.x *= p.out_scale; sqv.y *= p.out_scale; sqv.z *= p.out_scale; sqv.w *= p.out_scale;
            asm volatile("" : "+v"(sqv.x), "+v"(sqv.y), "+v"(sqv.z), "+v"(sqv.w), "+v"(bqv.x), "+v"(bqv.y), "+v"(bqv.z), "+v"(bqv.w)::"memory");
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
            float t_ = sqv.x + bqv.x;
#pragma unroll
            for (int j = 0; j < NB; ++j)
#pragma unroll
                for (int i = 0; i < 2; ++i) t_ += acc[j][i][0] + acc[j][i][1] + acc[j][i][2] + acc[j][i][3];
            if (t_ == 1.2345e-30f) p.C[0] = t_;
        } else {
            sgemm_epilogue<NT>(p, acc, sqv, bqv, lds + 2 * ABY + wave * 4096, wave, lane, m0, n0);
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
    DuaKTimer kt((void*)st);
    if (p.range_ctr) {
        if (nt == 4) hipLaunchKernelGGL((k_sgemm<4, true>), dim3(grid), 512, 0, st, p);
        else hipLaunchKernelGGL((k_sgemm<2, true>), dim3(grid), 512, 0, st, p);
    } else {
        if (nt == 4) hipLaunchKernelGGL((k_sgemm<4, false>), dim3(grid), 512, 0, st, p);
        else hipLaunchKernelGGL((k_sgemm<2, false>), dim3(grid), 512, 0, st, p);
    }
    {   // algorithmic: 2 M N K FLOPs; bytes = the raw fp32 operand once + the packed weights + the fp32 output
        const double K = (double)(p.K1 + p.K2), M = (double)p.M, N = (double)p.N;
        kt.done("k_sgemm", 4.0 * M * K + 4.0 * N * K + 4.0 * M * N, 2.0 * M * N * K);
    }
    return dua_check_launch(what);
}

// (f16x3 range check, common.h: this translation unit keeps the counter pointer on the host and hands it to the kernel's checking instantiation)
int dua_range_set_sgemm(unsigned long long* p) { sg_range_host = p; return 0; }
