/*
 * dua_hip.h - C ABI of the MI355X (gfx950) hot-path library libdua_hip.so.
 *
 * The reference (Michedev/diffusion-uncertainty) is 100% Python: it has no FFI
 * layer, its "kernels" are stock ATen ops.  This header is therefore the
 * boundary a maintainer would bind with ctypes: one entry
 * point per fused operation of the uncertainty-guided sampling hot path, each
 * citing the reference lines it replaces (paths relative to
 * diffusion_uncertainty/ in the reference tree).
 *
 * Conventions
 *  - All pointers are DEVICE pointers (HBM) unless the name ends in _host.
 *  - fp32 everywhere; activations inside the denoiser are NHWC
 *    ([N][H][W][C], C fastest), NCHW at the sampler edge.
 *  - `stream` is a hipStream_t passed as void* (NULL = default stream).
 *    Calls are asynchronous; nothing allocates, frees or synchronises.
 *  - Return value: 0 on success, a negative DUA_E* code otherwise
 *    (dua_last_error() returns a static description).  Caller owns all memory.
 */
#ifndef DUA_HIP_H
#define DUA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DUA_OK 0
#define DUA_EINVAL (-1) /* bad shape / argument */
#define DUA_EHIP (-2)   /* HIP runtime error at launch */

int dua_version(void);
const char* dua_last_error(void);

/* ---------------- sampler (scheduler) kernels: HBM-bound ---------------- */

/* DDIM update. schedulers_uncertainty/scheduling_ddim_uncertainty_zigzag_centered.py:473
 * (x0), :490-493 (clip), :507-510 (prev), :523-525 (eta>0: + sigma*noise).
 * noise may be NULL when sigma == 0. */
int dua_ddim_step(const float* eps, const float* x, const float* noise, float* prev, float* x0, size_t n,
                  float sqrt_beta_prod_t, float sqrt_alpha_prod_t, float sqrt_alpha_prod_prev, float dir_coef,
                  float sigma, int clip, float clip_range, void* stream);

/* out = a*src + b*noise.  Re-noising of x_{t-1} (zigzag_centered.py:538, a=sqrt(1-beta_t), b=sqrt(beta_t)),
 * DDPM add_noise (:575-604) and the SD guidance re-noise (uncertainty_guidance.py:88). */
int dua_axpby(const float* src, const float* noise, float* out, size_t n, float a, float b, void* stream);

/* DPM-Solver++ data prediction from an epsilon model output,
 * schedulers_uncertainty/scheduling_dpm_2_uncertainty_centered.py:523-528 (convert_model_output):
 * x0 = (x - sigma_t*eps) / alpha_t.  Also the (quirky) pred_original_sample of its UC window, :946. */
int dua_x0_from_eps(const float* eps, const float* x, float* x0, size_t n, float sigma_t, float alpha_t, void* stream);

/* Second-order multistep DPM-Solver++ update, midpoint form, scheduling_dpm_2_uncertainty_centered.py:709-727:
 * D1 = rinv*(m0 - m1); out = (c1*sample - c2*m0) - c3*D1 with c1 = sigma_t/sigma_s0, c2 = alpha_t*(exp(-h)-1),
 * c3 = 0.5*c2, rinv = 1/r0 (all evaluated by the caller as fp32 scalars, like the reference's 0-dim tensors).
 * The first-order update (:624-632) is dua_axpby(sample, m0, c1, -c2). */
int dua_dpmpp_2m_update(const float* sample, const float* m0, const float* m1, float* out, size_t n, float c1, float c2,
                        float c3, float rinv, void* stream);

/* Streaming centred second moment, zigzag_centered.py:549 / uncertainty_centered.py:540:
 * acc = (m==0 ? 0 : acc) + (eps_m - eps)^2 ; when m == M-1 the sum is divided by M. */
int dua_mc_centered_accum(const float* eps_m, const float* eps, float* acc, size_t n, int m, int M, void* stream);

/* Same estimator from a materialised stack [M][n]. */
int dua_mc_reduce_centered(const float* stack, const float* eps, float* u, size_t n, int M, void* stream);

/* Unbiased variance over the leading axis of stack [M][n] (torch.var, scheduling_ddim_mc_dropout.py:506,
 * uncertainty_guidance.py:106). */
int dua_mc_reduce_var(const float* stack, float* u, size_t n, int M, void* stream);

/* Per-sample quantile threshold, uncertainty_guidance.py:112: q[b] = torch.quantile(u[b].flatten(), p)
 * ('linear', fp32 rank arithmetic) and mask = u > q (strict), bit-exact.  The caller passes the rank
 * split computed in fp32 on the host: lo = floor(p32*(n-1)), hi = ceil(..), w = rank - lo.
 * u: [B][n]; mask: [B][n] uint8 (0/1); q: [B]. */
int dua_quantile_mask(const float* u, int B, int n, int lo, int hi, float w, uint8_t* mask, float* q, void* stream);
/* A/B switch (returns the previous value): 0 = default: samples of up to 65536 elements are held in registers (u read once) - by one
 * workgroup (min/max-ranged LDS histogram + wavefront scans), or, for 16384 < n <= 65536 and 16-byte aligned rows, by two workgroups
 * that bracket the target from a common sample and meet once through a per-stream workspace; 1 = always the streaming multi-pass kernel
 * (six passes over u); 2 = never the two-workgroup kernel; 3 = the two-workgroup kernel with its bracket declared missed (its fallback);
 * 4 = ... with every workgroup acting as if its partner had never arrived (the way out of its bounded wait). */
int dua_set_quantile_stream(int on);
/* The two-workgroup kernel keeps one small counter workspace per (device, stream), zeroed when allocated and left zeroed by every launch
 * that completes (a rejected launch marks it for re-zeroing before the next use).  Releases the current device's workspace of `stream`
 * (call before destroying a stream: a recycled handle must not inherit it) or, with all != 0, of every stream; returns the number of
 * workspaces freed, or a negative DUA_E* code. */
int dua_quantile_workspace_release(void* stream, int all);

/* y[p][h][w] = x[p][H-1-h][w] over `planes` = N*C image planes: torch.flip(x, dims=[2]) of an NCHW tensor, the input / output
 * mirroring of the `flip` estimator (schedulers_uncertainty/scheduling_ddim_flip.py:486,491).  Out of place. */
int dua_flip_h(const float* x, float* y, size_t planes, int H, int W, void* stream);

/* Posterior blend, uncertainty_guidance.py:115-120.  eps,u: [B][n]; mask: [B][n] uint8. */
int dua_posterior_blend(const float* eps, const float* u, const uint8_t* mask, float* out, int B, int n, int M,
                        float alpha_hat_t, void* stream);

/* Class-conditional posterior guidance
 * (pipeline_uncertainty/pipeline_sampler_class_conditional_uncertainty_guided_posterior_distribution.py).
 * dua_threshold_map (:10-30): map = (lower ? u < thr : u > thr) as 0.0/1.0 floats; u, map [B][n];
 *   per_sample = 1: thr[B] (the per-sample quantile of the float-threshold branch), 0: thr[n] broadcast over the batch
 *   (threshold[i] of the tensor branch, converted to fp32 by the caller).
 * dua_batch_sum: out[n] = sum_b x[b][n] (pred_epsilon.sum(dim=0) at :65 reduces over the BATCH axis; reproduced as is).
 * dua_posterior_score (:62-65): inv = 1/u; out = 1/(M*inv + inv_alpha) * (inv * bsum), bsum [n] broadcast over B.
 * dua_map_blend (:150): out = a*map + b*(1 - map). */
int dua_threshold_map(const float* u, const float* thr, float* map, int B, size_t n, int per_sample, int lower, void* stream);
int dua_batch_sum(const float* x, float* out, int B, size_t n, void* stream);
int dua_posterior_score(const float* u, const float* bsum, float* out, int B, size_t n, int M, float inv_alpha, void* stream);
int dua_map_blend(const float* a, const float* b, const float* map, float* out, size_t n, void* stream);

/* ((x/2+.5).clamp(0,1)*255).round() -> uint8, generate_samples.py:203-215. */
int dua_to_uint8(const float* x, uint8_t* out, size_t n, void* stream);

/* Dataset-level pixel-wise threshold, scripts/compute_threshold_pixel_wise.py:143-165: for each of the SP = S*3*H*W
 * (timestep, pixel) columns, the k-th smallest (0-based, k = int(N*perc)) over the N samples of half(u[n][...]), as
 * selected there by argsort(dim=0)[k] + gather.  u: [N][SP] fp32; out: [SP] fp16 bit patterns. Exact. */
int dua_pixelwise_order_statistic(const float* u, int N, long SP, int k, uint16_t* out, void* stream);

/* Device normal generator (Philox4x32-10 + Box-Muller); replaces torch.randn_like
 * (zigzag_centered.py:500,532) when the caller does not need the CPU generator's stream. */
int dua_randn(float* out, size_t n, uint64_t seed, uint64_t offset, void* stream);

/* Bernoulli keep-mask scaled by 1/(1-p) in place of nn.Dropout's mask (unet_openai.py:210). out: float [n]. */
int dua_dropout_mask(float* out, size_t n, float p, uint64_t seed, uint64_t offset, void* stream);

/* ---------------- layout ---------------- */
/* NCHW [N][C][H][W] -> NHWC [N][H][W][Cpad] (channels >= C zero-filled). */
int dua_nchw_to_nhwc(const float* x, float* y, int N, int C, int H, int W, int Cpad, void* stream);
/* NHWC [N][H][W][C] -> NCHW [N][Ckeep][H][W] keeping the first Ckeep channels (traits.py:18 `[:, :3]`). */
int dua_nhwc_to_nchw(const float* x, float* y, int N, int C, int H, int W, int Ckeep, void* stream);

/* ---------------- denoiser kernels (ADM UNet / U-ViT) ---------------- */

/* GroupNorm statistics over the virtual channel-concat [x1 | x2] (x2 may be NULL, C2 = 0):
 * guided_diffusion/nn.py:17-19 (GroupNorm32, 32 groups, eps 1e-5) on the concatenated skip input of
 * unet_openai.py:662.  stats: [N][G][2] = (mean, rstd).  ws: >= dua_groupnorm_ws_floats(N, HW, C1+C2, G) floats. */
size_t dua_groupnorm_ws_floats(int N, int HW, int C, int G);
int dua_groupnorm_stats(const float* x1, int C1, const float* x2, int C2, int N, int HW, int G, float eps,
                        float* ws, float* stats, void* stream);

/* y = act(GN(x)*gamma+beta [*(1+scale)+shift]) [*dropmask], optionally followed by 2x2 average pooling
 * (resample=-1) or nearest x2 upsampling (resample=+1): unet_openai.py:183-187 (in_layers), :207-214 and
 * :249-253 (FiLM out_layers), :191-196 (h_upd).  film: [N][2*C] (scale | shift) rows film_ld floats apart (0 = dense 2*C; a column block of the
 * one-GEMM FiLM table of all ResBlocks, :245 for every block at once) or NULL. act: 1 = SiLU.
 * dropmask: [N][H][W][C] float keep/(1-p) or NULL.  y: NHWC at the resampled size. */
int dua_groupnorm_apply(const float* x1, int C1, const float* x2, int C2, int N, int H, int W, int G,
                        const float* stats, const float* gamma, const float* beta, const float* film, long film_ld, int act,
                        int resample, const float* dropmask, float* y, void* stream);

/* x_upd of an up/down ResBlock (unet_openai.py:191-196,243): resample=-1 avg-pool 2x2, +1 nearest x2. */
int dua_resample2x(const float* x, float* y, int N, int H, int W, int C, int resample, void* stream);

/* ---- latent UNet of Stable Diffusion 1.5 (diffusers==0.31.0 UNet2DConditionModel, un-vendored dependency of
 * pipeline_uncertainty/pipeline_stable_diffusion_uncertainty_guided.py:747-753; parity unpinned) ----
 * x[n][p][c] += bias[n][c] in place (ResnetBlock2D: hidden_states + time_emb_proj(silu(emb))[:, :, None, None]);
 * bias rows are bias_ld floats apart. */
int dua_add_channel_bias(float* x, const float* bias, long bias_ld, int N, int HW, int C, void* stream);
/* GEGLU feed-forward gate: y[m][f] = x[m][f] * gelu(x[m][F + f]) for x [M][2F] (exact erf GELU). */
int dua_geglu(const float* x, float* y, long M, int F, void* stream);
/* ... written as f16x3 operand planes ([M][F] fp16 hi | lo of y * scale, scale a power of two) for dua_gemm_f16x3_presplit. */
int dua_geglu_split_f16x3(const float* x, void* planes, long M, int F, float scale, void* stream);
/* y[n][i][j][c] = x[n][2i][2j][c]  (Downsample2D's stride-2 / pad-1 conv = the stride-1 conv at the even pixels). */
int dua_pick_even(const float* x, float* y, int N, int H, int W, int C, void* stream);

/* 3x3 / stride 1 / pad 1 convolution as an implicit GEMM on fp32 MFMA (unet_openai.py:186,212,483,616).
 * x: NHWC [N][H][W][Cin]; wp: weights packed by dua_pack_conv3x3_weight ([CoutPad][9][Cin], Cin%32==0, or the
 * small-Cin form [CoutPad][16][4] when Cin == 4); bias [Cout]; residual NHWC [N][H][W][Cout] or NULL (the
 * `skip_connection(x) + h` of :257).  nchw_out != 0 writes y as NCHW [N][Cout][H][W]. */
int dua_conv3x3(const float* x, const float* wp, const float* bias, const float* residual, float* y, int N, int H,
                int W, int Cin, int Cout, int nchw_out, float* colsum, void* stream);
/* `colsum` (here and in the other conv / GEMM entry points; may be NULL): [ceil(M/64)][2][Cout] floats that receive, per
 * block of 64 output rows (pixels), the column sums and sums of squares of the values written to y - the GroupNorm
 * statistics of the NEXT layer, produced in the epilogue so that layer does not re-read y.  Needs an NHWC / row-major
 * output with Cout % 4 == 0; consumed by dua_groupnorm_stats_from_colsums when H*W % 64 == 0. */
int dua_groupnorm_stats_from_colsums(const float* ws1, int C1, const float* ws2, int C2, int N, int HW, int G, float eps,
                                     float* stats, void* stream);
/* ... of x + t for a per-(sample, channel) bias t [N][tbias_ld] that is added in front of the norm (diffusers ResnetBlock2D :
 * hidden + time_emb_proj(silu(emb))[:, :, None, None] before norm2): folded into the block sums, x + t is never materialised. */
int dua_groupnorm_stats_from_colsums_bias(const float* ws, int C, int N, int HW, int G, float eps, const float* tbias, long tbias_ld,
                                          float* stats, void* stream);
/* OIHW [Cout][CinReal][3][3] -> packed form above (CinPad is 4 for CinReal<=4, else CinReal). */
size_t dua_conv3x3_packed_floats(int Cout, int Cin);
int dua_pack_conv3x3_weight(const float* w_oihw, float* wp, int Cout, int CinReal, void* stream);

/* Batched GEMM  C[b] = alpha * A[b] * op(B[b]) (+ bias) (+ residual), fp32 MFMA.
 * A is the virtual K-concat [A1 | A2] (row-major, leading dims lda1/lda2; A2 may be NULL with K2 = 0):
 * the 1x1 skip conv on the concatenated input (unet_openai.py:223,662), qkv / proj_out (:287,295),
 * the time / label MLPs (:472-476, :200-206) and the attention products (:350-354).
 * transB = 1: B is [N][K] (weights, K^T);  transB = 0: B is [K][N].
 * act: 0 none, 1 SiLU on the INPUT A (emb_layers' leading SiLU), 2 GELU on the output (timm.py:105).
 * Strides are in floats; batch strides may be 0 (shared operand). */
typedef struct {
    const float* A1; const float* A2; const float* B; const float* bias; const float* residual; float* C;
    int M, N, K1, K2;
    long lda1, lda2, ldb, ldc, ldr;
    long sA1, sA2, sB, sC, sR;          /* batch strides */
    int batch, batch_inner;              /* batch index b -> (b / batch_inner, b % batch_inner); see below */
    long sA1i, sBi, sCi;                 /* inner batch strides (heads inside one image) */
    int transB, act_in, act_out;
    float alpha;
    float* colsum;                       /* optional column sums of C (see dua_conv3x3); batch must be 1 */
} dua_gemm_args;
int dua_gemm(const dua_gemm_args* args_host, void* stream);

/* Row softmax in place over the first `cols` (<= 4096) entries of each row of S (row stride ld floats), fp32
 * (unet_openai.py:353 / :388; uvit/uvit.py:73). */
int dua_softmax_rows(float* S, long rows, int cols, long ld, void* stream);

/* Fused multi-head attention, exact fp32 on the matrix cores, logits never materialised:
 *   out[b][t][h*ch + c] = sum_s softmax_s(scale * q[b,t,h,:].k[b,s,h,:]) v[b,s,h,c]
 * qkv rows [B*T][3*heads*ch] = (q | k | v), head-major inside each third; out [B*T][heads*ch].  Any T >= 1;
 * ch a multiple of 8, <= 256; B*heads <= 65535.  Replaces the body of QKVAttentionLegacy.forward / QKVAttention.forward
 * (guided_diffusion/unet_openai.py:338-355, :371-390; scale = ch^-1/2 = the two ch^-1/4 operand scales) and
 * F.scaled_dot_product_attention in uvit/uvit.py:69-73. */
int dua_attention(const float* qkv, float* out, int B, int T, int heads, int ch, float scale, void* stream);
/* Same contract on the bf16 matrix cores with split operands (hi*hi + hi*lo + lo*hi, ~2^-17 per product, both q.k^T and P.v):
 * the attention of the split precision modes.  ch a multiple of 4, <= 128 (the queries live in registers). */
int dua_attention_bf16x3(const float* qkv, float* out, int B, int T, int heads, int ch, float scale, void* stream);

/* LayerNorm over the last axis (U-ViT, uvit/uvit.py:108-110,225): y = (x-mean)*rstd*gamma+beta, eps 1e-5. */
int dua_layernorm(const float* x, const float* gamma, const float* beta, float* y, long rows, int D, float eps,
                  void* stream);

/* emb[n][:] = sinusoidal timestep embedding (guided_diffusion/nn.py:103-121), t int64 [N]. */
int dua_timestep_embedding(const int64_t* t, float* emb, int N, int dim, void* stream);

/* out[n][:] = a[n][:] + table[idx[n]][:]  (label embedding add, unet_openai.py:654). a may be NULL. */
int dua_embedding_add(const float* a, const float* table, const int64_t* idx, float* out, int N, int D, void* stream);

/* ---------------- split-bf16 ("bf16x3") variants of the two FLOP carriers ----------------
 * Same contracts as dua_conv3x3 / dua_gemm (transB = 1 form), evaluated on the bf16 matrix cores with every fp32
 * operand split a = hi + lo (bf16 each) and hi*hi + hi*lo + lo*hi accumulated in fp32 (per-product relative error
 * about 2^-17; csrc/mfma_bf16x3.hip).  Weights are split once: `planes` holds two bf16 planes [Npad128][K]
 * (hi, then lo), dua_bf16x3_plane_elems(N, K) elements each.  K must be a multiple of 32. */
size_t dua_bf16x3_plane_elems(int N, int K);
/* w: [N][K] fp32, or OIHW [N][conv_cin][3][3] when conv_cin != 0 (then K = 9*conv_cin, packed tap-major). */
int dua_pack_weight_bf16x3(const float* w, void* planes, int N, int K, int conv_cin, void* stream);
int dua_conv3x3_bf16x3(const float* x, const void* planes, const float* bias, const float* residual, float* y, int N,
                       int H, int W, int Cin, int Cout, int nchw_out, float* colsum, void* stream);
/* Variant whose input was already split by dua_groupnorm_apply_split: xplanes = two bf16 NHWC planes (hi | lo). */
int dua_conv3x3_bf16x3_presplit(const void* xplanes, const void* planes, const float* bias, const float* residual,
                                float* y, int N, int H, int W, int Cin, int Cout, int nchw_out, float* colsum, void* stream);
/* dua_groupnorm_apply writing its result as two bf16 planes (hi = bf16(v), lo = bf16(v - hi)), N*Hy*Wy*C elements
 * each (Hy, Wy = resampled size): the same bytes as the fp32 tensor, already in the bf16x3 conv's operand format. */
int dua_groupnorm_apply_split(const float* x1, int C1, const float* x2, int C2, int N, int H, int W, int G,
                              const float* stats, const float* gamma, const float* beta, const float* film, long film_ld, int act,
                              int resample, const float* dropmask, void* yplanes, void* stream);
/* C[M][N] = A W^T (+bias) (+residual) with A as split bf16 planes [M][K] (hi | lo) from dua_groupnorm_apply_split and W
 * from dua_pack_weight_bf16x3: the LDS-DMA kernel with one tap (qkv projection after the attention GroupNorm). */
int dua_gemm_bf16x3_presplit(const void* aplanes, const void* planes, const float* bias, const float* residual, float* C,
                             void* Csplit, int M, int N, int K, int act_out, float* colsum, void* stream);
/* Csplit != NULL: the result is written as split bf16 planes ([M][N] hi | lo) instead of fp32 C - the operand format of
 * the next presplit GEMM (U-ViT fc1 -> GELU -> fc2).  act_out: 0 none, 2 GELU. */
/* LayerNorm writing split bf16 planes ([rows][D] hi | lo): operand of the presplit GEMM (uvit/uvit.py:108-110). */
int dua_layernorm_split(const float* x, const float* gamma, const float* beta, void* yplanes, long rows, int D, float eps,
                        void* stream);
/* ---- "f16x3" operand format: the fp32-EQUIVALENT split (22 significant operand bits) ---------------------------------
 * Same role and call shapes as the bf16x3 entry points (conv / 1x1 / linear products of
 * guided_diffusion/unet_openai.py:186,212,287,483,616): every fp32 operand is scaled by a power of two into fp16's normal
 * range and split s = hi + lo with hi = fp16(s), lo = fp16(s - hi) (11 + 11 significant bits); the product is
 * hi*hi + hi*lo + lo*hi on three v_mfma_f32_32x32x16_f16 with fp32 accumulation (the dropped lo*lo term is 2^-22 relative),
 * and the epilogue multiplies the accumulator by 2^-(weight row exponent) / act_scale before bias / residual
 * (csrc/mfma_bf16x3.hip, FMT 3).  Activation planes ([rows][C], 2 bytes per element): plane 0 fp16(x * act_scale)
 * (saturating at +-65504), plane 1 fp16(x * act_scale - plane 0); act_scale is a power of two chosen by the caller (the
 * UNet uses 2^6: GroupNorm+SiLU outputs up to 1023 in magnitude are exact to 22 bits down to 2^-9).  Packed weights
 * (dua_f16x3_weight_bytes): fp16(w * 2^s) and the fp16 remainder, stage-contiguous - [row tile of 128][K step of 32][hi | lo]
 * [128 rows][32]: the 16 KB a K step of a 128-row tile stages form one block in LDS-image order - then round_up(N, 256) floats
 * 2^-s with s per output row such that the row maximum lands in [2^14, 2^15). */
size_t dua_f16x3_weight_bytes(int N, int K);
int dua_pack_weight_f16x3(const float* w, void* packed, int N, int K, int conv_cin, void* stream);
int dua_groupnorm_apply_split_f16x3(const float* x1, int C1, const float* x2, int C2, int N, int H, int W, int G,
                                    const float* stats, const float* gamma, const float* beta, const float* film, long film_ld, int act,
                                    int resample, const float* dropmask, float act_scale, void* yplanes, void* stream);
/* y = act(GroupNorm(x + t)) as f16x3 planes, the bias t [N][tbias_ld] folded into the affine coefficients; stats are those of x + t. */
int dua_groupnorm_apply_prebias_split_f16x3(const float* x, int C, int N, int H, int W, int G, const float* stats, const float* gamma,
                                            const float* beta, const float* tbias, long tbias_ld, int act, float act_scale, void* yplanes,
                                            void* stream);
int dua_conv3x3_f16x3_presplit(const void* xplanes, const void* packed, const float* bias, const float* residual, float* y,
                               int N, int H, int W, int Cin, int Cout, int nchw_out, float* colsum, float act_scale, void* stream);
int dua_gemm_f16x3_presplit(const void* aplanes, const void* packed, const float* bias, const float* residual, float* C,
                            int M, int N, int K, int act_out, float* colsum, float act_scale, void* stream);
/* ---- Winograd F(2,3) along y for the f16x3 3x3 convolutions (csrc/conv_wino_f16x3.hip) ------------------------------------
 * The same convolution (unet_openai.py:186,212) with 6 instead of 9 matrix-core taps per output pixel at unchanged operand
 * precision: input rows d0..d3 = 2j-1 .. 2j+2 -> V0 = d0-d2, V1 = d1+d2, V2 = d2-d1, V3 = d1-d3; kernel rows g0..g2 ->
 * U0 = g0, U1 = (g0+g1+g2)/2, U2 = (g0-g1+g2)/2, U3 = g2; M_f = conv1x3(V_f, U_f); out row 2j = M0+M1+M2, row 2j+1 = M1-M2-M3.
 * dua_groupnorm_apply_wino_f16x3: dua_groupnorm_apply_split_f16x3 writing the four V_f images as f16x3 planes
 *   [f][tile of 128 positions of N*(Ho/2)*Wo][C/32][hi | lo][128][32] of V * v_scale - 16 KB contiguous per (tile, 32-channel
 *   block), the conv kernel's activation stage - (dua_f16x3_wino_act_bytes bytes; Ho, Wo = the conv's input size after `resample`);
 *   v_scale a power of two (|V| <= 2 max|activation|: the UNet uses act_scale / 2).
 * dua_pack_weight_f16x3_wino: OIHW fp32 weights -> U_f * 2^s as fp16 hi | lo images in the kernel's stage order
 *   ([column tile of 128][K step = (channel block, fi, dx)][hi | lo][group * 128 + channel][32]: 32 KB contiguous per step), then
 *   round_up(Cout, 256) floats 2^-s (s per output channel over all 12 (f, dx) taps); dua_f16x3_wino_weight_bytes bytes.
 * dua_conv3x3_f16x3_wino: y[N][H][W][Cout] = conv3x3(x) + bias (+ residual) from those two operands; colsum as dua_conv3x3 except
 *   that a 64-pixel slot holds the sums of every other row of a 128-pixel block (the per-image totals, which is what
 *   dua_groupnorm_stats_from_colsums folds, are exact).  Shapes: dua_conv3x3_f16x3_wino_supported returns 1 for H even, W a power
 *   of two <= 128, Cin % 32 == 0, Cout % 4 == 0, H*W % 128 == 0, and 2 for the same with a smaller map whose H*W is still a multiple
 *   of 32 (the 8 x 8 level of ADM-64 / 128: a wave's 64 positions then span several images, so colsum must be NULL);
 *   0 = not taken, the conv goes through dua_conv3x3_f16x3_presplit.  res_resample: 0 = residual is [N][H][W][Cout]; +1 = residual is
 *   [N][H/2][W/2][Cout] and is added through nearest-neighbour 2x upsampling, -1 = residual is [N][2H][2W][Cout] and is added through
 *   2x2 average pooling - the resampled identity skip of an ADM up / down block (unet_openai.py:238-243) read in place, bit-identical
 *   to dua_resample2x followed by res_resample = 0. */
/* 3x3 convolution with a handful of output channels (the UNet's `out` conv, unet_openai.py:616: 256 -> 6, of which traits.py:18 keeps 3)
 * as a one-tap GEMM plus a gather: T[M][ldt] = x W_t^T with W_t row (tap * cout + co) = w[co][:][tap] (any GEMM entry point of this
 * header, N = 9 * cout padded to its granularity), then out[n][co][y][x] = bias[co] + sum_tap T[pixel + offset(tap)][tap * cout + co]
 * (zero outside the image), NCHW.  The implicit-GEMM kernels would spend a 128-column tile on it. */
int dua_conv3x3_tap_gather(const float* T, int ldt, const float* bias, float* out_nchw, int N, int H, int W, int cout, void* stream);
size_t dua_f16x3_wino_act_bytes(int N, int Ho, int Wo, int C);
size_t dua_f16x3_wino_weight_bytes(int Cout, int Cin);
int dua_pack_weight_f16x3_wino(const float* w_oihw, void* packed, int Cout, int Cin, void* stream);
int dua_groupnorm_apply_wino_f16x3(const float* x1, int C1, const float* x2, int C2, int N, int H, int W, int G,
                                   const float* stats, const float* gamma, const float* beta, const float* film, long film_ld, int act,
                                   int resample, const float* dropmask, float v_scale, void* vplanes, void* stream);
int dua_conv3x3_f16x3_wino_supported(int N, int H, int W, int Cin, int Cout);
int dua_conv3x3_f16x3_wino(const void* vplanes, const void* packed, const float* bias, const float* residual, int res_resample, float* y,
                           int N, int H, int W, int Cin, int Cout, float* colsum, float v_scale, void* stream);
/* Mode "f16w": the Winograd form with one fp16 product per tap.  The writer stores only the hi images of the stage blocks (4 bytes
 * written per input element instead of 8), the conv fetches and multiplies only the hi images of the planes and of the packed weights
 * (same buffers, same packing: dua_f16x3_wino_act_bytes / dua_pack_weight_f16x3_wino). */
int dua_groupnorm_apply_wino_f16w(const float* x1, int C1, const float* x2, int C2, int N, int H, int W, int G,
                                  const float* stats, const float* gamma, const float* beta, const float* film, long film_ld, int act,
                                  int resample, const float* dropmask, float v_scale, void* vplanes, void* stream);
int dua_conv3x3_f16w_wino(const void* vplanes, const void* packed, const float* bias, const float* residual, int res_resample, float* y,
                          int N, int H, int W, int Cin, int Cout, float* colsum, float v_scale, void* stream);
/* Winograd F(4,3) along y for the f16x3 3x3 convolutions (csrc/conv_wino4_f16x3.hip): four output rows from six transformed rows -
 * 4.5 instead of 6 matrix-core taps per output pixel and 1.5 x instead of 2 x the input as transformed images, at the same 22-bit
 * operands (the transform's constants amplify rounding ~2.5 x against the direct fp32 convolution; replaces `self.conv2 = conv_nd(...)`
 * style 3x3 convolutions of unet_openai.py:183-187,207-214 exactly like dua_conv3x3_f16x3_wino).  Same operand layouts with six
 * frequency images of N*(Ho/4)*Wo positions and 16 KB weight stages; v_scale = the plane scale of the V images (a power of two; V reaches
 * 10 x the input).  dua_conv3x3_f16x3_wino4_supported: 1 for H %% 4 == 0, W a power of two <= 128, Cin %% 32 == 0, Cout %% 4 == 0,
 * H*W %% 64 == 0; 2 = without the column-sum output (H*W %% 256 != 0). */
size_t dua_f16x3_wino4_act_bytes(int N, int Ho, int Wo, int C);
size_t dua_f16x3_wino4_weight_bytes(int Cout, int Cin);
int dua_pack_weight_f16x3_wino4(const float* w_oihw, void* packed, int Cout, int Cin, void* stream);
int dua_groupnorm_apply_wino4_f16x3(const float* x1, int C1, const float* x2, int C2, int N, int H, int W, int G,
                                    const float* stats, const float* gamma, const float* beta, const float* film, long film_ld, int act,
                                    int resample, const float* dropmask, float v_scale, void* vplanes, void* stream);
int dua_conv3x3_f16x3_wino4_supported(int N, int H, int W, int Cin, int Cout);
/* 0 / 2 / 4: the Winograd form (none, F(2,3), F(4,3)) this shape should run in - whole rounds of tiles on 256 persistent workgroups
 * decide (F(4,3) has half as many, each 1.5 x as long). */
int dua_conv3x3_f16x3_wino_pick(int N, int H, int W, int Cin, int Cout);
int dua_conv3x3_f16x3_wino4(const void* vplanes, const void* packed, const float* bias, const float* residual, int res_resample, float* y,
                            int N, int H, int W, int Cin, int Cout, float* colsum, float v_scale, void* stream);
/* The same form in EXACT fp32 (csrc/conv_wino_f32.hip, v_mfma_f32_16x16x4_f32: products and sums in fp32 - the Winograd form of
 * dua_conv3x3): the transformed images and the packed weights hold floats in the same stage-block layout (same byte counts:
 * dua_f16x3_wino_act_bytes / dua_f32_wino_weight_bytes), shapes as dua_conv3x3_f16x3_wino_supported. */
size_t dua_f32_wino_weight_bytes(int Cout, int Cin);
int dua_pack_weight_f32_wino(const float* w_oihw, void* packed, int Cout, int Cin, void* stream);
int dua_groupnorm_apply_wino_f32(const float* x1, int C1, const float* x2, int C2, int N, int H, int W, int G,
                                 const float* stats, const float* gamma, const float* beta, const float* film, long film_ld, int act,
                                 int resample, const float* dropmask, void* vplanes, void* stream);
int dua_conv3x3_f32_wino(const void* vplanes, const void* packed, const float* bias, const float* residual, int res_resample, float* y,
                         int N, int H, int W, int Cin, int Cout, float* colsum, void* stream);
/* The F(4,3)-along-y form with fp32 products and sums (csrc/conv_wino4_f32.hip: k_wconv432 on v_mfma_f32_16x16x4_f32, 4.5 of 9 taps per
 * pixel - the fastest form of `precision = "fp32"`).  Every product is an fp32 product, but the Winograd transforms re-associate the
 * reference's sums (unet_openai.py:186,212) and F(4,3)'s transform matrices amplify rounding: a single layer carries up to ~4 x the direct
 * fp32 engine's error against fp64, and the full-size ADM UNets differ from the direct form by up to 5e-5 (7.9e-6 against the reference's
 * output, bar 1e-3).  `model.winograd = False` runs the direct fp32 kernel - the reference's own summation order - and `2` the milder
 * F(2,3) form; use those when the closest possible fp32 arithmetic matters more than throughput.  The six
 * transformed images hold floats in the stage-block layout of dua_groupnorm_apply_wino4_f16x3 (same bytes: dua_f16x3_wino4_act_bytes),
 * the packed weights [column tile of 64][K step = (channel block, fi, dx)][group * 64 + channel][32] floats; shapes as
 * dua_conv3x3_f16x3_wino4_supported; colsum needs H * W % 256 == 0. */
size_t dua_f32_wino4_weight_bytes(int Cout, int Cin);
int dua_pack_weight_f32_wino4(const float* w_oihw, void* packed, int Cout, int Cin, void* stream);
int dua_groupnorm_apply_wino4_f32(const float* x1, int C1, const float* x2, int C2, int N, int H, int W, int G,
                                  const float* stats, const float* gamma, const float* beta, const float* film, long film_ld, int act,
                                  int resample, const float* dropmask, void* vplanes, void* stream);
int dua_conv3x3_f32_wino4(const void* vplanes, const void* packed, const float* bias, const float* residual, int res_resample, float* y,
                          int N, int H, int W, int Cin, int Cout, float* colsum, void* stream);
/* dua_gemm_f16x3_presplit with the A operand concatenated along K from TWO plane sets (rows [M][K1] and [M][K2], each K % 32 == 0):
 * the 1x1 skip convolution of an ADM output block on torch.cat([h, hs.pop()]) (unet_openai.py:219-223,662) and U-ViT's
 * skip_linear(cat([x, skip])) (uvit/uvit.py:116-117) without materialising the concatenation.  a2planes may be NULL (K2 = 0).
 * extra_planes != NULL: the columns >= extra_from (a multiple of 32) of the fp32 result are ALSO written as f16x3 planes of C * split_scale
 * ([M][N] hi | lo; columns below extra_from stay unwritten there); extra_only != 0: those columns are written as planes ONLY (not to C) -
 * (tried for the k | v thirds of a qkv projection handed to the attention kernel as planes: slower end to end, profiles/LOG.md). */
int dua_gemm_f16x3_presplit2(const void* a1planes, int K1, const void* a2planes, int K2, const void* packed, const float* bias,
                             const float* residual, float* C, void* extra_planes, int extra_from, int extra_only, int M, int N, int act_out,
                             float* colsum, float act_scale, float split_scale, void* stream);
/* A/B switches of the one-tap f16x3 GEMM kernel behind the presplit entry points (csrc/gemm_f16x3.hip; correct kernels only, process-global):
 * nt = 0 picks the tile width per shape (default), 2 / 3 / 4 force 128 / 192 / 256 columns, -1 sends the calls to the previous engine
 * (k_xconv_dma<NT, 3> / k_hconv<NT, false>); bit 0 of streamk = 0 runs whole tiles only even when dua_set_splitk_workspace registered a
 * buffer; bits 1-2 are ignored; bits 3-4 the stream-K extent: 0 = per shape (default), 8 = the last partly filled round of tiles only, 16 = the whole launch (workgroups out of phase). */
int dua_set_hgemm_mode(int nt, int streamk);
/* The caller keeps several launches in flight on HIP streams of their own (the batch in chunks, uvit/uvit.py / unet_openai.py
 * _forward_streams): the tile width is then chosen by work, not by whole rounds of tiles on the 256 CUs - the other stream's workgroups
 * take the CUs a partly filled round leaves idle.  Process-global; returns the previous setting; results do not depend on it. */
int dua_set_hgemm_concurrent(int on);
/* ... the result as f16x3 planes of C * split_scale ([M][N] hi | lo) for the next presplit GEMM; act_out 2 = GELU
 * (U-ViT Mlp: fc1 -> GELU -> fc2, uvit/timm.py:96-111). */
int dua_gemm_f16x3_presplit_planes(const void* aplanes, const void* packed, const float* bias, void* Cplanes, int M, int N, int K,
                                   int act_out, float act_scale, float split_scale, void* stream);
/* LayerNorm writing f16x3 planes of y * act_scale (uvit/uvit.py:108-110 feeding the qkv / fc1 GEMMs); D % 32 == 0. */
int dua_layernorm_split_f16x3(const float* x, const float* gamma, const float* beta, void* yplanes, long rows, int D, float eps,
                              float act_scale, void* stream);
/* dua_gemm_bf16x3 on the f16x3 split: C = alpha * [A1 | A2] W^T (+bias) (+residual) with the fp32 A operand (raw residual-stream
 * activations: 1x1 skip convolution unet_openai.py:223, proj_out :295) multiplied by act_scale and split into fp16 hi + lo at
 * LDS-store time.  hi saturates at +-65504 and lo then carries the remainder, so an out-of-range operand loses precision
 * instead of becoming inf.  W from dua_pack_weight_f16x3. */
int dua_gemm_f16x3(const float* A1, int K1, long lda1, const float* A2, int K2, long lda2, const void* packed,
                   const float* bias, const float* residual, long ldr, float* C, long ldc, int M, int N, int act_in,
                   int act_out, float alpha, float* colsum, float act_scale, void* stream);
/* The plain form of dua_gemm_f16x3 (bias only, no activation / residual / column sums, dense C: the ADM 1x1 skip convolutions over the
 * virtual concat, unet_openai.py:219-223,257,662) runs on the persistent raw-operand kernel csrc/gemm_skip_f16x3.hip (k_sgemm: 256-row
 * tiles, LDS-DMA of the raw fp32 rows, split in registers after the fragment read).  A/B switch, process-global, correct kernels only:
 * nt = 0 picks the tile width per shape (default), 2 / 4 force 128 / 256 columns (same results bit for bit), -1 sends these calls back
 * to the previous engine (k_xtile; different rounding order). */
int dua_set_sgemm_mode(int nt);
/* Fused attention (dua_attention) with both products on the f16x3 split: q, k, P and v scaled by powers of two and split
 * into fp16 hi + lo, hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_f16 (2^-22 per product), softmax in fp32.  Head dims <= 256 (above 128: a
 * 512-register build of the same kernel, one workgroup of 128 queries per CU - ADM-128's 192- and 256-channel heads).
 * out_planes != NULL: the result is written as f16x3 planes of out * planes_scale ([B*T][heads*ch] hi | lo), the operand of
 * dua_gemm_f16x3_presplit (proj_out, unet_openai.py:295); otherwise fp32 `out`. */
int dua_attention_f16x3(const float* qkv, float* out, void* out_planes, float planes_scale, int B, int T, int heads, int ch,
                        float scale, void* stream);
/* ... with keys / values from their own tensors (cross-attention, diffusers Attention with encoder_hidden_states): q rows
 * [B*Tq][ldq], k / v rows [B*Tk][ldkv], head h at column h*ch of each; Tk any (77 text tokens). */
int dua_cross_attention_f16x3(const float* q, long ldq, const float* k, const float* v, long ldkv, float* out, void* out_planes,
                              float planes_scale, int B, int Tq, int Tk, int heads, int ch, float scale, void* stream);
/* A/B engine selector for the presplit conv / GEMM kernels (returns the previous mode, or DUA_EINVAL): 1 = LDS-DMA kernels,
 * 256x256 tile where it fits (default); 2 = LDS-DMA 256x128 only; 0 = register-staged kernel (bf16x3); 6 = the 32x32x16 loop for every f16x3 tile; 7 = the 16x16x32 kernel (conv_f16x3.hip) for every f16x3 tile.  Process-global and unsynchronised: set it before queueing work.  The zero-record-descriptor modes
 * 3 - 5 (timing-only, wrong results) exist only in the diagnostic library built with -DDUA_CONV_TIMING
 * (scripts/conv_tile_timing.py --build); the product library rejects them. */
int dua_set_dma_conv(int mode);
/* A/B switch of the GroupNorm-apply kernels (returns the previous value): 1 = 8 channels per lane with 16-byte stores on
 * every plane where the channel counts allow (default), 0 = always the 4-channel-per-lane kernel. */
int dua_set_gn_apply_wide(int on);
/* (diagnostic build only; the product library accepts phases <= 1 = off)
 * Start-up stagger of the persistent LDS-DMA workgroups (tuning knob): `phases` phases per XCD (0 / 1 = off), each
 * `percent` % of 1/phases of the estimated tile period; applied when every workgroup owns >= min_tiles tiles.  Spreads
 * the epilogues (output / residual HBM traffic) of the 256 workgroups over the tile period instead of one burst. */
int dua_set_conv_stagger(int phases, int percent, int min_tiles);
int dua_gemm_bf16x3(const float* A1, int K1, long lda1, const float* A2, int K2, long lda2, const void* planes,
                    const float* bias, const float* residual, long ldr, float* C, long ldc, int M, int N, int act_in,
                    int act_out, float alpha, float* colsum, void* stream);

/* Batched C[b] = alpha * A[b] * B[b]^T with BOTH operands fp32 activations split on the fly (the attention products
 * q.k^T and P.v^T of unet_openai.py:350-354 in bf16x3 mode).  A: [M][K] (lda), B: [N][K] (ldb), K % 32 == 0; batch
 * index b -> (b / batch_inner, b % batch_inner) with outer (s*) / inner (s*i) strides in floats. */
int dua_gemm_nt_bf16x3_batched(const float* A, long lda, long sA, long sAi, const float* B, long ldb, long sB, long sBi,
                               float* C, long ldc, long sC, long sCi, int M, int N, int K, int batch, int batch_inner,
                               float alpha, void* stream);
/* out[b][c][r] = in[b][r][c], r < R, c < Cc: lays V out K-major so P.V uses the K-major GEMM path. batch <= 65535. */
int dua_transpose_batched(const float* in, float* out, int R, int Cc, long ld_in, long ld_out, long sIn, long sIni,
                          long sOut, long sOuti, int batch, int batch_inner, void* stream);

/* ---------------- block-level entry points (adm_resblock_fwd, adm_attnblock_fwd) ----------------
 * One ADM block forward per call: host-side sequencing of the per-op entry points above on the caller's stream over a
 * caller-provided workspace (dua_*_workspace_bytes; nothing is allocated inside).  Same kernels, order and operands as the
 * Python composition, hence bit-identical results.  Activations NHWC fp32; weights packed by the caller for the chosen
 * precision: DUA_PREC_FP32 - 3x3 weights from dua_pack_conv3x3_weight, 1x1 / linear weights as stored ([N][K]);
 * DUA_PREC_F16X3 - every weight from dua_pack_weight_f16x3, act_scale / raw_scale the plane scales of GroupNorm outputs and of
 * raw residual-stream operands (the UNet uses 64 and 16).  *_colsums: the producer's epilogue column sums (see dua_conv3x3) or
 * NULL; y_colsums (optional, [ceil(N*H*W/64)][2][Cout] floats) receives those of the block's output. */
#define DUA_PREC_FP32 0
#define DUA_PREC_F16X3 3
/* dua_adm_resblock_fwd only: DUA_PREC_F16X3 with ONE fp16 product per tap in the Winograd-form 3x3 convs (conv*_wino given and the shape
 * supported) - fp16 operands, fp32 accumulation: the arithmetic of the reference's own GPU path (torch.autocast fp16, script :83); every
 * other operation as DUA_PREC_F16X3.  Not a parity mode (11-bit operands): mode "f16w" of the Python host, reported beside the headline. */
#define DUA_PREC_F16W 4
typedef struct {
    const float* x1; int C1;             /* input; (x2, C2) = the skip half of the channel concat of unet_openai.py:662, or (NULL, 0) */
    const float* x2; int C2;
    int N, H, W, Cout;
    const float *gn1_w, *gn1_b;          /* in_layers[0]  (GroupNorm32) */
    const void* conv1_packed; const float* conv1_b;          /* in_layers[2] */
    const float* film; long film_ld;     /* emb_layers(emb) rows [N][2*Cout] = (scale | shift), row stride film_ld floats   :245-253 */
    const float *gn2_w, *gn2_b;          /* out_layers[0] */
    const void* conv2_packed; const float* conv2_b;          /* out_layers[3] */
    const void* skip_packed; const float* skip_b;            /* skip_connection (1x1 conv) or NULL = Identity   :223-226 */
    const float *x1_colsums, *x2_colsums;
    int updown;                          /* 0, +1 nearest x2, -1 2x2 average (resblock_updown, :191-196,238-243) */
    const float* dropmask;               /* out_layers[2] keep-mask / (1-p) in NHWC, or NULL (eval) */
    int precision; float act_scale, raw_scale;
    void* workspace; size_t workspace_bytes;
    float* y; float* y_colsums;          /* [N][Ho][Wo][Cout] */
    const void *conv1_wino, *conv2_wino; /* optional: the 3x3 weights also packed by dua_pack_weight_f16x3_wino (DUA_PREC_F16X3) /
                                            dua_pack_weight_f32_wino (DUA_PREC_FP32); a conv whose shape dua_conv3x3_f16x3_wino_supported
                                            accepts then runs in the Winograd form, and its conv*_packed may be NULL (the direct-form
                                            weights serve every other shape and need not be packed for this one) */
    int conv1_wino_m, conv2_wino_m;      /* 0 / 2: conv*_wino is the F(2,3) packing; 4 (DUA_PREC_F16X3: dua_pack_weight_f16x3_wino4; DUA_PREC_FP32: dua_pack_weight_f32_wino4): the F(4,3) packing of
                                            dua_pack_weight_f16x3_wino4, for a shape dua_conv3x3_f16x3_wino4_supported accepts */
} dua_resblock_args;
/* ResBlock.forward, guided_diffusion/unet_openai.py:237-257 (use_scale_shift_norm, resblock_updown). */
size_t dua_adm_resblock_workspace_bytes(const dua_resblock_args* args_host);
int dua_adm_resblock_fwd(const dua_resblock_args* args_host, void* stream);
typedef struct {
    const float* x; int C; int N, H, W, heads;
    const float *gn_w, *gn_b;            /* norm */
    const void* qkv_packed; const float* qkv_b;   /* qkv 1x1 conv, rows ordered (q | k | v), head-major inside each third (the legacy
                                                     order of QKVAttentionLegacy is a row permutation applied by the caller)   :287,348 */
    const void* proj_packed; const float* proj_b; /* proj_out   :295 */
    const float* x_colsums;
    int precision; float act_scale, raw_scale;
    void* workspace; size_t workspace_bytes;
    float* y; float* y_colsums;          /* x + proj_out(attention(qkv(norm(x))))   :300-306 */
} dua_attnblock_args;
/* AttentionBlock.forward, unet_openai.py:300-306, on the fused attention kernels: head dims <= 128 (multiple of 8) in fp32, <= 256
 * (multiple of 4) in f16x3; wider heads are composed from dua_gemm / dua_softmax_rows by the caller. */
size_t dua_adm_attnblock_workspace_bytes(const dua_attnblock_args* args_host);
typedef struct {
    const float* x; const float* skip;   /* tokens [B*L][D]; skip = the long-skip tokens of an out-block or NULL   uvit.py:116-117 */
    int B, L, D, heads, hidden;
    const void* skip_packed; const float* skip_b;            /* skip_linear [D][2D] or NULL */
    const float *norm1_w, *norm1_b, *norm2_w, *norm2_b; float eps;
    const void* qkv_packed; const float* qkv_b;              /* attn.qkv [3D][D] (bias may be NULL) */
    const void* proj_packed; const float* proj_b;            /* attn.proj */
    const void* fc1_packed; const float* fc1_b;              /* mlp.fc1 [hidden][D] (+ GELU) */
    const void* fc2_packed; const float* fc2_b;              /* mlp.fc2 [D][hidden] */
    int precision; float act_scale, raw_scale;
    void* workspace; size_t workspace_bytes;
    float* y;
    /* DUA_PREC_F16X3, optional (all may be NULL): the f16x3 planes ([B*L][D] fp16 hi | lo of the tokens * raw_scale) of x and of skip -
     * with both given, skip_linear runs on the presplit two-source GEMM (dua_gemm_f16x3_presplit2) instead of splitting the fp32 rows on
     * the fly - and y_planes: the same planes of this block's output, written beside y by the epilogue of fc2 (for the block that will
     * take y as its x or as its long skip). */
    const void* x_planes; const void* skip_planes; void* y_planes;
} dua_uvitblock_args;
/* uvit_block_fwd: Block.forward of uvit/uvit.py:95-120 (skip_linear(cat) -> x + attn(norm1 x) -> x + mlp(norm2 x)), fp32 or f16x3;
 * weights as stored ([N][K], fp32) or packed by dua_pack_weight_f16x3; head dims multiple of 8, <= 128. */
size_t dua_uvit_block_workspace_bytes(const dua_uvitblock_args* args_host);
int dua_uvit_block_fwd(const dua_uvitblock_args* args_host, void* stream);
/* Bench facility: dua_conv_profile(1) makes the block functions bracket each of their conv3x3 launches with HIP events on the
 * launch stream (returns the previous setting); dua_conv_profile_collect synchronises on them, returns the summed launch time (ms),
 * the summed algorithmic FLOPs (2*N*H*W*9*Cin*Cout), the number of launches since the last collect and the matrix-core product
 * FLOPs those launches executed (3 fp16 products per tap in f16x3, 6 of 9 taps in the Winograd form), and frees the events. */
int dua_conv_profile(int on);
int dua_conv_profile_collect(double* total_ms, double* total_flops, long* launches, double* total_mfma_flops);
/* Bench facility for the kernels OUTSIDE the conv3x3 launches (bench.py's `hbm_kernels` list and the per-leg rooflines):
 * dua_kernel_profile(1) makes the sampler kernels (DDIM step, re-noising, Monte-Carlo accumulation, quantile mask, uint8 conversion), the
 * GroupNorm / LayerNorm operand writers (V writers of the Winograd forms, plane writers), the one-tap / skip GEMMs and the attention kernels
 * bracket their launches with HIP events on the launch stream, tagged with a kind and the launch's ALGORITHMIC bytes / FLOPs (returns the
 * previous setting).  dua_kernel_profile_collect synchronises on the events and writes one JSON object
 * {"kind": {"launches", "ms", "min_ms", "bytes", "flops"}, ...} (sums per kind) into buf; returns its length or DUA_EINVAL when cap is too small. */
int dua_kernel_profile(int on);
long dua_kernel_profile_collect(char* buf, size_t cap);
int dua_adm_attnblock_fwd(const dua_attnblock_args* args_host, void* stream);

/* ---------------- U-ViT token plumbing (uvit/uvit.py) ---------------- */
/* PatchEmbed's strided conv as im2row: x NCHW [B][C][H][W] -> rows [B*(H/P)*(W/P)][C*P*P] in the
 * (c, p1, p2) order of the flattened conv weight (uvit.py:123-135); the projection itself is dua_gemm. */
int dua_patchify(const float* x, float* rows, int B, int C, int H, int W, int P, void* stream);
/* tokens[b] = [label_emb[y[b]] ; time_tok[b] ; patches[b]] + pos_embed  (uvit.py:207-216).
 * y may be NULL (no label token, extras = 1). patches [B][L][D], pos_embed [extras+L][D], out [B][extras+L][D]. */
int dua_uvit_assemble(const float* patches, const float* time_tok, const float* label_table, const int64_t* y,
                      const float* pos_embed, float* out, int B, int L, int D, void* stream);
/* tok [B][extras+L][P*P*C] -> NCHW image, dropping the extras tokens:
 * 'B (h w) (p1 p2 C) -> B C (h p1) (w p2)' (uvit.py:46-51,228-229). */
int dua_unpatchify(const float* tok, float* img, int B, int L, int extras, int C, int P, void* stream);

/* f16x3 range check (debug facility, off by default; process-global, synchronises the device when switched or read).  While on, every
 * kernel that WRITES f16x3 operands - the GroupNorm / LayerNorm / GELU / attention plane writers, the Winograd V writer, the on-the-fly
 * split of raw residual-stream operands in dua_gemm_f16x3 - counts, per operand, (a) saturated: |x * plane scale| > 65504, the fp16 hi half
 * clamps and the hi + lo pair no longer carries 22 bits of x: the "fp32-equivalent" claim does not hold for that operand; (b) subnormal:
 * 0 < |x * plane scale| < 2^-14, hi is an fp16 subnormal - harmless (the pair stays exact to 2^-25 / scale absolutely), reported for
 * completeness.  The plane scales are static (64 for GroupNorm / SiLU / LayerNorm outputs: |x| <= 1023; 16 for raw-stream operands, GELU
 * and attention outputs: |x| <= 4094; 32 for the Winograd F(2,3) images: |x| <= 1023; 8 for the F(4,3) images, whose transform rows sum to 10
 * in absolute value: |x| <= 818): a non-zero saturation count on real checkpoints says the activations
 * exceed those bounds and the run should use precision "fp32" (INTEGRATION.md section 4).  dua_range_check returns the previous setting (0 / 1;
 * a negative DUA_E* code when the counters cannot be allocated or cleared) and zeroes the counters when switched on; dua_range_check_read copies them out (reset != 0 zeroes them). */
int dua_range_check(int on);
int dua_range_check_read(unsigned long long* saturated, unsigned long long* subnormal, int reset);

/* Split-K for the f16x3 LDS-DMA conv / presplit GEMM launches that cannot fill the chip (fewer than 128 tiles of 256 rows: small
 * batches, e.g. SD's CFG batch of 2): with a workspace registered, such a launch runs every tile as up to 16 K parts on separate
 * workgroups and a second kernel folds the parts in index order (deterministic) and applies bias / residual / statistics.
 * The one-tap GEMM kernel (csrc/gemm_f16x3.hip) uses the same buffer for its stream-K tail: the tiles of the last, partly filled round
 * are cut into K-step units spread evenly over the 256 workgroups (needs 2 * 256 tiles of 256 x 256 floats = 128 MiB; a smaller buffer
 * leaves that kernel on whole tiles).
 * Results differ from the single-pass kernel by fp32 summation order only; the choice depends on the row count, so outputs are no
 * longer bit-identical across batch sizes while it is on.  Process-global, one stream at a time; (NULL, 0) switches it off (default). */
int dua_set_splitk_workspace(void* ws, size_t bytes);

/* ---------------- input gradients (gradient-mode guidance, uncertainty_guidance.py:83-108,121-129) ----------------
 * use_posterior = False (the default of scripts/generate_with_uncertainty_threshold_stable_diffusion.py:16-17) calls
 * uncertainty.backward() through the M denoiser calls and reads pred_epsilon.grad.  Only gradients w.r.t. activations are
 * needed; convolutions / linears of that pass are dua_conv3x3 / dua_gemm on flipped / transposed weights. */

/* Input gradient of y = act(GroupNorm(cat(x1, x2))) (act 0: identity, 1: SiLU) given dy [N][HW][C1+C2] and the forward
 * statistics (mean, rstd per (sample, group)).  dx1 / dx2 receive the gradient (+ add1 / add2 when given: the other branch of
 * a fan-out).  torch.nn.functional.group_norm backward, input term only. */
int dua_groupnorm_bwd(const float* x1, int C1, const float* x2, int C2, int N, long HW, int G, const float* stats,
                      const float* gamma, const float* beta, int act, const float* dy, const float* add1, const float* add2,
                      float* dx1, float* dx2, void* stream);
/* Input gradient of LayerNorm over the last dim (rows x D, D <= 2048): dx = rstd*(dy*gamma - mean(.) - xh*mean(. * xh)) (+ add). */
int dua_layernorm_bwd(const float* x, const float* gamma, const float* dy, const float* add, float* dx, long rows, int D, float eps,
                      void* stream);
/* Input gradient of dua_geglu: x [M][2F], dy [M][F] -> dx [M][2F]. */
int dua_geglu_bwd(const float* x, const float* dy, float* dx, long M, int F, void* stream);
/* Softmax input gradient in place on dP: dS = P * (dP - sum_j dP_j P_j) per row (cols <= 4096). */
int dua_softmax_bwd_rows(const float* P, float* dP, long rows, int cols, long ld, void* stream);
/* Both in one pass: on entry S = logits, dP = d/dP; on exit S = P = softmax(S) and dP = dS (what the attention gradient needs:
 * P for dV, dS for dQ / dK). */
int dua_softmax_fwd_bwd_rows(float* S, float* dP, long rows, int cols, long ld, void* stream);
/* Adjoint of dua_pick_even: g [N][H/2][W/2][C] -> out [N][H][W][C], zero off the even pixels. */
int dua_place_even(const float* g, float* out, int N, int H, int W, int C, void* stream);
/* grad[m] = scale * (stack[m] - mean_m stack): d/d e_m of (scale/2) * sum_m (e_m - mean)^2, i.e. of the unbiased variance over
 * M samples summed over pixels and averaged over B samples when scale = 2 / ((M-1) * B) (uncertainty_guidance.py:106-107). */
int dua_mc_var_grad(const float* stack, float* grad, int M, size_t n, float scale, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DUA_HIP_H */
