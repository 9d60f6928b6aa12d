"""Input-gradient kernels (csrc/backward.hip), the SD UNet's vector-Jacobian product and gradient-mode guidance
(reference uncertainty_guidance.py:83-108,121-129, use_posterior = False) on the HIP library.

Checkers: torch autograd in fp32 on the CPU for the single kernels (a floating-point kernel vs the plain PyTorch op it
differentiates), oracle/sd_unet.input_vjp and oracle/guidance.guided_score_percentile_gradient (torch autograd through the
CPU restatement, exactly what the reference does with the diffusers UNet) for the network and the guidance step.
PARITY UNPINNED for the network itself (un-vendored diffusers), as in tests/test_gpu_sd.py; the guidance step itself - everything
uncertainty_guidance.py computes around the denoiser calls - is pinned to the reference's own run by G7b
(test_gradient_mode_guidance_vs_reference_g7b)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from test_gpu_sd import MID, TINY, build, inputs

pytestmark = pytest.mark.gpu
dev = "cuda"


def rel_err(a, b):
    return (a - b).abs().max().item() / max(b.abs().max().item(), 1e-30)


@pytest.mark.parametrize("C1,C2,G,act", [(64, 0, 32, 1), (96, 32, 32, 1), (320, 0, 32, 0), (1280, 640, 32, 1)])
def test_groupnorm_bwd_vs_autograd(C1, C2, G, act):
    from diffusion_uncertainty_amd import ops
    g = torch.Generator().manual_seed(C1 + C2)
    N, H, W = 3, 8, 8
    x = (torch.randn(N, C1 + C2, H, W, generator=g) * 1.7 + 0.3).requires_grad_(True)
    ga, be = torch.randn(C1 + C2, generator=g), torch.randn(C1 + C2, generator=g)
    dy = torch.randn(N, C1 + C2, H, W, generator=g)
    add = torch.randn(N, C1 + C2, H, W, generator=g)
    y = F.group_norm(x, G, ga, be, 1e-5)
    y = F.silu(y) if act else y
    (y * dy).sum().backward()
    want = x.grad + add
    nhwc = lambda t: t.detach().permute(0, 2, 3, 1).contiguous().to(dev)
    x1, x2 = nhwc(x[:, :C1]), (nhwc(x[:, C1:]) if C2 else None)
    st = ops.groupnorm_stats(x1, x2, groups=G, eps=1e-5)
    dx1, dx2 = ops.groupnorm_bwd(x1, st, ga.to(dev), be.to(dev), nhwc(dy), x2=x2, act=act, groups=G, add1=nhwc(add[:, :C1]),
                                 add2=nhwc(add[:, C1:]) if C2 else None)
    got = torch.cat([dx1] + ([dx2] if C2 else []), dim=3).permute(0, 3, 1, 2).cpu()
    assert rel_err(got, want) < 2e-5
    dx1n, _ = ops.groupnorm_bwd(x1, st, ga.to(dev), be.to(dev), nhwc(dy), x2=x2, act=act, groups=G)      # without the fan-out term
    assert rel_err(dx1n.permute(0, 3, 1, 2).cpu(), x.grad[:, :C1]) < 2e-5


@pytest.mark.parametrize("D", [32, 320, 1280])
def test_layernorm_geglu_bwd_vs_autograd(D):
    from diffusion_uncertainty_amd import ops
    g = torch.Generator().manual_seed(D)
    rows = 37
    x = (torch.randn(rows, D, generator=g) * 2 - 0.5).requires_grad_(True)
    ga, be, dy, add = (torch.randn(D, generator=g), torch.randn(D, generator=g), torch.randn(rows, D, generator=g),
                       torch.randn(rows, D, generator=g))
    (F.layer_norm(x, (D,), ga, be, 1e-5) * dy).sum().backward()
    got = ops.layernorm_bwd(x.detach().to(dev), ga.to(dev), dy.to(dev), eps=1e-5, add=add.to(dev)).cpu()
    assert rel_err(got, x.grad + add) < 2e-5
    u = torch.randn(rows, 2 * D, generator=g).requires_grad_(True)
    a, gate = u.chunk(2, dim=-1)
    (a * F.gelu(gate) * dy).sum().backward()
    got = ops.geglu_bwd(u.detach().to(dev), dy.to(dev)).cpu()
    assert rel_err(got, u.grad) < 2e-5


def test_place_even_and_var_grad():
    from diffusion_uncertainty_amd import ops
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 6, 8, 12, generator=g)
    out = ops.place_even(x.to(dev), 12, 16).cpu()
    want = torch.zeros(2, 12, 16, 12)
    want[:, ::2, ::2] = x
    np.testing.assert_array_equal(out.numpy(), want.numpy())
    # <pick_even(a), b> == <a, place_even(b)>: the two are adjoint
    a = torch.randn(2, 12, 16, 12, generator=g)
    assert abs((ops.pick_even(a.to(dev)).cpu() * x).sum().item() - (a * out).sum().item()) < 1e-3
    st = torch.randn(5, 1, 4, 16, 16, generator=g).requires_grad_(True)
    torch.var(st, dim=0).mean(dim=0).sum().backward()
    got = ops.mc_var_grad(st.detach().to(dev), 2.0 / (4 * 1)).cpu()
    assert rel_err(got, st.grad) < 1e-5


@pytest.mark.parametrize("B,T,heads,ch", [(2, 64, 2, 40), (3, 256, 4, 8), (1, 1024, 2, 80), (1, 4096, 1, 40)])
def test_self_attention_bwd_vs_autograd(B, T, heads, ch):
    from diffusion_uncertainty_amd import ops
    g = torch.Generator().manual_seed(T)
    C = heads * ch
    qkv = torch.randn(B * T, 3 * C, generator=g).requires_grad_(True)
    do = torch.randn(B * T, C, generator=g)
    q, k, v = qkv.view(B, T, 3, heads, ch).permute(2, 0, 3, 1, 4)
    o = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(B * T, C)
    (o * do).sum().backward()
    fwd = ops.qkv_attention(qkv.detach().to(dev), B, T, heads, ch)
    assert rel_err(fwd.cpu(), o.detach()) < 2e-5
    got = ops.qkv_attention_bwd(qkv.detach().to(dev), do.to(dev), B, T, heads, ch).cpu()
    assert rel_err(got, qkv.grad) < 5e-5
    old = ops.ATTN_BWD_CHUNK_FLOATS
    try:                                            # one sample per chunk: the same numbers
        ops.ATTN_BWD_CHUNK_FLOATS = 1
        got1 = ops.qkv_attention_bwd(qkv.detach().to(dev), do.to(dev), B, T, heads, ch).cpu()
    finally:
        ops.ATTN_BWD_CHUNK_FLOATS = old
    np.testing.assert_array_equal(got1.numpy(), got.numpy())


def test_cross_attention_bwd_q_vs_autograd():
    from diffusion_uncertainty_amd import ops
    g = torch.Generator().manual_seed(5)
    B, T, Tk, heads, ch = 2, 256, 77, 8, 40
    C = heads * ch
    q = torch.randn(B, T, C, generator=g).requires_grad_(True)
    k, v, do = torch.randn(B, Tk, C, generator=g), torch.randn(B, Tk, C, generator=g), torch.randn(B, T, C, generator=g)
    sp = lambda z: z.reshape(B, z.shape[1], heads, ch).transpose(1, 2)
    o = F.scaled_dot_product_attention(sp(q), sp(k), sp(v)).transpose(1, 2).reshape(B, T, C)
    (o * do).sum().backward()
    got = ops.attention_bwd_q(q.detach().to(dev), k.to(dev), v.to(dev), do.to(dev), heads).cpu()
    assert rel_err(got, q.grad) < 5e-5


@pytest.mark.parametrize("cfg,hw,Tk", [(TINY, 16, 7), (MID, 32, 77)])
@pytest.mark.parametrize("precision", ["fp32", "f16x3"])
def test_sd_unet_input_vjp_vs_oracle(cfg, hw, Tk, precision):
    """d<g, unet(x)>/dx of the HIP UNet (taped forward in the model's precision, exact-fp32 backward) vs torch autograd
    through the CPU restatement."""
    from oracle import sd_unet
    m, sd = build(cfg, 11)
    x, t, enc = inputs(cfg, 2, hw, Tk, 3)
    gout = torch.randn(x.shape, generator=torch.Generator().manual_seed(8))
    ref_out, ref_grad = sd_unet.input_vjp(sd, cfg, x, t, enc, gout)
    m.precision = precision
    out, tape = m.forward_with_tape(x.to(dev), t.to(dev), enc.to(dev))
    grad = m.backward_sample(tape, gout.to(dev))
    e_out, e_grad = (out.cpu() - ref_out).abs().max().item(), rel_err(grad.cpu(), ref_grad)
    print(f"sd-unet vjp {precision} hw={hw}: out err {e_out:.3e}, grad rel err {e_grad:.3e} (|grad|max {ref_grad.abs().max().item():.3e})")
    assert grad.shape == x.shape and e_out < 1e-3 and e_grad < 1e-3
    m.backward_precision = "bf16x3"                 # convolutions / linears of the backward pass on the split-bf16 engine
    out2, tape = m.forward_with_tape(x.to(dev), t.to(dev), enc.to(dev))
    e_grad16 = rel_err(m.backward_sample(tape, gout.to(dev)).cpu(), ref_grad)
    print(f"    backward_precision bf16x3: grad rel err {e_grad16:.3e}")
    assert e_grad16 < 1e-3
    # the taped forward is the plain forward
    plain = m(x.to(dev), t.to(dev), encoder_hidden_states=enc.to(dev), return_dict=False)[0]
    np.testing.assert_array_equal(plain.cpu().numpy(), out.cpu().numpy())


def test_gradient_mode_guidance_vs_oracle():
    """use_posterior = False: eps + lr * mask * d(sum var)/d eps on the HIP library vs the oracle's autograd restatement, same
    noise tape (CPU draws), M = 3, mid-size UNet."""
    from diffusion_uncertainty_amd import uncertainty_guidance as ug
    from diffusion_uncertainty_amd.schedulers_uncertainty.noise import HostNoise
    from oracle import guidance, sd_unet
    m, sd = build(MID, 21)
    m.precision = "f16x3"
    g = torch.Generator().manual_seed(13)
    lat = torch.randn(1, 4, 16, 16, generator=g)
    eps = torch.randn(1, 4, 16, 16, generator=g)
    y = torch.randn(2, 77, 96, generator=g)
    x_in = torch.cat([lat] * 2)
    t, a, p, M, s, lr = torch.tensor(499), 0.31, 0.9, 3, 7.5, 0.8
    torch.manual_seed(6)
    ref, ref_grad, ref_u, ref_mask = guidance.guided_score_percentile_gradient(
        eps, x_in, lambda xh, tt, yy: sd_unet.forward_autograd(sd, MID, xh, torch.full((2,), int(tt)), yy), t, y, s, a, p, M,
        torch.randn_like, lr=lr)
    old = ug.use_posterior, ug.noise_source
    try:
        ug.use_posterior, ug.noise_source = False, HostNoise()
        torch.manual_seed(6)
        out = ug.get_uncertainty_guided_score_with_percentile(eps.to(dev), x_in.to(dev), t, y.to(dev), m, torch.tensor(a), p,
                                                              "stable-diffusion", num_uncertainty_samples=M, guidance_scale=s, lr=lr)
        with pytest.raises(NotImplementedError):           # a third-party denoiser has no HIP backward
            ug.get_uncertainty_guided_score_with_percentile(eps.to(dev), x_in.to(dev), t, y.to(dev), lambda **k: None, torch.tensor(a),
                                                            p, "stable-diffusion", num_uncertainty_samples=M, guidance_scale=s)
    finally:
        ug.use_posterior, ug.noise_source = old
    d = (out.cpu() - ref).abs()
    upd = (ref - eps).abs().max().item()
    print(f"gradient-mode guidance: max|hip - oracle| = {d.max().item():.3e}, update size {upd:.3e}, mask px {int(ref_mask.sum())}")
    # a pixel whose variance sits at the quantile may flip side of the hard threshold: allow a handful, hold the rest to the bar
    assert upd > 1e-3 and (d > 1e-3 * max(1.0, upd)).float().mean().item() < 5e-3 and d.median().item() < 1e-5 * max(1.0, upd)


class _FakeCFGUNetHIP:
    """The fake CFG denoisers of tests/golden/make_golden.py (g7 / g7b) on the HIP library, with the forward_with_tape / backward_sample
    protocol gradient-mode guidance asks of a denoiser: conv3x3 [-> SiLU -> conv3x3] + the context's mean, convolutions forward and
    backward (flipped, transposed weights) on dua_conv3x3.  Test-side model: only the activation and the scalar shift are torch ops."""

    def __init__(self, ws, ctx_ndim=3):
        from diffusion_uncertainty_amd import ops
        self.ops = ops
        self.w = [w.to(dev) for w in ws]
        self.fwd = [ops.pack_conv3x3_weight(w.contiguous()) for w in self.w]
        self.bwd = [ops.pack_conv3x3_weight(w.transpose(0, 1).flip(2, 3).contiguous()) for w in self.w]

    def _conv(self, x, wp, cout):
        return self.ops.conv3x3(x, wp, None, cout)

    def forward_with_tape(self, sample, timestep, encoder_hidden_states):
        h = self.ops.nchw_to_nhwc(sample.float().contiguous(), 4)
        pre = []
        for i, w in enumerate(self.w):
            h = self._conv(h, self.fwd[i], w.shape[0])
            if i + 1 < len(self.w):
                pre.append(h)
                h = F.silu(h)
        out = self.ops.nhwc_to_nchw(h) + encoder_hidden_states.mean(dim=(1, 2))[:, None, None, None]
        return out.contiguous(), pre

    def backward_sample(self, tape, grad_out):
        g = self.ops.nchw_to_nhwc(grad_out.float().contiguous(), 4)
        for i in reversed(range(len(self.w))):
            if i + 1 < len(self.w):
                z = tape[i]
                sg = torch.sigmoid(z)
                g = (g * (sg * (1 + z * (1 - sg)))).contiguous()
            g = self._conv(g, self.bwd[i], self.w[i].shape[1])
        return self.ops.nhwc_to_nchw(g).contiguous()


@pytest.mark.parametrize("tag", ["sd", "nl"])
def test_gradient_mode_guidance_vs_reference_g7b(golden, tag):
    """use_posterior = False against the REFERENCE's own run of uncertainty_guidance.py:79-108,121-129 (G7b, plain autograd on the CPU):
    the product function - re-noising, CFG combine, variance over the M samples, its gradient (csrc/backward.hip dua_mc_var_grad), the batch
    sum, the quantile mask and the masked update, all on the HIP library - with the fixture's fake CFG denoisers, same noise tape.
    Bars: pred_epsilon.grad-sized update within 1e-3 (relative to the gradient's scale) outside a handful of pixels that sit on the
    quantile; the mask kernel bit-exact on the fixture's own variance map."""
    from conftest import T
    from diffusion_uncertainty_amd import ops
    from diffusion_uncertainty_amd import uncertainty_guidance as ug
    from diffusion_uncertainty_amd.schedulers_uncertainty.noise import HostNoise
    g = golden("g7b_guidance_gradient")
    model = _FakeCFGUNetHIP([T(g["sd_w"])] if tag == "sd" else [T(g["nl_w1"]), T(g["nl_w2"])])
    eps, inp, ctx = T(g["sd_eps"]), T(g["sd_input"]), T(g["sd_ctx"])
    old = ug.use_posterior, ug.noise_source
    try:
        ug.use_posterior, ug.noise_source = False, HostNoise()
        for lr in g["lrs"]:
            for i, p in enumerate(g["ps"] if tag == "sd" else g["nl_ps"]):
                ref, ref_grad, ref_u, ref_mask = (T(g[f"{tag}_final_{p}_lr{lr}"]), T(g[f"{tag}_grad_{p}"]), T(g[f"{tag}_u_{p}"]),
                                                  T(g[f"{tag}_mask_{p}"]))
                torch.manual_seed(int(g["seed_base"]) + i)
                out = ug.get_uncertainty_guided_score_with_percentile(
                    eps.to(dev), inp.to(dev), torch.tensor([500]), ctx.to(dev), model, torch.tensor(float(g["sd_alpha"])), float(p),
                    "stable-diffusion", num_uncertainty_samples=int(g["M"]), guidance_scale=float(g["guidance_scale"]), lr=float(lr)).cpu()
                scale = max(1.0, ref_grad.abs().max().item())
                d = (out - ref).abs()
                flips = (d > 1e-3 * scale).float().mean().item()
                print(f"g7b {tag} p={p} lr={lr}: max|hip - reference| = {d.max().item():.3e} (gradient scale {scale:.3g}), "
                      f"{int((d > 1e-3 * scale).sum())} px over the bar, median {d.median().item():.2e}")
                assert flips < 5e-4 and d.median().item() < 1e-5 * scale
                # the mask on the fixture's own variance map: bit-exact (a14), and the quantile itself
                mask, q = ops.quantile_mask(ref_u.to(dev), float(p))
                np.testing.assert_array_equal(mask.cpu().numpy().astype(np.uint8), ref_mask.numpy())
                np.testing.assert_array_equal(q.cpu().numpy().reshape(-1), g[f"{tag}_q_{p}"])
                # outside the reference's mask the score is untouched, bit for bit (final = eps + lr * grad * 0)
                same = (out == eps)[ref_mask == 0].float().mean().item()
                assert same > 0.999
    finally:
        ug.use_posterior, ug.noise_source = old


def test_gradient_mode_pipeline_vs_oracle():
    """PLMS + CFG + 2 gradient-guided steps (M = 3, lr = 0.5) on a mid-size UNet: final latents of the HIP pipeline vs the oracle
    loop (autograd through the restatement) on the same noise tape."""
    from diffusion_uncertainty_amd import uncertainty_guidance as ug
    from diffusion_uncertainty_amd.pipeline_uncertainty.pipeline_stable_diffusion_uncertainty_guided import StableDiffusionPipelineUncertainty
    from diffusion_uncertainty_amd.schedulers_uncertainty.noise import HostNoise
    from diffusion_uncertainty_amd.schedulers_uncertainty.scheduling_pndm import PNDMScheduler
    from oracle import sd_pipeline, sd_unet
    m, sd = build(MID, 21)
    m.precision = "f16x3"
    g = torch.Generator().manual_seed(9)
    lat = torch.randn(1, 4, 16, 16, generator=g)
    pe, ne = torch.randn(1, 77, 96, generator=g), torch.randn(1, 77, 96, generator=g)
    torch.manual_seed(4)
    ref = sd_pipeline.sample(lambda x, t, e: sd_unet.forward_autograd(sd, MID, x, t, e), lat, pe, ne, 6, 7.5, 2, 2, 0.9, 3,
                             use_posterior=False, lr=0.5)
    pipe = StableDiffusionPipelineUncertainty(m, PNDMScheduler())
    old = ug.use_posterior, ug.noise_source
    try:
        ug.use_posterior, ug.noise_source = False, HostNoise()
        torch.manual_seed(4)
        out = pipe(prompt_embeds=pe, negative_prompt_embeds=ne, num_inference_steps=6, guidance_scale=7.5, latents=lat,
                   start_step_uc=2, num_steps_uc=2, percentile=0.9, num_uncertainty_samples=3, height=128, width=128, lr=0.5)
    finally:
        ug.use_posterior, ug.noise_source = old
    d = (out.images.cpu() - ref).abs()
    print(f"sd pipeline, gradient update: max|hip - oracle| final latents = {d.max().item():.3e} (|ref|max {ref.abs().max().item():.2f})")
    assert (d > 1e-3).float().mean().item() < 5e-3 and d.median().item() < 1e-4


def test_sd15_full_size_input_vjp_vs_oracle():
    """The SD-1.5 configuration itself (latents 64x64, 77 text tokens, CFG batch 2, self-attention over 4096 tokens): one taped
    forward in f16x3 and its input gradient vs torch autograd through the CPU restatement."""
    from oracle import sd_unet
    cfg = dict(sd_unet.SD15)
    m, sd = build(cfg, 15)
    x, t, enc = inputs(cfg, 2, 64, 77, 5)
    gout = torch.randn(x.shape, generator=torch.Generator().manual_seed(8))
    torch.set_num_threads(min(32, torch.get_num_threads()))
    ref_out, ref_grad = sd_unet.input_vjp(sd, cfg, x, t, enc, gout)
    m.precision = "f16x3"
    out, tape = m.forward_with_tape(x.to(dev), t.to(dev), enc.to(dev))
    grad = m.backward_sample(tape, gout.to(dev))
    e_out, e_grad = (out.cpu() - ref_out).abs().max().item(), rel_err(grad.cpu(), ref_grad)
    print(f"sd-1.5 unet vjp f16x3: out err {e_out:.3e}, grad rel err {e_grad:.3e} (|grad|max {ref_grad.abs().max().item():.3e})")
    assert e_out < 1e-3 and e_grad < 1e-3
    m.backward_precision = "bf16x3"
    out, tape = m.forward_with_tape(x.to(dev), t.to(dev), enc.to(dev))
    e_grad16 = rel_err(m.backward_sample(tape, gout.to(dev)).cpu(), ref_grad)
    print(f"    backward_precision bf16x3: grad rel err {e_grad16:.3e}")
    assert e_grad16 < 1e-3
