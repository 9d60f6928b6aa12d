"""Latent UNet of Stable Diffusion 1.5 (BASELINE configs[4]) on the HIP library vs the CPU oracle.
PARITY UNPINNED: the network lives in the un-vendored diffusers==0.31.0; oracle/sd_unet.py restates its published
architecture and is what the HIP path is held to (tolerance 1e-3 per element, the north star's bar)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
dev = "cuda"
TOL = 1e-3

TINY = dict(in_channels=4, out_channels=4, block_out_channels=(32, 64), layers_per_block=1,
            down_block_types=("CrossAttnDownBlock2D", "DownBlock2D"), up_block_types=("UpBlock2D", "CrossAttnUpBlock2D"),
            attention_head_dim=4, cross_attention_dim=32, norm_num_groups=32, norm_eps=1e-5)
MID = dict(in_channels=4, out_channels=4, block_out_channels=(64, 128, 128), layers_per_block=2,
           down_block_types=("CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "DownBlock2D"),
           up_block_types=("UpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D"),
           attention_head_dim=8, cross_attention_dim=96, norm_num_groups=32, norm_eps=1e-5)


def build(cfg, seed):
    from diffusion_uncertainty_amd.stable_diffusion.unet_2d_condition import UNet2DConditionModel
    from diffusion_uncertainty_amd.utils import seeded_state_dict
    from oracle import sd_unet
    m = UNet2DConditionModel(**cfg)
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    assert shapes == {k: tuple(v) for k, v in sd_unet.state_shapes(cfg).items()}      # diffusers key / shape layout
    sd = seeded_state_dict(shapes, seed)
    m.load_state_dict(sd, strict=True)
    return m.eval().to(dev), sd


def inputs(cfg, B, hw, Tk, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, 4, hw, hw, generator=g)
    enc = torch.randn(B, Tk, cfg["cross_attention_dim"], generator=g)
    t = torch.tensor([981, 21][:B] if B <= 2 else [981] * B)
    return x, t, enc


@pytest.mark.parametrize("cfg,hw,Tk", [(TINY, 16, 7), (MID, 32, 77)])
@pytest.mark.parametrize("precision", ["fp32", "bf16x3", "f16f8"])
def test_sd_unet_vs_oracle(cfg, hw, Tk, precision):
    from oracle import sd_unet
    m, sd = build(cfg, 11)
    x, t, enc = inputs(cfg, 2, hw, Tk, 3)
    ref = sd_unet.forward(sd, cfg, x, t, enc)
    m.precision = precision
    out = m(x.to(dev), t.to(dev), encoder_hidden_states=enc.to(dev), return_dict=False)[0]
    err = (out.cpu() - ref).abs().max().item()
    print(f"sd-unet {precision} hw={hw}: max|hip - oracle| = {err:.3e} (|ref|max {ref.abs().max().item():.2f})")
    assert out.shape == ref.shape and err < TOL
    # scalar timestep (the pipeline passes a 0-dim tensor) and the record return
    o2 = m(x[:1].to(dev), torch.tensor(981), encoder_hidden_states=enc[:1].to(dev))
    assert (o2.sample.cpu() - ref[:1]).abs().max().item() < TOL
    with pytest.raises(RuntimeError):
        m(x, t, encoder_hidden_states=enc)                       # no CPU fallback


def test_sd15_full_size_vs_oracle():
    """The SD-1.5 configuration itself (860 M parameters, latents 64x64, 77 text tokens, CFG batch 2): one forward."""
    from oracle import sd_unet
    cfg = dict(sd_unet.SD15)
    m, sd = build(cfg, 15)
    assert sum(p.numel() for p in m.parameters()) == 859520964            # parameter count of the SD-1.5 UNet
    x, t, enc = inputs(cfg, 2, 64, 77, 5)
    torch.set_num_threads(min(32, torch.get_num_threads()))
    ref = sd_unet.forward(sd, cfg, x, t, enc)
    for precision in ("fp32", "f16f8"):
        m.precision = precision
        out = m(x.to(dev), t.to(dev), encoder_hidden_states=enc.to(dev), return_dict=False)[0]
        err = (out.cpu() - ref).abs().max().item()
        print(f"sd-1.5 unet {precision}: max|hip - oracle| = {err:.3e} (|ref|max {ref.abs().max().item():.2f})")
        assert err < TOL
