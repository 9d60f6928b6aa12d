"""Host-side logic that needs no GPU: config plumbing, factory, schedule tables, timestep grids, window errors."""
import argparse

import numpy as np
import pytest
import torch

from diffusion_uncertainty_amd.configuration import DDPMScheduler
from diffusion_uncertainty_amd.init_model import (init_scheduler_imagenet128, init_scheduler_imagenet64, load_uvit_scheduler)
from diffusion_uncertainty_amd.schedulers_uncertainty.get_uncertainty_scheduler import (get_uncertainty_scheduler,
                                                                                       instatiate_uc_scheduler)
from diffusion_uncertainty_amd.schedulers_uncertainty.mixin import SchedulerUncertaintyClassConditionedMixin, SchedulerUncertaintyMixin


def args(kind, **kw):
    d = dict(scheduler_type=kind, M=5, start_step_uc=4, num_steps_uc=3, predict_next=False, eta=0.0, num_zigzag=3, uncertainty_distance=1)
    d.update(kw)
    return argparse.Namespace(**d)


@pytest.mark.parametrize("name,base", [("adm128", init_scheduler_imagenet128), ("adm64", init_scheduler_imagenet64), ("uvit", load_uvit_scheduler)])
def test_tables_and_grids_match_reference(golden, name, base):
    g = golden("g1_schedules")
    for n in (10, 20, 50):
        after, num = (4, 3) if n == 10 else (n - 10, 10)
        s = get_uncertainty_scheduler(args("uncertainty_zigzag_centered", start_step_uc=after, num_steps_uc=num), None, None, base())
        s.set_timesteps(n)
        np.testing.assert_allclose(s.betas.numpy(), g[f"{name}_betas"], rtol=1e-6)
        np.testing.assert_allclose(s.alphas_cumprod.numpy(), g[f"{name}_alphas_cumprod"], rtol=1e-4)
        np.testing.assert_array_equal(s.timesteps.numpy(), g[f"{name}_timesteps_{n}"])
        assert [after, num, s.timestep_after_step, s.timestep_end_step] == g[f"{name}_window_{n}"].tolist()
        assert len(s) == 1000


def test_factory_semantics():
    base = init_scheduler_imagenet128()
    y = torch.tensor([1, 2])
    z = get_uncertainty_scheduler(args("uncertainty_zigzag_centered"), y, "UNET", base)
    assert z.prompt_embeds is None                 # upstream drops y= (zigzag __init__ has prompt_embed)
    assert (z.M, z.num_zigzag, z.unet, z.predict_next, z.config.after_step, z.config.num_steps_uc) == (5, 3, "UNET", True, 4, 3)
    assert z.config.clip_sample is True and z.config.set_alpha_to_one is True and z.config.steps_offset == 0
    c = get_uncertainty_scheduler(args("uncertainty_centered", predict_next=True), y, None, base)
    assert c.predict_next is True and c.prompt_embeds is y
    assert instatiate_uc_scheduler is get_uncertainty_scheduler
    for s in (z, c):
        assert not isinstance(s, SchedulerUncertaintyMixin)          # attrs appear with set_timesteps, like upstream
        s.set_timesteps(10)
        assert isinstance(s, SchedulerUncertaintyMixin) and isinstance(s, SchedulerUncertaintyClassConditionedMixin)
    u = load_uvit_scheduler()
    assert "set_alpha_to_one" not in u.config and u.config.steps_offset == 1 and u.config.clip_sample is False
    with pytest.raises(NotImplementedError):
        get_uncertainty_scheduler(args("flip_grad"), y, None, base)
    fl = get_uncertainty_scheduler(args("flip"), y, "UNET", base)                 # (no M; prompt_embeds= is taken)
    assert fl.prompt_embeds is y and fl.unet == "UNET" and fl.config.after_step == 4 and not hasattr(fl, "M")
    k = get_uncertainty_scheduler(args("dpm_2_uncertainty_centered"), y, "UNET", base)     # y= / eta= are dropped by from_config
    assert (k.M, k.unet, k.prompt_embeds, k.config.solver_order, k.config.timestep_spacing) == (5, "UNET", None, 2, "leading")
    assert k.config.variance_type == "fixed_small" and k.config.final_sigmas_type == "zero"
    k.set_timesteps(10)                                                      # N + 1 in the 'leading' grid (:305-310)
    assert k.timesteps.tolist() == [900, 810, 720, 630, 540, 450, 360, 270, 180, 90] and float(k.sigmas[-1]) == 0.0
    assert isinstance(k, SchedulerUncertaintyMixin) and (k.timestep_after_step, k.timestep_end_step) == (540, 360)
    d = get_uncertainty_scheduler(args("uncertainty_centered_d", uncertainty_distance=3), y, None, base)
    assert d.prompt_embeds is None and d.uncertainty_distance == 3       # y= is dropped: this __init__ takes prompt_embed
    d.set_timesteps(10)
    assert d.timestep_index[900] == 0 and d.index_timestep[9] == 0
    im = get_uncertainty_scheduler(args("uncertainty_image", predict_next=True), y, None, base)
    assert im.predict_next is True and im.prompt_embeds is y
    assert get_uncertainty_scheduler(args("uncertainty_original", predict_next=True), y, None, base).predict_next is False
    assert type(get_uncertainty_scheduler(args("uncertainty"), y, None, base)).__module__.endswith("scheduling_ddim_uncertainty")


def test_mc_dropout_factory_overrides_dropout_p():
    unet = torch.nn.Sequential(torch.nn.Dropout(0.5), torch.nn.Linear(2, 2))
    from diffusion_uncertainty_amd.schedulers_uncertainty.scheduling_ddim_mc_dropout import DDIMSchedulerUncertaintyImagenetClassConditioned as D
    s = D.from_config(init_scheduler_imagenet128().config, unet=unet, M=3, after_step=0, num_steps_uc=2, dropout=0.2, bogus=1)
    assert unet[0].p == 0.2 and s.class_conditioned is True
    s = get_uncertainty_scheduler(args("mc_dropout"), torch.tensor([0]), unet, init_scheduler_imagenet128())
    assert s.prompt_embeds is None                 # the factory passes prompt_embeds=, which this __init__ does not take


def test_errors():
    s = get_uncertainty_scheduler(args("uncertainty_zigzag_centered"), None, None, init_scheduler_imagenet128())
    with pytest.raises(ValueError):
        s.set_timesteps(1001)
    with pytest.raises(IndexError):
        get_uncertainty_scheduler(args("uncertainty_zigzag_centered", start_step_uc=8), None, None, init_scheduler_imagenet128()).set_timesteps(10)
    s.set_timesteps(10)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        s.step(torch.zeros(1, 3, 4, 4), 900, torch.zeros(1, 3, 4, 4))
    assert DDPMScheduler().config.num_train_timesteps == 1000


def test_seeded_inputs_and_starting_points():
    from diffusion_uncertainty_amd.generate_samples import _seeded_inputs
    from diffusion_uncertainty_amd.utils import starting_points
    x, y = _seeded_inputs(5, 2, 3, 8, 10, init_seed_rng=3, skip_seed=1)
    assert x.shape == (6, 3, 8, 8) and y.shape == (6,)
    g = torch.Generator().manual_seed(4)
    assert torch.equal(x[2:4], torch.randn(2, 3, 8, 8, generator=g))
    a, b = starting_points(4, (3, 8, 8))
    g = torch.Generator().manual_seed(49394)
    assert torch.equal(a, torch.randn(4, 3, 8, 8, generator=g)) and torch.equal(b, torch.randint(0, 1000, (4,), generator=g))


def test_cli_flags_mirror_reference_defaults():
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("gen_cli", os.path.join(root, "scripts", "generate_dataset_score_uncertainty_imagenet.py"))
    cli = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cli)
    a = cli.parse_args(["--image-size", "128", "--scheduler-type", "uncertainty_zigzag_centered", "--model-type", "unet", "--start-index", "0"])
    assert (a.num_samples, a.batch_size, a.generation_steps, a.M, a.start_step_uc, a.num_steps_uc, a.seed, a.eta, a.dropout, a.num_zigzag) == \
        (300, 32, 20, 30, 0, 20, 0, 0.0, 0.1, 3)                       # reference parse_args :150-174
    assert a.precision == "f16x3"                                      # default engine mode: the fp32-equivalent split
    b = cli.parse_args(["--config", "imagenet128_readme_example", "-M", "7"])
    assert (b.M, b.generation_steps, b.start_step_uc, b.num_steps_uc, b.batch_size, b.image_size) == (7, 50, 40, 10, 128, 128)
    with pytest.raises(AssertionError):
        cli.parse_args(["--image-size", "128"])


def test_plain_ddim_scheduler_and_pipeline_surface():
    """Plain DDIM scheduler (what the guided-sampling script hands to the pipelines) and the pipelines' host-side checks."""
    from diffusion_uncertainty_amd.init_model import init_scheduler_imagenet128
    from diffusion_uncertainty_amd.pipeline_uncertainty.pipeline_sampler_class_conditional import DiffusionClassConditional
    from diffusion_uncertainty_amd.pipeline_uncertainty.pipeline_sampler_class_conditional_uncertainty_guided_posterior_distribution import \
        DiffusionClassConditionalGuidedPosteriorDistribution as Guided
    from diffusion_uncertainty_amd.schedulers_uncertainty.mixin import SchedulerUncertaintyMixin
    from diffusion_uncertainty_amd.schedulers_uncertainty.scheduling_ddim import DDIMScheduler
    s = DDIMScheduler.from_config(init_scheduler_imagenet128().config)
    with pytest.raises(ValueError):
        s._ddim(None, 0, None, 0.0, False, None)                       # set_timesteps not called
    s.set_timesteps(10)
    assert s.timesteps.tolist() == list(range(900, -1, -100))
    assert not isinstance(s, SchedulerUncertaintyMixin)                # no uncertainty window on the plain scheduler
    s.config.after_step, s.config.num_steps_uc = 6, 2                  # the guided pipeline assigns these (:117-118)
    s.set_timesteps(len(s.timesteps))
    assert s.timesteps.tolist() == list(range(900, -1, -100)) and s.config.after_step == 6
    model = torch.nn.Identity()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        DiffusionClassConditional(model, s, 16, torch.device("cpu"), 2, 0)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        Guided(model, s, 0.9, 16, torch.device("cpu"), 2, 0)
    with pytest.raises(AssertionError):
        Guided(model, s, 2, 16, torch.device("cuda"), 2, 0)            # threshold must be a float or a tensor


def test_pndm_scheduler_tables():
    """SD-1.5 scheduler_config: scaled_linear 0.00085..0.012, steps_offset 1, skip_prk_steps (second timestep twice)."""
    from diffusion_uncertainty_amd.schedulers_uncertainty.scheduling_pndm import PNDMScheduler
    s = PNDMScheduler()
    with pytest.raises(ValueError):
        s.step(torch.zeros(1), 1, torch.zeros(1))
    s.set_timesteps(20)
    assert s.timesteps.tolist() == [951, 901, 901] + list(range(851, 0, -50))
    assert len(s) == 1000 and abs(float(s.alphas_cumprod[0]) - 0.99915) < 1e-6 and abs(float(s.alphas_cumprod[-1]) - 0.004660) < 1e-5
    assert float(s.final_alpha_cumprod) == float(s.alphas_cumprod[0])
    with pytest.raises(NotImplementedError):
        PNDMScheduler(skip_prk_steps=False)


def test_sd15_unet_key_layout_matches_diffusers_checkpoint():
    """The SD-1.5 latent UNet holds its parameters under diffusers' UNet2DConditionModel keys (strict-loadable): 686
    tensors, 859 520 964 parameters - the published size of runwayml/stable-diffusion-v1-5's UNet - and the host module
    agrees with the oracle's independent enumeration of the layout."""
    from diffusion_uncertainty_amd.stable_diffusion.unet_2d_condition import UNet2DConditionModel
    from oracle import sd_unet
    m = UNet2DConditionModel()
    have = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    want = {k: tuple(v) for k, v in sd_unet.state_shapes(sd_unet.SD15).items()}
    assert have == want and len(have) == 686
    assert sum(p.numel() for p in m.parameters()) == 859520964
    for k in ("down_blocks.0.attentions.0.transformer_blocks.0.attn2.to_k.weight", "mid_block.resnets.1.time_emb_proj.bias",
              "up_blocks.3.resnets.2.conv_shortcut.weight", "up_blocks.0.upsamplers.0.conv.weight", "conv_norm_out.weight"):
        assert k in have
    assert have["down_blocks.0.attentions.0.transformer_blocks.0.attn2.to_k.weight"] == (320, 768)
    assert have["up_blocks.0.resnets.0.conv1.weight"] == (1280, 2560, 3, 3)
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 4, 8, 8), 1, encoder_hidden_states=torch.zeros(1, 77, 768))      # no CPU fallback


def test_oracle_sd_unet_runs_on_cpu():
    """The (unpinned) oracle restatement itself: a tiny configuration runs and is deterministic."""
    from diffusion_uncertainty_amd.utils import seeded_state_dict
    from oracle import sd_unet
    cfg = dict(in_channels=4, out_channels=4, block_out_channels=(32, 64), layers_per_block=1,
               down_block_types=("CrossAttnDownBlock2D", "DownBlock2D"), up_block_types=("UpBlock2D", "CrossAttnUpBlock2D"),
               attention_head_dim=4, cross_attention_dim=32, norm_num_groups=32, norm_eps=1e-5)
    sd = seeded_state_dict(sd_unet.state_shapes(cfg), 3)
    g = torch.Generator().manual_seed(0)
    x, enc = torch.randn(2, 4, 8, 8, generator=g), torch.randn(2, 5, 32, generator=g)
    a = sd_unet.forward(sd, cfg, x, torch.tensor([10, 900]), enc)
    b = sd_unet.forward(sd, cfg, x, torch.tensor([10, 900]), enc)
    assert a.shape == (2, 4, 8, 8) and torch.equal(a, b) and torch.isfinite(a).all()
    assert not torch.allclose(a[0], sd_unet.forward(sd, cfg, x, torch.tensor([900, 900]), enc)[0])     # the timestep matters


def test_oracle_sd_input_vjp_and_gradient_guidance_on_cpu():
    """The oracle's input gradient (torch autograd through the restatement) agrees with a central finite difference along a
    random direction (float64), and the gradient-mode guidance restatement (uncertainty_guidance.py:83-108,121-129) only moves
    pixels inside the percentile mask."""
    from diffusion_uncertainty_amd.utils import seeded_state_dict
    from oracle import guidance, sd_unet
    cfg = dict(in_channels=4, out_channels=4, block_out_channels=(32, 64), layers_per_block=1,
               down_block_types=("CrossAttnDownBlock2D", "DownBlock2D"), up_block_types=("UpBlock2D", "CrossAttnUpBlock2D"),
               attention_head_dim=4, cross_attention_dim=32, norm_num_groups=32, norm_eps=1e-5)
    sd = {k: v.double() for k, v in seeded_state_dict(sd_unet.state_shapes(cfg), 3).items()}
    g = torch.Generator().manual_seed(0)
    x, enc = torch.randn(2, 4, 8, 8, generator=g).double(), torch.randn(2, 5, 32, generator=g).double()
    gout, d = torch.randn(2, 4, 8, 8, generator=g).double(), torch.randn(2, 4, 8, 8, generator=g).double()
    t = torch.tensor([10, 900])
    _, grad = sd_unet.input_vjp(sd, cfg, x, t, enc, gout)
    h = 1e-5
    f = lambda z: (sd_unet.forward(sd, cfg, z, t, enc) * gout).sum().item()
    fd = (f(x + h * d) - f(x - h * d)) / (2 * h)
    assert abs(fd - (grad * d).sum().item()) < 1e-6 * max(1.0, abs(fd))
    sd32 = {k: v.float() for k, v in sd.items()}
    eps, lat = torch.randn(1, 4, 8, 8, generator=g), torch.randn(1, 4, 8, 8, generator=g)
    torch.manual_seed(1)
    final, grad, u, mask = guidance.guided_score_percentile_gradient(
        eps, torch.cat([lat] * 2), lambda xh, tt, yy: sd_unet.forward_autograd(sd32, cfg, xh, torch.full((2,), int(tt)), yy),
        torch.tensor(500), enc.float(), 7.5, 0.4, 0.9, 3, torch.randn_like, lr=0.5)
    assert final.shape == eps.shape and torch.isfinite(final).all() and grad.abs().max() > 0
    assert int(mask.sum()) == 26 and torch.equal(final[mask == 0], eps[mask == 0])      # 10 % of 256 pixels above the 0.9 quantile
    assert torch.allclose(final[mask == 1], (eps + 0.5 * grad)[mask == 1])


def test_precision_modes_pick_operand_formats():
    """Which operand format each GroupNorm in front of a matrix-core kernel writes, per precision mode (host logic only: no kernel
    runs)."""
    from diffusion_uncertainty_amd.guided_diffusion.unet_openai import UNetModel, _Ctx
    m = UNetModel(image_size=16, in_channels=3, model_channels=32, out_channels=6, num_res_blocks=1, attention_resolutions=(2,),
                  channel_mult=(1, 2), num_classes=10, num_head_channels=16, use_scale_shift_norm=True, resblock_updown=True)
    want = {"fp32": (False, False, False), "f16x3": ("f16x3", "f16x3", "f16x3"), "bf16x3": (True, True, True),
            "f16w": ("f16x3", "f16x3", "f16x3")}                 # (no conv shape given: the direct form's operand)
    for mode, (w16, w24, gemm) in want.items():
        m.precision = mode
        ctx = _Ctx(m)
        assert (ctx.conv_split(16), ctx.conv_split(24), ctx.gemm_split) == (w16, w24, gemm), mode
    # with a conv shape the Winograd operand of the mode: f16w is the ONE-product format on every call path (block ABI or not), never the
    # three-product one under the f16w name; the form rule is host logic, the shape test is the library's
    from diffusion_uncertainty_amd import ops
    orig = ops.conv3x3_wino_supported, ops.conv3x3_wino4_supported, ops.conv3x3_wino_pick
    try:
        ops.conv3x3_wino_supported = ops.conv3x3_wino4_supported = lambda *a: True
        ops.conv3x3_wino_pick = lambda *a: 4
        shape = (2, 16, 16, 64, 64)
        got = {}
        for mode in ("fp32", "f16x3", "f16w", "bf16x3"):
            m.precision = mode
            for abi in (True, False):
                m.use_block_abi = abi
                got[(mode, abi)] = _Ctx(m).conv_split(16, shape)
        assert got == {("fp32", True): "f32w4", ("fp32", False): "f32w4", ("f16x3", True): "f16x3w4", ("f16x3", False): "f16x3w4",
                       ("f16w", True): "f16w", ("f16w", False): "f16w", ("bf16x3", True): True, ("bf16x3", False): True}
    finally:
        ops.conv3x3_wino_supported, ops.conv3x3_wino4_supported, ops.conv3x3_wino_pick = orig
        m.use_block_abi = True
    m.precision = "int8"
    with pytest.raises(AssertionError):
        _Ctx(m)


def test_winograd_form_does_not_depend_on_the_batch():
    """with the default `winograd = True` the F(2,3) / F(4,3) choice of a layer is the library's whole-rounds rule evaluated at
    the model's planning batch (wino_plan_batch), so an image's arithmetic is the same whatever batch it runs in; the planning batch of
    128 reproduces the headline's per-shape pick (host logic only: the pick and the support predicates are host functions of the library)."""
    from diffusion_uncertainty_amd import ops
    from diffusion_uncertainty_amd.guided_diffusion.unet_openai import UNetModel, _Ctx
    m = UNetModel(image_size=16, in_channels=3, model_channels=32, out_channels=6, num_res_blocks=1, attention_resolutions=(2,),
                  channel_mult=(1, 2), num_classes=10, num_head_channels=16, use_scale_shift_norm=True, resblock_updown=True)
    m.precision = "f16x3"
    ctx = _Ctx(m)
    layers = [(128, 128, 256, 256), (128, 128, 512, 256), (64, 64, 256, 256), (32, 32, 512, 512), (16, 16, 768, 768), (16, 16, 1536, 768),
              (8, 8, 1024, 1024), (8, 8, 2048, 1024)]
    for H, W, cin, cout in layers:
        forms = {ctx.wino_form(N, H, W, cin, cout) for N in (1, 2, 5, 9, 128, 256, 640)}
        assert len(forms) == 1, (H, W, cin, cout, forms)
        assert forms.pop() == ops.conv3x3_wino_pick(128, H, W, cin, cout)
    # the rule itself does depend on the batch (that is why it is evaluated at a fixed one): 16^2 x 768 is F(2,3) at 128 images, F(4,3) at 640
    assert ops.conv3x3_wino_pick(128, 16, 16, 768, 768) == 2 and ops.conv3x3_wino_pick(640, 16, 16, 768, 768) == 4
    m.wino_plan_batch = 640
    assert {_Ctx(m).wino_form(N, 16, 16, 768, 768) for N in (1, 128, 640)} == {4}


def test_winograd_f23_along_y_identity():
    """The algebra csrc/conv_wino_f16x3.hip implements, restated in fp64 torch: with V = (d0-d2, d1+d2, d2-d1, d1-d3) over the input rows
    2j-1 .. 2j+2 (zero outside the image), U = (g0, (g0+g1+g2)/2, (g0-g1+g2)/2, g2) over the kernel rows and M_f = conv1x3(V_f, U_f) with
    zero padding in x, the output rows are M0+M1+M2 (even) and M1-M2-M3 (odd) - exactly conv3x3 with padding 1; |V| <= 2 max|x|, which is
    why the V planes carry half the activation planes' power-of-two scale."""
    import torch
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(0)
    for (N, H, W, Cin, Cout) in [(2, 8, 8, 5, 7), (1, 6, 16, 3, 4), (3, 2, 4, 2, 2)]:
        x = torch.randn(N, Cin, H, W, generator=g, dtype=torch.float64)
        w = torch.randn(Cout, Cin, 3, 3, generator=g, dtype=torch.float64)
        ref = F.conv2d(x, w, padding=1)
        xp = F.pad(x, (0, 0, 1, 1))
        d0, d1, d2, d3 = xp[:, :, 0:H:2], xp[:, :, 1:H + 1:2], xp[:, :, 2:H + 2:2], xp[:, :, 3:H + 3:2]
        V = [d0 - d2, d1 + d2, d2 - d1, d1 - d3]
        g0, g1, g2 = w[:, :, 0], w[:, :, 1], w[:, :, 2]
        U = [g0, (g0 + g1 + g2) / 2, (g0 - g1 + g2) / 2, g2]
        M = [F.conv2d(V[f], U[f][:, :, None, :], padding=(0, 1)) for f in range(4)]
        out = torch.empty_like(ref)
        out[:, :, 0::2] = M[0] + M[1] + M[2]
        out[:, :, 1::2] = M[1] - M[2] - M[3]
        assert (out - ref).abs().max().item() < 1e-12
        assert max(v.abs().max().item() for v in V) <= 2 * x.abs().max().item() + 1e-12
