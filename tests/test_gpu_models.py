"""Model-, scheduler- and driver-level parity of the HIP path against the CPU oracle and the committed
golden vectors.  Tolerance: the north star's 1e-3 fp32 per pixel (written below; observed far tighter)."""
import argparse

import numpy as np
import pytest
import torch

from conftest import T
from tests_support import adm_state_shapes, uvit_state_shapes

pytestmark = pytest.mark.gpu
dev = "cuda"
TOL = 1e-3

TINY_ADM = dict(image_size=16, in_channels=3, model_channels=32, out_channels=6, num_res_blocks=1,
                attention_resolutions=(2,), dropout=0.0, channel_mult=(1, 2), num_classes=10,
                num_heads=2, num_head_channels=-1, num_heads_upsample=-1,
                use_scale_shift_norm=True, resblock_updown=True, use_new_attention_order=False)
TINY_UVIT = dict(img_size=8, patch_size=2, in_chans=4, embed_dim=64, depth=4, num_heads=4, num_classes=11, conv=False)


def build_adm(cfg, seed):
    from diffusion_uncertainty_amd.guided_diffusion.unet_openai import UNetModel
    from diffusion_uncertainty_amd.utils import seeded_state_dict
    m = UNetModel(**cfg)
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    assert shapes == {k: tuple(v) for k, v in adm_state_shapes(cfg).items()}       # checkpoint key/shape compatibility
    sd = seeded_state_dict(shapes, seed)
    m.load_state_dict(sd, strict=True)
    return m.eval().to(dev), sd


# (score abs, uncertainty rel, image LSB) at ~3x the observed end-of-chain deviations: fp32 4.8e-4 / 7e-3 / 1, bf16x3 4.4e-3 / 9e-2 / 1,
# (the random-weight chain amplifies a per-call deviation 50-100x in every mode)
# "f16x3" (the fp32-equivalent split, the bench headline) is held to the SAME row as exact fp32: no per-mode loosening.
CHAIN_TOL = {"fp32": (1e-3, 2.5e-2, 1), "f16x3": (1e-3, 2.5e-2, 1), "bf16x3": (1.5e-2, 0.3, 2)}


def test_tiny_adm_vs_golden(golden):
    g = golden("g5_adm_blocks")
    m, _ = build_adm(TINY_ADM, 5)
    o = m(T(g["tiny_x"]).to(dev), T(g["tiny_t"]).to(dev), y=T(g["tiny_y"]).to(dev))
    np.testing.assert_allclose(o.cpu().numpy(), g["tiny_out"], rtol=0, atol=TOL)
    assert np.abs(o.cpu().numpy() - g["tiny_out"]).max() < 1e-4
    o3 = m(T(g["tiny_x"]).to(dev), T(g["tiny_t"]).to(dev), y=T(g["tiny_y"]).to(dev), out_channels_keep=3)
    np.testing.assert_array_equal(o3.cpu().numpy(), o[:, :3].cpu().numpy())
    with pytest.raises(AssertionError):                      # reference tests/test_generation_adm_imagenet.py:47-81
        m(T(g["tiny_x"]).to(dev), T(g["tiny_t"]).to(dev))
    with pytest.raises(RuntimeError):                        # no CPU fallback
        m(T(g["tiny_x"]), T(g["tiny_t"]), y=T(g["tiny_y"]))


@pytest.mark.parametrize("name", ["adm64", "adm128"])
def test_full_size_adm_vs_golden(golden, name):
    """Full-size ADM-64 / ADM-128, B=1, seeded weights: HIP output vs the reference's own output."""
    from oracle import adm
    g = golden(f"g5_{name}_full")
    cfg = dict(adm.ADM64 if name == "adm64" else adm.ADM128)
    m, _ = build_adm(cfg, int(g["seed"]))
    assert sum(p.numel() for p in m.parameters()) == int(g["n_params"])
    o = m(T(g["x"]).to(dev), T(g["t"]).to(dev), y=T(g["y"]).to(dev))
    err = np.abs(o.cpu().numpy() - g["out"]).max()
    print(f"{name}: max|hip - reference| = {err:.3e} (|ref|max {np.abs(g['out']).max():.2f})")
    assert err < TOL


def test_full_size_adm64_split_k(golden):
    """UNetModel.split_k (f16x3, small batches): full-size ADM-64 at B=1 with the under-filled conv launches split along K, against
    the reference's own output and against the single-pass kernels; the HIP-graph replay of the split forward is bit-identical."""
    from oracle import adm
    g = golden("g5_adm64_full")
    m, _ = build_adm(dict(adm.ADM64), int(g["seed"]))
    m.precision = "f16x3"
    x, t, y = T(g["x"]).to(dev), T(g["t"]).to(dev), T(g["y"]).to(dev)
    single = m(x, t, y=y)
    m.split_k = True
    split = m(x, t, y=y)
    e_single, e_split = np.abs(single.cpu().numpy() - g["out"]).max(), np.abs(split.cpu().numpy() - g["out"]).max()
    print(f"adm64 f16x3: single-pass err {e_single:.3e}, split-K err {e_split:.3e}, differ: {not torch.equal(single, split)}")
    assert e_split < TOL and e_split < 2 * e_single + 1e-6 and not torch.equal(single, split)      # (the split path did run)
    m.enable_hip_graph()
    assert torch.equal(m(x, t, y=y), split) and torch.equal(m(x, t, y=y), split)
    m.split_k = False
    assert torch.equal(m(x, t, y=y), single)                          # another capture; the workspace registration was undone


def test_adm_batch_consistency():
    """Batched forward == per-sample forwards (ragged tile edges, B*HW not a multiple of the 128-row tile)."""
    m, _ = build_adm(TINY_ADM, 5)
    x = torch.randn(5, 3, 16, 16, generator=torch.Generator().manual_seed(3)).to(dev)
    t, y = torch.tensor([999, 10, 500, 0, 250]).to(dev), torch.tensor([0, 1, 2, 3, 4]).to(dev)
    full = m(x, t, y=y)
    for i in range(5):
        one = m(x[i:i + 1], t[i:i + 1], y=y[i:i + 1])
        np.testing.assert_allclose(one.cpu().numpy(), full[i:i + 1].cpu().numpy(), rtol=0, atol=2e-5)


def test_full_size_adm128_batch_consistency():
    """Full-size ADM-128 in the headline mode (f16x3, Winograd-form convs, block-level C ABI) at batch 9: sample i of the batched forward
    against its own batch-1 forward.  Tiles of the batched launches straddle images (9 x 128^2 positions are not a multiple of the
    persistent grid, the 8^2 level packs two images into one 128-position tile); every accumulation order is per output row, so the
    results must agree to fp32-rounding level - observed bit-identical, held to 2e-5."""
    from oracle import adm
    m, _ = build_adm(dict(adm.ADM128), 128)
    m.precision = "f16x3"
    assert m.winograd and m.use_block_abi
    B = 9
    g = torch.Generator().manual_seed(12)
    x = torch.randn(B, 3, 128, 128, generator=g).to(dev)
    t = torch.tensor([999, 980, 700, 500, 333, 200, 60, 20, 0]).to(dev)
    y = torch.randint(0, 1000, (B,), generator=g).to(dev)
    full = m(x, t, y=y)
    worst, same = 0.0, 0
    for i in (0, 3, 4, 8):
        one = m(x[i:i + 1].contiguous(), t[i:i + 1], y=y[i:i + 1])
        d = (one - full[i:i + 1]).abs().max().item()
        worst = max(worst, d)
        same += int(torch.equal(one, full[i:i + 1]))
    print(f"ADM-128 f16x3 batch 9 vs batch 1: max abs difference {worst:.3e} ({same} of 4 samples bit-identical); output scale {full.abs().max().item():.3f}")
    assert worst < 2e-5
    # the default evaluates the form rule at the model's planning batch, not at the batch of the call - every layer runs the
    # same form at batch 9 and at batch 1, and (every kernel's accumulation order being per output row) the results are the same bits;
    # a batch of 9 equals a batch of 5 followed by a batch of 4
    assert same == 4, "default dispatch: a sample's result must not depend on the batch it runs in"
    parts = torch.cat([m(x[:5].contiguous(), t[:5], y=y[:5]), m(x[5:].contiguous(), t[5:], y=y[5:])])
    assert torch.equal(parts, full)
    # ... and so does one form everywhere (winograd = 4: F(4,3) where its kernel takes the shape)
    m.winograd = 4
    full4 = m(x, t, y=y)
    for i in (0, 4, 8):
        assert torch.equal(m(x[i:i + 1].contiguous(), t[i:i + 1], y=y[i:i + 1]), full4[i:i + 1]), i
    assert (full4 - full).abs().max().item() < 2e-5
    del m
    torch.cuda.empty_cache()


def test_headline_dispatch_batch128(golden):
    """the dispatch the driver times, asserted as a whole.  Full-size ADM-128, f16x3, default form pick, block-level
    C ABI at B = 128 (the headline's batch): (1) bench.py's own post-timing check - samples 0, 77, 127 of the batched forward are the bits of
    their own batch-1 forwards; (2) G5's input (the reference's full-size ADM-128 case) in rows 0 and 127 of an otherwise random batch:
    both rows against the REFERENCE's output (1e-3, north star) and against each other bit for bit; (3) the forms the levels run in at
    this batch are the headline's (F(4,3) from 128^2 to 32^2, F(2,3) at 8^2)."""
    import importlib.util
    import os
    from oracle import adm
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod_gpu", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    g = golden("g5_adm128_full")
    m, _ = build_adm(dict(adm.ADM128), int(g["seed"]))
    m.precision = "f16x3"
    assert m.winograd is True and m.use_block_abi and m.wino_plan_batch == 128
    B = 128
    gen = torch.Generator().manual_seed(77)
    x = torch.randn(B, 3, 128, 128, generator=gen)
    y = torch.randint(0, 1000, (B,), generator=gen)
    chk = bench.dispatch_check(m, x.to(dev), y.to(dev), torch.device(dev))
    print("dispatch check:", {k: v for k, v in chk.items() if k != "what"})
    assert chk["ok"] and chk["bit_identical"] == 3 and chk["max_abs_diff"] == 0.0
    f = chk["conv3x3_forms"]
    assert f["128^2:256->256"] == "F(4,3)" and f["64^2:256->256"] == "F(4,3)" and f["32^2:512->512"] == "F(4,3)" and f["8^2:1024->1024"] == "F(2,3)"
    t = torch.linspace(999, 0, B).round().to(torch.int64)
    for row in (0, 127):
        x[row], t[row], y[row] = T(g["x"])[0], T(g["t"])[0], T(g["y"])[0]
    with torch.no_grad():
        o = m(x.to(dev), t.to(dev), y=y.to(dev))
    err = max(np.abs(o[row:row + 1].cpu().numpy() - g["out"]).max() for row in (0, 127))
    print(f"ADM-128 f16x3 at batch 128, rows 0 / 127 = G5's input: max|hip - reference| = {err:.3e}")
    assert err < TOL and torch.equal(o[0], o[127])
    del m
    torch.cuda.empty_cache()


def test_uvit_vs_golden(golden):
    from diffusion_uncertainty_amd.uvit.uvit import UViT
    from diffusion_uncertainty_amd.utils import seeded_state_dict
    g = golden("g6_uvit")
    m = UViT(**TINY_UVIT)
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    assert shapes == {k: tuple(v) for k, v in uvit_state_shapes(TINY_UVIT).items()}
    m.load_state_dict(seeded_state_dict(shapes, 6))
    m.eval().to(dev)
    o = m(T(g["tiny_x"]).to(dev), T(g["tiny_t"]).to(dev), T(g["tiny_y"]).to(dev))
    np.testing.assert_allclose(o.cpu().numpy(), g["tiny_out"], rtol=0, atol=1e-4)
    tok = T(g["block_in"]).to(dev).reshape(-1, 64).contiguous()
    o = m.in_blocks[0].run(m._lin, tok, 2, 18)
    np.testing.assert_allclose(o.cpu().reshape(2, 18, 64).numpy(), g["block_noskip"], rtol=0, atol=1e-4)
    o = m.out_blocks[0].run(m._lin, tok, 2, 18, T(g["block_skip_in"]).to(dev).reshape(-1, 64).contiguous())
    np.testing.assert_allclose(o.cpu().reshape(2, 18, 64).numpy(), g["block_skip"], rtol=0, atol=1e-4)


def test_uvit_h2_full_vs_golden(golden):
    from diffusion_uncertainty_amd.init_model import load_uvit
    from diffusion_uncertainty_amd.utils import seeded_state_dict
    g = golden("g6_uvit_h2_full")
    m = load_uvit(256, device="cpu", load_checkpoint=False)
    m.load_state_dict(seeded_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, int(g["seed"])))
    m.to(dev)
    o = m(T(g["x"]).to(dev), T(g["t"]).to(dev), T(g["y"]).to(dev))
    err = np.abs(o.cpu().numpy() - g["out"]).max()
    print(f"uvit-h/2: max|hip - reference| = {err:.3e}")
    assert err < TOL
    m.precision, m.split_k = "f16x3", True                 # one image = 258 token rows: every GEMM launch runs split along K
    o2 = m(T(g["x"]).to(dev), T(g["t"]).to(dev), T(g["y"]).to(dev))
    err2 = np.abs(o2.cpu().numpy() - g["out"]).max()
    m.split_k = False
    o3 = m(T(g["x"]).to(dev), T(g["t"]).to(dev), T(g["y"]).to(dev))
    print(f"uvit-h/2 f16x3: split-K {err2:.3e}, single-pass {np.abs(o3.cpu().numpy() - g['out']).max():.3e}, differ: {not torch.equal(o2, o3)}")
    assert err2 < TOL and not torch.equal(o2, o3)


def _sched(kind, model, **kw):
    from diffusion_uncertainty_amd.init_model import init_scheduler_imagenet128
    from diffusion_uncertainty_amd.schedulers_uncertainty.get_uncertainty_scheduler import get_uncertainty_scheduler
    args = argparse.Namespace(scheduler_type=kind, M=kw.get("M", 2), start_step_uc=4, num_steps_uc=3, predict_next=kw.get("predict_next", False),
                              eta=0.0, num_zigzag=kw.get("num_zigzag", 3), uncertainty_distance=1)
    return get_uncertainty_scheduler(args, None, model, init_scheduler_imagenet128())


def test_uc_estimators_vs_golden(golden):
    """Scheduler.step() on the GPU vs the reference's step() on the same seed (noise tape from the CPU generator)."""
    g = golden("g3_uc_estimators")
    x, y = T(g["x"]).to(dev), T(g["y"]).to(dev)
    m, _ = build_adm(TINY_ADM, 5)
    for c in range(int(g["n_cases"])):
        model_name, kind, M, nz, t, seed = g[f"c{c}_meta"]
        if model_name != "tiny_adm":
            continue
        if kind == "zigzag_centered":
            s = _sched("uncertainty_zigzag_centered", m, M=int(M), num_zigzag=int(nz))
        else:
            s = _sched("uncertainty_centered", m, M=int(M), predict_next=(kind == "centered_next"))
        s.set_timesteps(10)
        s.prompt_embeds = y
        eps = m(x, torch.full((2,), int(t), dtype=torch.long, device=dev), y=y, out_channels_keep=3)
        np.testing.assert_allclose(eps.cpu().numpy(), g[f"c{c}_eps"], rtol=0, atol=1e-4)
        torch.manual_seed(int(seed))
        o = s.step(T(g[f"c{c}_eps"]).to(dev), int(t), x)            # teacher-forced eps
        np.testing.assert_allclose(o.prev_sample.cpu().numpy(), g[f"c{c}_prev"], rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(o.uncertainty.cpu().numpy(), g[f"c{c}_unc"], rtol=1e-2, atol=1e-6)
        assert o.pred_epsilon is not None and "uncertainty" in o
        if kind == "zigzag_centered" and int(nz) > 1:
            # faithful mode executes the dead forwards and must give the identical map
            s.faithful_zigzag = True
            calls = s.denoiser_calls
            torch.manual_seed(int(seed))
            o2 = s.step(T(g[f"c{c}_eps"]).to(dev), int(t), x)
            assert s.denoiser_calls - calls == int(M) * int(nz)
            np.testing.assert_array_equal(o2.uncertainty.cpu().numpy(), o.uncertainty.cpu().numpy())


def test_step_outside_window_and_errors():
    m, _ = build_adm(TINY_ADM, 5)
    s = _sched("uncertainty_zigzag_centered", m)
    x = torch.zeros(2, 3, 16, 16, device=dev)
    with pytest.raises(ValueError):
        s.step(x, 900, x)                                          # set_timesteps not called
    s.set_timesteps(10)
    assert (s.timestep_after_step, s.timestep_end_step) == (500, 300)
    o = s.step(x, 900, x)
    assert "uncertainty" not in o and o.uncertainty is None
    assert s.step(x, 900, x, return_dict=False)[0].shape == x.shape
    with pytest.raises(IndexError):
        s.config.after_step = 9
        s.set_timesteps(10)
    with pytest.raises(NotImplementedError):
        _sched("flip_grad", m)                                     # needs denoiser backward


@pytest.mark.parametrize("kind", ["zigzag_centered", "centered"])
def test_driver_vs_golden(golden, kind):
    from diffusion_uncertainty_amd.generate_samples import generate_samples_model_scheduler_class_conditioned_from_tensor as gen
    g = golden("g4_driver")
    m, _ = build_adm(TINY_ADM, 5)
    s = _sched("uncertainty_" + kind, m)
    s.set_timesteps(10)
    torch.manual_seed(0)
    r = gen(T(g["X_T"]), T(g["y"]), 3, torch.device(dev), m, s)       # bs=3 -> ragged last batch
    assert r["gen_images"].dtype == torch.uint8 and r["uncertainty"].shape == (4, 3, 3, 16, 16)
    np.testing.assert_allclose(r["score"].numpy(), g[f"{kind}_score"], rtol=0, atol=TOL)
    np.testing.assert_allclose(r["uncertainty"].numpy(), g[f"{kind}_uncertainty"], rtol=2e-2, atol=1e-5)
    d = np.abs(r["gen_images"].numpy().astype(int) - g[f"{kind}_gen_images"].astype(int))
    assert d.max() <= 1 and (d > 0).mean() < 2e-2      # +-1 LSB where a pixel sits on a rounding boundary
    with pytest.raises(RuntimeError):
        gen(T(g["X_T"]), T(g["y"]), 3, torch.device("cpu"), m, s)


@pytest.mark.parametrize("precision", ["fp32", "f16x3", "bf16x3"])
def test_driver_split_modes_vs_golden(golden, precision):
    """The free-running 10-step guided chain (reference driver output G4) in every precision mode.  The 1e-3 bar is per denoiser
    call (test_bf16x3_mode_adm); over a 10-step chain of a random-weight model the per-call deviations compound 50-100x (the
    exact fp32 mode ends at 4.8e-4 from a 7e-6 per-call deviation), so this is a regression guard at ~3x the observed
    end-of-chain deviation (printed), not the parity bar."""
    from diffusion_uncertainty_amd.generate_samples import generate_samples_model_scheduler_class_conditioned_from_tensor as gen
    g = golden("g4_driver")
    m, _ = build_adm(TINY_ADM, 5)
    m.precision = precision
    s = _sched("uncertainty_zigzag_centered", m)
    s.set_timesteps(10)
    torch.manual_seed(0)
    r = gen(T(g["X_T"]), T(g["y"]), 3, torch.device(dev), m, s)
    es = np.abs(r["score"].numpy() - g["zigzag_centered_score"]).max()
    eu = (np.abs(r["uncertainty"].numpy() - g["zigzag_centered_uncertainty"]) / (np.abs(g["zigzag_centered_uncertainty"]) + 1e-3)).max()
    d = np.abs(r["gen_images"].numpy().astype(int) - g["zigzag_centered_gen_images"].astype(int))
    print(f"{precision}: chain score err {es:.2e}, uncertainty rel err {eu:.2e}, image LSB diffs {(d > 0).mean():.2e} (max {d.max()})")
    assert es < CHAIN_TOL[precision][0] and eu < CHAIN_TOL[precision][1]
    assert d.max() <= CHAIN_TOL[precision][2]


def test_mc_dropout_vs_golden(golden):
    from diffusion_uncertainty_amd.generate_samples import generate_samples_model_scheduler_class_conditioned_from_tensor as gen
    g = golden("g8_mc_dropout")
    m, _ = build_adm(dict(TINY_ADM, dropout=0.1), 5)
    s = _sched("mc_dropout", m, M=3)
    s.set_timesteps(10)
    s.prompt_embeds = T(g["y"]).to(dev)
    torch.manual_seed(808)
    o = s.step(T(g["eps"]).to(dev), 500, T(g["x"]).to(dev))
    assert not m.training
    np.testing.assert_allclose(o.prev_sample.cpu().numpy(), g["prev"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(o.uncertainty.cpu().numpy(), g["unc"], rtol=2e-2, atol=1e-6)
    s = _sched("mc_dropout", m, M=2)
    s.set_timesteps(10)
    torch.manual_seed(0)
    r = gen(T(g["drv_X_T"]), T(g["drv_y"]), 3, torch.device(dev), m, s)
    np.testing.assert_allclose(r["score"].numpy(), g["drv_score"], rtol=0, atol=TOL)
    np.testing.assert_allclose(r["uncertainty"].numpy(), g["drv_uncertainty"], rtol=5e-2, atol=1e-5)


def test_posterior_guidance_vs_golden(golden):
    import diffusion_uncertainty_amd.uncertainty_guidance as ug
    g = golden("g7_guidance")
    w, ctx = T(g["sd_w"]), T(g["sd_ctx"]).to(dev)

    class FakeCFGUNet(torch.nn.Module):            # third-party UNet stand-in: runs in torch on the GPU (not the product)
        def forward(self, sample, timestep, encoder_hidden_states, **kw):
            h = torch.nn.functional.conv2d(sample.cpu(), w, padding=1).to(dev)
            return (h + encoder_hidden_states.mean(dim=(1, 2))[:, None, None, None],)

    for batched in (True, False):              # the M denoiser calls as one call of batch 2M, or one by one as upstream
        ug.batch_mc_samples = batched
        for i, p in enumerate((0.85, 0.9, 0.95, 0.99)):
            torch.manual_seed(700 + i)
            fe = ug.get_uncertainty_guided_score_with_percentile(T(g["sd_eps"]).to(dev), T(g["sd_input"]).to(dev), torch.tensor([500]), ctx,
                                                                 FakeCFGUNet(), torch.tensor(0.3), p, "stable-diffusion",
                                                                 num_uncertainty_samples=5, guidance_scale=7.5)
            np.testing.assert_allclose(fe.cpu().numpy(), g[f"sd_final_{p}"], rtol=2e-4, atol=1e-5)
    ug.batch_mc_samples = True
    ug.use_posterior = False
    try:
        with pytest.raises(NotImplementedError):
            ug.get_uncertainty_guided_score_with_percentile(T(g["sd_eps"]).to(dev), T(g["sd_input"]).to(dev), torch.tensor([500]), ctx,
                                                            FakeCFGUNet(), torch.tensor(0.3), 0.9, "stable-diffusion", guidance_scale=7.5)
    finally:
        ug.use_posterior = True


def test_device_noise_mode():
    """Device Philox noise: reproducible, and statistically equivalent uncertainty maps (not bitwise vs host tape)."""
    from diffusion_uncertainty_amd.schedulers_uncertainty.noise import DeviceNoise
    m, _ = build_adm(TINY_ADM, 5)
    x = torch.randn(2, 3, 16, 16, generator=torch.Generator().manual_seed(1)).to(dev)
    y = torch.tensor([1, 2]).to(dev)
    outs = []
    for _ in range(2):
        s = _sched("uncertainty_zigzag_centered", m, M=4)
        s.noise_source = DeviceNoise(seed=7)
        s.set_timesteps(10)
        s.prompt_embeds = y
        eps = m(x, torch.full((2,), 500, dtype=torch.long, device=dev), y=y, out_channels_keep=3)
        outs.append(s.step(eps, 500, x).uncertainty.cpu())
    assert torch.equal(outs[0], outs[1]) and (outs[0] >= 0).all() and outs[0].mean() > 0


# ---------------------------------------------------------------- split-bf16 (bf16x3) precision mode
@pytest.mark.parametrize("precision", ["f16x3", "bf16x3"])
@pytest.mark.parametrize("name", ["tiny", "adm64", "adm128"])
def test_bf16x3_mode_adm(golden, name, precision):
    """Convs / weight GEMMs on the bf16 matrix cores with split fp32 operands (bf16x3) and on the fp32-equivalent fp16 split (f16x3):
    within the 1e-3 bar against the reference's fp32 CPU output."""
    from oracle import adm
    if name == "tiny":
        g = golden("g5_adm_blocks")
        m, _ = build_adm(TINY_ADM, 5)
        x, t, y, ref = g["tiny_x"], g["tiny_t"], g["tiny_y"], g["tiny_out"]
    else:
        g = golden(f"g5_{name}_full")
        m, _ = build_adm(dict(adm.ADM64 if name == "adm64" else adm.ADM128), int(g["seed"]))
        x, t, y, ref = g["x"], g["t"], g["y"], g["out"]
    m.precision = precision
    o = m(T(x).to(dev), T(t).to(dev), y=T(y).to(dev))
    err = np.abs(o.cpu().numpy() - ref).max()
    print(f"{precision} {name}: max|hip - reference| = {err:.3e} (|ref|max {np.abs(ref).max():.2f})")
    assert err < TOL
    # observed 3e-5 (bf16x3): regression guards well inside the bar; f16x3 is held to 1.5e-5 (exact fp32: 7.4e-6)
    assert err < {"bf16x3": 1e-4, "f16x3": 1.5e-5}.get(precision, 2.5e-4)


@pytest.mark.parametrize("precision", ["f16x3", "bf16x3"])
def test_bf16x3_mode_uvit(golden, precision):
    from diffusion_uncertainty_amd.init_model import load_uvit
    from diffusion_uncertainty_amd.utils import seeded_state_dict
    g = golden("g6_uvit_h2_full")
    m = load_uvit(256, device="cpu", load_checkpoint=False)
    m.load_state_dict(seeded_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, int(g["seed"])))
    m.to(dev)
    m.precision = precision
    o = m(T(g["x"]).to(dev), T(g["t"]).to(dev), T(g["y"]).to(dev))
    err = np.abs(o.cpu().numpy() - g["out"]).max()
    print(f"{precision} uvit-h/2: max|hip - reference| = {err:.3e}")
    assert err < TOL and err < (2e-5 if precision == "f16x3" else 3e-4)       # f16x3: the fp32-equivalent bound (exact fp32 mode: 8.9e-6)
    if precision == "f16x3":               # batch whose token rows do not fill the 256-row tiles; tiny model with ragged shapes
        from diffusion_uncertainty_amd.uvit.uvit import UViT
        tm = UViT(**TINY_UVIT)
        tm.load_state_dict(seeded_state_dict({k: tuple(v.shape) for k, v in tm.state_dict().items()}, 6))
        tm.eval().to(dev)
        gt = golden("g6_uvit")
        tm.precision = "f16x3"
        ot = tm(T(gt["tiny_x"]).to(dev), T(gt["tiny_t"]).to(dev), T(gt["tiny_y"]).to(dev))
        np.testing.assert_allclose(ot.cpu().numpy(), gt["tiny_out"], rtol=0, atol=2e-5)


@pytest.mark.parametrize("precision", ["fp32", "f16x3", "bf16x3"])
def test_vae_decode_vs_golden(golden, precision):
    """KL-VAE decoder: full-size f8 decoder, 32x32 latents -> 256x256, vs the reference's output."""
    from diffusion_uncertainty_amd.uvit.autoencoder import FrozenAutoencoderKL, get_model
    from diffusion_uncertainty_amd.utils import seeded_state_dict
    g = golden("g9_vae_decode")
    tiny = dict(double_z=True, z_channels=4, resolution=32, in_channels=3, out_ch=3, ch=32, ch_mult=[1, 2], num_res_blocks=1,
                attn_resolutions=[], dropout=0.0)
    for name, m in (("tiny", FrozenAutoencoderKL(tiny, 4)), ("full", get_model(None))):
        sd = seeded_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, 9)
        assert len(sd) == int(g[f"{name}_nkeys"])
        m.load_state_dict(sd, strict=True)
        m.to(dev)
        m.precision = precision
        o = m.decode(T(g[f"{name}_z"]).to(dev))
        err = np.abs(o.cpu().numpy() - g[f"{name}_out"]).max()
        print(f"vae {name} {precision}: max|hip - reference| = {err:.3e} (|ref|max {np.abs(g[name + '_out']).max():.2f})")
        assert err < TOL
    with pytest.raises(RuntimeError):
        m.decode(T(g["tiny_z"]))


def test_uvit_driver_with_vae():
    """generate_samples_..._uvit_from_tensor end to end: tiny U-ViT + tiny VAE decode + uint8, vs the oracle."""
    from diffusion_uncertainty_amd.generate_samples import generate_samples_model_scheduler_class_conditioned_uvit_from_tensor as gen
    from diffusion_uncertainty_amd.init_model import load_uvit_scheduler
    from diffusion_uncertainty_amd.schedulers_uncertainty.get_uncertainty_scheduler import get_uncertainty_scheduler
    from diffusion_uncertainty_amd.uvit.autoencoder import FrozenAutoencoderKL
    from diffusion_uncertainty_amd.uvit.uvit import UViT
    from diffusion_uncertainty_amd.uvit.uvit_ae import UViTAE
    from diffusion_uncertainty_amd.utils import seeded_state_dict
    from oracle import driver, schedule, uvit as ouvit, vae as ovae
    ucfg = dict(TINY_UVIT)
    nn_ = UViT(**ucfg)
    usd = seeded_state_dict({k: tuple(v.shape) for k, v in nn_.state_dict().items()}, 6)
    nn_.load_state_dict(usd)
    tiny = dict(double_z=True, z_channels=4, resolution=32, in_channels=3, out_ch=3, ch=32, ch_mult=[1, 2], num_res_blocks=1,
                attn_resolutions=[], dropout=0.0)
    ae = FrozenAutoencoderKL(tiny, 4)
    asd = seeded_state_dict({k: tuple(v.shape) for k, v in ae.state_dict().items()}, 9)
    ae.load_state_dict(asd)
    model = UViTAE(nn_, ae).to(dev)
    args = argparse.Namespace(scheduler_type="uncertainty_zigzag_centered", M=2, start_step_uc=2, num_steps_uc=2, predict_next=True, eta=0.0,
                              num_zigzag=2, uncertainty_distance=1)
    s = get_uncertainty_scheduler(args, None, model, load_uvit_scheduler())
    s.set_timesteps(5)
    X_T = torch.randn(3, 4, 8, 8, generator=torch.Generator().manual_seed(2))
    y = torch.tensor([1, 5, 10])
    torch.manual_seed(0)
    r = gen(X_T, y, 2, model, s, device=dev)
    assert r["gen_images"].shape == (3, 3, 16, 16) and r["gen_images"].dtype == torch.uint8
    tab = schedule.Tables(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear")
    vcfg = dict(ovae.KL_F8, ch=32, ch_mult=(1, 2), num_res_blocks=1)
    torch.manual_seed(0)
    ref = driver.generate(X_T, y, 2, lambda x, t, yy, train=False: ouvit.forward(usd, ucfg, x, t, yy), tab, 5, "zigzag_centered", M=2,
                          after_step=2, num_steps_uc=2, num_zigzag=2, clip_sample=False, steps_offset=1,
                          decode=lambda z: ovae.decode(asd, vcfg, z), slice3=False)
    np.testing.assert_allclose(r["score"].numpy(), ref["score"].numpy(), rtol=0, atol=TOL)
    d = np.abs(r["gen_images"].numpy().astype(int) - ref["gen_images"].numpy().astype(int))
    assert d.max() <= 1 and (d > 0).mean() < 2e-2


@pytest.mark.parametrize("precision", ["fp32", "f16x3", "bf16x3"])
def test_fused_groupnorm_stats_equivalent(precision):
    """fuse_gn_stats (statistics from producer epilogues) vs the stand-alone statistics kernel: same model output."""
    m, _ = build_adm(TINY_ADM, 5)
    m.precision = precision
    x = torch.randn(3, 3, 16, 16, generator=torch.Generator().manual_seed(3)).to(dev)
    t, y = torch.tensor([999, 10, 500]).to(dev), torch.tensor([0, 1, 2]).to(dev)
    a = m(x, t, y=y)
    m.fuse_gn_stats = False
    b = m(x, t, y=y)
    np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=0, atol=2e-5 if precision in ("fp32", "f16x3") else 2e-4)


class _StubDenoiser(torch.nn.Module):
    """The stand-in denoiser of the golden generator (make_golden.py StubDenoiser)."""

    def forward(self, x, t, y=None):
        b = (t.float() / 1000.0)[:, None, None, None] + 0.01 * y.float()[:, None, None, None]
        h = 0.7 * x + 0.1 * torch.roll(x, 1, dims=-1) + b
        return torch.cat([h, -h], dim=1)


def test_threshold_map_bit_exact(golden):
    from diffusion_uncertainty_amd.pipeline_uncertainty import \
        pipeline_sampler_class_conditional_uncertainty_guided_posterior_distribution as pp
    g = golden("g10_posterior_pipeline")
    u, thr = T(g["tm_u"]).to(dev), T(g["tm_thr"]).to(dev)
    for c in range(int(g["tm_n"])):
        tt, th = g[f"tm{c}_meta"]
        m = pp.calculate_threshold_map(thr if th == "tensor" else float(th), 2, u, str(tt))
        assert m.dtype == torch.float32
        np.testing.assert_array_equal(m.cpu().numpy(), g[f"tm{c}_map"])
    for tt in ("higher", "lower"):
        np.testing.assert_array_equal(pp.calculate_threshold_map(thr, 4, u[1], tt).cpu().numpy(), g[f"tm3d_{tt}"])
    with pytest.raises(RuntimeError):
        pp.calculate_threshold_map(thr[:, :, :4], 2, u, "higher")


def test_posterior_score_update_vs_golden(golden):
    from diffusion_uncertainty_amd.pipeline_uncertainty import \
        pipeline_sampler_class_conditional_uncertainty_guided_posterior_distribution as pp
    from oracle import schedule
    g = golden("g10_posterior_pipeline")
    x, y = T(g["x"]).to(dev), T(g["y"]).to(dev)
    m, _ = build_adm(TINY_ADM, 5)
    acp = schedule.Tables().alphas_cumprod
    for c in range(int(g["e_n"])):
        model_name, M, t, ai, seed = g[f"e{c}_meta"]
        model = _StubDenoiser() if model_name == "stub" else m
        tv = torch.full((2,), int(t), dtype=torch.long, device=dev)
        torch.manual_seed(int(seed))
        u, score = pp.estimate_score_update_posterior(int(M), model, None, x, y, tv, T(g[f"e{c}_eps"]).to(dev), x, acp[int(ai)])
        np.testing.assert_allclose(u.cpu().numpy(), g[f"e{c}_u"], rtol=2e-2, atol=1e-7)
        # new_score = prec * (inv_var * sum): its scale is |sum| / M; compare against that scale
        ref = g[f"e{c}_score"]
        assert np.abs(score.cpu().numpy() - ref).max() <= 2e-2 * np.abs(ref).max()
        assert np.median(np.abs(score.cpu().numpy() - ref) / (np.abs(ref) + 1e-6)) < 1e-3


def test_posterior_pipelines_vs_golden(golden):
    from diffusion_uncertainty_amd.init_model import init_scheduler_imagenet128
    from diffusion_uncertainty_amd.pipeline_uncertainty.pipeline_sampler_class_conditional import DiffusionClassConditional
    from diffusion_uncertainty_amd.pipeline_uncertainty.pipeline_sampler_class_conditional_uncertainty_guided_posterior_distribution import \
        DiffusionClassConditionalGuidedPosteriorDistribution as Guided
    from diffusion_uncertainty_amd.schedulers_uncertainty.scheduling_ddim import DDIMScheduler
    from oracle import adm as oadm, posterior_pipeline as opp, schedule
    g = golden("g10_posterior_pipeline")
    m, sd = build_adm(TINY_ADM, 5)
    X_T, y = T(g["X_T"]), T(g["yy"])
    s = DDIMScheduler.from_config(init_scheduler_imagenet128().config)
    s.set_timesteps(10)
    r = DiffusionClassConditional(m, s, 16, torch.device(dev), 3, 0)(X_T=X_T, y=y)
    # per-step parity is 1e-4 (first steps); rounding differences grow along the 10-step trajectory of the random-weight net
    np.testing.assert_allclose(r["score"].numpy()[:, :3], g["plain_score"][:, :3], rtol=0, atol=2e-4)
    np.testing.assert_allclose(r["score"].numpy(), g["plain_score"], rtol=0, atol=5e-3)
    d = np.abs(r["gen_images"].numpy().astype(int) - g["plain_gen_images"].astype(int))
    assert d.max() <= 1 and (d > 0).mean() < 2e-2
    assert set(r) == {"y", "x_t", "timestep", "gen_images", "score"}
    r2 = DiffusionClassConditional(m, s, 16, torch.device(dev), 3, 0).sample(X_T, y)
    np.testing.assert_array_equal(r2["gen_images"].numpy(), r["gen_images"].numpy())
    with pytest.raises(RuntimeError):
        DiffusionClassConditional(m, s, 16, torch.device("cpu"), 3, 0)
    # guided loop vs the reference run: same scheduler class as the fixture (uncertainty_centered, M=1) so that step()
    # consumes the CPU noise tape identically
    thr10 = T(g["thr10"])
    for name, th in (("p90", 0.9), ("tensor", thr10)):
        sc = _sched("uncertainty_centered", m, M=1)
        sc.set_timesteps(10)
        torch.manual_seed(7)
        out = Guided(m, sc, th, 16, torch.device(dev), 3, 0, M=2)(X_T=X_T, y=y, start_step=6, num_steps=2)
        ref = g[f"guided_{name}_gen_images"]
        d = np.abs(out["gen_images"].numpy().astype(int) - ref.astype(int))
        # The update blends where u > threshold: a pixel whose uncertainty sits within rounding of the threshold flips with ANY
        # fp32-rounding-level change of the denoiser output and then moves by a visible amount, which the remaining steps spread.
        # Measured on this fixture over four arithmetic variants of the exact fp32 mode (3x3 convs direct / Winograd, `out` conv
        # implicit-GEMM / tap GEMM, all within 2e-6 of each other per call): 0.0000 ... 0.036 of the pixels beyond 1 LSB.  The bar
        # therefore sits at 5e-2 (a wrong update moves ~all pixels of the window); per-call parity is pinned elsewhere (G5, G7).
        assert (d > 1).mean() < 5e-2, (name, (d > 1).mean(), d.max())
        assert set(out) == {"y", "x_t", "timestep", "gen_images"}
    # plain DDIM scheduler (what the reference script passes) vs the oracle loop on the same noise tape
    tab, ts = schedule.Tables(), schedule.leading_timesteps(1000, 10)

    def den(x, t, yb):
        return oadm.forward(sd, TINY_ADM, x, torch.full((x.shape[0],), int(t), dtype=torch.long), yb)[:, :3]

    torch.manual_seed(11)
    want = opp.sample_posterior_guided(tab, den, X_T, y, 3, 10, ts, 0.9, M=2, start_step=6, num_steps=2, threshold_type="lower")
    s.set_timesteps(10)
    torch.manual_seed(11)
    out = Guided(m, s, 0.9, 16, torch.device(dev), 3, 0, M=2, threshold_type="lower")(X_T=X_T, y=y, start_step=6, num_steps=2)
    d = np.abs(out["gen_images"].numpy().astype(int) - want.numpy().astype(int))
    assert (d > 1).mean() < 5e-2, ((d > 1).mean(), d.max())
    # ---- deterministic gate: the loose bars above are smoke checks - a pixel within rounding of the threshold flips with any
    # fp32-level change.  Here the FIRST guided step is teacher-forced from the oracle's own state (same noise tape) and the pixels
    # whose uncertainty lies within 1e-3 (relative) of the per-sample quantile are excluded: everywhere else the threshold map must
    # agree EXACTLY and the blended score to the per-call bars of test_posterior_score_update_vs_golden.
    from oracle import ddim as oddim
    from diffusion_uncertainty_amd import ops
    from diffusion_uncertainty_amd.generate_samples import predict_model
    from diffusion_uncertainty_amd.pipeline_uncertainty import \
        pipeline_sampler_class_conditional_uncertainty_guided_posterior_distribution as pp
    x, yb = X_T[:3].clone(), y[:3]
    for t in ts[:6]:
        x, _, _ = oddim.ddim_update(tab, den(x, int(t), yb), int(t), x, 10)
    i, t = 6, int(ts[6])
    eps_ref = den(x, t, yb)
    torch.manual_seed(123)
    u_ref, post_ref = opp.score_update_posterior(2, lambda xh: den(xh, t, yb), x, eps_ref, tab.alphas_cumprod[i])
    m_ref = opp.threshold_map(0.9, i, u_ref, "higher")
    xd, yd, tv = x.to(dev), yb.to(dev), torch.full((3,), t, dtype=torch.long, device=dev)
    eps_hip = ops._f32c(predict_model(m, xd, tv, yd))
    assert (eps_hip.cpu() - eps_ref).abs().max().item() < 1e-4
    torch.manual_seed(123)
    u_hip, post_hip = pp.estimate_score_update_posterior(2, m, None, xd, yd, tv, eps_hip, xd, tab.alphas_cumprod[i])
    m_hip = pp.calculate_threshold_map(0.9, i, u_hip, "higher")
    q = torch.quantile(u_ref.flatten(1).float(), 0.9, dim=1).view(-1, 1, 1, 1)
    clear = (u_ref - q).abs() > 1e-3 * q
    assert clear.float().mean().item() > 0.99
    assert torch.equal(m_hip.cpu()[clear], m_ref[clear])
    blend_ref = post_ref * m_ref + eps_ref * (1 - m_ref)
    blend_hip = ops.map_blend(post_hip, eps_hip, m_hip).cpu()
    err = (blend_hip - blend_ref).abs()[clear].max().item()
    print(f"guided step, {100 * clear.float().mean().item():.2f} % of the pixels clear of the threshold: maps identical, blended score err {err:.2e} "
          f"(scale {blend_ref.abs().max().item():.2f})")
    assert err <= 2e-2 * blend_ref.abs().max().item()
    with pytest.raises(AssertionError):
        Guided(m, s, 1.5, 16, torch.device(dev), 3, 0)


DEEP_ADM = dict(TINY_ADM, channel_mult=(1, 1, 2, 2), num_res_blocks=3, attention_resolutions=(4,))


def test_more_estimators_vs_golden(golden):
    """uncertainty (activation-noise hooks) / uncertainty_image / uncertainty_centered_d: step() vs the reference's step()."""
    from diffusion_uncertainty_amd.init_model import init_scheduler_imagenet128
    from diffusion_uncertainty_amd.schedulers_uncertainty.get_uncertainty_scheduler import get_uncertainty_scheduler
    from diffusion_uncertainty_amd.generate_samples import generate_samples_model_scheduler_class_conditioned_from_tensor as gen
    g = golden("g11_more_estimators")
    x, y = T(g["x"]).to(dev), T(g["y"]).to(dev)
    m, _ = build_adm(DEEP_ADM, 6)

    def sched(kind, M, **kw):
        a = argparse.Namespace(scheduler_type=kind, M=M, start_step_uc=kw.get("start", 4), num_steps_uc=3,
                               predict_next=kw.get("predict_next", False), eta=0.0, num_zigzag=3,
                               uncertainty_distance=kw.get("d", 1))
        s = get_uncertainty_scheduler(a, None, m, init_scheduler_imagenet128())
        s.set_timesteps(10)
        return s

    for c in range(int(g["n_cases"])):
        kind, M, t, seed = g[f"c{c}_meta"]
        M, t = int(M), int(t)
        if kind == "hooks":
            s = sched("uncertainty", M)
        elif kind.startswith("image"):
            s = sched("uncertainty_image", M, predict_next=(kind == "image_next"))
        else:
            s = sched("uncertainty_centered_d", M, d=int(kind[-1]))
        s.prompt_embeds = y
        torch.manual_seed(int(seed))
        o = s.step(T(g[f"c{c}_eps"]).to(dev), t, x)                   # teacher-forced eps
        np.testing.assert_allclose(o.prev_sample.cpu().numpy(), g[f"c{c}_prev"], rtol=1e-6, atol=1e-6)
        u, ref = o.uncertainty.cpu().numpy(), g[f"c{c}_unc"]
        assert np.abs(u - ref).max() <= 5e-2 * ref.max(), (kind, np.abs(u - ref).max(), ref.max())
        assert np.median(np.abs(u - ref) / (ref + 1e-12)) < 2e-2, kind
        assert s.denoiser_calls == M and m.activation_noise is None      # hooks removed after the step
    s = sched("uncertainty_centered_d", 2, start=7, d=2)
    torch.manual_seed(0)
    r = gen(T(g["X_T"]), T(g["yy"]), 2, torch.device(dev), m, s)
    np.testing.assert_allclose(r["score"].numpy(), g["drv_score"], rtol=0, atol=5e-3)
    ref = g["drv_uncertainty"]
    assert np.abs(r["uncertainty"].numpy() - ref).max() <= 5e-2 * ref.max()
    d = np.abs(r["gen_images"].numpy().astype(int) - g["drv_gen_images"].astype(int))
    assert d.max() <= 1 and (d > 0).mean() < 2e-2


def test_dpm_solver_vs_golden(golden):
    """`dpm_2_uncertainty_centered` (multistep DPM-Solver++ + centred estimator): tables, a teacher-forced trajectory and a
    driver run against the reference's own class."""
    from diffusion_uncertainty_amd.init_model import init_scheduler_imagenet128, load_uvit_scheduler
    from diffusion_uncertainty_amd.schedulers_uncertainty.get_uncertainty_scheduler import get_uncertainty_scheduler
    from diffusion_uncertainty_amd.schedulers_uncertainty.mixin import SchedulerUncertaintyMixin
    from diffusion_uncertainty_amd.generate_samples import generate_samples_model_scheduler_class_conditioned_from_tensor as gen
    g = golden("g12_dpm_solver")
    m, _ = build_adm(TINY_ADM, 5)

    def sched(M, start, num, base=None):
        a = argparse.Namespace(scheduler_type="dpm_2_uncertainty_centered", M=M, start_step_uc=start, num_steps_uc=num,
                               predict_next=False, eta=0.0, num_zigzag=3, uncertainty_distance=1)
        return get_uncertainty_scheduler(a, None, m, base or init_scheduler_imagenet128())

    for name, n, base in (("adm128", 10, None), ("adm128", 20, None), ("uvit", 50, load_uvit_scheduler())):
        s = sched(2, 3, 4, base)
        s.set_timesteps(n)
        np.testing.assert_array_equal(s.timesteps.numpy(), g[f"tab_{name}_{n}_timesteps"])
        np.testing.assert_allclose(s.sigmas.numpy(), g[f"tab_{name}_{n}_sigmas"], rtol=1e-5, atol=0)
        assert (s.timestep_after_step, s.timestep_end_step) == tuple(g[f"tab_{name}_{n}_window"])
        assert isinstance(s, SchedulerUncertaintyMixin)
    x, y = T(g["x"]).to(dev), T(g["y"]).to(dev)
    s = sched(3, 5, 3)
    with pytest.raises(ValueError):
        s.step(x, 900, x)                                            # set_timesteps not called (:919-922)
    s.set_timesteps(10)
    s.prompt_embeds = y
    cur, seen = x, 0
    for i, t in enumerate(s.timesteps.tolist()):
        torch.manual_seed(4000 + i)
        o = s.step(T(g[f"traj{i}_eps"]).to(dev), t, cur)
        # schedule scalars are rebuilt on this host: an ulp of table difference is amplified by 1/alpha_t at high noise
        np.testing.assert_allclose(o.prev_sample.cpu().numpy(), g[f"traj{i}_prev"], rtol=2e-5, atol=2e-5)
        assert ("uncertainty" in o) == (f"traj{i}_unc" in g)
        if "uncertainty" in o:
            seen += 1
            np.testing.assert_allclose(o.pred_original_sample.cpu().numpy(), g[f"traj{i}_x0"], rtol=2e-5, atol=2e-5)
            np.testing.assert_allclose(o.pred_epsilon.cpu().numpy(), g[f"traj{i}_score"], rtol=2e-5, atol=2e-5)
            u, ref = o.uncertainty.cpu().numpy(), g[f"traj{i}_unc"]
            assert np.abs(u - ref).max() <= 2e-3 * ref.max(), (np.abs(u - ref).max(), ref.max())
        cur = T(g[f"traj{i}_prev"]).to(dev)
    assert seen == 3 and s.denoiser_calls == 9
    s = sched(2, 6, 3)
    s.set_timesteps(10)
    torch.manual_seed(0)
    r = gen(T(g["X_T"]), T(g["yy"]), 2, torch.device(dev), m, s)
    np.testing.assert_allclose(r["score"].numpy(), g["drv_score"], rtol=0, atol=TOL)
    ref = g["drv_uncertainty"]
    assert np.abs(r["uncertainty"].numpy() - ref).max() <= 1e-2 * ref.max()
    d = np.abs(r["gen_images"].numpy().astype(int) - g["drv_gen_images"].astype(int))
    assert d.max() <= 1 and (d > 0).mean() < 1e-2
    s2 = sched(2, 6, 3)
    s2.predict_next = True
    s2.set_timesteps(10)
    s2.prompt_embeds = T(g["yy"])[:2].to(dev)
    cur = T(g["X_T"])[:2].to(dev)
    with pytest.raises(NotImplementedError):                         # :955
        for t in s2.timesteps.tolist():
            cur = s2.step(torch.zeros_like(cur), t, cur).prev_sample


# ---------------------------------------------------------------- full-size chain (G13) and the round-2 fixtures
# per precision: (teacher-forced eps abs, free-running score abs, free-running uncertainty rel, image LSB, min mask agreement)
FULL_CHAIN_TOL = {"fp32": (1e-3, 1e-3, 2.5e-2, 1, 0.99), "f16x3": (1e-3, 1e-3, 2.5e-2, 1, 0.99),
                  "bf16x3": (1e-3, 5e-3, 0.3, 2, 0.97),
                  # "f16w": one fp16 product per tap in the Winograd convs (11-bit operands, the reference's own GPU arithmetic class): a side
                  # mode, NOT a parity mode - documented regression bars, reported by bench.py under other_precision_modes only
                  "f16w": (1.5e-2, 0.15, 1.0, 16, 0.98)}      # observed 6.0e-3 / 6.1e-2 / 0.39 / 7 / 99.75 %


@pytest.mark.parametrize("precision", ["fp32", "f16x3", "f16x3/F(4,3)", "bf16x3", "f16w"])
def test_full_size_chain_vs_golden(golden, precision):
    """Full-size ADM-128 through the driver, 10 DDIM steps, M=2, num_zigzag=3, window (6, 3), against the reference driver's own
    run (G13): teacher-forced per step (the fixture's x_t in, eps compared; scheduler.step on the fixture's eps with the same
    noise tape), free-running end to end, and the agreement of the percentile-0.9 mask of the final uncertainty map
    (uncertainty_guidance.py:112).  The fp32-equivalent modes (fp32, f16x3) are held to the north star's 1e-3 END TO END.
    "f16x3/F(4,3)": f16x3 with every 3x3 conv the F(4,3) kernel takes in that form (at this batch the default pick would choose F(2,3):
    the headline job's batch 128 runs F(4,3) at the 128^2 .. 32^2 levels) - same bars as f16x3."""
    from oracle import adm
    from diffusion_uncertainty_amd import ops
    from diffusion_uncertainty_amd.generate_samples import generate_samples_model_scheduler_class_conditioned_from_tensor as gen
    from diffusion_uncertainty_amd.init_model import init_scheduler_imagenet128
    from diffusion_uncertainty_amd.schedulers_uncertainty.get_uncertainty_scheduler import get_uncertainty_scheduler
    g = golden("g13_adm128_chain")
    m, _ = build_adm(dict(adm.ADM128), int(g["seed"]))
    if precision == "f16x3/F(4,3)":
        precision = "f16x3"
        m.winograd = 4
    m.precision = precision
    tol_tf, tol_s, tol_u, tol_lsb, tol_mask = FULL_CHAIN_TOL[precision]
    after, num = int(g["window"][0]), int(g["window"][1])
    a = argparse.Namespace(scheduler_type="uncertainty_zigzag_centered", M=2, start_step_uc=after, num_steps_uc=num, predict_next=True,
                           eta=0.0, num_zigzag=3, uncertainty_distance=1)
    s = get_uncertainty_scheduler(a, None, m, init_scheduler_imagenet128())
    s.set_timesteps(10)
    y = T(g["y"]).to(dev)
    s.prompt_embeds = y
    ts = [int(t) for t in g["timesteps"]]
    assert s.timesteps.tolist() == ts and (s.timestep_after_step, s.timestep_end_step) == (int(g["window"][2]), int(g["window"][3]))
    # ---- teacher-forced: every kept step from the reference's own x_t
    worst = 0.0
    for i in [int(v) for v in g["kept"]]:
        x = T(g[f"x{i}"]).to(dev)
        eps = m(x, torch.full((1,), ts[i], dtype=torch.long, device=dev), y=y, out_channels_keep=3)
        worst = max(worst, np.abs(eps.cpu().numpy() - g[f"eps{i}"]).max())
        if after <= i < after + num:
            torch.manual_seed(6000 + i)
            o = s.step(T(g[f"eps{i}"]).to(dev), ts[i], x)
            ref_u = g["uncertainty"][:, i - after]
            eu = np.abs(o.uncertainty.cpu().numpy() - ref_u).max() / np.abs(ref_u).max()
            assert eu < tol_u, (precision, i, eu)
            np.testing.assert_allclose(o.pred_epsilon.cpu().numpy(), g["score"][:, i - after], rtol=0, atol=1e-6)
    print(f"{precision}: teacher-forced max |eps - reference| over {len(g['kept'])} steps = {worst:.2e}")
    assert worst < tol_tf
    # ---- free-running end to end, same noise tape (the fixture re-seeds the CPU generator before every scheduler.step)
    s = get_uncertainty_scheduler(a, None, m, init_scheduler_imagenet128())
    s.set_timesteps(10)
    inner, k = s.step, [0]

    def seeded_step(*aa, **kw):
        torch.manual_seed(6000 + k[0])
        k[0] += 1
        return inner(*aa, **kw)

    s.step = seeded_step
    r = gen(T(g["X_T"]), T(g["y"]), 1, torch.device(dev), m, s)
    es = np.abs(r["score"].numpy() - g["score"]).max()
    eu = (np.abs(r["uncertainty"].numpy() - g["uncertainty"]) / (np.abs(g["uncertainty"]) + 1e-3 * np.abs(g["uncertainty"]).max())).max()
    d = np.abs(r["gen_images"].numpy().astype(int) - g["gen_images"].astype(int))
    mask, _q = ops.quantile_mask(r["uncertainty"][:, -1].contiguous().to(dev), 0.9)
    agree = (mask.cpu().numpy() == g["final_mask"]).mean()
    print(f"{precision}: end-to-end score err {es:.2e}, uncertainty rel err {eu:.2e}, image LSB diffs {(d > 0).mean():.2e} (max {d.max()}), "
          f"percentile-0.9 mask agreement {100 * agree:.3f} %")
    assert es < tol_s and eu < tol_u and d.max() <= tol_lsb and agree >= tol_mask


# BASELINE configs[1]'s own schedule against the reference's run of it (G20).  With SEEDED RANDOM weights the 50-step map is chaotic - the
# exact-fp32 engine itself is 2e-5 from the reference's CPU run at step 10, 8e-4 at step 25, 0.1 at step 40 (different summation orders
# only) - so the free-running part is held at steps 10 and 25, and the UC window is teacher-forced from the reference's own x_t:
# (x_t drift at step 10, at step 25, teacher-forced eps, score, uncertainty (relative)).  The same bars for the exact-fp32 engine and f16x3.
HEADLINE_CHAIN_TOL = {"fp32": (2e-4, 1e-2, 1e-3, 1e-3, 2.5e-2), "f16x3": (2e-4, 1e-2, 1e-3, 1e-3, 2.5e-2)}


@pytest.mark.parametrize("precision", ["fp32", "f16x3"])
def test_headline_schedule_chain_vs_golden(golden, precision):
    """Full-size ADM-128 through the driver on the headline job's OWN schedule - 50 DDIM steps, M=5, num_zigzag=3, UC window (40, 10) -
    for one image, against the reference driver's run of the same schedule (G20: reference UNet, scheduler and driver on the CPU, 200
    forwards; same seeded weights, same noise tape).  Free-running: x_t at steps 10 and 25 of the plain stretch (later steps are printed,
    not asserted: the map is chaotic with random weights).  Teacher-forced from the reference's x_t at the first and the last step of
    the window (timesteps of the real schedule, M=5, three zig-zag passes): the denoiser's eps, then scheduler.step on it with the
    reference's noise tape - score and uncertainty map against the reference's."""
    from oracle import adm
    from diffusion_uncertainty_amd.generate_samples import generate_samples_model_scheduler_class_conditioned_from_tensor as gen
    from diffusion_uncertainty_amd.init_model import init_scheduler_imagenet128
    from diffusion_uncertainty_amd.schedulers_uncertainty.get_uncertainty_scheduler import get_uncertainty_scheduler
    g = golden("g20_adm128_headline_schedule")
    m, _ = build_adm(dict(adm.ADM128), int(g["seed"]))
    m.precision = precision
    tol_x10, tol_x25, tol_tf, tol_s, tol_u = HEADLINE_CHAIN_TOL[precision]
    after, num = int(g["window"][0]), int(g["window"][1])
    a = argparse.Namespace(scheduler_type="uncertainty_zigzag_centered", M=5, start_step_uc=after, num_steps_uc=num, predict_next=True,
                           eta=0.0, num_zigzag=3, uncertainty_distance=1)
    ts = [int(t) for t in g["timesteps"]]
    y = T(g["y"]).to(dev)
    # ---- teacher-forced: first and last step of the window from the reference's own x_t
    s = get_uncertainty_scheduler(a, None, m, init_scheduler_imagenet128())
    s.set_timesteps(50)
    s.prompt_embeds = y
    assert s.timesteps.tolist() == ts and (s.timestep_after_step, s.timestep_end_step) == (int(g["window"][2]), int(g["window"][3]))
    for i in (40, 49):
        x = T(g[f"x{i}"]).to(dev)
        eps = m(x, torch.full((1,), ts[i], dtype=torch.long, device=dev), y=y, out_channels_keep=3)
        e_tf = np.abs(eps.cpu().numpy() - g["score"][:, i - after]).max()
        torch.manual_seed(int(g["step_seed0"]) + i)
        o = s.step(eps, ts[i], x)
        ref_u = g["uncertainty"][:, i - after]
        e_s = np.abs(o.pred_epsilon.cpu().numpy() - g["score"][:, i - after]).max()
        e_u = np.abs(o.uncertainty.cpu().numpy() - ref_u).max() / np.abs(ref_u).max()
        print(f"{precision}: headline schedule, step {i} (t = {ts[i]}) teacher-forced: eps {e_tf:.2e}, score {e_s:.2e}, uncertainty rel {e_u:.2e}")
        assert e_tf < tol_tf and e_s < tol_s and e_u < tol_u, (precision, i, e_tf, e_s, e_u)
    # ---- free-running, same noise tape
    s = get_uncertainty_scheduler(a, None, m, init_scheduler_imagenet128())
    s.set_timesteps(50)
    inner, k = s.step, [0]

    def seeded_step(*aa, **kw):
        torch.manual_seed(int(g["step_seed0"]) + k[0])
        k[0] += 1
        return inner(*aa, **kw)

    s.step = seeded_step
    seen = {}                                             # timestep -> x of the FIRST forward at it (the main pass of that step)

    def record(_mod, args, kwargs):
        t = int((args[1] if len(args) > 1 else kwargs["timesteps"]).flatten()[0])
        seen.setdefault(t, (args[0] if args else kwargs["x"]).detach().float().cpu().clone())

    h = m.register_forward_pre_hook(record, with_kwargs=True)
    try:
        r = gen(T(g["X_T"]), T(g["y"]), 1, torch.device(dev), m, s)
    finally:
        h.remove()
    drift = {}
    for i in [int(v) for v in g["kept"]]:
        ref = g[f"x{i}"]
        drift[i] = float(np.abs(seen[ts[i]].numpy() - ref).max() / np.abs(ref).max())
    d = np.abs(r["gen_images"].numpy().astype(int) - g["gen_images"].astype(int))
    print(f"{precision}: headline schedule, free-running: x_t drift by step {({k_: float(f'{v:.1e}') for k_, v in drift.items()})}, final image: "
          f"{100 * (d > 0).mean():.1f} % of the bytes differ (max {d.max()})")
    assert drift[10] < tol_x10 and drift[25] < tol_x25, (precision, drift)


@pytest.mark.parametrize("precision", ["fp32", "f16x3"])
def test_uvit_full_size_chain_vs_golden(golden, precision):
    """BASELINE configs[2] end to end at full size: U-ViT-H/2 + the full KL-VAE decoder through the latent driver, 10 DDIM steps,
    M=2, num_zigzag=3, window (6, 3), free-running against the REFERENCE driver's own run (G17: reference U-ViT, scheduler, driver and
    VAE; same seeded weights, same noise tape).  The fp32-equivalent modes are held to the north star's 1e-3 on the score maps."""
    from diffusion_uncertainty_amd import ops
    from diffusion_uncertainty_amd.generate_samples import generate_samples_model_scheduler_class_conditioned_uvit_from_tensor as gen
    from diffusion_uncertainty_amd.init_model import load_autoencoder_uvit, load_uvit, load_uvit_scheduler
    from diffusion_uncertainty_amd.schedulers_uncertainty.get_uncertainty_scheduler import get_uncertainty_scheduler
    from diffusion_uncertainty_amd.utils import seeded_state_dict
    from diffusion_uncertainty_amd.uvit.uvit_ae import UViTAE
    g = golden("g17_uvit_h2_chain")
    uvit = load_uvit(256, device="cpu", load_checkpoint=False)
    uvit.load_state_dict(seeded_state_dict({k: tuple(v.shape) for k, v in uvit.state_dict().items()}, int(g["uvit_seed"])))
    ae = load_autoencoder_uvit(device="cpu", load_checkpoint=False)
    ae.load_state_dict(seeded_state_dict({k: tuple(v.shape) for k, v in ae.state_dict().items()}, int(g["vae_seed"])), strict=True)
    uvit.precision = precision
    ae.precision = precision if precision in ("fp32", "f16x3") else "bf16x3"
    model = UViTAE(uvit, ae).to(dev)
    after, num = int(g["window"][0]), int(g["window"][1])
    a = argparse.Namespace(scheduler_type="uncertainty_zigzag_centered", M=2, start_step_uc=after, num_steps_uc=num, predict_next=True,
                           eta=0.0, num_zigzag=3, uncertainty_distance=1)
    s = get_uncertainty_scheduler(a, None, model, load_uvit_scheduler())
    s.set_timesteps(10)
    assert s.timesteps.tolist() == [int(t) for t in g["timesteps"]]
    assert (s.timestep_after_step, s.timestep_end_step) == (int(g["window"][2]), int(g["window"][3]))
    inner, k = s.step, [0]

    def seeded_step(*aa, **kw):
        torch.manual_seed(7000 + k[0])
        k[0] += 1
        return inner(*aa, **kw)

    s.step = seeded_step
    r = gen(T(g["X_T"]), T(g["y"]), 1, model, s, device=dev)
    es = np.abs(r["score"].numpy() - g["score"]).max()
    eu = (np.abs(r["uncertainty"].numpy() - g["uncertainty"]) / (np.abs(g["uncertainty"]) + 1e-3 * np.abs(g["uncertainty"]).max())).max()
    d = np.abs(r["gen_images"].numpy().astype(int) - g["gen_images"].astype(int))
    mask, _q = ops.quantile_mask(r["uncertainty"][:, -1].contiguous().to(dev), 0.9)
    agree = (mask.cpu().numpy() == g["final_mask"]).mean()
    print(f"uvit-h/2 chain {precision}: score err {es:.2e} (|score|max {np.abs(g['score']).max():.2f}), uncertainty rel err {eu:.2e}, "
          f"image LSB diffs {(d > 0).mean():.2e} (max {d.max()}), percentile-0.9 mask agreement {100 * agree:.3f} %")
    tol_s, tol_u, tol_lsb, tol_mask = {"fp32": (1e-3, 2.5e-2, 1, 0.99), "f16x3": (1e-3, 2.5e-2, 1, 0.99)}[precision]
    assert es < tol_s and eu < tol_u and d.max() <= tol_lsb and agree >= tol_mask


@pytest.mark.parametrize("precision", ["fp32", "f16x3"])
def test_headline_schedule_free_running_on_a_damped_network(golden, precision):
    """G22: the headline job's schedule (50 DDIM steps, M=5, num_zigzag=3, window (40, 10)) through the reference driver on a network
    that is NOT chaotic - the seeded weights with the layers the reference zero-initialises (the residual branches' last convs, the
    attention blocks' proj_out, the output conv) scaled by 0.1 - so that the free-running chain can be held END TO END against the
    reference's own run: x_t at steps 10 / 25 / 40 / 49, all ten score and uncertainty maps of the window, the final uint8 image, the
    percentile-0.9 mask of the last map.  The bars of the ten-step chain (G13), for fp32 and f16x3 alike."""
    from oracle import adm
    from diffusion_uncertainty_amd import ops
    from diffusion_uncertainty_amd.generate_samples import generate_samples_model_scheduler_class_conditioned_from_tensor as gen
    from diffusion_uncertainty_amd.init_model import init_scheduler_imagenet128
    from diffusion_uncertainty_amd.schedulers_uncertainty.get_uncertainty_scheduler import get_uncertainty_scheduler
    g = golden("g22_adm128_headline_schedule_damped")
    m, sd = build_adm(dict(adm.ADM128), int(g["seed"]))
    gain = float(g["gain"])
    m.load_state_dict({k: (v * gain if any(t in k for t in (".out_layers.3.", ".proj_out.", "out.2.")) else v) for k, v in sd.items()}, strict=True)
    m.precision = precision
    _tol_tf, tol_s, tol_u, tol_lsb, tol_mask = FULL_CHAIN_TOL[precision]
    after, num = int(g["window"][0]), int(g["window"][1])
    a = argparse.Namespace(scheduler_type="uncertainty_zigzag_centered", M=5, start_step_uc=after, num_steps_uc=num, predict_next=True,
                           eta=0.0, num_zigzag=3, uncertainty_distance=1)
    ts = [int(t) for t in g["timesteps"]]
    s = get_uncertainty_scheduler(a, None, m, init_scheduler_imagenet128())
    s.set_timesteps(50)
    assert s.timesteps.tolist() == ts and (s.timestep_after_step, s.timestep_end_step) == (int(g["window"][2]), int(g["window"][3]))
    inner, k, seed0 = s.step, [0], int(g["step_seed0"])

    def seeded_step(*aa, **kw):
        torch.manual_seed(seed0 + k[0])
        k[0] += 1
        return inner(*aa, **kw)

    s.step = seeded_step
    seen = {}

    def record(_mod, args, kwargs):
        t = int((args[1] if len(args) > 1 else kwargs["timesteps"]).flatten()[0])
        seen.setdefault(t, (args[0] if args else kwargs["x"]).detach().float().cpu().clone())

    h = m.register_forward_pre_hook(record, with_kwargs=True)
    try:
        r = gen(T(g["X_T"]), T(g["y"]), 1, torch.device(dev), m, s)
    finally:
        h.remove()
    drift = {}
    for i in [int(v) for v in g["kept"]]:
        ref = g[f"x{i}"]
        drift[i] = float(np.abs(seen[ts[i]].numpy() - ref).max() / np.abs(ref).max())
    es = np.abs(r["score"].numpy() - g["score"]).max()
    eu = (np.abs(r["uncertainty"].numpy() - g["uncertainty"]) / (np.abs(g["uncertainty"]) + 1e-3 * np.abs(g["uncertainty"]).max())).max()
    d = np.abs(r["gen_images"].numpy().astype(int) - g["gen_images"].astype(int))
    mask, _q = ops.quantile_mask(r["uncertainty"][:, -1].contiguous().to(dev), 0.9)
    agree = (mask.cpu().numpy() == g["final_mask"]).mean()
    print(f"{precision}: headline schedule on the damped network, free-running: x_t drift {({k_: float(f'{v:.1e}') for k_, v in drift.items()})}, "
          f"score err {es:.2e}, uncertainty rel err {eu:.2e}, image LSB diffs {(d > 0).mean():.2e} (max {d.max()}), mask agreement {100 * agree:.3f} %")
    assert max(drift.values()) < 1e-3 and es < tol_s and eu < tol_u and d.max() <= tol_lsb and agree >= tol_mask


@pytest.mark.parametrize("precision", ["fp32", "f16x3"])
def test_uvit_headline_schedule_chain_vs_golden(golden, precision):
    """BASELINE configs[2]'s own schedule for one image - full-size U-ViT-H/2, 50 DDIM steps, M=5, num_zigzag=3, UC window (40, 10) -
    against the reference latent driver's run of it (G21; 200 forwards, same seeded weights, same noise tape), like the ADM test above:
    free-running latents at steps 10, 25, 40 and 49 (observed 1e-6 throughout: unlike the ADM's, this random-weight chain is not
    chaotic), and the window's first and last step teacher-forced from the reference's own x_t (the real timesteps, M=5, three zig-zag passes): eps, score, uncertainty map."""
    from diffusion_uncertainty_amd.generate_samples import generate_samples_model_scheduler_class_conditioned_uvit_from_tensor as gen
    from diffusion_uncertainty_amd.init_model import load_autoencoder_uvit, load_uvit, load_uvit_scheduler
    from diffusion_uncertainty_amd.schedulers_uncertainty.get_uncertainty_scheduler import get_uncertainty_scheduler
    from diffusion_uncertainty_amd.utils import seeded_state_dict
    from diffusion_uncertainty_amd.uvit.uvit_ae import UViTAE
    g = golden("g21_uvit_h2_headline_schedule")
    uvit = load_uvit(256, device="cpu", load_checkpoint=False)
    uvit.load_state_dict(seeded_state_dict({k: tuple(v.shape) for k, v in uvit.state_dict().items()}, int(g["uvit_seed"])))
    ae = load_autoencoder_uvit(device="cpu", load_checkpoint=False)
    ae.load_state_dict(seeded_state_dict({k: tuple(v.shape) for k, v in ae.state_dict().items()}, int(g["vae_seed"])), strict=True)
    uvit.precision, ae.precision = precision, precision
    model = UViTAE(uvit, ae).to(dev)
    tol_x10, tol_x25, tol_tf, tol_s, tol_u = HEADLINE_CHAIN_TOL[precision]
    after, num = int(g["window"][0]), int(g["window"][1])
    a = argparse.Namespace(scheduler_type="uncertainty_zigzag_centered", M=5, start_step_uc=after, num_steps_uc=num, predict_next=True,
                           eta=0.0, num_zigzag=3, uncertainty_distance=1)
    ts = [int(t) for t in g["timesteps"]]
    y = T(g["y"]).to(dev)
    s = get_uncertainty_scheduler(a, None, model, load_uvit_scheduler())
    s.set_timesteps(50)
    s.prompt_embeds = y
    assert s.timesteps.tolist() == ts and (s.timestep_after_step, s.timestep_end_step) == (int(g["window"][2]), int(g["window"][3]))
    for i in (40, 49):
        x = T(g[f"x{i}"]).to(dev)
        eps = s.predict_model(x, ts[i])
        e_tf = np.abs(eps.cpu().numpy() - g["score"][:, i - after]).max()
        torch.manual_seed(9500 + i)
        o = s.step(eps, ts[i], x)
        ref_u = g["uncertainty"][:, i - after]
        e_s = np.abs(o.pred_epsilon.cpu().numpy() - g["score"][:, i - after]).max()
        e_u = np.abs(o.uncertainty.cpu().numpy() - ref_u).max() / np.abs(ref_u).max()
        print(f"uvit-h/2 {precision}: headline schedule, step {i} (t = {ts[i]}) teacher-forced: eps {e_tf:.2e}, score {e_s:.2e}, uncertainty rel {e_u:.2e}")
        assert e_tf < tol_tf and e_s < tol_s and e_u < tol_u, (precision, i, e_tf, e_s, e_u)
    s = get_uncertainty_scheduler(a, None, model, load_uvit_scheduler())
    s.set_timesteps(50)
    inner, k = s.step, [0]

    def seeded_step(*aa, **kw):
        torch.manual_seed(9500 + k[0])
        k[0] += 1
        return inner(*aa, **kw)

    s.step = seeded_step
    seen = {}

    def record(_mod, args, kwargs):
        t = int((args[1] if len(args) > 1 else kwargs["timesteps"]).flatten()[0])
        seen.setdefault(t, (args[0] if args else kwargs["x"]).detach().float().cpu().clone())

    h = uvit.register_forward_pre_hook(record, with_kwargs=True)
    try:
        gen(T(g["X_T"]), T(g["y"]), 1, model, s, device=dev)
    finally:
        h.remove()
    drift = {}
    for i in [int(v) for v in g["kept"]]:
        ref = g[f"x{i}"]
        drift[i] = float(np.abs(seen[ts[i]].numpy() - ref).max() / np.abs(ref).max())
    print(f"uvit-h/2 {precision}: headline schedule, free-running: x_t drift by step {({k_: float(f'{v:.1e}') for k_, v in drift.items()})}")
    # (this chain is not chaotic: the latents stay within 1e-6 of the reference's through all 50 steps, the window included - all four held)
    assert max(drift.values()) < 1e-4, (precision, drift)


@pytest.mark.parametrize("precision", ["fp32", "f16x3"])
@pytest.mark.parametrize("kind", ["uncertainty_zigzag_centered", "mc_dropout"])
def test_baseline_config0_vs_reference_driver(golden, kind, precision):
    """BASELINE configs[0] as it stands: full-size ADM-64, 10 DDIM steps, M=2, batch 4, UC window over all steps, the script's
    synthetic starting points and seed, against the REFERENCE driver's own CPU run (G18), free-running: zig-zag centred (70
    forwards per batch in the reference, dead ones skipped here) and mc_dropout (30 forwards, dropout masks from the CPU
    generator in the reference's order).  Also with the small-batch split-K path (f16x3)."""
    from oracle import adm
    from diffusion_uncertainty_amd.generate_samples import generate_samples_model_scheduler_class_conditioned_from_tensor as gen
    from diffusion_uncertainty_amd.init_model import init_scheduler_imagenet64
    from diffusion_uncertainty_amd.schedulers_uncertainty.get_uncertainty_scheduler import get_uncertainty_scheduler
    g = golden("g18_adm64_config0")
    tag = "zz" if kind == "uncertainty_zigzag_centered" else "mc"
    m, _ = build_adm(dict(adm.ADM64), int(g["seed"]))
    m.precision = precision
    a = argparse.Namespace(scheduler_type=kind, M=2, start_step_uc=0, num_steps_uc=10, predict_next=True, eta=0.0, num_zigzag=3,
                           uncertainty_distance=1)
    kept = [int(v) for v in g["kept"]]
    for split_k in ((False, True) if precision == "f16x3" else (False,)):
        m.split_k = split_k
        s = get_uncertainty_scheduler(a, None, m, init_scheduler_imagenet64())
        s.set_timesteps(10)
        assert s.timesteps.tolist() == [int(t) for t in g[f"{tag}_timesteps"]]
        torch.manual_seed(0)
        r = gen(T(g["X_T"]), T(g["y"]), 4, torch.device(dev), m, s)
        assert r["uncertainty"].shape == (4, 10, 3, 64, 64)
        es = np.abs(r["score"][:, kept].numpy() - g[f"{tag}_score"]).max()
        ref_u = g[f"{tag}_uncertainty"]
        eu = (np.abs(r["uncertainty"][:, kept].numpy() - ref_u) / (np.abs(ref_u) + 1e-3 * np.abs(ref_u).max())).max()
        em = np.abs(r["uncertainty"].mean(dim=(0, 2, 3, 4)).numpy() / g[f"{tag}_unc_step_mean"] - 1).max()
        d = np.abs(r["gen_images"].numpy().astype(int) - g[f"{tag}_gen_images"].astype(int))
        print(f"configs[0] {kind} {precision} split_k={split_k}: score err {es:.2e}, uncertainty rel err {eu:.2e} (per-step means {em:.2e}), "
              f"image LSB diffs {(d > 0).mean():.2e} (max {d.max()})")
        assert es < 1e-3 and eu < 5e-2 and em < 1e-3 and d.max() <= 1
    m.split_k = False


@pytest.mark.parametrize("precision", ["fp32", "f16x3", "f16w"])
def test_uc_step_m5_m10_full_size_vs_golden(golden, precision):
    """One uncertainty step of the headline scheduler at the Monte-Carlo counts of BASELINE configs[1] (M=5) and configs[3] (M=10) on
    full-size ADM-128 (50-step grid, window (40, 10), num_zigzag=3), against the reference scheduler's own step() (G19), same
    noise tape; one chain per denoiser call and the chains stacked (mc_group)."""
    from oracle import adm
    from diffusion_uncertainty_amd.init_model import init_scheduler_imagenet128
    from diffusion_uncertainty_amd.schedulers_uncertainty.get_uncertainty_scheduler import get_uncertainty_scheduler
    g = golden("g19_adm128_uc_step_m5_m10")
    m, _ = build_adm(dict(adm.ADM128), int(g["seed"]))
    m.precision = precision
    x, y = T(g["x"]).to(dev), T(g["y"]).to(dev)
    tol_eps, tol_u = {"fp32": (1e-3, 1e-2), "f16x3": (1e-3, 1e-2), "f16w": (4e-3, 6e-2)}[precision]      # f16w observed: 1.5e-3 / 2.5e-2
    for M in (5, 10):
        t = int(g[f"M{M}_t"])
        a = argparse.Namespace(scheduler_type="uncertainty_zigzag_centered", M=M, start_step_uc=40, num_steps_uc=10, predict_next=True,
                               eta=0.0, num_zigzag=3, uncertainty_distance=1)
        eps = m(x, torch.full((1,), t, dtype=torch.long, device=dev), y=y, out_channels_keep=3)
        e_eps = np.abs(eps.cpu().numpy() - g[f"M{M}_eps"]).max()
        assert e_eps < tol_eps
        for group in (1, M):
            s = get_uncertainty_scheduler(a, None, m, init_scheduler_imagenet128())
            s.set_timesteps(50)
            s.prompt_embeds = y
            s.mc_group = group
            assert s.timestep_after_step >= t >= s.timestep_end_step
            torch.manual_seed(1900 + M)
            o = s.step(T(g[f"M{M}_eps"]).to(dev), t, x)
            ref_u = g[f"M{M}_unc"]
            eu = (np.abs(o.uncertainty.cpu().numpy() - ref_u) / (np.abs(ref_u) + 1e-3 * np.abs(ref_u).max())).max()
            np.testing.assert_allclose(o.prev_sample.cpu().numpy(), g[f"M{M}_prev"], rtol=0, atol=1e-5)
            print(f"M={M} mc_group={group} {precision}: eps err {e_eps:.2e}, uncertainty rel err {eu:.2e}")
            assert eu < tol_u


def test_thresholds_vs_reference_script(golden):
    """a15: dua_pixelwise_order_statistic against the reference script's own load_uncertainty_datasets outputs (G14), bit for bit."""
    from diffusion_uncertainty_amd.thresholds import compute_threshold_pixel_wise
    g = golden("g14_thresholds")
    n = 0
    for c in range(int(g["n_cases"])):
        if g[f"c{c}_thr"].size == 0:
            continue
        thr = compute_threshold_pixel_wise(T(g[f"c{c}_u"]).to(dev), float(g[f"c{c}_meta"][3]))
        assert thr.dtype == torch.float16
        got, ref = thr.cpu().numpy(), g[f"c{c}_thr"]
        np.testing.assert_array_equal(got, ref)                      # the selected VALUE, exactly
        # bit for bit too, except where the k-th value is a zero tied between +0 and -0 samples (which of the tied samples an
        # unstable argsort returns is not defined; uncertainty maps are sums of squares and never hold -0)
        nz = ref != 0
        np.testing.assert_array_equal(got.view(np.uint16)[nz], ref.view(np.uint16)[nz])
        n += 1
    assert n >= 5


def test_zigzag_x0_branch_vs_golden(golden):
    """zigzag_centered with predict_next = False (set after construction, as upstream allows): x_t1 zig-zag, every forward live."""
    g = golden("g15_zigzag_x0")
    x, y = T(g["x"]).to(dev), T(g["y"]).to(dev)
    m, _ = build_adm(TINY_ADM, 5)
    for c in range(int(g["n_cases"])):
        M, nz, t, seed = [int(v) for v in g[f"c{c}_meta"]]
        s = _sched("uncertainty_zigzag_centered", m, M=M, num_zigzag=nz)
        s.predict_next = False
        s.set_timesteps(10)
        s.prompt_embeds = y
        torch.manual_seed(seed)
        o = s.step(T(g[f"c{c}_eps"]).to(dev), t, x)
        np.testing.assert_allclose(o.prev_sample.cpu().numpy(), g[f"c{c}_prev"], rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(o.uncertainty.cpu().numpy(), g[f"c{c}_unc"], rtol=1e-2, atol=1e-6)
        assert s.denoiser_calls == M * nz                  # nothing skipped on this branch


def test_inplace_weight_edit_repacks():
    """Packed-weight caches follow in-place edits of the parameters: the output after `p.mul_()` equals a freshly
    loaded model's, in a split mode whose packed planes would otherwise go stale; `.data` edits need invalidate_packed_weights()."""
    m, sd = build_adm(TINY_ADM, 5)
    m.precision = "f16x3"
    x = torch.randn(2, 3, 16, 16, generator=torch.Generator().manual_seed(3)).to(dev)
    t, y = torch.tensor([500, 10]).to(dev), torch.tensor([1, 2]).to(dev)
    a = m(x, t, y=y)
    name = "input_blocks.1.0.in_layers.2.weight"               # (a perturbation, not a rescaling: GroupNorm follows this conv)
    delta = torch.randn(sd[name].shape, generator=torch.Generator().manual_seed(9)) * 0.05
    with torch.no_grad():
        dict(m.named_parameters())[name].add_(delta.to(dev))
    b = m(x, t, y=y)
    sd2 = {k: v.clone() for k, v in sd.items()}
    sd2[name] = sd2[name] + delta
    m2, _ = build_adm(TINY_ADM, 5)
    m2.load_state_dict(sd2)
    m2.precision = "f16x3"
    c = m2(x, t, y=y)
    assert (a - b).abs().max().item() > 1e-3                      # the edit is visible ...
    np.testing.assert_allclose(b.cpu().numpy(), c.cpu().numpy(), rtol=0, atol=1e-6)     # ... and equals a fresh pack
    dict(m.named_parameters())[name].data.add_(delta.to(dev))      # invisible to the version counter
    m.invalidate_packed_weights()
    d = m(x, t, y=y)
    assert (d - b).abs().max().item() > 1e-3


def test_flip_estimator_vs_golden(golden):
    """`flip` (mirror-consistency) scheduler: step() on the GPU vs the reference's own unconditional class (G16), and the factory's
    class-conditioned variant end to end through the driver."""
    from diffusion_uncertainty_amd.init_model import init_scheduler_imagenet128
    from diffusion_uncertainty_amd.schedulers_uncertainty.scheduling_ddim_flip import DDIMSchedulerUncertaintyImagenet
    from diffusion_uncertainty_amd.generate_samples import generate_samples_model_scheduler_class_conditioned_from_tensor as gen
    g = golden("g16_flip")
    x = T(g["x"]).to(dev)
    m, _ = build_adm(dict(TINY_ADM, num_classes=None), 16)
    seen = 0
    for c in range(int(g["n_cases"])):
        name, t, seed = g[f"c{c}_meta"]
        if name != "tiny_adm_uncond":
            continue
        s = DDIMSchedulerUncertaintyImagenet.from_config(init_scheduler_imagenet128().config, after_step=4, num_steps_uc=3, unet=m)
        s.set_timesteps(10)
        eps = m(x, torch.full((2,), int(t), dtype=torch.long, device=dev))[:, :3]
        np.testing.assert_allclose(eps.cpu().numpy(), g[f"c{c}_eps"], rtol=0, atol=1e-4)
        torch.manual_seed(int(seed))
        o = s.step(T(g[f"c{c}_eps"]).to(dev), int(t), x)
        np.testing.assert_allclose(o.prev_sample.cpu().numpy(), g[f"c{c}_prev"], rtol=1e-6, atol=1e-6)
        assert not o.pred_original_sample.is_cuda and not o.score.is_cuda            # moved to the CPU inside step(), as upstream
        np.testing.assert_allclose(o.pred_original_sample.numpy(), g[f"c{c}_x0"], rtol=1e-6, atol=1e-6)
        assert (o.uncertainty is not None) == (f"c{c}_unc" in g)
        if o.uncertainty is not None:
            seen += 1
            ref = g[f"c{c}_unc"]
            assert np.abs(o.uncertainty.numpy() - ref).max() <= 2e-3 * ref.max()
    assert seen == 2
    mc, _ = build_adm(TINY_ADM, 5)
    sc = _sched("flip", mc)
    sc.set_timesteps(10)
    torch.manual_seed(0)
    r = gen(torch.randn(3, 3, 16, 16, generator=torch.Generator().manual_seed(1)), torch.tensor([1, 2, 3]), 2, torch.device(dev), mc, sc)
    assert r["uncertainty"].shape == (3, 3, 3, 16, 16) and r["score"].shape == (3, 3, 3, 16, 16) and sc.denoiser_calls == 6
    assert bool((r["uncertainty"] >= 0).all())


@pytest.mark.parametrize("precision", ["fp32", "f16x3", "f16w"])
def test_block_level_abi_bit_identical(precision):
    """dua_adm_resblock_fwd / dua_adm_attnblock_fwd (one C-ABI call per block) against the per-op composition in
    Python: same kernels in the same order, so the UNet output must be bit-identical - eval and train (dropout masks) mode,
    batch sizes that leave partial tiles, the up / down / channel-changing / attention blocks of the tiny ADM.  "f16w": the
    single-product Winograd conv on BOTH paths (the mode's arithmetic does not depend on the call path) - and it is
    not the three-product conv under another name: the f16w output differs from f16x3's."""
    from diffusion_uncertainty_amd.schedulers_uncertainty.noise import DeviceNoise
    m, _ = build_adm(dict(TINY_ADM, dropout=0.1), 5)
    m.precision = precision
    x = torch.randn(5, 3, 16, 16, generator=torch.Generator().manual_seed(3)).to(dev)
    t, y = torch.tensor([999, 10, 500, 0, 250]).to(dev), torch.tensor([0, 1, 2, 3, 4]).to(dev)
    outs = {}
    for train in (False, True):
        m.train(train)
        for blk in (True, False):
            m.use_block_abi = blk
            m.dropout_mask_source = DeviceNoise(seed=11).dropout_mask          # same mask stream for both paths
            outs[(train, blk)] = m(x, t, y=y)
        assert torch.equal(outs[(train, True)], outs[(train, False)]), (precision, train)
    assert not torch.equal(outs[(True, True)], outs[(False, True)])           # dropout was live
    if precision == "f16w":
        m.train(False)
        m.use_block_abi = False
        m.precision = "f16x3"
        assert not torch.equal(m(x, t, y=y), outs[(False, False)])
    # argument validation happens on the host, before any launch
    from diffusion_uncertainty_amd import _lib
    import ctypes
    a = _lib.ResBlockArgs()
    assert _lib.lib().dua_adm_resblock_fwd(ctypes.byref(a), None) == -1 and b"null" in _lib.lib().dua_last_error()


@pytest.mark.parametrize("precision", ["fp32", "f16x3"])
def test_grouped_mc_chains_match_sequential(golden, precision):
    """scheduler.mc_group: the M Monte-Carlo chains of a UC step as one stacked denoiser call.  Same noise tape, same per-sample
    arithmetic: the uncertainty map must match the chain-by-chain run to fp32 rounding and the reference's own step() (G3)."""
    g = golden("g3_uc_estimators")
    x, y = T(g["x"]).to(dev), T(g["y"]).to(dev)
    m, _ = build_adm(TINY_ADM, 5)
    m.precision = precision
    n = 0
    for c in range(int(g["n_cases"])):
        model_name, kind, M, nz, t, seed = g[f"c{c}_meta"]
        if model_name != "tiny_adm" or kind != "zigzag_centered":
            continue
        outs = []
        for group in (1, 2, 8):
            s = _sched("uncertainty_zigzag_centered", m, M=int(M), num_zigzag=int(nz))
            s.mc_group = group
            s.set_timesteps(10)
            s.prompt_embeds = y
            torch.manual_seed(int(seed))
            o = s.step(T(g[f"c{c}_eps"]).to(dev), int(t), x)
            assert s.denoiser_calls == int(M)
            outs.append(o.uncertainty.cpu().numpy())
            np.testing.assert_allclose(o.uncertainty.cpu().numpy(), g[f"c{c}_unc"], rtol=1e-2, atol=1e-6)
            np.testing.assert_allclose(o.prev_sample.cpu().numpy(), g[f"c{c}_prev"], rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(outs[1], outs[0], rtol=1e-4, atol=1e-7)
        np.testing.assert_allclose(outs[2], outs[0], rtol=1e-4, atol=1e-7)
        n += 1
    assert n >= 2


@pytest.mark.parametrize("precision", ["fp32", "f16x3"])
def test_uvit_block_abi_bit_identical(precision):
    """dua_uvit_block_fwd (one C-ABI call per U-ViT block, SURVEY 8(b) uvit_block_fwd) against the per-op composition: bit-identical."""
    from diffusion_uncertainty_amd.uvit.uvit import UViT
    from diffusion_uncertainty_amd.utils import seeded_state_dict
    m = UViT(**TINY_UVIT)
    m.load_state_dict(seeded_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, 6))
    m.eval().to(dev)
    m.precision = precision
    x = torch.randn(3, 4, 8, 8, generator=torch.Generator().manual_seed(4)).to(dev)
    t, y = torch.tensor([999, 3, 500]).to(dev), torch.tensor([10, 0, 5]).to(dev)
    m.use_block_abi = True
    a = m(x, t, y)
    m.use_block_abi = False
    b = m(x, t, y)
    assert torch.equal(a, b)
    assert float(a.abs().max()) > 0


@pytest.mark.parametrize("precision", ["fp32", "f16x3"])
def test_stream_split_bit_identical(precision):
    """UViT.stream_split / UNetModel.stream_split: the batch in chunks on HIP streams of their own equals the single-stream forward bit
    for bit - also on the very first call, when the chunks fill the shared packed-weight cache (the first chunk's stream is synchronised
    before the next chunk reads it) - and for ragged chunk sizes."""
    from diffusion_uncertainty_amd.uvit.uvit import UViT
    from diffusion_uncertainty_amd.utils import seeded_state_dict
    x = torch.randn(5, 4, 8, 8, generator=torch.Generator().manual_seed(4)).to(dev)
    t, y = torch.tensor([999, 3, 500, 40, 7]).to(dev), torch.tensor([10, 0, 5, 2, 9]).to(dev)
    outs = []
    for n in (2, 1, 2, 3):
        m = UViT(**TINY_UVIT)
        m.load_state_dict(seeded_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, 6))
        m.eval().to(dev)
        m.precision, m.stream_split = precision, n
        outs.append(m(x, t, y))                        # (fresh model: the split run packs the weights itself)
        outs.append(m(x, t, y))
    for o in outs[1:]:
        assert torch.equal(o, outs[0])
    a, _ = build_adm(TINY_ADM, 5)
    a.precision = precision
    xa = torch.randn(5, 3, 16, 16, generator=torch.Generator().manual_seed(1)).to(dev)
    ya = torch.tensor([0, 1, 2, 3, 4]).to(dev)
    a.stream_split = 2
    first = a(xa, t, y=ya, out_channels_keep=3)        # packs on the chunk streams
    a.stream_split = 1
    assert torch.equal(a(xa, t, y=ya, out_channels_keep=3), first)
    torch.cuda.synchronize()


@pytest.mark.parametrize("precision", ["fp32", "f16x3", "f16w"])
def test_adm_hip_graph_replay_bit_identical(precision):
    """UNetModel.enable_hip_graph(): the eval forward captured into a HIP graph and replayed equals the eager forward bit for
    bit, follows new inputs, a precision switch (new capture) and an in-place weight edit (captures dropped)."""
    m, _ = build_adm(TINY_ADM, 5)
    m.precision = precision
    xs = [torch.randn(3, 3, 16, 16, generator=torch.Generator().manual_seed(s)).to(dev) for s in (1, 2)]
    t, y = torch.tensor([999, 10, 500]).to(dev), torch.tensor([0, 1, 2]).to(dev)
    eager = [m(x, t, y=y, out_channels_keep=3) for x in xs]
    m.enable_hip_graph()
    for rep in range(2):
        for x, e in zip(xs, eager):
            assert torch.equal(m(x, t, y=y, out_channels_keep=3), e)
    assert len(m._graphs) == 1
    assert m(xs[0], t, y=y).shape == (3, 6, 16, 16) and len(m._graphs) == 2          # another kept-channel count: another capture
    with torch.no_grad():
        m.out[2].weight.mul_(2.0)
    after = m(xs[0], t, y=y, out_channels_keep=3)                  # the edit dropped the captures: re-captured with the new weights
    m.enable_hip_graph(False)
    assert torch.equal(m(xs[0], t, y=y, out_channels_keep=3), after) and not torch.equal(after, eager[0])
