"""f16x3 range check (dua_range_check / dua_range_check_read, csrc/common.h): the counters of the kernels that WRITE f16x3 operands.

The "fp32-equivalent" claim of the f16x3 mode rests on static power-of-two plane scales (64 for GroupNorm / SiLU / LayerNorm outputs, 16 for
raw residual-stream operands, 32 for the Winograd images): an operand beyond 65504 / scale saturates its hi half.  These tests plant such
operands and read the counts back, and run a model with checkpoint-like statistics (FiLM scale ~ +-5 as `1 + scale` of
unet_openai.py:249-253, GroupNorm gamma up to ~10 as nn.py:17-19, residual stream up to ~1e3) with the check on."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
dev = "cuda"


@pytest.fixture(scope="module")
def ops():
    from diffusion_uncertainty_amd import ops as o
    return o


def rnd(seed, *shape):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed))


def counted(ops, fn):
    prev = ops.range_check(True)
    try:
        fn()
        return ops.range_check_read()
    finally:
        ops.range_check(prev)


def test_range_counters_per_writer(ops):
    """Planted out-of-range / sub-normal operands are counted exactly by each writer; clean inputs count zero; off = no counting."""
    N, H, W, C = 2, 16, 16, 64
    st = torch.tensor([0.0, 1.0], device=dev).repeat(N * 32, 1).contiguous()
    one, zero = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    x = rnd(1, N, H, W, C).clamp(-8, 8)
    x[x.abs() < 1e-3] = 0.5                                       # no accidental sub-normals
    xb = x.clone()
    xb[0, 3, 4, 5] = 2000.0                                       # * 64 > 65504
    xb[1, 7, 7, 9] = -1500.0
    xb[1, 2, 2, 2] = 1e-7                                         # * 64 < 2^-14
    # GroupNorm-apply planes (identity statistics), 4- and 8-channel kernels
    assert counted(ops, lambda: ops.groupnorm_apply(x.to(dev), st, one, zero, act=0, split="f16x3")) == (0, 0)
    assert counted(ops, lambda: ops.groupnorm_apply(xb.to(dev), st, one, zero, act=0, split="f16x3")) == (2, 1)
    # Winograd V writer: the counts of the transformed images computed in torch
    xb[0, 3, 4, 5] = 5000.0                                       # (the Winograd images carry scale 32: |V| > 2047 saturates)
    sat, sub = counted(ops, lambda: ops.groupnorm_apply(xb.to(dev), st, one, zero, act=0, split="f16x3w"))
    a = torch.nn.functional.pad(xb.double(), (0, 0, 0, 0, 1, 1))
    d0, d1, d2, d3 = a[:, 0:H:2], a[:, 1:H + 1:2], a[:, 2:H + 2:2], a[:, 3:H + 3:2]
    v = torch.stack([d0 - d2, d1 + d2, d2 - d1, d1 - d3]).float() * 32.0
    assert sat == int((v.abs() > 65504).sum()) and sat > 0
    assert sub == int(((v.abs() > 0) & (v.abs() < 2.0 ** -14)).sum())
    # LayerNorm planes: rows with a huge gamma
    rows = rnd(2, 40, 64)
    g = torch.ones(64)
    g[10] = 5000.0
    sat, _ = counted(ops, lambda: ops.layernorm(rows.to(dev), g.to(dev), torch.zeros(64, device=dev), split="f16x3"))
    ref = torch.nn.functional.layer_norm(rows, (64,), g, torch.zeros(64)) * 64.0
    assert sat == int((ref.abs() > 65504).sum()) and sat > 0
    # on-the-fly split of a raw operand (1x1 skip convolution): scale 16
    a1 = rnd(3, 300, 64)
    a1[5, 6] = 6000.0
    w = rnd(4, 32, 64) / 8
    assert counted(ops, lambda: ops.gemm_f16x3(a1.to(dev), ops.pack_weight_f16x3(w.to(dev)), 32)) == (1, 0)
    # off: nothing is counted
    ops.range_check(True)
    ops.range_check(False)
    ops.groupnorm_apply(xb.to(dev), st, one, zero, act=0, split="f16x3")
    ops.range_check(True)
    assert ops.range_check_read() == (0, 0)
    ops.range_check(False)


def checkpoint_like_(m, t, y, gamma_max, film_gain, stream_gain, seed=3):
    """Give a seeded random model the statistics of a trained checkpoint: GroupNorm gammas in [0.3, gamma_max]; every ResBlock's FiLM
    projection rescaled so that, for THESE timesteps / labels, its `scale` half peaks at +-film_gain and its `shift` half at film_gain / 2
    (the embedding path is two Linear layers: evaluated here on the CPU); an input convolution that puts the residual stream at
    ~stream_gain at its peaks.  Returns the achieved FiLM peak."""
    import torch.nn.functional as F
    from oracle.adm import timestep_embedding          # (test infrastructure: the CPU restatement of nn.py:103-121)
    g = torch.Generator().manual_seed(seed)
    sd = {k: v.detach().float().cpu() for k, v in m.state_dict().items()}
    mc = sd["time_embed.0.weight"].shape[1]
    emb = timestep_embedding(t.cpu(), mc).float()
    emb = F.linear(F.silu(F.linear(emb, sd["time_embed.0.weight"], sd["time_embed.0.bias"])), sd["time_embed.2.weight"], sd["time_embed.2.bias"])
    emb = emb + sd["label_emb.weight"][y.cpu()]
    peak = 0.0
    with torch.no_grad():
        for name, p in m.named_parameters():
            if (name.endswith("in_layers.0.weight") or name.endswith("out_layers.0.weight") or name.endswith("norm.weight") or name == "out.0.weight"):
                p.copy_((0.3 + (gamma_max - 0.3) * torch.rand(p.shape, generator=g)).to(p.device))
            elif name.endswith("emb_layers.1.weight"):
                b = dict(m.named_parameters())[name[:-6] + "bias"]
                film = F.linear(F.silu(emb), p.detach().float().cpu(), b.detach().float().cpu())
                C = film.shape[1] // 2
                ks, kh = film_gain / film[:, :C].abs().max().item(), 0.5 * film_gain / film[:, C:].abs().max().item()
                p[:C].mul_(ks); b[:C].mul_(ks)
                p[C:].mul_(kh); b[C:].mul_(kh)
                peak = max(peak, (film[:, :C] * ks).abs().max().item())
            elif name == "input_blocks.0.0.weight":
                p.mul_(0.25 * stream_gain / max(1e-6, p.float().std().item() * math.sqrt(27.0)))      # std ~ gain / 4: peaks at ~gain
    m.invalidate_packed_weights()
    return peak


def test_model_with_checkpoint_like_statistics(ops):
    """A tiny ADM whose parameters have checkpoint-like magnitudes (FiLM scale ~ +-5, gamma up to 10, stream ~1e3): the f16x3 forward with
    the range check on counts ZERO saturated operands and stays at the parity modes' distance from the exact-fp32 engine.  The same model
    pushed beyond the static scales (gamma up to 150: activations past 1023): the check reports saturations and the output stays finite
    (never inf / NaN) - hi clamps at 65504 and lo carries the remainder, so an operand up to twice the range keeps ~11 bits instead of 22
    and beyond that it clips: a non-zero count means "use precision fp32" (INTEGRATION.md section 4)."""
    from test_gpu_models import TINY_ADM, build_adm
    m, _ = build_adm(dict(TINY_ADM), 5)
    x = rnd(7, 3, 3, 16, 16).to(dev)
    t, y = torch.tensor([900, 400, 20]).to(dev), torch.tensor([1, 5, 9]).to(dev)
    peak = checkpoint_like_(m, t, y, gamma_max=10.0, film_gain=5.0, stream_gain=1000.0)
    assert 4.9 < peak < 5.1
    m.precision = "fp32"
    ref = m(x, t, y=y)
    m.precision = "f16x3"
    out = None

    def run():
        nonlocal out
        out = m(x, t, y=y)

    sat, sub = counted(ops, run)
    scale = ref.abs().max().item()
    err = (out - ref).abs().max().item() / scale
    print(f"checkpoint-like tiny ADM: saturated {sat}, subnormal {sub}; |f16x3 - fp32| / max|fp32| = {err:.2e} (output scale {scale:.3g})")
    assert sat == 0
    assert err < 2e-5
    # far beyond the static scales
    checkpoint_like_(m, t, y, gamma_max=150.0, film_gain=5.0, stream_gain=1000.0, seed=4)
    m.precision = "fp32"
    ref = m(x, t, y=y)
    m.precision = "f16x3"
    sat, sub = counted(ops, run)
    scale = ref.abs().max().item()
    err = (out - ref).abs().max().item() / scale
    print(f"gamma up to 150: saturated {sat}, subnormal {sub}; |f16x3 - fp32| / max|fp32| = {err:.2e}")
    assert sat > 0 and torch.isfinite(out).all()
    # (a saturated hi half keeps ~11 bits up to twice the range, then clips - observed here: 0.36 of the output scale.  There is no bound to
    # assert once operands clip: the non-zero count IS the signal, and the remedy is precision "fp32")


# ---- the default-on check of a model's first forward (range_guard.py)
def test_first_forward_is_range_checked_once(ops):
    """f16x3: the first forward after a weight load runs with the counters on and leaves model.range_report; clean weights: no warning,
    no second check; the checked forward returns the bits of an unchecked one; a reload re-arms the check; fp32 never checks."""
    import warnings

    from diffusion_uncertainty_amd.range_guard import F16RangeWarning
    from diffusion_uncertainty_amd.utils import randomize_parameters_
    from diffusion_uncertainty_amd.uvit.uvit import UViT
    from test_gpu_models import TINY_ADM, TINY_UVIT, build_adm
    x = rnd(7, 3, 3, 16, 16).to(dev)
    t, y = torch.tensor([900, 400, 20]).to(dev), torch.tensor([1, 5, 9]).to(dev)
    m, sd = build_adm(dict(TINY_ADM), 5)
    m.precision = "f16x3"
    assert not m._range_checked and not ops.range_check_is_on()
    with warnings.catch_warnings():
        warnings.simplefilter("error", F16RangeWarning)
        a = m(x, t, y=y)
    assert m._range_checked and m.range_report["saturated"] == 0 and not ops.range_check_is_on()
    m.range_report = None
    b = m(x, t, y=y)                                                   # not checked again
    assert m.range_report is None and torch.equal(a, b)
    m.load_state_dict(sd)
    assert not m._range_checked
    m.auto_range_check = False
    c = m(x, t, y=y)
    assert not m._range_checked and torch.equal(a, c)
    m.precision = "fp32"
    m.auto_range_check = True
    m(x, t, y=y)
    assert not m._range_checked                                        # the exact-fp32 engines have no plane scales to leave
    # U-ViT: same contract, LayerNorm planes
    u = randomize_parameters_(UViT(**TINY_UVIT), 3).eval().to(dev)
    u.precision = "f16x3"
    xl = rnd(8, 3, 4, 8, 8).to(dev)
    with warnings.catch_warnings():
        warnings.simplefilter("error", F16RangeWarning)
        u(xl, t, y)
    assert u._range_checked and u.range_report["saturated"] == 0
    with torch.no_grad():
        u.in_blocks[1].norm2.weight.mul_(40000.0)                      # LayerNorm output * 64 beyond 65504
    u.load_state_dict(u.state_dict())
    with pytest.warns(F16RangeWarning, match=r"in_blocks\.1 \[Block\]"):
        u(xl, t, y)
    assert u.range_report["saturated"] > 0
    # KL-VAE decoder: same contract on decode()
    from diffusion_uncertainty_amd.uvit.autoencoder import FrozenAutoencoderKL
    tiny = dict(double_z=True, z_channels=4, resolution=32, in_channels=3, out_ch=3, ch=32, ch_mult=[1, 2], num_res_blocks=1, attn_resolutions=[], dropout=0.0)
    v = randomize_parameters_(FrozenAutoencoderKL(tiny, 4), 9).to(dev)
    v.precision = "f16x3"
    z = rnd(9, 2, 4, 16, 16).to(dev)
    with warnings.catch_warnings():
        warnings.simplefilter("error", F16RangeWarning)
        img = v.decode(z)
    assert v._range_checked and v.range_report["saturated"] == 0 and torch.equal(v.decode(z), img)
    with torch.no_grad():
        v.decoder.norm_out.weight.mul_(50000.0)
    v.load_state_dict(v.state_dict())
    with pytest.warns(F16RangeWarning, match="FrozenAutoencoderKL"):
        v.decode(z)


def test_first_forward_warns_on_a_saturating_groupnorm(ops):
    """One GroupNorm gamma scaled x2000 (activations ~ +-6000 against the 1023 the plane scale 64 leaves room for): the default path
    - no flag, no call of range_check - warns once, names the block and its layer classes, and still returns finite values."""
    from diffusion_uncertainty_amd.range_guard import F16RangeWarning
    from test_gpu_models import TINY_ADM, build_adm
    m, _ = build_adm(dict(TINY_ADM), 5)
    m.precision = "f16x3"
    with torch.no_grad():
        m.input_blocks[1][0].in_layers[0].weight.mul_(2000.0)
    x = rnd(7, 3, 3, 16, 16).to(dev)
    t, y = torch.tensor([900, 400, 20]).to(dev), torch.tensor([1, 5, 9]).to(dev)
    with pytest.warns(F16RangeWarning, match=r"input_blocks\.1 \[ResBlock\]") as rec:
        out = m(x, t, y=y)
    assert len([w for w in rec if issubclass(w.category, F16RangeWarning)]) == 1 and torch.isfinite(out).all()
    rep = m.range_report
    assert rep["saturated"] > 0 and rep["blocks"][0][0] == "input_blocks.1" and not ops.range_check_is_on()
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("error", F16RangeWarning)
        m(x, t, y=y)                                                   # once per weight load


def test_range_violation_policies(ops):
    """model.on_range_violation: "raise" turns the first-forward finding into an F16RangeError; "fp32" warns, switches the model to the
    fp32 engines and returns THEIR result for this forward (and every later one)."""
    from diffusion_uncertainty_amd.range_guard import F16RangeError, F16RangeWarning
    from test_gpu_models import TINY_ADM, build_adm
    x = rnd(7, 3, 3, 16, 16).to(dev)
    t, y = torch.tensor([900, 400, 20]).to(dev), torch.tensor([1, 5, 9]).to(dev)
    m, _ = build_adm(dict(TINY_ADM), 5)
    with torch.no_grad():
        m.input_blocks[1][0].in_layers[0].weight.mul_(2000.0)
    m.precision = "fp32"
    want = m(x, t, y=y)
    m.precision, m.on_range_violation = "f16x3", "raise"
    for _ in range(2):                                             # stays armed: not a one-shot
        with pytest.raises(F16RangeError, match=r"input_blocks\.1"):
            m(x, t, y=y)
    m.on_range_violation = "fp32"
    with pytest.warns(F16RangeWarning, match="switching this model to precision 'fp32'"):
        got = m(x, t, y=y)
    assert m.precision == "fp32" and torch.equal(got, want) and torch.equal(m(x, t, y=y), want) and not ops.range_check_is_on()


def test_default_cli_warns_on_saturating_weights(tmp_path, monkeypatch):
    """The caller script with its default flags (no --check-range): a model whose first GroupNorm gamma is x2000 triggers the warning."""
    import importlib.util
    import os

    from diffusion_uncertainty_amd.range_guard import F16RangeWarning
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("gen_cli", os.path.join(root, "scripts", "generate_dataset_score_uncertainty_imagenet.py"))
    cli = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cli)
    seeded = cli.randomize_parameters_

    def checkpoint_with_a_hot_layer(module, seed=0):
        seeded(module, seed)
        with torch.no_grad():
            module.input_blocks[1][0].in_layers[0].weight.mul_(2000.0)
        return module

    monkeypatch.setattr(cli, "randomize_parameters_", checkpoint_with_a_hot_layer)
    args = cli.parse_args(["--image-size", "64", "--scheduler-type", "uncertainty_zigzag_centered", "--model-type", "unet", "--start-index", "0",
                           "--num-samples", "2", "--batch-size", "2", "--generation-steps", "2", "-M", "1", "--start-step-uc", "1",
                           "--num-steps-uc", "1", "--random-weights", "--synthetic"])
    assert not args.check_range
    with pytest.warns(F16RangeWarning, match="input_blocks.1"):
        cli.generate_and_save(0, args, 2, tmp_path)
    assert (tmp_path / "gen_images_0.pth").exists()


@pytest.mark.parametrize("name", ["adm128", "adm64"])
def test_full_size_adm_with_checkpoint_like_magnitudes_vs_reference(golden, ops, name):
    """G23: full-size ADM-128 / ADM-64 whose weights carry checkpoint-like magnitudes - GroupNorm gammas up to 10, FiLM
    scale +-5, a residual stream peaking at 1.6e3 - 2e3 - against the REFERENCE's own forward on the same weights.  The default first-forward range check
    finds NO operand outside fp16's range (the static plane scales hold at these magnitudes) and the f16x3 output stays inside the north
    star's 1e-3 per pixel at an output scale of 18 - as close to the reference as the exact-fp32 engines."""
    import warnings

    from diffusion_uncertainty_amd.range_guard import F16RangeWarning
    from oracle import adm
    from test_gpu_models import build_adm
    from tests_support import checkpoint_like_state_dict
    from conftest import T
    g = golden(f"g23_{name}_checkpoint_like")
    m, sd = build_adm(dict(adm.ADM128 if name == "adm128" else adm.ADM64), int(g["seed"]))
    sd, _ = checkpoint_like_state_dict(sd, T(g["t"]), T(g["y"]), adm.timestep_embedding, float(g["gamma_max"]), float(g["film_gain"]),
                                       float(g["stream_gain"]), int(g["like_seed"]))
    m.load_state_dict(sd)
    m.to(dev)
    scale = float(np.abs(g["out"]).max())
    for precision in ("f16x3", "fp32"):
        m.precision = precision
        with warnings.catch_warnings():
            warnings.simplefilter("error", F16RangeWarning)
            out = m(T(g["x"]).to(dev), T(g["t"]).to(dev), y=T(g["y"]).to(dev))
        if precision == "f16x3":
            assert m._range_checked and m.range_report["saturated"] == 0, m.range_report
        err = np.abs(out.cpu().numpy() - g["out"]).max()
        print(f"checkpoint-like {name} {precision}: max|hip - reference| = {err:.3e} at output scale {scale:.1f} (stream max {float(g['stream_max'].max()):.0f})"
              + (f", range report {m.range_report['saturated']} saturated / {m.range_report['subnormal']} subnormal" if precision == "f16x3" else ""))
        assert err < 1e-3


def test_full_size_uvit_with_checkpoint_like_magnitudes_vs_reference(golden, ops):
    """G24: U-ViT-H/2 with LayerNorm gammas up to 3 and a 3 x positional table (token magnitudes up to ~34) against the
    REFERENCE's forward on the same weights: no operand of the LayerNorm / GELU / attention planes leaves fp16's range, and f16x3 is as
    close to the reference as the exact-fp32 engines."""
    import warnings

    from conftest import T
    from diffusion_uncertainty_amd.init_model import load_uvit
    from diffusion_uncertainty_amd.range_guard import F16RangeWarning
    from diffusion_uncertainty_amd.utils import seeded_state_dict
    from tests_support import uvit_checkpoint_like_state_dict
    g = golden("g24_uvit_h2_checkpoint_like")
    m = load_uvit(256, device="cpu", load_checkpoint=False)
    sd = seeded_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, int(g["seed"]))
    m.load_state_dict(uvit_checkpoint_like_state_dict(sd, float(g["gamma_max"]), float(g["qkv_gain"]), int(g["like_seed"])))
    m.to(dev)
    scale = float(np.abs(g["out"]).max())
    for precision in ("f16x3", "fp32"):
        m.precision = precision
        with warnings.catch_warnings():
            warnings.simplefilter("error", F16RangeWarning)
            out = m(T(g["x"]).to(dev), T(g["t"]).to(dev), T(g["y"]).to(dev))
        if precision == "f16x3":
            assert m._range_checked and m.range_report["saturated"] == 0, m.range_report
        err = np.abs(out.cpu().numpy() - g["out"]).max()
        print(f"checkpoint-like U-ViT-H/2 {precision}: max|hip - reference| = {err:.3e} at output scale {scale:.1f} (token max {float(g['token_max'].max()):.0f})")
        assert err < 1e-3
