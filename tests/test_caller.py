"""The caller of the hot path: scripts/generate_dataset_score_uncertainty_imagenet.py against the reference script
(/root/reference/scripts/generate_dataset_score_uncertainty_imagenet.py: import block :14-27, generate_and_save :33-93,
main :137-144) - the import surface on the CPU, ``generate_and_save`` itself under ``-m gpu``."""
import argparse
import ast
import importlib
import importlib.util
import json
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCRIPT = os.path.join(ROOT, "scripts", "generate_dataset_score_uncertainty_imagenet.py")
REF_SCRIPT = "/root/reference/scripts/generate_dataset_score_uncertainty_imagenet.py"

# the reference script's lines 14-27, as (module, symbol) pairs
REFERENCE_IMPORTS = [
    ("diffusion_uncertainty.utils", "load_config"),
    ("diffusion_uncertainty.uvit.autoencoder", "FrozenAutoencoderKL"),
    ("diffusion_uncertainty.uvit.load_pretrained_models", "load_autoencoder_uvit"),
    ("diffusion_uncertainty.uvit.load_pretrained_models", "load_uvit"),
    ("diffusion_uncertainty.uvit.load_pretrained_models", "load_uvit_scheduler"),
    ("diffusion_uncertainty.uvit.uvit", "UViT"),
    ("diffusion_uncertainty.uvit.uvit_ae", "UViTAE"),
    ("diffusion_uncertainty.schedulers_uncertainty.get_uncertainty_scheduler", "get_uncertainty_scheduler"),
    ("diffusion_uncertainty.generate_samples", "generate_samples_model_scheduler_class_conditioned"),
    ("diffusion_uncertainty.generate_samples", "generate_samples_model_scheduler_class_conditioned_from_tensor"),
    ("diffusion_uncertainty.generate_samples", "generate_samples_model_scheduler_class_conditioned_uvit"),
    ("diffusion_uncertainty.generate_samples", "generate_samples_model_scheduler_class_conditioned_uvit_from_tensor"),
    ("diffusion_uncertainty.init_model", "init_guided_diffusion_imagenet64_and_scheduler"),
    ("diffusion_uncertainty.init_model", "init_guided_diffusion_imagenet128_and_scheduler"),
    ("diffusion_uncertainty.paths", "CONFIG"),
    ("diffusion_uncertainty.paths", "DIFFUSION_STARTING_POINTS"),
    ("diffusion_uncertainty.paths", "RESULTS"),
]


@pytest.fixture
def aliased():
    """the package under the reference's name."""
    import diffusion_uncertainty_amd
    saved = {k: v for k, v in sys.modules.items() if k == "diffusion_uncertainty" or k.startswith("diffusion_uncertainty.")}
    meta = list(sys.meta_path)
    yield diffusion_uncertainty_amd.alias_as_reference()
    sys.meta_path[:] = meta
    for k in [k for k in sys.modules if k == "diffusion_uncertainty" or k.startswith("diffusion_uncertainty.")]:
        del sys.modules[k]
    sys.modules.update(saved)


def _cli():
    spec = importlib.util.spec_from_file_location("gen_cli", SCRIPT)
    cli = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cli)
    return cli


def test_reference_import_block_resolves(aliased):
    for module, symbol in REFERENCE_IMPORTS:
        m = importlib.import_module(module)
        assert hasattr(m, symbol), f"from {module} import {symbol}"
        # one copy of every module: the alias and the package name give the same object (isinstance checks hold across them)
        assert m is importlib.import_module(module.replace("diffusion_uncertainty", "diffusion_uncertainty_amd", 1))
    with pytest.raises(ModuleNotFoundError):
        importlib.import_module("diffusion_uncertainty.no_such_module")


@pytest.mark.skipif(not os.path.exists(REF_SCRIPT), reason="reference checkout not present (GPU box)")
def test_import_table_is_the_reference_scripts(aliased):
    """The table above is what the reference script imports from its own package: parsed from its text, nothing executed."""
    tree = ast.parse(open(REF_SCRIPT).read())
    pairs = [(n.module, a.name) for n in ast.walk(tree) if isinstance(n, ast.ImportFrom) and n.module
             and n.module.startswith("diffusion_uncertainty") for a in n.names]
    assert sorted(pairs) == sorted(REFERENCE_IMPORTS)


def test_paths_follow_the_reference_layout():
    from diffusion_uncertainty_amd import paths
    assert paths.RESULTS == paths.ROOT / "results"                                        # paths.py:12
    assert paths.DIFFUSION_STARTING_POINTS == paths.RESULTS / "diffusion_starting_points"  # paths.py:36
    assert paths.MODELS == paths.ROOT / "models"                                           # paths.py:172
    assert paths.SCORE_UNCERTAINTY == paths.RESULTS / "score-uncertainty" and paths.THRESHOLD == paths.RESULTS / "thresholds"
    files = paths.CONFIG.joinpath("generation").files("*.yaml")                            # the calls of script :174-175
    names = [x.basename().replace(".yaml", "") for x in files]
    assert "imagenet128_readme_example" in names
    assert not paths.RESULTS.joinpath("no-such-folder").files("*.yaml")


def test_load_config_semantics(tmp_path):
    from diffusion_uncertainty_amd.paths import CONFIG
    from diffusion_uncertainty_amd.utils import get_generation_configs, load_config
    a = load_config(CONFIG / "generation", "imagenet128_readme_example")
    b = load_config(CONFIG / "generation", "configs/imagenet128_readme_example.yaml")      # utils.py:53-56
    assert isinstance(a, argparse.Namespace) and a == b and (a.M, a.image_size, a.model_type) == (5, 128, "unet")
    assert "imagenet128_readme_example" in get_generation_configs("generation")
    with pytest.raises(FileNotFoundError):
        load_config(tmp_path, "missing")


def test_uvit_loaders_signatures_and_errors():
    from diffusion_uncertainty_amd import init_model
    from diffusion_uncertainty_amd.configuration import DDPMScheduler
    from diffusion_uncertainty_amd.uvit import load_pretrained_models as lpm
    assert init_model.load_uvit is lpm.load_uvit and init_model.load_autoencoder_uvit is lpm.load_autoencoder_uvit
    s = lpm.load_uvit_scheduler()
    assert isinstance(s, DDPMScheduler)
    assert (s.config.beta_start, s.config.beta_end, s.config.beta_schedule, s.config.steps_offset, s.config.clip_sample) == \
        (0.00085, 0.012, "scaled_linear", 1, False)                                        # load_pretrained_models.py:45-57
    with pytest.raises(TypeError):
        lpm.load_uvit(128, "cpu")
    with pytest.raises(FileNotFoundError, match="imagenet256_uvit_huge.pth"):              # no download: a named missing file
        lpm.load_uvit(256, "cpu")
    with pytest.raises(FileNotFoundError, match="autoencoder_kl_ema.pth"):
        lpm.load_autoencoder_uvit("cpu")


def test_rank_slices_and_spawn_arguments(monkeypatch, tmp_path):
    """main() with two visible GPUs: two processes, n = num_samples // 2 each, rank r on [start + r n, start + (r+1) n)
    (reference :51,137-144)."""
    cli = _cli()
    seen = {}

    def fake_spawn(fn, args, nprocs, join):
        seen.update(fn=fn, args=args, nprocs=nprocs, join=join)

    monkeypatch.setattr(torch.cuda, "device_count", lambda: 2)
    monkeypatch.setattr(torch.multiprocessing, "spawn", fake_spawn)
    monkeypatch.setattr(sys, "argv", ["x", "--image-size", "128", "--scheduler-type", "uncertainty_zigzag_centered", "--model-type", "unet",
                                      "--start-index", "10", "--num-samples", "301", "--dest", str(tmp_path / "out")])
    cli.main()
    assert seen["fn"] is cli.generate_and_save and seen["nprocs"] == 2 and seen["join"] is True
    args, n, dest = seen["args"]
    assert n == 150 and str(dest) == str(tmp_path / "out") and (tmp_path / "out" / "args.yaml").exists()
    r0, r1 = cli.index_range(0, args, n), cli.index_range(1, args, n)
    assert (r0.start, r0.stop, r1.start, r1.stop) == (10, 160, 160, 310)                   # the remainder image is dropped by //
    assert args.dataset == "imagenet128"                                                   # :128


def test_missing_starting_points_is_an_error(monkeypatch, tmp_path):
    """No silent synthetic fallback: without --synthetic the starting-point files must exist (checked before any GPU work)."""
    cli = _cli()
    args = cli.parse_args(["--image-size", "64", "--scheduler-type", "mc_dropout", "--model-type", "unet", "--start-index", "0"])
    assert not args.synthetic
    monkeypatch.setattr(cli, "DIFFUSION_STARTING_POINTS", cli.Path(tmp_path))
    monkeypatch.setattr(torch.cuda, "set_device", lambda d: None)
    with pytest.raises(FileNotFoundError, match="y.pth"):
        cli.generate_and_save(0, args, 4, tmp_path)


def test_scheduler_choices_are_the_references():
    cli = _cli()
    for name in ("mc_dropout", "flip", "uncertainty", "uncertainty_centered", "uncertainty_centered_d", "uncertainty_image",
                 "uncertainty_original", "uncertainty_zigzag_centered", "dpm_2_uncertainty_centered", "flip_grad"):
        a = cli.parse_args(["--image-size", "64", "--scheduler-type", name, "--model-type", "unet", "--start-index", "0"])
        assert a.scheduler_type == name
    with pytest.raises(SystemExit):
        cli.parse_args(["--image-size", "64", "--scheduler-type", "nope", "--model-type", "unet", "--start-index", "0"])


# ------------------------------------------------------------------------------------------------------------------ GPU
def _direct(cli, args, n):
    """The same job through the driver API, built the way generate_and_save builds it."""
    from diffusion_uncertainty_amd.generate_samples import (generate_samples_model_scheduler_class_conditioned_from_tensor,
                                                            generate_samples_model_scheduler_class_conditioned_uvit_from_tensor)
    from diffusion_uncertainty_amd.init_model import init_guided_diffusion_imagenet64_and_scheduler
    from diffusion_uncertainty_amd.schedulers_uncertainty.get_uncertainty_scheduler import get_uncertainty_scheduler
    from diffusion_uncertainty_amd.utils import randomize_parameters_, seed_everything, starting_points
    from diffusion_uncertainty_amd.uvit.load_pretrained_models import load_autoencoder_uvit, load_uvit, load_uvit_scheduler
    from diffusion_uncertainty_amd.uvit.uvit_ae import UViTAE
    device = torch.device("cuda", 0)
    seed_everything(args.seed)
    if args.model_type == "unet":
        X_T, y = starting_points(n, (3, 64, 64), dataset_index=0)                       # imagenet64: the script's first dataset
        unet, scheduler = init_guided_diffusion_imagenet64_and_scheduler(load_checkpoint=False, dropout=args.dropout)
        randomize_parameters_(unet, seed=64)
        unet.eval().to(device)
        unet.precision, unet.split_k, unet.winograd = "f16x3", True, True
        model = unet
    else:
        X_T, y = starting_points(n, (4, 32, 32), dataset_index=3)                       # imagenet256: its fourth
        uvit = randomize_parameters_(load_uvit(256, device="cpu", load_checkpoint=False), seed=7)
        ae = randomize_parameters_(load_autoencoder_uvit(device="cpu", load_checkpoint=False), seed=9)
        uvit.precision, uvit.split_k, ae.precision = "f16x3", True, "f16x3"
        model = UViTAE(uvit, ae).to(device)
        scheduler = load_uvit_scheduler()
    s = get_uncertainty_scheduler(args, y.to(device), model, scheduler)
    if args.noise == "device":                                   # the script's --noise device branch: Philox noise / dropout masks drawn in HBM
        from diffusion_uncertainty_amd.schedulers_uncertainty.noise import DeviceNoise
        s.noise_source = DeviceNoise(seed=args.seed)
        if args.model_type == "unet":
            model.dropout_mask_source = DeviceNoise(seed=args.seed + 1000).dropout_mask
    s.set_timesteps(args.generation_steps)
    if args.model_type == "unet":
        out = generate_samples_model_scheduler_class_conditioned_from_tensor(X_T, y, args.batch_size, device, model, s)
    else:
        out = generate_samples_model_scheduler_class_conditioned_uvit_from_tensor(X_T, y, args.batch_size, model, s, device)
    return s, out


@pytest.mark.gpu
@pytest.mark.parametrize("model_type,image_size,scheduler_type", [("unet", 64, "uncertainty_zigzag_centered"), ("unet", 64, "mc_dropout"),
                                                                  ("uvit", 256, "uncertainty_zigzag_centered")])
def test_generate_and_save_writes_the_reference_files(tmp_path, model_type, image_size, scheduler_type):
    cli = _cli()
    n, steps, window = 4, 4, 2
    args = cli.parse_args(["--image-size", str(image_size), "--scheduler-type", scheduler_type, "--model-type", model_type,
                           "--start-index", "0", "--num-samples", str(n), "--batch-size", "3", "--generation-steps", str(steps), "-M", "2",
                           "--start-step-uc", "1", "--num-steps-uc", str(window), "--random-weights", "--synthetic"]
                          + (["--noise", "device"] if scheduler_type == "mc_dropout" else []))       # (host dropout tapes of ADM-64: 30 s on the CPU)
    cli.generate_and_save(0, args, n, tmp_path)
    names = sorted(os.listdir(tmp_path))
    assert names == sorted(["timestep_0.pth", "timesteps.pth", "score_0.pth", "score.pth", f"uncertainty_{scheduler_type}_0.pth",
                            "gen_images_0.pth"])
    t = torch.load(tmp_path / "timestep_0.pth")
    score = torch.load(tmp_path / "score_0.pth")
    unc = torch.load(tmp_path / f"uncertainty_{scheduler_type}_0.pth")
    img = torch.load(tmp_path / "gen_images_0.pth")
    lat = (4, 32, 32) if model_type == "uvit" else (3, 64, 64)
    assert t.shape == (steps,) and t.dtype == torch.int64
    assert score.shape == (n, window) + lat and score.dtype == torch.float32 and score.device.type == "cpu"
    assert unc.shape == (n, window) + lat and unc.dtype == torch.float32
    assert img.shape == (n, 3, image_size, image_size) and img.dtype == torch.uint8
    assert torch.isfinite(score).all() and torch.isfinite(unc).all() and (unc >= 0).all() and unc.max() > 0
    assert torch.equal(torch.load(tmp_path / "score.pth"), score) and torch.equal(torch.load(tmp_path / "timesteps.pth"), t)
    s, out = _direct(cli, args, n)
    assert torch.equal(s.timesteps.cpu(), t)
    assert torch.equal(out["gen_images"], img) and torch.equal(out["score"], score) and torch.equal(out["uncertainty"], unc)


@pytest.mark.gpu
def test_cli_without_starting_points_fails_loudly_subprocess(tmp_path):
    """The caller as a user starts it, in a fresh process, WITHOUT --synthetic and without the starting-point files: a non-zero exit that
    names the missing file (no silent synthetic fallback).  (The complete run of the script in a fresh process - argument parsing, args.yaml,
    the one-GPU path of main(), the output files - is the first stage of test_maps_to_thresholds_to_guided_sampling_cli_chain.)"""
    import subprocess
    cmd = [sys.executable, SCRIPT, "--image-size", "64", "--scheduler-type", "uncertainty_zigzag_centered", "--model-type", "unet", "--start-index", "1",
           "--num-samples", "3", "--batch-size", "2", "--generation-steps", "3", "-M", "1", "--random-weights", "--dest", str(tmp_path / "out")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=dict(os.environ, DIFFUSION_UNCERTAINTY_ROOT=str(tmp_path / "empty_root")))
    assert r.returncode != 0 and "y.pth" in r.stderr and "starting points not found" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("update", ["gradient"])          # (the script's default, as upstream; the posterior update: tests/test_gpu_sd.py, bench.py)
def test_sd_cli_end_to_end_subprocess(tmp_path, update):
    """The second caller (reference scripts/generate_with_uncertainty_threshold_stable_diffusion.py: percentile-thresholded guidance on the
    SD-1.5 latent UNet) in a fresh process with its default update - gradient mode, as upstream (:16-17,88): 4 PLMS steps, 1 guided step,
    M = 2; the saved latents are finite."""
    import subprocess
    script = os.path.join(ROOT, "scripts", "generate_with_uncertainty_threshold_stable_diffusion.py")
    out = tmp_path / f"latents_{update}.pth"
    cmd = [sys.executable, script, "--num-steps", "4", "--num-steps-threshold", "1", "-M", "2", "--repeats", "0", "--no-graph", "--out", str(out)]
    if update == "posterior":
        cmd.append("--use-posterior")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    assert f"{update} update" in r.stdout
    lat = torch.load(out)
    assert lat.shape == (1, 4, 64, 64) and torch.isfinite(lat).all() and lat.abs().max() > 0


def test_threshold_cli_arguments_and_folder_checks(tmp_path):
    """scripts/compute_threshold_pixel_wise.py (reference :21-40,167-186): flags and defaults, and the folder validation that runs before
    any GPU work - a missing args.yaml, folders that disagree on the scheduler type, a folder without uncertainty files."""
    import yaml
    spec = importlib.util.spec_from_file_location("thr_cli", os.path.join(ROOT, "scripts", "compute_threshold_pixel_wise.py"))
    thr = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(thr)
    a = thr.parse_args(["--dataset-folders", "x", "y"])
    assert (a.perc, a.batch_size, a.on_cpu, a.dataset_folders) == (0.15, 128, False, ["x", "y"])
    with pytest.raises(AssertionError):
        thr.parse_args([])
    f1, f2 = tmp_path / "a", tmp_path / "b"
    for f, st in ((f1, "uncertainty_zigzag_centered"), (f2, "mc_dropout")):
        f.mkdir()
        yaml.safe_dump({"dataset": "imagenet64", "scheduler_type": st}, open(f / "args.yaml", "w"))
        torch.save(torch.zeros(1), f / "gen_images_0.pth")
        torch.save(torch.zeros(1), f / f"uncertainty_{st}_0.pth")
    thr.assert_validity_folders_(argparse.Namespace(dataset_folders=[f1]))
    with pytest.raises(AssertionError, match="scheduler types do not match"):
        thr.assert_validity_folders_(argparse.Namespace(dataset_folders=[f1, f2]))
    (f1 / "uncertainty_uncertainty_zigzag_centered_0.pth").unlink()
    with pytest.raises(AssertionError, match="no uncertainty files"):
        thr.assert_validity_folders_(argparse.Namespace(dataset_folders=[f1]))
    (f2 / "args.yaml").unlink()
    with pytest.raises(AssertionError, match="args.yaml not found"):
        thr.assert_validity_folders_(argparse.Namespace(dataset_folders=[f2]))


def test_guided_generation_cli_arguments():
    """scripts/generate_images_with_uncertainty_threshold.py: the reference's flags (:85-109) with its defaults - except --guidance-type,
    whose upstream default (gradient) is not built - and config files with terminal arguments taking priority (:113-118)."""
    spec = importlib.util.spec_from_file_location("gen_thr_cli", os.path.join(ROOT, "scripts", "generate_images_with_uncertainty_threshold.py"))
    g = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(g)
    a = g.parse_args(["--dataset", "imagenet64"])
    assert (a.num_samples, a.batch_size, a.scheduler_type, a.num_steps, a.start_step_guidance, a.num_steps_guidance, a.start_index, a.seed, a.percentile,
            a.use_percentile, a.skip_ddim, a.threshold_type, a.lambda_update) == (1000, 128, "uncertainty", 20, 0, 20, 0, 491, 0.95, False, False, "higher", 0.1)
    assert a.guidance_type == "posterior"
    b = g.parse_args(["--config", "imagenet128_0_95", "--num-samples", "7"])
    assert (b.dataset_name, b.num_samples, b.scheduler_type, b.num_steps_guidance, b.use_percentile, b.seed) == ("imagenet128", 7, "uncertainty_centered", 3, True, 853983298)
    with pytest.raises(AssertionError):
        g.parse_args([])


@pytest.mark.gpu
def test_maps_to_thresholds_to_guided_sampling_cli_chain(tmp_path, monkeypatch):
    """SURVEY 8f rank 2 end to end through the three CLIs, ADM-64 with seeded random weights and synthetic starting points: (1) a dataset of
    104 uncertainty maps (4 steps, UC window = all 4), (2) pixel-wise thresholds at perc 0.5 - equal, bit for bit, to torch's argsort +
    gather on the fp16 maps (the reference's arithmetic, :143-165), (3) posterior-guided sampling on 16 // 4 = 4 timesteps with those
    thresholds: the guided images differ from the plain DDIM images of the same starting points and both are valid uint8 batches."""
    import subprocess

    import yaml
    root = tmp_path / "root"
    env = dict(os.environ, DIFFUSION_UNCERTAINTY_ROOT=str(root))
    ds = tmp_path / "dataset"
    gen = [sys.executable, SCRIPT, "--image-size", "64", "--scheduler-type", "uncertainty_zigzag_centered", "--model-type", "unet", "--start-index", "0",
           "--num-samples", "104", "--batch-size", "52", "--generation-steps", "4", "-M", "1", "--start-step-uc", "0", "--num-steps-uc", "4",
           "--random-weights", "--synthetic", "--noise", "device", "--dest", str(ds)]
    r = subprocess.run(gen, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    thr_cmd = [sys.executable, os.path.join(ROOT, "scripts", "compute_threshold_pixel_wise.py"), "--dataset-folders", str(ds), "--perc", "0.5"]
    r = subprocess.run(thr_cmd, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    thr_file = root / "results" / "thresholds" / "imagenet64" / "thresholds_uncertainty_zigzag_centered_perc=0.5.pth"
    cfg_file = root / "results" / "thresholds" / "imagenet64" / "config_uncertainty_zigzag_centered_perc=0.5.yaml"
    thr = torch.load(thr_file)
    u = torch.load(ds / "uncertainty_uncertainty_zigzag_centered_0.pth").half()
    want = torch.stack([u[:, i].gather(0, u[:, i].argsort(dim=0)[int(104 * 0.5)].unsqueeze(0)).squeeze(0) for i in range(4)])
    assert thr.dtype == torch.float16 and thr.shape == (4, 3, 64, 64) and torch.equal(thr.view(torch.int16), want.view(torch.int16))
    cfg = yaml.safe_load(open(cfg_file))
    assert cfg["perc"] == 0.5 and cfg["dataset_config"]["dataset"] == "imagenet64" and cfg["dataset_config"]["generation_steps"] == 4
    out = tmp_path / "guided"
    g_cmd = [sys.executable, os.path.join(ROOT, "scripts", "generate_images_with_uncertainty_threshold.py"), "--dataset", "imagenet64", "--scheduler-type",
             "uncertainty_zigzag_centered", "--percentile", "0.5", "--num-samples", "6", "--batch-size", "4", "--num-steps", "16", "--start-step-guidance", "1",
             "--num-steps-guidance", "1", "--random-weights", "--synthetic", "--dest", str(out)]
    r = subprocess.run(g_cmd, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    run = [p for p in (out / "imagenet64_uncertainty_zigzag_centered_perc=0.5").iterdir()][0]
    plain, guided = torch.load(run / "gen_images.pth"), torch.load(run / "gen_images_threshold.pth")
    assert plain.shape == guided.shape == (6, 3, 64, 64) and plain.dtype == guided.dtype == torch.uint8
    # 'higher' touches the pixels whose Monte-Carlo variance exceeds the dataset's threshold, 'lower' the others: between them every pixel
    # of the guided window is revised once, so at least one of the two runs must leave the plain trajectory (which one is data)
    r = subprocess.run(g_cmd + ["--threshold-type", "lower", "--dest", str(tmp_path / "guided_lower")], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    run_l = [p for p in (tmp_path / "guided_lower" / "imagenet64_uncertainty_zigzag_centered_perc=0.5").iterdir()][0]
    guided_l = torch.load(run_l / "gen_images_threshold.pth")
    assert torch.equal(torch.load(run_l / "gen_images.pth"), plain)                              # same plain run (same seed, same starting points)
    changed = [(plain != g_).float().mean().item() for g_ in (guided, guided_l)]
    print("fraction of bytes changed by the guided window: higher", changed[0], "lower", changed[1])
    assert max(changed) > 0.01 and not torch.equal(guided, guided_l)
    rec = json.load(open(out / "results.json"))
    assert rec[-1]["guidance_type"] == "posterior" and rec[-1]["fid_score"] is None and rec[-1]["use_percentile"] is False
    spec = importlib.util.spec_from_file_location("gen_thr_cli2", os.path.join(ROOT, "scripts", "generate_images_with_uncertainty_threshold.py"))
    gcli = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gcli)
    with pytest.raises(NotImplementedError, match="autograd through"):                          # (raised before any model is built)
        gcli.main(g_cmd[2:] + ["--guidance-type", "gradient"])

@pytest.mark.gpu
def test_measure_times_cli_subprocess(tmp_path):
    """scripts/measure_times_imagenet.py (the reference script the metric's span is taken from, :88-118): time_uc and time_normal of the
    driver on ADM-64, one record appended to the times file; the uncertainty window costs time (time_uc > time_normal)."""
    import subprocess
    times = tmp_path / "times.json"
    cmd = [sys.executable, os.path.join(ROOT, "scripts", "measure_times_imagenet.py"), "--image-size", "64", "--scheduler-type", "uncertainty_zigzag_centered",
           "--model-type", "unet", "--start-index", "0", "--num-samples", "8", "--batch-size", "8", "--generation-steps", "6", "-M", "2", "--start-step-uc", "2",
           "--num-steps-uc", "3", "--random-weights", "--synthetic", "--noise", "device", "--times-file", str(times)]
    json.dump([{"dataset": "earlier run"}], open(times, "w"))             # the script APPENDS to an existing file (:128-136)
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    rec = json.load(open(times))
    assert len(rec) == 2 and rec[0] == {"dataset": "earlier run"}
    last = rec[-1]
    assert (last["dataset"], last["image_size"], last["model_type"], last["scheduler_type"]) == ("imagenet", 64, "adm", "uncertainty_zigzag_centered")
    assert last["time_uc"] > last["time_normal"] > 0 and last["executed_forwards_per_batch_uc"] == 6 + 3 * 2


@pytest.mark.gpu
def test_measure_times_plain_scheduler_families():
    """The plain schedulers scripts/measure_times_imagenet.py times (reference :103-107): DDIM for the ADM models, DPM-Solver-2 for the U-ViT
    ones - the uncertainty scheduler of the family with an EMPTY window behind PlainScheduler: the driver collects no maps, the images are
    valid, and for DDIM the result equals the plain DDIMScheduler's, bit for bit (tiny models, in process)."""
    spec = importlib.util.spec_from_file_location("measure_cli", os.path.join(ROOT, "scripts", "measure_times_imagenet.py"))
    mt = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mt)
    from diffusion_uncertainty_amd.generate_samples import (generate_samples_model_scheduler_class_conditioned_from_tensor,
                                                            generate_samples_model_scheduler_class_conditioned_uvit_from_tensor)
    from diffusion_uncertainty_amd.init_model import init_scheduler_imagenet128, load_uvit_scheduler
    from diffusion_uncertainty_amd.schedulers_uncertainty.mixin import SchedulerUncertaintyMixin
    from diffusion_uncertainty_amd.schedulers_uncertainty.scheduling_ddim import DDIMScheduler
    from diffusion_uncertainty_amd.utils import randomize_parameters_
    from diffusion_uncertainty_amd.uvit.autoencoder import FrozenAutoencoderKL
    from diffusion_uncertainty_amd.uvit.uvit import UViT
    from diffusion_uncertainty_amd.uvit.uvit_ae import UViTAE
    from test_gpu_models import TINY_ADM, TINY_UVIT, build_adm
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(3)
    base = argparse.Namespace(generation_steps=6, M=2, start_step_uc=0, num_steps_uc=6, predict_next=True, eta=0.0, num_zigzag=3, uncertainty_distance=1, seed=0)
    # ADM: DDIM
    m, _ = build_adm(dict(TINY_ADM), 5)
    X, y = torch.randn(3, 3, 16, 16, generator=g), torch.tensor([1, 5, 9])
    a = argparse.Namespace(**dict(base.__dict__, image_size=64, model_type="unet", scheduler_type="uncertainty_zigzag_centered"))
    s = mt.plain_scheduler(a, y.to(dev), m, init_scheduler_imagenet128())
    assert not isinstance(s, SchedulerUncertaintyMixin)
    r = generate_samples_model_scheduler_class_conditioned_from_tensor(X, y, 3, dev, m, s)
    assert set(r) == {"gen_images"} and r["gen_images"].dtype == torch.uint8
    d = DDIMScheduler.from_config(init_scheduler_imagenet128().config)
    d.set_timesteps(6)
    assert torch.equal(generate_samples_model_scheduler_class_conditioned_from_tensor(X, y, 3, dev, m, d)["gen_images"], r["gen_images"])
    # U-ViT: DPM-Solver-2
    tiny_vae = dict(double_z=True, z_channels=4, resolution=32, in_channels=3, out_ch=3, ch=32, ch_mult=[1, 2], num_res_blocks=1, attn_resolutions=[], dropout=0.0)
    u = UViTAE(randomize_parameters_(UViT(**TINY_UVIT), 3).eval(), randomize_parameters_(FrozenAutoencoderKL(tiny_vae, 4), 9)).to(dev)
    Xl, yl = torch.randn(2, 4, 8, 8, generator=g), torch.tensor([2, 7])
    a = argparse.Namespace(**dict(base.__dict__, image_size=256, model_type="uvit", scheduler_type="uncertainty_zigzag_centered"))
    s = mt.plain_scheduler(a, yl.to(dev), u, load_uvit_scheduler())
    assert type(s._inner).__name__.startswith("KDPM2") and not isinstance(s, SchedulerUncertaintyMixin)
    r = generate_samples_model_scheduler_class_conditioned_uvit_from_tensor(Xl, yl, 2, u, s, dev)
    assert "uncertainty" not in r and r["gen_images"].shape == (2, 3, 16, 16) and r["gen_images"].dtype == torch.uint8


def test_starting_data_script_draws_like_the_reference(tmp_path, monkeypatch):
    """scripts/generate_diffusion_starting_data.py (reference :25-69): the k-th dataset's generator is seeded 49394 + k, noise first, labels
    after it from the same generator; files land under DIFFUSION_STARTING_POINTS/<name>."""
    monkeypatch.setenv("DIFFUSION_UNCERTAINTY_ROOT", str(tmp_path))
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "generate_diffusion_starting_data.py"), "--datasets", "imagenet128_uvit", "cifar10",
                        "--num-samples", "9"], capture_output=True, text=True, timeout=300, env=dict(os.environ, DIFFUSION_UNCERTAINTY_ROOT=str(tmp_path)))
    assert r.returncode == 0, r.stderr[-1500:]
    base = tmp_path / "results" / "diffusion_starting_points"
    for name, k, shape, classes in (("imagenet128_uvit", 2, (4, 16, 16), 1000), ("cifar10", 5, (3, 32, 32), 10)):
        g = torch.Generator().manual_seed(49394 + k)
        x = torch.randn(9, *shape, generator=g)
        y = torch.randint(0, classes, (9,), generator=g)
        assert torch.equal(torch.load(base / name / "X_T.pth"), x) and torch.equal(torch.load(base / name / "y.pth"), y)
    from diffusion_uncertainty_amd.utils import STARTING_POINT_DATASETS, starting_points
    assert list(STARTING_POINT_DATASETS) == ["imagenet64", "imagenet128", "imagenet128_uvit", "imagenet256", "imagenet512", "cifar10"]
    # X_T of a short draw is a prefix of a longer one (the labels are not: they follow the noise in the same stream)
    a, _ = starting_points(5, (3, 8, 8), dataset_index=1)
    b, _ = starting_points(40, (3, 8, 8), dataset_index=1)
    assert torch.equal(a, b[:5])


@pytest.mark.skipif(not os.path.exists(REF_SCRIPT), reason="reference checkout not present (GPU box)")
def test_reference_caller_source_runs_against_this_package():
    """The reference's OWN script text, unmodified, against this package (BASELINE.json north_star: "scripts/generate_dataset_score_uncertainty_
    imagenet.py runs unchanged"): in a fresh process the package is aliased as `diffusion_uncertainty`, the script's two third-party imports
    that this image lacks get stand-ins (`path.Path` -> the Path of paths.py, `pytorch_lightning.seed_everything` -> utils.seed_everything),
    and the script file is executed as a module: its whole import block (:14-31) resolves against this package, its `parse_args` (:147-196)
    lists this checkout's config/generation/*.yaml through `CONFIG.joinpath(..).files(..)` / `.basename()`, loads one through `load_config` and
    applies the terminal overrides.  (Everything after that point needs a GPU, which this container does not have and where the
    reference checkout does not travel: tests/test_caller.py's -m gpu tests run this package's own copy of the flow.)"""
    import subprocess
    code = f"""
import sys, types, importlib.util
sys.path.insert(0, {ROOT!r})
import diffusion_uncertainty_amd
diffusion_uncertainty_amd.alias_as_reference()
from diffusion_uncertainty_amd import paths, utils
sys.modules["path"] = types.SimpleNamespace(Path=paths.Path)
sys.modules["pytorch_lightning"] = types.SimpleNamespace(seed_everything=utils.seed_everything)
spec = importlib.util.spec_from_file_location("reference_caller", {REF_SCRIPT!r})
mod = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mod)                                   # the import block and the function definitions, as written upstream
import diffusion_uncertainty.generate_samples as gs, diffusion_uncertainty_amd.generate_samples as ours
assert gs is ours and mod.generate_samples_model_scheduler_class_conditioned_from_tensor is ours.generate_samples_model_scheduler_class_conditioned_from_tensor
assert mod.get_uncertainty_scheduler.__module__.startswith("diffusion_uncertainty_amd")
sys.argv = ["x", "--config", "imagenet128_readme_example", "-M", "7", "--batch-size", "16"]
a = mod.parse_args()
assert (a.M, a.batch_size, a.generation_steps, a.start_step_uc, a.num_steps_uc, a.image_size, a.scheduler_type, a.model_type, a.num_zigzag) == \\
    (7, 16, 50, 40, 10, 128, "uncertainty_zigzag_centered", "unet", 3), a
sys.argv = ["x", "--image-size", "64", "--scheduler-type", "mc_dropout", "--model-type", "unet", "--start-index", "5"]
b = mod.parse_args()
assert (b.num_samples, b.batch_size, b.generation_steps, b.M, b.dropout, b.start_index) == (300, 32, 20, 30, 0.1, 5)
print("REFERENCE_CALLER_OK")
"""
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "REFERENCE_CALLER_OK" in r.stdout, (r.stdout[-500:], r.stderr[-2500:])


@pytest.mark.skipif(not os.path.exists("/root/reference/scripts/compute_threshold_pixel_wise.py"), reason="reference checkout not present (GPU box)")
def test_reference_threshold_script_source_runs_against_this_package(tmp_path):
    """The reference's scripts/compute_threshold_pixel_wise.py, unmodified, in a fresh process against the aliased package (its imports:
    `utils.load_config`, `paths.{CONFIG, FID, RESULTS, ROOT}`): it finds a dataset folder written in this package's output format
    (args.yaml, uncertainty_*_0.pth, gen_images_0.pth), runs its own CPU selection and writes thresholds + config under
    RESULTS/thresholds/<dataset>; `thresholds.compute_threshold_pixel_wise` - this package's GPU path - must produce the same bits, which the
    -m gpu chain test asserts against the same torch arithmetic."""
    import subprocess

    import yaml
    root = tmp_path / "root"
    ds = tmp_path / "dataset"
    ds.mkdir()
    u = torch.randn(120, 2, 3, 8, 8, generator=torch.Generator().manual_seed(5)).square()
    torch.save(u, ds / "uncertainty_uncertainty_zigzag_centered_0.pth")
    torch.save(torch.zeros(120, 3, 8, 8, dtype=torch.uint8), ds / "gen_images_0.pth")
    yaml.safe_dump({"dataset": "imagenet64", "scheduler_type": "uncertainty_zigzag_centered", "generation_steps": 2}, open(ds / "args.yaml", "w"))
    code = f"""
import sys, types, importlib.util
sys.path.insert(0, {ROOT!r})
import diffusion_uncertainty_amd
diffusion_uncertainty_amd.alias_as_reference()
from diffusion_uncertainty_amd import paths
diffusion_uncertainty_amd.alias_as_reference(create_dirs=True)     # (ROOT is this test's scratch folder: DIFFUSION_UNCERTAINTY_ROOT)
class P(paths.Path):                                           # the two further path.Path methods this script calls (:33-35,111)
    def relpath(self, start):
        import os
        return os.path.relpath(str(self), str(start))
sys.modules["path"] = types.SimpleNamespace(Path=P)
spec = importlib.util.spec_from_file_location("reference_threshold", "/root/reference/scripts/compute_threshold_pixel_wise.py")
mod = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mod)
sys.argv = ["x", "--dataset-folders", {str(ds)!r}, "--perc", "0.25", "--on-cpu"]
mod.main()
print("REFERENCE_THRESHOLD_OK")
"""
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=dict(os.environ, DIFFUSION_UNCERTAINTY_ROOT=str(root)))
    assert r.returncode == 0 and "REFERENCE_THRESHOLD_OK" in r.stdout, (r.stdout[-500:], r.stderr[-2500:])
    thr = torch.load(root / "results" / "thresholds" / "imagenet64" / "thresholds_uncertainty_zigzag_centered_perc=0.25.pth")
    h = u.half()
    want = torch.stack([h[:, i].gather(0, h[:, i].argsort(dim=0)[int(120 * 0.25)].unsqueeze(0)).squeeze(0) for i in range(2)])
    assert thr.dtype == torch.float16 and torch.equal(thr, want)
    cfg = yaml.safe_load(open(root / "results" / "thresholds" / "imagenet64" / "config_uncertainty_zigzag_centered_perc=0.25.yaml"))
    assert cfg["dataset_config"]["dataset"] == "imagenet64" and cfg["perc"] == 0.25
