"""Diagnostic (GPU): the full 50-step headline job on two images in four arithmetic variants (exact fp32, exact fp32 with the stand-alone
GroupNorm statistics, f16x3 direct, f16x3 Winograd), pairwise score distances per seed: python scripts/diag/chain_divergence.py [steps]"""
import sys, os, argparse
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from oracle import adm
from test_gpu_models import build_adm
from diffusion_uncertainty_amd.generate_samples import generate_samples_model_scheduler_class_conditioned_from_tensor as gen
from diffusion_uncertainty_amd.init_model import init_scheduler_imagenet128
from diffusion_uncertainty_amd.schedulers_uncertainty.get_uncertainty_scheduler import get_uncertainty_scheduler
from diffusion_uncertainty_amd.schedulers_uncertainty.noise import DeviceNoise
dev = "cuda"
m, _ = build_adm(dict(adm.ADM128), 128)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
for seed in (1, 2, 3, 4):
    g = torch.Generator().manual_seed(1000 + seed)
    X_T, y = torch.randn(2, 3, 128, 128, generator=g), torch.randint(0, 1000, (2,), generator=g)
    res = {}
    for name, prec, wino, fuse in (("fp32", "fp32", True, True), ("fp32b", "fp32", True, False), ("direct", "f16x3", False, True), ("wino", "f16x3", True, True)):
        m.precision, m.winograd, m.fuse_gn_stats = prec, wino, fuse
        args = argparse.Namespace(scheduler_type="uncertainty_zigzag_centered", M=5, start_step_uc=steps - 10, num_steps_uc=10, predict_next=True,
                                  eta=0.0, num_zigzag=3, uncertainty_distance=1)
        s = get_uncertainty_scheduler(args, None, m, init_scheduler_imagenet128())
        s.noise_source = DeviceNoise(seed=seed)
        s.set_timesteps(steps)
        res[name] = gen(X_T, y, 2, torch.device(dev), m, s)
    def cmp(a, b):
        return (res[a]["score"] - res[b]["score"]).abs().max().item(), (res[a]["score"][:, 0] - res[b]["score"][:, 0]).abs().max().item()
    print(f"seed {seed}: fp32b-fp32 {cmp('fp32b','fp32')[0]:.2e} (first UC step {cmp('fp32b','fp32')[1]:.2e}) | direct-fp32 {cmp('direct','fp32')[0]:.2e} ({cmp('direct','fp32')[1]:.2e}) | "
          f"wino-fp32 {cmp('wino','fp32')[0]:.2e} ({cmp('wino','fp32')[1]:.2e}) | wino-direct {cmp('wino','direct')[0]:.2e}", flush=True)
