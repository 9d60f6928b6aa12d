"""U-ViT loaders, at the address the reference's callers import them from
(reference uvit/load_pretrained_models.py: ``load_uvit`` :13-40, ``load_uvit_scheduler`` :44-62,
``load_autoencoder_uvit`` :66-84, ``sample_uvit`` :87-112).

Same names, positional arguments and hyper-parameters.  Differences: there is no network here, so a missing checkpoint
is a ``FileNotFoundError`` naming the expected file under ``paths.MODELS`` instead of a ``gdown`` download;
``load_checkpoint=False`` (keyword only in spirit: the reference has no such argument) builds the architecture with its
constructor initialisation, for synthetic-weight runs.
"""
import torch

from ..configuration import DDPMScheduler
from ..paths import MODELS
from .uvit import UViT


def _checkpoint(name):
    path = MODELS / name
    if not path.exists():
        raise FileNotFoundError(f"{path}: checkpoint not found (no download on this platform; place the file the reference's "
                                f"loader would fetch there, or pass load_checkpoint=False for seeded random weights)")
    return path


def load_uvit(image_size=256, device="cuda", load_checkpoint=True):
    """U-ViT-H: patch 2 on 32x32 latents (ImageNet-256) or patch 4 on 64x64 (ImageNet-512); 1001 classes (:34-36)."""
    if image_size not in (256, 512):                                     # the reference's beartype Literal[256, 512]
        raise TypeError(f"image_size must be 256 or 512, got {image_size!r}")
    patch_size = 2 if image_size == 256 else 4
    z_size = image_size // 8
    nnet = UViT(img_size=z_size, patch_size=patch_size, in_chans=4, embed_dim=1152, depth=28, num_heads=16,
                num_classes=1001, conv=False)
    nnet.to(device)
    if load_checkpoint:
        nnet.load_state_dict(torch.load(_checkpoint(f"imagenet{image_size}_uvit_huge.pth"), map_location=device))
    nnet.eval()
    return nnet


def load_uvit_scheduler():
    """:44-62.  Only the DDPM keys survive ``DDPMScheduler.from_config`` (``set_alpha_to_one`` / ``skip_prk_steps`` are
    dropped, so ``set_alpha_to_one`` takes its default, True, in the uncertainty scheduler built from this config)."""
    scheduler_config = {"_class_name": "PNDMScheduler", "_diffusers_version": "0.6.0", "beta_end": 0.012,
                        "beta_schedule": "scaled_linear", "beta_start": 0.00085, "num_train_timesteps": 1000,
                        "set_alpha_to_one": False, "skip_prk_steps": True, "steps_offset": 1, "trained_betas": None,
                        "clip_sample": False}
    ddpm_scheduler = DDPMScheduler.from_config(scheduler_config)
    assert isinstance(ddpm_scheduler, DDPMScheduler), f"Expected DDPMScheduler, got {type(ddpm_scheduler)}"
    return ddpm_scheduler


def load_autoencoder_uvit(device="cuda", load_checkpoint=True):
    """:66-84: the f8 KL-VAE of the U-ViT release (``uvit/autoencoder.py`` ``get_model``); decoder on the HIP library."""
    from .autoencoder import get_model
    autoencoder = get_model(str(_checkpoint("autoencoder_kl_ema.pth")) if load_checkpoint else None)
    autoencoder.eval()
    autoencoder.to(device)
    return autoencoder


@torch.no_grad()
def sample_uvit(uvit, scheduler, autoencoder, y, x_T=None):
    """:87-112: plain sampling loop over ``scheduler.timesteps`` followed by one VAE decode."""
    if x_T is None:
        x_T = torch.randn(y.shape[0], 4, uvit.img_size, uvit.img_size).to(y.device)
    assert x_T.shape[0] == y.shape[0], \
        f"Batch size of x_T and y must be the same, got {x_T.shape[0]=} and {y.shape[0]=} respectively"
    image = x_T.clone()
    for t in scheduler.timesteps:
        t_tensor = torch.zeros(y.shape[0], dtype=torch.long, device=y.device).fill_(t)
        eps = uvit(image, t_tensor, y)
        image = scheduler.step(eps, t, image, return_dict=True).prev_sample
    return autoencoder.decode(image)
