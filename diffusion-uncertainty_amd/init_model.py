"""Model / base-scheduler factories (reference init_model.py:20-56,128-157; uvit/load_pretrained_models.py:13-62).

Same function names, arguments and hyper-parameters; checkpoints are loaded with a strict
``load_state_dict`` from ``paths.MODELS`` (state-dict keys identical to the OpenAI / U-ViT releases).
"""
import math

import numpy as np
import torch

from .configuration import DDPMScheduler
from .guided_diffusion.unet_openai import UNetModel
from .paths import MODELS
from .uvit.uvit import UViT
from .uvit.uvit_ae import UViTAE

ADM128_PARAMS = dict(image_size=128, in_channels=3, model_channels=256, out_channels=6, num_res_blocks=2,
                     attention_resolutions=(4, 8, 16), dropout=0.0, channel_mult=(1, 1, 2, 3, 4), num_classes=1000,
                     use_checkpoint=False, use_fp16=False, num_heads=4, num_head_channels=-1, num_heads_upsample=4,
                     use_scale_shift_norm=True, resblock_updown=True, use_new_attention_order=False)       # init_model.py:21
ADM64_PARAMS = dict(image_size=64, in_channels=3, model_channels=192, out_channels=6, num_res_blocks=3,
                    attention_resolutions=(2, 4, 8), dropout=0.1, channel_mult=(1, 2, 3, 4), num_classes=1000,
                    use_checkpoint=False, use_fp16=False, num_heads=4, num_head_channels=64, num_heads_upsample=4,
                    use_scale_shift_norm=True, resblock_updown=True, use_new_attention_order=True)         # init_model.py:45-47


def init_guided_diffusion_imagenet128(load_checkpoint=True):
    ddpm = UNetModel(**ADM128_PARAMS)
    if load_checkpoint:
        ddpm.load_state_dict(torch.load(MODELS / "128x128_diffusion.pt", map_location="cpu"))
    return ddpm


def init_scheduler_imagenet128():
    return DDPMScheduler(beta_start=0.0001, beta_end=0.02, num_train_timesteps=1000, beta_schedule="linear")


def init_guided_diffusion_imagenet128_and_scheduler(load_checkpoint=True):
    return init_guided_diffusion_imagenet128(load_checkpoint=load_checkpoint), init_scheduler_imagenet128()


def init_guided_diffusion_imagenet64(load_checkpoint=True, dropout: float = 0.1):
    model = UNetModel(**dict(ADM64_PARAMS, dropout=dropout))
    if load_checkpoint:
        model.load_state_dict(state_dict=torch.load(MODELS / "64x64_diffusion.pt", map_location="cpu"))
    return model.to(torch.float32)


def betas_for_alpha_bar(num_diffusion_timesteps, alpha_bar, max_beta=0.999):
    """init_model.py:140-157."""
    betas = []
    for i in range(num_diffusion_timesteps):
        t1, t2 = i / num_diffusion_timesteps, (i + 1) / num_diffusion_timesteps
        betas.append(min(1 - alpha_bar(t2) / alpha_bar(t1), max_beta))
    return np.array(betas)


def init_scheduler_imagenet64():
    betas = betas_for_alpha_bar(1000, lambda t: math.cos((t + 0.008) / 1.008 * math.pi / 2) ** 2)
    return DDPMScheduler(trained_betas=betas, num_train_timesteps=1000, beta_schedule="cosine")


def init_guided_diffusion_imagenet64_and_scheduler(load_checkpoint=True, dropout=0.1):
    return init_guided_diffusion_imagenet64(load_checkpoint=load_checkpoint, dropout=dropout), init_scheduler_imagenet64()


# ---- U-ViT: the loaders live where the reference keeps them (uvit/load_pretrained_models.py:13,44,66); re-exported here
from .uvit.load_pretrained_models import load_autoencoder_uvit, load_uvit, load_uvit_scheduler  # noqa: E402,F401


def instantiate_model_scheduler(args, load_checkpoint=True, device="cuda"):
    """(model, base scheduler) by dataset name - or by the ``args.yaml`` of ``args.dataset_folders[0]`` when given a Namespace
    (reference init_model.py:222-262, a ``singledispatch`` pair; the CIFAR-10 branch is the diffusers UNet2DModel path: not built)."""
    if not isinstance(args, str):
        import os

        import yaml
        with open(os.path.join(str(args.dataset_folders[0]), "args.yaml")) as f:
            args = yaml.safe_load(f)["dataset"]
    dataset_name = args
    if dataset_name == "imagenet64":
        return init_guided_diffusion_imagenet64_and_scheduler(load_checkpoint=load_checkpoint)
    if dataset_name == "imagenet128":
        return init_guided_diffusion_imagenet128_and_scheduler(load_checkpoint=load_checkpoint)
    if dataset_name in ("imagenet256", "imagenet512"):
        size = int(dataset_name[len("imagenet"):])
        return (UViTAE(load_uvit(size, device, load_checkpoint=load_checkpoint), load_autoencoder_uvit(device, load_checkpoint=load_checkpoint)),
                load_uvit_scheduler())
    if dataset_name == "cifar10":
        raise NotImplementedError("cifar10: the diffusers UNet2DModel path is outside the accelerated hot path (SURVEY 8f)")
    raise ValueError(f"invalid dataset name {dataset_name}")


init_model_scheduler = instantiate_model_scheduler

__all__ = ["UNetModel", "UViT", "UViTAE", "DDPMScheduler"]
