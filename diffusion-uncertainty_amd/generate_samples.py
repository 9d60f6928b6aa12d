"""Sampling drivers (batch loop -> timestep loop -> denoiser -> scheduler.step -> collect -> uint8).

Mirror of reference ``generate_samples.py``:
``generate_samples_model_scheduler_class_conditioned_from_tensor`` :127-233,
``generate_samples_model_scheduler_class_conditioned_uvit_from_tensor`` :469-570 and the two seed-driven
variants the CLI script also imports (:18-124, :573-...).  Same signatures, same result dict
(``gen_images`` uint8, ``uncertainty`` / ``score`` stacked on dim 1, CPU tensors).  The denoiser, the
scheduler arithmetic and the uint8 conversion (dua_to_uint8) all run in libdua_hip.so; ``device`` must be a
HIP device.  The reference's progress ``print`` calls are dropped (quiet by default).
"""
from typing import Any

import torch

from . import ops
from .guided_diffusion.unet_openai import UNetModel
from .schedulers_uncertainty.mixin import SchedulerUncertaintyMixin


def _require_gpu(device):
    if torch.device(device).type != "cuda":
        raise RuntimeError(f"generate_samples: device={device!r}; the HIP path needs a GPU device (no CPU fallback)")


def _eps3(model, x, t_tensor, y):
    """`model(x, t, y=y)[:, :3]` (:184); the HIP UNet writes only the three epsilon channels."""
    if isinstance(model, UNetModel):
        return model(x, t_tensor, y=y, out_channels_keep=3)
    return model(x, t_tensor, y=y)[:, :3]


def predict_model(model, sample, t_tensor, y):
    """Model-agnostic denoiser call, reference generate_samples.py:670-676 (the diffusers UNet2DModel branch :671-672 is
    the CIFAR-10 path, SURVEY 8f, not built)."""
    from .uvit.uvit_ae import UViTAE
    if isinstance(model, UViTAE):
        return model(sample, t_tensor, y)
    return _eps3(model, sample, t_tensor, y)


def _to_host_async(t):
    """Stream-ordered D2H into pinned memory (torch's caching host allocator): the host does not stall the launch queue the way
    the reference's ``.cpu()`` on pageable memory does; _loop synchronises once before it returns."""
    h = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
    h.copy_(t, non_blocking=True)
    return h


def _loop(x, y_batch, model_call, scheduler, device, intermediates=None):
    is_uc = isinstance(scheduler, SchedulerUncertaintyMixin)
    uncertanties, scores = [], []
    if is_uc:
        scheduler.prompt_embeds = y_batch
    scheduler.set_timesteps(len(scheduler.timesteps))                     # :177
    for t in scheduler.timesteps:
        t = t.item()
        t_tensor = torch.full((y_batch.shape[0],), t, device=device, dtype=torch.long)
        x = scheduler.scale_model_input(x, t)
        noisy_residual = model_call(x, t_tensor, y_batch)
        output = scheduler.step(noisy_residual, t, x)
        if intermediates is not None:
            intermediates.append(output.prev_sample.cpu())
        if is_uc and scheduler.timestep_after_step >= t >= scheduler.timestep_end_step:
            uncertanties.append(_to_host_async(output.uncertainty))       # :193-194 (D2H every UC step)
            scores.append(_to_host_async(output.pred_epsilon))
        x = output.prev_sample
    if uncertanties:
        torch.cuda.current_stream(x.device).synchronize()
    return x, uncertanties, scores


def _finish(results, scheduler, fid_evaluator, samples_uncertanties, samples_scores):
    if fid_evaluator is not None:
        results["fid"] = fid_evaluator.compute()
    if isinstance(scheduler, SchedulerUncertaintyMixin) and samples_uncertanties:
        results["uncertainty"] = torch.cat(samples_uncertanties, dim=0).cpu()
        results["score"] = torch.cat(samples_scores, dim=0).cpu()
    return results


@torch.no_grad()
def generate_samples_model_scheduler_class_conditioned_from_tensor(X_T: torch.Tensor, y: torch.Tensor, batch_size: int,
                                                                   device, model: torch.nn.Module, scheduler,
                                                                   fid_evaluator: Any = None, save_intermediates: bool = False):
    assert X_T.shape[0] == y.shape[0], f"{X_T.shape=} {y.shape=}"
    _require_gpu(device)
    num_samples = X_T.shape[0]
    samples_gen_images, samples_uncertanties, samples_scores, samples_intermediates = [], [], [], []
    for start in range(0, num_samples, batch_size):                       # ragged last batch :162-163
        x = X_T[start:start + batch_size].to(device).float().contiguous()
        y_batch = y[start:start + batch_size].to(device)
        inter = [] if save_intermediates else None
        x, unc, sc = _loop(x, y_batch, lambda a, tt, yy: _eps3(model, a, tt, yy), scheduler, device, inter)
        if save_intermediates:
            samples_intermediates.append(torch.stack(inter, dim=1))
        if isinstance(scheduler, SchedulerUncertaintyMixin):
            samples_uncertanties.append(torch.stack(unc, dim=1))
            samples_scores.append(torch.stack(sc, dim=1))
        gen_images = ops.to_uint8(x)                                      # :203-215
        if fid_evaluator is not None:
            fid_evaluator.update(gen_images, real=False)
        samples_gen_images.append(gen_images)
    results = {"gen_images": torch.cat(samples_gen_images, dim=0).cpu()}
    if save_intermediates:
        results["intermediates"] = torch.cat(samples_intermediates, dim=0).cpu()
    return _finish(results, scheduler, fid_evaluator, samples_uncertanties, samples_scores)


@torch.no_grad()
def generate_samples_model_scheduler_class_conditioned_uvit_from_tensor(X_T: torch.Tensor, y: torch.Tensor, batch_size: int,
                                                                        uvit_ae: torch.nn.Module, scheduler,
                                                                        device="cuda", fid_evaluator: Any = None):
    assert X_T.shape[0] == y.shape[0], f"{X_T.shape=} {y.shape=}"
    _require_gpu(device)
    num_samples = X_T.shape[0]
    samples_gen_images, samples_uncertanties, samples_scores = [], [], []
    for start in range(0, num_samples, batch_size):
        x = X_T[start:start + batch_size].to(device).float().contiguous()
        y_batch = y[start:start + batch_size].to(device)
        x, unc, sc = _loop(x, y_batch, lambda a, tt, yy: uvit_ae(a, tt, yy), scheduler, device)      # :520
        if isinstance(scheduler, SchedulerUncertaintyMixin):
            samples_uncertanties.append(torch.stack(unc, dim=1))
            samples_scores.append(torch.stack(sc, dim=1))
        x = uvit_ae.decode(x)                                             # :534 (KL-VAE decoder: SURVEY 8f "next")
        gen_images = ops.to_uint8(ops._f32c(x))
        if fid_evaluator is not None:
            fid_evaluator.update(gen_images, real=False)
        samples_gen_images.append(gen_images)
    results = {"timestep": scheduler.timesteps, "gen_images": torch.cat(samples_gen_images, dim=0).cpu()}
    return _finish(results, scheduler, fid_evaluator, samples_uncertanties, samples_scores)


def _seeded_inputs(num_samples, batch_size, channels, image_size, num_classes, init_seed_rng, skip_seed):
    """Per-batch starting noise / labels of the seed-driven drivers (:46-60): one generator re-seeded with
    init_seed_rng + i_batch*skip_seed before each draw.  Drawn on the CPU generator (device-independent)."""
    xs, ys = [], []
    g = torch.Generator(device="cpu")
    done, i_batch = 0, 0
    while done < num_samples:
        seed = init_seed_rng + i_batch * skip_seed
        x = torch.randn(batch_size, channels, image_size, image_size, dtype=torch.float32, generator=g.manual_seed(seed))
        if isinstance(num_classes, int):
            yb = torch.randint(0, num_classes, (batch_size,), generator=g.manual_seed(seed))
        else:
            assert num_samples == num_classes.shape[0]
            yb = num_classes[done:done + batch_size]
            x = x[:yb.shape[0]]
        xs.append(x)
        ys.append(yb)
        done += x.shape[0]
        i_batch += 1
    return torch.cat(xs), torch.cat(ys)


@torch.no_grad()
def generate_samples_model_scheduler_class_conditioned(num_samples: int, batch_size: int, image_size: int, model, scheduler,
                                                       num_classes, device=None, fid_evaluator: Any = None,
                                                       init_seed_rng: int = 0, is_uvit: bool = False, skip_seed: int = 1,
                                                       is_cifar10: bool = False):
    """:18-124.  Note: like the reference this variant does NOT re-call set_timesteps per batch beyond the shared loop."""
    assert not is_cifar10, "the CIFAR-10 diffusers UNet2DModel is outside the accelerated hot path"
    X_T, y = _seeded_inputs(num_samples, batch_size, 4 if is_uvit else 3, image_size, num_classes, init_seed_rng, skip_seed)
    if is_uvit:
        r = generate_samples_model_scheduler_class_conditioned_uvit_from_tensor(X_T, y, batch_size, model, scheduler, device, fid_evaluator)
    else:
        r = generate_samples_model_scheduler_class_conditioned_from_tensor(X_T, y, batch_size, device, model, scheduler, fid_evaluator)
    r.update(y=y, x_t=X_T, timestep=scheduler.timesteps)
    return r


@torch.no_grad()
def generate_samples_model_scheduler_class_conditioned_uvit(num_samples: int, batch_size: int, image_size: int, uvit_ae, scheduler,
                                                            num_classes, device="cuda", fid_evaluator: Any = None,
                                                            init_seed_rng: int = 0):
    return generate_samples_model_scheduler_class_conditioned(num_samples, batch_size, image_size, uvit_ae, scheduler, num_classes,
                                                              device, fid_evaluator, init_seed_rng, is_uvit=True)
