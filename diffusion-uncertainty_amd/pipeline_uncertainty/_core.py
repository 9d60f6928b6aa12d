"""Shared sampling skeleton of the class-conditional pipelines (host side).

The reference spells the same three loops out in every pipeline class (batch loop over X_T / labels or over seeds, timestep
loop, result collation: pipeline_sampler_class_conditional.py:75-140,
pipeline_sampler_class_conditional_uncertainty_guided_posterior_distribution.py:88-175).  Here they exist once:

  * ``starting_batches``  - the batch source: slices of caller-provided tensors, or draws from a device generator re-seeded
                            with ``seed0 + batch index`` before every draw (the reference's ``get_X_T_batch`` / ``get_y_batch``);
  * ``sample_chain``      - one batch through ``scheduler.timesteps`` (denoiser call -> ``scheduler.step``) with an optional
                            per-timestep ``revise`` callback that may replace the step's output (the guided pipelines);
  * ``run_pipeline``      - drives both and collates the result dict (``y``, ``x_t``, ``timestep``, ``gen_images``, extras).

All tensor arithmetic runs in libdua_hip.so (denoiser, scheduler kernels, dua_to_uint8).
"""
import torch

from .. import ops
from ..generate_samples import predict_model


def starting_batches(X_T, y, num_samples, num_classes, batch_size, device, shape_chw, seed0):
    """Yield (x_T batch, label batch) until ``num_samples`` starting points have been handed out."""
    assert num_samples is not None or X_T is not None, "Either num_samples or X_T must be provided"
    assert num_classes is not None or y is not None, "Either num_classes or y must be provided"
    total = X_T.shape[0] if num_samples is None else num_samples
    gen = torch.Generator(device=device)
    done = k = 0
    while done < total:
        if X_T is not None:
            xb = X_T[done:done + batch_size]
        else:
            xb = torch.randn((batch_size,) + tuple(shape_chw), device=device, dtype=torch.float32, generator=gen.manual_seed(seed0 + k))
        if y is not None:
            yb = y[done:done + batch_size]
        else:
            yb = torch.randint(0, num_classes, (batch_size,), device=device, generator=gen.manual_seed(seed0 + k))
        yield xb, yb.to(device)
        done += xb.shape[0]
        k += 1


def sample_chain(model, scheduler, x, yb, device, revise=None, keep_scores=False, keep=None):
    """x through every timestep of ``scheduler``.  ``revise(i, t, t_tensor, x, eps, out)`` may return a replacement for the
    step output.  Returns (x_0, [eps per timestep on the CPU] if keep_scores, [out.<keep> per timestep on the CPU] if keep)."""
    scores, kept = [], []
    scheduler.prompt_embeds = yb
    for i, t in enumerate(scheduler.timesteps):
        t = t.item()
        t_tensor = torch.full((x.shape[0],), t, device=device, dtype=torch.long)
        eps = predict_model(model, x, t_tensor, yb)
        out = scheduler.step(eps, t, x)
        if revise is not None:
            revised = revise(i, t, t_tensor, x, eps, out)
            out = out if revised is None else revised
        if keep_scores:
            scores.append(eps.cpu())
        if keep is not None:
            kept.append(getattr(out, keep).cpu())          # AttributeError unless the scheduler provides it, as upstream
        x = out.prev_sample
    return x, scores, kept


def run_pipeline(pipe, X_T, y, num_samples, num_classes, chain):
    """``chain(x_T batch on the device, labels) -> (uint8 images, {extra name: per-batch tensor})``; collates like the reference."""
    shape = (4 if pipe.is_uvit else 3, pipe.image_size, pipe.image_size)
    xs, ys, imgs, extras = [], [], [], {}
    for xb, yb in starting_batches(X_T, y, num_samples, num_classes, pipe.batch_size, pipe.device, shape, pipe.init_seed_rng):
        xs.append(xb.cpu().clone())
        ys.append(yb)
        g, ex = chain(xb.to(pipe.device).float().contiguous(), yb)
        if pipe.fid_evaluator is not None:
            pipe.fid_evaluator.update(g, real=False)
        imgs.append(g)
        for k, v in ex.items():
            extras.setdefault(k, []).append(v.cpu())
    res = {"y": torch.cat(ys, dim=0).cpu(), "x_t": torch.cat(xs, dim=0).cpu(), "timestep": pipe.scheduler.timesteps,
           "gen_images": torch.cat(imgs, dim=0).cpu()}
    if pipe.fid_evaluator is not None:
        res["fid"] = pipe.fid_evaluator.compute()
    for k, v in extras.items():
        res[k] = torch.cat(v, dim=0)
    return res


def to_images(x):
    return ops.to_uint8(ops._f32c(x))
