"""Plain class-conditional DDIM sampling pipeline on the HIP library.

Mirror of reference pipeline_uncertainty/pipeline_sampler_class_conditional.py: ``DiffusionClassConditional``
(``__init__`` :11-21, ``generate`` :25-60, ``sample`` :63-73, ``__call__`` :75-140, batch helpers :142-181,
``predict_score`` :183-203).  Same constructor, same result dict (``y``, ``x_t``, ``timestep``, ``gen_images``, ``score``
[, ``intermediates``, ``fid``]); every tensor operation runs in libdua_hip.so (denoiser, DDIM step, uint8 conversion).
"""
from functools import singledispatchmethod
from typing import Dict, Optional

import torch

from .. import ops
from ..generate_samples import _require_gpu, predict_model
from ..uvit.uvit_ae import UViTAE


class DiffusionClassConditional:

    def __init__(self, model, scheduler, image_size: int, device, batch_size: int, init_seed_rng: int,
                 fid_evaluator: Optional[object] = None, return_intermediates: bool = False):
        _require_gpu(device)
        self.model = model.to(device)
        self.device = device
        self.image_size = image_size
        self.fid_evaluator = fid_evaluator
        self.batch_size = batch_size
        self.is_uvit = isinstance(model, UViTAE)
        self.is_cifar10 = False                     # diffusers UNet2DModel (CIFAR-10) is not on the accelerated path
        self.init_seed_rng = init_seed_rng
        self.scheduler = scheduler
        self.return_intermediates = return_intermediates

    @torch.no_grad()
    def generate(self, X_T, y_batch):
        X_t = X_T.cpu().clone().to(self.device)
        scores, intermediates = [], []
        self.scheduler.prompt_embeds = y_batch
        for t in self.scheduler.timesteps:
            t = t.item()
            t_tensor = torch.full((X_t.shape[0],), t, device=self.device, dtype=torch.long)
            noisy_residual = self.predict_score(X_t, y_batch, t_tensor)
            output = self.scheduler.step(noisy_residual, t, X_t)
            scores.append(noisy_residual.cpu())
            if self.return_intermediates:
                intermediates.append(output.intermediate.cpu())      # AttributeError unless the scheduler provides it (:52)
            X_t = output.prev_sample
        if self.is_uvit:
            X_t = self.model.decode(X_t)
        gen_images = ops.to_uint8(ops._f32c(X_t))                    # :55-58
        if self.return_intermediates:
            return gen_images, torch.stack(scores, dim=1), torch.stack(intermediates, dim=1)
        return gen_images, torch.stack(scores, dim=1)

    @singledispatchmethod
    def sample(self, samples, classes):
        raise NotImplementedError

    @sample.register
    def _(self, samples: int, classes: int):
        return self(num_samples=samples, num_classes=classes)

    @sample.register
    def _(self, X_T: torch.Tensor, y: torch.Tensor):
        return self(X_T=X_T, y=y)

    def __call__(self, /, num_samples: Optional[int] = None, num_classes: Optional[int] = None,
                 X_T: Optional[torch.Tensor] = None, y: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
        assert num_samples is not None or X_T is not None, "Either num_samples or X_T must be provided"
        assert num_classes is not None or y is not None, "Either num_classes or y must be provided"
        num_generated_samples = 0
        samples_scores, samples_X_t, samples_y, samples_gen_images, samples_intermediates = [], [], [], [], []
        i_batch = 0
        if num_samples is None:
            num_samples = X_T.shape[0]
        generator = torch.Generator(device=self.device)
        while num_samples > num_generated_samples:
            X_T_batch = self.get_X_T_batch(X_T, num_generated_samples, i_batch, generator)
            y_batch = self.get_y_batch(num_classes, y, num_generated_samples, i_batch, generator=generator)
            y_batch = y_batch.to(self.device)
            samples_X_t.append(X_T_batch.cpu().clone())
            samples_y.append(y_batch)
            self.scheduler.prompt_embeds = y_batch
            if self.return_intermediates:
                gen_images, scores, intermediates = self.generate(X_T=X_T_batch, y_batch=y_batch)
                samples_intermediates.append(intermediates.cpu())
            else:
                gen_images, scores = self.generate(X_T=X_T_batch, y_batch=y_batch)
            samples_scores.append(scores)
            num_generated_samples += gen_images.shape[0]
            if self.fid_evaluator is not None:
                self.fid_evaluator.update(gen_images, real=False)
            samples_gen_images.append(gen_images)
            i_batch += 1
        results = {"y": torch.cat(samples_y, dim=0).cpu(),
                   "x_t": torch.cat(samples_X_t, dim=0).cpu(),
                   "timestep": self.scheduler.timesteps,
                   "gen_images": torch.cat(samples_gen_images, dim=0).cpu()}
        if self.return_intermediates:
            results["intermediates"] = torch.cat(samples_intermediates, dim=0).cpu()
        if self.fid_evaluator is not None:
            results["fid"] = self.fid_evaluator.compute()
        results["score"] = torch.cat(samples_scores, dim=0).cpu()
        return results

    def get_y_batch(self, num_classes, y, num_generated_samples, i_batch, generator):
        if y is not None:
            return y[num_generated_samples:num_generated_samples + self.batch_size]
        return torch.randint(0, num_classes, (self.batch_size,), device=self.device,
                             generator=generator.manual_seed(self.init_seed_rng + i_batch))

    def get_X_T_batch(self, X_T, num_generated_samples, i_batch, generator):
        if X_T is not None:
            return X_T[num_generated_samples:num_generated_samples + self.batch_size]
        c = 4 if self.is_uvit else 3
        return torch.randn(self.batch_size, c, self.image_size, self.image_size, device=self.device, dtype=torch.float32,
                           generator=generator.manual_seed(self.init_seed_rng + i_batch))

    def predict_score(self, input, y_batch, t_tensor):
        return predict_model(self.model, input, t_tensor, y_batch)
