"""Plain class-conditional DDIM sampling pipeline on the HIP library.

Same surface as reference pipeline_uncertainty/pipeline_sampler_class_conditional.py (``DiffusionClassConditional``:
constructor :11-21, ``__call__`` :75-140, ``sample`` :63-73, ``generate`` :25-60) and the same result dict (``y``, ``x_t``,
``timestep``, ``gen_images``, ``score`` [, ``intermediates``, ``fid``]).  The loops are the shared skeleton of ``_core``;
this class only says what a batch's chain returns.
"""
from typing import Dict, Optional

import torch

from ..generate_samples import _require_gpu
from ..uvit.uvit_ae import UViTAE
from . import _core


class DiffusionClassConditional:

    def __init__(self, model, scheduler, image_size: int, device, batch_size: int, init_seed_rng: int,
                 fid_evaluator: Optional[object] = None, return_intermediates: bool = False):
        _require_gpu(device)
        self.model, self.scheduler, self.device = model.to(device), scheduler, device
        self.image_size, self.batch_size, self.init_seed_rng = image_size, batch_size, init_seed_rng
        self.fid_evaluator, self.return_intermediates = fid_evaluator, return_intermediates
        self.is_uvit = isinstance(model, UViTAE)

    @torch.no_grad()
    def generate(self, X_T, y_batch):
        """One batch -> (uint8 images, scores [B, steps, ...] [, intermediates])."""
        x, scores, inter = _core.sample_chain(self.model, self.scheduler, X_T.to(self.device).float().contiguous(), y_batch, self.device,
                                              keep_scores=True, keep="intermediate" if self.return_intermediates else None)
        if self.is_uvit:
            x = self.model.decode(x)
        out = (_core.to_images(x), torch.stack(scores, dim=1))
        return out + (torch.stack(inter, dim=1),) if self.return_intermediates else out

    def __call__(self, /, num_samples: Optional[int] = None, num_classes: Optional[int] = None,
                 X_T: Optional[torch.Tensor] = None, y: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
        def chain(xb, yb):
            r = self.generate(xb, yb)
            ex = {"intermediates": r[2]} if self.return_intermediates else {}
            ex["score"] = r[1]
            return r[0], ex
        return _core.run_pipeline(self, X_T, y, num_samples, num_classes, chain)

    def sample(self, samples, classes):
        """(num_samples, num_classes) or (X_T, y), as the reference's singledispatch pair."""
        if isinstance(samples, torch.Tensor):
            return self(X_T=samples, y=classes)
        return self(num_samples=samples, num_classes=classes)
