"""Import harness for the upstream reference (THIS container only).

Used by ``tests/golden/make_golden.py`` to generate the committed golden
vectors.  ``/root/reference`` does not exist on the GPU box, so nothing under
``tests/`` imports this module at test time; tests only read the ``.npz``
fixtures next to it.

The reference's scheduler/driver files import eight symbols from ``diffusers``
(not installed, no network).  SURVEY.md Appendix C lists the minimal behaviour
that is sufficient; this file provides exactly that as stand-in modules placed
in ``sys.modules``.  All arithmetic of ``step()`` lives in the reference's own
files - the stand-in contributes only config plumbing.
"""
import enum
import glob
import inspect
import os
import sys
import types
from collections import OrderedDict

import torch

REFERENCE_ROOT = "/root/reference"


class _Config(dict):
    """attr + item accessible mapping (diffusers FrozenDict look-alike)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


class ConfigMixin:
    @property
    def config(self):
        if not hasattr(self, "_internal_dict"):
            self._internal_dict = _Config()
        return self._internal_dict

    def register_to_config(self, **kw):
        self.config.update(kw)

    @classmethod
    def from_config(cls, config, **kwargs):
        sig = inspect.signature(cls.__init__)
        names = [p for p in sig.parameters if p != "self"]
        merged = {k: v for k, v in dict(config).items() if k in names}
        merged.update({k: v for k, v in kwargs.items() if k in names})
        return cls(**merged)


def register_to_config(init):
    sig = inspect.signature(init)

    def wrapped(self, *args, **kwargs):
        bound = sig.bind(self, *args, **kwargs)
        bound.apply_defaults()
        cfg = {k: v for k, v in bound.arguments.items() if k != "self"}
        self.register_to_config(**cfg)
        init(self, *args, **kwargs)

    wrapped.__signature__ = sig
    return wrapped


class BaseOutput(OrderedDict):
    def __post_init__(self):
        for f in self.__dataclass_fields__:
            v = getattr(self, f)
            if v is not None:
                self[f] = v


def randn_tensor(shape, generator=None, device=None, dtype=None, layout=None):
    return torch.randn(shape, generator=generator, device=device, dtype=dtype)


class SchedulerMixin:
    pass


class SchedulerOutput(OrderedDict):
    """diffusers.schedulers.scheduling_utils.SchedulerOutput: a record with ``prev_sample``."""

    def __init__(self, prev_sample=None):
        super().__init__(prev_sample=prev_sample)
        self.prev_sample = prev_sample


class KarrasDiffusionSchedulers(enum.Enum):
    DDIMScheduler = 1
    DDPMScheduler = 2


class UNet2DModel(torch.nn.Module):
    pass


class ResnetBlock2D(torch.nn.Module):
    pass


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install():
    if "diffusers" in sys.modules:
        return
    _mod("diffusers")
    _mod("diffusers.configuration_utils", ConfigMixin=ConfigMixin, register_to_config=register_to_config)
    _mod("diffusers.utils", BaseOutput=BaseOutput)
    _mod("diffusers.utils.outputs", BaseOutput=BaseOutput)
    _mod("diffusers.utils.torch_utils", randn_tensor=randn_tensor)
    _mod("diffusers.schedulers")
    _mod("diffusers.schedulers.scheduling_utils", SchedulerMixin=SchedulerMixin,
         KarrasDiffusionSchedulers=KarrasDiffusionSchedulers, SchedulerOutput=SchedulerOutput)
    _mod("diffusers.models")
    _mod("diffusers.models.unets", UNet2DModel=UNet2DModel)
    _mod("diffusers.models.resnet", ResnetBlock2D=ResnetBlock2D)
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)


class _PathStandIn(str):
    """The handful of ``path.Path`` methods scripts/compute_threshold_pixel_wise.py touches (``path`` is not installed)."""

    def __truediv__(self, other):
        return _PathStandIn(os.path.join(self, str(other)))

    def joinpath(self, *parts):
        return _PathStandIn(os.path.join(self, *[str(q) for q in parts]))

    def exists(self):
        return os.path.exists(self)

    def absolute(self):
        return _PathStandIn(os.path.abspath(self))

    @property
    def parent(self):
        return _PathStandIn(os.path.dirname(self))

    def files(self, pattern="*"):
        return [_PathStandIn(f) for f in sorted(glob.glob(os.path.join(self, pattern))) if os.path.isfile(f)]

    def mkdir(self):
        os.makedirs(self, exist_ok=True)


def install_script_standins(scratch_dir):
    """Stand-ins that let ``scripts/compute_threshold_pixel_wise.py`` be imported from /root/reference (read-only) without
    touching it: ``path`` (not installed), ``diffusion_uncertainty.utils`` (imports omegaconf, not installed; only ``load_config``
    is imported, never called by ``load_uncertainty_datasets``) and ``diffusion_uncertainty.paths`` (the real module creates
    directories under the reference root at import: replaced by constants under ``scratch_dir``)."""
    install()
    _mod("path", Path=_PathStandIn)
    import diffusion_uncertainty  # noqa: F401  (namespace of the reference package; the two submodules below are stand-ins)
    root = _PathStandIn(scratch_dir)
    _mod("diffusion_uncertainty.utils", load_config=lambda *a, **k: (_ for _ in ()).throw(RuntimeError("stand-in")))
    _mod("diffusion_uncertainty.paths", ROOT=root, CONFIG=root / "config", FID=root / "results" / "fid", RESULTS=root / "results")


def load_reference_script(name):
    """Import /root/reference/scripts/<name>.py as a module (its ``main`` is guarded by ``__name__``)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_script_" + name, os.path.join(REFERENCE_ROOT, "scripts", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def ddpm_base_config(**over):
    """The keys a diffusers ``DDPMScheduler().config`` would carry."""
    cfg = _Config(num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear",
                  trained_betas=None, variance_type="fixed_small", clip_sample=True,
                  prediction_type="epsilon", thresholding=False, dynamic_thresholding_ratio=0.995,
                  clip_sample_range=1.0, sample_max_value=1.0, timestep_spacing="leading",
                  steps_offset=0, rescale_betas_zero_snr=False)
    cfg.update(over)
    return cfg
