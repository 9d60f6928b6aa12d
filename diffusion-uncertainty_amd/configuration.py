"""Config plumbing the reference takes from diffusers (ConfigMixin / register_to_config / BaseOutput /
DDPMScheduler config container).  The reference builds every uncertainty scheduler through
``Cls.from_config(scheduler.config, **kw)`` (get_uncertainty_scheduler.py:13-35), which keeps only keyword
arguments named in ``Cls.__init__`` - reproduced here so the factory call sites read unchanged."""
import enum
import inspect
from collections import OrderedDict


class ConfigDict(dict):
    """attr + item accessible, mutable (reference drivers assign config.after_step / num_steps_uc)."""

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
        if "_internal_dict" not in self.__dict__:
            self.__dict__["_internal_dict"] = ConfigDict()
        return self.__dict__["_internal_dict"]

    def register_to_config(self, **kw):
        self.config.update(kw)

    @classmethod
    def from_config(cls, config, **kwargs):
        names = [p for p in inspect.signature(cls.__init__).parameters if p != "self"]
        merged = {k: v for k, v in dict(config).items() if k in names}
        merged.update({k: v for k, v in kwargs.items() if k in names})
        return cls(**merged)


def register_to_config(init):
    sig = inspect.signature(init)

    def wrapped(self, *args, **kwargs):
        bound = sig.bind(self, *args, **kwargs)
        bound.apply_defaults()
        self.register_to_config(**{k: v for k, v in bound.arguments.items() if k != "self"})
        init(self, *args, **kwargs)

    wrapped.__signature__ = sig
    wrapped.__doc__ = init.__doc__
    return wrapped


class SchedulerMixin:
    pass


class KarrasDiffusionSchedulers(enum.Enum):
    DDIMScheduler = 1
    DDPMScheduler = 2


class BaseOutput(OrderedDict):
    """Record with attribute AND mapping access; undeclared attributes may be attached
    (the reference sets ``output.pred_epsilon`` ad hoc, zigzag_centered.py:557)."""

    def __init__(self, **kw):
        super().__init__()
        for k, v in kw.items():
            setattr(self, k, v)

    def __setattr__(self, k, v):
        object.__setattr__(self, k, v)
        if v is not None:
            OrderedDict.__setitem__(self, k, v)

    def to_tuple(self):
        return tuple(self.values())


class DDPMScheduler(ConfigMixin, SchedulerMixin):
    """Base-config carrier (the role diffusers.DDPMScheduler plays at init_model.py:39,136 and
    load_pretrained_models.py:60): only ``.config`` is consumed downstream."""

    @register_to_config
    def __init__(self, num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear",
                 trained_betas=None, variance_type="fixed_small", clip_sample=True, prediction_type="epsilon",
                 thresholding=False, dynamic_thresholding_ratio=0.995, clip_sample_range=1.0, sample_max_value=1.0,
                 timestep_spacing="leading", steps_offset=0, rescale_betas_zero_snr=False):
        pass
