"""A scheduler's plain steps behind an object that is NOT a ``SchedulerUncertaintyMixin``.

The drivers collect uncertainty maps when the scheduler satisfies the runtime-checkable Protocol of ``mixin.py`` (the two window attributes)
and none otherwise (reference generate_samples.py:176,190-194).  The reference times its ``time_normal`` with a diffusers ``DDIMScheduler`` /
``KDPM2DiscreteScheduler`` built from the uncertainty scheduler's config (scripts/measure_times_imagenet.py:103-118).  Here the same
arithmetic is an uncertainty scheduler whose window is EMPTY - every ``step()`` is then the plain update of its family (``k_ddim_step``;
the two-stage DPM-Solver-2 update) - wrapped so that the Protocol check fails and the driver takes its plain path.
"""


class PlainScheduler:
    def __init__(self, inner):
        self._inner = inner
        self.config, self.timesteps = inner.config, inner.timesteps

    prompt_embeds = property(lambda self: self._inner.prompt_embeds, lambda self, v: setattr(self._inner, "prompt_embeds", v))

    def set_timesteps(self, *a, **k):
        self._inner.set_timesteps(*a, **k)
        self.timesteps = self._inner.timesteps

    def scale_model_input(self, x, t):
        return self._inner.scale_model_input(x, t)

    def step(self, *a, **k):
        return self._inner.step(*a, **k)
