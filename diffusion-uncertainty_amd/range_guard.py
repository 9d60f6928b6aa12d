"""First-forward range check of the f16x3 operand planes.

The fp32-equivalence of precision ``"f16x3"`` (outputs within 1e-3 fp32 per pixel of the reference CPU path, BASELINE.json
north_star) rests on static power-of-two plane scales: an operand whose scaled magnitude leaves fp16's range saturates its
hi half and the pair no longer carries 22 bits (``ops.py`` F16X3_ACT_SCALE, ``csrc/common.h``).  Seeded random weights stay
far inside the range; a real checkpoint need not.  So the FIRST forward a model runs after its weights were loaded or edited
executes with the library's counters on (``dua_range_check``), block by block, and a non-zero saturation count is reported
once as an ``F16RangeWarning`` naming the blocks and their layer classes; the counters are then switched off again (their
cost when on is ~0.6 % of a forward, and one forward in hundreds pays it).  The checked forward computes the same bits as an
unchecked one.  ``model.auto_range_check = False`` opts out; a caller that has the counters on itself (``--check-range``)
is left alone.  ``model.on_range_violation`` chooses what a non-zero count does: ``"warn"`` (default), ``"raise"``
(``F16RangeError``: a batch job that must not produce silently degraded maps) or ``"fp32"`` (warn, switch the model to the
fp32 engines - 2.5x slower, no plane scales to leave - and re-run this forward there).
"""
import warnings

import torch

from . import ops


class F16RangeWarning(RuntimeWarning):
    """Operands of an f16x3 forward left fp16's range after their static scale: the run is not fp32-equivalent."""


class F16RangeError(RuntimeError):
    """Raised instead of the warning when ``model.on_range_violation == "raise"``."""


class RangeProbe:
    """Collects (block name, layer classes, saturated, subnormal) while a checked forward runs."""

    def __init__(self):
        self.rows = []

    def note(self, name, module):
        sat, sub = ops.range_check_read(reset=True)                     # synchronises: once per block of ONE forward
        if sat or sub:
            kinds = "+".join(type(m).__name__ for m in module) if isinstance(module, torch.nn.Sequential) else type(module).__name__
            self.rows.append((name, kinds, sat, sub))


def due(model):
    return (model.precision == "f16x3" and getattr(model, "auto_range_check", True) and not getattr(model, "_range_checked", False)
            and not ops.range_check_is_on() and not torch.cuda.is_current_stream_capturing())


def checked_forward(model, run):
    """``run()`` = the model's eager forward on the current stream, with ``model._range_probe`` visible to its block loop."""
    probe = RangeProbe()
    ops.range_check(True)
    model._range_probe = probe
    try:
        out = run()
        probe.note("head", model)                                       # what ran after the last block
    finally:
        model._range_probe = None
        ops.range_check(False)
    model._range_checked = True
    sat = sum(r[2] for r in probe.rows)
    sub = sum(r[3] for r in probe.rows)
    model.range_report = dict(saturated=sat, subnormal=sub, blocks=list(probe.rows))
    if sat:
        worst = sorted((r for r in probe.rows if r[2]), key=lambda r: -r[2])[:6]
        where = ", ".join(f"{n} [{k}]: {s}" for n, k, s, _ in worst)
        policy = getattr(model, "on_range_violation", "warn")
        msg = (f"{type(model).__name__} (precision f16x3): {sat} operand(s) left fp16's range after the static plane scale in the "
               f"first forward after the weights were loaded ({where}); the fp32-equivalent claim does NOT hold for these weights")
        if policy == "raise":
            model._range_checked = False                                # stays armed: every attempt on these weights raises
            raise F16RangeError(msg + " - use precision 'fp32'")
        if policy == "fp32":
            warnings.warn(msg + " - switching this model to precision 'fp32' (on_range_violation = 'fp32') and re-running the forward there",
                          F16RangeWarning, stacklevel=3)
            model.precision = "fp32"
            return run()
        warnings.warn(msg + " - use precision 'fp32'", F16RangeWarning, stacklevel=3)
    return out
