"""Scheduler factory keyed on ``args.scheduler_type`` (reference
schedulers_uncertainty/get_uncertainty_scheduler.py:13-40).  The three estimators of the accelerated hot
path are built here; the reference's other 12 scheduler_type values are out of scope
and raise instead of silently falling back."""
from .scheduling_ddim_mc_dropout import DDIMSchedulerUncertaintyImagenetClassConditioned as DDIMSchedulerUncertaintyImagenetClassConditionedMCDropout
from .scheduling_ddim_uncertainty_centered import DDIMSchedulerUncertaintyImagenetClassConditioned as DDIMSchedulerUncertaintyUncertaintyCentered
from .scheduling_ddim_uncertainty_zigzag_centered import DDIMSchedulerUncertaintyImagenetClassConditioned as DDIMSchedulerUncertaintyUncertaintyZigZagCentered

_UNSUPPORTED = ("uncertainty", "uncertainty_image", "flip", "flip_grad", "uncertainty_original", "uncertainty_centered_d",
                "dpm_2_uncertainty_centered")


def get_uncertainty_scheduler(args, y, unet, scheduler):
    if args.scheduler_type == "uncertainty_centered":
        return DDIMSchedulerUncertaintyUncertaintyCentered.from_config(
            scheduler.config, M=args.M, after_step=args.start_step_uc, num_steps_uc=args.num_steps_uc, unet=unet,
            predict_next=args.predict_next, y=y, eta=args.eta)
    if args.scheduler_type == "uncertainty_zigzag_centered":
        return DDIMSchedulerUncertaintyUncertaintyZigZagCentered.from_config(
            scheduler.config, M=args.M, after_step=args.start_step_uc, num_steps_uc=args.num_steps_uc, unet=unet, y=y,
            eta=args.eta, num_zigzag=args.num_zigzag)
    if args.scheduler_type in _UNSUPPORTED:
        raise NotImplementedError(f"scheduler_type={args.scheduler_type!r} is outside the accelerated hot path "
                                  "(uncertainty_zigzag_centered, uncertainty_centered, mc_dropout)")
    # the reference's catch-all else-branch (:33-34) is MC dropout
    return DDIMSchedulerUncertaintyImagenetClassConditionedMCDropout.from_config(
        scheduler.config, prompt_embeds=y, unet=unet, M=args.M, after_step=args.start_step_uc,
        num_steps_uc=args.num_steps_uc, eta=args.eta)


# aliases (:39-40)
instatiate_uc_scheduler = get_uncertainty_scheduler
instatiate_uncertainty_scheduler = get_uncertainty_scheduler
