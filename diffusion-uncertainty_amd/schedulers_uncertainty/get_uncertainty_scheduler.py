"""Scheduler factory keyed on ``args.scheduler_type`` (reference
schedulers_uncertainty/get_uncertainty_scheduler.py:13-40).  The forward-only estimators are built here
(zigzag_centered, centered, mc_dropout: the hot path; uncertainty / uncertainty_original / uncertainty_image /
uncertainty_centered_d, dpm_2_uncertainty_centered, flip: SURVEY 8f rank 4).  ``flip`` dispatches the denoiser through the
class-conditioned trait (upstream's class keeps the diffusers-style ``predict_model`` and fails on an ADM / U-ViT denoiser, see
scheduling_ddim_flip.py here); ``flip_grad`` (needs denoiser backward) raises instead of silently falling back."""
from .scheduling_ddim_flip import DDIMSchedulerUncertaintyImagenetClassConditioned as DDIMSchedulerUncertaintyFlip
from .scheduling_ddim_mc_dropout import DDIMSchedulerUncertaintyImagenetClassConditioned as DDIMSchedulerUncertaintyImagenetClassConditionedMCDropout
from .scheduling_ddim_uncertainty import DDIMSchedulerUncertaintyImagenetClassConditioned as DDIMSchedulerUncertaintyUncertainty
from .scheduling_ddim_uncertainty_centered_d import DDIMSchedulerUncertaintyImagenetClassConditioned as DDIMSchedulerUncertaintyUncertaintyCenteredD
from .scheduling_ddim_uncertainty_image import DDIMSchedulerUncertaintyImagenetClassConditioned as DDIMSchedulerUncertaintyUncertaintyImage
from .scheduling_ddim_uncertainty_centered import DDIMSchedulerUncertaintyImagenetClassConditioned as DDIMSchedulerUncertaintyUncertaintyCentered
from .scheduling_dpm_2_uncertainty_centered import KDPM2SchedulerUncertaintyImagenetClassConditioned as DPM2UncertaintyCentered
from .scheduling_ddim_uncertainty_zigzag_centered import DDIMSchedulerUncertaintyImagenetClassConditioned as DDIMSchedulerUncertaintyUncertaintyZigZagCentered

DDIMSchedulerUncertaintyUncertaintyOriginal = DDIMSchedulerUncertaintyUncertainty
_UNSUPPORTED = ("flip_grad",)


def get_uncertainty_scheduler(args, y, unet, scheduler):
    if args.scheduler_type == "uncertainty_centered":
        return DDIMSchedulerUncertaintyUncertaintyCentered.from_config(
            scheduler.config, M=args.M, after_step=args.start_step_uc, num_steps_uc=args.num_steps_uc, unet=unet,
            predict_next=args.predict_next, y=y, eta=args.eta)
    if args.scheduler_type == "uncertainty_zigzag_centered":
        return DDIMSchedulerUncertaintyUncertaintyZigZagCentered.from_config(
            scheduler.config, M=args.M, after_step=args.start_step_uc, num_steps_uc=args.num_steps_uc, unet=unet, y=y,
            eta=args.eta, num_zigzag=args.num_zigzag)
    if args.scheduler_type == "uncertainty":
        return DDIMSchedulerUncertaintyUncertainty.from_config(
            scheduler.config, M=args.M, after_step=args.start_step_uc, num_steps_uc=args.num_steps_uc, unet=unet,
            predict_next=args.predict_next, y=y, eta=args.eta)
    if args.scheduler_type == "uncertainty_original":
        return DDIMSchedulerUncertaintyUncertaintyOriginal.from_config(
            scheduler.config, M=args.M, after_step=args.start_step_uc, num_steps_uc=args.num_steps_uc, unet=unet,
            predict_next=False, y=y, eta=args.eta)
    if args.scheduler_type == "uncertainty_image":
        return DDIMSchedulerUncertaintyUncertaintyImage.from_config(
            scheduler.config, M=args.M, after_step=args.start_step_uc, num_steps_uc=args.num_steps_uc, unet=unet,
            predict_next=args.predict_next, y=y, eta=args.eta)
    if args.scheduler_type == "uncertainty_centered_d":
        return DDIMSchedulerUncertaintyUncertaintyCenteredD.from_config(
            scheduler.config, M=args.M, after_step=args.start_step_uc, num_steps_uc=args.num_steps_uc, unet=unet, y=y,
            eta=args.eta, uncertainty_distance=args.uncertainty_distance)
    if args.scheduler_type == "dpm_2_uncertainty_centered":
        return DPM2UncertaintyCentered.from_config(
            scheduler.config, M=args.M, after_step=args.start_step_uc, num_steps_uc=args.num_steps_uc, unet=unet, y=y,
            eta=args.eta)
    if args.scheduler_type == "flip":
        return DDIMSchedulerUncertaintyFlip.from_config(
            scheduler.config, after_step=args.start_step_uc, num_steps_uc=args.num_steps_uc, unet=unet, eta=args.eta,
            prompt_embeds=y)
    if args.scheduler_type in _UNSUPPORTED:
        raise NotImplementedError(f"scheduler_type={args.scheduler_type!r} is outside the accelerated hot path "
                                  "(built: uncertainty_zigzag_centered, uncertainty_centered, mc_dropout, flip, uncertainty, "
                                  "uncertainty_original, uncertainty_image, uncertainty_centered_d, dpm_2_uncertainty_centered)")
    # the reference's catch-all else-branch (:33-34) is MC dropout
    return DDIMSchedulerUncertaintyImagenetClassConditionedMCDropout.from_config(
        scheduler.config, prompt_embeds=y, unet=unet, M=args.M, after_step=args.start_step_uc,
        num_steps_uc=args.num_steps_uc, eta=args.eta)


# aliases (:39-40)
instatiate_uc_scheduler = get_uncertainty_scheduler
instatiate_uncertainty_scheduler = get_uncertainty_scheduler
