"""Uncertainty-threshold guided Stable-Diffusion sampling loop on the HIP library (BASELINE configs[4]).

Mirror of the denoising loop of reference ``pipeline_uncertainty/pipeline_stable_diffusion_uncertainty_guided.py``
``StableDiffusionPipelineUncertainty.__call__`` (:719-776): CFG-doubled latent, UNet call, CFG combine (:757-759), inside
the window ``start_step_uc <= i < start_step_uc + num_steps_uc`` the percentile-thresholded posterior update of the score
(``get_uncertainty_guided_score_with_percentile`` with ``alpha_hat_t = scheduler.alphas_cumprod[t]`` :770), scheduler step
(:773-774).  The latent UNet and the scheduler are this package's HIP implementations
(stable_diffusion/unet_2d_condition.py, schedulers_uncertainty/scheduling_pndm.py).

Outside the hot path and NOT built (third-party models of the un-vendored diffusers / transformers): the CLIP tokenizer and
text encoder (``encode_prompt``), the VAE decode and the safety checker - the caller passes ``prompt_embeds`` /
``negative_prompt_embeds`` ([1, 77, 768] each) and receives latents (``output_type='latent'``, the reference's own
branch :778-782).  Both updates of the guidance function are available: the posterior blend and the gradient update
(``uncertainty_guidance.use_posterior = False``, the reference script's default), the latter through the UNet's explicit input
gradient (``forward_with_tape`` / ``backward_sample``).
"""
import torch

from .. import ops, uncertainty_guidance
from ..schedulers_uncertainty.noise import HostNoise


class StableDiffusionPipelineOutput(dict):
    def __init__(self, images, nsfw_content_detected=None):
        super().__init__(images=images, nsfw_content_detected=nsfw_content_detected)
        self.images, self.nsfw_content_detected = images, nsfw_content_detected


class StableDiffusionPipelineUncertainty:
    vae_scale_factor = 8

    def __init__(self, unet, scheduler):
        self.unet, self.scheduler = unet, scheduler
        self.noise_source = HostNoise()          # prepare_latents: randn in the reference's order (torch generator on the host)

    @torch.no_grad()
    def __call__(self, prompt=None, height=512, width=512, num_inference_steps=50, guidance_scale=7.5, negative_prompt=None,
                 num_images_per_prompt=1, eta=0.0, generator=None, latents=None, prompt_embeds=None, negative_prompt_embeds=None,
                 output_type="latent", return_dict=True, start_step_uc=0, num_steps_uc=0, percentile=0.95, lr=1.0,
                 num_uncertainty_samples=5, cross_attention_kwargs=None, device="cuda"):
        if prompt is not None or prompt_embeds is None:
            raise NotImplementedError("the CLIP text encoder is outside the accelerated path: pass prompt_embeds "
                                      "(and negative_prompt_embeds for classifier-free guidance)")
        if output_type != "latent":
            raise NotImplementedError("the VAE decode is outside the accelerated path: output_type must be 'latent'")
        do_cfg = guidance_scale > 1.0                                                       # :697
        batch = prompt_embeds.shape[0] * num_images_per_prompt
        pe = prompt_embeds.to(device=device, dtype=torch.float32)
        if do_cfg:
            assert negative_prompt_embeds is not None, "classifier-free guidance needs negative_prompt_embeds"
            pe = torch.cat([negative_prompt_embeds.to(device=device, dtype=torch.float32), pe])   # :715-716
        self.scheduler.set_timesteps(num_inference_steps, device=None)                     # :719
        timesteps = self.scheduler.timesteps
        if latents is None:                                                                 # prepare_latents
            shape = (batch, self.unet.config["in_channels"], height // self.vae_scale_factor, width // self.vae_scale_factor)
            if generator is not None:
                latents = torch.randn(shape, generator=generator, dtype=torch.float32).to(device)
            else:
                latents = self.noise_source.draw(shape, device)
        latents = ops._f32c(latents.to(device))                                            # init_noise_sigma is 1 for PNDM
        for i, t in enumerate(timesteps):
            x_in = torch.cat([latents] * 2) if do_cfg else latents                          # :744
            x_in = self.scheduler.scale_model_input(x_in, t)
            noise_pred = self.unet(x_in, t, encoder_hidden_states=pe, cross_attention_kwargs=cross_attention_kwargs,
                                   return_dict=False)[0]                                    # :747-753
            if do_cfg:                                                                      # :756-759
                un, tx = noise_pred.chunk(2)
                un, tx = ops._f32c(un), ops._f32c(tx)
                noise_pred = ops.axpby(un, ops.axpby(tx, un, 1.0, -1.0), 1.0, guidance_scale)
            if start_step_uc <= i < start_step_uc + num_steps_uc:                           # :765-770
                lr_i = lr(i) if callable(lr) else lr
                noise_pred = uncertainty_guidance.get_uncertainty_guided_score_with_percentile(
                    noise_pred, x_in, t, pe, self.unet, alpha_hat_t=self.scheduler.alphas_cumprod[int(t)], percentile=percentile,
                    guidance_scale=guidance_scale, model_type="stable-diffusion", lr=lr_i,
                    num_uncertainty_samples=num_uncertainty_samples,
                    extra_diffusion_kwargs=dict(cross_attention_kwargs=cross_attention_kwargs, return_dict=False))
            latents = self.scheduler.step(noise_pred, t, latents, return_dict=True)["prev_sample"]   # :773-774
        if not return_dict:
            return (latents, None)
        return StableDiffusionPipelineOutput(images=latents, nsfw_content_detected=None)
