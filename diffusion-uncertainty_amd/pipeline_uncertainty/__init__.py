"""Class-conditional sampling pipelines (reference ``diffusion_uncertainty/pipeline_uncertainty/``): the plain DDIM
pipeline and the forward-only posterior-guided pipeline.  The gradient / second-order variants need a
backward pass through the denoiser and are out of scope."""
