"""Model-type dispatch of the extra denoiser calls (reference schedulers_uncertainty/traits.py:8-18)."""
import torch


class PredictorClassConditionedTrait:

    def predict_model(self, x, t):
        from ..guided_diffusion.unet_openai import UNetModel
        from ..uvit.uvit import UViT
        from ..uvit.uvit_ae import UViTAE
        if isinstance(t, float):
            t = round(t)
        if isinstance(t, int):
            t = torch.zeros(size=(x.shape[0],), dtype=torch.int64, device=x.device).fill_(t)
        if isinstance(self.unet, (UViT, UViTAE)):
            return self.unet(x, t, self.prompt_embeds)
        if isinstance(self.unet, UNetModel):
            # `[:, :3]` of the reference: the HIP output conv writes only the 3 epsilon channels
            return self.unet(x, t, y=self.prompt_embeds, out_channels_keep=3)
        return self.unet(x, t, y=self.prompt_embeds)[:, :3]
