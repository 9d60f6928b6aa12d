"""U-ViT + latent decoder wrapper (reference uvit/uvit_ae.py:4-25)."""
import torch


class UViTAE(torch.nn.Module):
    def __init__(self, uvit, autoencoder):
        super().__init__()
        self.uvit = uvit
        self.autoencoder = autoencoder
        self.in_chans = uvit.in_chans
        self.img_size = uvit.img_size
        self.patch_size = uvit.patch_size
        self.embed_dim = uvit.embed_dim

    def forward(self, x, t, y):
        return self.uvit(x, t, y)

    def decode(self, x):
        return self.autoencoder.decode(x)

    def to(self, device):
        self.uvit.to(device)
        if self.autoencoder is not None:
            self.autoencoder.to(device)
        return self
