"""KL-VAE (f8) of the U-ViT latent models: decoder on the gfx950 HIP library.

Host-side mirror of reference ``uvit/autoencoder.py`` (``FrozenAutoencoderKL`` :412-450, ``Decoder`` :292-407,
``ResnetBlock`` :74-137, ``AttnBlock`` :145-197, ``Upsample`` :29-41, ``get_model`` :463-476): same constructor
arguments and the same module nesting, so the released ``autoencoder_kl_ema.pth`` (encoder + decoder + quant convs)
loads with a strict ``load_state_dict``.  ``decode`` - the only method on the sampling path
(generate_samples.py:534) - runs in libdua_hip.so: NHWC activations, GroupNorm(32, eps 1e-6)+SiLU fused, 3x3 convs
and 1x1 / attention products on the MFMA tile engines (fp32 or split-bf16, ``precision``), nearest x2 upsampling.
The encoder modules only hold parameters (encoding is not on the path).  No CPU fallback.
"""
import torch
import torch.nn as nn

from .. import ops, range_guard


def Normalize(in_channels, num_groups=32):
    return nn.GroupNorm(num_groups=num_groups, num_channels=in_channels, eps=1e-6, affine=True)


class Upsample(nn.Module):
    def __init__(self, in_channels, with_conv):
        super().__init__()
        self.with_conv = with_conv
        if with_conv:
            self.conv = nn.Conv2d(in_channels, in_channels, kernel_size=3, stride=1, padding=1)


class Downsample(nn.Module):
    def __init__(self, in_channels, with_conv):
        super().__init__()
        self.with_conv = with_conv
        if with_conv:
            self.conv = nn.Conv2d(in_channels, in_channels, kernel_size=3, stride=2, padding=0)


class ResnetBlock(nn.Module):
    def __init__(self, *, in_channels, out_channels=None, conv_shortcut=False, dropout, temb_channels=512):
        super().__init__()
        assert not conv_shortcut and temb_channels == 0
        self.in_channels = in_channels
        self.out_channels = in_channels if out_channels is None else out_channels
        self.norm1 = Normalize(in_channels)
        self.conv1 = nn.Conv2d(in_channels, self.out_channels, kernel_size=3, stride=1, padding=1)
        self.norm2 = Normalize(self.out_channels)
        self.dropout = nn.Dropout(dropout)
        self.conv2 = nn.Conv2d(self.out_channels, self.out_channels, kernel_size=3, stride=1, padding=1)
        if self.in_channels != self.out_channels:
            self.nin_shortcut = nn.Conv2d(in_channels, self.out_channels, kernel_size=1, stride=1, padding=0)

    def run(self, ae, x):
        N, H, W, _ = x.shape
        st = ops.groupnorm_stats(x, eps=1e-6)
        h = ops.groupnorm_apply(x, st, self.norm1.weight, self.norm1.bias, act=1, split=ae.split_acts)
        h = ae.conv3x3(h, self.conv1, self.out_channels)
        st = ops.groupnorm_stats(h, eps=1e-6)
        h = ops.groupnorm_apply(h, st, self.norm2.weight, self.norm2.bias, act=1, split=ae.split_acts)
        res = x
        if self.in_channels != self.out_channels:
            res = ae.linear(x.view(N * H * W, -1), self.nin_shortcut).view(N, H, W, self.out_channels)
        return ae.conv3x3(h, self.conv2, self.out_channels, residual=res)


class AttnBlock(nn.Module):
    def __init__(self, in_channels):
        super().__init__()
        self.in_channels = in_channels
        self.norm = Normalize(in_channels)
        self.q = nn.Conv2d(in_channels, in_channels, kernel_size=1)
        self.k = nn.Conv2d(in_channels, in_channels, kernel_size=1)
        self.v = nn.Conv2d(in_channels, in_channels, kernel_size=1)
        self.proj_out = nn.Conv2d(in_channels, in_channels, kernel_size=1)

    def run(self, ae, x):
        N, H, W, C = x.shape
        T = H * W
        st = ops.groupnorm_stats(x, eps=1e-6)
        xn = ops.groupnorm_apply(x, st, self.norm.weight, self.norm.bias, act=0)
        qkv = ae.qkv(xn.view(N * T, C), self)
        a = ops.qkv_attention(qkv, N, T, 1, C, precision=ae.precision)            # single head, scale c^-1/2 (:181)
        return ae.linear(a, self.proj_out, residual=x.view(N * T, C)).view(N, H, W, C)


class _Stage(nn.Module):
    pass


class Encoder(nn.Module):
    """Parameter holder only (``Encoder.__init__`` :209-270): keeps the checkpoint's encoder.* keys loadable."""

    def __init__(self, *, ch, out_ch, ch_mult=(1, 2, 4, 8), num_res_blocks, attn_resolutions, dropout=0.0, resamp_with_conv=True,
                 in_channels, resolution, z_channels, double_z=True, **ignore_kwargs):
        super().__init__()
        assert not attn_resolutions
        self.conv_in = nn.Conv2d(in_channels, ch, kernel_size=3, stride=1, padding=1)
        in_ch_mult = (1,) + tuple(ch_mult)
        self.down = nn.ModuleList()
        block_in = ch
        for i_level in range(len(ch_mult)):
            block = nn.ModuleList()
            block_in = ch * in_ch_mult[i_level]
            block_out = ch * ch_mult[i_level]
            for _ in range(num_res_blocks):
                block.append(ResnetBlock(in_channels=block_in, out_channels=block_out, temb_channels=0, dropout=dropout))
                block_in = block_out
            down = _Stage()
            down.block = block
            down.attn = nn.ModuleList()
            if i_level != len(ch_mult) - 1:
                down.downsample = Downsample(block_in, resamp_with_conv)
            self.down.append(down)
        self.mid = _Stage()
        self.mid.block_1 = ResnetBlock(in_channels=block_in, out_channels=block_in, temb_channels=0, dropout=dropout)
        self.mid.attn_1 = AttnBlock(block_in)
        self.mid.block_2 = ResnetBlock(in_channels=block_in, out_channels=block_in, temb_channels=0, dropout=dropout)
        self.norm_out = Normalize(block_in)
        self.conv_out = nn.Conv2d(block_in, 2 * z_channels if double_z else z_channels, kernel_size=3, stride=1, padding=1)


class Decoder(nn.Module):
    def __init__(self, *, ch, out_ch, ch_mult=(1, 2, 4, 8), num_res_blocks, attn_resolutions, dropout=0.0, resamp_with_conv=True,
                 in_channels, resolution, z_channels, give_pre_end=False, tanh_out=False, **ignorekwargs):
        super().__init__()
        assert not attn_resolutions and not give_pre_end and not tanh_out and resamp_with_conv
        self.num_resolutions = len(ch_mult)
        self.num_res_blocks = num_res_blocks
        block_in = ch * ch_mult[self.num_resolutions - 1]
        self.conv_in = nn.Conv2d(z_channels, block_in, kernel_size=3, stride=1, padding=1)
        self.mid = _Stage()
        self.mid.block_1 = ResnetBlock(in_channels=block_in, out_channels=block_in, temb_channels=0, dropout=dropout)
        self.mid.attn_1 = AttnBlock(block_in)
        self.mid.block_2 = ResnetBlock(in_channels=block_in, out_channels=block_in, temb_channels=0, dropout=dropout)
        self.up = nn.ModuleList()
        for i_level in reversed(range(self.num_resolutions)):
            block = nn.ModuleList()
            block_out = ch * ch_mult[i_level]
            for _ in range(num_res_blocks + 1):
                block.append(ResnetBlock(in_channels=block_in, out_channels=block_out, temb_channels=0, dropout=dropout))
                block_in = block_out
            up = _Stage()
            up.block = block
            up.attn = nn.ModuleList()
            if i_level != 0:
                up.upsample = Upsample(block_in, resamp_with_conv)
            self.up.insert(0, up)
        self.norm_out = Normalize(block_in)
        self.conv_out = nn.Conv2d(block_in, out_ch, kernel_size=3, stride=1, padding=1)


class FrozenAutoencoderKL(nn.Module):
    def __init__(self, ddconfig, embed_dim, pretrained_path=None, scale_factor=0.18215):
        super().__init__()
        self.encoder = Encoder(**ddconfig)
        self.decoder = Decoder(**ddconfig)
        assert ddconfig["double_z"]
        self.quant_conv = nn.Conv2d(2 * ddconfig["z_channels"], 2 * embed_dim, 1)
        self.post_quant_conv = nn.Conv2d(embed_dim, ddconfig["z_channels"], 1)
        self.embed_dim = embed_dim
        self.scale_factor = scale_factor
        self.precision = "fp32"          # or "f16x3" (fp32-equivalent fp16 hi + lo split) or "bf16x3"
        self.auto_range_check = True     # f16x3: the first decode after a weight load runs with the range counters on (range_guard.py)
        self._range_checked = False
        self._range_probe = None
        self._packed = {}
        if pretrained_path is not None:
            m, u = self.load_state_dict(torch.load(pretrained_path, map_location="cpu"))
            assert len(m) == 0 and len(u) == 0
        self.eval()
        self.requires_grad_(False)

    def _apply(self, fn, *a, **k):
        self._packed = {}
        self._range_checked = False
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._packed = {}
        self._range_checked = False
        return super().load_state_dict(*a, **k)

    # ---- precision-aware FLOP carriers (same contract as guided_diffusion.unet_openai._Ctx)
    @property
    def split_acts(self):
        return "f16x3" if self.precision == "f16x3" else self.precision == "bf16x3"

    def raw_planes(self, h, resample=0):
        """Raw fp32 NHWC map (no norm in front of the up-sampler conv) -> f16x3 operand planes, optionally nearest x2: the
        GroupNorm-apply kernel with identity statistics (exact) at the raw-stream scale."""
        N, H, W, C = h.shape
        if not (self.precision == "f16x3" and C % 32 == 0):
            return ops.resample2x(h, 1) if resample > 0 else h
        key = ("identity_gn", N, C)
        if key not in self._packed:
            st = torch.zeros((N, 32, 2), device=h.device, dtype=torch.float32)
            st[:, :, 1] = 1.0
            self._packed[key] = (st, torch.ones(C, device=h.device), torch.zeros(C, device=h.device))
        st, one, zero = self._packed[key]
        return ops.groupnorm_apply(h, st, one, zero, act=0, resample=resample, groups=32, split="f16x3", split_scale=ops.F16X3_RAW_SCALE)

    def _pack(self, key, fn):
        if key not in self._packed:
            self._packed[key] = fn()
        return self._packed[key]

    def conv3x3(self, h, conv, cout, residual=None, nchw_out=False):
        if isinstance(h, ops.SplitPlanes) and h.fmt == "f16x3":
            packed = self._pack((id(conv), "w3h"), lambda: ops.pack_weight_f16x3(conv.weight.detach().contiguous()))
            return ops.conv3x3_bf16x3(h, packed, conv.bias, cout, residual=residual, nchw_out=nchw_out)
        if isinstance(h, ops.SplitPlanes) or (self.precision == "bf16x3" and h.shape[3] % 32 == 0):
            planes = self._pack((id(conv), "w3x"), lambda: ops.pack_weight_bf16x3(conv.weight.detach().contiguous()))
            return ops.conv3x3_bf16x3(h, planes, conv.bias, cout, residual=residual, nchw_out=nchw_out)
        wp = self._pack((id(conv), "w3"), lambda: ops.pack_conv3x3_weight(conv.weight.detach().contiguous()))
        return ops.conv3x3(h, wp, conv.bias, cout, residual=residual, nchw_out=nchw_out)

    def linear(self, a, mod, residual=None, alpha=1.0):
        n_out = mod.weight.shape[0]
        if self.precision == "f16x3" and a.shape[1] % 32 == 0:
            packed = self._pack((id(mod), "w1h"), lambda: ops.pack_weight_f16x3(mod.weight.detach().reshape(n_out, -1).contiguous()))
            return ops.gemm_f16x3(a, packed, n_out, bias=mod.bias, residual=residual, alpha=alpha)
        if self.precision == "bf16x3" and a.shape[1] % 32 == 0:
            planes = self._pack((id(mod), "w1x"), lambda: ops.pack_weight_bf16x3(mod.weight.detach().reshape(n_out, -1).contiguous()))
            return ops.gemm_bf16x3(a, planes, n_out, bias=mod.bias, residual=residual, alpha=alpha)
        w = self._pack((id(mod), "w1"), lambda: mod.weight.detach().reshape(n_out, -1).contiguous())
        return ops.gemm(a, w, bias=mod.bias, residual=residual, alpha=alpha)

    def qkv(self, xn, blk):
        def cat():
            C = blk.in_channels
            w = torch.cat([blk.q.weight.detach().reshape(C, C), blk.k.weight.detach().reshape(C, C), blk.v.weight.detach().reshape(C, C)])
            b = torch.cat([blk.q.bias.detach(), blk.k.bias.detach(), blk.v.bias.detach()])
            return w.contiguous(), b.contiguous()
        w, b = self._pack((id(blk), "qkv"), cat)
        if self.precision == "f16x3" and xn.shape[1] % 32 == 0:
            return ops.gemm_f16x3(xn, self._pack((id(blk), "qkvh"), lambda: ops.pack_weight_f16x3(w)), w.shape[0], bias=b,
                                  act_scale=ops.F16X3_ACT_SCALE)
        if self.precision == "bf16x3" and xn.shape[1] % 32 == 0:
            planes = self._pack((id(blk), "qkvx"), lambda: ops.pack_weight_bf16x3(w))
            return ops.gemm_bf16x3(xn, planes, w.shape[0], bias=b)
        return ops.gemm(xn, w, bias=b)

    def encode(self, x):
        raise NotImplementedError("encoding is not on the sampling hot path (SURVEY.md 8f)")

    @torch.no_grad()
    def decode(self, z):
        """uvit/autoencoder.py:446-450 + Decoder.forward :380-407."""
        if not z.is_cuda:
            raise RuntimeError("FrozenAutoencoderKL.decode: the HIP path needs GPU tensors; this package has no CPU fallback")
        if range_guard.due(self):
            return range_guard.checked_forward(self, lambda: self._decode(z))
        return self._decode(z)

    def _decode(self, z):
        d = self.decoder
        probe = self._range_probe
        B, Cz, H, W = z.shape
        zz = ops.nchw_to_nhwc(z.float().contiguous(), 4 if Cz <= 4 else Cz)                       # NHWC (padded to 4)
        w = self._pack("pq", lambda: torch.nn.functional.pad(self.post_quant_conv.weight.detach().reshape(self.post_quant_conv.out_channels, Cz),
                                                            (0, zz.shape[3] - Cz)).contiguous())
        h = ops.gemm(zz.view(B * H * W, -1), w, bias=self.post_quant_conv.bias, alpha=1.0 / self.scale_factor)   # (1/sf) z, 1x1 conv
        co = self.post_quant_conv.out_channels
        if co < 4:
            raise NotImplementedError("z_channels < 4")
        h = h.view(B, H, W, co)
        h = self.conv3x3(h, d.conv_in, d.conv_in.out_channels)
        h = d.mid.block_1.run(self, h)
        h = d.mid.attn_1.run(self, h)
        h = d.mid.block_2.run(self, h)
        if probe is not None:
            probe.note("decoder.mid", d.mid.block_1)
        for i_level in reversed(range(d.num_resolutions)):
            for blk in d.up[i_level].block:
                h = blk.run(self, h)
            if i_level != 0:
                up = d.up[i_level].upsample.conv                                                 # nearest x2 (:38)
                h = self.conv3x3(self.raw_planes(h, resample=1), up, up.out_channels)
            if probe is not None:
                probe.note(f"decoder.up.{i_level}", d.up[i_level].block[0])
        st = ops.groupnorm_stats(h, eps=1e-6)
        h = ops.groupnorm_apply(h, st, d.norm_out.weight, d.norm_out.bias, act=1, split=self.split_acts)
        return self.conv3x3(h, d.conv_out, d.conv_out.out_channels, nchw_out=True)

    def forward(self, inputs, fn):
        if fn == "decode":
            return self.decode(inputs)
        raise NotImplementedError(fn)


def get_model(pretrained_path=None, scale_factor=0.18215):
    """uvit/autoencoder.py:463-476."""
    ddconfig = dict(double_z=True, z_channels=4, resolution=256, in_channels=3, out_ch=3, ch=128, ch_mult=[1, 2, 4, 4],
                    num_res_blocks=2, attn_resolutions=[], dropout=0.0)
    return FrozenAutoencoderKL(ddconfig, 4, pretrained_path, scale_factor)
