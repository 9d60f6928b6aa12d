"""ADM (guided-diffusion) UNet on the gfx950 HIP library.

Host-side mirror of reference ``guided_diffusion/unet_openai.py``: same constructor arguments
(``UNetModel.__init__`` :425-617), same module nesting and therefore the same state-dict keys as the
OpenAI checkpoints (``input_blocks.N.0.in_layers.2.weight``, ``...1.qkv.weight`` ...), same call
convention ``model(x[B,C,H,W], t[B] int64, y=labels) -> [B,out_channels,H,W]`` and the same assertion on a
missing/extra ``y`` (:644-646).  The nn modules below only HOLD parameters; ``forward`` drives the
C ABI (include/dua_hip.h): activations are NHWC fp32 in HBM, every conv / linear / attention product
runs on the fp32 MFMA tile engine, GroupNorm+SiLU(+FiLM, +dropout mask, +2x resampling) is one fused
kernel, the skip concat of :662 is never materialised (two-source GroupNorm / GEMM operands).

Only the configuration family the reference instantiates (init_model.py:21,45-47) is supported:
dims=2, use_scale_shift_norm=True, resblock_updown=True.  There is no CPU path.
"""
import torch
import torch.nn as nn

from .. import _lib, ops, range_guard
from ..utils import params_generation
from .nn import normalization


def zero_module(module):
    """guided_diffusion/nn.py:68-74 - kept so a fresh model matches the reference's init."""
    for p in module.parameters():
        p.detach().zero_()
    return module


class TimestepBlock(nn.Module):
    pass


class TimestepEmbedSequential(nn.Sequential, TimestepBlock):
    def run(self, ctx, h, skip=None, name=None):
        hooks = ctx.model.activation_noise
        for j, layer in enumerate(self):
            if isinstance(layer, ResBlock):
                h = layer.run(ctx, h, skip)
                skip = None
            elif isinstance(layer, AttentionBlock):
                h = layer.run(ctx, h)
            else:                                   # input conv (Cin padded to 4: always the fp32 small-Cin kernel)
                h = ops.conv3x3(h, ctx.packed(layer, "w3"), layer.bias, layer.out_channels, colsum=ctx.model.fuse_gn_stats)
            if hooks and f"{name}.{j}" in hooks:
                h = ctx.add_activation_noise(h, hooks[f"{name}.{j}"])
        return h


class Upsample(nn.Module):
    def __init__(self, channels, use_conv, dims=2, out_channels=None):
        super().__init__()
        assert not use_conv and dims == 2, "only the conv-free resampler of resblock_updown=True is built"
        self.channels = channels


class Downsample(Upsample):
    pass


class ResBlock(TimestepBlock):
    """unet_openai.py:144-257."""

    def __init__(self, channels, emb_channels, dropout, out_channels=None, use_conv=False,
                 use_scale_shift_norm=False, dims=2, use_checkpoint=False, up=False, down=False):
        super().__init__()
        assert dims == 2 and use_scale_shift_norm and not use_conv, "unsupported ResBlock variant"
        self.channels, self.emb_channels, self.dropout = channels, emb_channels, dropout
        self.out_channels = out_channels or channels
        self.use_scale_shift_norm = use_scale_shift_norm
        self.in_layers = nn.Sequential(normalization(channels), nn.SiLU(),
                                       nn.Conv2d(channels, self.out_channels, 3, padding=1))
        self.updown = up or down
        self.resample = 1 if up else (-1 if down else 0)
        if up:
            self.h_upd, self.x_upd = Upsample(channels, False), Upsample(channels, False)
        elif down:
            self.h_upd, self.x_upd = Downsample(channels, False), Downsample(channels, False)
        else:
            self.h_upd = self.x_upd = nn.Identity()
        self.emb_layers = nn.Sequential(nn.SiLU(), nn.Linear(emb_channels, 2 * self.out_channels))
        self.out_layers = nn.Sequential(normalization(self.out_channels), nn.SiLU(), nn.Dropout(p=dropout),
                                        zero_module(nn.Conv2d(self.out_channels, self.out_channels, 3, padding=1)))
        if self.out_channels == channels:
            self.skip_connection = nn.Identity()
        else:
            self.skip_connection = nn.Conv2d(channels, self.out_channels, 1)

    def run(self, ctx, x, skip=None):
        """x (and optional skip = second half of the channel concat): NHWC."""
        gn1, conv1 = self.in_layers[0], self.in_layers[2]
        gn2, drop, conv2 = self.out_layers[0], self.out_layers[2], self.out_layers[3]
        N, H, W, _ = x.shape
        if ctx.block_abi and x.shape[3] % 32 == 0 and (skip is None or skip.shape[3] % 32 == 0) and self.out_channels % 32 == 0:
            # the whole block as ONE C-ABI call (dua_adm_resblock_fwd: the same kernels in the same order, bit-identical)
            kind3, kind1 = ("w3h", "w1h") if ctx.model.precision == "f16x3" else ("w3", "w1")
            ident = isinstance(self.skip_connection, nn.Identity)
            mask = ctx.dropout_mask((N, H, W, self.out_channels), drop.p) if (drop.training and drop.p > 0 and not self.updown) else None
            if not (drop.training and drop.p > 0 and self.updown):          # (a dropout mask on a resampling block: composed path)
                Hb, Wb = (2 * H, 2 * W) if self.resample > 0 else ((H // 2, W // 2) if self.resample < 0 else (H, W))
                cin = x.shape[3] + (0 if skip is None else skip.shape[3])
                kw = {0: None, 2: "w3w" if ctx.model.precision == "f16x3" else "w3wf", 4: "w3w4" if ctx.model.precision == "f16x3" else "w3w4f"}
                m1, m2 = ctx.wino_form(N, Hb, Wb, cin, self.out_channels), ctx.wino_form(N, Hb, Wb, self.out_channels, self.out_channels)
                w1 = ctx.packed(conv1, kw[m1]) if m1 else None
                w2 = ctx.packed(conv2, kw[m2]) if m2 else None
                # (the direct-form weights are packed only for a conv that does not run in the Winograd form: ~2 GB less for ADM-128)
                return ops.adm_resblock(x, skip, (gn1.weight, gn1.bias), ctx.packed(conv1, kind3) if w1 is None else None, conv1.bias, ctx.film(self),
                                        (gn2.weight, gn2.bias), ctx.packed(conv2, kind3) if w2 is None else None, conv2.bias,
                                        None if ident else ctx.packed(self.skip_connection, kind1),
                                        None if ident else self.skip_connection.bias, self.out_channels, updown=self.resample,
                                        dropmask=mask, precision=("f16w" if ctx.model.wino_single else ctx.model.precision),
                                        want_colsum=ctx.model.fuse_gn_stats, conv1_wino=w1, conv2_wino=w2, conv1_wino_m=m1, conv2_wino_m=m2)
        # (mode "f16w": ctx.conv_split hands the GroupNorm writer the hi-images-only operand here too, so this per-op composition runs the
        # single-product Winograd conv exactly like the block-level call above)
        st = ops.groupnorm_stats(x, skip)
        Hc = H * 2 if self.resample > 0 else (H // 2 if self.resample < 0 else H)
        Wc = W * 2 if self.resample > 0 else (W // 2 if self.resample < 0 else W)
        cin = x.shape[3] + (0 if skip is None else skip.shape[3])
        h = ops.groupnorm_apply(x, st, gn1.weight, gn1.bias, x2=skip, act=1, resample=self.resample,
                                split=ctx.conv_split(Wc, (N, Hc, Wc, cin, self.out_channels) if x.shape[3] % 32 == 0 else None))
        if self.updown:
            x = ops.resample2x(x, self.resample)                                   # x_upd :243
        h = ctx.conv3x3(h, conv1, self.out_channels, colsum=True)      # epilogue leaves GroupNorm statistics for out_layers[0]
        film = ctx.film(self)                                                              # :245 emb_layers(emb)
        st2 = ops.groupnorm_stats(h)
        mask = None
        if drop.training and drop.p > 0:
            mask = ctx.dropout_mask(h.shape, drop.p)
        h = ops.groupnorm_apply(h, st2, gn2.weight, gn2.bias, film=film, act=1, dropmask=mask,
                                split=ctx.conv_split(Wc, (N, Hc, Wc, self.out_channels, self.out_channels)))                  # :249-253
        if isinstance(self.skip_connection, nn.Identity):
            assert skip is None
            res = x
        else:
            sc = self.skip_connection
            Hx, Wx = x.shape[1], x.shape[2]
            res = ctx.linear(x.view(N * Hx * Wx, -1), sc, a2=(skip.view(N * Hx * Wx, -1) if skip is not None else None))
            res = res.view(N, Hx, Wx, self.out_channels)
        return ctx.conv3x3(h, conv2, self.out_channels, residual=res, colsum=True)                      # :257


class QKVAttentionLegacy(nn.Module):
    """Marker for the head-first qkv split (unet_openai.py:329-355)."""

    def __init__(self, n_heads):
        super().__init__()
        self.n_heads = n_heads


class QKVAttention(nn.Module):
    """Marker for the chunk-first qkv split (unet_openai.py:362-390)."""

    def __init__(self, n_heads):
        super().__init__()
        self.n_heads = n_heads


class AttentionBlock(nn.Module):
    """unet_openai.py:260-306."""

    def __init__(self, channels, num_heads=1, num_head_channels=-1, use_checkpoint=False, use_new_attention_order=False):
        super().__init__()
        self.channels = channels
        if num_head_channels == -1:
            self.num_heads = num_heads
        else:
            assert channels % num_head_channels == 0, \
                f"q,k,v channels {channels} is not divisible by num_head_channels {num_head_channels}"
            self.num_heads = channels // num_head_channels
        self.norm = normalization(channels)
        self.qkv = nn.Conv1d(channels, channels * 3, 1)
        self.attention = QKVAttention(self.num_heads) if use_new_attention_order else QKVAttentionLegacy(self.num_heads)
        self.proj_out = zero_module(nn.Conv1d(channels, channels, 1))

    def run(self, ctx, x):
        N, H, W, C = x.shape
        T = H * W
        ch = C // self.num_heads
        chmax = 256 if ctx.model.precision == "f16x3" else 128          # (fp32 flash kernel: the GEMM pair is faster above 128)
        if ctx.block_abi and C % 32 == 0 and ch <= chmax and ch % 8 == 0 and ctx.model.presplit_qkv and ops.ATTENTION_MODE == "auto":
            f16 = ctx.model.precision == "f16x3"
            qk, bq = ctx.packed(self, "qkvh" if f16 else "qkv")
            return ops.adm_attnblock(x, self.num_heads, (self.norm.weight, self.norm.bias), qk, bq,
                                     ctx.packed(self.proj_out, "w1h" if f16 else "w1"), self.proj_out.bias,
                                     precision=ctx.model.precision, want_colsum=ctx.model.fuse_gn_stats)
        st = ops.groupnorm_stats(x)
        # bf16x3 mode: the normalised tokens go straight out as split bf16 planes and the qkv projection runs on the
        # LDS-DMA kernel (one tap)
        xn = ops.groupnorm_apply(x, st, self.norm.weight, self.norm.bias, act=0,
                                 split=ctx.gemm_split if (ctx.model.presplit_qkv and C % 32 == 0) else False)
        qkv = ctx.qkv(xn if isinstance(xn, ops.SplitPlanes) else xn.view(N * T, C), self)
        a = ops.qkv_attention(qkv, N, T, self.num_heads, C // self.num_heads,
                              precision="f16x3" if ctx.model.precision == "f16x3" else ("bf16x3" if ctx.split_mode else "fp32"),
                              planes_out=(ctx.gemm_split == "f16x3"))
        if isinstance(a, ops.SplitPlanes):        # attention output already in the operand format: proj_out on the LDS-DMA kernel
            key = (id(self.proj_out), "w1hp")
            if key not in ctx.model._packed:
                w = self.proj_out.weight.detach()
                ctx.model._packed[key] = ops.pack_weight_f16x3(w.reshape(w.shape[0], w.shape[1]).contiguous())
            out2d = ops.gemm_bf16x3_presplit(a, ctx.model._packed[key], C, bias=self.proj_out.bias, residual=x.view(N * T, C),
                                             colsum=ctx.model.fuse_gn_stats)
        else:
            out2d = ctx.linear(a, self.proj_out, residual=x.view(N * T, C), colsum=True)
        out = out2d.view(N, H, W, C)
        ws = getattr(out2d, "_dua_colsums_rows", None)
        if ws is not None and T % 64 == 0:
            out._dua_colsums = ws                     # statistics for the next block's GroupNorm
        return out


class _Ctx:
    """Per-forward state: timestep embedding, packed-weight cache, dropout-mask source."""

    def __init__(self, model):
        self.model = model
        self.emb = None
        self.film_all = None
        # split modes: GroupNorm+SiLU writes the conv input directly in the operand format of the conv kernel
        # (False | True = bf16x3 planes | "f16x3" planes)
        # "f16x3" (fp32-equivalent): 3x3 convs and qkv on fp16 hi / lo planes written by the GroupNorm kernels, 1x1 skip / proj_out
        # with the raw fp32 operand split on the fly (same 22-bit split), attention (head dims <= 256) on the split flash kernel in the same format
        assert model.precision in ("fp32", "bf16x3", "f16x3"), model.precision
        self.split_mode = model.precision not in ("fp32", "f16x3")          # on-the-fly bf16 split of raw fp32 operands
        self.presplit = model.precision != "fp32"
        self.split_acts = model.precision if model.precision == "f16x3" else model.precision == "bf16x3"
        self.gemm_split = self.split_acts                                                  # one-tap GEMMs (qkv, proj_out)
        # parity modes: whole ResBlocks / AttentionBlocks as single C-ABI calls (dua_adm_resblock_fwd / dua_adm_attnblock_fwd)
        self.block_abi = bool(getattr(model, "use_block_abi", True)) and model.precision in ("fp32", "f16x3") and not model.activation_noise

    def wino_form(self, N, H, W, cin, cout):
        """0 / 2 / 4: this 3x3 conv (input / output size H x W) runs in the direct form / the Winograd F(2,3) / F(4,3)-along-y form.
        model.winograd: True = the kernel library's choice per shape (f16x3: F(4,3) where whole rounds of tiles favour it), 2 / 4 = that
        form wherever its kernel takes the shape, False = direct everywhere.  fp32: F(4,3) wherever its kernel takes the shape (the fp32
        matrix pipe is the only limit there: 4.5 against 6 taps, whatever the number of tiles), else F(2,3); f16w has the F(2,3) form only."""
        w = getattr(self.model, "winograd", True)
        if self.model.precision not in ("f16x3", "fp32") or not w:
            return 0
        if self.model.precision == "fp32" and w != 2 and ops.conv3x3_wino4_supported(N, H, W, cin, cout):
            return 4
        if self.model.precision == "f16x3" and not self.model.wino_single:
            s4 = ops.conv3x3_wino4_supported(N, H, W, cin, cout)
            if w is True:
                # The library's rule counts whole rounds of tiles, i.e. it depends on the batch.  It is evaluated at the model's PLANNING
                # batch (wino_plan_batch, default 128 = BASELINE configs[1] per GPU), never at the batch of the call: the form of a layer
                # - and with it every image's arithmetic - is then the same whatever batch the image runs in.
                s2 = ops.conv3x3_wino_supported(N, H, W, cin, cout)
                form = ops.conv3x3_wino_pick(int(getattr(self.model, "wino_plan_batch", 128)), H, W, cin, cout)
                if form == 4 and s4:
                    return 4
                return 2 if s2 else (4 if s4 else 0)
            if w == 4 and s4:
                return 4
        return 2 if ops.conv3x3_wino_supported(N, H, W, cin, cout) else 0

    def wino_ok(self, N, H, W, cin, cout):
        return self.wino_form(N, H, W, cin, cout) != 0

    def conv_split(self, width, conv_shape=None):
        """Operand format the GroupNorm in front of a 3x3 conv of this output width writes (conv_shape = (N, H, W, Cin, Cout) lets
        the f16x3 mode pick the Winograd operand where the kernel takes the shape)."""
        m = self.wino_form(*conv_shape) if conv_shape is not None else 0
        if m:
            if self.model.precision == "f16x3":
                # mode "f16w" (one fp16 product per tap, F(2,3) only): the hi-images-only operand on EVERY call path - the block-level
                # C ABI (DUA_PREC_F16W) and this per-op composition run the same kernel on the same operand, bit for bit
                return "f16x3w4" if m == 4 else ("f16w" if self.model.wino_single else "f16x3w")
            return "f32w4" if m == 4 else "f32w"
        return self.split_acts

    def packed(self, mod, kind):
        cache = self.model._packed
        key = (id(mod), kind)
        if key not in cache:
            if kind == "w3":
                cache[key] = ops.pack_conv3x3_weight(mod.weight.detach().contiguous())
            elif kind == "w1":
                w = mod.weight.detach()
                cache[key] = w.reshape(w.shape[0], w.shape[1]).contiguous()
            elif kind == "w3x":
                cache[key] = ops.pack_weight_bf16x3(mod.weight.detach().contiguous())
            elif kind == "w3h":
                cache[key] = ops.pack_weight_f16x3(mod.weight.detach().contiguous())
            elif kind == "w3w":
                cache[key] = ops.pack_weight_f16x3_wino(mod.weight.detach().contiguous())
            elif kind == "w3w4":
                cache[key] = ops.pack_weight_f16x3_wino4(mod.weight.detach().contiguous())
            elif kind == "w3wf":
                cache[key] = ops.pack_weight_f32_wino(mod.weight.detach().contiguous())
            elif kind == "w3w4f":
                cache[key] = ops.pack_weight_f32_wino4(mod.weight.detach().contiguous())
            elif kind == "qkvh":
                wq, bq = self.packed(mod, "qkv")
                cache[key] = (ops.pack_weight_f16x3(wq), bq)
            elif kind == "w1h":
                w = mod.weight.detach()
                cache[key] = ops.pack_weight_f16x3(w.reshape(w.shape[0], w.shape[1]).contiguous())
            elif kind == "w1x":
                w = mod.weight.detach()
                cache[key] = ops.pack_weight_bf16x3(w.reshape(w.shape[0], w.shape[1]).contiguous())
            elif kind == "qkvx":
                wq, bq = self.packed(mod, "qkv")
                cache[key] = (ops.pack_weight_bf16x3(wq), bq)
            elif kind == "qkv":
                w = mod.qkv.weight.detach().reshape(3 * mod.channels, mod.channels)
                b = mod.qkv.bias.detach()
                if isinstance(mod.attention, QKVAttentionLegacy):
                    # rows (head, part, c) -> (part, head, c): one attention layout for both split orders
                    Hh, ch = mod.num_heads, mod.channels // mod.num_heads
                    w = w.reshape(Hh, 3, ch, mod.channels).permute(1, 0, 2, 3).reshape(3 * mod.channels, mod.channels)
                    b = b.reshape(Hh, 3, ch).permute(1, 0, 2).reshape(-1)
                cache[key] = (w.contiguous(), b.contiguous())
        return cache[key]

    def film(self, blk):
        """(scale | shift) rows of one ResBlock: a column block of the FiLM table.  `emb_layers` of EVERY ResBlock is the same
        Linear(SiLU(emb)) on the same [B, 4*model_channels] embedding (unet_openai.py:200-206,245), so the forward evaluates
        all of them with ONE GEMM against the row-concatenated weights (36 launches of ~0.1 ms -> 1 for ADM-128)."""
        tab = self.model._film_table()
        if self.film_all is None:
            self.film_all = ops.gemm(self.emb, tab["w"], bias=tab["b"], act_in=1)
        o, n = tab["slices"][id(blk)]
        return self.film_all[:, o:o + n]

    def dropout_mask(self, shape_nhwc, p):
        return self.model.dropout_mask_source(shape_nhwc, p)

    def add_activation_noise(self, h, std):
        """h + std * randn_like(h): the forward hooks of schedulers_uncertainty/scheduling_ddim_uncertainty.py:36-40,
        520-529.  The draw has the reference's NCHW shape (same tape order as torch.randn_like on the module output)."""
        N, H, W, C = h.shape
        src = self.model.activation_noise_source
        if src is None:
            raise RuntimeError("UNetModel.activation_noise is set but activation_noise_source is None")
        noise = ops.nchw_to_nhwc(src.draw((N, C, H, W), h.device))
        return ops.axpby(h, noise, 1.0, std)

    # ---- the FLOP carriers, in the model's precision mode
    def conv3x3(self, h, conv, cout, residual=None, nchw_out=False, colsum=False):
        colsum = colsum and self.model.fuse_gn_stats
        if isinstance(h, ops.SplitPlanes) or (self.split_mode and h.shape[3] % 32 == 0):
            kind = {"f16x3": "w3h", "f16x3w": "w3w", "f16w": "w3w", "f16x3w4": "w3w4", "f32w": "w3wf", "f32w4": "w3w4f"}.get(h.fmt, "w3x") if isinstance(h, ops.SplitPlanes) else "w3x"
            return ops.conv3x3_bf16x3(h, self.packed(conv, kind), conv.bias, cout, residual=residual, nchw_out=nchw_out, colsum=colsum)
        return ops.conv3x3(h, self.packed(conv, "w3"), conv.bias, cout, residual=residual, nchw_out=nchw_out, colsum=colsum)

    def linear(self, a, mod, a2=None, residual=None, colsum=False):
        """1x1 conv / conv1d(k=1) as a GEMM over rows."""
        n_out = mod.weight.shape[0]
        K = a.shape[1] + (0 if a2 is None else a2.shape[1])
        colsum = colsum and self.model.fuse_gn_stats
        if self.model.precision == "f16x3" and K % 32 == 0 and a.shape[1] % 32 == 0:     # fp32 A split on the fly into fp16 hi + lo
            return ops.gemm_f16x3(a, self.packed(mod, "w1h"), n_out, bias=mod.bias, a2=a2, residual=residual, colsum=colsum)
        if self.split_mode and K % 32 == 0 and a.shape[1] % 32 == 0:      # fp32 A split on the fly: the bf16x3 tile engine in both modes
            return ops.gemm_bf16x3(a, self.packed(mod, "w1x"), n_out, bias=mod.bias, a2=a2, residual=residual, colsum=colsum)
        return ops.gemm(a, self.packed(mod, "w1"), bias=mod.bias, a2=a2, residual=residual, colsum=colsum)

    def qkv(self, xn, blk):
        if isinstance(xn, ops.SplitPlanes):
            planes, bq = self.packed(blk, "qkvh" if xn.fmt == "f16x3" else "qkvx")
            return ops.gemm_bf16x3_presplit(xn, planes, 3 * blk.channels, bias=bq)
        if self.split_mode and xn.shape[1] % 32 == 0:
            planes, bq = self.packed(blk, "qkvx")
            return ops.gemm_bf16x3(xn, planes, 3 * blk.channels, bias=bq)
        wq, bq = self.packed(blk, "qkv")
        return ops.gemm(xn, wq, bias=bq)


class UNetModel(nn.Module):
    """unet_openai.py:397-665."""

    def __init__(self, image_size, in_channels, model_channels, out_channels, num_res_blocks, attention_resolutions,
                 dropout=0, channel_mult=(1, 2, 4, 8), conv_resample=True, dims=2, num_classes=None,
                 use_checkpoint=False, use_fp16=False, num_heads=1, num_head_channels=-1, num_heads_upsample=-1,
                 use_scale_shift_norm=False, resblock_updown=False, use_new_attention_order=False):
        super().__init__()
        assert dims == 2 and resblock_updown and use_scale_shift_norm and not use_fp16, \
            "the HIP path covers the configuration family of init_model.py (resblock_updown + scale-shift norm, fp32)"
        assert in_channels <= 4
        if num_heads_upsample == -1:
            num_heads_upsample = num_heads
        self.image_size, self.in_channels, self.model_channels = image_size, in_channels, model_channels
        self.out_channels, self.num_res_blocks = out_channels, num_res_blocks
        self.attention_resolutions, self.dropout, self.channel_mult = attention_resolutions, dropout, channel_mult
        self.conv_resample, self.num_classes, self.use_checkpoint = conv_resample, num_classes, use_checkpoint
        self.dtype = torch.float32
        self.num_heads, self.num_head_channels, self.num_heads_upsample = num_heads, num_head_channels, num_heads_upsample

        ted = model_channels * 4
        self.time_embed = nn.Sequential(nn.Linear(model_channels, ted), nn.SiLU(), nn.Linear(ted, ted))
        if num_classes is not None:
            self.label_emb = nn.Embedding(num_classes, ted)
        ch = input_ch = int(channel_mult[0] * model_channels)
        self.input_blocks = nn.ModuleList([TimestepEmbedSequential(nn.Conv2d(in_channels, ch, 3, padding=1))])
        chans, ds = [ch], 1
        rb = dict(use_scale_shift_norm=use_scale_shift_norm)
        ab = dict(num_head_channels=num_head_channels, use_new_attention_order=use_new_attention_order)
        for level, mult in enumerate(channel_mult):
            for _ in range(num_res_blocks):
                layers = [ResBlock(ch, ted, dropout, out_channels=int(mult * model_channels), **rb)]
                ch = int(mult * model_channels)
                if ds in attention_resolutions:
                    layers.append(AttentionBlock(ch, num_heads=num_heads, **ab))
                self.input_blocks.append(TimestepEmbedSequential(*layers))
                chans.append(ch)
            if level != len(channel_mult) - 1:
                self.input_blocks.append(TimestepEmbedSequential(ResBlock(ch, ted, dropout, out_channels=ch, down=True, **rb)))
                chans.append(ch)
                ds *= 2
        self.middle_block = TimestepEmbedSequential(ResBlock(ch, ted, dropout, **rb), AttentionBlock(ch, num_heads=num_heads, **ab),
                                                    ResBlock(ch, ted, dropout, **rb))
        self.output_blocks = nn.ModuleList([])
        for level, mult in list(enumerate(channel_mult))[::-1]:
            for i in range(num_res_blocks + 1):
                ich = chans.pop()
                layers = [ResBlock(ch + ich, ted, dropout, out_channels=int(model_channels * mult), **rb)]
                ch = int(model_channels * mult)
                if ds in attention_resolutions:
                    layers.append(AttentionBlock(ch, num_heads=num_heads_upsample, **ab))
                if level and i == num_res_blocks:
                    layers.append(ResBlock(ch, ted, dropout, out_channels=ch, up=True, **rb))
                    ds //= 2
                self.output_blocks.append(TimestepEmbedSequential(*layers))
        self.out = nn.Sequential(normalization(ch), nn.SiLU(), zero_module(nn.Conv2d(input_ch, out_channels, 3, padding=1)))
        self._packed = {}
        # "fp32": exact fp32 MFMA (v_mfma_f32_32x32x2_f32).  "bf16x3": convs and weight GEMMs on the bf16 matrix cores
        # with split fp32 operands (csrc/mfma_bf16x3.hip); attention products, embeddings, norms stay fp32.
        # "f16x3": the fp32-equivalent split (fp16 hi + lo of the power-of-two-scaled operand, three fp16 MFMAs per product): the headline.
        # "f16w" (side mode, NOT a parity mode): the f16x3 engine with ONE fp16 product per tap in the Winograd-form 3x3 convs (fp16
        # operands, fp32 accumulation - the arithmetic of the reference's own GPU path under torch.autocast, script :83; GroupNorm,
        # softmax, the sampler and every GEMM outside those convs as in f16x3).  `precision` then reads "f16x3", `wino_single` True.
        self.precision = "fp32"
        # GroupNorm statistics produced in the epilogue of the conv / GEMM that writes the tensor (no re-read); set False to
        # use the stand-alone statistics kernel everywhere.
        self.fuse_gn_stats = True
        self.presplit_qkv = True         # bf16x3: attention GroupNorm writes split planes, qkv runs on the LDS-DMA kernel
        self.use_block_abi = True        # fp32 / f16x3: ResBlocks and AttentionBlocks run as ONE C-ABI call each (bit-identical to the composition)
        self._graphs = None              # enable_hip_graph(): {key: captured forward}
        # f16x3, small batches: conv / presplit-GEMM launches with fewer than 128 tiles run split along K (ops.set_split_k; deterministic,
        # but the result then depends on the batch size at fp32-rounding level, so it is opt-in here: the Monte-Carlo chains of a
        # UC step stay bit-identical whether they run one by one or stacked)
        self.split_k = False
        # f16x3 / fp32: 3x3 convs in a Winograd-along-y form where a kernel takes the shape (csrc/conv_wino_f16x3.hip, conv_wino_f32.hip:
        # F(2,3), 6 instead of 9 matrix-core taps per pixel; csrc/conv_wino4_f16x3.hip: F(4,3), 4.5 taps, f16x3 only) at the same operand
        # precision; results differ from the direct form at fp32-rounding level.  True = the library picks the form per shape
        # (dua_conv3x3_f16x3_wino_pick, a rule over whole rounds of tiles) evaluated at `wino_plan_batch`, NOT at the batch of the call:
        # a layer's form - and every image's arithmetic - does not depend on the batch the image runs in (bit-identical per image across
        # batch sizes, stacked Monte-Carlo chains == one by one); 2 / 4 = that form wherever it applies; False = the direct kernels.
        self.wino_plan_batch = 128
        self.winograd = True
        # fp32 / f16x3: the few-channel `out` conv as one-tap GEMM + tap gather (see _forward_body)
        self.out_conv_taps = True
        # mc_dropout: masks come from the CPU generator in ResBlock execution order by default (parity with the
        # reference CPU path); set to a device generator for speed (see schedulers_uncertainty/noise.py).
        self.dropout_mask_source = host_dropout_mask
        # {module name: std}: Gaussian noise added to that module's output (scheduler_type 'uncertainty'); drawn from
        # activation_noise_source.draw(shape, device)
        self.activation_noise = None
        self.activation_noise_source = None
        # f16x3: the first forward after the weights were loaded / edited runs with the library's range counters on and warns when
        # an operand left fp16's range after its static plane scale (range_guard.py); False opts out
        self.auto_range_check = True
        self._range_checked = False
        self._range_probe = None

    @property
    def precision(self):
        return self._precision

    @precision.setter
    def precision(self, v):
        self.wino_single = v == "f16w"
        self._precision = "f16x3" if v == "f16w" else v

    @property
    def precision_mode(self):
        """The user-facing mode name ("f16w" when the single-product Winograd convs are on)."""
        return "f16w" if self.wino_single else self._precision

    # ---- weight cache hygiene
    def invalidate_packed_weights(self):
        self._packed = {}
        self._range_checked = False
        if getattr(self, "_graphs", None):
            self._graphs = {}

    def _film_table(self):
        """Row-concatenated emb_layers weights / biases of all ResBlocks (cached with the packed weights)."""
        tab = self._packed.get("film_table")
        if tab is None:
            blocks = [m for m in self.modules() if isinstance(m, ResBlock)]
            slices, o = {}, 0
            for b in blocks:
                n = b.emb_layers[1].weight.shape[0]
                slices[id(b)] = (o, n)
                o += n
            tab = dict(w=torch.cat([b.emb_layers[1].weight.detach() for b in blocks]).contiguous(),
                       b=torch.cat([b.emb_layers[1].bias.detach() for b in blocks]).contiguous(), slices=slices)
            self._packed["film_table"] = tab
        return tab

    def _apply(self, fn, *a, **k):
        self.invalidate_packed_weights()
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self.invalidate_packed_weights()
        return super().load_state_dict(*a, **k)

    def enable_hip_graph(self, on=True):
        """Capture the eval-mode forward once per (input shape, precision, kept channels) into a HIP graph and replay it afterwards:
        ~330 launches of a forward become one graph launch - what the small-batch cases (BASELINE configs[0]: ADM-64, batch 4) are
        bound by.  Bit-identical to the eager forward.  Train mode (host dropout masks) and activation-noise hooks stay eager."""
        self._graphs = {} if on else None
        return self

    @torch.no_grad()
    def forward(self, x, timesteps, y=None, out_channels_keep=None):
        """out_channels_keep: write only the first k output channels (the `[:, :3]` of traits.py:18)."""
        assert (y is not None) == (self.num_classes is not None), \
            "must specify y if and only if the model is class-conditional"
        if not x.is_cuda:
            raise RuntimeError("UNetModel.forward: the HIP path needs GPU tensors; this package has no CPU fallback")
        gen = params_generation(self)                # in-place weight edits (param.data.copy_, LoRA merges) repack
        if gen != getattr(self, "_packed_gen", None):
            if self._packed:
                self._packed = {}
            if getattr(self, "_graphs", None):
                self._graphs = {}
            self._packed_gen = gen
            self._range_checked = False
        if range_guard.due(self):
            return range_guard.checked_forward(self, lambda: self._forward_impl(x, timesteps, y, out_channels_keep))
        graphs = getattr(self, "_graphs", None)
        n = int(getattr(self, "stream_split", 1))
        if n > 1 and graphs is None and not self.training and not self.activation_noise and x.shape[0] >= 2 * n:
            return self._forward_streams(x, timesteps, y, out_channels_keep, n)
        if graphs is None or self.training or self.activation_noise:
            return self._forward_impl(x, timesteps, y, out_channels_keep)
        x = x.float().contiguous()
        t = timesteps.to(torch.int64).contiguous()
        yy = y.to(torch.int64).contiguous() if y is not None else None
        key = (tuple(x.shape), self.precision_mode, out_channels_keep, self.use_block_abi, self.fuse_gn_stats, ops.ATTENTION_MODE, y is not None,
               bool(self.split_k), getattr(self, "winograd", True), bool(getattr(self, "out_conv_taps", True)))
        entry = graphs.get(key)
        if entry is None:
            self._forward_impl(x, t, yy, out_channels_keep)            # eager once: packs the weights, warms the allocator
            sx, st, sy = x.clone(), t.clone(), (yy.clone() if yy is not None else None)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                so = self._forward_impl(sx, st, sy, out_channels_keep)
            entry = graphs[key] = (graph, sx, st, sy, so)
        graph, sx, st, sy, so = entry
        sx.copy_(x)
        st.copy_(t)
        if sy is not None:
            sy.copy_(yy)
        graph.replay()
        return so.clone()

    def _forward_streams(self, x, timesteps, y, out_channels_keep, n):
        """The batch in n chunks, each chunk's whole forward on a HIP stream of its own (results identical per image: every kernel is
        batch-invariant; split-K stays off: its workspace is process-global).  See uvit/uvit.py::_forward_streams."""
        cur = torch.cuda.current_stream(x.device)
        if len(getattr(self, "_streams", [])) < n:
            self._streams = [torch.cuda.Stream(device=x.device) for _ in range(n)]
        xs, ts = torch.chunk(x, n), torch.chunk(timesteps, n)
        ys = torch.chunk(y, n) if y is not None else [None] * len(xs)
        outs = []
        # the split-K / stream-K workspace is ONE process-global buffer: concurrent chunks would fold each other's parts.  Force it off
        # here - also when the CALLER had registered it (ops.set_split_k(True)) - and restore the caller's setting afterwards.
        prev_splitk = ops.set_split_k(False, x.device)
        prev_conc = _lib.lib().dua_set_hgemm_concurrent(1)       # tile widths by work: the other chunk's kernels fill partly empty rounds
        try:
            return self._forward_chunks(xs, ts, ys, cur, outs, out_channels_keep)
        finally:
            _lib.lib().dua_set_hgemm_concurrent(prev_conc)
            ops.set_split_k(prev_splitk, x.device)

    def _forward_chunks(self, xs, ts, ys, cur, outs, out_channels_keep=None):
        for i, (xc, tc, yc) in enumerate(zip(xs, ts, ys)):
            st = self._streams[i]
            st.wait_stream(cur)
            n_packed = len(self._packed)
            with torch.cuda.stream(st):
                outs.append(self._forward_body(xc.contiguous(), tc.contiguous(), yc.contiguous() if yc is not None else None, out_channels_keep))
            if len(self._packed) != n_packed:
                st.synchronize()         # this chunk packed weights on ITS stream: the other chunks read them from the shared cache
        for o, st in zip(outs, self._streams):
            cur.wait_stream(st)
            o.record_stream(cur)
        return torch.cat(outs)

    def _forward_impl(self, x, timesteps, y=None, out_channels_keep=None):
        if not (self.split_k and self.precision == "f16x3"):
            return self._forward_body(x, timesteps, y, out_channels_keep)
        prev = ops.set_split_k(True, x.device)
        try:
            return self._forward_body(x, timesteps, y, out_channels_keep)
        finally:
            ops.set_split_k(prev, x.device)

    def _forward_body(self, x, timesteps, y=None, out_channels_keep=None):
        ctx = _Ctx(self)
        B = x.shape[0]
        te, te2 = self.time_embed[0], self.time_embed[2]
        emb = ops.timestep_embedding(timesteps.to(torch.int64).contiguous(), self.model_channels)
        emb = ops.gemm(emb, te.weight, bias=te.bias)
        emb = ops.gemm(emb, te2.weight, bias=te2.bias, act_in=1)
        if self.num_classes is not None:
            assert y.shape == (B,), f"{y.shape=} != {x.shape[0]=}"
            emb = ops.embedding_add(emb, self.label_emb.weight, y.to(torch.int64).contiguous())
        ctx.emb = emb
        h = ops.nchw_to_nhwc(x.float().contiguous(), 4)
        hs = []
        probe = self._range_probe
        for i, module in enumerate(self.input_blocks):
            h = module.run(ctx, h, name=f"input_blocks.{i}")
            hs.append(h)
            if probe is not None:
                probe.note(f"input_blocks.{i}", module)
        h = self.middle_block.run(ctx, h, name="middle_block")
        if probe is not None:
            probe.note("middle_block", self.middle_block)
        for i, module in enumerate(self.output_blocks):
            h = module.run(ctx, h, hs.pop(), name=f"output_blocks.{i}")
            if probe is not None:
                probe.note(f"output_blocks.{i}", module)
        gn, conv = self.out[0], self.out[2]
        st = ops.groupnorm_stats(h)
        keep = self.out_channels if out_channels_keep is None else out_channels_keep
        if self.precision in ("fp32", "f16x3") and keep <= 8 and h.shape[3] % 32 == 0 and bool(getattr(self, "out_conv_taps", True)):
            # the out conv (:616) keeps 3 - 6 channels: nine tap columns per channel from ONE one-tap GEMM over the pixels, then a
            # gather of the neighbours' columns (ops.conv3x3_tap_gather) - the implicit-GEMM conv kernels would compute a 128-column
            # tile for it (2.8 ms of a 170 ms forward at batch 128)
            N_, H_, W_, C_ = h.shape
            key = (id(conv), "taps", keep, self.precision)
            if key not in self._packed:
                wt = ops.tap_column_weight(conv.weight.detach(), keep)
                self._packed[key] = (ops.pack_weight_f16x3(wt) if self.precision == "f16x3" else wt, wt.shape[0])
            wt, ncol = self._packed[key]
            if self.precision == "f16x3":
                hp = ops.groupnorm_apply(h, st, gn.weight, gn.bias, act=1, split="f16x3")
                T = ops.gemm_bf16x3_presplit(hp, wt, ncol)
            else:
                hp = ops.groupnorm_apply(h, st, gn.weight, gn.bias, act=1)
                T = ops.gemm(hp.view(N_ * H_ * W_, C_), wt)
            return ops.conv3x3_tap_gather(T, conv.bias.detach()[:keep].contiguous(), N_, H_, W_, keep)
        h = ops.groupnorm_apply(h, st, gn.weight, gn.bias, act=1, split=ctx.conv_split(h.shape[2]))
        return ctx.conv3x3(h, conv, keep, nchw_out=True)


def host_dropout_mask(shape_nhwc, p):
    """Bernoulli keep-mask / (1-p) drawn from the global CPU generator in NCHW order, exactly the draw
    nn.Dropout makes on the reference CPU path, uploaded as NHWC."""
    N, H, W, C = shape_nhwc
    m = torch.empty((N, C, H, W), dtype=torch.float32).bernoulli_(1 - p).div_(1 - p)
    return m.permute(0, 2, 3, 1).contiguous().to("cuda", non_blocking=True)
