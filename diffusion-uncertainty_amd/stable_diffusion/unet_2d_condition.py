"""Latent UNet of Stable Diffusion 1.5 on the gfx950 HIP library (BASELINE configs[4]).

The reference takes this network from the un-vendored ``diffusers==0.31.0`` (``UNet2DConditionModel``,
``pipeline_uncertainty/pipeline_stable_diffusion_uncertainty_guided.py:747-753``); this module holds parameters under
diffusers' state-dict keys (``down_blocks.0.resnets.0.norm1.weight``, ``...attentions.0.transformer_blocks.0.attn2.to_k.weight``
...) so the ``runwayml/stable-diffusion-v1-5`` UNet checkpoint loads strictly, keeps its call convention
``unet(sample[B,4,h,w], timestep, encoder_hidden_states=[B,77,768], return_dict=False)[0]`` and drives the C ABI
(include/dua_hip.h): NHWC activations, 3x3 convs / 1x1 convs / linears on the MFMA tile engines in the model's precision
mode, GroupNorm+SiLU fused with the operand split, the skip concat of the up blocks never materialised, self-attention in the
fused flash kernel, cross-attention (77 text tokens) through the batched GEMM + softmax path, GEGLU / time-embedding add as
small HBM-bound kernels.  PARITY UNPINNED: no diffusers here and no vector in /root/reference; checked against
oracle/sd_unet.py (a restatement of the published architecture) only.  There is no CPU path.
"""
import torch
import torch.nn as nn

from .. import ops, range_guard
from ..utils import params_generation


class ResnetBlock2D(nn.Module):
    def __init__(self, cin, cout, temb, groups):
        super().__init__()
        self.cin, self.cout = cin, cout
        self.norm1, self.conv1 = nn.GroupNorm(groups, cin), nn.Conv2d(cin, cout, 3, padding=1)
        self.time_emb_proj = nn.Linear(temb, cout)
        self.norm2, self.conv2 = nn.GroupNorm(groups, cout), nn.Conv2d(cout, cout, 3, padding=1)
        if cin != cout:
            self.conv_shortcut = nn.Conv2d(cin, cout, 1)

    def run(self, ctx, x, skip=None, tape=None):
        """x (+ skip = second half of the channel concat of the up blocks): NHWC.  tape: list that receives what run_bwd needs."""
        N, H, W, _ = x.shape
        G = self.norm1.num_groups
        st = ops.groupnorm_stats(x, skip, groups=G, eps=self.norm1.eps)
        h = ops.groupnorm_apply(x, st, self.norm1.weight, self.norm1.bias, x2=skip, act=1, groups=G, split=ctx.split_acts)
        fold = ctx.f16x3 and ctx.model.fold_temb
        h = ctx.conv3x3(h, self.conv1, self.cout, colsum=fold)
        hp = ops.groupnorm_prebias_planes(h, ctx.temb(self), self.norm2.weight, self.norm2.bias, act=1, groups=G, eps=self.norm2.eps) if fold else None
        if hp is not None and tape is not None:                         # a taped forward keeps hidden + temb for the backward pass
            ops.add_channel_bias_(h, ctx.temb(self))                   # (after the apply above has read h, in stream order)
            tape.append((x, skip, st, h, hp.stats))
        if hp is None:
            ops.add_channel_bias_(h, ctx.temb(self))                   # + time_emb_proj(silu(emb))[:, :, None, None]
            st2 = ops.groupnorm_stats(h, groups=G, eps=self.norm2.eps)
            if tape is not None:
                tape.append((x, skip, st, h, st2))
            h = ops.groupnorm_apply(h, st2, self.norm2.weight, self.norm2.bias, act=1, groups=G, split=ctx.split_acts)
        else:
            h = hp                                                     # norm2(hidden + temb) from the conv's column sums: no add / statistics passes
        if hasattr(self, "conv_shortcut"):
            res = ctx.linear(x.view(N * H * W, -1), self.conv_shortcut, a2=(skip.view(N * H * W, -1) if skip is not None else None))
            res = res.view(N, H, W, self.cout)
        else:
            assert skip is None
            res = x
        return ctx.conv3x3(h, self.conv2, self.cout, residual=res, colsum=True)     # epilogue leaves the next GroupNorm's statistics

    def run_bwd(self, ctx, saved, g):
        """Gradient of run() w.r.t. (x, skip) given g = d/d output (NHWC)."""
        x, skip, st, h1, st2 = saved
        N, H, W, C1 = x.shape
        G = self.norm1.num_groups
        g_h2 = ctx.conv3x3_bwd(g, self.conv2)
        g_h1, _ = ops.groupnorm_bwd(h1, st2, self.norm2.weight, self.norm2.bias, g_h2, act=1, groups=G)
        g_n1 = ctx.conv3x3_bwd(g_h1, self.conv1)                      # the time-embedding add passes the gradient through
        if hasattr(self, "conv_shortcut"):
            parts = ctx.linear_bwd(g.view(N * H * W, self.cout), self.conv_shortcut.weight,
                                   split=None if skip is None else (C1, skip.shape[3]))
            add1 = parts[0].view(N, H, W, C1)
            add2 = parts[1].view(N, H, W, -1) if skip is not None else None
        else:
            add1, add2 = g, None
        return ops.groupnorm_bwd(x, st, self.norm1.weight, self.norm1.bias, g_n1, x2=skip, act=1, groups=G, add1=add1, add2=add2)


class Attention(nn.Module):
    def __init__(self, dim, ctx_dim, heads):
        super().__init__()
        self.heads = heads
        self.to_q, self.to_k, self.to_v = nn.Linear(dim, dim, bias=False), nn.Linear(ctx_dim, dim, bias=False), nn.Linear(ctx_dim, dim, bias=False)
        self.to_out = nn.ModuleList([nn.Linear(dim, dim), nn.Dropout(0.0)])


class GEGLU(nn.Module):
    def __init__(self, dim, inner):
        super().__init__()
        self.proj = nn.Linear(dim, 2 * inner)


class FeedForward(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.net = nn.ModuleList([GEGLU(dim, 4 * dim), nn.Dropout(0.0), nn.Linear(4 * dim, dim)])


class BasicTransformerBlock(nn.Module):
    def __init__(self, dim, ctx_dim, heads):
        super().__init__()
        self.norm1, self.attn1 = nn.LayerNorm(dim), Attention(dim, dim, heads)
        self.norm2, self.attn2 = nn.LayerNorm(dim), Attention(dim, ctx_dim, heads)
        self.norm3, self.ff = nn.LayerNorm(dim), FeedForward(dim)

    def run(self, ctx, h, B, T, enc, tape=None):
        """h: [B*T, C] token rows; enc: [B, Tk, ctx_dim]."""
        C = h.shape[1]
        h0 = h
        heads = self.attn1.heads
        n1 = ctx.norm_rows(h, self.norm1)
        qkv = ctx.gemm(n1, ctx.cat_weight(self.attn1, ("to_q", "to_k", "to_v")), None)
        a = ops.qkv_attention(qkv, B, T, heads, C // heads, precision=ctx.attn_precision, planes_out=(ctx.split_acts == "f16x3"))
        h = h1 = ctx.gemm(a, self.attn1.to_out[0].weight, self.attn1.to_out[0].bias, residual=h)
        n2 = ctx.norm_rows(h, self.norm2)
        q = ctx.gemm(n2, self.attn2.to_q.weight, None)
        Tk = enc.shape[1]
        k, v = ctx.context_kv(self, enc)
        a = ops.attention(q.view(B, T, C), k.view(B, Tk, C), v.view(B, Tk, C), heads, precision=ctx.attn_precision,
                          planes_out=ctx.f16x3)
        if not isinstance(a, ops.SplitPlanes):
            a = a.view(B * T, C)
        h = ctx.gemm(a, self.attn2.to_out[0].weight, self.attn2.to_out[0].bias, residual=h)
        n3 = ctx.norm_rows(h, self.norm3)
        u = ctx.gemm(n3, self.ff.net[0].proj.weight, self.ff.net[0].proj.bias)
        if tape is not None:
            tape.append((B, T, h0, qkv, h1, q, k, v, h, u))
        g = ops.geglu(u, split="f16x3" if ctx.f16x3 and u.shape[1] % 64 == 0 else False)
        return ctx.gemm(g, self.ff.net[2].weight, self.ff.net[2].bias, residual=h)

    def run_bwd(self, ctx, saved, g):
        """Gradient of run() w.r.t. the token rows h given g = d/d output [B*T, C]; exact fp32."""
        B, T, h0, qkv, h1, q, k, v, h2, u = saved
        C = h0.shape[1]
        heads = self.attn1.heads
        g_u = ops.geglu_bwd(u, ctx.linear_bwd(g, self.ff.net[2].weight)[0])
        g_h2 = ops.layernorm_bwd(h2, self.norm3.weight, ctx.linear_bwd(g_u, self.ff.net[0].proj.weight)[0], eps=self.norm3.eps, add=g)
        g_a2 = ctx.linear_bwd(g_h2, self.attn2.to_out[0].weight)[0]
        Tk = k.shape[0] // B
        g_q = ops.attention_bwd_q(q.view(B, T, C), k.view(B, Tk, C), v.view(B, Tk, C), g_a2.view(B, T, C), heads).view(B * T, C)
        g_h1 = ops.layernorm_bwd(h1, self.norm2.weight, ctx.linear_bwd(g_q, self.attn2.to_q.weight)[0], eps=self.norm2.eps, add=g_h2)
        g_a = ctx.linear_bwd(g_h1, self.attn1.to_out[0].weight)[0]
        g_qkv = ops.qkv_attention_bwd(qkv, g_a, B, T, heads, C // heads)
        g_n1 = ctx.linear_bwd(g_qkv, ctx.cat_weight(self.attn1, ("to_q", "to_k", "to_v")), key=(id(self.attn1), "qkv_t"))[0]
        return ops.layernorm_bwd(h0, self.norm1.weight, g_n1, eps=self.norm1.eps, add=g_h1)


class Transformer2DModel(nn.Module):
    def __init__(self, dim, ctx_dim, heads, groups):
        super().__init__()
        self.norm = nn.GroupNorm(groups, dim, eps=1e-6)
        self.proj_in = nn.Conv2d(dim, dim, 1)
        self.transformer_blocks = nn.ModuleList([BasicTransformerBlock(dim, ctx_dim, heads)])
        self.proj_out = nn.Conv2d(dim, dim, 1)

    def run(self, ctx, x, enc, tape=None):
        N, H, W, C = x.shape
        st = ops.groupnorm_stats(x, groups=self.norm.num_groups, eps=self.norm.eps)
        if tape is not None:
            tape.append((x, st))
        xn = ops.groupnorm_apply(x, st, self.norm.weight, self.norm.bias, act=0, groups=self.norm.num_groups,
                                 split="f16x3" if ctx.f16x3 and C % 32 == 0 else False)
        if isinstance(xn, ops.SplitPlanes):
            xn = ops.SplitPlanes(xn.buf, (N * H * W, C), xn.fmt, xn.scale)
        else:
            xn = xn.view(N * H * W, C)
        h = ctx.linear(xn, self.proj_in)
        for blk in self.transformer_blocks:
            h = blk.run(ctx, h, N, H * W, enc, tape=tape)
        out2d = ctx.linear(h, self.proj_out, residual=x.view(N * H * W, C), colsum=True)
        out = out2d.view(N, H, W, C)
        ws = getattr(out2d, "_dua_colsums_rows", None)
        if ws is not None and (H * W) % 64 == 0:
            out._dua_colsums = ws                     # statistics for the next block's GroupNorm
        return out

    def run_bwd(self, ctx, tape, g):
        """Pops this block's records off the tape (reverse order of run) and returns d/d x."""
        N, H, W, C = g.shape
        g_h = ctx.linear_bwd(g.view(N * H * W, C), self.proj_out.weight)[0]
        for blk in reversed(self.transformer_blocks):
            g_h = blk.run_bwd(ctx, tape.pop(), g_h)
        x, st = tape.pop()
        g_xn = ctx.linear_bwd(g_h, self.proj_in.weight)[0].view(N, H, W, C)
        return ops.groupnorm_bwd(x, st, self.norm.weight, self.norm.bias, g_xn, act=0, groups=self.norm.num_groups, add1=g)[0]


class Downsample2D(nn.Module):
    def __init__(self, ch):
        super().__init__()
        self.conv = nn.Conv2d(ch, ch, 3, stride=2, padding=1)


class Upsample2D(nn.Module):
    def __init__(self, ch):
        super().__init__()
        self.conv = nn.Conv2d(ch, ch, 3, padding=1)


class _Block(nn.Module):
    """One down / mid / up block: ``resnets`` (+ ``attentions``) (+ ``downsamplers`` / ``upsamplers``)."""

    def __init__(self, resnets, attentions=None, down=None, up=None):
        super().__init__()
        self.resnets = nn.ModuleList(resnets)
        if attentions is not None:
            self.attentions = nn.ModuleList(attentions)
        if down is not None:
            self.downsamplers = nn.ModuleList([down])
        if up is not None:
            self.upsamplers = nn.ModuleList([up])


class TimestepEmbedding(nn.Module):
    def __init__(self, cin, ted):
        super().__init__()
        self.linear_1, self.linear_2 = nn.Linear(cin, ted), nn.Linear(ted, ted)


class _Ctx:
    def __init__(self, model):
        self.model = model
        assert model.precision in ("fp32", "bf16x3", "f16x3"), model.precision
        self.f16x3 = model.precision == "f16x3"                      # fp32-equivalent fp16 hi + lo split (22 operand bits)
        self.split_mode = model.precision == "bf16x3"                # on-the-fly bf16 split of raw fp32 operands
        self.split_acts = model.precision if model.precision == "f16x3" else model.precision == "bf16x3"
        self.attn_precision = {"fp32": "fp32", "f16x3": "f16x3"}.get(model.precision, "bf16x3")
        self.emb = None
        self.temb_all = None
        self.kv = None                   # {id(transformer block): (k, v)} of the current text context, or None

    def packed(self, key, make):
        cache = self.model._packed
        if key not in cache:
            cache[key] = make()
        return cache[key]

    def conv3x3(self, h, conv, cout, residual=None, nchw_out=False, colsum=False):
        if isinstance(h, ops.SplitPlanes):
            if h.fmt == "f16x3":
                w = self.packed((id(conv), "w3h"), lambda: ops.pack_weight_f16x3(conv.weight.detach().contiguous()))
            else:
                w = self.packed((id(conv), "w3x"), lambda: ops.pack_weight_bf16x3(conv.weight.detach().contiguous()))
            return ops.conv3x3_bf16x3(h, w, conv.bias, cout, residual=residual, nchw_out=nchw_out, colsum=colsum)
        if self.split_mode and h.shape[3] % 32 == 0:
            w = self.packed((id(conv), "w3x"), lambda: ops.pack_weight_bf16x3(conv.weight.detach().contiguous()))
            return ops.conv3x3_bf16x3(h, w, conv.bias, cout, residual=residual, nchw_out=nchw_out, colsum=colsum)
        w = self.packed((id(conv), "w3"), lambda: ops.pack_conv3x3_weight(conv.weight.detach().contiguous()))
        return ops.conv3x3(h, w, conv.bias, cout, residual=residual, nchw_out=nchw_out, colsum=colsum)

    def gemm(self, a, weight, bias, a2=None, residual=None, colsum=False):
        """rows x weight^T in the model's precision (fp32 MFMA, or the split-bf16 tile engine with the operand split on the fly)."""
        if isinstance(a, ops.SplitPlanes):           # operand already split (norm / attention output): LDS-DMA GEMM
            n_out = weight.shape[0]
            kind, pack = ("w1hp", ops.pack_weight_f16x3) if a.fmt == "f16x3" else ("w1xp", ops.pack_weight_bf16x3)
            packed = self.packed((id(weight), kind), lambda: pack(weight.detach().reshape(n_out, -1).contiguous()))
            return ops.gemm_bf16x3_presplit(a, packed, n_out, bias=bias, residual=residual, colsum=colsum)
        n_out, K = weight.shape[0], a.shape[1] + (0 if a2 is None else a2.shape[1])
        if self.f16x3 and self.model.split_k and K % 32 == 0 and a.shape[1] % 4 == 0 and -(-a.shape[0] // 256) * -(-n_out // 128) <= 128:
            # few rows (the CFG batch): one small pass writes the rows as f16x3 planes, then the LDS-DMA GEMM, which splits along K when
            # its tiles cannot fill the chip - the on-the-fly engine below would run these on 6 - 80 of the 256 CUs
            kind = "w1hp"
            packed = self.packed((id(weight), kind), lambda: ops.pack_weight_f16x3(weight.detach().reshape(n_out, K).contiguous()))
            return ops.gemm_bf16x3_presplit(self.row_planes(a, a2), packed, n_out, bias=bias, residual=residual, colsum=colsum)
        if self.f16x3 and K % 32 == 0 and a.shape[1] % 32 == 0:       # raw fp32 rows split on the fly into fp16 hi + lo
            packed = self.packed((id(weight), "w1h"), lambda: ops.pack_weight_f16x3(weight.detach().reshape(n_out, K).contiguous()))
            return ops.gemm_f16x3(a, packed, n_out, bias=bias, a2=a2, residual=residual, colsum=colsum)
        if self.split_mode and K % 32 == 0 and a.shape[1] % 32 == 0:
            planes = self.packed((id(weight), "w1x"), lambda: ops.pack_weight_bf16x3(weight.detach().reshape(n_out, K).contiguous()))
            return ops.gemm_bf16x3(a, planes, n_out, bias=bias, a2=a2, residual=residual, colsum=colsum)
        w = self.packed((id(weight), "w1"), lambda: weight.detach().reshape(n_out, K).contiguous())
        return ops.gemm(a, w, bias=bias, a2=a2, residual=residual, colsum=colsum)

    def norm_rows(self, h, norm):
        """LayerNorm of token rows; in f16x3 mode straight into the operand planes of the next projection."""
        split = "f16x3" if self.f16x3 and h.shape[1] % 32 == 0 else False
        return ops.layernorm(h, norm.weight, norm.bias, eps=norm.eps, split=split)

    def raw_planes(self, h, resample=0):
        """Raw fp32 NHWC map (no norm in front of a down / up-sampler conv) -> f16x3 operand planes, optionally nearest x2:
        the GroupNorm-apply kernel with identity statistics (mean 0, rstd 1, gamma 1, beta 0: exact) at the raw-stream scale."""
        N, H, W, C = h.shape
        if not (self.f16x3 and C % 32 == 0):
            return ops.resample2x(h, 1) if resample > 0 else h
        key = ("identity_gn", N, C)
        if key not in self.model._packed:
            st = torch.zeros((N, 32, 2), device=h.device, dtype=torch.float32)
            st[:, :, 1] = 1.0
            self.model._packed[key] = (st, torch.ones(C, device=h.device), torch.zeros(C, device=h.device))
        st, one, zero = self.model._packed[key]
        return ops.groupnorm_apply(h, st, one, zero, act=0, resample=resample, groups=32, split="f16x3", split_scale=ops.F16X3_RAW_SCALE)

    def conv3x3_bwd(self, g, conv):
        """d/d input of a stride-1 / pad-1 3x3 convolution: the same convolution with the kernels rotated by 180 degrees and the
        channel roles swapped, on the exact fp32 MFMA engine (gradients have no fixed scale to split at)."""
        def flipped():
            return conv.weight.detach().permute(1, 0, 2, 3).flip(2, 3).contiguous()
        if self.model.backward_precision == "bf16x3" and g.shape[3] % 32 == 0:
            return ops.conv3x3_bf16x3(g, self.packed((id(conv), "w3tx"), lambda: ops.pack_weight_bf16x3(flipped())), None, conv.in_channels)
        return ops.conv3x3(g, self.packed((id(conv), "w3t"), lambda: ops.pack_conv3x3_weight(flipped())), None, conv.in_channels)

    def linear_bwd(self, g, weight, split=None, key=None):
        """d/d rows of rows x weight^T: g [M, n_out] x weight [n_out, K] -> ([M, K],), or its column blocks (K1, K2) as two dense
        tensors when the forward operand was a concat."""
        n_out = weight.shape[0]
        w2 = weight.detach().reshape(n_out, -1)
        kid = key if key is not None else (id(weight), "w1t")
        split16 = self.model.backward_precision == "bf16x3" and n_out % 32 == 0      # gradient rows split on the fly into bf16 hi + lo

        def mm(sub, cols):
            wt = lambda: cols.t().contiguous()
            if split16:
                return ops.gemm_bf16x3(g, self.packed((kid, sub, "x"), lambda: ops.pack_weight_bf16x3(wt())), cols.shape[1])
            return ops.gemm(g, self.packed((kid, sub), wt))
        if split is None:
            return (mm(-1, w2),)
        K1 = split[0]
        return mm(0, w2[:, :K1]), mm(1, w2[:, K1:])

    def row_planes(self, a, a2=None):
        """Raw fp32 rows [M, K1] (| [M, K2]) -> f16x3 operand planes [M, K1 + K2] at the raw-stream scale (identity GroupNorm-apply
        over a [1, M, 1, K] view, the concat never materialised)."""
        M, K1 = a.shape
        K = K1 + (0 if a2 is None else a2.shape[1])
        if a2 is None and K > 2048:                      # the kernel takes up to 3072 channels: fold wide rows (same memory)
            r = next(r for r in (2, 4, 8, 16) if K % (32 * r) == 0 and K // r <= 2048)
            pl = self.row_planes(a.view(M * r, K // r))
            return ops.SplitPlanes(pl.buf, (M, K), pl.fmt, pl.scale)
        key = ("identity_gn", 1, K)
        if key not in self.model._packed:
            st = torch.zeros((1, 32, 2), device=a.device, dtype=torch.float32)
            st[:, :, 1] = 1.0
            self.model._packed[key] = (st, torch.ones(K, device=a.device), torch.zeros(K, device=a.device))
        st, one, zero = self.model._packed[key]
        pl = ops.groupnorm_apply(a.view(1, M, 1, K1), st, one, zero, x2=None if a2 is None else a2.view(1, M, 1, -1), act=0, groups=32,
                                 split="f16x3", split_scale=ops.F16X3_RAW_SCALE)
        return ops.SplitPlanes(pl.buf, (M, K), pl.fmt, pl.scale)

    def linear(self, a, conv1x1, a2=None, residual=None, colsum=False):
        return self.gemm(a, conv1x1.weight, conv1x1.bias, a2=a2, residual=residual, colsum=colsum)

    def cat_weight(self, attn, names):
        """Row-concatenated projection weights (q | k | v of a self-attention): one GEMM, packed qkv rows for dua_attention."""
        key = (id(attn), names)
        if key not in self.model._packed:
            self.model._packed[key] = torch.cat([getattr(attn, n).weight.detach() for n in names]).contiguous()
        return self.model._packed[key]

    def context_kv(self, blk, enc):
        """to_k / to_v of the text context for one transformer block: from the per-context cache when the forward was given one
        (they depend on neither the sample nor the timestep), else computed here."""
        if self.kv is not None:
            return self.kv[id(blk)]
        e2 = enc.reshape(enc.shape[0] * enc.shape[1], -1)
        return self.gemm(e2, blk.attn2.to_k.weight, None), self.gemm(e2, blk.attn2.to_v.weight, None)

    def temb(self, resnet):
        """time_emb_proj(silu(emb)) of one ResnetBlock2D: a column block of ONE GEMM over all blocks (same emb everywhere)."""
        tab = self.model._temb_table()
        if self.temb_all is None:
            self.temb_all = ops.gemm(self.emb, tab["w"], bias=tab["b"], act_in=1)
        o, n = tab["slices"][id(resnet)]
        return self.temb_all[:, o:o + n]


class UNet2DConditionModel(nn.Module):
    """diffusers.UNet2DConditionModel for the SD-1.x configuration family (conv proj_in/out, one transformer block per
    attention, additive time embedding, 8 heads)."""

    def __init__(self, in_channels=4, out_channels=4, block_out_channels=(320, 640, 1280, 1280), layers_per_block=2,
                 down_block_types=("CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "DownBlock2D"),
                 up_block_types=("UpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D"),
                 attention_head_dim=8, cross_attention_dim=768, norm_num_groups=32, norm_eps=1e-5):
        super().__init__()
        assert norm_eps == 1e-5, "GroupNorm eps of the resnets is the nn.GroupNorm default in this build"
        boc, L, G, heads, cd = tuple(block_out_channels), layers_per_block, norm_num_groups, attention_head_dim, cross_attention_dim
        self.config = dict(in_channels=in_channels, out_channels=out_channels, block_out_channels=boc, layers_per_block=L,
                           down_block_types=tuple(down_block_types), up_block_types=tuple(up_block_types),
                           attention_head_dim=heads, cross_attention_dim=cd, norm_num_groups=G, norm_eps=norm_eps)
        ted = boc[0] * 4
        self.conv_in = nn.Conv2d(in_channels, boc[0], 3, padding=1)
        self.time_embedding = TimestepEmbedding(boc[0], ted)
        self.down_blocks = nn.ModuleList()
        ch = boc[0]
        for i, (bt, co) in enumerate(zip(down_block_types, boc)):
            rs, at = [], []
            for _ in range(L):
                rs.append(ResnetBlock2D(ch, co, ted, G))
                ch = co
                if bt.startswith("CrossAttn"):
                    at.append(Transformer2DModel(co, cd, heads, G))
            self.down_blocks.append(_Block(rs, at or None, down=Downsample2D(co) if i != len(boc) - 1 else None))
        self.mid_block = _Block([ResnetBlock2D(ch, ch, ted, G), ResnetBlock2D(ch, ch, ted, G)], [Transformer2DModel(ch, cd, heads, G)])
        self.up_blocks = nn.ModuleList()
        rev = list(reversed(boc))
        out_ch = rev[0]
        for i, bt in enumerate(up_block_types):
            prev, out_ch = out_ch, rev[i]
            skip_in = rev[min(i + 1, len(boc) - 1)]
            rs, at = [], []
            for j in range(L + 1):
                rs.append(ResnetBlock2D((prev if j == 0 else out_ch) + (skip_in if j == L else out_ch), out_ch, ted, G))
                if bt.startswith("CrossAttn"):
                    at.append(Transformer2DModel(out_ch, cd, heads, G))
            self.up_blocks.append(_Block(rs, at or None, up=Upsample2D(out_ch) if i != len(boc) - 1 else None))
        self.conv_norm_out = nn.GroupNorm(G, boc[0])
        self.conv_out = nn.Conv2d(boc[0], out_channels, 3, padding=1)
        self._packed = {}
        self.precision = "fp32"          # "fp32" | "f16x3" | "bf16x3" (see guided_diffusion/unet_openai.py)
        self.auto_range_check = True     # f16x3: first forward after a weight load runs with the range counters on (range_guard.py)
        self._range_checked = False
        self._range_probe = None
        # hipGraph replay of the forward (enable_hip_graph): at the CFG batch of 2 one forward is ~1000 launches of a few
        # microseconds each and the host cannot issue them fast enough; a captured graph removes the launch path.
        self._graphs = None
        # products of backward_sample(): "fp32" exact fp32 MFMA, "bf16x3" split-bf16 (fp32 exponent range, so gradients need no
        # scale; ~2^-16 per product) for the convolutions / linears; the attention gradients are exact fp32 in both
        self.backward_precision = "fp32"
        # f16x3: launches that cannot fill the chip (the CFG batch of 2 gives the 32^2 / 16^2 / 8^2 levels 10 - 40 tiles for 256 CUs)
        # run split along K with a deterministic second-stage sum (dua_set_splitk_workspace): ~1.6x on the whole sampling job
        self.split_k = True
        # to_k / to_v of the text context computed once per context tensor instead of in every denoiser call (identical values)
        # f16x3: norm2(hidden + temb) of a ResnetBlock2D from the column sums of conv1's epilogue with the time-embedding bias folded into
        # the statistics and the affine coefficients (no add pass, no statistics passes)
        self.fold_temb = True
        self.cache_context = True
        self._kv_cache = []

    def _context_kv(self, enc):
        """Keys / values of every cross-attention for this text context, computed once per (context tensor, precision): the reference
        pipeline passes the same prompt embeddings to all 21 denoiser calls of a job.  Held per context TENSOR OBJECT (identity +
        in-place version counter; the cache keeps the tensor alive, so its identity cannot be recycled); two entries (prompt, stacked
        Monte-Carlo prompt)."""
        if not self.cache_context:
            return None
        key = (id(enc), enc._version, tuple(enc.shape), self.precision, bool(self.split_k))
        for k_, ref, kv in self._kv_cache:
            if k_ == key and ref is enc:
                return kv
        ctx = _Ctx(self)
        prev = ops.set_split_k(self.split_k and self.precision == "f16x3", enc.device)
        try:
            blocks = [b for m in self.modules() if isinstance(m, Transformer2DModel) for b in m.transformer_blocks]
            kv = {id(b): ctx.context_kv(b, enc) for b in blocks}
        finally:
            ops.set_split_k(prev, enc.device)
        self._kv_cache = [(key, enc, kv)] + self._kv_cache[:1]
        return kv

    def invalidate_packed_weights(self):
        """Drop the packed weights (and the captured graphs that hold them); the next forward repacks."""
        self._packed = {}
        self._range_checked = False
        self._kv_cache = []
        if getattr(self, "_graphs", None) is not None:
            self._graphs = {}

    def enable_hip_graph(self, on=True):
        """Capture the forward once per (input shapes, precision) into a HIP graph and replay it afterwards."""
        self._graphs = {} if on else None
        return self

    def _apply(self, fn, *a, **k):
        self._packed = {}
        self._range_checked = False
        self._kv_cache = []
        if getattr(self, "_graphs", None) is not None:
            self._graphs = {}
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._packed = {}
        self._range_checked = False
        self._kv_cache = []
        if getattr(self, "_graphs", None) is not None:
            self._graphs = {}
        return super().load_state_dict(*a, **k)

    def _temb_table(self):
        tab = self._packed.get("temb_table")
        if tab is None:
            blocks = [m for m in self.modules() if isinstance(m, ResnetBlock2D)]
            slices, o = {}, 0
            for b in blocks:
                slices[id(b)] = (o, b.cout)
                o += b.cout
            tab = dict(w=torch.cat([b.time_emb_proj.weight.detach() for b in blocks]).contiguous(),
                       b=torch.cat([b.time_emb_proj.bias.detach() for b in blocks]).contiguous(), slices=slices)
            self._packed["temb_table"] = tab
        return tab

    @torch.no_grad()
    def forward(self, sample, timestep, encoder_hidden_states=None, cross_attention_kwargs=None, return_dict=True, **kw):
        if not sample.is_cuda:
            raise RuntimeError("UNet2DConditionModel.forward: the HIP path needs GPU tensors; this package has no CPU fallback")
        assert encoder_hidden_states is not None, "encoder_hidden_states is required (cross-attention UNet)"
        B = sample.shape[0]
        t = torch.as_tensor(timestep, device=sample.device)
        t = (t.reshape(1).expand(B) if t.ndim == 0 else t).to(torch.int64).contiguous()
        enc = encoder_hidden_states.to(torch.float32).contiguous()
        sample = sample.float().contiguous()
        gen = params_generation(self)                                  # in-place weight edits invalidate packed weights and graphs
        if gen != getattr(self, "_packed_gen", None):
            if self._packed:
                self.invalidate_packed_weights()
            self._packed_gen = gen
        kv = self._context_kv(enc)                                     # cross-attention keys / values of this context (cached), or None
        if range_guard.due(self):                                      # first forward after a weight load: range counters on (range_guard.py)
            out = range_guard.checked_forward(self, lambda: self._forward_impl(sample, t, enc, kv=kv))
        elif self._graphs is None:
            out = self._forward_impl(sample, t, enc, kv=kv)
        else:
            key = (tuple(sample.shape), tuple(enc.shape), self.precision, ops.ATTENTION_MODE, bool(self.split_k), kv is not None, bool(self.fold_temb))
            entry = self._graphs.get(key)
            if entry is None:
                self._forward_impl(sample, t, enc, kv=kv)              # eager once: packs the weights, warms the allocator
                sx, st, se = sample.clone(), t.clone(), enc.clone()
                skv = None if kv is None else {b: (k_.clone(), v_.clone()) for b, (k_, v_) in kv.items()}
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    so = self._forward_impl(sx, st, se, kv=skv)
                entry = self._graphs[key] = [graph, sx, st, se, so, skv, kv]
            graph, sx, st, se, so, skv, last_kv = entry
            sx.copy_(sample), st.copy_(t)
            if skv is None:
                se.copy_(enc)
            elif kv is not last_kv:                                    # another context: refresh the graph's static keys / values
                for b, (k_, v_) in kv.items():
                    skv[b][0].copy_(k_), skv[b][1].copy_(v_)
                entry[6] = kv
            graph.replay()
            out = so.clone()
        if return_dict:
            return UNet2DConditionOutput(sample=out)
        return (out,)

    def _forward_impl(self, sample, t, enc, tape=None, kv=None):
        """sample [B,4,h,w] fp32, t [B] int64, enc [B,Tk,ctx] fp32, all on the GPU -> [B,4,h,w].  tape: list receiving the
        activations backward_sample() needs (None: nothing is kept); kv: cached cross-attention keys / values of enc."""
        prev_split_k = ops.set_split_k(self.split_k and self.precision == "f16x3", sample.device)
        try:
            return self._forward_body(sample, t, enc, tape, kv)
        finally:
            ops.set_split_k(prev_split_k, sample.device)

    def _forward_body(self, sample, t, enc, tape, kv=None):
        ctx = _Ctx(self)
        ctx.kv = kv
        te = self.time_embedding
        emb = ops.timestep_embedding(t, self.conv_in.out_channels)        # [cos | sin], freq_shift 0: the ADM embedding
        emb = ops.gemm(emb, te.linear_1.weight, bias=te.linear_1.bias)
        ctx.emb = ops.gemm(emb, te.linear_2.weight, bias=te.linear_2.bias, act_in=1)
        h = ops.nchw_to_nhwc(sample, 4)
        h = ops.conv3x3(h, ctx.packed((id(self.conv_in), "w3"), lambda: ops.pack_conv3x3_weight(self.conv_in.weight.detach().contiguous())),
                        self.conv_in.bias, self.conv_in.out_channels, colsum=True)
        res = [h]
        probe = getattr(self, "_range_probe", None)
        for i, blk in enumerate(self.down_blocks):
            if probe is not None and i:
                probe.note(f"down_blocks.{i - 1}", self.down_blocks[i - 1])
            for j, r in enumerate(blk.resnets):
                h = r.run(ctx, h, tape=tape)
                if hasattr(blk, "attentions"):
                    h = blk.attentions[j].run(ctx, h, enc, tape=tape)
                res.append(h)
            if hasattr(blk, "downsamplers"):
                conv = blk.downsamplers[0].conv
                h = ops.pick_even(ctx.conv3x3(ctx.raw_planes(h), conv, conv.out_channels))      # stride 2 / pad 1 = stride 1 at the even pixels
                res.append(h)
        if probe is not None:
            probe.note(f"down_blocks.{len(self.down_blocks) - 1}", self.down_blocks[-1])
        h = self.mid_block.resnets[0].run(ctx, h, tape=tape)
        h = self.mid_block.attentions[0].run(ctx, h, enc, tape=tape)
        h = self.mid_block.resnets[1].run(ctx, h, tape=tape)
        if probe is not None:
            probe.note("mid_block", self.mid_block)
        for i, blk in enumerate(self.up_blocks):
            if probe is not None and i:
                probe.note(f"up_blocks.{i - 1}", self.up_blocks[i - 1])
            for j, r in enumerate(blk.resnets):
                h = r.run(ctx, h, res.pop(), tape=tape)
                if hasattr(blk, "attentions"):
                    h = blk.attentions[j].run(ctx, h, enc, tape=tape)
            if hasattr(blk, "upsamplers"):
                conv = blk.upsamplers[0].conv
                h = ctx.conv3x3(ctx.raw_planes(h, resample=1), conv, conv.out_channels, colsum=True)
        if probe is not None:
            probe.note(f"up_blocks.{len(self.up_blocks) - 1}", self.up_blocks[-1])
        G = self.conv_norm_out.num_groups
        st = ops.groupnorm_stats(h, groups=G, eps=self.conv_norm_out.eps)
        if tape is not None:
            tape.append((h, st))
        h = ops.groupnorm_apply(h, st, self.conv_norm_out.weight, self.conv_norm_out.bias, act=1, groups=G, split=ctx.split_acts)
        return ctx.conv3x3(h, self.conv_out, self.conv_out.out_channels, nchw_out=True)

    # ---- input gradient (gradient-mode guidance, reference uncertainty_guidance.py:83-108,121-129)
    @torch.no_grad()
    def forward_with_tape(self, sample, timestep, encoder_hidden_states):
        """Eager forward in the model's precision that also returns the tape backward_sample() consumes:
        ``out, tape = unet.forward_with_tape(x, t, enc)``.  The tape holds fp32 activations (GroupNorm / LayerNorm inputs,
        qkv rows, GEGLU pre-activations) of this call; it is used once."""
        if not sample.is_cuda:
            raise RuntimeError("UNet2DConditionModel.forward_with_tape: the HIP path needs GPU tensors; this package has no CPU fallback")
        B = sample.shape[0]
        t = torch.as_tensor(timestep, device=sample.device)
        t = (t.reshape(1).expand(B) if t.ndim == 0 else t).to(torch.int64).contiguous()
        gen = params_generation(self)
        if gen != getattr(self, "_packed_gen", None):
            if self._packed:
                self.invalidate_packed_weights()
            self._packed_gen = gen
        tape = []
        enc = encoder_hidden_states.to(torch.float32).contiguous()
        out = self._forward_impl(sample.float().contiguous(), t, enc, tape=tape, kv=self._context_kv(enc))
        return out, tape

    @torch.no_grad()
    def backward_sample(self, tape, grad_out):
        """Vector-Jacobian product of the forward that recorded ``tape``: d<grad_out, unet(sample)>/d sample, [B,4,h,w] fp32.
        Every product of this pass runs on the exact fp32 engines (dua_conv3x3 / dua_gemm with flipped / transposed weights,
        attention probabilities recomputed from the taped qkv rows); weights get no gradient."""
        ctx = _Ctx(self)
        tape = list(tape)
        G = self.conv_norm_out.num_groups
        g = ctx.conv3x3_bwd(ops.nchw_to_nhwc(ops._f32c(grad_out), self.conv_out.out_channels), self.conv_out)
        h, st = tape.pop()
        g = ops.groupnorm_bwd(h, st, self.conv_norm_out.weight, self.conv_norm_out.bias, g, act=1, groups=G)[0]
        gres = []                                                     # gradients of the skip tensors, in the order the down path consumes them
        for blk in reversed(self.up_blocks):
            if hasattr(blk, "upsamplers"):
                gu = ctx.conv3x3_bwd(g, blk.upsamplers[0].conv)       # adjoint of nearest x2: sum over each 2x2 cell
                pooled = ops.resample2x(gu, -1)
                g = ops.axpby(pooled, pooled, 4.0, 0.0)
            for j in reversed(range(len(blk.resnets))):
                if hasattr(blk, "attentions"):
                    g = blk.attentions[j].run_bwd(ctx, tape, g)
                g, gs = blk.resnets[j].run_bwd(ctx, tape.pop(), g)
                gres.append(gs)
        g, _ = self.mid_block.resnets[1].run_bwd(ctx, tape.pop(), g)
        g = self.mid_block.attentions[0].run_bwd(ctx, tape, g)
        g, _ = self.mid_block.resnets[0].run_bwd(ctx, tape.pop(), g)
        for blk in reversed(self.down_blocks):
            if hasattr(blk, "downsamplers"):
                g = ops.axpby(g, gres.pop(), 1.0, 1.0)
                N, Ho, Wo, _ = g.shape
                g = ctx.conv3x3_bwd(ops.place_even(g, 2 * Ho, 2 * Wo), blk.downsamplers[0].conv)
            for j in reversed(range(len(blk.resnets))):
                g = ops.axpby(g, gres.pop(), 1.0, 1.0)
                if hasattr(blk, "attentions"):
                    g = blk.attentions[j].run_bwd(ctx, tape, g)
                g, _ = blk.resnets[j].run_bwd(ctx, tape.pop(), g)
        g = ops.axpby(g, gres.pop(), 1.0, 1.0)
        assert not tape and not gres
        return ops.nhwc_to_nchw(ctx.conv3x3_bwd(g, self.conv_in))


class UNet2DConditionOutput(dict):
    def __init__(self, sample):
        super().__init__(sample=sample)
        self.sample = sample
