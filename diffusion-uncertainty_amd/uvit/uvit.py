"""U-ViT on the gfx950 HIP library.

Host-side mirror of reference ``uvit/uvit.py`` (``UViT.__init__`` :138-201, ``forward`` :203-231, ``Block``
:95-120, ``Attention`` :54-92, ``PatchEmbed`` :123-135) and ``uvit/timm.py`` ``Mlp`` :96-111: same
constructor, same module nesting / state-dict keys as the released checkpoints
(``in_blocks.N.attn.qkv.weight`` ...), same call ``model(x[B,C,h,w], t[B], y[B]) -> [B,C,h,w]``.
nn modules hold parameters only; the arithmetic runs in libdua_hip.so: LayerNorm kernel, fp32-MFMA GEMMs
with fused bias / GELU / residual epilogues, two-source GEMM for ``skip_linear(cat([x, skip]))`` (:116-117),
fp32 attention (the reference upcasts qkv to fp32 and calls SDPA, :71-73).  No CPU path.
"""
import torch
import torch.nn as nn

from .. import _lib, ops, range_guard


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features, out_features)
        self.drop = nn.Dropout(drop)


class Attention(nn.Module):
    def __init__(self, dim, num_heads=8, qkv_bias=False, qk_scale=None, attn_drop=0., proj_drop=0.):
        super().__init__()
        assert qk_scale is None, "custom qk_scale is not on the accelerated path"
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)


class Block(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio=4., qkv_bias=False, qk_scale=None, act_layer=nn.GELU,
                 norm_layer=nn.LayerNorm, skip=False, use_checkpoint=False):
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias, qk_scale=qk_scale)
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer)
        self.skip_linear = nn.Linear(2 * dim, dim) if skip else None

    def packed(self, cache, precision):
        """Weights of this block in the layout dua_uvit_block_fwd takes for `precision` (cached on the model)."""
        key = (id(self), "block", precision)
        if key not in cache:
            pk = (lambda lin_: ops.pack_weight_f16x3(lin_.weight.detach().contiguous())) if precision == "f16x3" else \
                (lambda lin_: lin_.weight.detach().contiguous())
            d = dict(norm1_w=self.norm1.weight, norm1_b=self.norm1.bias, norm2_w=self.norm2.weight, norm2_b=self.norm2.bias,
                     qkv_w=pk(self.attn.qkv), qkv_b=self.attn.qkv.bias, proj_w=pk(self.attn.proj), proj_b=self.attn.proj.bias,
                     fc1_w=pk(self.mlp.fc1), fc1_b=self.mlp.fc1.bias, fc2_w=pk(self.mlp.fc2), fc2_b=self.mlp.fc2.bias,
                     hidden=self.mlp.fc1.weight.shape[0])
            if self.skip_linear is not None:
                d.update(skip_w=pk(self.skip_linear), skip_b=self.skip_linear.bias)
            cache[key] = d
        return cache[key]

    def run(self, lin, x, B, L, skip=None, precision="fp32", block_cache=None, want_planes=False):
        """x, skip: [B*L, D]; lin(a, linear_module, **kw) is the model's precision-aware GEMM."""
        D = x.shape[1]
        ch = D // self.attn.num_heads
        if (block_cache is not None and precision in ("fp32", "f16x3") and D % 32 == 0 and self.mlp.fc1.weight.shape[0] % 32 == 0 and
                ch % 8 == 0 and ch <= 128 and ops.ATTENTION_MODE == "auto"):
            # the whole block as ONE C-ABI call (dua_uvit_block_fwd: the same kernels in the same order, bit-identical)
            return ops.uvit_block(x, skip if self.skip_linear is not None else None, B, L, self.attn.num_heads,
                                  self.packed(block_cache, precision), precision=precision, eps=self.norm1.eps, want_planes=want_planes)
        if self.skip_linear is not None:
            xp, sp = getattr(x, "_dua_planes", None), getattr(skip, "_dua_planes", None)
            if precision == "f16x3" and xp is not None and sp is not None:       # both operands exist as planes: presplit two-source GEMM
                key = (id(self.skip_linear), "f16x3")
                cache = block_cache if block_cache is not None else lin.__self__._planes          # (the model's packed-weight cache)
                if key not in cache:
                    cache[key] = ops.pack_weight_f16x3(self.skip_linear.weight.detach().contiguous())
                pl = lambda b: ops.SplitPlanes(b, tuple(x.shape), "f16x3", ops.F16X3_RAW_SCALE)      # noqa: E731
                x = ops.gemm_f16x3_presplit2(pl(xp), pl(sp), cache[key], D, bias=self.skip_linear.bias)
            else:
                x = lin(x, self.skip_linear, a2=skip)                                              # :116-117
        # split modes: LayerNorm / GELU outputs go out in the operand format of the LDS-DMA GEMMs (bf16 planes)
        # ("f16x3": the fp32-equivalent split, fp16 hi / lo planes)
        fast = (precision if precision == "f16x3" else precision == "bf16x3") if D % 32 == 0 else False
        h = ops.layernorm(x, self.norm1.weight, self.norm1.bias, self.norm1.eps, split=fast)
        qkv = lin(h, self.attn.qkv)
        a = ops.qkv_attention(qkv, B, L, self.attn.num_heads, D // self.attn.num_heads,
                              precision=precision if precision in ("fp32", "f16x3") else "bf16x3",
                              planes_out=(fast == "f16x3"))                                               # :69-73
        x = lin(a, self.attn.proj, residual=x)                                                     # :118
        h = ops.layernorm(x, self.norm2.weight, self.norm2.bias, self.norm2.eps, split=fast)
        h = lin(h, self.mlp.fc1, act_out=2, split_out=bool(fast))                                  # fc1 + GELU
        if want_planes and precision == "f16x3" and isinstance(h, ops.SplitPlanes):
            key = (id(self.mlp.fc2), "f16x3")
            cache = block_cache if block_cache is not None else lin.__self__._planes          # (the model's packed-weight cache)
            if key not in cache:
                cache[key] = ops.pack_weight_f16x3(self.mlp.fc2.weight.detach().contiguous())
            y, yp = ops.gemm_f16x3_presplit2(h, None, cache[key], D, bias=self.mlp.fc2.bias, residual=x, planes_out=True)
            y._dua_planes = yp.buf
            return y
        return lin(h, self.mlp.fc2, residual=x)                                                    # :119


class PatchEmbed(nn.Module):
    def __init__(self, patch_size, in_chans=3, embed_dim=768):
        super().__init__()
        self.patch_size = patch_size
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)


class UViT(nn.Module):
    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768, depth=12, num_heads=12, mlp_ratio=4.,
                 qkv_bias=False, qk_scale=None, norm_layer=nn.LayerNorm, mlp_time_embed=False, num_classes=-1,
                 use_checkpoint=False, conv=True, skip=True):
        super().__init__()
        assert not conv, "final 3x3 conv (conv=True) is not used by the released ImageNet models (load_pretrained_models.py:36)"
        self.num_features = self.embed_dim = embed_dim
        self.num_classes = num_classes
        self.in_chans = in_chans
        self.patch_embed = PatchEmbed(patch_size=patch_size, in_chans=in_chans, embed_dim=embed_dim)
        num_patches = (img_size // patch_size) ** 2
        self.time_embed = nn.Sequential(nn.Linear(embed_dim, 4 * embed_dim), nn.SiLU(),
                                        nn.Linear(4 * embed_dim, embed_dim)) if mlp_time_embed else nn.Identity()
        if self.num_classes > 0:
            self.label_emb = nn.Embedding(self.num_classes, embed_dim)
            self.extras = 2
        else:
            self.extras = 1
        self.pos_embed = nn.Parameter(torch.zeros(1, self.extras + num_patches, embed_dim))
        kw = dict(dim=embed_dim, num_heads=num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, qk_scale=qk_scale, norm_layer=norm_layer)
        self.in_blocks = nn.ModuleList([Block(**kw) for _ in range(depth // 2)])
        self.mid_block = Block(**kw)
        self.out_blocks = nn.ModuleList([Block(skip=skip, **kw) for _ in range(depth // 2)])
        self.norm = norm_layer(embed_dim)
        self.patch_dim = patch_size ** 2 * in_chans
        self.decoder_pred = nn.Linear(embed_dim, self.patch_dim, bias=True)
        self.final_layer = nn.Identity()
        nn.init.trunc_normal_(self.pos_embed, std=.02)
        self.apply(self._init_weights)
        self.img_size = img_size
        self.patch_size = patch_size
        self.precision = "fp32"          # or "f16x3" (fp32-equivalent) / "bf16x3" (see guided_diffusion/unet_openai.py)
        self.use_block_abi = True        # fp32 / f16x3: every Block runs as ONE C-ABI call (bit-identical to the composition)
        # f16x3: register the split-K workspace for the forward (ops.set_split_k): GEMM launches whose last round of tiles would be
        # mostly idle (N = 1152 projections at batch 64: 585 tiles on 256 CUs) run their tail rows split along K.  Opt-in: results
        # then depend on the batch size at fp32-rounding level
        self.split_k = False
        self.stream_split = 1            # > 1: the batch in that many chunks on HIP streams of their own (see _forward_streams)
        self.stream_split_max_batch = 1 << 30
        self._streams = []
        self.auto_range_check = True     # f16x3: first forward after a weight load runs with the range counters on (range_guard.py)
        self._range_checked = False
        self._range_probe = None
        self.skip_planes = True          # f16x3: block outputs that feed a skip_linear also leave the fc2 epilogue as f16x3 planes (A/B switch)
        self._planes = {}

    def _apply(self, fn, *a, **k):
        self._planes = {}
        self._range_checked = False
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._planes = {}
        self._range_checked = False
        return super().load_state_dict(*a, **k)

    def _lin(self, a, mod, a2=None, residual=None, act_out=0, split_out=False):
        if isinstance(a, ops.SplitPlanes):                # operand already in a split format: LDS-DMA GEMM
            key = (id(mod), a.fmt)
            if key not in self._planes:
                pack = ops.pack_weight_f16x3 if a.fmt == "f16x3" else ops.pack_weight_bf16x3
                self._planes[key] = pack(mod.weight.detach().contiguous())
            return ops.gemm_bf16x3_presplit(a, self._planes[key], mod.weight.shape[0], bias=mod.bias, residual=residual,
                                            act_out=act_out, split_out=split_out)
        K = a.shape[1] + (0 if a2 is None else a2.shape[1])
        if self.precision == "f16x3" and K % 32 == 0 and a.shape[1] % 32 == 0:           # raw fp32 operand split on the fly
            key = (id(mod), "f16x3")
            if key not in self._planes:
                self._planes[key] = ops.pack_weight_f16x3(mod.weight.detach().contiguous())
            return ops.gemm_f16x3(a, self._planes[key], mod.weight.shape[0], bias=mod.bias, a2=a2, residual=residual, act_out=act_out)
        if self.precision != "fp32" and K % 32 == 0 and a.shape[1] % 32 == 0:
            if id(mod) not in self._planes:
                self._planes[id(mod)] = ops.pack_weight_bf16x3(mod.weight.detach().contiguous())
            return ops.gemm_bf16x3(a, self._planes[id(mod)], mod.weight.shape[0], bias=mod.bias, a2=a2, residual=residual, act_out=act_out)
        return ops.gemm(a, mod.weight, bias=mod.bias, a2=a2, residual=residual, act_out=act_out)

    def _init_weights(self, m):
        if isinstance(m, nn.Linear):
            nn.init.trunc_normal_(m.weight, std=.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    @torch.no_grad()
    def forward(self, x, timesteps, y=None):
        if not x.is_cuda:
            raise RuntimeError("UViT.forward: the HIP path needs GPU tensors; this package has no CPU fallback")
        if range_guard.due(self):
            return range_guard.checked_forward(self, lambda: self._forward_single(x, timesteps, y))
        n = int(self.stream_split)
        if n > 1 and 2 * n <= x.shape[0] <= self.stream_split_max_batch:
            return self._forward_streams(x, timesteps, y, n)
        return self._forward_single(x, timesteps, y)

    def _forward_single(self, x, timesteps, y):
        if not (self.split_k and self.precision == "f16x3"):
            return self._forward_body(x, timesteps, y)
        prev = ops.set_split_k(True, x.device)
        try:
            return self._forward_body(x, timesteps, y)
        finally:
            ops.set_split_k(prev, x.device)

    def _forward_streams(self, x, timesteps, y, n):
        """The batch in n chunks, each chunk's whole forward on a HIP stream of its own (results identical per image: every kernel is
        batch-invariant).  At batch 64 a linear of U-ViT-H/2 is 390 - 1170 tiles for 256 persistent workgroups - 1.5 to 4.6 rounds, the
        last one partly empty - and the HBM-bound kernels between the GEMMs leave the matrix cores idle; with two half batches in flight
        the workgroups of one chunk's kernel start on the CUs the other chunk's kernel has left.  Stream-K / split-K use ONE
        process-global workspace, so they stay off on this path."""
        cur = torch.cuda.current_stream(x.device)
        if len(self._streams) < n:
            self._streams = [torch.cuda.Stream(device=x.device) for _ in range(n)]
        xs, ts = torch.chunk(x, n), torch.chunk(timesteps, n)
        ys = torch.chunk(y, n) if y is not None else [None] * len(xs)
        outs = []
        # the split-K / stream-K workspace is ONE process-global buffer: concurrent chunks would fold each other's parts.  Force it off
        # here - also when the CALLER had registered it (ops.set_split_k(True)) - and restore the caller's setting afterwards.
        prev_splitk = ops.set_split_k(False, x.device)
        prev_conc = _lib.lib().dua_set_hgemm_concurrent(1)       # tile widths by work: the other chunk's kernels fill partly empty rounds
        try:
            return self._forward_chunks(xs, ts, ys, cur, outs)
        finally:
            _lib.lib().dua_set_hgemm_concurrent(prev_conc)
            ops.set_split_k(prev_splitk, x.device)

    def _forward_chunks(self, xs, ts, ys, cur, outs):
        for i, (xc, tc, yc) in enumerate(zip(xs, ts, ys)):
            st = self._streams[i]
            st.wait_stream(cur)
            n_packed = len(self._planes)
            with torch.cuda.stream(st):
                outs.append(self._forward_body(xc.contiguous(), tc.contiguous(), yc.contiguous() if yc is not None else None))
            if len(self._planes) != n_packed:
                st.synchronize()         # this chunk packed weights on ITS stream: the other chunks read them from the shared cache
        for o, st in zip(outs, self._streams):
            cur.wait_stream(st)
            o.record_stream(cur)
        return torch.cat(outs)

    def _forward_body(self, x, timesteps, y=None):
        B, C, H, W = x.shape
        P, D = self.patch_size, self.embed_dim
        rows = ops.patchify(x.float().contiguous(), P)                                     # [B*L, C*P*P]
        L = rows.shape[0] // B
        w = self.patch_embed.proj.weight
        tok = ops.gemm(rows, w.reshape(D, -1), bias=self.patch_embed.proj.bias)            # :205
        tt = ops.timestep_embedding(timesteps.to(torch.int64).contiguous(), D)
        if not isinstance(self.time_embed, nn.Identity):
            tt = ops.gemm(tt, self.time_embed[0].weight, bias=self.time_embed[0].bias)
            tt = ops.gemm(tt, self.time_embed[2].weight, bias=self.time_embed[2].bias, act_in=1)
        yy = y.to(torch.int64).contiguous() if y is not None else None
        assert (yy is not None) == (self.extras == 2), "label token count does not match the positional table"
        h = ops.uvit_assemble(tok, tt, self.label_emb.weight if yy is not None else None, yy,
                              self.pos_embed.reshape(-1, D), B, L, D)                      # :207-216
        T = L + self.extras
        h = h.view(B * T, D)
        skips = []
        bc = self._planes if self.use_block_abi else None
        # f16x3: a block whose output is the x or the long skip of a skip_linear also leaves it as f16x3 planes (the epilogue of fc2 writes
        # them beside the fp32 tokens), so that skip_linear(cat([x, skip])) runs on the presplit two-source GEMM
        wp = self.precision == "f16x3" and self.skip_planes and bool(self.out_blocks) and self.out_blocks[0].skip_linear is not None
        probe = self._range_probe
        for i, blk in enumerate(self.in_blocks):
            h = blk.run(self._lin, h, B, T, precision=self.precision, block_cache=bc, want_planes=wp)
            skips.append(h)
            if probe is not None:
                probe.note(f"in_blocks.{i}", blk)
        h = self.mid_block.run(self._lin, h, B, T, precision=self.precision, block_cache=bc, want_planes=wp)
        if probe is not None:
            probe.note("mid_block", self.mid_block)
        for i, blk in enumerate(self.out_blocks):
            h = blk.run(self._lin, h, B, T, skips.pop(), precision=self.precision, block_cache=bc,
                        want_planes=wp and i + 1 < len(self.out_blocks))
            if probe is not None:
                probe.note(f"out_blocks.{i}", blk)
        h = ops.layernorm(h, self.norm.weight, self.norm.bias, self.norm.eps)
        h = self._lin(h, self.decoder_pred)                                                # [B*T, P*P*C]
        return ops.unpatchify(h, B, L, self.extras, C, P)                                  # :228-229
