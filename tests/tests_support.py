"""State-dict shape tables (names -> shapes) of the checkpoints the path loads,
derived from the oracle's block enumeration (SURVEY.md Appendix A key patterns)."""
from oracle import adm


def resblock_shapes(cin, cout, emb):
    s = {"in_layers.0.weight": (cin,), "in_layers.0.bias": (cin,),
         "in_layers.2.weight": (cout, cin, 3, 3), "in_layers.2.bias": (cout,),
         "emb_layers.1.weight": (2 * cout, emb), "emb_layers.1.bias": (2 * cout,),
         "out_layers.0.weight": (cout,), "out_layers.0.bias": (cout,),
         "out_layers.3.weight": (cout, cout, 3, 3), "out_layers.3.bias": (cout,)}
    if cin != cout:
        s["skip_connection.weight"] = (cout, cin, 1, 1)
        s["skip_connection.bias"] = (cout,)
    return s


def attn_shapes(C):
    return {"norm.weight": (C,), "norm.bias": (C,), "qkv.weight": (3 * C, C, 1), "qkv.bias": (3 * C,),
            "proj_out.weight": (C, C, 1), "proj_out.bias": (C,)}


def adm_state_shapes(cfg):
    mc = cfg["model_channels"]
    ted = 4 * mc
    s = {"time_embed.0.weight": (ted, mc), "time_embed.0.bias": (ted,),
         "time_embed.2.weight": (ted, ted), "time_embed.2.bias": (ted,)}
    if cfg.get("num_classes") is not None:
        s["label_emb.weight"] = (cfg["num_classes"], ted)
    inp, mid, out = adm.layout(cfg)
    for op in [o for blk in inp for o in blk] + mid + [o for blk in out for o in blk]:
        if op[0] == "conv":
            s[op[1] + ".weight"] = (op[3], op[2], 3, 3)
            s[op[1] + ".bias"] = (op[3],)
        elif op[0] == "res":
            s.update({op[1] + "." + k: v for k, v in resblock_shapes(op[2], op[3], ted).items()})
        else:
            s.update({op[1] + "." + k: v for k, v in attn_shapes(op[2]).items()})
    ch = int(cfg["channel_mult"][0] * mc)
    s.update({"out.0.weight": (ch,), "out.0.bias": (ch,), "out.2.weight": (cfg["out_channels"], ch, 3, 3),
              "out.2.bias": (cfg["out_channels"],)})
    return s


def uvit_state_shapes(cfg):
    D, P, C = cfg["embed_dim"], cfg["patch_size"], cfg["in_chans"]
    L = (cfg["img_size"] // P) ** 2
    extras = 2 if cfg.get("num_classes", -1) > 0 else 1
    s = {"pos_embed": (1, extras + L, D), "patch_embed.proj.weight": (D, C, P, P), "patch_embed.proj.bias": (D,)}
    if cfg.get("num_classes", -1) > 0:
        s["label_emb.weight"] = (cfg["num_classes"], D)

    def blk(p, skip):
        b = {f"{p}.norm1.weight": (D,), f"{p}.norm1.bias": (D,), f"{p}.attn.qkv.weight": (3 * D, D),
             f"{p}.attn.proj.weight": (D, D), f"{p}.attn.proj.bias": (D,), f"{p}.norm2.weight": (D,),
             f"{p}.norm2.bias": (D,), f"{p}.mlp.fc1.weight": (4 * D, D), f"{p}.mlp.fc1.bias": (4 * D,),
             f"{p}.mlp.fc2.weight": (D, 4 * D), f"{p}.mlp.fc2.bias": (D,)}
        if skip:
            b[f"{p}.skip_linear.weight"] = (D, 2 * D)
            b[f"{p}.skip_linear.bias"] = (D,)
        return b
    for i in range(cfg["depth"] // 2):
        s.update(blk(f"in_blocks.{i}", False))
        s.update(blk(f"out_blocks.{i}", True))
    s.update(blk("mid_block", False))
    s.update({"norm.weight": (D,), "norm.bias": (D,), "decoder_pred.weight": (P * P * C, D), "decoder_pred.bias": (P * P * C,)})
    return s


def checkpoint_like_state_dict(sd, t, y, timestep_embedding, gamma_max=10.0, film_gain=5.0, stream_gain=1000.0, seed=3):
    """Give a seeded random ADM state dict the magnitudes of a trained checkpoint - a pure function of (names, values, t, y, seed), so the
    reference side (tests/golden/make_golden.py g23) and the HIP side build the SAME weights: every GroupNorm gamma uniform in
    [0.3, gamma_max] (per-tensor generator seeded from its name); every ResBlock's FiLM projection rescaled so that, for THESE timesteps /
    labels, its `scale` half peaks at +-film_gain and its `shift` half at film_gain / 2 (unet_openai.py:249-253 applies `1 + scale`); the
    input convolution rescaled so that the residual stream peaks at ~stream_gain.  ``timestep_embedding(t, dim)`` is nn.py:103-121 (the
    reference's own function on its side, the oracle's restatement on ours: G5 pins them to each other).  Returns (new state dict, FiLM peak)."""
    import math
    import zlib

    import torch
    import torch.nn.functional as F
    out = {k: v.detach().clone().float() for k, v in sd.items()}
    mc = out["time_embed.0.weight"].shape[1]
    emb = timestep_embedding(t, mc).float()
    emb = F.linear(F.silu(F.linear(emb, out["time_embed.0.weight"], out["time_embed.0.bias"])), out["time_embed.2.weight"], out["time_embed.2.bias"])
    emb = emb + out["label_emb.weight"][y]
    peak = 0.0
    for name in sorted(out):
        p = out[name]
        if name.endswith("in_layers.0.weight") or name.endswith("out_layers.0.weight") or name.endswith("norm.weight") or name == "out.0.weight":
            g = torch.Generator().manual_seed((seed * 1000003 + zlib.crc32(name.encode())) % (2 ** 62))
            out[name] = 0.3 + (gamma_max - 0.3) * torch.rand(p.shape, generator=g)
        elif name.endswith("emb_layers.1.weight"):
            bname = name[:-6] + "bias"
            film = F.linear(F.silu(emb), p, out[bname])
            C = film.shape[1] // 2
            ks, kh = film_gain / film[:, :C].abs().max().item(), 0.5 * film_gain / film[:, C:].abs().max().item()
            w, b = p.clone(), out[bname].clone()
            w[:C] *= ks; b[:C] *= ks
            w[C:] *= kh; b[C:] *= kh
            out[name], out[bname] = w, b
            peak = max(peak, (film[:, :C] * ks).abs().max().item())
        elif name == "input_blocks.0.0.weight":
            out[name] = p * (0.25 * stream_gain / max(1e-6, p.std().item() * math.sqrt(27.0)))       # std ~ gain / 4: peaks at ~gain
    return out, peak


def uvit_checkpoint_like_state_dict(sd, gamma_max=3.0, qkv_gain=1.0, seed=4):
    """The U-ViT counterpart of checkpoint_like_state_dict: LayerNorm gammas uniform in [0.3, gamma_max] (per-tensor generator seeded from
    the name), optionally sharper attention (qkv weights x qkv_gain) and a stronger positional table (x 3).  Pure function of names, values
    and the seed.  (Keep the gains modest: see tests/golden/make_golden.py g24 - the network amplifies rounding chaotically beyond them.)"""
    import zlib

    import torch
    out = {k: v.detach().clone().float() for k, v in sd.items()}
    for name in sorted(out):
        p = out[name]
        if name.endswith("norm1.weight") or name.endswith("norm2.weight") or name == "norm.weight":
            g = torch.Generator().manual_seed((seed * 1000003 + zlib.crc32(name.encode())) % (2 ** 62))
            out[name] = 0.3 + (gamma_max - 0.3) * torch.rand(p.shape, generator=g)
        elif name.endswith("attn.qkv.weight"):
            out[name] = p * qkv_gain
        elif name == "pos_embed":
            out[name] = p * 3.0
    return out
