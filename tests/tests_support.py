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
