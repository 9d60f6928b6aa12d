"""Batch / timestep driver loop (oracle; test infrastructure).

Follows reference ``generate_samples.py``
``generate_samples_model_scheduler_class_conditioned_from_tensor`` :127-233 and
``..._uvit_from_tensor`` :469-570.
"""
import torch

from . import ddim, dpm


def to_uint8(x):
    """generate_samples.py:203-215."""
    return ((x / 2 + 0.5).clamp(0, 1) * 255.0).round().to(torch.uint8)


@torch.no_grad()
def generate(X_T, y, batch_size, model, tab, n_steps, kind, M, after_step, num_steps_uc,
             num_zigzag=3, predict_next=False, eta=0.0, clip_sample=True, steps_offset=0,
             decode=None, slice3=True):
    """``model(x, t_vec, y, train=False) -> eps_full``; kind in {'zigzag_centered','centered','mc_dropout','dpm'}."""
    from .schedule import leading_timesteps, uc_window
    if kind == "dpm":
        st = dpm.DPMState(tab, n_steps, after_step, num_steps_uc, steps_offset=steps_offset)
        ts, window = st.timesteps, st.window
    else:
        ts = leading_timesteps(tab.num_train_timesteps, n_steps, steps_offset)
        window = uc_window(ts, after_step, num_steps_uc)
    imgs, ucs, scs = [], [], []
    for s in range(0, X_T.shape[0], batch_size):
        x, yb = X_T[s:s + batch_size].clone(), y[s:s + batch_size]

        def call(xx, t, train=False):
            tv = torch.full((xx.shape[0],), int(t), dtype=torch.long)
            o = model(xx, tv, yb, train)
            return o[:, :3] if slice3 else o

        u_l, s_l = [], []
        if kind == "dpm":                 # the reference driver re-calls set_timesteps per batch (generate_samples.py:177)
            st = dpm.DPMState(tab, n_steps, after_step, num_steps_uc, steps_offset=steps_offset)
        for t in ts:
            t = int(t)
            eps = call(x, t)
            if kind == "zigzag_centered":
                o = ddim.step_zigzag_centered(tab, call, eps, t, x, n_steps, window, M, num_zigzag, eta, clip_sample)
            elif kind == "dpm":
                o = dpm.step(st, call, eps, t, x, M)
            elif kind == "centered":
                o = ddim.step_centered(tab, call, eps, t, x, n_steps, window, M, predict_next, eta, clip_sample)
            else:
                o = ddim.step_mc_dropout(tab, lambda a, b: call(a, b, True), eps, t, x, n_steps, window, M, eta, clip_sample)
            if window[0] >= t >= window[1]:
                u_l.append(o["uncertainty"])
                s_l.append(o["pred_epsilon"])
            x = o["prev_sample"]
        ucs.append(torch.stack(u_l, 1))
        scs.append(torch.stack(s_l, 1))
        if decode is not None:
            x = decode(x)
        imgs.append(to_uint8(x))
    return dict(gen_images=torch.cat(imgs), uncertainty=torch.cat(ucs), score=torch.cat(scs))
