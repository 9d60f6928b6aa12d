"""Multistep DPM-Solver++ with the centred uncertainty estimator (oracle; test infrastructure).

Follows reference ``schedulers_uncertainty/scheduling_dpm_2_uncertainty_centered.py`` (the class behind
``scheduler_type='dpm_2_uncertainty_centered'``): ``set_timesteps`` :285-367, ``_sigma_to_alpha_sigma_t`` :436-440,
``convert_model_output`` :523-528, first-order update :624-632, second-order midpoint update :694-727, ``step`` :876-977.
Scalars are 0-dim fp32 tensors evaluated as the reference writes them.
"""
import numpy as np
import torch


class DPMState:
    """Grid, sigmas and the multistep history (``model_outputs``, ``lower_order_nums``, ``_step_index``)."""

    def __init__(self, tab, n_steps, after_step, num_steps_uc, timestep_spacing="leading", steps_offset=0, solver_order=2):
        T = tab.num_train_timesteps
        if timestep_spacing == "leading":                                         # :305-310 (note N + 1)
            ratio = T // (n_steps + 1)
            ts = (np.arange(0, n_steps + 1) * ratio).round()[::-1][:-1].copy().astype(np.int64) + steps_offset
        elif timestep_spacing == "linspace":                                      # :301-304
            ts = np.linspace(0, T - 1, n_steps + 1).round()[::-1][:-1].copy().astype(np.int64)
        else:
            raise NotImplementedError(timestep_spacing)
        sig = (((1 - tab.alphas_cumprod) / tab.alphas_cumprod) ** 0.5).numpy()
        sig = np.interp(ts, np.arange(0, len(sig)), sig)                          # :336
        self.sigmas = torch.from_numpy(np.concatenate([sig, [0]]).astype(np.float32))   # final_sigmas_type == 'zero' :338-349
        self.timesteps = ts
        self.window = (int(ts[after_step]), int(ts[after_step + num_steps_uc - 1]))
        self.tab, self.order = tab, solver_order
        self.model_outputs = [None] * solver_order
        self.lower_order_nums, self.step_index = 0, None


def _alpha_sigma(sigma):
    alpha_t = 1 / ((sigma ** 2 + 1) ** 0.5)
    return alpha_t, sigma * alpha_t


def step(st, denoise, eps, t, x, M, draw=torch.randn_like):
    """One ``step()`` of the reference (:876-977) -> dict(prev_sample[, uncertainty, pred_original_sample, pred_epsilon])."""
    if st.step_index is None:                                                     # _init_step_index :857-874
        idx = np.nonzero(st.timesteps == t)[0]
        st.step_index = len(st.timesteps) - 1 if len(idx) == 0 else int(idx[1] if len(idx) > 1 else idx[0])
    i, n = st.step_index, len(st.timesteps)
    lower_order_final = i == n - 1                                                # final_sigmas_type == 'zero' :929-933
    alpha_t, sigma_t = _alpha_sigma(st.sigmas[i])
    m0 = (x - sigma_t * eps) / alpha_t                                            # convert_model_output :523-528
    for k in range(st.order - 1):
        st.model_outputs[k] = st.model_outputs[k + 1]
    st.model_outputs[-1] = m0
    a_t, s_t = _alpha_sigma(st.sigmas[i + 1])
    a_s0, s_s0 = _alpha_sigma(st.sigmas[i])
    lam_t, lam_s0 = torch.log(a_t) - torch.log(s_t), torch.log(a_s0) - torch.log(s_s0)
    h = lam_t - lam_s0
    if st.order == 1 or st.lower_order_nums < 1 or lower_order_final:
        prev = (s_t / s_s0) * x - (a_t * (torch.exp(-h) - 1.0)) * m0              # :624-632
    else:
        a_s1, s_s1 = _alpha_sigma(st.sigmas[i - 1])
        lam_s1 = torch.log(a_s1) - torch.log(s_s1)
        r0 = (lam_s0 - lam_s1) / h
        D0, D1 = m0, (1.0 / r0) * (m0 - st.model_outputs[-2])
        prev = (s_t / s_s0) * x - (a_t * (torch.exp(-h) - 1.0)) * D0 - 0.5 * (a_t * (torch.exp(-h) - 1.0)) * D1   # :716-727
    if st.lower_order_nums < st.order:
        st.lower_order_nums += 1
    out = dict(prev_sample=prev)
    hi, lo = st.window
    if lo <= t <= hi:                                                             # :941-963 (model_output IS m0 here)
        a = st.tab.alphas_cumprod[int(t)]
        x0 = (x - (1 - a) ** 0.5 * m0) / a ** 0.5
        scores = []
        for _ in range(M):
            x_hat = torch.sqrt(a) * x0 + torch.sqrt(1 - a) * draw(x0)
            scores.append(denoise(x_hat, t))
        out.update(uncertainty=(torch.stack(scores, 0) - m0.unsqueeze(0)).pow(2).mean(0), pred_original_sample=x0,
                   pred_epsilon=m0)
    st.step_index += 1
    return out
