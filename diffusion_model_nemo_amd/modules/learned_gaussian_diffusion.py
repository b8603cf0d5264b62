"""Improved-DDPM sampler with a learned, per-pixel interpolated variance.

Mirror of diffusion_model_nemo/modules/learned_gaussian_diffusion.py:14-53. The network emits 2*C channels
(eps | v); log sigma^2 = f log beta_t + (1 - f) log beta~_t with f = (v + 1) / 2. The sampling loop is inherited
and runs the same fused device kernel with ``learned_variance`` set. As in the reference, ``objective`` does not
change the parameterisation here: the first half of the output is always treated as predicted noise.
"""
from typing import Optional

import torch

from .gaussian_diffusion import GaussianDiffusion


class LearnedGaussianDiffusion(GaussianDiffusion):
    learned_variance = True

    def __init__(self, timesteps: int, schedule_name: str, schedule_cfg: Optional[dict] = None,
                 objective: str = "pred_noise"):
        super().__init__(timesteps=timesteps, schedule_name=schedule_name, schedule_cfg=schedule_cfg,
                         objective=objective)
        self._declared_objective = objective
        self.objective = "pred_noise"

    def p_mean_variance(self, model, x, t, model_output=None, return_pred_x_start=False):
        if model_output is None:
            model_output = model(x, t)
        pred_noise, frac_raw = model_output.chunk(2, dim=1)
        min_log = self.extract(self.posterior_log_variance_clipped, t, x.shape)
        max_log = self.extract(torch.log(self.betas), t, x.shape)
        frac = (frac_raw + 1) * 0.5
        model_log_variance = frac * max_log + (1 - frac) * min_log
        model_variance = model_log_variance.exp()
        x_start = self.predict_start_from_noise(x_t=x, t=t, noise=pred_noise).clamp(-1.0, 1.0)
        model_mean, _ = self.q_posterior(x_start, x, t)
        if return_pred_x_start:
            return model_mean, model_variance, model_log_variance, x_start
        return model_mean, model_variance, model_log_variance
