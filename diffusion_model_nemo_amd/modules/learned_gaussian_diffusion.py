"""Improved-DDPM sampler: the network predicts, per pixel, where between the two analytic extremes the reverse
variance lies (diffusion_model_nemo/modules/learned_gaussian_diffusion.py:14-53).

The network emits 2*C channels: the noise estimate and a raw interpolation value v in [-1, 1]. With w = (v + 1) / 2,
log sigma^2 = w * log beta_t + (1 - w) * log beta~_t (Nichol & Dhariwal, eq. 15). The sampling loop is the inherited
device loop with the `learned_variance` switch of the update kernel; this module only has to provide the host-side
`p_mean_variance` that the step-by-step API and the variational-bound terms use. As in the reference, `objective`
does not change the parameterisation: the first half of the output is always read as predicted noise.
"""
from typing import Optional, Tuple

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

    def interpolated_variance(self, v: torch.Tensor, t: torch.Tensor, shape) -> Tuple[torch.Tensor, torch.Tensor]:
        """(sigma^2, log sigma^2) for the raw interpolation values `v` at steps `t`
        (learned_gaussian_diffusion.py:38-43): the bounds are log beta~_t (clipped posterior) and log beta_t."""
        lower, upper = (self.extract(table, t, shape) for table in (self.posterior_log_variance_clipped, self.betas.log()))
        w = (v + 1) * 0.5
        log_variance = w * upper + (1 - w) * lower
        return log_variance.exp(), log_variance

    def p_mean_variance(self, model, x, t, model_output=None, return_pred_x_start=False):
        out = model(x, t) if model_output is None else model_output
        eps, v = torch.split(out, out.shape[1] // 2, dim=1)
        variance, log_variance = self.interpolated_variance(v, t, x.shape)
        x0 = self.predict_start_from_noise(x_t=x, t=t, noise=eps).clamp(-1.0, 1.0)
        mean = self.q_posterior(x0, x, t)[0]
        result = (mean, variance, log_variance)
        return result + (x0,) if return_pred_x_start else result
