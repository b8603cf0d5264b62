"""Forward SDE definitions used by the score-based samplers.

Mirror of diffusion_model_nemo/modules/sde_lib/{sde_lib,ve_sde,vp_sde,sub_vp_sde}.py (interfaces from
https://github.com/yang-song/score_sde_pytorch). These classes only carry scalars and closed-form per-time
coefficients; the per-iteration coefficients a sampler needs are evaluated once on the host
(`linear_coefficients`, `discretize_coefficients`) and uploaded as a table, instead of being rebuilt from Python
objects on every step as `SDE.reverse()` does in the reference (sde_lib.py:69-107).
"""
import abc
import math

import numpy as np
import torch


class SDE(abc.ABC):
    sampling_epsilon: float = None

    def __init__(self, N):
        super().__init__()
        self.N = N
        if self.sampling_epsilon is None:
            raise ValueError("Sampling epsilon cannot be None ! Must be set as a class variable !")

    @property
    @abc.abstractmethod
    def T(self):
        ...

    @abc.abstractmethod
    def sde(self, x, t):
        """(drift, diffusion) of the forward SDE at time t."""

    @abc.abstractmethod
    def marginal_prob(self, x, t):
        """(mean, std) of p_t(x | x_0)."""

    @abc.abstractmethod
    def prior_sampling(self, shape):
        ...

    @abc.abstractmethod
    def prior_logp(self, z):
        ...

    @abc.abstractmethod
    def linear_coefficients(self, t):
        """(d, g) with drift(x, t) = d(t) * x and diffusion g(t); t is a 1-D float32 tensor."""

    def discretize_coefficients(self, t):
        """(fc, G) with f(x, t) = fc(t) * x: Euler-Maruyama discretisation x_{i+1} = x_i + f + G z (sde_lib.py:53-67)."""
        dt = 1 / self.N
        d, g = self.linear_coefficients(t)
        return d * dt, g * torch.sqrt(torch.tensor(dt))

    def discretize(self, x, t):
        fc, G = self.discretize_coefficients(t.detach().cpu().float())
        fc, G = fc.to(x.device), G.to(x.device)
        return fc[:, None, None, None] * x, G

    @property
    def prior_scale(self) -> float:
        return 1.0


class VESDE(SDE):
    """Variance-exploding SDE (ve_sde.py:7-59)."""
    sampling_epsilon = 1e-5

    def __init__(self, sigma_min=0.01, sigma_max=50, N=1000):
        super().__init__(N)
        self.sigma_min = sigma_min
        self.sigma_max = sigma_max
        self.discrete_sigmas = torch.exp(torch.linspace(np.log(self.sigma_min), np.log(self.sigma_max), N))
        self.N = N

    @property
    def T(self):
        return 1

    def _sigma(self, t):
        return self.sigma_min * (self.sigma_max / self.sigma_min) ** t

    def linear_coefficients(self, t):
        g = self._sigma(t) * torch.sqrt(torch.tensor(2 * (np.log(self.sigma_max) - np.log(self.sigma_min))))
        return torch.zeros_like(t), g

    def sde(self, x, t):
        sigma = self._sigma(t)
        diffusion = sigma * torch.sqrt(
            torch.tensor(2 * (np.log(self.sigma_max) - np.log(self.sigma_min)), device=t.device))
        return torch.zeros_like(x), diffusion

    def marginal_prob(self, x, t):
        return x, self._sigma(t)

    def prior_sampling(self, shape):
        return torch.randn(*shape) * self.sigma_max

    @property
    def prior_scale(self) -> float:
        return float(self.sigma_max)

    def prior_logp(self, z):
        n = np.prod(z.shape[1:])
        return (-n / 2.0 * np.log(2 * np.pi * self.sigma_max ** 2)
                - torch.sum(z ** 2, dim=(1, 2, 3)) / (2 * self.sigma_max ** 2))

    def sigma_pair(self, t):
        """(sigma_i, sigma_{i-1}) of the SMLD discretisation at time t (ve_sde.py:52-57)."""
        step = (t * (self.N - 1) / self.T).long()
        sigma = self.discrete_sigmas[step]
        adjacent = torch.where(step == 0, torch.zeros_like(t), self.discrete_sigmas[step - 1])
        return sigma, adjacent

    def discretize_coefficients(self, t):
        sigma, adjacent = self.sigma_pair(t)
        return torch.zeros_like(t), torch.sqrt(sigma ** 2 - adjacent ** 2)


class VPSDE(SDE):
    """Variance-preserving SDE (vp_sde.py:11-71)."""
    sampling_epsilon = 1e-3

    def __init__(self, beta_min=0.1, beta_max=20, N=1000):
        super().__init__(N)
        self.beta_0 = beta_min
        self.beta_1 = beta_max
        self.N = N
        self.compute_constants(N)

    def compute_constants(self, timesteps):
        self.betas = torch.linspace(self.beta_0 / timesteps, self.beta_1 / timesteps, timesteps)
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.sqrt_alphas_cumprod = torch.sqrt(self.alphas_cumprod)
        self.sqrt_1m_alphas_cumprod = torch.sqrt(1.0 - self.alphas_cumprod)

    @property
    def T(self):
        return 1

    def _beta_t(self, t):
        return self.beta_0 + t * (self.beta_1 - self.beta_0)

    def _log_mean_coeff(self, t):
        return -0.25 * t ** 2 * (self.beta_1 - self.beta_0) - 0.5 * t * self.beta_0

    def linear_coefficients(self, t):
        beta_t = self._beta_t(t)
        return -0.5 * beta_t, torch.sqrt(beta_t)

    def sde(self, x, t):
        beta_t = self._beta_t(t)
        return -0.5 * beta_t[:, None, None, None] * x, torch.sqrt(beta_t)

    def marginal_prob(self, x, t):
        lmc = self._log_mean_coeff(t)
        return torch.exp(lmc[:, None, None, None]) * x, torch.sqrt(1.0 - torch.exp(2.0 * lmc))

    def prior_sampling(self, shape):
        return torch.randn(*shape)

    def prior_logp(self, z):
        n = np.prod(z.shape[1:])
        return -n / 2.0 * np.log(2 * np.pi) - torch.sum(z ** 2, dim=(1, 2, 3)) / 2.0

    def discretize_coefficients(self, t):
        """DDPM discretisation (vp_sde.py:63-71): f = (sqrt(alpha_i) - 1) x, G = sqrt(beta_i)."""
        step = (t * (self.N - 1) / self.T).long()
        return torch.sqrt(self.alphas[step]) - 1.0, torch.sqrt(self.betas[step])


class subVPSDE(SDE):
    """sub-VP SDE (sub_vp_sde.py:7-45). Its std is 1 - exp(2 log_mean_coeff) (no square root), as shipped."""
    sampling_epsilon = 1e-3

    def __init__(self, beta_min=0.1, beta_max=20, N=1000):
        super().__init__(N)
        self.beta_0 = beta_min
        self.beta_1 = beta_max
        self.N = N

    @property
    def T(self):
        return 1

    def linear_coefficients(self, t):
        beta_t = self.beta_0 + t * (self.beta_1 - self.beta_0)
        discount = 1.0 - torch.exp(-2 * self.beta_0 * t - (self.beta_1 - self.beta_0) * t ** 2)
        return -0.5 * beta_t, torch.sqrt(beta_t * discount)

    def sde(self, x, t):
        d, g = self.linear_coefficients(t)
        return d[:, None, None, None] * x, g

    def marginal_prob(self, x, t):
        lmc = -0.25 * t ** 2 * (self.beta_1 - self.beta_0) - 0.5 * t * self.beta_0
        return torch.exp(lmc)[:, None, None, None] * x, 1 - torch.exp(2.0 * lmc)

    def prior_sampling(self, shape):
        return torch.randn(*shape)

    def prior_logp(self, z):
        n = np.prod(z.shape[1:])
        return -n / 2.0 * np.log(2 * np.pi) - torch.sum(z ** 2, dim=(1, 2, 3)) / 2.0
