"""Beta schedules and the base class of the discrete-time samplers.

Host-side mirror of diffusion_model_nemo/modules/diffusion_process.py:8-117. The schedule vectors are tiny
(T floats) and are evaluated once on the host in float32 with the same operation order as the reference, so
the per-step coefficients uploaded to the device are bit-identical to the reference's.
"""
from dataclasses import dataclass
from typing import Dict, Optional

import torch


def cosine_beta_schedule(timesteps, s=0.008, min_clip=0.0001, max_clip=0.999):
    """Improved-DDPM cosine schedule (https://arxiv.org/abs/2102.09672), clipped to [min_clip, max_clip]."""
    grid = torch.linspace(0, timesteps, timesteps + 1)
    f = torch.cos(((grid / timesteps) + s) / (1 + s) * torch.pi * 0.5) ** 2
    f = f / f[0]
    return torch.clip(1 - (f[1:] / f[:-1]), min_clip, max_clip)


def linear_beta_schedule(timesteps, beta_start=0.0001, beta_end=0.02):
    return torch.linspace(beta_start, beta_end, timesteps)


def quadratic_beta_schedule(timesteps, beta_start=0.0001, beta_end=0.02):
    return torch.linspace(beta_start ** 0.5, beta_end ** 0.5, timesteps) ** 2


def sigmoid_beta_schedule(timesteps, beta_start=0.0001, beta_end=0.02):
    return torch.sigmoid(torch.linspace(-6, 6, timesteps)) * (beta_end - beta_start) + beta_start


SCHEDULES = {
    "linear": linear_beta_schedule,
    "quadratic": quadratic_beta_schedule,
    "sigmoid": sigmoid_beta_schedule,
    "cosine": cosine_beta_schedule,
}


@dataclass
class CosineSchedule:
    s: float = 0.008
    min_clip: float = 0.0001
    max_clip: float = 0.999


@dataclass
class LinearSchedule:
    beta_start: float = 0.0001
    beta_end: float = 0.02


@dataclass
class QuadraticSchedule:
    beta_start: float = 0.0001
    beta_end: float = 0.02


@dataclass
class SigmoidSchedule:
    beta_start: float = 0.0001
    beta_end: float = 0.02


class AbstractDiffusionProcess(torch.nn.Module):
    """Common interface of the samplers (diffusion_process.py:39-91)."""
    use_class_conditioning: bool = False

    def __init__(self, timesteps, schedule_name, schedule_cfg: Optional[Dict] = None):
        super().__init__()
        self.timesteps = timesteps
        self.schedule_name = schedule_name
        self.schedule_cfg = schedule_cfg if schedule_cfg is not None else {}
        self.schedule_fn = None

    def compute_constants(self, timesteps):
        raise NotImplementedError()

    def q_mean_variance(self, x_start, t):
        raise NotImplementedError()

    def q_posterior(self, x_start, x, t):
        raise NotImplementedError()

    def q_sample(self, x_start, t, noise=None):
        raise NotImplementedError()

    def p_mean_variance(self, model, x, t, model_output=None):
        raise NotImplementedError()

    def p_sample(self, model, x, t):
        raise NotImplementedError()

    def sample(self, model, shape, device=None):
        raise NotImplementedError()

    def interpolate(self, model, x1, x2, t=None, lambd=0.0):
        raise NotImplementedError()

    def extract(self, a: torch.Tensor, t: torch.Tensor, x_shape):
        """Per-sample gather of a schedule vector, broadcastable against ``x_shape``. The tables are cached on the
        device of ``t`` (the reference gathers on the CPU and copies, one host sync per call)."""
        tab = self._device_table(a, t.device)
        return tab.gather(-1, t.long()).reshape(t.shape[0], *((1,) * (len(x_shape) - 1)))

    def _device_table(self, a: torch.Tensor, device):
        """Device copy of a schedule vector. Only the vectors that are attributes of this object are cached (the cache
        keeps its keys alive); a temporary such as `1 - alphas_cumprod` is copied per call and dropped."""
        if a.device == device:
            return a
        cache = self.__dict__.setdefault("_tab_cache", {})
        key = (id(a), str(device))
        hit = cache.get(key)
        if hit is not None and hit[0] is a:
            return hit[1]
        moved = a.to(device)
        if any(v is a for v in self.__dict__.values()):
            cache[key] = (a, moved)
        return moved

    def forward(self, *args, **kwargs):
        raise RuntimeWarning(f"{self.__class__.__name__} should not be used with forward(), please explicitly call "
                             f"the methods of this module.")
