"""`ScoreSDE` model wrapper. Mirror of diffusion_model_nemo/models/score_sde.py:26-55,130-153 for sampling."""
from typing import Optional

import torch

from ..config import instantiate
from .ddpm import AbstractDiffusionModel


class ScoreSDE(AbstractDiffusionModel):
    def __init__(self, cfg, trainer=None):
        super().__init__(cfg, trainer=trainer)
        self.continuous = self.cfg.get('continuous', True)
        self.likelihood_weighting = self.cfg.get('likelihood_weighting', False)
        self.diffusion_model = instantiate(self.cfg.diffusion_model)
        sde_type = self.cfg.sde.get('sde_type').lower()
        self.sde = instantiate(self.cfg.sde.get(sde_type))
        self.sampler = instantiate(self.cfg.sampler)
        self.sampler.update_sde(self.sde)

    def forward(self, x_t: torch.Tensor, t: torch.Tensor, classes: torch.Tensor = None):
        return self.diffusion_model(x_t, t)

    def sample(self, batch_size: int, image_size: int, device: torch.device = None, **kwargs):
        with torch.inference_mode():
            self.eval()
            if device is None:
                device = next(self.parameters()).device
            shape = [batch_size, self.channels, image_size, image_size]
            return self.sampler.sample(self.diffusion_model, shape=shape, device=device, **kwargs)

    def interpolate(self, x1, x2, t: Optional[int] = None, lambd: float = 0.5, **kwargs):
        raise NotImplementedError()

    def change_sampler(self, sampler_cfg):
        self.sampler = instantiate(sampler_cfg)
        self.sampler.update_sde(self.sde)
        self.cfg.sampler = sampler_cfg
