"""`ConditionalDDPM`: class-conditional sampling with the reference's interface
(mirror of diffusion_model_nemo/models/conditional_ddpm.py:23-111).

The network adds a learned class-embedding row after the ingest convolution (modules/unet.py:134-141); row
`num_classes` is the "no class" row used for unconditional generation and, during the reference's training, for the
randomly dropped labels. Sampling hands the sampler `functools.partial(self.forward, classes=label)`, exactly like
the reference; the samplers of this package recognise that callable and keep the loop on the device.
"""
from functools import partial
from typing import Dict, Optional

import torch

from .ddpm import DDPM


class ConditionalDDPM(DDPM):
    def __init__(self, cfg, trainer=None):
        super().__init__(cfg, trainer=trainer)
        if 'num_classes' not in self.cfg or self.cfg.num_classes is None:
            raise RuntimeWarning("Conditional ddpm must have the `num_classes` value inside cfg.model !")
        self.num_classes = self.cfg.num_classes
        self.random_class_index = self.num_classes
        self.sampler.use_class_conditioning = True  # force set

    def forward(self, x_t: torch.Tensor, t: torch.Tensor, classes: torch.Tensor = None):
        if self.sampler.use_class_conditioning:
            if classes is None:
                classes = torch.ones(x_t.size(0), device=x_t.device, dtype=torch.long) * self.random_class_index
        return self.diffusion_model(x_t, t, classes)

    def get_diffusion_model(self, batch: Dict):
        if self.sampler.use_class_conditioning:
            device = next(self.parameters()).device
            label = batch["label"].to(device=device)
            if self.training:
                # a random half of the labels becomes the "no class" index (conditional_ddpm.py:58-61)
                sample_mask = torch.randint(0, 2, size=[label.size(0)], dtype=torch.bool, device=label.device)
                label = torch.masked_fill(label, sample_mask, value=self.random_class_index)
            return partial(self.forward, classes=label)
        return self.forward

    def change_sampler(self, sampler_cfg):
        super().change_sampler(sampler_cfg)
        self.sampler.use_class_conditioning = True

    def _labels(self, batch_size: int, label, device):
        if label is None:
            return torch.ones(batch_size, dtype=torch.long, device=device) * self.random_class_index
        return torch.full([batch_size], fill_value=int(label), dtype=torch.long, device=device)

    def sample(self, batch_size: int, image_size: int, label=None, **kwargs):
        with torch.inference_mode():
            self.eval()
            device = next(self.parameters()).device
            shape = [batch_size, self.channels, image_size, image_size]
            fn = self.get_diffusion_model({'label': self._labels(batch_size, label, device)})
            return self.sampler.sample(fn, shape=shape, device=device, **kwargs)

    def interpolate(self, x1: torch.Tensor, x2: torch.Tensor, t: Optional[int] = None, lambd: float = 0.5, label=None,
                    **kwargs):
        with torch.inference_mode():
            self.eval()
            device = next(self.parameters()).device
            assert x1.ndim == 4, f"x1 is not a batch of tensors ! Given shape {x1.shape}"
            assert x2.ndim == 4, f"x2 is not a batch of tensors ! Given shape {x2.shape}"
            fn = self.get_diffusion_model({'label': self._labels(x1.shape[0], label, device)})
            return self.sampler.interpolate(fn, x1=x1.to(device), x2=x2.to(device), t=t, lambd=lambd)
