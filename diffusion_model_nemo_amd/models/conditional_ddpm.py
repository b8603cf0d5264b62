"""`ConditionalDDPM`: class-conditional sampling and training with the reference's interface
(diffusion_model_nemo/models/conditional_ddpm.py:23-111).

The network adds a learned class-embedding row after the ingest convolution (modules/unet.py:134-141). Row
`num_classes` is the "no class" row: it stands for unconditional generation and, in training, for the labels that
the reference drops at random. Everything that produces labels goes through `resolve_labels`, so the sampler, the
interpolation and the training step agree on defaults and on the drop-out. The samplers receive
`functools.partial(self.forward, classes=labels)` as in the reference; the samplers of this package recognise that
callable and keep the whole loop on the device.
"""
from functools import partial
from typing import Dict, Optional, Union

import torch

from .ddpm import DDPM


class ConditionalDDPM(DDPM):
    def __init__(self, cfg, trainer=None):
        super().__init__(cfg, trainer=trainer)
        n = self.cfg.get('num_classes', None)
        if n is None:   # same exception type and text as conditional_ddpm.py:27-28
            raise RuntimeWarning("Conditional ddpm must have the `num_classes` value inside cfg.model !")
        self.num_classes = n
        self._force_conditioning()

    @property
    def random_class_index(self) -> int:
        """Index of the "no class" embedding row (conditional_ddpm.py:31)."""
        return self.num_classes

    def _force_conditioning(self):
        self.sampler.use_class_conditioning = True

    def change_sampler(self, sampler_cfg):
        super().change_sampler(sampler_cfg)
        self._force_conditioning()

    # ---- labels ---------------------------------------------------------------------------------------------------
    def resolve_labels(self, batch_size: int, device, label: Union[None, int, torch.Tensor] = None,
                       drop: bool = False) -> torch.Tensor:
        """int64 labels [batch_size] on `device`. `label` None: the "no class" row for every image; an int: that class
        for every image (conditional_ddpm.py:78-81); a tensor: used as given. `drop`: every label is replaced by the
        "no class" row with probability 1/2, the classifier-free style drop-out of the training step
        (conditional_ddpm.py:58-61; one `torch.randint(0, 2)` draw of the batch size on the labels' device)."""
        if label is None:
            out = torch.full([batch_size], self.random_class_index, dtype=torch.long, device=device)
        elif isinstance(label, torch.Tensor):
            out = label.to(device=device)
        else:
            out = torch.full([batch_size], int(label), dtype=torch.long, device=device)
        if drop:
            dropped = torch.randint(0, 2, size=[out.size(0)], dtype=torch.bool, device=out.device)
            out = out.masked_fill(dropped, self.random_class_index)
        return out

    def forward(self, x_t: torch.Tensor, t: torch.Tensor, classes: torch.Tensor = None):
        if classes is None and self.sampler.use_class_conditioning:
            classes = self.resolve_labels(x_t.size(0), x_t.device)
        return self.diffusion_model(x_t, t, classes)

    def get_diffusion_model(self, batch: Dict):
        """The callable the samplers and the loss evaluate: `forward` with the batch's labels bound
        (conditional_ddpm.py:51-69); in training mode half of them are dropped first."""
        if not self.sampler.use_class_conditioning:
            return self.forward
        given = batch["label"]
        labels = self.resolve_labels(given.size(0), next(self.parameters()).device, given, drop=self.training)
        return partial(self.forward, classes=labels)

    def _bound_network(self, batch_size: int, label):
        device = next(self.parameters()).device
        return self.get_diffusion_model({'label': self.resolve_labels(batch_size, device, label)}), device

    # ---- generation -----------------------------------------------------------------------------------------------
    def sample(self, batch_size: int, image_size: int, label=None, **kwargs):
        with torch.inference_mode():
            self.eval()
            fn, device = self._bound_network(batch_size, label)
            return self.sampler.sample(fn, shape=[batch_size, self.channels, image_size, image_size], device=device, **kwargs)

    def interpolate(self, x1: torch.Tensor, x2: torch.Tensor, t: Optional[int] = None, lambd: float = 0.5, label=None,
                    **kwargs):
        with torch.inference_mode():
            self.eval()
            for name, x in (("x1", x1), ("x2", x2)):
                assert x.ndim == 4, f"{name} is not a batch of tensors ! Given shape {x.shape}"
            fn, device = self._bound_network(x1.shape[0], label)
            return self.sampler.interpolate(fn, x1=x1.to(device), x2=x2.to(device), t=t, lambd=lambd)
