"""Host-side mirror of `diffusion_model_nemo.models` (the classes the three example directories drive)."""
from .ddpm import AbstractDiffusionModel, DDPM, ImprovedDDPM
from .conditional_ddpm import ConditionalDDPM
from .score_sde import ScoreSDE

__all__ = ["AbstractDiffusionModel", "DDPM", "ImprovedDDPM", "ConditionalDDPM", "ScoreSDE"]
