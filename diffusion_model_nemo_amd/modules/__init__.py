"""Host-side mirror of `diffusion_model_nemo.modules` for the accelerated hot path."""
from .diffusion_process import (AbstractDiffusionProcess, CosineSchedule, LinearSchedule, QuadraticSchedule,
                                SigmoidSchedule, cosine_beta_schedule, linear_beta_schedule, quadratic_beta_schedule,
                                sigmoid_beta_schedule)
from .gaussian_diffusion import GaussianDiffusion
from .generalized_gaussian_diffusion import GeneralizedGaussianDiffusion
from .learned_gaussian_diffusion import LearnedGaussianDiffusion
from .sde_lib import SDE, VESDE, VPSDE, subVPSDE
from .sde_samplers import PredictorCorrectorSampler, get_corrector, get_predictor
from .unet import Unet

__all__ = [
    "AbstractDiffusionProcess", "CosineSchedule", "LinearSchedule", "QuadraticSchedule", "SigmoidSchedule",
    "cosine_beta_schedule", "linear_beta_schedule", "quadratic_beta_schedule", "sigmoid_beta_schedule",
    "GaussianDiffusion", "GeneralizedGaussianDiffusion", "LearnedGaussianDiffusion", "SDE", "VESDE", "VPSDE", "subVPSDE",
    "PredictorCorrectorSampler", "get_predictor", "get_corrector", "Unet",
]
