"""Predictor-corrector sampler for score-SDE models on the gfx950 kernels.

Mirror of diffusion_model_nemo/modules/sde_samplers/predictor_corrector_sampler.py:18-174 together with the
predictor / corrector registries (sde_predictors/*.py, sde_correctors/*.py) and the score adapter
(loss/sde_loss/score_function_loss.py:48-91).

Every predictor of the reference is an affine map of (x, network output) plus scaled noise whose coefficients
depend only on the iteration index, so the whole configuration collapses into one per-iteration table
(`step_table`) evaluated on the host in float32, and the loop runs on the device as a replayed hipGraph:
per iteration  n_steps x {U-Net, Langevin update}  then  {U-Net, predictor update}  (corrector first, as in the
reference, predictor_corrector_sampler.py:105-106).
"""
import math
from typing import List, Optional

import torch

from .. import _native as N
from . import sde_lib
from .gaussian_diffusion import _fresh_seed
from .unet import Unet

PREDICTORS = {"euler_maruyama": 1, "reverse_diffusion": 2, "ancestral_sampling": 3, "none": 0, "null": 0, None: 0}
CORRECTORS = {"langevin": 1, "ald": 2, "none": 0, "null": 0, None: 0}


def get_predictor(name):
    return name if name in PREDICTORS else None


def get_corrector(name):
    return name if name in CORRECTORS else None


def score_labels_and_scale(sde, t: torch.Tensor, continuous: bool):
    """Network time input and the factor turning the network output into a score, per time value
    (score_function_loss.py:48-91): VP/sub-VP: labels = t (N-1), score = -net / std; VE: labels = sigma(t), score = net."""
    if isinstance(sde, (sde_lib.VPSDE, sde_lib.subVPSDE)):
        labels = t * (sde.N - 1)
        if continuous or isinstance(sde, sde_lib.subVPSDE):
            std = sde.marginal_prob(torch.zeros(t.shape[0], 1, 1, 1), t)[1]
        else:
            std = sde.sqrt_1m_alphas_cumprod[labels.long()]
        return labels, -1.0 / std
    if isinstance(sde, sde_lib.VESDE):
        if continuous:
            labels = sde.marginal_prob(None, t)[1]
        else:
            labels = torch.round((sde.T - t) * (sde.N - 1)).long().float()
        return labels, torch.ones_like(t)
    raise NotImplementedError(f"SDE class {sde.__class__.__name__} not yet supported.")


class PredictorCorrectorSampler(torch.nn.Module):
    def __init__(self, predictor: str, corrector: str, snr: float, n_steps: int = 1, probability_flow: bool = False,
                 continuous: bool = True, denoise: bool = True, eps: float = None):
        super().__init__()
        self.predictor = predictor
        self.corrector = corrector
        self.snr = snr
        self.n_steps = n_steps
        self.probability_flow = probability_flow
        self.continuous = continuous
        self.denoise = denoise
        self.eps = eps
        self.sde: Optional[sde_lib.SDE] = None
        self.seed: Optional[int] = None

    def update_sde(self, sde: sde_lib.SDE):
        self.sde = sde
        self.__dict__.pop("_tables", None)

    # ---- per-iteration coefficients ---------------------------------------------------------------------------------
    def _validate(self):
        sde = self.sde
        if self.predictor not in PREDICTORS:
            raise ValueError(f"unknown predictor `{self.predictor}`")
        if self.corrector not in CORRECTORS:
            raise ValueError(f"unknown corrector `{self.corrector}`")
        pk, ck = PREDICTORS[self.predictor], CORRECTORS[self.corrector]
        if pk == 3:
            if not isinstance(sde, (sde_lib.VPSDE, sde_lib.VESDE)):
                raise NotImplementedError(f"SDE class {sde.__class__.__name__} not yet supported.")
            assert not self.probability_flow, "Probability flow not supported by ancestral sampling"
            if isinstance(sde, sde_lib.VPSDE):
                # the reference reads `sde.discrete_betas`, which VPSDE never defines
                # (sde_predictors/ancestral_sampling_predictor.py:32)
                raise AttributeError("'VPSDE' object has no attribute 'discrete_betas'")
        if pk == 1 and self.probability_flow:
            # the reverse SDE hands back the Python float 0.0 as diffusion (sde_lib.py:97) and the Euler-Maruyama
            # update subscripts it (euler_maruyama_predictor.py:16)
            raise TypeError("'float' object is not subscriptable")
        if ck and isinstance(sde, sde_lib.subVPSDE):
            # langevin_corrector.py:20-22 reads `sde.alphas`, which subVPSDE lacks
            raise AttributeError("'subVPSDE' object has no attribute 'alphas'")
        return pk, ck

    def step_table(self, device=None, times: Optional[torch.Tensor] = None) -> torch.Tensor:
        """[N][12] float32 table (layout: csrc/dmn_sampler.h), iteration i at time linspace(T, eps, N)[i]
        (or at the explicit float32 ``times``)."""
        sde = self.sde
        if sde is None:
            raise ValueError("Must explicitly set `update_sde(sde)` for this module prior to calling forward()")
        pk, ck = self._validate()
        eps = sde.sampling_epsilon if self.eps is None else self.eps
        Nn = sde.N
        t = torch.linspace(sde.T, eps, Nn) if times is None else times.detach().cpu().float()
        rows = t.shape[0]
        labels, sscale = score_labels_and_scale(sde, t, self.continuous)
        pf = 0.5 if self.probability_flow else 1.0
        a = torch.ones(rows)
        b = torch.zeros(rows)
        c = torch.zeros(rows)
        if pk == 1:    # Euler-Maruyama on the reverse SDE (euler_maruyama_predictor.py:11-17, sde_lib.py:91-98)
            d, g = sde.linear_coefficients(t)
            dt = -1.0 / Nn
            a = 1.0 + d * dt
            b = -(g ** 2) * pf * dt * sscale
            c = torch.zeros(rows) if self.probability_flow else g * float(math.sqrt(-dt))
        elif pk == 2:  # reverse diffusion (reverse_diffusion_predictor.py:11-16, sde_lib.py:100-105)
            fc, G = sde.discretize_coefficients(t)
            a = 1.0 - fc
            b = (G ** 2) * pf * sscale
            c = torch.zeros(rows) if self.probability_flow else G
        elif pk == 3:  # ancestral sampling, VE (ancestral_sampling_predictor.py:17-28)
            sigma, adj = sde.sigma_pair(t)
            b = (sigma ** 2 - adj ** 2) * sscale
            c = torch.sqrt((adj ** 2 * (sigma ** 2 - adj ** 2)) / (sigma ** 2))
        alpha = torch.ones(rows)
        if ck and isinstance(sde, sde_lib.VPSDE):
            alpha = sde.alphas[(t * (sde.N - 1) / sde.T).long()]
        ald = torch.zeros(rows)
        if ck == 2:    # annealed Langevin: fixed step (annealed_langevin_dynamics_corrector.py:32-37)
            std = sde.marginal_prob(torch.zeros(rows, 1, 1, 1), t)[1]
            ald = (self.snr * std) ** 2 * 2 * alpha
        tab = torch.zeros(rows, N.TABLE_COLS, dtype=torch.float32)
        tab[:, 0], tab[:, 1], tab[:, 2], tab[:, 3] = labels, a, b, c
        tab[:, 4], tab[:, 5], tab[:, 6] = sscale, alpha, ald
        tab[:, 8] = torch.arange(rows).float()
        return tab if device is None else tab.to(device)

    # ---- the loop --------------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def forward(self, model, shape: List[int], device, return_nfe: bool = True, use_tqdm: bool = True,
                keep: str = "all", seed: Optional[int] = None, image_offset: int = 0,
                return_device_tensor: bool = False, use_graph: bool = True, check: bool = True):
        """Returns (imgs, nfe) like the reference: imgs is a list of N CPU tensors ((x_mean or x) + 1) / 2, or a
        one-element list with keep='last'. nfe keeps the reference's formula N * (n_steps + 1).
        ``return_device_tensor=True`` returns the final image on the device; ``check=False`` then skips the wait for
        the chain and leaves ``Unet.check_handoffs()`` to the caller."""
        if self.sde is None:
            raise ValueError("Must explicitly set `update_sde(sde)` for this module prior to calling forward()")
        if keep not in ("all", "last"):
            raise ValueError("keep must be 'all' or 'last'")
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError("PredictorCorrectorSampler runs only on an MI355X device (no CPU implementation)")
        if not isinstance(model, Unet):
            raise TypeError("the device loop needs the native Unet of this package as score network")
        pk, ck = self._validate()
        sde = self.sde
        B, Cc, H, W = (int(s) for s in shape)
        seed = int(seed) if seed is not None else (int(self.seed) if self.seed is not None else _fresh_seed())
        table = self.step_table(device)
        x = torch.empty((B, Cc, H, W), dtype=torch.float32, device=device)
        x_mean = torch.empty_like(x)
        hist = torch.empty((sde.N, B, Cc, H, W), dtype=torch.float32, device=device) if keep == "all" else None
        ws = model.workspace(B, H, W, device, sampling=True)
        handle = model.native_handle()
        cur = torch.cuda.current_stream(device)
        # graph capture needs a non-default stream: the caller's own when it already is one
        if cur != torch.cuda.default_stream(device):
            side = cur
        else:
            side = self.__dict__.setdefault("_streams", {}).setdefault(str(device), torch.cuda.Stream(device))
        side.wait_stream(cur)
        with torch.cuda.device(device), torch.cuda.stream(side):   # the library launches on the current device
            N.check(N.lib().dmn_pc_sample_loop(
                handle, table.data_ptr(), sde.N, x.data_ptr(), B, H, W, 1 if pk else 0, ck, self.n_steps if ck else 0,
                float(self.snr), 1 if self.denoise else 0, 1, float(sde.prior_scale), seed, image_offset,
                None if hist is None else hist.data_ptr(), x_mean.data_ptr(), ws.data_ptr(), ws.numel(),
                side.cuda_stream, 1 if use_graph else 0))
        cur.wait_stream(side)
        if not pk:
            x_mean = x  # NonePredictor returns (x, x): the kept image is the corrector's x
        final = x_mean if self.denoise else x
        nfe = sde.N * (self.n_steps + 1)
        if return_device_tensor:
            out = final
        elif keep == "last" or not (pk or ck):
            out = [((final + 1.0) * 0.5).cpu()] * (1 if keep == "last" else sde.N)
        else:
            out = list(((hist + 1.0) * 0.5).cpu().unbind(0))
        if check or not return_device_tensor:
            model.check_handoffs()   # waits for the chain; a starved in-launch exchange invalidates the images
        return (out, nfe) if return_nfe else out

    def sample(self, model, shape, device=None, return_nfe: bool = False, **kwargs):
        if device is None:
            device = next(model.parameters()).device
        return self.forward(model=model, shape=shape, device=device, return_nfe=return_nfe, **kwargs)
