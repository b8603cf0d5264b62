"""`GeneralizedGaussianDiffusion` (DDIM sampler) with the reference's interface, executed by libdmn_hip.so.

Mirror of diffusion_model_nemo/modules/generalized_gaussian_diffusion.py:19-141: a strided subsequence of the
T training timesteps, x0 predicted from the network output and clamped, then
x_next = sqrt(a_next) x0 + c1 z + c2 model_output with c1 = eta sqrt((1 - a_t / a_next)(1 - a_next) / (1 - a_t)),
c2 = sqrt(1 - a_next - c1^2). The per-step scalars are computed on the host in the reference's float32 operation
order and travel to the device as rows of the step table (layout in csrc/dmn_sampler.h); the loop is the same
graph-replayed loop as ancestral sampling.
"""
from typing import Optional

import torch

from .. import _native as N
from .gaussian_diffusion import GaussianDiffusion


class GeneralizedGaussianDiffusion(GaussianDiffusion):
    _step_flags = 2

    def __init__(self, timesteps: int, schedule_name: str, schedule_cfg: Optional[dict] = None,
                 objective: str = "pred_noise", eta: float = 0.0, ddim_timesteps: int = -1):
        super().__init__(timesteps=timesteps, schedule_name=schedule_name, schedule_cfg=schedule_cfg, objective=objective)
        if not (0.0 <= eta <= 1.0):
            raise ValueError("`eta` must be a value in [0, 1]. 0 = DDIM and 1 = DDPM mode")
        self.eta = eta
        self.ddim_timesteps = ddim_timesteps if ddim_timesteps > 0 else self.timesteps
        self.compute_constants(self.timesteps)

    # ---- the strided schedule (generalized_gaussian_diffusion.py:105-113) ---------------------------------------
    def sequence(self):
        stride = self.timesteps // self.ddim_timesteps
        seq = list(range(0, self.timesteps, stride))
        nxt = [-1] + seq[:-1]
        return list(zip(reversed(seq), reversed(nxt)))

    def step_labels(self):
        return [t for t, _ in self.sequence()]

    def step_table(self, device) -> torch.Tensor:
        """[steps][12] float32 device table in execution order; DDIM row layout of csrc/dmn_sampler.h."""
        cache = self.__dict__.setdefault("_step_tables", {})
        key = ("ddim", str(device), self.eta, self.ddim_timesteps)
        if key not in cache:
            betas_ext = torch.cat([torch.zeros(1), self.betas], dim=0)
            acp_ext = (1.0 - betas_ext).cumprod(dim=0)
            seq = self.sequence()
            tab = torch.zeros(len(seq), N.TABLE_COLS, dtype=torch.float32)
            for r, (t, t_next) in enumerate(seq):
                at, at_next = acp_ext[t + 1], acp_ext[t_next + 1]
                c1 = self.eta * torch.sqrt((1.0 - at / at_next) * (1.0 - at_next) / (1.0 - at))
                c2 = torch.sqrt((1.0 - at_next) - c1 ** 2)
                tab[r, 0] = float(t)
                tab[r, 1] = (1.0 - at).sqrt()
                tab[r, 2] = at.sqrt()
                tab[r, 3] = at_next.sqrt()
                tab[r, 4] = c1
                tab[r, 5] = c2
                tab[r, 8] = float(t)   # noise key: the timestep, like the ancestral sampler
            cache[key] = tab.to(device)
        return cache[key]

    def generalized_predict_start_from_noise(self, x_t, t, noise):
        assert x_t.shape == noise.shape, f"{x_t.shape} != {noise.shape}"
        acp_ext = (1.0 - torch.cat([torch.zeros(1), self.betas], dim=0)).cumprod(dim=0)
        a = self.extract(acp_ext, t + 1, x_t.shape)
        return (x_t - noise * (1.0 - a).sqrt()) / a.sqrt()

    @torch.no_grad()
    def p_sample(self, model, x, t, t_next, noise: Optional[torch.Tensor] = None, seed: Optional[int] = None,
                 image_offset: int = 0):
        """One generalized step (generalized_gaussian_diffusion.py:78-97): returns (x_next, x0_t). All samples must
        share t and t_next, and (t, t_next) must be a step of this sampler's sequence."""
        if not x.is_cuda:
            raise RuntimeError("p_sample runs only on MI355X device tensors (no CPU implementation)")
        tv, tn = int(t.reshape(-1)[0].item()), int(t_next.reshape(-1)[0].item())
        if not bool((t == tv).all().item()) or not bool((t_next == tn).all().item()):
            raise ValueError("the generalized sampler steps every sample from the same t to the same t_next")
        seq = self.sequence()
        if (tv, tn) not in seq:
            raise ValueError(f"({tv}, {tn}) is not a step of the {self.ddim_timesteps}-step sequence")
        row = seq.index((tv, tn))
        B, Cc, H, W = x.shape
        x = x.contiguous().float()
        net = model(x, t).contiguous().float()
        out = torch.empty_like(x)
        if noise is not None:
            noise = noise.to(x.device, torch.float32).contiguous()
        pred_x0 = (0 if self.objective == 'pred_noise' else 1) | self._step_flags
        with torch.cuda.device(x.device):
            N.check(N.lib().dmn_ddpm_step(self.step_table(x.device).data_ptr(), row, x.data_ptr(), net.data_ptr(),
                                          None if noise is None else noise.data_ptr(), out.data_ptr(), B, Cc, H, W, 0,
                                          pred_x0, self._resolve_seed(seed), image_offset, N.current_stream_handle(x.device)))
        if self.objective == 'pred_noise':
            x0 = self.generalized_predict_start_from_noise(x, t, net).clamp(-1.0, 1.0)
        else:
            x0 = net.clamp(-1.0, 1.0)
        return out, x0

    @torch.no_grad()
    def p_sample_loop(self, model, shape, use_tqdm=True, img=None, device=None, keep: str = "all",
                      seed: Optional[int] = None, image_offset: int = 0, return_device_tensor: bool = False,
                      use_graph: bool = True):
        """generalized_gaussian_diffusion.py:100-137: list of CPU images in [0, 1], one per executed step."""
        if keep not in ("all", "last"):
            raise ValueError("keep must be 'all' or 'last'")
        if device is None:
            from .gaussian_diffusion import _unwrap_conditional
            device = next(_unwrap_conditional(model)[0].parameters()).device
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError("p_sample_loop runs only on an MI355X device (no CPU implementation)")
        seed = self._resolve_seed(seed)
        if img is not None:
            x = img.to(device, torch.float32).contiguous().clone()
        else:
            x = torch.empty([int(s) for s in shape], dtype=torch.float32, device=device)
        nsteps = len(self.sequence())
        # the x_T draw is keyed by T (as in the ancestral sampler), the per-step draws by their timestep
        if img is None:
            with torch.cuda.device(device):
                N.check(N.lib().dmn_fill_normal(x.data_ptr(), x.shape[0], x[0].numel(), seed, self.timesteps, image_offset,
                                                1.0, N.current_stream_handle(device)))
        hist = self._run_chain(model, x, 0, nsteps, False, keep, seed, image_offset, use_graph=use_graph)
        if return_device_tensor:
            return x
        if keep == "last":
            return [((x + 1) * 0.5).cpu()]
        return list(((hist + 1) * 0.5).cpu().unbind(0))

    @torch.no_grad()
    def sample(self, model, shape, device=None, **kwargs):
        return self.p_sample_loop(model, shape=shape, device=device, **kwargs)

    @torch.no_grad()
    def interpolate(self, model, x, t: Optional[int] = None, **kwargs):
        """generalized_gaussian_diffusion.py:140-141: denoise the given tensor with the whole sequence."""
        return self.p_sample_loop(model, x.shape, img=x, **kwargs)
