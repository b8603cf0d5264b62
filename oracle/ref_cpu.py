"""CPU oracle: a plain-PyTorch fp32 restatement of the reference's denoising hot path.

TEST INFRASTRUCTURE ONLY. Nothing under ``diffusion_model_nemo_amd/`` imports this module; it
is used by ``tests/``, by ``__graft_entry__.smoke()`` as the checker and by ``bench.py`` for the
reported CPU baseline. It is pinned to the reference by the golden vectors in
``tests/golden/`` (captured from the reference itself by ``tools/gen_golden.py``).

Everything is functional: the U-Net takes a ``{name: tensor}`` mapping with the reference's
``state_dict`` names. All citations are paths relative to the reference repository root.
"""
import math
from typing import Callable, Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor
Params = Dict[str, Tensor]


# ----------------------------------------------------------------------------------------------
# U-Net (diffusion_model_nemo/modules/unet.py:131-168)
# ----------------------------------------------------------------------------------------------
def sinusoidal_embedding(time: Tensor, dim: int) -> Tensor:
    """diffusion_model_nemo/parts/positional_encoding.py:11-18. ``time`` may be long or float."""
    half = dim // 2
    k = math.log(10000) / (half - 1)
    freqs = torch.exp(torch.arange(half, device=time.device) * -k)
    e = time[:, None] * freqs[None, :]
    return torch.cat((e.sin(), e.cos()), dim=-1)


def time_mlp(p: Params, time: Tensor, dim: int) -> Tensor:
    """diffusion_model_nemo/modules/unet.py:61-66 (Linear -> exact-erf GELU -> Linear)."""
    e = sinusoidal_embedding(time, dim)
    e = F.linear(e, p["time_mlp.1.weight"], p["time_mlp.1.bias"])
    e = F.gelu(e)
    return F.linear(e, p["time_mlp.3.weight"], p["time_mlp.3.bias"])


def block(p: Params, pre: str, x: Tensor, groups: int) -> Tensor:
    """conv3x3 -> GroupNorm -> SiLU; both block orders dispatch here
    (diffusion_model_nemo/parts/convnext.py:25-45)."""
    x = F.conv2d(x, p[pre + ".proj.weight"], p[pre + ".proj.bias"], padding=1)
    x = F.group_norm(x, groups, p[pre + ".norm.weight"], p[pre + ".norm.bias"], eps=1e-5)
    return F.silu(x)


def resnet_block(p: Params, pre: str, x: Tensor, temb: Optional[Tensor], groups: int) -> Tensor:
    """diffusion_model_nemo/parts/convnext.py:78-86: time bias is added after block1's SiLU."""
    h = block(p, pre + ".block1", x, groups)
    if temb is not None and (pre + ".mlp.1.weight") in p:
        tb = F.linear(F.silu(temb), p[pre + ".mlp.1.weight"], p[pre + ".mlp.1.bias"])
        h = tb[:, :, None, None] + h
    h = block(p, pre + ".block2", h, groups)
    if (pre + ".res_conv.weight") in p:
        x = F.conv2d(x, p[pre + ".res_conv.weight"], p[pre + ".res_conv.bias"])
    return h + x


def linear_attention(p: Params, pre: str, x: Tensor, heads: int = 4) -> Tensor:
    """diffusion_model_nemo/parts/mha.py:44-59 (without the PreNorm / Residual wrappers)."""
    b, c, h, w = x.shape
    qkv = F.conv2d(x, p[pre + ".to_qkv.weight"])
    hidden = qkv.shape[1] // 3
    d = hidden // heads
    q, k, v = (t.reshape(b, heads, d, h * w) for t in qkv.chunk(3, dim=1))
    q = q.softmax(dim=-2) * (d ** -0.5)
    k = k.softmax(dim=-1)
    context = torch.einsum("bhdn,bhen->bhde", k, v)
    out = torch.einsum("bhde,bhdn->bhen", context, q)
    out = out.reshape(b, hidden, h, w)
    out = F.conv2d(out, p[pre + ".to_out.0.weight"], p[pre + ".to_out.0.bias"])
    return F.group_norm(out, 1, p[pre + ".to_out.1.weight"], p[pre + ".to_out.1.bias"], eps=1e-5)


def attention(p: Params, pre: str, x: Tensor, heads: int = 4) -> Tensor:
    """diffusion_model_nemo/parts/mha.py:16-30."""
    b, c, h, w = x.shape
    qkv = F.conv2d(x, p[pre + ".to_qkv.weight"])
    hidden = qkv.shape[1] // 3
    d = hidden // heads
    q, k, v = (t.reshape(b, heads, d, h * w) for t in qkv.chunk(3, dim=1))
    q = q * (d ** -0.5)
    sim = torch.einsum("bhdi,bhdj->bhij", q, k)
    sim = sim - sim.amax(dim=-1, keepdim=True)
    attn = sim.softmax(dim=-1)
    out = torch.einsum("bhij,bhdj->bhid", attn, v)
    out = out.permute(0, 1, 3, 2).reshape(b, hidden, h, w)
    return F.conv2d(out, p[pre + ".to_out.weight"], p[pre + ".to_out.bias"])


def prenorm_residual(p: Params, pre: str, x: Tensor, fn: Callable) -> Tensor:
    """Residual(PreNorm(dim, fn)) (diffusion_model_nemo/utils.py:68-93): fn(GroupNorm(1)(x)) + x."""
    xn = F.group_norm(x, 1, p[pre + ".fn.norm.weight"], p[pre + ".fn.norm.bias"], eps=1e-5)
    return fn(p, pre + ".fn.fn", xn) + x


def unet_forward(p: Params, x: Tensor, time: Tensor, *, dim: int = 32, groups: int = 8,
                 classes: Optional[Tensor] = None) -> Tensor:
    """diffusion_model_nemo/modules/unet.py:131-168 for ``use_convnext=False``.

    The number of levels, the learned-variance head and the tail variant are read off the parameter
    names. The top-level skip is pushed but never popped (the ``ups`` list has one level fewer than
    ``downs``), and every up level ends with a transposed convolution (unet.py:95-104)."""
    x = F.conv2d(x, p["init_conv.weight"], p["init_conv.bias"], padding=3)
    if "class_embed.weight" in p:
        num_classes = p["class_embed.weight"].shape[0] - 1
        if classes is None:
            classes = torch.full((x.size(0),), num_classes, dtype=torch.long, device=x.device)
        x = x + F.embedding(classes, p["class_embed.weight"])[:, :, None, None]
    t = time_mlp(p, time, dim) if "time_mlp.1.weight" in p else None

    n_down = 0
    while f"downs.{n_down}.0.block1.proj.weight" in p:
        n_down += 1
    hs = []
    for i in range(n_down):
        x = resnet_block(p, f"downs.{i}.0", x, t, groups)
        x = resnet_block(p, f"downs.{i}.1", x, t, groups)
        x = prenorm_residual(p, f"downs.{i}.2", x, linear_attention)
        hs.append(x)
        if f"downs.{i}.3.weight" in p:
            x = F.conv2d(x, p[f"downs.{i}.3.weight"], p[f"downs.{i}.3.bias"], stride=2, padding=1)

    x = resnet_block(p, "mid_block1", x, t, groups)
    x = prenorm_residual(p, "mid_attn", x, attention)
    x = resnet_block(p, "mid_block2", x, t, groups)

    i = 0
    while f"ups.{i}.0.block1.proj.weight" in p:
        x = torch.cat((x, hs.pop()), dim=1)
        x = resnet_block(p, f"ups.{i}.0", x, t, groups)
        x = resnet_block(p, f"ups.{i}.1", x, t, groups)
        x = prenorm_residual(p, f"ups.{i}.2", x, linear_attention)
        if f"ups.{i}.3.weight" in p:
            x = F.conv_transpose2d(x, p[f"ups.{i}.3.weight"], p[f"ups.{i}.3.bias"], stride=2, padding=1)
        i += 1

    x = resnet_block(p, "final_conv.0", x, None, groups)
    if "final_conv.3.weight" in p:  # resnet_block_order == 'bn_act_conv' (unet.py:112-113)
        x = F.group_norm(x, groups, p["final_conv.1.weight"], p["final_conv.1.bias"], eps=1e-5)
        x = F.silu(x)
        return F.conv2d(x, p["final_conv.3.weight"], p["final_conv.3.bias"])
    return F.conv2d(x, p["final_conv.1.weight"], p["final_conv.1.bias"])


# ----------------------------------------------------------------------------------------------
# Beta schedules and DDPM constants
# (diffusion_model_nemo/modules/diffusion_process.py:8-36, gaussian_diffusion.py:44-83)
# ----------------------------------------------------------------------------------------------
def beta_schedule(name: str, timesteps: int, **cfg) -> Tensor:
    if name == "cosine":
        s = cfg.get("s", 0.008)
        min_clip = cfg.get("min_clip", 0.0001)
        max_clip = cfg.get("max_clip", 0.999)
        x = torch.linspace(0, timesteps, timesteps + 1)
        ac = torch.cos(((x / timesteps) + s) / (1 + s) * torch.pi * 0.5) ** 2
        ac = ac / ac[0]
        betas = 1 - (ac[1:] / ac[:-1])
        return torch.clip(betas, min_clip, max_clip)
    b0 = cfg.get("beta_start", 0.0001)
    b1 = cfg.get("beta_end", 0.02)
    if name == "linear":
        return torch.linspace(b0, b1, timesteps)
    if name == "quadratic":
        return torch.linspace(b0 ** 0.5, b1 ** 0.5, timesteps) ** 2
    if name == "sigmoid":
        return torch.sigmoid(torch.linspace(-6, 6, timesteps)) * (b1 - b0) + b0
    raise ValueError(f"Invalid schedule `{name}`")


CONSTANT_NAMES = (
    "betas", "alphas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_recip_alphas",
    "sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod", "sqrt_recip_alphas_cumprod",
    "sqrt_recipm1_alphas_cumprod", "log_one_minus_alphas_cumprod", "posterior_variance",
    "posterior_log_variance_clipped", "posterior_mean_coef1", "posterior_mean_coef2",
)


def ddpm_constants(timesteps: int, schedule_name: str = "cosine", **cfg) -> Dict[str, Tensor]:
    """gaussian_diffusion.py:44-83, same operation order in fp32."""
    c = {}
    betas = beta_schedule(schedule_name, timesteps, **cfg)
    c["betas"] = betas
    c["alphas"] = 1.0 - betas
    c["alphas_cumprod"] = torch.cumprod(c["alphas"], dim=0)
    c["alphas_cumprod_prev"] = F.pad(c["alphas_cumprod"][:-1], (1, 0), value=1.0)
    c["sqrt_recip_alphas"] = torch.sqrt(1.0 / c["alphas"])
    c["sqrt_alphas_cumprod"] = torch.sqrt(c["alphas_cumprod"])
    c["sqrt_one_minus_alphas_cumprod"] = torch.sqrt(1.0 - c["alphas_cumprod"])
    c["sqrt_recip_alphas_cumprod"] = torch.sqrt(1.0 / c["alphas_cumprod"])
    c["sqrt_recipm1_alphas_cumprod"] = torch.sqrt(1.0 / c["alphas_cumprod"] - 1)
    c["log_one_minus_alphas_cumprod"] = torch.log(1.0 - c["alphas_cumprod"])
    c["posterior_variance"] = betas * (1.0 - c["alphas_cumprod_prev"]) / (1.0 - c["alphas_cumprod"])
    pv = c["posterior_variance"]
    c["posterior_log_variance_clipped"] = torch.log(torch.cat([pv[1].unsqueeze(0), pv[1:]]))
    c["posterior_mean_coef1"] = betas * torch.sqrt(c["alphas_cumprod_prev"]) / (1.0 - c["alphas_cumprod"])
    c["posterior_mean_coef2"] = (
        (1.0 - c["alphas_cumprod_prev"]) * torch.sqrt(c["alphas"]) / (1.0 - c["alphas_cumprod"])
    )
    return c


def _ex(a: Tensor, t: Tensor, ndim: int) -> Tensor:
    """diffusion_process.py:84-87."""
    return a.gather(-1, t.cpu()).reshape(t.shape[0], *((1,) * (ndim - 1)))


def q_sample(c, x_start: Tensor, t: Tensor, noise: Tensor) -> Tensor:
    """gaussian_diffusion.py:104-116."""
    return (_ex(c["sqrt_alphas_cumprod"], t, x_start.ndim) * x_start
            + _ex(c["sqrt_one_minus_alphas_cumprod"], t, x_start.ndim) * noise)


def p_sample_update(c, x: Tensor, t: Tensor, model_output: Tensor, noise: Tensor,
                    learned_variance: bool = False, objective: str = "pred_noise") -> Tensor:
    """One ancestral step given the network output and the noise draw.

    Fixed variance: gaussian_diffusion.py:118-123,137-154,157-167.
    Learned variance: learned_gaussian_diffusion.py:35-48 feeding the same p_sample."""
    nd = x.ndim
    if learned_variance:
        eps, v = model_output.chunk(2, dim=1)
        min_log = _ex(c["posterior_log_variance_clipped"], t, nd)
        max_log = _ex(torch.log(c["betas"]), t, nd)
        frac = (v + 1) * 0.5
        log_var = frac * max_log + (1 - frac) * min_log
        x0 = _ex(c["sqrt_recip_alphas_cumprod"], t, nd) * x - _ex(c["sqrt_recipm1_alphas_cumprod"], t, nd) * eps
    else:
        if objective == "pred_noise":
            x0 = (_ex(c["sqrt_recip_alphas_cumprod"], t, nd) * x
                  - _ex(c["sqrt_recipm1_alphas_cumprod"], t, nd) * model_output)
        else:
            x0 = model_output.clone()
        log_var = _ex(c["posterior_log_variance_clipped"], t, nd)
    x0 = x0.clamp(-1.0, 1.0)
    mean = _ex(c["posterior_mean_coef1"], t, nd) * x0 + _ex(c["posterior_mean_coef2"], t, nd) * x
    mask = (1 - (t == 0).float()).reshape(x.shape[0], *((1,) * (nd - 1)))
    return mean + mask * torch.exp(0.5 * log_var) * noise


def ddpm_sample_chain(model_fn: Callable[[Tensor, Tensor], Tensor], c, shape, noise_fn,
                      learned_variance: bool = False, keep=None, timesteps: Optional[int] = None):
    """gaussian_diffusion.py:171-189 with injected noise.

    ``noise_fn(step)`` returns the N(0,1) draw of that step; ``step == T`` is the initial x_T draw and
    ``step == i`` the draw consumed by ``p_sample`` at timestep i. Returns {timestep: x after that step}
    for timesteps in ``keep`` (default: only 0) plus key ``"x_T"``; values are in model range, i.e. before
    the reference's ``(x + 1) / 2`` display mapping."""
    T = int(c["betas"].shape[0]) if timesteps is None else timesteps
    keep = {0} if keep is None else set(keep)
    x = noise_fn(T)
    out = {"x_T": x.clone()}
    for i in reversed(range(T)):
        t = torch.full((shape[0],), i, dtype=torch.long)
        eps = model_fn(x, t)
        x = p_sample_update(c, x, t, eps, noise_fn(i), learned_variance=learned_variance)
        if i in keep:
            out[i] = x.clone()
    return out


# ----------------------------------------------------------------------------------------------
# Variational bound / bits per dimension (diffusion_model_nemo/utils.py:24-56,
# loss/variational_bound_loss.py:31-52, models/abstract_diffusion_model.py:137-197)
# ----------------------------------------------------------------------------------------------
def normal_kl(mean1, logvar1, mean2, logvar2):
    """utils.py:28-34."""
    return 0.5 * (-1.0 + logvar2 - logvar1 + torch.exp(logvar1 - logvar2) + ((mean1 - mean2) ** 2) * torch.exp(-logvar2))


def discretized_gaussian_log_likelihood(x, means, log_scales, thres=0.999):
    """utils.py:37-56 (tanh approximation of the normal CDF, bins of width 2/255)."""
    def cdf(v):
        return 0.5 * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (v + 0.044715 * (v ** 3))))

    def log(v):
        return torch.log(v.clamp(min=1e-12))

    centered = x - means
    inv_stdv = torch.exp(-log_scales)
    cdf_plus = cdf(inv_stdv * (centered + 1.0 / 255.0))
    cdf_min = cdf(inv_stdv * (centered - 1.0 / 255.0))
    return torch.where(x < -thres, log(cdf_plus), torch.where(x > thres, log(1.0 - cdf_min), log(cdf_plus - cdf_min)))


def vlb_terms(samples, model_mean, model_log_variance, true_mean, true_log_variance_clipped, t):
    """VariationalBoundLoss.compute_variation_loss_terms (variational_bound_loss.py:31-52): per-sample KL in bits,
    replaced by the discretised decoder NLL at t == 0."""
    def mean_flat(v):
        return v.mean(dim=tuple(range(1, v.ndim)))

    kl = mean_flat(normal_kl(true_mean, true_log_variance_clipped, model_mean, model_log_variance)) * (1.0 / math.log(2.0))
    nll = -discretized_gaussian_log_likelihood(samples, model_mean, 0.5 * model_log_variance)
    nll = mean_flat(nll) * (1.0 / math.log(2.0))
    return torch.where(t == 0, nll, kl)


def bits_per_dimension(model_fn, c, x_start: Tensor, noise_fn, learned_variance: bool = False):
    """AbstractDiffusionModel.calculate_bits_per_dimension (abstract_diffusion_model.py:137-197) for the
    'pred_noise' objective. ``noise_fn(t)`` is the q_sample draw of timestep t. Returns total_bpd [B],
    terms_bpd [B, T], prior_bpd [B]."""
    T = int(c["betas"].shape[0])
    B, nd = x_start.shape[0], x_start.ndim
    terms = torch.zeros(B, T)
    for ti in range(T - 1, -1, -1):
        t = torch.full((B,), ti, dtype=torch.long)
        x_t = q_sample(c, x_start, t, noise_fn(ti))
        true_mean = _ex(c["posterior_mean_coef1"], t, nd) * x_start + _ex(c["posterior_mean_coef2"], t, nd) * x_t
        true_lv = _ex(c["posterior_log_variance_clipped"], t, nd)
        out = model_fn(x_t, t)
        if learned_variance:
            eps, v = out.chunk(2, dim=1)
            frac = (v + 1) * 0.5
            model_lv = frac * _ex(torch.log(c["betas"]), t, nd) + (1 - frac) * true_lv
        else:
            eps, model_lv = out, true_lv.expand(-1, *x_start.shape[1:])
        x0 = (_ex(c["sqrt_recip_alphas_cumprod"], t, nd) * x_t - _ex(c["sqrt_recipm1_alphas_cumprod"], t, nd) * eps).clamp(-1.0, 1.0)
        model_mean = _ex(c["posterior_mean_coef1"], t, nd) * x0 + _ex(c["posterior_mean_coef2"], t, nd) * x_t
        terms[:, ti] = vlb_terms(x_start, model_mean, model_lv, true_mean, true_lv, t)
    tp = torch.full((B,), T - 1, dtype=torch.long)
    qt_mean = _ex(c["sqrt_alphas_cumprod"], tp, nd) * x_start
    qt_lv = _ex(c["log_one_minus_alphas_cumprod"], tp, nd)
    zero = torch.tensor(0.0)
    kl_prior = normal_kl(qt_mean, qt_lv, zero, zero)
    prior = kl_prior.mean(dim=tuple(range(1, nd))) / math.log(2.0)
    return {"total_bpd": terms.sum(dim=1) + prior, "terms_bpd": terms, "prior_bpd": prior}


def ddpm_interpolate(model_fn, c, x1: Tensor, x2: Tensor, t: int, lambd: float, q_noise, noise_fn):
    """GaussianDiffusion.interpolate (gaussian_diffusion.py:195-218): both images diffused to step t with the draws
    ``q_noise = (n1, n2)``, mixed (1 - lambd) / lambd, then ancestral steps t-1 .. 0 with ``noise_fn(i)``. Returns the
    list of x after every step, in model range."""
    T = int(c["betas"].shape[0])
    if t >= T:
        raise ValueError(f"`t` must be < {T} during interpolation")
    B = x1.shape[0]
    tb = torch.full((B,), t, dtype=torch.long)
    img = (1 - lambd) * q_sample(c, x1, tb, q_noise[0]) + lambd * q_sample(c, x2, tb, q_noise[1])
    out = []
    for i in reversed(range(t)):
        ti = torch.full((B,), i, dtype=torch.long)
        img = p_sample_update(c, img, ti, model_fn(img, ti), noise_fn(i))
        out.append(img.clone())
    return out


def ddim_sequence(timesteps: int, ddim_timesteps: int):
    """(t, t_next) pairs in execution order (generalized_gaussian_diffusion.py:111-113,121-123)."""
    n = ddim_timesteps if ddim_timesteps > 0 else timesteps
    stride = timesteps // n
    seq = list(range(0, timesteps, stride))
    nxt = [-1] + seq[:-1]
    return list(zip(reversed(seq), reversed(nxt)))


def ddim_step_coefs(betas: Tensor, t: int, t_next: int, eta: float) -> Dict[str, Tensor]:
    """Scalars of one DDIM step in the reference's fp32 operation order
    (generalized_gaussian_diffusion.py:43-46,84-94,105-108)."""
    acp_ext = torch.cumprod(1.0 - torch.cat([torch.zeros(1), betas]), dim=0)
    at, at_next = acp_ext[t + 1], acp_ext[t_next + 1]
    c1 = eta * torch.sqrt((1.0 - at / at_next) * (1.0 - at_next) / (1.0 - at))
    c2 = torch.sqrt((1.0 - at_next) - c1 ** 2)
    return {"sqrt_one_minus_at": (1.0 - at).sqrt(), "sqrt_at": at.sqrt(), "sqrt_at_next": at_next.sqrt(), "c1": c1, "c2": c2}


def ddim_sample_update(k: Dict[str, Tensor], x: Tensor, model_output: Tensor, noise: Tensor,
                       objective: str = "pred_noise") -> Tensor:
    """generalized_gaussian_diffusion.py:48-97. With objective 'pred_x0' the reference clamps the network output
    in place, so the 'direction' term also uses the clamped tensor."""
    if objective == "pred_noise":
        x0 = ((x - model_output * k["sqrt_one_minus_at"]) / k["sqrt_at"]).clamp(-1.0, 1.0)
    else:
        x0 = model_output = model_output.clamp(-1.0, 1.0)
    return k["sqrt_at_next"] * x0 + k["c1"] * noise + k["c2"] * model_output


def ddim_sample_chain(model_fn: Callable[[Tensor, Tensor], Tensor], betas: Tensor, shape, noise_fn, eta: float = 0.0,
                      ddim_timesteps: int = -1, objective: str = "pred_noise", img: Tensor = None):
    """generalized_gaussian_diffusion.py:100-137 with injected noise: ``noise_fn(T)`` is the x_T draw and
    ``noise_fn(t)`` the draw of the step at timestep t (drawn even when eta == 0). Returns the list of x after
    every step, in model range."""
    T = int(betas.shape[0])
    x = noise_fn(T) if img is None else img   # `img`: GeneralizedGaussianDiffusion.interpolate (:138-140), no x_T draw
    out = []
    for t, t_next in ddim_sequence(T, ddim_timesteps):
        tt = torch.full((shape[0],), t, dtype=torch.long)
        x = ddim_sample_update(ddim_step_coefs(betas, t, t_next, eta), x, model_fn(x, tt), noise_fn(t), objective)
        out.append(x.clone())
    return out


# ----------------------------------------------------------------------------------------------
# Score-SDE (diffusion_model_nemo/modules/sde_lib/*.py, sde_predictors/*.py, sde_correctors/*.py,
# sde_samplers/predictor_corrector_sampler.py, loss/sde_loss/score_function_loss.py:48-91)
# ----------------------------------------------------------------------------------------------
class VESDE:
    """sde_lib/ve_sde.py:7-59."""
    sampling_epsilon = 1e-5
    T = 1

    def __init__(self, sigma_min=0.01, sigma_max=50, N=1000):
        self.sigma_min, self.sigma_max, self.N = sigma_min, sigma_max, N
        self.discrete_sigmas = torch.exp(torch.linspace(np.log(sigma_min), np.log(sigma_max), N))

    def sde(self, x, t):
        sigma = self.sigma_min * (self.sigma_max / self.sigma_min) ** t
        diffusion = sigma * torch.sqrt(torch.tensor(2 * (np.log(self.sigma_max) - np.log(self.sigma_min))))
        return torch.zeros_like(x), diffusion

    def marginal_prob(self, x, t):
        return x, self.sigma_min * (self.sigma_max / self.sigma_min) ** t

    def discretize(self, x, t):
        timestep = (t * (self.N - 1) / self.T).long()
        sigma = self.discrete_sigmas[timestep]
        adjacent = torch.where(timestep == 0, torch.zeros_like(t), self.discrete_sigmas[timestep - 1])
        return torch.zeros_like(x), torch.sqrt(sigma ** 2 - adjacent ** 2)


class VPSDE:
    """sde_lib/vp_sde.py:11-71."""
    sampling_epsilon = 1e-3
    T = 1

    def __init__(self, beta_min=0.1, beta_max=20, N=1000):
        self.beta_0, self.beta_1, self.N = beta_min, beta_max, N
        self.betas = torch.linspace(beta_min / N, beta_max / N, N)
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.sqrt_alphas_cumprod = torch.sqrt(self.alphas_cumprod)
        self.sqrt_1m_alphas_cumprod = torch.sqrt(1.0 - self.alphas_cumprod)

    def sde(self, x, t):
        beta_t = self.beta_0 + t * (self.beta_1 - self.beta_0)
        return -0.5 * beta_t[:, None, None, None] * x, torch.sqrt(beta_t)

    def marginal_prob(self, x, t):
        lmc = -0.25 * t ** 2 * (self.beta_1 - self.beta_0) - 0.5 * t * self.beta_0
        return torch.exp(lmc[:, None, None, None]) * x, torch.sqrt(1.0 - torch.exp(2.0 * lmc))

    def discretize(self, x, t):
        timestep = (t * (self.N - 1) / self.T).long()
        beta = self.betas[timestep]
        alpha = self.alphas[timestep]
        return torch.sqrt(alpha)[:, None, None, None] * x - x, torch.sqrt(beta)


def score_fn_for(sde, model_fn, continuous: bool = True):
    """loss/sde_loss/score_function_loss.py:48-91 (VP and VE branches)."""
    if isinstance(sde, VPSDE):
        def score_fn(x, t):
            labels = t * (sde.N - 1)
            score = model_fn(x, labels)
            if continuous:
                std = sde.marginal_prob(torch.zeros_like(x), t)[1]
            else:
                std = sde.sqrt_1m_alphas_cumprod[labels.long()]
            return -score / std[:, None, None, None]
        return score_fn

    def score_fn(x, t):
        if continuous:
            labels = sde.marginal_prob(torch.zeros_like(x), t)[1]
        else:
            labels = torch.round((sde.T - t) * (sde.N - 1)).long()
        return model_fn(x, labels)
    return score_fn


def reverse_sde(sde, score_fn, x, t, probability_flow: bool = False):
    """RSDE.sde (sde_lib/sde_lib.py:91-98). With probability_flow the score term is halved and the diffusion is the
    Python float 0.0, as in the reference."""
    drift, diffusion = sde.sde(x, t)
    drift = drift - diffusion[:, None, None, None] ** 2 * score_fn(x, t) * (0.5 if probability_flow else 1.0)
    return drift, (0.0 if probability_flow else diffusion)


def reverse_discretize(sde, score_fn, x, t, probability_flow: bool = False):
    """RSDE.discretize (sde_lib/sde_lib.py:100-105)."""
    f, G = sde.discretize(x, t)
    rev_f = f - G[:, None, None, None] ** 2 * score_fn(x, t) * (0.5 if probability_flow else 1.0)
    return rev_f, (torch.zeros_like(G) if probability_flow else G)


def predictor_step(name, sde, score_fn, x, t, z, probability_flow: bool = False):
    """One predictor update with injected noise ``z``; returns (x, x_mean).
    euler_maruyama_predictor.py:11-17, reverse_diffusion_predictor.py:11-16,
    ancestral_sampling_predictor.py:17-28 (VE branch). Euler-Maruyama with probability_flow raises TypeError like the
    reference, which subscripts the float diffusion 0.0 (euler_maruyama_predictor.py:16)."""
    if name in (None, "none", "null"):
        return x, x
    if name == "euler_maruyama":
        dt = -1.0 / sde.N
        drift, diffusion = reverse_sde(sde, score_fn, x, t, probability_flow)
        x_mean = x + drift * dt
        return x_mean + diffusion[:, None, None, None] * np.sqrt(-dt) * z, x_mean
    if name == "reverse_diffusion":
        f, G = reverse_discretize(sde, score_fn, x, t, probability_flow)
        x_mean = x - f
        return x_mean + G[:, None, None, None] * z, x_mean
    if name == "ancestral_sampling":
        if not isinstance(sde, VESDE):
            raise NotImplementedError("VP ancestral sampling reads sde.discrete_betas, which VPSDE lacks")
        assert not probability_flow, "Probability flow not supported by ancestral sampling"
        timestep = (t * (sde.N - 1) / sde.T).long()
        sigma = sde.discrete_sigmas[timestep]
        adjacent = torch.where(timestep == 0, torch.zeros_like(t), sde.discrete_sigmas[timestep - 1])
        score = score_fn(x, t)
        x_mean = x + score * (sigma ** 2 - adjacent ** 2)[:, None, None, None]
        std = torch.sqrt((adjacent ** 2 * (sigma ** 2 - adjacent ** 2)) / (sigma ** 2))
        return x_mean + std[:, None, None, None] * z, x_mean
    raise ValueError(name)


def corrector_step(name, sde, score_fn, x, t, z_fn, snr, n_steps):
    """``n_steps`` corrector updates; ``z_fn(k)`` is the noise of inner step k. Returns (x, x_mean).
    langevin_corrector.py:15-35, annealed_langevin_dynamics_corrector.py:21-41."""
    if name in (None, "none", "null"):
        return x, x
    if isinstance(sde, VPSDE):
        timestep = (t * (sde.N - 1) / sde.T).long()
        alpha = sde.alphas[timestep]
    else:
        alpha = torch.ones_like(t)
    x_mean = x
    if name == "langevin":
        for k in range(n_steps):
            grad = score_fn(x, t)
            noise = z_fn(k)
            grad_norm = torch.norm(grad.reshape(grad.shape[0], -1), dim=-1).mean()
            noise_norm = torch.norm(noise.reshape(noise.shape[0], -1), dim=-1).mean()
            step_size = (snr * noise_norm / grad_norm) ** 2 * 2 * alpha
            x_mean = x + step_size[:, None, None, None] * grad
            x = x_mean + torch.sqrt(step_size * 2)[:, None, None, None] * noise
        return x, x_mean
    if name == "ald":
        std = sde.marginal_prob(x, t)[1]
        for k in range(n_steps):
            grad = score_fn(x, t)
            noise = z_fn(k)
            step_size = (snr * std) ** 2 * 2 * alpha
            x_mean = x + step_size[:, None, None, None] * grad
            x = x_mean + noise * torch.sqrt(step_size * 2)[:, None, None, None]
        return x, x_mean
    raise ValueError(name)


def pc_sample_chain(model_fn, sde, shape, noise_fn, predictor="euler_maruyama", corrector=None,
                    snr=0.16, n_steps=1, continuous=True, denoise=True, eps=None, n_iter=None, keep=None,
                    probability_flow=False):
    """sde_samplers/predictor_corrector_sampler.py:58-120 with injected noise.

    ``noise_fn(step, stream)``: stream 0 = prior draw (step == N), stream 1 = predictor noise of
    iteration ``step``, stream 2+k = corrector inner step k. The prior draw is scaled by sigma_max for
    VE (ve_sde.py:40-41). ``n_iter`` truncates the loop (first n_iter of the N timesteps) for short
    golden chains. Returns {iteration: x_mean or x} for iterations in ``keep`` (default: last)."""
    eps = sde.sampling_epsilon if eps is None else eps
    score_fn = score_fn_for(sde, model_fn, continuous)
    x = noise_fn(sde.N, 0)
    if isinstance(sde, VESDE):
        x = x * sde.sigma_max
    timesteps = torch.linspace(sde.T, eps, sde.N)
    n_iter = sde.N if n_iter is None else n_iter
    keep = {n_iter - 1} if keep is None else set(keep)
    out = {"x_T": x.clone()}
    for i in range(n_iter):
        vec_t = torch.ones(shape[0]) * timesteps[i]
        x, x_mean = corrector_step(corrector, sde, score_fn, x, vec_t,
                                   lambda k, i=i: noise_fn(i, 2 + k), snr, n_steps)
        x, x_mean = predictor_step(predictor, sde, score_fn, x, vec_t,
                                   None if predictor in (None, "none", "null") else noise_fn(i, 1), probability_flow)
        if i in keep:
            out[i] = (x_mean if denoise else x).clone()
    out["x_last"] = x.clone()
    return out


# ----------------------------------------------------------------------------------------------
# Training step (diffusion_model_nemo/models/ddpm.py:43-57, loss/simple_loss.py:37-45)
# ----------------------------------------------------------------------------------------------
def ddpm_training_loss(model_fn, c, x_start: Tensor, t: Tensor, noise: Tensor, loss_type: str = "l2") -> Tensor:
    x_t = q_sample(c, x_start, t, noise)
    out = model_fn(x_t, t)
    fn = {"l1": F.l1_loss, "l2": F.mse_loss, "huber": F.smooth_l1_loss}[loss_type]
    return fn(out, noise, reduction="mean")


def to_torch(params_np) -> Params:
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in params_np.items()}
