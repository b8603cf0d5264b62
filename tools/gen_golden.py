#!/usr/bin/env python3
"""Capture golden vectors from the reference implementation (runs only in the build container).

The reference (read-only at /root/reference) is imported as a library; the third-party packages it
imports but this image lacks (``nemo``, ``omegaconf``) are replaced by inert placeholders, since the
hot path only uses them as base classes / decorators:
``NeuralModule`` is a ``torch.nn.Module``, ``Loss`` a ``torch.nn.modules.loss._Loss`` and ``typecheck``
an identity decorator. No reference source is copied: the script only CALLS the reference and stores
inputs and outputs as small ``.npz`` fixtures under ``tests/golden/``.

Weights come from ``oracle/weights.py`` (name-hashed, deterministic) and every Gaussian draw the
reference makes through ``torch.randn`` / ``torch.randn_like`` is replaced, for the duration of a
capture, by the counter-based noise of ``oracle/philox.py`` so that the oracle and the GPU path can
replay exactly the same chain.

Usage:  python tools/gen_golden.py [--ref /root/reference] [--out tests/golden] [--only NAME ...]
"""
import argparse
import logging
import os
import sys
import time
import types

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import philox  # noqa: E402
from oracle.weights import make_param, make_state_dict  # noqa: E402


def install_placeholders():
    from torch.nn.modules.loss import _Loss

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    def typecheck(*a, **k):
        def deco(f):
            return f
        return deco

    class NeuralType:
        def __init__(self, *a, **k):
            pass

    class LossType:
        pass

    class DictConfig(dict):
        pass

    class OmegaConf:
        pass

    nemo = mod("nemo")
    core = mod("nemo.core", NeuralModule=nn.Module, Loss=_Loss, typecheck=typecheck)
    nt = mod("nemo.core.neural_types", NeuralType=NeuralType, LossType=LossType)
    utils = mod("nemo.utils", logging=logging.getLogger("nemo"))
    nemo.core, nemo.utils, core.neural_types = core, utils, nt
    mod("omegaconf", DictConfig=DictConfig, OmegaConf=OmegaConf)


class NoiseInjector:
    """Context manager: torch.randn / torch.randn_like return Philox noise following ``plan``,
    an iterator of (step, stream) pairs in the order the reference draws."""

    def __init__(self, plan, seed, image_offset=0):
        self.plan = iter(plan)
        self.seed = seed
        self.image_offset = image_offset
        self.calls = 0

    def _draw(self, shape):
        step, stream = next(self.plan)
        self.calls += 1
        return torch.from_numpy(philox.normal_like(tuple(shape), self.seed, step, stream, self.image_offset))

    def __enter__(self):
        self._randn, self._randn_like = torch.randn, torch.randn_like

        def randn(*shape, **kw):
            if len(shape) == 1 and isinstance(shape[0], (list, tuple, torch.Size)):
                shape = tuple(shape[0])
            return self._draw(shape)

        def randn_like(x, **kw):
            return self._draw(x.shape)

        torch.randn, torch.randn_like = randn, randn_like
        return self

    def __exit__(self, *exc):
        torch.randn, torch.randn_like = self._randn, self._randn_like


def load_named(module: nn.Module, prefix: str = "", seed: int = 0):
    """Fill ``module`` with name-hashed weights; names are ``prefix + state_dict key``."""
    sd = module.state_dict()
    new = {k: torch.from_numpy(make_param(prefix + k, tuple(v.shape), seed=seed)) for k, v in sd.items()}
    module.load_state_dict(new)
    return module


def npz(out_dir, name, **arrays):
    path = os.path.join(out_dir, name + ".npz")
    conv = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        conv[k] = np.asarray(v)
    np.savez_compressed(path, **conv)
    print(f"  wrote {path} ({os.path.getsize(path) / 1024:.1f} KiB)")


def x_input(shape, seed, step=0, stream=7):
    return torch.from_numpy(philox.normal_like(shape, seed, step, stream))


# ---------------------------------------------------------------------------------------------
def gen_schedules(M, out):
    arrays = {}
    names = ("betas", "alphas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_recip_alphas",
             "sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod", "sqrt_recip_alphas_cumprod",
             "sqrt_recipm1_alphas_cumprod", "log_one_minus_alphas_cumprod", "posterior_variance",
             "posterior_log_variance_clipped", "posterior_mean_coef1", "posterior_mean_coef2")
    for sched in ("cosine", "linear", "quadratic", "sigmoid"):
        for T in (10, 1000):
            gd = M.GaussianDiffusion(timesteps=T, schedule_name=sched)
            for n in names:
                arrays[f"{sched}_{T}_{n}"] = getattr(gd, n)
    # non-default schedule arguments reach the schedule function through schedule_cfg
    gd = M.GaussianDiffusion(timesteps=50, schedule_name="linear",
                             schedule_cfg={"linear": {"beta_start": 0.001, "beta_end": 0.05}})
    for n in names:
        arrays[f"linearcfg_50_{n}"] = getattr(gd, n)
    npz(out, "schedules", **arrays)


def gen_blocks(M, out):
    from diffusion_model_nemo.parts import convnext, mha, positional_encoding
    from diffusion_model_nemo import utils as rutils
    arrays = {}
    B = 2
    temb = x_input((B, 128), seed=11)
    arrays["temb"] = temb
    # Block / ResnetBlock
    for tag, cin, cout, hw, groups, with_t in (("rb_32_32", 32, 32, 16, 8, True), ("rb_32_64", 32, 64, 8, 8, True),
                                               ("rb_64_32_g4", 64, 32, 8, 4, True), ("rb_32_32_not", 32, 32, 8, 8, False)):
        m = convnext.ResnetBlock(cin, cout, time_emb_dim=128 if with_t else None, groups=groups)
        load_named(m, tag + ".")
        x = x_input((B, cin, hw, hw), seed=12)
        with torch.no_grad():
            y = m(x, temb if with_t else None)
        arrays[tag + "_x"], arrays[tag + "_y"] = x, y
    m = convnext.Block(32, 64, groups=8)
    load_named(m, "blk.")
    x = x_input((B, 32, 8, 8), seed=13)
    with torch.no_grad():
        arrays["blk_x"], arrays["blk_y"] = x, m(x)
    # attention variants (bare and wrapped in Residual(PreNorm(.)))
    for tag, c, hw in (("la_32", 32, 16), ("la_64", 64, 8)):
        m = rutils.Residual(rutils.PreNorm(c, mha.LinearAttention(c)))
        load_named(m, tag + ".")
        x = x_input((B, c, hw, hw), seed=14)
        with torch.no_grad():
            arrays[tag + "_x"], arrays[tag + "_y"] = x, m(x)
            arrays[tag + "_bare_y"] = m.fn.fn(x)
    for tag, c, hw in (("at_64_4", 64, 4), ("at_256_8", 256, 8)):
        m = rutils.Residual(rutils.PreNorm(c, mha.Attention(c)))
        load_named(m, tag + ".")
        x = x_input((B, c, hw, hw), seed=15)
        with torch.no_grad():
            arrays[tag + "_x"], arrays[tag + "_y"] = x, m(x)
            arrays[tag + "_bare_y"] = m.fn.fn(x)
    # down / up sample
    m = rutils.Downsample(32)
    load_named(m, "down.")
    x = x_input((B, 32, 16, 16), seed=16)
    with torch.no_grad():
        arrays["down_x"], arrays["down_y"] = x, m(x)
    m = rutils.Upsample(32)
    m.load_state_dict({"weight": torch.from_numpy(make_param("ups.9.3.weight", (32, 32, 4, 4))),
                       "bias": torch.from_numpy(make_param("ups.9.3.bias", (32,)))})
    x = x_input((B, 32, 8, 8), seed=17)
    with torch.no_grad():
        arrays["up_x"], arrays["up_y"] = x, m(x)
    # sinusoidal embedding for integer and float time
    pe = positional_encoding.SinusoidalPositionEmbeddings(32)
    t_long = torch.tensor([0, 1, 500, 999], dtype=torch.long)
    t_float = torch.tensor([0.01, 0.37, 12.5, 50.0, 999.0 * 0.731], dtype=torch.float32)
    arrays["pe_t_long"], arrays["pe_long"] = t_long, pe(t_long)
    arrays["pe_t_float"], arrays["pe_float"] = t_float, pe(t_float)
    npz(out, "blocks", **arrays)


def build_unet(M, learned_variance=False, groups=8, order="bn_act_conv", seed=0):
    u = M.Unet(input_dim=32, dim=32, dim_mults=[1, 2, 4, 8], channels=3, with_time_emb=True,
               resnet_block_groups=groups, use_convnext=False, resnet_block_order=order, dropout=0.0,
               learned_variance=learned_variance)
    load_named(u, "", seed=seed)
    u.eval()
    return u


def gen_unet(M, out):
    from oracle.weights import unet_param_shapes
    arrays = {}
    B = 2
    x = x_input((B, 3, 32, 32), seed=21)
    arrays["x"] = x
    for tag, lv, groups, order in (("ddpm", False, 8, "bn_act_conv"), ("improved", True, 8, "bn_act_conv"),
                                   ("sde_g4", False, 4, "bn_act_conv"), ("convbnact", False, 8, "conv_bn_act")):
        u = build_unet(M, lv, groups, order)
        # the oracle's shape table must list exactly the reference's parameters
        ref_shapes = {k: tuple(v.shape) for k, v in u.state_dict().items()}
        mine = {k: tuple(v) for k, v in unet_param_shapes(learned_variance=lv, resnet_block_order=order).items()}
        assert ref_shapes == mine, (set(ref_shapes) ^ set(mine))
        with torch.no_grad():
            for t in (0, 500, 999):
                arrays[f"{tag}_t{t}"] = u(x, torch.full((B,), t, dtype=torch.long))
            arrays[f"{tag}_tmixed"] = u(x, torch.tensor([3, 977], dtype=torch.long))
            tf = torch.tensor([0.0123, 37.5], dtype=torch.float32)
            arrays[f"{tag}_tfloat"] = u(x, tf)
        arrays["t_float"] = tf
    # 64x64 input (bottleneck attention over 64 tokens), one image
    u = build_unet(M)
    x64 = x_input((1, 3, 64, 64), seed=22)
    with torch.no_grad():
        arrays["x64"], arrays["ddpm64_t250"] = x64, u(x64, torch.tensor([250]))
    npz(out, "unet", **arrays)


def gen_psample(M, out):
    arrays = {}
    B = 3
    shape = (B, 3, 8, 8)
    x = x_input(shape, seed=31)
    eps = x_input(shape, seed=32)
    eps6 = x_input((B, 6, 8, 8), seed=33)
    z = x_input(shape, seed=34)
    arrays.update(x=x, eps=eps, eps6=eps6, z=z)
    for sched in ("cosine", "linear"):
        gd = M.GaussianDiffusion(timesteps=1000, schedule_name=sched)
        lgd = M.LearnedGaussianDiffusion(timesteps=1000, schedule_name=sched)
        for tt in ([999, 999, 999], [1, 1, 1], [0, 0, 0], [0, 17, 640]):
            t = torch.tensor(tt, dtype=torch.long)
            tag = f"{sched}_t{'_'.join(map(str, tt))}"
            with NoiseInjector(iter([(0, 0)] * 4), seed=0) as inj:
                torch.randn_like = lambda _x, **kw: z.clone()
                arrays["fixed_" + tag] = gd.p_sample(lambda a, b: eps.clone(), x.clone(), t)
                arrays["learned_" + tag] = lgd.p_sample(lambda a, b: eps6.clone(), x.clone(), t)
        tq = torch.tensor([0, 499, 999], dtype=torch.long)
        arrays[f"qsample_{sched}"] = gd.q_sample(x, tq, noise=z)
    gd0 = M.GaussianDiffusion(timesteps=1000, schedule_name="cosine", objective="pred_x0")
    t = torch.tensor([5, 50, 500])
    with NoiseInjector(iter([(0, 0)] * 4), seed=0):
        torch.randn_like = lambda _x, **kw: z.clone()
        arrays["predx0_cosine"] = gd0.p_sample(lambda a, b: eps.clone(), x.clone(), t)
    npz(out, "p_sample", **arrays)


def gen_chain(M, out, seed=0):
    """Full T=1000 ancestral chains, B=2 @ 32x32 (fixed variance) and a T=100 learned-variance chain."""
    B, T = 2, 1000
    shape = (B, 3, 32, 32)
    keep = (999, 900, 500, 100, 0)
    u = build_unet(M)
    gd = M.GaussianDiffusion(timesteps=T, schedule_name="cosine")
    plan = [(T, 0)] + [(i, 1) for i in reversed(range(T))]
    t0 = time.time()
    with NoiseInjector(plan, seed=seed), torch.no_grad():
        imgs = gd.p_sample_loop(u, shape, device=torch.device("cpu"), use_tqdm=False)
    print(f"  1000-step reference chain: {time.time() - t0:.1f} s")
    arrays = {"x_T": philox.normal_like(shape, seed, T, 0)}
    for i in keep:
        arrays[f"x_{i}"] = imgs[T - 1 - i] * 2.0 - 1.0  # undo the (x+1)/2 display mapping
        arrays[f"img_{i}"] = imgs[T - 1 - i]
    npz(out, "chain_ddpm_T1000_B2_32", **arrays)

    # learned variance, T=100 cosine
    T2 = 100
    u2 = build_unet(M, learned_variance=True)
    lgd = M.LearnedGaussianDiffusion(timesteps=T2, schedule_name="cosine")
    plan = [(T2, 0)] + [(i, 1) for i in reversed(range(T2))]
    with NoiseInjector(plan, seed=seed + 1), torch.no_grad():
        imgs = lgd.p_sample_loop(u2, shape, device=torch.device("cpu"), use_tqdm=False)
    arrays = {f"x_{i}": imgs[T2 - 1 - i] * 2.0 - 1.0 for i in (99, 50, 10, 0)}
    npz(out, "chain_improved_T100_B2_32", **arrays)

    # shard invariance fixture: image index 1 of the chain above regenerated alone with offset 1
    # is covered by the oracle tests (no reference run needed).


def gen_sde(M, out, seed=0):
    from diffusion_model_nemo.loss import SDEScoreFunctionLoss
    arrays = {}
    B = 2
    shape = (B, 3, 8, 8)
    x = x_input(shape, seed=41)
    z = x_input(shape, seed=42)
    score = x_input(shape, seed=43) * 0.3
    arrays.update(x=x, z=z, score=score)
    tvals = torch.tensor([1.0, 0.5005], dtype=torch.float32)
    arrays["t"] = tvals
    ve = M.VESDE(sigma_min=0.01, sigma_max=50.0, N=1000)
    vp = M.VPSDE(beta_min=0.1, beta_max=20.0, N=1000)
    arrays["ve_discrete_sigmas"] = ve.discrete_sigmas
    arrays["vp_betas"], arrays["vp_alphas"] = vp.betas, vp.alphas
    arrays["vp_sqrt_1m_alphas_cumprod"] = vp.sqrt_1m_alphas_cumprod
    tt = torch.linspace(1.0, 1e-5, 9)
    xx = x_input((9, 3, 8, 8), seed=44)
    arrays["tt"], arrays["xx"] = tt, xx
    for name, sde in (("ve", ve), ("vp", vp)):
        d, g = sde.sde(xx, tt)
        m, s = sde.marginal_prob(xx, tt)
        f, G = sde.discretize(xx, tt)
        arrays.update({f"{name}_drift": d, f"{name}_diffusion": g, f"{name}_mean": m, f"{name}_std": s,
                       f"{name}_f": f, f"{name}_G": G})
    score_fn = lambda a, b: score.clone()  # noqa: E731
    for sname, sde in (("ve", ve), ("vp", vp)):
        for pname in ("euler_maruyama", "reverse_diffusion", "ancestral_sampling"):
            if sname == "vp" and pname == "ancestral_sampling":
                continue  # reads sde.discrete_betas, which VPSDE never defines
            pred = M.get_predictor(pname)(sde, score_fn, False)
            with NoiseInjector(iter([(0, 0)] * 2), seed=0):
                torch.randn_like = lambda _x, **kw: z.clone()
                xo, xm = pred.update_fn(x.clone(), tvals)
            arrays[f"{sname}_{pname}_x"], arrays[f"{sname}_{pname}_mean"] = xo, xm
        for cname in ("langevin", "ald"):
            corr = M.get_corrector(cname)(sde, score_fn, 0.16, 2)
            zs = [z.clone(), x_input(shape, seed=45)]
            arrays["z2"] = zs[1]
            it = iter(zs)
            with NoiseInjector(iter([(0, 0)] * 2), seed=0):
                torch.randn_like = lambda _x, **kw: next(it)
                xo, xm = corr.update_fn(x.clone(), tvals)
            arrays[f"{sname}_{cname}_x"], arrays[f"{sname}_{cname}_mean"] = xo, xm
    # score-function adapters around a real network (groups=4 as in the score_sde YAML)
    u = build_unet(M, groups=4)
    x32 = x_input((B, 3, 32, 32), seed=46)
    arrays["x32"] = x32
    with torch.no_grad():
        for sname, sde in (("ve", ve), ("vp", vp)):
            for cont in (True, False):
                fn = SDEScoreFunctionLoss.resolve_score_function(u, sde, continuous=cont)
                arrays[f"score_{sname}_{'cont' if cont else 'disc'}"] = fn(x32, tvals.clone())
    npz(out, "sde_units", **arrays)

    # short predictor-corrector chains with N = 20
    N = 20
    shape = (B, 3, 32, 32)
    arrays = {}
    for tag, sde, pred, corr, n_steps in (
            ("ve_rd_langevin", M.VESDE(0.01, 50.0, N), "reverse_diffusion", "langevin", 1),
            ("ve_em_none", M.VESDE(0.01, 50.0, N), "euler_maruyama", None, 1),
            ("vp_em_none", M.VPSDE(0.1, 20.0, N), "euler_maruyama", None, 1),
            ("vp_rd_langevin2", M.VPSDE(0.1, 20.0, N), "reverse_diffusion", "langevin", 2),
            ("ve_anc_ald", M.VESDE(0.01, 50.0, N), "ancestral_sampling", "ald", 1)):
        sampler = M.PredictorCorrectorSampler(predictor=pred, corrector=corr, snr=0.16, n_steps=n_steps)
        sampler.update_sde(sde)
        plan = [(N, 0)]
        for i in range(N):
            if corr is not None:
                plan += [(i, 2 + k) for k in range(n_steps)]
            plan += [(i, 1)]
        with NoiseInjector(plan, seed=seed + 5):
            imgs, nfe = sampler.forward(u, list(shape), torch.device("cpu"), return_nfe=True, use_tqdm=False)
        for i in (0, 9, 19):
            arrays[f"{tag}_mean_{i}"] = imgs[i] * 2.0 - 1.0
        arrays[f"{tag}_nfe"] = np.int64(nfe)
    npz(out, "chain_sde_N20_B2_32", **arrays)


def gen_train(M, out):
    from diffusion_model_nemo.loss import DiffusionLoss
    B = 2
    u = build_unet(M)
    u.train()
    gd = M.GaussianDiffusion(timesteps=1000, schedule_name="cosine")
    x0 = torch.tanh(x_input((B, 3, 32, 32), seed=51))
    noise = x_input((B, 3, 32, 32), seed=52)
    t = torch.tensor([17, 803], dtype=torch.long)
    loss_fn = DiffusionLoss("l2", "mean")
    x_t = gd.q_sample(x_start=x0, t=t, noise=noise)
    loss = loss_fn(input=u(x_t, t), target=noise)
    loss.backward()
    gn = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in u.parameters())).float()
    arrays = dict(x0=x0, noise=noise, t=t, x_t=x_t, loss=loss.detach(), grad_norm=gn,
                  grad_init_conv_bias=u.init_conv.bias.grad, grad_final_bias=u.final_conv[3].bias.grad,
                  grad_mid_qkv=u.mid_attn.fn.fn.to_qkv.weight.grad.reshape(-1)[:256])
    npz(out, "train_step", **arrays)


def gen_cond(M, out):
    """Class-conditional Unet (unet.py:118-141): 10 classes; labelled, null-class and default (classes=None) forwards."""
    from oracle.weights import unet_param_shapes
    u = M.Unet(input_dim=32, dim=32, dim_mults=[1, 2, 4, 8], channels=3, with_time_emb=True, resnet_block_groups=8,
               use_convnext=False, dropout=0.0, num_classes=10)
    load_named(u, "", seed=0)   # the padding row gets ordinary values too: the forward just looks the row up
    u.eval()
    ref_shapes = {k: tuple(v.shape) for k, v in u.state_dict().items()}
    assert ref_shapes == {k: tuple(v) for k, v in unet_param_shapes(num_classes=10).items()}
    x = x_input((3, 3, 32, 32), seed=23)
    t = torch.tensor([5, 400, 990], dtype=torch.long)
    labels = torch.tensor([3, 10, 0], dtype=torch.long)
    with torch.no_grad():
        arrays = {"x": x, "t": t, "labels": labels, "y_labels": u(x, t, labels), "y_default": u(x, t)}
    npz(out, "unet_cond", **arrays)


def gen_bpd(M, out, seed=4):
    """Bits per dimension (models/abstract_diffusion_model.py:137-197). `models/` cannot be imported without NeMo /
    Lightning, so the 30-line loop is re-driven here over the reference's own pieces: the samplers' q_sample /
    q_posterior / p_mean_variance / q_mean_variance, VariationalBoundLoss.compute_variation_loss_terms and
    utils.normal_kl / mean_flattened. T=20, B=2 @ 32x32, fixed and learned variance."""
    import math
    from diffusion_model_nemo import utils
    from diffusion_model_nemo.loss.variational_bound_loss import VariationalBoundLoss
    B, T = 2, 20
    shape = (B, 3, 32, 32)
    x0 = (x_input(shape, seed=41) * 0.5).clamp(-1, 1)
    x0[0, :, :4] = 1.0    # exercise both saturated branches of the discretised likelihood
    x0[1, :, :4] = -1.0
    arrays = {"x0": x0}
    for tag, lv in (("fixed", False), ("learned", True)):
        u = build_unet(M, learned_variance=lv)
        sampler = (M.LearnedGaussianDiffusion if lv else M.GaussianDiffusion)(timesteps=T, schedule_name="cosine")
        terms = torch.zeros(B, T)
        plan = [(t, 3) for t in range(T - 1, -1, -1)]
        with NoiseInjector(plan, seed=seed), torch.no_grad():
            for t in range(T - 1, -1, -1):
                t_b = torch.full([B], t, dtype=torch.long)
                x_t = sampler.q_sample(x_start=x0, t=t_b)
                true_mean, true_lv = sampler.q_posterior(x_start=x0, x=x_t, t=t_b)
                model_mean, _, model_lv, _ = sampler.p_mean_variance(u, x=x_t, t=t_b, return_pred_x_start=True)
                if model_lv.shape != model_mean.shape:
                    model_lv = model_lv.expand(-1, *model_mean.size()[1:])
                terms[:, t] = VariationalBoundLoss.compute_variation_loss_terms(
                    samples=x0, model_mean=model_mean, model_log_variance=model_lv, true_mean=true_mean,
                    true_log_variance_clipped=true_lv, t=t_b)
            t_prior = torch.full([B], T - 1, dtype=torch.long)
            qt_mean, _, qt_lv = sampler.q_mean_variance(x_start=x0, t=t_prior)
            zero = torch.tensor(0.0)
            prior = utils.mean_flattened(utils.normal_kl(mean1=qt_mean, logvar1=qt_lv, mean2=zero, logvar2=zero)) / math.log(2.0)
        arrays[f"{tag}_terms"] = terms
        arrays[f"{tag}_prior"] = prior
        arrays[f"{tag}_total"] = terms.sum(dim=1) + prior
    npz(out, "bpd_T20_B2_32", **arrays)


def gen_ddim(M, out, seed=3):
    """GeneralizedGaussianDiffusion (DDIM) chains, B=2 @ 32x32, T=1000 cosine: 20 strided steps with eta=0 and
    eta=0.5, plus a pred_x0-objective chain of 10 steps; every x_t kept (model range)."""
    B, T = 2, 1000
    shape = (B, 3, 32, 32)
    u = build_unet(M)
    arrays = {}
    for tag, steps, eta, objective in (("eta0_s20", 20, 0.0, "pred_noise"), ("eta05_s20", 20, 0.5, "pred_noise"),
                                       ("x0_s10", 10, 0.3, "pred_x0")):
        g = M.GeneralizedGaussianDiffusion(timesteps=T, schedule_name="cosine", objective=objective, eta=eta,
                                           ddim_timesteps=steps)
        stride = T // steps
        seq = list(range(0, T, stride))
        # the reference draws x_T, then one randn_like per step (also when eta == 0), keyed here by the timestep
        plan = [(T, 0)] + [(i, 1) for i in reversed(seq)]
        with NoiseInjector(plan, seed=seed), torch.no_grad():
            imgs = g.p_sample_loop(u, shape, use_tqdm=False, device=torch.device("cpu"))
        arrays[f"{tag}_xs"] = torch.stack(imgs) * 2.0 - 1.0
        arrays[f"{tag}_meta"] = np.array([steps, eta, 1.0 if objective == "pred_x0" else 0.0], dtype=np.float64)
    arrays["x_T"] = philox.normal_like(shape, seed, T, 0)
    npz(out, "chain_ddim_T1000_B2_32", **arrays)


def gen_switches(M, out, seed=6):
    """Boundary methods and sampler options the other fixtures leave out: GaussianDiffusion.interpolate
    (gaussian_diffusion.py:195-218), GeneralizedGaussianDiffusion.interpolate (generalized_gaussian_diffusion.py:138-140),
    PredictorCorrectorSampler with continuous=False / probability_flow=True / denoise=False
    (predictor_corrector_sampler.py:19-29,97-113; sde_lib.py:91-105; score_function_loss.py:60-72,80-90), and Langevin /
    annealed-Langevin corrector steps in which every sample shares one time (langevin_corrector.py:15-35)."""
    B = 2
    shape = (B, 3, 32, 32)
    arrays = {}
    u = build_unet(M)
    # DDPM interpolation: q_sample of both images at t (two randn_like draws), mix, then t ancestral steps
    T, t_mix, lam = 1000, 50, 0.3
    gd = M.GaussianDiffusion(timesteps=T, schedule_name="cosine")
    x1 = torch.tanh(x_input(shape, seed=61))
    x2 = torch.tanh(x_input(shape, seed=62))
    plan = [(t_mix, 4), (t_mix, 5)] + [(i, 1) for i in reversed(range(t_mix))]
    with NoiseInjector(plan, seed=seed), torch.no_grad():
        imgs = gd.interpolate(u, x1, x2, t=t_mix, lambd=lam)
    assert len(imgs) == t_mix
    arrays.update(interp_x1=x1, interp_x2=x2, interp_meta=np.array([T, t_mix, lam], dtype=np.float64),
                  interp_xs=torch.stack([imgs[i] for i in (0, 24, 49)]) * 2.0 - 1.0)
    # DDIM "interpolation": the whole strided sequence applied to a given latent (no x_T draw)
    g = M.GeneralizedGaussianDiffusion(timesteps=T, schedule_name="cosine", eta=0.0, ddim_timesteps=10)
    lat = x_input(shape, seed=63)
    seq = list(range(0, T, T // 10))
    with NoiseInjector([(i, 1) for i in reversed(seq)], seed=seed), torch.no_grad():
        imgs = g.interpolate(u, lat)
    arrays.update(ddim_interp_latent=lat, ddim_interp_xs=torch.stack(imgs) * 2.0 - 1.0)

    # predictor-corrector options, N = 20, the score network of the score_sde YAML (groups = 4)
    N = 20
    u4 = build_unet(M, groups=4)
    for tag, sde, pred, corr, kw in (
            ("vp_rd_langevin_disc", M.VPSDE(0.1, 20.0, N), "reverse_diffusion", "langevin", dict(continuous=False)),
            ("ve_em_none_disc", M.VESDE(0.01, 50.0, N), "euler_maruyama", None, dict(continuous=False)),
            ("ve_rd_langevin_pf", M.VESDE(0.01, 50.0, N), "reverse_diffusion", "langevin", dict(probability_flow=True)),
            # (Euler-Maruyama with probability_flow=True raises in the reference: RSDE.sde returns the float 0.0 as diffusion,
            #  sde_lib.py:97, which euler_maruyama_predictor.py:16 then subscripts)
            ("vp_rd_none_pf", M.VPSDE(0.1, 20.0, N), "reverse_diffusion", None, dict(probability_flow=True)),
            ("ve_rd_langevin_raw", M.VESDE(0.01, 50.0, N), "reverse_diffusion", "langevin", dict(denoise=False)),
            ("vp_none_langevin", M.VPSDE(0.1, 20.0, N), None, "langevin", dict())):
        sampler = M.PredictorCorrectorSampler(predictor=pred, corrector=corr, snr=0.16, n_steps=1, **kw)
        sampler.update_sde(sde)
        plan = [(N, 0)]
        for i in range(N):
            if corr is not None:
                plan += [(i, 2)]
            if pred is not None:
                plan += [(i, 1)]   # NonePredictor draws nothing
        with NoiseInjector(plan, seed=seed + 1):
            imgs, nfe = sampler.forward(u4, list(shape), torch.device("cpu"), return_nfe=True, use_tqdm=False)
        for i in (0, 9, 19):
            arrays[f"{tag}_img_{i}"] = imgs[i] * 2.0 - 1.0
        arrays[f"{tag}_nfe"] = np.int64(nfe)

    # corrector steps at ONE time for the whole batch (what a sampling loop feeds them): two inner steps
    sshape = (B, 3, 8, 8)
    x = x_input(sshape, seed=64)
    score = x_input(sshape, seed=65) * 0.3
    zs = [x_input(sshape, seed=66), x_input(sshape, seed=67)]
    tv = torch.tensor([0.5005, 0.5005], dtype=torch.float32)
    arrays.update(corr_x=x, corr_score=score, corr_z0=zs[0], corr_z1=zs[1], corr_t=tv)
    for sname, sde in (("ve", M.VESDE(0.01, 50.0, 1000)), ("vp", M.VPSDE(0.1, 20.0, 1000))):
        for cname in ("langevin", "ald"):
            for n_inner in (1, 2):
                corr = M.get_corrector(cname)(sde, lambda a, b: score.clone(), 0.16, n_inner)
                it = iter([z.clone() for z in zs])
                with NoiseInjector(iter([(0, 0)] * 2), seed=0):
                    torch.randn_like = lambda _x, **kw: next(it)
                    xo, xm = corr.update_fn(x.clone(), tv)
                arrays[f"corr_{sname}_{cname}_n{n_inner}_x"] = xo
                arrays[f"corr_{sname}_{cname}_n{n_inner}_mean"] = xm
    npz(out, "switches", **arrays)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden"))
    ap.add_argument("--only", nargs="*", default=None)
    args = ap.parse_args()
    install_placeholders()
    sys.path.insert(0, args.ref)
    import diffusion_model_nemo.modules as M
    torch.set_num_threads(8)
    os.makedirs(args.out, exist_ok=True)
    jobs = dict(schedules=gen_schedules, blocks=gen_blocks, unet=gen_unet, p_sample=gen_psample,
                sde=gen_sde, train=gen_train, chain=gen_chain, ddim=gen_ddim, cond=gen_cond, bpd=gen_bpd,
                switches=gen_switches)
    for name, fn in jobs.items():
        if args.only and name not in args.only:
            continue
        print(f"[{name}]")
        fn(M, args.out)


if __name__ == "__main__":
    main()
