"""The CPU oracle (oracle/ref_cpu.py) against golden vectors captured from the reference
(tools/gen_golden.py). These pin the oracle; the GPU tests then compare the HIP path with the oracle."""
import numpy as np
import pytest
import torch

from oracle import philox, ref_cpu as R
from oracle.weights import make_param, make_state_dict, unet_param_shapes
from conftest import load_golden

torch.set_num_threads(8)


def T(a):
    return torch.from_numpy(np.asarray(a))


def close(a, b, atol, rtol=0.0):
    a = a.detach().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = np.asarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    err = np.abs(a.astype(np.float64) - b.astype(np.float64))
    tol = atol + rtol * np.abs(b)
    assert (err <= tol).all(), f"max err {err.max():.3e} (tol {atol:g}+{rtol:g}*|ref|)"


def named(prefix, shapes):
    return {k: T(make_param(prefix + k, s)) for k, s in shapes.items()}


def test_schedules_match_reference():
    g = load_golden("schedules")
    for sched in ("cosine", "linear", "quadratic", "sigmoid"):
        for Tn in (10, 1000):
            c = R.ddpm_constants(Tn, sched)
            for n in R.CONSTANT_NAMES:
                # same float32 operation order as the reference; libm/vector-width differences between host CPUs
                # move single results by one ulp
                np.testing.assert_allclose(c[n].numpy(), g[f"{sched}_{Tn}_{n}"], rtol=3e-7, atol=1e-9,
                                           err_msg=f"{sched} {Tn} {n}")
    c = R.ddpm_constants(50, "linear", beta_start=0.001, beta_end=0.05)
    for n in R.CONSTANT_NAMES:
        np.testing.assert_allclose(c[n].numpy(), g[f"linearcfg_50_{n}"], rtol=3e-7, atol=1e-9)


def _rb_shapes(cin, cout, with_t):
    s = {}
    if with_t:
        s["mlp.1.weight"], s["mlp.1.bias"] = (cout, 128), (cout,)
    for b, ci in (("block1", cin), ("block2", cout)):
        s[f"{b}.proj.weight"], s[f"{b}.proj.bias"] = (cout, ci, 3, 3), (cout,)
        s[f"{b}.norm.weight"], s[f"{b}.norm.bias"] = (cout,), (cout,)
    if cin != cout:
        s["res_conv.weight"], s["res_conv.bias"] = (cout, cin, 1, 1), (cout,)
    return s


@pytest.mark.parametrize("tag,cin,cout,groups,with_t", [
    ("rb_32_32", 32, 32, 8, True), ("rb_32_64", 32, 64, 8, True),
    ("rb_64_32_g4", 64, 32, 4, True), ("rb_32_32_not", 32, 32, 8, False)])
def test_resnet_block(tag, cin, cout, groups, with_t):
    g = load_golden("blocks")
    p = {"rb." + k: v for k, v in named(tag + ".", _rb_shapes(cin, cout, with_t)).items()}
    y = R.resnet_block(p, "rb", T(g[tag + "_x"]), T(g["temb"]) if with_t else None, groups)
    close(y, g[tag + "_y"], 2e-6, 1e-5)


def test_block():
    g = load_golden("blocks")
    p = {"b." + k: v for k, v in named("blk.", {"proj.weight": (64, 32, 3, 3), "proj.bias": (64,),
                                                 "norm.weight": (64,), "norm.bias": (64,)}).items()}
    close(R.block(p, "b", T(g["blk_x"]), 8), g["blk_y"], 2e-6, 1e-5)


@pytest.mark.parametrize("tag,c", [("la_32", 32), ("la_64", 64)])
def test_linear_attention(tag, c):
    g = load_golden("blocks")
    shapes = {"fn.norm.weight": (c,), "fn.norm.bias": (c,), "fn.fn.to_qkv.weight": (384, c, 1, 1),
              "fn.fn.to_out.0.weight": (c, 128, 1, 1), "fn.fn.to_out.0.bias": (c,),
              "fn.fn.to_out.1.weight": (c,), "fn.fn.to_out.1.bias": (c,)}
    p = {"a." + k: v for k, v in named(tag + ".", shapes).items()}
    x = T(g[tag + "_x"])
    close(R.linear_attention(p, "a.fn.fn", x), g[tag + "_bare_y"], 2e-6, 1e-5)
    close(R.prenorm_residual(p, "a", x, R.linear_attention), g[tag + "_y"], 2e-6, 1e-5)


@pytest.mark.parametrize("tag,c", [("at_64_4", 64), ("at_256_8", 256)])
def test_attention(tag, c):
    g = load_golden("blocks")
    shapes = {"fn.norm.weight": (c,), "fn.norm.bias": (c,), "fn.fn.to_qkv.weight": (384, c, 1, 1),
              "fn.fn.to_out.weight": (c, 128, 1, 1), "fn.fn.to_out.bias": (c,)}
    p = {"a." + k: v for k, v in named(tag + ".", shapes).items()}
    x = T(g[tag + "_x"])
    close(R.attention(p, "a.fn.fn", x), g[tag + "_bare_y"], 2e-6, 1e-5)
    close(R.prenorm_residual(p, "a", x, R.attention), g[tag + "_y"], 2e-6, 1e-5)


def test_down_up_sinusoid():
    import torch.nn.functional as F
    g = load_golden("blocks")
    w, b = T(make_param("down.weight", (32, 32, 4, 4))), T(make_param("down.bias", (32,)))
    close(F.conv2d(T(g["down_x"]), w, b, stride=2, padding=1), g["down_y"], 2e-6, 1e-5)
    w, b = T(make_param("ups.9.3.weight", (32, 32, 4, 4))), T(make_param("ups.9.3.bias", (32,)))
    close(F.conv_transpose2d(T(g["up_x"]), w, b, stride=2, padding=1), g["up_y"], 2e-6, 1e-5)
    close(R.sinusoidal_embedding(T(g["pe_t_long"]), 32), g["pe_long"], 1e-6, 0)
    close(R.sinusoidal_embedding(T(g["pe_t_float"]), 32), g["pe_float"], 1e-6, 0)


@pytest.mark.parametrize("tag,lv,groups,order", [
    ("ddpm", False, 8, "bn_act_conv"), ("improved", True, 8, "bn_act_conv"),
    ("sde_g4", False, 4, "bn_act_conv"), ("convbnact", False, 8, "conv_bn_act")])
def test_unet_forward(tag, lv, groups, order):
    g = load_golden("unet")
    p = R.to_torch(make_state_dict(unet_param_shapes(learned_variance=lv, resnet_block_order=order)))
    x = T(g["x"])
    for t in (0, 500, 999):
        y = R.unet_forward(p, x, torch.full((2,), t, dtype=torch.long), groups=groups)
        close(y, g[f"{tag}_t{t}"], 1e-5, 1e-5)
    close(R.unet_forward(p, x, torch.tensor([3, 977]), groups=groups), g[f"{tag}_tmixed"], 1e-5, 1e-5)
    close(R.unet_forward(p, x, T(g["t_float"]), groups=groups), g[f"{tag}_tfloat"], 1e-5, 1e-5)


def test_unet_forward_64():
    g = load_golden("unet")
    p = R.to_torch(make_state_dict(unet_param_shapes()))
    close(R.unet_forward(p, T(g["x64"]), torch.tensor([250])), g["ddpm64_t250"], 1e-5, 1e-5)


def test_p_sample_updates():
    g = load_golden("p_sample")
    x, eps, eps6, z = (T(g[k]) for k in ("x", "eps", "eps6", "z"))
    for sched in ("cosine", "linear"):
        c = R.ddpm_constants(1000, sched)
        for tt in ([999, 999, 999], [1, 1, 1], [0, 0, 0], [0, 17, 640]):
            t = torch.tensor(tt)
            tag = f"{sched}_t{'_'.join(map(str, tt))}"
            close(R.p_sample_update(c, x, t, eps, z), g["fixed_" + tag], 1e-6, 1e-6)
            close(R.p_sample_update(c, x, t, eps6, z, learned_variance=True), g["learned_" + tag], 1e-6, 1e-6)
        close(R.q_sample(c, x, torch.tensor([0, 499, 999]), z), g[f"qsample_{sched}"], 1e-7, 1e-6)
    c = R.ddpm_constants(1000, "cosine")
    close(R.p_sample_update(c, x, torch.tensor([5, 50, 500]), eps, z, objective="pred_x0"),
          g["predx0_cosine"], 1e-6, 1e-6)


def test_sde_units():
    g = load_golden("sde_units")
    ve, vp = R.VESDE(0.01, 50.0, 1000), R.VPSDE(0.1, 20.0, 1000)
    np.testing.assert_allclose(ve.discrete_sigmas.numpy(), g["ve_discrete_sigmas"], rtol=3e-7)
    np.testing.assert_allclose(vp.betas.numpy(), g["vp_betas"], rtol=3e-7)
    np.testing.assert_allclose(vp.sqrt_1m_alphas_cumprod.numpy(), g["vp_sqrt_1m_alphas_cumprod"], rtol=3e-7)
    tt, xx = T(g["tt"]), T(g["xx"])
    for name, sde in (("ve", ve), ("vp", vp)):
        d, gg = sde.sde(xx, tt)
        m, s = sde.marginal_prob(xx, tt)
        f, G = sde.discretize(xx, tt)
        for k, v in (("drift", d), ("diffusion", gg), ("mean", m), ("std", s), ("f", f), ("G", G)):
            close(v, g[f"{name}_{k}"], 1e-7, 1e-6)
    x, z, z2, score, t = (T(g[k]) for k in ("x", "z", "z2", "score", "t"))
    sfn = lambda a, b: score.clone()  # noqa: E731
    for sname, sde in (("ve", ve), ("vp", vp)):
        for pname in ("euler_maruyama", "reverse_diffusion", "ancestral_sampling"):
            if sname == "vp" and pname == "ancestral_sampling":
                with pytest.raises(NotImplementedError):
                    R.predictor_step(pname, sde, sfn, x, t, z)
                continue
            xo, xm = R.predictor_step(pname, sde, sfn, x, t, z)
            close(xo, g[f"{sname}_{pname}_x"], 1e-6, 1e-6)
            close(xm, g[f"{sname}_{pname}_mean"], 1e-6, 1e-6)
        for cname in ("langevin", "ald"):
            zs = [z, z2]
            xo, xm = R.corrector_step(cname, sde, sfn, x, t, lambda k: zs[k], 0.16, 2)
            close(xo, g[f"{sname}_{cname}_x"], 1e-6, 2e-6)
            close(xm, g[f"{sname}_{cname}_mean"], 1e-6, 2e-6)


def test_score_function_adapters():
    g = load_golden("sde_units")
    p = R.to_torch(make_state_dict(unet_param_shapes()))
    model = lambda x, t: R.unet_forward(p, x, t, groups=4)  # noqa: E731
    x32, t = T(g["x32"]), T(g["t"])
    for sname, sde in (("ve", R.VESDE(0.01, 50.0, 1000)), ("vp", R.VPSDE(0.1, 20.0, 1000))):
        for cont in (True, False):
            s = R.score_fn_for(sde, model, cont)(x32, t.clone())
            ref = g[f"score_{sname}_{'cont' if cont else 'disc'}"]
            close(s, ref, 2e-5 * max(1.0, float(np.abs(ref).max())), 1e-5)


@pytest.mark.parametrize("tag,kind,pred,corr,n_steps", [
    ("ve_rd_langevin", "ve", "reverse_diffusion", "langevin", 1),
    ("ve_em_none", "ve", "euler_maruyama", None, 1),
    ("vp_em_none", "vp", "euler_maruyama", None, 1),
    ("vp_rd_langevin2", "vp", "reverse_diffusion", "langevin", 2),
    ("ve_anc_ald", "ve", "ancestral_sampling", "ald", 1)])
def test_pc_chain_20(tag, kind, pred, corr, n_steps):
    g = load_golden("chain_sde_N20_B2_32")
    N, shape, seed = 20, (2, 3, 32, 32), 5
    sde = R.VESDE(0.01, 50.0, N) if kind == "ve" else R.VPSDE(0.1, 20.0, N)
    p = R.to_torch(make_state_dict(unet_param_shapes()))
    model = lambda x, t: R.unet_forward(p, x, t, groups=4)  # noqa: E731
    noise = lambda step, stream: T(philox.normal_like(shape, seed, step, stream))  # noqa: E731
    with torch.no_grad():
        out = R.pc_sample_chain(model, sde, shape, noise, pred, corr, 0.16, n_steps, keep=(0, 9, 19))
    for i in (0, 9, 19):
        ref = g[f"{tag}_mean_{i}"]
        scale = max(1.0, float(np.abs(ref).max()))
        close(out[i], ref, 5e-5 * scale, 0)
    assert int(g[f"{tag}_nfe"]) == N * (n_steps + 1)


def test_training_step_loss_and_grads():
    g = load_golden("train_step")
    p = R.to_torch(make_state_dict(unet_param_shapes()))
    for v in p.values():
        v.requires_grad_(True)
    c = R.ddpm_constants(1000, "cosine")
    x0, noise, t = T(g["x0"]), T(g["noise"]), T(g["t"])
    close(R.q_sample(c, x0, t, noise), g["x_t"], 1e-7, 1e-6)
    loss = R.ddpm_training_loss(lambda x, tt: R.unet_forward(p, x, tt), c, x0, t, noise)
    loss.backward()
    close(loss, g["loss"], 1e-6, 1e-5)
    gn = torch.sqrt(sum((v.grad.double() ** 2).sum() for v in p.values())).float()
    close(gn, g["grad_norm"], 1e-5, 1e-4)
    close(p["init_conv.bias"].grad, g["grad_init_conv_bias"], 1e-6, 1e-4)
    close(p["final_conv.3.bias"].grad, g["grad_final_bias"], 1e-6, 1e-4)


def test_ddpm_chain_1000_steps():
    """Full 1000-step ancestral chain, B=2 @ 32x32, against the reference's own run."""
    g = load_golden("chain_ddpm_T1000_B2_32")
    T_, shape, seed = 1000, (2, 3, 32, 32), 0
    p = R.to_torch(make_state_dict(unet_param_shapes()))
    c = R.ddpm_constants(T_, "cosine")
    noise = lambda step: T(philox.normal_like(shape, seed, step, 0 if step == T_ else 1))  # noqa: E731
    with torch.no_grad():
        out = R.ddpm_sample_chain(lambda x, t: R.unet_forward(p, x, t), c, shape, noise, keep=(999, 900, 500, 100, 0))
    np.testing.assert_allclose(out["x_T"].numpy(), g["x_T"], rtol=0, atol=1e-6)
    for i in (999, 900, 500, 100, 0):
        close(out[i], g[f"x_{i}"], 1e-4, 0)


def test_improved_chain_100_steps():
    g = load_golden("chain_improved_T100_B2_32")
    T_, shape, seed = 100, (2, 3, 32, 32), 1
    p = R.to_torch(make_state_dict(unet_param_shapes(learned_variance=True)))
    c = R.ddpm_constants(T_, "cosine")
    noise = lambda step: T(philox.normal_like(shape, seed, step, 0 if step == T_ else 1))  # noqa: E731
    with torch.no_grad():
        out = R.ddpm_sample_chain(lambda x, t: R.unet_forward(p, x, t), c, shape, noise,
                                  learned_variance=True, keep=(99, 50, 10, 0))
    for i in (99, 50, 10, 0):
        close(out[i], g[f"x_{i}"], 1e-4, 0)


def test_philox_known_answer_and_shard_invariance():
    # Known-answer vector of Philox4x32-10 (Random123 kat_vectors: counter = key = 0)
    r = philox.philox4x32_10(np.uint32(0), np.uint32(0), np.uint32(0), np.uint32(0), 0, 0)
    assert [int(v) for v in r] == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    r = philox.philox4x32_10(np.uint32(0xFFFFFFFF), np.uint32(0xFFFFFFFF), np.uint32(0xFFFFFFFF),
                             np.uint32(0xFFFFFFFF), 0xFFFFFFFF, 0xFFFFFFFF)
    assert [int(v) for v in r] == [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]
    full = philox.normal_like((4, 3, 8, 8), seed=3, step=17, stream=1)
    part = philox.normal_like((2, 3, 8, 8), seed=3, step=17, stream=1, image_offset=2)
    np.testing.assert_array_equal(full[2:], part)
    big = philox.normal_like((64, 3, 32, 32), seed=9, step=1, stream=1)
    assert abs(float(big.mean())) < 0.01 and abs(float(big.std()) - 1.0) < 0.01


def test_ddim_chains_match_reference(golden):
    """GeneralizedGaussianDiffusion.p_sample_loop (generalized_gaussian_diffusion.py:100-137), three variants."""
    g = golden("chain_ddim_T1000_B2_32")
    p = R.to_torch(make_state_dict(unet_param_shapes()))
    betas = R.beta_schedule("cosine", 1000)
    shape = (2, 3, 32, 32)

    def noise_fn(step):
        return torch.from_numpy(philox.normal_like(shape, 3, step, 0 if step == 1000 else 1))

    for tag in ("eta0_s20", "eta05_s20", "x0_s10"):
        steps, eta, x0 = g[f"{tag}_meta"]
        with torch.no_grad():
            xs = R.ddim_sample_chain(lambda x, t: R.unet_forward(p, x, t), betas, shape, noise_fn, eta=float(eta),
                                     ddim_timesteps=int(steps), objective="pred_x0" if x0 else "pred_noise")
        want = torch.from_numpy(g[f"{tag}_xs"])
        assert len(xs) == want.shape[0]
        err = max(float((a - b).abs().max()) for a, b in zip(xs, want))
        assert err <= 2e-5, (tag, err)


def test_class_conditional_unet_matches_reference(golden):
    """unet.py:131-141: the class embedding row is added after init_conv; classes=None means the extra 'null' row."""
    g = golden("unet_cond")
    p = R.to_torch(make_state_dict(unet_param_shapes(num_classes=10)))
    x, t, labels = T(g["x"]), T(g["t"]), T(g["labels"])
    with torch.no_grad():
        close(R.unet_forward(p, x, t, classes=labels), g["y_labels"], atol=2e-5)
        close(R.unet_forward(p, x, t), g["y_default"], atol=2e-5)


def test_bits_per_dimension_matches_reference(golden):
    """abstract_diffusion_model.py:137-197 driven over the reference's samplers and VariationalBoundLoss."""
    g = golden("bpd_T20_B2_32")
    x0 = T(g["x0"])
    c = R.ddpm_constants(20, "cosine")

    def noise_fn(t):
        return torch.from_numpy(philox.normal_like(tuple(x0.shape), 4, t, 3))

    for tag, lv in (("fixed", False), ("learned", True)):
        p = R.to_torch(make_state_dict(unet_param_shapes(learned_variance=lv)))
        with torch.no_grad():
            r = R.bits_per_dimension(lambda x, t: R.unet_forward(p, x, t), c, x0, noise_fn, learned_variance=lv)
        close(r["terms_bpd"], g[f"{tag}_terms"], atol=2e-5, rtol=2e-5)
        close(r["prior_bpd"], g[f"{tag}_prior"], atol=1e-6, rtol=1e-6)
        close(r["total_bpd"], g[f"{tag}_total"], atol=1e-4, rtol=2e-5)


def test_interpolation_matches_reference(golden):
    """GaussianDiffusion.interpolate (gaussian_diffusion.py:195-218) and GeneralizedGaussianDiffusion.interpolate
    (generalized_gaussian_diffusion.py:138-140)."""
    g = golden("switches")
    p = R.to_torch(make_state_dict(unet_param_shapes()))
    model = lambda x, t: R.unet_forward(p, x, t)  # noqa: E731
    Tn, t_mix, lam = g["interp_meta"]
    Tn, t_mix, seed, shape = int(Tn), int(t_mix), 6, (2, 3, 32, 32)
    c = R.ddpm_constants(Tn, "cosine")
    qn = [T(philox.normal_like(shape, seed, t_mix, s)) for s in (4, 5)]
    with torch.no_grad():
        xs = R.ddpm_interpolate(model, c, T(g["interp_x1"]), T(g["interp_x2"]), t_mix, float(lam), qn,
                                lambda i: T(philox.normal_like(shape, seed, i, 1)))
    assert len(xs) == t_mix
    for k, i in enumerate((0, 24, 49)):
        close(xs[i], g["interp_xs"][k], 2e-5)
    with pytest.raises(ValueError):
        R.ddpm_interpolate(model, c, T(g["interp_x1"]), T(g["interp_x2"]), Tn, 0.5, qn, None)
    betas = R.beta_schedule("cosine", Tn)
    with torch.no_grad():
        xs = R.ddim_sample_chain(model, betas, shape, lambda i: T(philox.normal_like(shape, seed, i, 1)), eta=0.0,
                                 ddim_timesteps=10, img=T(g["ddim_interp_latent"]))
    want = g["ddim_interp_xs"]
    assert len(xs) == want.shape[0] == 10
    for i in range(10):
        close(xs[i], want[i], 2e-5)


@pytest.mark.parametrize("tag,kind,pred,corr,kw", [
    ("vp_rd_langevin_disc", "vp", "reverse_diffusion", "langevin", dict(continuous=False)),
    ("ve_em_none_disc", "ve", "euler_maruyama", None, dict(continuous=False)),
    ("ve_rd_langevin_pf", "ve", "reverse_diffusion", "langevin", dict(probability_flow=True)),
    ("vp_rd_none_pf", "vp", "reverse_diffusion", None, dict(probability_flow=True)),
    ("ve_rd_langevin_raw", "ve", "reverse_diffusion", "langevin", dict(denoise=False)),
    ("vp_none_langevin", "vp", None, "langevin", dict())])
def test_pc_chain_options_match_reference(golden, tag, kind, pred, corr, kw):
    """PredictorCorrectorSampler with continuous=False, probability_flow=True, denoise=False and without a predictor
    (predictor_corrector_sampler.py:19-29,97-113)."""
    g = golden("switches")
    N, shape, seed = 20, (2, 3, 32, 32), 7
    sde = R.VESDE(0.01, 50.0, N) if kind == "ve" else R.VPSDE(0.1, 20.0, N)
    p = R.to_torch(make_state_dict(unet_param_shapes()))
    model = lambda x, t: R.unet_forward(p, x, t, groups=4)  # noqa: E731
    noise = lambda step, stream: T(philox.normal_like(shape, seed, step, stream))  # noqa: E731
    with torch.no_grad():
        out = R.pc_sample_chain(model, sde, shape, noise, pred, corr, 0.16, 1, keep=(0, 9, 19), **kw)
    for i in (0, 9, 19):
        ref = g[f"{tag}_img_{i}"]
        close(out[i], ref, 5e-5 * max(1.0, float(np.abs(ref).max())), 0)
    assert int(g[f"{tag}_nfe"]) == N * 2
    with pytest.raises(TypeError):   # the reference subscripts the float diffusion 0.0 (euler_maruyama_predictor.py:16)
        R.pc_sample_chain(model, sde, shape, noise, "euler_maruyama", None, probability_flow=True, n_iter=1)


def test_corrector_steps_at_one_time_match_reference(golden):
    """Langevin / annealed-Langevin inner steps with one time for the whole batch (langevin_corrector.py:15-35,
    annealed_langevin_dynamics_corrector.py:21-41)."""
    g = golden("switches")
    x, score, t = T(g["corr_x"]), T(g["corr_score"]), T(g["corr_t"])
    zs = [T(g["corr_z0"]), T(g["corr_z1"])]
    for sname, sde in (("ve", R.VESDE(0.01, 50.0, 1000)), ("vp", R.VPSDE(0.1, 20.0, 1000))):
        for cname in ("langevin", "ald"):
            for n in (1, 2):
                xo, xm = R.corrector_step(cname, sde, lambda a, b: score.clone(), x, t, lambda k: zs[k], 0.16, n)
                close(xo, g[f"corr_{sname}_{cname}_n{n}_x"], 1e-6, 2e-6)
                close(xm, g[f"corr_{sname}_{cname}_n{n}_mean"], 1e-6, 2e-6)
