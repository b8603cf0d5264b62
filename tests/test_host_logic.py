"""Host-side logic of the product package (no GPU needed): schedule constants, step tables, SDE coefficient
tables, config loading / `_target_` instantiation, state_dict naming, and the C-ABI library's exported symbols."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT, load_golden
from oracle import ref_cpu as R
from oracle.weights import unet_param_shapes


def test_schedule_constants_match_reference():
    from diffusion_model_nemo_amd.modules import GaussianDiffusion
    g = load_golden("schedules")
    for sched in ("cosine", "linear", "quadratic", "sigmoid"):
        for Tn in (10, 1000):
            gd = GaussianDiffusion(timesteps=Tn, schedule_name=sched)
            for n in R.CONSTANT_NAMES:
                np.testing.assert_allclose(getattr(gd, n).numpy(), g[f"{sched}_{Tn}_{n}"], rtol=3e-7, atol=1e-9)
    gd = GaussianDiffusion(timesteps=50, schedule_name="linear",
                           schedule_cfg={"linear": {"beta_start": 0.001, "beta_end": 0.05}})
    for n in R.CONSTANT_NAMES:
        np.testing.assert_allclose(getattr(gd, n).numpy(), g[f"linearcfg_50_{n}"], rtol=3e-7, atol=1e-9)
    with pytest.raises(AssertionError):
        GaussianDiffusion(timesteps=10, schedule_name="nope")


def test_ddpm_step_table_rows_follow_execution_order():
    from diffusion_model_nemo_amd.modules import GaussianDiffusion
    gd = GaussianDiffusion(timesteps=10, schedule_name="linear")
    tab = gd.step_table("cpu")
    assert tab.shape == (10, 12)
    assert tab[:, 0].tolist() == list(range(9, -1, -1))
    assert torch.equal(tab[:, 3], gd.posterior_mean_coef1.flip(0))
    assert tab[-1, 7] == 0 and bool((tab[:-1, 7] == 1).all())
    assert torch.equal(tab[:, 6], torch.log(gd.betas).flip(0))


def _affine_from_oracle(name, sde_o, t, x, z, score):
    sfn = lambda a, b: score  # noqa: E731
    return R.predictor_step(name, sde_o, sfn, x, t, z)


@pytest.mark.parametrize("kind", ["ve", "vp"])
@pytest.mark.parametrize("pred", ["euler_maruyama", "reverse_diffusion", "ancestral_sampling"])
def test_sde_step_table_reproduces_reference_predictors(kind, pred):
    """The affine coefficients (a, b, c) of the step table, applied on the CPU, equal the reference predictor
    updates (golden) for both per-sample times."""
    from diffusion_model_nemo_amd.modules import VESDE, VPSDE, PredictorCorrectorSampler
    g = load_golden("sde_units")
    sde = VESDE(0.01, 50.0, 1000) if kind == "ve" else VPSDE(0.1, 20.0, 1000)
    s = PredictorCorrectorSampler(predictor=pred, corrector=None, snr=0.16)
    s.update_sde(sde)
    if kind == "vp" and pred == "ancestral_sampling":
        with pytest.raises(AttributeError):
            s.step_table()
        return
    t = torch.from_numpy(g["t"])
    tab = s.step_table(times=t)
    x, z, score = (torch.from_numpy(g[k]) for k in ("x", "z", "score"))
    a, b, c, ss = (tab[:, i][:, None, None, None] for i in (1, 2, 3, 4))
    net = score / ss
    mean = a * x + b * net
    xo = mean + c * z
    sc = max(1.0, float(np.abs(g[f"{kind}_{pred}_x"]).max()))
    assert float((mean - torch.from_numpy(g[f"{kind}_{pred}_mean"])).abs().max()) <= 3e-6 * sc
    assert float((xo - torch.from_numpy(g[f"{kind}_{pred}_x"])).abs().max()) <= 3e-6 * sc


def test_sde_tables_labels_and_errors():
    from diffusion_model_nemo_amd.modules import VESDE, VPSDE, PredictorCorrectorSampler, subVPSDE
    g = load_golden("sde_units")
    ve, vp = VESDE(0.01, 50.0, 1000), VPSDE(0.1, 20.0, 1000)
    np.testing.assert_allclose(ve.discrete_sigmas.numpy(), g["ve_discrete_sigmas"], rtol=3e-7)
    np.testing.assert_allclose(vp.sqrt_1m_alphas_cumprod.numpy(), g["vp_sqrt_1m_alphas_cumprod"], rtol=3e-7)
    tt, xx = torch.from_numpy(g["tt"]), torch.from_numpy(g["xx"])
    for name, sde in (("ve", ve), ("vp", vp)):
        d, gg = sde.sde(xx, tt)
        m, sd = sde.marginal_prob(xx, tt)
        f, G = sde.discretize(xx, tt)
        for k, v in (("drift", d), ("diffusion", gg), ("mean", m), ("std", sd), ("f", f), ("G", G)):
            np.testing.assert_allclose(v.numpy(), g[f"{name}_{k}"], rtol=2e-6, atol=5e-7)
    s = PredictorCorrectorSampler(predictor="euler_maruyama", corrector="langevin", snr=0.16)
    with pytest.raises(ValueError):
        s.step_table()
    s.update_sde(ve)
    tab = s.step_table()
    assert tab.shape == (1000, 12)
    assert abs(float(tab[0, 0]) - 50.0) < 1e-3 and abs(float(tab[-1, 0]) - 0.01) < 1e-4   # label = sigma(t)
    s.update_sde(vp)
    tab = s.step_table()
    assert abs(float(tab[0, 0]) - 999.0) < 1e-3 and float(tab[0, 4]) < 0                   # label = 999 t, score = -net/std
    s.update_sde(subVPSDE())
    with pytest.raises(AttributeError):
        s.step_table()
    with pytest.raises(ValueError):
        PredictorCorrectorSampler(predictor="bogus", corrector=None, snr=0.1).update_sde(ve) or \
            PredictorCorrectorSampler(predictor="bogus", corrector=None, snr=0.1)._validate()


REFERENCE_STYLE_CFG = """
name: "DDPM-UNet"
model:
  image_size: ???
  timesteps: 1000
  channels: 3
  num_classes: null
  diffusion_model:
    _target_: diffusion_model_nemo.modules.Unet
    input_dim: ${model.image_size}
    dim: 32
    channels: ${model.channels}
    num_classes: ${model.num_classes}
    dim_mults: [1, 2, 4, 8]
    with_time_emb: True
    resnet_block_order: 'bn_act_conv'
    resnet_block_groups: 8
    use_convnext: False
    convnext_mult: 2
    out_dim: null
    dropout: 0.0
  sampler:
    _target_: diffusion_model_nemo.modules.GaussianDiffusion
    timesteps: ${model.timesteps}
    schedule_name: "cosine"
    schedule_cfg:
      cosine: {s: 0.008, min_clip: 0.0001, max_clip: 0.999}
      linear: {beta_start: 0.0001, beta_end: 0.02}
  loss:
    _target_: diffusion_model_nemo.loss.DiffusionLoss
    loss_type: "l2"
    reduction: "mean"
exp_manager:
  name: ${name}
"""


def test_config_loading_and_target_instantiation(tmp_path):
    from diffusion_model_nemo_amd.config import MissingMandatoryValue, load_config
    from diffusion_model_nemo_amd.models import DDPM
    from diffusion_model_nemo_amd.modules import GaussianDiffusion, Unet
    p = tmp_path / "cfg.yaml"
    p.write_text(REFERENCE_STYLE_CFG)
    cfg = load_config(str(p))
    with pytest.raises(MissingMandatoryValue):
        _ = cfg.model.image_size
    cfg = load_config(str(p), ["model.image_size=64", "model.timesteps=250"])
    assert cfg.model.diffusion_model.input_dim == 64 and cfg.model.sampler.timesteps == 250
    assert cfg.exp_manager.name == "DDPM-UNet"
    m = DDPM.from_config_file(str(p), ["model.image_size=64"])
    assert isinstance(m.diffusion_model, Unet) and isinstance(m.sampler, GaussianDiffusion)
    assert m.sampler.timesteps == 1000 and m.image_size == 64
    ref = {k: tuple(v) for k, v in unet_param_shapes().items()}
    mine = {k: tuple(v.shape) for k, v in m.diffusion_model.state_dict().items()}
    assert mine == ref
    assert all(k.startswith(("diffusion_model.",)) for k in m.state_dict())
    # runtime sampler swap keeps the network (abstract_diffusion_model.py:130-135)
    m.change_sampler({"_target_": "diffusion_model_nemo.modules.LearnedGaussianDiffusion", "timesteps": 10,
                      "schedule_name": "linear"})
    assert type(m.sampler).__name__ == "LearnedGaussianDiffusion"


def test_unsupported_configurations_fail_loudly():
    from diffusion_model_nemo_amd.modules import Unet
    with pytest.raises(NotImplementedError):
        Unet(input_dim=32, dim=32, use_convnext=True)
    with pytest.raises(NotImplementedError):
        Unet(input_dim=32, dim=32, use_convnext=False, dropout=0.1)
    uc = Unet(input_dim=32, dim=32, use_convnext=False, num_classes=10)
    assert tuple(uc.state_dict()["class_embed.weight"].shape) == (11, 32)
    u = Unet(input_dim=32, dim=32, use_convnext=False)
    with pytest.raises(RuntimeError):
        u(torch.zeros(1, 3, 32, 32), torch.zeros(1, dtype=torch.long))   # no CPU path


def test_library_loads_and_exports_every_declared_symbol():
    from diffusion_model_nemo_amd import _native
    header = open(os.path.join(ROOT, "include", "dmn.h")).read()
    declared = set(re.findall(r"\b(dmn_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    assert declared == set(_native.SIGNATURES), declared ^ set(_native.SIGNATURES)
    lib = _native.lib()
    for name in declared:
        assert hasattr(lib, name), name
    assert b"gfx950" in lib.dmn_version()
    # structure sizes agree with the C declarations (int fields only / pointer + int + 4 x int64)
    assert ctypes.sizeof(_native.DmnUnetCfg) == 4 * (11 + 8)
    assert ctypes.sizeof(_native.DmnTensorDesc) == 8 + 8 + 8 + 32


def test_shipped_library_holds_no_packed_fp32_instruction():
    """The gfx950 code objects of libdmn_hip.so are disassembled here: no v_pk_add/mul/fma_f32 (with them, kernels of this
    library gave wrong sums in lanes 48..63 beside a foreign stream on MI355X; build.py switches them off and checks the
    same thing), and the convolution / attention kernels really are matrix-core code."""
    from diffusion_model_nemo_amd import _native, isa_check
    census = isa_check.assert_no_packed_fp32(_native.LIB_PATH)
    assert len(census) >= 8, sorted(census)                      # one code object per .hip translation unit
    assert sum(c["forbidden"] for c in census.values()) == 0
    assert sum(c["mfma"] for c in census.values()) > 10000, census
    # the pattern itself: it must catch the instructions it is there for
    assert isa_check.FORBIDDEN.search("\tv_pk_fma_f32 v[4:5], v[18:19], v[32:33], v[4:5] op_sel_hi:[1,0,1]")
    assert isa_check.FORBIDDEN.search("v_pk_mul_f32 v[16:17], v[10:11], v[16:17]")
    assert not isa_check.FORBIDDEN.search("v_pk_add_f16 v1, v2, v3") and not isa_check.FORBIDDEN.search("v_fma_f32 v1, v2, v3, v4")


def test_kernels_that_wait_inside_a_launch_fit_two_workgroups_per_cu():
    """Workgroups that exchange GroupNorm partial sums inside a launch keep their place on the CU while they wait; the
    residency rule (two exchanging launches of <= 256 workgroups resident together) needs two of them per CU, i.e. at
    most 256 of a SIMD's 512 registers per wave. Read from the code-object metadata of the shipped library."""
    from diffusion_model_nemo_amd import _native, isa_check
    regs = isa_check.assert_exchanging_kernels_fit_half_a_cu(_native.LIB_PATH)
    assert len(regs) >= 12, sorted(regs)                                 # bf16 + fp32, conv tiles + attention widths
    assert any("Li32ELi16ELi32ELi4ELi1E" in k for k in regs)             # the 32x16-pixel tile among them
    assert max(regs.values()) <= 256
    # kernels that never wait (one workgroup per image: TAIL && ONE) are not held to it
    assert not isa_check.EXCHANGING.search("_ZN12_GLOBAL__N_119la_fused_out_kernelIDF16bLi256ELb1ELb1EEEv16LinAttnFusedArgs")


def test_nemo_archive_round_trip(tmp_path):
    """`.nemo` layout (tar: model_config.yaml + model_weights.ckpt) written and read back: same config, same weights,
    and the reader also accepts the gzip-compressed form without the './' member prefix."""
    import tarfile
    from diffusion_model_nemo_amd.config import load_config
    from diffusion_model_nemo_amd.models import DDPM
    p = tmp_path / "cfg.yaml"
    p.write_text(REFERENCE_STYLE_CFG)
    m = DDPM(load_config(str(p), ["model.image_size=32", "model.timesteps=50"]).model)
    with torch.no_grad():
        for i, prm in enumerate(m.parameters()):
            prm.add_(0.001 * (i % 7))
    path = str(tmp_path / "DDPM.nemo")
    m.save_to(path)
    with tarfile.open(path) as tar:
        assert sorted(n.lstrip("./") for n in tar.getnames()) == ["model_config.yaml", "model_weights.ckpt"]
    r = DDPM.restore_from(path)
    assert r.timesteps == 50 and r.image_size == 32 and r.sampler.timesteps == 50
    a, b = m.state_dict(), r.state_dict()
    assert a.keys() == b.keys() and all(torch.equal(a[k], b[k]) for k in a)
    gz = str(tmp_path / "DDPM_gz.nemo")
    with tarfile.open(path) as src, tarfile.open(gz, "w:gz") as dst:
        for mem in src.getmembers():
            data = src.extractfile(mem).read()
            mem.name = mem.name.lstrip("./")
            import io
            dst.addfile(mem, io.BytesIO(data))
    r2 = DDPM.restore_from(gz)
    assert all(torch.equal(a[k], r2.state_dict()[k]) for k in a)
    with pytest.raises(FileNotFoundError):
        with tarfile.open(str(tmp_path / "bad.nemo"), "w:") as t:
            pass
        DDPM.restore_from(str(tmp_path / "bad.nemo"))
    # Optimiser state travels as a member of its own: `model_weights.ckpt` stays the pure state_dict the reference
    # restores with a strict load_state_dict; a restore followed by a save without training in between keeps it; the
    # entries that earlier archives of this package kept inside the weights file are still read.
    opt = {"exp_avg": torch.arange(6.0), "exp_avg_sq": torch.ones(6), "step_count": torch.tensor(3), "layout": "digest"}
    r._pending_optimizer_state = opt
    p2 = str(tmp_path / "with_opt.nemo")
    r.save_to(p2)
    with tarfile.open(p2) as tar:
        names = sorted(n.lstrip("./") for n in tar.getnames())
        assert names == ["model_config.yaml", "model_weights.ckpt", "optimizer_state.ckpt"]
        import io
        weights = torch.load(io.BytesIO(tar.extractfile("./model_weights.ckpt").read()), weights_only=True)
    assert set(weights) == set(m.state_dict())
    r3 = DDPM.restore_from(p2)
    got = r3._pending_optimizer_state
    assert torch.equal(got["exp_avg"], opt["exp_avg"]) and int(got["step_count"]) == 3 and got["layout"] == "digest"
    p3 = str(tmp_path / "again.nemo")
    r3.save_to(p3)
    assert torch.equal(DDPM.restore_from(p3)._pending_optimizer_state["exp_avg_sq"], opt["exp_avg_sq"])
    legacy = str(tmp_path / "legacy.nemo")
    with tarfile.open(path) as src, tarfile.open(legacy, "w:") as dst:
        for mem in src.getmembers():
            data = src.extractfile(mem).read()
            if mem.name.endswith("model_weights.ckpt"):
                sd = torch.load(io.BytesIO(data), weights_only=True)
                sd.update({"__optimizer__." + k: v for k, v in opt.items() if k != "layout"})
                buf = io.BytesIO()
                torch.save(sd, buf)
                data = buf.getvalue()
                mem.size = len(data)
            dst.addfile(mem, io.BytesIO(data))
    assert int(DDPM.restore_from(legacy)._pending_optimizer_state["step_count"]) == 3


def test_pixel_values_adapter_round_trip():
    """Image column -> `pixel_values` in [-1, 1] (CHW float32), image column dropped; flip only in train mode; reverse
    transform gives the uint8 HWC image back (data/hf_vision_data.py:14-54,77-97)."""
    from diffusion_model_nemo_amd import data as D
    rng = np.random.default_rng(0)
    imgs = [rng.integers(0, 256, size=(12, 10, 3), dtype=np.uint8) for _ in range(4)]
    ex = D.test_transforms({"img": list(imgs), "label": [1, 2, 3, 4]})
    assert "img" not in ex and len(ex["pixel_values"]) == 4
    pv = ex["pixel_values"][0]
    assert pv.dtype == torch.float32 and tuple(pv.shape) == (3, 12, 10)
    np.testing.assert_allclose(pv.numpy(), imgs[0].transpose(2, 0, 1) / 255.0 * 2 - 1, atol=1e-6)
    back = D.get_reverse_transform(uint=True, as_pil=False)(pv)
    assert back.dtype == np.uint8 and np.abs(back.astype(int) - imgs[0].astype(int)).max() <= 1
    # grey-scale (HW) images get a channel axis; train mode flips some images left-right and nothing else
    g = torch.Generator().manual_seed(3)
    grey = [rng.integers(0, 256, size=(8, 8), dtype=np.uint8) for _ in range(16)]
    tr = D.train_transforms({"image": list(grey)}, generator=g)["pixel_values"]
    flipped = 0
    for t, src in zip(tr, grey):
        ref = torch.from_numpy(src).float()[None] / 255.0 * 2 - 1
        assert tuple(t.shape) == (1, 8, 8)
        if torch.equal(t, ref):
            continue
        assert torch.equal(t, ref.flip(-1))
        flipped += 1
    assert 0 < flipped < 16
    with pytest.raises(ValueError):
        D.test_transforms({"picture": []})
    ds = D.PixelValuesDataset({"img": imgs, "label": [0, 1, 2, 3]}, mode="test")
    batch = D.collate_pixel_values([ds[i] for i in range(len(ds))])
    assert tuple(batch["pixel_values"].shape) == (4, 3, 12, 10) and batch["label"].tolist() == [0, 1, 2, 3]
    assert float(batch["pixel_values"].min()) >= -1.0 and float(batch["pixel_values"].max()) <= 1.0
    # resize (shorter side) + centre crop to a square network input
    t = D.get_transform(8, center_crop=True)(imgs[0])
    assert tuple(t.shape) == (3, 8, 8)


def test_training_parameter_table_matches_reference_state_dict_and_lr_schedule():
    """The flat parameter buffer of the training step holds exactly the reference's state_dict entries (names and
    PyTorch shapes; diffusion_model_nemo/modules/unet.py:41-120), 16-byte aligned and non-overlapping; creating the
    trainer handle and sizing its workspace need no GPU."""
    import ctypes as C
    from diffusion_model_nemo_amd import _native as N
    from diffusion_model_nemo_amd.training import cosine_annealing_lr, flat_layout
    from oracle.weights import unet_param_shapes
    for dim, mults, order in ((32, (1, 2, 4, 8), "bn_act_conv"), (64, (1, 2, 4), "conv_bn_act")):
        cfg = N.DmnUnetCfg()
        cfg.dim, cfg.n_mults, cfg.channels, cfg.out_dim, cfg.groups = dim, len(mults), 3, 3, 8
        for i, m in enumerate(mults):
            cfg.dim_mults[i] = m
        cfg.heads, cfg.dim_head, cfg.with_time_emb = 4, 32, 1
        cfg.tail_norm = 1 if order == "bn_act_conv" else 0
        layout, n = flat_layout(cfg)
        shapes = unet_param_shapes(dim=dim, dim_mults=mults, resnet_block_order=order)
        assert {k: tuple(v) for k, v in shapes.items()} == {k: v[1] for k, v in layout.items()}
        spans = sorted((off, off + int(np.prod(shape))) for off, shape in layout.values())
        assert all(a % 4 == 0 for a, _ in spans) and spans[-1][1] <= n
        assert all(spans[i][1] <= spans[i + 1][0] for i in range(len(spans) - 1))
        h = C.c_void_p()
        N.check(N.lib().dmn_train_create(C.byref(cfg), 1, C.byref(h)))
        b = C.c_size_t()
        N.check(N.lib().dmn_train_workspace_bytes(h, 2, 32, 32, C.byref(b)))
        assert b.value > 4 * n
        assert N.lib().dmn_train_workspace_bytes(h, 2, 34, 34, C.byref(b)) != 0   # not a multiple of 2^(levels-1)
        N.lib().dmn_train_destroy(h)
    # learned-variance networks are refused, not silently trained on a different loss
    cfg.out_dim = 6
    h = C.c_void_p()
    assert N.lib().dmn_train_create(C.byref(cfg), 1, C.byref(h)) != 0
    assert cosine_annealing_lr(0, 100, 1e-3, 1e-4) == pytest.approx(1e-3)
    assert cosine_annealing_lr(50, 100, 1e-3, 1e-4) == pytest.approx(5.5e-4)
    assert cosine_annealing_lr(100, 100, 1e-3, 1e-4) == pytest.approx(1e-4)


def test_instantiate_refuses_targets_outside_the_package(tmp_path):
    """A config (also one taken from a `.nemo` archive) may only name classes of this package or of the reference
    package it mirrors: anything else must not even be imported."""
    import io
    import tarfile

    import yaml
    from diffusion_model_nemo_amd.config import instantiate, resolve_target
    from diffusion_model_nemo_amd.models import DDPM
    for bad in ("subprocess.call", "os.system", "builtins.eval", "diffusion_model_nemo_amd.config.load_config",
                "diffusion_model_nemo_amd_evil.Thing", "torch.load"):
        with pytest.raises(ValueError):
            resolve_target(bad)
    with pytest.raises(ValueError):
        instantiate({"_target_": "subprocess.call", "args": ["true"]})
    assert resolve_target("diffusion_model_nemo.modules.GaussianDiffusion").__name__ == "GaussianDiffusion"
    # the same through restore_from: a crafted archive whose sampler names a foreign callable
    cfg = {"image_size": 32, "timesteps": 4, "channels": 3,
           "diffusion_model": {"_target_": "diffusion_model_nemo.modules.Unet", "input_dim": 32, "dim": 32,
                               "use_convnext": False, "dropout": 0.0},
           "sampler": {"_target_": "subprocess.call", "args": ["true"]},
           "loss": {"_target_": "diffusion_model_nemo.loss.DiffusionLoss", "loss_type": "l2", "reduction": "mean"}}
    path = str(tmp_path / "evil.nemo")
    with tarfile.open(path, "w:") as tar:
        for name, data in (("model_config.yaml", yaml.safe_dump(cfg).encode()), ("model_weights.ckpt", b"unused")):
            info = tarfile.TarInfo("./" + name)
            info.size = len(data)
            tar.addfile(info, io.BytesIO(data))
    with pytest.raises(ValueError, match="refusing to import"):
        DDPM.restore_from(path)


def test_learning_rate_and_clip_follow_the_config():
    """`train_batch` takes AdamW arguments, the CosineAnnealing policy (min_lr, no warm-up) and the clip value from the
    config sections the reference's trainer reads (examples/configs/ddpm/unet_small.yaml:65-78,83-84,90)."""
    import math
    from diffusion_model_nemo_amd.models import DDPM
    base = {"image_size": 32, "timesteps": 10, "channels": 3,
            "diffusion_model": {"_target_": "diffusion_model_nemo.modules.Unet", "input_dim": 32, "dim": 32,
                                "use_convnext": False, "dropout": 0.0},
            "sampler": {"_target_": "diffusion_model_nemo.modules.GaussianDiffusion", "timesteps": 10, "schedule_name": "linear"},
            "loss": {"_target_": "diffusion_model_nemo.loss.DiffusionLoss", "loss_type": "l2", "reduction": "mean"},
            "optim": {"name": "adamw", "lr": 0.001, "betas": [0.9, 0.98], "weight_decay": 0.001,
                      "sched": {"name": "CosineAnnealing", "warmup_steps": None, "warmup_ratio": None, "min_lr": 1e-4}}}
    m = DDPM(base)
    m.trainer_cfg = {"gradient_clip_val": 1.0, "max_steps": 100}
    o = m._optim_settings()
    assert o["betas"] == (0.9, 0.98) and o["weight_decay"] == 0.001 and o["clip"] == 1.0 and o["max_steps"] == 100
    got = [m.learning_rate(s) for s in (0, 1, 50, 99, 100, 500)]
    want = [1e-4 + 0.5 * (1e-3 - 1e-4) * (1 + math.cos(math.pi * s / 100)) for s in (0, 1, 50, 99)] + [1e-4, 1e-4]
    assert all(abs(a - b) < 1e-12 for a, b in zip(got, want)), (got, want)
    assert got[0] == 1e-3 and got[0] > got[1] > got[2] > got[3] > got[4]
    # a trainer object handed to the constructor wins over the YAML section
    m2 = DDPM(base, trainer=type("T", (), {"gradient_clip_val": 0.5, "max_steps": 10})())
    assert m2._optim_settings()["clip"] == 0.5 and m2.learning_rate(10) == 1e-4
    # no horizon known: constant learning rate; no trainer section: no clipping
    m3 = DDPM(base)
    assert m3.learning_rate(12345) == 1e-3 and m3._optim_settings()["clip"] == 0.0
    with pytest.raises(NotImplementedError):
        DDPM({**base, "optim": {"name": "sgd"}})._optim_settings()


def test_conditional_ddpm_labels_come_from_one_helper():
    from diffusion_model_nemo_amd.models import ConditionalDDPM
    cfg = {"image_size": 32, "timesteps": 10, "channels": 3, "num_classes": 10,
           "diffusion_model": {"_target_": "diffusion_model_nemo.modules.Unet", "input_dim": 32, "dim": 32, "num_classes": 10,
                               "use_convnext": False, "dropout": 0.0},
           "sampler": {"_target_": "diffusion_model_nemo.modules.GaussianDiffusion", "timesteps": 10, "schedule_name": "linear"},
           "loss": {"_target_": "diffusion_model_nemo.loss.DiffusionLoss", "loss_type": "l2", "reduction": "mean"}}
    m = ConditionalDDPM(cfg)
    assert m.sampler.use_class_conditioning and m.random_class_index == 10
    assert m.resolve_labels(3, "cpu").tolist() == [10, 10, 10]
    assert m.resolve_labels(2, "cpu", 4).tolist() == [4, 4]
    given = torch.tensor([1, 2, 3, 4, 5, 6, 7, 8])
    assert m.resolve_labels(8, "cpu", given).tolist() == given.tolist()
    torch.manual_seed(0)
    dropped = m.resolve_labels(8, "cpu", given, drop=True)
    torch.manual_seed(0)
    mask = torch.randint(0, 2, size=[8], dtype=torch.bool)     # the reference's draw (conditional_ddpm.py:60)
    assert dropped.tolist() == torch.masked_fill(given, mask, 10).tolist() and mask.any() and not mask.all()
    m.train()
    torch.manual_seed(0)
    fn = m.get_diffusion_model({"label": given})
    assert fn.keywords["classes"].tolist() == dropped.tolist()
    m.eval()
    assert m.get_diffusion_model({"label": given}).keywords["classes"].tolist() == given.tolist()
    m.change_sampler({"_target_": "diffusion_model_nemo.modules.GaussianDiffusion", "timesteps": 10, "schedule_name": "cosine"})
    assert m.sampler.use_class_conditioning
    with pytest.raises(RuntimeWarning):
        ConditionalDDPM({**cfg, "num_classes": None})
