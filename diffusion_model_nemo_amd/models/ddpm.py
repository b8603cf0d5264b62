"""`DDPM` / `ImprovedDDPM` model wrappers with the reference's configuration surface.

Mirror of diffusion_model_nemo/models/abstract_diffusion_model.py:26-135, models/ddpm.py:23-109 and
models/improved_ddpm.py:21-25, without NeMo / Lightning (neither is needed to sample): the constructor takes the
`model:` section of the reference's YAML (a mapping with `_target_` entries) and instantiates network, sampler and
loss exactly like `ModelPT.__init__` + `hydra.utils.instantiate` do, so `examples/configs/*/unet_small.yaml` load
unchanged. Parameter names match the reference (`diffusion_model.*`), so its checkpoints' state_dicts load.
"""
from typing import Dict, Iterable, Optional

import torch
import torch.nn as nn

from ..config import Config, instantiate, load_config, to_config


class AbstractDiffusionModel(nn.Module):
    def __init__(self, cfg, trainer=None):
        super().__init__()
        self.cfg = cfg if isinstance(cfg, Config) else to_config(cfg)
        self.trainer = trainer
        # Timesteps must be defined at the global level
        self.image_size = self.cfg.image_size
        self.timesteps = self.cfg.timesteps
        self.channels = self.cfg.channels
        self.save_and_sample_every = self.cfg.get('save_every', 1000)

    @classmethod
    def from_config_file(cls, path: str, overrides: Optional[Iterable[str]] = None, trainer=None):
        """Loads a reference YAML (e.g. examples/configs/ddpm/unet_small.yaml) with Hydra-style overrides such as
        ``model.image_size=64`` and builds the model from its ``model:`` section."""
        full = load_config(path, overrides)
        model = cls(full.model, trainer=trainer)
        # the trainer section carries what Lightning applies around training_step (gradient_clip_val, max_steps)
        model.trainer_cfg = full.get('trainer', None)
        return model

    def get_diffusion_model(self, batch: Dict):
        return self.forward

    # ---- `.nemo` archives (what ModelPT.save_to / restore_from exchange: a tar with model_config.yaml and
    # model_weights.ckpt; used by examples/ddpm/eval_ddpm.py:70 and the other example scripts) --------------------
    CONFIG_MEMBER = "model_config.yaml"
    WEIGHTS_MEMBER = "model_weights.ckpt"
    # AdamW moments, step count and the digest of the flat-buffer layout they belong to travel as a member of their own:
    # `model_weights.ckpt` stays the pure state_dict that the reference restores with a strict load_state_dict, and a
    # reader that does not know this member ignores it
    OPTIM_MEMBER = "optimizer_state.ckpt"
    OPTIM_PREFIX = "__optimizer__."   # where earlier archives of this package kept the same entries: still read

    def save_to(self, save_path: str):
        """Writes the model section of the config and the state_dict in the `.nemo` layout (uncompressed tar)."""
        import io
        import tarfile

        import yaml

        def plain(o):   # Config -> builtin containers (raw access: no attribute wrapping on the way)
            if isinstance(o, dict):
                return {k: plain(v) for k, v in dict.items(o)}
            if isinstance(o, (list, tuple)):
                return [plain(v) for v in o]
            return o

        blobs = {self.CONFIG_MEMBER: yaml.safe_dump(plain(self.cfg), sort_keys=False).encode()}
        buf = io.BytesIO()
        tensors = {k: v.detach().cpu() for k, v in self.state_dict().items()}
        torch.save(tensors, buf)
        blobs[self.WEIGHTS_MEMBER] = buf.getvalue()
        tr = getattr(self, "_native_trainer", None)
        opt_state = None
        if tr is not None and tr.step_count > 0:   # AdamW moments and step count, so a restored run resumes them
            opt_state = {k: (v.cpu() if isinstance(v, torch.Tensor) else v) for k, v in tr.state_dict().items()}
        elif getattr(self, "_pending_optimizer_state", None):
            opt_state = self._pending_optimizer_state   # restored and saved again without a training step in between
        if opt_state:
            obuf = io.BytesIO()
            torch.save(opt_state, obuf)
            blobs[self.OPTIM_MEMBER] = obuf.getvalue()
        with tarfile.open(save_path, "w:") as tar:
            for name, data in blobs.items():
                info = tarfile.TarInfo("./" + name)
                info.size = len(data)
                tar.addfile(info, io.BytesIO(data))

    @classmethod
    def restore_from(cls, restore_path: str, trainer=None, map_location=None, strict: bool = True, overrides=None):
        """Builds the model from a `.nemo` archive (plain or gzip-compressed tar). The weights are read with
        `torch.load(..., weights_only=True)` and the config with `yaml.SafeLoader`; its `_target_` entries may only
        name classes of this package (or their `diffusion_model_nemo.*` aliases), anything else raises ValueError."""
        import io
        import tarfile

        import yaml
        with tarfile.open(restore_path, "r:*") as tar:
            members = {m.name.lstrip("./"): m for m in tar.getmembers() if m.isfile()}
            for need in (cls.CONFIG_MEMBER, cls.WEIGHTS_MEMBER):
                if need not in members:
                    raise FileNotFoundError(f"{restore_path} holds no {need}")
            cfg = yaml.load(tar.extractfile(members[cls.CONFIG_MEMBER]).read().decode(), Loader=yaml.SafeLoader)
            blob = tar.extractfile(members[cls.WEIGHTS_MEMBER]).read()
            oblob = tar.extractfile(members[cls.OPTIM_MEMBER]).read() if cls.OPTIM_MEMBER in members else None
        if isinstance(cfg, dict) and "model" in cfg and "diffusion_model" not in cfg:
            cfg = cfg["model"]            # a full experiment config rather than its model section
        model = cls(to_config(cfg, overrides), trainer=trainer)
        state = torch.load(io.BytesIO(blob), map_location=map_location or "cpu", weights_only=True)
        if isinstance(state, dict) and "state_dict" in state and not any(k.startswith("diffusion_model.") for k in state):
            state = state["state_dict"]
        opt_state = {k[len(cls.OPTIM_PREFIX):]: state.pop(k) for k in [k for k in state if k.startswith(cls.OPTIM_PREFIX)]}
        model.load_state_dict(state, strict=strict)
        if oblob is not None:
            opt_state = torch.load(io.BytesIO(oblob), map_location="cpu", weights_only=True)
        if opt_state:
            model._pending_optimizer_state = opt_state   # handed to the trainer when it is created on a device
        return model

    def change_sampler(self, sampler_cfg):
        """abstract_diffusion_model.py:130-135."""
        self.sampler = instantiate(sampler_cfg)
        self.cfg.sampler = sampler_cfg

    def set_compute_dtype(self, compute_dtype: str):
        self.diffusion_model.set_compute_dtype(compute_dtype)
        return self

    def calculate_bits_per_dimension(self, x_start: torch.Tensor, diffusion_model_fn=None, max_batch_size: int = 32,
                                     seed: Optional[int] = None):
        """abstract_diffusion_model.py:137-197: the variational bound in bits per dimension, one network evaluation
        per timestep, all T steps replayed on the device. Returns {'total_bpd' [B], 'terms_bpd' [B, T],
        'prior_bpd' [B]}. `diffusion_model_fn` is what `get_diffusion_model(batch)` returns (default: the network)."""
        import ctypes as C
        from .. import _native as N
        from ..modules.gaussian_diffusion import _unwrap_conditional
        from ..modules.unet import Unet
        sampler = self.sampler
        if getattr(sampler, "objective", "pred_noise") != "pred_noise":
            raise NotImplementedError("bits per dimension is built for the 'pred_noise' objective")
        net, labels = _unwrap_conditional(self.diffusion_model if diffusion_model_fn is None else diffusion_model_fn)
        if not isinstance(net, Unet):
            raise NotImplementedError("bits per dimension needs the accelerated network (no Python-callable path)")
        if not x_start.is_cuda:
            raise RuntimeError("calculate_bits_per_dimension runs only on MI355X device tensors")
        B = x_start.size(0) if max_batch_size <= 0 else min(max_batch_size, x_start.size(0))
        x0 = x_start[:B].contiguous().float()
        _, Cc, H, W = x0.shape
        T = sampler.timesteps
        dev = x0.device
        table = sampler.bpd_table(dev)
        terms = torch.zeros(B, T, dtype=torch.float32, device=dev)
        prior = torch.zeros(B, dtype=torch.float32, device=dev)
        net.set_class_labels(labels, B, dev)
        ws = net.workspace(B, H, W, dev, sampling=True)
        seed = sampler._resolve_seed(seed)
        N.check(N.lib().dmn_bpd_loop(
            net.native_handle(), table.data_ptr(), T, x0.data_ptr(), B, H, W, 1 if sampler.learned_variance else 0, seed, 0,
            terms.data_ptr(), T, prior.data_ptr(), float(sampler.sqrt_alphas_cumprod[T - 1]),
            float(sampler.log_one_minus_alphas_cumprod[T - 1]), ws.data_ptr(), ws.numel(),
            N.current_stream_handle(dev), 1))
        net.check_handoffs()   # waits for the loop; terms computed from starved statistics must not be handed on
        return {'total_bpd': terms.sum(dim=1) + prior, 'terms_bpd': terms, 'prior_bpd': prior}


class DDPM(AbstractDiffusionModel):
    def __init__(self, cfg, trainer=None):
        super().__init__(cfg, trainer=trainer)
        self.diffusion_model = instantiate(self.cfg.diffusion_model)
        self.sampler = instantiate(self.cfg.sampler)
        self.loss = instantiate(self.cfg.loss)

    def forward(self, x_t: torch.Tensor, t: torch.Tensor, classes: torch.Tensor = None):
        return self.diffusion_model(x_t, t)

    def sample(self, batch_size: int, image_size: int, device: torch.device = None, **kwargs):
        """models/ddpm.py:105-109; extra keyword arguments go to the sampler (keep=, seed=, image_offset=, ...)."""
        with torch.inference_mode():
            self.eval()
            shape = [batch_size, self.channels, image_size, image_size]
            return self.sampler.sample(self.diffusion_model, shape=shape, device=device, **kwargs)

    def interpolate(self, x1, x2, t: Optional[int] = None, lambd: float = 0.5, **kwargs):
        with torch.inference_mode():
            self.eval()
            assert x1.ndim == 4, f"x1 is not a batch of tensors ! Given shape {x1.shape}"
            assert x2.ndim == 4, f"x2 is not a batch of tensors ! Given shape {x2.shape}"
            dev = next(self.parameters()).device
            return self.sampler.interpolate(self.diffusion_model, x1=x1.to(dev), x2=x2.to(dev), t=t, lambd=lambd)

    def training_loss(self, batch, t: Optional[torch.Tensor] = None, noise: Optional[torch.Tensor] = None):
        """Forward half of models/ddpm.py:43-57: t ~ U{0..T-1}, x_t = q_sample(x_0, t, eps), loss(net(x_t, t), eps).
        The value comes from the sampling path's forward (any compute_dtype) and carries no autograd graph;
        `training_step` / `train_batch` are the entry points that also compute the gradients."""
        samples = batch["pixel_values"]
        B = samples.shape[0]
        if t is None:
            t = torch.randint(0, self.timesteps, size=(B,), device=samples.device, dtype=torch.long)
        if noise is None:
            noise = torch.randn_like(samples)
        with torch.no_grad():
            x_t = self.sampler.q_sample(x_start=samples, t=t, noise=noise)
            out = self.forward(x_t, t)
            if getattr(self.diffusion_model, "learned_variance", False):
                out, _ = out.chunk(2, dim=1)
            return self.loss(input=out, target=noise)

    # ---- training (models/ddpm.py:43-57 + the Lightning trainer's backward / clip / optimiser step) -------------
    def native_trainer(self):
        """The `UnetTrainer` that owns the flat parameter / gradient / AdamW buffers of `diffusion_model`."""
        from ..training import UnetTrainer
        tr = getattr(self, "_native_trainer", None)
        if tr is None:
            if getattr(self.diffusion_model, "learned_variance", False):
                raise NotImplementedError("the training step is built for L_simple (out_dim == channels) only")
            loss_type = getattr(self.loss, "loss_type", "l2")
            if getattr(self.loss, "_reduction", "mean") != "mean":
                raise NotImplementedError("the training step is built for reduction='mean'")
            tr = UnetTrainer(self.diffusion_model, loss_type=loss_type)
            pending = self.__dict__.pop("_pending_optimizer_state", None)
            if pending:
                tr.load_state_dict(pending)
            object.__setattr__(self, "_native_trainer", tr)
        return tr

    def _optim_settings(self):
        """AdamW arguments, learning-rate policy and clip value as the reference's trainer would apply them
        (examples/configs/ddpm/unet_small.yaml:65-78,83-84,90)."""
        optim = self.cfg.get('optim', None) or {}
        name = optim.get('name', 'adamw')
        if name != 'adamw':
            raise NotImplementedError(f"optimizer '{name}': only adamw is built")
        sched = optim.get('sched', None) or {}
        def trainer_value(key):   # a trainer object / mapping handed to the constructor wins over the YAML's section
            for src in (self.trainer, getattr(self, 'trainer_cfg', None)):
                if src is None:
                    continue
                v = src.get(key, None) if isinstance(src, dict) else getattr(src, key, None)
                if v is not None:
                    return v
            return None

        sname = sched.get('name', None)
        if sname not in (None, 'CosineAnnealing'):
            raise NotImplementedError(f"learning-rate policy '{sname}': only CosineAnnealing is built")
        max_steps = sched.get('max_steps', None) or trainer_value('max_steps')
        if max_steps is not None and int(max_steps) <= 0:
            max_steps = None
        clip = trainer_value('gradient_clip_val')
        return {"lr": float(optim.get('lr', 1e-3)), "betas": tuple(optim.get('betas', (0.9, 0.999))),
                "eps": float(optim.get('eps', 1e-8)), "weight_decay": float(optim.get('weight_decay', 0.01)),
                "sched": sname, "min_lr": float(sched.get('min_lr', 0.0) or 0.0),
                "warmup_steps": int(sched.get('warmup_steps', 0) or 0),
                "max_steps": None if max_steps is None else int(max_steps),
                "clip": 0.0 if clip is None else float(clip)}

    def learning_rate(self, step: int) -> float:
        """Learning rate of optimisation step `step` (0-based) under the config's policy: CosineAnnealing from `lr` to
        `min_lr` over `optim.sched.max_steps` / `trainer.max_steps` when a policy and a horizon are configured (Lightning
        derives the horizon from the data loader when max_steps is null; without a loader here it must be given),
        constant `lr` otherwise."""
        from ..training import cosine_annealing_lr
        o = self._optim_settings()
        if o["sched"] == 'CosineAnnealing' and o["max_steps"]:
            return cosine_annealing_lr(step, o["max_steps"], o["lr"], o["min_lr"], o["warmup_steps"])
        return o["lr"]

    def _training_classes(self, batch):
        """Class labels of a training batch: what `get_diffusion_model(batch)` binds to `classes` (none here; the
        class-conditional subclass applies its label drop-out there, models/conditional_ddpm.py:51-67)."""
        fn = self.get_diffusion_model(batch)
        return getattr(fn, "keywords", {}).get("classes") if hasattr(fn, "keywords") else None

    def training_step(self, batch, batch_nb=0, t: Optional[torch.Tensor] = None, noise: Optional[torch.Tensor] = None):
        """models/ddpm.py:43-57: t ~ U{0..T-1}, eps ~ N(0, I), x_t = q_sample(x_0, t, eps), loss(net(x_t, t), eps).
        Forward AND backward run in the HIP kernels during this call; the returned scalar carries an autograd node
        whose backward hands the finished gradients to every parameter's `.grad` (scaled by the incoming gradient),
        so `loss.backward()` followed by any torch optimiser behaves like the reference. `train_batch` is the
        all-device alternative (all-reduce + clip + AdamW without leaving the GPU)."""
        samples = batch["pixel_values"]
        B = samples.shape[0]
        dev = samples.device
        if t is None:
            t = torch.randint(0, self.timesteps, size=(B,), device=dev, dtype=torch.long)
        if noise is None:
            noise = torch.randn_like(samples)
        tr = self.native_trainer()
        s = self.sampler
        value = tr.loss_and_grads(samples, t, noise, s.sqrt_alphas_cumprod, s.sqrt_one_minus_alphas_cumprod,
                                  classes=self._training_classes(batch))
        params = [p for p in self.diffusion_model.parameters()]
        names = [n for n, _ in self.diffusion_model.named_parameters()]
        return _NativeLoss.apply(value, tr, names, *params)

    def train_batch(self, batch, lr: Optional[float] = None, t=None, noise=None):
        """One complete optimisation step on the device: loss + gradients, ONE all-reduce of the flat gradient
        buffer when torch.distributed is initialised, gradient clipping (`trainer.gradient_clip_val`; off when no
        trainer section is known) and AdamW with the `optim` section of the config, the learning rate following
        `optim.sched` (examples/configs/ddpm/unet_small.yaml:65-78,90) unless `lr` overrides it. Returns the loss."""
        samples = batch["pixel_values"]
        B = samples.shape[0]
        dev = samples.device
        if t is None:
            t = torch.randint(0, self.timesteps, size=(B,), device=dev, dtype=torch.long)
        if noise is None:
            noise = torch.randn_like(samples)
        tr = self.native_trainer()
        s = self.sampler
        value = tr.loss_and_grads(samples, t, noise, s.sqrt_alphas_cumprod, s.sqrt_one_minus_alphas_cumprod,
                                  classes=self._training_classes(batch))
        scale = tr.all_reduce_grads()
        o = self._optim_settings()
        tr.optimizer_step(lr=self.learning_rate(tr.step_count) if lr is None else lr, betas=o["betas"], eps=o["eps"],
                          weight_decay=o["weight_decay"], max_grad_norm=o["clip"], grad_scale=scale)
        return value


class _NativeLoss(torch.autograd.Function):
    """Ties the loss value computed by the HIP training step to the parameters: backward returns the gradients the
    kernels already produced (views of the trainer's flat gradient buffer) times the incoming gradient."""

    @staticmethod
    def forward(ctx, value, trainer, names, *params):
        # the trainer's gradient buffer is overwritten by its next step: keep this step's own copy
        ctx.layout = [trainer.layout[n] for n in names]
        ctx.flat = trainer.grads.clone()
        return value.clone()

    @staticmethod
    def backward(ctx, grad_out):
        import math
        flat = ctx.flat * grad_out
        grads = tuple(flat[off:off + int(math.prod(shape))].view(shape) for off, shape in ctx.layout)
        return (None, None, None) + grads


class ImprovedDDPM(DDPM):
    """Learned-variance DDPM: same wrapper, `LearnedGaussianDiffusion` sampler and a 2C-channel network
    (models/improved_ddpm.py:21-25; examples/configs/improved_ddpm/unet_small.yaml:25,40). The variational-bound
    loss entry of the config (`vb_loss`) is only needed for training and is not instantiated."""
