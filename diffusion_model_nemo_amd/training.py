"""Training step of the DDPM U-Net on the MI355X kernels (libdmn_hip.so, `dmn_train_*`).

Host side of what the reference does in `DDPM.training_step` (diffusion_model_nemo/models/ddpm.py:43-57) plus its
Lightning trainer's backward / clip / AdamW / cosine learning-rate schedule
(examples/configs/ddpm/unet_small.yaml:65-78,90). `UnetTrainer` keeps parameters, gradients and the AdamW moments in
one flat float32 device tensor each (every parameter in the reference's own layout, at offsets the library reports);
the `Unet` module's parameters are re-pointed at views of the flat parameter tensor, so `state_dict()`, checkpoints and
the sampling path keep working, and one RCCL all-reduce of the flat gradient tensor is the whole data-parallel
exchange. There is no torch autograd implementation behind this: a missing library or a CPU tensor raises.
"""
import ctypes as C
import math
from typing import Dict, Optional

import torch

from . import _native as N

LOSS_KINDS = {"l1": 0, "l2": 1, "huber": 2}


def flat_layout(unet_cfg: "N.DmnUnetCfg", loss_kind: int = 1):
    """(layout, flat_size) of the library's flat parameter buffer for a network configuration:
    layout[name] = (offset in floats, PyTorch shape). Needs the library but no GPU."""
    lib = N.lib()
    h = C.c_void_p()
    N.check(lib.dmn_train_create(C.byref(unet_cfg), loss_kind, C.byref(h)))
    try:
        layout = {}
        for i in range(lib.dmn_train_num_tensors(h)):
            name, off, nd, sh = C.c_char_p(), C.c_int64(), C.c_int(), (C.c_int64 * 4)()
            N.check(lib.dmn_train_tensor_info(h, i, C.byref(name), C.byref(off), C.byref(nd), C.byref(sh)))
            layout[name.value.decode()] = (int(off.value), tuple(int(s) for s in sh[:nd.value]))
        return layout, int(lib.dmn_train_flat_size(h))
    finally:
        lib.dmn_train_destroy(h)


class UnetTrainer:
    def __init__(self, unet, loss_type: str = "l2", device=None, activations: Optional[str] = None):
        """`unet.compute_dtype` selects the matrix-core operand type ('fp32' exact, 'bf16' rounded operands with fp32
        accumulation). `activations` selects how the saved activations and their gradients are stored: 'fp32', or
        'bf16' (only with bf16 operands; halves the workspace and cuts the step's memory traffic by a third; in the band test its
        gradients are as close to the fp32 oracle as those of fp32 storage, the operand rounding dominates; parameters, their gradients, statistics and the optimiser stay fp32). Default 'fp32', the parity mode; on one MI355X the bf16
        storage moves 30 GB instead of 46 GB per step of 32 images at 128x128 and takes 9.5 ms instead of 10.1."""
        if loss_type not in LOSS_KINDS:
            raise ValueError("loss_type must be 'l1', 'l2' or 'huber'")
        if activations is None:
            activations = "fp32"
        if activations not in ("fp32", "bf16") or (activations == "bf16" and unet.compute_dtype != "bf16"):
            raise ValueError("activations must be 'fp32', or 'bf16' together with compute_dtype='bf16'")
        self.activations = activations
        self.unet = unet
        self.device = torch.device(device) if device is not None else next(unet.parameters()).device
        if self.device.type != "cuda":
            raise RuntimeError("UnetTrainer needs the network on an MI355X device (no CPU implementation): "
                               "move it with .cuda() / .to(device) first")
        self._synced = False   # data-parallel replicas agree on parameters and optimiser state
        lib = N.lib()
        h = C.c_void_p()
        cfg = unet._cfg()
        if activations == "bf16":
            cfg.dtype = 2
        with torch.cuda.device(self.device):   # the library works on the current device (streams, descriptors)
            N.check(lib.dmn_train_create(C.byref(cfg), LOSS_KINDS[loss_type], C.byref(h)))
        self._handle = h
        self.layout: Dict[str, tuple] = {}
        for i in range(lib.dmn_train_num_tensors(h)):
            name, off, nd, sh = C.c_char_p(), C.c_int64(), C.c_int(), (C.c_int64 * 4)()
            N.check(lib.dmn_train_tensor_info(h, i, C.byref(name), C.byref(off), C.byref(nd), C.byref(sh)))
            self.layout[name.value.decode()] = (int(off.value), tuple(int(s) for s in sh[:nd.value]))
        self.flat_size = int(lib.dmn_train_flat_size(h))
        state = unet.state_dict()
        if set(state) != set(self.layout):
            raise RuntimeError(f"parameter names differ: {sorted(set(state) ^ set(self.layout))[:5]} ...")
        self.params = torch.zeros(self.flat_size, dtype=torch.float32, device=self.device)
        self.grads = torch.zeros_like(self.params)
        self.exp_avg = torch.zeros_like(self.params)
        self.exp_avg_sq = torch.zeros_like(self.params)
        self._scratch = torch.zeros(1024, dtype=torch.float32, device=self.device)
        self._loss = torch.zeros(1, dtype=torch.float32, device=self.device)
        self._ws = None
        self._ws_key = None
        self.step_count = 0
        # the module's parameters become views of the flat buffer; their .grad stays untouched (the autograd route of
        # DDPM.training_step accumulates there like torch does, `grad_of` reads the kernels' own buffer)
        named = dict(unet.named_parameters())
        for name, (off, shape) in self.layout.items():
            n = int(math.prod(shape))
            view = self.params[off:off + n].view(shape)
            view.copy_(state[name].to(self.device, torch.float32))
            p = named[name]
            p.data = view

    def __del__(self):
        try:
            if self._handle is not None:
                N.lib().dmn_train_destroy(self._handle)
        except Exception:
            pass

    # ---- optimiser state (AdamW moments + step count) for checkpoints -------------------------------------------
    def layout_digest(self) -> str:
        """Names, offsets and shapes of the flat buffers in one string: the moments of a checkpoint only mean something
        under the layout they were written with."""
        import hashlib
        text = ";".join(f"{n}@{off}:{'x'.join(map(str, shape))}" for n, (off, shape) in sorted(self.layout.items()))
        return hashlib.sha1(f"{self.flat_size}|{text}".encode()).hexdigest()

    def state_dict(self) -> Dict[str, torch.Tensor]:
        return {"exp_avg": self.exp_avg.detach().clone(), "exp_avg_sq": self.exp_avg_sq.detach().clone(),
                "step_count": torch.tensor(self.step_count, dtype=torch.int64), "layout": self.layout_digest()}

    def load_state_dict(self, state: Dict[str, torch.Tensor]):
        if "layout" in state and state["layout"] != self.layout_digest():
            raise ValueError("optimizer state was written under another flat-buffer layout (names / offsets / shapes differ): "
                             "its moments would be applied to the wrong parameters")
        for k in ("exp_avg", "exp_avg_sq"):
            if state[k].numel() != self.flat_size:
                raise ValueError(f"optimizer state '{k}' has {state[k].numel()} elements, expected {self.flat_size}")
            getattr(self, k).copy_(state[k].to(self.device, torch.float32).reshape(-1))
        self.step_count = int(state["step_count"])

    def _check_views(self):
        """The module's parameters must still be the views of the flat buffer made at construction: `.to()`, `.float()`
        or `set_compute_dtype` on the module after that re-allocate them and the kernels would train a stale copy."""
        p = self.unet.init_conv.weight
        off, shape = self.layout["init_conv.weight"]
        if p.data_ptr() != self.params.data_ptr() + 4 * off or p.device != self.device:
            raise RuntimeError("the network's parameters no longer alias the trainer's flat buffer (the module was moved "
                               "or converted after UnetTrainer was created): build a new UnetTrainer")

    def sync_replicas(self, group=None):
        """Data parallel: rank 0's parameters, AdamW moments and step count to every rank (what Lightning DDP does for
        parameters when it wraps the module, i.e. BEFORE the first forward). Runs in front of the first gradient
        computation once a process group of more than one rank exists (`_before_step`); a no-op, to be tried again at
        the next step, while torch.distributed is not initialised."""
        import torch.distributed as dist
        from .parallel import broadcast_training_state
        if not (dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1):
            return False
        self.step_count = broadcast_training_state((self.params, self.exp_avg, self.exp_avg_sq), self.step_count, group)
        self.unet._weights_version = None   # the sampling handle repacks from the (possibly replaced) parameters
        self._synced = True
        return True

    def _before_step(self):
        """In front of every gradient computation: replicas that have not been brought together yet are now, so that no
        rank ever differentiates at parameters the others do not hold."""
        if not self._synced:
            self.sync_replicas()

    def grad_of(self, name: str) -> torch.Tensor:
        off, shape = self.layout[name]
        return self.grads[off:off + int(math.prod(shape))].view(shape)

    def _workspace(self, B, H, W):
        key = (B, H, W)
        if self._ws_key != key:
            n = C.c_size_t()
            with torch.cuda.device(self.device):
                N.check(N.lib().dmn_train_workspace_bytes(self._handle, B, H, W, C.byref(n)))
            self._ws = torch.empty(n.value + 256, dtype=torch.uint8, device=self.device)
            self._ws_key = key
        return self._ws

    def loss_and_grads(self, x0: torch.Tensor, t: torch.Tensor, noise: torch.Tensor,
                       sqrt_alphas_cumprod: Optional[torch.Tensor] = None,
                       sqrt_one_minus_alphas_cumprod: Optional[torch.Tensor] = None,
                       model_out: Optional[torch.Tensor] = None, classes: Optional[torch.Tensor] = None) -> torch.Tensor:
        """loss = DiffusionLoss(Unet(q_sample(x0, t, noise), t), noise) and its gradient into `self.grads`
        (overwritten). Without the two schedule tables x0 is taken as x_t. Returns the loss as a device scalar."""
        if not x0.is_cuda:
            raise RuntimeError("the training step runs only on MI355X device tensors (no CPU implementation)")
        if x0.device != self.device:
            raise RuntimeError(f"batch on {x0.device}, trainer on {self.device}")
        self._check_views()
        self._before_step()
        B, Cc, H, W = x0.shape
        if Cc != self.unet.channels:
            raise ValueError(f"expected {self.unet.channels} channels, got {Cc}")
        x0 = x0.contiguous().float()
        noise = noise.contiguous().float()
        t = t.to(self.device, torch.int64).contiguous()
        if t.shape != (B,) or noise.shape != x0.shape:
            raise ValueError("t must be [B] and noise must match x0")
        sa = sb = None
        if sqrt_alphas_cumprod is not None:
            sa = sqrt_alphas_cumprod.to(self.device, torch.float32).contiguous()
            sb = sqrt_one_minus_alphas_cumprod.to(self.device, torch.float32).contiguous()
            # no range check on device tensors (it would stall the stream every step): the kernel clamps the label for
            # the table look-up
        labels = None
        if self.unet.num_classes is not None:
            if classes is not None:
                labels = classes.to(self.device, torch.int64).contiguous()
                if labels.shape != (B,):
                    raise ValueError("classes must have shape [B]")
            N.check(N.lib().dmn_train_set_class_labels(self._handle, N.ptr(labels)))
        ws = self._workspace(B, H, W)
        base = ws.data_ptr()
        aligned = (base + 255) & ~255
        loss = torch.zeros(1, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            N.check(N.lib().dmn_train_loss_and_grads(
                self._handle, self.params.data_ptr(), self.grads.data_ptr(), x0.data_ptr(), noise.data_ptr(), t.data_ptr(),
                N.ptr(sa), N.ptr(sb), 0 if sa is None else sa.numel(), B, H, W, loss.data_ptr(), N.ptr(model_out), aligned,
                ws.numel() - (aligned - base), N.current_stream_handle(self.device)))
        self._keep = (x0, noise, t, sa, sb, labels)   # borrowed by the kernels until the stream has run them
        return loss[0]

    def all_reduce_grads(self):
        """Data parallel: ONE summing all-reduce (RCCL over xGMI under backend 'nccl') of the flat gradient tensor.
        Returns the factor that turns the sum into the mean; `optimizer_step` applies it on the device."""
        from .parallel import all_reduce_flat_grads
        return all_reduce_flat_grads(self.grads)

    def optimizer_step(self, lr: float, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.01,
                       max_grad_norm: float = 0.0, grad_scale: float = 1.0):
        """clip_grad_norm_(max_grad_norm) + torch.optim.AdamW semantics on the flat buffers, on the device."""
        self.step_count += 1
        with torch.cuda.device(self.device):
            N.check(N.lib().dmn_adamw_step(
                self.params.data_ptr(), self.grads.data_ptr(), self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(),
                self.flat_size, lr, betas[0], betas[1], eps, weight_decay, self.step_count, max_grad_norm, grad_scale,
                self._scratch.data_ptr(), N.current_stream_handle(self.device)))
        self.unet._weights_version = None   # the sampling handle repacks from the updated parameters


def cosine_annealing_lr(step: int, max_steps: int, base_lr: float, min_lr: float, warmup_steps: int = 0) -> float:
    """Learning rate of NeMo's `CosineAnnealing` policy as configured by the reference
    (examples/configs/ddpm/unet_small.yaml:73-78: no warm-up, min_lr 1e-4): linear warm-up when asked for, then
    min_lr + (base - min_lr) * (1 + cos(pi * progress)) / 2, and min_lr from max_steps on."""
    if warmup_steps and step < warmup_steps:
        return base_lr * (step + 1) / (warmup_steps + 1)
    if step >= max_steps:
        return min_lr
    prog = (step - warmup_steps) / max(1, max_steps - warmup_steps)
    return min_lr + 0.5 * (base_lr - min_lr) * (1.0 + math.cos(math.pi * prog))
