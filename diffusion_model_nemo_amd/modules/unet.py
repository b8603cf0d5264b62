"""`Unet` with the reference's constructor and `state_dict` names, executed by the gfx950 kernels.

Mirror of diffusion_model_nemo/modules/unet.py:13-168 for ``use_convnext=False``. The torch modules
created here only HOLD parameters (so checkpoints of the reference load unchanged and optimizers see
ordinary ``nn.Parameter`` objects); ``forward`` hands raw device pointers to libdmn_hip.so. There is no
torch / CPU implementation of the forward: a CPU tensor, or a missing library, raises.
"""
import ctypes as C
from typing import List, Optional

import torch
import torch.nn as nn

from .. import _native as N


class _ParamOnly(nn.Module):
    def forward(self, *a, **k):  # pragma: no cover - never executed
        raise RuntimeError("parameter container: the computation runs in libdmn_hip.so")


def _conv(cin, cout, k, bias=True, transposed=False):
    cls = nn.ConvTranspose2d if transposed else nn.Conv2d
    return cls(cin, cout, kernel_size=k, bias=bias)


class _Block(_ParamOnly):
    def __init__(self, cin, cout, groups):
        super().__init__()
        self.proj = _conv(cin, cout, 3)
        self.norm = nn.GroupNorm(groups, cout)


class _ResnetBlock(_ParamOnly):
    """Parameter layout of parts/convnext.py:66-76."""

    def __init__(self, cin, cout, time_emb_dim=None, groups=8):
        super().__init__()
        self.mlp = nn.Sequential(nn.SiLU(), nn.Linear(time_emb_dim, cout)) if time_emb_dim is not None else None
        self.block1 = _Block(cin, cout, groups)
        self.block2 = _Block(cout, cout, groups)
        self.res_conv = _conv(cin, cout, 1) if cin != cout else nn.Identity()


class _LinearAttention(_ParamOnly):
    def __init__(self, dim, heads=4, dim_head=32):
        super().__init__()
        hidden = heads * dim_head
        self.to_qkv = _conv(dim, hidden * 3, 1, bias=False)
        self.to_out = nn.Sequential(_conv(hidden, dim, 1), nn.GroupNorm(1, dim))


class _Attention(_ParamOnly):
    def __init__(self, dim, heads=4, dim_head=32):
        super().__init__()
        hidden = heads * dim_head
        self.to_qkv = _conv(dim, hidden * 3, 1, bias=False)
        self.to_out = _conv(hidden, dim, 1)


class _PreNorm(_ParamOnly):
    def __init__(self, dim, fn):
        super().__init__()
        self.fn = fn
        self.norm = nn.GroupNorm(1, dim)


class _Residual(_ParamOnly):
    def __init__(self, fn):
        super().__init__()
        self.fn = fn


class Unet(nn.Module):
    HEADS, DIM_HEAD = 4, 32  # parts/mha.py:8,34

    def __init__(
        self,
        input_dim: None,
        dim: int,
        out_dim: Optional[int] = None,
        dim_mults: Optional[List[int]] = None,
        channels: int = 3,
        with_time_emb: bool = True,
        resnet_block_groups: int = 8,
        use_convnext: bool = True,
        convnext_mult: int = 2,
        resnet_block_order: str = 'bn_act_conv',
        dropout: Optional[float] = None,
        learned_variance: bool = False,
        num_classes: Optional[int] = None,
        compute_dtype: str = "bf16",
    ):
        super().__init__()
        if use_convnext:
            raise NotImplementedError("use_convnext=True is not part of the accelerated path "
                                      "(every shipped config sets use_convnext: False)")
        if dropout not in (None, 0, 0.0):
            raise NotImplementedError("dropout > 0 is not supported by the HIP forward (sampling path)")
        if resnet_block_order not in ('conv_bn_act', 'bn_act_conv'):
            raise ValueError("Valid ordering for block are : ['conv_bn_act', 'bn_act_conv']")
        if compute_dtype not in ("bf16", "fp32"):
            raise ValueError("compute_dtype must be 'bf16' or 'fp32'")
        dim_mults = tuple(dim_mults) if dim_mults is not None else (1, 2, 4, 8)
        self.channels = channels
        self.learned_variance = learned_variance
        self.dim = dim
        self.resnet_block_order = resnet_block_order
        self.resnet_block_groups = resnet_block_groups
        self.num_classes = num_classes
        self.compute_dtype = compute_dtype
        self._labels = None   # device labels of the current forward / sampling chain (kept alive here)
        self._label_flag = None      # device bool: a label handed over as a device tensor was out of range
        self._giveups_seen = 0       # hand-off give-ups already reported by check_handoffs
        self.init_conv = nn.Conv2d(channels, dim, kernel_size=7, padding=3)

        dims = [dim, *[dim * m for m in dim_mults]]
        in_out = list(zip(dims[:-1], dims[1:]))
        self.dim_list, self.in_out_list, self.dim_mults = dims, in_out, dim_mults
        g = resnet_block_groups
        if with_time_emb:
            time_dim = dim * 4
            # index 0 is the sinusoidal embedding (no parameters), so the Linear layers are .1 and .3
            self.time_mlp = nn.Sequential(nn.Identity(), nn.Linear(dim, time_dim), nn.GELU(), nn.Linear(time_dim, time_dim))
        else:
            raise NotImplementedError("with_time_emb=False is not part of the accelerated path")

        self.downs = nn.ModuleList([])
        self.ups = nn.ModuleList([])
        n_res = len(in_out)
        for ind, (ci, co) in enumerate(in_out):
            last = ind >= n_res - 1
            self.downs.append(nn.ModuleList([
                _ResnetBlock(ci, co, time_dim, g), _ResnetBlock(co, co, time_dim, g),
                _Residual(_PreNorm(co, _LinearAttention(co, self.HEADS, self.DIM_HEAD))),
                nn.Conv2d(co, co, 4, 2, 1) if not last else nn.Identity()]))
        mid = dims[-1]
        self.mid_block1 = _ResnetBlock(mid, mid, time_dim, g)
        self.mid_attn = _Residual(_PreNorm(mid, _Attention(mid, self.HEADS, self.DIM_HEAD)))
        self.mid_block2 = _ResnetBlock(mid, mid, time_dim, g)
        for ind, (ci, co) in enumerate(reversed(in_out[1:])):
            # every up level ends with a transposed convolution (unet.py:95-104)
            self.ups.append(nn.ModuleList([
                _ResnetBlock(co * 2, ci, time_dim, g), _ResnetBlock(ci, ci, time_dim, g),
                _Residual(_PreNorm(ci, _LinearAttention(ci, self.HEADS, self.DIM_HEAD))),
                nn.ConvTranspose2d(ci, ci, 4, 2, 1)]))
        default_out = channels * (2 if learned_variance else 1)
        self.out_dim = out_dim if out_dim is not None else default_out
        if resnet_block_order == 'bn_act_conv':
            tail = [nn.GroupNorm(g, dim), nn.SiLU(), nn.Conv2d(dim, self.out_dim, 1)]
        else:
            tail = [nn.Conv2d(dim, self.out_dim, 1)]
        self.final_conv = nn.Sequential(_ResnetBlock(dim, dim, None, g), *tail)

        if self.num_classes is not None:
            # unet.py:118-120: one extra row (index num_classes) stands for "no class"
            self.class_embed = nn.Embedding(self.num_classes + 1, embedding_dim=self.dim, padding_idx=self.num_classes)

        self._handle = None
        self._handle_device = None
        self._weights_version = None
        self._workspace = None
        self._ws_key = None
        self._plan_batch = 64
        self._sample_chains = None   # None: the library's default (two half-batch chains in the DDPM loop)

    def set_sample_chains(self, chains: int, skew_us: int = 0):
        """How the DDPM sampling loop runs a batch: 2 = two half-batch launch chains on two streams (default for
        even batches of 16 images or more; per-image results are the same bits), 1 = one chain."""
        if chains not in (1, 2):
            raise ValueError("chains must be 1 or 2")
        self._sample_chains = (int(chains), int(skew_us))
        self._workspace = None
        self._ws_key = None
        if self._handle is not None:
            with torch.cuda.device(self._handle_device):
                N.check(N.lib().dmn_unet_set_sample_chains(self._handle, *self._sample_chains))
        return self

    def set_plan_batch(self, batch: int):
        """Batch the launch heuristics plan for (default 64). With a fixed value the result for an image does not
        depend on how the batch holding it is split over calls or ranks; 0 plans for the batch of each call."""
        if batch < 0:
            raise ValueError("plan batch must be >= 0")
        self._plan_batch = int(batch)
        self._workspace = None
        self._ws_key = None
        if self._handle is not None:
            with torch.cuda.device(self._handle_device):
                N.check(N.lib().dmn_unet_set_plan_batch(self._handle, self._plan_batch))
        return self

    def handoff_giveups(self) -> int:
        """Bounded waits of the in-launch GroupNorm hand-offs that ran out since the handle was created (synchronises
        the device; the count never goes down). A count that grew means an evaluation in between used incomplete
        statistics."""
        if self._handle is None:
            return 0
        n = C.c_uint()
        with torch.cuda.device(self._handle_device):
            N.check(N.lib().dmn_unet_handoff_giveups(self._handle, C.byref(n)))
        return int(n.value)

    def check_handoffs(self):
        """Call where results are handed on (the sampling loops and the bits-per-dimension evaluation do; after a bare
        `forward` it is the caller's turn): waits for the device, then raises if an in-launch hand-off timed out since
        the last check -- only the results produced in between are invalid, later calls are not affected -- or if a
        class label given as a device tensor was out of range (the kernels clamp it; the reference raises IndexError)."""
        n = self.handoff_giveups()
        new = n - self._giveups_seen
        self._giveups_seen = n
        bad_labels = self._label_flag is not None and bool(self._label_flag.item())
        self._label_flag = None
        if new > 0:
            raise RuntimeError(f"{new} in-launch GroupNorm hand-offs timed out: the results since the last check are invalid")
        if bad_labels:
            raise IndexError("class label out of range")

    def _device(self) -> torch.device:
        dev = self.init_conv.weight.device
        if dev.type != "cuda":
            raise RuntimeError("the network's parameters must live on an MI355X device (no CPU implementation)")
        return dev

    # ---- native handle -------------------------------------------------------------------------------
    def _cfg(self) -> N.DmnUnetCfg:
        c = N.DmnUnetCfg()
        c.dim = self.dim
        c.n_mults = len(self.dim_mults)
        for i, m in enumerate(self.dim_mults):
            c.dim_mults[i] = int(m)
        c.channels = self.channels
        c.out_dim = self.out_dim
        c.groups = self.resnet_block_groups
        c.heads, c.dim_head = self.HEADS, self.DIM_HEAD
        c.tail_norm = 1 if self.resnet_block_order == 'bn_act_conv' else 0
        c.with_time_emb = 1
        c.dtype = N.BF16 if self.compute_dtype == "bf16" else N.F32
        c.num_classes = 0 if self.num_classes is None else int(self.num_classes)
        return c

    def set_class_labels(self, classes: Optional[torch.Tensor], batch: int, device):
        """Labels used by the following forwards / native sampling loops; None = the 'null' class for every image
        (unet.py:135-137). Ignored by unconditional networks, like the reference ignores `classes` then."""
        if self.num_classes is None:
            return
        if classes is None:
            self._labels = None
        else:
            if classes.dim() != 1 or classes.numel() != batch:
                raise ValueError("classes must have shape [B]")
            on_host = not classes.is_cuda
            if on_host and (int(classes.min()) < 0 or int(classes.max()) > self.num_classes):
                raise IndexError("class label out of range")   # labels on the host are checked where they are
            classes = classes.to(device=device, dtype=torch.int64).contiguous()
            if not on_host:
                # device labels: no round trip to the host here (it would stall the stream before every forward); the
                # kernels clamp, and the verdict is read at the next synchronisation point (check_handoffs)
                bad = ((classes < 0) | (classes > self.num_classes)).any()
                self._label_flag = bad if self._label_flag is None else (self._label_flag | bad)
            self._labels = classes
        with torch.cuda.device(device):
            N.check(N.lib().dmn_unet_set_class_labels(self.native_handle(), None if self._labels is None
                                                      else self._labels.data_ptr()))

    def _version(self):
        return tuple((p.data_ptr(), p._version) for p in self.parameters())

    def native_handle(self) -> int:
        """Creates the DmnUnet handle / repacks the weights when parameters changed. The library works on the current
        HIP device: the handle, its packed weights and every call on it are bound to the device of the parameters."""
        lib = N.lib()
        dev = self._device()
        if self._handle is not None and self._handle_device != dev:
            self.release()   # the module moved to another GPU: packed weights and graphs belong to the old one
        with torch.cuda.device(dev):
            if self._handle is None:
                h = C.c_void_p()
                cfg = self._cfg()
                N.check(lib.dmn_unet_create(C.byref(cfg), C.byref(h)))
                N.check(lib.dmn_unet_set_plan_batch(h, self._plan_batch))
                if self._sample_chains is not None:
                    N.check(lib.dmn_unet_set_sample_chains(h, *self._sample_chains))
                self._handle = h
                self._handle_device = dev
            ver = self._version()
            if ver != self._weights_version:
                params = {k: v.detach().to("cpu", torch.float32).contiguous().numpy() for k, v in self.state_dict().items()}
                descs, keep = N.make_tensor_descs(params)
                N.check(lib.dmn_unet_load_weights(self._handle, descs, len(params)))
                del keep
                self._weights_version = ver
        return self._handle

    def set_compute_dtype(self, compute_dtype: str):
        if compute_dtype not in ("bf16", "fp32"):
            raise ValueError("compute_dtype must be 'bf16' or 'fp32'")
        if compute_dtype != self.compute_dtype:
            self.release()
            self.compute_dtype = compute_dtype
        return self

    def release(self):
        if self._handle is not None:
            with torch.cuda.device(self._handle_device):
                torch.cuda.synchronize()
                N.lib().dmn_unet_destroy(self._handle)
        self._handle = None
        self._handle_device = None
        self._weights_version = None
        self._workspace = None
        self._ws_key = None
        self._giveups_seen = 0   # a new handle counts from zero

    def __del__(self):
        try:
            if self._handle is not None:
                N.lib().dmn_unet_destroy(self._handle)
        except Exception:
            pass

    def workspace(self, B, H, W, device, sampling=False) -> torch.Tensor:
        key = (B, H, W, str(device), sampling, self._plan_batch, self._sample_chains)
        if self._ws_key != key:
            lib = N.lib()
            n = C.c_size_t()
            fn = lib.dmn_sample_workspace_bytes if sampling else lib.dmn_unet_workspace_bytes
            N.check(fn(self.native_handle(), B, H, W, C.byref(n)))
            self._workspace = torch.empty(n.value, dtype=torch.uint8, device=device)
            self._ws_key = key
        return self._workspace

    # ---- forward -----------------------------------------------------------------------------------------
    def forward(self, x: torch.Tensor, time: torch.Tensor, classes=None) -> torch.Tensor:
        if not x.is_cuda:
            raise RuntimeError("Unet.forward runs only on an MI355X device tensor (no CPU implementation)")
        if x.dim() != 4 or x.size(1) != self.channels:
            raise ValueError(f"expected [B, {self.channels}, H, W], got {tuple(x.shape)}")
        B, _, H, W = x.shape
        x = x.contiguous().float()
        if time.dim() != 1 or time.numel() != B:
            raise ValueError("time must have shape [B]")
        if time.is_floating_point():
            t, is_float = time.to(device=x.device, dtype=torch.float32).contiguous(), 1
        else:
            t, is_float = time.to(device=x.device, dtype=torch.int64).contiguous(), 0
        if x.device != self._device():
            raise RuntimeError(f"input on {x.device}, network on {self._device()}")
        handle = self.native_handle()
        self.set_class_labels(classes, B, x.device)
        ws = self.workspace(B, H, W, x.device)
        out = torch.empty((B, self.out_dim, H, W), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            N.check(N.lib().dmn_unet_forward(handle, x.data_ptr(), t.data_ptr(), is_float, out.data_ptr(), B, H, W,
                                             ws.data_ptr(), ws.numel(), N.current_stream_handle(x.device)))
        return out
