"""Deterministic, name-hashed parameter generator (TEST INFRASTRUCTURE, not product code).

Fills a ``state_dict``-shaped mapping ``{name: shape}`` with reproducible float32 values that
depend only on the parameter NAME, its shape and a seed - not on the order in which a
framework happens to initialise modules. The same generator feeds the reference when golden
vectors are captured, the CPU oracle, and the GPU parity tests, so a 32 MB weight fixture
never has to be stored.

Scales are chosen so activations stay O(1) through the U-Net of
diffusion_model_nemo/modules/unet.py:14-120: conv / linear weights are uniform with
variance gain^2 / fan_in, norm scales sit near 1, biases and norm shifts near 0.
"""
import zlib
from collections import OrderedDict

import numpy as np

_M1 = np.uint64(0x9E3779B97F4A7C15)
_M2 = np.uint64(0xD6E8FEB86659FD93)


def _hash_uniform(name: str, n: int, seed: int) -> np.ndarray:
    """n float64 values in [0,1), a pure function of (name, index, seed)."""
    base = np.uint64((zlib.crc32(name.encode("utf-8")) + 0x1000193 * (seed + 1)) & 0xFFFFFFFF)
    i = np.arange(n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        x = (i + (base << np.uint64(20)) + base) * _M1
        x ^= x >> np.uint64(32)
        x = x * _M2
        x ^= x >> np.uint64(29)
        x = x * _M1
        x ^= x >> np.uint64(32)
    return (x >> np.uint64(40)).astype(np.float64) * (1.0 / 16777216.0)


def make_param(name: str, shape, seed: int = 0, gain: float = 1.0) -> np.ndarray:
    shape = tuple(int(s) for s in shape)
    n = int(np.prod(shape)) if len(shape) else 1
    u = _hash_uniform(name, n, seed) * 2.0 - 1.0  # [-1, 1)
    leaf = name.rsplit(".", 1)[-1]
    if len(shape) >= 2:
        if len(shape) == 4 and (name.endswith(".3.weight") and "ups." in name):
            # ConvTranspose2d weight [Cin, Cout, kh, kw]: each output sums Cin * (kh*kw/4) taps
            fan_in = shape[0] * shape[2] * shape[3] / 4.0
        else:
            fan_in = float(np.prod(shape[1:]))
        scale = gain * np.sqrt(3.0 / fan_in)
        v = u * scale
    elif leaf == "weight":      # norm scale
        v = 1.0 + 0.1 * u
    else:                       # bias / norm shift
        v = 0.1 * u
    return v.reshape(shape).astype(np.float32)


def make_state_dict(shapes, seed: int = 0, gain: float = 1.0):
    """shapes: mapping name -> shape. Returns OrderedDict name -> float32 ndarray."""
    out = OrderedDict()
    for name, shape in shapes.items():
        out[name] = make_param(name, shape, seed=seed, gain=gain)
    return out


def unet_param_shapes(dim=32, dim_mults=(1, 2, 4, 8), channels=3, learned_variance=False,
                      out_dim=None, resnet_block_order="bn_act_conv", num_classes=None,
                      heads=4, dim_head=32):
    """Parameter names and shapes of the reference U-Net with ``use_convnext=False`` and a time MLP
    (diffusion_model_nemo/modules/unet.py:41-120; parts/convnext.py:66-76; parts/mha.py:8-42;
    utils.py:77-89). Order follows the reference's ``state_dict()``."""
    shapes = OrderedDict()
    hidden = heads * dim_head
    time_dim = dim * 4

    def conv(name, cout, cin, k, bias=True):
        shapes[name + ".weight"] = (cout, cin, k, k)
        if bias:
            shapes[name + ".bias"] = (cout,)

    def norm(name, c):
        shapes[name + ".weight"] = (c,)
        shapes[name + ".bias"] = (c,)

    def linear(name, cout, cin):
        shapes[name + ".weight"] = (cout, cin)
        shapes[name + ".bias"] = (cout,)

    def resblock(name, cin, cout, time=True):
        if time:
            linear(name + ".mlp.1", cout, time_dim)
        conv(name + ".block1.proj", cout, cin, 3)
        norm(name + ".block1.norm", cout)
        conv(name + ".block2.proj", cout, cout, 3)
        norm(name + ".block2.norm", cout)
        if cin != cout:
            conv(name + ".res_conv", cout, cin, 1)

    def lin_attn(name, c):
        norm(name + ".fn.norm", c)
        conv(name + ".fn.fn.to_qkv", hidden * 3, c, 1, bias=False)
        conv(name + ".fn.fn.to_out.0", c, hidden, 1)
        norm(name + ".fn.fn.to_out.1", c)

    def attn(name, c):
        norm(name + ".fn.norm", c)
        conv(name + ".fn.fn.to_qkv", hidden * 3, c, 1, bias=False)
        conv(name + ".fn.fn.to_out", c, hidden, 1)

    conv("init_conv", dim, channels, 7)
    linear("time_mlp.1", time_dim, dim)
    linear("time_mlp.3", time_dim, time_dim)
    dims = [dim] + [dim * m for m in dim_mults]
    in_out = list(zip(dims[:-1], dims[1:]))
    nres = len(in_out)
    for i, (ci, co) in enumerate(in_out):
        resblock(f"downs.{i}.0", ci, co)
        resblock(f"downs.{i}.1", co, co)
        lin_attn(f"downs.{i}.2", co)
        if i < nres - 1:
            conv(f"downs.{i}.3", co, co, 4)
    mid = dims[-1]
    resblock("mid_block1", mid, mid)
    attn("mid_attn", mid)
    resblock("mid_block2", mid, mid)
    for i, (ci, co) in enumerate(reversed(in_out[1:])):
        resblock(f"ups.{i}.0", co * 2, ci)
        resblock(f"ups.{i}.1", ci, ci)
        lin_attn(f"ups.{i}.2", ci)
        # ConvTranspose2d(ci, ci, 4, 2, 1): weight [Cin, Cout, 4, 4]
        shapes[f"ups.{i}.3.weight"] = (ci, ci, 4, 4)
        shapes[f"ups.{i}.3.bias"] = (ci,)
    default_out = channels * (2 if learned_variance else 1)
    od = out_dim if out_dim is not None else default_out
    resblock("final_conv.0", dim, dim, time=False)
    if resnet_block_order == "bn_act_conv":
        norm("final_conv.1", dim)
        conv("final_conv.3", od, dim, 1)
    else:
        conv("final_conv.1", od, dim, 1)
    if num_classes is not None:
        shapes["class_embed.weight"] = (num_classes + 1, dim)
    return shapes
