"""Per-kernel parity of the HIP implicit-GEMM convolution (through the C-ABI test hook) against
torch.nn.functional on the CPU in fp32. Tolerances: the f32 path is an exact-f32 fma chain in a different
summation order than MKLDNN (abs 2e-5 * scale); the bf16 path rounds inputs/weights/outputs to bf16
(abs 2e-2 * scale, the reference's own bf16-autocast error band)."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import philox  # noqa: E402


def _N():
    from diffusion_model_nemo_amd import _native
    return _native


def rnd(shape, seed, scale=1.0):
    return torch.from_numpy(philox.normal_like(tuple(shape), seed, 0, 9)) * scale


def to_dev_nhwc(x_nchw, dtype):
    t = x_nchw.permute(0, 2, 3, 1).contiguous().cuda()
    return t.to(torch.bfloat16) if dtype == 1 else t


def from_dev_nhwc(t):
    return t.float().cpu().permute(0, 3, 1, 2).contiguous()


def moments(g):
    """Partial moments of the last axis in the kernels' slot format: {sum, squared deviations from the own mean, count, 0}."""
    s = g.sum(-1)
    n = torch.full_like(s, g.shape[-1])
    m2 = ((g - (s / n).unsqueeze(-1)) ** 2).sum(-1)
    return torch.stack([s, m2, n, torch.zeros_like(s)], -1).float()


def combine_moments(slots):
    """[..., slots, 4] partial moments -> [..., 3] = {sum, squared deviations from the mean of the union, count} (float64)."""
    slots = slots.double()
    s, m2, n = slots[..., 0], slots[..., 1], slots[..., 2]
    S, N = s.sum(-1), n.sum(-1)
    mean = (S / N).unsqueeze(-1)
    own = torch.where(n > 0, s / n.clamp(min=1), mean.expand_as(s))   # a slot of a row outside the image counts nothing
    M2 = (m2 + n * (own - mean) ** 2).sum(-1)
    return torch.stack([S, M2, N], -1)


def run_conv(dtype, x0, x1, w, b, stride, pad, transposed, res=None, pro=None, epi_G=0, tile=0, bn=0):
    N = _N()
    lib = N.lib()
    B, C0, H, W = x0.shape
    C1 = 0 if x1 is None else x1.shape[1]
    if transposed:
        Cout, kh, kw = w.shape[1], w.shape[2], w.shape[3]
        Ho, Wo = 2 * H, 2 * W
    else:
        Cout, kh, kw = w.shape[0], w.shape[2], w.shape[3]
        Ho = (H + 2 * pad - kh) // stride + 1
        Wo = (W + 2 * pad - kw) // stride + 1
    d0 = to_dev_nhwc(x0, dtype)
    d1 = None if x1 is None else to_dev_nhwc(x1, dtype)
    out = torch.zeros((B, Ho, Wo, Cout), device="cuda", dtype=torch.bfloat16 if dtype == 1 else torch.float32)
    dres = None if res is None else to_dev_nhwc(res, dtype)
    wn = np.ascontiguousarray(w.numpy(), dtype=np.float32)
    bn_ = None if b is None else np.ascontiguousarray(b.numpy(), dtype=np.float32)
    pg = pb = ptb = pst = None
    pG = pslots = psilu = 0
    if pro is not None:
        pg = np.ascontiguousarray(pro["gamma"].numpy(), dtype=np.float32)
        pb = np.ascontiguousarray(pro["beta"].numpy(), dtype=np.float32)
        ptb = None if pro.get("tbias") is None else pro["tbias"].cuda().contiguous()
        pst = pro["stats"].cuda().contiguous()
        pG, pslots, psilu = pro["G"], pro["slots"], pro["silu"]
    est = None
    if epi_G:
        est = torch.full((B * epi_G * 1024 * 4,), float("nan"), device="cuda", dtype=torch.float32)
    slots = C.c_int(0)
    torch.cuda.synchronize()
    N.check(lib.dmn_test_conv2d(
        dtype, d0.data_ptr(), None if d1 is None else d1.data_ptr(), C0, C1, B, H, W, wn.ctypes.data,
        None if bn_ is None else bn_.ctypes.data, Cout, kh, kw, stride, pad, 1 if transposed else 0, out.data_ptr(),
        None if dres is None else dres.data_ptr(), None if pg is None else pg.ctypes.data,
        None if pb is None else pb.ctypes.data, None if ptb is None else ptb.data_ptr(),
        None if pst is None else pst.data_ptr(), pG, pslots, psilu, None if est is None else est.data_ptr(), epi_G,
        C.byref(slots), tile, bn, None))
    torch.cuda.synchronize()
    stats = None
    if epi_G:
        stats = combine_moments(est[: B * epi_G * slots.value * 4].cpu().reshape(B, epi_G, slots.value, 4))
    return from_dev_nhwc(out), stats


def tol(dtype, ref):
    scale = max(1.0, float(ref.abs().max()))
    return (2e-5 if dtype == 0 else 2.5e-2) * scale


def q(x, dtype):
    """what the kernel sees: inputs rounded to the activation dtype"""
    return x.to(torch.bfloat16).float() if dtype == 1 else x


# every (Cout, Cin) pair of the 3x3 convolutions of the 64x64 network, at small spatial sizes
PAIRS_3X3 = [(32, 32), (64, 32), (64, 64), (32, 128), (128, 64), (128, 128), (64, 256), (256, 128), (256, 256), (128, 512)]


@pytest.mark.parametrize("dtype", [0, 1])
@pytest.mark.parametrize("cout,cin", PAIRS_3X3)
def test_conv3x3_all_channel_pairs(dtype, cout, cin):
    B, H, W = 2, 8, 8
    x = rnd((B, cin, H, W), 1)
    w = rnd((cout, cin, 3, 3), 2, (1.0 / (9 * cin)) ** 0.5)
    b = rnd((cout,), 3, 0.1)
    ref = F.conv2d(q(x, dtype), q(w, dtype), b, padding=1)
    got, _ = run_conv(dtype, x, None, w, b, 1, 1, False)
    assert (got - ref).abs().max() <= tol(dtype, ref)


@pytest.mark.parametrize("dtype", [0, 1])
@pytest.mark.parametrize("tile,bn", [(16, 32), (16, 64), (8, 32), (8, 64)])
@pytest.mark.parametrize("H,W", [(16, 16), (32, 16), (24, 40)])
def test_conv3x3_tile_configs_and_ragged_edges(dtype, tile, bn, H, W):
    B, cin, cout = 2, 32, 64
    x = rnd((B, cin, H, W), 4)
    w = rnd((cout, cin, 3, 3), 5, (1.0 / (9 * cin)) ** 0.5)
    b = rnd((cout,), 6, 0.1)
    ref = F.conv2d(q(x, dtype), q(w, dtype), b, padding=1)
    got, stats = run_conv(dtype, x, None, w, b, 1, 1, False, epi_G=8, tile=tile, bn=bn)
    assert (got - ref).abs().max() <= tol(dtype, ref)
    # groups are 8 consecutive channels: [B, 64, H, W] -> [B, 8, 8 * H * W]
    ref_stats = combine_moments(moments(ref.reshape(B, 8, -1).double()).unsqueeze(2))
    assert torch.equal(stats[..., 2], ref_stats[..., 2])                      # every element counted once
    for k in (0, 1):
        assert (stats[..., k] - ref_stats[..., k]).abs().max() <= 2e-3 * max(1.0, float(ref_stats[..., k].abs().max())) * (1 if dtype == 0 else 10)


@pytest.mark.parametrize("dtype", [0, 1])
def test_conv1x1_qkv_and_concat_sources(dtype):
    B, H, W = 2, 8, 8
    x0, x1 = rnd((B, 64, H, W), 7), rnd((B, 64, H, W), 8)
    w = rnd((384, 128, 1, 1), 9, (1.0 / 128) ** 0.5)
    ref = F.conv2d(torch.cat((q(x0, dtype), q(x1, dtype)), 1), q(w, dtype))
    got, _ = run_conv(dtype, x0, x1, w, None, 1, 0, False)
    assert (got - ref).abs().max() <= tol(dtype, ref)
    # 3x3 over two sources (decoder block1 reads cat(x, skip))
    w3 = rnd((64, 128, 3, 3), 10, (1.0 / (9 * 128)) ** 0.5)
    b3 = rnd((64,), 11, 0.1)
    ref = F.conv2d(torch.cat((q(x0, dtype), q(x1, dtype)), 1), q(w3, dtype), b3, padding=1)
    got, _ = run_conv(dtype, x0, x1, w3, b3, 1, 1, False)
    assert (got - ref).abs().max() <= tol(dtype, ref)


@pytest.mark.parametrize("dtype", [0, 1])
@pytest.mark.parametrize("C", [32, 128])
def test_downsample_and_upsample(dtype, C):
    B, H, W = 2, 16, 16
    x = rnd((B, C, H, W), 12)
    w = rnd((C, C, 4, 4), 13, (1.0 / (16 * C)) ** 0.5)
    b = rnd((C,), 14, 0.1)
    ref = F.conv2d(q(x, dtype), q(w, dtype), b, stride=2, padding=1)
    got, _ = run_conv(dtype, x, None, w, b, 2, 1, False)
    assert (got - ref).abs().max() <= tol(dtype, ref)
    wt = rnd((C, C, 4, 4), 15, (1.0 / (4 * C)) ** 0.5)
    ref = F.conv_transpose2d(q(x, dtype), q(wt, dtype), b, stride=2, padding=1)
    got, _ = run_conv(dtype, x, None, wt, b, 2, 1, True)
    assert got.shape == ref.shape
    assert (got - ref).abs().max() <= tol(dtype, ref)


@pytest.mark.parametrize("dtype", [0, 1])
@pytest.mark.parametrize("G,silu,use_tb", [(8, 1, True), (4, 1, False), (1, 0, False)])
def test_conv_with_groupnorm_prologue_and_residual(dtype, G, silu, use_tb):
    """conv(SiLU(GN(x)) + time_bias) + residual: Block2 of a ResnetBlock / the qkv projection behind PreNorm."""
    B, Cn, H, W = 2, 64, 16, 16
    x = rnd((B, Cn, H, W), 16) * 1.7 + 0.3
    gamma, beta = 1 + 0.1 * rnd((Cn,), 17), 0.1 * rnd((Cn,), 18)
    tb = rnd((B, Cn), 19, 0.3) if use_tb else None
    w = rnd((64, Cn, 3, 3), 20, (1.0 / (9 * Cn)) ** 0.5)
    b = rnd((64,), 21, 0.1)
    res = rnd((B, 64, H, W), 22)
    xq = q(x, dtype)
    h = F.group_norm(xq, G, gamma, beta, eps=1e-5)
    if silu:
        h = F.silu(h)
    if tb is not None:
        h = h + tb[:, :, None, None]
    ref = F.conv2d(q(h, dtype), q(w, dtype), b, padding=1) + q(res, dtype)
    g = xq.reshape(B, G, -1)
    # two slots per group: the consumer must add them
    half = g.shape[-1] // 2
    stats = torch.stack([moments(g[..., :half]), moments(g[..., half:])], dim=2)
    pro = dict(gamma=gamma, beta=beta, tbias=tb, stats=stats.contiguous(), G=G, slots=2, silu=silu)
    got, _ = run_conv(dtype, x, None, w, b, 1, 1, False, res=res, pro=pro)
    assert (got - ref).abs().max() <= tol(dtype, ref) * (1 if dtype == 0 else 2)


def test_philox_matches_cpu_generator():
    N = _N()
    B, per = 3, 3 * 8 * 8 + 0
    x = torch.empty((B, per), device="cuda", dtype=torch.float32)
    N.check(N.lib().dmn_fill_normal(x.data_ptr(), B, per, 1234567890123, 17, 5, 1.0, None))
    torch.cuda.synchronize()
    ref = philox.normal(1234567890123, 17, 0, np.arange(B) + 5, per)
    got = x.cpu().numpy()
    # float64 Box-Muller rounded once: identical except for rare last-ulp differences of log/sincos
    assert np.abs(got - ref).max() <= 1e-6
    assert (got == ref).mean() > 0.99


# ---- fused linear attention against the oracle (parts/mha.py:44-59) ---------------------------------------------
@pytest.mark.parametrize("dtype", [0, 1])
@pytest.mark.parametrize("n_side,cch", [(16, 32), (8, 64), (4, 128), (8, 256), (32, 32)])
def test_fused_linear_attention_matches_oracle(dtype, n_side, cch):
    from oracle import ref_cpu as R
    N = _N()
    lib = N.lib()
    B, n = 3, n_side * n_side
    x = rnd((B, cch, n_side, n_side), 11)
    gamma, beta = 1.0 + 0.1 * rnd((cch,), 12), 0.1 * rnd((cch,), 13)
    p = {"a.to_qkv.weight": rnd((384, cch, 1, 1), 14, cch ** -0.5) * 2.0,
         "a.to_out.0.weight": rnd((cch, 128, 1, 1), 15, 128 ** -0.5), "a.to_out.0.bias": 0.1 * rnd((cch,), 16),
         "a.to_out.1.weight": torch.ones(cch), "a.to_out.1.bias": torch.zeros(cch)}
    xq = q(x, dtype)
    xn = F.group_norm(xq, 1, gamma, beta, eps=1e-5)
    # reference value of to_out.0(...) before its GroupNorm: undo nothing, recompute the pieces
    qkv = F.conv2d(xn, q(p["a.to_qkv.weight"], dtype))
    qq, kk, vv = (t.reshape(B, 4, 32, n) for t in qkv.chunk(3, dim=1))
    ctx = torch.einsum("bhdn,bhen->bhde", kk.softmax(-1), vv)
    o = torch.einsum("bhde,bhdn->bhen", ctx, qq.softmax(-2) * 32 ** -0.5).reshape(B, 128, n_side, n_side)
    ref = F.conv2d(o, q(p["a.to_out.0.weight"], dtype), p["a.to_out.0.bias"])
    assert torch.allclose(F.group_norm(ref, 1), R.linear_attention({k: (q(v, dtype) if "weight" in k and v.ndim == 4 else v)
                                                                   for k, v in p.items()}, "a", xn), atol=1e-4)
    xd = to_dev_nhwc(x, dtype)
    xs = moments(xq.reshape(B, -1)).contiguous().cuda()
    y = torch.zeros((B, n_side, n_side, cch), device="cuda", dtype=torch.bfloat16 if dtype == 1 else torch.float32)
    ystats = torch.full((B * 1024 * 4,), float("nan"), device="cuda")
    scratch = torch.zeros((B * 4 * ((n + 31) // 32) * 1088,), device="cuda")
    slices = C.c_int(0)
    wqkv = np.ascontiguousarray(p["a.to_qkv.weight"].reshape(384, cch).numpy())
    wout = np.ascontiguousarray(p["a.to_out.0.weight"].reshape(cch, 128).numpy())
    bout = np.ascontiguousarray(p["a.to_out.0.bias"].numpy())
    N.check(lib.dmn_test_linattn(dtype, xd.data_ptr(), xs.data_ptr(), B, n, cch, np.ascontiguousarray(gamma.numpy()).ctypes.data,
                                 np.ascontiguousarray(beta.numpy()).ctypes.data, wqkv.ctypes.data, wout.ctypes.data,
                                 bout.ctypes.data, y.data_ptr(), ystats.data_ptr(), C.byref(slices), scratch.data_ptr(),
                                 scratch.numel() * 4, None))
    torch.cuda.synchronize()
    got = from_dev_nhwc(y)
    assert (got - ref).abs().max() <= (5e-5 if dtype == 0 else 4e-2) * max(1.0, float(ref.abs().max()))
    st = combine_moments(ystats[: B * slices.value * 4].reshape(B, slices.value, 4).cpu())
    want = combine_moments(moments(got.double().reshape(B, -1)).unsqueeze(1))
    assert torch.equal(st[:, 2], want[:, 2])
    if dtype == 0:
        assert torch.allclose(st, want, rtol=1e-4, atol=1e-2)
    else:   # statistics are taken from the fp32 values before rounding to bf16
        assert torch.allclose(st, want, rtol=3e-2, atol=1.0)


@pytest.mark.parametrize("dtype", [0, 1])
@pytest.mark.parametrize("n", [16, 48, 64, 128, 192, 256, 320])
def test_softmax_attention_core_matches_torch(dtype, n):
    """Attention core (parts/mha.py:21-28) on [B][n][384]: the matrix-core kernel (n = 64, 128, 192, 256) and the vector
    kernel (other n) against fp32 torch. Large logits on purpose (a row maximum far from 0 must not overflow)."""
    N = _N()
    B, heads, d = 3, 4, 32
    qkv = rnd((B, n, 3 * heads * d), 31 + n, 1.5)
    qkv[0, :, :heads * d] *= 3.0                      # sharper soft-max rows in image 0
    q, k, v = (t.reshape(B, n, heads, d).permute(0, 2, 1, 3) for t in qkv.chunk(3, dim=2))
    if dtype == 1:
        q, k, v = (t.to(torch.bfloat16).float() for t in (q, k, v))
    sim = torch.einsum("bhid,bhjd->bhij", q * d ** -0.5, k)
    attn = (sim - sim.amax(dim=-1, keepdim=True)).softmax(dim=-1)
    ref = torch.einsum("bhij,bhjd->bhid", attn, v).permute(0, 2, 1, 3).reshape(B, n, heads * d)
    dev = qkv.cuda().to(torch.bfloat16 if dtype == 1 else torch.float32).contiguous()
    out = torch.full((B, n, heads * d), float("nan"), device="cuda", dtype=dev.dtype)
    N.check(N.lib().dmn_test_attention(dtype, dev.data_ptr(), out.data_ptr(), B, n, heads, None))
    torch.cuda.synchronize()
    err = float((out.float().cpu() - ref).abs().max())
    assert err <= (5e-5 if dtype == 0 else 3e-2), (dtype, n, err)   # outputs reach |7|: 1e-5 relative in f32


@pytest.mark.parametrize("n,adt,bf16", [(256, 0, 1), (64, 0, 1), (128, 1, 1), (256, 0, 0), (48, 0, 1)])
def test_training_attention_core_forward_and_backward_match_autograd(n, adt, bf16):
    """parts/mha.py:20-29 as the training step runs it: out and d(qkv) against torch autograd in float64. bf16 operands
    on the matrix cores (64 <= n <= 256, n % 64 == 0): 2e-2 of each tensor's largest entry; the vector-unit kernels
    (fp32 operands, or a size the matrix-core kernels do not take): 2e-5."""
    from diffusion_model_nemo_amd import _native as N
    B, heads, d = 3, 4, 32
    g = torch.Generator().manual_seed(n + adt)
    qkv = torch.randn(B, n, 3 * heads * d, generator=g) * 1.5
    dout = torch.randn(B, n, heads * d, generator=g)
    ref_in = qkv.double().requires_grad_(True)
    q, k, v = [t.reshape(B, n, heads, d).permute(0, 2, 1, 3) for t in ref_in.chunk(3, dim=-1)]
    attn = torch.softmax(torch.einsum("bhid,bhjd->bhij", q * d ** -0.5, k), dim=-1)
    ref = torch.einsum("bhij,bhjd->bhid", attn, v).permute(0, 2, 1, 3).reshape(B, n, heads * d)
    ref.backward(dout.double())
    dt = torch.bfloat16 if adt else torch.float32
    dq = qkv.cuda().to(dt).contiguous()
    do = dout.cuda().to(dt).contiguous()
    out = torch.full((B, n, heads * d), float("nan"), device="cuda", dtype=dt)
    dqkv = torch.full((B, n, 3 * heads * d), float("nan"), device="cuda", dtype=dt)
    lse = torch.zeros(2 * B * heads * n, device="cuda")
    N.check(N.lib().dmn_test_attn_train(adt, dq.data_ptr(), do.data_ptr(), out.data_ptr(), dqkv.data_ptr(), lse.data_ptr(), B, n,
                                        heads, bf16, None))
    torch.cuda.synchronize()
    exact = not bf16 and not adt
    tol = 2e-5 if (exact or (not adt and n == 48)) else (2e-2 if not adt else 4e-2)
    for name, got, want in (("out", out, ref.detach()), ("dqkv", dqkv, ref_in.grad)):
        err = float((got.double().cpu() - want).abs().max())
        scale = float(want.abs().max())
        assert err <= tol * scale, (name, n, adt, bf16, err, scale)


def _block_tail(dtype, B, H, W, C_, G, out_G, fused, plan_B=64, tile=0, bn=0, seed=3, offset=0.0, want_ref=False):
    """block2.proj + closing SiLU(GroupNorm(h)) + res through the C-ABI hook; returns (out NCHW fp32, statistics of out
    combined over the partial slots [B][out_G][{sum, squared deviations, count}], give-ups of the hand-off)."""
    N = _N()
    lib = N.lib()
    tdt = torch.bfloat16 if dtype == 1 else torch.float32
    # offset: the input of the block and the output of its convolution sit `offset` standard deviations off zero
    x, res = rnd((B, C_, H, W), seed) + offset, rnd((B, C_, H, W), seed + 1)
    w = rnd((C_, C_, 3, 3), seed + 2, (1.0 / (9 * C_)) ** 0.5).numpy().astype(np.float32)
    bias = (rnd((C_,), seed + 3) + offset).numpy().astype(np.float32)
    pg, pb = (1.0 + 0.1 * rnd((C_,), seed + 4)).numpy().astype(np.float32), (0.1 * rnd((C_,), seed + 5)).numpy().astype(np.float32)
    tg, tb = (1.0 + 0.1 * rnd((C_,), seed + 6)).numpy().astype(np.float32), (0.1 * rnd((C_,), seed + 7)).numpy().astype(np.float32)
    dx, dres = to_dev_nhwc(x, dtype), to_dev_nhwc(res, dtype)
    xs = dx.float().view(B, H * W, G, C_ // G)
    pst = moments(xs.permute(0, 2, 1, 3).reshape(B, G, -1)).view(B, G, 1, 4).contiguous()   # statistics of the input, one slot
    out = torch.zeros((B, H, W, C_), device="cuda", dtype=tdt)
    h2 = torch.zeros_like(out)
    ost = torch.zeros((B * out_G * 4096 * 4,), device="cuda")
    slots, give = C.c_int(0), C.c_uint(0)
    N.check(lib.dmn_test_conv_block_tail(
        dtype, dx.data_ptr(), C_, B, H, W, w.ctypes.data, bias.ctypes.data, C_, pg.ctypes.data, pb.ctypes.data, pst.data_ptr(), G, 1,
        tg.ctypes.data, tb.ctypes.data, G, dres.data_ptr(), h2.data_ptr(), out.data_ptr(), ost.data_ptr(), out_G, C.byref(slots),
        fused, plan_B, tile, bn, C.byref(give), None))
    torch.cuda.synchronize()
    n = slots.value
    if want_ref:   # the same block in float64 on the CPU, from the inputs as the kernel sees them
        xr, rr = from_dev_nhwc(dx).double(), from_dev_nhwc(dres).double()
        wq = torch.from_numpy(w).to(tdt).double()
        h = F.silu(F.group_norm(xr, G, torch.from_numpy(pg).double(), torch.from_numpy(pb).double(), eps=1e-5))
        h2 = F.conv2d(h, wq, torch.from_numpy(bias).double(), padding=1)
        ref = F.silu(F.group_norm(h2, G, torch.from_numpy(tg).double(), torch.from_numpy(tb).double(), eps=1e-5)) + rr
        return out, ref
    return out, combine_moments(ost[:B * out_G * n * 4].view(B, out_G, n, 4).cpu()), give.value


@pytest.mark.parametrize("B,H,W,C_,fused", [
    (3, 16, 16, 128, 1),    # partial moments exchanged inside the launch (8 workgroups per image)
    (3, 16, 16, 128, 0),    # the same through the statistics buffer and the separate closing pass
    (3, 32, 32, 64, 0),     # 16x16-pixel tiles
    (3, 8, 8, 256, 1),      # one workgroup holds whole groups
    (2, 64, 64, 32, 0),     # full resolution: 16 partials per group
    (2, 24, 40, 64, 0),     # ragged tiles: partials of unequal counts
])
def test_block_tail_keeps_parity_when_group_means_dwarf_the_spread(B, H, W, C_, fused):
    """GroupNorm statistics travel as partial moments (sum, squared deviations from the partial's own mean, count), so a
    group whose mean lies 50 standard deviations off zero -- which a sum / sum-of-squares pair resolves to three digits --
    is normalised as accurately as one centred on zero: the fp32 block against float64 at the same tolerance (2e-4 of the
    output's scale) with and without the offset, through the prologue, both closing forms and the output statistics."""
    for offset in (0.0, 50.0):
        out, ref = _block_tail(0, B, H, W, C_, 8, 1, fused, offset=offset, want_ref=True)
        err = float((from_dev_nhwc(out).double() - ref).abs().max())
        assert err <= 2e-4 * max(1.0, float(ref.abs().max())), (offset, err)


@pytest.mark.parametrize("dtype", [1, 0])
@pytest.mark.parametrize("B,H,W,C_,G,out_G,tile,bn", [
    (6, 32, 32, 64, 8, 1, 0, 0),      # 16x16 tiles, 64 channels per workgroup: 4 workgroups per image exchange 8 groups
    (70, 32, 32, 64, 8, 1, 0, 0),     # more images than one launch of 256 workgroups takes: two launches
    (5, 32, 32, 64, 8, 8, 0, 0),      # statistics of the output in 8 groups
    (7, 16, 16, 128, 8, 1, 0, 0),     # 8 workgroups per image: the exchange's maximum
    (40, 16, 16, 128, 4, 1, 0, 0),
    (9, 8, 8, 256, 8, 1, 0, 0),       # one workgroup holds whole groups: no exchange
    (9, 8, 8, 256, 4, 1, 0, 0),       # groups of 64 channels = the channel tile
    (4, 8, 8, 256, 2, 1, 0, 0),       # groups of 128 channels: two workgroups per group
    (3, 24, 16, 64, 8, 1, 0, 0),      # ragged tiles
    (3, 32, 32, 32, 8, 1, 0, 0),      # the decoder's last level
    (4, 64, 64, 32, 8, 1, -1, 0),     # full resolution (tile -1: 32x16-pixel tiles allowed): 8 workgroups per image
    (3, 64, 64, 32, 8, 8, -1, 0),     # ... with the output's statistics in 8 groups (what the egress normalises with)
    (40, 64, 64, 32, 4, 1, -1, 0),    # more images than one launch takes
    (3, 48, 40, 32, 8, 1, -1, 0),     # ragged 32x16 tiles (2 x 3 per image)
    (2, 40, 56, 32, 8, 8, -1, 0),
])
def test_block_tail_inside_the_convolution_equals_the_two_launch_form(dtype, B, H, W, C_, G, out_G, tile, bn):
    """The ResnetBlock tail computed by block2.proj itself (GroupNorm partial sums exchanged between the workgroups of
    the launch) gives the bits of the convolution followed by the separate closing pass, and statistics of the output
    that agree to summation order; repeated launches agree with themselves (no stale partial sum is ever read)."""
    # a waiting workgroup may need at most half a CU's LDS: an fp32 halo tile beyond that
    # (e.g. 64 channels at 16x16 pixels: 93 KB, 256 channels at 8x8: 105 KB) is refused where workgroups would have to wait
    fused = 1
    if tile < 0:
        fused, tile = 3, 0
    try:
        a, sa, give = _block_tail(dtype, B, H, W, C_, G, out_G, fused, tile=tile, bn=bn)
    except _N().DmnError:
        halo = 612 if fused == 3 else 324 if min(H, W) >= 32 else 100
        assert dtype == 0 and halo * (C_ * 4 + 32) > 80 * 1024
        return
    b, sb, _ = _block_tail(dtype, B, H, W, C_, G, out_G, fused & 2, tile=tile, bn=bn)   # same tiles, separate closing pass
    assert give == 0
    assert torch.equal(a, b), float((a.float() - b.float()).abs().max())
    assert float((sa - sb).abs().max()) <= 1e-5 * float(sb.abs().max())
    # out is [B, H, W, C]: group g holds channels g * C/out_G .. of every pixel
    grouped = a.double().cpu().view(B, H * W, out_G, C_ // out_G).permute(0, 2, 1, 3).reshape(B, out_G, -1)
    want = combine_moments(moments(grouped).unsqueeze(2))
    assert torch.equal(sa[..., 2], want[..., 2])
    assert float((sa - want).abs().max()) <= 1e-4 * float(want.abs().max())
    for _ in range(5):
        again, _, give = _block_tail(dtype, B, H, W, C_, G, out_G, fused, tile=tile, bn=bn)
        assert give == 0 and torch.equal(again, a)


def test_block_tail_with_too_many_workgroups_per_image_is_refused():
    """Launches whose workgroups wait for each other stay small enough to be resident as a whole; a full-resolution
    image (16 workgroups) is left to the two-launch form."""
    N = _N()
    with pytest.raises(N.DmnError):
        _block_tail(1, 2, 64, 64, 32, 8, 1, 1)
    out, _, _ = _block_tail(1, 2, 64, 64, 32, 8, 1, 0)
    assert torch.isfinite(out.float()).all()


@pytest.mark.parametrize("dtype", [1, 0])
@pytest.mark.parametrize("B,side_h,side_w,cch", [
    (6, 32, 32, 64),     # 8 pixel slices per image exchange their sums (the maximum)
    (40, 32, 32, 32),    # more images than one launch of 256 workgroups takes: two launches
    (7, 16, 16, 128),
    (9, 8, 8, 256),      # one workgroup per image: no exchange, the whole attention in one launch
    (5, 8, 8, 128),
    (4, 7, 7, 64),       # one workgroup per image, ragged (49 of 64 staged rows)
    (4, 4, 4, 32),
    (3, 24, 40, 64),     # last slice ragged
    (3, 64, 64, 32),     # full resolution: 8 slices of 512 pixels (256-pixel slices would be 16, beyond the exchange)
    (34, 64, 64, 32),    # ... and more images than one launch takes
])
def test_linear_attention_closed_inside_its_second_launch_equals_the_separate_pass(dtype, B, side_h, side_w, cch):
    """Residual(PreNorm(LinearAttention)) with the closing GroupNorm(1)(y) + x computed by the attention's second launch
    (partial sums exchanged between the workgroups of the launch) gives the bits of the separate closing pass over y;
    repeated launches agree with themselves."""
    N = _N()
    lib = N.lib()
    n = side_h * side_w
    x = rnd((B, cch, side_h, side_w), 21)
    arr = lambda t: np.ascontiguousarray(t.numpy(), dtype=np.float32)   # noqa: E731
    gamma, beta = arr(1.0 + 0.1 * rnd((cch,), 22)), arr(0.1 * rnd((cch,), 23))
    og, ob = arr(1.0 + 0.1 * rnd((cch,), 24)), arr(0.1 * rnd((cch,), 25))
    wqkv, wout, bout = arr(rnd((384, cch), 26, cch ** -0.5) * 2.0), arr(rnd((cch, 128), 27, 128 ** -0.5)), arr(0.1 * rnd((cch,), 28))
    xd = to_dev_nhwc(x, dtype)
    xq = xd.float()
    xs = moments(xq.reshape(B, -1)).contiguous()
    scratch = torch.zeros((B * 4 * ((n + 31) // 32) * 1088,), device="cuda")

    def run(fused):
        y = torch.zeros_like(xd)
        out = torch.zeros_like(xd)
        ystats = torch.zeros((B * 1024 * 4,), device="cuda")
        give = C.c_uint(0)
        N.check(lib.dmn_test_linattn_block(dtype, xd.data_ptr(), xs.data_ptr(), B, n, cch, gamma.ctypes.data, beta.ctypes.data,
                                           wqkv.ctypes.data, wout.ctypes.data, bout.ctypes.data, og.ctypes.data, ob.ctypes.data,
                                           y.data_ptr(), ystats.data_ptr(), out.data_ptr(), scratch.data_ptr(), scratch.numel() * 4,
                                           fused, 64, C.byref(give), None))
        torch.cuda.synchronize()
        assert give.value == 0
        return out

    try:
        a = run(1)
    except N.DmnError:   # fp32 rows double the LDS of a workgroup: beyond half a CU the exchanging form is refused
        assert dtype == 0 and side_h * side_w > 64
        return
    b = run(0)
    assert torch.isfinite(a.float()).all() and float(a.float().abs().max()) > 0.5
    assert torch.equal(a, b), float((a.float() - b.float()).abs().max())
    # where one workgroup holds an image the whole attention is one launch: same bits as context pass + second launch
    assert torch.equal(run(3), a) and torch.equal(run(2), a)
    for _ in range(5):
        assert torch.equal(run(1), a)
