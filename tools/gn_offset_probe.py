"""Experiment (GPU): error of the fp32 handle against the CPU oracle when every GroupNorm input carries a mean that is large
against its spread (a constant added to the bias of every convolution that feeds a GroupNorm).
usage: python tools/gn_offset_probe.py [offsets ...]"""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from diffusion_model_nemo_amd.modules import Unet  # noqa: E402
from oracle import philox, ref_cpu as R  # noqa: E402
from oracle.weights import make_state_dict, unet_param_shapes  # noqa: E402

offsets = [float(a) for a in sys.argv[1:]] or [0.0, 10.0, 30.0, 100.0]
torch.set_num_threads(16)
x = torch.from_numpy(philox.normal_like((2, 3, 64, 64), 5, 0, 7))
t = torch.tensor([17, 803])
for off in offsets:
    sd = make_state_dict(unet_param_shapes())
    for k in sd:
        if k.endswith(".proj.bias") or k.endswith("to_out.0.bias"):
            sd[k] = (sd[k] + np.float32(off)).astype(np.float32)
    p = R.to_torch(sd)
    with torch.no_grad():
        ref = R.unet_forward(p, x, t)
    u = Unet(input_dim=64, dim=32, dim_mults=[1, 2, 4, 8], use_convnext=False, dropout=0.0, compute_dtype="fp32")
    u.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    u = u.cuda().eval()
    got = u(x.cuda(), t.cuda()).cpu()
    d = (got - ref).abs()
    print(f"offset {off:6.1f}: max abs err {float(d.max()):.3e} mean {float(d.mean()):.3e} |ref|max {float(ref.abs().max()):.2f}", flush=True)
