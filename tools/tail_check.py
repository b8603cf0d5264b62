"""GPU check of the in-launch block tails: fused (default) against the separate closing passes (DMN_UNET_PATHS=128)."""
import os
import sys
import time

import torch

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from test_gpu_unet_sampling import build_unet  # noqa: E402


def build64(dtype, mask):
    if mask:
        os.environ["DMN_UNET_PATHS"] = str(mask)
    u = build_unet(dtype)
    os.environ.pop("DMN_UNET_PATHS", None)
    return u


g = torch.Generator(device="cuda").manual_seed(4)
for dtype in ("fp32", "bf16"):
    for B, S in ((4, 64), (64, 64), (3, 32), (2, 128), (5, 48)):
        x = torch.randn(B, 3, S, S, device="cuda", generator=g)
        t = torch.randint(0, 1000, (B,), device="cuda", generator=g)
        uf, uu = build64(dtype, 0), build64(dtype, 128)
        a, b = uf(x, t), uu(x, t)
        torch.cuda.synchronize()
        d = (a - b).abs()
        rep = all(torch.equal(uf(x, t), a) for _ in range(20))
        print(f"{dtype} B={B} {S}x{S}: fused vs separate max {float(d.max()):.3e} mean {float(d.mean()):.3e} "
              f"|out| {float(b.abs().max()):.2f} repeatable {rep} giveups {uf.handoff_giveups()}", flush=True)
        assert rep and uf.handoff_giveups() == 0
        assert float(d.max()) < (1e-4 if dtype == "fp32" else 0.15)
print("ok")
