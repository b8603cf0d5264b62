"""Which launches of the network forward read LDS they did not write: the forward is run alone, with every CU's LDS
overwritten by a pattern in front of each launch, for several patterns; a launch whose workspace checksum depends on
the pattern read some of it. Arguments: dtype, image size, dim_mults..."""
import sys

import torch

sys.path.insert(0, ".")
DTYPE = sys.argv[1] if len(sys.argv) > 1 else "bf16"
SIZE = int(sys.argv[2]) if len(sys.argv) > 2 else 64
MULTS = [int(a) for a in sys.argv[3:]] or [1, 2, 4]
from diffusion_model_nemo_amd import _native as N  # noqa: E402
from diffusion_model_nemo_amd.modules import Unet  # noqa: E402

dev = torch.device("cuda", 0)
B = 32
torch.manual_seed(1234)
net = Unet(input_dim=SIZE, dim=32, dim_mults=MULTS, channels=3, resnet_block_groups=8, use_convnext=False, dropout=0.0,
           compute_dtype=DTYPE).to(dev).eval()
lib = N.lib()
g = torch.Generator(device=dev).manual_seed(3)
x = torch.randn(B, 3, SIZE, SIZE, device=dev, generator=g)
t = torch.full((B,), 500, device=dev, dtype=torch.long)
slots = torch.zeros(256, dtype=torch.int64, device=dev)
net(x, t)
h = net.native_handle()
N.check(lib.dmn_test_unet_trace(h, slots.data_ptr(), 256))
res = {}
outs = {}
for pat in (0x00000000, 0xFFFFFFFF, 0x3F803F80, 0x42F642F6, 0x7F7F7F7F):
    N.check(lib.dmn_test_unet_trace_poison(h, pat if pat else 0))
    if pat == 0:   # pattern 0 means "off" to the library: use a pattern of zeros through a second name
        N.check(lib.dmn_test_unet_trace_poison(h, 0x80000000))   # -0.0f / bf16 pairs (-0.0, 0.0)
        pat = 0x80000000
    net._workspace.zero_()
    y = net(x, t).clone()
    torch.cuda.synchronize()
    n = lib.dmn_test_unet_trace_count(h)
    res[pat] = slots[:n].clone()
    outs[pat] = y
names = [lib.dmn_test_unet_trace_name(h, k).decode() for k in range(n)]
pats = list(res)
base = pats[0]
print(f"{DTYPE} size {SIZE} mults {MULTS}: {n} launches")
for p in pats[1:]:
    d = (res[p] != res[base]).nonzero().flatten().cpu().tolist()
    nonfin = int((~torch.isfinite(outs[p])).sum())
    print(f"pattern {p:#010x} vs {base:#010x}: {len(d)} launch checksums differ, first {d[:8]}; output max |diff| "
          f"{float((outs[p] - outs[base]).abs().nan_to_num(1e9).max()):.3g}, non-finite outputs {nonfin}")
    for k in d[:4]:
        print(f"      launch {k}: {names[k]}")
