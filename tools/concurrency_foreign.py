"""Does the bf16 network forward change its results when a FOREIGN kernel stream (plain torch work) runs beside it?
Arguments: repetitions, kind of foreign work (none | elementwise | matmul | small)."""
import sys
import threading

import torch

sys.path.insert(0, ".")
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 100
KIND = sys.argv[2] if len(sys.argv) > 2 else "elementwise"
import os as _os
from diffusion_model_nemo_amd import _native as _N
if _os.environ.get("DMN_LIB_VARIANT"):
    _N.LIB_PATH = _N.LIB_PATH.replace("libdmn_hip.so", f"libdmn_hip_{_os.environ['DMN_LIB_VARIANT']}.so")
from diffusion_model_nemo_amd.modules import Unet  # noqa: E402

dev = torch.device("cuda", 0)
torch.manual_seed(1234)
net = Unet(input_dim=64, dim=32, dim_mults=[1, 2, 4], channels=3, resnet_block_groups=8, use_convnext=False, dropout=0.0,
           compute_dtype="bf16").to(dev).eval()
g = torch.Generator(device=dev).manual_seed(3)
x = torch.randn(32, 3, 64, 64, device=dev, generator=g)
t = torch.full((32,), 500, device=dev, dtype=torch.long)
ref = net(x, t).clone()
torch.cuda.synchronize()
streams = [torch.cuda.Stream(dev) for _ in range(2)]
stop = threading.Event()
bad = [0]
count = [0]


def forward_loop():
    with torch.cuda.stream(streams[0]):
        for _ in range(REPS):
            y = net(x, t)
            if float((y - ref).abs().max()) > 0:
                bad[0] += 1
    stop.set()


def foreign_loop():
    with torch.cuda.stream(streams[1]):
        a = torch.randn(4096, 4096, device=dev, dtype=torch.bfloat16)
        b = torch.randn(4096, 4096, device=dev, dtype=torch.bfloat16)
        big = torch.randn(16 << 20, device=dev)
        small = torch.randn(4096, device=dev)
        while not stop.is_set():
            if KIND == "elementwise":
                for _ in range(20):
                    big.mul_(1.0000001)
            elif KIND == "matmul":
                for _ in range(5):
                    torch.mm(a, b)
            elif KIND == "small":
                for _ in range(200):
                    small.mul_(1.0000001)
            count[0] += 1
            streams[1].synchronize()


th = [threading.Thread(target=forward_loop)]
if KIND != "none":
    th.append(threading.Thread(target=foreign_loop))
for x_ in th:
    x_.start()
for x_ in th:
    x_.join()
torch.cuda.synchronize()
print(f"foreign work {KIND}: {bad[0]} of {REPS} forwards differ from the forward run alone ({count[0]} foreign batches)", flush=True)
