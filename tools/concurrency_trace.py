"""Locates the first launch of the network forward whose result differs when two handles run at the same time on two
streams: a checksum of the whole workspace is taken after every launch (dmn_test_unet_trace) and compared with the
checksums of the same forward run alone. Arguments: repetitions, dtype, DMN_UNET_PATHS mask."""
import collections
import os
import sys
import threading

import torch

sys.path.insert(0, ".")
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 30
DTYPE = sys.argv[2] if len(sys.argv) > 2 else "bf16"
os.environ["DMN_UNET_PATHS"] = sys.argv[3] if len(sys.argv) > 3 else "0"
from diffusion_model_nemo_amd import _native as N  # noqa: E402
if os.environ.get("DMN_LIB_VARIANT"):
    N.LIB_PATH = N.LIB_PATH.replace("libdmn_hip.so", f"libdmn_hip_{os.environ['DMN_LIB_VARIANT']}.so")
from diffusion_model_nemo_amd.modules import Unet  # noqa: E402

dev = torch.device("cuda", 0)
SIZE, MULTS, B, SLOTS = 64, (1, 2, 4), 32, 256


def build():
    torch.manual_seed(1234)
    return Unet(input_dim=SIZE, dim=32, dim_mults=list(MULTS), channels=3, resnet_block_groups=8, use_convnext=False,
                dropout=0.0, compute_dtype=DTYPE).to(dev).eval()


lib = N.lib()
nets = [build() for _ in range(2)]
g = torch.Generator(device=dev).manual_seed(3)
xs = [torch.randn(B, 3, SIZE, SIZE, device=dev, generator=g) for _ in range(2)]
t = torch.full((B,), 500, device=dev, dtype=torch.long)
slots = [torch.zeros(SLOTS, dtype=torch.int64, device=dev) for _ in range(2)]
refs, names = [], []
for i in range(2):
    nets[i](xs[i], t)   # creates the handle
    N.check(lib.dmn_test_unet_trace(nets[i].native_handle(), slots[i].data_ptr(), SLOTS))
    # the workspace is cleared before every forward, so the state after launch k depends on this forward alone
    nets[i]._workspace.zero_()
    y = nets[i](xs[i], t).clone()
    torch.cuda.synchronize()
    a = slots[i].clone()
    nets[i]._workspace.zero_()
    nets[i](xs[i], t)
    torch.cuda.synchronize()
    n = lib.dmn_test_unet_trace_count(nets[i].native_handle())
    assert torch.equal(a[:n], slots[i][:n]), "checksums differ between two forwards run alone"
    refs.append((y, a, n))
    names.append([lib.dmn_test_unet_trace_name(nets[i].native_handle(), k).decode() for k in range(n)])
print(f"{refs[0][2]} launches per forward", flush=True)

streams = [torch.cuda.Stream(dev) for _ in range(2)]
first = [collections.Counter(), collections.Counter()]
outbad = [0, 0]
nd = [[], []]


def work(i):
    y0, a, n = refs[i]
    with torch.cuda.stream(streams[i]):
        for _ in range(REPS):
            nets[i]._workspace.zero_()
            y = nets[i](xs[i], t)
            cur = slots[i][:n].clone()
            if float((y - y0).abs().max()) > 0:
                outbad[i] += 1
            diff = (cur != a[:n]).nonzero()
            if diff.numel():
                first[i][int(diff[0])] += 1
                nd[i].append(int(diff.numel()))


th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
for x in th:
    x.start()
for x in th:
    x.join()
torch.cuda.synchronize()
for i in range(2):
    print(f"handle {i}: {outbad[i]} of {REPS} outputs differ; first differing launch:")
    for k, c in sorted(first[i].items()):
        print(f"   launch {k:3d} x{c}: {names[i][k]}   (before it: {names[i][k - 1] if k else '-'})")
if os.environ.get("DMN_TRACE_LIST"):
    print("launch list:")
    for k, nm in enumerate(names[0]):
        print(f"   {k:3d} {nm}")
