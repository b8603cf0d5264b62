"""Measurement (GPU): what a dependent launch costs when it does nothing -- 61 launches (one chain's step) of a kernel with
one store per workgroup, captured into a graph and replayed; on one stream, and on two streams side by side.
python tools/launch_floor.py"""
import ctypes as C
import sys
import threading

import torch

sys.path.insert(0, ".")
from diffusion_model_nemo_amd import _native as N  # noqa: E402

dev = torch.device("cuda", 0)
sink = [torch.zeros(65536, device=dev) for _ in range(2)]
streams = [torch.cuda.Stream(dev) for _ in range(2)]


def run(i, wgs, out):
    us = C.c_float()
    with torch.cuda.device(dev):
        N.check(N.lib().dmn_bench_launch_floor(61, wgs, 200, sink[i].data_ptr(), streams[i].cuda_stream, C.byref(us)))
    out[i] = us.value


for wgs in (1, 256, 512, 1024, 4096):
    one = [0.0, 0.0]
    run(0, wgs, one)
    two = [0.0, 0.0]
    th = [threading.Thread(target=run, args=(i, wgs, two)) for i in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    print(f"FLOOR {wgs:5d} workgroups per launch: one stream {one[0]:.2f} us per launch ({61 * one[0]:.0f} us per 61); "
          f"two streams side by side {two[0]:.2f} / {two[1]:.2f} us per launch", flush=True)
