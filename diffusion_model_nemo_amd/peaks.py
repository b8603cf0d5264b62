"""Measured ceilings of the GPU the library runs on: streaming copy / read bandwidth at several sizes and the bare
matrix-core rates, to quote next to the nominal figures (8 TB/s HBM3E, 2.5 PFLOP/s dense bf16, 157 TFLOP/s fp32 matrix).
Measurement only; nothing on the sampling or training path calls this."""
import ctypes as C

import torch

from . import _native as N


def measure(dev=None):
    dev = dev or torch.device("cuda", 0)
    lib = N.lib()
    stream = torch.cuda.current_stream(dev).cuda_stream
    ms = C.c_float()
    res = {"device": torch.cuda.get_device_name(dev), "stream": [], "mfma": []}
    big = 1 << 30
    src = torch.empty(big, dtype=torch.uint8, device=dev).random_(0, 255)
    dst = torch.empty(big, dtype=torch.uint8, device=dev)
    # 16.78 MB is one level-0 activation of the headline configuration (64 x 64x64 x 32 channels, bf16): a copy of it
    # moves exactly the bytes of the dominant convolution launch
    for nbytes, iters in ((64 * 64 * 64 * 32 * 2, 200), (1 << 26, 100), (1 << 28, 40), (big, 20)):
        for mode, name in ((0, "copy"), (1, "read")):
            N.check(lib.dmn_bench_stream(mode, dst.data_ptr(), src.data_ptr(), nbytes, iters, stream, C.byref(ms)))
            traffic = nbytes * (2 if mode == 0 else 1)
            res["stream"].append({"kind": name, "bytes": nbytes, "traffic_bytes": traffic, "launch_us": ms.value * 1e3,
                                  "GBs": traffic / (ms.value / 1e3) / 1e9})
    sink = torch.empty(65536 * 8, dtype=torch.float32, device=dev)
    fl = C.c_double()
    for kind, name in ((0, "v_mfma_f32_16x16x32_bf16"), (1, "v_mfma_f32_16x16x4_f32")):
        for wgs in (1, 2, 4):
            N.check(lib.dmn_bench_mfma(kind, wgs, 4000, 10, sink.data_ptr(), stream, C.byref(ms), C.byref(fl)))
            res["mfma"].append({"instruction": name, "waves_per_simd": wgs, "launch_ms": ms.value,
                                "TFLOPs": fl.value / (ms.value / 1e3) / 1e12})
    torch.cuda.synchronize(dev)
    return res
