"""Experiment (GPU): one 64-image chain vs. two (four) concurrent 32- (16-) image chains on separate streams."""
import sys
import time

import torch

sys.path.insert(0, ".")
from diffusion_model_nemo_amd.modules import GaussianDiffusion, Unet  # noqa: E402

dev = torch.device("cuda", 0)
T = int(sys.argv[1]) if len(sys.argv) > 1 else 200


def build():
    torch.manual_seed(1234)
    return Unet(input_dim=64, dim=32, dim_mults=[1, 2, 4, 8], channels=3, resnet_block_groups=8, use_convnext=False,
                dropout=0.0, compute_dtype="bf16").to(dev).eval()


_all_streams = []


def distinct_streams(n):
    """n torch streams that a timing probe shows on hardware queues of their own (as far as the device has them)."""
    import ctypes as C
    from diffusion_model_nemo_amd import _native as N
    out = []
    for _ in range(16):
        if len(out) == n:
            break
        st = torch.cuda.Stream(dev)
        _all_streams.append(st)
        with torch.cuda.stream(st):
            torch.zeros(4, device=dev).add_(1)
        sh = C.c_int(0)
        ok = True
        for o in out:
            N.check(N.lib().dmn_streams_share_queue(st.cuda_stream, o.cuda_stream, C.byref(sh)))
            if sh.value:
                ok = False
                break
        if ok:
            out.append(st)
    print(f"  {len(out)} of {n} streams on separate queues ({len(_all_streams)} created)", flush=True)
    while len(out) < n:
        out.append(torch.cuda.Stream(dev))
    return out


def run(nsplit, skew_us=0.0):
    B = 64 // nsplit
    nets = [build() for _ in range(nsplit)]
    gds = [GaussianDiffusion(timesteps=T, schedule_name="cosine") for _ in range(nsplit)]
    streams = distinct_streams(nsplit)
    outs = [None] * nsplit

    def chain(i):
        with torch.cuda.stream(streams[i]):
            if skew_us and i:   # delay chain i by i * skew_us on the device, so the chains run out of phase
                torch.cuda._sleep(int(i * skew_us * 1e-6 * 2.0e9))
            outs[i] = gds[i].p_sample_loop(nets[i], (B, 3, 64, 64), device=dev, seed=7, image_offset=i * B,
                                           return_device_tensor=True)

    def once(threads=True):
        if not threads:   # first pass: handle creation, weight upload and graph capture must not overlap another capture
            for i in range(nsplit):
                chain(i)
            torch.cuda.synchronize()
            return
        # one host thread per chain (the native loop releases the GIL): a chain's graph launches may block its thread
        import threading
        th = [threading.Thread(target=chain, args=(i,)) for i in range(nsplit)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        torch.cuda.synchronize()

    once(threads=False)
    t0 = time.perf_counter()
    once()
    el = time.perf_counter() - t0
    print(f"{nsplit} x B={B} skew {skew_us:.0f} us: {el * 1e3 / T:.3f} ms per step for 64 images ({64 / (el * 1000 / T):.1f} img/s at 1000 steps)", flush=True)
    return torch.cat(outs)


a = run(1)
b = run(2)
for sk in (100.0, 200.0, 300.0, 400.0):
    b2 = run(2, sk)
c = run(4)
c2 = run(4, 200.0)
print("max |1x - 2x| =", float((a - b).abs().max()), " max |1x - 2x skewed| =", float((a - b2).abs().max()),
      " max |1x - 4x| =", float((a - c).abs().max()))
