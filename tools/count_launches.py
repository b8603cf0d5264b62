import ctypes as C, sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from test_gpu_unet_sampling import build_unet64
from diffusion_model_nemo_amd import _native as N
u = build_unet64("bf16")
x = torch.randn(32, 3, 64, 64, device="cuda"); t = torch.full((32,), 5, device="cuda")
u(x, t); torch.cuda.synchronize()
slots = torch.zeros(256, dtype=torch.int64, device="cuda")
lib = N.lib(); h = u.native_handle()
N.check(lib.dmn_test_unet_trace(h, slots.data_ptr(), 256))
u(x, t); torch.cuda.synchronize()
n = lib.dmn_test_unet_trace_count(h)
print("launch sites in one evaluation of 32 images:", n)
for k in range(n):
    print(k, lib.dmn_test_unet_trace_name(h, k).decode()[:70])
N.check(lib.dmn_test_unet_trace(h, None, 0))
