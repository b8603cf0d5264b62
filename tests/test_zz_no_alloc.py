"""dmn_sde_corrector_step_static (the entry point that allocates nothing) against dmn_sde_corrector_step
(caller-provided workspace): bit-identical x_out on two calls in a row, the second with a smaller input, through
the C-ABI. Collected last (file name) so that a failure here hides no other test."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run_pair(B, C, H, W, use_norms, seed):
    from diffusion_model_nemo_amd import _native as N
    from diffusion_model_nemo_amd.modules import VESDE, PredictorCorrectorSampler
    lib = N.lib()
    s = PredictorCorrectorSampler(predictor="reverse_diffusion", corrector="langevin" if use_norms else "ald", snr=0.16)
    s.update_sde(VESDE(0.01, 50.0, 20))
    tab = s.step_table("cuda")
    g = torch.Generator(device="cuda").manual_seed(seed)
    x = torch.randn(B, C, H, W, device="cuda", generator=g)
    net = torch.randn(B, C, H, W, device="cuda", generator=g) * 0.3
    want = torch.full_like(x, float("nan"))
    got = torch.full_like(x, float("nan"))
    scratch = torch.empty(lib.dmn_sde_scratch_bytes(B, C, H, W), dtype=torch.uint8, device="cuda")
    row, inner = 7, 1
    N.check(lib.dmn_sde_corrector_step(tab.data_ptr(), row, x.data_ptr(), net.data_ptr(), None, want.data_ptr(), B, C, H, W,
                                       0.16, inner, use_norms, 11, 3, scratch.data_ptr(), None))
    N.check(lib.dmn_sde_corrector_step_static(tab.data_ptr(), row, x.data_ptr(), net.data_ptr(), None, got.data_ptr(), B, C,
                                              H, W, 0.16, inner, use_norms, 11, 3, None))
    torch.cuda.synchronize()
    assert torch.isfinite(want).all()
    assert torch.equal(got, want), (B, C, H, W, use_norms, float((got - want).abs().max()))


def test_static_corrector_step_equals_workspace_form_twice():
    _run_pair(8, 3, 64, 64, 1, seed=1)     # 12 slots per image
    _run_pair(3, 3, 24, 16, 1, seed=2)     # smaller: fewer images, 2 slots per image; stale partials beside it
    _run_pair(2, 3, 8, 8, 0, seed=3)       # annealed Langevin: fixed step, the partial sums are not used


def test_static_corrector_step_rejects_oversize_and_bad_arguments():
    from diffusion_model_nemo_amd import _native as N
    lib = N.lib()
    x = torch.zeros(1, 3, 64, 64, device="cuda")
    tab = torch.zeros(1, N.TABLE_COLS, device="cuda")
    # 1024 images x 12 slots x 8 bytes > 64 KiB: refused before anything is launched
    rc = lib.dmn_sde_corrector_step_static(tab.data_ptr(), 0, x.data_ptr(), x.data_ptr(), None, x.data_ptr(), 1024, 3, 64, 64,
                                           0.16, 0, 1, 0, 0, None)
    assert rc == -5 and b"scratch" in lib.dmn_last_error()
    assert lib.dmn_sde_corrector_step_static(None, 0, x.data_ptr(), x.data_ptr(), None, x.data_ptr(), 1, 3, 64, 64, 0.16, 0,
                                             1, 0, 0, None) == -1
