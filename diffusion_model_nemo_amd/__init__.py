"""MI355X-native implementation of the DDPM / score-SDE denoising hot path of titu1994/diffusion_model_nemo.

`diffusion_model_nemo_amd.modules`, `.models` and `.loss` mirror the reference's class names and constructor
arguments, so its Hydra `_target_` strings resolve here through `config.instantiate`; the computation runs in
`lib/libdmn_hip.so` (hand-written HIP kernels for gfx950, C-ABI in include/dmn.h).
"""
__version__ = "0.1.0"
