"""Counter-based Gaussian noise for parity tests (TEST INFRASTRUCTURE, not product code).

Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11)
followed by a Box-Muller transform evaluated in float64 and rounded to float32 once.
The HIP kernels in ``diffusion_model_nemo_amd/csrc/dmn_rng.h`` implement exactly the same
function, so the CPU oracle and the GPU path see the same x_T and the same per-step noise
without either side depending on a framework generator.

Keying: ``key = (seed_lo, seed_hi)``; ``counter = (elem >> 2, image_index, step, stream)``
where ``elem`` is the NCHW linear index of the element inside its image (so the value does
not depend on the memory layout or on how a batch is sharded over ranks), ``image_index``
is the GLOBAL image index, ``step`` the diffusion step the draw belongs to and ``stream`` a
small integer separating independent draws of one step (prior / predictor / corrector k).
Each counter yields 4 normals; element ``elem`` takes normal ``elem & 3``.

The reference draws its noise with ``torch.randn`` / ``torch.randn_like``
(diffusion_model_nemo/modules/gaussian_diffusion.py:164,177); those generators differ between
CPU and GPU, which is why parity runs inject this noise on both sides instead.
"""
import numpy as np

PHILOX_M0 = np.uint64(0xD2511F53)
PHILOX_M1 = np.uint64(0xCD9E8D57)
PHILOX_W0 = np.uint32(0x9E3779B9)
PHILOX_W1 = np.uint32(0xBB67AE85)

STREAM_PRIOR = 0
STREAM_STEP = 1       # DDPM p_sample noise / SDE predictor noise
STREAM_CORRECTOR = 2  # + corrector inner step index


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised Philox4x32-10. All inputs uint32 arrays (broadcastable); returns 4 uint32 arrays."""
    c0 = np.asarray(c0, dtype=np.uint32)
    c1 = np.asarray(c1, dtype=np.uint32)
    c2 = np.asarray(c2, dtype=np.uint32)
    c3 = np.asarray(c3, dtype=np.uint32)
    c0, c1, c2, c3 = np.broadcast_arrays(c0, c1, c2, c3)
    k0 = np.uint32(k0)
    k1 = np.uint32(k1)
    mask = np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0 = PHILOX_M0 * c0.astype(np.uint64)
        p1 = PHILOX_M1 * c2.astype(np.uint64)
        hi0 = (p0 >> np.uint64(32)).astype(np.uint32)
        lo0 = (p0 & mask).astype(np.uint32)
        hi1 = (p1 >> np.uint64(32)).astype(np.uint32)
        lo1 = (p1 & mask).astype(np.uint32)
        n0 = hi1 ^ c1 ^ k0
        n1 = lo1
        n2 = hi0 ^ c3 ^ k1
        n3 = lo0
        c0, c1, c2, c3 = n0, n1, n2, n3
        with np.errstate(over="ignore"):
            k0 = np.uint32((int(k0) + int(PHILOX_W0)) & 0xFFFFFFFF)
            k1 = np.uint32((int(k1) + int(PHILOX_W1)) & 0xFFFFFFFF)
    return c0, c1, c2, c3


def _box_muller(r0, r1):
    u1 = (r0.astype(np.float64) + 1.0) * (1.0 / 4294967296.0)  # (0, 1]
    u2 = r1.astype(np.float64) * (1.0 / 4294967296.0)          # [0, 1)
    rad = np.sqrt(-2.0 * np.log(u1))
    ang = 2.0 * np.pi * u2
    return rad * np.cos(ang), rad * np.sin(ang)


def normal(seed, step, stream, image_indices, elems_per_image):
    """N(0,1) float32 array [len(image_indices), elems_per_image] (NCHW element order)."""
    image_indices = np.asarray(image_indices, dtype=np.uint32)
    n4 = (elems_per_image + 3) // 4
    c0 = np.arange(n4, dtype=np.uint32)[None, :]
    c1 = image_indices[:, None]
    r0, r1, r2, r3 = philox4x32_10(c0, c1, np.uint32(step), np.uint32(stream),
                                   seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    a, b = _box_muller(r0, r1)
    c, d = _box_muller(r2, r3)
    out = np.stack([a, b, c, d], axis=-1).reshape(len(image_indices), n4 * 4)
    return out[:, :elems_per_image].astype(np.float32)


def normal_like(shape, seed, step, stream, image_offset=0):
    """N(0,1) float32 array of NCHW ``shape``; image b has global index ``image_offset + b``."""
    b = shape[0]
    per = int(np.prod(shape[1:]))
    idx = np.arange(b, dtype=np.uint32) + np.uint32(image_offset)
    return normal(seed, step, stream, idx, per).reshape(shape)
