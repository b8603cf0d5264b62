// Counter-based Gaussian noise on the device: Philox4x32-10 + Box-Muller in float64, rounded to float32
// once. Bit-for-bit the same function as the CPU checker's generator (keyed by seed, step, stream, GLOBAL
// image index and the NCHW element index inside the image), so a sampling chain does not depend on the
// memory layout, on the batch split over ranks, or on a framework RNG. The reference draws with
// torch.randn / torch.randn_like (diffusion_model_nemo/modules/gaussian_diffusion.py:164,177).
#pragma once
#include "dmn_common.h"

__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3,
                                               unsigned k0, unsigned k1, unsigned out[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = 0xD2511F53ull * c0;
        const unsigned long long p1 = 0xCD9E8D57ull * c2;
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0;
        const unsigned n1 = (unsigned)p1;
        const unsigned n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1;
        const unsigned n3 = (unsigned)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// Four N(0,1) values for elements 4*quad .. 4*quad+3 of image `img`.
__device__ __forceinline__ void normal4(unsigned long long seed, unsigned step, unsigned stream,
                                         unsigned img, unsigned quad, float z[4]) {
    unsigned r[4];
    philox4x32_10(quad, img, step, stream, (unsigned)seed, (unsigned)(seed >> 32), r);
    const double k = 1.0 / 4294967296.0;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const double u1 = ((double)r[2 * h] + 1.0) * k;
        const double u2 = (double)r[2 * h + 1] * k;
        const double rad = sqrt(-2.0 * log(u1));
        double sn, cs;
        sincos(6.283185307179586476925286766559 * u2, &sn, &cs);
        z[2 * h] = (float)(rad * cs);
        z[2 * h + 1] = (float)(rad * sn);
    }
}
