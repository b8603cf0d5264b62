// Device helper: turn the partial group-norm statistics of a producer kernel into per-channel
// affine coefficients  y = x * a[c] + b[c]  (then optional SiLU, then + tb[c]) in LDS.
// Matches GroupNorm(eps=1e-5, biased variance) as used by the reference blocks
// (diffusion_model_nemo/parts/convnext.py:12,35; utils.py:89; parts/mha.py:42).
#pragma once
#include "dmn_kernels.h"

// sCoef: 3*C floats (a | b | tb). sTmp: 2*G floats of scratch. C <= 4 * blockDim.x. Ends with a __syncthreads().
// The per-channel parameters are requested before the statistics are reduced, so the two global round trips overlap.
__device__ __forceinline__ void norm_coefs_to_lds(const NormIn& n, int b, int C, float* sCoef, float* sTmp) {
    const int tid = threadIdx.x;
    const int nthr = blockDim.x;
    float pg[4], pb[4], pt[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = tid + i * nthr;
        pg[i] = pb[i] = pt[i] = 0.f;
        if (c < C) {
            pg[i] = n.gamma[c];
            pb[i] = n.beta[c];
            if (n.tbias) pt[i] = n.tbias[(size_t)b * n.tb_stride + c];
        }
    }
    // Statistics: a wave reduces 64 / seg groups at once, seg lanes per group (seg = 16, 32 or 64 >= slots when
    // possible), so the slot loads of all its groups are in flight together. Lanes stride over the slots, then a
    // fixed-order butterfly inside the segment (DPP inside rows of 16 lanes): deterministic.
    const int wave = tid >> 6, lane = tid & 63, nwaves = nthr >> 6;
    const int seg = n.slots <= 16 ? 16 : n.slots <= 32 ? 32 : 64;
    const int gpw = 64 / seg;
    const int sl = lane & (seg - 1);
    for (int g0 = wave * gpw; g0 < n.G; g0 += nwaves * gpw) {
        const int g = g0 + lane / seg;
        float s = 0.f, ss = 0.f;
        if (g < n.G) {
            const float* p = n.stats + ((size_t)(b * n.G + g) * n.slots) * 2;
            for (int i = sl; i < n.slots; i += seg) {
                const float2 v = *reinterpret_cast<const float2*>(p + 2 * i);
                s += v.x;
                ss += v.y;
            }
        }
        s = row16_sum(s);
        ss = row16_sum(ss);
        if (seg >= 32) {
            s += __shfl_xor(s, 16, 64);
            ss += __shfl_xor(ss, 16, 64);
        }
        if (seg == 64) {
            s += __shfl_xor(s, 32, 64);
            ss += __shfl_xor(ss, 32, 64);
        }
        if (sl == 0 && g < n.G) {
            const float mean = s * n.inv_count;
            float var = ss * n.inv_count - mean * mean;
            var = var > 0.f ? var : 0.f;
            sTmp[2 * g] = mean;
            sTmp[2 * g + 1] = __builtin_amdgcn_rsqf(var + 1e-5f);   // 1 ulp; the IEEE sqrt + divide pair is ~35 instructions
        }
    }
    __syncthreads();
    const int cpg = C / n.G;
    const bool pow2 = (cpg & (cpg - 1)) == 0;
    const int lg = 31 - __builtin_clz(cpg);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = tid + i * nthr;
        if (c < C) {
            const int g = pow2 ? c >> lg : c / cpg;
            const float aa = sTmp[2 * g + 1] * pg[i];
            sCoef[c] = aa;
            sCoef[C + c] = pb[i] - sTmp[2 * g] * aa;
            sCoef[2 * C + c] = pt[i];
        }
    }
    __syncthreads();
}
