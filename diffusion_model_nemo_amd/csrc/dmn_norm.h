// Device helpers for GroupNorm statistics that travel between kernels as per-workgroup partial MOMENTS, and for turning
// them into per-channel affine coefficients  y = x * a[c] + b[c]  (then optional SiLU, then + tb[c]) in LDS.
// Matches GroupNorm(eps=1e-5, biased variance) as used by the reference blocks
// (diffusion_model_nemo/parts/convnext.py:12,35; utils.py:89; parts/mha.py:42).
//
// A partial is {s = sum, m2 = sum of squared deviations from the partial's OWN mean, n = element count}; partials of
// one (image, group) are combined in two passes:
//     S = sum_i s_i,  mean = S / N,  M2 = sum_i [ m2_i + (s_i - n_i * mean)^2 / n_i ],  var = M2 / N.
// Unlike sum / sum-of-squares pairs (var = E[x^2] - E[x]^2 loses log10(mean^2 / var) digits to cancellation: a group
// whose mean is 50 standard deviations off zero keeps 3 of float's 7 digits) every term here is a sum of squares of
// DEVIATIONS, so the variance keeps its precision whatever the mean. The sums run in a fixed order (lane stride over
// the slots, then a fixed butterfly), so the result is bit-deterministic and the same for every consumer of a buffer.
#pragma once
#include "dmn_kernels.h"   // moments_term / seg_sum / moments_finish / moments_single live in dmn_common.h

// Lanes per group and slots a lane keeps in registers when partial moments are combined: up to 64 slots take 16 lanes with
// up to four slots each (all loads in flight together, the butterflies stay inside rows of 16 lanes: DPP only); more
// slots take a whole wave per group, and beyond 256 a lane walks the rest in a loop (reading them twice).
constexpr int kCombineKeep = 4;
__device__ __forceinline__ int combine_seg(int slots) { return slots <= 16 * kCombineKeep ? 16 : 64; }

// mean / rstd of groups [0, ng) from their partial moments, into sMR[2 * group] / sMR[2 * group + 1]. load(g, i) returns slot
// i of group g as {s, m2, n, -}. Called by every wave of the workgroup; no barrier inside. A wave reduces 64 / seg groups
// at once, seg lanes per group, so the slot loads of all its groups are in flight together. Lanes stride over the slots,
// then a fixed-order butterfly inside the segment for the sum and one for the squared deviations: the result depends on
// the slot VALUES and their order only, whether they come from a statistics buffer, from the granules of an in-launch
// exchange (handoff_group_stats follows the same scheme) or from LDS.
template <typename Load>
__device__ __forceinline__ void combine_groups(int ng, int slots, float inv_count, float* sMR, Load load) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, nwaves = blockDim.x >> 6;
    const int seg = combine_seg(slots);
    const int gpw = 64 / seg;
    const int sl = lane & (seg - 1);
    for (int g0 = wave * gpw; g0 < ng; g0 += nwaves * gpw) {
        const int g = g0 + lane / seg;
        const bool live = g < ng;
        f32x4 v[kCombineKeep];
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < kCombineKeep; ++k) {
            v[k] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (live && sl + k * seg < slots) v[k] = load(g, sl + k * seg);
        }
#pragma unroll
        for (int k = 0; k < kCombineKeep; ++k) s += v[k][0];
        if (live)
            for (int i = sl + kCombineKeep * seg; i < slots; i += seg) s += load(g, i)[0];
        const float mean = seg_sum(s, seg) * inv_count;
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < kCombineKeep; ++k) t += moments_term(v[k][0], v[k][1], v[k][2], mean);
        if (live)
            for (int i = sl + kCombineKeep * seg; i < slots; i += seg) {
                const f32x4 w = load(g, i);
                t += moments_term(w[0], w[1], w[2], mean);
            }
        const float M2 = seg_sum(t, seg);
        if (sl == 0 && live) {
            float rstd;
            moments_finish(M2, inv_count, rstd);
            sMR[2 * g] = mean;
            sMR[2 * g + 1] = rstd;
        }
    }
}

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
    const f32x4* base = reinterpret_cast<const f32x4*>(n.stats) + (size_t)b * n.G * n.slots;
    const int slots = n.slots;
    combine_groups(n.G, slots, n.inv_count, sTmp, [&](int g, int i) { return base[(size_t)g * slots + i]; });
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
