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

// One round of the combination: this lane's (up to kCombineKeep) slots of group g are in v (zero where there is none),
// more slots (beyond kCombineKeep * seg) are fetched through load. Every lane of the wave takes part.
template <typename Load>
__device__ __forceinline__ void combine_round(const f32x4 (&v)[kCombineKeep], int g, bool live, int slots, int seg, int sl, float inv_count,
                                              float* sMR, Load load) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < kCombineKeep; ++k) s += v[k][0];
    const bool more = slots > kCombineKeep * seg;   // uniform
    if (more && live)
        for (int i = sl + kCombineKeep * seg; i < slots; i += seg) s += load(g, i)[0];
    const float mean = seg_sum(s, seg) * inv_count;
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < kCombineKeep; ++k) t += moments_term(v[k][0], v[k][1], v[k][2], mean);
    if (more && live)
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

// mean / rstd of groups [0, ng) from their partial moments, into sMR[2 * group] / sMR[2 * group + 1]. load(g, i) returns slot
// i of group g as {s, m2, n, -}. Called by every wave of the workgroup; no barrier inside. A wave reduces 64 / seg groups
// at once, seg lanes per group, so the slot loads of all its groups are in flight together. Lanes stride over the slots,
// then a fixed-order butterfly inside the segment for the sum and one for the squared deviations: the result depends on
// the slot VALUES and their order only, whether they come from a statistics buffer, from the granules of an in-launch
// exchange (handoff_group_stats follows the same scheme) or from LDS.
// pre: optional, the kCombineKeep slots of this lane for the FIRST round (group wave * gpw + lane / seg, slots sl + k * seg;
// entries past the group's slots or groups are ignored), fetched ahead by the caller (norm_issue): that round then
// contains no load at all, so the compiler can let it run while later loads of the caller are still in flight.
template <typename Load>
__device__ __forceinline__ void combine_groups(int ng, int slots, float inv_count, float* sMR, Load load, const f32x4* pre = nullptr) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, nwaves = blockDim.x >> 6;
    const int seg = combine_seg(slots);
    const int gpw = 64 / seg;
    const int sl = lane & (seg - 1);
    int g0 = wave * gpw;
    if (pre) {
        const int g = g0 + lane / seg;
        const bool live = g < ng;
        f32x4 v[kCombineKeep];
#pragma unroll
        for (int k = 0; k < kCombineKeep; ++k) {
            const bool have = live && sl + k * seg < slots;
#pragma unroll
            for (int e = 0; e < 3; ++e) v[k][e] = have ? pre[k][e] : 0.f;
            v[k][3] = 0.f;
        }
        if (g0 < ng) combine_round(v, g, live, slots, seg, sl, inv_count, sMR, load);   // wave-uniform condition
        g0 += nwaves * gpw;
    }
    for (; g0 < ng; g0 += nwaves * gpw) {
        const int g = g0 + lane / seg;
        const bool live = g < ng;
        f32x4 v[kCombineKeep];
#pragma unroll
        for (int k = 0; k < kCombineKeep; ++k) {
            v[k] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (live && sl + k * seg < slots) v[k] = load(g, sl + k * seg);
        }
        combine_round(v, g, live, slots, seg, sl, inv_count, sMR, load);
    }
}

// The global loads of a normalising prologue, issued as early as the caller likes (ahead of its bulk loads: vector loads
// return in order, so whatever is requested first can be consumed while the rest is still in flight): the channel
// parameters of this thread and the first kCombineKeep slots of the groups its lane serves in the first round of
// combine_groups. Every load is unconditional, from a clamped address: inside a branch the compiler loses count of the
// loads in flight and waits for all of them.
struct NormLoads {
    float pg[4], pb[4], pt[4];
    f32x4 v[kCombineKeep];
};
__device__ __forceinline__ void norm_issue(const NormIn& n, int b, int C, NormLoads& L) {
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int seg = combine_seg(n.slots);
    const int sl = lane & (seg - 1);
    const int g = wave * (64 / seg) + lane / seg;
    const f32x4* p = reinterpret_cast<const f32x4*>(n.stats) + ((size_t)b * n.G + (g < n.G ? g : n.G - 1)) * n.slots;
#pragma unroll
    for (int k = 0; k < kCombineKeep; ++k) {
        const int i = sl + k * seg;
        L.v[k] = p[i < n.slots ? i : n.slots - 1];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = tid + i * nthr;
        const int cc = c < C ? c : C - 1;
        L.pg[i] = n.gamma[cc];
        L.pb[i] = n.beta[cc];
        L.pt[i] = n.tbias ? n.tbias[(size_t)b * n.tb_stride + cc] : 0.f;
    }
}

// sCoef: 3*C floats (a | b | tb). sTmp: 2*G floats of scratch. C <= 4 * blockDim.x. Ends with a __syncthreads().
__device__ __forceinline__ void norm_finish(const NormIn& n, int b, int C, const NormLoads& L, float* sCoef, float* sTmp) {
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int slots = n.slots;
    const f32x4* base = reinterpret_cast<const f32x4*>(n.stats) + (size_t)b * n.G * slots;
    combine_groups(n.G, slots, n.inv_count, sTmp, [&](int g, int i) { return base[(size_t)g * slots + i]; }, L.v);
    __syncthreads();
    const int cpg = C / n.G;
    const bool pow2 = (cpg & (cpg - 1)) == 0;
    const int lg = 31 - __builtin_clz(cpg);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = tid + i * nthr;
        if (c < C) {
            const int g = pow2 ? c >> lg : c / cpg;
            const float aa = sTmp[2 * g + 1] * L.pg[i];
            sCoef[c] = aa;
            sCoef[C + c] = L.pb[i] - sTmp[2 * g] * aa;
            sCoef[2 * C + c] = L.pt[i];
        }
    }
    __syncthreads();
}

// Both halves back to back, for kernels that have nothing to put between them.
__device__ __forceinline__ void norm_coefs_to_lds(const NormIn& n, int b, int C, float* sCoef, float* sTmp) {
    NormLoads L;
    norm_issue(n, b, C, L);
    norm_finish(n, b, C, L, sCoef, sTmp);
}

// ---- the coefficients in registers -------------------------------------------------------------------------------------
// A thread that normalises PER consecutive channels c0 .. c0 + PER - 1 of every pixel it stages (the convolution's
// prologue) can keep their coefficients in registers: the groups are combined as in norm_finish (split over the waves),
// and after ONE barrier every thread picks mean / rstd of the groups of its channels from sTmp. Against norm_finish this
// removes the second barrier and the round trip of the coefficients through LDS. Same arithmetic: same bits.
template <int PER>
struct NormDirect {
    f32x4 v[kCombineKeep];
    f32x4 pg[PER / 4], pb[PER / 4], pt[PER / 4];
};
// gamma, beta and (when given) tbias + b * tb_stride must be 16-byte aligned, c0 a multiple of 4, c0 + PER <= C.
template <int PER>
__device__ __forceinline__ void norm_direct_issue(const NormIn& n, int b, int c0, NormDirect<PER>& L) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int seg = combine_seg(n.slots);
    const int sl = lane & (seg - 1);
    const int g = wave * (64 / seg) + lane / seg;
    const f32x4* p = reinterpret_cast<const f32x4*>(n.stats) + ((size_t)b * n.G + (g < n.G ? g : n.G - 1)) * n.slots;
#pragma unroll
    for (int k = 0; k < kCombineKeep; ++k) {
        const int i = sl + k * seg;
        L.v[k] = p[i < n.slots ? i : n.slots - 1];
    }
    const float* tb = n.tbias ? n.tbias + (size_t)b * n.tb_stride : n.gamma;   // any mapped address when there is none
#pragma unroll
    for (int q = 0; q < PER / 4; ++q) {
        L.pg[q] = *reinterpret_cast<const f32x4*>(n.gamma + c0 + 4 * q);
        L.pb[q] = *reinterpret_cast<const f32x4*>(n.beta + c0 + 4 * q);
        L.pt[q] = *reinterpret_cast<const f32x4*>(tb + c0 + 4 * q);
    }
}
// sTmp: 2 * G floats. ca / cb / ct: y = x * ca + cb (then the activation, then + ct). Contains one __syncthreads().
template <int PER>
__device__ __forceinline__ void norm_direct_finish(const NormIn& n, int b, int C, int c0, const NormDirect<PER>& L, float* sTmp,
                                                   float (&ca)[PER], float (&cb)[PER], float (&ct)[PER]) {
    const int slots = n.slots;
    const f32x4* base = reinterpret_cast<const f32x4*>(n.stats) + (size_t)b * n.G * slots;
    combine_groups(n.G, slots, n.inv_count, sTmp, [&](int g, int i) { return base[(size_t)g * slots + i]; }, L.v);
    __syncthreads();
    const int cpg = C / n.G;
#pragma unroll
    for (int q = 0; q < PER / 4; ++q) {
        const int g = (c0 + 4 * q) / cpg;   // a quad of channels never straddles a group (groups hold a multiple of 4)
        const float mean = sTmp[2 * g], rstd = sTmp[2 * g + 1];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float aa = rstd * L.pg[q][e];
            ca[4 * q + e] = aa;
            cb[4 * q + e] = L.pb[q][e] - mean * aa;
            ct[4 * q + e] = n.tbias ? L.pt[q][e] : 0.f;
        }
    }
}
