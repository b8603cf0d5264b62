// Shared device/host helpers for the gfx950 kernels of the DDPM / score-SDE denoising path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define DMN_WAVE 64


// Activation element type of a network instance: float (exact-f32 MFMA path, used for parity with the
// reference's fp32 CPU results) or __bf16 (bf16 MFMA path, fp32 accumulate, the fast path).
template <typename T> struct ElemTraits;
template <> struct ElemTraits<float> {
    static constexpr int kPer16B = 4;   // elements per 16-byte lane fragment
    static constexpr int kChunk = 16;   // channels per 64-byte LDS pixel row (one k-step)
};
template <> struct ElemTraits<__bf16> {
    static constexpr int kPer16B = 8;
    static constexpr int kChunk = 32;
};

__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(__bf16 v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ __bf16 from_f32<__bf16>(float v) { return (__bf16)v; }

__device__ __forceinline__ float silu_f(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }
// bf16 path: y = x*a + b as one fused multiply-add whatever the translation unit's contraction setting
// (the fp32 parity path keeps the separate multiply and add of the reference's kernels).
__device__ __forceinline__ float silu_affine_fast(float x, float a, float b, float c) {
    const float y = __builtin_fmaf(x, a, b);
    return __builtin_fmaf(y, __builtin_amdgcn_rcpf(1.0f + __expf(-y)), c);
}
// The fp32 parity path uses the full-precision exponential so SiLU agrees with the reference's
// ATen kernel to rounding; the bf16 path may use the fast one.
__device__ __forceinline__ float silu_precise(float v) { return v / (1.0f + expf(-v)); }

// 16-byte vector of T, unpacked to / packed from float.
template <typename T> struct Vec16;
template <> struct Vec16<float> {
    f32x4 v;
    static constexpr int N = 4;
    __device__ __forceinline__ float get(int i) const { return v[i]; }
    __device__ __forceinline__ void set(int i, float x) { v[i] = x; }
};
template <> struct Vec16<__bf16> {
    bf16x8 v;
    static constexpr int N = 8;
    __device__ __forceinline__ float get(int i) const { return (float)v[i]; }
    __device__ __forceinline__ void set(int i, float x) { v[i] = (__bf16)x; }
};

// Store of a 16-byte (or 8-byte) vector as ONE instruction at agent scope (`sc1`): the data goes through the L2 to the
// memory side while the kernel runs instead of waiting, dirty, for the write-back at its end, and the next launch --
// every kernel here produces a tensor that the NEXT launch reads, from any XCD -- finds it in the Infinity Cache.
// Measured on MI355X with two launch chains, per denoising step: default policy +0.3 %, default policy with the
// non-temporal hint +0.65 %, `sc0 sc1 nt` +5.7 %, `sc0 sc1` as agent scope. Inline assembly because the language has
// no scoped store of 16 bytes; the trailing s_nop covers the wait states between a wide vector-memory store and a
// write to its data registers, which the compiler inserts for stores it knows. dst must point to GLOBAL memory (a
// tensor in device memory): the instruction takes the pointer's value as a global address.
template <typename V> __device__ __forceinline__ void store_vec(V* dst, const V& v) {
    if constexpr (sizeof(V) == 16) {
        const u32x4 w = __builtin_bit_cast(u32x4, v);
        asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(dst), "v"(w) : "memory");
    } else {
        static_assert(sizeof(V) == 8, "8- or 16-byte vectors");
        const u32x2 w = __builtin_bit_cast(u32x2, v);
        asm volatile("global_store_dwordx2 %0, %1, off sc1\n\ts_nop 1" : : "v"(dst), "v"(w) : "memory");
    }
}

// One k-step of the 16x16 MFMA on two 16-byte fragments: D += A(16 x k) * B(k x 16).
// bf16: a single v_mfma_f32_16x16x32_bf16 (lane holds k = 8*(lane>>4) .. +7).
// f32 : four v_mfma_f32_16x16x4_f32; lane group g = lane>>4 holds channels 4g..4g+3 of a 16-channel
//       step and instruction s consumes channel 4g+s of every group. A and B use the same channel
//       permutation, so the sum over k is unchanged; numerics are an exact f32 fma chain.
__device__ __forceinline__ void mma_step(const Vec16<__bf16>& a, const Vec16<__bf16>& b, f32x4& acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, b.v, acc, 0, 0, 0);
}
__device__ __forceinline__ void mma_step(const Vec16<float>& a, const Vec16<float>& b, f32x4& acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.v[0], b.v[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.v[1], b.v[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.v[2], b.v[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.v[3], b.v[3], acc, 0, 0, 0);
}

// Cross-lane moves inside an aligned row of 16 lanes as DPP modifiers (a few cycles) instead of LDS-routed
// permutes (~100 cycles each). Every lane of the wave must be active.
template <int CTRL> __device__ __forceinline__ float dpp_move(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
// Sum / max over each aligned row of 16 lanes; every lane of the row receives the same, order-fixed result.
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_move<0xB1>(v);    // quad_perm [1,0,3,2]
    v += dpp_move<0x4E>(v);    // quad_perm [2,3,0,1]
    v += dpp_move<0x141>(v);   // row_half_mirror
    v += dpp_move<0x140>(v);   // row_mirror
    return v;
}
__device__ __forceinline__ float row16_max(float v) {
    v = fmaxf(v, dpp_move<0xB1>(v));
    v = fmaxf(v, dpp_move<0x4E>(v));
    v = fmaxf(v, dpp_move<0x141>(v));
    v = fmaxf(v, dpp_move<0x140>(v));
    return v;
}
// Sum / max over the wave: the four row results are read as scalars (v_readlane, a few cycles each) and combined as
// (r0 + r1) + (r2 + r3) -- the values and order of a butterfly over lanes xor 16, xor 32, without its two LDS-routed
// permutes (~100 cycles each, one behind the other, at the very end of a kernel's epilogue).
__device__ __forceinline__ float wave_sum(float v) {
    const int iv = __builtin_bit_cast(int, row16_sum(v));
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 48));
    return (r0 + r1) + (r2 + r3);
}
__device__ __forceinline__ float wave_max(float v) {
    const int iv = __builtin_bit_cast(int, row16_max(v));
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 48));
    return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}

// ---- GroupNorm statistics as partial moments (layout, combination rule and rationale: dmn_norm.h) -----------------
// A partial is {s = sum, m2 = sum of squared deviations from the partial's own mean, n = element count}.
// Value of the first lane of each aligned row of 16 lanes, in every lane of that row (four scalar reads and three
// selects: no LDS routing). Reads the four lanes whatever the exec mask is.
__device__ __forceinline__ float row16_first(float v) {
    const int iv = __builtin_bit_cast(int, v);
    const int s0 = __builtin_amdgcn_readlane(iv, 0), s1 = __builtin_amdgcn_readlane(iv, 16);
    const int s2 = __builtin_amdgcn_readlane(iv, 32), s3 = __builtin_amdgcn_readlane(iv, 48);
    const int q = (threadIdx.x & 63) >> 4;
    return __builtin_bit_cast(float, q == 0 ? s0 : q == 1 ? s1 : q == 2 ? s2 : s3);
}

// Sums of deviations from a shift K (any value of the right magnitude: a sample of the distribution) over n elements,
// sd = sum (x - K), qd = sum (x - K)^2, as a partial {s, m2, n}. The squares summed are those of differences of the order
// of the spread, so qd - sd^2 / n cancels digits of the spread's scale only, not of the mean's.
__device__ __forceinline__ void moments_from_shifted(float sd, float qd, float n, float K, float& s, float& m2) {
    s = n > 0.f ? __builtin_fmaf(n, K, sd) : 0.f;
    const float m = n > 0.f ? qd - sd * sd * __builtin_amdgcn_rcpf(n) : 0.f;
    m2 = m > 0.f ? m : 0.f;
}

// Contribution of one partial to the M2 of the union whose mean is `mean`.
__device__ __forceinline__ float moments_term(float s, float m2, float n, float mean) {
    const float d = __builtin_fmaf(-n, mean, s);   // n * (mean_i - mean) with one rounding
    return n > 0.f ? m2 + d * d * __builtin_amdgcn_rcpf(n) : 0.f;
}

// Sum over an aligned segment of `seg` lanes (16, 32 or 64), the same order-fixed result in every lane of the segment.
// (seg is wave-uniform. Segments of 32 or 64 lanes combine the row results read as scalars, in the order of a butterfly
// over lanes xor 16, xor 32.)
__device__ __forceinline__ float seg_sum(float v, int seg) {
    v = row16_sum(v);
    if (seg >= 32) {
        const int iv = __builtin_bit_cast(int, v);
        const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 16));
        const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 48));
        const float lo = r0 + r1, hi = r2 + r3;
        v = seg == 64 ? lo + hi : ((threadIdx.x & 32) ? hi : lo);
    }
    return v;
}

// mean / rstd of a group from its total sum S and total M2 (both already combined over all partials).
__device__ __forceinline__ void moments_finish(float M2, float inv_count, float& rstd) {
    float var = M2 * inv_count;
    var = var > 0.f ? var : 0.f;
    rstd = __builtin_amdgcn_rsqf(var + 1e-5f);   // 1 ulp; the IEEE sqrt + divide pair is ~35 instructions
}

// A group that has ONE partial (the workgroup that needs the statistics holds every element of the group): the
// arithmetic of the general combination with one slot, so both forms give the same bits.
__device__ __forceinline__ void moments_single(float s, float m2, float n, float inv_count, float& mean, float& rstd) {
    mean = s * inv_count;
    moments_finish(moments_term(s, m2, n, mean), inv_count, rstd);
}

// ---- error plumbing shared by the C-ABI translation units -------------------------------------
enum DmnStatus {
    DMN_OK = 0,
    DMN_ERR_INVALID = -1,     // bad argument / unsupported shape
    DMN_ERR_HIP = -2,         // a HIP runtime call failed
    DMN_ERR_WORKSPACE = -3,   // workspace too small
    DMN_ERR_WEIGHTS = -4,     // missing / mis-shaped parameter
    DMN_ERR_LIMIT = -5        // size beyond a stated maximum
};
void dmn_set_error(const char* fmt, ...);
#define DMN_HIP_CHECK(expr)                                                                   \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) {                                                               \
            dmn_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return DMN_ERR_HIP;                                                               \
        }                                                                                     \
    } while (0)
#define DMN_REQUIRE(cond, code, ...)      \
    do {                                  \
        if (!(cond)) {                    \
            dmn_set_error(__VA_ARGS__);   \
            return code;                  \
        }                                 \
    } while (0)
