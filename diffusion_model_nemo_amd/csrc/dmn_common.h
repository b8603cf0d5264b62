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

// Streaming store of a 16-byte (or 8-byte) vector: written through the L2 instead of left dirty in it. Every kernel
// here produces a tensor that the NEXT launch reads, possibly from another XCD, so the data has to reach memory
// before the kernel can retire; written through, that drain overlaps the kernel instead of extending its end.
template <typename V> __device__ __forceinline__ void store_stream(V* dst, const V& v) {
    if constexpr (sizeof(V) == 16) {
        __builtin_nontemporal_store(__builtin_bit_cast(u32x4, v), reinterpret_cast<u32x4*>(dst));
    } else {
        static_assert(sizeof(V) == 8, "8- or 16-byte vectors");
        __builtin_nontemporal_store(__builtin_bit_cast(u32x2, v), reinterpret_cast<u32x2*>(dst));
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
__device__ __forceinline__ float wave_sum(float v) {
    v = row16_sum(v);
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
    v = row16_max(v);
    v = fmaxf(v, __shfl_xor(v, 16, 64));
    v = fmaxf(v, __shfl_xor(v, 32, 64));
    return v;
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
