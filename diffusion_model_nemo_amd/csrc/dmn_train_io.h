// Typed access to the activation tensors of the training step: float (parity mode) or bfloat16 storage
// (converted to float on load, rounded to nearest even on store; all arithmetic stays float).
#pragma once
#include "dmn_common.h"

template <typename T> struct Io;
template <> struct Io<float> {
    static __device__ __forceinline__ float ld(const float* p) { return *p; }
    static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
    static __device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
    static __device__ __forceinline__ void st4(float* p, const f32x4& v) { *reinterpret_cast<f32x4*>(p) = v; }
    // one 16-byte access: VEC elements
    static constexpr int VEC = 4;
    static __device__ __forceinline__ void ldv(const float* p, float* v) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(p);
        v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
    }
    static __device__ __forceinline__ void stv(float* p, const float* v) { *reinterpret_cast<f32x4*>(p) = f32x4{v[0], v[1], v[2], v[3]}; }
};
template <> struct Io<__bf16> {
    static __device__ __forceinline__ float ld(const __bf16* p) { return (float)*p; }
    static __device__ __forceinline__ void st(__bf16* p, float v) { *p = (__bf16)v; }
    static __device__ __forceinline__ f32x4 ld4(const __bf16* p) {
        const bf16x4 v = *reinterpret_cast<const bf16x4*>(p);
        return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
    }
    static __device__ __forceinline__ void st4(__bf16* p, const f32x4& v) {
        bf16x4 o;
        o[0] = (__bf16)v[0];
        o[1] = (__bf16)v[1];
        o[2] = (__bf16)v[2];
        o[3] = (__bf16)v[3];
        *reinterpret_cast<bf16x4*>(p) = o;
    }
    static constexpr int VEC = 8;
    static __device__ __forceinline__ void ldv(const __bf16* p, float* v) {
        const bf16x8 t = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (float)t[i];
    }
    static __device__ __forceinline__ void stv(__bf16* p, const float* v) {
        bf16x8 o;
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = (__bf16)v[i];
        *reinterpret_cast<bf16x8*>(p) = o;
    }
};

// runs `call` with T = float (adt == 0) or __bf16 (adt == 1)
#define DMN_DISPATCH_T(adt, ...)      \
    do {                              \
        if ((adt) == 1) {             \
            using T = __bf16;         \
            __VA_ARGS__;              \
        } else {                      \
            using T = float;          \
            __VA_ARGS__;              \
        }                             \
    } while (0)
