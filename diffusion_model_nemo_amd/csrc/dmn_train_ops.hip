// Non-GEMM kernels of the training step: GroupNorm forward / backward, activations, the attention cores with their
// backward passes, loss, layout changes and the optimiser. NHWC float32 throughout. Reference semantics:
// torch.nn.GroupNorm / SiLU / GELU (diffusion_model_nemo/parts/convnext.py:11-23, modules/unet.py:61-66),
// LinearAttention / Attention (parts/mha.py:16-30,44-59), DiffusionLoss (loss/simple_loss.py:37-45),
// q_sample (modules/gaussian_diffusion.py:104-116).
#include "dmn_common.h"
#include "dmn_train.h"
#include "dmn_train_io.h"

#define DMN_LAUNCH_CHECK()                                                        \
    do {                                                                          \
        hipError_t _e = hipGetLastError();                                        \
        if (_e != hipSuccess) {                                                   \
            dmn_set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(_e), __FILE__, __LINE__); \
            return DMN_ERR_HIP;                                                   \
        }                                                                         \
    } while (0)

namespace {

constexpr float kGnEps = 1e-5f;

// sum over the 256 threads of a workgroup; every thread receives the result. sh: >= 4 floats, reusable after return.
__device__ __forceinline__ float block_sum(float v, float* sh) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}

// the same in float64 (GroupNorm statistics); sh: >= 4 doubles
__device__ __forceinline__ double block_sum_d(double v, double* sh) {
    for (int o = 1; o < 64; o <<= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}

__device__ __forceinline__ float sigmoid_f(float z) { return 1.0f / (1.0f + expf(-z)); }
__device__ __forceinline__ float silu_grad(float z) {
    const float sg = sigmoid_f(z);
    return sg * (1.0f + z * (1.0f - sg));
}
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad(float x) {
    return 0.5f * (1.0f + erff(x * 0.70710678118654752f)) + x * 0.3989422804014327f * expf(-0.5f * x * x);
}

// ---- GroupNorm ----------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void gn_stats_kernel(const T* __restrict__ x, int HW, int C, int G, double* __restrict__ stats) {
    __shared__ double sh[4];
    const int bg = blockIdx.x, b = bg / G, g = bg - b * G;
    const int cg4 = (C / G) >> 2;
    const long long total = (long long)HW * cg4;
    const T* base = x + (size_t)b * HW * C + g * (C / G);
    double s = 0.0, ss = 0.0;   // four elements at a time in float32, the running sums in float64 (dmn_train.h)
    for (long long i = (long long)blockIdx.y * 256 + threadIdx.x; i < total; i += (long long)gridDim.y * 256) {
        const long long p = i / cg4;
        const int c4 = (int)(i - p * cg4);
        const f32x4 v = Io<T>::ld4(base + p * C + 4 * c4);
        s += (double)((v[0] + v[1]) + (v[2] + v[3]));
        ss += (double)((v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]));
    }
    s = block_sum_d(s, sh);
    ss = block_sum_d(ss, sh);
    if (threadIdx.x == 0) {
        unsafeAtomicAdd(stats + 2 * bg, s);
        unsafeAtomicAdd(stats + 2 * bg + 1, ss);
    }
}

__device__ __forceinline__ void gn_mean_rstd(const double* stats, int bg, float inv_cnt, float& mean, float& rstd) {
    const double s = stats[2 * bg], ss = stats[2 * bg + 1];
    const double m = s * (double)inv_cnt;
    const double var = fmax(ss * (double)inv_cnt - m * m, 0.0);   // float64: the cancellation costs digits it can spare
    mean = (float)m;
    rstd = 1.0f / sqrtf((float)var + kGnEps);
}

// grid (blocks, B): a workgroup stays inside one image, so the statistics of its output need one atomic pair
template <typename T> __global__ __launch_bounds__(256) void gn_apply_kernel(const GnArgs a) {
    const T* x = reinterpret_cast<const T*>(a.x);
    const T* res = reinterpret_cast<const T*>(a.res);
    T* y = reinterpret_cast<T*>(a.y);
    __shared__ double sh[4];
    const int c4n = a.C >> 2, b = blockIdx.y;
    const long long total = (long long)a.HW * c4n;
    const int Cg = a.C / a.G;
    const float inv_cnt = 1.0f / ((float)Cg * (float)a.HW);
    const size_t img = (size_t)b * a.HW * a.C;
    double s1 = 0.0, s2 = 0.0;
    if (256 % c4n == 0) {
        // A thread keeps its four channels (the stride of the walk is a multiple of the channel vectors per pixel): the
        // group's mean / rstd and the channel parameters are fetched once, and four pixels' loads are in flight together.
        const int c = 4 * (threadIdx.x % c4n), ppb = 256 / c4n;
        float mean, rstd;
        gn_mean_rstd(a.stats, b * a.G + c / Cg, inv_cnt, mean, rstd);
        const f32x4 ga = *reinterpret_cast<const f32x4*>(a.gamma + c), be = *reinterpret_cast<const f32x4*>(a.beta + c);
        f32x4 tbv = f32x4{0.f, 0.f, 0.f, 0.f};
        if (a.tb) tbv = *reinterpret_cast<const f32x4*>(a.tb + (size_t)b * a.tb_ld + c);
        const T* xp = x + img + c;
        const T* rp = res ? res + img + c : nullptr;
        T* yp = y + img + c;
        auto one = [&](const f32x4& v, const f32x4& rv, int p) {
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float z = (v[e] - mean) * rstd * ga[e] + be[e];
                if (a.act) z = z * sigmoid_f(z);
                o[e] = z + tbv[e] + rv[e];
            }
            Io<T>::st4(yp + (size_t)p * a.C, o);
            s1 += (double)((o[0] + o[1]) + (o[2] + o[3]));
            s2 += (double)((o[0] * o[0] + o[1] * o[1]) + (o[2] * o[2] + o[3] * o[3]));
        };
        const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};
        int p = blockIdx.x * ppb + threadIdx.x / c4n;
        const int st = gridDim.x * ppb;
        for (; p + 3 * st < a.HW; p += 4 * st) {
            f32x4 v[4], rv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                v[u] = Io<T>::ld4(xp + (size_t)(p + u * st) * a.C);
                rv[u] = rp ? Io<T>::ld4(rp + (size_t)(p + u * st) * a.C) : zero;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) one(v[u], rv[u], p + u * st);
        }
        for (; p < a.HW; p += st) one(Io<T>::ld4(xp + (size_t)p * a.C), rp ? Io<T>::ld4(rp + (size_t)p * a.C) : zero, p);
    } else
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long pix = i / c4n;
        const int c = 4 * (int)(i - pix * c4n);
        float mean, rstd;
        gn_mean_rstd(a.stats, b * a.G + c / Cg, inv_cnt, mean, rstd);
        const size_t off = img + pix * a.C + c;
        const f32x4 v = Io<T>::ld4(x + off);
        const f32x4 ga = *reinterpret_cast<const f32x4*>(a.gamma + c), be = *reinterpret_cast<const f32x4*>(a.beta + c);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float z = (v[e] - mean) * rstd * ga[e] + be[e];
            if (a.act) z = z * sigmoid_f(z);
            if (a.tb) z += a.tb[(size_t)b * a.tb_ld + c + e];
            o[e] = z;
        }
        if (res) {
            const f32x4 rv = Io<T>::ld4(res + off);
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] += rv[e];
        }
        Io<T>::st4(y + off, o);
        s1 += (double)((o[0] + o[1]) + (o[2] + o[3]));
        s2 += (double)((o[0] * o[0] + o[1] * o[1]) + (o[2] * o[2] + o[3] * o[3]));
    }
    if (a.ystats) {
        s1 = block_sum_d(s1, sh);
        s2 = block_sum_d(s2, sh);
        if (threadIdx.x == 0) {
            unsafeAtomicAdd(a.ystats + 2 * b, s1);
            unsafeAtomicAdd(a.ystats + 2 * b + 1, s2);
        }
    }
}

// per (image, channel): sum dz * xhat, sum dz, sum dy. grid (splits, B); C/4 divides 256 (checked by the launcher): a thread keeps four
// fixed channels and walks pixels with a stride, four pixels' loads issued before the first use.
template <typename T> __global__ __launch_bounds__(256) void gn_bwd_sums_kernel(const GnArgs a) {
    const T* x = reinterpret_cast<const T*>(a.x);
    const T* dy = reinterpret_cast<const T*>(a.dy);
    __shared__ float sh[256 * 12];
    const int b = blockIdx.y, c4n = a.C >> 2, tid = threadIdx.x;
    const int c = 4 * (tid % c4n), pl = tid / c4n, npl = 256 / c4n;
    const int Cg = a.C / a.G;
    const float inv_cnt = 1.0f / ((float)Cg * (float)a.HW);
    float mean, rstd;
    gn_mean_rstd(a.stats, b * a.G + c / Cg, inv_cnt, mean, rstd);
    const f32x4 ga = *reinterpret_cast<const f32x4*>(a.gamma + c), be = *reinterpret_cast<const f32x4*>(a.beta + c);
    float s0[4] = {0, 0, 0, 0}, s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
    const size_t img = (size_t)b * a.HW * a.C;
    const T* xp = x + img + c;
    const T* dp = dy + img + c;
    auto one = [&](const f32x4& v, const f32x4& d) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float xh = (v[e] - mean) * rstd;
            float dz = d[e];
            if (a.act) dz *= silu_grad(xh * ga[e] + be[e]);
            s0[e] += dz * xh;
            s1[e] += dz;
            s2[e] += d[e];
        }
    };
    int p = blockIdx.x * npl + pl;
    const int st = gridDim.x * npl;
    for (; p + 3 * st < a.HW; p += 4 * st) {
        f32x4 v[4], d[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            v[u] = Io<T>::ld4(xp + (size_t)(p + u * st) * a.C);
            d[u] = Io<T>::ld4(dp + (size_t)(p + u * st) * a.C);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) one(v[u], d[u]);
    }
    for (; p < a.HW; p += st) one(Io<T>::ld4(xp + (size_t)p * a.C), Io<T>::ld4(dp + (size_t)p * a.C));
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        sh[tid * 12 + e] = s0[e];
        sh[tid * 12 + 4 + e] = s1[e];
        sh[tid * 12 + 8 + e] = s2[e];
    }
    __syncthreads();
    // thread (channel group, which of 12) sums over the pixel lanes
    for (int o = tid; o < c4n * 12; o += 256) {
        const int cg = o / 12, k = o - cg * 12;
        float t = 0.f;
        for (int l = 0; l < npl; ++l) t += sh[(l * c4n + cg) * 12 + k];
        const int ch = 4 * cg + (k & 3), which = k >> 2;
        unsafeAtomicAdd(a.sums + ((size_t)b * a.C + ch) * 3 + which, t);
    }
}

template <typename T> __global__ __launch_bounds__(256) void gn_bwd_apply_kernel(const GnArgs a) {
    const T* x = reinterpret_cast<const T*>(a.x);
    const T* dy = reinterpret_cast<const T*>(a.dy);
    T* dx = reinterpret_cast<T*>(a.dx);
    __shared__ float S1[64], S2[64];
    __shared__ float P1[1024], P2[1024];   // C <= 1024 (checked by the launcher)
    const int b = blockIdx.y, tid = threadIdx.x, Cg = a.C / a.G, c4n = a.C >> 2;
    // gamma * sums per channel by all threads (every load in flight at once), then one thread per group adds its
    // channels from LDS in channel order: the sums of the plain loop, without one memory round trip per channel
    for (int c = tid; c < a.C; c += 256) {
        const float g = a.gamma[c];
        P1[c] = g * a.sums[((size_t)b * a.C + c) * 3 + 1];
        P2[c] = g * a.sums[((size_t)b * a.C + c) * 3 + 0];
    }
    __syncthreads();
    if (tid < a.G) {
        float t1 = 0.f, t2 = 0.f;
        for (int c = tid * Cg; c < (tid + 1) * Cg; ++c) {
            t1 += P1[c];
            t2 += P2[c];
        }
        S1[tid] = t1;
        S2[tid] = t2;
    }
    if (blockIdx.x == 0) {   // parameter gradients of this image (summed over images by the atomics) and its time-bias share
        for (int c = tid; c < a.C; c += 256) {
            const float* sm = a.sums + ((size_t)b * a.C + c) * 3;
            unsafeAtomicAdd(a.dgamma + c, sm[0]);
            unsafeAtomicAdd(a.dbeta + c, sm[1]);
            if (a.dtb) a.dtb[(size_t)b * a.tb_ld + c] += sm[2];
        }
    }
    __syncthreads();
    const float inv_cnt = 1.0f / ((float)Cg * (float)a.HW);
    const size_t img = (size_t)b * a.HW * a.C;
    const long long total = (long long)a.HW * c4n;
    if (256 % c4n == 0) {   // fixed channels per thread, as in the forward kernel
        const int c = 4 * (tid % c4n), g = c / Cg, ppb = 256 / c4n;
        float mean, rstd;
        gn_mean_rstd(a.stats, b * a.G + g, inv_cnt, mean, rstd);
        const float m1 = S1[g] * inv_cnt, m2 = S2[g] * inv_cnt;
        const f32x4 ga = *reinterpret_cast<const f32x4*>(a.gamma + c), be = *reinterpret_cast<const f32x4*>(a.beta + c);
        const T* xp = x + img + c;
        const T* dp = dy + img + c;
        T* op = dx + img + c;
        const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};
        auto one = [&](const f32x4& v, const f32x4& d, f32x4 o, int p) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float xh = (v[e] - mean) * rstd;
                float dz = d[e];
                if (a.act) dz *= silu_grad(xh * ga[e] + be[e]);
                o[e] += rstd * (dz * ga[e] - m1 - xh * m2);
            }
            Io<T>::st4(op + (size_t)p * a.C, o);
        };
        int p = blockIdx.x * ppb + tid / c4n;
        const int st = gridDim.x * ppb;
        for (; p + 3 * st < a.HW; p += 4 * st) {
            f32x4 v[4], d[4], o[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                v[u] = Io<T>::ld4(xp + (size_t)(p + u * st) * a.C);
                d[u] = Io<T>::ld4(dp + (size_t)(p + u * st) * a.C);
                o[u] = a.accumulate ? Io<T>::ld4(op + (size_t)(p + u * st) * a.C) : zero;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) one(v[u], d[u], o[u], p + u * st);
        }
        for (; p < a.HW; p += st)
            one(Io<T>::ld4(xp + (size_t)p * a.C), Io<T>::ld4(dp + (size_t)p * a.C), a.accumulate ? Io<T>::ld4(op + (size_t)p * a.C) : zero, p);
        return;
    }
    for (long long i = (long long)blockIdx.x * 256 + tid; i < total; i += (long long)gridDim.x * 256) {
        const long long p = i / c4n;
        const int c = 4 * (int)(i - p * c4n), g = c / Cg;
        float mean, rstd;
        gn_mean_rstd(a.stats, b * a.G + g, inv_cnt, mean, rstd);
        const float m1 = S1[g] * inv_cnt, m2 = S2[g] * inv_cnt;
        const size_t off = img + (size_t)p * a.C + c;
        const f32x4 v = Io<T>::ld4(x + off), d = Io<T>::ld4(dy + off);
        const f32x4 ga = *reinterpret_cast<const f32x4*>(a.gamma + c), be = *reinterpret_cast<const f32x4*>(a.beta + c);
        f32x4 o = f32x4{0.f, 0.f, 0.f, 0.f};
        if (a.accumulate) o = Io<T>::ld4(dx + off);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float xh = (v[e] - mean) * rstd;
            float dz = d[e];
            if (a.act) dz *= silu_grad(xh * ga[e] + be[e]);
            o[e] += rstd * (dz * ga[e] - m1 - xh * m2);
        }
        Io<T>::st4(dx + off, o);
    }
}

// ---- elementwise --------------------------------------------------------------------------------------------
// n is a multiple of 4 (channel counts are); four elements per thread and trip
template <typename T>
__global__ __launch_bounds__(256) void axpby_kernel(T* __restrict__ dst, const T* __restrict__ src, long long n4, int acc) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        f32x4 v = Io<T>::ld4(src + 4 * i);
        if (acc) {
            const f32x4 d = Io<T>::ld4(dst + 4 * i);
            v = f32x4{v[0] + d[0], v[1] + d[1], v[2] + d[2], v[3] + d[3]};
        }
        Io<T>::st4(dst + 4 * i, v);
    }
}
template <typename T>
__global__ __launch_bounds__(256) void add_kernel(T* __restrict__ out, const T* __restrict__ a, const T* __restrict__ b, long long n4) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const f32x4 x = Io<T>::ld4(a + 4 * i), y = Io<T>::ld4(b + 4 * i);
        Io<T>::st4(out + 4 * i, f32x4{x[0] + y[0], x[1] + y[1], x[2] + y[2], x[3] + y[3]});
    }
}
template <typename T>
__global__ __launch_bounds__(256) void copy_slice_kernel(T* __restrict__ dst, int ldd, int doff, const T* __restrict__ src, int lds,
                                                         int soff, int C, long long rows, int acc) {
    const int c4n = C >> 2;
    const long long total = rows * c4n;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long row = i / c4n;
        const int c = 4 * (int)(i - row * c4n);
        f32x4 v = Io<T>::ld4(src + row * lds + soff + c);
        T* d = dst + row * ldd + doff + c;
        if (acc) {
            const f32x4 o = Io<T>::ld4(d);
            v = f32x4{v[0] + o[0], v[1] + o[1], v[2] + o[2], v[3] + o[3]};
        }
        Io<T>::st4(d, v);
    }
}
template <typename T>
__global__ __launch_bounds__(256) void act_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, long long n, int act) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float v = Io<T>::ld(x + i);
        Io<T>::st(y + i, act == 0 ? v * sigmoid_f(v) : gelu_f(v));
    }
}
template <typename T>
__global__ __launch_bounds__(256) void act_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dy, T* __restrict__ dx,
                                                      long long n, int act, int acc) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float v = Io<T>::ld(x + i);
        const float g = Io<T>::ld(dy + i) * (act == 0 ? silu_grad(v) : gelu_grad(v));
        Io<T>::st(dx + i, acc ? Io<T>::ld(dx + i) + g : g);
    }
}
template <typename T>
__global__ __launch_bounds__(256) void class_bias_fwd_kernel(T* __restrict__ x, const float* __restrict__ table,
                                                             const long long* __restrict__ labels, int pad, int HW, int C) {
    const int b = blockIdx.y;
    const long long row = labels ? min(max(labels[b], 0ll), (long long)pad) : pad;
    const float* e = table + row * C;
    T* xb = x + (size_t)b * HW * C;
    const long long total = (long long)HW * C;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256)
        Io<T>::st(xb + i, Io<T>::ld(xb + i) + e[i % C]);
}
// grid (splits, B); thread = channel (c = tid % C) x pixel lane
template <typename T>
__global__ __launch_bounds__(256) void class_bias_bwd_kernel(const T* __restrict__ dy, const long long* __restrict__ labels, int pad,
                                                             float* __restrict__ dtable, int HW, int C) {
    __shared__ float sh[256];
    const int b = blockIdx.y, tid = threadIdx.x;
    const long long row = labels ? min(max(labels[b], 0ll), (long long)pad) : pad;
    if (row == pad) return;
    const int c = tid % C, pl = tid / C, npl = 256 / C;
    float acc = 0.f;
    if (pl < npl)
        for (int p = blockIdx.x * npl + pl; p < HW; p += gridDim.x * npl) acc += Io<T>::ld(dy + ((size_t)b * HW + p) * C + c);
    sh[tid] = acc;
    __syncthreads();
    if (pl == 0) {
        float t = 0.f;
        for (int l = 0; l < npl; ++l) t += sh[l * C + c];
        unsafeAtomicAdd(dtable + row * C + c, t);
    }
}

// parts/positional_encoding.py:11-18: frequencies exp(i * -(ln 1e4 / (half - 1))) in float32, [sin | cos]
__global__ void sinus_kernel(const void* t, int t_is_float, int B, int dim, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * dim) return;
    const int b = i / dim, j = i - b * dim, half = dim >> 1;
    const float tv = t_is_float ? reinterpret_cast<const float*>(t)[b] : (float)reinterpret_cast<const long long*>(t)[b];
    const float k = (float)(9.210340371976184 / (double)(half - 1));
    const float f = expf((float)(j < half ? j : j - half) * -k);
    const float arg = tv * f;
    out[i] = j < half ? sinf(arg) : cosf(arg);
}

template <typename T>
__global__ __launch_bounds__(256) void train_ingest_kernel(const float* __restrict__ x0, const float* __restrict__ eps,
                                                           const long long* __restrict__ t, const float* __restrict__ sa,
                                                           const float* __restrict__ sb, T* __restrict__ xt, float* __restrict__ tgt,
                                                           int B, int C, int HW, int nT) {
    const long long total = (long long)B * C * HW;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        // i indexes NHWC
        const int c = (int)(i % C);
        const long long pix = i / C;
        const int b = (int)(pix / HW), p = (int)(pix - (long long)b * HW);
        const size_t src = ((size_t)b * C + c) * HW + p;
        const float e = eps ? eps[src] : 0.f;
        float v = x0[src];
        if (sa) {
            const long long tb = min(max(t[b], 0ll), (long long)nT - 1);   // an out-of-range label must not read outside the tables
            v = sa[tb] * v + sb[tb] * e;
        }
        Io<T>::st(xt + i, v);
        if (tgt) tgt[i] = e;
    }
}

// out / dout in the activation type; the target stays float32 (it is the caller's noise, only reordered)
template <typename T>
__global__ __launch_bounds__(256) void loss_kernel(const T* __restrict__ out, const float* __restrict__ tgt, long long n, long long n_mean,
                                                   int kind, float* __restrict__ loss, T* __restrict__ dout) {
    __shared__ float sh[4];
    const float inv = 1.0f / (float)n_mean;
    float acc = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float d = Io<T>::ld(out + i) - tgt[i];
        float l, g;
        if (kind == 1) {
            l = d * d;
            g = 2.0f * d;
        } else if (kind == 0) {
            l = fabsf(d);
            g = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
        } else {
            const float ad = fabsf(d);
            l = ad < 1.f ? 0.5f * d * d : ad - 0.5f;
            g = ad < 1.f ? d : (d > 0.f ? 1.f : -1.f);
        }
        acc += l;
        Io<T>::st(dout + i, g * inv);
    }
    acc = block_sum(acc, sh);
    if (threadIdx.x == 0) unsafeAtomicAdd(loss, acc * inv);
}

template <typename T>
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const T* __restrict__ in, float* __restrict__ out, int B, int C, int HW) {
    const long long total = (long long)B * C * HW;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int p = (int)(i % HW);
        const long long bc = i / HW;
        const int c = (int)(bc % C), b = (int)(bc / C);
        out[i] = Io<T>::ld(in + ((size_t)b * HW + p) * C + c);
    }
}

// ---- linear attention core ------------------------------------------------------------------------------------
// One workgroup per (image, head). qkv row of pixel n: [q (heads*32) | k (heads*32) | v (heads*32)].
constexpr int kD = 32;
constexpr float kScale = 0.17677669529663687f;   // 32^-0.5

// softmax over the 32 channels of q for 64 pixels staged in LDS rows of 33 floats: thread (p = tid>>2, part = tid&3)
// owns channels 8*part..+7; afterwards Qs holds the UNSCALED softmax.
template <int LD = 33> __device__ __forceinline__ void softmax_q_rows(float* Qs, int tid) {
    const int p = tid >> 2, part = tid & 3;
    float* row = Qs + p * LD + 8 * part;
    float mx = row[0];
#pragma unroll
    for (int i = 1; i < 8; ++i) mx = fmaxf(mx, row[i]);
    mx = fmaxf(mx, __shfl_xor(mx, 1, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 2, 64));
    float e[8], sum = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        e[i] = expf(row[i] - mx);
        sum += e[i];
    }
    sum += __shfl_xor(sum, 1, 64);
    sum += __shfl_xor(sum, 2, 64);
    const float inv = 1.0f / sum;
#pragma unroll
    for (int i = 0; i < 8; ++i) row[i] = e[i] * inv;
}

// loads a [64 pixels][32] slice (channel offset coff of the row) into LDS rows of 33 floats; pixels >= n become `fill`
template <typename T, int LD = 33>
__device__ __forceinline__ void load_tile(float* S, const T* base, int ld, int coff, int p0, int n, int tid, float fill) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int idx = tid + 256 * i, p = idx >> 5, d = idx & 31;
        S[p * LD + d] = (p0 + p < n) ? Io<T>::ld(base + (size_t)(p0 + p) * ld + coff + d) : fill;
    }
}

// C[d][e] += sum over the 64 staged pixels of A[p][d] * B[p][e] on the matrix cores (exact fp32): wave w owns the
// 16x16 block (w >> 1, w & 1); row pitch 48 floats keeps the operand reads of a half-wave on 32 different banks.
// With SUM the waves of column block 0 also accumulate sum_p A[p][d] (every column of `sums` holds it).
constexpr int kLdT = 48;
template <bool SUM>
__device__ __forceinline__ void tile_outer_acc(const float* As, const float* Bs, int lane, int wave, f32x4& acc, f32x4& sums) {
    const int r = lane & 15, q = lane >> 4, di = wave >> 1, ej = wave & 1;
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        const float av = As[(4 * s + q) * kLdT + 16 * di + r];
        const float bv = Bs[(4 * s + q) * kLdT + 16 * ej + r];
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc, 0, 0, 0);
        if (SUM && ej == 0) sums = __builtin_amdgcn_mfma_f32_16x16x4f32(av, 1.0f, sums, 0, 0, 0);
    }
}

// Every kernel below runs on a grid (image * head, pixel split): a split owns the 64-pixel tiles
// [split * per, (split + 1) * per) of its image; reductions over pixels meet through float atomics in zeroed memory.
__device__ __forceinline__ void split_range(int n, int& p_begin, int& p_end) {
    const int tiles = (n + 63) >> 6, per = (tiles + gridDim.y - 1) / gridDim.y;
    p_begin = min(n, (int)blockIdx.y * per * 64);
    p_end = min(n, ((int)blockIdx.y + 1) * per * 64);
}

// partial max over the split's pixels of every k channel: pmax[bh][split][32]
template <typename T>
__global__ __launch_bounds__(256) void linattn_kmax_kernel(const T* __restrict__ qkv, float* __restrict__ pmax, int n, int heads) {
    __shared__ float red[8 * 32];
    const int bh = blockIdx.x, b = bh / heads, h = bh - b * heads, tid = threadIdx.x;
    const int hid = heads * kD, ld = 3 * hid;
    const T* base = qkv + (size_t)b * n * ld;
    int pb, pe;
    split_range(n, pb, pe);
    const int d = tid & 31, pl = tid >> 5;
    float mx = -INFINITY;
    {   // eight pixels' loads in flight together (unconditional, from clamped addresses); max is order-independent
        const T* col = base + hid + h * kD + d;
        int p = pb + pl;
        for (; p + 56 < pe; p += 64) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = Io<T>::ld(col + (size_t)(p + 8 * u) * ld);
#pragma unroll
            for (int u = 0; u < 8; ++u) mx = fmaxf(mx, v[u]);
        }
        for (; p < pe; p += 8) mx = fmaxf(mx, Io<T>::ld(col + (size_t)p * ld));
    }
    red[pl * 32 + d] = mx;
    __syncthreads();
    if (tid < 32) {
        float m = red[tid];
        for (int i = 1; i < 8; ++i) m = fmaxf(m, red[i * 32 + tid]);
        pmax[((size_t)bh * gridDim.y + blockIdx.y) * 32 + tid] = m;
    }
}

// ctx_un[d][e] += sum_n exp(k[n][d] - max[d]) v[n][e];  kstat[bh] = {max[32], sum of exp[32]}
template <typename T>
__global__ __launch_bounds__(256) void linattn_ctx_kernel(const T* __restrict__ qkv, const float* __restrict__ pmax,
                                                          float* __restrict__ ctx_un, float* __restrict__ kstat, int n, int heads) {
    __shared__ float Ks[64 * kLdT], Vs[64 * kLdT], kmx[32];
    const int bh = blockIdx.x, b = bh / heads, h = bh - b * heads, tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int hid = heads * kD, ld = 3 * hid;
    const T* base = qkv + (size_t)b * n * ld;
    int pb, pe;
    split_range(n, pb, pe);
    {   // the maximum over all splits (up to 32 of them): eight threads per channel share the splits instead of one
        // thread walking them one memory round trip at a time; Ks is free until the first barrier of the loop below
        float m = -INFINITY;
        for (int i = tid >> 5; i < (int)gridDim.y; i += 8) m = fmaxf(m, pmax[((size_t)bh * gridDim.y + i) * 32 + (tid & 31)]);
        Ks[tid] = m;
        __syncthreads();
        if (tid < 32) {
            m = Ks[tid];
            for (int i = 1; i < 8; ++i) m = fmaxf(m, Ks[i * 32 + tid]);
            kmx[tid] = m;
            if (blockIdx.y == 0) kstat[(size_t)bh * 64 + tid] = m;
        }
    }
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f}, sums = acc;
    // the next tile's k and v are requested (unconditionally, from clamped rows) before the current tile's outer
    // products, so that a tile costs its arithmetic instead of arithmetic plus a memory round trip
    float kraw[8], vraw[8];
    auto fetch = [&](int p0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int idx = tid + 256 * i, p = idx >> 5, dd = idx & 31;
            const int pr = p0 + p < pe ? p0 + p : pe - 1;
            const T* row = base + (size_t)pr * ld;
            kraw[i] = Io<T>::ld(row + hid + h * kD + dd);
            vraw[i] = Io<T>::ld(row + 2 * hid + h * kD + dd);
        }
    };
    if (pb < pe) fetch(pb);
    for (int p0 = pb; p0 < pe; p0 += 64) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int idx = tid + 256 * i, p = idx >> 5, dd = idx & 31;
            const bool ok = p0 + p < pe;
            Ks[p * kLdT + dd] = ok ? expf(kraw[i] - kmx[dd]) : 0.f;
            Vs[p * kLdT + dd] = ok ? vraw[i] : 0.f;
        }
        if (p0 + 64 < pe) fetch(p0 + 64);
        __syncthreads();
        tile_outer_acc<true>(Ks, Vs, lane, wave, acc, sums);
    }
    if (pb < pe) {   // D[4q + i][r]: row d = 16*(wave>>1) + 4q + i, column e = 16*(wave&1) + r
        const int r = lane & 15, q = lane >> 4, di = wave >> 1, ej = wave & 1;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int d = 16 * di + 4 * q + i;
            unsafeAtomicAdd(ctx_un + (size_t)bh * 1024 + d * 32 + 16 * ej + r, acc[i]);
            if (ej == 0 && r == 0) unsafeAtomicAdd(kstat + (size_t)bh * 64 + 32 + d, sums[i]);
        }
    }
}

// ---- per-pixel 32x32 products on the matrix cores ----------------------------------------------------------------
// A wave owns 16 pixels; lane (r = lane & 15, q = lane >> 4) holds, for pixel r, the eight channels
// 16c + 4q + s (c < 2, s < 4) of a 32-channel head, in slot order t = 4c + s: two float4 of its row. For a 32x32 matrix
// Wm, y = Wm x of every pixel comes out of the MFMA in the SAME slot layout when the matrix is the A operand and the
// lanes' own values the B operand: D[4q + i][r] is channel 16j + 4q + i of pixel r. Everything between the products
// (softmax, its backward) is therefore lane-local plus reductions over the four q lanes; no LDS for pixel data and no
// barrier in the pixel loop. Wm sits in LDS with row pitch 33 and its columns permuted so that the 32 lanes of a
// half-wave read 32 different banks.
__device__ __forceinline__ int perm_col(int ch) {
    const int c = ch >> 4, q = (ch >> 2) & 3, s = ch & 3;
    return ((q & 1) << 4) + ((q >> 1) << 3) + 4 * c + s;
}
__device__ __forceinline__ bf16x8 pack8_bf16(const float* v) {
    bf16x8 o;
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = (__bf16)v[i];
    return o;
}
template <bool BF> __device__ __forceinline__ void mat32_apply(const float* Wm, const float* x, float* y, int r, int q) {
    const int cb = ((q & 1) << 4) + ((q >> 1) << 3);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
        const float* row = Wm + (16 * j + r) * 33 + cb;
        if constexpr (BF) {
            float a8[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) a8[t] = row[t];
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pack8_bf16(a8), pack8_bf16(x), acc, 0, 0, 0);
        } else {
#pragma unroll
            for (int t = 0; t < 8; ++t) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(row[t], x[t], acc, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) y[4 * j + i] = acc[i];
    }
}
__device__ __forceinline__ float qsum(float v) {   // over the four lanes that share a pixel
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}
__device__ __forceinline__ float qmax(float v) {
    v = fmaxf(v, __shfl_xor(v, 16, 64));
    v = fmaxf(v, __shfl_xor(v, 32, 64));
    return v;
}
template <typename T> __device__ __forceinline__ void load8(const T* row, int q, bool ok, float* x) {
    f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f}, b = a;
    if (ok) {
        a = Io<T>::ld4(row + 4 * q);
        b = Io<T>::ld4(row + 16 + 4 * q);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        x[i] = a[i];
        x[4 + i] = b[i];
    }
}
template <typename T> __device__ __forceinline__ void store8(T* row, int q, const float* x) {
    Io<T>::st4(row + 4 * q, f32x4{x[0], x[1], x[2], x[3]});
    Io<T>::st4(row + 16 + 4 * q, f32x4{x[4], x[5], x[6], x[7]});
}
// unscaled softmax over the 32 channels of the lane's pixel
__device__ __forceinline__ void softmax8(const float* x, float* sm) {
    float mx = x[0];
#pragma unroll
    for (int i = 1; i < 8; ++i) mx = fmaxf(mx, x[i]);
    mx = qmax(mx);
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        sm[i] = expf(x[i] - mx);
        sum += sm[i];
    }
    const float inv = 1.0f / qsum(sum);
#pragma unroll
    for (int i = 0; i < 8; ++i) sm[i] *= inv;
}

// out[n][e] = scale * sum_d ctx[d][e] * softmax_d(q[n])[d],  ctx = ctx_un / sum of exp
template <typename T, bool BF>
__global__ __launch_bounds__(256) void linattn_out_kernel(const T* __restrict__ qkv, const float* __restrict__ ctx_un,
                                                          const float* __restrict__ kstat, T* __restrict__ out, int n, int heads) {
    __shared__ float CxT[32 * 33];
    const int bh = blockIdx.x, b = bh / heads, h = bh - b * heads, tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6, r = lane & 15, q = lane >> 4;
    const int hid = heads * kD, ld = 3 * hid;
    const T* base = qkv + (size_t)b * n * ld;
    int pb, pe;
    split_range(n, pb, pe);
    for (int i = tid; i < 1024; i += 256) {
        const int d = i >> 5, e = i & 31;
        CxT[e * 33 + perm_col(d)] = ctx_un[(size_t)bh * 1024 + i] / kstat[(size_t)bh * 64 + 32 + d];
    }
    __syncthreads();
    float nq[8];   // the next 16 pixels' q, requested before the current ones are worked on
    if (pb + 16 * wave < pe) load8(base + (size_t)(pb + 16 * wave + r < pe ? pb + 16 * wave + r : pb) * ld + h * kD, q, pb + 16 * wave + r < pe, nq);
    for (int p0 = pb + 16 * wave; p0 < pe; p0 += 64) {
        const int pix = p0 + r;
        const bool ok = pix < pe;
        float qv[8], qs[8], o[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) qv[i] = nq[i];
        if (p0 + 64 < pe) {
            const int np = p0 + 64 + r;
            load8(base + (size_t)(np < pe ? np : pb) * ld + h * kD, q, np < pe, nq);
        }
        softmax8(qv, qs);
        mat32_apply<BF>(CxT, qs, o, r, q);
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] *= kScale;
        if (ok) store8(out + ((size_t)b * n + pix) * hid + h * kD, q, o);
    }
}

// dctx[d][e] += scale * sum_n softmax(q[n])[d] * dout[n][e]
template <typename T>
__global__ __launch_bounds__(256) void linattn_dctx_kernel(const T* __restrict__ qkv, const T* __restrict__ dout,
                                                           float* __restrict__ dctx, int n, int heads) {
    __shared__ float As[64 * kLdT], Bs[64 * kLdT];
    const int bh = blockIdx.x, b = bh / heads, h = bh - b * heads, tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int hid = heads * kD, ld = 3 * hid;
    const T* base = qkv + (size_t)b * n * ld;
    const T* dob = dout + (size_t)b * n * hid;
    int pb, pe;
    split_range(n, pb, pe);
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f}, unused = acc;
    // the next tile's q and dout rows are requested (from clamped rows) before the current tile is worked on
    float qa[8], da[8];
    auto fetch = [&](int p0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int idx = tid + 256 * i, p = idx >> 5, d = idx & 31;
            const int pr = p0 + p < pe ? p0 + p : pe - 1;
            qa[i] = Io<T>::ld(base + (size_t)pr * ld + h * kD + d);
            da[i] = Io<T>::ld(dob + (size_t)pr * hid + h * kD + d);
        }
    };
    if (pb < pe) fetch(pb);
    for (int p0 = pb; p0 < pe; p0 += 64) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int idx = tid + 256 * i, p = idx >> 5, d = idx & 31;
            const bool ok = p0 + p < pe;   // rows beyond the range are zero: they add nothing
            As[p * kLdT + d] = ok ? qa[i] : 0.f;
            Bs[p * kLdT + d] = ok ? da[i] : 0.f;
        }
        if (p0 + 64 < pe) fetch(p0 + 64);
        __syncthreads();
        softmax_q_rows<kLdT>(As, tid);
        __syncthreads();
        tile_outer_acc<false>(As, Bs, lane, wave, acc, unused);
    }
    if (pb < pe) {
        const int r = lane & 15, q = lane >> 4, di = wave >> 1, ej = wave & 1;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            unsafeAtomicAdd(dctx + (size_t)bh * 1024 + (16 * di + 4 * q + i) * 32 + 16 * ej + r, acc[i] * kScale);
    }
}

// per pixel dq, dk, dv on the matrix cores (layout: see mat32_apply)
template <typename T, bool BF>
__global__ __launch_bounds__(256) void linattn_bwd_pix_kernel(const T* __restrict__ qkv, const T* __restrict__ dout,
                                                              const float* __restrict__ ctx_un, const float* __restrict__ kstat,
                                                              const float* __restrict__ dctx, T* __restrict__ dqkv, int n, int heads) {
    __shared__ float Cx[32 * 33], Dc[32 * 33], DcT[32 * 33], sv[32];
    const int bh = blockIdx.x, b = bh / heads, h = bh - b * heads, tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6, r = lane & 15, q = lane >> 4;
    const int hid = heads * kD, ld = 3 * hid;
    const T* base = qkv + (size_t)b * n * ld;
    const T* dob = dout + (size_t)b * n * hid;
    T* dqb = dqkv + (size_t)b * n * ld;
    int pb, pe;
    split_range(n, pb, pe);
    for (int i = tid; i < 1024; i += 256) {
        const int d = i >> 5, e = i & 31;
        const float g = dctx[(size_t)bh * 1024 + i];
        Cx[d * 33 + perm_col(e)] = ctx_un[(size_t)bh * 1024 + i] / kstat[(size_t)bh * 64 + 32 + d];
        Dc[d * 33 + perm_col(e)] = g;
        DcT[e * 33 + perm_col(d)] = g;
    }
    __syncthreads();
    if (tid < 32) {
        float t = 0.f;
        for (int e = 0; e < 32; ++e) t += Dc[tid * 33 + e] * Cx[tid * 33 + e];
        sv[tid] = t;
    }
    __syncthreads();
    float kmx8[8], kin8[8], sv8[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const int ch = 16 * (t >> 2) + 4 * q + (t & 3);
        kmx8[t] = kstat[(size_t)bh * 64 + ch];
        kin8[t] = 1.0f / kstat[(size_t)bh * 64 + 32 + ch];
        sv8[t] = sv[ch];
    }
    // the next 16 pixels' operands are requested before the current ones are worked on (a memory round trip per
    // iteration otherwise)
    float nq[8], nk[8], nv[8], nd[8];
    auto fetch = [&](int p0) {
        const int pix = p0 + r;
        const bool ok = pix < pe;
        const T* row = base + (size_t)(ok ? pix : pb) * ld + h * kD;
        load8(row, q, ok, nq);
        load8(row + hid, q, ok, nk);
        load8(row + 2 * hid, q, ok, nv);
        load8(dob + (size_t)(ok ? pix : pb) * hid + h * kD, q, ok, nd);
    };
    if (pb + 16 * wave < pe) fetch(pb + 16 * wave);
    for (int p0 = pb + 16 * wave; p0 < pe; p0 += 64) {
        const int pix = p0 + r;
        const bool ok = pix < pe;
        float qv[8], kv[8], vv[8], dov[8], qs[8], y[8], res[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            qv[i] = nq[i];
            kv[i] = nk[i];
            vv[i] = nv[i];
            dov[i] = nd[i];
        }
        if (p0 + 64 < pe) fetch(p0 + 64);
        softmax8(qv, qs);
        // dqhat[d] = sum_e ctx[d][e] dout[e]; dq = scale * s[d] * (dqhat[d] - sum_d' s[d'] dqhat[d'])
        mat32_apply<BF>(Cx, dov, y, r, q);
        float dot = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) dot += qs[i] * y[i];
        dot = qsum(dot);
#pragma unroll
        for (int i = 0; i < 8; ++i) res[i] = kScale * qs[i] * (y[i] - dot);
        T* orow = dqb + (size_t)pix * ld + h * kD;
        if (ok) store8(orow, q, res);
        // dkhat[d] = sum_e dctx[d][e] v[e]; dk = khat[d] * (dkhat[d] - sv[d])
        mat32_apply<BF>(Dc, vv, y, r, q);
        float kh[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            kh[i] = ok ? expf(kv[i] - kmx8[i]) * kin8[i] : 0.f;
            res[i] = kh[i] * (y[i] - sv8[i]);
        }
        if (ok) store8(orow + hid, q, res);
        // dv[e] = sum_d khat[d] dctx[d][e]
        mat32_apply<BF>(DcT, kh, y, r, q);
        if (ok) store8(orow + 2 * hid, q, y);
    }
}

// ---- softmax attention core (bottleneck) ------------------------------------------------------------------------
// One workgroup per (image, head); thread = query row (rows beyond 256 in further rounds); keys in LDS tiles of 64.
template <typename T>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const T* __restrict__ qkv, T* __restrict__ out, float* __restrict__ lse,
                                                       int n, int heads) {
    __shared__ float Ks[64 * 33], Vs[64 * 33];
    const int bh = blockIdx.x, b = bh / heads, h = bh - b * heads, tid = threadIdx.x;
    const int hid = heads * kD, ld = 3 * hid;
    const T* base = qkv + (size_t)b * n * ld;
    for (int i0 = 0; i0 < n; i0 += 256) {
        const int i = i0 + tid;
        const bool ok = i < n;
        float qv[32], acc[32], m = -INFINITY, l = 0.f;
#pragma unroll
        for (int d = 0; d < 32; ++d) {
            qv[d] = ok ? Io<T>::ld(base + (size_t)i * ld + h * kD + d) * kScale : 0.f;
            acc[d] = 0.f;
        }
        for (int j0 = 0; j0 < n; j0 += 64) {
            __syncthreads();
            load_tile(Ks, base, ld, hid + h * kD, j0, n, tid, 0.f);
            load_tile(Vs, base, ld, 2 * hid + h * kD, j0, n, tid, 0.f);
            __syncthreads();
            const int jn = min(64, n - j0);
            for (int j = 0; j < jn; ++j) {
                float sc = 0.f;
#pragma unroll
                for (int d = 0; d < 32; ++d) sc += qv[d] * Ks[j * 33 + d];
                const float mn = fmaxf(m, sc);
                const float corr = expf(m - mn), pj = expf(sc - mn);
                l = l * corr + pj;
#pragma unroll
                for (int d = 0; d < 32; ++d) acc[d] = acc[d] * corr + pj * Vs[j * 33 + d];
                m = mn;
            }
        }
        if (ok) {
            const float inv = 1.0f / l;
            T* orow = out + ((size_t)b * n + i) * hid + h * kD;
#pragma unroll
            for (int d = 0; d < 32; ++d) Io<T>::st(orow + d, acc[d] * inv);
            lse[(size_t)bh * n + i] = m + logf(l);
        }
    }
}

// Backward of the softmax attention in two launches on a grid (image * head, blocks of 64 rows): thread (row = tid & 63,
// part = tid >> 6) owns one query (first launch: dq and the per-query sum_e dout*out) or one key (second launch:
// dk, dv) and a quarter (16 of every staged 64) of the other side; the four partial rows are added through LDS.
template <typename T>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(const T* __restrict__ qkv, const T* __restrict__ out, const T* __restrict__ dout,
                                                          const float* __restrict__ lse, T* __restrict__ dqkv, float* __restrict__ dsum,
                                                          int n, int heads) {
    __shared__ float As[64 * 33], Bs[64 * 33], red[4 * 64 * 33];
    const int bh = blockIdx.x, b = bh / heads, h = bh - b * heads, tid = threadIdx.x;
    const int row = tid & 63, part = tid >> 6;
    const int hid = heads * kD, ld = 3 * hid;
    const T* base = qkv + (size_t)b * n * ld;
    const T* ob = out + (size_t)b * n * hid;
    const T* dob = dout + (size_t)b * n * hid;
    const int i = blockIdx.y * 64 + row;
    const bool ok = i < n;
    float qv[32], dov[32], dq[32], Di = 0.f;
    const float li = ok ? lse[(size_t)bh * n + i] : 0.f;
#pragma unroll
    for (int d = 0; d < 32; ++d) {
        qv[d] = ok ? Io<T>::ld(base + (size_t)i * ld + h * kD + d) * kScale : 0.f;
        dov[d] = ok ? Io<T>::ld(dob + (size_t)i * hid + h * kD + d) : 0.f;
        Di += dov[d] * (ok ? Io<T>::ld(ob + (size_t)i * hid + h * kD + d) : 0.f);
        dq[d] = 0.f;
    }
    if (ok && part == 0) dsum[(size_t)bh * n + i] = Di;
    for (int j0 = 0; j0 < n; j0 += 64) {
        __syncthreads();
        load_tile(As, base, ld, hid + h * kD, j0, n, tid, 0.f);
        load_tile(Bs, base, ld, 2 * hid + h * kD, j0, n, tid, 0.f);
        __syncthreads();
        const int jn = min(64, n - j0);
        for (int j = 16 * part; j < min(jn, 16 * part + 16); ++j) {
            float sc = 0.f, dp = 0.f;
#pragma unroll
            for (int d = 0; d < 32; ++d) {
                sc += qv[d] * As[j * 33 + d];
                dp += dov[d] * Bs[j * 33 + d];
            }
            const float ds = expf(sc - li) * (dp - Di);
#pragma unroll
            for (int d = 0; d < 32; ++d) dq[d] += ds * As[j * 33 + d];
        }
    }
#pragma unroll
    for (int d = 0; d < 32; ++d) red[(part * 64 + row) * 33 + d] = dq[d];
    __syncthreads();
    if (ok && part == 0) {
        T* orow = dqkv + ((size_t)b * n + i) * ld + h * kD;
#pragma unroll
        for (int d = 0; d < 32; ++d) {
            const float v = (red[row * 33 + d] + red[(64 + row) * 33 + d]) + (red[(128 + row) * 33 + d] + red[(192 + row) * 33 + d]);
            Io<T>::st(orow + d, v * kScale);
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(const T* __restrict__ qkv, const T* __restrict__ dout, const float* __restrict__ lse,
                                                           const float* __restrict__ dsum, T* __restrict__ dqkv, int n, int heads) {
    __shared__ float As[64 * 33], Bs[64 * 33], Ls[64], Ds[64], red[4 * 64 * 33];
    const int bh = blockIdx.x, b = bh / heads, h = bh - b * heads, tid = threadIdx.x;
    const int row = tid & 63, part = tid >> 6;
    const int hid = heads * kD, ld = 3 * hid;
    const T* base = qkv + (size_t)b * n * ld;
    const T* dob = dout + (size_t)b * n * hid;
    const int j = blockIdx.y * 64 + row;
    const bool ok = j < n;
    float kv[32], vv[32], dk[32], dv[32];
#pragma unroll
    for (int d = 0; d < 32; ++d) {
        kv[d] = ok ? Io<T>::ld(base + (size_t)j * ld + hid + h * kD + d) : 0.f;
        vv[d] = ok ? Io<T>::ld(base + (size_t)j * ld + 2 * hid + h * kD + d) : 0.f;
        dk[d] = dv[d] = 0.f;
    }
    for (int i0 = 0; i0 < n; i0 += 64) {
        __syncthreads();
        load_tile(As, base, ld, h * kD, i0, n, tid, 0.f);     // q (unscaled)
        load_tile(Bs, dob, hid, h * kD, i0, n, tid, 0.f);    // dout
        if (tid < 64) {
            const bool in = i0 + tid < n;
            Ls[tid] = in ? lse[(size_t)bh * n + i0 + tid] : 0.f;
            Ds[tid] = in ? dsum[(size_t)bh * n + i0 + tid] : 0.f;
        }
        __syncthreads();
        const int in_ = min(64, n - i0);
        for (int i = 16 * part; i < min(in_, 16 * part + 16); ++i) {
            float sc = 0.f, dp = 0.f;
#pragma unroll
            for (int d = 0; d < 32; ++d) {
                sc += As[i * 33 + d] * kv[d];
                dp += Bs[i * 33 + d] * vv[d];
            }
            const float pij = expf(sc * kScale - Ls[i]);
            const float ds = pij * (dp - Ds[i]) * kScale;
#pragma unroll
            for (int d = 0; d < 32; ++d) {
                dv[d] += pij * Bs[i * 33 + d];
                dk[d] += ds * As[i * 33 + d];
            }
        }
    }
    T* orow = dqkv + ((size_t)b * n + j) * ld + hid + h * kD;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        __syncthreads();
#pragma unroll
        for (int d = 0; d < 32; ++d) red[(part * 64 + row) * 33 + d] = pass == 0 ? dk[d] : dv[d];
        __syncthreads();
        if (ok && part == 0) {
#pragma unroll
            for (int d = 0; d < 32; ++d) {
                const float v = (red[row * 33 + d] + red[(64 + row) * 33 + d]) + (red[(128 + row) * 33 + d] + red[(192 + row) * 33 + d]);
                Io<T>::st(orow + pass * hid + d, v);
            }
        }
    }
}

// ---- softmax attention core on the matrix cores (bf16 operands, fp32 accumulation), 64 <= n <= 256, n % 64 == 0 ----
// Same orientation in all three kernels: a 16x16 product tile has one side of the score matrix on the lane (lane & 15)
// and four entries of the other side in the registers of a lane group, so (a) the per-query / per-key row quantities
// (log-sum-exp, sum_e dout*out) are one value per lane or per register, and (b) two stacked tiles, converted to bf16,
// ARE the 32-deep operand of the product that contracts over that other side -- no lane movement. The operand that
// pairs with them is the transpose of a [n][32] tensor, staged once per workgroup in LDS.
// grid (heads, B, n / 64); a wave owns 16 queries (forward, dq) or 16 keys (dk/dv).
constexpr int kAtMaxN = 256;
constexpr int kAtPitch = kAtMaxN + 8;   // bf16 elements per transposed row (+16 bytes: rows start on different banks)

// eight consecutive channels of a row as a bf16 operand fragment
__device__ __forceinline__ bf16x8 at_frag(const float* p) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
    return bf16x8{(__bf16)a[0], (__bf16)a[1], (__bf16)a[2], (__bf16)a[3], (__bf16)b[0], (__bf16)b[1], (__bf16)b[2], (__bf16)b[3]};
}
__device__ __forceinline__ bf16x8 at_frag(const __bf16* p) { return *reinterpret_cast<const bf16x8*>(p); }

// transposes columns [coff, coff + 32) of rows [0, n) of `src` (row stride ld) into sT[32][kAtPitch] (bf16)
template <typename T>
__device__ __forceinline__ void at_stage_transposed(__bf16* sT, const T* src, int ld, int coff, int n, int tid) {
    for (int u = tid; u < n * 4; u += 256) {
        const int row = u >> 2, part = u & 3;
        const bf16x8 v = at_frag(src + (size_t)row * ld + coff + part * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) sT[(part * 8 + j) * kAtPitch + row] = v[j];
    }
}

// transposed fragment for contraction step u (32 entries of the long side): positions {32u+4g..+3, 32u+16+4g..+3}
__device__ __forceinline__ bf16x8 at_tfrag(const __bf16* sT, int row, int u, int lq) {
    const __bf16* p = sT + row * kAtPitch + 32 * u + 4 * lq;
    const bf16x4 lo = *reinterpret_cast<const bf16x4*>(p), hi = *reinterpret_cast<const bf16x4*>(p + 16);
    return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

__device__ __forceinline__ bf16x8 at_stack(const f32x4& t0, const f32x4& t1) {
    return bf16x8{(__bf16)t0[0], (__bf16)t0[1], (__bf16)t0[2], (__bf16)t0[3], (__bf16)t1[0], (__bf16)t1[1], (__bf16)t1[2], (__bf16)t1[3]};
}

#define AT_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)

template <typename T>
__global__ __launch_bounds__(256) void attn_fwd_mfma_kernel(const T* __restrict__ qkv, T* __restrict__ out, float* __restrict__ lse,
                                                            int n, int heads) {
    __shared__ __attribute__((aligned(16))) __bf16 sVt[32 * kAtPitch];
    const int h = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lq = lane >> 4;
    const int hid = heads * kD, ld = 3 * hid, bh = b * heads + h;
    const T* base = qkv + (size_t)b * n * ld;
    const int qi = (blockIdx.z * 4 + wave) * 16 + l15;
    const bf16x8 qf = at_frag(base + (size_t)qi * ld + h * kD + lq * 8);
    at_stage_transposed(sVt, base, ld, 2 * hid + h * kD, n, tid);
    const int nt = n / 16;
    f32x4 sreg[kAtMaxN / 16];
    float m = -INFINITY;
#pragma unroll
    for (int t = 0; t < kAtMaxN / 16; ++t) {
        if (t < nt) {
            const bf16x8 kf = at_frag(base + (size_t)(t * 16 + l15) * ld + hid + h * kD + lq * 8);
            f32x4 acc = AT_MFMA(kf, qf, (f32x4{0.f, 0.f, 0.f, 0.f}));
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                acc[r] *= kScale;
                m = fmaxf(m, acc[r]);
            }
            sreg[t] = acc;
        }
    }
    m = fmaxf(m, __shfl_xor(m, 16, 64));
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float l = 0.f;
#pragma unroll
    for (int t = 0; t < kAtMaxN / 16; ++t) {
        if (t < nt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e = (float)(__bf16)__expf(sreg[t][r] - m);   // the denominator sums what the product multiplies
                sreg[t][r] = e;
                l += e;
            }
        }
    }
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    __syncthreads();
    f32x4 o[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int u = 0; u < kAtMaxN / 32; ++u) {
        if (2 * u < nt) {
            const bf16x8 pf = at_stack(sreg[2 * u], sreg[2 * u + 1]);
#pragma unroll
            for (int et = 0; et < 2; ++et) o[et] = AT_MFMA(at_tfrag(sVt, et * 16 + l15, u, lq), pf, o[et]);
        }
    }
    const float inv = 1.0f / l;
    T* orow = out + ((size_t)b * n + qi) * hid + h * kD;
#pragma unroll
    for (int et = 0; et < 2; ++et) {
        f32x4 v = o[et];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] *= inv;
        Io<T>::st4(orow + et * 16 + lq * 4, v);
    }
    if (lq == 0) lse[(size_t)bh * n + qi] = m + logf(l);
}

// dq (and the per-query sum_e dout*out, kept for the second launch): a wave owns 16 queries and sweeps the keys
template <typename T>
__global__ __launch_bounds__(256) void attn_bwd_dq_mfma_kernel(const T* __restrict__ qkv, const T* __restrict__ out,
                                                               const T* __restrict__ dout, const float* __restrict__ lse,
                                                               T* __restrict__ dqkv, float* __restrict__ dsum, int n, int heads) {
    __shared__ __attribute__((aligned(16))) __bf16 sKt[32 * kAtPitch];
    const int h = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lq = lane >> 4;
    const int hid = heads * kD, ld = 3 * hid, bh = b * heads + h;
    const T* base = qkv + (size_t)b * n * ld;
    const int qi = (blockIdx.z * 4 + wave) * 16 + l15;
    const bf16x8 qf = at_frag(base + (size_t)qi * ld + h * kD + lq * 8);
    const T* dorow = dout + ((size_t)b * n + qi) * hid + h * kD + lq * 8;
    const T* orow = out + ((size_t)b * n + qi) * hid + h * kD + lq * 8;
    const bf16x8 dof = at_frag(dorow);
    float Di = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) Di += Io<T>::ld(dorow + j) * Io<T>::ld(orow + j);
    Di += __shfl_xor(Di, 16, 64);
    Di += __shfl_xor(Di, 32, 64);
    if (lq == 0) dsum[(size_t)bh * n + qi] = Di;
    const float li = lse[(size_t)bh * n + qi];
    at_stage_transposed(sKt, base, ld, hid + h * kD, n, tid);
    __syncthreads();
    f32x4 dq[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    for (int u = 0; u < n / 32; ++u) {
        f32x4 ds[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const T* krow = base + (size_t)((2 * u + i) * 16 + l15) * ld + h * kD + lq * 8;
            const f32x4 sc = AT_MFMA(at_frag(krow + hid), qf, (f32x4{0.f, 0.f, 0.f, 0.f}));
            const f32x4 dp = AT_MFMA(at_frag(krow + 2 * hid), dof, (f32x4{0.f, 0.f, 0.f, 0.f}));
#pragma unroll
            for (int r = 0; r < 4; ++r) ds[i][r] = __expf(sc[r] * kScale - li) * (dp[r] - Di);
        }
        const bf16x8 dsf = at_stack(ds[0], ds[1]);
#pragma unroll
        for (int et = 0; et < 2; ++et) dq[et] = AT_MFMA(at_tfrag(sKt, et * 16 + l15, u, lq), dsf, dq[et]);
    }
    T* drow = dqkv + ((size_t)b * n + qi) * ld + h * kD;
#pragma unroll
    for (int et = 0; et < 2; ++et) {
        f32x4 v = dq[et];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] *= kScale;
        Io<T>::st4(drow + et * 16 + lq * 4, v);
    }
}

// dk and dv: a wave owns 16 keys and sweeps the queries
template <typename T>
__global__ __launch_bounds__(256) void attn_bwd_dkv_mfma_kernel(const T* __restrict__ qkv, const T* __restrict__ dout,
                                                                const float* __restrict__ lse, const float* __restrict__ dsum,
                                                                T* __restrict__ dqkv, int n, int heads) {
    __shared__ __attribute__((aligned(16))) __bf16 sQt[32 * kAtPitch], sDt[32 * kAtPitch];
    const int h = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lq = lane >> 4;
    const int hid = heads * kD, ld = 3 * hid, bh = b * heads + h;
    const T* base = qkv + (size_t)b * n * ld;
    const T* dob = dout + (size_t)b * n * hid;
    const int kj = (blockIdx.z * 4 + wave) * 16 + l15;
    const bf16x8 kf = at_frag(base + (size_t)kj * ld + hid + h * kD + lq * 8);
    const bf16x8 vf = at_frag(base + (size_t)kj * ld + 2 * hid + h * kD + lq * 8);
    at_stage_transposed(sQt, base, ld, h * kD, n, tid);
    at_stage_transposed(sDt, dob, hid, h * kD, n, tid);
    __syncthreads();
    f32x4 dk[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    f32x4 dv[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    for (int u = 0; u < n / 32; ++u) {
        f32x4 pp[2], ds[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int q0 = (2 * u + i) * 16;
            // rows = queries (registers: query q0 + 4 lq + r), column = this lane's key
            const f32x4 sc = AT_MFMA(at_frag(base + (size_t)(q0 + l15) * ld + h * kD + lq * 8), kf, (f32x4{0.f, 0.f, 0.f, 0.f}));
            const f32x4 dp = AT_MFMA(at_frag(dob + (size_t)(q0 + l15) * hid + h * kD + lq * 8), vf, (f32x4{0.f, 0.f, 0.f, 0.f}));
            const f32x4 lr = *reinterpret_cast<const f32x4*>(lse + (size_t)bh * n + q0 + 4 * lq);
            const f32x4 dr = *reinterpret_cast<const f32x4*>(dsum + (size_t)bh * n + q0 + 4 * lq);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                pp[i][r] = __expf(sc[r] * kScale - lr[r]);
                ds[i][r] = pp[i][r] * (dp[r] - dr[r]) * kScale;
            }
        }
        const bf16x8 pf = at_stack(pp[0], pp[1]), dsf = at_stack(ds[0], ds[1]);
#pragma unroll
        for (int et = 0; et < 2; ++et) {
            dv[et] = AT_MFMA(at_tfrag(sDt, et * 16 + l15, u, lq), pf, dv[et]);
            dk[et] = AT_MFMA(at_tfrag(sQt, et * 16 + l15, u, lq), dsf, dk[et]);
        }
    }
    T* drow = dqkv + ((size_t)b * n + kj) * ld + h * kD;
#pragma unroll
    for (int et = 0; et < 2; ++et) {
        Io<T>::st4(drow + hid + et * 16 + lq * 4, dk[et]);
        Io<T>::st4(drow + 2 * hid + et * 16 + lq * 4, dv[et]);
    }
}
#undef AT_MFMA

static bool attn_train_on_matrix_cores(int n, int bf16) { return bf16 && n >= 64 && n <= kAtMaxN && n % 64 == 0; }

// ---- parameter layout changes -----------------------------------------------------------------------------------
__device__ __forceinline__ void repack_index(const RepackDesc& d, long long idx, int& co, int& ci, int& tap) {
    tap = (int)(idx % d.taps);
    const long long r = idx / d.taps;
    if (d.transposed) {
        co = (int)(r % d.n_out);
        ci = (int)(r / d.n_out);
    } else {
        ci = (int)(r % d.n_in);
        co = (int)(r / d.n_in);
    }
}
// Tiles of 32 outer x 32 (16 for 4x4 kernels) inner channels with all their taps pass through LDS, so that the reference
// layout ([outer][inner][tap], rows of inner * taps contiguous floats) and both GEMM layouts are read and written in
// contiguous 64-128 byte runs; element by element the two transposed copies were 4-byte writes a row apart each.
// Kernels with more than 16 taps (the 7x7 ingest convolution: a few thousand elements) keep the element-wise walk.
constexpr int kRpMaxTaps = 16;
constexpr int kRpBatch = 12;
constexpr int kRpLds = 32 * 297;   // 32 rows of taps * (CT + 1) floats, made odd: 9 * 33 = 297, 16 * 17 + 1 = 273
struct RepackTile {
    int R, Cn, taps, CT, CTP, ldl, tc, ntiles;
    __device__ __forceinline__ RepackTile(const RepackDesc& d) {
        R = d.transposed ? d.n_in : d.n_out;     // outer dimension of the reference tensor
        Cn = d.transposed ? d.n_out : d.n_in;    // inner
        taps = d.taps;
        CT = taps <= 9 ? 32 : 16;
        CTP = CT + 1;
        ldl = (taps * CTP) | 1;
        tc = (Cn + CT - 1) / CT;
        ntiles = ((R + 31) / 32) * tc;
    }
};
__global__ __launch_bounds__(256) void repack_kernel(const RepackDesc* __restrict__ descs, const float* __restrict__ params,
                                                     float* __restrict__ wf, float* __restrict__ wb) {
    __shared__ float L[kRpLds];
    const RepackDesc d = descs[blockIdx.y];
    const int tid = threadIdx.x;
    if (d.taps > kRpMaxTaps) {
        const long long numel = (long long)d.n_out * d.n_in * d.taps;
        for (long long idx = (long long)blockIdx.x * 256 + tid; idx < numel; idx += (long long)gridDim.x * 256) {
            int co, ci, tap;
            repack_index(d, idx, co, ci, tap);
            const float v = params[d.src_off + idx];
            wf[d.f_off + ((long long)tap * d.n_in + ci) * d.f_ld + co] = v;
            wb[d.b_off + ((long long)tap * d.n_out + co) * d.b_ld + ci] = v;
        }
        return;
    }
    const RepackTile t(d);
    // destination A: rows (tap, inner), contiguous over outer; destination B: rows (tap, outer), contiguous over inner
    float* A = d.transposed ? wb + d.b_off : wf + d.f_off;
    float* Bp = d.transposed ? wf + d.f_off : wb + d.b_off;
    const int ldA = d.transposed ? d.b_ld : d.f_ld, ldB = d.transposed ? d.f_ld : d.b_ld;
    const int grp = tid >> 5, l = tid & 31;
    for (int tile = blockIdx.x; tile < t.ntiles; tile += gridDim.x) {
        const int r0 = (tile / t.tc) * 32, c0 = (tile % t.tc) * t.CT;
        const int rw = min(32, t.R - r0), cw = min(t.CT, t.Cn - c0), rowlen = cw * t.taps;
        __syncthreads();   // the previous tile has been written out
        // kRpBatch loads in flight per thread (one after the other they cost a memory round trip each)
        const float* src = params + d.src_off + ((long long)r0 * t.Cn + c0) * t.taps;
        const int total = rw * rowlen, rstride = t.Cn * t.taps;
        for (int e0 = tid; e0 < total; e0 += 256 * kRpBatch) {
            float v[kRpBatch];
            int at[kRpBatch];
#pragma unroll
            for (int u = 0; u < kRpBatch; ++u) {
                const int e = min(e0 + 256 * u, total - 1);
                const int r = e / rowlen, j = e - r * rowlen, c = j / t.taps, tap = j - c * t.taps;
                v[u] = src[(long long)r * rstride + j];
                at[u] = r * t.ldl + tap * t.CTP + c;
            }
#pragma unroll
            for (int u = 0; u < kRpBatch; ++u)
                if (e0 + 256 * u < total) L[at[u]] = v[u];
        }
        __syncthreads();
        for (int row = grp; row < t.taps * cw; row += 8) {
            const int tap = row / cw, c = row - tap * cw;
            if (l < rw) A[((long long)tap * t.Cn + c0 + c) * ldA + r0 + l] = L[l * t.ldl + tap * t.CTP + c];
        }
        for (int row = grp; row < t.taps * rw; row += 8) {
            const int tap = row / rw, r = row - tap * rw;
            if (l < cw) Bp[((long long)tap * t.R + r0 + r) * ldB + c0 + l] = L[r * t.ldl + tap * t.CTP + l];
        }
    }
}
__global__ __launch_bounds__(256) void unpack_grads_kernel(const RepackDesc* __restrict__ descs, const float* __restrict__ dwf,
                                                           float* __restrict__ grads) {
    __shared__ float L[kRpLds];
    const RepackDesc d = descs[blockIdx.y];
    const int tid = threadIdx.x;
    if (d.taps > kRpMaxTaps) {
        const long long numel = (long long)d.n_out * d.n_in * d.taps;
        for (long long idx = (long long)blockIdx.x * 256 + tid; idx < numel; idx += (long long)gridDim.x * 256) {
            int co, ci, tap;
            repack_index(d, idx, co, ci, tap);
            grads[d.src_off + idx] = dwf[d.f_off + ((long long)tap * d.n_in + ci) * d.f_ld + co];
        }
        return;
    }
    const RepackTile t(d);
    const float* src = dwf + d.f_off;   // rows (tap, cin) of f_ld floats: contiguous over the outer dimension unless transposed
    const int grp = tid >> 5, l = tid & 31;
    for (int tile = blockIdx.x; tile < t.ntiles; tile += gridDim.x) {
        const int r0 = (tile / t.tc) * 32, c0 = (tile % t.tc) * t.CT;
        const int rw = min(32, t.R - r0), cw = min(t.CT, t.Cn - c0), rowlen = cw * t.taps;
        __syncthreads();
        // rows of 32 lanes: (tap, inner) x outer, or (tap, outer) x inner when transposed; kRpBatch loads in flight
        const int per = d.transposed ? rw : cw, lanes = d.transposed ? cw : rw, nrows = t.taps * per;
        const long long base = d.transposed ? (long long)r0 * d.f_ld + c0 : (long long)c0 * d.f_ld + r0;
        const long long tapstride = (long long)(d.transposed ? t.R : t.Cn) * d.f_ld;
        for (int row0 = grp; row0 < nrows; row0 += 8 * kRpBatch) {
            float v[kRpBatch];
            int at[kRpBatch];
#pragma unroll
            for (int u = 0; u < kRpBatch; ++u) {
                const int row = min(row0 + 8 * u, nrows - 1);
                const int tap = row / per, i = row - tap * per, ll = min(l, lanes - 1);
                v[u] = src[base + tap * tapstride + (long long)i * d.f_ld + ll];
                at[u] = d.transposed ? i * t.ldl + tap * t.CTP + ll : ll * t.ldl + tap * t.CTP + i;
            }
#pragma unroll
            for (int u = 0; u < kRpBatch; ++u)
                if (row0 + 8 * u < nrows && l < lanes) L[at[u]] = v[u];
        }
        __syncthreads();
        for (int r = grp; r < rw; r += 8) {
            float* dst = grads + d.src_off + ((long long)(r0 + r) * t.Cn + c0) * t.taps;
            for (int j = l; j < rowlen; j += 32) {
                const int c = j / t.taps, tap = j - c * t.taps;
                dst[j] = L[r * t.ldl + tap * t.CTP + c];
            }
        }
    }
}

// ---- optimiser ----------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sqnorm_kernel(const float* __restrict__ g, long long n, float* __restrict__ out) {
    __shared__ float sh[4];
    float acc = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) acc += g[i] * g[i];
    acc = block_sum(acc, sh);
    if (threadIdx.x == 0) out[blockIdx.x] = acc;
}
__global__ __launch_bounds__(256) void adamw_kernel(const AdamWArgs a) {
    __shared__ float sh[4];
    float coef = a.grad_scale;
    if (a.max_norm > 0.f) {
        // every workgroup adds the same partial sums in the same order: one norm, bit for bit, everywhere
        float acc = 0.f;
        for (int i = threadIdx.x; i < a.n_parts; i += 256) acc += a.sq_parts[i];
        const float norm = sqrtf(block_sum(acc, sh)) * a.grad_scale;
        coef *= fminf(1.0f, a.max_norm / (norm + 1e-6f));
    }
    const float step_size = a.lr / a.bias1, inv_sqrt_b2 = 1.0f / sqrtf(a.bias2);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < a.n; i += (long long)gridDim.x * 256) {
        const float g = a.g[i] * coef;
        float p = a.p[i] * (1.0f - a.lr * a.weight_decay);
        const float m = a.beta1 * a.m[i] + (1.0f - a.beta1) * g;
        const float v = a.beta2 * a.v[i] + (1.0f - a.beta2) * g * g;
        a.m[i] = m;
        a.v[i] = v;
        p -= step_size * m / (sqrtf(v) * inv_sqrt_b2 + a.eps);
        a.p[i] = p;
    }
}

inline int grid_for(long long n, int per_block = 256, int cap = 8192) {
    long long g = (n + per_block - 1) / per_block;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

}   // namespace

int launch_gn_stats(const void* x, int adt, int B, int HW, int C, int G, double* stats, hipStream_t s) {
    DMN_REQUIRE(C % G == 0 && (C / G) % 4 == 0, DMN_ERR_INVALID, "gn: C=%d G=%d", C, G);
    const long long per = (long long)HW * (C / G) / 4;
    int split = (int)((per + 2047) / 2048);
    const int cap = 2048 / (B * G) > 1 ? 2048 / (B * G) : 1;
    if (split > cap) split = cap;
    if (split < 1) split = 1;
    DMN_DISPATCH_T(adt, gn_stats_kernel<T><<<dim3(B * G, split), 256, 0, s>>>((const T*)x, HW, C, G, stats));
    DMN_LAUNCH_CHECK();
    return DMN_OK;
}

int launch_gn_apply(const GnArgs& a, hipStream_t s) {
    DMN_REQUIRE(a.C % 4 == 0 && a.C % a.G == 0 && (a.C / a.G) % 4 == 0, DMN_ERR_INVALID, "gn: C=%d G=%d", a.C, a.G);
    const int cap = 4096 / a.B > 1 ? 4096 / a.B : 1;
    const dim3 grid(grid_for((long long)a.HW * a.C / 4, 256 * 2, cap), a.B);
    DMN_DISPATCH_T(a.adt, gn_apply_kernel<T><<<grid, 256, 0, s>>>(a));
    DMN_LAUNCH_CHECK();
    return DMN_OK;
}

int launch_gn_bwd(const GnArgs& a, hipStream_t s) {
    const int c4n = a.C / 4;
    DMN_REQUIRE(a.C % 4 == 0 && c4n <= 256 && 256 % c4n == 0 && a.G <= 64, DMN_ERR_INVALID, "gn backward: C=%d G=%d", a.C, a.G);
    DMN_REQUIRE(a.dx && a.dgamma && a.dbeta, DMN_ERR_INVALID, "gn backward: null output");
    const int npl = 256 / c4n;
    int split = (a.HW + npl * 8 - 1) / (npl * 8);
    const int cap = 2048 / a.B > 1 ? 2048 / a.B : 1;
    if (split > cap) split = cap;
    if (split < 1) split = 1;
    DMN_DISPATCH_T(a.adt, gn_bwd_sums_kernel<T><<<dim3(split, a.B), 256, 0, s>>>(a));
    DMN_LAUNCH_CHECK();
    const int nb = grid_for((long long)a.HW * c4n, 256 * 4, cap);
    DMN_DISPATCH_T(a.adt, gn_bwd_apply_kernel<T><<<dim3(nb, a.B), 256, 0, s>>>(a));
    DMN_LAUNCH_CHECK();
    return DMN_OK;
}

int launch_axpby(void* dst, const void* src, int adt, long long n, int accumulate, hipStream_t s) {
    DMN_REQUIRE((n & 3) == 0, DMN_ERR_INVALID, "axpby: %lld elements", n);
    DMN_DISPATCH_T(adt, axpby_kernel<T><<<grid_for(n / 4, 512), 256, 0, s>>>((T*)dst, (const T*)src, n / 4, accumulate));
    DMN_LAUNCH_CHECK();
    return DMN_OK;
}
int launch_add(void* out, const void* a, const void* b, int adt, long long n, hipStream_t s) {
    DMN_REQUIRE((n & 3) == 0, DMN_ERR_INVALID, "add: %lld elements", n);
    DMN_DISPATCH_T(adt, add_kernel<T><<<grid_for(n / 4, 512), 256, 0, s>>>((T*)out, (const T*)a, (const T*)b, n / 4));
    DMN_LAUNCH_CHECK();
    return DMN_OK;
}
int launch_copy_slice(void* dst, int ldd, int doff, const void* src, int lds, int soff, int C, long long rows, int adt,
                      int accumulate, hipStream_t s) {
    DMN_REQUIRE(((C | ldd | doff | lds | soff) & 3) == 0, DMN_ERR_INVALID, "copy_slice: widths must be multiples of 4");
    DMN_DISPATCH_T(adt, copy_slice_kernel<T><<<grid_for(rows * C / 4, 512), 256, 0, s>>>((T*)dst, ldd, doff, (const T*)src, lds, soff, C,
                                                                                          rows, accumulate));
    DMN_LAUNCH_CHECK();
    return DMN_OK;
}
int launch_act_fwd(const void* x, void* y, int adt, long long n, int act, hipStream_t s) {
    DMN_DISPATCH_T(adt, act_fwd_kernel<T><<<grid_for(n, 1024), 256, 0, s>>>((const T*)x, (T*)y, n, act));
    DMN_LAUNCH_CHECK();
    return DMN_OK;
}
int launch_act_bwd(const void* x, const void* dy, void* dx, int adt, long long n, int act, int accumulate, hipStream_t s) {
    DMN_DISPATCH_T(adt, act_bwd_kernel<T><<<grid_for(n, 1024), 256, 0, s>>>((const T*)x, (const T*)dy, (T*)dx, n, act, accumulate));
    DMN_LAUNCH_CHECK();
    return DMN_OK;
}
int launch_class_bias_fwd(void* x, int adt, const float* table, const long long* labels, int pad, int B, int HW, int C, hipStream_t s) {
    const dim3 grid(grid_for((long long)HW * C, 1024, 256), B);
    DMN_DISPATCH_T(adt, class_bias_fwd_kernel<T><<<grid, 256, 0, s>>>((T*)x, table, labels, pad, HW, C));
    DMN_LAUNCH_CHECK();
    return DMN_OK;
}
int launch_class_bias_bwd(const void* dy, int adt, const long long* labels, int pad, float* dtable, int B, int HW, int C, hipStream_t s) {
    DMN_REQUIRE(C <= 256, DMN_ERR_LIMIT, "class embedding width %d", C);
    const int npl = 256 / C;
    int split = (HW + npl * 16 - 1) / (npl * 16);
    if (split > 64) split = 64;
    DMN_DISPATCH_T(adt, class_bias_bwd_kernel<T><<<dim3(split, B), 256, 0, s>>>((const T*)dy, labels, pad, dtable, HW, C));
    DMN_LAUNCH_CHECK();
    return DMN_OK;
}
int launch_sinus(const void* t, int t_is_float, int B, int dim, float* out, hipStream_t s) {
    sinus_kernel<<<(B * dim + 255) / 256, 256, 0, s>>>(t, t_is_float, B, dim, out);
    DMN_LAUNCH_CHECK();
    return DMN_OK;
}
int launch_train_ingest(const float* x0, const float* eps, const long long* t, const float* sa, const float* sb, int n_timesteps,
                        void* xt, int adt, float* tgt, int B, int C, int H, int W, hipStream_t s) {
    DMN_DISPATCH_T(adt, train_ingest_kernel<T><<<grid_for((long long)B * C * H * W, 1024), 256, 0, s>>>(x0, eps, t, sa, sb, (T*)xt, tgt, B,
                                                                                                         C, H * W, n_timesteps));
    DMN_LAUNCH_CHECK();
    return DMN_OK;
}
int launch_loss(const void* out, const float* tgt, int adt, long long n, long long n_mean, int kind, float* loss, void* dout,
                hipStream_t s) {
    DMN_DISPATCH_T(adt, loss_kernel<T><<<grid_for(n, 1024, 1024), 256, 0, s>>>((const T*)out, tgt, n, n_mean, kind, loss, (T*)dout));
    DMN_LAUNCH_CHECK();
    return DMN_OK;
}
int launch_nhwc_to_nchw(const void* in, int adt, float* out, int B, int C, int HW, hipStream_t s) {
    DMN_DISPATCH_T(adt, nhwc_to_nchw_kernel<T><<<grid_for((long long)B * C * HW, 1024), 256, 0, s>>>((const T*)in, out, B, C, HW));
    DMN_LAUNCH_CHECK();
    return DMN_OK;
}
int linattn_train_splits(int B, int n, int heads) {
    const int tiles = (n + 63) / 64;
    static int target = 0;   // workgroups a launch aims for (tuning aid: DMN_LINATTN_WGS)
    if (!target) {
        const char* e = getenv("DMN_LINATTN_WGS");
        target = e ? atoi(e) : 2048;
        if (target < 64) target = 64;
    }
    int splits = target / (B * heads);
    if (splits > tiles) splits = tiles;
    if (splits < 1) splits = 1;
    if (splits > 256) splits = 256;
    return splits;
}
int launch_linattn_train_fwd(const void* qkv, void* out, int adt, float* ctx, float* kstat, float* pmax, int B, int n, int heads,
                             int bf16, hipStream_t s) {
    const dim3 grid(B * heads, linattn_train_splits(B, n, heads));
    DMN_DISPATCH_T(adt, linattn_kmax_kernel<T><<<grid, 256, 0, s>>>((const T*)qkv, pmax, n, heads));
    DMN_LAUNCH_CHECK();
    DMN_DISPATCH_T(adt, linattn_ctx_kernel<T><<<grid, 256, 0, s>>>((const T*)qkv, pmax, ctx, kstat, n, heads));
    DMN_LAUNCH_CHECK();
    if (adt == 1) linattn_out_kernel<__bf16, true><<<grid, 256, 0, s>>>((const __bf16*)qkv, ctx, kstat, (__bf16*)out, n, heads);
    else if (bf16) linattn_out_kernel<float, true><<<grid, 256, 0, s>>>((const float*)qkv, ctx, kstat, (float*)out, n, heads);
    else linattn_out_kernel<float, false><<<grid, 256, 0, s>>>((const float*)qkv, ctx, kstat, (float*)out, n, heads);
    DMN_LAUNCH_CHECK();
    return DMN_OK;
}
int launch_linattn_train_bwd(const void* qkv, const void* dout, int adt, const float* ctx, const float* kstat, float* dctx, void* dqkv,
                             int B, int n, int heads, int bf16, hipStream_t s) {
    const dim3 grid(B * heads, linattn_train_splits(B, n, heads));
    DMN_DISPATCH_T(adt, linattn_dctx_kernel<T><<<grid, 256, 0, s>>>((const T*)qkv, (const T*)dout, dctx, n, heads));
    DMN_LAUNCH_CHECK();
    if (adt == 1)
        linattn_bwd_pix_kernel<__bf16, true><<<grid, 256, 0, s>>>((const __bf16*)qkv, (const __bf16*)dout, ctx, kstat, dctx, (__bf16*)dqkv, n,
                                                                  heads);
    else if (bf16)
        linattn_bwd_pix_kernel<float, true><<<grid, 256, 0, s>>>((const float*)qkv, (const float*)dout, ctx, kstat, dctx, (float*)dqkv, n, heads);
    else
        linattn_bwd_pix_kernel<float, false><<<grid, 256, 0, s>>>((const float*)qkv, (const float*)dout, ctx, kstat, dctx, (float*)dqkv, n, heads);
    DMN_LAUNCH_CHECK();
    return DMN_OK;
}
int launch_attn_train_fwd(const void* qkv, void* out, int adt, float* lse, int B, int n, int heads, int bf16, hipStream_t s) {
    if (attn_train_on_matrix_cores(n, bf16)) {
        const dim3 grid(heads, B, n / 64);
        DMN_DISPATCH_T(adt, attn_fwd_mfma_kernel<T><<<grid, 256, 0, s>>>((const T*)qkv, (T*)out, lse, n, heads));
        DMN_LAUNCH_CHECK();
        return DMN_OK;
    }
    DMN_DISPATCH_T(adt, attn_fwd_kernel<T><<<B * heads, 256, 0, s>>>((const T*)qkv, (T*)out, lse, n, heads));
    DMN_LAUNCH_CHECK();
    return DMN_OK;
}
int launch_attn_train_bwd(const void* qkv, const void* out, const void* dout, int adt, const float* lse, void* dqkv, int B, int n,
                          int heads, int bf16, hipStream_t s) {
    // lse: [B*heads][2][n] -- the second half holds the per-query sums between the two launches
    float* dsum = const_cast<float*>(lse) + (size_t)B * heads * n;
    if (attn_train_on_matrix_cores(n, bf16)) {
        const dim3 grid(heads, B, n / 64);
        DMN_DISPATCH_T(adt, attn_bwd_dq_mfma_kernel<T><<<grid, 256, 0, s>>>((const T*)qkv, (const T*)out, (const T*)dout, lse, (T*)dqkv, dsum, n, heads));
        DMN_LAUNCH_CHECK();
        DMN_DISPATCH_T(adt, attn_bwd_dkv_mfma_kernel<T><<<grid, 256, 0, s>>>((const T*)qkv, (const T*)dout, lse, dsum, (T*)dqkv, n, heads));
        DMN_LAUNCH_CHECK();
        return DMN_OK;
    }
    const dim3 grid(B * heads, (n + 63) / 64);
    DMN_DISPATCH_T(adt, attn_bwd_dq_kernel<T><<<grid, 256, 0, s>>>((const T*)qkv, (const T*)out, (const T*)dout, lse, (T*)dqkv, dsum, n, heads));
    DMN_LAUNCH_CHECK();
    DMN_DISPATCH_T(adt, attn_bwd_dkv_kernel<T><<<grid, 256, 0, s>>>((const T*)qkv, (const T*)dout, lse, dsum, (T*)dqkv, n, heads));
    DMN_LAUNCH_CHECK();
    return DMN_OK;
}
int launch_repack(const RepackDesc* descs_dev, int n_desc, long long max_numel, const float* params, float* wf, float* wb,
                  hipStream_t s) {
    repack_kernel<<<dim3(grid_for(max_numel, 4096, 128), n_desc), 256, 0, s>>>(descs_dev, params, wf, wb);
    DMN_LAUNCH_CHECK();
    return DMN_OK;
}
int launch_unpack_grads(const RepackDesc* descs_dev, int n_desc, long long max_numel, const float* dwf, float* grads, hipStream_t s) {
    unpack_grads_kernel<<<dim3(grid_for(max_numel, 4096, 128), n_desc), 256, 0, s>>>(descs_dev, dwf, grads);
    DMN_LAUNCH_CHECK();
    return DMN_OK;
}
namespace {
__global__ void spin_kernel(unsigned long long ticks) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < (1 << 18); ++i) {   // leaves after at most 2^18 polls whatever the counter does
        if (__builtin_amdgcn_s_memrealtime() - t0 >= ticks) break;
        __builtin_amdgcn_s_sleep(16);
    }
}
}   // namespace

int launch_spin(int us, hipStream_t s) {
    DMN_REQUIRE(us >= 0 && us <= 5000, DMN_ERR_INVALID, "spin: %d us", us);
    spin_kernel<<<1, 64, 0, s>>>((unsigned long long)us * 100ull);
    DMN_LAUNCH_CHECK();
    return DMN_OK;
}

int launch_sqnorm(const float* g, long long n, float* parts, int* n_parts, hipStream_t s) {
    const int grid = grid_for(n, 2048, kSqnormMaxParts);
    sqnorm_kernel<<<grid, 256, 0, s>>>(g, n, parts);
    DMN_LAUNCH_CHECK();
    *n_parts = grid;
    return DMN_OK;
}
int launch_adamw(const AdamWArgs& a, hipStream_t s) {
    adamw_kernel<<<grid_for(a.n, 1024), 256, 0, s>>>(a);
    DMN_LAUNCH_CHECK();
    return DMN_OK;
}
