// Fused linear attention for gfx950: PreNorm GroupNorm(1) -> to_qkv (1x1) -> softmax_d(q) * d^-1/2,
// softmax_n(k) -> ctx = k v^T -> out = ctx^T q -> to_out (1x1) + bias, without ever writing q/k/v or the
// attention output to HBM (diffusion_model_nemo/parts/mha.py:33-59, utils.py:85-93). 4 heads x 32 channels.
//
//   pass 1 (la_fused_ctx_kernel)  grid (splits, B): a workgroup stages its pixel slice of x (normalised) in LDS;
//       wave h owns head h: K and V projections as MFMA GEMMs D[px][cout], a first sweep for the per-channel
//       maximum over the slice, a second sweep that exponentiates and accumulates ctx[d][e] += P^T V with the
//       accumulator tiles fed straight back as MFMA operands (pixels are the k dimension). Writes
//       {ctx, max, sum} per (image, head, split). No inter-wave communication.
//   pass 2 (la_fused_out_kernel)  grid (slices, B): rescales and sums the split partials into ctx^T (LDS), then
//       per 16-pixel tile chains three GEMMs in registers: Q = Wq x, softmax over d, out = ctx^T q,
//       y = Wout out + bias; stores y (NHWC) and GroupNorm(1) partial statistics of y.
//
// "Accumulator as operand": a 16x16 accumulator tile holds rows 4*(lane>>4)+r of column lane&15. Two tiles
// (bf16) or one tile (f32) form the 16-byte operand fragment of the next MFMA whose k dimension is the tile's
// row index, in the k order  pos -> 4*(pos>>3) + (pos&3) + 16*((pos>>2)&1)  (bf16; identity for f32). The
// other operand of that MFMA (ctx^T rows, Wout columns) is stored in the same order.
#include "dmn_kernels.h"
#include "dmn_norm.h"

namespace {

constexpr int kHeads = 4, kD = 32, kHidden = 128;
constexpr int kPartStride = kD * kD + 2 * kD;

template <typename T> struct Sub;   // k-steps per 32 logical k values
template <> struct Sub<__bf16> { static constexpr int N = 1; };
template <> struct Sub<float> { static constexpr int N = 2; };

// operand fragment of k-step `sub` built from the accumulator tiles t0 (rows 0..15) and t1 (rows 16..31)
__device__ __forceinline__ Vec16<__bf16> acc_operand(const f32x4& t0, const f32x4& t1, int, __bf16*) {
    Vec16<__bf16> v;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        v.v[j] = (__bf16)t0[j];
        v.v[4 + j] = (__bf16)t1[j];
    }
    return v;
}
__device__ __forceinline__ Vec16<float> acc_operand(const f32x4& t0, const f32x4& t1, int sub, float*) {
    Vec16<float> v;
    v.v = sub == 0 ? t0 : t1;
    return v;
}

// position (0..31) inside a 32-wide k block -> logical index, for the operand that pairs with acc_operand
template <typename T> __device__ __host__ inline int perm32(int pos) {
    if (sizeof(T) == 4) return pos;
    return 4 * (pos >> 3) + (pos & 3) + 16 * ((pos >> 2) & 1);
}

template <typename T>
__device__ __forceinline__ decltype(Vec16<T>::v) load_frag(const T* base, int idx, int lane) {
    using V = decltype(Vec16<T>::v);
    return *reinterpret_cast<const V*>(base + ((size_t)idx * 64 + lane) * ElemTraits<T>::kPer16B);
}

// stage rows [p0, p0 + rows) of x[b] (C channels) into LDS, normalised with GroupNorm(1); rows >= n are zero
template <typename T>
__device__ __forceinline__ void stage_x(const LinAttnFusedArgs& a, int b, int p0, int rows, char* sX, int row_bytes,
                                        const float* sCoef) {
    constexpr int PER = ElemTraits<T>::kPer16B;
    using V = decltype(Vec16<T>::v);
    const int C = a.C;
    const int parts = C / PER;                 // 256 % parts == 0: a thread keeps its channel offset
    const int part = threadIdx.x % parts;
    float ca[PER], cb[PER];
#pragma unroll
    for (int e = 0; e < PER; ++e) {
        ca[e] = sCoef[part * PER + e];
        cb[e] = sCoef[C + part * PER + e];
    }
    const T* x = static_cast<const T*>(a.x) + (size_t)b * a.n * C;
    const int units = rows * parts;
    for (int u = threadIdx.x; u < units; u += 256) {
        const int r = u / parts;
        const int p = p0 + r;
        Vec16<T> v;
        if (p < a.n) {
            v.v = *reinterpret_cast<const V*>(x + (size_t)p * C + part * PER);
#pragma unroll
            for (int e = 0; e < PER; ++e) v.set(e, v.get(e) * ca[e] + cb[e]);
        } else {
#pragma unroll
            for (int e = 0; e < PER; ++e) v.set(e, 0.f);
        }
        *reinterpret_cast<V*>(sX + (size_t)r * row_bytes + part * 16) = v.v;
    }
}

// ------------------------------------------------------------------------------------------------------------
// pass 1: grid (splits, B); slice = nsub * 128 pixels
// ------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) la_fused_ctx_kernel(const LinAttnFusedArgs a) {
    constexpr int PER = ElemTraits<T>::kPer16B;
    constexpr int KCH = ElemTraits<T>::kChunk;
    constexpr int SUB = Sub<T>::N;
    using V = decltype(Vec16<T>::v);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int C = a.C, KS = C / KCH;
    const int row_bytes = C * (int)sizeof(T) + 32;
    const int slice = a.slice1;
    float* sCoef = reinterpret_cast<float*>(smem);            // 3*C + 2
    char* sX = smem + ((3 * C + 8) * 4 + 15) / 16 * 16;
    const int split = blockIdx.x, b = blockIdx.y;
    const int lane = threadIdx.x & 63, h = threadIdx.x >> 6;
    const int l15 = lane & 15, lq = lane >> 4;
    const int p0 = split * slice;

    norm_coefs_to_lds(a.norm, b, C, sCoef, sCoef + 3 * C);
    stage_x<T>(a, b, p0, slice, sX, row_bytes, sCoef);
    __syncthreads();

    const T* wk = static_cast<const T*>(a.wk);   // fragment order [ct 0..7][ks][lane][PER]; head h owns ct 2h, 2h+1
    const T* wv = static_cast<const T*>(a.wv);
    const int npt = slice / 16;                  // pixel tiles in the slice

    // ---- sweep 1: per-channel maximum of k over the slice ------------------------------------------------
    float mx[2] = {-INFINITY, -INFINITY};
    for (int pt = 0; pt < npt; ++pt) {
        f32x4 kacc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
        const char* xrow = sX + (size_t)(pt * 16 + l15) * row_bytes + lq * 16;
        for (int ks = 0; ks < KS; ++ks) {
            Vec16<T> xf;
            xf.v = *reinterpret_cast<const V*>(xrow + ks * 64);
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                Vec16<T> wf;
                wf.v = load_frag<T>(wk, (2 * h + ct) * KS + ks, lane);
                mma_step(xf, wf, kacc[ct]);       // D[px][cout]
            }
        }
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool valid = p0 + pt * 16 + lq * 4 + r < a.n;
                mx[ct] = valid ? fmaxf(mx[ct], kacc[ct][r]) : mx[ct];
            }
    }
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
        mx[ct] = fmaxf(mx[ct], __shfl_xor(mx[ct], 16, 64));
        mx[ct] = fmaxf(mx[ct], __shfl_xor(mx[ct], 32, 64));
    }

    // ---- sweep 2: P = exp(k - max), Z = sum P, ctx += P^T V ------------------------------------------------
    f32x4 ctx[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) ctx[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float zs[2] = {0.f, 0.f};
    for (int pp = 0; pp < npt; pp += 2) {        // 32 pixels = one bf16 k-step (two f32 k-steps)
        f32x4 kacc[2][2], vacc[2][2];             // [pixel tile of the pair][channel tile]
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) kacc[i][j] = vacc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int ks = 0; ks < KS; ++ks) {
            Vec16<T> wkf[2], wvf[2];
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                wkf[ct].v = load_frag<T>(wk, (2 * h + ct) * KS + ks, lane);
                wvf[ct].v = load_frag<T>(wv, (2 * h + ct) * KS + ks, lane);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                Vec16<T> xf;
                xf.v = *reinterpret_cast<const V*>(sX + (size_t)((pp + i) * 16 + l15) * row_bytes + lq * 16 + ks * 64);
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) {
                    mma_step(xf, wkf[ct], kacc[i][ct]);
                    mma_step(xf, wvf[ct], vacc[i][ct]);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool valid = p0 + (pp + i) * 16 + lq * 4 + r < a.n;
                    const float e = valid ? expf(kacc[i][ct][r] - mx[ct]) : 0.f;
                    kacc[i][ct][r] = e;
                    zs[ct] += e;
                }
#pragma unroll
        for (int sub = 0; sub < SUB; ++sub)
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                const Vec16<T> pa = acc_operand(kacc[0][dt], kacc[1][dt], sub, (T*)nullptr);   // A[d][k = px]
#pragma unroll
                for (int et = 0; et < 2; ++et) {
                    const Vec16<T> vb = acc_operand(vacc[0][et], vacc[1][et], sub, (T*)nullptr);   // B[k = px][e]
                    mma_step(pa, vb, ctx[dt][et]);
                }
            }
    }
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
        zs[ct] += __shfl_xor(zs[ct], 16, 64);
        zs[ct] += __shfl_xor(zs[ct], 32, 64);
    }
    float* dst = a.ctx_part + ((size_t)(b * kHeads + h) * a.splits + split) * kPartStride;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int et = 0; et < 2; ++et)
#pragma unroll
            for (int r = 0; r < 4; ++r) dst[(dt * 16 + lq * 4 + r) * kD + et * 16 + l15] = ctx[dt][et][r];
    if (lq == 0) {
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            dst[kD * kD + ct * 16 + l15] = mx[ct];
            dst[kD * kD + kD + ct * 16 + l15] = zs[ct];
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// pass 2: grid (slices, B); slice = 64 * tpw pixels, wave w owns pixel tiles w*tpw .. w*tpw + tpw-1
// ------------------------------------------------------------------------------------------------------------
template <typename T, int C>
__global__ void __launch_bounds__(256) la_fused_out_kernel(const LinAttnFusedArgs a) {
    constexpr int PER = ElemTraits<T>::kPer16B;
    constexpr int KCH = ElemTraits<T>::kChunk;
    constexpr int SUB = Sub<T>::N;
    constexpr int KS = C / KCH;
    constexpr int YT = C / 16;
    constexpr int CTX_ROW = kD * (int)sizeof(T) + 32;
    using V = decltype(Vec16<T>::v);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int row_bytes = C * (int)sizeof(T) + 32;
    const int slice = 64 * a.tpw;
    float* sCoef = reinterpret_cast<float*>(smem);                         // 3*C + 8
    char* sCtx = smem + ((3 * C + 8) * 4 + 15) / 16 * 16;                  // [4][32 rows e][CTX_ROW]
    float* sFac = reinterpret_cast<float*>(sCtx + kHeads * kD * CTX_ROW);  // [128][splits]
    char* sX = reinterpret_cast<char*>(sFac) + ((kHidden * a.splits * 4 + 15) / 16) * 16;
    float* sRed = reinterpret_cast<float*>(sX + (size_t)slice * row_bytes);   // 8 floats
    const int b = blockIdx.y, tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lq = lane >> 4;
    const int p0 = blockIdx.x * slice;

    norm_coefs_to_lds(a.norm, b, C, sCoef, sCoef + 3 * C);
    stage_x<T>(a, b, p0, slice, sX, row_bytes, sCoef);

    // ---- merge the split partials: factor_s[d] = exp(m_s[d] - M[d]) * 32^-1/2 / sum_s Z_s[d] exp(m_s[d] - M[d])
    const float* part = a.ctx_part + (size_t)b * kHeads * a.splits * kPartStride;
    for (int hd = tid; hd < kHidden; hd += 256) {
        const int h = hd >> 5, d = hd & 31;
        const float* base = part + (size_t)h * a.splits * kPartStride + kD * kD + d;
        float M = -INFINITY;
        for (int s = 0; s < a.splits; ++s) M = fmaxf(M, base[(size_t)s * kPartStride]);
        float Z = 0.f;
        for (int s = 0; s < a.splits; ++s) {
            const float f = expf(base[(size_t)s * kPartStride] - M);
            sFac[hd * a.splits + s] = f;
            Z += base[(size_t)s * kPartStride + kD] * f;
        }
        const float inv = 0.17677669529663687f / Z;
        for (int s = 0; s < a.splits; ++s) sFac[hd * a.splits + s] *= inv;
    }
    __syncthreads();
    // ctx^T in operand order: row e, position pos holds ctx[d = perm32(pos)][e]
    for (int i = tid; i < kHeads * kD * kD; i += 256) {
        const int h = i >> 10, d = (i >> 5) & 31, e = i & 31;   // consecutive threads -> consecutive e (coalesced)
        const float* base = part + (size_t)h * a.splits * kPartStride + d * kD + e;
        float acc = 0.f;
        for (int s = 0; s < a.splits; ++s) acc = fmaf(base[(size_t)s * kPartStride], sFac[(h * kD + d) * a.splits + s], acc);
        int pos = 0;   // inverse of perm32
        if (sizeof(T) == 4) pos = d;
        else pos = ((d & 15) >> 2) * 8 + (d >> 4) * 4 + (d & 3);
        *reinterpret_cast<T*>(sCtx + (size_t)(h * kD + e) * CTX_ROW + pos * sizeof(T)) = from_f32<T>(acc);
    }
    __syncthreads();

    const T* wq = static_cast<const T*>(a.wq);     // [ct 0..7][ks][lane][PER]
    const T* wo = static_cast<const T*>(a.wo);     // [yt][h][sub][lane][PER], k in operand order
    float ssum = 0.f, ssq = 0.f;
    T* yout = static_cast<T*>(a.y) + (size_t)b * a.n * C;
    for (int t = 0; t < a.tpw; ++t) {
        const int ptile = wave * a.tpw + t;
        const int px = p0 + ptile * 16 + l15;       // this lane's pixel (column)
        // Q[cout][px] = Wq x
        f32x4 q[8];
#pragma unroll
        for (int ct = 0; ct < 8; ++ct) q[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
        const char* xrow = sX + (size_t)(ptile * 16 + l15) * row_bytes + lq * 16;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            Vec16<T> xf;
            xf.v = *reinterpret_cast<const V*>(xrow + ks * 64);
#pragma unroll
            for (int ct = 0; ct < 8; ++ct) {
                Vec16<T> wf;
                wf.v = load_frag<T>(wq, ct * KS + ks, lane);
                mma_step(wf, xf, q[ct]);
            }
        }
        // softmax over the 32 channels of each head (rows of tiles 2h, 2h+1; spread over the 4 lane quarters)
        f32x4 o[8];
#pragma unroll
        for (int h = 0; h < kHeads; ++h) {
            float m = q[2 * h][0];
#pragma unroll
            for (int r = 0; r < 4; ++r) m = fmaxf(m, fmaxf(q[2 * h][r], q[2 * h + 1][r]));
            m = fmaxf(m, __shfl_xor(m, 16, 64));
            m = fmaxf(m, __shfl_xor(m, 32, 64));
            float s = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                q[2 * h][r] = expf(q[2 * h][r] - m);
                q[2 * h + 1][r] = expf(q[2 * h + 1][r] - m);
                s += q[2 * h][r] + q[2 * h + 1][r];
            }
            s += __shfl_xor(s, 16, 64);
            s += __shfl_xor(s, 32, 64);
            const float inv = 1.0f / s;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                q[2 * h][r] *= inv;
                q[2 * h + 1][r] *= inv;
            }
            // out[e][px] = sum_d ctx[d][e] q[d][px]
            o[2 * h] = o[2 * h + 1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int sub = 0; sub < SUB; ++sub) {
                const Vec16<T> qb = acc_operand(q[2 * h], q[2 * h + 1], sub, (T*)nullptr);
#pragma unroll
                for (int et = 0; et < 2; ++et) {
                    Vec16<T> cf;
                    cf.v = *reinterpret_cast<const V*>(sCtx + (size_t)(h * kD + et * 16 + l15) * CTX_ROW + sub * 64 + lq * 16);
                    mma_step(cf, qb, o[2 * h + et]);
                }
            }
        }
        // y[c][px] = Wout out + bias
        f32x4 y[YT];
#pragma unroll
        for (int yt = 0; yt < YT; ++yt) y[yt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int h = 0; h < kHeads; ++h)
#pragma unroll
            for (int sub = 0; sub < SUB; ++sub) {
                const Vec16<T> ob = acc_operand(o[2 * h], o[2 * h + 1], sub, (T*)nullptr);
#pragma unroll
                for (int yt = 0; yt < YT; ++yt) {
                    Vec16<T> wf;
                    wf.v = load_frag<T>(wo, (yt * kHeads + h) * SUB + sub, lane);
                    mma_step(wf, ob, y[yt]);
                }
            }
        const bool valid = px < a.n;
#pragma unroll
        for (int yt = 0; yt < YT; ++yt) {
            const int c = yt * 16 + lq * 4;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = y[yt][r] + a.bias[c + r];
            if (valid) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    ssum += v[r];
                    ssq += v[r] * v[r];
                }
                if constexpr (sizeof(T) == 4) {
                    *reinterpret_cast<f32x4*>(yout + (size_t)px * C + c) = f32x4{v[0], v[1], v[2], v[3]};
                } else {
                    bf16x4 ov;
#pragma unroll
                    for (int r = 0; r < 4; ++r) ov[r] = (__bf16)v[r];
                    *reinterpret_cast<bf16x4*>(yout + (size_t)px * C + c) = ov;
                }
            }
        }
    }
    // GroupNorm(1) partial statistics of y: one slot per workgroup
    ssum = wave_sum(ssum);
    ssq = wave_sum(ssq);
    if (lane == 0) {
        sRed[2 * wave] = ssum;
        sRed[2 * wave + 1] = ssq;
    }
    __syncthreads();
    if (tid == 0) {
        float s = 0.f, ss = 0.f;
        for (int w = 0; w < 4; ++w) {
            s += sRed[2 * w];
            ss += sRed[2 * w + 1];
        }
        float* dst = a.y_stats + ((size_t)b * gridDim.x + blockIdx.x) * 2;
        dst[0] = s;
        dst[1] = ss;
    }
}

template <typename T, int C>
int launch_out(const LinAttnFusedArgs& a, hipStream_t s) {
    const int slice = 64 * a.tpw;
    const size_t row = (size_t)C * sizeof(T) + 32;
    const size_t lds = ((3 * C + 8) * 4 + 15) / 16 * 16 + (size_t)kHeads * kD * (kD * sizeof(T) + 32) +
                       ((size_t)kHidden * a.splits * 4 + 15) / 16 * 16 + slice * row + 64;
    DMN_REQUIRE(lds <= 160 * 1024, DMN_ERR_LIMIT, "linear attention: LDS request %zu too large", lds);
    static bool attr = false;
    if (!attr) {
        DMN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&la_fused_out_kernel<T, C>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr = true;
    }
    hipLaunchKernelGGL((la_fused_out_kernel<T, C>), dim3((a.n + slice - 1) / slice, a.B), dim3(256), lds, s, a);
    DMN_HIP_CHECK(hipGetLastError());
    return DMN_OK;
}

template <typename T>
int launch_both(LinAttnFusedArgs& a, hipStream_t s) {
    const size_t row = (size_t)a.C * sizeof(T) + 32;
    const size_t lds1 = ((3 * a.C + 8) * 4 + 15) / 16 * 16 + (size_t)a.slice1 * row + 64;
    DMN_REQUIRE(lds1 <= 160 * 1024, DMN_ERR_LIMIT, "linear attention: LDS request %zu too large", lds1);
    static bool attr = false;
    if (!attr) {
        DMN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&la_fused_ctx_kernel<T>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr = true;
    }
    hipLaunchKernelGGL(la_fused_ctx_kernel<T>, dim3(a.splits, a.B), dim3(256), lds1, s, a);
    DMN_HIP_CHECK(hipGetLastError());
    switch (a.C) {
        case 32: return launch_out<T, 32>(a, s);
        case 64: return launch_out<T, 64>(a, s);
        case 128: return launch_out<T, 128>(a, s);
        case 256: return launch_out<T, 256>(a, s);
    }
    dmn_set_error("fused linear attention: C=%d not instantiated", a.C);
    return DMN_ERR_INVALID;
}

}  // namespace

bool linattn_fused_supported(int C, int heads, int dim_head) {
    return heads == kHeads && dim_head == kD && (C == 32 || C == 64 || C == 128 || C == 256);
}

// pixel-slice geometry: pass 1 takes slice1 pixels per workgroup (largest that still gives >= 256 workgroups),
// pass 2 takes 64 * tpw
void linattn_fused_plan(int n, int B, int C, int dtype, int* slice1, int* splits, int* tpw, int* slices) {
    const int row = C * (dtype == DMN_F32 ? 4 : 2) + 32;
    int s1 = 512;
    while (s1 > 32 && ((long long)((n + s1 - 1) / s1) * B < 256 || s1 * row > 72 * 1024 || s1 / 2 >= n)) s1 /= 2;
    *slice1 = s1;
    *splits = (n + s1 - 1) / s1;
    int t = 4;
    while (t > 1 && ((long long)((n + 64 * t - 1) / (64 * t)) * B < 512 || 64 * t * row > 72 * 1024)) t /= 2;
    *tpw = t;
    *slices = (n + 64 * t - 1) / (64 * t);
}

int launch_linattn_fused(LinAttnFusedArgs& a, hipStream_t s) {
    DMN_REQUIRE(linattn_fused_supported(a.C, kHeads, kD), DMN_ERR_INVALID, "fused linear attention: C=%d unsupported", a.C);
    DMN_REQUIRE(a.slice1 >= 32 && a.slice1 % 32 == 0 && a.tpw >= 1 && a.splits == (a.n + a.slice1 - 1) / a.slice1 &&
                    a.slices == (a.n + 64 * a.tpw - 1) / (64 * a.tpw),
                DMN_ERR_INVALID, "fused linear attention: inconsistent plan");
    DMN_REQUIRE(a.norm.stats && a.norm.G == 1, DMN_ERR_INVALID, "fused linear attention: GroupNorm(1) statistics of x required");
    if (a.dtype == DMN_F32) return launch_both<float>(a, s);
    return launch_both<__bf16>(a, s);
}
