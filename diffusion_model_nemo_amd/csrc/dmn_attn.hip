// Attention cores on a materialised qkv tensor [B][n][3*hidden] (channels: q | k | v, each head-major,
// hidden = heads * d, d = 32), fp32 arithmetic.
//
//   linear attention  (diffusion_model_nemo/parts/mha.py:44-59):
//       q = softmax_d(q) * d^-1/2 ; k = softmax_n(k) ; ctx[d][e] = sum_n k[d][n] v[e][n] ;
//       out[e][n] = sum_d ctx[d][e] q[d][n]
//   softmax attention (parts/mha.py:16-30): out[i] = softmax_j((q_i * d^-1/2) . k_j) v_j
//
// The softmax over n of linear attention is split over workgroups: every workgroup takes a slice of the
// pixels, uses its LOCAL per-channel maximum, and writes {ctx, max, sum}; the consumer rescales the
// partials to the global maximum (exact in real arithmetic, deterministic slot order).
#include "dmn_kernels.h"

namespace {

constexpr int kD = 32;
constexpr int kLaChunk = 256;           // pixels per workgroup in the context pass
constexpr int kPartStride = kD * kD + 2 * kD;

template <typename T>
__device__ __forceinline__ void load32(const T* p, float* dst) {
    constexpr int PER = ElemTraits<T>::kPer16B;
    using V = decltype(Vec16<T>::v);
#pragma unroll
    for (int i = 0; i < kD / PER; ++i) {
        Vec16<T> v;
        v.v = *reinterpret_cast<const V*>(p + i * PER);
#pragma unroll
        for (int j = 0; j < PER; ++j) dst[i * PER + j] = v.get(j);
    }
}
template <typename T>
__device__ __forceinline__ void store32(T* p, const float* src) {
    constexpr int PER = ElemTraits<T>::kPer16B;
    using V = decltype(Vec16<T>::v);
#pragma unroll
    for (int i = 0; i < kD / PER; ++i) {
        Vec16<T> v;
#pragma unroll
        for (int j = 0; j < PER; ++j) v.set(j, src[i * PER + j]);
        *reinterpret_cast<V*>(p + i * PER) = v.v;
    }
}

// grid (splits, heads, B); 256 threads
template <typename T>
__global__ void __launch_bounds__(256) linattn_ctx_kernel(const LinAttnArgs a) {
    __shared__ float sK[kLaChunk][kD + 1];
    __shared__ float sV[kLaChunk][kD + 4];
    __shared__ float sRed[8][kD];
    __shared__ float sM[kD];
    const int split = blockIdx.x, h = blockIdx.y, b = blockIdx.z, tid = threadIdx.x;
    const int hidden = a.heads * kD;
    const int p0 = split * kLaChunk;
    const int np = min(kLaChunk, a.n - p0);
    {
        float kk[kD], vv[kD];
        if (tid < np) {
            const T* row = static_cast<const T*>(a.qkv) + ((size_t)b * a.n + p0 + tid) * 3 * hidden;
            load32(row + hidden + h * kD, kk);
            load32(row + 2 * hidden + h * kD, vv);
        } else {
#pragma unroll
            for (int j = 0; j < kD; ++j) { kk[j] = -INFINITY; vv[j] = 0.f; }
        }
#pragma unroll
        for (int j = 0; j < kD; ++j) { sK[tid][j] = kk[j]; sV[tid][j] = vv[j]; }
    }
    __syncthreads();
    const int d = tid & 31, part = tid >> 5;   // 8 parts x 32 pixels
    float m = -INFINITY;
    for (int p = part * 32; p < part * 32 + 32; ++p) m = fmaxf(m, sK[p][d]);
    sRed[part][d] = m;
    __syncthreads();
    if (tid < kD) {
        float mm = sRed[0][tid];
        for (int i = 1; i < 8; ++i) mm = fmaxf(mm, sRed[i][tid]);
        sM[tid] = mm;
    }
    __syncthreads();
    m = sM[d];
    float z = 0.f;
    for (int p = part * 32; p < part * 32 + 32; ++p) {
        const float e = expf(sK[p][d] - m);   // exp(-inf) = 0 for the padded pixels
        sK[p][d] = e;
        z += e;
    }
    __syncthreads();
    sRed[part][d] = z;
    __syncthreads();
    float* dst = a.ctx_part + ((size_t)(b * a.heads + h) * a.splits + split) * kPartStride;
    if (tid < kD) {
        float zz = 0.f;
        for (int i = 0; i < 8; ++i) zz += sRed[i][tid];
        dst[kD * kD + tid] = sM[tid];
        dst[kD * kD + kD + tid] = zz;
    }
    // ctx[d][e]: thread -> (d = tid>>3, e0 = (tid&7)*4)
    const int dd = tid >> 3, e0 = (tid & 7) * 4;
    float c0 = 0.f, c1 = 0.f, c2 = 0.f, c3 = 0.f;
    for (int p = 0; p < kLaChunk; ++p) {
        const float kv = sK[p][dd];
        const f32x4 v = *reinterpret_cast<const f32x4*>(&sV[p][e0]);
        c0 = fmaf(kv, v[0], c0);
        c1 = fmaf(kv, v[1], c1);
        c2 = fmaf(kv, v[2], c2);
        c3 = fmaf(kv, v[3], c3);
    }
    *reinterpret_cast<f32x4*>(dst + dd * kD + e0) = f32x4{c0, c1, c2, c3};
}

// grid (ceil(n/256), B); 256 threads, one pixel per thread
template <typename T>
__global__ void __launch_bounds__(256) linattn_out_kernel(const LinAttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sCtx = reinterpret_cast<float*>(smem);          // [heads][d][e]
    float* sScale = sCtx + a.heads * kD * kD;              // [heads][d][splits] rescale factors / Z
    const int b = blockIdx.y, tid = threadIdx.x;
    const int hidden = a.heads * kD;
    // combine the partial contexts: factor_s[d] = exp(m_s[d] - M[d]) / sum_s Z_s[d] exp(m_s[d] - M[d])
    for (int hd = tid; hd < a.heads * kD; hd += 256) {
        const int h = hd / kD, d = hd % kD;
        const float* base = a.ctx_part + ((size_t)(b * a.heads + h) * a.splits) * kPartStride;
        float M = -INFINITY;
        for (int s = 0; s < a.splits; ++s) M = fmaxf(M, base[(size_t)s * kPartStride + kD * kD + d]);
        float Z = 0.f;
        for (int s = 0; s < a.splits; ++s) {
            const float f = expf(base[(size_t)s * kPartStride + kD * kD + d] - M);
            sScale[hd * a.splits + s] = f;
            Z += base[(size_t)s * kPartStride + kD * kD + kD + d] * f;
        }
        const float inv = 1.0f / Z;
        for (int s = 0; s < a.splits; ++s) sScale[hd * a.splits + s] *= inv;
    }
    __syncthreads();
    for (int i = tid; i < a.heads * kD * kD; i += 256) {
        const int h = i / (kD * kD), de = i % (kD * kD), d = de / kD;
        const float* base = a.ctx_part + ((size_t)(b * a.heads + h) * a.splits) * kPartStride;
        float acc = 0.f;
        for (int s = 0; s < a.splits; ++s) acc = fmaf(base[(size_t)s * kPartStride + de], sScale[(h * kD + d) * a.splits + s], acc);
        sCtx[i] = acc;
    }
    __syncthreads();
    const int p = blockIdx.x * 256 + tid;
    if (p >= a.n) return;
    const T* row = static_cast<const T*>(a.qkv) + ((size_t)b * a.n + p) * 3 * hidden;
    T* orow = static_cast<T*>(a.out) + ((size_t)b * a.n + p) * hidden;
    const float scale = 0.17677669529663687f;  // 32^-0.5
    for (int h = 0; h < a.heads; ++h) {
        float q[kD], o[kD];
        load32(row + h * kD, q);
        float m = q[0];
#pragma unroll
        for (int j = 1; j < kD; ++j) m = fmaxf(m, q[j]);
        float z = 0.f;
#pragma unroll
        for (int j = 0; j < kD; ++j) { q[j] = expf(q[j] - m); z += q[j]; }
        const float inv = scale / z;
#pragma unroll
        for (int e = 0; e < kD; ++e) o[e] = 0.f;
        const float* ctx = sCtx + h * kD * kD;
#pragma unroll
        for (int dq = 0; dq < kD; ++dq) {
            const float qq = q[dq] * inv;
#pragma unroll
            for (int e = 0; e < kD; e += 4) {
                const f32x4 c = *reinterpret_cast<const f32x4*>(ctx + dq * kD + e);
                o[e] = fmaf(c[0], qq, o[e]);
                o[e + 1] = fmaf(c[1], qq, o[e + 1]);
                o[e + 2] = fmaf(c[2], qq, o[e + 2]);
                o[e + 3] = fmaf(c[3], qq, o[e + 3]);
            }
        }
        store32(orow + h * kD, o);
    }
}

// softmax attention; grid (heads, B, query blocks); K/V of a 128-key chunk in LDS; online softmax.
// A workgroup serves 256 / parts queries; the `parts` threads of a query (parts = 1, 2 or 4, chosen so that small
// images still use every thread) walk interleaved keys and merge their (max, sum, accumulator) through LDS.
template <typename T>
__global__ void __launch_bounds__(256) attention_kernel(const AttnArgs a, int parts) {
    constexpr int KC = 128;   // keys per LDS chunk
    constexpr int ROW = kD + 4;
    __shared__ __attribute__((aligned(16))) float sKV[2 * KC * ROW];   // K rows, then V rows; reused for the merge
    float(*sK)[ROW] = reinterpret_cast<float(*)[ROW]>(sKV);
    float(*sV)[ROW] = reinterpret_cast<float(*)[ROW]>(sKV + KC * ROW);
    const int h = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int hidden = a.heads * kD;
    const int qpw = 256 / parts;                  // queries per workgroup
    const int ql = tid % qpw, part = tid / qpw;
    const int qi = blockIdx.z * qpw + ql;
    const bool active = qi < a.n;
    const T* base = static_cast<const T*>(a.qkv) + (size_t)b * a.n * 3 * hidden;
    float q[kD], acc[kD];
    const float scale = 0.17677669529663687f;
    if (active) {
        load32(base + (size_t)qi * 3 * hidden + h * kD, q);
#pragma unroll
        for (int j = 0; j < kD; ++j) q[j] *= scale;
    } else {
#pragma unroll
        for (int j = 0; j < kD; ++j) q[j] = 0.f;
    }
#pragma unroll
    for (int j = 0; j < kD; ++j) acc[j] = 0.f;
    float m = -INFINITY, l = 0.f;
    for (int k0 = 0; k0 < a.n; k0 += KC) {
        const int nk = min(KC, a.n - k0);
        __syncthreads();
        if (tid < nk) {
            float kk[kD], vv[kD];
            const T* row = base + (size_t)(k0 + tid) * 3 * hidden;
            load32(row + hidden + h * kD, kk);
            load32(row + 2 * hidden + h * kD, vv);
#pragma unroll
            for (int j = 0; j < kD; ++j) { sK[tid][j] = kk[j]; sV[tid][j] = vv[j]; }
        }
        __syncthreads();
        for (int j = part; j < nk; j += parts) {
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < kD; c += 4) {
                const f32x4 kv = *reinterpret_cast<const f32x4*>(&sK[j][c]);
                s = fmaf(q[c], kv[0], s);
                s = fmaf(q[c + 1], kv[1], s);
                s = fmaf(q[c + 2], kv[2], s);
                s = fmaf(q[c + 3], kv[3], s);
            }
            const float mn = fmaxf(m, s);
            const float alpha = expf(m - mn);
            const float pj = expf(s - mn);
            l = l * alpha + pj;
#pragma unroll
            for (int c = 0; c < kD; c += 4) {
                const f32x4 vv = *reinterpret_cast<const f32x4*>(&sV[j][c]);
                acc[c] = fmaf(pj, vv[0], acc[c] * alpha);
                acc[c + 1] = fmaf(pj, vv[1], acc[c + 1] * alpha);
                acc[c + 2] = fmaf(pj, vv[2], acc[c + 2] * alpha);
                acc[c + 3] = fmaf(pj, vv[3], acc[c + 3] * alpha);
            }
            m = mn;
        }
    }
    if (parts > 1) {
        // merge the partial softmaxes of a query: rows of 35 floats (max, sum, 32 accumulators), odd pitch
        constexpr int MR = kD + 3;
        __syncthreads();
        float* mine = sKV + (size_t)tid * MR;
        mine[0] = m;
        mine[1] = l;
#pragma unroll
        for (int j = 0; j < kD; ++j) mine[2 + j] = acc[j];
        __syncthreads();
        if (part != 0) return;
        float M = m;
        for (int p = 1; p < parts; ++p) M = fmaxf(M, sKV[(size_t)(p * qpw + ql) * MR]);
        const float f0 = m == -INFINITY ? 0.f : expf(m - M);
        l *= f0;
#pragma unroll
        for (int j = 0; j < kD; ++j) acc[j] *= f0;
        for (int p = 1; p < parts; ++p) {
            const float* o = sKV + (size_t)(p * qpw + ql) * MR;
            const float f = o[0] == -INFINITY ? 0.f : expf(o[0] - M);
            l = fmaf(o[1], f, l);
#pragma unroll
            for (int j = 0; j < kD; ++j) acc[j] = fmaf(o[2 + j], f, acc[j]);
        }
    }
    if (active) {
        const float inv = 1.0f / l;
#pragma unroll
        for (int j = 0; j < kD; ++j) acc[j] *= inv;
        store32(static_cast<T*>(a.out) + ((size_t)b * a.n + qi) * hidden + h * kD, acc);
    }
}

// Softmax attention on the matrix cores, for 64 <= n <= 256 keys with n % 32 == 0 (the bottleneck of 64x64 and 128x128
// inputs; smaller bottlenecks are not dense contractions and keep the VALU kernel above). grid (heads, B, n / 64);
// a wave owns 16 queries, a workgroup 64.
//   S^T = K Q^T per 16-key block: A = 16 keys x 32 channels, B = 32 channels x 16 queries, both 16-byte fragments read
//         straight from the qkv rows; the result has the query on the lane (lane & 15) and four keys per lane group in
//         its registers, so the soft-max over keys is a reduction over registers plus two cross-group shuffles;
//   O^T = V^T P^T: the exponentials, still in their accumulator registers, ARE the B operand (32 keys x 16 queries) of
//         the second product, no lane movement; only V is transposed, once per workgroup, through LDS.
// Two stacked 16-key tiles form one 32-deep bf16 step in the order (tile 0 regs 0..3, tile 1 regs 0..3) of a lane
// group, i.e. keys {4g..4g+3, 16+4g..16+4g+3}; the V^T fragment is read in the same order. In f32 mode each 16-key tile
// is one step of four exact-f32 instructions in natural key order.
template <typename T>
__global__ void __launch_bounds__(256) attention_mfma_kernel(const AttnArgs a) {
    constexpr int PER = ElemTraits<T>::kPer16B;      // elements per 16-byte fragment: 8 (bf16) / 4 (f32)
    constexpr int KCH = ElemTraits<T>::kChunk;       // channels per MFMA step: 32 / 16
    constexpr int QS = kD / KCH;                     // steps over the 32 head channels: 1 / 2
    constexpr int MAXT = 16;                         // 16-key tiles kept in registers (n <= 256)
    using V = decltype(Vec16<T>::v);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* sVt = reinterpret_cast<T*>(smem);             // V^T [32 channels][n + pad] of this (image, head)
    const int h = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lq = lane >> 4;
    const int n = a.n, hidden = a.heads * kD, pitch = n + 16 / (int)sizeof(T);   // +16 bytes: rows on different banks
    const T* base = static_cast<const T*>(a.qkv) + (size_t)b * n * 3 * hidden;
    const int q0 = (blockIdx.z * 4 + wave) * 16;

    // Q fragments of the wave's queries first (B operand: query on the lane, 16 bytes of its channels)
    Vec16<T> qf[QS];
#pragma unroll
    for (int ks = 0; ks < QS; ++ks)
        qf[ks].v = *reinterpret_cast<const V*>(base + (size_t)(q0 + l15) * 3 * hidden + h * kD + ks * KCH + lq * PER);
    // V^T into LDS: a thread takes 16 bytes of a key's channels and scatters them over the channel rows
    for (int u = tid; u < n * (kD / PER); u += 256) {
        const int key = u / (kD / PER), part = u % (kD / PER);
        Vec16<T> v;
        v.v = *reinterpret_cast<const V*>(base + (size_t)key * 3 * hidden + 2 * hidden + h * kD + part * PER);
#pragma unroll
        for (int j = 0; j < PER; ++j) sVt[(part * PER + j) * pitch + key] = v.v[j];
    }
    const int nt = n / 16;
    const float scale = 0.17677669529663687f;        // 32^-0.5, applied to the products (the reference scales q)
    f32x4 sreg[MAXT];
    float m = -INFINITY;
#pragma unroll
    for (int t = 0; t < MAXT; ++t) {
        if (t < nt) {
            f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < QS; ++ks) {
                Vec16<T> kf;   // A operand: key on the lane
                kf.v = *reinterpret_cast<const V*>(base + (size_t)(t * 16 + l15) * 3 * hidden + hidden + h * kD + ks * KCH + lq * PER);
                mma_step(kf, qf[ks], acc);
            }
            acc *= scale;
            sreg[t] = acc;
            m = fmaxf(fmaxf(m, fmaxf(acc[0], acc[1])), fmaxf(acc[2], acc[3]));
        }
    }
    m = fmaxf(m, __shfl_xor(m, 16, 64));
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float l = 0.f;
#pragma unroll
    for (int t = 0; t < MAXT; ++t) {
        if (t < nt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float e;
                if constexpr (sizeof(T) == 2) {
                    e = __expf(sreg[t][r] - m);
                    e = (float)(__bf16)e;             // the denominator sums what the second product multiplies
                } else {
                    e = expf(sreg[t][r] - m);
                }
                sreg[t][r] = e;
                l += e;
            }
        }
    }
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    __syncthreads();                                  // V^T complete
    f32x4 o[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    if constexpr (sizeof(T) == 2) {
#pragma unroll
        for (int u = 0; u < MAXT / 2; ++u) {
            if (2 * u < nt) {
                Vec16<T> pf;   // B operand: keys {4g..4g+3} of tile 2u, then of tile 2u+1
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    pf.v[r] = (__bf16)sreg[2 * u][r];
                    pf.v[4 + r] = (__bf16)sreg[2 * u + 1][r];
                }
#pragma unroll
                for (int et = 0; et < 2; ++et) {
                    const T* row = sVt + (et * 16 + l15) * pitch + 32 * u + 4 * lq;
                    const bf16x4 lo = *reinterpret_cast<const bf16x4*>(row);
                    const bf16x4 hi = *reinterpret_cast<const bf16x4*>(row + 16);
                    Vec16<T> vf;
                    vf.v = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    mma_step(vf, pf, o[et]);
                }
            }
        }
    } else {
#pragma unroll
        for (int t = 0; t < MAXT; ++t) {
            if (t < nt) {
                Vec16<T> pf;
                pf.v = sreg[t];
#pragma unroll
                for (int et = 0; et < 2; ++et) {
                    Vec16<T> vf;
                    vf.v = *reinterpret_cast<const V*>(sVt + (et * 16 + l15) * pitch + 16 * t + 4 * lq);
                    mma_step(vf, pf, o[et]);
                }
            }
        }
    }
    // O^T: query on the lane, channels 16 et + 4 g + r in the registers
    const float inv = 1.0f / l;
    T* orow = static_cast<T*>(a.out) + ((size_t)b * n + q0 + l15) * hidden + h * kD;
#pragma unroll
    for (int et = 0; et < 2; ++et) {
        if constexpr (sizeof(T) == 2) {
            bf16x4 ov;
#pragma unroll
            for (int r = 0; r < 4; ++r) ov[r] = (__bf16)(o[et][r] * inv);
            *reinterpret_cast<bf16x4*>(orow + et * 16 + lq * 4) = ov;
        } else {
            *reinterpret_cast<f32x4*>(orow + et * 16 + lq * 4) = o[et] * inv;
        }
    }
}

}  // namespace

int linear_attention_splits(int n) { return (n + kLaChunk - 1) / kLaChunk; }

int launch_linear_attention(LinAttnArgs& a, hipStream_t s) {
    DMN_REQUIRE(a.d == kD, DMN_ERR_INVALID, "linear attention: dim_head=%d (only 32)", a.d);
    a.splits = linear_attention_splits(a.n);
    const size_t lds = ((size_t)a.heads * kD * kD + (size_t)a.heads * kD * a.splits) * 4;
    DMN_REQUIRE(lds <= 64 * 1024, DMN_ERR_LIMIT, "linear attention: %d pixels exceed the supported maximum", a.n);
    dim3 g1(a.splits, a.heads, a.B), g2((a.n + 255) / 256, a.B);
    if (a.dtype == DMN_F32) {
        hipLaunchKernelGGL(linattn_ctx_kernel<float>, g1, dim3(256), 0, s, a);
        hipLaunchKernelGGL(linattn_out_kernel<float>, g2, dim3(256), lds, s, a);
    } else {
        hipLaunchKernelGGL(linattn_ctx_kernel<__bf16>, g1, dim3(256), 0, s, a);
        hipLaunchKernelGGL(linattn_out_kernel<__bf16>, g2, dim3(256), lds, s, a);
    }
    DMN_HIP_CHECK(hipGetLastError());
    return DMN_OK;
}

int launch_attention(AttnArgs& a, hipStream_t s) {
    DMN_REQUIRE(a.d == kD, DMN_ERR_INVALID, "attention: dim_head=%d (only 32)", a.d);
    if (!a.no_mfma && a.n >= 64 && a.n <= 256 && a.n % 64 == 0) {   // a dense contraction: matrix cores
        const size_t esz = a.dtype == DMN_F32 ? 4 : 2;
        const size_t lds = (size_t)kD * (a.n * esz + 16);
        dim3 grid(a.heads, a.B, a.n / 64);
        if (a.dtype == DMN_F32)
            hipLaunchKernelGGL(attention_mfma_kernel<float>, grid, dim3(256), lds, s, a);
        else
            hipLaunchKernelGGL(attention_mfma_kernel<__bf16>, grid, dim3(256), lds, s, a);
        DMN_HIP_CHECK(hipGetLastError());
        return DMN_OK;
    }
    const int parts = a.n <= 64 ? 4 : a.n <= 128 ? 2 : 1;
    const int qpw = 256 / parts;
    dim3 grid(a.heads, a.B, (a.n + qpw - 1) / qpw);
    if (a.dtype == DMN_F32)
        hipLaunchKernelGGL(attention_kernel<float>, grid, dim3(256), 0, s, a, parts);
    else
        hipLaunchKernelGGL(attention_kernel<__bf16>, grid, dim3(256), 0, s, a, parts);
    DMN_HIP_CHECK(hipGetLastError());
    return DMN_OK;
}
