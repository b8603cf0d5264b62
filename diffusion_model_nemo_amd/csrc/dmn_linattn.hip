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
#include "dmn_handoff.h"
#include "dmn_kernels.h"
#include "dmn_norm.h"

#include <type_traits>

namespace {

// optional phase timestamps (tuning aid): slot 0 wall clock at start, 7 cycle counter at start, 1..4 cycle counter at
// the end of each phase, 5 wall clock at the end
#define LA_STAMP(base, i)                                                                     \
    if ((base) && threadIdx.x == 0) {                                                         \
        long long* d_ = (base) + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8;            \
        d_[(i)] = (i) == 0 ? (long long)wall_clock64() : (long long)clock64();                 \
        if ((i) == 0) d_[7] = clock64();                                                      \
        if ((i) == 4) d_[5] = wall_clock64();                                                 \
    }

constexpr int kHeads = 4, kD = 32, kHidden = 128;
constexpr int kMergeInOutMaxSplits = 4;   // up to this many split partials per image are merged by the second launch itself
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

// Stages rows [p0, p0 + rows) of x[b] (C channels) into LDS, normalised with GroupNorm(1); rows >= n become zero.
// The first batch of loads is issued by `begin` before the normalisation coefficients exist, so its latency
// overlaps the statistics round trip; `finish` applies the coefficients and writes everything.
template <typename T, int C>
struct XStager {
    static constexpr int PER = ElemTraits<T>::kPer16B;
    static constexpr int PARTS = C / PER;          // 16-byte units per pixel row; divides 256
    static constexpr int SB = 8;
    using V = decltype(Vec16<T>::v);
    V r[SB];
    const T* x;
    int p0, rows, n, part;
    __device__ __forceinline__ void issue(int j0) {
#pragma unroll
        for (int j = 0; j < SB; ++j) {
            const int u = threadIdx.x + 256 * (j0 + j);
            const int p = p0 + u / PARTS;
            const bool ok = u < rows * PARTS && p < n;
            r[j] = *reinterpret_cast<const V*>(x + (size_t)(ok ? p : 0) * C + part * PER);
        }
    }
    __device__ __forceinline__ void begin(const void* xb, int b, int n_, int p0_, int rows_) {
        x = static_cast<const T*>(xb) + (size_t)b * n_ * C;
        n = n_;
        p0 = p0_;
        rows = rows_;
        part = threadIdx.x % PARTS;
        issue(0);
    }
    __device__ __forceinline__ void finish(char* sX, int row_bytes, const float* sCoef) {
        float ca[PER], cb[PER];
#pragma unroll
        for (int e = 0; e < PER; ++e) {
            ca[e] = sCoef[part * PER + e];
            cb[e] = sCoef[C + part * PER + e];
        }
        const int units = rows * PARTS;
        for (int j0 = 0; j0 * 256 < units; j0 += SB) {
#pragma unroll
            for (int j = 0; j < SB; ++j) {
                const int u = threadIdx.x + 256 * (j0 + j);
                const int row = u / PARTS;
                const bool ok = p0 + row < n;
                Vec16<T> v;
                v.v = r[j];
#pragma unroll
                for (int e = 0; e < PER; ++e) v.set(e, ok ? v.get(e) * ca[e] + cb[e] : 0.f);
                if (u < units) *reinterpret_cast<V*>(sX + (size_t)row * row_bytes + part * 16) = v.v;
            }
            if ((j0 + SB) * 256 < units) issue(j0 + SB);
        }
    }
};

// ------------------------------------------------------------------------------------------------------------
// The context of one head over a pixel slice staged in LDS, held by one wave: K and V projections, the per-channel
// maximum of k (sweep 1), then P = exp(k - max), Z = sum P and ctx += P^T V (sweep 2). Shared by the first pass and by
// the single-launch form for images that one workgroup holds.
// ------------------------------------------------------------------------------------------------------------
template <typename T, int C>
struct HeadContext {
    static constexpr int KCH = ElemTraits<T>::kChunk;
    static constexpr int SUB = Sub<T>::N;
    static constexpr int KS = C / KCH;
    static constexpr bool WREG = KS <= 4;          // projection weights of the head stay in registers
    static constexpr int WR = WREG ? KS : 1;
    using V = decltype(Vec16<T>::v);
    const T *wk, *wv;                              // fragment order [ct 0..7][ks][lane][PER]; head h owns ct 2h, 2h+1
    int h, lane;
    Vec16<T> wkr[2][WR], wvr[2][WR];
    float mx[2];
    f32x4 ctx[2][2];                               // [d tile][e tile], rows d = 16 * dt + 4 * (lane >> 4) + r, column e = lane & 15
    f32x4 zacc[2];                                 // Z[d] in the same lanes and registers as row d of the ctx tiles

    // the requests that depend on nothing: issue before the pixels are staged
    __device__ __forceinline__ void preload(const void* wk_, const void* wv_, int h_, int lane_) {
        wk = static_cast<const T*>(wk_);
        wv = static_cast<const T*>(wv_);
        h = h_;
        lane = lane_;
        if constexpr (WREG) {
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    wkr[ct][ks].v = load_frag<T>(wk, (2 * h + ct) * KS + ks, lane);
                    wvr[ct][ks].v = load_frag<T>(wv, (2 * h + ct) * KS + ks, lane);
                }
        }
    }
    __device__ __forceinline__ Vec16<T> kfrag(int ct, int ks) const {
        if constexpr (WREG) return wkr[ct][ks];
        Vec16<T> f;
        f.v = load_frag<T>(wk, (2 * h + ct) * KS + ks, lane);
        return f;
    }
    __device__ __forceinline__ Vec16<T> vfrag(int ct, int ks) const {
        if constexpr (WREG) return wvr[ct][ks];
        Vec16<T> f;
        f.v = load_frag<T>(wv, (2 * h + ct) * KS + ks, lane);
        return f;
    }

    // sX: `slice` rows (multiple of 32) of row_bytes, pixel p0 + row; rows beyond the image (n) hold zeros and are masked
    __device__ __forceinline__ void run(const char* sX, int row_bytes, int slice, int p0, int n, long long* dbg) {
        const int l15 = lane & 15, lq = lane >> 4;
        const int npt = slice / 16;                  // pixel tiles in the slice (even)
        // ---- sweep 1: per-channel maximum of k over the slice ------------------------------------------------
        // A slice that lies wholly inside the image (every size the network produces) takes the unmasked bodies.
        const bool full = p0 + slice <= n;
        mx[0] = mx[1] = -INFINITY;
        auto sweep1 = [&](auto masked_t) {
            constexpr bool MASKED = decltype(masked_t)::value;
#pragma unroll 2
            for (int pp = 0; pp < npt; pp += 2) {
                f32x4 kacc[2][2];
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) kacc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const Vec16<T> k0 = kfrag(0, ks), k1 = kfrag(1, ks);
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        Vec16<T> xf;
                        xf.v = *reinterpret_cast<const V*>(sX + (size_t)((pp + i) * 16 + l15) * row_bytes + lq * 16 + ks * 64);
                        mma_step(xf, k0, kacc[i][0]);     // D[px][cout]
                        mma_step(xf, k1, kacc[i][1]);
                    }
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const bool valid = !MASKED || p0 + (pp + i) * 16 + lq * 4 + r < n;
                            mx[ct] = valid ? fmaxf(mx[ct], kacc[i][ct][r]) : mx[ct];
                        }
            }
        };
        if (full) sweep1(std::false_type{});
        else sweep1(std::true_type{});
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            mx[ct] = fmaxf(mx[ct], __shfl_xor(mx[ct], 16, 64));
            mx[ct] = fmaxf(mx[ct], __shfl_xor(mx[ct], 32, 64));
        }

        LA_STAMP(dbg, 3)
        // ---- sweep 2: P = exp(k - max), Z = sum P, ctx += P^T V ------------------------------------------------
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) ctx[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        // The softmax denominators come from the matrix pipe too (P against a matrix of ones): row d of zacc holds Z[d]
        // in the same lanes and registers as row d of the ctx tiles.
        zacc[0] = zacc[1] = f32x4{0.f, 0.f, 0.f, 0.f};
        Vec16<T> ones;
#pragma unroll
        for (int e = 0; e < ElemTraits<T>::kPer16B; ++e) ones.set(e, 1.0f);
        // exp(k - max) as exp2(k * log2(e) - max * log2(e)): one fused multiply-add ahead of the transcendental
        constexpr float kLog2e = 1.4426950408889634f;
        const float mxl[2] = {mx[0] * kLog2e, mx[1] * kLog2e};
        auto sweep2 = [&](auto masked_t) {
            constexpr bool MASKED = decltype(masked_t)::value;
#pragma unroll 2
            for (int pp = 0; pp < npt; pp += 2) {        // 32 pixels = one bf16 k-step (two f32 k-steps)
                f32x4 kacc[2][2], vacc[2][2];             // [pixel tile of the pair][channel tile]
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) kacc[i][j] = vacc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const Vec16<T> k0 = kfrag(0, ks), k1 = kfrag(1, ks), v0 = vfrag(0, ks), v1 = vfrag(1, ks);
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        Vec16<T> xf;
                        xf.v = *reinterpret_cast<const V*>(sX + (size_t)((pp + i) * 16 + l15) * row_bytes + lq * 16 + ks * 64);
                        mma_step(xf, k0, kacc[i][0]);
                        mma_step(xf, k1, kacc[i][1]);
                        mma_step(xf, v0, vacc[i][0]);
                        mma_step(xf, v1, vacc[i][1]);
                    }
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const bool valid = !MASKED || p0 + (pp + i) * 16 + lq * 4 + r < n;
                            float e;
                            if constexpr (sizeof(T) == 4) e = expf(kacc[i][ct][r] - mx[ct]);
                            else e = __builtin_amdgcn_exp2f(__builtin_fmaf(kacc[i][ct][r], kLog2e, -mxl[ct]));
                            kacc[i][ct][r] = valid ? e : 0.f;
                        }
#pragma unroll
                for (int sub = 0; sub < SUB; ++sub)
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt) {
                        const Vec16<T> pa = acc_operand(kacc[0][dt], kacc[1][dt], sub, (T*)nullptr);   // A[d][k = px]
                        mma_step(pa, ones, zacc[dt]);     // Z[d] = sum_px P[px][d], in every column of row d
#pragma unroll
                        for (int et = 0; et < 2; ++et) {
                            const Vec16<T> vb = acc_operand(vacc[0][et], vacc[1][et], sub, (T*)nullptr);   // B[k = px][e]
                            mma_step(pa, vb, ctx[dt][et]);
                        }
                    }
            }
        };
        if (full) sweep2(std::false_type{});
        else sweep2(std::true_type{});
    }

    // The slice was the whole image: normalise, scale by d^-1/2 and write ctx^T in operand order (row e, position pos
    // holds ctx[d = perm32(pos)][e]); `row_elems` elements of T between the rows of `out` (this head's 32 rows).
    __device__ __forceinline__ void write_transposed(T* out, int row_elems) const {
        const int l15 = lane & 15, lq = lane >> 4;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int d = dt * 16 + lq * 4 + r;
                const float inv = 0.17677669529663687f / zacc[dt][r];
                const int pos = sizeof(T) == 4 ? d : ((d & 15) >> 2) * 8 + (d >> 4) * 4 + (d & 3);   // inverse of perm32
#pragma unroll
                for (int et = 0; et < 2; ++et) out[(et * 16 + l15) * row_elems + pos] = from_f32<T>(ctx[dt][et][r] * inv);
            }
    }
};

// ------------------------------------------------------------------------------------------------------------
// pass 1: grid (splits, B); slice = a.slice1 pixels (multiple of 32); wave h owns head h
// ------------------------------------------------------------------------------------------------------------
template <typename T, int C>
__global__ void __launch_bounds__(256) la_fused_ctx_kernel(const LinAttnFusedArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int row_bytes = C * (int)sizeof(T) + 32;
    const int slice = a.slice1;
    float* sCoef = reinterpret_cast<float*>(smem);            // 3*C + 8
    char* sX = smem + ((3 * C + 8) * 4 + 15) / 16 * 16;       // later reused for the merge factors
    const int split = blockIdx.x, b = blockIdx.y;
    const int lane = threadIdx.x & 63, h = threadIdx.x >> 6;
    const int l15 = lane & 15, lq = lane >> 4;
    const int p0 = split * slice;

    LA_STAMP(a.dbg, 0)
    HeadContext<T, C> hc;
    hc.preload(a.wk, a.wv, h, lane);
    XStager<T, C> st;
    st.begin(a.x, b, a.n, p0, slice);
    norm_coefs_to_lds(a.norm, b, C, sCoef, sCoef + 3 * C);
    LA_STAMP(a.dbg, 1)
    st.finish(sX, row_bytes, sCoef);
    __syncthreads();
    LA_STAMP(a.dbg, 2)
    hc.run(sX, row_bytes, slice, p0, a.n, a.dbg);
    if (a.splits == 1) {
        hc.write_transposed(static_cast<T*>(a.ctxT) + ((size_t)b * kHeads + h) * kD * kD, kD);
    } else {
        float* dst = a.ctx_part + (((size_t)b * kHeads + h) * a.splits + split) * kPartStride;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int et = 0; et < 2; ++et)
#pragma unroll
                for (int r = 0; r < 4; ++r) dst[(dt * 16 + lq * 4 + r) * kD + et * 16 + l15] = hc.ctx[dt][et][r];
        if (lq == 0) {
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) dst[kD * kD + ct * 16 + l15] = hc.mx[ct];
        }
        if (l15 == 0) {
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int r = 0; r < 4; ++r) dst[kD * kD + kD + dt * 16 + lq * 4 + r] = hc.zacc[dt][r];
        }
    }
    LA_STAMP(a.dbg, 4)
}

// ------------------------------------------------------------------------------------------------------------
// merge of the split partials (only launched when the second pass does not merge them itself): grid (B, heads), one
// workgroup per image and head (an image's partials are 17 KB per split: four workgroups halve the launch's latency)
//   factor_s[d] = exp(m_s[d] - M[d]) * 32^-1/2 / sum_s Z_s[d] exp(m_s[d] - M[d]);  ctx^T = sum_s factor_s * ctx_s
// ------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) la_merge_kernel(const LinAttnFusedArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sFac = reinterpret_cast<float*>(smem);             // [128][splits] (this head's 32 rows are used)
    const int b = blockIdx.x, head = blockIdx.y, S = a.splits;
    const float* part_b = a.ctx_part + (size_t)b * kHeads * S * kPartStride;
    // Up to kBatch splits (the full-resolution level has 8) are fetched together instead of one round trip per split and
    // pass; the arithmetic keeps the split order, so the bits are those of the plain loops below.
    constexpr int kBatch = 8;
    for (int hd = head * kD + threadIdx.x; hd < (head + 1) * kD; hd += 256) {
        const float* base = part_b + (size_t)(hd >> 5) * S * kPartStride + kD * kD + (hd & 31);
        float M = -INFINITY, Z = 0.f;
        if (S <= kBatch) {
            float mv[kBatch], zv[kBatch];
#pragma unroll
            for (int s = 0; s < kBatch; ++s) {
                const size_t o = (size_t)(s < S ? s : S - 1) * kPartStride;
                mv[s] = base[o];
                zv[s] = base[o + kD];
            }
#pragma unroll
            for (int s = 0; s < kBatch; ++s)
                if (s < S) M = fmaxf(M, mv[s]);
#pragma unroll
            for (int s = 0; s < kBatch; ++s)
                if (s < S) {
                    const float f = expf(mv[s] - M);
                    sFac[hd * S + s] = f;
                    Z += zv[s] * f;
                }
        } else {
            for (int s = 0; s < S; ++s) M = fmaxf(M, base[(size_t)s * kPartStride]);
            for (int s = 0; s < S; ++s) {
                const float f = expf(base[(size_t)s * kPartStride] - M);
                sFac[hd * S + s] = f;
                Z += base[(size_t)s * kPartStride + kD] * f;
            }
        }
        const float inv = 0.17677669529663687f / Z;
        for (int s = 0; s < S; ++s) sFac[hd * S + s] *= inv;
    }
    // the context rows this thread merges, requested before the barrier (they do not depend on the factors)
    const int i0 = head * (kD * kD / 4) + threadIdx.x;   // kD * kD / 4 == 256: one unit per thread
    f32x4 pv[kBatch];
    if (S <= kBatch) {
        const int hh = i0 >> 8, d = (i0 >> 3) & 31, e4 = (i0 & 7) * 4;
        const float* base = part_b + (size_t)hh * S * kPartStride + d * kD + e4;
#pragma unroll
        for (int s = 0; s < kBatch; ++s) pv[s] = *reinterpret_cast<const f32x4*>(base + (size_t)(s < S ? s : S - 1) * kPartStride);
    }
    __syncthreads();
    // ctx^T in operand order: row e, position pos holds ctx[d = perm32(pos)][e]
    T* out = static_cast<T*>(a.ctxT) + (size_t)b * kHeads * kD * kD;
    for (int i = head * (kD * kD / 4) + threadIdx.x; i < (head + 1) * (kD * kD / 4); i += 256) {
        const int hh = i >> 8, d = (i >> 3) & 31, e4 = (i & 7) * 4;
        const float* base = part_b + (size_t)hh * S * kPartStride + d * kD + e4;
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
        if (S <= kBatch && i == i0) {
#pragma unroll
            for (int s = 0; s < kBatch; ++s)
                if (s < S) {
                    const float f = sFac[(hh * kD + d) * S + s];
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[r] = fmaf(pv[s][r], f, acc[r]);
                }
        } else
        for (int s = 0; s < S; ++s) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(base + (size_t)s * kPartStride);
            const float f = sFac[(hh * kD + d) * S + s];
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] = fmaf(v[r], f, acc[r]);
        }
        const int pos = sizeof(T) == 4 ? d : ((d & 15) >> 2) * 8 + (d >> 4) * 4 + (d & 3);   // inverse of perm32
#pragma unroll
        for (int r = 0; r < 4; ++r) out[(hh * kD + e4 + r) * kD + pos] = from_f32<T>(acc[r]);
    }
}

// ------------------------------------------------------------------------------------------------------------
// pass 2: grid (slices, B); slice = 64 * tpw pixels, processed in groups of 64 (four pixel tiles).
//   phase A  wave h = head h:  Q_h = Wq_h x  ->  softmax over d  ->  out_h = ctx_h^T q_h  ->  LDS (k-major rows)
//   phase B  waves split the output-channel tiles:  y = Wout out + bias  ->  store + statistics
// ------------------------------------------------------------------------------------------------------------
// TAIL: the launch also closes the attention -- out = GroupNorm(1)(y) + x (parts/mha.py:41-42, utils.py:73-74). y stays
//   in the workgroup's LDS (in the rows of the staged pixels it was computed from) while the image's workgroups exchange
//   their partial sums inside the launch (dmn_handoff.h); neither y nor a closing pass over it exists.
// ONE: the workgroup's slice is the whole image (grid (1, B)) and no first pass ran: after staging, wave h computes the
//   context of head h itself (HeadContext, the arithmetic of the first pass) and leaves ctx^T in LDS where the merged
//   context would have been copied to. The pixels are staged once and the whole attention is one launch.
// Workgroups that may wait for each other (TAIL with several slices per image, i.e. not ONE) keep to half a SIMD's
// registers: two of them per CU must be able to be resident together (dmn_handoff.h).
template <typename T, int C, bool TAIL, bool ONE>
__global__ void __launch_bounds__(256, (TAIL && !ONE) ? 2 : 1) la_fused_out_kernel(const LinAttnFusedArgs a) {
    constexpr int PER = ElemTraits<T>::kPer16B;
    constexpr int KCH = ElemTraits<T>::kChunk;
    constexpr int SUB = Sub<T>::N;
    constexpr int KS = C / KCH;
    constexpr int KO = kHidden / KCH;              // k-steps of to_out
    constexpr int YT = C / 16;
    constexpr int YPW = YT >= 4 ? YT / 4 : 1;      // output-channel tiles per wave
    constexpr int PTW = YT >= 4 ? 4 : YT;          // pixel tiles per wave in phase B (waves share a tile row when YT < 4)
    constexpr bool QREG = KS <= 2;                 // Wq of the head stays in registers across pixel groups
    constexpr int QR = QREG ? KS : 1;
    constexpr int CTX_ROW = kD * (int)sizeof(T) + 32;
    constexpr int O_ROW = kHidden * (int)sizeof(T) + 32;
    using V = decltype(Vec16<T>::v);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int row_bytes = C * (int)sizeof(T) + 32;
    const int slice = 64 * a.tpw;
    float* sCoef = reinterpret_cast<float*>(smem);                         // 3*C + 8
    char* sCtx = smem + ((3 * C + 8) * 4 + 15) / 16 * 16;                  // [4][32 rows e][CTX_ROW]
    char* sO = sCtx + kHeads * kD * CTX_ROW;                               // [64 px][O_ROW]
    char* sX = sO + 64 * O_ROW;
    float* sRed = reinterpret_cast<float*>(sX + (size_t)slice * row_bytes);   // 16 floats: partial moments of the four waves
    const int b = a.b_first + blockIdx.y, tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lq = lane >> 4;
    const int p0 = blockIdx.x * slice;
    long long* dbg2 = a.dbg ? a.dbg + (size_t)65536 * 8 : nullptr;

    LA_STAMP(dbg2, 0)
    // ---- requests that depend on nothing: weights, merged context, first pixels ------------------------------
    const T* wq = static_cast<const T*>(a.wq);     // [ct 0..7][ks][lane][PER]; head h owns ct 2h, 2h+1
    const T* wo = static_cast<const T*>(a.wo);     // [yt][ko][lane][PER]
    Vec16<T> wqr[2][QR];
    if constexpr (QREG) {
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) wqr[ct][ks].v = load_frag<T>(wq, (2 * wave + ct) * KS + ks, lane);
    }
    const int yt0 = YT >= 4 ? wave : wave % YT;    // this wave's tiles: yt0 + 4*i
    const int ptb = YT >= 4 ? 0 : (wave / YT) * PTW;
    Vec16<T> wor[YPW][KO];
    f32x4 bias[YPW];
    auto load_out_weights = [&]() {
#pragma unroll
        for (int i = 0; i < YPW; ++i)
#pragma unroll
            for (int ko = 0; ko < KO; ++ko) wor[i][ko].v = load_frag<T>(wo, (yt0 + 4 * i) * KO + ko, lane);
#pragma unroll
        for (int i = 0; i < YPW; ++i) bias[i] = *reinterpret_cast<const f32x4*>(a.bias + (yt0 + 4 * i) * 16 + lq * 4);
    };
    HeadContext<T, C> hc;                          // ONE only
    if constexpr (ONE) hc.preload(a.wk, a.wv, wave, lane);   // to_out's weights are requested after the context: fewer live registers
    else load_out_weights();
    constexpr int CU = kHeads * kD * kD / PER / 256;   // 16-byte units of ctx^T per thread
    V cr[CU];
    // merge_splits > 0: the first pass left that many split partials per image and no merge launch ran; this workgroup
    // merges them itself (the arithmetic of la_merge_kernel, every workgroup of the image redundantly; worth it for a few
    // splits, where the partials of an image are a few tens of KB from the L2)
    const int S = ONE ? 0 : a.merge_splits;
    const float* part_b = a.ctx_part + (size_t)b * kHeads * (S > 0 ? S : 1) * kPartStride;
    float* sFac = reinterpret_cast<float*>(sO);    // [128][S] (sO is first written in phase A)
    if (!ONE && S == 0) {
        const T* cT = static_cast<const T*>(a.ctxT) + (size_t)b * kHeads * kD * kD;
#pragma unroll
        for (int j = 0; j < CU; ++j) cr[j] = *reinterpret_cast<const V*>(cT + (size_t)(tid + 256 * j) * PER);
    }
    XStager<T, C> st;
    st.begin(a.x, b, a.n, p0, slice);
    // With up to two splits the partial contexts this thread will merge are requested here, ahead of the coefficient
    // round trip and the staging arithmetic, instead of one round trip per split behind them (same values, same order).
    constexpr int kEarlySplits = 2;
    const bool early = S > 0 && S <= kEarlySplits;
    f32x4 pv[4][kEarlySplits];
    if (early) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float* base = part_b + (size_t)j * S * kPartStride + ((tid >> 3) & 31) * kD + (tid & 7) * 4;
#pragma unroll
            for (int s = 0; s < kEarlySplits; ++s) pv[j][s] = *reinterpret_cast<const f32x4*>(base + (size_t)(s < S ? s : S - 1) * kPartStride);
        }
    }
    if (S > 0 && tid < kHidden) {
        const int hd = tid;
        const float* base = part_b + (size_t)(hd >> 5) * S * kPartStride + kD * kD + (hd & 31);
        // the maxima and sums of all splits in flight together (S <= kMergeInOutMaxSplits); arithmetic in split order
        float mv[kMergeInOutMaxSplits], zv[kMergeInOutMaxSplits];
#pragma unroll
        for (int s = 0; s < kMergeInOutMaxSplits; ++s) {
            const size_t o = (size_t)(s < S ? s : S - 1) * kPartStride;
            mv[s] = base[o];
            zv[s] = base[o + kD];
        }
        float M = -INFINITY;
#pragma unroll
        for (int s = 0; s < kMergeInOutMaxSplits; ++s)
            if (s < S) M = fmaxf(M, mv[s]);
        float Z = 0.f;
#pragma unroll
        for (int s = 0; s < kMergeInOutMaxSplits; ++s)
            if (s < S) {
                const float f = expf(mv[s] - M);
                sFac[hd * S + s] = f;
                Z += zv[s] * f;
            }
        const float inv = 0.17677669529663687f / Z;
        for (int s = 0; s < S; ++s) sFac[hd * S + s] *= inv;
    }
    norm_coefs_to_lds(a.norm, b, C, sCoef, sCoef + 3 * C);   // its barriers also publish sFac
    LA_STAMP(dbg2, 1)
    st.finish(sX, row_bytes, sCoef);
    if constexpr (ONE) {
        __syncthreads();
        hc.run(sX, row_bytes, slice, 0, a.n, nullptr);
        load_out_weights();
        hc.write_transposed(reinterpret_cast<T*>(sCtx + (size_t)wave * kD * CTX_ROW), CTX_ROW / (int)sizeof(T));
    } else if (S == 0) {
#pragma unroll
        for (int j = 0; j < CU; ++j) {
            const int u = tid + 256 * j;               // unit -> row (h*32 + e), 16-byte column
            constexpr int UPR = kD / PER;
            *reinterpret_cast<V*>(sCtx + (size_t)(u / UPR) * CTX_ROW + (u % UPR) * 16) = cr[j];
        }
    } else {
        // ctx^T in operand order: row e, position pos holds ctx[d = perm32(pos)][e]
        static_assert(kHeads * kD * kD / 4 == 4 * 256, "four units per thread");
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = tid + 256 * j;
            const int hh = i >> 8, d = (i >> 3) & 31, e4 = (i & 7) * 4;
            const float* base = part_b + (size_t)hh * S * kPartStride + d * kD + e4;
            f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
            if (early) {
#pragma unroll
                for (int s = 0; s < kEarlySplits; ++s)
                    if (s < S) {
                        const float f = sFac[(hh * kD + d) * S + s];
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc[r] = fmaf(pv[j][s][r], f, acc[r]);
                    }
            } else
            for (int s = 0; s < S; ++s) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(base + (size_t)s * kPartStride);
                const float f = sFac[(hh * kD + d) * S + s];
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[r] = fmaf(v[r], f, acc[r]);
            }
            const int pos = sizeof(T) == 4 ? d : ((d & 15) >> 2) * 8 + (d >> 4) * 4 + (d & 3);   // inverse of perm32
#pragma unroll
            for (int r = 0; r < 4; ++r)
                *reinterpret_cast<T*>(sCtx + (size_t)(hh * kD + e4 + r) * CTX_ROW + pos * sizeof(T)) = from_f32<T>(acc[r]);
        }
    }
    __syncthreads();
    LA_STAMP(dbg2, 2)

    Vec16<T> cf[2][SUB];                           // A fragments of ctx_h^T
#pragma unroll
    for (int et = 0; et < 2; ++et)
#pragma unroll
        for (int sub = 0; sub < SUB; ++sub)
            cf[et][sub].v = *reinterpret_cast<const V*>(sCtx + (size_t)(wave * kD + et * 16 + l15) * CTX_ROW + sub * 64 + lq * 16);

    Vec16<T> ones;
#pragma unroll
    for (int e = 0; e < ElemTraits<T>::kPer16B; ++e) ones.set(e, 1.0f);
    // GroupNorm(1) statistics of y as partial moments (dmn_norm.h), one per wave. The values leave the registers as they
    // are produced, so the wave sums deviations from ONE shift, the first value its first lane produces (the wave's
    // lowest pixel: inside the image whenever any of the wave's pixels is; any value of the distribution will do as a
    // shift: what matters is that the squares summed are those of differences of the order of the spread, not of the
    // mean). With a common shift the lanes' sums simply add, and the shift is taken out once at the end.
    float ssum = 0.f, ssq = 0.f, scnt = 0.f, sshift = 0.f;
    T* yout = static_cast<T*>(a.y) + (size_t)b * a.n * C;
    for (int g = 0; g < a.tpw; ++g) {
        // ---- phase A: head `wave` on the four pixel tiles of the group ------------------------------------------
        f32x4 q[4][2];
#pragma unroll
        for (int pt = 0; pt < 4; ++pt) q[pt][0] = q[pt][1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            Vec16<T> w0, w1;
            if constexpr (QREG) {
                w0 = wqr[0][ks];
                w1 = wqr[1][ks];
            } else {
                w0.v = load_frag<T>(wq, (2 * wave) * KS + ks, lane);
                w1.v = load_frag<T>(wq, (2 * wave + 1) * KS + ks, lane);
            }
#pragma unroll
            for (int pt = 0; pt < 4; ++pt) {
                Vec16<T> xf;
                xf.v = *reinterpret_cast<const V*>(sX + (size_t)(g * 64 + pt * 16 + l15) * row_bytes + lq * 16 + ks * 64);
                mma_step(w0, xf, q[pt][0]);       // D[cout][px]
                mma_step(w1, xf, q[pt][1]);
            }
        }
#pragma unroll
        for (int pt = 0; pt < 4; ++pt) {
            // softmax over the 32 channels of the head: rows of the two tiles, spread over the 4 lane quarters
            float m = q[pt][0][0];
#pragma unroll
            for (int r = 0; r < 4; ++r) m = fmaxf(m, fmaxf(q[pt][0][r], q[pt][1][r]));
            m = fmaxf(m, __shfl_xor(m, 16, 64));
            m = fmaxf(m, __shfl_xor(m, 32, 64));
            f32x4 o[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
            if constexpr (sizeof(T) == 4) {
                float s = 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    q[pt][0][r] = expf(q[pt][0][r] - m);
                    q[pt][1][r] = expf(q[pt][1][r] - m);
                    s += q[pt][0][r] + q[pt][1][r];
                }
                s += __shfl_xor(s, 16, 64);
                s += __shfl_xor(s, 32, 64);
                const float inv = 1.0f / s;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    q[pt][0][r] *= inv;
                    q[pt][1][r] *= inv;
                }
                // out[e][px] = sum_d ctx[d][e] q[d][px]
#pragma unroll
                for (int sub = 0; sub < SUB; ++sub) {
                    const Vec16<T> qb = acc_operand(q[pt][0], q[pt][1], sub, (T*)nullptr);
                    mma_step(cf[0][sub], qb, o[0]);
                    mma_step(cf[1][sub], qb, o[1]);
                }
            } else {
                // bf16: exp2 with the log2(e) factor folded into one fma; the softmax denominator is a third MFMA of
                // the same operand against a matrix of ones (the sum over d of exactly the bf16 values the
                // numerator uses, delivered to every lane of the pixel: no cross-lane traffic), applied after.
                constexpr float kLog2e = 1.4426950408889634f;
                const float ml = m * kLog2e;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    q[pt][0][r] = __builtin_amdgcn_exp2f(__builtin_fmaf(q[pt][0][r], kLog2e, -ml));
                    q[pt][1][r] = __builtin_amdgcn_exp2f(__builtin_fmaf(q[pt][1][r], kLog2e, -ml));
                }
                const Vec16<T> qb = acc_operand(q[pt][0], q[pt][1], 0, (T*)nullptr);
                f32x4 den = f32x4{0.f, 0.f, 0.f, 0.f};
                mma_step(cf[0][0], qb, o[0]);
                mma_step(cf[1][0], qb, o[1]);
                mma_step(ones, qb, den);
                const float inv = __builtin_amdgcn_rcpf(den[0]);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    o[0][r] *= inv;
                    o[1][r] *= inv;
                }
            }
            // lane: pixel l15, channels e = et*16 + 4*lq .. +3 -> k-major row of the pixel
#pragma unroll
            for (int et = 0; et < 2; ++et) {
                char* dstp = sO + (size_t)(pt * 16 + l15) * O_ROW + (wave * kD + et * 16 + lq * 4) * sizeof(T);
                if constexpr (sizeof(T) == 4) {
                    *reinterpret_cast<f32x4*>(dstp) = o[et];
                } else {
                    bf16x4 ov;
#pragma unroll
                    for (int r = 0; r < 4; ++r) ov[r] = (__bf16)o[et][r];
                    *reinterpret_cast<bf16x4*>(dstp) = ov;
                }
            }
        }
        __syncthreads();
        // ---- phase B: y[c][px] = Wout out + bias for this wave's channel tiles ------------------------------------
        f32x4 y[YPW][PTW];
#pragma unroll
        for (int i = 0; i < YPW; ++i)
#pragma unroll
            for (int pt = 0; pt < PTW; ++pt) y[i][pt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ko = 0; ko < KO; ++ko) {
#pragma unroll
            for (int pt = 0; pt < PTW; ++pt) {
                Vec16<T> of;
                of.v = *reinterpret_cast<const V*>(sO + (size_t)((ptb + pt) * 16 + l15) * O_ROW + ko * 64 + lq * 16);
#pragma unroll
                for (int i = 0; i < YPW; ++i) mma_step(wor[i][ko], of, y[i][pt]);
            }
        }
        if (g == 0) sshift = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, y[0][0][0] + bias[0][0])));
#pragma unroll
        for (int pt = 0; pt < PTW; ++pt) {
            const int px = p0 + g * 64 + (ptb + pt) * 16 + l15;
            if (px < a.n) {
#pragma unroll
                for (int i = 0; i < YPW; ++i) {
                    const int c = (yt0 + 4 * i) * 16 + lq * 4;
                    float v[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = y[i][pt][r] + bias[i][r];
                    scnt += 4.f;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float d = v[r] - sshift;
                        ssum += d;
                        ssq += d * d;
                    }
                    // TAIL: y replaces the staged pixel it came from (every wave is done with the group's rows of sX
                    // since the barrier between the phases)
                    char* yl = sX + (size_t)(g * 64 + (ptb + pt) * 16 + l15) * row_bytes + c * sizeof(T);
                    if constexpr (sizeof(T) == 4) {
                        const f32x4 ov = f32x4{v[0], v[1], v[2], v[3]};
                        if constexpr (TAIL) *reinterpret_cast<f32x4*>(yl) = ov;
                        else store_vec(reinterpret_cast<f32x4*>(yout + (size_t)px * C + c), ov);
                    } else {
                        bf16x4 ov;
#pragma unroll
                        for (int r = 0; r < 4; ++r) ov[r] = (__bf16)v[r];
                        if constexpr (TAIL) *reinterpret_cast<bf16x4*>(yl) = ov;
                        else store_vec(reinterpret_cast<bf16x4*>(yout + (size_t)px * C + c), ov);
                    }
                }
            }
        }
        if (g + 1 < a.tpw) __syncthreads();       // sO is rewritten by the next group
    }
    LA_STAMP(dbg2, 3)
    // one partial per WAVE (slot 4 * workgroup + wave); the consumer combines them
    float pws, pwq, pwn;   // this wave's partial moments: three independent cross-lane sums, then the shift taken out
    {
        const float sd = wave_sum(ssum), qd = wave_sum(ssq);
        pwn = wave_sum(scnt);
        moments_from_shifted(sd, qd, pwn, sshift, pws, pwq);
    }
    if constexpr (!TAIL) {
        if (lane == 0) {
            f32x4* dst = reinterpret_cast<f32x4*>(a.y_stats) + ((size_t)b * gridDim.x + blockIdx.x) * 4 + wave;
            *dst = f32x4{pws, pwq, pwn, 0.f};
        }
    } else {
        // ---- closing GroupNorm(1) + residual ------------------------------------------------------------------------
        constexpr int PARTS = C / PER;                 // 16-byte units per pixel
        constexpr int XB = 4;                          // residual units requested ahead of the wait
        const unsigned epoch = *a.tail.epoch;
        const int slots = 4 * gridDim.x;               // one partial per wave of every workgroup of the image
        float* sMR = sCoef + 3 * C;
        const T* xres = static_cast<const T*>(a.x) + (size_t)b * a.n * C;
        T* outp = static_cast<T*>(a.out) + (size_t)b * a.n * C;
        const int rows = min(slice, a.n - p0);         // pixels of this workgroup inside the image
        const int units = rows * PARTS;
        V xr[XB];
        auto x_issue = [&](int j0) {
#pragma unroll
            for (int j = 0; j < XB; ++j) {
                const int u = tid + 256 * (j0 + j);
                xr[j] = *reinterpret_cast<const V*>(xres + (size_t)(p0 + (u < units ? u / PARTS : 0)) * C + (u % PARTS) * PER);
            }
        };
        x_issue(0);
        float pgam = 0.f, pbet = 0.f;
        if (tid < C) {
            pgam = a.tail_gamma[tid];
            pbet = a.tail_beta[tid];
        }
        if (slots > 4) {   // several workgroups per image: exchange
            if (lane == 0) handoff_publish(a.tail, epoch, b, 1, slots, 0, blockIdx.x * 4 + wave, pws, pwq, pwn);
            handoff_group_stats(a.tail, epoch, b, 1, slots, 0, 1, a.tail_inv_count, sMR);
        } else {           // the arithmetic of the exchange on this workgroup's own four partials
            if (lane == 0) *reinterpret_cast<f32x4*>(sRed + 4 * wave) = f32x4{pws, pwq, pwn, 0.f};
            __syncthreads();
            combine_groups(1, 4, a.tail_inv_count, sMR, [&](int, int i) { return *reinterpret_cast<const f32x4*>(sRed + 4 * i); });
        }
        __syncthreads();
        for (int c = tid; c < C; c += 256) {           // C <= 256: one channel per thread
            const float aa = sMR[1] * pgam;
            sCoef[c] = aa;
            sCoef[C + c] = pbet - sMR[0] * aa;
        }
        __syncthreads();
        const int part = tid % PARTS;                  // 256 % PARTS == 0: a thread keeps its channel offset
        float ca[PER], cb[PER];
#pragma unroll
        for (int e = 0; e < PER; ++e) {
            ca[e] = sCoef[part * PER + e];
            cb[e] = sCoef[C + part * PER + e];
        }
        for (int j0 = 0; j0 * 256 < units; j0 += XB) {
            if (j0) x_issue(j0);
#pragma unroll
            for (int j = 0; j < XB; ++j) {
                const int u = tid + 256 * (j0 + j);
                if (u < units) {
                    const int row = u / PARTS;
                    Vec16<T> v, r;
                    v.v = *reinterpret_cast<const V*>(sX + (size_t)row * row_bytes + part * 16);
                    r.v = xr[j];
#pragma unroll
                    for (int e = 0; e < PER; ++e) {
                        float y;
                        if constexpr (sizeof(T) == 2) y = __builtin_fmaf(v.get(e), ca[e], cb[e]) + 0.f;
                        else {
                            y = v.get(e) * ca[e] + cb[e];
                            y += 0.f;
                        }
                        v.set(e, y + r.get(e));
                    }
                    store_vec(reinterpret_cast<V*>(outp + (size_t)(p0 + row) * C + part * PER), v.v);
                }
            }
        }
    }
    LA_STAMP(dbg2, 4)
}

template <typename T, int C>
int launch_pair(const LinAttnFusedArgs& a, hipStream_t s) {
    const size_t row = (size_t)C * sizeof(T) + 32;
    const size_t coef = ((3 * C + 8) * 4 + 15) / 16 * 16;
    const size_t lds1 = coef + (size_t)a.slice1 * row + 64;
    const size_t ldsm = (size_t)kHidden * a.splits * 4;
    DMN_REQUIRE(ldsm <= 64 * 1024, DMN_ERR_LIMIT, "linear attention: %d splits", a.splits);
    const int slice = 64 * a.tpw;
    const size_t lds2 = coef + (size_t)kHeads * kD * (kD * sizeof(T) + 32) + 64 * ((size_t)kHidden * sizeof(T) + 32) +
                        slice * row + 64;
    DMN_REQUIRE(lds1 <= 160 * 1024 && lds2 <= 160 * 1024, DMN_ERR_LIMIT, "linear attention: LDS request %zu / %zu too large",
                lds1, lds2);
    static bool attr = false;
    if (!attr) {
        DMN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&la_fused_ctx_kernel<T, C>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        DMN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&la_fused_out_kernel<T, C, false, false>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        DMN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&la_fused_out_kernel<T, C, true, false>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        DMN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&la_fused_out_kernel<T, C, false, true>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        DMN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&la_fused_out_kernel<T, C, true, true>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr = true;
    }
    // an image that one workgroup holds in both passes: one launch for the whole attention
    if (a.splits == 1 && a.slices == 1 && !a.separate_ctx) {
        LinAttnFusedArgs c = a;
        c.merge_splits = 0;
        c.b_first = 0;
        c.b_count = a.B;
        if (a.tail.gran) hipLaunchKernelGGL((la_fused_out_kernel<T, C, true, true>), dim3(1, a.B), dim3(256), lds2, s, c);
        else hipLaunchKernelGGL((la_fused_out_kernel<T, C, false, true>), dim3(1, a.B), dim3(256), lds2, s, c);
        DMN_HIP_CHECK(hipGetLastError());
        return DMN_OK;
    }
    hipLaunchKernelGGL((la_fused_ctx_kernel<T, C>), dim3(a.splits, a.B), dim3(256), lds1, s, a);
    DMN_HIP_CHECK(hipGetLastError());
    // a few splits: the second launch merges them itself (no merge launch); more: one merge per image
    LinAttnFusedArgs c = a;
    c.merge_splits = (a.splits > 1 && a.splits <= kMergeInOutMaxSplits && !a.separate_merge) ? a.splits : 0;
    if (a.splits > 1 && !c.merge_splits) {
        hipLaunchKernelGGL(la_merge_kernel<T>, dim3(a.B, kHeads), dim3(256), ldsm, s, a);
        DMN_HIP_CHECK(hipGetLastError());
    }
    // a second launch whose workgroups wait for each other stays small enough to be resident as a whole (dmn_handoff.h)
    int per_launch = a.B;
    if (a.tail.gran && a.slices > 1) {
        DMN_REQUIRE(lds2 <= (size_t)kHandoffMaxLdsBytes && a.slices <= kHandoffMaxWgsPerImage, DMN_ERR_LIMIT,
                    "fused linear attention: closing GroupNorm outside the in-launch exchange's limits");
        per_launch = kHandoffMaxWgsPerLaunch / a.slices;
    }
    for (int b0 = 0; b0 < a.B; b0 += per_launch) {
        c.b_first = b0;
        c.b_count = a.B - b0 < per_launch ? a.B - b0 : per_launch;
        if (a.tail.gran) hipLaunchKernelGGL((la_fused_out_kernel<T, C, true, false>), dim3(a.slices, c.b_count), dim3(256), lds2, s, c);
        else hipLaunchKernelGGL((la_fused_out_kernel<T, C, false, false>), dim3(a.slices, c.b_count), dim3(256), lds2, s, c);
        DMN_HIP_CHECK(hipGetLastError());
    }
    return DMN_OK;
}

template <typename T>
int launch_both(LinAttnFusedArgs& a, hipStream_t s) {
    switch (a.C) {
        case 32: return launch_pair<T, 32>(a, s);
        case 64: return launch_pair<T, 64>(a, s);
        case 128: return launch_pair<T, 128>(a, s);
        case 256: return launch_pair<T, 256>(a, s);
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
    if (n <= 256 && (n + 31) / 32 * 32 * row <= 128 * 1024) {   // small images: one workgroup per image writes the merged context itself (no merge launch)
        *slice1 = (n + 31) / 32 * 32;
        *splits = 1;
        s1 = 0;
    }
    // halve the slice for more workgroups, but never beyond 64 splits per image (the merge holds their factors in LDS)
    while (s1 > 32 && (s1 * row > 72 * 1024 || s1 / 2 >= n ||
                       ((long long)((n + s1 - 1) / s1) * B < 256 && (n + s1 / 2 - 1) / (s1 / 2) <= 64)))
        s1 /= 2;
    while (s1 && (n + s1 - 1) / s1 > 64) s1 *= 2;
    if (s1) {
        *slice1 = s1;
        *splits = (n + s1 - 1) / s1;
    }
    int t = 4;
    while (t > 1 && ((long long)((n + 64 * t - 1) / (64 * t)) * B < 512 || 64 * t * row > 72 * 1024)) t /= 2;
    // Images that would keep more pixel slices than the in-launch exchange admits take slices twice as long when that
    // brings them under the limit (64x64 pixels of 32 channels: 8 slices of 512 pixels): the second launch then closes
    // the attention itself and no pass over y remains.
    static const bool long_slices = getenv("DMN_LA_SHORT_SLICES") == nullptr;
    if (long_slices && t == 4 && (n + 255) / 256 > kHandoffMaxWgsPerImage && (n + 511) / 512 <= kHandoffMaxWgsPerImage &&
        64 * 8 * row <= 56 * 1024)
        t = 8;
    *tpw = t;
    *slices = (n + 64 * t - 1) / (64 * t);
}

bool linattn_tail_fits(int n, int B, int C, int dtype) {
    int slice1, splits, tpw, slices;
    linattn_fused_plan(n, B, C, dtype, &slice1, &splits, &tpw, &slices);
    if (slices == 1) return true;   // one workgroup per image: nothing is exchanged, nothing waits
    const size_t esz = dtype == DMN_F32 ? 4 : 2;
    const size_t lds2 = ((3 * C + 8) * 4 + 15) / 16 * 16 + (size_t)kHeads * kD * (kD * esz + 32) + 64 * ((size_t)kHidden * esz + 32) +
                        (size_t)64 * tpw * (C * esz + 32) + 64;
    return slices <= kHandoffMaxWgsPerImage && lds2 <= (size_t)kHandoffMaxLdsBytes;
}

int launch_linattn_fused(LinAttnFusedArgs& a, hipStream_t s) {
    DMN_REQUIRE(linattn_fused_supported(a.C, kHeads, kD), DMN_ERR_INVALID, "fused linear attention: C=%d unsupported", a.C);
    DMN_REQUIRE(a.slice1 >= 32 && a.slice1 % 32 == 0 && a.tpw >= 1 && a.splits == (a.n + a.slice1 - 1) / a.slice1 &&
                    a.slices == (a.n + 64 * a.tpw - 1) / (64 * a.tpw),
                DMN_ERR_INVALID, "fused linear attention: inconsistent plan");
    DMN_REQUIRE(a.norm.stats && a.norm.G == 1, DMN_ERR_INVALID, "fused linear attention: GroupNorm(1) statistics of x required");
    DMN_REQUIRE(a.ctxT, DMN_ERR_INVALID, "fused linear attention: needs the merged-context buffer");
    DMN_REQUIRE(a.slices <= 65535 && a.splits <= 65535, DMN_ERR_LIMIT, "fused linear attention: image too large");
    if (a.tail.gran)
        DMN_REQUIRE(4 * a.slices <= kHandoffMaxSlots && a.out && a.tail.epoch && a.tail.giveups && a.tail_gamma && a.tail_beta,
                    DMN_ERR_INVALID, "fused linear attention: closing GroupNorm needs <= %d pixel slices, parameters and an output",
                    kHandoffMaxSlots);
    else DMN_REQUIRE(a.y && a.y_stats, DMN_ERR_INVALID, "fused linear attention: needs y and its statistics buffer");
    a.dbg = kernel_debug_buffer();
    if (a.dtype == DMN_F32) return launch_both<float>(a, s);
    return launch_both<__bf16>(a, s);
}
