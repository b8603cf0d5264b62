// Memory-bound kernels of the U-Net: GroupNorm apply (+SiLU, +residual), time-embedding path,
// 7x7 ingest (NCHW fp32 -> NHWC) and the 1x1 egress (NHWC -> NCHW fp32).
#include "dmn_kernels.h"
#include "dmn_norm.h"

namespace {

// Row of the class-embedding table for image b (unet.py:134-141): the label, or the extra "no class" row when no labels are
// given. A label outside [0, num_classes] is clamped so that the kernel never reads outside the table (the host reports such
// labels as an error at its next synchronisation point instead of paying a device round trip before every launch).
__device__ __forceinline__ long long ingest_label(const IngestArgs& a, int b) {
    if (!a.labels) return a.num_classes;
    const long long l = a.labels[b];
    return l < 0 ? 0 : (l > a.num_classes ? a.num_classes : l);
}

// ---------------------------------------------------------------------------------------------------
// out = [SiLU](GN(in)) [+ res], 16 bytes per lane, with optional partial statistics of `out`.
// Closes a ResnetBlock (parts/convnext.py:85-86: block2's norm+act, then + res_conv(x)) and a
// LinearAttention (parts/mha.py:41-42 GroupNorm(1) after to_out, utils.py:73-74 residual).
// Grid: (chunks, B); every workgroup owns kPixPerWg consecutive pixels of one image.
// ---------------------------------------------------------------------------------------------------
constexpr int kNormThreads = 256;
constexpr int kNormElems = 8192;        // elements per workgroup for large tensors ...
constexpr int kNormElemsSmall = 2048;   // ... and when that would leave most CUs without a workgroup

template <typename T, int ELEMS>
__global__ void __launch_bounds__(kNormThreads) norm_apply_kernel(const NormApplyArgs a, int pix_per_wg) {
    constexpr int PER = ElemTraits<T>::kPer16B;
    using V = decltype(Vec16<T>::v);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sCoef = reinterpret_cast<float*>(smem);
    float* sTmp = sCoef + 3 * a.C;            // 2*G_in
    const int b = blockIdx.y;
    const int C = a.C;
    const int vec_per_pix = C / PER;
    const int p0 = blockIdx.x * pix_per_wg;
    const int p1 = min(p0 + pix_per_wg, a.HW);
    const int nvec = (p1 - p0) * vec_per_pix;
    const T* in = static_cast<const T*>(a.in) + ((size_t)b * a.HW + p0) * C;
    const T* res = a.res ? static_cast<const T*>(a.res) + ((size_t)b * a.HW + p0) * C : nullptr;
    T* out = static_cast<T*>(a.out) + ((size_t)b * a.HW + p0) * C;

    // issue every data load first (a workgroup owns at most kNormElems elements), then derive the
    // normalisation coefficients while they are in flight
    constexpr int NV = ELEMS / PER / kNormThreads;
    V vin[NV], vres[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int i = threadIdx.x + j * kNormThreads;
        if (i < nvec) {
            vin[j] = *reinterpret_cast<const V*>(in + (size_t)i * PER);
            if (res) vres[j] = *reinterpret_cast<const V*>(res + (size_t)i * PER);
        }
    }
    norm_coefs_to_lds(a.norm, b, C, sCoef, sTmp);

    const bool want = a.epi.stats != nullptr;
    const int Go = want ? a.epi.G : 1;
    const int cpg_o = C / Go;
    // A thread always visits the same channel offset because kNormThreads % vec_per_pix == 0 (checked by the
    // launcher), so the group of each of its quads is fixed. Statistics are accumulated per 4-channel quad of
    // the thread's 16-byte vector (a quad never straddles an output group: groups hold a multiple of 4 channels).
    constexpr int NQ = PER / 4;
    int qgroup[NQ];
    const int cb = (threadIdx.x % vec_per_pix) * PER;
#pragma unroll
    for (int k = 0; k < NQ; ++k) qgroup[k] = (cb + 4 * k) / cpg_o;
    float ca[PER], cbeta[PER], ct[PER];
#pragma unroll
    for (int e = 0; e < PER; ++e) {
        ca[e] = sCoef[cb + e];
        cbeta[e] = sCoef[C + cb + e];
        ct[e] = sCoef[2 * C + cb + e];
    }
    // Statistics of the result as sums of deviations from a per-thread shift, the first value of each of the thread's
    // 4-channel quads (dmn_common.h: moments_from_shifted): one subtraction per element more than plain sums.
    int nmine = 0;   // vectors of this thread inside the slice
    float sd[NQ], qd[NQ], kshift[NQ];
#pragma unroll
    for (int k = 0; k < NQ; ++k) sd[k] = qd[k] = kshift[k] = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int i = threadIdx.x + j * kNormThreads;
        if (i < nvec) {
            ++nmine;
            Vec16<T> v, r;
            v.v = vin[j];
            if (res) r.v = vres[j];
#pragma unroll
            for (int e = 0; e < PER; ++e) {
                float y;
                if constexpr (sizeof(T) == 2) {
                    if (a.norm.silu) {
                        y = silu_affine_fast(v.get(e), ca[e], cbeta[e], ct[e]);
                    } else {
                        y = __builtin_fmaf(v.get(e), ca[e], cbeta[e]) + ct[e];
                    }
                } else {
                    y = v.get(e) * ca[e] + cbeta[e];
                    if (a.norm.silu) y = silu_precise(y);
                    y += ct[e];
                }
                if (res) y += r.get(e);
                v.set(e, y);
                const float yr = v.get(e);  // statistics of the value the consumer will read
                if (j == 0 && e % 4 == 0) kshift[e / 4] = yr;   // a thread's vectors are valid from j = 0 on
                const float d = yr - kshift[e / 4];
                sd[e / 4] += d;
                qd[e / 4] += d * d;
            }
            store_vec(reinterpret_cast<V*>(out + (size_t)i * PER), v.v);
        }
    }
    if (want) {
        // thread -> wave by the two-pass rule of dmn_norm.h, four groups at a time so that their cross-lane sums are in
        // flight together; every wave writes its own slot (4 * workgroup + wave): no LDS, no barrier, the consumer combines
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        float ts[NQ], tm2[NQ];
        const float tn = 4.f * nmine;
#pragma unroll
        for (int k = 0; k < NQ; ++k) moments_from_shifted(sd[k], qd[k], tn, kshift[k], ts[k], tm2[k]);
        for (int g0 = 0; g0 < Go; g0 += 4) {
            float ws[4], wn[4], wq[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float ms = 0.f, mn = 0.f;
#pragma unroll
                for (int k = 0; k < NQ; ++k) {
                    ms += (qgroup[k] == g0 + u) ? ts[k] : 0.f;
                    mn += (qgroup[k] == g0 + u) ? tn : 0.f;
                }
                ws[u] = wave_sum(ms);
                wn[u] = wave_sum(mn);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float wmean = wn[u] > 0.f ? ws[u] * __builtin_amdgcn_rcpf(wn[u]) : 0.f;
                float mq = 0.f;
#pragma unroll
                for (int k = 0; k < NQ; ++k) mq += (qgroup[k] == g0 + u) ? moments_term(ts[k], tm2[k], tn, wmean) : 0.f;
                wq[u] = wave_sum(mq);
            }
            if (lane == 0) {
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (g0 + u < Go) {
                        f32x4* dst = reinterpret_cast<f32x4*>(a.epi.stats) + (size_t)(b * Go + g0 + u) * a.epi.slots + blockIdx.x * 4 + wave;
                        *dst = f32x4{ws[u], wq[u], wn[u], 0.f};
                    }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Time path, one workgroup per sample, all fp32:
//   SinusoidalPositionEmbeddings (parts/positional_encoding.py:11-18) -> Linear -> GELU(erf) -> Linear
//   (modules/unet.py:61-66) -> for every ResnetBlock: SiLU -> Linear(4*dim, Cout) (parts/convnext.py:69,82)
// The label is either read from `t` (int64 or float32) or from a per-step schedule table indexed by a
// device-resident step counter (so a captured graph can be replayed for every step).
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ void time_mlp_body(const TimeArgs& a, int bx, int b) {
    // grid (chunks, rows): every workgroup recomputes the two small dense layers (20k MACs) and then produces
    // its slice of the per-block biases. Weights are stored TRANSPOSED ([in][out]) so that thread o reads
    // column o coalesced and all its loads are independent (the kernel is pure latency otherwise).
    __shared__ float sE[64];
    __shared__ float sH[512];
    __shared__ float sT[512];
    const int tid = threadIdx.x;
    const int dim = a.dim, hid = 4 * a.dim, half = a.dim / 2;
    float label;
    if (a.h2_out) {            // tabulating: workgroup b takes table row b
        label = a.table[(size_t)b * a.table_cols + a.label_col];
    } else if (a.t) {
        label = a.t_is_float ? static_cast<const float*>(a.t)[b] : (float)static_cast<const long long*>(a.t)[b];
    } else {
        const int row = *a.step_idx;
        label = a.table[(size_t)row * a.table_cols + a.label_col];
        if (a.step_copy && bx == 0 && b == 0 && tid == 0) *a.step_copy = row;
        if (a.h2) {            // the hidden vector of this row was tabulated before the loop
            for (int o = tid; o < hid; o += 256) sT[o] = a.h2[(size_t)row * hid + o];
            __syncthreads();
            const int oo = bx * 256 + tid;
            if (oo < a.nout) {
                float acc = 0.f;
#pragma unroll 16
                for (int k = 0; k < hid; ++k) acc = fmaf(sT[k], a.wb[(size_t)k * a.nout + oo], acc);
                a.out[(size_t)b * a.nout + oo] = acc + a.bb[oo];
            }
            return;
        }
    }
    if (tid < half) {
        const float e = label * a.freqs[tid];
        sE[tid] = sinf(e);
        sE[half + tid] = cosf(e);
    }
    __syncthreads();
    for (int o = tid; o < hid; o += 256) {
        float acc = 0.f;
#pragma unroll 16
        for (int k = 0; k < dim; ++k) acc = fmaf(sE[k], a.w1[(size_t)k * hid + o], acc);
        acc += a.b1[o];
        sH[o] = 0.5f * acc * (1.0f + erff(acc * 0.70710678118654752440f));
    }
    __syncthreads();
    for (int o = tid; o < hid; o += 256) {
        float acc = 0.f;
#pragma unroll 16
        for (int k = 0; k < hid; ++k) acc = fmaf(sH[k], a.w2[(size_t)k * hid + o], acc);
        sT[o] = silu_precise(acc + a.b2[o]);
        if (a.h2_out) a.h2_out[(size_t)b * hid + o] = sT[o];
    }
    if (a.h2_out) return;
    __syncthreads();
    const int o = bx * 256 + tid;
    if (o < a.nout) {
        float acc = 0.f;
#pragma unroll 16
        for (int k = 0; k < hid; ++k) acc = fmaf(sT[k], a.wb[(size_t)k * a.nout + o], acc);
        a.out[(size_t)b * a.nout + o] = acc + a.bb[o];
    }
}

__global__ void __launch_bounds__(256) time_mlp_kernel(const TimeArgs a) { time_mlp_body(a, blockIdx.x, blockIdx.y); }

// ---------------------------------------------------------------------------------------------------
// Ingest: init_conv 7x7 pad 3 (modules/unet.py:41,132) reading the reference's NCHW fp32 layout and
// writing NHWC T. Cin is 3 (too thin for an MFMA K dimension): a direct fp32 VALU convolution with the
// weights ([Cin*49][Cout], transposed on the host) read through uniform (scalar) loads. 16x16 output pixels per workgroup, one pixel per thread.
// ---------------------------------------------------------------------------------------------------
template <typename T, int COUT>
__global__ void __launch_bounds__(256) ingest_kernel(const IngestArgs a) {
    constexpr int TS = 16, HALO = 3, IW = TS + 2 * HALO;
    __shared__ float sIn[4][IW][IW + 1];
    const int tiles_x = (a.W + TS - 1) / TS;
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x % tiles_x;
    const int b = blockIdx.y;
    const int y0 = ty * TS - HALO, x0 = tx * TS - HALO;
    if (a.epoch && blockIdx.x == 0 && b == 0 && threadIdx.x == 0) {   // a new network evaluation; 0 is "never written"
        const unsigned e = *a.epoch + 1u;
        *a.epoch = e ? e : 1u;
        if (a.trace) a.trace[e & 8191u] = (long long)wall_clock64();
    }
    for (int i = threadIdx.x; i < a.Cin * IW * IW; i += 256) {
        const int c = i / (IW * IW), r = i % (IW * IW);
        const int ly = r / IW, lx = r % IW;
        const int gy = y0 + ly, gx = x0 + lx;
        float v = 0.f;
        if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) v = a.x[((size_t)(b * a.Cin + c) * a.H + gy) * a.W + gx];
        sIn[c][ly][lx] = v;
    }
    __syncthreads();
    const int py = threadIdx.x / TS, px = threadIdx.x % TS;
    float acc[COUT];
    const float* emb = a.class_embed ? a.class_embed + (size_t)ingest_label(a, b) * COUT : nullptr;
#pragma unroll
    for (int o = 0; o < COUT; ++o) acc[o] = a.bias[o] + (emb ? emb[o] : 0.f);
    for (int c = 0; c < a.Cin; ++c)
        for (int ky = 0; ky < 7; ++ky)
#pragma unroll
            for (int kx = 0; kx < 7; ++kx) {
                const float v = sIn[c][py + ky][px + kx];
                const float* w = a.w + (size_t)((c * 7 + ky) * 7 + kx) * COUT;   // uniform address: scalar loads
#pragma unroll
                for (int o = 0; o < COUT; ++o) acc[o] = fmaf(v, w[o], acc[o]);
            }
    const int oy = ty * TS + py, ox = tx * TS + px;
    if (oy < a.H && ox < a.W) {
        T* dst = static_cast<T*>(a.out) + ((size_t)(b * a.H + oy) * a.W + ox) * COUT;
#pragma unroll
        for (int o = 0; o < COUT; ++o) dst[o] = from_f32<T>(acc[o]);
    }
}

// bf16 MFMA form of the ingest: the image tile is staged as [row][pixel][4 channels] bf16 (channel 3 = 0), so for
// kernel row ky the 8 pixels x 4 channels to the right of an output pixel are 64 contiguous bytes = one k-step
// (the eighth pixel meets zero weights). 7 k-steps per output tile instead of 147 scalar taps.
template <int COUT>
__device__ __forceinline__ void ingest_mfma_body(const IngestArgs& a, int bx, int b) {
    constexpr int TS = 16, HALO = 3, IH = TS + 2 * HALO, RW = 24, NT = COUT / 16;
    __shared__ __attribute__((aligned(16))) __bf16 sIn[IH * RW * 4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lq = lane >> 4;
    const int tiles_x = (a.W + TS - 1) / TS;
    const int ty = bx / tiles_x, tx = bx % tiles_x;
    const int y0 = ty * TS - HALO, x0 = tx * TS - HALO;
    if (a.epoch && bx == 0 && b == 0 && tid == 0) {   // a new network evaluation; 0 is "never written"
        const unsigned e = *a.epoch + 1u;
        *a.epoch = e ? e : 1u;
        if (a.trace) a.trace[e & 8191u] = (long long)wall_clock64();
    }
    // weights first: 7 * NT fragments per wave, straight into registers
    bf16x8 wf[NT][7];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int ky = 0; ky < 7; ++ky)
            wf[nt][ky] = *reinterpret_cast<const bf16x8*>(static_cast<const __bf16*>(a.wfrag) + ((size_t)(nt * 7 + ky) * 64 + lane) * 8);
    f32x4 bias[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bias[nt] = *reinterpret_cast<const f32x4*>(a.bias + nt * 16 + lq * 4);
    if (a.class_embed) {   // class embedding row of this image, added like a second bias (unet.py:139-141)
        const float* emb = a.class_embed + (size_t)ingest_label(a, b) * COUT;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bias[nt] += *reinterpret_cast<const f32x4*>(emb + nt * 16 + lq * 4);
    }
    for (int i = tid; i < IH * RW; i += 256) {
        const int ly = i / RW, lx = i % RW;
        const int gy = y0 + ly, gx = x0 + lx;
        const bool in = lx < IH && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
        bf16x4 v;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float f = 0.f;
            if (in && c < a.Cin) f = a.x[((size_t)(b * a.Cin + c) * a.H + gy) * a.W + gx];
            v[c] = (__bf16)f;
        }
        *reinterpret_cast<bf16x4*>(&sIn[i * 4]) = v;
    }
    __syncthreads();
    // wave w: output rows 4w .. 4w+3 (one 16-pixel tile each), all output channels
    f32x4 acc[4][NT];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[r][nt] = bias[nt];
#pragma unroll
    for (int ky = 0; ky < 7; ++ky)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const __bf16* src = &sIn[(((wave * 4 + r) + ky) * RW + l15 + 2 * lq) * 4];   // 8-byte aligned
            const bf16x4 lo = *reinterpret_cast<const bf16x4*>(src);
            const bf16x4 hi = *reinterpret_cast<const bf16x4*>(src + 4);
            const bf16x8 xv = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                acc[r][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt][ky], xv, acc[r][nt], 0, 0, 0);
        }
    // lane: pixel l15 of the row, output channels 16*nt + 4*lq .. +3
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int oy = ty * TS + wave * 4 + r, ox = tx * TS + l15;
        if (oy < a.H && ox < a.W) {
            __bf16* dst = static_cast<__bf16*>(a.out) + ((size_t)(b * a.H + oy) * a.W + ox) * COUT;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                bf16x4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = (__bf16)acc[r][nt][j];
                store_vec(reinterpret_cast<bf16x4*>(dst + nt * 16 + lq * 4), o);
            }
        }
    }
}

template <int COUT>
__global__ void __launch_bounds__(256) ingest_mfma_kernel(const IngestArgs a) {
    ingest_mfma_body<COUT>(a, blockIdx.x, blockIdx.y);
}

// The time path and the ingest convolution are independent and open every network evaluation: one launch runs both
// (blockIdx.x < ingest_blocks: an ingest tile of image blockIdx.y; beyond: a slice of the time path of row blockIdx.y).
template <int COUT>
__global__ void __launch_bounds__(256) ingest_time_kernel(const IngestArgs a, const TimeArgs t, int ingest_blocks) {
    if ((int)blockIdx.x < ingest_blocks) {
        ingest_mfma_body<COUT>(a, blockIdx.x, blockIdx.y);
    } else if ((int)blockIdx.y < t.rows) {
        time_mlp_body(t, blockIdx.x - ingest_blocks, blockIdx.y);
    }
}

// ---------------------------------------------------------------------------------------------------
// Egress: final_conv tail (modules/unet.py:112-116): [GroupNorm(groups) -> SiLU ->] 1x1 conv to 3 or 6
// channels, written in the reference's NCHW fp32 layout. One pixel per thread.
// ---------------------------------------------------------------------------------------------------
template <typename T, int CO>   // CO > 0: compile-time output channels; CO == 0: a.Cout <= 8 at run time
__global__ void __launch_bounds__(256) egress_kernel(const EgressArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sCoef = reinterpret_cast<float*>(smem);   // 3*C
    float* sW = sCoef + 3 * a.C;                     // Cout*C + Cout, then 2*G scratch
    const int b = blockIdx.y, C = a.C;
    const int Cout = CO ? CO : a.Cout;
    constexpr int NO = CO ? CO : 8;
    constexpr int PER = ElemTraits<T>::kPer16B;
    constexpr int NVMAX = 8;                         // vectors of one pixel kept in registers (C <= 64 bf16 / 32 f32)
    using V = decltype(Vec16<T>::v);
    const bool norm = a.norm.stats != nullptr;
    const int p = blockIdx.x * 256 + threadIdx.x;
    const bool live = p < a.HW;
    float* sEps = sW + Cout * C + Cout + 2 * (norm ? a.norm.G : 0) + 16;   // [Cout][256] network outputs of the workgroup's pixels (fuse_step)
    const int nv = C / PER;
    const T* in = static_cast<const T*>(a.in) + ((size_t)b * a.HW + (live ? p : 0)) * C;
    // the pixel's channels are requested before the statistics round trip
    V vin[NVMAX];
#pragma unroll
    for (int j = 0; j < NVMAX; ++j)
        if (j < nv) vin[j] = *reinterpret_cast<const V*>(in + j * PER);
    if (norm) norm_coefs_to_lds(a.norm, b, C, sCoef, sW);
    for (int i = threadIdx.x; i < Cout * C; i += 256) sW[i] = a.w[i];
    for (int i = threadIdx.x; i < Cout; i += 256) sW[Cout * C + i] = a.bias[i];
    __syncthreads();
    if (!live && !a.fuse_step) return;
    float acc[NO];
#pragma unroll
    for (int o = 0; o < NO; ++o) acc[o] = o < Cout ? sW[Cout * C + o] : 0.f;
    auto consume = [&](const V& raw, int cb) {
        Vec16<T> v;
        v.v = raw;
        float y[PER];
#pragma unroll
        for (int q = 0; q < PER; q += 4) {
            f32x4 ca{1.f, 1.f, 1.f, 1.f}, cbeta{0.f, 0.f, 0.f, 0.f};
            if (norm) {
                ca = *reinterpret_cast<const f32x4*>(sCoef + cb + q);
                cbeta = *reinterpret_cast<const f32x4*>(sCoef + C + cb + q);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float x = v.get(q + j);
                if (norm) {
                    if constexpr (sizeof(T) == 2) {
                        x = __builtin_fmaf(x, ca[j], cbeta[j]);
                        if (a.norm.silu) x = silu_f(x);
                    } else {
                        x = x * ca[j] + cbeta[j];
                        if (a.norm.silu) x = silu_precise(x);
                    }
                }
                y[q + j] = x;
            }
        }
#pragma unroll
        for (int o = 0; o < NO; ++o) {
            if (CO == 0 && o >= Cout) break;
#pragma unroll
            for (int q = 0; q < PER; q += 4) {
                const f32x4 w = *reinterpret_cast<const f32x4*>(sW + o * C + cb + q);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[o] = fmaf(y[q + j], w[j], acc[o]);
            }
        }
    };
    if (nv <= NVMAX) {
#pragma unroll
        for (int j = 0; j < NVMAX; ++j)
            if (j < nv) consume(vin[j], j * PER);
    } else {
        for (int cb = 0; cb < C; cb += PER) consume(*reinterpret_cast<const V*>(in + cb), cb);
    }
    if (!a.fuse_step) {
#pragma unroll
        for (int o = 0; o < NO; ++o)
            if (o < Cout) __builtin_nontemporal_store(acc[o], &a.out[((size_t)b * Cout + o) * a.HW + p]);
        return;
    }
    // ---- sampler update on the workgroup's 256 pixels: thread (channel c, pixel quad j) takes the four elements
    //      c * HW + p0 + 4j .. +3 of the image, i.e. exactly one quad of the noise generator (HW and p0 are multiples of 4)
#pragma unroll
    for (int o = 0; o < NO; ++o)
        if (o < Cout) sEps[o * 256 + threadIdx.x] = acc[o];
    __syncthreads();
    const DdpmStepArgs& st = a.step;
    const StepCommon& sc = st.c;
    const int row = sc.step_idx ? *sc.step_idx : sc.step;
    if (st.step_next && blockIdx.x == 0 && b == 0 && threadIdx.x == 0) *st.step_next = row + 1;
    const float* tr = sc.table + (size_t)row * DMN_TABLE_COLS;
    const int per = sc.C * a.HW;
    const int p0 = blockIdx.x * 256;
    for (int t = threadIdx.x; t < sc.C * 64; t += 256) {
        const int c = t >> 6, j = t & 63;
        const int pq = p0 + 4 * j;
        if (pq >= a.HW) continue;
        const int e0 = c * a.HW + pq;
        float z[4];
        load_noise(sc, (unsigned)tr[8], 1u, b, e0 >> 2, per, z);
        const size_t xi = (size_t)b * per + e0;
        const f32x4 x = *reinterpret_cast<const f32x4*>(sc.x + xi);
        f32x4 xn;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float v = st.learned_variance ? sEps[(sc.C + c) * 256 + 4 * j + r] : 0.f;
            xn[r] = ddpm_update_element(st, tr, x[r], sEps[c * 256 + 4 * j + r], v, z[r]);
        }
        store_vec(reinterpret_cast<f32x4*>(sc.x_out + xi), xn);
        // the history row is not read again on the device: keep it out of the way of what the next launch reads
        if (sc.keep) __builtin_nontemporal_store(xn, reinterpret_cast<f32x4*>(sc.keep + (size_t)row * sc.keep_stride + xi));
    }
}

__global__ void __launch_bounds__(256) checksum_kernel(const uint4* p, size_t n16, unsigned long long* slot) {
    unsigned long long acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
        const uint4 v = p[i];
        acc += (unsigned long long)v.x + v.y + v.z + v.w;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(slot, acc);
}

// Fills the whole LDS of the CU a workgroup lands on with `pattern` (one workgroup per CU: it asks for all 160 KiB).
__global__ void __launch_bounds__(256) lds_poison_kernel(unsigned pattern, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned* w = reinterpret_cast<unsigned*>(smem);
    for (int i = threadIdx.x; i < 160 * 1024 / 4; i += 256) w[i] = pattern;
    __syncthreads();
    if (sink && w[threadIdx.x] != pattern) *sink = 1;   // keeps the stores alive
}

// span of one launch from the stamps its workgroups left ([wg][8]: slot 0 clock at start, slot 5 clock at end; unused
// entries are zero): acc[0] += latest end - earliest start, acc[1] += 1; the stamps are cleared for the next launch.
__global__ void __launch_bounds__(256) stamp_span_kernel(long long* stamps, int wgs, unsigned long long* acc) {
    __shared__ long long sMin[256], sMax[256];
    long long lo = 0x7fffffffffffffffll, hi = 0;
    for (int i = threadIdx.x; i < wgs; i += 256) {
        const long long b = stamps[(size_t)i * 8], e = stamps[(size_t)i * 8 + 5];
        if (b && e) {
            lo = b < lo ? b : lo;
            hi = e > hi ? e : hi;
        }
        stamps[(size_t)i * 8] = 0;
        stamps[(size_t)i * 8 + 5] = 0;
    }
    sMin[threadIdx.x] = lo;
    sMax[threadIdx.x] = hi;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < 256; ++i) {
            lo = sMin[i] < lo ? sMin[i] : lo;
            hi = sMax[i] > hi ? sMax[i] : hi;
        }
        if (hi > lo) {
            acc[0] += (unsigned long long)(hi - lo);
            acc[1] += 1;
        }
    }
}

}  // namespace

int launch_stamp_span(long long* stamps, int wgs, unsigned long long* acc, hipStream_t s) {
    hipLaunchKernelGGL(stamp_span_kernel, dim3(1), dim3(256), 0, s, stamps, wgs, acc);
    DMN_HIP_CHECK(hipGetLastError());
    return DMN_OK;
}

int launch_lds_poison(unsigned pattern, hipStream_t s) {
    static bool attr = false;
    if (!attr) {
        DMN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&lds_poison_kernel),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr = true;
    }
    hipLaunchKernelGGL(lds_poison_kernel, dim3(1024), dim3(256), 160 * 1024, s, pattern, (unsigned*)nullptr);
    DMN_HIP_CHECK(hipGetLastError());
    return DMN_OK;
}

int launch_checksum(const void* p, size_t bytes, unsigned long long* slot, hipStream_t s) {
    hipLaunchKernelGGL(checksum_kernel, dim3(2048), dim3(256), 0, s, static_cast<const uint4*>(p), bytes / 16, slot);
    DMN_HIP_CHECK(hipGetLastError());
    return DMN_OK;
}

static int norm_apply_elems(int B, int HW, int C) {
    return (long long)B * HW * C / kNormElems >= 1024 ? kNormElems : kNormElemsSmall;
}

static int norm_apply_wgs(int B, int HW, int C) {   // workgroups (pixel slices) per image
    int pix = norm_apply_elems(B, HW, C) / C;
    if (pix < 1) pix = 1;
    if (pix > HW) pix = HW;
    return (HW + pix - 1) / pix;
}
// partial moments per (image, group) of the output statistics: one per wave of every workgroup
int norm_apply_slots(int B, int HW, int C) { return 4 * norm_apply_wgs(B, HW, C); }

int launch_norm_apply(NormApplyArgs& a, hipStream_t s) {
    const int per = a.dtype == DMN_F32 ? 4 : 8;
    DMN_REQUIRE(a.C % per == 0 && a.C % a.norm.G == 0, DMN_ERR_INVALID, "norm_apply: C=%d groups=%d", a.C, a.norm.G);
    DMN_REQUIRE(kNormThreads % (a.C / per) == 0 && a.C <= kNormElemsSmall, DMN_ERR_INVALID, "norm_apply: C=%d must be a power-of-two multiple of %d", a.C, per);
    const int planB = a.plan_B > 0 ? a.plan_B : a.B;
    const int slots = norm_apply_wgs(planB, a.HW, a.C);
    const bool small = norm_apply_elems(planB, a.HW, a.C) == kNormElemsSmall;
    const int pix = (a.HW + slots - 1) / slots;
    int Go = 1;
    if (a.epi.stats) {
        Go = a.epi.G;
        DMN_REQUIRE(a.C % Go == 0 && (a.C / Go) % 4 == 0, DMN_ERR_INVALID, "norm_apply: output groups %d", Go);
        a.epi.slots = 4 * slots;
    }
    const size_t lds = (3 * (size_t)a.C + 2 * (size_t)a.norm.G) * 4;
    dim3 grid(slots, a.B);
    if (a.dtype == DMN_F32) {
        if (small) hipLaunchKernelGGL((norm_apply_kernel<float, kNormElemsSmall>), grid, dim3(kNormThreads), lds, s, a, pix);
        else hipLaunchKernelGGL((norm_apply_kernel<float, kNormElems>), grid, dim3(kNormThreads), lds, s, a, pix);
    } else {
        if (small) hipLaunchKernelGGL((norm_apply_kernel<__bf16, kNormElemsSmall>), grid, dim3(kNormThreads), lds, s, a, pix);
        else hipLaunchKernelGGL((norm_apply_kernel<__bf16, kNormElems>), grid, dim3(kNormThreads), lds, s, a, pix);
    }
    DMN_HIP_CHECK(hipGetLastError());
    return DMN_OK;
}

int launch_time_mlp(TimeArgs& a, hipStream_t s) {
    DMN_REQUIRE(a.dim <= 64 && a.dim % 2 == 0 && 4 * a.dim <= 512, DMN_ERR_LIMIT, "time_mlp: dim=%d exceeds 64", a.dim);
    hipLaunchKernelGGL(time_mlp_kernel, dim3((a.nout + 255) / 256, a.rows), dim3(256), 0, s, a);
    DMN_HIP_CHECK(hipGetLastError());
    return DMN_OK;
}

int launch_time_hidden(TimeArgs& a, int table_rows, hipStream_t s) {
    DMN_REQUIRE(a.table && a.h2_out && table_rows > 0 && table_rows <= 65535, DMN_ERR_INVALID, "time path: nothing to tabulate");
    DMN_REQUIRE(a.dim <= 64 && a.dim % 2 == 0 && 4 * a.dim <= 512, DMN_ERR_LIMIT, "time_mlp: dim=%d exceeds 64", a.dim);
    hipLaunchKernelGGL(time_mlp_kernel, dim3(1, table_rows), dim3(256), 0, s, a);
    DMN_HIP_CHECK(hipGetLastError());
    return DMN_OK;
}

bool ingest_time_fusable(const IngestArgs& a, const TimeArgs& t) {
    return a.dtype == DMN_BF16 && a.wfrag && (a.Cout == 32 || a.Cout == 64) && t.rows <= a.B && t.dim <= 64;
}

int launch_ingest_time(IngestArgs& a, TimeArgs& t, hipStream_t s) {
    DMN_REQUIRE(ingest_time_fusable(a, t), DMN_ERR_INVALID, "ingest + time path: not fusable in this configuration");
    const int ib = ((a.H + 15) / 16) * ((a.W + 15) / 16);
    dim3 grid(ib + (t.nout + 255) / 256, a.B);
    if (a.Cout == 32) hipLaunchKernelGGL(ingest_time_kernel<32>, grid, dim3(256), 0, s, a, t, ib);
    else hipLaunchKernelGGL(ingest_time_kernel<64>, grid, dim3(256), 0, s, a, t, ib);
    DMN_HIP_CHECK(hipGetLastError());
    return DMN_OK;
}

int launch_ingest(IngestArgs& a, hipStream_t s) {
    DMN_REQUIRE(a.Cin >= 1 && a.Cin <= 4, DMN_ERR_LIMIT, "ingest: Cin=%d (max 4)", a.Cin);
    dim3 grid(((a.H + 15) / 16) * ((a.W + 15) / 16), a.B);
#define DMN_INGEST(T, N) hipLaunchKernelGGL((ingest_kernel<T, N>), grid, dim3(256), 0, s, a)
    if (a.dtype == DMN_BF16 && a.wfrag && (a.Cout == 32 || a.Cout == 64)) {
        if (a.Cout == 32) hipLaunchKernelGGL(ingest_mfma_kernel<32>, grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL(ingest_mfma_kernel<64>, grid, dim3(256), 0, s, a);
    } else if (a.Cout == 32) {
        if (a.dtype == DMN_F32) DMN_INGEST(float, 32); else DMN_INGEST(__bf16, 32);
    } else if (a.Cout == 64) {
        if (a.dtype == DMN_F32) DMN_INGEST(float, 64); else DMN_INGEST(__bf16, 64);
    } else {
        dmn_set_error("ingest: Cout=%d not instantiated (32 or 64)", a.Cout);
        return DMN_ERR_INVALID;
    }
#undef DMN_INGEST
    DMN_HIP_CHECK(hipGetLastError());
    return DMN_OK;
}

int launch_egress(EgressArgs& a, hipStream_t s) {
    DMN_REQUIRE(a.Cout >= 1 && a.Cout <= 8, DMN_ERR_LIMIT, "egress: Cout=%d (max 8)", a.Cout);
    size_t lds = (3 * (size_t)a.C + (size_t)a.Cout * a.C + a.Cout + 2 * (size_t)(a.norm.stats ? a.norm.G : 0) + 16) * 4;
    if (a.fuse_step) {
        const int ci = a.step.c.C;
        DMN_REQUIRE(a.HW % 4 == 0 && a.Cout == ci * (a.step.learned_variance ? 2 : 1) && a.step.c.HW == a.HW && a.step.c.B == a.B &&
                        a.step.c.x && a.step.c.x_out && a.step.c.table,
                    DMN_ERR_INVALID, "egress: sampler update needs HW %% 4 == 0 and %d network channels", ci * (a.step.learned_variance ? 2 : 1));
        lds += (size_t)a.Cout * 256 * 4;
    }
    dim3 grid((a.HW + 255) / 256, a.B);
    DMN_REQUIRE(a.C % 8 == 0, DMN_ERR_INVALID, "egress: C=%d must be a multiple of 8", a.C);
#define DMN_EGRESS(T, CO) hipLaunchKernelGGL((egress_kernel<T, CO>), grid, dim3(256), lds, s, a)
#define DMN_EGRESS_T(T)                                   \
    switch (a.Cout) {                                     \
        case 1: DMN_EGRESS(T, 1); break;                  \
        case 2: DMN_EGRESS(T, 2); break;                  \
        case 3: DMN_EGRESS(T, 3); break;                  \
        case 6: DMN_EGRESS(T, 6); break;                  \
        default: DMN_EGRESS(T, 0); break;                 \
    }
    if (a.dtype == DMN_F32) {
        DMN_EGRESS_T(float)
    } else {
        DMN_EGRESS_T(__bf16)
    }
#undef DMN_EGRESS_T
#undef DMN_EGRESS
    DMN_HIP_CHECK(hipGetLastError());
    return DMN_OK;
}
