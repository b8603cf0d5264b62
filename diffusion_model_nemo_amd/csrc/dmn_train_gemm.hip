// Matrix-core kernels of the training step: the implicit GEMM over pixels that serves every convolution /
// linear layer forward and its input gradient, and the pixel-reduction GEMM of the weight gradients
// (what torch.nn.functional.conv2d / conv_transpose2d / linear and their autograd do in the reference:
// diffusion_model_nemo/parts/convnext.py:11-13, utils.py:77-82, parts/mha.py:13-14,40-42, modules/unet.py:41,61-66).
// gfx950, 256-thread workgroups = 4 wave64, v_mfma_f32_16x16x4_f32 (exact fp32 products, fp32 accumulate).
#include <stdlib.h>

#include "dmn_common.h"
#include "dmn_train.h"
#include "dmn_train_io.h"

namespace {

__device__ __forceinline__ void decode_m(int m, int Ho, int Wo, int pm, int& b, int& oy, int& ox) {
    const int hw = Ho * Wo;
    b = m / hw;
    const int rem = m - b * hw;
    if (pm) {   // parity-major order inside an image: all pixels of one (oy & 1, ox & 1) class are consecutive
        const int qn = hw >> 2, w2 = Wo >> 1;
        const int ph = rem / qn, idx = rem - ph * qn;
        const int yy = idx / w2, xx = idx - yy * w2;
        oy = 2 * yy + (ph >> 1);
        ox = 2 * xx + (ph & 1);
    } else {
        oy = rem / Wo;
        ox = rem - oy * Wo;
    }
}

// source coordinate of output coordinate o under kernel offset k; false when the tap falls outside / between pixels
__device__ __forceinline__ bool src_coord(int o, int k, int stride, int pad, int tmode, int lim, int& i) {
    if (tmode == 0) {
        i = o * stride - pad + k;
        return i >= 0 && i < lim;
    }
    const int t = o + pad - k;
    if (t < 0) return false;
    if (stride == 2) {
        if (t & 1) return false;
        i = t >> 1;
    } else {
        if (t % stride) return false;
        i = t / stride;
    }
    return i < lim;
}

// BN: output channels per workgroup. 64 output pixels per workgroup, 16 per wave. The A operand (pixels x channels)
// goes from global memory straight into the lanes that feed it to the matrix core: lane (r = lane & 15, q = lane >> 4)
// loads channels 4q..4q+3 of a 16-channel step of pixel r, and instruction s of the step consumes element s, i.e.
// the MFMA's k index q stands for channel 4q + s. The B operand (weights) is staged through LDS for the four waves,
// read with the same channel permutation. One stage = 32 channels of one tap; the next stage's loads are in flight
// while the current one is multiplied.
// SLOW: Cin is not a multiple of 32 (the 7x7 ingest on 3 channels): k runs over the flattened (tap, channel) index
// and every element is gathered by itself.
// BF: both operands are rounded to bfloat16 on their way into the matrix core and one v_mfma_f32_16x16x32_bf16 per
// stage replaces the eight fp32 instructions: k slot i of lane group q stands for channel 16*(i>>2) + 4q + (i&3) of
// the stage, so the global loads and the LDS reads are those of the fp32 variant.
__device__ __forceinline__ bf16x8 pack_bf16(const float* v) {
    bf16x8 o;
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = (__bf16)v[i];
    return o;
}

// GroupNorm statistics of a workgroup's 64-pixel x BN-channel output tile (all pixels in image b): per-lane sums over
// its four pixels, the four lane groups of a wave, the four waves (through `red`, >= 8 * BN floats of LDS that
// nothing reads any more), then one atomic pair per group the tile touches.
template <int BN>
__device__ __forceinline__ void tile_stats(const float (&s1)[BN / 16], const float (&s2)[BN / 16], float* red, int n0, int N, int Cg,
                                           double* stats_b) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, q = lane >> 4;
#pragma unroll
    for (int j = 0; j < BN / 16; ++j) {
        float a = s1[j], b = s2[j];
        a += __shfl_xor(a, 16, 64);
        a += __shfl_xor(a, 32, 64);
        b += __shfl_xor(b, 16, 64);
        b += __shfl_xor(b, 32, 64);
        if (q == 0) {
            red[(wave * BN + 16 * j + r) * 2] = a;
            red[(wave * BN + 16 * j + r) * 2 + 1] = b;
        }
    }
    __syncthreads();
    if (tid < BN) {
        const int n = n0 + tid;
        if (n < N && (tid == 0 || n % Cg == 0)) {
            double a = 0.0, b = 0.0;   // 16-pixel partials in float32 so far; from here on float64 (dmn_train.h)
            const int g = n / Cg;
            for (int t = tid; t < BN && n0 + t < N && (n0 + t) / Cg == g; ++t)
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    a += (double)red[(w * BN + t) * 2];
                    b += (double)red[(w * BN + t) * 2 + 1];
                }
            unsafeAtomicAdd(stats_b + 2 * g, a);
            unsafeAtomicAdd(stats_b + 2 * g + 1, b);
        }
    }
}

// Output tile of a pixel GEMM, 64 pixels x BN channels: the accumulators (+ bias) go through LDS so that every thread
// writes (and, when accumulating, first reads) ONE 16-byte piece of a pixel's channel row -- whole cache lines per
// pixel instead of a 4- or 2-byte element per lane. Es: 64 * (BN + 4) floats that nothing else uses any more.
// rowptr(p): output row of tile pixel p, or null. Also returns the per-lane statistics sums of the biased values.
template <typename T, int BN, typename RowPtr>
__device__ __forceinline__ void store_tile(float* Es, const f32x4 (&acc)[BN / 16], const float* bias, int accumulate, int n0, int N,
                                           RowPtr rowptr, float (&s1)[BN / 16], float (&s2)[BN / 16]) {
    constexpr int LDE = BN + 4, V = Io<T>::VEC, PER_ROW = BN / V;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, q = lane >> 4;
#pragma unroll
    for (int j = 0; j < BN / 16; ++j) {
        const int n = n0 + 16 * j + r;
        const float bv = (bias && n < N) ? bias[n] : 0.f;
        s1[j] = s2[j] = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float v = acc[j][i] + bv;
            Es[(16 * wave + 4 * q + i) * LDE + 16 * j + r] = v;
            if (n < N) {
                s1[j] += v;
                s2[j] += v * v;
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < (64 * PER_ROW + 255) / 256; ++k) {
        const int idx = tid + 256 * k;
        if (idx >= 64 * PER_ROW) break;
        const int p = idx / PER_ROW, cv = (idx - p * PER_ROW) * V;
        T* o = rowptr(p);
        if (!o || n0 + cv >= N) continue;
        float v[V];
#pragma unroll
        for (int e = 0; e < V; ++e) v[e] = Es[p * LDE + cv + e];
        if (accumulate) {
            float old[V];
            Io<T>::ldv(o + n0 + cv, old);
#pragma unroll
            for (int e = 0; e < V; ++e) v[e] += old[e];
        }
        Io<T>::stv(o + n0 + cv, v);
    }
    __syncthreads();
}
// whether the vector epilogue applies: channel rows made of whole 16-byte pieces
template <typename T> __device__ __forceinline__ bool tile_store_ok(const PGemmArgs& a) {
    return (a.N % Io<T>::VEC) == 0 && (a.ldout % Io<T>::VEC) == 0 && (reinterpret_cast<uintptr_t>(a.out) & 15) == 0;
}

template <typename T, int BN, bool SLOW, bool BF>
__global__ __launch_bounds__(256) void pgemm_kernel(const PGemmArgs a, const int M, const int pm) {
    const T* in = reinterpret_cast<const T*>(a.in);
    T* out = reinterpret_cast<T*>(a.out);
    constexpr int LDB = BN + 4, NJ = BN / 16, TPR = BN / 4, RPP = 256 / TPR, NB = 32 / RPP;
    __shared__ float Bs[64 * LDB];   // weight tile (32 rows) during the loop, the output tile (64 rows) afterwards
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, q = lane >> 4;
    const int m0 = blockIdx.x * 64, n0 = blockIdx.y * BN;
    const int mA = m0 + 16 * wave + r;
    const bool mvalid = mA < M;
    int pb = 0, poy = 0, pox = 0;
    if (mvalid) decode_m(mA, a.Ho, a.Wo, pm, pb, poy, pox);
    const int K = a.kh * a.kw * a.Cin;
    int ky0 = 0, kx0 = 0, kst = 1;
    if (pm) {   // the whole tile shares one parity class: only every second kernel row / column can contribute
        const int hw = a.Ho * a.Wo;
        const int ph = (m0 % hw) / (hw >> 2);
        ky0 = ((ph >> 1) + a.pad) & 1;
        kx0 = ((ph & 1) + a.pad) & 1;
        kst = 2;
    }
    const int nky = (a.kh - ky0 + kst - 1) / kst, nkx = (a.kw - kx0 + kst - 1) / kst, nch = a.Cin >> 5;
    const int nstage = SLOW ? (K + 31) / 32 : nky * nkx * nch;
    const bool wvec = ((reinterpret_cast<uintptr_t>(a.w) & 15) == 0) && ((a.ldw & 3) == 0);

    f32x4 acc[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 an0, an1, bn[NB];

    auto load_stage = [&](int it) {
        int krow0;
        if constexpr (!SLOW) {
            const int tapi = it / nch, cc = it - tapi * nch;
            const int kyi = tapi / nkx;
            const int ky = ky0 + kst * kyi, kx = kx0 + kst * (tapi - kyi * nkx);
            int iy = 0, ix = 0;
            const bool oky = src_coord(poy, ky, a.stride, a.pad, a.tmode, a.Hi, iy);
            const bool okx = src_coord(pox, kx, a.stride, a.pad, a.tmode, a.Wi, ix);
            an0 = an1 = f32x4{0.f, 0.f, 0.f, 0.f};
            if (mvalid & oky & okx) {
                const T* p = in + ((size_t)(pb * a.Hi + iy) * a.Wi + ix) * a.ldin + cc * 32 + 4 * q;
                an0 = Io<T>::ld4(p);
                an1 = Io<T>::ld4(p + 16);
            }
            krow0 = (ky * a.kw + kx) * a.Cin + cc * 32;
        } else {
            krow0 = it * 32;
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const int kk = krow0 + 16 * c + 4 * q + s;
                    float v = 0.f;
                    if (mvalid && kk < K) {
                        const int tap = kk / a.Cin, ci = kk - tap * a.Cin;
                        const int ky = tap / a.kw, kx = tap - ky * a.kw;
                        int iy = 0, ix = 0;
                        if (src_coord(poy, ky, a.stride, a.pad, a.tmode, a.Hi, iy) &&
                            src_coord(pox, kx, a.stride, a.pad, a.tmode, a.Wi, ix))
                            v = Io<T>::ld(in + ((size_t)(pb * a.Hi + iy) * a.Wi + ix) * a.ldin + ci);
                    }
                    if (c == 0) an0[s] = v; else an1[s] = v;
                }
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int row = tid / TPR + i * RPP, n = n0 + (tid % TPR) * 4;
            const int krow = krow0 + row;
            f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
            if (krow < K) {
                const float* p = a.w + (size_t)krow * a.ldw + n;
                if (wvec && n + 3 < a.N) {
                    v = *reinterpret_cast<const f32x4*>(p);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (n + e < a.N) v[e] = p[e];
                }
            }
            bn[i] = v;
        }
    };

    load_stage(0);
    for (int it = 0; it < nstage; ++it) {
#pragma unroll
        for (int i = 0; i < NB; ++i)
            *reinterpret_cast<f32x4*>(&Bs[(tid / TPR + i * RPP) * LDB + (tid % TPR) * 4]) = bn[i];
        const f32x4 ac0 = an0, ac1 = an1;
        __syncthreads();
        if (it + 1 < nstage) load_stage(it + 1);
        if constexpr (BF) {
            const float af[8] = {ac0[0], ac0[1], ac0[2], ac0[3], ac1[0], ac1[1], ac1[2], ac1[3]};
            const bf16x8 a8 = pack_bf16(af);
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                float bf[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) bf[i] = Bs[(16 * (i >> 2) + 4 * q + (i & 3)) * LDB + r + 16 * j];
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, pack_bf16(bf), acc[j], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const float av = c == 0 ? ac0[s] : ac1[s];
                    const float* brow = &Bs[(16 * c + 4 * q + s) * LDB + r];
#pragma unroll
                    for (int j = 0; j < NJ; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, brow[16 * j], acc[j], 0, 0, 0);
                }
        }
        __syncthreads();
    }

    // D[4q + i][r]: output pixel 16*wave + 4q + i of the tile, output channel 16j + r
    float s1[NJ], s2[NJ];
    if (tile_store_ok<T>(a)) {
        store_tile<T, BN>(Bs, acc, a.bias, a.accumulate, n0, a.N, [&](int p) -> T* {
            const int m = m0 + p;
            if (m >= M) return nullptr;
            size_t pl = (size_t)m;
            if (pm) {
                int b, oy, ox;
                decode_m(m, a.Ho, a.Wo, pm, b, oy, ox);
                pl = (size_t)(b * a.Ho + oy) * a.Wo + ox;
            }
            return out + pl * a.ldout;
        }, s1, s2);
        if (a.stats) tile_stats<BN>(s1, s2, Bs, n0, a.N, a.N / a.statsG, a.stats + (size_t)(m0 / (a.Ho * a.Wo)) * a.statsG * 2);
        return;
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) s1[j] = s2[j] = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + 16 * wave + 4 * q + i;
        if (m >= M) continue;
        size_t pl = (size_t)m;
        if (pm) {
            int b, oy, ox;
            decode_m(m, a.Ho, a.Wo, pm, b, oy, ox);
            pl = (size_t)(b * a.Ho + oy) * a.Wo + ox;
        }
        T* o = out + pl * a.ldout;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int n = n0 + 16 * j + r;
            if (n < a.N) {
                float v = acc[j][i];
                if (a.bias) v += a.bias[n];
                s1[j] += v;
                s2[j] += v * v;
                if (a.accumulate) v += Io<T>::ld(o + n);
                Io<T>::st(o + n, v);
            }
        }
    }
    if (a.stats) tile_stats<BN>(s1, s2, Bs, n0, a.N, a.N / a.statsG, a.stats + (size_t)(m0 / (a.Ho * a.Wo)) * a.statsG * 2);
}

// 3x3, stride 1, pad 1 with 32 input and <= 32 output channels (the full-resolution level, where a workgroup's nine
// weight taps -- 36 KB -- outweigh the 13 KB of input it stages): the weights are loaded ONCE per workgroup and stay in
// LDS while it walks over `tpw` consecutive 8x8-pixel tiles, the next tile's halo in flight during the current tile's
// multiplications. Needs the vector epilogue (tile_store_ok), which the launcher checks.
template <typename T, bool BF>
__global__ __launch_bounds__(256) void pgemm_halo_c32_kernel(const PGemmArgs a, const int ntiles, const int tpw) {
    const T* in = reinterpret_cast<const T*>(a.in);
    T* out = reinterpret_cast<T*>(a.out);
    constexpr int BN = 32, LDB = BN + 4, NJ = 2, LDH = 36;
    // BF: both operands live in LDS as bfloat16 in the order the matrix core wants them -- a lane's eight k slots
    // (channels 16*(i>>2) + 4q + (i&3)) are 16 contiguous bytes, for the weights per (tap, channel tile, lane), for the
    // halo per pixel at a pitch of 80 bytes -- so an operand is ONE 16-byte LDS read and the loop holds no conversions
    constexpr int HP = 40;   // halo pitch in bfloat16 elements
    constexpr int WS_FLOATS = BF ? 9 * NJ * 64 * 4 : 9 * 32 * LDB, HS_FLOATS = BF ? 100 * HP / 2 : 100 * LDH;
    __shared__ __attribute__((aligned(16))) float Ws[WS_FLOATS];
    __shared__ __attribute__((aligned(16))) float Hs[HS_FLOATS];
    __shared__ float Es[64 * LDB];
    __bf16* Wf = reinterpret_cast<__bf16*>(Ws);
    __bf16* Hb = reinterpret_cast<__bf16*>(Hs);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, q = lane >> 4;
    const int tw = a.Wo >> 3, th = a.Ho >> 3, tpi = tw * th;
    const int pA = 16 * wave + r, py = pA >> 3, px = pA & 7;
    const bool wvec = ((reinterpret_cast<uintptr_t>(a.w) & 15) == 0) && ((a.ldw & 3) == 0);
    const int t0 = blockIdx.x * tpw, t1 = min(t0 + tpw, ntiles);
    f32x4 hr[4];
    auto load_halo = [&](int t) {
        const int b = t / tpi, trem = t - b * tpi;
        const int oy0 = (trem / tw) * 8, ox0 = (trem % tw) * 8;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + 256 * i;
            f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
            if (idx < 800) {
                const int pix = idx >> 3, c4 = (idx & 7) * 4;
                const int hy = pix / 10, hx = pix - hy * 10;
                const int iy = oy0 - 1 + hy, ix = ox0 - 1 + hx;
                if (iy >= 0 && iy < a.Hi && ix >= 0 && ix < a.Wi)
                    v = Io<T>::ld4(in + ((size_t)(b * a.Hi + iy) * a.Wi + ix) * a.ldin + c4);
            }
            hr[i] = v;
        }
    };
    load_halo(t0);
    for (int i = tid; i < 288 * 8; i += 256) {
        const int row = i >> 3, n = (i & 7) * 4;
        f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
        const float* p = a.w + (size_t)row * a.ldw + n;
        if (wvec && n + 3 < a.N) {
            v = *reinterpret_cast<const f32x4*>(p);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (n + e < a.N) v[e] = p[e];
        }
        if constexpr (BF) {
            const int tap = row >> 5, c = row & 31, qq = (c >> 2) & 3, ii = 4 * (c >> 4) + (c & 3);
#pragma unroll
            for (int e = 0; e < 4; ++e) Wf[(((tap * NJ + ((n + e) >> 4)) * 64 + qq * 16 + ((n + e) & 15)) << 3) + ii] = (__bf16)v[e];
        } else {
            *reinterpret_cast<f32x4*>(&Ws[row * LDB + n]) = v;
        }
    }
    for (int t = t0; t < t1; ++t) {
        const int b = t / tpi, trem = t - b * tpi;
        const int oy0 = (trem / tw) * 8, ox0 = (trem % tw) * 8;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + 256 * i;
            if (idx < 800) {
                if constexpr (BF) {
                    const int c4 = (idx & 7) * 4;
                    *reinterpret_cast<bf16x4*>(&Hb[(idx >> 3) * HP + ((c4 >> 2) & 3) * 8 + 4 * (c4 >> 4)]) =
                        bf16x4{(__bf16)hr[i][0], (__bf16)hr[i][1], (__bf16)hr[i][2], (__bf16)hr[i][3]};
                } else {
                    *reinterpret_cast<f32x4*>(&Hs[(idx >> 3) * LDH + (idx & 7) * 4]) = hr[i];
                }
            }
        }
        __syncthreads();   // halo (and, the first time, the weights) visible; the previous tile's Es reads are over
        if (t + 1 < t1) load_halo(t + 1);
        f32x4 acc[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int hy = py + (a.tmode ? 2 - ky : ky);
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int hx = px + (a.tmode ? 2 - kx : kx);
                if constexpr (BF) {
                    const bf16x8 a8 = *reinterpret_cast<const bf16x8*>(&Hb[(hy * 10 + hx) * HP + q * 8]);
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        const bf16x8 b8 = *reinterpret_cast<const bf16x8*>(&Wf[(((ky * 3 + kx) * NJ + j) * 64 + lane) << 3]);
                        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, acc[j], 0, 0, 0);
                    }
                } else {
                    const float* hp = &Hs[(hy * 10 + hx) * LDH + 4 * q];
                    const f32x4 ac0 = *reinterpret_cast<const f32x4*>(hp), ac1 = *reinterpret_cast<const f32x4*>(hp + 16);
                    const float* Bt = &Ws[(ky * 3 + kx) * 32 * LDB];
#pragma unroll
                    for (int c = 0; c < 2; ++c)
#pragma unroll
                        for (int s = 0; s < 4; ++s) {
                            const float av = c == 0 ? ac0[s] : ac1[s];
                            const float* brow = &Bt[(16 * c + 4 * q + s) * LDB + r];
#pragma unroll
                            for (int j = 0; j < NJ; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, brow[16 * j], acc[j], 0, 0, 0);
                        }
                }
            }
        }
        float s1[NJ], s2[NJ];
        // the barrier inside store_tile also ends every wave's reads of Hs before the next tile's halo is written
        store_tile<T, BN>(Es, acc, a.bias, a.accumulate, 0, a.N, [&](int p) -> T* {
            return out + ((size_t)(b * a.Ho + oy0 + (p >> 3)) * a.Wo + ox0 + (p & 7)) * a.ldout;
        }, s1, s2);
        if (a.stats) tile_stats<BN>(s1, s2, Es, 0, a.N, a.N / a.statsG, a.stats + (size_t)b * a.statsG * 2);
    }
}

// 3x3, stride 1, pad 1 (forward of the ResnetBlock convolutions and, with the taps mirrored, their input gradient):
// one workgroup = 8x8 output pixels. For every 32-channel chunk the 10x10 halo of the tile is staged in LDS once and
// all nine taps read their A operand from it (1.56 x the input instead of 9 x through L2); weights as in pgemm_kernel.
template <typename T, int BN, bool BF>
__global__ __launch_bounds__(256) void pgemm_halo_kernel(const PGemmArgs a) {
    const T* in = reinterpret_cast<const T*>(a.in);
    T* out = reinterpret_cast<T*>(a.out);
    constexpr int LDB = BN + 4, NJ = BN / 16, TPR = BN / 4, RPP = 256 / TPR, NB = 32 / RPP, LDH = 36;
    // one stage = the three taps of a kernel row for 32 channels: three weight tiles in flight per round trip, a third of
    // the barriers, and the next chunk's halo rides along with the last row of the current one
    __shared__ float Bs[3 * 32 * LDB];   // >= 64 * LDB: doubles as the output tile of the epilogue
    __shared__ float Hs[100 * LDH];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, q = lane >> 4;
    const int tw = a.Wo >> 3, th = a.Ho >> 3;
    const int tile = blockIdx.x, b = tile / (tw * th), trem = tile - b * tw * th;
    const int oy0 = (trem / tw) * 8, ox0 = (trem % tw) * 8;
    const int n0 = blockIdx.y * BN;
    const int pA = 16 * wave + r, py = pA >> 3, px = pA & 7;
    const int nch = a.Cin >> 5, nstage = nch * 3;

    f32x4 acc[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // two stages of weight tiles in flight (register sets bn0 / bn1 alternate): at the low-resolution levels a launch
    // has two workgroups per CU and a stage's three MFMAs per 16 channels hide nothing, so the round trip of the weight
    // loads is what a stage costs; two in flight halve it
    f32x4 bn0[3 * NB], bn1[3 * NB], hr[4];

    // Every global load below is issued unconditionally from a clamped address and masked afterwards: a load inside a
    // branch makes the compiler wait for ALL outstanding loads (vmcnt(0)) before each use, which would turn the two
    // stages of weight tiles in flight into none. The launcher guarantees 16-byte aligned weight rows and N % 4 == 0.
    const int wn = n0 + (tid % TPR) * 4;
    const bool wok = wn + 3 < a.N;
    const float* wbase = a.w + (wok ? wn : 0);
    auto load_b = [&](int it, f32x4 (&bn)[3 * NB]) {
        const int cc = it / 3, ky = it - cc * 3;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int krow0 = (ky * 3 + kx) * a.Cin + cc * 32;
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int row = tid / TPR + i * RPP;
                bn[kx * NB + i] = *reinterpret_cast<const f32x4*>(wbase + (size_t)(krow0 + row) * a.ldw);   // masked when stored
            }
        }
    };
    // halo element of this thread: coordinates clamped into the image, value masked when outside (or past the 800 pieces)
    const T* hptr[4];
    bool hok[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = min(tid + 256 * i, 799);
        const int pix = idx >> 3, c4 = (idx & 7) * 4;
        const int hy = pix / 10, hx = pix - hy * 10;
        const int iy = oy0 - 1 + hy, ix = ox0 - 1 + hx;
        hok[i] = iy >= 0 && iy < a.Hi && ix >= 0 && ix < a.Wi;
        const int cy = min(max(iy, 0), a.Hi - 1), cx = min(max(ix, 0), a.Wi - 1);
        hptr[i] = in + ((size_t)(b * a.Hi + cy) * a.Wi + cx) * a.ldin + c4;
    }
    auto load_halo = [&](int cc) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            hr[i] = Io<T>::ld4(hptr[i] + cc * 32);   // masked when stored: a select here would wait for the load at once
        }
    };

    auto stage = [&](int it, f32x4 (&bn)[3 * NB]) {
        const int cc = it / 3, ky = it - cc * 3;
        if (ky == 0) {   // the previous chunk's last reads ended at the barrier that closed its stage
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int idx = tid + 256 * i;
                if (idx < 800) *reinterpret_cast<f32x4*>(&Hs[(idx >> 3) * LDH + (idx & 7) * 4]) = hok[i] ? hr[i] : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int i = 0; i < NB; ++i)
                *reinterpret_cast<f32x4*>(&Bs[kx * 32 * LDB + (tid / TPR + i * RPP) * LDB + (tid % TPR) * 4]) =
                    wok ? bn[kx * NB + i] : f32x4{0.f, 0.f, 0.f, 0.f};
        __syncthreads();
        load_b(min(it + 2, nstage - 1), bn);           // past the end: a repeated, unused load instead of a branch
        if (ky == 2) load_halo(min(cc + 1, nch - 1));
        const int hy = py + (a.tmode ? 2 - ky : ky);
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int hx = px + (a.tmode ? 2 - kx : kx);
            const float* hp = &Hs[(hy * 10 + hx) * LDH + 4 * q];
            const f32x4 ac0 = *reinterpret_cast<const f32x4*>(hp), ac1 = *reinterpret_cast<const f32x4*>(hp + 16);
            const float* Bt = &Bs[kx * 32 * LDB];
            if constexpr (BF) {
                const float af[8] = {ac0[0], ac0[1], ac0[2], ac0[3], ac1[0], ac1[1], ac1[2], ac1[3]};
                const bf16x8 a8 = pack_bf16(af);
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    float bf[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) bf[i] = Bt[(16 * (i >> 2) + 4 * q + (i & 3)) * LDB + r + 16 * j];
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, pack_bf16(bf), acc[j], 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        const float av = c == 0 ? ac0[s] : ac1[s];
                        const float* brow = &Bt[(16 * c + 4 * q + s) * LDB + r];
#pragma unroll
                        for (int j = 0; j < NJ; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, brow[16 * j], acc[j], 0, 0, 0);
                    }
            }
        }
        __syncthreads();
    };
    load_halo(0);
    load_b(0, bn0);
    if (nstage > 1) load_b(1, bn1);
    for (int it = 0; it < nstage; it += 2) {
        stage(it, bn0);
        if (it + 1 < nstage) stage(it + 1, bn1);
    }
    float s1[NJ], s2[NJ];
    if (tile_store_ok<T>(a)) {
        store_tile<T, BN>(Bs, acc, a.bias, a.accumulate, n0, a.N, [&](int p) -> T* {
            return out + ((size_t)(b * a.Ho + oy0 + (p >> 3)) * a.Wo + ox0 + (p & 7)) * a.ldout;
        }, s1, s2);
        if (a.stats) tile_stats<BN>(s1, s2, Bs, n0, a.N, a.N / a.statsG, a.stats + (size_t)b * a.statsG * 2);
        return;
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) s1[j] = s2[j] = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int pO = 16 * wave + 4 * q + i;
        T* o = out + ((size_t)(b * a.Ho + oy0 + (pO >> 3)) * a.Wo + ox0 + (pO & 7)) * a.ldout;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int n = n0 + 16 * j + r;
            if (n < a.N) {
                float v = acc[j][i];
                if (a.bias) v += a.bias[n];
                s1[j] += v;
                s2[j] += v * v;
                if (a.accumulate) v += Io<T>::ld(o + n);
                Io<T>::st(o + n, v);
            }
        }
    }
    if (a.stats) tile_stats<BN>(s1, s2, Bs, n0, a.N, a.N / a.statsG, a.stats + (size_t)b * a.statsG * 2);
}

// Weight gradient tile TM (weight rows = (tap, cin)) x TN (cout) per workgroup, reduced over a contiguous range of
// output pixels; 32 pixels per stage staged through LDS pixel-major for both operands (the MFMA's k index is the
// pixel). Partial sums of the pixel splits meet in dw through hardware float atomics.
template <typename T, int TM, int TN, bool SLOW, bool BF, bool DEEP = false>
__global__ __launch_bounds__(256) void wgemm_kernel(const WGemmArgs a, const int M, const int chunk) {
    const T* x = reinterpret_cast<const T*>(a.x);
    const T* dy = reinterpret_cast<const T*>(a.dy);
    constexpr int LDX = TM + 16, LDY = TN + 16, NI = TM / 32, NJ = TN / 32;
    constexpr int TPX = TM / 4, PPX = 256 / TPX, NX = 32 / PPX;   // threads per pixel row, pixels per pass, passes
    constexpr int TPY = TN / 4, PPY = 256 / TPY, NY = 32 / PPY;
    __shared__ float Xs[32 * LDX], Ys[32 * LDY];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, q = lane >> 4;
    const int wm = wave >> 1, wn = wave & 1;
    const int K = a.kh * a.kw * a.Cin;
    const int krow0 = blockIdx.y * TM, n0 = blockIdx.z * TN;
    const int ms = blockIdx.x * chunk, me = min(M, ms + chunk);
    int tky = 0, tkx = 0, ci0 = 0;
    if constexpr (!SLOW) {
        const int tap = krow0 / a.Cin;
        ci0 = krow0 - tap * a.Cin;
        tky = tap / a.kw;
        tkx = tap - tky * a.kw;
    }
    const bool yvec = ((reinterpret_cast<uintptr_t>(a.dy) & 15) == 0) && ((a.lddy & 3) == 0);
    const int hw = a.Ho * a.Wo;

    f32x4 acc[NI][NJ];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 xn[NX], yn[NY];
    const bool do_bias = a.dbias != nullptr && blockIdx.y == 0 && tid < TN;
    float bsum = 0.f;

    auto load_stage = [&](int mb) {
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const int m = mb + tid / TPX + i * PPX, c4 = (tid % TPX) * 4;
            f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
            if (m < me) {
                const int b = m / hw, rem = m - b * hw;
                const int oy = rem / a.Wo, ox = rem - oy * a.Wo;
                if constexpr (!SLOW) {
                    int iy = 0, ix = 0;
                    if (src_coord(oy, tky, a.stride, a.pad, a.tmode, a.Hi, iy) && src_coord(ox, tkx, a.stride, a.pad, a.tmode, a.Wi, ix))
                        v = Io<T>::ld4(x + ((size_t)(b * a.Hi + iy) * a.Wi + ix) * a.ldx + ci0 + c4);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int row = krow0 + c4 + e;
                        if (row < K) {
                            const int tap = row / a.Cin, ci = row - tap * a.Cin;
                            const int ky = tap / a.kw, kx = tap - ky * a.kw;
                            int iy = 0, ix = 0;
                            if (src_coord(oy, ky, a.stride, a.pad, a.tmode, a.Hi, iy) && src_coord(ox, kx, a.stride, a.pad, a.tmode, a.Wi, ix))
                                v[e] = Io<T>::ld(x + ((size_t)(b * a.Hi + iy) * a.Wi + ix) * a.ldx + ci);
                        }
                    }
                }
            }
            xn[i] = v;
        }
#pragma unroll
        for (int i = 0; i < NY; ++i) {
            const int m = mb + tid / TPY + i * PPY, n = n0 + (tid % TPY) * 4;
            f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
            if (m < me) {
                const T* p = dy + (size_t)m * a.lddy + n;
                if (yvec && n + 3 < a.N) {
                    v = Io<T>::ld4(p);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (n + e < a.N) v[e] = Io<T>::ld(p + e);
                }
            }
            yn[i] = v;
        }
    };

    // the stage in LDS: bias sums and matrix products
    auto compute = [&]() {
        if (do_bias) {
#pragma unroll 8
            for (int p = 0; p < 32; ++p) bsum += Ys[p * LDY + tid];
        }
        if constexpr (BF) {   // k slot i of lane group q = pixel 4i + q of the stage, for both operands
            bf16x8 a8[NI], b8[NJ];
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                float v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = Xs[(4 * k + q) * LDX + wm * (TM / 2) + 16 * i + r];
                a8[i] = pack_bf16(v);
            }
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                float v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = Ys[(4 * k + q) * LDY + wn * (TN / 2) + 16 * j + r];
                b8[j] = pack_bf16(v);
            }
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8[i], b8[j], acc[i][j], 0, 0, 0);
        } else {
#pragma unroll
            for (int k4 = 0; k4 < 8; ++k4) {
                const int p = 4 * k4 + q;
                float av[NI], bv[NJ];
#pragma unroll
                for (int i = 0; i < NI; ++i) av[i] = Xs[p * LDX + wm * (TM / 2) + 16 * i + r];
#pragma unroll
                for (int j = 0; j < NJ; ++j) bv[j] = Ys[p * LDY + wn * (TN / 2) + 16 * j + r];
#pragma unroll
                for (int i = 0; i < NI; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv[j], acc[i][j], 0, 0, 0);
            }
        }
    };
    if constexpr (DEEP) {
        // Two stages' operands in flight (register sets A / B alternate), loads unconditional from clamped addresses and
        // masked when stored (see wgemm_halo_kernel); the launcher guarantees vector-aligned dy rows and N % 4 == 0.
        f32x4 xb[NX], yb[NY];
        unsigned mxa = 0, mya = 0, mxb = 0, myb = 0;
        const int yn0 = n0 + (tid % TPY) * 4;
        const bool ynok = yn0 < a.N;
        auto load_u = [&](int mb, f32x4 (&X)[NX], f32x4 (&Y)[NY], unsigned& mx, unsigned& my) {
            mx = my = 0;
#pragma unroll
            for (int i = 0; i < NX; ++i) {
                const int m = mb + tid / TPX + i * PPX, c4 = (tid % TPX) * 4;
                const int mc = min(m, me - 1);
                const int b = mc / hw, rem = mc - b * hw;
                const int oy = rem / a.Wo, ox = rem - oy * a.Wo;
                int iy = 0, ix = 0;
                const bool oky = src_coord(oy, tky, a.stride, a.pad, a.tmode, a.Hi, iy);
                const bool okx = src_coord(ox, tkx, a.stride, a.pad, a.tmode, a.Wi, ix);
                const bool ok = oky & okx;
                mx |= (unsigned)(ok & (m < me)) << i;
                X[i] = Io<T>::ld4(x + ((size_t)(b * a.Hi + (ok ? iy : 0)) * a.Wi + (ok ? ix : 0)) * a.ldx + ci0 + c4);
            }
#pragma unroll
            for (int i = 0; i < NY; ++i) {
                const int m = mb + tid / TPY + i * PPY;
                my |= (unsigned)(ynok & (m < me)) << i;
                Y[i] = Io<T>::ld4(dy + (size_t)min(m, me - 1) * a.lddy + (ynok ? yn0 : 0));
            }
        };
        const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};
        const int last = ms < me ? ms + ((me - ms - 1) / 32) * 32 : ms;
        auto work = [&](int mb, f32x4 (&X)[NX], f32x4 (&Y)[NY], unsigned& mx, unsigned& my) {
#pragma unroll
            for (int i = 0; i < NX; ++i)
                *reinterpret_cast<f32x4*>(&Xs[(tid / TPX + i * PPX) * LDX + (tid % TPX) * 4]) = ((mx >> i) & 1u) ? X[i] : zero;
#pragma unroll
            for (int i = 0; i < NY; ++i)
                *reinterpret_cast<f32x4*>(&Ys[(tid / TPY + i * PPY) * LDY + (tid % TPY) * 4]) = ((my >> i) & 1u) ? Y[i] : zero;
            __syncthreads();
            load_u(min(mb + 64, last), X, Y, mx, my);   // past the end: a repeated, unused load instead of a branch
            compute();
            __syncthreads();
        };
        if (ms < me) {
            load_u(ms, xn, yn, mxa, mya);
            load_u(min(ms + 32, last), xb, yb, mxb, myb);
        }
        for (int mb = ms; mb < me; mb += 64) {
            work(mb, xn, yn, mxa, mya);
            if (mb + 32 < me) work(mb + 32, xb, yb, mxb, myb);
        }
    } else {
        if (ms < me) load_stage(ms);
        for (int mb = ms; mb < me; mb += 32) {
#pragma unroll
            for (int i = 0; i < NX; ++i) *reinterpret_cast<f32x4*>(&Xs[(tid / TPX + i * PPX) * LDX + (tid % TPX) * 4]) = xn[i];
#pragma unroll
            for (int i = 0; i < NY; ++i) *reinterpret_cast<f32x4*>(&Ys[(tid / TPY + i * PPY) * LDY + (tid % TPY) * 4]) = yn[i];
            __syncthreads();
            if (mb + 32 < me) load_stage(mb + 32);
            compute();
            __syncthreads();
        }
    }
    if (ms >= me) return;
    if (do_bias && n0 + tid < a.N) unsafeAtomicAdd(a.dbias + n0 + tid, bsum);
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int row = krow0 + wm * (TM / 2) + 16 * i + 4 * q + e;
            if (row >= K) continue;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int n = n0 + wn * (TN / 2) + 16 * j + r;
                if (n < a.N) unsafeAtomicAdd(a.dw + (size_t)row * a.lddw + n, acc[i][j][e]);
            }
        }
}

// Weight gradient of a 3x3 / stride 1 / pad 1 convolution, all nine taps of a 32 (cin) x 32 (cout) block per workgroup:
// per 8x8 pixel tile the 10x10 halo of x and the tile of dy are staged once and every tap takes its A operand from the
// halo at its own offset (the per-tap kernel above reads both tensors nine times). Wave w owns the 16x16 block
// (w >> 1, w & 1) of every tap: nine accumulators.
template <typename T, bool BF, bool DEEP>
__global__ __launch_bounds__(256) void wgemm_halo_kernel(const WGemmArgs a, const int tiles, const int per) {
    const T* x = reinterpret_cast<const T*>(a.x);
    const T* dy = reinterpret_cast<const T*>(a.dy);
    constexpr int LDX = 48, LDY = 48;
    __shared__ float Xs[100 * LDX], Ys[64 * LDY];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, q = lane >> 4;
    const int wm = wave >> 1, wn = wave & 1;
    const int ci0 = blockIdx.y * 32, n0 = blockIdx.z * 32;
    const int t0 = blockIdx.x * per, t1 = min(tiles, t0 + per);
    const int tw = a.Wo >> 3, th = a.Ho >> 3;
    const bool do_bias = a.dbias != nullptr && blockIdx.y == 0 && tid < 32;
    float bsum = 0.f;
    f32x4 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

    // DEEP: two tiles' operands in flight (register sets A / B alternate) instead of one: a workgroup walks its tiles one
    // after the other and a tile's matrix work hides nothing of the next round trip. It pays where the step is not
    // bound by memory bandwidth as a whole (bf16 activation storage: -2.3 % step time) and costs where it is (fp32
    // storage: +1 %), so the launcher picks by storage type. Every load is issued unconditionally from a clamped
    // address and masked when stored: a load inside a branch makes the compiler wait for ALL outstanding loads before
    // each use, which would turn two tiles in flight into none. The launcher guarantees N % 4 == 0.
    f32x4 xa[4], ya[2], xb[4], yb[2];
    unsigned ma = 0, mb = 0;                 // bit i: halo piece i of the set lies inside the image
    const int yn = n0 + (tid & 7) * 4;
    const bool yok = yn < a.N;
    auto load_tile = [&](int tile, f32x4 (&xr)[4], f32x4 (&yr)[2], unsigned& m) {
        const int b = tile / (tw * th), trem = tile - b * tw * th;
        const int oy0 = (trem / tw) * 8, ox0 = (trem % tw) * 8;
        m = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = min(tid + 256 * i, 799);
            const int pix = idx >> 3, c4 = (idx & 7) * 4;
            const int hy = pix / 10, hx = pix - hy * 10;
            const int iy = oy0 - 1 + hy, ix = ox0 - 1 + hx;
            const bool in = (iy >= 0) & (iy < a.Hi) & (ix >= 0) & (ix < a.Wi);
            m |= (unsigned)in << i;
            const int cy = min(max(iy, 0), a.Hi - 1), cx = min(max(ix, 0), a.Wi - 1);
            xr[i] = Io<T>::ld4(x + ((size_t)(b * a.Hi + cy) * a.Wi + cx) * a.ldx + ci0 + c4);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int pix = (tid + 256 * i) >> 3;
            yr[i] = Io<T>::ld4(dy + ((size_t)(b * a.Ho + oy0 + (pix >> 3)) * a.Wo + ox0 + (pix & 7)) * a.lddy + (yok ? yn : 0));
        }
    };
    auto work = [&](int tile, f32x4 (&xr)[4], f32x4 (&yr)[2], unsigned& m) {
        __syncthreads();
        const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + 256 * i;
            if (idx < 800) *reinterpret_cast<f32x4*>(&Xs[(idx >> 3) * LDX + (idx & 7) * 4]) = ((m >> i) & 1u) ? xr[i] : zero;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = tid + 256 * i;
            *reinterpret_cast<f32x4*>(&Ys[(idx >> 3) * LDY + (idx & 7) * 4]) = yok ? yr[i] : zero;
        }
        __syncthreads();
        load_tile(min(tile + (DEEP ? 2 : 1), t1 - 1), xr, yr, m);   // past the end: a repeated, unused load instead of a branch
        if (do_bias) {
#pragma unroll 8
            for (int p = 0; p < 64; ++p) bsum += Ys[p * LDY + tid];
        }
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            // k slot i of lane group q = pixel 32*st + 4i + q of the tile
            float yv[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) yv[i] = Ys[(32 * st + 4 * i + q) * LDY + 16 * wn + r];
            if constexpr (BF) {
                const bf16x8 b8 = pack_bf16(yv);
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const int ky = t / 3, kx = t - ky * 3;
                    float xv[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const int p = 32 * st + 4 * i + q;
                        xv[i] = Xs[(((p >> 3) + ky) * 10 + (p & 7) + kx) * LDX + 16 * wm + r];
                    }
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pack_bf16(xv), b8, acc[t], 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const int ky = t / 3, kx = t - ky * 3;
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const int p = 32 * st + 4 * i + q;
                        const float xv = Xs[(((p >> 3) + ky) * 10 + (p & 7) + kx) * LDX + 16 * wm + r];
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(xv, yv[i], acc[t], 0, 0, 0);
                    }
                }
            }
        }
    };
    if (t0 < t1) {
        load_tile(t0, xa, ya, ma);
        if constexpr (DEEP) load_tile(min(t0 + 1, t1 - 1), xb, yb, mb);
    }
    if constexpr (DEEP) {
        for (int tile = t0; tile < t1; tile += 2) {
            work(tile, xa, ya, ma);
            if (tile + 1 < t1) work(tile + 1, xb, yb, mb);
        }
    } else {
        for (int tile = t0; tile < t1; ++tile) work(tile, xa, ya, ma);
    }
    if (t0 >= t1) return;
    if (do_bias && n0 + tid < a.N) unsafeAtomicAdd(a.dbias + n0 + tid, bsum);
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int row = t * a.Cin + ci0 + 16 * wm + 4 * q + e;
            const int n = n0 + 16 * wn + r;
            if (n < a.N) unsafeAtomicAdd(a.dw + (size_t)row * a.lddw + n, acc[t][e]);
        }
}

}   // namespace

#define DMN_LAUNCH_CHECK()                                                        \
    do {                                                                          \
        hipError_t _e = hipGetLastError();                                        \
        if (_e != hipSuccess) {                                                   \
            dmn_set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(_e), __FILE__, __LINE__); \
            return DMN_ERR_HIP;                                                   \
        }                                                                         \
    } while (0)

template <typename T, bool BF> static void pgemm_dispatch(const PGemmArgs& a, int M, int pm, bool slow, int bn, dim3 grid, hipStream_t s) {
    if (bn == 64) {
        if (slow) pgemm_kernel<T, 64, true, BF><<<grid, 256, 0, s>>>(a, M, pm);
        else pgemm_kernel<T, 64, false, BF><<<grid, 256, 0, s>>>(a, M, pm);
    } else {
        if (slow) pgemm_kernel<T, 32, true, BF><<<grid, 256, 0, s>>>(a, M, pm);
        else pgemm_kernel<T, 32, false, BF><<<grid, 256, 0, s>>>(a, M, pm);
    }
}
template <typename T, bool BF> static void pgemm_halo_dispatch(const PGemmArgs& a, int bn, dim3 grid, hipStream_t s) {
    if (bn == 64) pgemm_halo_kernel<T, 64, BF><<<grid, 256, 0, s>>>(a);
    else pgemm_halo_kernel<T, 32, BF><<<grid, 256, 0, s>>>(a);
}

static int pgemm_c32_tiles() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("DMN_PGEMM_C32_TILES");   // 0 switches the weight-resident kernel off
        v = e ? atoi(e) : 8;
        if (v > 64) v = 64;
    }
    return v;
}

static int pgemm_small_below() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("DMN_PGEMM_SMALL");
        v = e ? atoi(e) : 512;
    }
    return v;
}

int pgemm_stats_supported(const PGemmArgs& a) {
    const int hw = a.Ho * a.Wo;
    const bool halo = a.kh == 3 && a.kw == 3 && a.stride == 1 && a.pad == 1 && (a.Ho & 7) == 0 && (a.Wo & 7) == 0;
    return a.statsG > 0 && a.N % a.statsG == 0 && !a.accumulate && (halo || (hw % 64 == 0 && !(a.tmode == 1 && a.stride == 2)));
}

int launch_pgemm(const PGemmArgs& a, hipStream_t s) {
    DMN_REQUIRE(a.in && a.w && a.out && a.B > 0 && a.N > 0 && a.Cin > 0, DMN_ERR_INVALID, "pgemm: bad argument");
    DMN_REQUIRE(!a.stats || pgemm_stats_supported(a), DMN_ERR_INVALID, "pgemm: statistics not supported for this shape");
    const long long Ml = (long long)a.B * a.Ho * a.Wo;
    DMN_REQUIRE(Ml < (1ll << 31) - 64, DMN_ERR_LIMIT, "pgemm: %lld output pixels", Ml);
    const int M = (int)Ml;
    const bool slow = (a.Cin & 31) != 0 || (a.ldin & 3) != 0 || (reinterpret_cast<uintptr_t>(a.in) & 15) != 0;
    const int hw = a.Ho * a.Wo;
    const int pm = (!slow && a.tmode == 1 && a.stride == 2 && (a.Ho & 1) == 0 && (a.Wo & 1) == 0 && ((hw >> 2) % 64) == 0) ? 1 : 0;
    // launches with few workgroups (the low-resolution levels) take the narrow channel tile: twice the workgroups, so
    // that a CU holds two whose load latencies overlap
    const bool is_halo = !slow && a.kh == 3 && a.kw == 3 && a.stride == 1 && a.pad == 1 && (a.Ho & 7) == 0 && (a.Wo & 7) == 0 &&
                         a.Hi == a.Ho && a.Wi == a.Wo;
    const long long wgs64 = (is_halo ? (long long)a.B * (a.Ho >> 3) * (a.Wo >> 3) : (long long)(M + 63) / 64) * ((a.N + 63) / 64);
    const int bn = (a.N > 32 && !(a.N % 32 == 0 && wgs64 < pgemm_small_below())) ? 64 : 32;
    if (is_halo && (a.N & 3) == 0 && (a.ldw & 3) == 0 && (reinterpret_cast<uintptr_t>(a.w) & 15) == 0) {
        const int vec = a.adt == 1 ? 8 : 4;
        const int ntiles = a.B * (a.Ho >> 3) * (a.Wo >> 3);
        if (a.Cin == 32 && a.N <= 32 && a.N % vec == 0 && a.ldout % vec == 0 && (reinterpret_cast<uintptr_t>(a.out) & 15) == 0 &&
            pgemm_c32_tiles() > 0) {
            // tiles per workgroup: as many as leave the launch with at least 512 workgroups (two per CU fit by LDS),
            // but never fewer than two, so that small launches run the same multi-tile path as large ones
            int tpw = pgemm_c32_tiles();
            if (tpw > ntiles / 512) tpw = ntiles / 512;
            if (tpw < 2) tpw = 2;
            const int wgs = (ntiles + tpw - 1) / tpw;
            if (a.adt == 1) pgemm_halo_c32_kernel<__bf16, true><<<wgs, 256, 0, s>>>(a, ntiles, tpw);
            else if (a.bf16) pgemm_halo_c32_kernel<float, true><<<wgs, 256, 0, s>>>(a, ntiles, tpw);
            else pgemm_halo_c32_kernel<float, false><<<wgs, 256, 0, s>>>(a, ntiles, tpw);
            DMN_LAUNCH_CHECK();
            return DMN_OK;
        }
        dim3 hgrid(ntiles, (a.N + bn - 1) / bn);
        if (a.adt == 1) pgemm_halo_dispatch<__bf16, true>(a, bn, hgrid, s);
        else if (a.bf16) pgemm_halo_dispatch<float, true>(a, bn, hgrid, s);
        else pgemm_halo_dispatch<float, false>(a, bn, hgrid, s);
        DMN_LAUNCH_CHECK();
        return DMN_OK;
    }
    dim3 grid((M + 63) / 64, (a.N + bn - 1) / bn);
    if (a.adt == 1) pgemm_dispatch<__bf16, true>(a, M, pm, slow, bn, grid, s);
    else if (a.bf16) pgemm_dispatch<float, true>(a, M, pm, slow, bn, grid, s);
    else pgemm_dispatch<float, false>(a, M, pm, slow, bn, grid, s);
    DMN_LAUNCH_CHECK();
    return DMN_OK;
}

template <typename T, bool BF, bool DEEP = false>
static void wgemm_dispatch(const WGemmArgs& a, int M, int chunk, bool slow, int tm, int tn, dim3 grid, hipStream_t s) {
    if (slow) {
        if (tn == 64) wgemm_kernel<T, 32, 64, true, BF><<<grid, 256, 0, s>>>(a, M, chunk);
        else wgemm_kernel<T, 32, 32, true, BF><<<grid, 256, 0, s>>>(a, M, chunk);
    } else if (tm == 64) {
        if (tn == 64) wgemm_kernel<T, 64, 64, false, BF, DEEP><<<grid, 256, 0, s>>>(a, M, chunk);
        else wgemm_kernel<T, 64, 32, false, BF, DEEP><<<grid, 256, 0, s>>>(a, M, chunk);
    } else {
        if (tn == 64) wgemm_kernel<T, 32, 64, false, BF, DEEP><<<grid, 256, 0, s>>>(a, M, chunk);
        else wgemm_kernel<T, 32, 32, false, BF, DEEP><<<grid, 256, 0, s>>>(a, M, chunk);
    }
}

// Workgroups a weight-gradient launch aims for: every workgroup ends with one float atomic per element of its output
// tile, so more pixel splits buy occupancy with atomic traffic (2048 splits of a 64x64 tile are 8 M atomics). These
// launches run beside the chain of pixel kernels and the step ends when the last of them does, so the step time follows
// this number closely (measured, ms per step at 128 / 256 / 320 / 384 / 512 / 1024: fp32 activations 11.6 / 10.42 /
// 10.44 / 10.46 / 10.75 / 11.8; bf16 activations, whose 8-byte loads leave a workgroup with half the bytes in flight,
// 14.6 / 10.63 / 10.16 / 10.18 / 10.38 / 11.2 with one stage in flight; with two, as the kernels run now, 160 / 192 /
// 224 / 256 / 288 / 448 / 512: 9.98 / 9.53 / 9.53 / 9.59 / 9.61 / 9.64 / 9.75).
static int wgemm_wg_target(int adt) {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("DMN_WGEMM_WGS");
        v = e ? atoi(e) : 0;
        if (v && v < 64) v = 64;
    }
    return v ? v : (adt == 1 ? 224 : 256);
}
// ... and for launches whose whole weight block is one or two tiles (the 32-channel layers at full resolution): every
// split adds to the same one or two thousand addresses, and that contention costs more than the occupancy buys
// (measured, ms per step at 64 / 128 / 192 / 256 / 512 / 1024 / 4096 for these launches alone: fp32 activations 10.70 /
// 10.41 / 10.38 / 10.43 / 10.61 / 11.03 / 12.9; bf16 activations, two stages in flight, 128 / 192 / 256 / 320: 9.46 / 9.47 /
// 9.51 / 9.54).
static int wgemm_small_target(int adt, int tiles_kn) {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("DMN_WGEMM_SMALL_WGS");
        v = e ? atoi(e) : 0;
        if (v && v < 32) v = 32;
    }
    if (tiles_kn > 2) return wgemm_wg_target(adt);
    (void)adt;
    return v ? v : 192;
}

// ... and for the element-wise gather path (the 3-channel ingest convolution): its output gradient is the last tensor
// of the backward pass, so this launch runs alone at the end of the step, where only its own length counts
static int wgemm_slow_target() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("DMN_WGEMM_SLOW_WGS");
        v = e ? atoi(e) : 1024;   // ms per step at 256 / 512 / 1024 / 2048 / 4096 / 8192: 10.49 / 10.39 / 10.35 / 10.38 / 10.37 / 10.44
        if (v < 64) v = 64;
    }
    return v;
}

int launch_wgemm(const WGemmArgs& a, hipStream_t s) {
    DMN_REQUIRE(a.x && a.dy && a.dw && a.B > 0 && a.N > 0 && a.Cin > 0, DMN_ERR_INVALID, "wgemm: bad argument");
    const long long Ml = (long long)a.B * a.Ho * a.Wo;
    DMN_REQUIRE(Ml < (1ll << 31) - 64, DMN_ERR_LIMIT, "wgemm: %lld output pixels", Ml);
    const int M = (int)Ml;
    const int K = a.kh * a.kw * a.Cin;
    const bool big_m = (a.Cin % 64) == 0, big_n = a.N > 32;
    const bool slow = (a.Cin & 31) != 0 || (a.ldx & 3) != 0 || (reinterpret_cast<uintptr_t>(a.x) & 15) != 0;
    if (!slow && a.kh == 3 && a.kw == 3 && a.stride == 1 && a.pad == 1 && a.tmode == 0 && (a.Ho & 7) == 0 && (a.Wo & 7) == 0 &&
        a.Hi == a.Ho && a.Wi == a.Wo && (a.lddy & 3) == 0 && (a.N & 3) == 0 && (reinterpret_cast<uintptr_t>(a.dy) & 15) == 0) {
        const int tiles = a.B * (a.Ho >> 3) * (a.Wo >> 3), cit = a.Cin / 32, cot = (a.N + 31) / 32;
        int ns = wgemm_small_target(a.adt, cit * cot) / (cit * cot);
        if (ns > tiles) ns = tiles;
        if (ns < 1) ns = 1;
        const int per = (tiles + ns - 1) / ns;
        ns = (tiles + per - 1) / per;
        DMN_REQUIRE(cit <= 65535 && cot <= 65535, DMN_ERR_LIMIT, "wgemm: grid %d x %d", cit, cot);
        if (a.adt == 1) wgemm_halo_kernel<__bf16, true, true><<<dim3(ns, cit, cot), 256, 0, s>>>(a, tiles, per);
        else if (a.bf16) wgemm_halo_kernel<float, true, false><<<dim3(ns, cit, cot), 256, 0, s>>>(a, tiles, per);
        else wgemm_halo_kernel<float, false, false><<<dim3(ns, cit, cot), 256, 0, s>>>(a, tiles, per);
        DMN_LAUNCH_CHECK();
        return DMN_OK;
    }
    const int tm = (!slow && big_m) ? 64 : 32, tn = big_n ? 64 : 32;
    const int rt = (K + tm - 1) / tm, ct = (a.N + tn - 1) / tn;
    int nsplit = (slow ? wgemm_slow_target() : wgemm_small_target(a.adt, rt * ct)) / (rt * ct);
    const int max_split = (M + 127) / 128;
    if (nsplit > max_split) nsplit = max_split;
    if (nsplit < 1) nsplit = 1;
    // bf16 activation storage: no workgroup walks more than 64 stages of 32 pixels (the 4x4 down / up-sampling layers
    // at full resolution: 16 weight tiles x 14 splits walked ~150 each on a side stream), up to 1024 workgroups per
    // launch; measured 9.44 against 9.50-9.58 ms per step, no difference with fp32 storage (tuning aid:
    // DMN_WGEMM_MAX_STAGES, 0 = no bound)
    {
        static int max_stages = -1;
        if (max_stages < 0) {
            const char* e = getenv("DMN_WGEMM_MAX_STAGES");
            max_stages = e ? atoi(e) : 64;
        }
        if (max_stages > 0 && a.adt == 1 && !slow && rt * ct > 2) {
            int want = (M + 32 * max_stages - 1) / (32 * max_stages);
            const int cap = 1024 / (rt * ct) > 1 ? 1024 / (rt * ct) : 1;
            if (want > cap) want = cap;
            if (want > nsplit) nsplit = want;
        }
    }
    int chunk = (M + nsplit - 1) / nsplit;
    chunk = (chunk + 31) & ~31;
    nsplit = (M + chunk - 1) / chunk;
    dim3 grid(nsplit, rt, ct);
    DMN_REQUIRE(rt <= 65535 && ct <= 65535, DMN_ERR_LIMIT, "wgemm: grid %d x %d", rt, ct);
    // bf16 activation storage: two stages in flight where every dy row can be fetched as whole vectors (see the kernel)
    const bool deep = a.adt == 1 && !slow && (a.N & 3) == 0 && (a.lddy & 3) == 0 && (reinterpret_cast<uintptr_t>(a.dy) & 15) == 0;
    if (deep) wgemm_dispatch<__bf16, true, true>(a, M, chunk, slow, tm, tn, grid, s);
    else if (a.adt == 1) wgemm_dispatch<__bf16, true>(a, M, chunk, slow, tm, tn, grid, s);
    else if (a.bf16) wgemm_dispatch<float, true>(a, M, chunk, slow, tm, tn, grid, s);
    else wgemm_dispatch<float, false>(a, M, chunk, slow, tm, tn, grid, s);
    DMN_LAUNCH_CHECK();
    return DMN_OK;
}
