// Implicit-GEMM convolution for gfx950 (CDNA4): NHWC activations, LDS-staged input halo tile and
// weight slab, 16x16 MFMA (bf16 x32 or exact-f32 x4) with fp32 accumulation.
//
// One kernel serves every convolution of the reference U-Net except the 3-channel 7x7 ingest and the
// 3/6-channel tail:
//   3x3 s1 p1   Block.proj                      diffusion_model_nemo/parts/convnext.py:11
//   1x1         res_conv / to_qkv / to_out      parts/convnext.py:76, parts/mha.py:13-14,39-41
//   4x4 s2 p1   Downsample                      diffusion_model_nemo/utils.py:81-82
//   4x4 s2 p1^T Upsample, as four 2x2 sub-pixel phases   utils.py:77-78
// and the channel concatenation of the decoder (modules/unet.py:162) is read from two base pointers
// instead of being materialised.
//
// GEMM orientation: D[cout][pixel] = W[cout][k] * X[k][pixel]; the weights are the MFMA "A" operand
// and the pixels the "B" operand, so each lane ends up with 4 CONSECUTIVE output channels of one pixel
// and the NHWC store is an 8-byte (bf16) / 16-byte (f32) vector per lane.
//
// Fusions: GroupNorm + SiLU + time-embedding bias of the PREVIOUS Block applied while the input tile is
// staged (parts/convnext.py:35-41,81-85), bias, optional residual add, and partial GroupNorm statistics
// of the output for the next consumer.
#include "dmn_kernels.h"
#include "dmn_norm.h"

namespace {

constexpr int kRowBytes = 80;  // 64 data bytes + 16 pad: consecutive pixels land on distinct 16-B LDS slots

template <typename T, int TH, int TW, int BN, int WM, int WN>
__global__ void __launch_bounds__(256) conv_igemm_kernel(const ConvArgs a) {
    static_assert(WM * WN == 4, "four waves per workgroup");
    constexpr int BM = TH * TW;
    constexpr int MT = BM / (16 * WM);
    constexpr int NT = BN / (16 * WN);
    static_assert(MT >= 1 && NT >= 1, "tile too small for the wave grid");
    constexpr int KCH = ElemTraits<T>::kChunk;
    constexpr int PER = ElemTraits<T>::kPer16B;
    using V = decltype(Vec16<T>::v);

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int l15 = lane & 15, lq = lane >> 4;

    const int tiles_x = (a.Wo + TW - 1) / TW;
    const int tile_y = blockIdx.x / tiles_x, tile_x = blockIdx.x % tiles_x;
    const int n0 = blockIdx.y * BN;
    const int b = blockIdx.z / a.nphase, ph = blockIdx.z % a.nphase;
    const ConvPhase P = a.phase[ph];
    const int S = a.stride;
    const int IH = (TH - 1) * S + a.ext_y, IW = (TW - 1) * S + a.ext_x;
    const int iy0 = tile_y * TH * S - P.pad_y, ix0 = tile_x * TW * S - P.pad_x;
    const int Cin = a.C0 + a.C1;
    const int ntaps = a.ntaps;

    char* sA = smem;
    char* sW = sA + IH * IW * kRowBytes;
    float* sCoef = reinterpret_cast<float*>(sW + ntaps * BN * kRowBytes);
    const bool pro = a.pro.stats != nullptr;
    float* sRed = sCoef + (pro ? 3 * Cin : 0);  // >= max(2*G, 2*WM*BN/4) floats

    if (pro) norm_coefs_to_lds(a.pro, b, Cin, sCoef, sRed);

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // per-lane LDS pixel index of each m-tile's pixel (before the tap offset)
    int pixbase[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int p = (wm * MT + mt) * 16 + l15;
        pixbase[mt] = ((p / TW) * S) * IW + (p % TW) * S;
    }

    const int nchunks = Cin / KCH;
    const T* wphase = static_cast<const T*>(a.w) + P.w_off;
    for (int ch = 0; ch < nchunks; ++ch) {
        const int c0 = ch * KCH;
        const T* src;
        int cs, Csrc;
        if (c0 < a.C0) {
            src = static_cast<const T*>(a.in0);
            cs = c0;
            Csrc = a.C0;
        } else {
            src = static_cast<const T*>(a.in1);
            cs = c0 - a.C0;
            Csrc = a.C1;
        }
        // ---- stage the input halo tile: IH*IW pixels x 64 bytes -----------------------------------
        for (int i = tid; i < IH * IW * 4; i += 256) {
            const int pix = i >> 2, part = i & 3;
            const int ly = pix / IW, lx = pix - ly * IW;
            const int gy = iy0 + ly, gx = ix0 + lx;
            Vec16<T> v;
#pragma unroll
            for (int j = 0; j < PER; ++j) v.set(j, 0.f);
            if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) {
                v.v = *reinterpret_cast<const V*>(src + ((size_t)(b * a.H + gy) * a.W + gx) * Csrc + cs + part * PER);
                if (pro) {
                    const int cb = c0 + part * PER;
#pragma unroll
                    for (int j = 0; j < PER; ++j) {
                        float y = v.get(j) * sCoef[cb + j] + sCoef[Cin + cb + j];
                        if (a.pro.silu) y = (sizeof(T) == 4) ? silu_precise(y) : silu_f(y);
                        v.set(j, y + sCoef[2 * Cin + cb + j]);
                    }
                }
            }
            *reinterpret_cast<V*>(sA + pix * kRowBytes + part * 16) = v.v;
        }
        // ---- stage the weight slab: BN couts x ntaps x 64 bytes, contiguous in the packed layout ---
        const T* wsrc = wphase + ((size_t)ch * a.Cout + n0) * ntaps * KCH;
        for (int i = tid; i < ntaps * BN * 4; i += 256) {
            const int row = i >> 2, part = i & 3;
            const int nl = row / ntaps, tap = row - nl * ntaps;
            *reinterpret_cast<V*>(sW + (tap * BN + nl) * kRowBytes + part * 16) =
                *reinterpret_cast<const V*>(wsrc + (size_t)row * KCH + part * PER);
        }
        __syncthreads();
        // ---- MFMA over the taps of this channel chunk ----------------------------------------------
        for (int t = 0; t < ntaps; ++t) {
            const int toff = a.tap_dy[t] * IW + a.tap_dx[t];
            Vec16<T> wf[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int nl = (wn * NT + nt) * 16 + l15;
                wf[nt].v = *reinterpret_cast<const V*>(sW + (t * BN + nl) * kRowBytes + lq * 16);
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                Vec16<T> xf;
                xf.v = *reinterpret_cast<const V*>(sA + (pixbase[mt] + toff) * kRowBytes + lq * 16);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) mma_step(wf[nt], xf, acc[mt][nt]);
            }
        }
        __syncthreads();
    }

    // ---- epilogue: bias, residual, statistics, NHWC store ---------------------------------------------
    const bool want_stats = a.epi.stats != nullptr;
    float ssum[NT], ssq[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) ssum[nt] = ssq[nt] = 0.f;
    T* outp = static_cast<T*>(a.out);
    const T* resp = static_cast<const T*>(a.res);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int p = (wm * MT + mt) * 16 + l15;
        const int oyt = tile_y * TH + p / TW, oxt = tile_x * TW + p % TW;
        const bool valid = oyt < a.Ho && oxt < a.Wo;
        const int oy = oyt * a.out_sy + P.out_oy, ox = oxt * a.out_sx + P.out_ox;
        const size_t obase = ((size_t)(b * a.out_H + oy) * a.out_W + ox) * a.Cout;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int n = n0 + (wn * NT + nt) * 16 + lq * 4;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = acc[mt][nt][r] + (a.bias ? a.bias[n + r] : 0.f);
            if (valid) {
                if (want_stats) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        ssum[nt] += v[r];
                        ssq[nt] += v[r] * v[r];
                    }
                }
                if (resp) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] += to_f32(resp[obase + n + r]);
                }
                if constexpr (sizeof(T) == 4) {
                    *reinterpret_cast<f32x4*>(outp + obase + n) = f32x4{v[0], v[1], v[2], v[3]};
                } else {
                    bf16x4 o;
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = (__bf16)v[r];
                    *reinterpret_cast<bf16x4*>(outp + obase + n) = o;
                }
            }
        }
    }
    if (want_stats) {
        // reduce over the 16 pixel lanes; lane (lane&15)==0 of each quarter holds the sums of a 4-channel quad
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) {
                ssum[nt] += __shfl_xor(ssum[nt], o, 64);
                ssq[nt] += __shfl_xor(ssq[nt], o, 64);
            }
            if (l15 == 0) {
                const int quad = (wn * NT + nt) * 4 + lq;  // local quad index in [0, BN/4)
                sRed[(wm * (BN / 4) + quad) * 2] = ssum[nt];
                sRed[(wm * (BN / 4) + quad) * 2 + 1] = ssq[nt];
            }
        }
        __syncthreads();
        const int G = a.epi.G;
        const int cpg = a.Cout / G;
        const int ngl = cpg >= BN ? 1 : BN / cpg;      // groups this workgroup touches
        const int qpg = (cpg >= BN ? BN : cpg) / 4;    // quads per touched group inside this workgroup
        const int sub = cpg > BN ? cpg / BN : 1;       // workgroups (along N) sharing one group
        if (tid < ngl) {
            float s = 0.f, ss = 0.f;
            for (int w = 0; w < WM; ++w)
                for (int q = 0; q < qpg; ++q) {
                    s += sRed[(w * (BN / 4) + tid * qpg + q) * 2];
                    ss += sRed[(w * (BN / 4) + tid * qpg + q) * 2 + 1];
                }
            const int g = n0 / cpg + tid;
            const int slot = blockIdx.x * sub + (cpg > BN ? (n0 % cpg) / BN : 0);
            float* dst = a.epi.stats + ((size_t)(b * G + g) * a.epi.slots + slot) * 2;
            dst[0] = s;
            dst[1] = ss;
        }
    }
}

struct TileCfg {
    int th, tw, bn;
};

template <typename T, int TH, int TW, int BN, int WM, int WN>
int launch_cfg(const ConvArgs& a, size_t lds, hipStream_t s) {
    const int tiles = ((a.Ho + TH - 1) / TH) * ((a.Wo + TW - 1) / TW);
    dim3 grid(tiles, a.Cout / BN, a.B * a.nphase);
    hipLaunchKernelGGL((conv_igemm_kernel<T, TH, TW, BN, WM, WN>), grid, dim3(256), lds, s, a);
    DMN_HIP_CHECK(hipGetLastError());
    return DMN_OK;
}

TileCfg pick_cfg(const ConvArgs& a) {
    TileCfg c;
    if (a.cfg_tile && a.cfg_bn) {
        c.th = c.tw = a.cfg_tile;
        c.bn = a.cfg_bn;
        return c;
    }
    const bool big = a.stride == 1 && a.Ho >= 16 && a.Wo >= 16;
    c.th = c.tw = big ? 16 : 8;
    c.bn = (a.Cout % 64 == 0) ? 64 : 32;
    auto blocks = [&](const TileCfg& t) {
        return (long long)((a.Ho + t.th - 1) / t.th) * ((a.Wo + t.tw - 1) / t.tw) * (a.Cout / t.bn) * a.B * a.nphase;
    };
    // keep at least ~2 workgroups per CU when the problem allows it
    if (blocks(c) < 512 && c.bn == 64) c.bn = 32;
    if (blocks(c) < 512 && c.th == 16) {
        c.th = c.tw = 8;
        if (a.Cout % 64 == 0 && blocks(TileCfg{8, 8, 64}) >= 512) c.bn = 64;
    }
    return c;
}

size_t lds_bytes(const ConvArgs& a, const TileCfg& c) {
    const int IH = (c.th - 1) * a.stride + a.ext_y, IW = (c.tw - 1) * a.stride + a.ext_x;
    size_t n = (size_t)IH * IW * kRowBytes + (size_t)a.ntaps * c.bn * kRowBytes;
    const int Cin = a.C0 + a.C1;
    if (a.pro.stats) n += (size_t)3 * Cin * 4;
    size_t red = 2 * 4 * (c.bn / 4);
    if (a.pro.stats && (size_t)2 * a.pro.G > red) red = 2 * a.pro.G;
    n += red * 4 + 64;
    return n;
}

template <typename T>
int dispatch(const ConvArgs& a, const TileCfg& c, size_t lds, hipStream_t s) {
    if (c.th == 16 && c.bn == 32) return launch_cfg<T, 16, 16, 32, 4, 1>(a, lds, s);
    if (c.th == 16 && c.bn == 64) return launch_cfg<T, 16, 16, 64, 4, 1>(a, lds, s);
    if (c.th == 8 && c.bn == 32) return launch_cfg<T, 8, 8, 32, 2, 2>(a, lds, s);
    return launch_cfg<T, 8, 8, 64, 2, 2>(a, lds, s);
}

template <typename T, int TH, int TW, int BN, int WM, int WN>
int set_attr() {
    DMN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_kernel<T, TH, TW, BN, WM, WN>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    return DMN_OK;
}

}  // namespace

int conv_init_attributes() {
    int rc;
    if ((rc = set_attr<float, 16, 16, 32, 4, 1>())) return rc;
    if ((rc = set_attr<float, 16, 16, 64, 4, 1>())) return rc;
    if ((rc = set_attr<float, 8, 8, 32, 2, 2>())) return rc;
    if ((rc = set_attr<float, 8, 8, 64, 2, 2>())) return rc;
    if ((rc = set_attr<__bf16, 16, 16, 32, 4, 1>())) return rc;
    if ((rc = set_attr<__bf16, 16, 16, 64, 4, 1>())) return rc;
    if ((rc = set_attr<__bf16, 8, 8, 32, 2, 2>())) return rc;
    if ((rc = set_attr<__bf16, 8, 8, 64, 2, 2>())) return rc;
    return DMN_OK;
}

int conv_epi_slots(const ConvArgs& a) {
    const TileCfg c = pick_cfg(a);
    const int tiles = ((a.Ho + c.th - 1) / c.th) * ((a.Wo + c.tw - 1) / c.tw);
    const int cpg = a.Cout / a.epi.G;
    return tiles * (cpg > c.bn ? cpg / c.bn : 1);
}

int launch_conv(ConvArgs& a, hipStream_t s) {
    const int kch = a.dtype == DMN_F32 ? 16 : 32;
    const int Cin = a.C0 + a.C1;
    DMN_REQUIRE(a.C0 % kch == 0 && a.C1 % kch == 0 && Cin > 0, DMN_ERR_INVALID,
                "conv: input channels (%d + %d) must be multiples of %d", a.C0, a.C1, kch);
    DMN_REQUIRE(a.Cout % 32 == 0, DMN_ERR_INVALID, "conv: Cout=%d must be a multiple of 32", a.Cout);
    DMN_REQUIRE(a.ntaps >= 1 && a.ntaps <= 16 && a.nphase >= 1 && a.nphase <= 4, DMN_ERR_INVALID, "conv: bad tap/phase count");
    DMN_REQUIRE(!(a.pro.stats && a.C1), DMN_ERR_INVALID, "conv: input normalisation needs a single source");
    const TileCfg c = pick_cfg(a);
    if (a.pro.stats) DMN_REQUIRE(Cin % a.pro.G == 0, DMN_ERR_INVALID, "conv: Cin %% groups != 0");
    if (a.epi.stats) {
        DMN_REQUIRE(a.nphase == 1 && a.Cout % a.epi.G == 0 && (a.Cout / a.epi.G) % 4 == 0, DMN_ERR_INVALID,
                    "conv: output statistics need one phase and >= 4 channels per group");
        const int cpg = a.Cout / a.epi.G;
        DMN_REQUIRE(cpg >= c.bn ? cpg % c.bn == 0 : c.bn % cpg == 0, DMN_ERR_INVALID, "conv: group size %d vs tile %d", cpg, c.bn);
        a.epi.slots = conv_epi_slots(a);
    }
    const size_t lds = lds_bytes(a, c);
    DMN_REQUIRE(lds <= 160 * 1024, DMN_ERR_INVALID, "conv: LDS request %zu too large", lds);
    if (a.dtype == DMN_F32) return dispatch<float>(a, c, lds, s);
    return dispatch<__bf16>(a, c, lds, s);
}
