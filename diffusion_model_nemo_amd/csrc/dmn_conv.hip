// Implicit-GEMM convolution for gfx950 (CDNA4): NHWC activations, LDS-staged input halo tile and
// weight slab, 16x16 MFMA (bf16 x32 or exact-f32 x4) with fp32 accumulation.
//
// One kernel template serves every convolution of the reference U-Net except the 3-channel 7x7 ingest
// and the 3/6-channel tail:
//   K3X3    3x3 s1 p1   Block.proj                      diffusion_model_nemo/parts/convnext.py:11
//   K1X1    1x1         res_conv / to_qkv / to_out      parts/convnext.py:76, parts/mha.py:13-14,39-41
//   K4X4S2  4x4 s2 p1   Downsample                      diffusion_model_nemo/utils.py:81-82
//   KT2X2   4x4 s2 p1^T Upsample, as four 2x2 sub-pixel phases   utils.py:77-78
// The channel concatenation of the decoder (modules/unet.py:162) is read from two base pointers instead
// of being materialised.
//
// GEMM orientation: D[cout][pixel] = W[cout][k] * X[k][pixel]; the weights are the MFMA "A" operand and
// the pixels the "B" operand, so each lane ends up with 4 CONSECUTIVE output channels of one pixel and the
// NHWC store is an 8-byte (bf16) / 16-byte (f32) vector per lane.
//
// Pipeline per workgroup (256 threads, 4 waves): the K dimension is walked in channel chunks of 64 bytes
// per pixel; while the MFMAs of chunk c run out of LDS, the global loads of chunk c+1 (input halo tile and
// weight slab) are already in flight into registers and are written to LDS behind the next barrier. All tap
// offsets, tile extents and per-thread staging slots are compile-time constants.
//
// Fusions: GroupNorm + SiLU + time-embedding bias of the PREVIOUS Block applied while the input tile is
// written to LDS (parts/convnext.py:35-41,81-85), bias, optional residual add, and partial GroupNorm
// statistics of the output for the next consumer.
#include "dmn_kernels.h"
#include "dmn_norm.h"

namespace {

constexpr int kRowBytes = 96;  // 64 data bytes + 32 pad: the 16-lane groups of a ds_read_b128 over consecutive rows
                               // (slot = 6 * row + quarter mod 16) hit 16 distinct 16-byte slots of the 256-byte bank row

template <int KIND> struct KindTraits;
template <> struct KindTraits<DMN_K3X3> { static constexpr int NTAPS = 9, EXT = 3, S = 1, KW = 3; };
template <> struct KindTraits<DMN_K1X1> { static constexpr int NTAPS = 1, EXT = 1, S = 1, KW = 1; };
template <> struct KindTraits<DMN_K4X4S2> { static constexpr int NTAPS = 16, EXT = 4, S = 2, KW = 4; };
template <> struct KindTraits<DMN_KT2X2> { static constexpr int NTAPS = 4, EXT = 2, S = 1, KW = 2; };

template <typename T, int KIND, int TH, int TW, int BN, int WM, int WN>
__global__ void __launch_bounds__(256) conv_igemm_kernel(const ConvArgs a) {
    using KT = KindTraits<KIND>;
    static_assert(WM * WN == 4, "four waves per workgroup");
    constexpr int NTAPS = KT::NTAPS, S = KT::S;
    constexpr int BM = TH * TW;
    constexpr int MT = BM / (16 * WM);
    constexpr int NT = BN / (16 * WN);
    static_assert(MT >= 1 && NT >= 1, "tile too small for the wave grid");
    constexpr int KCH = ElemTraits<T>::kChunk;
    constexpr int PER = ElemTraits<T>::kPer16B;
    constexpr int IH = (TH - 1) * S + KT::EXT, IW = (TW - 1) * S + KT::EXT;
    constexpr int A_UNITS = IH * IW * 4;            // 16-byte staging units of the input tile per chunk
    constexpr int W_UNITS = NTAPS * BN * 4;         // ... of the weight slab
    constexpr int NUA = (A_UNITS + 255) / 256;
    constexpr int NUW = (W_UNITS + 255) / 256;
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
    const int iy0 = tile_y * TH * S - P.pad_y, ix0 = tile_x * TW * S - P.pad_x;
    const int Cin = a.C0 + a.C1;

    char* sA = smem;
    char* sW = sA + IH * IW * kRowBytes;
    float* sCoef = reinterpret_cast<float*>(sW + NTAPS * BN * kRowBytes);
    const bool pro = a.pro.stats != nullptr;
    float* sRed = sCoef + (pro ? 3 * Cin : 0);  // >= max(2*G, 2*WM*BN/4) floats

    if (pro) norm_coefs_to_lds(a.pro, b, Cin, sCoef, sRed);

    // ---- per-thread staging slots (fixed for the whole K loop) -------------------------------------
    // input: unit u -> halo pixel (u >> 2), 16-byte part (u & 3) == (tid & 3)
    int gpix[NUA];   // linear pixel index in the image batch, or -1 outside the image / beyond the tile
#pragma unroll
    for (int j = 0; j < NUA; ++j) {
        const int u = tid + 256 * j;
        const int pix = u >> 2;
        const int ly = pix / IW, lx = pix - ly * IW;
        const int gy = iy0 + ly, gx = ix0 + lx;
        const bool ok = u < A_UNITS && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
        gpix[j] = ok ? (b * a.H + gy) * a.W + gx : -1;
    }
    const int part = tid & 3;

    V ra[NUA], rw[NUW];
    const T* wphase = static_cast<const T*>(a.w) + P.w_off;

    auto issue_loads = [&](int ch) {
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
#pragma unroll
        for (int j = 0; j < NUA; ++j) {
            // out-of-image slots read pixel 0 (always mapped) and are zeroed when written to LDS: no branch here
            const int gp = gpix[j] < 0 ? 0 : gpix[j];
            ra[j] = *reinterpret_cast<const V*>(src + (size_t)gp * Csrc + cs + part * PER);
        }
        // weight slab of this chunk: [tap][Cout][KCH]; the workgroup takes couts n0 .. n0+BN of every tap
        const T* wsrc = wphase + (size_t)ch * NTAPS * a.Cout * KCH;
#pragma unroll
        for (int j = 0; j < NUW; ++j) {
            const int u = tid + 256 * j;
            if (NUW * 256 == W_UNITS || u < W_UNITS) {
                const int row = u >> 2;
                const int tap = row / BN, nl = row % BN;
                rw[j] = *reinterpret_cast<const V*>(wsrc + ((size_t)tap * a.Cout + n0 + nl) * KCH + part * PER);
            }
        }
    };

    auto write_lds = [&](int ch) {
        float ca[PER], cb[PER], ct[PER];
        if (pro) {
            const int cbase = ch * KCH + part * PER;
#pragma unroll
            for (int e = 0; e < PER; ++e) {
                ca[e] = sCoef[cbase + e];
                cb[e] = sCoef[Cin + cbase + e];
                ct[e] = sCoef[2 * Cin + cbase + e];
            }
        }
#pragma unroll
        for (int j = 0; j < NUA; ++j) {
            const int u = tid + 256 * j;
            if (NUA * 256 == A_UNITS || u < A_UNITS) {
                Vec16<T> v;
                v.v = ra[j];
                const bool inside = gpix[j] >= 0;
                if (pro) {
#pragma unroll
                    for (int e = 0; e < PER; ++e) {
                        float y = v.get(e) * ca[e] + cb[e];
                        if (a.pro.silu) y = (sizeof(T) == 4) ? silu_precise(y) : silu_f(y);
                        v.set(e, inside ? y + ct[e] : 0.f);
                    }
                } else if (!inside) {
#pragma unroll
                    for (int e = 0; e < PER; ++e) v.set(e, 0.f);
                }
                *reinterpret_cast<V*>(sA + (u >> 2) * kRowBytes + part * 16) = v.v;
            }
        }
#pragma unroll
        for (int j = 0; j < NUW; ++j) {
            const int u = tid + 256 * j;
            if (NUW * 256 == W_UNITS || u < W_UNITS)
                *reinterpret_cast<V*>(sW + (u >> 2) * kRowBytes + part * 16) = rw[j];
        }
    };

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // per-lane LDS byte offsets of the fragments (before the compile-time tap offset)
    int xoff[MT], woff[NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int p = (wm * MT + mt) * 16 + l15;
        xoff[mt] = (((p / TW) * S) * IW + (p % TW) * S) * kRowBytes + lq * 16;
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) woff[nt] = ((wn * NT + nt) * 16 + l15) * kRowBytes + lq * 16;

    const int nchunks = Cin / KCH;
    issue_loads(0);
    write_lds(0);
    __syncthreads();
    for (int ch = 0; ch < nchunks; ++ch) {
        const bool more = ch + 1 < nchunks;
        if (more) issue_loads(ch + 1);
        {
            // software-pipelined over the taps: the fragments of tap t+1 are read while tap t multiplies
            constexpr int KW = KT::KW;
            Vec16<T> wf[2][NT], xf[2][MT];
            auto load_frags = [&](int t, int buf) {
                const int toff = ((t / KW) * IW + (t % KW)) * kRowBytes;   // an immediate after unrolling
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    wf[buf][nt].v = *reinterpret_cast<const V*>(sW + t * BN * kRowBytes + woff[nt]);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) xf[buf][mt].v = *reinterpret_cast<const V*>(sA + xoff[mt] + toff);
            };
            load_frags(0, 0);
#pragma unroll
            for (int t = 0; t < NTAPS; ++t) {
                if (t + 1 < NTAPS) load_frags(t + 1, (t + 1) & 1);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) mma_step(wf[t & 1][nt], xf[t & 1][mt], acc[mt][nt]);
            }
        }
        if (more) {
            __syncthreads();
            write_lds(ch + 1);
            __syncthreads();
        }
    }

    // ---- epilogue: bias, residual, statistics, NHWC store ---------------------------------------------
    const bool want_stats = a.epi.stats != nullptr;
    float ssum[NT], ssq[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) ssum[nt] = ssq[nt] = 0.f;
    T* outp = static_cast<T*>(a.out);
    const T* resp = static_cast<const T*>(a.res);
    float bias[NT][4];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int n = n0 + (wn * NT + nt) * 16 + lq * 4;
#pragma unroll
        for (int r = 0; r < 4; ++r) bias[nt][r] = a.bias ? a.bias[n + r] : 0.f;
    }
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
            for (int r = 0; r < 4; ++r) v[r] = acc[mt][nt][r] + bias[nt][r];
            if (valid) {
                if (want_stats) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        ssum[nt] += v[r];
                        ssq[nt] += v[r] * v[r];
                    }
                }
                if (resp) {
                    if constexpr (sizeof(T) == 4) {
                        const f32x4 rv = *reinterpret_cast<const f32x4*>(resp + obase + n);
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] += rv[r];
                    } else {
                        const bf16x4 rv = *reinterpret_cast<const bf16x4*>(resp + obase + n);
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] += (float)rv[r];
                    }
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
        __syncthreads();   // sRed may alias the normalisation scratch read above by slower waves
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

template <typename T, int KIND, int TH, int TW, int BN, int WM, int WN>
int launch_cfg(const ConvArgs& a, size_t lds, hipStream_t s) {
    const int tiles = ((a.Ho + TH - 1) / TH) * ((a.Wo + TW - 1) / TW);
    dim3 grid(tiles, a.Cout / BN, a.B * a.nphase);
    hipLaunchKernelGGL((conv_igemm_kernel<T, KIND, TH, TW, BN, WM, WN>), grid, dim3(256), lds, s, a);
    DMN_HIP_CHECK(hipGetLastError());
    return DMN_OK;
}

int kind_ext(int kind) { return kind == DMN_K3X3 ? 3 : kind == DMN_K1X1 ? 1 : kind == DMN_K4X4S2 ? 4 : 2; }
int kind_taps(int kind) { return kind == DMN_K3X3 ? 9 : kind == DMN_K1X1 ? 1 : kind == DMN_K4X4S2 ? 16 : 4; }
int kind_stride(int kind) { return kind == DMN_K4X4S2 ? 2 : 1; }

TileCfg pick_cfg(const ConvArgs& a) {
    TileCfg c;
    if (a.cfg_tile && a.cfg_bn) {
        c.th = c.tw = a.cfg_tile;
        c.bn = a.cfg_bn;
        return c;
    }
    auto blocks = [&](const TileCfg& t) {
        return (long long)((a.Ho + t.th - 1) / t.th) * ((a.Wo + t.tw - 1) / t.tw) * (a.Cout / t.bn) * a.B * a.nphase;
    };
    const bool big = a.kind != DMN_K4X4S2 && a.Ho >= 16 && a.Wo >= 16;
    c.th = c.tw = big ? 16 : 8;
    c.bn = (a.Cout % 64 == 0 && a.kind != DMN_K4X4S2) ? 64 : 32;
    // prefer at least one workgroup per CU (256), two when a smaller channel tile gets there
    if (c.th == 16 && blocks(c) < 256) {
        if (c.bn == 64 && blocks(TileCfg{16, 16, 32}) >= 256) c.bn = 32;
        else c.th = c.tw = 8;
    }
    if (c.th == 8 && c.bn == 64 && blocks(c) < 256) c.bn = 32;
    return c;
}

size_t lds_bytes(const ConvArgs& a, const TileCfg& c) {
    const int ext = kind_ext(a.kind), st = kind_stride(a.kind);
    const int IH = (c.th - 1) * st + ext, IW = (c.tw - 1) * st + ext;
    size_t n = (size_t)IH * IW * kRowBytes + (size_t)kind_taps(a.kind) * c.bn * kRowBytes;
    const int Cin = a.C0 + a.C1;
    if (a.pro.stats) n += (size_t)3 * Cin * 4;
    size_t red = 2 * 4 * (c.bn / 4);
    if (a.pro.stats && (size_t)2 * a.pro.G > red) red = 2 * a.pro.G;
    n += red * 4 + 64;
    return n;
}

template <typename T, int KIND>
int dispatch_tile(const ConvArgs& a, const TileCfg& c, size_t lds, hipStream_t s) {
    if (c.th == 16 && c.bn == 32) return launch_cfg<T, KIND, 16, 16, 32, 4, 1>(a, lds, s);
    if (c.th == 16 && c.bn == 64) return launch_cfg<T, KIND, 16, 16, 64, 4, 1>(a, lds, s);
    if (c.th == 8 && c.bn == 32) return launch_cfg<T, KIND, 8, 8, 32, 2, 2>(a, lds, s);
    if (c.th == 8 && c.bn == 64) return launch_cfg<T, KIND, 8, 8, 64, 2, 2>(a, lds, s);
    dmn_set_error("conv: unsupported tile %dx%d / %d", c.th, c.tw, c.bn);
    return DMN_ERR_INVALID;
}

template <typename T>
int dispatch(const ConvArgs& a, const TileCfg& c, size_t lds, hipStream_t s) {
    switch (a.kind) {
        case DMN_K3X3: return dispatch_tile<T, DMN_K3X3>(a, c, lds, s);
        case DMN_K1X1: return dispatch_tile<T, DMN_K1X1>(a, c, lds, s);
        case DMN_KT2X2: return dispatch_tile<T, DMN_KT2X2>(a, c, lds, s);
        case DMN_K4X4S2:
            if (c.th == 8 && c.bn == 32) return launch_cfg<T, DMN_K4X4S2, 8, 8, 32, 2, 2>(a, lds, s);
            if (c.th == 8 && c.bn == 64) return launch_cfg<T, DMN_K4X4S2, 8, 8, 64, 2, 2>(a, lds, s);
            dmn_set_error("conv: the stride-2 kernel only has 8x8 pixel tiles");
            return DMN_ERR_INVALID;
    }
    dmn_set_error("conv: unknown kind %d", a.kind);
    return DMN_ERR_INVALID;
}

template <typename T, int KIND, int TH, int TW, int BN, int WM, int WN>
int set_attr() {
    // opt in to more than 64 KiB of dynamic LDS (done once at handle creation, never inside a stream capture)
    DMN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_kernel<T, KIND, TH, TW, BN, WM, WN>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    return DMN_OK;
}

template <typename T, int KIND>
int set_attr_kind() {
    int rc;
    if (KIND != DMN_K4X4S2) {
        if ((rc = set_attr<T, KIND == DMN_K4X4S2 ? DMN_K3X3 : KIND, 16, 16, 32, 4, 1>())) return rc;
        if ((rc = set_attr<T, KIND == DMN_K4X4S2 ? DMN_K3X3 : KIND, 16, 16, 64, 4, 1>())) return rc;
    }
    if ((rc = set_attr<T, KIND, 8, 8, 32, 2, 2>())) return rc;
    if ((rc = set_attr<T, KIND, 8, 8, 64, 2, 2>())) return rc;
    return DMN_OK;
}

}  // namespace

int conv_init_attributes() {
    static bool done = false;
    if (done) return DMN_OK;
    int rc;
    if ((rc = set_attr_kind<float, DMN_K3X3>())) return rc;
    if ((rc = set_attr_kind<float, DMN_K1X1>())) return rc;
    if ((rc = set_attr_kind<float, DMN_K4X4S2>())) return rc;
    if ((rc = set_attr_kind<float, DMN_KT2X2>())) return rc;
    if ((rc = set_attr_kind<__bf16, DMN_K3X3>())) return rc;
    if ((rc = set_attr_kind<__bf16, DMN_K1X1>())) return rc;
    if ((rc = set_attr_kind<__bf16, DMN_K4X4S2>())) return rc;
    if ((rc = set_attr_kind<__bf16, DMN_KT2X2>())) return rc;
    done = true;
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
    DMN_REQUIRE(a.kind >= 0 && a.kind <= 3 && a.nphase >= 1 && a.nphase <= 4, DMN_ERR_INVALID, "conv: bad kind/phase count");
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
