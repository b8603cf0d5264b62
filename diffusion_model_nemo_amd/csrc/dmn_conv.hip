// Implicit-GEMM convolution for gfx950 (CDNA4): NHWC activations, the input halo tile of ALL input channels
// staged once in LDS, weights streamed from L2 straight into MFMA operand registers, 16x16 MFMA (bf16 x32 or
// exact-f32 x4) with fp32 accumulation.
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
// Per workgroup (256 threads, 4 waves): (1) the halo tile is loaded for a whole channel group (normally every
// input channel), normalised + activated on the way (see Fusions) and written to LDS with a row pitch of
// channel-bytes + 32, which keeps the 16-lane groups of ds_read_b128 on distinct banks; (2) ONE barrier; (3) a
// barrier-free loop over (k-step, tap): the wave's weight fragments arrive from global memory in the exact
// lane order of the MFMA A operand (the host packs them so) through a register ring that is refilled one
// k-step ahead, the pixel fragments come from LDS. Channel groups only repeat (1)-(3) when the tile of all
// channels would not fit in LDS (f32 parity mode, 512-channel decoder inputs).
//
// Fusions: GroupNorm + SiLU + time-embedding bias of the PREVIOUS Block applied while the input tile is
// written to LDS (parts/convnext.py:35-41,81-85), bias, optional residual add, and partial GroupNorm
// statistics of the output for the next consumer.
#include <stdio.h>
#include <stdlib.h>

#include <type_traits>

#include "dmn_handoff.h"
#include "dmn_kernels.h"
#include "dmn_norm.h"

namespace {

constexpr int kRowPad = 32;    // bytes added to every LDS pixel row
constexpr int kLdsSlack = 512; // fragment reads of skipped k-steps may run past the last row

// 3x3 tap (dy * 3 + dx) of ring entry e = dx * 3 + i: dx outer; inside a column the 8-wide tiles take dy = 0, 2, 1
// (both even rows pairs first), the 16-wide tiles dy = 0, 1, 2.
template <int TW> __host__ __device__ constexpr int k3_tap(int e) {
    const int dx = e / 3, i = e % 3;
    const int dy = TW == 16 ? i : (i == 0 ? 0 : i == 1 ? 2 : 1);
    return dy * 3 + dx;
}

template <int KIND> struct KindTraits;
template <> struct KindTraits<DMN_K3X3> { static constexpr int NTAPS = 9, EXT = 3, S = 1, KW = 3, GK = 1; };
template <> struct KindTraits<DMN_K1X1> { static constexpr int NTAPS = 1, EXT = 1, S = 1, KW = 1, GK = 4; };
template <> struct KindTraits<DMN_K4X4S2> { static constexpr int NTAPS = 16, EXT = 4, S = 2, KW = 4, GK = 1; };
template <> struct KindTraits<DMN_KT2X2> { static constexpr int NTAPS = 4, EXT = 2, S = 1, KW = 2, GK = 2; };

// PRO: 0 = input used as is, 1 = GroupNorm affine, 2 = GroupNorm affine + SiLU (+ time bias) while staging,
//      3 = the tail of the producer applied while staging: SiLU(GroupNorm(in0)) + pro_res for a ResnetBlock
//          (parts/convnext.py:85-86; pro.silu set) or GroupNorm(in0) + pro_res for a linear attention
//          (parts/mha.py:41-42 + utils.py:73-74; pro.silu clear). When pro_out is given, the pixels of the tile proper
//          are also written there, so the tensor exists in memory for its other consumers (the next block's residual,
//          a skip connection) without a launch of its own.
// RES: also apply a 1x1 convolution (w2, bias2 -> out2) to the same input tile -- the ResnetBlock shortcut
// res_conv(x) beside block1.proj(x) (parts/convnext.py:76,79,86) -- sharing the staged tile; 3x3 kind only.
// TAIL: the launch is block2.proj of a ResnetBlock and also closes the block (parts/convnext.py:85-86): the
// accumulators stay in registers while the image's workgroups exchange their GroupNorm partial sums (dmn_handoff.h),
// then out = SiLU(GroupNorm(conv + bias)) + res is stored, with partial statistics of THAT tensor for the next
// consumer. The convolution's own output never reaches memory and the block needs no closing pass.
// A launch whose workgroups may wait for each other (TAIL) keeps to half a SIMD's registers: two such workgroups per CU
// must be able to be resident together (dmn_handoff.h).
template <typename T, int KIND, int TH, int TW, int BN, int WM, int WN, int PRO, bool RES, bool TAIL = false>
__global__ void __launch_bounds__(256, TAIL ? 2 : 1) conv_igemm_kernel(const ConvArgs a) {
    static_assert(!RES || KIND == DMN_K3X3, "the fused 1x1 shortcut rides on the 3x3 kernel");
    static_assert(!TAIL || (KIND == DMN_K3X3 && !RES), "the fused block tail rides on the plain 3x3 kernel");
    using KT = KindTraits<KIND>;
    static_assert(WM * WN == 4, "four waves per workgroup");
    constexpr int NTAPS = KT::NTAPS, S = KT::S, GK = KT::GK, KW = KT::KW;
    constexpr int E = NTAPS * GK;                   // (k-step, tap) pairs per iteration of the k loop
    // weight ring depth: the 16x16-pixel tiles run at high occupancy and only look RS fragments ahead (few
    // registers, latency hidden by the other waves); the 8x8 tiles of the wide layers run one workgroup per CU
    // and keep a whole iteration in flight
    // ... and for 3x3 even TWO iterations (k-steps): a k-step is 18 matrix instructions per wave (290 cycles), an L2
    // round trip two to three times that, so with one k-step in flight the loop ran at the latency of the weights
    constexpr bool DEEP = KIND == DMN_K3X3 && TW == 8;
    constexpr int RS = TW == 16 ? (NTAPS == 9 ? 3 : (E >= 4 ? 4 : E)) : (DEEP ? 2 * E : E);
    static_assert(E % RS == 0 || DEEP, "ring depth must divide the iteration");
    constexpr int BM = TH * TW;
    constexpr int MT = BM / (16 * WM);
    constexpr int NT = BN / (16 * WN);
    static_assert(MT >= 1 && NT >= 1, "tile too small for the wave grid");
    constexpr int XB = 2;                          // pixel-fragment buffers (double buffering when cheap)
    constexpr int KCH = ElemTraits<T>::kChunk;
    constexpr int PER = ElemTraits<T>::kPer16B;
    constexpr int IH = (TH - 1) * S + KT::EXT, IW = (TW - 1) * S + KT::EXT;
    constexpr int NPIX = IH * IW;
    constexpr int SB = TH == 8 ? 8 : 6;             // staging loads in flight per thread (8x8 tiles: 100 halo pixels of
                                                    // 128 / 256 channels are 7 / 13 units per thread: one / two batches;
                                                    // the 10 units of a 32x16 tile in one batch measured 1 % slower per step)
    using V = decltype(Vec16<T>::v);

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int l15 = lane & 15, lq = lane >> 4;

    const int tiles_x = (a.Wo + TW - 1) / TW;
    const int tile_y = blockIdx.x / tiles_x, tile_x = blockIdx.x % tiles_x;
    const int n0 = blockIdx.y * BN;
    const int b = a.b_first + blockIdx.z / a.nphase, ph = blockIdx.z % a.nphase;
    const ConvPhase P = a.phase[ph];
    const int iy0 = tile_y * TH * S - P.pad_y, ix0 = tile_x * TW * S - P.pad_x;
    const int Cin = a.C0 + a.C1;
    const int CG = a.cg;                            // channels per staged group
    const int KSG = CG / KCH, KS = Cin / KCH;       // k-steps per group / in total
    const int lgu = a.lg_upp;                       // log2 of the 16-byte units per pixel row (<= 8)
    const int rowB = CG * (int)sizeof(T) + kRowPad;

    const size_t wgid = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    // optional phase timestamps (tuning aid): [0] wall clock at start, [7] cycle counter at start, [1..4] cycle
    // counter after set-up / staging / MFMA loop / epilogue, [5] wall clock at the end
#define DMN_STAMP(i)                                                         \
    if (a.dbg && tid == 0) {                                                 \
        a.dbg[wgid * 8 + (i)] = (i) == 0 ? (long long)wall_clock64() : (long long)clock64(); \
        if ((i) == 0) a.dbg[wgid * 8 + 7] = clock64();                       \
        if ((i) == 4) a.dbg[wgid * 8 + 5] = wall_clock64();                  \
    }
    DMN_STAMP(0)
    char* sX = smem;
    float* sCoef = reinterpret_cast<float*>(smem + (size_t)NPIX * rowB + kLdsSlack);
    float* sTmp = sCoef + (PRO ? 3 * Cin : 0);      // 2*G floats of normalisation scratch
    float* sRed = sTmp + (PRO ? (2 * a.pro.G + 3) & ~3 : 0);   // 4*WM*BN/4 floats of statistics scratch, 16-byte aligned

    // ---- everything that only depends on the block index is requested first -------------------------------
    // weight ring: fragment (ct, flat = ks*NTAPS + tap) is 64 lanes x 16 B, contiguous in flat
    const T* wfrag[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int ct = n0 / 16 + wn * NT + nt;
        wfrag[nt] = static_cast<const T*>(a.w) + P.w_off + ((size_t)ct * KS * NTAPS * 64 + lane) * PER;
    }
    V w[RS][NT];
    const int nflat = KS * NTAPS;                   // fragments per output-channel tile
    // Unconditional on purpose (past the end the last fragment is fetched again and never used): with a branch
    // around the load the compiler can no longer count the loads in flight and waits for ALL of them
    // (vmcnt(0)) before every use, which serialises the ring on the L2 latency.
    auto ring_fill = [&](int slot, int flat) {
        const int f = flat < nflat ? flat : nflat - 1;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) w[slot][nt] = *reinterpret_cast<const V*>(wfrag[nt] + (size_t)f * 64 * PER);
    };
    // 3x3: the taps of a k-step are visited column by column (dx outer) so that the pixel fragments read from LDS for
    // one dx serve all three dy (see the MFMA loop); ring entry e of a k-step is tap k3_tap(e).
    auto ring_fill3 = [&](int slot, int ks, int e) {   // e may run past 8 into the following k-steps
        ring_fill(slot, (ks + e / 9) * 9 + k3_tap<TW>(e % 9));
    };
#pragma unroll
    for (int e = 0; e < RS; ++e) {
        if constexpr (KIND == DMN_K3X3) ring_fill3(e, 0, e);
        else ring_fill(e, e);
    }

    f32x4 bias[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        bias[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (a.bias) bias[nt] = *reinterpret_cast<const f32x4*>(a.bias + n0 + (wn * NT + nt) * 16 + lq * 4);
    }
    // fused shortcut: fragment (ct, ks) of the 1x1 weights, fetched one k-step ahead of its use at the centre tap
    const T* w2frag[RES ? NT : 1];
    V w2v[RES ? NT : 1];
    f32x4 acc2[RES ? MT : 1][RES ? NT : 1];
    if constexpr (RES) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int ct = n0 / 16 + wn * NT + nt;
            w2frag[nt] = static_cast<const T*>(a.w2) + ((size_t)ct * KS * 64 + lane) * PER;
            w2v[nt] = *reinterpret_cast<const V*>(w2frag[nt]);
        }
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc2[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    // tile staging: unit u = tid + 256*j is the 16-byte column `part` of halo pixel u >> lgu
    const int part = tid & ((1 << lgu) - 1);
    const int nunits = ((NPIX << lgu) + 255) >> 8;
    V sr[SB];
    V sr2[PRO == 3 ? SB : 1];                       // residual operand of the fused block tail
    unsigned smask = 0;                             // bit j: unit j of the batch lies inside the image
    // Branch-free on purpose: with short-circuit tests the compiler wraps every load in nested exec-mask regions
    // that re-fetch kernel arguments, several hundred cycles of set-up per workgroup.
    const int inH = a.H, inW = a.W, inC0 = a.C0, inC1 = a.C1;
    const T* const in0p = static_cast<const T*>(a.in0);
    const T* const in1p = static_cast<const T*>(a.in1);
    const T* const resp3 = static_cast<const T*>(a.pro_res);
    const bool pro_silu = a.pro.silu != 0;
    const bool pro_store = PRO == 3 && a.pro_out != nullptr && blockIdx.y == 0 && ph == 0;
    auto stage_issue = [&](int cg0, int j0) {
        const int c = cg0 + part * PER;
        const bool first = c < inC0;
        const T* src = first ? in0p : in1p;
        const unsigned cs = first ? c : c - inC0, Csrc = first ? inC0 : inC1;
        smask = 0;
#pragma unroll
        for (int j = 0; j < SB; ++j) {
            const int pix = (tid + 256 * (j0 + j)) >> lgu;
            const int ly = pix / IW, lx = pix - ly * IW;
            const int gy = iy0 + ly, gx = ix0 + lx;
            const bool in = (pix < NPIX) & ((unsigned)gy < (unsigned)inH) & ((unsigned)gx < (unsigned)inW);
            smask |= (unsigned)in << j;
            // slots outside the image read pixel 0 (always mapped) and are zeroed when written
            const unsigned poff = (unsigned)((b * inH + gy) * inW + gx) * Csrc + cs;
            const unsigned off = in ? poff : cs;
            sr[j] = *reinterpret_cast<const V*>(src + off);
            if constexpr (PRO == 3) sr2[j] = *reinterpret_cast<const V*>(resp3 + off);
        }
    };
    // the statistics and channel parameters of the normalising prologue are requested AHEAD of the halo tile: vector loads
    // return in order, so the coefficient round trip (load, cross-lane combination, LDS, barrier) runs while the halo
    // pixels are still on their way instead of behind them
    // One staged channel group (the rule): a thread normalises the same PER channels of every pixel and keeps their
    // coefficients in registers (norm_direct_*: one barrier less and no round trip through LDS at the head of the launch).
    const bool direct = PRO && a.pro_direct;
    NormLoads nl;
    NormDirect<PER> nd;
    float dca[PER], dcb[PER], dct[PER];
    if (PRO) {
        if (direct) norm_direct_issue<PER>(a.pro, b, part * PER, nd);
        else norm_issue(a.pro, b, Cin, nl);
    }
    stage_issue(0, 0);
    if (PRO) {
        if (direct) norm_direct_finish<PER>(a.pro, b, Cin, part * PER, nd, sTmp, dca, dcb, dct);
        else norm_finish(a.pro, b, Cin, nl, sCoef, sTmp);
    }
    DMN_STAMP(1)

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int cg0 = 0; cg0 < Cin; cg0 += CG) {
        if (cg0) {
            __syncthreads();                        // every wave is done with the previous group's tile
            stage_issue(cg0, 0);
        }
        // ---- write the tile of channels [cg0, cg0 + CG) ----------------------------------------------------
        {
            const int c = cg0 + part * PER;
            float ca[PER], cb[PER], ct[PER];
            if (PRO) {
#pragma unroll
                for (int e = 0; e < PER; ++e) {
                    ca[e] = direct ? dca[e] : sCoef[c + e];
                    cb[e] = direct ? dcb[e] : sCoef[Cin + c + e];
                    ct[e] = direct ? dct[e] : sCoef[2 * Cin + c + e];
                }
            }
            for (int j0 = 0; j0 < nunits; j0 += SB) {
#pragma unroll
                for (int j = 0; j < SB; ++j) {
                    const int pix = (tid + 256 * (j0 + j)) >> lgu;
                    const bool in = (smask >> j) & 1u;
                    Vec16<T> v, r2;
                    v.v = sr[j];
                    if constexpr (PRO == 3) r2.v = sr2[j];
                    // padding skips the prologue's arithmetic: a 16-byte unit belongs to one pixel and the units of a
                    // wave to one or two, so mostly whole waves skip (the halo rows above and below an image)
                    if (!PRO || in)
#pragma unroll
                    for (int e = 0; e < PER; ++e) {
                        float y = v.get(e);
                        if constexpr (PRO >= 2 && sizeof(T) == 2) {
                            if (PRO == 2 || pro_silu) y = silu_affine_fast(y, ca[e], cb[e], PRO == 2 ? ct[e] : r2.get(e));
                            else y = __builtin_fmaf(y, ca[e], cb[e]) + r2.get(e);
                        } else if (PRO) {
                            y = y * ca[e] + cb[e];
                            if (PRO == 2) y = silu_precise(y) + ct[e];
                            if constexpr (PRO == 3) y = (pro_silu ? silu_precise(y) : y) + r2.get(e);
                        }
                        v.set(e, y);
                    }
                    if (!in) v.v = V{};                  // padding is zero AFTER the prologue
                    if (pix < NPIX) *reinterpret_cast<V*>(sX + pix * rowB + part * 16) = v.v;
                    if constexpr (PRO == 3) {
                        // the tile's own pixels (not the halo), staged by the first channel slice and phase, materialise
                        // the tensor
                        const int ly = pix / IW, lx = pix - ly * IW;
                        if (in && pro_store && ly >= P.pad_y && ly < P.pad_y + TH * S && lx >= P.pad_x && lx < P.pad_x + TW * S) {
                            const unsigned off = (unsigned)((b * a.H + iy0 + ly) * a.W + ix0 + lx) * (unsigned)a.C0 + (unsigned)c;
                            store_vec(reinterpret_cast<V*>(static_cast<T*>(a.pro_out) + off), v.v);
                        }
                    }
                }
                if (j0 + SB < nunits) stage_issue(cg0, j0 + SB);
            }
        }
        __syncthreads();
        DMN_STAMP(2)

        // ---- MFMA loop over this group's k-steps, no barriers ------------------------------------------
        // per-lane LDS byte offsets of the pixel fragments (before tap and k-step offsets)
        int xoff[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int p = (wm * MT + mt) * 16 + l15;
            xoff[mt] = (((p / TW) * S) * IW + (p % TW) * S) * rowB + lq * 16;
        }
        const int ksbase = cg0 / KCH;
        if constexpr (KIND == DMN_K3X3) {
            // A 16-pixel fragment is one row of a 16-wide tile or two rows of an 8-wide tile. For a fixed dx the three
            // dy of a pixel tile are fragments of the SAME column window shifted by whole rows, so the wave reads each
            // distinct fragment once per (k-step, dx) and feeds it to every (pixel tile, dy) that needs it:
            //   16 wide: rows r0 .. r0+MT+1                      -> MT + 2 reads for 3 * MT MFMA operands
            //    8 wide: row pairs (2j, 2j+1), j = 0..MT, for dy = 0 / 2 and (2j+1, 2j+2), j < MT, for dy = 1
            //                                                   -> 2 * MT + 1 reads
            // (the loop is bound by LDS reads otherwise: 1 KB per MFMA and wave).
            constexpr int NR = TW == 16 ? MT + 2 : 2 * MT + 1;
            const int rstride = IW * rowB;
            const int rb0 = TW == 16 ? ((wm * MT) * IW + l15) * rowB + lq * 16
                                     : ((2 * wm * MT + (l15 >> 3)) * IW + (l15 & 7)) * rowB + lq * 16;
            // one k-step; with the deep ring, k-steps alternate between the ring's halves (par), so the slots stay
            // compile-time register indices
            auto k3_step = [&](const int ksl, auto par_c) {
                constexpr int par = decltype(par_c)::value;
                const int ks = ksbase + ksl;
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    Vec16<T> R[NR];
#pragma unroll
                    for (int j = 0; j < NR; ++j) {
                        const int rows = TW == 16 ? j : (j <= MT ? 2 * j : 2 * (j - MT - 1) + 1);
                        R[j].v = *reinterpret_cast<const V*>(sX + rb0 + rows * rstride + dx * rowB + ksl * 64);
                    }
#pragma unroll
                    for (int i = 0; i < 3; ++i) {
                        const int e = dx * 3 + i;
                        const int dy = k3_tap<TW>(e) / 3;
                        const int slot = DEEP ? par * E + e : e % RS;
                        Vec16<T> wv[NT];
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) wv[nt].v = w[slot][nt];
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) {
                            const int fj = TW == 16 ? mt + dy : (dy == 1 ? MT + 1 + mt : mt + dy / 2);
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt) mma_step(wv[nt], R[fj], acc[mt][nt]);
                        }
                        if constexpr (RES) {
                            if (dy == 1 && dx == 1) {   // centre tap: the pixel itself
#pragma unroll
                                for (int nt = 0; nt < NT; ++nt) {
                                    Vec16<T> w2;
                                    w2.v = w2v[nt];
#pragma unroll
                                    for (int mt = 0; mt < MT; ++mt)
                                        mma_step(w2, R[TW == 16 ? mt + 1 : MT + 1 + mt], acc2[mt][nt]);
                                    const int ksn = ks + 1 < KS ? ks + 1 : KS - 1;   // unconditional, as the ring
                                    w2v[nt] = *reinterpret_cast<const V*>(w2frag[nt] + (size_t)ksn * 64 * PER);
                                }
                            }
                        }
                        ring_fill3(slot, ks, e + RS);   // the slot just consumed
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            };
            if constexpr (DEEP) {
                int ksl = 0;
                for (; ksl + 1 < KSG; ksl += 2) {
                    k3_step(ksl, std::integral_constant<int, 0>{});
                    k3_step(ksl + 1, std::integral_constant<int, 1>{});
                }
                // an odd number of k-steps only occurs where all channels are one staged group (no ring continues)
                if (ksl < KSG) k3_step(ksl, std::integral_constant<int, 0>{});
            } else {
                for (int ksl = 0; ksl < KSG; ++ksl) k3_step(ksl, std::integral_constant<int, 0>{});
            }
        } else
        for (int ksl = 0; ksl < KSG; ksl += GK) {
            const int flat0 = (ksbase + ksl) * NTAPS; // fragment index of entry 0 of this iteration
            Vec16<T> xf[XB][MT];
            auto load_x = [&](int e, int buf) {
                const int dk = e / NTAPS, t = e % NTAPS;
                const int toff = ((t / KW) * IW + (t % KW)) * rowB;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    xf[buf][mt].v = *reinterpret_cast<const V*>(sX + xoff[mt] + toff + (ksl + dk) * 64);
            };
            if (XB == 2) load_x(0, 0);
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const int dk = e / NTAPS;
                if (XB == 2) {
                    if (e + 1 < E) load_x(e + 1, (e + 1) & 1);
                } else {
                    load_x(e, 0);
                }
                if (GK == 1 || ksl + dk < KSG) {
                    Vec16<T> wv[NT];
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) wv[nt].v = w[e % RS][nt];
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) mma_step(wv[nt], xf[XB == 2 ? (e & 1) : 0][mt], acc[mt][nt]);
                    if constexpr (RES) {
                        if (e == 4) {   // centre tap: the pixel itself
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt) {
                                Vec16<T> w2;
                                w2.v = w2v[nt];
#pragma unroll
                                for (int mt = 0; mt < MT; ++mt) mma_step(w2, xf[XB == 2 ? (e & 1) : 0][mt], acc2[mt][nt]);
                                const int ksn = ksbase + ksl + 1 < KS ? ksbase + ksl + 1 : KS - 1;   // unconditional, as the ring
                                w2v[nt] = *reinterpret_cast<const V*>(w2frag[nt] + (size_t)ksn * 64 * PER);
                            }
                        }
                    }
                }
                ring_fill(e % RS, flat0 + e + RS);   // the slot just consumed
                __builtin_amdgcn_sched_barrier(0);   // keep later taps' LDS reads from piling up in registers
            }
        }
    }
    {
        DMN_STAMP(3)
        // Waves that are about to finish go first: what a launch chain waits for is the END of every launch, and on a CU
        // shared with the other chain's kernels the epilogue otherwise queues behind their staging arithmetic (-0.5 % step
        // time; raising the priority in the matrix loop as well gains less).
        __builtin_amdgcn_s_setprio(3);
        // ---- epilogue: bias, residual, statistics, NHWC store ---------------------------------------------
        T* outp = static_cast<T*>(a.out);
        const T* resp = static_cast<const T*>(a.res);
        // Output offsets in 32 bits (the launcher bounds the tensor), linear in mt: pixel tile mt of the wave lies
        // ROWS rows below tile 0 at the same column, so one scalar stride replaces the per-tile multiplications.
        constexpr int ROWS = 16 / TW;
        const int prow = (wm * MT * 16 + l15) / TW, pcol = (wm * MT * 16 + l15) % TW;
        const int oyt0 = tile_y * TH + prow, oxt = tile_x * TW + pcol;
        const unsigned obase0 = (unsigned)((b * a.out_H + oyt0 * a.out_sy + P.out_oy) * a.out_W + oxt * a.out_sx + P.out_ox) *
                                (unsigned)a.Cout;
        const unsigned ostep = (unsigned)(ROWS * a.out_sy * a.out_W) * (unsigned)a.Cout;
        const bool col_ok = oxt < a.Wo;
        // Partial moments {sum, squared deviations from the partial's own mean, count} (dmn_norm.h) of value(mt, nt, r): one
        // partial per ROW of 16 pixel lanes and 4-channel quad, formed without LDS or barrier. The 16 lanes sum deviations
        // from ONE shift, the first value of the row's first lane (a valid pixel whenever the row holds one: tiles start
        // inside the image), so their sums simply add -- independent butterflies as for plain sums -- and the row's
        // first lane takes the shift out. Slots of a group are numbered (workgroup, wave row wm, quad inside the group);
        // combining them is the consumer's work (combine_groups: one lane per slot).
        auto mt_valid = [&](int mt) { return col_ok && oyt0 + mt * ROWS < a.Ho; };
        auto row_moments = [&](auto value, float (&rs)[NT], float (&rm2)[NT], float& nrow) {
            float ps[NT], pq[NT], pk[NT];
            float cnt = 0.f;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                ps[nt] = pq[nt] = 0.f;
                pk[nt] = row16_first(value(0, nt, 0));
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                if (mt_valid(mt)) {
                    cnt += 4.f;
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float d = value(mt, nt, r) - pk[nt];
                            ps[nt] += d;
                            pq[nt] += d * d;
                        }
                }
            nrow = row16_sum(cnt);   // elements of a 4-channel quad over the row's valid pixels
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) moments_from_shifted(row16_sum(ps[nt]), row16_sum(pq[nt]), nrow, pk[nt], rs[nt], rm2[nt]);
        };
        // the row partials as slots: sink(g, slot, s, m2, n) runs in the first lane of every row
        auto rows_to_slots = [&](auto value, int G, auto sink) {
            float rs[NT], rm2[NT], nrow;
            row_moments(value, rs, rm2, nrow);
            if (l15 == 0) {
                const int cpg = a.Cout / G;
                const int qpg = (cpg >= BN ? BN : cpg) / 4;    // quads per touched group inside this workgroup
                const int sub = cpg > BN ? cpg / BN : 1;       // workgroups (along N) sharing one group
                const int wgslot = blockIdx.x * sub + (cpg > BN ? (n0 % cpg) / BN : 0);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const int quad = (wn * NT + nt) * 4 + lq;  // local quad index in [0, BN/4)
                    sink((n0 + quad * 4) / cpg, (wgslot * WM + wm) * qpg + quad % qpg, rs[nt], rm2[nt], nrow);
                }
            }
        };
        // A workgroup that holds every pixel and channel of its groups: the row partials go to LDS in slot order and are
        // combined there, by the consumer's own routine (same values, same order: same bits as through a buffer).
        auto local_stats = [&](auto value, int G, float inv_count, float* sMR) {
            const int cpg = a.Cout / G;
            const int qpg = (cpg >= BN ? BN : cpg) / 4;
            const int ngl = cpg >= BN ? 1 : BN / cpg;
            rows_to_slots(value, G, [&](int g, int slot, float s_, float m2_, float n_) {
                *reinterpret_cast<f32x4*>(sRed + ((g - n0 / cpg) * (WM * qpg) + slot) * 4) = f32x4{s_, m2_, n_, 0.f};
            });
            __syncthreads();
            combine_groups(ngl, WM * qpg, inv_count, sMR, [&](int g, int i) { return *reinterpret_cast<const f32x4*>(sRed + (g * (WM * qpg) + i) * 4); });
        };
        auto conv_value = [&](int mt, int nt, int r) { return acc[mt][nt][r] + bias[nt][r]; };
        auto store_partial = [&](const NormOut& o, int g, int slot, float s_, float m2_, float n_) {
            f32x4* dst = reinterpret_cast<f32x4*>(o.stats) + (size_t)(b * o.G + g) * o.slots + slot;
            *dst = f32x4{s_, m2_, n_, 0.f};
        };
        if constexpr (TAIL) {
            // (0) requests whose answers are only needed behind the wait: the residual operand of every output
            //     fragment, this thread's channel of the closing GroupNorm
            using RV = typename std::conditional<sizeof(T) == 4, f32x4, bf16x4>::type;
            RV rv[MT][NT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const bool valid = col_ok && oyt0 + mt * ROWS < a.Ho;
                const unsigned obase = valid ? obase0 + (unsigned)mt * ostep : 0u;   // outside the image: any mapped address
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    rv[mt][nt] = *reinterpret_cast<const RV*>(resp + obase + n0 + (wn * NT + nt) * 16 + lq * 4);
            }
            const unsigned epoch = *a.tail.epoch;
            const int G = a.epi.G;
            const int cpg = a.Cout / G;
            float pgam = 0.f, pbet = 0.f;
            if (tid < BN) {
                pgam = a.tail_gamma[n0 + tid];
                pbet = a.tail_beta[n0 + tid];
            }
            // (1) statistics of conv + bias over this workgroup's pixels, published as granules -- unless this workgroup
            //     holds every pixel and channel of its groups (one slot per group: nothing to exchange)
            const bool exchange = a.epi_wgs > 1;
            // The halo tile is dead once every wave has left its MFMA loop: its LDS then holds mean / rstd of this
            // workgroup's groups and the per-channel coefficients. The statistics scratch sRed lies behind the tile.
            float* sMR = reinterpret_cast<float*>(smem);
            float* sC = sMR + 2 * kHandoffMaxGroups;
            const int ng = cpg >= BN ? 1 : BN / cpg;
            if (exchange) {
                rows_to_slots(conv_value, G, [&](int g, int slot, float s_, float m2_, float n_) {
                    handoff_publish(a.tail, epoch, b, G, a.epi.slots, g, slot, s_, m2_, n_);
                });
                __syncthreads();   // every wave is done with the halo tile
                // (2) wait for the image's other workgroups
                handoff_group_stats(a.tail, epoch, b, G, a.epi.slots, n0 / cpg, ng, a.tail_inv_count, sMR);
            } else {   // the arithmetic of the exchange on this workgroup's own partials
                local_stats(conv_value, G, a.tail_inv_count, sMR);
            }
            __syncthreads();
            if (a.dbg && tid == 0) a.dbg[wgid * 8 + 6] = clock64();
            if (tid < BN) {
                const int gl = cpg >= BN ? 0 : tid / cpg;
                const float aa = sMR[2 * gl + 1] * pgam;
                sC[tid] = aa;
                sC[BN + tid] = pbet - sMR[2 * gl] * aa;
            }
            __syncthreads();
            // (3) out = SiLU(GroupNorm(conv + bias)) + res, in the arithmetic of the stand-alone closing pass (which
            //     reads the convolution's output rounded to the activation type)
            // A wave that holds every channel of its pixels (WN == 1, 32 channels = 64 bytes per pixel) passes its results
            // through LDS (the dead halo tile) and stores whole pixel rows, 16 bytes per lane and 1 KB contiguous per
            // instruction, instead of 8-byte pieces that fill every 64-byte segment in two instructions.
            constexpr bool ROWSTORE = WN == 1 && sizeof(T) == 2 && BN == 32;
            constexpr int OROW = BN * 2 + 16;
            char* sOut = smem + 1024 + (size_t)wave * (MT * 16) * OROW;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const bool valid = col_ok && oyt0 + mt * ROWS < a.Ho;
                const unsigned obase = obase0 + (unsigned)mt * ostep;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const int cl = (wn * NT + nt) * 16 + lq * 4;
                    const int n = n0 + cl;
                    const f32x4 ca = *reinterpret_cast<const f32x4*>(sC + cl);
                    const f32x4 cb = *reinterpret_cast<const f32x4*>(sC + BN + cl);
                    if (valid) {
                        // the result replaces the accumulator: the statistics below take it from there
                        if constexpr (sizeof(T) == 4) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                float t = (acc[mt][nt][r] + bias[nt][r]) * ca[r] + cb[r];
                                t = silu_precise(t);
                                t += 0.f;
                                acc[mt][nt][r] = t + rv[mt][nt][r];
                            }
                            store_vec(reinterpret_cast<f32x4*>(outp + obase + n), acc[mt][nt]);
                        } else {
                            bf16x4 o;
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const float x = (float)(__bf16)(acc[mt][nt][r] + bias[nt][r]);
                                o[r] = (__bf16)(silu_affine_fast(x, ca[r], cb[r], 0.f) + (float)rv[mt][nt][r]);
                                acc[mt][nt][r] = (float)o[r];   // statistics of the value the consumer will read
                            }
                            if constexpr (ROWSTORE) *reinterpret_cast<bf16x4*>(sOut + (mt * 16 + l15) * OROW + cl * 2) = o;
                            else store_vec(reinterpret_cast<bf16x4*>(outp + obase + n), o);
                        }
                    }
                }
            }
            if constexpr (ROWSTORE) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                static_assert(!ROWSTORE || TW == 16, "a pixel tile is one row of the tile");
#pragma unroll
                for (int j = 0; j < MT; ++j) {   // MT * 16 pixels x 4 units of 16 bytes = MT units per lane
                    const int u = lane + 64 * j;
                    const int pl = u >> 2, part = u & 3;
                    const int oy = tile_y * TH + wm * MT + (pl >> 4), ox = tile_x * TW + (pl & 15);
                    const uint4 v = *reinterpret_cast<const uint4*>(sOut + pl * OROW + part * 16);
                    if (oy < a.Ho && ox < a.Wo) {
                        const unsigned off = (unsigned)((b * a.out_H + oy * a.out_sy + P.out_oy) * a.out_W + ox * a.out_sx + P.out_ox) *
                                             (unsigned)a.Cout + (unsigned)(n0 + part * 8);
                        store_vec(reinterpret_cast<uint4*>(outp + off), v);
                    }
                }
            }
            // (4) partial statistics of the block's output
            if (a.tail_stats.stats) {
                rows_to_slots([&](int mt, int nt, int r) { return acc[mt][nt][r]; }, a.tail_stats.G,
                              [&](int g, int slot, float s_, float m2_, float n_) { store_partial(a.tail_stats, g, slot, s_, m2_, n_); });
            }
        } else {
        const bool want_stats = a.epi.stats != nullptr;
        bool activated = false;
        if constexpr (KIND == DMN_K3X3 && TH == 8) {
            // block1 of a ResnetBlock whose workgroups each hold every pixel and channel of their groups (8x8 images): the
            // epilogue applies the block's own GroupNorm + SiLU + time bias (parts/convnext.py:33-45,82-84) to the
            // accumulators -- statistics, coefficients and arithmetic exactly those of a consumer that normalises the
            // stored convolution while staging it (same bits), but once per element instead of once per consumer
            // workgroup and halo pixel, and the consumer needs neither the statistics round trip nor the prologue.
            if (a.act_gamma) {
                activated = true;
                const int G = a.epi.G;
                const int cpg = a.Cout / G;
                float pgam = 0.f, pbet = 0.f, ptb = 0.f;
                if (tid < BN) {
                    pgam = a.act_gamma[n0 + tid];
                    pbet = a.act_beta[n0 + tid];
                    if (a.act_tbias) ptb = a.act_tbias[(size_t)b * a.act_tb_stride + n0 + tid];
                }
                // the halo tile is dead behind the barrier inside local_stats
                float* sMR = reinterpret_cast<float*>(smem);
                float* sC = sMR + 2 * kHandoffMaxGroups;
                local_stats(conv_value, G, a.act_inv_count, sMR);
                __syncthreads();
                if (tid < BN) {
                    const int gl = cpg >= BN ? 0 : tid / cpg;
                    const float aa = sMR[2 * gl + 1] * pgam;
                    sC[tid] = aa;
                    sC[BN + tid] = pbet - sMR[2 * gl] * aa;
                    sC[2 * BN + tid] = ptb;
                }
                __syncthreads();
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const bool valid = col_ok && oyt0 + mt * ROWS < a.Ho;
                    const unsigned obase = obase0 + (unsigned)mt * ostep;
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        const int cl = (wn * NT + nt) * 16 + lq * 4;
                        const int n = n0 + cl;
                        const f32x4 ca = *reinterpret_cast<const f32x4*>(sC + cl);
                        const f32x4 cb = *reinterpret_cast<const f32x4*>(sC + BN + cl);
                        const f32x4 ct = *reinterpret_cast<const f32x4*>(sC + 2 * BN + cl);
                        if (valid) {
                            if constexpr (sizeof(T) == 4) {
                                float y[4];
#pragma unroll
                                for (int r = 0; r < 4; ++r) {
                                    float t = (acc[mt][nt][r] + bias[nt][r]) * ca[r] + cb[r];
                                    y[r] = silu_precise(t) + ct[r];
                                }
                                store_vec(reinterpret_cast<f32x4*>(outp + obase + n), f32x4{y[0], y[1], y[2], y[3]});
                            } else {
                                bf16x4 o;
#pragma unroll
                                for (int r = 0; r < 4; ++r) {
                                    const float x = (float)(__bf16)(acc[mt][nt][r] + bias[nt][r]);
                                    o[r] = (__bf16)silu_affine_fast(x, ca[r], cb[r], ct[r]);
                                }
                                store_vec(reinterpret_cast<bf16x4*>(outp + obase + n), o);
                            }
                        }
                    }
                }
            }
        }
        if (!activated)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const bool valid = col_ok && oyt0 + mt * ROWS < a.Ho;
            const unsigned obase = obase0 + (unsigned)mt * ostep;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int n = n0 + (wn * NT + nt) * 16 + lq * 4;
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = acc[mt][nt][r] + bias[nt][r];
                if (valid) {
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
                    // the sub-pixel phases of the transposed convolution interleave their pixels: those partial
                    // lines are left to merge in the L2, everything else streams out
                    constexpr bool STREAM = KIND != DMN_KT2X2;
                    if constexpr (sizeof(T) == 4) {
                        const f32x4 o = f32x4{v[0], v[1], v[2], v[3]};
                        if constexpr (STREAM) store_vec(reinterpret_cast<f32x4*>(outp + obase + n), o);
                        else *reinterpret_cast<f32x4*>(outp + obase + n) = o;
                    } else {
                        bf16x4 o;
#pragma unroll
                        for (int r = 0; r < 4; ++r) o[r] = (__bf16)v[r];
                        if constexpr (STREAM) store_vec(reinterpret_cast<bf16x4*>(outp + obase + n), o);
                        else *reinterpret_cast<bf16x4*>(outp + obase + n) = o;
                    }
                }
            }
        }
        if constexpr (RES) {
            T* out2 = static_cast<T*>(a.out2);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int n = n0 + (wn * NT + nt) * 16 + lq * 4;
                const f32x4 b2 = *reinterpret_cast<const f32x4*>(a.bias2 + n);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    if (col_ok && oyt0 + mt * ROWS < a.Ho) {
                        // the shortcut is a stride-1 convolution: its output grid is the input grid
                        const unsigned o = (unsigned)((b * a.out_H + oyt0 + mt * ROWS) * a.out_W + oxt) * (unsigned)a.Cout + n;
                        if constexpr (sizeof(T) == 4) {
                            store_vec(reinterpret_cast<f32x4*>(out2 + o), acc2[mt][nt] + b2);
                        } else {
                            bf16x4 ov;
#pragma unroll
                            for (int r = 0; r < 4; ++r) ov[r] = (__bf16)(acc2[mt][nt][r] + b2[r]);
                            store_vec(reinterpret_cast<bf16x4*>(out2 + o), ov);
                        }
                    }
                }
            }
        }
        if (want_stats && !activated) {   // of conv + bias, before the residual
            rows_to_slots(conv_value, a.epi.G,
                          [&](int g, int slot, float s_, float m2_, float n_) { store_partial(a.epi, g, slot, s_, m2_, n_); });
        }
        }
    }
    // a clocked launch's end is when its results have left: every wave waits until its stores have been answered, then
    // the workgroup stamps (a profiler's dispatch duration includes that drain; without the wait the stamps end 1-3 us
    // before it, most of all where an exchange makes all workgroups store at the same time)
    if (a.dbg) {
        __builtin_amdgcn_s_waitcnt(0);   // every counter at zero
        __syncthreads();
    }
    DMN_STAMP(4)
#undef DMN_STAMP
}

struct TileCfg {
    int th, tw, bn;
};

template <typename T, int KIND, int TH, int TW, int BN, int WM, int WN, int PRO, bool RES = false, bool TAIL = false>
int launch_cfg(const ConvArgs& a, size_t lds, hipStream_t s) {
    const int tiles = ((a.Ho + TH - 1) / TH) * ((a.Wo + TW - 1) / TW);
    dim3 grid(tiles, a.Cout / BN, a.b_count * a.nphase);
    hipLaunchKernelGGL((conv_igemm_kernel<T, KIND, TH, TW, BN, WM, WN, PRO, RES, TAIL>), grid, dim3(256), lds, s, a);
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
    {   // tuning override: DMN_CONV_TILE_<Ho>_<Cout>_<kind>="<tile>,<bn>" (tile 8|16, bn 32|64) for launches of that shape
        char key[64];
        snprintf(key, sizeof(key), "DMN_CONV_TILE_%d_%d_%d", a.Ho, a.Cout, a.kind);
        const char* v = getenv(key);
        int t = 0, n = 0;
        if (v && sscanf(v, "%d,%d", &t, &n) == 2 && (t == 8 || t == 16) && (n == 32 || n == 64) &&
            !(a.kind == DMN_K4X4S2 && t == 16) && !(a.w2 && t == 16 && n == 64)) {
            c.th = c.tw = t;
            c.bn = n;
            return c;
        }
    }
    const int planB = a.plan_B > 0 ? a.plan_B : a.B;
    auto blocks = [&](const TileCfg& t) {
        return (long long)((a.Ho + t.th - 1) / t.th) * ((a.Wo + t.tw - 1) / t.tw) * (a.Cout / t.bn) * planB * a.nphase;
    };
    const bool big = a.kind != DMN_K4X4S2 && a.Ho >= 32 && a.Wo >= 32;   // 16x16 images: four 8x8 tiles beat one big tile split over channels
    c.th = c.tw = big ? 16 : 8;
    c.bn = (a.Cout % 64 == 0 && a.kind != DMN_K4X4S2) ? 64 : 32;
    // prefer at least one workgroup per CU (256), two when a smaller channel tile gets there
    if (c.th == 16 && blocks(c) < 256) {
        if (c.bn == 64 && blocks(TileCfg{16, 16, 32}) >= 256) c.bn = 32;
        else c.th = c.tw = 8;
    }
    if (c.th == 8 && c.bn == 64 && blocks(c) < 256) c.bn = 32;
    // The sampling loop launches half the planning batch per chain: 3x3 launches that would leave half the CUs of a chain
    // without a workgroup take the 32-channel tile (measured with two chains on MI355X: 64 channels at 32x32 pixels
    // -0.7 % of the step, 256 channels at 8x8 -0.6 %, 64 channels at 16x16 -0.7 %; 128 channels at 16x16 already have 512
    // workgroups and lose 1 % with the narrow tile, and for the 1x1 kernels the wider tile stays ahead).
    // Several 8x8 tiles per image only while the image keeps at most kHandoffMaxWgsPerImage workgroups, so that the
    // launch still fits the in-launch exchange.
    const int tiles8 = ((a.Ho + 7) / 8) * ((a.Wo + 7) / 8);
    if (a.kind == DMN_K3X3 && c.bn == 64 && blocks(c) < 512 && !(a.w2 && c.th == 16) &&
        (c.th == 16 || tiles8 == 1 || tiles8 * (a.Cout / 32) <= kHandoffMaxWgsPerImage))
        c.bn = 32;
    if (a.w2 && c.th == 16 && c.bn == 64) c.th = c.tw = 8;   // two accumulator sets: keep the per-wave tile small
    // A launch that wants to close its ResnetBlock itself (tall_tail) but has too many 16x16 tiles per image for the
    // in-launch exchange takes tiles of 32 rows x 16 columns when those fit: full-resolution images (64x64: 8
    // workgroups per image instead of 16) then need no closing pass either.
    if (a.tall_tail && a.kind == DMN_K3X3 && !a.w2 && c.th == 16 && c.bn == 32 && a.Cout == 32) {
        const int t16 = ((a.Ho + 15) / 16) * ((a.Wo + 15) / 16), t32 = ((a.Ho + 31) / 32) * ((a.Wo + 15) / 16);
        if (t16 > kHandoffMaxWgsPerImage && t32 <= kHandoffMaxWgsPerImage) c.th = 32;
    }
    return c;
}

size_t tile_pixels(const ConvArgs& a, const TileCfg& c) {
    const int ext = kind_ext(a.kind), st = kind_stride(a.kind);
    const int IH = (c.th - 1) * st + ext, IW = (c.tw - 1) * st + ext;
    return (size_t)IH * IW;
}

size_t lds_bytes(const ConvArgs& a, const TileCfg& c, int cg) {
    const size_t esz = a.dtype == DMN_F32 ? 4 : 2;
    size_t n = tile_pixels(a, c) * (cg * esz + kRowPad) + kLdsSlack;
    const int Cin = a.C0 + a.C1;
    if (a.pro.stats) n += (size_t)3 * Cin * 4;
    size_t red = 4 * 4 * (c.bn / 4) + 4;
    if (a.pro.stats) red += 2 * a.pro.G;
    n += red * 4 + 64;
    return n;
}

// channels staged per group: all of them when the tile fits, else the largest power-of-two fraction that does
int pick_cg(const ConvArgs& a, const TileCfg& c) {
    const int kch = a.dtype == DMN_F32 ? 16 : 32;
    int cg = a.C0 + a.C1;
    while (lds_bytes(a, c, cg) > 150 * 1024 && cg % 2 == 0 && (cg / 2) % kch == 0) cg /= 2;
    return cg;
}

// wave grids per tile: 16x16 pixels -> 2x2 waves (8 pixel tiles per wave), 8x8 pixels -> all four waves share the
// pixels and split the output channels when there are 64 of them
template <typename T, int KIND, int PRO>
int dispatch_tile(const ConvArgs& a, const TileCfg& c, size_t lds, hipStream_t s) {
    if constexpr (KIND == DMN_K3X3 && PRO == 2) {   // the tall tile without the tail: the two-launch form of the same partition
        if (c.th == 32 && c.tw == 16 && c.bn == 32) return launch_cfg<T, KIND, 32, 16, 32, 4, 1, PRO>(a, lds, s);
    }
    if constexpr (KIND != DMN_K4X4S2) {
        if (c.th == 16 && c.bn == 32) return launch_cfg<T, KIND, 16, 16, 32, 2, 2, PRO>(a, lds, s);
        if (c.th == 16 && c.bn == 64) return launch_cfg<T, KIND, 16, 16, 64, 2, 2, PRO>(a, lds, s);
    }
    if (c.th == 8 && c.bn == 32) return launch_cfg<T, KIND, 8, 8, 32, 2, 2, PRO>(a, lds, s);
    if (c.th == 8 && c.bn == 64) return launch_cfg<T, KIND, 8, 8, 64, 1, 4, PRO>(a, lds, s);
    dmn_set_error("conv: unsupported tile %dx%d / %d for kind %d", c.th, c.tw, c.bn, KIND);
    return DMN_ERR_INVALID;
}

template <typename T>
int dispatch_fused_shortcut(const ConvArgs& a, const TileCfg& c, size_t lds, hipStream_t s) {
    if (c.th == 16 && c.bn == 32) return launch_cfg<T, DMN_K3X3, 16, 16, 32, 2, 2, 0, true>(a, lds, s);
    if (c.th == 8 && c.bn == 32) return launch_cfg<T, DMN_K3X3, 8, 8, 32, 2, 2, 0, true>(a, lds, s);
    if (c.th == 8 && c.bn == 64) return launch_cfg<T, DMN_K3X3, 8, 8, 64, 1, 4, 0, true>(a, lds, s);
    dmn_set_error("conv: no fused-shortcut kernel for tile %dx%d / %d", c.th, c.tw, c.bn);
    return DMN_ERR_INVALID;
}

// block2.proj that also closes its ResnetBlock (TAIL): 3x3, GroupNorm + SiLU prologue
template <typename T>
int dispatch_fused_tail(const ConvArgs& a, const TileCfg& c, size_t lds, hipStream_t s, int pro) {
    if (pro == 0) {   // the input was normalised and activated by its producer (one 8x8 tile per image)
        if (c.th == 8 && c.bn == 32) return launch_cfg<T, DMN_K3X3, 8, 8, 32, 2, 2, 0, false, true>(a, lds, s);
        if (c.th == 8 && c.bn == 64) return launch_cfg<T, DMN_K3X3, 8, 8, 64, 1, 4, 0, false, true>(a, lds, s);
        dmn_set_error("conv: no fused-tail kernel without prologue for tile %dx%d / %d", c.th, c.tw, c.bn);
        return DMN_ERR_INVALID;
    }
    if (c.th == 32 && c.tw == 16 && c.bn == 32) return launch_cfg<T, DMN_K3X3, 32, 16, 32, 4, 1, 2, false, true>(a, lds, s);
    if (c.th == 16 && c.bn == 32) return launch_cfg<T, DMN_K3X3, 16, 16, 32, 2, 2, 2, false, true>(a, lds, s);
    if (c.th == 16 && c.bn == 64) return launch_cfg<T, DMN_K3X3, 16, 16, 64, 2, 2, 2, false, true>(a, lds, s);
    if (c.th == 8 && c.bn == 32) return launch_cfg<T, DMN_K3X3, 8, 8, 32, 2, 2, 2, false, true>(a, lds, s);
    if (c.th == 8 && c.bn == 64) return launch_cfg<T, DMN_K3X3, 8, 8, 64, 1, 4, 2, false, true>(a, lds, s);
    dmn_set_error("conv: no fused-tail kernel for tile %dx%d / %d", c.th, c.tw, c.bn);
    return DMN_ERR_INVALID;
}

template <typename T>
int dispatch(const ConvArgs& a, const TileCfg& c, size_t lds, hipStream_t s) {
    const int pro = a.pro.stats ? (a.pro_res ? 3 : a.pro.silu ? 2 : 1) : 0;
    if (c.th == 32 && !(pro == 2 && a.kind == DMN_K3X3 && !a.w2)) {
        dmn_set_error("conv: the 32x16 tile only serves block2.proj (3x3 with the GroupNorm + SiLU prologue)");
        return DMN_ERR_INVALID;
    }
    if (a.tail.gran) {
        if (a.kind != DMN_K3X3 || (pro != 2 && pro != 0) || a.w2 || a.C1) {
            dmn_set_error("conv: the fused block tail needs a single-source 3x3 convolution with the GroupNorm + SiLU prologue "
                          "or an input activated by its producer");
            return DMN_ERR_INVALID;
        }
        return dispatch_fused_tail<T>(a, c, lds, s, pro);
    }
    if (pro == 3) {
        if (a.kind == DMN_K1X1 || a.w2 || a.C1 || a.pro.tbias) {
            dmn_set_error("conv: the fused producer tail needs a plain single-source 3x3 or resampling convolution");
            return DMN_ERR_INVALID;
        }
        if (a.kind == DMN_K4X4S2) return dispatch_tile<T, DMN_K4X4S2, 3>(a, c, lds, s);
        if (a.kind == DMN_KT2X2) return dispatch_tile<T, DMN_KT2X2, 3>(a, c, lds, s);
        return dispatch_tile<T, DMN_K3X3, 3>(a, c, lds, s);
    }
    if (a.w2) {
        if (a.kind != DMN_K3X3 || pro != 0) {
            dmn_set_error("conv: the fused 1x1 shortcut needs a plain 3x3 convolution");
            return DMN_ERR_INVALID;
        }
        return dispatch_fused_shortcut<T>(a, c, lds, s);
    }
    switch (a.kind) {
        case DMN_K3X3:
            if (pro == 0) return dispatch_tile<T, DMN_K3X3, 0>(a, c, lds, s);
            if (pro == 1) return dispatch_tile<T, DMN_K3X3, 1>(a, c, lds, s);
            return dispatch_tile<T, DMN_K3X3, 2>(a, c, lds, s);
        case DMN_K1X1:
            if (pro == 0) return dispatch_tile<T, DMN_K1X1, 0>(a, c, lds, s);
            if (pro == 1) return dispatch_tile<T, DMN_K1X1, 1>(a, c, lds, s);
            return dispatch_tile<T, DMN_K1X1, 2>(a, c, lds, s);
        case DMN_KT2X2:
            if (pro == 0) return dispatch_tile<T, DMN_KT2X2, 0>(a, c, lds, s);
            break;
        case DMN_K4X4S2:
            if (pro == 0) return dispatch_tile<T, DMN_K4X4S2, 0>(a, c, lds, s);
            break;
        default:
            dmn_set_error("conv: unknown kind %d", a.kind);
            return DMN_ERR_INVALID;
    }
    dmn_set_error("conv: input normalisation is only built for the 3x3 and 1x1 kernels");
    return DMN_ERR_INVALID;
}

template <typename T, int KIND, int TH, int TW, int BN, int WM, int WN, int PRO, bool RES = false, bool TAIL = false>
int set_attr() {
    // opt in to more than 64 KiB of dynamic LDS (done once at handle creation, never inside a stream capture)
    DMN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_kernel<T, KIND, TH, TW, BN, WM, WN, PRO, RES, TAIL>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    return DMN_OK;
}

template <typename T, int KIND, int PRO>
int set_attr_kind() {
    int rc;
    if constexpr (KIND != DMN_K4X4S2) {
        if ((rc = set_attr<T, KIND, 16, 16, 32, 2, 2, PRO>())) return rc;
        if ((rc = set_attr<T, KIND, 16, 16, 64, 2, 2, PRO>())) return rc;
    }
    if ((rc = set_attr<T, KIND, 8, 8, 32, 2, 2, PRO>())) return rc;
    if ((rc = set_attr<T, KIND, 8, 8, 64, 1, 4, PRO>())) return rc;
    return DMN_OK;
}

template <typename T>
int set_attr_all() {
    int rc;
    if ((rc = set_attr_kind<T, DMN_K3X3, 0>())) return rc;
    if ((rc = set_attr_kind<T, DMN_K3X3, 1>())) return rc;
    if ((rc = set_attr_kind<T, DMN_K3X3, 2>())) return rc;
    if ((rc = set_attr_kind<T, DMN_K3X3, 3>())) return rc;
    if ((rc = set_attr_kind<T, DMN_K1X1, 0>())) return rc;
    if ((rc = set_attr_kind<T, DMN_K1X1, 1>())) return rc;
    if ((rc = set_attr_kind<T, DMN_K1X1, 2>())) return rc;
    if ((rc = set_attr_kind<T, DMN_K4X4S2, 0>())) return rc;
    if ((rc = set_attr_kind<T, DMN_KT2X2, 0>())) return rc;
    if ((rc = set_attr_kind<T, DMN_K4X4S2, 3>())) return rc;
    if ((rc = set_attr_kind<T, DMN_KT2X2, 3>())) return rc;
    if ((rc = set_attr<T, DMN_K3X3, 16, 16, 32, 2, 2, 0, true>())) return rc;
    if ((rc = set_attr<T, DMN_K3X3, 8, 8, 32, 2, 2, 0, true>())) return rc;
    if ((rc = set_attr<T, DMN_K3X3, 8, 8, 64, 1, 4, 0, true>())) return rc;
    if ((rc = set_attr<T, DMN_K3X3, 32, 16, 32, 4, 1, 2, false, true>())) return rc;
    if ((rc = set_attr<T, DMN_K3X3, 32, 16, 32, 4, 1, 2>())) return rc;
    if ((rc = set_attr<T, DMN_K3X3, 16, 16, 32, 2, 2, 2, false, true>())) return rc;
    if ((rc = set_attr<T, DMN_K3X3, 16, 16, 64, 2, 2, 2, false, true>())) return rc;
    if ((rc = set_attr<T, DMN_K3X3, 8, 8, 32, 2, 2, 2, false, true>())) return rc;
    if ((rc = set_attr<T, DMN_K3X3, 8, 8, 64, 1, 4, 2, false, true>())) return rc;
    if ((rc = set_attr<T, DMN_K3X3, 8, 8, 32, 2, 2, 0, false, true>())) return rc;
    if ((rc = set_attr<T, DMN_K3X3, 8, 8, 64, 1, 4, 0, false, true>())) return rc;
    return DMN_OK;
}

}  // namespace

static long long* g_conv_dbg = nullptr;
void conv_set_debug_buffer(long long* dev) { g_conv_dbg = dev; }
long long* kernel_debug_buffer() { return g_conv_dbg; }

int conv_init_attributes() {
    static bool done = false;
    if (done) return DMN_OK;
    int rc;
    if ((rc = set_attr_all<float>())) return rc;
    if ((rc = set_attr_all<__bf16>())) return rc;
    done = true;
    return DMN_OK;
}

// partials per (image, group): one per workgroup that touches the group, wave row and 4-channel quad of the group inside it
static int rows_per_wg(const TileCfg& c) { return c.th == 32 ? 4 : (c.th == 8 && c.bn == 64) ? 1 : 2; }   // WM of the kernel that serves the tile
static int slots_for_groups(const ConvArgs& a, int G) {
    const TileCfg c = pick_cfg(a);
    const int tiles = ((a.Ho + c.th - 1) / c.th) * ((a.Wo + c.tw - 1) / c.tw);
    const int cpg = a.Cout / G;
    return tiles * (cpg > c.bn ? cpg / c.bn : 1) * rows_per_wg(c) * ((cpg >= c.bn ? c.bn : cpg) / 4);
}
// workgroups that touch a group of an image: 1 = the workgroup holds the whole group (nothing to exchange)
static int wgs_per_group(const ConvArgs& a, int G) {
    const TileCfg c = pick_cfg(a);
    const int tiles = ((a.Ho + c.th - 1) / c.th) * ((a.Wo + c.tw - 1) / c.tw);
    const int cpg = a.Cout / G;
    return tiles * (cpg > c.bn ? cpg / c.bn : 1);
}
int conv_epi_slots(const ConvArgs& a) { return slots_for_groups(a, a.epi.G); }
int conv_epi_wgs(const ConvArgs& a) { return wgs_per_group(a, a.epi.G); }
int conv_tail_slots(const ConvArgs& a) { return slots_for_groups(a, a.tail_stats.G); }
// workgroups of one image in the launch
static int wgs_per_image(const ConvArgs& a) {
    const TileCfg c = pick_cfg(a);
    return ((a.Ho + c.th - 1) / c.th) * ((a.Wo + c.tw - 1) / c.tw) * (a.Cout / c.bn) * a.nphase;
}
// block1 may apply its own GroupNorm + SiLU (+ time bias): 3x3, one 8x8 tile per image, every group inside one workgroup
bool conv_act_fits(const ConvArgs& a) {
    if (a.kind != DMN_K3X3 || a.nphase != 1 || a.epi.G < 1 || a.epi.G > kHandoffMaxGroups || a.Cout % a.epi.G) return false;
    const TileCfg c = pick_cfg(a);
    const int cpg = a.Cout / a.epi.G;
    return c.th == 8 && cpg % 4 == 0 && cpg <= c.bn && c.bn % cpg == 0 && wgs_per_group(a, a.epi.G) == 1;
}
bool conv_tail_fits(const ConvArgs& a) {
    if (a.epi.G < 1 || a.epi.G > kHandoffMaxGroups) return false;
    const int slots = conv_epi_slots(a);
    if (wgs_per_group(a, a.epi.G) == 1) return true;   // every group lies inside one workgroup: nothing is exchanged, nothing waits
    ConvArgs t = a;
    t.pro.stats = t.pro.stats ? t.pro.stats : reinterpret_cast<const float*>(1);   // the tail rides on the prologue kernel
    const TileCfg c = pick_cfg(t);
    return slots <= kHandoffMaxSlots && wgs_per_image(a) <= kHandoffMaxWgsPerImage &&
           lds_bytes(t, c, pick_cg(t, c)) <= (size_t)kHandoffMaxLdsBytes;
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
    if (a.tail.gran) {
        DMN_REQUIRE(!a.epi.stats && a.epi.G >= 1 && a.tail.epoch && a.tail.giveups && a.tail_gamma && a.tail_beta && a.res,
                    DMN_ERR_INVALID, "conv: fused block tail needs groups, epoch, parameters and the residual tensor");
        DMN_REQUIRE(conv_tail_fits(a), DMN_ERR_LIMIT, "conv: fused block tail limited to %d groups of %d partial sums",
                    kHandoffMaxGroups, kHandoffMaxSlots);
    }
    if (a.act_gamma) {
        DMN_REQUIRE(!a.epi.stats && !a.tail.gran && !a.res && a.act_beta && conv_act_fits(a), DMN_ERR_INVALID,
                    "conv: the activating epilogue needs a 3x3 convolution whose workgroups hold whole groups of an image");
        a.epi.slots = 1;
        a.epi_wgs = 1;
    }
    if (a.epi.stats || a.tail.gran) {
        DMN_REQUIRE(a.nphase == 1 && a.Cout % a.epi.G == 0 && (a.Cout / a.epi.G) % 4 == 0, DMN_ERR_INVALID,
                    "conv: output statistics need one phase and >= 4 channels per group");
        const int cpg = a.Cout / a.epi.G;
        DMN_REQUIRE(cpg >= c.bn ? cpg % c.bn == 0 : c.bn % cpg == 0, DMN_ERR_INVALID, "conv: group size %d vs tile %d", cpg, c.bn);
        a.epi.slots = conv_epi_slots(a);
        a.epi_wgs = conv_epi_wgs(a);
    }
    if (a.tail.gran && a.tail_stats.stats) {
        DMN_REQUIRE(a.tail_stats.G >= 1 && a.Cout % a.tail_stats.G == 0 && (a.Cout / a.tail_stats.G) % 4 == 0, DMN_ERR_INVALID,
                    "conv: statistics of the block output need >= 4 channels per group");
        const int cpg = a.Cout / a.tail_stats.G;
        DMN_REQUIRE(cpg >= c.bn ? cpg % c.bn == 0 : c.bn % cpg == 0, DMN_ERR_INVALID, "conv: group size %d vs tile %d", cpg, c.bn);
        a.tail_stats.slots = conv_tail_slots(a);
    }
    if (!a.dbg) a.dbg = g_conv_dbg;
    a.cg = pick_cg(a, c);
    const int upp = a.cg / (a.dtype == DMN_F32 ? 4 : 8);
    DMN_REQUIRE(upp <= 256 && (upp & (upp - 1)) == 0, DMN_ERR_INVALID,
                "conv: %d channels per staged group (need a power of two <= %d)", a.cg, a.dtype == DMN_F32 ? 1024 : 2048);
    DMN_REQUIRE(a.cg == Cin || (a.cg / kch) % 4 == 0, DMN_ERR_INVALID, "conv: channel group %d too small for %d channels", a.cg, Cin);
    a.lg_upp = 0;
    while ((1 << a.lg_upp) < upp) ++a.lg_upp;
    // coefficients per thread, without a barrier: one staged channel group, 16-byte aligned parameter vectors
    static const bool lds_coefs = getenv("DMN_CONV_LDS_COEFS") != nullptr;   // A/B: the coefficients through LDS everywhere
    a.pro_direct = a.pro.stats && a.cg == Cin && !lds_coefs && (Cin / a.pro.G) % 4 == 0 &&
                   (((uintptr_t)a.pro.gamma | (uintptr_t)a.pro.beta | (uintptr_t)a.pro.tbias) & 15) == 0 && (a.pro.tb_stride & 3) == 0;
    DMN_REQUIRE((long long)a.B * a.H * a.W * (a.C0 > a.C1 ? a.C0 : a.C1) < (1ll << 31), DMN_ERR_LIMIT,
                "conv: input tensor beyond 2^31 elements");
    DMN_REQUIRE((long long)a.B * a.out_H * a.out_W * a.Cout < (1ll << 31), DMN_ERR_LIMIT, "conv: output tensor beyond 2^31 elements");
    if (a.pro.stats) DMN_REQUIRE(Cin <= 1024, DMN_ERR_LIMIT, "conv: input normalisation limited to 1024 channels");
    const size_t lds = lds_bytes(a, c, a.cg);
    DMN_REQUIRE(lds <= 160 * 1024, DMN_ERR_INVALID, "conv: LDS request %zu too large", lds);
    // a launch whose workgroups wait for each other stays small enough to be resident as a whole (dmn_handoff.h)
    int per_launch = a.B;
    if (a.tail.gran && a.epi_wgs > 1) {
        per_launch = kHandoffMaxWgsPerLaunch / wgs_per_image(a);
        if (per_launch < 1) per_launch = 1;
    }
    for (int b0 = 0; b0 < a.B; b0 += per_launch) {
        a.b_first = b0;
        a.b_count = a.B - b0 < per_launch ? a.B - b0 : per_launch;
        const int rc = a.dtype == DMN_F32 ? dispatch<float>(a, c, lds, s) : dispatch<__bf16>(a, c, lds, s);
        if (rc) return rc;
    }
    return DMN_OK;
}
