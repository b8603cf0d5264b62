// U-Net handle, weight packing and forward orchestration behind the C-ABI (include/dmn.h).
// Follows Unet.forward of the reference (diffusion_model_nemo/modules/unet.py:131-168) for
// use_convnext=False; every tensor op is one of the gfx950 kernels declared in dmn_kernels.h.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <cmath>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/dmn.h"
#include "dmn_kernels.h"
#include "dmn_sampler.h"
#include "dmn_unet_internal.h"

#include <stdlib.h>

// ---- error string ----------------------------------------------------------------------------------
static thread_local char g_err[1024] = "";
void dmn_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* dmn_last_error(void) { return g_err; }
extern "C" const char* dmn_version(void) { return "dmn-hip 0.1 (gfx950)"; }

// ---- host-side packing ---------------------------------------------------------------------------------
static inline uint16_t f32_to_bf16_rne(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (uint16_t)((u >> 16) | 0x40);  // NaN stays NaN
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

static int upload(const void* host, size_t bytes, void** dev) {
    DMN_HIP_CHECK(hipMalloc(dev, bytes));
    DMN_HIP_CHECK(hipMemcpy(*dev, host, bytes, hipMemcpyHostToDevice));
    return DMN_OK;
}
static int upload_f32(const float* host, size_t n, float** dev) { return upload(host, n * 4, (void**)dev); }

// w: [Cout][Cin][kh][kw] (or [Cin][Cout][4][4] when transposed). Packed layout per phase:
// [Cout/16][Cin/kch][ntaps][64 lanes][per], tap = ky*kw + kx (conv) or a*2+b (transposed phase); lane
// (q = lane>>4, r = lane&15) holds output channel 16*tile + r and input channels kstep*kch + q*per + j, i.e.
// one wave-wide 16-byte load yields the MFMA A operand of a (16 cout x kch cin) block.
int pack_conv(PackedConv& pc, int dtype, const float* w, const float* bias, int Cout, int Cin, int kh, int kw, int stride,
              int pad, bool transposed) {
    const int kch = dtype == DMN_F32 ? 16 : 32, per = dtype == DMN_F32 ? 4 : 8;
    DMN_REQUIRE(Cin % kch == 0 && Cout % 16 == 0, DMN_ERR_WEIGHTS, "conv weight: Cin=%d not a multiple of %d or Cout=%d not of 16",
                Cin, kch, Cout);
    pc.Cin = Cin;
    pc.Cout = Cout;
    pc.transposed = transposed;
    pc.pad = pad;
    std::vector<float> packed;
    if (!transposed) {
        if (kh == 3 && kw == 3 && stride == 1 && pad == 1) pc.kind = DMN_K3X3;
        else if (kh == 1 && kw == 1 && stride == 1 && pad == 0) pc.kind = DMN_K1X1;
        else if (kh == 4 && kw == 4 && stride == 2 && pad == 1) pc.kind = DMN_K4X4S2;
        else {
            dmn_set_error("conv %dx%d stride %d pad %d has no kernel specialisation", kh, kw, stride, pad);
            return DMN_ERR_WEIGHTS;
        }
        pc.stride = stride;
        pc.ntaps = kh * kw;
        pc.ext = kh;
        pc.nphase = 1;
        pc.phase[0] = ConvPhase{pad, pad, 0, 0, 0};
        packed.resize((size_t)Cin * Cout * pc.ntaps);
        size_t o = 0;
        for (int ct = 0; ct < Cout / 16; ++ct)
            for (int ch = 0; ch < Cin / kch; ++ch)
                for (int t = 0; t < pc.ntaps; ++t)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int j = 0; j < per; ++j)
                            packed[o++] = w[((size_t)(ct * 16 + (lane & 15)) * Cin + ch * kch + (lane >> 4) * per + j) * pc.ntaps + t];
    } else {
        DMN_REQUIRE(kh == 4 && kw == 4 && stride == 2 && pad == 1, DMN_ERR_WEIGHTS, "transposed conv: only 4x4 s2 p1");
        pc.kind = DMN_KT2X2;
        pc.stride = 1;
        pc.ntaps = 4;
        pc.ext = 2;
        pc.nphase = 4;
        const size_t per_phase = (size_t)Cin * Cout * 4;
        packed.resize(per_phase * 4);
        for (int ph = 0; ph < 4; ++ph) {
            const int py = ph / 2, px = ph % 2;
            pc.phase[ph] = ConvPhase{1 - py, 1 - px, py, px, (long long)(per_phase * ph)};
            size_t o = per_phase * ph;
            for (int ct = 0; ct < Cout / 16; ++ct)
                for (int ch = 0; ch < Cin / kch; ++ch)
                    for (int t = 0; t < 4; ++t) {
                        // output row 2y+py receives input row y-1+dy through kernel row ky (oy = 2 iy - 1 + ky)
                        const int dy = t / 2, dx = t % 2;
                        const int ky = py == 0 ? 3 - 2 * dy : 2 - 2 * dy;
                        const int kx = px == 0 ? 3 - 2 * dx : 2 - 2 * dx;
                        for (int lane = 0; lane < 64; ++lane)
                            for (int j = 0; j < per; ++j)
                                packed[o++] = w[(((size_t)(ch * kch + (lane >> 4) * per + j) * Cout + ct * 16 + (lane & 15)) * 4 + ky) * 4 + kx];
                    }
        }
    }
    int rc;
    if (dtype == DMN_F32) {
        if ((rc = upload(packed.data(), packed.size() * 4, &pc.w))) return rc;
    } else {
        std::vector<uint16_t> h(packed.size());
        for (size_t i = 0; i < packed.size(); ++i) h[i] = f32_to_bf16_rne(packed[i]);
        if ((rc = upload(h.data(), h.size() * 2, &pc.w))) return rc;
    }
    pc.bias = nullptr;
    if (bias && (rc = upload_f32(bias, Cout, &pc.bias))) return rc;
    return DMN_OK;
}

void free_conv(PackedConv& pc) {
    if (pc.w) (void)hipFree(pc.w);
    if (pc.bias) (void)hipFree(pc.bias);
    pc.w = nullptr;
    pc.bias = nullptr;
}

void conv_args_from(const PackedConv& pc, int dtype, ConvArgs& a, int B, int H, int W) {
    memset(&a, 0, sizeof(a));
    a.dtype = dtype;
    a.B = B;
    a.H = H;
    a.W = W;
    a.kind = pc.kind;
    a.nphase = pc.nphase;
    for (int i = 0; i < 4; ++i) a.phase[i] = pc.phase[i];
    a.w = pc.w;
    a.bias = pc.bias;
    a.Cout = pc.Cout;
    if (pc.transposed) {
        a.Ho = H;
        a.Wo = W;
        a.out_H = 2 * H;
        a.out_W = 2 * W;
        a.out_sy = a.out_sx = 2;
    } else {
        a.Ho = (H + 2 * pc.pad - pc.ext) / pc.stride + 1;
        a.Wo = (W + 2 * pc.pad - pc.ext) / pc.stride + 1;
        a.out_H = a.Ho;
        a.out_W = a.Wo;
        a.out_sy = a.out_sx = 1;
    }
}

// ---- workspace arena: first-fit blocks over the caller's buffer -------------------------------------
struct Arena {
    struct Blk {
        size_t off, size;
        bool used;
    };
    char* base = nullptr;
    size_t cap = 0, high = 0;
    bool dry = false;
    std::vector<Blk> blks;
    void reset(void* b, size_t c, bool d) {
        base = d ? reinterpret_cast<char*>(4096) : (char*)b;   // dry run: offsets only, any non-null base
        cap = c;
        dry = d;
        high = 0;
        blks.clear();
        blks.push_back(Blk{0, dry ? (size_t)1 << 60 : c, false});
    }
    void* alloc(size_t n) {
        n = (n + 255) & ~(size_t)255;
        for (size_t i = 0; i < blks.size(); ++i) {
            if (!blks[i].used && blks[i].size >= n) {
                if (blks[i].size > n) {
                    Blk rest{blks[i].off + n, blks[i].size - n, false};
                    blks[i].size = n;
                    blks.insert(blks.begin() + i + 1, rest);
                }
                blks[i].used = true;
                if (blks[i].off + n > high) high = blks[i].off + n;
                return base + blks[i].off;   // in a dry run only the offsets matter
            }
        }
        return nullptr;
    }
    void release(void* p) {
        if (!p) return;
        const size_t off = (char*)p - base;
        for (size_t i = 0; i < blks.size(); ++i)
            if (blks[i].off == off && blks[i].used) {
                blks[i].used = false;
                if (i + 1 < blks.size() && !blks[i + 1].used) {
                    blks[i].size += blks[i + 1].size;
                    blks.erase(blks.begin() + i + 1);
                }
                if (i > 0 && !blks[i - 1].used) {
                    blks[i - 1].size += blks[i].size;
                    blks.erase(blks.begin() + i);
                }
                return;
            }
    }
};

struct Act {            // an NHWC activation + optional statistics of it
    void* p = nullptr;
    int C = 0, H = 0, W = 0;
    float* stats = nullptr;
    int sG = 0, sSlots = 0;
    bool activated = false;   // block1.proj output that already went through the block's GroupNorm + SiLU + time bias
};

// GroupNorm + SiLU (+ time bias) that a convolution may apply to its own output (ConvArgs::act_*)
struct ActEpi {
    const float *gamma, *beta, *tbias;
    int tb_stride;
};
static void set_act_epilogue(ConvArgs& a, const ActEpi& e, int G) {
    a.act_gamma = e.gamma;
    a.act_beta = e.beta;
    a.act_tbias = e.tbias;
    a.act_tb_stride = e.tb_stride;
    a.act_inv_count = 1.0f / ((float)(a.Cout / G) * a.out_H * a.out_W);
}

constexpr int kTraceStop = 1;   // Fwd::rc of a traced forward that was asked to end after a given launch

struct Fwd {
    DmnUnet* u;
    Arena ar;
    hipStream_t s;
    int B;
    bool dry;
    int rc = DMN_OK;
    size_t esz() const { return u->cfg.dtype == DMN_F32 ? 4 : 2; }
    void* alloc(size_t n) {
        void* p = ar.alloc(n);
        if (!p && rc == DMN_OK) {
            dmn_set_error("workspace too small (need more than %zu bytes)", ar.cap);
            rc = DMN_ERR_WORKSPACE;
        }
        return p;
    }
    Act new_act(int C, int H, int W) {
        Act a;
        a.C = C;
        a.H = H;
        a.W = W;
        a.p = alloc((size_t)B * H * W * C * esz());
        return a;
    }
    float* new_stats(int G, int slots) { return (float*)alloc((size_t)B * G * slots * 4 * sizeof(float)); }   // {sum, m2, n, -} per slot
    void free_act(Act& a) {
        ar.release(a.p);
        ar.release(a.stats);
        a.p = nullptr;
        a.stats = nullptr;
    }
    int planB() const { return u->plan_batch > 0 ? u->plan_batch : B; }
    // tuning aid: the convolution launches of an evaluation are numbered; the one selected by DmnUnet::probe_site gets the
    // per-workgroup phase stamps (in place, also inside captured steps)
    int conv_seq = 0;
    long long* site_stamps() {
        const int seq = conv_seq++;
        if (dry || u->probe_on != 2 || !u->probe_stamps || u->probe_site < 0 || seq != u->probe_site) return nullptr;
        return u->probe_stamps + (size_t)u->chain_id * kProbeMaxWgs * 8;
    }
    bool ok() const { return rc == DMN_OK; }
    // granule area of this chain (dmn_handoff.h): every fused launch takes the next range. The dry run only adds up;
    // a real run returns nullptr when the area is missing or too small and the caller takes the two-launch form.
    char* ho_base = nullptr;
    size_t ho_cap = 0, ho_off = kHandoffHeaderBytes;
    bool ho_enabled = false;
    unsigned long long* handoff_take(size_t bytes) {
        bytes = (bytes + 255) & ~(size_t)255;
        const size_t at = ho_off;
        ho_off += bytes;
        if (dry || !ho_base || at + bytes > ho_cap) return nullptr;
        return reinterpret_cast<unsigned long long*>(ho_base + at);
    }
    HandoffArgs handoff_args(unsigned long long* gran) const {
        HandoffArgs h;
        h.gran = gran;
        h.epoch = reinterpret_cast<const unsigned*>(ho_base);
        h.giveups = reinterpret_cast<unsigned*>(ho_base) + 1;
        return h;
    }
    // every launch of the forward reports here; with a trace buffer attached, a checksum of the whole workspace is
    // taken behind it (slot k = state after launch k), which locates the first launch whose result differs between runs
    void run(int r, const char* what) {
        if (rc == DMN_OK && r != DMN_OK) rc = r;
        if (u->trace_slots && !dry && rc == DMN_OK && u->trace_n < u->trace_max) {
            if ((int)u->trace_names.size() <= u->trace_n) u->trace_names.resize(u->trace_n + 1);
            u->trace_names[u->trace_n] = what;
            unsigned long long* slot = u->trace_slots + u->trace_n++;
            if (hipMemsetAsync(slot, 0, 8, s) != hipSuccess) rc = DMN_ERR_HIP;
            else rc = launch_checksum(ar.base, ar.cap, slot, s);
            if (rc == DMN_OK && u->trace_poison) rc = launch_lds_poison(u->trace_poison, s);   // in front of the next launch
            if (rc == DMN_OK && u->trace_n - 1 == u->trace_stop) rc = kTraceStop;   // skips every later launch
        }
    }
};

static NormIn norm_in(const Act& a, const float* gamma, const float* beta, int B, int silu, const float* tbias, int tb_stride) {
    NormIn n;
    memset(&n, 0, sizeof(n));
    n.stats = a.stats;
    n.G = a.sG;
    n.slots = a.sSlots;
    n.inv_count = 1.0f / ((float)(a.C / a.sG) * a.H * a.W);
    n.gamma = gamma;
    n.beta = beta;
    n.tbias = tbias;
    n.tb_stride = tb_stride;
    n.silu = silu;
    return n;
}

// conv with optional: second source, input normalisation, residual, output statistics
// Launches a convolution of the network, with the measurement hooks around it: the launch a probe site selects (or,
// without a site, the dominant full-resolution one: block2.proj of downs.0.0) is timed by events (eager runs) or
// clocked by its own workgroups (also inside captured steps).
static void run_conv(Fwd& f, const PackedConv& pc, ConvArgs& a, const char* what) {
    DmnUnet* u = f.u;
    long long* const site = f.site_stamps();   // non-null: this launch is the one a probe site selects
    if (site) a.dbg = site;
    const bool target = u->probe_on && (u->probe_site < 0 ? &pc == &u->downs[0].rb[0].c2 : site != nullptr);
    bool probe = target && u->probe_on == 1 && u->probe_n < kProbePairs;
    if (probe) {   // never inside a stream capture: a captured event record carries no time
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        probe = hipStreamIsCapturing(f.s, &cs) == hipSuccess && cs == hipStreamCaptureStatusNone;
    }
    if (probe) {   // two markers back to back (what a marker itself costs), then the launch, then a third
        (void)hipEventRecord(u->probe_ev[3 * u->probe_n], f.s);
        (void)hipEventRecord(u->probe_ev[3 * u->probe_n + 1], f.s);
    }
    // device-clock form (also inside captured steps): every workgroup of this launch stamps its start and end
    const bool clocked = target && u->probe_on == 2 && u->probe_stamps;
    long long* stamps = clocked ? u->probe_stamps + (size_t)u->chain_id * kProbeMaxWgs * 8 : nullptr;   // per launch chain
    if (clocked) a.dbg = stamps;
    f.run(launch_conv(a, f.s), what);
    if (clocked) {
        const int tiles = ((a.Ho + 15) / 16) * ((a.Wo + 15) / 16);   // upper bound of the launch's workgroups (8x8 tiles: 4x)
        int wgs = tiles * 4 * (a.Cout / 32) * f.B * a.nphase;
        if (wgs > kProbeMaxWgs) wgs = kProbeMaxWgs;
        f.run(launch_stamp_span(stamps, wgs, u->probe_acc + 2 * u->chain_id, f.s), "probe: span of the stamps");
    }
    if (probe || clocked) {
        // What the clocked launch must move: input, output, weights, and the residual tensor where it closes its block.
        // A launch whose workgroups exchange is issued in pieces of at most 256 workgroups (launch_conv): the stamps
        // then hold the last piece, of a.b_count images.
        const int nb = a.b_count > 0 ? a.b_count : f.B;
        u->probe_bytes = ((double)nb * a.H * a.W * pc.Cin + (double)nb * a.out_H * a.out_W * pc.Cout * (a.res ? 2 : 1)) * f.esz() +
                         (double)pc.Cin * pc.Cout * pc.ntaps * f.esz();
        u->probe_shape[0] = 2.0 * nb * a.out_H * a.out_W * pc.Cin * pc.Cout * pc.ntaps;
        u->probe_shape[1] = pc.Cin;
        u->probe_shape[2] = pc.Cout;
        u->probe_shape[3] = a.out_H;
        u->probe_shape[4] = pc.ntaps;
    }
    if (probe) {
        (void)hipEventRecord(u->probe_ev[3 * u->probe_n + 2], f.s);
        ++u->probe_n;
    }
}

static Act do_conv(Fwd& f, const PackedConv& pc, const Act& in0, const Act* in1, const NormIn* pro, const Act* res, int epiG,
                   const PackedConv* shortcut = nullptr, Act* shortcut_out = nullptr, const void* pro_res = nullptr,
                   void* pro_out = nullptr, const ActEpi* act = nullptr) {
    ConvArgs a;
    conv_args_from(pc, f.u->cfg.dtype, a, f.B, in0.H, in0.W);
    a.plan_B = f.planB();
    a.in0 = in0.p;
    a.C0 = in0.C;
    if (in1) {
        a.in1 = in1->p;
        a.C1 = in1->C;
    }
    if (pro) a.pro = *pro;
    if (pro_res) {   // the producer's tail (normalisation of `pro`, then + pro_res) is applied while staging
        a.pro_res = pro_res;
        a.pro_out = pro_out;
    }
    Act out = f.new_act(pc.Cout, a.out_H, a.out_W);
    a.out = out.p;
    if (res) a.res = res->p;
    if (shortcut) {   // 1x1 convolution of the same input, computed by the same launch
        *shortcut_out = f.new_act(pc.Cout, a.out_H, a.out_W);
        a.w2 = shortcut->w;
        a.bias2 = shortcut->bias;
        a.out2 = shortcut_out->p;
    }
    if (epiG > 0) {
        a.epi.G = epiG;
        if (act && conv_act_fits(a)) {   // the launch normalises and activates its own output: no statistics to hand on
            set_act_epilogue(a, *act, epiG);
            out.activated = true;
        } else {
            out.sG = epiG;
            out.sSlots = conv_epi_slots(a);
            out.stats = f.new_stats(epiG, out.sSlots);
            a.epi.stats = out.stats;
        }
    }
    if (f.ok() && !f.dry) {
        char what[160];
        snprintf(what, sizeof(what), "conv kind %d %d->%d at %dx%d%s%s%s out@%zu stats@%zu", pc.kind, pc.Cin, pc.Cout, in0.H,
                 in0.W, pro ? (pro->silu ? " norm+silu" : " norm") : "", res ? " +res" : "", shortcut ? " +shortcut" : "",
                 (size_t)((char*)a.out - f.ar.base), out.stats ? (size_t)((char*)out.stats - f.ar.base) : (size_t)0);
        run_conv(f, pc, a, what);
    }
    return out;
}

// A ResnetBlock whose closing `SiLU(GroupNorm(h2)) + res` has not been applied yet: the next block's first
// convolution applies it while staging its input and writes the result to `out` on the way.
struct LazyTail {
    bool active = false;
    Act h2, r, out;           // block2.proj output with statistics; shortcut tensor (when has_r); target of the tail
    bool has_r = false;
    const void* res = nullptr;   // residual operand: r.p or the block input
    const float *g2 = nullptr, *b2 = nullptr;
};

// ResnetBlock (parts/convnext.py:78-86). statsG > 0: also produce GroupNorm(statsG) statistics of the output.
// `pending`: the producer of x left its tail to this block's first convolution (x.p is allocated, not written).
// `defer`: leave this block's tail to the consumer (returned Act is allocated, not written; no statistics).
static Act resblock(Fwd& f, const ResBlockW& w, Act& x, Act* x2, const float* tb, int tb_stride, int statsG,
                    LazyTail* pending = nullptr, LazyTail* defer = nullptr) {
    const int G = f.u->cfg.groups;
    Act r;
    const Act* resp = &x;
    if (w.has_res) resp = &r;   // res_conv(x) comes out of block1's launch (same input tile)
    // block1 applies its own GroupNorm + SiLU + time bias where a workgroup holds whole groups of its image (8x8 pixels);
    // block2.proj then reads a finished tensor
    const ActEpi act1 = {w.g1, w.b1, w.tb_off >= 0 ? tb + w.tb_off : nullptr, tb_stride};
    const bool want_act = !(f.u->path_mask & 4096);
    Act h1;
    if (pending && pending->active) {
        // conv1 reads SiLU(GN(h2_prev)) + res_prev and materialises it into x.p
        ConvArgs a;
        conv_args_from(w.c1, f.u->cfg.dtype, a, f.B, x.H, x.W);
        a.plan_B = f.planB();
        a.in0 = pending->h2.p;
        a.C0 = x.C;
        a.pro = norm_in(pending->h2, pending->g2, pending->b2, f.B, 1, nullptr, 0);
        a.pro_res = pending->res;
        a.pro_out = x.p;
        h1 = f.new_act(w.c1.Cout, a.out_H, a.out_W);
        a.out = h1.p;
        a.epi.G = G;
        if (want_act && conv_act_fits(a)) {
            set_act_epilogue(a, act1, G);
            h1.activated = true;
        } else {
            h1.sG = G;
            h1.sSlots = conv_epi_slots(a);
            h1.stats = f.new_stats(G, h1.sSlots);
            a.epi.stats = h1.stats;
        }
        if (long long* st = f.site_stamps()) a.dbg = st;
        if (f.ok() && !f.dry) f.run(launch_conv(a, f.s), "conv1 + previous block's tail");
        f.free_act(pending->h2);
        if (pending->has_r) f.free_act(pending->r);
        pending->active = false;
    } else {
        h1 = do_conv(f, w.c1, x, x2, nullptr, nullptr, G, w.has_res ? &w.res : nullptr, w.has_res ? &r : nullptr, nullptr, nullptr,
                     want_act ? &act1 : nullptr);
    }
    NormIn n1;
    memset(&n1, 0, sizeof(n1));
    if (!h1.activated) n1 = norm_in(h1, w.g1, w.b1, f.B, 1, act1.tbias, tb_stride);
    if (!defer && f.ho_enabled && w.c2.kind == DMN_K3X3) {
        // block2.proj closes the block itself: the image's workgroups exchange their GroupNorm partial sums inside
        // the launch, so neither the convolution's output nor a closing pass over it exists
        ConvArgs a;
        conv_args_from(w.c2, f.u->cfg.dtype, a, f.B, h1.H, h1.W);
        a.plan_B = f.planB();
        a.in0 = h1.p;
        a.C0 = h1.C;
        a.pro = n1;
        a.epi.G = G;
        // full-resolution images may take the tall tile that brings them under the exchange's limit (DMN_UNET_PATHS bit
        // 131072: never; block_tail_is_local asks with the same flag)
        a.tall_tail = !(f.u->path_mask & 131072) && !h1.activated;
        if (conv_tail_fits(a)) {
            unsigned long long* gran = f.handoff_take((size_t)f.B * G * conv_epi_slots(a) * kHandoffGranulesPerSlot * sizeof(unsigned long long));
            if (gran || f.dry) {
                Act out = f.new_act(w.c2.Cout, a.out_H, a.out_W);
                if (f.dry) {   // plan for the two-launch form as well (its peak is the larger one)
                    Act h2 = f.new_act(w.c2.Cout, a.out_H, a.out_W);
                    float* st = f.new_stats(G, conv_epi_slots(a));
                    f.ar.release(st);
                    f.free_act(h2);
                }
                a.out = out.p;
                a.res = resp->p;
                a.tail = f.handoff_args(gran);
                a.tail_gamma = w.g2;
                a.tail_beta = w.b2;
                a.tail_inv_count = 1.0f / ((float)(w.c2.Cout / G) * a.out_H * a.out_W);
                if (statsG > 0) {
                    a.tail_stats.G = statsG;
                    out.sG = statsG;
                    out.sSlots = conv_tail_slots(a);
                    out.stats = f.new_stats(statsG, out.sSlots);
                    a.tail_stats.stats = out.stats;
                }
                if (f.ok() && !f.dry) run_conv(f, w.c2, a, "conv2 + block tail (in-launch statistics)");
                else (void)f.site_stamps();
                f.free_act(h1);
                if (w.has_res) f.free_act(r);
                return out;
            }
        }
    }
    Act h2 = do_conv(f, w.c2, h1, nullptr, h1.activated ? nullptr : &n1, nullptr, G);
    f.free_act(h1);
    Act out = f.new_act(h2.C, h2.H, h2.W);
    if (defer) {
        defer->active = true;
        defer->h2 = h2;
        defer->has_r = w.has_res;
        defer->r = r;
        defer->res = resp->p;
        defer->g2 = w.g2;
        defer->b2 = w.b2;
        return out;
    }
    NormApplyArgs na;
    memset(&na, 0, sizeof(na));
    na.dtype = f.u->cfg.dtype;
    na.in = h2.p;
    na.out = out.p;
    na.res = resp->p;
    na.B = f.B;
    na.HW = h2.H * h2.W;
    na.C = h2.C;
    na.norm = norm_in(h2, w.g2, w.b2, f.B, 1, nullptr, 0);
    na.plan_B = f.planB();
    if (statsG > 0) {
        out.sG = statsG;
        out.sSlots = norm_apply_slots(na.plan_B, na.HW, na.C);
        out.stats = f.new_stats(statsG, out.sSlots);
        na.epi.stats = out.stats;
        na.epi.G = statsG;
    }
    if (f.ok() && !f.dry) f.run(launch_norm_apply(na, f.s), "block tail");
    f.free_act(h2);
    if (w.has_res) f.free_act(r);
    return out;
}

// Whether the first ResnetBlock of a level closes itself inside block2.proj (in-launch statistics, or none to exchange
// where every GroupNorm group of an image lies inside one workgroup: 8x8 pixels) instead of leaving its tail to the
// second block's first convolution, whose workgroups would each redo the closing arithmetic, with two operand reads,
// on their whole halo tile. Measured with two launch chains on MI355X: -1.6 % of the step for the 8x8 levels, -1.8 %
// more for the 16x16 and 32x32 levels (where the exchange costs 2-3 us per launch).
static bool block_tail_is_local(Fwd& f, const ResBlockW& w, int H, int W) {
    if (!f.ho_enabled || w.c2.kind != DMN_K3X3 || (f.u->path_mask & 8192)) return false;
    ConvArgs a;
    conv_args_from(w.c2, f.u->cfg.dtype, a, f.B, H, W);
    a.plan_B = f.planB();
    a.C0 = w.c2.Cin;
    a.epi.G = f.u->cfg.groups;
    a.tall_tail = !(f.u->path_mask & 131072);
    return conv_tail_fits(a) && (conv_epi_wgs(a) == 1 || !(f.u->path_mask & 16384));
}

// A linear attention whose closing `GroupNorm(y) + x` has not been applied: the resampling convolution that consumes it
// applies it while staging its input (and stores the result when a skip connection needs the tensor).
struct LazyAttn {
    bool active = false;
    Act y;                        // to_out output with GroupNorm(1) statistics
    const float *og = nullptr, *ob = nullptr;
};

static Act lin_attn_tail(Fwd& f, const AttnW& w, Act& x, Act& y, LazyAttn* defer) {
    if (defer) {
        defer->active = true;
        defer->y = y;
        defer->og = w.og;
        defer->ob = w.ob;
        return Act();
    }
    Act out = f.new_act(x.C, x.H, x.W);
    NormApplyArgs na;
    memset(&na, 0, sizeof(na));
    na.dtype = f.u->cfg.dtype;
    na.in = y.p;
    na.out = out.p;
    na.res = x.p;
    na.B = f.B;
    na.HW = x.H * x.W;
    na.C = x.C;
    na.norm = norm_in(y, w.og, w.ob, f.B, 0, nullptr, 0);
    na.plan_B = f.planB();
    if (f.ok() && !f.dry) f.run(launch_norm_apply(na, f.s), "linear attention tail");
    f.free_act(y);
    return out;
}

// Residual(PreNorm(dim, LinearAttention(dim))) (utils.py:68-93, parts/mha.py:44-59). x must carry GN(1) statistics.
static Act lin_attn(Fwd& f, const AttnW& w, Act& x, LazyAttn* defer = nullptr) {
    const DmnUnetCfg& c = f.u->cfg;
    const int hidden = c.heads * c.dim_head;
    if (w.fq && !(f.u->path_mask & 2)) {
        LinAttnFusedArgs fa;
        memset(&fa, 0, sizeof(fa));
        fa.dtype = c.dtype;
        fa.x = x.p;
        fa.norm = norm_in(x, w.ng, w.nb, f.B, 0, nullptr, 0);
        fa.wq = w.fq;
        fa.wk = w.fk;
        fa.wv = w.fv;
        fa.wo = w.fo;
        fa.bias = w.out.bias;
        fa.B = f.B;
        fa.n = x.H * x.W;
        fa.C = x.C;
        linattn_fused_plan(fa.n, f.planB(), fa.C, c.dtype, &fa.slice1, &fa.splits, &fa.tpw, &fa.slices);
        fa.separate_merge = (f.u->path_mask & 1024) != 0;
        fa.separate_ctx = (f.u->path_mask & 512) != 0;
        if (!defer && f.ho_enabled && !(f.u->path_mask & 256) && linattn_tail_fits(fa.n, f.planB(), fa.C, c.dtype)) {
            // the second launch closes the attention itself (GroupNorm(1)(y) + x, partial sums exchanged inside the launch)
            unsigned long long* gran = f.handoff_take((size_t)f.B * 4 * fa.slices * kHandoffGranulesPerSlot * sizeof(unsigned long long));   // a partial per wave
            if (gran || f.dry) {
                Act out = f.new_act(x.C, x.H, x.W);
                float* part = (float*)f.alloc((size_t)f.B * 4 * fa.splits * (32 * 32 + 64) * 4);
                void* ctxT = f.alloc((size_t)f.B * 4 * 32 * 32 * f.esz());
                if (f.dry) {   // plan for the three-launch form as well (its peak is the larger one)
                    Act y = f.new_act(x.C, x.H, x.W);
                    float* st = f.new_stats(1, 4 * fa.slices);
                    f.ar.release(st);
                    f.free_act(y);
                }
                fa.ctxT = ctxT;
                fa.ctx_part = part;
                fa.out = out.p;
                fa.tail = f.handoff_args(gran);
                fa.tail_gamma = w.og;
                fa.tail_beta = w.ob;
                fa.tail_inv_count = 1.0f / ((float)x.C * x.H * x.W);
                if (f.ok() && !f.dry) f.run(launch_linattn_fused(fa, f.s), "fused linear attention + closing GroupNorm (in-launch statistics)");
                f.ar.release(part);
                f.ar.release(ctxT);
                return out;
            }
        }
        Act y = f.new_act(x.C, x.H, x.W);
        y.sG = 1;
        y.sSlots = 4 * fa.slices;   // a partial per wave of every pixel slice
        y.stats = f.new_stats(1, y.sSlots);
        float* part = (float*)f.alloc((size_t)f.B * 4 * fa.splits * (32 * 32 + 64) * 4);
        void* ctxT = f.alloc((size_t)f.B * 4 * 32 * 32 * f.esz());
        fa.ctxT = ctxT;
        fa.y = y.p;
        fa.y_stats = y.stats;
        fa.ctx_part = part;
        char what[160];
        snprintf(what, sizeof(what), "fused linear attention C=%d n=%d y@%zu stats@%zu part@%zu ctxT@%zu", fa.C, fa.n,
                 (size_t)((char*)y.p - f.ar.base), (size_t)((char*)y.stats - f.ar.base), (size_t)((char*)part - f.ar.base),
                 (size_t)((char*)ctxT - f.ar.base));
        if (f.ok() && !f.dry) f.run(launch_linattn_fused(fa, f.s), what);
        f.ar.release(part);
        f.ar.release(ctxT);
        return lin_attn_tail(f, w, x, y, defer);
    }
    NormIn pn = norm_in(x, w.ng, w.nb, f.B, 0, nullptr, 0);
    Act qkv = do_conv(f, w.qkv, x, nullptr, &pn, nullptr, 0);
    Act la = f.new_act(hidden, x.H, x.W);
    LinAttnArgs aa;
    memset(&aa, 0, sizeof(aa));
    aa.dtype = c.dtype;
    aa.qkv = qkv.p;
    aa.out = la.p;
    aa.B = f.B;
    aa.n = x.H * x.W;
    aa.heads = c.heads;
    aa.d = c.dim_head;
    aa.splits = linear_attention_splits(aa.n);
    float* part = (float*)f.alloc((size_t)f.B * c.heads * aa.splits * (c.dim_head * c.dim_head + 2 * c.dim_head) * 4);
    aa.ctx_part = part;
    if (f.ok() && !f.dry) f.run(launch_linear_attention(aa, f.s), "linear attention core");
    f.free_act(qkv);
    f.ar.release(part);
    Act y = do_conv(f, w.out, la, nullptr, nullptr, nullptr, 1);
    f.free_act(la);
    return lin_attn_tail(f, w, x, y, defer);
}

// Residual(PreNorm(dim, Attention(dim))) (parts/mha.py:16-30)
static Act full_attn(Fwd& f, const AttnW& w, Act& x) {
    const DmnUnetCfg& c = f.u->cfg;
    const int hidden = c.heads * c.dim_head;
    NormIn pn = norm_in(x, w.ng, w.nb, f.B, 0, nullptr, 0);
    Act qkv = do_conv(f, w.qkv, x, nullptr, &pn, nullptr, 0);
    Act o = f.new_act(hidden, x.H, x.W);
    AttnArgs aa;
    memset(&aa, 0, sizeof(aa));
    aa.dtype = c.dtype;
    aa.qkv = qkv.p;
    aa.out = o.p;
    aa.B = f.B;
    aa.n = x.H * x.W;
    aa.heads = c.heads;
    aa.d = c.dim_head;
    aa.no_mfma = (f.u->path_mask & 8) != 0;
    if (f.ok() && !f.dry) f.run(launch_attention(aa, f.s), "attention core");
    f.free_act(qkv);
    Act out = do_conv(f, w.out, o, nullptr, nullptr, &x, 0);
    f.free_act(o);
    return out;
}

// The whole forward. `t`/`t_is_float` or (`table`, `step_idx`) provide the time label.
int unet_reserve_handoff(DmnUnet* u, size_t bytes) {
    if (bytes <= u->handoff_bytes && u->handoff_mem[0] && u->handoff_mem[1]) return DMN_OK;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;   // a capture on the legacy stream forbids allocation
    DMN_REQUIRE(!(hipStreamIsCapturing(nullptr, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone), DMN_ERR_INVALID,
                "the in-launch exchange areas cannot grow inside a stream capture: size the workspace for this shape before capturing");
    const size_t want = bytes < ((size_t)1 << 20) ? ((size_t)1 << 20) : bytes + bytes / 2;
    DMN_HIP_CHECK(hipDeviceSynchronize());   // evaluations in flight still poll the old areas
    if (u->graph_exec) {                     // captured steps hold the old addresses
        (void)hipGraphExecDestroy(u->graph_exec);
        if (u->graph_exec2) (void)hipGraphExecDestroy(u->graph_exec2);
        u->graph_exec = u->graph_exec2 = nullptr;
    }
    for (int c = 0; c < 2; ++c) {
        if (u->handoff_mem[c]) (void)hipFree(u->handoff_mem[c]);
        u->handoff_mem[c] = nullptr;
    }
    u->handoff_bytes = 0;
    for (int c = 0; c < 2; ++c) {
        DMN_HIP_CHECK(hipMalloc(&u->handoff_mem[c], want));
        DMN_HIP_CHECK(hipMemset(u->handoff_mem[c], 0, want));   // epoch 0, no granule carries a tag
    }
    DMN_HIP_CHECK(hipDeviceSynchronize());
    u->handoff_bytes = want;
    u->epoch_used = 0;
    return DMN_OK;
}

int unet_epoch_reserve(DmnUnet* u, unsigned long long evals, hipStream_t s) {
    if (!u->handoff_mem[0] || !u->handoff_mem[1]) return DMN_OK;
    if (u->epoch_used + evals < u->epoch_limit) {
        u->epoch_used += evals;
        return DMN_OK;
    }
    // Every earlier loop of this handle has joined `s`; once `s` is idle nothing polls the areas, and what the coming
    // work finds is epoch 0 and no tag at all, as after allocation. The give-up counts go to the host first.
    DMN_HIP_CHECK(hipStreamSynchronize(s));
    for (int c = 0; c < 2; ++c) {
        unsigned hdr[2] = {0, 0};
        DMN_HIP_CHECK(hipMemcpy(hdr, u->handoff_mem[c], sizeof(hdr), hipMemcpyDeviceToHost));
        u->giveups_carry += hdr[1];
        DMN_HIP_CHECK(hipMemsetAsync(u->handoff_mem[c], 0, u->handoff_bytes, s));
    }
    u->epoch_used = evals;
    return DMN_OK;
}

// ---- which handles have exchanging work in flight ---------------------------------------------------------------
namespace {
struct GuardEntry {
    DmnUnet* u;
    int device;
    bool in_call;          // a host call of the handle is issuing launches (and may be capturing its stream)
    bool enqueuing;        // ... and those launches exchange
    bool pending;          // `done` has been recorded and not yet been seen complete
    hipEvent_t done;
    hipStream_t done_on;   // the stream `done` was last recorded on
};
std::mutex g_guard_mu;
std::vector<GuardEntry> g_guard;
}  // namespace

bool exchange_guard_enter(DmnUnet* u) {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    std::lock_guard<std::mutex> lock(g_guard_mu);
    GuardEntry* mine = nullptr;
    bool busy = false;
    for (auto& e : g_guard) {
        if (e.u == u) {
            mine = &e;
            continue;
        }
        if (e.device != dev) continue;
        // An event may not be queried while the stream it was last recorded on is being captured: the query fails, its
        // error stays behind in this thread, and the other thread's capture is invalidated. A handle captures only
        // inside its own calls (in_call); a caller's own capture around dmn_unet_forward shows in the stream's state.
        // Either way the entry then counts as still pending.
        if (e.pending && !e.in_call) {
            hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
            if (hipStreamIsCapturing(e.done_on, &st) == hipSuccess && st == hipStreamCaptureStatusNone && hipEventQuery(e.done) == hipSuccess)
                e.pending = false;
            (void)hipGetLastError();   // "not ready" is an answer, not a failure of this thread's next launch
        }
        if (e.enqueuing || e.pending) busy = true;
    }
    if (!mine) {
        g_guard.push_back(GuardEntry{u, dev, false, false, false, nullptr, nullptr});
        mine = &g_guard.back();
    }
    mine->device = dev;
    mine->in_call = true;
    mine->enqueuing = !busy;
    return !busy;
}

// `entered`: what exchange_guard_enter returned for this call (every enter is followed by one leave).
void exchange_guard_leave(DmnUnet* u, hipStream_t s, bool entered) {
    std::lock_guard<std::mutex> lock(g_guard_mu);
    for (auto& e : g_guard)
        if (e.u == u) {
            e.in_call = false;
            e.enqueuing = false;
            if (!entered) return;   // nothing of this call exchanges; an earlier call's event stays as it is
            if (!e.done && hipEventCreateWithFlags(&e.done, hipEventDisableTiming) != hipSuccess) e.done = nullptr;
            // without an event the work cannot be followed: it then counts as finished (the spins stay bounded)
            e.pending = e.done && hipEventRecord(e.done, s) == hipSuccess;
            e.done_on = s;
            return;
        }
}

void exchange_guard_forget(DmnUnet* u) {
    std::lock_guard<std::mutex> lock(g_guard_mu);
    for (size_t i = 0; i < g_guard.size(); ++i)
        if (g_guard[i].u == u) {
            if (g_guard[i].done) (void)hipEventDestroy(g_guard[i].done);
            g_guard.erase(g_guard.begin() + i);
            return;
        }
}

int unet_forward_impl(DmnUnet* u, const float* x, const void* t, int t_is_float, const float* table, int* step_idx,
                      float* out, int B, int H, int W, void* ws, size_t ws_bytes, hipStream_t s, bool dry, size_t* high,
                      size_t* handoff_need, const DdpmStepArgs* fuse_step) {
    const DmnUnetCfg& c = u->cfg;
    DMN_REQUIRE(u->loaded || dry, DMN_ERR_WEIGHTS, "dmn_unet_forward: weights not loaded");
    const int nlev = c.n_mults;
    const int div = 1 << (nlev - 1);
    DMN_REQUIRE(B >= 1 && H % div == 0 && W % div == 0 && H >= div && W >= div, DMN_ERR_INVALID,
                "unet: image %dx%d must be a multiple of %d", H, W, div);
    Fwd f;
    f.u = u;
    f.s = s;
    f.B = B;
    f.dry = dry;
    f.ar.reset(ws, ws_bytes, dry);
    f.ho_enabled = !(u->path_mask & 128) && !u->exchange_off;
    if (!dry && f.ho_enabled) {
        f.ho_base = static_cast<char*>(u->handoff_mem[u->chain_id & 1]);
        f.ho_cap = u->handoff_bytes;
        if (!f.ho_base) f.ho_enabled = false;
    }
    if (!dry) {
        u->trace_n = 0;
        if (u->trace_slots && u->trace_poison) (void)launch_lds_poison(u->trace_poison, s);
    }

    float* tb = nullptr;
    int tb_rows = B;
    TimeArgs ta;
    memset(&ta, 0, sizeof(ta));
    if (c.with_time_emb) {
        // in the sampling loops every image shares the step's label: one row, broadcast with stride 0
        tb_rows = (t == nullptr && table != nullptr) ? 1 : B;
        tb = (float*)f.alloc((size_t)tb_rows * u->tb_total * 4);
        ta.t = t;
        ta.t_is_float = t_is_float;
        ta.table = table;
        ta.step_idx = step_idx;
        // the counter is copied 16 ints further on: kernels behind the network may read the copy while the last of
        // them already writes the next value to the original
        ta.step_copy = step_idx ? step_idx + kStepCopyOffset : nullptr;
        ta.table_cols = DMN_TABLE_COLS;
        ta.label_col = 0;
        ta.rows = tb_rows;
        ta.dim = c.dim;
        ta.freqs = u->freqs;
        ta.w1 = u->tw1; ta.b1 = u->tb1;
        ta.w2 = u->tw2; ta.b2 = u->tb2;
        ta.wb = u->twb; ta.bb = u->tbb;
        ta.nout = u->tb_total;
        ta.out = tb;
        // a sampling loop has tabulated the hidden vectors of its table (unet_time_hidden_table)
        if (t == nullptr && table != nullptr && table == u->h2_table_of) ta.h2 = u->h2_table;
    }
    const int tbs = tb_rows == 1 ? 0 : u->tb_total;

    Act cur = f.new_act(c.dim, H, W);
    {
        IngestArgs ia;
        memset(&ia, 0, sizeof(ia));
        ia.dtype = c.dtype;
        ia.x = x;
        ia.w = u->init_w;
        ia.wfrag = u->init_wfrag;
        ia.bias = u->init_b;
        ia.out = cur.p;
        ia.B = B; ia.Cin = c.channels; ia.H = H; ia.W = W; ia.Cout = c.dim;
        if (c.num_classes > 0) {
            ia.class_embed = u->class_embed;
            ia.labels = u->labels ? u->labels + u->label_offset : nullptr;
            ia.num_classes = c.num_classes;
        }
        if (f.ho_enabled && !dry) ia.epoch = reinterpret_cast<unsigned*>(f.ho_base);
        if (f.ho_enabled && !dry && u->step_trace) ia.trace = u->step_trace + (size_t)u->chain_id * 8192;
        if (f.ok() && !dry) {
            if (c.with_time_emb && !(u->path_mask & 1) && ingest_time_fusable(ia, ta)) {
                f.run(launch_ingest_time(ia, ta, s), "ingest + time");   // the two independent openers of the network in one launch
            } else {
                if (c.with_time_emb) f.run(launch_time_mlp(ta, s), "time");
                f.run(launch_ingest(ia, s), "ingest");
            }
        }
    }

    std::vector<Act> skips;
    for (int i = 0; i < nlev; ++i) {
        // the first block's tail is applied by the second block's first convolution (which also stores it)
        LazyTail lt;
        const bool lazy = !(u->path_mask & 4) && u->downs[i].rb[1].c1.kind == DMN_K3X3 && !u->downs[i].rb[1].has_res &&
                          !block_tail_is_local(f, u->downs[i].rb[0], cur.H, cur.W);
        Act a1 = resblock(f, u->downs[i].rb[0], cur, nullptr, tb, tbs, 0, nullptr, lazy ? &lt : nullptr);
        Act a2 = resblock(f, u->downs[i].rb[1], a1, nullptr, tb, tbs, 1, &lt, nullptr);
        f.free_act(cur);   // the block input stays alive as the deferred tail's residual
        f.free_act(a1);
        // Opt-in (DMN_UNET_PATHS bit 4): below the last level the attention's tail (GroupNorm(y) + a2) is left to the
        // downsampling convolution, and the tensor itself is only stored where a skip connection reads it. Same bits,
        // 6 launches fewer per step, but measured SLOWER on MI355X (64x64, 2 x 32 images: 883 us per step against 863):
        // the resampling kernels stage every input pixel 1.3x (halo) to 4x (sub-pixel phases) and now pay the tail's
        // arithmetic and a second operand read each time, which costs more than the streaming pass it replaces.
        const bool fold = (u->path_mask & 16) && !((u->path_mask & 64) && i > 0) && i < nlev - 1;
        LazyAttn la;
        Act a3 = lin_attn(f, u->downs[i].attn, a2, fold ? &la : nullptr);
        if (fold) {
            Act keep;
            if (i > 0) keep = f.new_act(a2.C, a2.H, a2.W);
            NormIn tn = norm_in(la.y, la.og, la.ob, f.B, 0, nullptr, 0);
            cur = do_conv(f, u->downs[i].resample, la.y, nullptr, &tn, nullptr, 0, nullptr, nullptr, a2.p, i > 0 ? keep.p : nullptr);
            f.free_act(la.y);
            f.free_act(a2);
            if (i > 0) skips.push_back(keep);
            continue;
        }
        f.free_act(a2);
        if (i < nlev - 1) {
            cur = do_conv(f, u->downs[i].resample, a3, nullptr, nullptr, nullptr, 0);
            if (i == 0) f.free_act(a3);   // the top-level skip is never consumed (unet.py:95: ups uses in_out[1:])
            else skips.push_back(a3);
        } else {
            skips.push_back(a3);
            cur = a3;                     // Identity downsample: the bottleneck reads the skip tensor itself
        }
    }
    {
        Act m1 = resblock(f, u->mid1, cur, nullptr, tb, tbs, 1);
        Act m2 = full_attn(f, u->mid_attn, m1);
        f.free_act(m1);
        cur = resblock(f, u->mid2, m2, nullptr, tb, tbs, 0);
        f.free_act(m2);
    }
    for (int i = 0; i < nlev - 1; ++i) {
        Act sk = skips.back();
        skips.pop_back();
        LazyTail lt;
        const bool lazy = !(u->path_mask & 4) && u->ups[i].rb[1].c1.kind == DMN_K3X3 && !u->ups[i].rb[1].has_res &&
                          !block_tail_is_local(f, u->ups[i].rb[0], cur.H, cur.W);
        Act a1 = resblock(f, u->ups[i].rb[0], cur, &sk, tb, tbs, 0, nullptr, lazy ? &lt : nullptr);
        f.free_act(cur);
        f.free_act(sk);
        Act a2 = resblock(f, u->ups[i].rb[1], a1, nullptr, tb, tbs, 1, &lt, nullptr);
        f.free_act(a1);
        if ((u->path_mask & 16) && !(u->path_mask & 32)) {   // opt-in: the upsampling convolution applies the attention's tail
            LazyAttn la;
            lin_attn(f, u->ups[i].attn, a2, &la);
            NormIn tn = norm_in(la.y, la.og, la.ob, f.B, 0, nullptr, 0);
            cur = do_conv(f, u->ups[i].resample, la.y, nullptr, &tn, nullptr, 0, nullptr, nullptr, a2.p, nullptr);
            f.free_act(la.y);
            f.free_act(a2);
            continue;
        }
        Act a3 = lin_attn(f, u->ups[i].attn, a2);
        f.free_act(a2);
        cur = do_conv(f, u->ups[i].resample, a3, nullptr, nullptr, nullptr, 0);
        f.free_act(a3);
    }
    {
        Act fin = resblock(f, u->final_rb, cur, nullptr, nullptr, 0, c.tail_norm ? c.groups : 0);
        f.free_act(cur);
        EgressArgs ea;
        memset(&ea, 0, sizeof(ea));
        ea.dtype = c.dtype;
        ea.in = fin.p;
        if (c.tail_norm) ea.norm = norm_in(fin, u->tail_g, u->tail_b, B, 1, nullptr, 0);
        ea.w = u->tail_w;
        ea.bias = u->tail_bias;
        ea.out = out;
        ea.B = B; ea.HW = H * W; ea.C = c.dim; ea.Cout = c.out_dim;
        if (fuse_step) {   // the sampler's update rides on the network's last kernel: one launch fewer per step
            ea.fuse_step = 1;
            ea.step = *fuse_step;
        }
        if (f.ok() && !dry) f.run(launch_egress(ea, s), fuse_step ? "egress + sampler update" : "egress");
        f.free_act(fin);
    }
    if (high) *high = f.ar.high;
    if (handoff_need) *handoff_need = f.ho_off;
    return f.rc == kTraceStop ? DMN_OK : f.rc;
}

static int pack_fragments(int dtype, const float* W, int rows, int K, bool acc_order, void** dev);

// Fused linear attention on caller-provided tensors (unit parity tests, tuning). Packs + uploads the weights per call.
// closing: 0 = stop at y = to_out(...) + bias with its statistics; 1 = out = GroupNorm(1)(y) + x computed by the second
// launch itself (in-launch statistics hand-off); 2 = the same tensor from a separate closing pass over y.
static int test_linattn_impl(int dtype, const void* x, const float* x_stats, int B, int n, int C, const float* gamma_host,
                             const float* beta_host, const float* wqkv_host, const float* wout_host, const float* bout_host,
                             void* y, float* y_stats, int* slices_out, void* scratch, size_t scratch_bytes, int closing,
                             const float* og_host, const float* ob_host, void* out, unsigned* giveups, int plan_B, hipStream_t stream,
                             int separate_ctx = 0) {
    LinAttnFusedArgs fa;
    memset(&fa, 0, sizeof(fa));
    fa.separate_ctx = separate_ctx;
    void *fq = nullptr, *fk = nullptr, *fv = nullptr, *fo = nullptr, *ctxT = nullptr, *area = nullptr;
    float *g = nullptr, *bt = nullptr, *bo = nullptr, *og = nullptr, *ob = nullptr;
    auto cleanup = [&]() {
        (void)hipStreamSynchronize(stream);
        void* fr[] = {fq, fk, fv, fo, g, bt, bo, og, ob, ctxT, area};
        for (void* q : fr)
            if (q) (void)hipFree(q);
    };
    int rc;
    if ((rc = pack_fragments(dtype, wqkv_host, 128, C, false, &fq)) || (rc = pack_fragments(dtype, wqkv_host + (size_t)128 * C, 128, C, false, &fk)) ||
        (rc = pack_fragments(dtype, wqkv_host + (size_t)256 * C, 128, C, false, &fv)) || (rc = pack_fragments(dtype, wout_host, C, 128, false, &fo)) ||
        (rc = upload_f32(gamma_host, C, &g)) || (rc = upload_f32(beta_host, C, &bt)) || (rc = upload_f32(bout_host, C, &bo)) ||
        (closing && ((rc = upload_f32(og_host, C, &og)) || (rc = upload_f32(ob_host, C, &ob))))) {
        cleanup();
        return rc;
    }
    fa.dtype = dtype;
    fa.x = x;
    fa.y = y;
    fa.norm.stats = x_stats;
    fa.norm.G = 1;
    fa.norm.slots = 1;
    fa.norm.inv_count = 1.0f / ((float)C * n);
    fa.norm.gamma = g;
    fa.norm.beta = bt;
    fa.wq = fq; fa.wk = fk; fa.wv = fv; fa.wo = fo;
    fa.bias = bo;
    fa.B = B; fa.n = n; fa.C = C;
    linattn_fused_plan(n, plan_B > 0 ? plan_B : B, C, dtype, &fa.slice1, &fa.splits, &fa.tpw, &fa.slices);
    const size_t need = (size_t)B * 4 * fa.splits * (32 * 32 + 64) * 4;
    if (scratch_bytes < need || hipMalloc(&ctxT, (size_t)B * 4 * 32 * 32 * 4) != hipSuccess) {
        cleanup();
        dmn_set_error("dmn_test_linattn: scratch %zu < %zu, or allocation failed", scratch_bytes, need);
        return DMN_ERR_WORKSPACE;
    }
    fa.ctx_part = (float*)scratch;
    fa.y_stats = y_stats;
    if (slices_out) *slices_out = 4 * fa.slices;   // statistic slots per image: one per wave of every pixel slice
    fa.ctxT = ctxT;
    if (giveups) *giveups = 0;
    if (closing == 1) {
        if (!linattn_tail_fits(n, plan_B > 0 ? plan_B : B, C, dtype)) {
            cleanup();
            dmn_set_error("dmn_test_linattn_block: shape outside the in-launch exchange's limits");
            return DMN_ERR_LIMIT;
        }
        const size_t bytes = kHandoffHeaderBytes + (size_t)B * 4 * fa.slices * kHandoffGranulesPerSlot * sizeof(unsigned long long);
        unsigned hdr[2] = {1u, 0u};   // epoch 1 over a zeroed area
        if (hipMalloc(&area, bytes) != hipSuccess || hipMemset(area, 0, bytes) != hipSuccess ||
            hipMemcpy(area, hdr, sizeof(hdr), hipMemcpyHostToDevice) != hipSuccess) {
            cleanup();
            return DMN_ERR_HIP;
        }
        fa.tail.gran = reinterpret_cast<unsigned long long*>((char*)area + kHandoffHeaderBytes);
        fa.tail.epoch = reinterpret_cast<const unsigned*>(area);
        fa.tail.giveups = reinterpret_cast<unsigned*>(area) + 1;
        fa.tail_gamma = og;
        fa.tail_beta = ob;
        fa.tail_inv_count = 1.0f / ((float)C * n);
        fa.out = out;
    }
    rc = launch_linattn_fused(fa, stream);
    if (rc == DMN_OK && closing == 2) {
        NormApplyArgs na;
        memset(&na, 0, sizeof(na));
        na.dtype = dtype;
        na.in = y;
        na.out = out;
        na.res = x;
        na.B = B;
        na.HW = n;
        na.C = C;
        na.norm.stats = y_stats;
        na.norm.G = 1;
        na.norm.slots = 4 * fa.slices;
        na.norm.inv_count = 1.0f / ((float)C * n);
        na.norm.gamma = og;
        na.norm.beta = ob;
        rc = launch_norm_apply(na, stream);
    }
    (void)hipStreamSynchronize(stream);
    if (giveups && area) (void)hipMemcpy(giveups, (char*)area + 4, 4, hipMemcpyDeviceToHost);
    cleanup();
    return rc;
}

extern "C" int dmn_test_linattn(int dtype, const void* x, const float* x_stats, int B, int n, int C, const float* gamma_host,
                                const float* beta_host, const float* wqkv_host, const float* wout_host,
                                const float* bout_host, void* y, float* y_stats, int* slices_out, void* scratch,
                                size_t scratch_bytes, void* stream) {
    DMN_REQUIRE(x && x_stats && y && y_stats && scratch, DMN_ERR_INVALID, "dmn_test_linattn: null argument");
    DMN_REQUIRE(linattn_fused_supported(C, 4, 32), DMN_ERR_INVALID, "dmn_test_linattn: C=%d unsupported", C);
    return test_linattn_impl(dtype, x, x_stats, B, n, C, gamma_host, beta_host, wqkv_host, wout_host, bout_host, y, y_stats,
                             slices_out, scratch, scratch_bytes, 0, nullptr, nullptr, nullptr, nullptr, 0, (hipStream_t)stream);
}

// The whole Residual(PreNorm(LinearAttention)) block (utils.py:68-93, parts/mha.py:44-59): out = GroupNorm(1)(y) + x behind
// the fused attention, fused = 1 inside its second launch, fused = 0 as a separate closing pass over y (same bits);
// + 2: the first pass always as a launch of its own, also where one workgroup holds an image and one launch would do.
extern "C" int dmn_test_linattn_block(int dtype, const void* x, const float* x_stats, int B, int n, int C, const float* gamma_host,
                                      const float* beta_host, const float* wqkv_host, const float* wout_host,
                                      const float* bout_host, const float* out_gamma_host, const float* out_beta_host, void* y,
                                      float* y_stats, void* out, void* scratch, size_t scratch_bytes, int fused, int plan_B,
                                      unsigned* giveups, void* stream) {
    DMN_REQUIRE(x && x_stats && y && y_stats && out && scratch && out_gamma_host && out_beta_host, DMN_ERR_INVALID,
                "dmn_test_linattn_block: null argument");
    DMN_REQUIRE(linattn_fused_supported(C, 4, 32), DMN_ERR_INVALID, "dmn_test_linattn_block: C=%d unsupported", C);
    return test_linattn_impl(dtype, x, x_stats, B, n, C, gamma_host, beta_host, wqkv_host, wout_host, bout_host, y, y_stats, nullptr,
                             scratch, scratch_bytes, (fused & 1) ? 1 : 2, out_gamma_host, out_beta_host, out, giveups, plan_B,
                             (hipStream_t)stream, (fused & 2) != 0);
}

// Times the kernel that dominates a denoising step -- the level-0 3x3 convolution with the fused GroupNorm +
// SiLU + time-bias prologue and the statistics epilogue (block2.proj of downs.0.0) -- by itself: `iters`
// back-to-back launches bracketed by HIP events on `stream`.
extern "C" int dmn_bench_dominant_kernel(DmnUnet* u, int B, int H, int W, int iters, void* ws, size_t ws_bytes, void* stream,
                                         float* avg_ms, double* alg_bytes) {
    DMN_REQUIRE(u && u->loaded && ws && avg_ms && iters > 0 && B > 0, DMN_ERR_INVALID, "dmn_bench_dominant_kernel: bad argument");
    hipStream_t s = (hipStream_t)stream;
    const ResBlockW& w = u->downs[0].rb[0];
    const int G = u->cfg.groups, C = w.c2.Cin;
    Fwd f;
    f.u = u;
    f.s = s;
    f.B = B;
    f.dry = false;
    f.ar.reset(ws, ws_bytes, false);
    Act x = f.new_act(C, H, W);
    x.sG = G;
    x.sSlots = 1;
    x.stats = f.new_stats(G, 1);
    float* tb = (float*)f.alloc((size_t)u->tb_total * 4 + 256);
    if (!f.ok()) return f.rc;
    // zero activations with statistics {sum 0, sum of squares = count}: mean 0, variance 1
    DMN_HIP_CHECK(hipMemsetAsync(x.p, 0, (size_t)B * H * W * C * f.esz(), s));
    DMN_HIP_CHECK(hipMemsetAsync(tb, 0, (size_t)u->tb_total * 4, s));
    std::vector<float> st((size_t)B * G * 4);
    for (size_t i = 0; i < st.size(); i += 4) {   // one partial per group: sum 0, squared deviations = count
        st[i] = 0.f;
        st[i + 1] = (float)(C / G) * H * W;
        st[i + 2] = (float)(C / G) * H * W;
        st[i + 3] = 0.f;
    }
    DMN_HIP_CHECK(hipMemcpyAsync(x.stats, st.data(), st.size() * 4, hipMemcpyHostToDevice, s));
    DMN_HIP_CHECK(hipStreamSynchronize(s));
    NormIn n1 = norm_in(x, w.g1, w.b1, B, 1, w.tb_off >= 0 ? tb + w.tb_off : nullptr, 0);
    hipEvent_t e0, e1;
    DMN_HIP_CHECK(hipEventCreate(&e0));
    DMN_HIP_CHECK(hipEventCreate(&e1));
    int rc = DMN_OK;
    for (int i = 0; i < 3 + iters && rc == DMN_OK; ++i) {
        if (i == 3) (void)hipEventRecord(e0, s);
        Act h = do_conv(f, w.c2, x, nullptr, &n1, nullptr, G);
        rc = f.rc;
        f.free_act(h);
    }
    (void)hipEventRecord(e1, s);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *avg_ms = ms / iters;
    if (alg_bytes)
        *alg_bytes = (double)B * H * W * (w.c2.Cin + w.c2.Cout) * f.esz() + (double)w.c2.Cin * w.c2.Cout * 9 * f.esz();
    return rc;
}

// Event pairs around the dominant convolution as it runs INSIDE eager (not graph-replayed) network evaluations: the
// launch then follows and precedes the same kernels as in a sampling step and finds its operands in the same cache
// state, unlike back-to-back launches of the kernel by itself. Enable, run evaluations / an eager sampling loop on one
// stream, then read the averages. The host enqueues faster than the device executes, so the markers sit in the queue
// directly around the kernel. Three markers per probed launch: the distance of the first two (nothing between them) is
// what a marker packet itself costs, the distance of the last two is the kernel plus one marker.
extern "C" int dmn_unet_probe_dominant(DmnUnet* u, int enable) {
    DMN_REQUIRE(u && enable >= 0 && enable <= 2, DMN_ERR_INVALID, "dmn_unet_probe_dominant: handle / mode");
    if (enable == 1 && u->probe_ev.empty()) {
        u->probe_ev.resize(3 * kProbePairs, nullptr);
        for (auto& e : u->probe_ev) DMN_HIP_CHECK(hipEventCreate(&e));
    }
    if (enable == 2) {
        // Device-clock form: the workgroups of the probed launch stamp the constant 100 MHz clock when they start and
        // when they end; a one-workgroup kernel behind the launch adds (latest end - earliest start) to a running sum.
        // Works in captured steps too, so it times the launch where the timed chains run it.
        if (!u->probe_stamps) {
            DMN_HIP_CHECK(hipMalloc(&u->probe_stamps, (size_t)2 * kProbeMaxWgs * 8 * sizeof(long long)));
            DMN_HIP_CHECK(hipMalloc(&u->probe_acc, 4 * sizeof(unsigned long long)));
        }
        DMN_HIP_CHECK(hipMemset(u->probe_stamps, 0, (size_t)2 * kProbeMaxWgs * 8 * sizeof(long long)));
        DMN_HIP_CHECK(hipMemset(u->probe_acc, 0, 4 * sizeof(unsigned long long)));
    }
    if ((u->probe_on == 2) != (enable == 2) && u->graph_exec) {   // captured steps hold (or lack) the stamping launch
        DMN_HIP_CHECK(hipDeviceSynchronize());
        (void)hipGraphExecDestroy(u->graph_exec);
        if (u->graph_exec2) (void)hipGraphExecDestroy(u->graph_exec2);
        u->graph_exec = u->graph_exec2 = nullptr;
    }
    u->probe_on = enable;
    if (enable == 1) u->probe_n = 0;
    return DMN_OK;
}

extern "C" int dmn_unet_probe_result(DmnUnet* u, float* avg_ms, float* marker_ms, int* launches, double* alg_bytes) {
    DMN_REQUIRE(u && avg_ms && marker_ms && launches, DMN_ERR_INVALID, "dmn_unet_probe_result: null argument");
    if (u->probe_acc && u->probe_on != 1 && u->probe_n == 0) {   // device-clock form
        unsigned long long acc[4];
        DMN_HIP_CHECK(hipDeviceSynchronize());
        DMN_HIP_CHECK(hipMemcpy(acc, u->probe_acc, sizeof(acc), hipMemcpyDeviceToHost));
        DMN_REQUIRE(acc[1] + acc[3] > 0, DMN_ERR_INVALID, "dmn_unet_probe_result: no probed launch recorded");
        *avg_ms = (float)((double)(acc[0] + acc[2]) / (double)(acc[1] + acc[3]) * 1e-5);   // ticks of 10 ns, both chains
        *marker_ms = 0.f;
        *launches = (int)(acc[1] + acc[3]);
        if (alg_bytes) *alg_bytes = u->probe_bytes;
        return DMN_OK;
    }
    DMN_REQUIRE(u->probe_n > 0, DMN_ERR_INVALID, "dmn_unet_probe_result: no probed launch recorded");
    double sum = 0.0, empty = 0.0;
    for (int i = 0; i < u->probe_n; ++i) {
        DMN_HIP_CHECK(hipEventSynchronize(u->probe_ev[3 * i + 2]));
        float ms = 0.f;
        DMN_HIP_CHECK(hipEventElapsedTime(&ms, u->probe_ev[3 * i], u->probe_ev[3 * i + 1]));
        empty += ms;
        DMN_HIP_CHECK(hipEventElapsedTime(&ms, u->probe_ev[3 * i + 1], u->probe_ev[3 * i + 2]));
        sum += ms;
    }
    *avg_ms = (float)(sum / u->probe_n);
    *marker_ms = (float)(empty / u->probe_n);
    *launches = u->probe_n;
    u->probe_n = 0;
    if (alg_bytes) *alg_bytes = u->probe_bytes;
    return DMN_OK;
}

// ---- weights ---------------------------------------------------------------------------------------------
struct ParamMap {
    std::map<std::string, const DmnTensorDesc*> m;
    const DmnTensorDesc* get(const std::string& n, int ndim, const int64_t* shape) const {
        auto it = m.find(n);
        if (it == m.end()) {
            dmn_set_error("missing parameter '%s'", n.c_str());
            return nullptr;
        }
        const DmnTensorDesc* d = it->second;
        bool ok = d->ndim == ndim;
        for (int i = 0; ok && i < ndim; ++i) ok = d->shape[i] == shape[i];
        if (!ok) {
            dmn_set_error("parameter '%s' has the wrong shape", n.c_str());
            return nullptr;
        }
        return d;
    }
};

#define GETP(var, name, ...)                                  \
    const DmnTensorDesc* var;                                 \
    {                                                         \
        const int64_t _s[] = {__VA_ARGS__};                   \
        var = pm.get(name, (int)(sizeof(_s) / 8), _s);        \
        if (!var) return DMN_ERR_WEIGHTS;                     \
    }

static int load_vec(const ParamMap& pm, const std::string& name, int n, float** dev) {
    GETP(d, name, n);
    return upload_f32(d->data, n, dev);
}

static int load_conv(const ParamMap& pm, int dtype, const std::string& name, int Cout, int Cin, int k, int stride, int pad,
                     bool bias, bool transposed, PackedConv& pc) {
    const DmnTensorDesc* w;
    {
        const int64_t s0[] = {Cout, Cin, k, k}, s1[] = {Cin, Cout, k, k};
        w = pm.get(name + ".weight", 4, transposed ? s1 : s0);
        if (!w) return DMN_ERR_WEIGHTS;
    }
    const float* b = nullptr;
    if (bias) {
        GETP(bd, name + ".bias", Cout);
        b = bd->data;
    }
    return pack_conv(pc, dtype, w->data, b, Cout, Cin, k, k, stride, pad, transposed);
}

static int load_resblock(const ParamMap& pm, DmnUnet* u, const std::string& pre, int cin, int cout, bool time,
                         std::vector<float>& twb, std::vector<float>& tbb, ResBlockW& rb) {
    const int dtype = u->cfg.dtype;
    int rc;
    rb.tb_off = -1;
    if (time && u->cfg.with_time_emb) {
        const int td = 4 * u->cfg.dim;
        GETP(w, pre + ".mlp.1.weight", cout, td);
        GETP(b, pre + ".mlp.1.bias", cout);
        rb.tb_off = (int)tbb.size();
        twb.insert(twb.end(), w->data, w->data + (size_t)cout * td);
        tbb.insert(tbb.end(), b->data, b->data + cout);
    }
    if ((rc = load_conv(pm, dtype, pre + ".block1.proj", cout, cin, 3, 1, 1, true, false, rb.c1))) return rc;
    if ((rc = load_vec(pm, pre + ".block1.norm.weight", cout, &rb.g1))) return rc;
    if ((rc = load_vec(pm, pre + ".block1.norm.bias", cout, &rb.b1))) return rc;
    if ((rc = load_conv(pm, dtype, pre + ".block2.proj", cout, cout, 3, 1, 1, true, false, rb.c2))) return rc;
    if ((rc = load_vec(pm, pre + ".block2.norm.weight", cout, &rb.g2))) return rc;
    if ((rc = load_vec(pm, pre + ".block2.norm.bias", cout, &rb.b2))) return rc;
    rb.has_res = cin != cout;
    if (rb.has_res && (rc = load_conv(pm, dtype, pre + ".res_conv", cout, cin, 1, 1, 0, true, false, rb.res))) return rc;
    return DMN_OK;
}

// W: [rows][K] row-major. Output [rows/16][K/kch][64 lanes][per] where lane (q = lane>>4, r = lane&15) holds row
// 16*tile + r and the k values kstep*kch + q*per + j; `acc_order` stores each 32-wide k block in the order an
// accumulator-built operand presents it (bf16: pos -> 4*(pos>>3) + (pos&3) + 16*((pos>>2)&1); f32: identity).
static int pack_fragments(int dtype, const float* W, int rows, int K, bool acc_order, void** dev) {
    const int per = dtype == DMN_F32 ? 4 : 8, kch = dtype == DMN_F32 ? 16 : 32;
    DMN_REQUIRE(rows % 16 == 0 && K % 32 == 0, DMN_ERR_WEIGHTS, "fragment packing: %dx%d", rows, K);
    std::vector<float> p((size_t)rows * K);
    size_t o = 0;
    for (int rt = 0; rt < rows / 16; ++rt)
        for (int ks = 0; ks < K / kch; ++ks)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < per; ++j) {
                    int k = ks * kch + (lane >> 4) * per + j;
                    if (acc_order && dtype != DMN_F32) {
                        const int pos = k & 31;
                        k = (k & ~31) + 4 * (pos >> 3) + (pos & 3) + 16 * ((pos >> 2) & 1);
                    }
                    p[o++] = W[(size_t)(rt * 16 + (lane & 15)) * K + k];
                }
    if (dtype == DMN_F32) return upload(p.data(), p.size() * 4, dev);
    std::vector<uint16_t> h(p.size());
    for (size_t i = 0; i < p.size(); ++i) h[i] = f32_to_bf16_rne(p[i]);
    return upload(h.data(), h.size() * 2, dev);
}

static int load_attn(const ParamMap& pm, DmnUnet* u, const std::string& pre, int c, bool linear, AttnW& a) {
    const int dtype = u->cfg.dtype;
    const int hidden = u->cfg.heads * u->cfg.dim_head;
    int rc;
    if ((rc = load_vec(pm, pre + ".fn.norm.weight", c, &a.ng))) return rc;
    if ((rc = load_vec(pm, pre + ".fn.norm.bias", c, &a.nb))) return rc;
    if ((rc = load_conv(pm, dtype, pre + ".fn.fn.to_qkv", 3 * hidden, c, 1, 1, 0, false, false, a.qkv))) return rc;
    if (linear) {
        if ((rc = load_conv(pm, dtype, pre + ".fn.fn.to_out.0", c, hidden, 1, 1, 0, true, false, a.out))) return rc;
        if ((rc = load_vec(pm, pre + ".fn.fn.to_out.1.weight", c, &a.og))) return rc;
        if ((rc = load_vec(pm, pre + ".fn.fn.to_out.1.bias", c, &a.ob))) return rc;
        if (linattn_fused_supported(c, u->cfg.heads, u->cfg.dim_head)) {
            const int64_t sq[] = {3 * hidden, c, 1, 1}, so[] = {c, hidden, 1, 1};
            const DmnTensorDesc* wq = pm.get(pre + ".fn.fn.to_qkv.weight", 4, sq);
            const DmnTensorDesc* wo = pm.get(pre + ".fn.fn.to_out.0.weight", 4, so);
            if (!wq || !wo) return DMN_ERR_WEIGHTS;
            if ((rc = pack_fragments(dtype, wq->data, hidden, c, false, &a.fq))) return rc;
            if ((rc = pack_fragments(dtype, wq->data + (size_t)hidden * c, hidden, c, false, &a.fk))) return rc;
            if ((rc = pack_fragments(dtype, wq->data + (size_t)2 * hidden * c, hidden, c, false, &a.fv))) return rc;
            if ((rc = pack_fragments(dtype, wo->data, c, hidden, false, &a.fo))) return rc;
        }
    } else {
        if ((rc = load_conv(pm, dtype, pre + ".fn.fn.to_out", c, hidden, 1, 1, 0, true, false, a.out))) return rc;
    }
    return DMN_OK;
}

static void free_rb(ResBlockW& rb) {
    free_conv(rb.c1);
    free_conv(rb.c2);
    free_conv(rb.res);
    float** v[] = {&rb.g1, &rb.b1, &rb.g2, &rb.b2};
    for (auto p : v) {
        if (*p) (void)hipFree(*p);
        *p = nullptr;
    }
}
static void free_attn(AttnW& a) {
    free_conv(a.qkv);
    free_conv(a.out);
    float** v[] = {&a.ng, &a.nb, &a.og, &a.ob};
    for (auto p : v) {
        if (*p) (void)hipFree(*p);
        *p = nullptr;
    }
    void** f[] = {&a.fq, &a.fk, &a.fv, &a.fo};
    for (auto p : f) {
        if (*p) (void)hipFree(*p);
        *p = nullptr;
    }
}
static void free_weights(DmnUnet* u) {
    for (auto& l : u->downs) { free_rb(l.rb[0]); free_rb(l.rb[1]); free_attn(l.attn); free_conv(l.resample); }
    for (auto& l : u->ups) { free_rb(l.rb[0]); free_rb(l.rb[1]); free_attn(l.attn); free_conv(l.resample); }
    free_rb(u->mid1);
    free_rb(u->mid2);
    free_rb(u->final_rb);
    free_attn(u->mid_attn);
    float** v[] = {&u->freqs, &u->tw1, &u->tb1, &u->tw2, &u->tb2, &u->twb, &u->tbb, &u->init_w, &u->init_b,
                   &u->tail_g, &u->tail_b, &u->tail_w, &u->tail_bias};
    for (auto p : v) {
        if (*p) (void)hipFree(*p);
        *p = nullptr;
    }
    if (u->init_wfrag) (void)hipFree(u->init_wfrag);
    u->init_wfrag = nullptr;
    if (u->class_embed) (void)hipFree(u->class_embed);
    u->class_embed = nullptr;
    u->loaded = false;
}

extern "C" int dmn_unet_create(const DmnUnetCfg* cfg, DmnUnet** out) {
    DMN_REQUIRE(cfg && out, DMN_ERR_INVALID, "dmn_unet_create: null argument");
    DMN_REQUIRE(cfg->n_mults >= 1 && cfg->n_mults <= DMN_MAX_MULTS, DMN_ERR_INVALID, "n_mults=%d", cfg->n_mults);
    DMN_REQUIRE(cfg->dtype == DMN_F32 || cfg->dtype == DMN_BF16, DMN_ERR_INVALID, "dtype=%d", cfg->dtype);
    DMN_REQUIRE(cfg->num_classes >= 0, DMN_ERR_INVALID, "num_classes=%d", cfg->num_classes);
    DMN_REQUIRE(cfg->dim % 32 == 0 && cfg->dim <= 64, DMN_ERR_LIMIT, "dim=%d must be 32 or 64", cfg->dim);
    DMN_REQUIRE(cfg->dim_head == 32 && cfg->heads >= 1, DMN_ERR_INVALID, "dim_head=%d (only 32 supported)", cfg->dim_head);
    DMN_REQUIRE(cfg->channels >= 1 && cfg->channels <= 4 && cfg->out_dim >= 1 && cfg->out_dim <= 8, DMN_ERR_LIMIT,
                "channels=%d out_dim=%d", cfg->channels, cfg->out_dim);
    DMN_REQUIRE(cfg->groups >= 1 && cfg->dim % cfg->groups == 0 && (cfg->dim / cfg->groups) % 4 == 0, DMN_ERR_INVALID,
                "groups=%d needs >= 4 channels per group", cfg->groups);
    for (int i = 0; i < cfg->n_mults; ++i)
        DMN_REQUIRE(cfg->dim_mults[i] >= 1 && (i == 0 || true), DMN_ERR_INVALID, "dim_mults[%d]=%d", i, cfg->dim_mults[i]);
    int rc = conv_init_attributes();
    if (rc) return rc;
    DmnUnet* u = new DmnUnet();
    u->cfg = *cfg;
    if (const char* e = getenv("DMN_UNET_PATHS")) u->path_mask = atoi(e);
    if (const char* e = getenv("DMN_SAMPLE_CHAINS")) u->sample_chains = atoi(e) == 1 ? 1 : 2;
    if (const char* e = getenv("DMN_PROBE_SITE")) u->probe_site = atoi(e);
    if (const char* e = getenv("DMN_HANDOFF_EPOCH_LIMIT")) {
        const unsigned long long v = strtoull(e, nullptr, 0);
        if (v >= 1 && v <= 0xFFF00000ull) u->epoch_limit = v;
    }
    if (const char* e = getenv("DMN_LOOP_PACE")) u->loop_pace = atoi(e) < 0 ? 0 : (atoi(e) > 4096 ? 4096 : atoi(e));
    if (const char* e = getenv("DMN_LOOP_GRAPH_STEPS")) u->loop_graph_steps = atoi(e) < 1 ? 1 : (atoi(e) > 64 ? 64 : atoi(e));
    if (const char* e = getenv("DMN_LOOP_LOCKSTEP")) u->loop_lockstep = atoi(e) != 0;
    if (const char* e = getenv("DMN_LOOP_LOCKSTEP_EVERY")) u->loop_lockstep_every = atoi(e) < 1 ? 1 : (atoi(e) > 1024 ? 1024 : atoi(e));
    if (const char* e = getenv("DMN_LOOP_DEPTH")) u->loop_depth = atoi(e) < 2 ? 2 : (atoi(e) > DmnUnet::kPaceSlots ? DmnUnet::kPaceSlots : atoi(e));
    if (const char* e = getenv("DMN_SAMPLE_SKEW_US")) u->sample_skew_us = atoi(e) < 0 ? 0 : (atoi(e) > 5000 ? 5000 : atoi(e));
    u->downs.resize(cfg->n_mults);
    u->ups.resize(cfg->n_mults - 1);
    *out = u;
    return DMN_OK;
}

// Diagnosis: attach a device array of `max_slots` 64-bit slots; every forward from then on writes a checksum of its
// whole workspace after each of its launches (slot k: after launch k) and restarts at slot 0. nullptr detaches.
extern "C" int dmn_test_unet_trace(DmnUnet* u, unsigned long long* slots, int max_slots) {
    DMN_REQUIRE(u && max_slots >= 0, DMN_ERR_INVALID, "dmn_test_unet_trace: bad argument");
    u->trace_slots = slots;
    u->trace_max = slots ? max_slots : 0;
    u->trace_n = 0;
    u->trace_stop = -1;
    return DMN_OK;
}

// Diagnosis: traced forwards overwrite the LDS of every CU with `pattern` in front of each launch (0: off).
extern "C" int dmn_test_unet_trace_poison(DmnUnet* u, unsigned pattern) {
    DMN_REQUIRE(u, DMN_ERR_INVALID, "dmn_test_unet_trace_poison: null handle");
    u->trace_poison = pattern;
    return DMN_OK;
}

// Diagnosis: the following forwards end after launch k (the workspace then holds the state behind it); -1: run all.
extern "C" int dmn_test_unet_trace_stop(DmnUnet* u, int k) {
    DMN_REQUIRE(u, DMN_ERR_INVALID, "dmn_test_unet_trace_stop: null handle");
    u->trace_stop = k;
    return DMN_OK;
}

// Launches traced by the last forward, and the description of launch k (valid until the next forward).
extern "C" int dmn_test_unet_trace_count(const DmnUnet* u) { return u ? u->trace_n : 0; }
extern "C" const char* dmn_test_unet_trace_name(const DmnUnet* u, int k) {
    return (u && k >= 0 && k < (int)u->trace_names.size()) ? u->trace_names[k].c_str() : "";
}

extern "C" int dmn_unet_set_class_labels(DmnUnet* u, const int64_t* labels) {
    DMN_REQUIRE(u, DMN_ERR_INVALID, "dmn_unet_set_class_labels: null handle");
    DMN_REQUIRE(u->cfg.num_classes > 0 || labels == nullptr, DMN_ERR_INVALID, "the network has no class embedding");
    u->labels = reinterpret_cast<const long long*>(labels);
    return DMN_OK;
}

extern "C" int dmn_unet_set_plan_batch(DmnUnet* u, int batch) {
    DMN_REQUIRE(u && batch >= 0, DMN_ERR_INVALID, "dmn_unet_set_plan_batch: bad argument");
    if (batch != u->plan_batch && u->graph_exec) {   // a captured step holds the old launch geometry
        (void)hipGraphExecDestroy(u->graph_exec);
        if (u->graph_exec2) (void)hipGraphExecDestroy(u->graph_exec2);
        u->graph_exec = u->graph_exec2 = nullptr;
    }
    u->plan_batch = batch;
    return DMN_OK;
}

extern "C" void dmn_unet_destroy(DmnUnet* u) {
    if (!u) return;
    exchange_guard_forget(u);
    if (u->graph_exec) (void)hipGraphExecDestroy(u->graph_exec);
    if (u->graph_exec2) (void)hipGraphExecDestroy(u->graph_exec2);
    for (hipStream_t cs : u->chain_stream)
        if (cs) (void)hipStreamDestroy(cs);
    if (u->ev_fork) (void)hipEventDestroy(u->ev_fork);
    if (u->ev_join) (void)hipEventDestroy(u->ev_join);
    if (u->ev_join0) (void)hipEventDestroy(u->ev_join0);
    for (auto& slot : u->ev_pace)
        for (hipEvent_t e : slot)
            if (e) (void)hipEventDestroy(e);
    if (u->ev_start) (void)hipEventDestroy(u->ev_start);
    if (u->ev_stop) (void)hipEventDestroy(u->ev_stop);
    for (auto e : u->probe_ev)
        if (e) (void)hipEventDestroy(e);
    if (u->step_trace) (void)hipFree(u->step_trace);
    if (u->h2_table) (void)hipFree(u->h2_table);
    if (u->probe_stamps) (void)hipFree(u->probe_stamps);
    if (u->probe_acc) (void)hipFree(u->probe_acc);
    for (int c = 0; c < 2; ++c)
        if (u->handoff_mem[c]) (void)hipFree(u->handoff_mem[c]);
    free_weights(u);
    delete u;
}

extern "C" int dmn_unet_load_weights(DmnUnet* u, const DmnTensorDesc* params, int n) {
    DMN_REQUIRE(u && params && n > 0, DMN_ERR_INVALID, "dmn_unet_load_weights: null argument");
    DMN_HIP_CHECK(hipDeviceSynchronize());
    free_weights(u);
    if (u->graph_exec) {
        (void)hipGraphExecDestroy(u->graph_exec);
        if (u->graph_exec2) (void)hipGraphExecDestroy(u->graph_exec2);
        u->graph_exec = u->graph_exec2 = nullptr;
    }
    ParamMap pm;
    for (int i = 0; i < n; ++i) pm.m[params[i].name] = &params[i];
    const DmnUnetCfg& c = u->cfg;
    const int dim = c.dim, td = 4 * dim;
    int rc;
    std::vector<int> dims(1, dim);
    for (int i = 0; i < c.n_mults; ++i) dims.push_back(dim * c.dim_mults[i]);
    std::vector<float> twb, tbb;

    {   // ingest weights stay fp32, transposed to [Cin*49][Cout] so a tap's Cout weights are contiguous
        GETP(w, "init_conv.weight", dim, c.channels, 7, 7);
        GETP(b, "init_conv.bias", dim);
        const int taps = c.channels * 49;
        std::vector<float> wt((size_t)dim * taps);
        for (int o = 0; o < dim; ++o)
            for (int k = 0; k < taps; ++k) wt[(size_t)k * dim + o] = w->data[(size_t)o * taps + k];
        if ((rc = upload_f32(wt.data(), wt.size(), &u->init_w))) return rc;
        if ((rc = upload_f32(b->data, dim, &u->init_b))) return rc;
        if (c.dtype == DMN_BF16) {
            // MFMA form: per kernel row ky one k-step of 8 pixels x 4 channels (k = 4*dx + ci, zero for dx = 7 and
            // ci >= channels), A-operand fragments [Cout/16][7][64 lanes][8]
            std::vector<uint16_t> fr((size_t)(dim / 16) * 7 * 64 * 8);
            size_t o = 0;
            for (int ct = 0; ct < dim / 16; ++ct)
                for (int ky = 0; ky < 7; ++ky)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int j = 0; j < 8; ++j) {
                            const int k = (lane >> 4) * 8 + j, dx = k >> 2, ci = k & 3;
                            const int co = ct * 16 + (lane & 15);
                            const float v = (dx < 7 && ci < c.channels) ? w->data[(((size_t)co * c.channels + ci) * 7 + ky) * 7 + dx] : 0.f;
                            fr[o++] = f32_to_bf16_rne(v);
                        }
            if ((rc = upload(fr.data(), fr.size() * 2, &u->init_wfrag))) return rc;
        }
    }
    if (c.num_classes > 0) {
        GETP(ce, "class_embed.weight", c.num_classes + 1, dim);
        if ((rc = upload_f32(ce->data, (size_t)(c.num_classes + 1) * dim, &u->class_embed))) return rc;
    }
    if (c.with_time_emb) {
        GETP(w1, "time_mlp.1.weight", td, dim);
        GETP(b1, "time_mlp.1.bias", td);
        GETP(w2, "time_mlp.3.weight", td, td);
        GETP(b2, "time_mlp.3.bias", td);
        // dense layers are stored transposed ([in][out]) for the one-thread-per-output time kernel
        std::vector<float> w1t((size_t)td * dim), w2t((size_t)td * td);
        for (int o = 0; o < td; ++o) {
            for (int k = 0; k < dim; ++k) w1t[(size_t)k * td + o] = w1->data[(size_t)o * dim + k];
            for (int k = 0; k < td; ++k) w2t[(size_t)k * td + o] = w2->data[(size_t)o * td + k];
        }
        if ((rc = upload_f32(w1t.data(), w1t.size(), &u->tw1))) return rc;
        if ((rc = upload_f32(b1->data, td, &u->tb1))) return rc;
        if ((rc = upload_f32(w2t.data(), w2t.size(), &u->tw2))) return rc;
        if ((rc = upload_f32(b2->data, td, &u->tb2))) return rc;
        // exp(arange(half) * -(ln 1e4 / (half-1))) evaluated in float32 like the reference
        // (parts/positional_encoding.py:13-15)
        const int half = dim / 2;
        std::vector<float> fr(half);
        const float k = (float)(std::log(10000.0) / (half - 1));
        for (int i = 0; i < half; ++i) fr[i] = expf((float)i * -k);
        if ((rc = upload_f32(fr.data(), half, &u->freqs))) return rc;
    }
    const int nlev = c.n_mults;
    for (int i = 0; i < nlev; ++i) {
        const int ci = dims[i], co = dims[i + 1];
        const std::string p = "downs." + std::to_string(i);
        if ((rc = load_resblock(pm, u, p + ".0", ci, co, true, twb, tbb, u->downs[i].rb[0]))) return rc;
        if ((rc = load_resblock(pm, u, p + ".1", co, co, true, twb, tbb, u->downs[i].rb[1]))) return rc;
        if ((rc = load_attn(pm, u, p + ".2", co, true, u->downs[i].attn))) return rc;
        if (i < nlev - 1 && (rc = load_conv(pm, c.dtype, p + ".3", co, co, 4, 2, 1, true, false, u->downs[i].resample))) return rc;
    }
    const int mid = dims[nlev];
    if ((rc = load_resblock(pm, u, "mid_block1", mid, mid, true, twb, tbb, u->mid1))) return rc;
    if ((rc = load_attn(pm, u, "mid_attn", mid, false, u->mid_attn))) return rc;
    if ((rc = load_resblock(pm, u, "mid_block2", mid, mid, true, twb, tbb, u->mid2))) return rc;
    for (int i = 0; i < nlev - 1; ++i) {
        // reversed(in_out[1:]): level j = nlev-1-i has (dim_in, dim_out) = (dims[j], dims[j+1])
        const int j = nlev - 1 - i;
        const int ci = dims[j], co = dims[j + 1];
        const std::string p = "ups." + std::to_string(i);
        if ((rc = load_resblock(pm, u, p + ".0", co * 2, ci, true, twb, tbb, u->ups[i].rb[0]))) return rc;
        if ((rc = load_resblock(pm, u, p + ".1", ci, ci, true, twb, tbb, u->ups[i].rb[1]))) return rc;
        if ((rc = load_attn(pm, u, p + ".2", ci, true, u->ups[i].attn))) return rc;
        if ((rc = load_conv(pm, c.dtype, p + ".3", ci, ci, 4, 2, 1, true, true, u->ups[i].resample))) return rc;
    }
    if ((rc = load_resblock(pm, u, "final_conv.0", dim, dim, false, twb, tbb, u->final_rb))) return rc;
    {
        const std::string last = c.tail_norm ? "final_conv.3" : "final_conv.1";
        if (c.tail_norm) {
            if ((rc = load_vec(pm, "final_conv.1.weight", dim, &u->tail_g))) return rc;
            if ((rc = load_vec(pm, "final_conv.1.bias", dim, &u->tail_b))) return rc;
        }
        GETP(w, last + ".weight", c.out_dim, dim, 1, 1);
        GETP(b, last + ".bias", c.out_dim);
        if ((rc = upload_f32(w->data, (size_t)c.out_dim * dim, &u->tail_w))) return rc;
        if ((rc = upload_f32(b->data, c.out_dim, &u->tail_bias))) return rc;
    }
    u->tb_total = (int)tbb.size();
    if (u->tb_total > 0) {
        std::vector<float> twbt(twb.size());
        const int nout = u->tb_total;
        for (int o = 0; o < nout; ++o)
            for (int k = 0; k < td; ++k) twbt[(size_t)k * nout + o] = twb[(size_t)o * td + k];
        if ((rc = upload_f32(twbt.data(), twbt.size(), &u->twb))) return rc;
        if ((rc = upload_f32(tbb.data(), tbb.size(), &u->tbb))) return rc;
    }
    u->loaded = true;
    return DMN_OK;
}

extern "C" int dmn_unet_workspace_bytes(DmnUnet* u, int B, int H, int W, size_t* bytes) {
    DMN_REQUIRE(u && bytes, DMN_ERR_INVALID, "dmn_unet_workspace_bytes: null argument");
    size_t high = 0;
    DmnUnet* uu = u;
    int saved_total = uu->tb_total;
    if (!u->loaded) {   // time-bias width is known from the configuration alone
        int total = 0;
        std::vector<int> dims(1, u->cfg.dim);
        for (int i = 0; i < u->cfg.n_mults; ++i) dims.push_back(u->cfg.dim * u->cfg.dim_mults[i]);
        for (int i = 0; i < u->cfg.n_mults; ++i) total += 2 * dims[i + 1];
        total += 2 * dims[u->cfg.n_mults];
        for (int i = 0; i < u->cfg.n_mults - 1; ++i) total += 2 * dims[u->cfg.n_mults - 1 - i];
        uu->tb_total = total;
    }
    size_t handoff = 0;
    int rc = unet_forward_impl(uu, nullptr, nullptr, 0, nullptr, nullptr, nullptr, B, H, W, nullptr, 0, nullptr, true, &high, &handoff);
    uu->tb_total = saved_total;
    if (rc) return rc;
    *bytes = high + 4096;
    // the granule areas of the in-launch hand-offs belong to the handle and grow with the largest shape asked about
    if (!(u->path_mask & 128) && (rc = unet_reserve_handoff(uu, handoff))) return rc;
    return DMN_OK;
}

// Tuning aid: the raw per-workgroup stamps ([n][8] 64-bit words, layout in dmn_conv.hip) of the launch selected by
// DMN_PROBE_SITE, as its last execution on launch chain `chain` left them.
extern "C" int dmn_test_probe_read(DmnUnet* u, int chain, long long* host, int nwgs) {
    DMN_REQUIRE(u && host && chain >= 0 && chain < 2 && nwgs > 0 && nwgs <= kProbeMaxWgs && u->probe_stamps, DMN_ERR_INVALID,
                "dmn_test_probe_read: bad argument or probe never enabled");
    DMN_HIP_CHECK(hipDeviceSynchronize());
    DMN_HIP_CHECK(hipMemcpy(host, u->probe_stamps + (size_t)chain * kProbeMaxWgs * 8, (size_t)nwgs * 8 * sizeof(long long),
                            hipMemcpyDeviceToHost));
    return DMN_OK;
}

// Tuning / measurement aid: choose the convolution launch (its number among an evaluation's convolution launches, -1 =
// the dominant level-0 one) that dmn_unet_probe_dominant(u, 2) clocks, and read what the launch clocked last was:
// shape[5] = {multiply-add work 2 * images * output pixels * Cin * Cout * taps, Cin, Cout, output height, taps}.
// Captured steps are dropped: they hold the old choice.
extern "C" int dmn_test_probe_site(DmnUnet* u, int site, double* shape) {
    DMN_REQUIRE(u, DMN_ERR_INVALID, "dmn_test_probe_site: null handle");
    if (shape)
        for (int i = 0; i < 5; ++i) shape[i] = u->probe_shape[i];
    if (site == u->probe_site) return DMN_OK;
    DMN_HIP_CHECK(hipDeviceSynchronize());
    if (u->graph_exec) (void)hipGraphExecDestroy(u->graph_exec);
    if (u->graph_exec2) (void)hipGraphExecDestroy(u->graph_exec2);
    u->graph_exec = u->graph_exec2 = nullptr;
    u->probe_site = site;
    return DMN_OK;
}

// The hidden vectors of the time path for every row of a sampling loop's table (TimeArgs::h2), computed by one launch
// on `s` before the loop; the evaluations of the loop then skip the two dense layers. The buffer belongs to the handle
// and only grows (never while a stream is capturing: the caller is a loop's entry). rows == 0 forgets the table.
int unet_time_hidden_table(DmnUnet* u, const float* table, int rows, hipStream_t s) {
    u->h2_table_of = nullptr;
    if (!table || rows <= 0 || !u->cfg.with_time_emb || (u->path_mask & 32768) || rows > 65535) return DMN_OK;
    const size_t need = (size_t)rows * 4 * u->cfg.dim;
    if (need > u->h2_cap) {
        DMN_HIP_CHECK(hipDeviceSynchronize());   // an earlier loop may still read the old buffer
        if (u->h2_table) (void)hipFree(u->h2_table);
        u->h2_table = nullptr;
        u->h2_cap = 0;
        DMN_HIP_CHECK(hipMalloc(&u->h2_table, need * sizeof(float)));
        u->h2_cap = need;
    }
    TimeArgs ta;
    memset(&ta, 0, sizeof(ta));
    ta.table = table;
    ta.table_cols = DMN_TABLE_COLS;
    ta.label_col = 0;
    ta.dim = u->cfg.dim;
    ta.freqs = u->freqs;
    ta.w1 = u->tw1; ta.b1 = u->tb1;
    ta.w2 = u->tw2; ta.b2 = u->tb2;
    ta.h2_out = u->h2_table;
    int rc = launch_time_hidden(ta, rows, s);
    if (rc == DMN_OK) u->h2_table_of = table;
    return rc;
}

// Tuning aid: the 100 MHz clock at the start of every network evaluation, per launch chain (a ring of 8192 entries indexed
// by the chain's epoch). host == nullptr: switch the recording on (buffers zeroed; captured steps are dropped because they
// hold the old choice); otherwise copy chain `chain`'s ring to host[8192].
extern "C" int dmn_test_step_trace(DmnUnet* u, int chain, long long* host) {
    DMN_REQUIRE(u && chain >= 0 && chain < 2, DMN_ERR_INVALID, "dmn_test_step_trace: bad argument");
    DMN_HIP_CHECK(hipDeviceSynchronize());
    if (!host) {
        if (!u->step_trace) {
            DMN_HIP_CHECK(hipMalloc(&u->step_trace, (size_t)2 * 8192 * sizeof(long long)));
            if (u->graph_exec) (void)hipGraphExecDestroy(u->graph_exec);
            if (u->graph_exec2) (void)hipGraphExecDestroy(u->graph_exec2);
            u->graph_exec = u->graph_exec2 = nullptr;
        }
        DMN_HIP_CHECK(hipMemset(u->step_trace, 0, (size_t)2 * 8192 * sizeof(long long)));
        return DMN_OK;
    }
    DMN_REQUIRE(u->step_trace, DMN_ERR_INVALID, "dmn_test_step_trace: recording was never switched on");
    DMN_HIP_CHECK(hipMemcpy(host, u->step_trace + (size_t)chain * 8192, 8192 * sizeof(long long), hipMemcpyDeviceToHost));
    return DMN_OK;
}

// Spins of the in-launch hand-offs that ran into their time limit since the handle was created (both chains; waits for
// the device). A count that has grown since the caller last looked means that an evaluation in between used incomplete
// GroupNorm statistics: its results are invalid.
extern "C" int dmn_unet_handoff_giveups(DmnUnet* u, unsigned* count) {
    DMN_REQUIRE(u && count, DMN_ERR_INVALID, "dmn_unet_handoff_giveups: null argument");
    DMN_HIP_CHECK(hipDeviceSynchronize());   // the loops run on non-blocking streams, which a plain copy does not wait for
    *count = u->giveups_carry;
    for (int c = 0; c < 2; ++c)
        if (u->handoff_mem[c]) {
            unsigned hdr[16] = {0};
            DMN_HIP_CHECK(hipMemcpy(hdr, u->handoff_mem[c], sizeof(hdr), hipMemcpyDeviceToHost));
            *count += hdr[1];
            if (hdr[1] && getenv("DMN_HANDOFF_DEBUG"))
                fprintf(stderr, "handoff chain %d: epoch now %u giveups %u; first miss: epoch %u tags %u %u image %u group %u slot %u of %u "
                        "block %x epoch word then %u G %u\n", c, hdr[0], hdr[1], hdr[3], hdr[4], hdr[5], hdr[6], hdr[7], hdr[8], hdr[9],
                        hdr[10], hdr[11], hdr[12]);
        }
    return DMN_OK;
}

extern "C" int dmn_unet_forward(DmnUnet* u, const float* x, const void* t, int t_is_float, float* out, int B, int H, int W,
                                void* workspace, size_t workspace_bytes, void* stream) {
    DMN_REQUIRE(u && x && t && out && workspace, DMN_ERR_INVALID, "dmn_unet_forward: null argument");
    hipStream_t s = (hipStream_t)stream;
    // Inside a caller's stream capture neither an event nor a memset of the exchange areas may be issued: the capture
    // keeps the exchanges and the caller answers for not replaying it beside another handle's sampling.
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    const bool capturing = hipStreamIsCapturing(s, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone;
    bool entered = false;
    if (!capturing && !(u->path_mask & 128)) {
        entered = exchange_guard_enter(u);
        u->exchange_off = !entered;
        const int rc = unet_epoch_reserve(u, 1, s);
        if (rc) {
            u->exchange_off = false;
            exchange_guard_leave(u, s, entered);
            return rc;
        }
    }
    const int rc = unet_forward_impl(u, x, t, t_is_float, nullptr, nullptr, out, B, H, W, workspace, workspace_bytes, s, false, nullptr);
    u->exchange_off = false;
    if (!capturing && !(u->path_mask & 128)) exchange_guard_leave(u, s, entered);
    return rc;
}
