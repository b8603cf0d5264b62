// Training step of the DDPM U-Net behind the C-ABI (include/dmn.h, dmn_train_*): forward with saved activations,
// L_simple, backward through every layer, and the fused clip + AdamW update. Follows DDPM.training_step
// (diffusion_model_nemo/models/ddpm.py:43-57), Unet.forward (modules/unet.py:131-168, use_convnext=False),
// ResnetBlock / Block (parts/convnext.py:8-45,63-86), LinearAttention / Attention (parts/mha.py:7-59) and
// DiffusionLoss (loss/simple_loss.py:37-45); what autograd derives in the reference is written out here as a tape of
// backward closures over the kernels of dmn_train.h. Parameters and gradients live in ONE flat float32 buffer each, in
// the reference's per-tensor layouts (so state_dicts interchange and one RCCL all-reduce covers all gradients); the
// GEMM-ordered copies the kernels consume are rebuilt from it at the start of every step.
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <cmath>

#include <deque>
#include <functional>
#include <string>
#include <vector>

#include "../../include/dmn.h"
#include "dmn_common.h"
#include "dmn_train.h"

namespace {

struct PInfo {
    std::string name;
    int ndim = 0;
    int64_t shape[4] = {0, 0, 0, 0};
    long long off = 0, numel = 0;
};

struct ConvP {
    int w = -1, b = -1;            // parameter indices (bias optional)
    long long f_off = 0, b_off = 0;
    int f_ld = 0, b_ld = 0;
    int Cin = 0, Cout = 0, k = 1, stride = 1, pad = 0;
    bool transposed = false;
};
struct NormP {
    int g = -1, b = -1;
};
struct ResP {
    ConvP c1, c2, res;
    NormP n1, n2;
    bool has_res = false;
    int tb_off = -1;
};
struct AttnP {
    NormP pre, post;
    ConvP qkv, out;
    bool linear = true;
};
struct LevelP {
    ResP rb[2];
    AttnP attn;
    ConvP resample;
    bool has_resample = false;
};

}   // namespace

struct DmnTrainer {
    DmnUnetCfg cfg;
    int loss_kind = 1;
    std::vector<PInfo> params;
    long long n_flat = 0;
    std::vector<RepackDesc> descs;
    RepackDesc* descs_dev = nullptr;
    long long gemm_numel = 0, max_numel = 0;
    // structure
    ConvP init, t1, t2, tball, tail;
    NormP tail_norm;
    std::vector<LevelP> downs, ups;
    ResP mid1, mid2, final_rb;
    AttnP mid_attn;
    int class_embed = -1;                 // parameter index of class_embed.weight (class-conditional networks)
    const long long* labels = nullptr;    // borrowed device labels of the coming steps
    int tb_total = 0;
    long long tb_bias_off = 0;   // the per-block time-linear biases are consecutive in the flat buffer
    // The weight-gradient kernels depend on nothing but their convolution's output gradient and feed nothing but the
    // final layout change, so they run on a second stream beside the chain of input-gradient kernels: one event from
    // the main stream per launch, one join at the end of the step.
    // A step also runs as up to kMaxChains part-batch chains: chain 0 on the caller's stream, the others on `chain[k]`,
    // each with its own side stream.
    static constexpr int kMaxChains = 4;
    hipStream_t side[kMaxChains] = {}, chain[kMaxChains] = {};
    // The device exposes few hardware queues (four by default) and the runtime maps every stream of the process onto
    // one of them; two streams on one queue run their kernels in turn. A process that already holds other streams
    // (a framework's side streams, a collective library's) would otherwise find a chain sharing a queue with its own
    // weight-gradient stream or with the other chain (measured: 12.4 -> 13.8 or 16.7 ms per step). The trainer therefore
    // creates candidate streams and keeps those that a timing probe shows on queues of their own.
    std::vector<hipStream_t> pool;          // every stream created (kept until destroy, also the rejected ones:
                                            // freeing one would hand its queue slot to the next candidate)
    hipStream_t picked_for = nullptr;       // caller stream the roles were picked against
    int picked = 0;                         // roles filled: chain[1], side[0], side[1], chain[2], side[2], ...
    bool have_pick = false;
    int max_chains = 2;
    std::vector<hipEvent_t> events;
    size_t ev_next = 0;
    hipEvent_t join = nullptr, fork = nullptr;
    int use_side = 1, use_dual = 1;
    // workspace region sizes of the last shape
    int sz_B = 0, sz_H = 0, sz_W = 0;
    size_t sz_zero = 0, sz_arena = 0, sz_total = 0;
};

namespace {

struct Builder {
    DmnTrainer* tr;
    long long flat = 0, gemm = 0;
    int add_param(const std::string& name, std::initializer_list<int64_t> shape) {
        PInfo p;
        p.name = name;
        p.ndim = (int)shape.size();
        p.numel = 1;
        int i = 0;
        for (int64_t s : shape) {
            p.shape[i++] = s;
            p.numel *= s;
        }
        p.off = flat;
        flat += (p.numel + 3) & ~3ll;
        tr->params.push_back(p);
        return (int)tr->params.size() - 1;
    }
    void add_desc(ConvP& c, int taps) {
        RepackDesc d;
        memset(&d, 0, sizeof(d));
        d.src_off = tr->params[c.w].off;
        d.n_out = c.Cout;
        d.n_in = c.Cin;
        d.taps = taps;
        d.transposed = c.transposed ? 1 : 0;
        d.f_off = c.f_off;
        d.f_ld = c.f_ld;
        d.b_off = c.b_off;
        d.b_ld = c.b_ld;
        tr->descs.push_back(d);
        const long long n = (long long)c.Cout * c.Cin * taps;
        if (n > tr->max_numel) tr->max_numel = n;
    }
    ConvP conv(const std::string& name, int cout, int cin, int k, int stride, int pad, bool bias, bool transposed = false) {
        ConvP c;
        c.Cin = cin;
        c.Cout = cout;
        c.k = k;
        c.stride = stride;
        c.pad = pad;
        c.transposed = transposed;
        if (transposed) c.w = add_param(name + ".weight", {cin, cout, k, k});
        else c.w = add_param(name + ".weight", {cout, cin, k, k});
        if (bias) c.b = add_param(name + ".bias", {cout});
        const long long n = (long long)cout * cin * k * k;
        c.f_off = c.b_off = gemm;
        c.f_ld = cout;
        c.b_ld = cin;
        gemm += (n + 3) & ~3ll;
        add_desc(c, k * k);
        return c;
    }
    ConvP linear(const std::string& name, int cout, int cin) {
        ConvP c;
        c.Cin = cin;
        c.Cout = cout;
        c.w = add_param(name + ".weight", {cout, cin});
        c.b = add_param(name + ".bias", {cout});
        c.f_off = c.b_off = gemm;
        c.f_ld = cout;
        c.b_ld = cin;
        gemm += ((long long)cout * cin + 3) & ~3ll;
        add_desc(c, 1);
        return c;
    }
    NormP norm(const std::string& name, int c) {
        NormP n;
        n.g = add_param(name + ".weight", {c});
        n.b = add_param(name + ".bias", {c});
        return n;
    }
    ResP resblock(const std::string& pre, int cin, int cout) {
        ResP r;
        r.c1 = conv(pre + ".block1.proj", cout, cin, 3, 1, 1, true);
        r.n1 = norm(pre + ".block1.norm", cout);
        r.c2 = conv(pre + ".block2.proj", cout, cout, 3, 1, 1, true);
        r.n2 = norm(pre + ".block2.norm", cout);
        r.has_res = cin != cout;
        if (r.has_res) r.res = conv(pre + ".res_conv", cout, cin, 1, 1, 0, true);
        return r;
    }
    AttnP attn(const std::string& pre, int c, bool linear_attn) {
        const int hidden = tr->cfg.heads * tr->cfg.dim_head;
        AttnP a;
        a.linear = linear_attn;
        a.pre = norm(pre + ".fn.norm", c);
        a.qkv = conv(pre + ".fn.fn.to_qkv", 3 * hidden, c, 1, 1, 0, false);
        if (linear_attn) {
            a.out = conv(pre + ".fn.fn.to_out.0", c, hidden, 1, 1, 0, true);
            a.post = norm(pre + ".fn.fn.to_out.1", c);
        } else {
            a.out = conv(pre + ".fn.fn.to_out", c, hidden, 1, 1, 0, true);
        }
        return a;
    }
};

struct TimeLin {   // one ResnetBlock's `mlp.1` inside the concatenated [time_dim][tb_total] matrix
    std::string pre;
    int cout;
    ResP* rb;
};

int build_structure(DmnTrainer* tr) {
    const DmnUnetCfg& c = tr->cfg;
    Builder b{tr};
    const int dim = c.dim, td = 4 * dim, nlev = c.n_mults;
    std::vector<int> dims(1, dim);
    for (int i = 0; i < nlev; ++i) dims.push_back(dim * c.dim_mults[i]);
    tr->downs.resize(nlev);
    tr->ups.resize(nlev - 1);
    std::vector<TimeLin> tl;
    tr->init = b.conv("init_conv", dim, c.channels, 7, 1, 3, true);
    tr->t1 = b.linear("time_mlp.1", td, dim);
    tr->t2 = b.linear("time_mlp.3", td, td);
    for (int i = 0; i < nlev; ++i) {
        const int ci = dims[i], co = dims[i + 1];
        const std::string p = "downs." + std::to_string(i);
        tr->downs[i].rb[0] = b.resblock(p + ".0", ci, co);
        tl.push_back({p + ".0", co, &tr->downs[i].rb[0]});
        tr->downs[i].rb[1] = b.resblock(p + ".1", co, co);
        tl.push_back({p + ".1", co, &tr->downs[i].rb[1]});
        tr->downs[i].attn = b.attn(p + ".2", co, true);
        tr->downs[i].has_resample = i < nlev - 1;
        if (i < nlev - 1) tr->downs[i].resample = b.conv(p + ".3", co, co, 4, 2, 1, true);
    }
    const int mid = dims[nlev];
    tr->mid1 = b.resblock("mid_block1", mid, mid);
    tl.push_back({"mid_block1", mid, &tr->mid1});
    tr->mid_attn = b.attn("mid_attn", mid, false);
    tr->mid2 = b.resblock("mid_block2", mid, mid);
    tl.push_back({"mid_block2", mid, &tr->mid2});
    for (int i = 0; i < nlev - 1; ++i) {
        const int j = nlev - 1 - i;
        const int ci = dims[j], co = dims[j + 1];
        const std::string p = "ups." + std::to_string(i);
        tr->ups[i].rb[0] = b.resblock(p + ".0", co * 2, ci);
        tl.push_back({p + ".0", ci, &tr->ups[i].rb[0]});
        tr->ups[i].rb[1] = b.resblock(p + ".1", ci, ci);
        tl.push_back({p + ".1", ci, &tr->ups[i].rb[1]});
        tr->ups[i].attn = b.attn(p + ".2", ci, true);
        tr->ups[i].has_resample = true;
        tr->ups[i].resample = b.conv(p + ".3", ci, ci, 4, 2, 1, true, true);
    }
    tr->final_rb = b.resblock("final_conv.0", dim, dim);
    if (c.tail_norm) {
        tr->tail_norm = b.norm("final_conv.1", dim);
        tr->tail = b.conv("final_conv.3", c.out_dim, dim, 1, 1, 0, true);
    } else {
        tr->tail = b.conv("final_conv.1", c.out_dim, dim, 1, 1, 0, true);
    }
    if (c.num_classes > 0) tr->class_embed = b.add_param("class_embed.weight", {c.num_classes + 1, dim});
    // per-block time linears: weights as column blocks of one [td][tb_total] matrix, biases consecutive
    int total = 0;
    for (auto& t : tl) total += t.cout;
    tr->tb_total = total;
    tr->tball.Cin = td;
    tr->tball.Cout = total;
    tr->tball.f_off = b.gemm;
    tr->tball.f_ld = total;
    b.gemm += (long long)td * total;
    tr->tball.b_off = b.gemm;
    tr->tball.b_ld = td;
    b.gemm += (long long)td * total;
    int col = 0;
    for (auto& t : tl) {
        ConvP cp;
        cp.Cin = td;
        cp.Cout = t.cout;
        cp.w = b.add_param(t.pre + ".mlp.1.weight", {t.cout, td});
        cp.f_off = tr->tball.f_off + col;
        cp.f_ld = total;
        cp.b_off = tr->tball.b_off + (long long)col * td;
        cp.b_ld = td;
        b.add_desc(cp, 1);
        t.rb->tb_off = col;
        col += t.cout;
    }
    tr->tb_bias_off = b.flat;
    for (auto& t : tl) {
        const long long before = b.flat;
        b.add_param(t.pre + ".mlp.1.bias", {t.cout});
        if (b.flat - before != t.cout) return DMN_ERR_INVALID;   // widths are multiples of 4: no padding between them
    }
    tr->n_flat = b.flat;
    tr->gemm_numel = b.gemm;
    return DMN_OK;
}

// ---- one step ---------------------------------------------------------------------------------------------------
struct Bump {
    char* base = nullptr;
    size_t cap = 0, used = 0;
    bool dry = false;
    void* alloc(size_t n) {
        n = (n + 255) & ~(size_t)255;
        void* p = dry ? reinterpret_cast<void*>(4096 + used) : (void*)(base + used);
        used += n;
        if (!dry && used > cap) return nullptr;
        return p;
    }
};

struct Ten {
    void* p = nullptr;
    void* g = nullptr;
    int adt = 0;              // 0 float32, 1 bfloat16 storage (value and gradient alike)
    int C = 0, H = 0, W = 0;
    bool gset = false;
    double* stats = nullptr;  // float64 {sum, sum of squares} per (image, group) left by the producing kernel
    int statsG = 0;
};

struct Run {
    DmnTrainer* tr;
    hipStream_t s;
    bool dry;
    int B;
    int bf16 = 0;             // matrix-core operand type of the GEMMs
    int img_adt = 0;          // storage type of the image-shaped activations (the time path's small tensors stay float32)
    int rc = DMN_OK;
    Bump a, z;
    const float* params;
    float* grads;
    float *wf, *wb, *dwf;
    std::vector<std::function<void()>> tape;
    std::deque<Ten> tens;
    hipStream_t side = nullptr;
    const long long* labels = nullptr;   // class labels of this chain's images
    bool side_used = false;
    // stream for a weight-gradient launch whose inputs this chain's stream has produced by now
    hipStream_t wgrad_stream() {
        if (!tr->use_side || !side || dry || rc != DMN_OK) return s;
        if (tr->ev_next >= tr->events.size()) {
            hipEvent_t e = nullptr;
            if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return s;
            tr->events.push_back(e);
        }
        hipEvent_t e = tr->events[tr->ev_next++];
        if (hipEventRecord(e, s) != hipSuccess || hipStreamWaitEvent(side, e, 0) != hipSuccess) return s;
        side_used = true;
        return side;
    }

    void fail_ws() {
        if (rc == DMN_OK) {
            dmn_set_error("training workspace too small");
            rc = DMN_ERR_WORKSPACE;
        }
    }
    float* alloc(size_t floats) {
        float* p = (float*)a.alloc(floats * 4);
        if (!p) fail_ws();
        return p;
    }
    float* zalloc(size_t floats) {
        float* p = (float*)z.alloc(floats * 4);
        if (!p) fail_ws();
        return p;
    }
    double* zalloc_stats(size_t pairs) { return reinterpret_cast<double*>(zalloc(pairs * 4)); }   // float64 {sum, sum of squares}
    size_t numel(const Ten* t) const { return (size_t)B * t->H * t->W * t->C; }
    void* alloc_elems(size_t n, int adt) {
        void* p = a.alloc(n * (adt == 1 ? 2 : 4));
        if (!p) fail_ws();
        return p;
    }
    Ten* nt(int C, int H, int W, int adt) {
        tens.emplace_back();
        Ten* t = &tens.back();
        t->C = C;
        t->H = H;
        t->W = W;
        t->adt = adt;
        t->p = alloc_elems(numel(t), adt);
        return t;
    }
    // gradient buffer of t and whether the coming write must accumulate
    void* G(Ten* t, int* acc) {
        if (!t->g) t->g = alloc_elems(numel(t), t->adt);
        *acc = t->gset ? 1 : 0;
        t->gset = true;
        return t->g;
    }
    bool live() const { return !dry && rc == DMN_OK; }
    void L(int r) {
        if (rc == DMN_OK && r != DMN_OK) rc = r;
    }
    const float* P(int idx) const { return idx < 0 ? nullptr : params + tr->params[idx].off; }
    float* GP(int idx) const { return idx < 0 ? nullptr : grads + tr->params[idx].off; }
};

#define RUN(r, expr)                 \
    do {                             \
        if ((r).live()) (r).L(expr); \
    } while (0)

// statsG > 0: the consumer is a GroupNorm with that many groups; the producing kernel leaves its statistics when it can
Ten* conv(Run& r, Ten* x, const ConvP& w, bool need_dx = true, const float* bias_override = nullptr, float* dbias_override = nullptr,
          int statsG = 0) {
    const int Ho = w.transposed ? (x->H - 1) * w.stride - 2 * w.pad + w.k : (x->H + 2 * w.pad - w.k) / w.stride + 1;
    const int Wo = w.transposed ? (x->W - 1) * w.stride - 2 * w.pad + w.k : (x->W + 2 * w.pad - w.k) / w.stride + 1;
    Ten* y = r.nt(w.Cout, Ho, Wo, x->adt);
    const float* bias = bias_override ? bias_override : r.P(w.b);
    float* dbias = dbias_override ? dbias_override : r.GP(w.b);
    PGemmArgs a;
    memset(&a, 0, sizeof(a));
    a.in = x->p;
    a.ldin = x->C;
    a.Cin = w.Cin;
    a.B = r.B;
    a.Hi = x->H;
    a.Wi = x->W;
    a.Ho = Ho;
    a.Wo = Wo;
    a.kh = a.kw = w.k;
    a.stride = w.stride;
    a.pad = w.pad;
    a.tmode = w.transposed ? 1 : 0;
    a.w = r.wf + w.f_off;
    a.ldw = w.f_ld;
    a.bias = bias;
    a.out = y->p;
    a.ldout = w.Cout;
    a.N = w.Cout;
    a.bf16 = r.bf16;
    a.adt = x->adt;
    a.statsG = statsG;
    if (statsG > 0 && pgemm_stats_supported(a)) {
        y->stats = a.stats = r.zalloc_stats((size_t)r.B * statsG);
        y->statsG = statsG;
    }
    RUN(r, launch_pgemm(a, r.s));
    Run* rp = &r;
    const ConvP* wp = &w;
    r.tape.push_back([rp, x, y, wp, need_dx, dbias, Ho, Wo]() {
        Run& r = *rp;
        const ConvP& w = *wp;
        WGemmArgs g;
        memset(&g, 0, sizeof(g));
        g.x = x->p;
        g.ldx = x->C;
        g.Cin = w.Cin;
        g.B = r.B;
        g.Hi = x->H;
        g.Wi = x->W;
        g.Ho = Ho;
        g.Wo = Wo;
        g.kh = g.kw = w.k;
        g.stride = w.stride;
        g.pad = w.pad;
        g.tmode = w.transposed ? 1 : 0;
        g.dy = y->g;
        g.lddy = w.Cout;
        g.N = w.Cout;
        g.dw = r.dwf + w.f_off;
        g.lddw = w.f_ld;
        g.bf16 = r.bf16;
        g.adt = x->adt;
        g.dbias = dbias;
        RUN(r, launch_wgemm(g, r.wgrad_stream()));
        if (need_dx) {
            PGemmArgs d;
            memset(&d, 0, sizeof(d));
            d.in = y->g;
            d.ldin = w.Cout;
            d.Cin = w.Cout;
            d.B = r.B;
            d.Hi = Ho;
            d.Wi = Wo;
            d.Ho = x->H;
            d.Wo = x->W;
            d.kh = d.kw = w.k;
            d.stride = w.stride;
            d.pad = w.pad;
            d.tmode = w.transposed ? 0 : 1;
            d.w = r.wb + w.b_off;
            d.ldw = w.b_ld;
            d.out = r.G(x, &d.accumulate);
            d.ldout = x->C;
            d.N = w.Cin;
            d.bf16 = r.bf16;
            d.adt = x->adt;
            RUN(r, launch_pgemm(d, r.s));
        }
    });
    return y;
}

// the gradient of `t` receives `src` (a gradient buffer nothing reads after the calling closure): the first writer
// simply takes the buffer over, later ones accumulate
void pass_grad(Run& r, Ten* t, void* src) {
    if (!t->gset) {
        t->g = src;
        t->gset = true;
        return;
    }
    RUN(r, launch_axpby(t->g, src, t->adt, (long long)r.numel(t), 1, r.s));
}

// y = act(GroupNorm(x)) + tb + res
// ystats: also leave the per-image statistics of y for a following GroupNorm(1)
Ten* gnorm(Run& r, Ten* x, const NormP& n, int G, int act, const float* tb, float* dtb, int tb_ld, Ten* res = nullptr,
           bool ystats = false) {
    Ten* y = r.nt(x->C, x->H, x->W, x->adt);
    const bool have = x->stats && x->statsG == G;
    double* stats = have ? x->stats : r.zalloc_stats((size_t)r.B * G);
    if (ystats) {
        y->stats = r.zalloc_stats((size_t)r.B);
        y->statsG = 1;
    }
    GnArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x->p;
    a.stats = stats;
    a.gamma = r.P(n.g);
    a.beta = r.P(n.b);
    a.tb = tb;
    a.tb_ld = tb_ld;
    a.res = res ? res->p : nullptr;
    a.y = y->p;
    a.B = r.B;
    a.HW = x->H * x->W;
    a.C = x->C;
    a.G = G;
    a.act = act;
    a.adt = x->adt;
    a.ystats = y->stats;
    if (!have) RUN(r, launch_gn_stats(x->p, x->adt, r.B, a.HW, a.C, G, stats, r.s));
    RUN(r, launch_gn_apply(a, r.s));
    Run* rp = &r;
    float* dgamma = r.GP(n.g);
    float* dbeta = r.GP(n.b);
    r.tape.push_back([rp, x, y, a, dgamma, dbeta, dtb, res]() mutable {
        Run& r = *rp;
        a.dy = y->g;
        a.sums = r.zalloc((size_t)r.B * a.C * 3);
        a.dgamma = dgamma;
        a.dbeta = dbeta;
        a.dtb = dtb;
        a.dx = r.G(x, &a.accumulate);
        RUN(r, launch_gn_bwd(a, r.s));
        if (res) pass_grad(r, res, y->g);   // after the kernels above are queued: they read y->g before anything adds to it
    });
    return y;
}

Ten* add(Run& r, Ten* a, Ten* b) {
    Ten* y = r.nt(a->C, a->H, a->W, a->adt);
    RUN(r, launch_add(y->p, a->p, b->p, y->adt, (long long)r.numel(y), r.s));
    Run* rp = &r;
    r.tape.push_back([rp, a, b, y]() {
        Run& r = *rp;
        int acc;
        void* gb = r.G(b, &acc);
        RUN(r, launch_axpby(gb, y->g, y->adt, (long long)r.numel(y), acc, r.s));
        pass_grad(r, a, y->g);
    });
    return y;
}

Ten* cat(Run& r, Ten* a, Ten* b) {
    Ten* y = r.nt(a->C + b->C, a->H, a->W, a->adt);
    const long long rows = (long long)r.B * a->H * a->W;
    RUN(r, launch_copy_slice(y->p, y->C, 0, a->p, a->C, 0, a->C, rows, y->adt, 0, r.s));
    RUN(r, launch_copy_slice(y->p, y->C, a->C, b->p, b->C, 0, b->C, rows, y->adt, 0, r.s));
    Run* rp = &r;
    r.tape.push_back([rp, a, b, y, rows]() {
        Run& r = *rp;
        int acc;
        void* ga = r.G(a, &acc);
        RUN(r, launch_copy_slice(ga, a->C, 0, y->g, y->C, 0, a->C, rows, y->adt, acc, r.s));
        void* gb = r.G(b, &acc);
        RUN(r, launch_copy_slice(gb, b->C, 0, y->g, y->C, a->C, b->C, rows, y->adt, acc, r.s));
    });
    return y;
}

Ten* activation(Run& r, Ten* x, int act) {
    Ten* y = r.nt(x->C, x->H, x->W, x->adt);
    RUN(r, launch_act_fwd(x->p, y->p, x->adt, (long long)r.numel(x), act, r.s));
    Run* rp = &r;
    r.tape.push_back([rp, x, y, act]() {
        Run& r = *rp;
        int acc;
        void* gx = r.G(x, &acc);
        RUN(r, launch_act_bwd(x->p, y->g, gx, x->adt, (long long)r.numel(x), act, acc, r.s));
    });
    return y;
}

Ten* attn_core(Run& r, Ten* qkv, bool linear) {
    const int heads = r.tr->cfg.heads, hidden = heads * 32, n = qkv->H * qkv->W;
    Ten* y = r.nt(hidden, qkv->H, qkv->W, qkv->adt);
    float* ctx = linear ? r.zalloc((size_t)r.B * heads * 1024) : nullptr;
    float* st = linear ? r.zalloc((size_t)r.B * heads * 64) : r.alloc((size_t)r.B * heads * n * 2);
    if (linear) {
        float* pmax = r.alloc((size_t)r.B * heads * linattn_train_splits(r.B, n, heads) * 32);
        RUN(r, launch_linattn_train_fwd(qkv->p, y->p, qkv->adt, ctx, st, pmax, r.B, n, heads, r.bf16, r.s));
    }
    else RUN(r, launch_attn_train_fwd(qkv->p, y->p, qkv->adt, st, r.B, n, heads, r.bf16, r.s));
    Run* rp = &r;
    r.tape.push_back([rp, qkv, y, ctx, st, linear, n, heads]() {
        Run& r = *rp;
        int acc;
        void* g = r.G(qkv, &acc);   // qkv has this single consumer: the kernels overwrite
        if (acc && r.rc == DMN_OK) {
            dmn_set_error("internal: attention input gradient already set");
            r.rc = DMN_ERR_INVALID;
        }
        if (linear) {
            float* dctx = r.zalloc((size_t)r.B * heads * 1024);
            RUN(r, launch_linattn_train_bwd(qkv->p, y->g, qkv->adt, ctx, st, dctx, g, r.B, n, heads, r.bf16, r.s));
        }
        else RUN(r, launch_attn_train_bwd(qkv->p, y->p, y->g, qkv->adt, st, g, r.B, n, heads, r.bf16, r.s));
    });
    return y;
}

// out_stats: the block's output feeds a GroupNorm(1) (the PreNorm of an attention)
Ten* resblock(Run& r, Ten* x, const ResP& w, Ten* tb, bool out_stats = false) {
    const int G = r.tr->cfg.groups;
    Ten* h = conv(r, x, w.c1, true, nullptr, nullptr, G);
    const bool time = tb && w.tb_off >= 0;
    h = gnorm(r, h, w.n1, G, 1, time ? (const float*)tb->p + w.tb_off : nullptr, time ? (float*)tb->g + w.tb_off : nullptr,
              r.tr->tb_total);
    h = conv(r, h, w.c2, true, nullptr, nullptr, G);
    // the shortcut is computed first so that its sum with the block's tail comes out of the GroupNorm kernel
    Ten* res = w.has_res ? conv(r, x, w.res) : x;
    return gnorm(r, h, w.n2, G, 1, nullptr, nullptr, 0, res, out_stats);
}

Ten* attention(Run& r, Ten* x, const AttnP& w) {
    Ten* xn = gnorm(r, x, w.pre, 1, 0, nullptr, nullptr, 0);
    Ten* qkv = conv(r, xn, w.qkv);
    Ten* o = attn_core(r, qkv, w.linear);
    Ten* y = conv(r, o, w.out, true, nullptr, nullptr, w.linear ? 1 : 0);
    if (w.linear) y = gnorm(r, y, w.post, 1, 0, nullptr, nullptr, 0);
    return add(r, y, x);
}

struct StepIO {
    const float *x0, *noise;
    const long long* t;
    const float *sa, *sb;
    int n_timesteps;
    float* loss;
    float* model_out;
};

struct Sizes {
    size_t zero = 0, arena = 0, total = 0;
};

// Forward, loss and backward of `r.B` images on stream r.s: everything between the shared prologue (zero fill, weight
// layout change) and the shared epilogue (gradient layout change). n_mean: element count the mean loss runs over (the
// whole batch, also when this chain holds a part of it).
int run_chain(Run& r, const StepIO& io, int H, int W, long long n_mean) {
    DmnTrainer* tr = r.tr;
    const DmnUnetCfg& c = tr->cfg;
    const int nlev = c.n_mults, B = r.B;
    hipStream_t s = r.s;
    // inputs
    Ten* xt = r.nt(c.channels, H, W, r.img_adt);
    Ten* tgt = r.nt(c.channels, H, W, 0);
    RUN(r, launch_train_ingest(io.x0, io.noise, io.t, io.sa, io.sb, io.n_timesteps, xt->p, xt->adt, (float*)tgt->p, B, c.channels, H, W, s));
    // time path (modules/unet.py:61-66, parts/convnext.py:68-72,81-83)
    Ten* emb = r.nt(c.dim, 1, 1, 0);
    RUN(r, launch_sinus(io.t, 0, B, c.dim, (float*)emb->p, s));
    Ten* h1 = conv(r, emb, tr->t1, false);
    Ten* h2 = activation(r, h1, 1);
    Ten* temb = conv(r, h2, tr->t2);
    Ten* st = activation(r, temb, 0);
    Ten* tb = conv(r, st, tr->tball, true, r.params + tr->tb_bias_off, r.grads + tr->tb_bias_off);
    tb->g = r.zalloc((size_t)B * tr->tb_total);   // every block adds its share
    tb->gset = true;

    Ten* cur = conv(r, xt, tr->init, false);
    if (tr->class_embed >= 0) {   // x = init_conv(x) + class_embed(classes)[:, :, None, None]  (unet.py:134-141), in place
        const long long* labels = r.labels;
        RUN(r, launch_class_bias_fwd(cur->p, cur->adt, r.P(tr->class_embed), labels, c.num_classes, B, H * W, c.dim, s));
        Run* rp = &r;
        r.tape.push_back([rp, cur, tr, labels, H, W]() {
            Run& r = *rp;
            RUN(r, launch_class_bias_bwd(cur->g, cur->adt, labels, tr->cfg.num_classes, r.GP(tr->class_embed), r.B, H * W, tr->cfg.dim,
                                         r.s));
        });
    }
    std::vector<Ten*> skips;
    for (int i = 0; i < nlev; ++i) {
        cur = resblock(r, cur, tr->downs[i].rb[0], tb);
        cur = resblock(r, cur, tr->downs[i].rb[1], tb, true);
        cur = attention(r, cur, tr->downs[i].attn);
        if (i > 0) skips.push_back(cur);   // the top-level skip is never consumed (unet.py:95)
        if (tr->downs[i].has_resample) cur = conv(r, cur, tr->downs[i].resample);
    }
    cur = resblock(r, cur, tr->mid1, tb, true);
    cur = attention(r, cur, tr->mid_attn);
    cur = resblock(r, cur, tr->mid2, tb);
    for (int i = 0; i < nlev - 1; ++i) {
        Ten* sk = skips.back();
        skips.pop_back();
        cur = cat(r, cur, sk);
        cur = resblock(r, cur, tr->ups[i].rb[0], tb);
        cur = resblock(r, cur, tr->ups[i].rb[1], tb, true);
        cur = attention(r, cur, tr->ups[i].attn);
        cur = conv(r, cur, tr->ups[i].resample);
    }
    cur = resblock(r, cur, tr->final_rb, nullptr);
    if (c.tail_norm) cur = gnorm(r, cur, tr->tail_norm, c.groups, 1, nullptr, nullptr, 0);
    Ten* out = conv(r, cur, tr->tail);
    DMN_REQUIRE(out->H == H && out->W == W, DMN_ERR_INVALID, "internal: output %dx%d", out->H, out->W);

    // loss and its gradient (the scalar was zeroed in the prologue)
    int acc;
    void* dout = r.G(out, &acc);
    RUN(r, launch_loss(out->p, (const float*)tgt->p, out->adt, (long long)r.numel(out), n_mean, tr->loss_kind, r.dry ? nullptr : io.loss,
                       dout, s));
    if (io.model_out) RUN(r, launch_nhwc_to_nchw(out->p, out->adt, io.model_out, B, c.out_dim, H * W, s));

    for (size_t i = r.tape.size(); i-- > 0;) r.tape[i]();
    return r.rc;
}

// chains a step of B images is cut into: two half-batch chains on two streams keep more of the chip busy through the
// low-resolution levels, whose launches have fewer workgroups than there are CUs
int chains_for(const DmnTrainer* tr, int B) {
    if (!tr->use_dual) return 1;
    for (int n = tr->max_chains; n > 1; n >>= 1)
        if (B % n == 0 && B / n >= 2) return n;
    return 1;
}

// true when kernels launched on a and b run one after the other (same hardware queue): two 300 us single-wave waits
// take ~300 us on separate queues and ~600 us on a shared one. Best of two trials.
bool streams_share_queue(hipStream_t a, hipStream_t b) {
    double best = 1e30;
    for (int trial = 0; trial < 2; ++trial) {
        if (hipStreamSynchronize(a) != hipSuccess || hipStreamSynchronize(b) != hipSuccess) return false;
        const auto t0 = std::chrono::steady_clock::now();
        if (launch_spin(300, a) != DMN_OK || launch_spin(300, b) != DMN_OK) return false;
        (void)hipStreamSynchronize(a);
        (void)hipStreamSynchronize(b);
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        if (us < best) best = us;
    }
    return best > 480.0;
}

// Fills chain[1..] and side[0..] for `nchain` chains against the caller's stream s. Roles are filled in the order of
// what a shared queue would cost: the second chain, then the weight-gradient streams.
int pick_streams(DmnTrainer* tr, hipStream_t s, int nchain) {
    const int need = (nchain - 1) + (tr->use_side ? nchain : 0);
    if (tr->have_pick && tr->picked_for == s && tr->picked >= need) return DMN_OK;
    std::vector<hipStream_t> distinct, rest;
    size_t next = 0;
    const size_t max_pool = 12;
    while ((int)distinct.size() < need) {
        hipStream_t c = nullptr;
        if (next < tr->pool.size()) {
            c = tr->pool[next];
        } else if (tr->pool.size() < max_pool) {
            DMN_HIP_CHECK(hipStreamCreateWithFlags(&c, hipStreamNonBlocking));
            tr->pool.push_back(c);
            if (launch_spin(0, c) != DMN_OK) return DMN_ERR_HIP;   // first use: the runtime binds the stream to a queue
        } else {
            break;
        }
        ++next;
        bool shared = streams_share_queue(c, s);
        for (size_t i = 0; i < distinct.size() && !shared; ++i) shared = streams_share_queue(c, distinct[i]);
        (shared ? rest : distinct).push_back(c);
    }
    // fewer separate queues than roles (more chains than queues, or a lower queue limit): the rest share
    for (size_t i = 0; (int)distinct.size() < need; ++i) {
        if (rest.empty()) return DMN_ERR_HIP;
        distinct.push_back(rest[i % rest.size()]);
    }
    int k = 0;
    auto take = [&]() { return distinct[k++]; };
    for (int c = 0; c < DmnTrainer::kMaxChains; ++c) tr->side[c] = tr->chain[c] = nullptr;
    if (nchain > 1) tr->chain[1] = take();
    if (tr->use_side) tr->side[0] = take();
    if (tr->use_side && nchain > 1) tr->side[1] = take();
    for (int c = 2; c < nchain; ++c) {
        tr->chain[c] = take();
        if (tr->use_side) tr->side[c] = take();
    }
    tr->picked_for = s;
    tr->picked = need;
    tr->have_pick = true;
    return DMN_OK;
}

int run_step(DmnTrainer* tr, const float* params, float* grads, const StepIO& io, int B, int H, int W, void* ws, size_t ws_bytes,
             hipStream_t s, bool dry, Sizes* sz) {
    const DmnUnetCfg& c = tr->cfg;
    const int nlev = c.n_mults, div = 1 << (nlev - 1);
    DMN_REQUIRE(B >= 1 && H % div == 0 && W % div == 0 && H >= div && W >= div, DMN_ERR_INVALID,
                "training: image %dx%d must be a multiple of %d", H, W, div);
    const int nchain = chains_for(tr, B), Bc = B / nchain;
    // workspace: [wf | wb | zero region: dwf + one zero arena per chain | one arena per chain]
    const size_t gemm_bytes = ((size_t)tr->gemm_numel * 4 + 255) & ~(size_t)255;
    if (!dry) {
        // region sizes come from a dry pass of the same code, cached per shape
        if (tr->sz_B != B || tr->sz_H != H || tr->sz_W != W) {
            Sizes q;
            int rc = run_step(tr, params, grads, io, B, H, W, nullptr, 0, s, true, &q);
            if (rc) return rc;
            tr->sz_B = B;
            tr->sz_H = H;
            tr->sz_W = W;
            tr->sz_zero = q.zero;
            tr->sz_arena = q.arena;
            tr->sz_total = q.total;
        }
        DMN_REQUIRE(ws_bytes >= tr->sz_total, DMN_ERR_WORKSPACE, "training workspace: %zu bytes given, %zu needed", ws_bytes,
                    tr->sz_total);
        DMN_REQUIRE((reinterpret_cast<uintptr_t>(ws) & 255) == 0, DMN_ERR_INVALID, "training workspace must be 256-byte aligned");
        if (!tr->join) {
            DMN_HIP_CHECK(hipEventCreateWithFlags(&tr->fork, hipEventDisableTiming));
            DMN_HIP_CHECK(hipEventCreateWithFlags(&tr->join, hipEventDisableTiming));
        }
        {
            int rcp = pick_streams(tr, s, nchain);
            if (rcp) return rcp;
        }
        if (!tr->descs_dev) {   // first step on the device: upload the layout table (creation itself needs no GPU)
            DMN_HIP_CHECK(hipMalloc((void**)&tr->descs_dev, tr->descs.size() * sizeof(RepackDesc)));
            DMN_HIP_CHECK(hipMemcpy(tr->descs_dev, tr->descs.data(), tr->descs.size() * sizeof(RepackDesc), hipMemcpyHostToDevice));
        }
    }
    char* base = (char*)ws;
    float* wf = (float*)base;
    float* wb = (float*)(base + gemm_bytes);
    float* dwf = (float*)(base + 2 * gemm_bytes);
    if (!dry) {
        DMN_HIP_CHECK(hipMemsetAsync(dwf, 0, gemm_bytes + nchain * tr->sz_zero, s));
        DMN_HIP_CHECK(hipMemsetAsync(grads, 0, (size_t)tr->n_flat * 4, s));
        DMN_HIP_CHECK(hipMemsetAsync(io.loss, 0, 4, s));
        int rc = launch_repack(tr->descs_dev, (int)tr->descs.size(), tr->max_numel, params, wf, wb, s);
        if (rc) return rc;
        if (nchain > 1) {
            DMN_HIP_CHECK(hipEventRecord(tr->fork, s));
            for (int k = 1; k < nchain; ++k) DMN_HIP_CHECK(hipStreamWaitEvent(tr->chain[k], tr->fork, 0));
        }
        tr->ev_next = 0;
    }
    const long long n_mean = (long long)B * H * W * c.out_dim;
    const size_t img = (size_t)c.channels * H * W;
    int rc = DMN_OK;
    bool side_used[DmnTrainer::kMaxChains] = {};
    Sizes one;
    for (int k = 0; k < (dry ? 1 : nchain) && rc == DMN_OK; ++k) {
        Run r;
        r.tr = tr;
        r.s = k == 0 ? s : tr->chain[k];
        r.side = tr->side[k];
        r.dry = dry;
        r.B = Bc;
        r.bf16 = c.dtype >= 1 ? 1 : 0;
        r.img_adt = c.dtype == 2 ? 1 : 0;
        r.params = params;
        r.grads = grads;
        r.wf = wf;
        r.wb = wb;
        r.dwf = dwf;
        r.labels = tr->labels ? tr->labels + (size_t)k * Bc : nullptr;
        if (dry) {
            r.z.dry = r.a.dry = true;
        } else {
            r.z.base = base + 3 * gemm_bytes + k * tr->sz_zero;
            r.z.cap = tr->sz_zero;
            r.a.base = base + 3 * gemm_bytes + nchain * tr->sz_zero + k * tr->sz_arena;
            r.a.cap = tr->sz_arena;
        }
        StepIO ck = io;
        if (!dry) {
            ck.x0 = io.x0 + (size_t)k * Bc * img;
            ck.noise = io.noise + (size_t)k * Bc * img;
            ck.t = io.t + (size_t)k * Bc;
            if (io.model_out) ck.model_out = io.model_out + (size_t)k * Bc * c.out_dim * H * W;
        }
        rc = run_chain(r, ck, H, W, n_mean);
        side_used[k] = r.side_used;
        if (dry) {
            one.zero = (r.z.used + 255) & ~(size_t)255;
            one.arena = (r.a.used + 255) & ~(size_t)255;
        }
    }
    if (rc) {
        // kernels already queued on the internal streams still use the caller's buffers: drain them before the
        // caller learns of the failure and may free the workspace
        if (!dry) {
            for (int k = 0; k < nchain; ++k) {
                if (tr->side[k]) (void)hipStreamSynchronize(tr->side[k]);
                if (k && tr->chain[k]) (void)hipStreamSynchronize(tr->chain[k]);
            }
            (void)hipStreamSynchronize(s);
        }
        return rc;
    }
    if (!dry) {
        // the layout change below reads what every stream accumulated
        for (int k = 0; k < nchain; ++k)
            if (side_used[k]) {
                DMN_HIP_CHECK(hipEventRecord(tr->join, tr->side[k]));
                DMN_HIP_CHECK(hipStreamWaitEvent(s, tr->join, 0));
            }
        for (int k = 1; k < nchain; ++k) {
            DMN_HIP_CHECK(hipEventRecord(tr->join, tr->chain[k]));
            DMN_HIP_CHECK(hipStreamWaitEvent(s, tr->join, 0));
        }
        rc = launch_unpack_grads(tr->descs_dev, (int)tr->descs.size(), tr->max_numel, dwf, grads, s);
    } else if (sz) {
        sz->zero = one.zero;
        sz->arena = one.arena;
        sz->total = 3 * gemm_bytes + nchain * (one.zero + one.arena);
    }
    return rc;
}

}   // namespace

extern "C" int dmn_streams_share_queue(void* a, void* b, int* shared) {
    DMN_REQUIRE(shared, DMN_ERR_INVALID, "dmn_streams_share_queue: null argument");
    *shared = streams_share_queue((hipStream_t)a, (hipStream_t)b) ? 1 : 0;
    return DMN_OK;
}

// Softmax attention core of the training step on caller-provided tensors (unit parity tests): forward (out, lse) and
// backward (dqkv) of out = softmax(q k^T / sqrt(32)) v for qkv [B][n][3*heads*32]. adt 0 = fp32 tensors, 1 = bf16;
// bf16_operands selects the matrix-core kernels where they apply. lse: [2][B*heads][n] floats.
extern "C" int dmn_test_attn_train(int adt, const void* qkv, const void* dout, void* out, void* dqkv, float* lse, int B, int n,
                                   int heads, int bf16_operands, void* stream) {
    DMN_REQUIRE(qkv && dout && out && dqkv && lse && B > 0 && n > 0 && heads > 0, DMN_ERR_INVALID, "dmn_test_attn_train: bad argument");
    int rc = launch_attn_train_fwd(qkv, out, adt, lse, B, n, heads, bf16_operands, (hipStream_t)stream);
    if (rc) return rc;
    return launch_attn_train_bwd(qkv, out, dout, adt, lse, dqkv, B, n, heads, bf16_operands, (hipStream_t)stream);
}

extern "C" int dmn_train_create(const DmnUnetCfg* cfg, int loss_kind, DmnTrainer** out) {
    DMN_REQUIRE(cfg && out, DMN_ERR_INVALID, "dmn_train_create: null argument");
    DMN_REQUIRE(cfg->n_mults >= 1 && cfg->n_mults <= DMN_MAX_MULTS, DMN_ERR_INVALID, "n_mults=%d", cfg->n_mults);
    DMN_REQUIRE(cfg->dim % 32 == 0 && cfg->dim >= 32, DMN_ERR_LIMIT, "dim=%d must be a multiple of 32", cfg->dim);
    DMN_REQUIRE(cfg->dim_head == 32 && cfg->heads >= 1, DMN_ERR_INVALID, "dim_head=%d (only 32 supported)", cfg->dim_head);
    DMN_REQUIRE(cfg->with_time_emb == 1, DMN_ERR_INVALID, "training needs the time embedding");
    DMN_REQUIRE(cfg->num_classes >= 0 && (cfg->num_classes == 0 || cfg->dim <= 256), DMN_ERR_INVALID, "num_classes=%d", cfg->num_classes);
    DMN_REQUIRE(cfg->out_dim == cfg->channels, DMN_ERR_INVALID, "L_simple needs out_dim == channels (got %d, %d)", cfg->out_dim, cfg->channels);
    DMN_REQUIRE(cfg->groups >= 1 && cfg->groups <= 64 && cfg->dim % cfg->groups == 0 && (cfg->dim / cfg->groups) % 4 == 0,
                DMN_ERR_INVALID, "groups=%d", cfg->groups);
    DMN_REQUIRE(loss_kind >= 0 && loss_kind <= 2, DMN_ERR_INVALID, "loss_kind=%d", loss_kind);
    DMN_REQUIRE(cfg->dtype >= 0 && cfg->dtype <= 2, DMN_ERR_INVALID, "dtype=%d", cfg->dtype);
    int maxc = cfg->dim;
    for (int i = 0; i < cfg->n_mults; ++i) {
        DMN_REQUIRE(cfg->dim_mults[i] >= 1, DMN_ERR_INVALID, "dim_mults[%d]", i);
        if (cfg->dim * cfg->dim_mults[i] > maxc) maxc = cfg->dim * cfg->dim_mults[i];
    }
    DMN_REQUIRE(maxc <= 1024 && 256 % (maxc / 4) == 0, DMN_ERR_LIMIT, "widest level %d channels: must divide 1024", maxc);
    DmnTrainer* tr = new DmnTrainer();
    tr->cfg = *cfg;
    tr->loss_kind = loss_kind;
    if (const char* e = getenv("DMN_TRAIN_SIDE_STREAM")) tr->use_side = atoi(e) != 0;   // tuning aids: 0 keeps one stream,
    if (const char* e = getenv("DMN_TRAIN_DUAL_CHAIN")) tr->use_dual = atoi(e) != 0;    // 0 keeps one chain
    if (const char* e = getenv("DMN_TRAIN_CHAINS")) {
        const int n = atoi(e);
        tr->max_chains = n >= 4 ? 4 : (n >= 2 ? 2 : 1);
    }
    int rc = build_structure(tr);
    if (rc) {
        delete tr;
        dmn_set_error("dmn_train_create: unsupported widths");
        return rc;
    }
    *out = tr;
    return DMN_OK;
}

extern "C" void dmn_train_destroy(DmnTrainer* tr) {
    if (!tr) return;
    if (tr->descs_dev) (void)hipFree(tr->descs_dev);
    for (hipEvent_t e : tr->events) (void)hipEventDestroy(e);
    if (tr->join) (void)hipEventDestroy(tr->join);
    if (tr->fork) (void)hipEventDestroy(tr->fork);
    for (hipStream_t st : tr->pool) (void)hipStreamDestroy(st);
    delete tr;
}

extern "C" int dmn_train_set_class_labels(DmnTrainer* tr, const int64_t* labels) {
    DMN_REQUIRE(tr, DMN_ERR_INVALID, "dmn_train_set_class_labels: null handle");
    DMN_REQUIRE(tr->cfg.num_classes > 0 || labels == nullptr, DMN_ERR_INVALID, "the network has no class embedding");
    tr->labels = reinterpret_cast<const long long*>(labels);
    return DMN_OK;
}

extern "C" int dmn_train_num_tensors(const DmnTrainer* tr) { return tr ? (int)tr->params.size() : 0; }
extern "C" int64_t dmn_train_flat_size(const DmnTrainer* tr) { return tr ? (int64_t)tr->n_flat : 0; }

extern "C" int dmn_train_tensor_info(const DmnTrainer* tr, int idx, const char** name, int64_t* offset, int* ndim, int64_t* shape4) {
    DMN_REQUIRE(tr && idx >= 0 && idx < (int)tr->params.size(), DMN_ERR_INVALID, "dmn_train_tensor_info: index %d", idx);
    const PInfo& p = tr->params[idx];
    if (name) *name = p.name.c_str();
    if (offset) *offset = p.off;
    if (ndim) *ndim = p.ndim;
    if (shape4)
        for (int i = 0; i < 4; ++i) shape4[i] = p.shape[i];
    return DMN_OK;
}

extern "C" int dmn_train_workspace_bytes(DmnTrainer* tr, int B, int H, int W, size_t* bytes) {
    DMN_REQUIRE(tr && bytes, DMN_ERR_INVALID, "dmn_train_workspace_bytes: null argument");
    StepIO io;
    memset(&io, 0, sizeof(io));
    Sizes q;
    int rc = run_step(tr, nullptr, nullptr, io, B, H, W, nullptr, 0, nullptr, true, &q);
    if (rc) return rc;
    *bytes = q.total;
    return DMN_OK;
}

extern "C" int dmn_train_loss_and_grads(DmnTrainer* tr, const float* params, float* grads, const float* x0, const float* noise,
                                        const int64_t* t, const float* sqrt_acp, const float* sqrt_1m_acp, int n_timesteps, int B,
                                        int H, int W, float* loss, float* model_out, void* workspace, size_t workspace_bytes, void* stream) {
    DMN_REQUIRE(tr && params && grads && x0 && noise && t && loss && workspace, DMN_ERR_INVALID,
                "dmn_train_loss_and_grads: null argument");
    DMN_REQUIRE((sqrt_acp == nullptr) == (sqrt_1m_acp == nullptr), DMN_ERR_INVALID, "schedule tables: both or neither");
    DMN_REQUIRE(sqrt_acp == nullptr || n_timesteps >= 1, DMN_ERR_INVALID, "n_timesteps=%d", n_timesteps);
    StepIO io;
    io.x0 = x0;
    io.noise = noise;
    io.t = reinterpret_cast<const long long*>(t);
    io.sa = sqrt_acp;
    io.sb = sqrt_1m_acp;
    io.n_timesteps = n_timesteps;
    io.loss = loss;
    io.model_out = model_out;
    return run_step(tr, params, grads, io, B, H, W, workspace, workspace_bytes, (hipStream_t)stream, false, nullptr);
}

extern "C" int dmn_adamw_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1,
                              float beta2, float eps, float weight_decay, int step, float max_norm, float grad_scale,
                              float* scratch, void* stream) {
    DMN_REQUIRE(params && grads && exp_avg && exp_avg_sq && scratch && n > 0 && step >= 1, DMN_ERR_INVALID, "dmn_adamw_step: bad argument");
    hipStream_t s = (hipStream_t)stream;
    AdamWArgs a;
    a.p = params;
    a.g = grads;
    a.m = exp_avg;
    a.v = exp_avg_sq;
    a.n = n;
    a.lr = lr;
    a.beta1 = beta1;
    a.beta2 = beta2;
    a.eps = eps;
    a.weight_decay = weight_decay;
    a.bias1 = (float)(1.0 - pow((double)beta1, (double)step));
    a.bias2 = (float)(1.0 - pow((double)beta2, (double)step));
    a.max_norm = max_norm;
    a.sq_parts = scratch;
    a.n_parts = 0;
    a.grad_scale = grad_scale;
    if (max_norm > 0.f) {
        int rc = launch_sqnorm(grads, n, scratch, &a.n_parts, s);
        if (rc) return rc;
    }
    return launch_adamw(a, s);
}
