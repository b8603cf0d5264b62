// Sampler entry points of the C-ABI (include/dmn.h): single-step updates, the graph-replayed sampling
// loops, and per-kernel test hooks.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>

#include <vector>

#include "../../include/dmn.h"
#include "dmn_kernels.h"
#include "dmn_sampler.h"
#include "dmn_unet_internal.h"

static StepCommon step_common(const float* table, const int* step_idx, int row, const float* x, const float* net,
                              const float* noise, float* x_out, int B, int C, int HW, uint64_t seed, int64_t img_offset,
                              float* keep) {
    StepCommon c;
    memset(&c, 0, sizeof(c));
    c.x = x;
    c.net = net;
    c.x_out = x_out;
    c.table = table;
    c.step_idx = step_idx;
    c.step = row;
    c.B = B;
    c.C = C;
    c.HW = HW;
    c.seed = seed;
    c.img_offset = img_offset;
    c.noise = noise;
    c.keep = keep;
    c.keep_stride = (long long)B * C * HW;
    return c;
}

extern "C" int dmn_ddpm_step(const float* table, int row, const float* x, const float* model_out, const float* noise,
                             float* x_out, int B, int C, int H, int W, int learned_variance, int pred_x0, uint64_t seed,
                             int64_t img_offset, void* stream) {
    DMN_REQUIRE(table && x && model_out && x_out && row >= 0, DMN_ERR_INVALID, "dmn_ddpm_step: null argument");
    DdpmStepArgs a;
    a.step_next = nullptr;
    a.c = step_common(table, nullptr, row, x, model_out, noise, x_out, B, C, H * W, seed, img_offset, nullptr);
    a.learned_variance = learned_variance;
    a.pred_x0 = pred_x0 & 1;
    a.ddim = (pred_x0 >> 1) & 1;
    return launch_ddpm_step(a, (hipStream_t)stream);
}

extern "C" int dmn_sde_predictor_step(const float* table, int row, const float* x, const float* model_out,
                                      const float* noise, float* x_out, float* x_mean, int B, int C, int H, int W,
                                      uint64_t seed, int64_t img_offset, void* stream) {
    DMN_REQUIRE(table && x && model_out && x_out && row >= 0, DMN_ERR_INVALID, "dmn_sde_predictor_step: null argument");
    AffineStepArgs a;
    a.c = step_common(table, nullptr, row, x, model_out, noise, x_out, B, C, H * W, seed, img_offset, nullptr);
    a.x_mean = x_mean;
    a.keep_mean = 0;
    return launch_affine_step(a, (hipStream_t)stream);
}

extern "C" size_t dmn_sde_scratch_bytes(int B, int C, int H, int W) {
    return (size_t)B * langevin_slots(C * H * W) * 2 * sizeof(float) + 256;
}

extern "C" int dmn_sde_corrector_step(const float* table, int row, const float* x, const float* model_out,
                                      const float* noise, float* x_out, int B, int C, int H, int W, float snr, int inner,
                                      int use_norms, uint64_t seed, int64_t img_offset, void* scratch, void* stream) {
    DMN_REQUIRE(table && x && model_out && x_out && scratch && row >= 0, DMN_ERR_INVALID,
                "dmn_sde_corrector_step: null argument");
    LangevinArgs a;
    memset(&a, 0, sizeof(a));
    a.c = step_common(table, nullptr, row, x, model_out, noise, x_out, B, C, H * W, seed, img_offset, nullptr);
    a.partials = (float*)scratch;
    a.inner = inner;
    a.use_norms = use_norms;
    a.snr = snr;
    return launch_langevin(a, (hipStream_t)stream);
}

extern "C" int dmn_fill_normal(float* x, int B, int per, uint64_t seed, uint32_t step, int64_t img_offset, float scale,
                               void* stream) {
    DMN_REQUIRE(x && B > 0 && per > 0, DMN_ERR_INVALID, "dmn_fill_normal: bad argument");
    return launch_prior(x, B, per, seed, step, img_offset, scale, (hipStream_t)stream);
}

// ---- sampling loops ------------------------------------------------------------------------------------
// Workspace layout: [0,256) control block (step counter), then net output, x_mean, Langevin partials, then the
// U-Net arena.
struct LoopBufs {
    int* step_idx;
    float* net;
    float* xmean;
    float* partials;
    void* arena;
    size_t arena_bytes;
};

static size_t align256(size_t n) { return (n + 255) & ~(size_t)255; }

// Langevin scratch of one launch chain of B images inside a sampling workspace: 256 bytes (the chain's step size), then
// either the plain partials of its images or -- in the first chain's region, when the batch runs as two chains -- the
// granules of BOTH chains' images, [2 buffers][2 B][slots][2] of 8 bytes (LangevinArgs::gran).
static size_t loop_granule_bytes(int B_all, int C, int H, int W) {
    return (size_t)2 * B_all * langevin_slots(C * H * W) * 2 * sizeof(unsigned long long);
}
static size_t loop_partials_bytes(int B, int C, int H, int W) { return 256 + loop_granule_bytes(2 * B, C, H, W); }

static int carve(DmnUnet* u, int B, int H, int W, void* ws, size_t ws_bytes, LoopBufs& lb, size_t* need) {
    const size_t img = (size_t)B * u->cfg.channels * H * W * 4;
    const size_t net = (size_t)B * u->cfg.out_dim * H * W * 4;
    size_t unet_bytes = 0;
    int rc = dmn_unet_workspace_bytes(u, B, H, W, &unet_bytes);
    if (rc) return rc;
    size_t off = 256;
    const size_t o_net = off;
    off += align256(net);
    const size_t o_mean = off;
    off += align256(img);
    const size_t o_part = off;
    off += align256(loop_partials_bytes(B, u->cfg.channels, H, W));
    const size_t o_arena = off;
    off += unet_bytes;
    if (need) *need = off;
    if (ws) {
        DMN_REQUIRE(ws_bytes >= off, DMN_ERR_WORKSPACE, "sampling workspace too small: %zu < %zu", ws_bytes, off);
        char* b = (char*)ws;
        lb.step_idx = (int*)b;
        lb.net = (float*)(b + o_net);
        lb.xmean = (float*)(b + o_mean);
        lb.partials = (float*)(b + o_part);
        lb.arena = b + o_arena;
        lb.arena_bytes = ws_bytes - o_arena;
    }
    return DMN_OK;
}

// Chains a DDPM loop over B images runs as: two half batches when the handle allows it and the halves are big enough
// to be worth a launch chain of their own.
static int ddpm_chains(const DmnUnet* u, int B) { return (u->sample_chains == 2 && B >= 16 && B % 2 == 0) ? 2 : 1; }

extern "C" int dmn_ddpm_loop_chains(const DmnUnet* u, int B) { return u ? ddpm_chains(u, B) : 0; }

extern "C" int dmn_unet_set_sample_chains(DmnUnet* u, int chains, int skew_us) {
    DMN_REQUIRE(u && (chains == 1 || chains == 2) && skew_us >= 0 && skew_us <= 5000, DMN_ERR_INVALID,
                "dmn_unet_set_sample_chains: chains must be 1 or 2, skew 0..5000 us");
    u->sample_chains = chains;
    u->sample_skew_us = skew_us;
    return DMN_OK;
}

extern "C" int dmn_sample_workspace_bytes(DmnUnet* u, int B, int H, int W, size_t* bytes) {
    DMN_REQUIRE(u && bytes, DMN_ERR_INVALID, "dmn_sample_workspace_bytes: null argument");
    LoopBufs lb;
    int rc = carve(u, B, H, W, nullptr, 0, lb, bytes);
    if (rc) return rc;
    if (ddpm_chains(u, B) == 2) {   // each chain has a control block, buffers and arena of its own
        size_t half = 0;
        if ((rc = carve(u, B / 2, H, W, nullptr, 0, lb, &half))) return rc;
        if (2 * align256(half) > *bytes) *bytes = 2 * align256(half);
    }
    return DMN_OK;
}

// Runs `body(chain, stream)` nsteps times per chain; when allowed, captures one step per chain into a graph (cached on
// the handle) and replays it. Two chains run on the handle's pair of chain streams, forked from and joined to the
// caller's stream with events, the second chain `sample_skew_us` behind the first.
template <typename Body>
static int run_loop_inner(DmnUnet* u, const SampleKey& key, int nsteps, hipStream_t s, int use_graph, int nchains, Body body);

// The handle's pair of chain streams. HIP spreads the streams of a process over a few hardware queues, and two streams
// on one queue run their kernels one behind the other: the chains would gain nothing, and chains that wait for each
// other on the device (the Langevin corrector's meetings) would wait for nothing. So the pair is probed when it is made
// (launch_probe_wait / launch_probe_set), the second stream is made again while the probe fails (the rejected ones are
// kept until the search ends, or the next one would land on the same queue), and the outcome is kept in
// chains_concurrent.
static int ensure_chain_streams(DmnUnet* u) {
    if (u->chain_stream[0]) return DMN_OK;
    hipStream_t a = nullptr, b = nullptr;
    std::vector<hipStream_t> rejected;
    unsigned* d = nullptr;
    bool ok = false;
    int rc = DMN_OK;
    auto fail = [&](int code) {
        for (hipStream_t r : rejected) (void)hipStreamDestroy(r);
        if (a) (void)hipStreamDestroy(a);
        if (b) (void)hipStreamDestroy(b);
        if (d) (void)hipFree(d);
        return code;
    };
    if (hipStreamCreateWithFlags(&a, hipStreamNonBlocking) != hipSuccess || hipMalloc(&d, 2 * sizeof(unsigned)) != hipSuccess) {
        dmn_set_error("sampling loop: cannot create the chain streams");
        return fail(DMN_ERR_HIP);
    }
    for (int attempt = 0; attempt < 8 && !ok; ++attempt) {
        hipStream_t c = nullptr;
        unsigned seen = 0;
        if (hipStreamCreateWithFlags(&c, hipStreamNonBlocking) != hipSuccess || hipMemset(d, 0, 2 * sizeof(unsigned)) != hipSuccess ||
            hipDeviceSynchronize() != hipSuccess) {
            if (c) rejected.push_back(c);
            dmn_set_error("sampling loop: cannot create the chain streams");
            return fail(DMN_ERR_HIP);
        }
        if ((rc = launch_probe_wait(d, d + 1, 20000, a)) || (rc = launch_probe_set(d, c))) {
            rejected.push_back(c);
            return fail(rc);
        }
        if (hipStreamSynchronize(a) != hipSuccess || hipStreamSynchronize(c) != hipSuccess ||
            hipMemcpy(&seen, d + 1, sizeof(seen), hipMemcpyDeviceToHost) != hipSuccess) {
            rejected.push_back(c);
            dmn_set_error("sampling loop: probe of the chain streams failed");
            return fail(DMN_ERR_HIP);
        }
        if (seen) {
            b = c;
            ok = true;
        } else {
            rejected.push_back(c);
        }
    }
    if (!ok) {   // no pair runs side by side (one hardware queue?): two chains still give the right bits, one behind the other
        b = rejected.back();
        rejected.pop_back();
    }
    for (hipStream_t r : rejected) (void)hipStreamDestroy(r);
    rejected.clear();
    (void)hipFree(d);
    d = nullptr;
    if (hipEventCreateWithFlags(&u->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&u->ev_join, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&u->ev_join0, hipEventDisableTiming) != hipSuccess) {
        dmn_set_error("sampling loop: cannot create the fork / join events");
        return fail(DMN_ERR_HIP);
    }
    u->chain_stream[0] = a;
    u->chain_stream[1] = b;
    u->chains_concurrent = ok;
    return DMN_OK;
}

extern "C" int dmn_unet_chains_concurrent(DmnUnet* u, int* concurrent) {
    DMN_REQUIRE(u && concurrent, DMN_ERR_INVALID, "dmn_unet_chains_concurrent: null argument");
    int rc = ensure_chain_streams(u);
    if (rc) return rc;
    *concurrent = u->chains_concurrent ? 1 : 0;
    return DMN_OK;
}

// `evals`: network evaluations one step of a chain makes (epoch housekeeping of the in-launch exchanges).
template <typename Body>
static int run_loop(DmnUnet* u, const SampleKey& key_in, int nsteps, int evals, hipStream_t s, int use_graph, int nchains, Body body) {
    if (!u->ev_start) {
        DMN_HIP_CHECK(hipEventCreate(&u->ev_start));
        DMN_HIP_CHECK(hipEventCreate(&u->ev_stop));
    }
    u->ev_valid = false;
    // The in-launch exchanges are only safe while the exchanging launches on the GPU are those of ONE handle's chains
    // (dmn_handoff.h): a loop that starts while another handle's exchanging work is in flight closes every block with
    // the separate passes instead (results agree to summation order). Its captured step differs, hence the extra key bit.
    const bool want_exchange = !(u->path_mask & 128);
    const bool entered = want_exchange && exchange_guard_enter(u);
    u->exchange_off = want_exchange && !entered;
    SampleKey key = key_in;
    if (u->exchange_off) key.kind |= 1 << 16;
    int rc = want_exchange ? unet_epoch_reserve(u, (unsigned long long)nsteps * (unsigned long long)evals, s) : DMN_OK;
    if (rc == DMN_OK) rc = unet_time_hidden_table(u, static_cast<const float*>(key.table), nsteps, s);
    key.h2 = u->h2_table_of ? u->h2_table : nullptr;
    if (rc == DMN_OK && hipEventRecord(u->ev_start, s) != hipSuccess) {
        dmn_set_error("hipEventRecord failed at the start of a sampling loop");
        rc = DMN_ERR_HIP;
    }
    if (rc == DMN_OK) rc = run_loop_inner(u, key, nsteps, s, use_graph, nchains, body);
    u->exchange_off = false;
    if (want_exchange) exchange_guard_leave(u, s, entered);   // behind the join: both chains have been enqueued on or joined to `s`
    if (rc) return rc;
    DMN_HIP_CHECK(hipEventRecord(u->ev_stop, s));
    u->ev_valid = true;
    return DMN_OK;
}

template <typename Body>
static int run_loop_inner(DmnUnet* u, const SampleKey& key_in, int nsteps, hipStream_t s, int use_graph, int nchains, Body body) {
    int rc;
    // steps per captured graph: a divisor of both the loop length and the pacing block
    int K = 1;
    if (use_graph && s != nullptr && u->loop_graph_steps > 1) {
        const int blk = u->loop_pace > 0 ? u->loop_pace : 8;
        for (int k = 1; k <= u->loop_graph_steps && k <= blk; ++k)
            if (nsteps % k == 0 && blk % k == 0) K = k;
    }
    SampleKey key = key_in;
    key.kind |= K << 20;
    hipStream_t cs[2] = {s, s};
    if (nchains == 2) {
        if ((rc = ensure_chain_streams(u))) return rc;
        cs[0] = u->chain_stream[0];
        cs[1] = u->chain_stream[1];
    }
    const bool graph = use_graph && s != nullptr;
    if (graph && !(u->graph_exec && u->graph_key == key)) {
        hipGraphExec_t* slot[2] = {&u->graph_exec, &u->graph_exec2};
        for (int c = 0; c < 2; ++c)
            if (*slot[c]) {
                (void)hipGraphExecDestroy(*slot[c]);
                *slot[c] = nullptr;
            }
        for (int c = 0; c < nchains; ++c) {   // both steps are captured on the caller's stream; a graph runs on any stream
            hipGraph_t g = nullptr;
            DMN_HIP_CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed));
            rc = DMN_OK;
            for (int k = 0; k < K && rc == DMN_OK; ++k) rc = body(c, s);
            hipError_t e = hipStreamEndCapture(s, &g);
            if (rc) {
                if (g) (void)hipGraphDestroy(g);
                return rc;
            }
            DMN_HIP_CHECK(e);
            DMN_HIP_CHECK(hipGraphInstantiate(slot[c], g, nullptr, nullptr, 0));
            (void)hipGraphDestroy(g);
        }
        u->graph_key = key;
        if (getenv("DMN_LOOP_HOST_TRACE")) fprintf(stderr, "loop: step graphs captured (x %p seed %llu)\n", (void*)key.x, (unsigned long long)key.seed);
    }
    if (nchains == 2) {   // the chain streams start behind everything queued on the caller's stream so far
        DMN_HIP_CHECK(hipEventRecord(u->ev_fork, s));
        DMN_HIP_CHECK(hipStreamWaitEvent(cs[0], u->ev_fork, 0));
        DMN_HIP_CHECK(hipStreamWaitEvent(cs[1], u->ev_fork, 0));
        if (u->sample_skew_us > 0 && (rc = launch_delay(u->sample_skew_us, cs[1]))) return rc;
    }
    // tuning aid (DMN_LOOP_HOST_TRACE): host time of the enqueue loop per block of 50 steps, to stderr
    static const bool host_trace = getenv("DMN_LOOP_HOST_TRACE") != nullptr;
    std::chrono::steady_clock::time_point t_block = std::chrono::steady_clock::now();
    // Pacing: the host enqueues a step in a seventh of the time the device needs for it, and while it runs hundreds of
    // steps ahead (until the hardware queue is full, about 300 steps) the device is measurably slower: 787 against 727 us
    // per step over the first 300 steps of a loop on MI355X, and a profiler's kernel trace of the same loop shows no
    // such trend in the kernels themselves -- the cost follows the host's burst of packet and kernel-argument writes.
    // So the host waits, every `pace` steps, until the block before the last one has finished: at most two blocks are
    // ever in flight (16 steps by default: about 11 ms ahead of the device; fewer steps in flight measure faster).
    const int pace = (s != nullptr && u->loop_pace > 0) ? u->loop_pace : 0;
    const int depth = u->loop_depth;
    if (pace)
        for (int k = 0; k < depth; ++k)
            for (int c = 0; c < nchains; ++c)
                if (!u->ev_pace[k][c]) DMN_HIP_CHECK(hipEventCreateWithFlags(&u->ev_pace[k][c], hipEventDisableTiming));
    for (int i = 0; i < nsteps; i += K) {
        if (pace && i % pace == 0 && i >= depth * pace) {
            const int slot = (i / pace) % depth;   // recorded `depth` blocks ago
            for (int c = 0; c < nchains; ++c) DMN_HIP_CHECK(hipEventSynchronize(u->ev_pace[slot][c]));
        }
        for (int c = 0; c < nchains; ++c) {
            if (graph) DMN_HIP_CHECK(hipGraphLaunch(c ? u->graph_exec2 : u->graph_exec, cs[c]));
            else if ((rc = body(c, cs[c]))) return rc;
        }
        if (pace && (i + K) % pace == 0) {
            const int slot = (i / pace) % depth;
            for (int c = 0; c < nchains; ++c) DMN_HIP_CHECK(hipEventRecord(u->ev_pace[slot][c], cs[c]));
            // Two chains run fastest in LOCKSTEP -- the same kernels side by side -- and slowest half a step apart
            // (733 against 750 us per step, tools/chain_phase.py); left alone they drift apart in about every second
            // loop. So each chain also waits, on the device, for the other one's end of the block: they start every
            // block together (the drift inside a block of 8-16 steps is a few hundredths of a step).
            // (at every loop_lockstep_every-th pacing point: the wait costs about as much as a pacing point itself, and
            // the chains need tens of steps to drift apart measurably)
            if (nchains == 2 && u->loop_lockstep && i + K < nsteps && (i / pace) % u->loop_lockstep_every == u->loop_lockstep_every - 1) {
                DMN_HIP_CHECK(hipStreamWaitEvent(cs[0], u->ev_pace[slot][1], 0));
                DMN_HIP_CHECK(hipStreamWaitEvent(cs[1], u->ev_pace[slot][0], 0));
                if (u->sample_skew_us > 0 && (rc = launch_delay(u->sample_skew_us, cs[1]))) return rc;   // tuning: a fixed lag
            }
        }
        if (K == 1 && host_trace && (i % 50 == 49 || i == nsteps - 1)) {
            const auto now = std::chrono::steady_clock::now();
            fprintf(stderr, "%s%.0f", i < 50 ? "loop host us per step:" : "", std::chrono::duration<double, std::micro>(now - t_block).count() / (i % 50 + 1));
            fprintf(stderr, i == nsteps - 1 ? "\n" : " ");
            t_block = now;
        }
    }
    if (nchains == 2) {   // and the caller's stream goes on behind both
        DMN_HIP_CHECK(hipEventRecord(u->ev_join0, cs[0]));
        DMN_HIP_CHECK(hipStreamWaitEvent(s, u->ev_join0, 0));
        DMN_HIP_CHECK(hipEventRecord(u->ev_join, cs[1]));
        DMN_HIP_CHECK(hipStreamWaitEvent(s, u->ev_join, 0));
    }
    return DMN_OK;
}

extern "C" int dmn_ddpm_sample_loop(DmnUnet* u, const float* table, int nsteps, float* x, int B, int H, int W,
                                    int learned_variance, int pred_x0, int draw_prior, uint64_t seed, int64_t img_offset,
                                    float* keep, void* workspace, size_t workspace_bytes, void* stream, int use_graph) {
    DMN_REQUIRE(u && table && x && workspace && nsteps > 0, DMN_ERR_INVALID, "dmn_ddpm_sample_loop: bad argument");
    DMN_REQUIRE(!((pred_x0 & 2) && learned_variance), DMN_ERR_INVALID, "the DDIM update has no learned-variance form");
    DMN_REQUIRE(u->cfg.out_dim == u->cfg.channels * (learned_variance ? 2 : 1), DMN_ERR_INVALID,
                "network emits %d channels, sampler expects %d", u->cfg.out_dim, u->cfg.channels * (learned_variance ? 2 : 1));
    hipStream_t s = (hipStream_t)stream;
    const int C = u->cfg.channels;
    const int nchains = s ? ddpm_chains(u, B) : 1;   // the legacy default stream synchronises with every other stream
    const int Bc = B / nchains;
    LoopBufs lb[2];
    int rc;
    size_t per_chain = 0;
    if (nchains == 2) {
        if ((rc = carve(u, Bc, H, W, nullptr, 0, lb[0], &per_chain))) return rc;
        per_chain = align256(per_chain);
        DMN_REQUIRE(workspace_bytes >= 2 * per_chain, DMN_ERR_WORKSPACE, "sampling workspace too small: %zu < %zu",
                    workspace_bytes, 2 * per_chain);
    } else {
        per_chain = workspace_bytes;
    }
    for (int c = 0; c < nchains; ++c) {
        if ((rc = carve(u, Bc, H, W, (char*)workspace + c * per_chain, per_chain, lb[c], nullptr))) return rc;
        DMN_HIP_CHECK(hipMemsetAsync(lb[c].step_idx, 0, 256, s));
    }
    if (draw_prior && (rc = launch_prior(x, B, C * H * W, seed, (unsigned)nsteps, img_offset, 1.0f, s))) return rc;
    SampleKey key = {x, table, workspace, keep, nullptr, 1 | (nchains << 8), B, H, W, learned_variance | (pred_x0 << 1), nsteps, seed,
                     img_offset, 0.f, u->labels};
    const size_t img = (size_t)C * H * W;
    auto body = [&](int c, hipStream_t cs) -> int {
        const int b0 = c * Bc;   // first image of the chain
        float* xc = x + b0 * img;
        // The update reads the copy of the counter that the network's first kernel made and writes the next value
        // to the counter itself: no launch of its own for the increment.
        // (A network without time embedding has no kernel that makes the copy: separate increment then.)
        const bool handoff = u->cfg.with_time_emb != 0;
        DdpmStepArgs a;
        a.c = step_common(table, handoff ? lb[c].step_idx + kStepCopyOffset : lb[c].step_idx, 0, xc, lb[c].net, nullptr, xc, Bc, C,
                          H * W, seed, img_offset + b0, keep ? keep + b0 * img : nullptr);
        a.c.keep_stride = (long long)B * C * H * W;   // kept images of all chains share one [n_kept][B] buffer
        a.step_next = handoff ? lb[c].step_idx : nullptr;
        a.learned_variance = learned_variance;
        a.pred_x0 = pred_x0 & 1;
        a.ddim = (pred_x0 >> 1) & 1;
        // the network's last kernel applies the update itself (same arithmetic and noise: same bits) where the pixel count
        // allows its quads of four
        const bool fuse = handoff && (H * W) % 4 == 0 && !(u->path_mask & 2048);
        u->label_offset = b0;
        u->chain_id = c;
        int r = unet_forward_impl(u, xc, nullptr, 1, table, lb[c].step_idx, lb[c].net, Bc, H, W, lb[c].arena, lb[c].arena_bytes, cs,
                                  false, nullptr, nullptr, fuse ? &a : nullptr);
        u->label_offset = 0;
        u->chain_id = 0;
        if (r) return r;
        if (fuse) return DMN_OK;
        if ((r = launch_ddpm_step(a, cs))) return r;
        return handoff ? DMN_OK : launch_advance(lb[c].step_idx, cs);
    };
    return run_loop(u, key, nsteps, 1, s, use_graph, nchains, body);
}

extern "C" int dmn_pc_sample_loop(DmnUnet* u, const float* table, int nsteps, float* x, int B, int H, int W, int predictor,
                                  int corrector, int n_corrector_steps, float snr, int denoise, int draw_prior,
                                  float prior_scale, uint64_t seed, int64_t img_offset, float* keep, float* x_mean_out,
                                  void* workspace, size_t workspace_bytes, void* stream, int use_graph) {
    DMN_REQUIRE(u && table && x && workspace && nsteps > 0, DMN_ERR_INVALID, "dmn_pc_sample_loop: bad argument");
    DMN_REQUIRE(u->cfg.out_dim == u->cfg.channels, DMN_ERR_INVALID, "score network must emit %d channels", u->cfg.channels);
    DMN_REQUIRE(corrector >= 0 && corrector <= 2 && predictor >= 0 && predictor <= 1 && n_corrector_steps >= 0 &&
                    n_corrector_steps <= 64, DMN_ERR_INVALID, "dmn_pc_sample_loop: bad predictor/corrector selection");
    hipStream_t s = (hipStream_t)stream;
    const int C = u->cfg.channels;
    // Two half-batch launch chains like the DDPM loop, with the bits of one chain. Only the Langevin corrector couples
    // the images of a batch (its step size is a batch mean; the predictors and the fixed-step annealed Langevin
    // dynamics are per element): there the chains meet once per corrector step and exchange their images' norms on the
    // device (LangevinArgs::gran). The waits of those meetings are bounded and counted like the in-launch exchanges',
    // in the handle's counter: without one (exchanges disabled) the Langevin corrector keeps to one chain.
    int nchains = s ? ddpm_chains(u, B) : 1;
    if (corrector == 1 && n_corrector_steps > 0 && nchains == 2) {
        size_t unused = 0;
        int rcw = dmn_unet_workspace_bytes(u, B / 2, H, W, &unused);   // reserves the handle's exchange areas
        if (rcw) return rcw;
        if ((rcw = ensure_chain_streams(u))) return rcw;
        // chains that cannot run side by side would wait for each other in vain at every meeting
        if (!u->handoff_mem[0] || !u->chains_concurrent) nchains = 1;
    }
    const bool meet = corrector == 1 && nchains == 2;
    const int Bc = B / nchains;
    LoopBufs lb[2];
    int rc;
    size_t per_chain = 0;
    if (nchains == 2) {
        if ((rc = carve(u, Bc, H, W, nullptr, 0, lb[0], &per_chain))) return rc;
        per_chain = align256(per_chain);
        DMN_REQUIRE(workspace_bytes >= 2 * per_chain, DMN_ERR_WORKSPACE, "sampling workspace too small: %zu < %zu",
                    workspace_bytes, 2 * per_chain);
    } else {
        per_chain = workspace_bytes;
    }
    for (int c = 0; c < nchains; ++c) {
        if ((rc = carve(u, Bc, H, W, (char*)workspace + c * per_chain, per_chain, lb[c], nullptr))) return rc;
        DMN_HIP_CHECK(hipMemsetAsync(lb[c].step_idx, 0, 256, s));
    }
    unsigned long long* gran = nullptr;
    if (meet) {   // tags of an earlier loop must not survive: the area starts every loop zeroed
        gran = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(lb[0].partials) + 256);
        DMN_HIP_CHECK(hipMemsetAsync(gran, 0, loop_granule_bytes(B, C, H, W), s));
    }
    if (draw_prior && (rc = launch_prior(x, B, C * H * W, seed, (unsigned)nsteps, img_offset, prior_scale, s))) return rc;
    SampleKey key = {x, table, workspace, keep, x_mean_out, 2 | (nchains << 8), B, H, W,
                     predictor | (corrector << 2) | (n_corrector_steps << 4) | (denoise << 12), nsteps, seed, img_offset, snr, u->labels};
    const size_t img = (size_t)C * H * W;
    auto body = [&](int c, hipStream_t cs) -> int {
        const int b0 = c * Bc;   // first image of the chain
        float* xc = x + b0 * img;
        float* keepc = keep ? keep + b0 * img : nullptr;
        float* xmean = x_mean_out ? x_mean_out + b0 * img : lb[c].xmean;
        const long long keep_stride = (long long)B * C * H * W;   // kept images of all chains share one [n_kept][B] buffer
        int r;
        u->label_offset = b0;
        u->chain_id = c;
        auto forward = [&]() {
            return unet_forward_impl(u, xc, nullptr, 1, table, lb[c].step_idx, lb[c].net, Bc, H, W, lb[c].arena, lb[c].arena_bytes, cs,
                                     false, nullptr);
        };
        auto done = [&](int code) {
            u->label_offset = 0;
            u->chain_id = 0;
            return code;
        };
        if (corrector) {
            for (int k = 0; k < n_corrector_steps; ++k) {
                if ((r = forward())) return done(r);
                LangevinArgs a;
                memset(&a, 0, sizeof(a));
                // without a predictor the image kept for an iteration is the corrector's x (NonePredictor returns
                // (x, x), predictor_corrector_sampler.py:105-113): the last inner step writes it
                a.c = step_common(table, lb[c].step_idx, 0, xc, lb[c].net, nullptr, xc, Bc, C, H * W, seed, img_offset + b0,
                                  (!predictor && k == n_corrector_steps - 1) ? keepc : nullptr);
                a.c.keep_stride = keep_stride;
                a.partials = lb[c].partials + 64;   // behind the 256 bytes of the chain's step size
                a.inner = k;
                a.use_norms = corrector == 1;
                a.snr = snr;
                if (meet) {
                    a.gran = gran;
                    a.step_io = lb[c].partials;
                    a.giveups = reinterpret_cast<unsigned*>(u->handoff_mem[0]) + 1;
                    a.B_all = B;
                    a.b_first = b0;
                    a.n_inner = n_corrector_steps;
                }
                if ((r = launch_langevin(a, cs))) return done(r);
            }
        }
        if (predictor) {
            if ((r = forward())) return done(r);
            AffineStepArgs a;
            a.c = step_common(table, lb[c].step_idx, 0, xc, lb[c].net, nullptr, xc, Bc, C, H * W, seed, img_offset + b0, keepc);
            a.c.keep_stride = keep_stride;
            a.x_mean = xmean;
            a.keep_mean = denoise;
            if ((r = launch_affine_step(a, cs))) return done(r);
        }
        return done(launch_advance(lb[c].step_idx, cs));
    };
    return run_loop(u, key, nsteps, n_corrector_steps + 1, s, use_graph, nchains, body);
}

extern "C" int dmn_bpd_loop(DmnUnet* u, const float* table, int nsteps, const float* x0, int B, int H, int W,
                            int learned_variance, uint64_t seed, int64_t img_offset, float* terms, int terms_stride,
                            float* prior, float prior_sqrt_acp, float prior_log_1m_acp, void* workspace,
                            size_t workspace_bytes, void* stream, int use_graph) {
    DMN_REQUIRE(u && table && x0 && terms && prior && workspace && nsteps > 0 && terms_stride >= nsteps, DMN_ERR_INVALID,
                "dmn_bpd_loop: bad argument");
    DMN_REQUIRE(u->cfg.out_dim == u->cfg.channels * (learned_variance ? 2 : 1), DMN_ERR_INVALID,
                "network emits %d channels, evaluation expects %d", u->cfg.out_dim, u->cfg.channels * (learned_variance ? 2 : 1));
    hipStream_t s = (hipStream_t)stream;
    const int C = u->cfg.channels;
    // every term is per image: two half-batch launch chains like the DDPM loop
    const int nchains = s ? ddpm_chains(u, B) : 1;
    const int Bc = B / nchains;
    LoopBufs lb[2];
    int rc;
    size_t per_chain = 0;
    if (nchains == 2) {
        if ((rc = carve(u, Bc, H, W, nullptr, 0, lb[0], &per_chain))) return rc;
        per_chain = align256(per_chain);
        DMN_REQUIRE(workspace_bytes >= 2 * per_chain, DMN_ERR_WORKSPACE, "sampling workspace too small: %zu < %zu",
                    workspace_bytes, 2 * per_chain);
    } else {
        per_chain = workspace_bytes;
    }
    for (int c = 0; c < nchains; ++c) {
        if ((rc = carve(u, Bc, H, W, (char*)workspace + c * per_chain, per_chain, lb[c], nullptr))) return rc;
        DMN_HIP_CHECK(hipMemsetAsync(lb[c].step_idx, 0, 256, s));
    }
    DMN_REQUIRE((size_t)Bc * bpd_slots(C * H * W) * 4 <= dmn_sde_scratch_bytes(Bc, C, H, W), DMN_ERR_WORKSPACE,
                "dmn_bpd_loop: scratch too small");
    if ((rc = launch_bpd_prior(x0, prior, B, C * H * W, prior_sqrt_acp, prior_log_1m_acp, s))) return rc;
    SampleKey key = {x0, table, workspace, terms, prior, 3 | (nchains << 8), B, H, W, learned_variance, nsteps, seed, img_offset,
                     (float)terms_stride, u->labels};
    const size_t img = (size_t)C * H * W;
    auto body = [&](int c, hipStream_t cs) -> int {
        const int b0 = c * Bc;   // first image of the chain
        BpdArgs a;
        memset(&a, 0, sizeof(a));
        a.c = step_common(table, lb[c].step_idx, 0, x0 + b0 * img, lb[c].net, nullptr, lb[c].xmean, Bc, C, H * W, seed,
                          img_offset + b0, nullptr);
        a.partials = lb[c].partials;
        a.terms = terms + (size_t)b0 * terms_stride;
        a.terms_stride = terms_stride;
        a.learned_variance = learned_variance;
        int r;
        if ((r = launch_bpd_qsample(a, cs))) return r;
        u->label_offset = b0;
        u->chain_id = c;
        r = unet_forward_impl(u, lb[c].xmean, nullptr, 1, table, lb[c].step_idx, lb[c].net, Bc, H, W, lb[c].arena, lb[c].arena_bytes,
                              cs, false, nullptr);
        u->label_offset = 0;
        u->chain_id = 0;
        if (r) return r;
        if ((r = launch_bpd_term(a, cs))) return r;
        return launch_advance(lb[c].step_idx, cs);
    };
    return run_loop(u, key, nsteps, 1, s, use_graph, nchains, body);
}

extern "C" int dmn_loop_last_ms(DmnUnet* u, float* ms) {
    DMN_REQUIRE(u && ms, DMN_ERR_INVALID, "dmn_loop_last_ms: null argument");
    DMN_REQUIRE(u->ev_valid, DMN_ERR_INVALID, "dmn_loop_last_ms: no sampling loop has run on this handle");
    DMN_HIP_CHECK(hipEventSynchronize(u->ev_stop));
    DMN_HIP_CHECK(hipEventElapsedTime(ms, u->ev_start, u->ev_stop));
    return DMN_OK;
}

// ---- per-kernel test hook ----------------------------------------------------------------------------------
extern "C" int dmn_test_set_conv_debug(long long* dev) {
    conv_set_debug_buffer(dev);
    return DMN_OK;
}

extern "C" int dmn_test_conv2d(int dtype, const void* in0, const void* in1, int C0, int C1, int B, int H, int W,
                               const float* w_host, const float* bias_host, int Cout, int kh, int kw, int stride, int pad,
                               int transposed, void* out, const void* res, const float* pro_gamma_host,
                               const float* pro_beta_host, const float* pro_tbias_dev, const float* pro_stats_dev, int pro_G,
                               int pro_slots, int pro_silu, float* epi_stats_dev, int epi_G, int* epi_slots_out,
                               int cfg_tile, int cfg_bn, void* stream) {
    int rc = conv_init_attributes();
    if (rc) return rc;
    PackedConv pc;
    if ((rc = pack_conv(pc, dtype, w_host, bias_host, Cout, C0 + C1, kh, kw, stride, pad, transposed != 0))) return rc;
    ConvArgs a;
    conv_args_from(pc, dtype, a, B, H, W);
    a.in0 = in0;
    a.in1 = in1;
    a.C0 = C0;
    a.C1 = C1;
    a.out = out;
    a.res = res;
    a.cfg_tile = cfg_tile;
    a.cfg_bn = cfg_bn;
    float *dg = nullptr, *db = nullptr;
    if (pro_stats_dev) {
        const int Cin = C0 + C1;
        if (hipMalloc((void**)&dg, Cin * 4) != hipSuccess || hipMalloc((void**)&db, Cin * 4) != hipSuccess) return DMN_ERR_HIP;
        (void)hipMemcpy(dg, pro_gamma_host, Cin * 4, hipMemcpyHostToDevice);
        (void)hipMemcpy(db, pro_beta_host, Cin * 4, hipMemcpyHostToDevice);
        a.pro.stats = pro_stats_dev;
        a.pro.G = pro_G;
        a.pro.slots = pro_slots;
        a.pro.inv_count = 1.0f / ((float)(Cin / pro_G) * H * W);
        a.pro.gamma = dg;
        a.pro.beta = db;
        a.pro.tbias = pro_tbias_dev;
        a.pro.tb_stride = Cin;
        a.pro.silu = pro_silu;
    }
    if (epi_stats_dev) {
        a.epi.stats = epi_stats_dev;
        a.epi.G = epi_G;
    }
    rc = launch_conv(a, (hipStream_t)stream);
    if (epi_slots_out) *epi_slots_out = a.epi.slots;
    (void)hipStreamSynchronize((hipStream_t)stream);
    free_conv(pc);
    if (dg) (void)hipFree(dg);
    if (db) (void)hipFree(db);
    return rc;
}

// block2.proj of a ResnetBlock together with the block's closing SiLU(GroupNorm(h)) + res on caller-provided tensors
// (unit parity tests, tuning): fused = 1 runs the one-launch form with the in-launch statistics hand-off, fused = 0 the
// convolution followed by the stand-alone closing pass (h2 [B][H][W][Cout] is its intermediate); + 2 (either form):
// full-resolution images take the 32x16-pixel tiles that bring them under the hand-off's limit, as the network does. out_stats_dev receives
// [B][out_G][slots][2] partial statistics of `out` (slots returned in out_slots). The launch may stamp phase times
// into the buffer set by dmn_test_set_conv_debug.
extern "C" int dmn_test_conv_block_tail(int dtype, const void* in, int Cin, int B, int H, int W, const float* w_host,
                                        const float* bias_host, int Cout, const float* pro_gamma_host,
                                        const float* pro_beta_host, const float* pro_stats_dev, int pro_G, int pro_slots,
                                        const float* tail_gamma_host, const float* tail_beta_host, int G, const void* res,
                                        void* h2, void* out, float* out_stats_dev, int out_G, int* out_slots, int fused,
                                        int plan_B, int cfg_tile, int cfg_bn, unsigned* giveups, void* stream) {
    DMN_REQUIRE(in && w_host && pro_stats_dev && tail_gamma_host && tail_beta_host && res && out && ((fused & 1) || h2), DMN_ERR_INVALID,
                "dmn_test_conv_block_tail: null argument");
    int rc = conv_init_attributes();
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    PackedConv pc;
    if ((rc = pack_conv(pc, dtype, w_host, bias_host, Cout, Cin, 3, 3, 1, 1, false))) return rc;
    float *dpg = nullptr, *dpb = nullptr, *dtg = nullptr, *dtb = nullptr, *h2_stats = nullptr;
    void* area = nullptr;
    auto cleanup = [&]() {
        (void)hipStreamSynchronize(s);
        free_conv(pc);
        for (void* p : {(void*)dpg, (void*)dpb, (void*)dtg, (void*)dtb, (void*)h2_stats, area})
            if (p) (void)hipFree(p);
    };
    auto up = [&](const float* h, int n, float** d) {
        return hipMalloc((void**)d, n * 4) == hipSuccess && hipMemcpy(*d, h, n * 4, hipMemcpyHostToDevice) == hipSuccess;
    };
    if (!up(pro_gamma_host, Cin, &dpg) || !up(pro_beta_host, Cin, &dpb) || !up(tail_gamma_host, Cout, &dtg) ||
        !up(tail_beta_host, Cout, &dtb)) {
        cleanup();
        dmn_set_error("dmn_test_conv_block_tail: device allocation failed");
        return DMN_ERR_HIP;
    }
    ConvArgs a;
    conv_args_from(pc, dtype, a, B, H, W);
    a.in0 = in;
    a.C0 = Cin;
    a.plan_B = plan_B;
    a.cfg_tile = cfg_tile;
    a.cfg_bn = cfg_bn;
    a.tall_tail = (fused & 2) != 0;
    a.pro.stats = pro_stats_dev;
    a.pro.G = pro_G;
    a.pro.slots = pro_slots;
    a.pro.inv_count = 1.0f / ((float)(Cin / pro_G) * H * W);
    a.pro.gamma = dpg;
    a.pro.beta = dpb;
    a.pro.silu = 1;
    a.epi.G = G;
    const float inv_count = 1.0f / ((float)(Cout / G) * H * W);
    if (giveups) *giveups = 0;
    if (fused & 1) {
        const size_t bytes = kHandoffHeaderBytes + (size_t)B * G * conv_epi_slots(a) * kHandoffGranulesPerSlot * sizeof(unsigned long long);
        unsigned hdr[2] = {1u, 0u};   // epoch 1 over a zeroed area
        if (!conv_tail_fits(a) || hipMalloc(&area, bytes) != hipSuccess || hipMemset(area, 0, bytes) != hipSuccess ||
            hipMemcpy(area, hdr, sizeof(hdr), hipMemcpyHostToDevice) != hipSuccess) {
            cleanup();
            dmn_set_error("dmn_test_conv_block_tail: shape outside the hand-off's limits, or allocation failed");
            return DMN_ERR_LIMIT;
        }
        a.out = out;
        a.res = res;
        a.tail.gran = reinterpret_cast<unsigned long long*>((char*)area + kHandoffHeaderBytes);
        a.tail.epoch = reinterpret_cast<const unsigned*>(area);
        a.tail.giveups = reinterpret_cast<unsigned*>(area) + 1;
        a.tail_gamma = dtg;
        a.tail_beta = dtb;
        a.tail_inv_count = inv_count;
        if (out_stats_dev) {
            a.tail_stats.stats = out_stats_dev;
            a.tail_stats.G = out_G;
        }
        rc = launch_conv(a, s);
        if (out_slots) *out_slots = a.tail_stats.slots;
        (void)hipStreamSynchronize(s);
        if (giveups) (void)hipMemcpy(giveups, (char*)area + 4, 4, hipMemcpyDeviceToHost);
    } else {
        const int slots = conv_epi_slots(a);
        if (hipMalloc((void**)&h2_stats, (size_t)B * G * slots * 4 * sizeof(float)) != hipSuccess) {
            cleanup();
            return DMN_ERR_HIP;
        }
        a.out = h2;
        a.epi.stats = h2_stats;
        rc = launch_conv(a, s);
        if (rc == DMN_OK) {
            NormApplyArgs na;
            memset(&na, 0, sizeof(na));
            na.dtype = dtype;
            na.in = h2;
            na.out = out;
            na.res = res;
            na.B = B;
            na.HW = H * W;
            na.C = Cout;
            na.norm.stats = h2_stats;
            na.norm.G = G;
            na.norm.slots = a.epi.slots;
            na.norm.inv_count = inv_count;
            na.norm.gamma = dtg;
            na.norm.beta = dtb;
            na.norm.silu = 1;
            na.plan_B = plan_B;
            if (out_stats_dev) {
                na.epi.stats = out_stats_dev;
                na.epi.G = out_G;
            }
            rc = launch_norm_apply(na, s);
            if (out_slots) *out_slots = na.epi.slots;
        }
    }
    cleanup();
    return rc;
}

// Softmax attention core on caller-provided tensors (unit parity tests).
extern "C" int dmn_test_attention(int dtype, const void* qkv, void* out, int B, int n, int heads, void* stream) {
    DMN_REQUIRE(qkv && out && B > 0 && n > 0 && heads > 0, DMN_ERR_INVALID, "dmn_test_attention: bad argument");
    AttnArgs a;
    memset(&a, 0, sizeof(a));
    a.dtype = dtype;
    a.qkv = qkv;
    a.out = out;
    a.B = B;
    a.n = n;
    a.heads = heads;
    a.d = 32;
    return launch_attention(a, (hipStream_t)stream);
}
