// Private layout of the DmnUnet handle (weights packed for the gfx950 kernels).
#pragma once
#include <string>
#include <vector>

#include "../../include/dmn.h"
#include "dmn_kernels.h"

struct PackedConv {
    void* w = nullptr;       // device, packed [phase][Cout/16][Cin/kch][ntaps][64][16 B] in the network dtype
    float* bias = nullptr;   // device fp32 [Cout] or nullptr
    int Cin = 0, Cout = 0, kind = 0, stride = 1, pad = 0, ntaps = 0, ext = 0, nphase = 1;
    bool transposed = false;
    ConvPhase phase[4] = {};
};

struct ResBlockW {
    PackedConv c1, c2, res;
    float *g1 = nullptr, *b1 = nullptr, *g2 = nullptr, *b2 = nullptr;
    bool has_res = false;
    int tb_off = -1;         // offset of this block's time bias inside the per-sample bias vector
};

struct AttnW {
    float *ng = nullptr, *nb = nullptr;   // PreNorm GroupNorm(1)
    PackedConv qkv, out;
    float *og = nullptr, *ob = nullptr;   // LinearAttention's GroupNorm(1) after to_out
    // fused linear attention: to_qkv rows (q | k | v) and to_out in MFMA fragment order, network dtype
    void *fq = nullptr, *fk = nullptr, *fv = nullptr, *fo = nullptr;
};

struct LevelW {
    ResBlockW rb[2];
    AttnW attn;
    PackedConv resample;     // Downsample conv or Upsample transposed conv
};

struct SampleKey {           // identifies a captured step graph
    const void* x; const void* table; const void* ws; const void* keep; const void* xmean;
    int kind, B, H, W, flags, nsteps;
    unsigned long long seed;
    long long img_offset;
    float snr;
    const void* labels;
    const void* h2 = nullptr;   // the handle's table of time-path hidden vectors the captured evaluations read (or nullptr)
    bool operator==(const SampleKey& o) const {
        return labels == o.labels && h2 == o.h2 && x == o.x && table == o.table && ws == o.ws && keep == o.keep && xmean == o.xmean && kind == o.kind &&
               B == o.B && H == o.H && W == o.W && flags == o.flags && nsteps == o.nsteps && seed == o.seed &&
               img_offset == o.img_offset && snr == o.snr;
    }
};

struct DmnUnet {
    DmnUnetCfg cfg;
    bool loaded = false;
    std::vector<LevelW> downs, ups;
    ResBlockW mid1, mid2, final_rb;
    AttnW mid_attn;
    // time path (fp32)
    float *freqs = nullptr, *tw1 = nullptr, *tb1 = nullptr, *tw2 = nullptr, *tb2 = nullptr, *twb = nullptr, *tbb = nullptr;
    int tb_total = 0;
    // ingest / egress (fp32)
    float *init_w = nullptr, *init_b = nullptr, *tail_g = nullptr, *tail_b = nullptr, *tail_w = nullptr, *tail_bias = nullptr;
    void* init_wfrag = nullptr;   // bf16 handles: ingest weights as MFMA fragments
    float* class_embed = nullptr; // [num_classes + 1][dim] fp32 (class-conditional networks)
    const long long* labels = nullptr;   // borrowed device labels of the coming forwards
    // Batch the launch heuristics (tile sizes, pixel slices) plan for, whatever the batch of a call: the partial sums
    // of the GroupNorm statistics are then formed in the same order for an image however its batch is sharded, so
    // per-image results are bit-identical across splits. 0: plan for the batch of each call.
    int plan_batch = 64;
    // Diagnosis switches (environment DMN_UNET_PATHS, read when the handle is created): bit 0 launches the ingest and
    // the time path separately, bit 1 uses the unfused linear attention, bit 2 closes every ResnetBlock with its own
    // pass instead of leaving it to the next convolution, bit 3 keeps the bottleneck attention off the matrix cores,
    // bit 4 leaves the tail of every linear attention to the resampling convolution behind it (fewer launches, slower;
    // bit 5: only on the way down, bit 6: only at the top level), bit 7 closes the ResnetBlocks in front of an attention
    // with a pass of their own instead of inside block2.proj (no in-launch hand-off anywhere), bit 8 does the same for
    // the closing GroupNorm of the linear attentions only, bit 10 merges the split partials of a linear attention in a
    // launch of its own even where its second launch would merge them itself, bit 11 keeps the DDPM update a launch of its
    // own instead of applying it in the network's last kernel, bit 12 leaves the GroupNorm + SiLU + time bias of every
    // block1 to the convolution behind it (block2.proj) instead of block1.proj's own epilogue, bit 13 leaves the tail of a level's
    // first ResnetBlock to the second block's first convolution (as everywhere above the levels that fit the in-launch
    // exchange) instead of closing it in block2.proj, bit 14 does so only where closing it needs an exchange.
    int path_mask = 0;
    // diagnosis: workspace checksums after every launch of a forward (dmn_test_unet_trace)
    unsigned long long* trace_slots = nullptr;
    int trace_max = 0, trace_n = 0, trace_stop = -1;
    unsigned trace_poison = 0;
    std::vector<std::string> trace_names;
    // cached step graph of the sampling loops
    hipGraphExec_t graph_exec = nullptr;
    hipGraphExec_t graph_exec2 = nullptr;   // second half-batch chain of a DDPM loop (see sample_chains)
    SampleKey graph_key = {};
    // The DDPM loop runs a batch as `sample_chains` chains of half the images on two streams (the second one may be held
    // back by `sample_skew_us`; on MI355X no skew measured best): a step is a chain of ~85 dependent launches, and a second chain fills the gaps at
    // the launch boundaries and in the small bottleneck kernels. With the fixed planning batch every image gets the
    // same bits as in one chain. Environment DMN_SAMPLE_CHAINS (1 or 2) / DMN_SAMPLE_SKEW_US at handle creation.
    int sample_chains = 2;
    int sample_skew_us = 0;
    // The two launch chains of a sampling loop run on two streams of the handle (never on the caller's: whether two
    // streams share a hardware queue -- and then run one behind the other -- depends on every stream the process has
    // made, so the pair is probed once, when it is made, and kept).
    hipStream_t chain_stream[2] = {nullptr, nullptr};
    bool chains_concurrent = false;      // the probe saw the two streams run side by side
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_join0 = nullptr;
    int label_offset = 0;                   // first image of the chain that is being launched (class labels)
    int chain_id = 0;                       // which chain is being launched (probe buffers)
    // HIP events bracketing the last sampling loop on its stream (device time of the loop)
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;
    // pacing of a loop's enqueue (dmn_api.cpp:run_loop_inner): [slot][chain], recorded every loop_pace steps
    static constexpr int kPaceSlots = 64;
    hipEvent_t ev_pace[kPaceSlots][2] = {};
    int loop_pace = 8;                   // steps between pacing events (0: unpaced; DMN_LOOP_PACE)
    bool loop_lockstep = true;           // the two chains start every pacing block together (DMN_LOOP_LOCKSTEP=0: left to drift)
    int loop_graph_steps = 1;            // denoising steps per captured graph, at most (DMN_LOOP_GRAPH_STEPS)
    int loop_lockstep_every = 4;         // lockstep wait at every n-th pacing point (DMN_LOOP_LOCKSTEP_EVERY)
    int loop_depth = 2;                  // blocks of loop_pace steps in flight at most (2..kPaceSlots; DMN_LOOP_DEPTH)
    bool ev_valid = false;
    // measurement hook: HIP event pairs around the launches of the step's dominant convolution (eager launches only)
    std::vector<hipEvent_t> probe_ev;    // 3 * kProbePairs events once enabled
    int probe_n = 0;                     // pairs recorded since the probe was switched on
    int probe_on = 0;                    // 1: HIP events around eager launches, 2: device clock stamps (also in graphs)
    int probe_site = -1;                 // tuning aid (environment DMN_PROBE_SITE): with probe_on == 2 the n-th convolution launch
                                         // of an evaluation stamps its phases instead of the dominant one (dmn_test_probe_read)
    long long* probe_stamps = nullptr;   // [2 chains][kProbeMaxWgs][8] per-workgroup stamps of the probed launch
    unsigned long long* probe_acc = nullptr;   // per chain {sum of spans in 10 ns ticks, launches}
    float* h2_table = nullptr;           // hidden vectors of the time path for the rows of h2_table_of ([rows][4 dim])
    size_t h2_cap = 0;                   // floats allocated
    const float* h2_table_of = nullptr;  // the step table they were computed from (nullptr: none valid)
    long long* step_trace = nullptr;     // tuning aid (dmn_test_step_trace): [2 chains][8192] evaluation start clocks
    double probe_bytes = 0.0;            // algorithmic bytes of one probed launch
    double probe_shape[5] = {0, 0, 0, 0, 0};   // its multiply-add work (2 * images * output pixels * Cin * Cout * taps), Cin, Cout, output height, taps
    // Granule areas of the in-launch statistics hand-off (dmn_handoff.h), one per launch chain, owned by the handle:
    // [256 B header: epoch, give-ups][granules of every fused launch of an evaluation, in launch order]. Sized by
    // dmn_unet_workspace_bytes for the largest shape asked about; an evaluation whose fused launches do not fit
    // falls back to the separate closing passes.
    void* handoff_mem[2] = {nullptr, nullptr};
    size_t handoff_bytes = 0;
    // Epoch housekeeping: the tag of a granule is a 32-bit epoch that the device increments once per evaluation. The host
    // keeps an upper bound of the epochs consumed since the areas were last zeroed and zeroes them again (stream-ordered,
    // in front of a loop or forward, never inside a capture) before the bound reaches `epoch_limit`, so a wrapped epoch
    // can never meet a stale tag of the same value. Environment DMN_HANDOFF_EPOCH_LIMIT (tests use a small one).
    unsigned long long epoch_used = 0;
    unsigned long long epoch_limit = 0xFFF00000ull;
    unsigned giveups_carry = 0;          // give-ups counted in areas that have been zeroed since
    // Set while a loop / forward of this handle is being enqueued or captured although another handle's exchanging work
    // is in flight on the device (exchange_guard_enter returned false): every closing then takes its separate pass, as
    // under DMN_UNET_PATHS bit 7 -- the same arithmetic up to the order in which partial sums meet, no waiting
    // workgroups -- because the residency guarantee of
    // dmn_handoff.h covers the exchanging launches of ONE handle's two chains only.
    bool exchange_off = false;
};
constexpr int kProbePairs = 256;
constexpr int kProbeMaxWgs = 16384;

int pack_conv(PackedConv& pc, int dtype, const float* w, const float* bias, int Cout, int Cin, int kh, int kw, int stride,
              int pad, bool transposed);
void free_conv(PackedConv& pc);
void conv_args_from(const PackedConv& pc, int dtype, ConvArgs& a, int B, int H, int W);
// With a device step counter (`table` + `step_idx`), the evaluation also copies the counter to
// step_idx[kStepCopyOffset]; step_idx must then address at least kStepCopyOffset + 1 ints.
constexpr int kStepCopyOffset = 16;
int unet_forward_impl(DmnUnet* u, const float* x, const void* t, int t_is_float, const float* table, int* step_idx,
                      float* out, int B, int H, int W, void* ws, size_t ws_bytes, hipStream_t s, bool dry, size_t* high,
                      size_t* handoff_need = nullptr, const struct DdpmStepArgs* fuse_step = nullptr);
// Makes both chains' granule areas at least `bytes` large (never inside a stream capture: it allocates).
int unet_reserve_handoff(DmnUnet* u, size_t bytes);
// Accounts for `evals` coming network evaluations per chain; zeroes the granule areas on `s` first when the epoch bound
// would pass the limit (see DmnUnet::epoch_used). Not inside a stream capture.
int unet_epoch_reserve(DmnUnet* u, unsigned long long evals, hipStream_t s);
// Tabulates the time path's hidden vectors for the `rows` rows of a sampling loop's step table (one launch on `s`); the
// loop's evaluations then read their row. Called at every loop's entry (it also forgets the previous loop's table).
int unet_time_hidden_table(DmnUnet* u, const float* table, int rows, hipStream_t s);
// Process-wide registry of handles whose exchanging work (in-launch GroupNorm hand-offs) is being enqueued or still in
// flight. enter: true when no OTHER handle has such work pending on the current device -- the caller may exchange and must
// call leave(u, s) after its last launch on `s`; false: run without exchanges (set DmnUnet::exchange_off), no leave.
bool exchange_guard_enter(DmnUnet* u);
void exchange_guard_leave(DmnUnet* u, hipStream_t s, bool entered);
void exchange_guard_forget(DmnUnet* u);   // handle destruction
