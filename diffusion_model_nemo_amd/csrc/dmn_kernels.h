// Internal launch interface of the gfx950 kernels (not part of the C-ABI; see include/dmn.h for that).
// All activations are NHWC, element type T = float or __bf16 selected by `dtype`.
#pragma once
#include "dmn_common.h"
#include "dmn_handoff.h"
#include "dmn_sampler.h"

enum { DMN_F32 = 0, DMN_BF16 = 1 };
// convolution shapes the implicit-GEMM kernel is specialised for
enum { DMN_K3X3 = 0, DMN_K1X1 = 1, DMN_K4X4S2 = 2, DMN_KT2X2 = 3 };

// ---- group-norm statistics passed between kernels -----------------------------------------------
// A producer writes, for every (image, group), `slots` partial moments {sum, squared deviations from the
// partial's own mean, element count, unused}, one per contributing workgroup, each slot written exactly
// once: layout [B][G][slots][4] fp32. A consumer combines the slots in a fixed order (dmn_norm.h:
// deterministic, and free of the cancellation of sum / sum-of-squares pairs) and derives mean / rstd.
// 1 / inv_count is the number of elements per (image, group).
struct NormIn {
    const float* stats;   // nullptr: no normalisation
    int G;
    int slots;
    float inv_count;
    const float* gamma;   // [C]
    const float* beta;    // [C]
    const float* tbias;   // optional [B][tb_stride] added AFTER the activation (time-embedding bias)
    int tb_stride;
    int silu;             // apply SiLU after the affine normalisation
};
struct NormOut {
    float* stats;         // nullptr: no statistics wanted
    int G;
    int slots;            // slots per (image, group) the launch will fill; set by the launcher
};

struct ConvPhase {
    int pad_y, pad_x;     // top/left padding of this phase
    int out_oy, out_ox;   // output pixel = tile-space pixel * out_s + out_o
    long long w_off;      // element offset of this phase's packed weights
};

struct ConvArgs {
    int dtype;
    const void* in0;      // [B][H][W][C0]
    const void* in1;      // optional second source, channel-concatenated after in0: [B][H][W][C1]
    int C0, C1;
    int B, H, W;
    int Ho, Wo;           // tile-space output extent (for a transposed conv: the input extent)
    int out_H, out_W;     // extent of the output tensor
    int out_sy, out_sx;
    int kind;             // DMN_K3X3 / DMN_K1X1 / DMN_K4X4S2 / DMN_KT2X2 (fixes taps, extent and stride)
    int nphase;
    ConvPhase phase[4];
    const void* w;        // packed [phase][Cout/16][Cin/kChunk][ntaps][64 lanes][16 B]: MFMA A-operand fragments
    const float* bias;    // [Cout] or nullptr
    void* out;            // [B][out_H][out_W][Cout]
    int Cout;
    NormIn pro;           // normalise + activate the input while staging it (in0 only)
    const void* pro_res;  // optional (3x3 / resampling kinds, single source): tensor added behind the normalisation [+ SiLU]
                          //   of `pro` -- the tail of the ResnetBlock or linear attention that produced in0
    void* pro_out;        // optional with pro_res: receives the resulting tensor (same shape as in0)
    const void* res;      // optional residual added in the epilogue, same shape as out
    NormOut epi;          // statistics of the output after bias (taken before the residual add)
    int pro_direct;          // set by the launcher: the prologue's coefficients are formed per thread (dmn_norm.h: norm_direct_*)
    int epi_wgs;          // workgroups that touch one group of an image (1: a workgroup holds whole groups); set by the launcher
    const void* w2;       // optional fused 1x1 convolution of the same input (3x3 kind, no input normalisation):
    const float* bias2;   //   weights packed like `w` with one tap, bias [Cout],
    void* out2;           //   output [B][out_H][out_W][Cout]
    // Fused ResnetBlock tail (3x3 kind with the GroupNorm + SiLU prologue, single source; tail.gran != nullptr):
    // out = SiLU(GroupNorm_{epi.G}(conv + bias)) + res with the image's partial sums exchanged inside the launch
    // (epi.stats stays nullptr; epi.G gives the groups), and partial statistics of `out` into tail_stats.
    HandoffArgs tail;
    const float* tail_gamma;   // [Cout] affine parameters of the closing GroupNorm
    const float* tail_beta;
    float tail_inv_count;      // 1 / elements per (image, group) of that GroupNorm
    NormOut tail_stats;        // optional statistics of `out`
    int b_first, b_count;      // images [b_first, b_first + b_count) of the batch in this launch (set by the launcher)
    // Activating epilogue (3x3 kind, conv_act_fits; act_gamma != nullptr): out = SiLU(GroupNorm_{epi.G}(conv + bias)) +
    // act_tbias -- block1 of a ResnetBlock applying its own norm / act / time bias where a workgroup holds whole groups
    // of its image (epi.stats stays nullptr: the consumer reads a finished tensor)
    const float* act_gamma;    // [Cout]
    const float* act_beta;
    const float* act_tbias;    // optional [B][act_tb_stride]
    int act_tb_stride;
    float act_inv_count;       // 1 / elements per (image, group)
    int cg, lg_upp;       // channels staged per group and log2 of its 16-byte units per pixel; set by the launcher
    long long* dbg;       // optional [workgroups][8] phase timestamps (tuning aid; nullptr in production)
    int tall_tail;        // the launch is meant to close its ResnetBlock (tail): 32x16-pixel tiles where 16x16 leave an image
                          // too many workgroups for the in-launch exchange
    int cfg_tile, cfg_bn; // 0 = choose automatically; otherwise force the pixel tile (8 or 16) / channel tile (32 or 64)
    int plan_B;           // batch the tile heuristic plans for (0: B). A fixed value makes the choice -- and with it the
                          // order in which GroupNorm partial sums are formed -- independent of how a batch is sharded
};
int launch_conv(ConvArgs& a, hipStream_t s);
int conv_epi_slots(const ConvArgs& a);   // slots per (image, group) launch_conv will write
int conv_epi_wgs(const ConvArgs& a);     // workgroups that contribute to one (image, group)
int conv_tail_slots(const ConvArgs& a);  // same for tail_stats
bool conv_act_fits(const ConvArgs& a);   // epi.G groups: the launch may apply its own GroupNorm + SiLU (act_* fields)
bool conv_tail_fits(const ConvArgs& a);  // epi.G groups / their partial sums within the hand-off's limits
int conv_init_attributes();
void conv_set_debug_buffer(long long* dev);
long long* kernel_debug_buffer();   // timestamps of every later launch_conv go here (nullptr: off)

// out = [SiLU](GN(in)) [+ tbias] [+ res], optional statistics of `out`.
struct NormApplyArgs {
    int dtype;
    const void* in;
    void* out;
    const void* res;      // optional
    int B, HW, C;
    NormIn norm;
    NormOut epi;
    int plan_B;           // batch the slice heuristic plans for (0: B)
};
int launch_norm_apply(NormApplyArgs& a, hipStream_t s);
int norm_apply_slots(int B, int HW, int C);   // B: the planning batch

// time path: sinusoid -> Linear -> GELU -> Linear -> SiLU -> all per-block Linear(4*dim, Cout_i)
struct TimeArgs {
    const void* t;          // [B] int64 or float32, or nullptr to read `table`
    int t_is_float;
    const float* table;     // optional [nsteps][table_cols]; label = table[*step_idx][label_col]
    const int* step_idx;
    int* step_copy;         // optional: receives *step_idx (one thread), for the kernels that run after the network
    int table_cols, label_col;
    int rows, dim;          // rows = B, or 1 when every sample shares the table label; hidden = 4*dim
    const float* freqs;     // [dim/2]
    const float* w1; const float* b1;   // transposed [dim][4dim]
    const float* w2; const float* b2;   // transposed [4dim][4dim]
    const float* wb; const float* bb;   // concatenated per-block linears, transposed [4dim][nout], [nout]
    int nout;
    float* out;             // [B][nout]
    // The hidden vector behind the two dense layers depends on the label alone. A sampling loop knows all its labels
    // beforehand: launch_time_hidden tabulates the vector for every table row (h2_out, [rows][4 dim]) before the loop,
    // and the evaluations (h2 set, table mode) read their row instead of recomputing two dependent layers -- the same
    // code produced the numbers, so the biases have the same bits.
    const float* h2;
    float* h2_out;
};
int launch_time_mlp(TimeArgs& a, hipStream_t s);
int launch_time_hidden(TimeArgs& a, int table_rows, hipStream_t s);   // a.table, a.h2_out set; one workgroup per table row

// 7x7 ingest: NCHW fp32 -> NHWC T, Cin <= 4
struct IngestArgs {
    int dtype;
    const float* x;   // [B][Cin][H][W]
    const float* w;   // [Cin][7][7][Cout] (transposed from the PyTorch layout on the host)
    const void* wfrag;  // optional (bf16 only): MFMA A fragments [Cout/16][7 rows][64 lanes][8], k = 4*dx + ci
    const float* bias;
    void* out;        // [B][H][W][Cout]
    int B, Cin, H, W, Cout;
    const float* class_embed;   // optional [num_classes + 1][Cout]: row labels[b] is added to every pixel of image b
    const long long* labels;    // [B] or nullptr (every image takes row num_classes)
    int num_classes;
    unsigned* epoch;            // optional: the hand-off epoch of the chain, incremented by one thread (dmn_handoff.h)
    long long* trace;           // optional tuning aid: [8192] ring, slot epoch & 8191 receives the 100 MHz clock when the evaluation starts
};
int launch_ingest(IngestArgs& a, hipStream_t s);
bool ingest_time_fusable(const IngestArgs& a, const TimeArgs& t);
int launch_ingest_time(IngestArgs& a, TimeArgs& t, hipStream_t s);   // both in one launch (bf16 networks)

// tail: [GN + SiLU] -> 1x1 conv -> NCHW fp32
struct EgressArgs {
    int dtype;
    const void* in;   // [B][HW][C]
    NormIn norm;      // stats == nullptr: plain 1x1
    const float* w;   // [Cout][C]
    const float* bias;
    float* out;       // [B][Cout][HW]; unused with fuse_step
    int B, HW, C, Cout;
    // The DDPM loop's last launch of a step: instead of storing the network output, apply the sampler update to it
    // (step.c.x -> step.c.x_out; step.c.net is ignored) -- the arithmetic and noise of ddpm_step_kernel, same bits.
    int fuse_step;
    DdpmStepArgs step;
};
int launch_egress(EgressArgs& a, hipStream_t s);

// linear attention core on a materialised qkv tensor [B][n][3*hidden] (q | k | v, head-major)
struct LinAttnArgs {
    int dtype;
    const void* qkv;
    void* out;          // [B][n][hidden]
    float* ctx_part;    // workspace [B][heads][splits][d*d + 2*d]
    int B, n, heads, d, splits;
};
int launch_linear_attention(LinAttnArgs& a, hipStream_t s);
int linear_attention_splits(int n);

// fused linear attention: GroupNorm(1)(x) -> to_qkv -> softmaxes -> context -> to_out + bias, plus GroupNorm(1)
// statistics of the result. 4 heads of 32 channels; C in {32, 64, 128, 256}.
struct LinAttnFusedArgs {
    int dtype;
    const void* x;        // [B][n][C]
    void* y;              // [B][n][C]  to_out(...) + bias, before its GroupNorm
    NormIn norm;          // PreNorm statistics / affine of x
    const void *wq, *wk, *wv;   // [128/16][C/kChunk][64 lanes][16 B] fragment order
    const void* wo;       // [C/16][128/kChunk][64][16 B] fragment order
    const float* bias;    // [C] to_out bias
    float* ctx_part;      // workspace [B][4][splits][32*32 + 64]
    void* ctxT;           // workspace [B][4][32][32] in the activation dtype: merged, scaled context, transposed
    float* y_stats;       // [B][1][slices][4] partial moments
    int B, n, C;
    int slice1, splits;   // pass 1: workgroup slice in pixels (multiple of 32)
    int tpw, slices;      // pass 2: workgroup slice = 64*tpw pixels
    long long* dbg;       // optional phase timestamps [2][workgroups][8] (tuning aid; nullptr in production)
    // Fused closing (tail.gran != nullptr): out = GroupNorm(1)(y) * gamma + beta + x with the image's partial sums
    // exchanged inside the second launch; y and y_stats are then unused.
    HandoffArgs tail;
    const float* tail_gamma;   // [C]
    const float* tail_beta;
    float tail_inv_count;      // 1 / (n * C)
    void* out;                 // [B][n][C]
    int b_first, b_count;      // images of the second launch (set by the launcher)
    int merge_splits;          // > 0: the second launch merges that many split partials itself (set by the launcher)
    int separate_merge;        // 1: always merge in a launch of its own (diagnosis)
    int separate_ctx;          // 1: always run the first pass as a launch of its own, also where one workgroup holds the image (diagnosis)
};
bool linattn_tail_fits(int n, int B, int C, int dtype);   // the closing GroupNorm fits the in-launch exchange's limits
bool linattn_fused_supported(int C, int heads, int dim_head);
void linattn_fused_plan(int n, int B, int C, int dtype, int* slice1, int* splits, int* tpw, int* slices);
int launch_linattn_fused(LinAttnFusedArgs& a, hipStream_t s);

// softmax attention core on qkv [B][n][3*hidden] -> out [B][n][hidden]
struct AttnArgs {
    int dtype;
    const void* qkv;
    void* out;
    int B, n, heads, d;
    bool no_mfma;   // keep to the VALU kernel (diagnosis)
};
int launch_attention(AttnArgs& a, hipStream_t s);

// *slot += sum of the 32-bit words of [p, p + bytes) (bytes a multiple of 16; integer sum: independent of the order)
int launch_checksum(const void* p, size_t bytes, unsigned long long* slot, hipStream_t s);
// overwrites the LDS of every CU with `pattern` (diagnosis of reads of LDS a kernel did not write)
int launch_lds_poison(unsigned pattern, hipStream_t s);
// adds the span (latest end - earliest start) of the per-workgroup stamps of one launch to acc[0], 1 to acc[1]
int launch_stamp_span(long long* stamps, int wgs, unsigned long long* acc, hipStream_t s);
