// Kernels of the training step (forward with saved activations, backward, optimiser) of the DDPM U-Net:
// DDPM.training_step of the reference (diffusion_model_nemo/models/ddpm.py:43-57) followed by what its
// Lightning trainer does with the loss (backward, gradient clipping, AdamW). Activations are NHWC float32.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// Implicit GEMM over output pixels: out[m][n] (+)= sum_{tap, c} gather(in, m, tap)[c] * w[(tap*Cin + c)][n] + bias[n].
// tmode 0 gathers like a convolution (iy = oy*stride - pad + ky), tmode 1 like a transposed convolution
// (iy = (oy + pad - ky) / stride where divisible). The first serves Conv2d forward and ConvTranspose2d input
// gradients, the second ConvTranspose2d forward and Conv2d input gradients; linear layers are 1x1 on 1x1 images.
// Activation tensors (and their gradients) are float32 (adt 0) or bfloat16 (adt 1: bf16 storage implies bf16 matrix-core
// operands); parameters, parameter gradients, statistics and the time path's small tensors may differ per call.
struct PGemmArgs {
    const void* in;
    int ldin, Cin;          // floats per input pixel row; channels consumed (from the row's start)
    int B, Hi, Wi, Ho, Wo;
    int kh, kw, stride, pad, tmode;
    const float* w;         // [kh*kw*Cin][ldw]
    int ldw;
    const float* bias;      // [N] or null
    void* out;
    int ldout, N;
    int accumulate;
    int bf16;               // 1: operands rounded to bf16 for the matrix cores (fp32 accumulate)
    double* stats;          // optional [B][statsG][2] float64, zeroed: += {sum, sum of squares} of the output per GroupNorm group
    int statsG;
    int adt;                // type of in / out
};
int launch_pgemm(const PGemmArgs& a, hipStream_t s);
// whether launch_pgemm can produce `stats` for these shapes (64-pixel tiles must not straddle images)
int pgemm_stats_supported(const PGemmArgs& a);

// Weight gradient: dw[(tap*Cin + c)][n] += sum_m gather(x, m, tap)[c] * dy[m][n]   (atomic over pixel splits)
struct WGemmArgs {
    const void* x;
    int ldx, Cin;
    int B, Hi, Wi, Ho, Wo;
    int kh, kw, stride, pad, tmode;
    const void* dy;
    int lddy, N;
    float* dw;
    int lddw;
    int bf16;
    float* dbias;           // optional [N]: += column sums of dy (the bias gradient), taken from the staged tiles
    int adt;                // type of x / dy
};
int launch_wgemm(const WGemmArgs& a, hipStream_t s);

// GroupNorm over NHWC: stats[b][g] = {sum, sum of squares} (atomic into zeroed memory). The sums are float64: the
// variance is E[x^2] - E[x]^2, which in float32 loses log10(mean^2 / var) of its 7 digits (a group 50 standard deviations
// off zero keeps 3); float64 atomics cost nothing measurable here and keep 12. Workgroup-level partials of a few
// elements are formed in float32 and widened before they meet.
int launch_gn_stats(const void* x, int adt, int B, int HW, int C, int G, double* stats, hipStream_t s);
// y = act(gn(x) * gamma + beta) + tb[b][c] + res   (act: 0 none, 1 SiLU; tb optional with row stride tb_ld)
struct GnArgs {
    const void* x;
    const double* stats;    // [B][G][2] float64 {sum, sum of squares}
    const float *gamma, *beta;
    const float* tb;
    int tb_ld;
    const void* res;        // optional residual added after the activation (same shape as y)
    void* y;
    double* ystats;         // optional [B][2] float64, zeroed: += {sum, sum of squares} of y per image (a following GroupNorm(1))
    int B, HW, C, G, act;
    int adt;                // type of x, res, y, dy, dx
    // backward
    const void* dy;
    void* dx;
    int accumulate;
    float* sums;            // [B][C][3] zeroed: sum dz*xhat, sum dz, sum dy
    float *dgamma, *dbeta;  // [C] accumulated atomically
    float* dtb;             // [B][tb_ld] slice, accumulated
    // the residual's gradient is dy itself: the caller aliases or accumulates it
};
int launch_gn_apply(const GnArgs& a, hipStream_t s);
int launch_gn_bwd(const GnArgs& a, hipStream_t s);

// elementwise helpers (element counts and widths are multiples of 4)
int launch_axpby(void* dst, const void* src, int adt, long long n, int accumulate, hipStream_t s);      // dst (+)= src
int launch_add(void* out, const void* a, const void* b, int adt, long long n, hipStream_t s);          // out = a + b
// dst[row][doff + c] (+)= src[row][soff + c], c < C
int launch_copy_slice(void* dst, int ldd, int doff, const void* src, int lds, int soff, int C, long long rows, int adt,
                      int accumulate, hipStream_t s);
// act: 0 SiLU, 1 exact GELU. Forward y = act(x); backward dx (+)= dy * act'(x).
int launch_act_fwd(const void* x, void* y, int adt, long long n, int act, hipStream_t s);
int launch_act_bwd(const void* x, const void* dy, void* dx, int adt, long long n, int act, int accumulate, hipStream_t s);
// Class conditioning (modules/unet.py:134-141): x[b][p][c] += table[label_b][c] in place (labels null: every image takes
// row `pad`, the "no class" row); backward: dtable[label_b][c] += sum_p dy[b][p][c] except for row `pad`
// (nn.Embedding's padding_idx keeps that row's gradient at zero).
int launch_class_bias_fwd(void* x, int adt, const float* table, const long long* labels, int pad, int B, int HW, int C, hipStream_t s);
int launch_class_bias_bwd(const void* dy, int adt, const long long* labels, int pad, float* dtable, int B, int HW, int C, hipStream_t s);
// Sinusoidal embedding of integer or float labels: out[b][dim]
int launch_sinus(const void* t, int t_is_float, int B, int dim, float* out, hipStream_t s);

// x_t = a[t_b] x0 + b[t_b] eps (a, b null: x_t = x0) from NCHW float32 to NHWC in the activation type, and eps to NHWC
// float32 as the target.
// Labels are clamped to [0, n_timesteps).
int launch_train_ingest(const float* x0, const float* eps, const long long* t, const float* sa, const float* sb, int n_timesteps,
                        void* xt, int adt, float* tgt, int B, int C, int H, int W, hipStream_t s);
// loss (+= into a zeroed scalar) of DiffusionLoss with reduction 'mean' (loss/simple_loss.py:37-45): kind 0 l1,
// 1 l2, 2 huber (smooth_l1, beta 1); dout = d loss / d out. out, tgt, dout: [n] in the same (NHWC) order. The mean runs
// over n_mean elements (>= n: a launch may hold a part of the batch).
int launch_loss(const void* out, const float* tgt, int adt, long long n, long long n_mean, int kind, float* loss, void* dout,
                hipStream_t s);
// NHWC [B][HW][C] -> NCHW float32 [B][C][HW]
int launch_nhwc_to_nchw(const void* in, int adt, float* out, int B, int C, int HW, hipStream_t s);

// LinearAttention core on qkv [B][n][3*heads*32] (parts/mha.py:48-56): out [B][n][heads*32]. The pixels of an image
// are split over linattn_train_splits() workgroups. ctx [B][heads][32][32] (unnormalised k^T v) and kstat
// [B][heads][2][32] (max, sum of exp) must be ZEROED by the caller and are kept for the backward; pmax
// [B][heads][splits][32] is scratch; dctx [B][heads][32][32] must be zeroed.
// bf16: the per-pixel 32x32 products take bfloat16 operands (fp32 accumulate) instead of exact fp32.
int linattn_train_splits(int B, int n, int heads);
int launch_linattn_train_fwd(const void* qkv, void* out, int adt, float* ctx, float* kstat, float* pmax, int B, int n, int heads,
                             int bf16, hipStream_t s);
int launch_linattn_train_bwd(const void* qkv, const void* dout, int adt, const float* ctx, const float* kstat, float* dctx, void* dqkv,
                             int B, int n, int heads, int bf16, hipStream_t s);
// Attention core (parts/mha.py:20-29): out [B][n][heads*32]; lse [2][B*heads][n] (log-sum-exp, then scratch). With bf16
// operands and 64 <= n <= 256 (n % 64 == 0) the three launches run on the matrix cores, else on the vector units.
int launch_attn_train_fwd(const void* qkv, void* out, int adt, float* lse, int B, int n, int heads, int bf16, hipStream_t s);
int launch_attn_train_bwd(const void* qkv, const void* out, const void* dout, int adt, const float* lse, void* dqkv, int B, int n,
                          int heads, int bf16, hipStream_t s);

// Layout changes between the reference's parameter layout and the GEMM layout, for all tensors in one launch.
struct RepackDesc {
    long long src_off;      // offset in the flat reference-layout buffer
    int n_out, n_in, taps;  // Cout, Cin, kh*kw
    int transposed;         // 1: ConvTranspose2d weight [Cin][Cout][kh][kw]
    long long f_off;        // forward GEMM layout [(tap*Cin + ci)][f_ld] at column 0 of this offset
    int f_ld;
    long long b_off;        // backward GEMM layout [(tap*Cout + co)][b_ld]
    int b_ld;
};
int launch_repack(const RepackDesc* descs_dev, int n_desc, long long max_numel, const float* params, float* wf, float* wb,
                  hipStream_t s);
// grads_ref[src_off + idx] = dwf[...] for every described tensor
int launch_unpack_grads(const RepackDesc* descs_dev, int n_desc, long long max_numel, const float* dwf, float* grads,
                        hipStream_t s);

// Global gradient norm and AdamW (torch.nn.utils.clip_grad_norm_, torch.optim.AdamW semantics).
// parts[i] = sum of g^2 over the elements workgroup i visits, i < *n_parts <= kSqnormMaxParts: every slot written once,
// summed in index order by the consumer, so the norm is the same bit pattern on every run and every rank
constexpr int kSqnormMaxParts = 1024;
int launch_sqnorm(const float* g, long long n, float* parts, int* n_parts, hipStream_t s);
struct AdamWArgs {
    float* p;
    const float* g;
    float *m, *v;
    long long n;
    float lr, beta1, beta2, eps, weight_decay;
    float bias1, bias2;     // 1 - beta^step
    float max_norm;         // <= 0: no clipping
    const float* sq_parts;  // device [n_parts]: partial sums of g^2 over ALL parameters (launch_sqnorm)
    int n_parts;
    float grad_scale;       // multiplies g before everything else (1/world_size after a summing all-reduce)
};
int launch_adamw(const AdamWArgs& a, hipStream_t s);
// One wave that waits `us` microseconds of the constant 100 MHz counter and exits (bounded by an iteration count as
// well): the probe with which the trainer finds out which of its streams share a hardware queue.
int launch_spin(int us, hipStream_t s);
