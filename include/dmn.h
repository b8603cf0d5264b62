/* C-ABI of libdmn_hip.so: the MI355X (gfx950) implementation of the DDPM / score-SDE denoising hot path
 * of titu1994/diffusion_model_nemo. Plain pointers and sizes only; no framework types.
 *
 * The reference has no native seam: its "plugin API" for this path is Hydra `_target_` instantiation of
 * Python classes (diffusion_model_nemo/models/ddpm.py:27-29). The Python mirror of those classes in
 * `diffusion_model_nemo_amd/` binds the entry points below through ctypes; each declaration cites the
 * reference code it replaces (paths relative to the reference repository root).
 *
 * Conventions
 *   - every function returns 0 on success or a negative DmnStatus; dmn_last_error() describes the failure
 *     (thread-local). No exception crosses this boundary.
 *   - device pointers are borrowed for the duration of the call (asynchronously, on `stream`); the library
 *     never frees or retains them, except the packed-weight copy owned by a DmnUnet handle.
 *   - images are NCHW float32 (the reference's layout); `dtype` selects the INTERNAL activation/weight
 *     type: 0 = float32 (exact-f32 MFMA, parity mode), 1 = bfloat16 (bf16 MFMA, fp32 accumulate).
 *   - kernels are enqueued on `stream` (a hipStream_t passed as void*); no hidden synchronisation.
 */
#ifndef DMN_H
#define DMN_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define DMN_MAX_MULTS 8
#define DMN_TABLE_COLS 12

typedef struct DmnUnet DmnUnet;

/* Constructor arguments of Unet (diffusion_model_nemo/modules/unet.py:14-29) for use_convnext=False. */
typedef struct {
    int dim;                       /* base width */
    int n_mults;
    int dim_mults[DMN_MAX_MULTS];
    int channels;                  /* image channels (<= 4) */
    int out_dim;                   /* channels * (learned_variance ? 2 : 1) unless overridden (<= 8) */
    int groups;                    /* resnet_block_groups */
    int heads, dim_head;           /* attention geometry: 4 x 32 in the reference (parts/mha.py:8,34) */
    int tail_norm;                 /* 1: resnet_block_order == 'bn_act_conv' => GN+SiLU before the last 1x1 */
    int with_time_emb;
    int dtype;                     /* 0 f32, 1 bf16 */
    int num_classes;               /* 0: unconditional; > 0: class_embed has num_classes + 1 rows (unet.py:118-120) */
} DmnUnetCfg;

/* One parameter of the reference state_dict: HOST float32 data, contiguous, PyTorch shape order. */
typedef struct {
    const char* name;              /* e.g. "downs.0.0.block1.proj.weight" */
    const float* data;
    int ndim;
    int64_t shape[4];
} DmnTensorDesc;

const char* dmn_last_error(void);
const char* dmn_version(void);

/* ---- U-Net: replaces Unet.__init__/forward (modules/unet.py:14-120,131-168) ------------------------------ */
int dmn_unet_create(const DmnUnetCfg* cfg, DmnUnet** out);
void dmn_unet_destroy(DmnUnet* u);
/* Packs the reference's parameters (names as in Unet.state_dict()) into the device layout. Synchronous. */
int dmn_unet_load_weights(DmnUnet* u, const DmnTensorDesc* params, int n_params);
/* Bytes of caller workspace one evaluation of this shape needs. It also changes the handle (hence no const): the handle's own
 * device areas for the in-launch exchanges (see dmn_unet_handoff_giveups) are made large enough for this shape; when they
 * have to grow, the device is synchronised, memory is allocated and cached step graphs are dropped. Call it before a stream
 * capture, not inside one (a capture on the legacy stream is detected and refused with DMN_ERR_INVALID; a capture on
 * another stream cannot be seen from here). */
int dmn_unet_workspace_bytes(DmnUnet* u, int B, int H, int W, size_t* bytes);
/* Class labels of the next forwards / sampling loops (unet.py:134-141): DEVICE int64 [B], borrowed until replaced;
 * NULL selects the extra "null" row num_classes for every image, like classes=None in the reference. */
int dmn_unet_set_class_labels(DmnUnet* u, const int64_t* labels);

/* Batch that the launch heuristics (convolution tiles, pixel slices of the normalisation and attention kernels) plan
 * for; default 64. The choice fixes the order in which per-image GroupNorm partial sums are formed, so with a fixed
 * planning batch the output for an image is bit-identical however the batch that contains it is sharded over calls or
 * ranks. 0 plans for the batch of each call (fastest for small batches; rounding then depends on the split).
 * Workspace sizes depend on it: set it before dmn_unet_workspace_bytes. No counterpart in the reference. */
int dmn_unet_set_plan_batch(DmnUnet* u, int batch);

/* A ResnetBlock's closing SiLU(GroupNorm(h)) + res (parts/convnext.py:85-86) is computed by the block's second
 * convolution itself: the workgroups of an image exchange their GroupNorm partial sums inside the launch through a
 * device area the handle owns (allocated by dmn_unet_workspace_bytes / dmn_sample_workspace_bytes for the largest shape
 * asked about; a shape never asked about runs the separate closing passes instead). The waits are bounded (0.5 s); one
 * that runs out is counted and the evaluation's result is invalid. Returns the count since the handle was created (it
 * never goes down: a caller compares it with the value it saw last); waits for the device. The exchanges are only
 * used while no OTHER handle has exchanging work in flight on the GPU: a loop or forward that starts beside one takes
 * the separate closing passes (the same arithmetic with the partial sums of the following GroupNorm formed per pixel
 * slice instead of per convolution tile: results agree to summation order). No counterpart in the reference. */
int dmn_unet_handoff_giveups(DmnUnet* u, unsigned* count);
/* Tuning aid, no counterpart in the reference: per-workgroup phase stamps of the convolution launch selected by the
 * environment variable DMN_PROBE_SITE (its number among an evaluation's convolution launches), taken inside the sampling
 * loops while dmn_unet_probe_dominant(u, 2) is on. host: [nwgs][8] 64-bit words. */
int dmn_test_probe_read(DmnUnet* u, int chain, long long* host, int nwgs);
/* Measurement aid, no counterpart in the reference: choose that launch at run time (site = -1: the dominant level-0
 * convolution again); dmn_unet_probe_dominant(u, 2) / dmn_unet_probe_result then clock it. shape (optional, 5 doubles)
 * receives what the launch clocked last was: {2 * images * output pixels * Cin * Cout * taps, Cin, Cout, output height, taps}. */
int dmn_test_probe_site(DmnUnet* u, int site, double* shape);
/* Tuning aid, no counterpart in the reference: the 100 MHz device clock at the start of every network evaluation, per
 * launch chain, in a ring of 8192 entries indexed by the chain's evaluation count. host == NULL switches the recording on
 * and zeroes the rings; otherwise chain `chain`'s ring is copied to host[8192]. */
int dmn_test_step_trace(DmnUnet* u, int chain, long long* host);

/* eps = Unet(x, t).  t: [B] int64 (t_is_float = 0) or float32 (1), device memory. out: [B][out_dim][H][W]. */
int dmn_unet_forward(DmnUnet* u, const float* x, const void* t, int t_is_float, float* out, int B, int H, int W,
                     void* workspace, size_t workspace_bytes, void* stream);

/* ---- samplers ------------------------------------------------------------------------------------------------
 * `table` is a DEVICE array [nsteps][DMN_TABLE_COLS] float32, one row per executed step in execution order,
 * built on the host from the reference's schedule formulas (gaussian_diffusion.py:44-83; sde_lib/).
 * Row layout: see diffusion_model_nemo_amd/csrc/dmn_sampler.h.
 * Noise: Philox4x32-10 + Box-Muller keyed by (seed, step id, stream, img_offset + b, element); pass `noise`
 * (device [B][C][H][W]) to inject a draw instead (single-step entry points only). */

/* One ancestral step: GaussianDiffusion.p_sample given the network output (gaussian_diffusion.py:157-167,
 * 125-154,118-123,91-101) or LearnedGaussianDiffusion.p_mean_variance (learned_gaussian_diffusion.py:35-48).
 * pred_x0: bit 0 = objective 'pred_x0'; bit 1 = generalized (DDIM) update of GeneralizedGaussianDiffusion.p_sample
 * (generalized_gaussian_diffusion.py:78-97) on a table with the DDIM row layout. Same meaning in the loop below. */
int dmn_ddpm_step(const float* table, int row, const float* x, const float* model_out, const float* noise, float* x_out,
                  int B, int C, int H, int W, int learned_variance, int pred_x0, uint64_t seed, int64_t img_offset,
                  void* stream);

/* GaussianDiffusion.p_sample_loop (gaussian_diffusion.py:171-189): x <- x_T (drawn here when draw_prior != 0)
 * then nsteps x { eps = Unet(x, t); x = p_sample(...) }. `keep` (optional, device [nsteps][B*C*H*W]) receives x
 * after every step. The step body is captured once into a hipGraph and replayed when use_graph != 0 and
 * `stream` is not the null stream. */
int dmn_ddpm_sample_loop(DmnUnet* u, const float* table, int nsteps, float* x, int B, int H, int W, int learned_variance,
                         int pred_x0, int draw_prior, uint64_t seed, int64_t img_offset, float* keep, void* workspace,
                         size_t workspace_bytes, void* stream, int use_graph);

/* One predictor update as an affine map x_mean = a x + b net, x = x_mean + c z (rows 1..3 of the table):
 * EulerMaruyamaPredictor / ReverseDiffusionPredictor / AncestralSamplingPredictor.update_fn
 * (sde_predictors/) composed with the score adapter (loss/sde_loss/score_function_loss.py:48-91). */
int dmn_sde_predictor_step(const float* table, int row, const float* x, const float* model_out, const float* noise,
                           float* x_out, float* x_mean, int B, int C, int H, int W, uint64_t seed, int64_t img_offset,
                           void* stream);
/* One Langevin (use_norms=1, sde_correctors/langevin_corrector.py:26-33) or annealed-Langevin (use_norms=0,
 * annealed_langevin_dynamics_corrector.py:34-39) inner step. scratch: >= dmn_sde_scratch_bytes(B, C, H, W). */
int dmn_sde_corrector_step(const float* table, int row, const float* x, const float* model_out, const float* noise,
                           float* x_out, int B, int C, int H, int W, float snr, int inner, int use_norms, uint64_t seed,
                           int64_t img_offset, void* scratch, void* stream);
size_t dmn_sde_scratch_bytes(int B, int C, int H, int W);
/* dmn_sde_corrector_step without the `scratch` argument, for callers that cannot allocate device memory (code under
 * HIP graph capture, serving loops, harnesses that forbid hipMalloc). No counterpart in the reference. The per-sample
 * partial sums live in a 64 KiB device array of the library (one copy per GPU, held until the process ends), so
 * B * min(16, ceil(C*H*W / 1024)) * 8 bytes must not exceed 65536 (DMN_ERR_LIMIT otherwise); the call allocates and
 * frees nothing. Results are bit-identical to dmn_sde_corrector_step. Every call on one GPU shares that GPU's array:
 * calls that overlap on one GPU (different streams) race; calls on one stream do not. The device that is current at
 * the call must be the one that owns the tensors. */
int dmn_sde_corrector_step_static(const float* table, int row, const float* x, const float* model_out, const float* noise,
                                  float* x_out, int B, int C, int H, int W, float snr, int inner, int use_norms,
                                  uint64_t seed, int64_t img_offset, void* stream);

/* PredictorCorrectorSampler.forward (sde_samplers/predictor_corrector_sampler.py:58-120): corrector then predictor
 * per iteration. corrector: 0 none, 1 langevin, 2 ald. predictor: 0 none, 1 affine (table rows). `keep` receives
 * x_mean (denoise != 0) or x after every iteration. prior_scale: sigma_max for VE, 1 for VP. Unless the corrector is
 * langevin (its step size is a batch mean) the images of a batch are independent and the loop runs as two half-batch
 * launch chains like dmn_ddpm_sample_loop
 * (same bits; the workspace of dmn_sample_workspace_bytes covers both layouts). */
int dmn_pc_sample_loop(DmnUnet* u, const float* table, int nsteps, float* x, int B, int H, int W, int predictor,
                       int corrector, int n_corrector_steps, float snr, int denoise, int draw_prior, float prior_scale,
                       uint64_t seed, int64_t img_offset, float* keep, float* x_mean_out, void* workspace,
                       size_t workspace_bytes, void* stream, int use_graph);
/* The sampling loops run a batch of B >= 16 (even) images as two half-batch launch chains on two streams of the handle
 * (forked from and joined to the caller's stream), which fills the gaps a single chain of ~70 dependent launches
 * leaves; every image gets the bits it gets in one chain. No counterpart in the reference (gaussian_diffusion.py:171-189
 * is one Python loop). The Langevin corrector's step size is a mean over the whole batch (langevin_corrector.py:26-33):
 * there the chains meet on the device once per corrector step and exchange their images' norms.
 * dmn_unet_set_sample_chains: chains 1 or 2 (default 2, environment DMN_SAMPLE_CHAINS), skew_us = head start of the
 * first chain (default 0, DMN_SAMPLE_SKEW_US). dmn_ddpm_loop_chains: chains a loop over B images will use. Workspaces
 * sized by dmn_sample_workspace_bytes hold either form.
 * dmn_unet_chains_concurrent: makes the handle's two chain streams if need be and reports whether a probe saw them run
 * side by side (1) or one behind the other on a shared hardware queue (0: two chains still give the right bits, but no
 * speed, and the Langevin corrector keeps to one chain). */
int dmn_unet_set_sample_chains(DmnUnet* u, int chains, int skew_us);
int dmn_ddpm_loop_chains(const DmnUnet* u, int B);
int dmn_unet_chains_concurrent(DmnUnet* u, int* concurrent);

/* AbstractDiffusionModel.calculate_bits_per_dimension (models/abstract_diffusion_model.py:137-197) for the
 * 'pred_noise' objective: for every row of `table` (BPD row layout, csrc/dmn_sampler.h; normally t = T-1 .. 0)
 * x_t = q_sample(x0, t) with generator stream 3, network evaluation, then the per-sample variational-bound term in
 * bits (VariationalBoundLoss.compute_variation_loss_terms, loss/variational_bound_loss.py:31-52) into
 * terms[b * terms_stride + t]. prior[b] receives the prior term computed from sqrt(a_{T-1}) and log(1 - a_{T-1}).
 * x0, terms, prior: device float32. workspace: >= dmn_sample_workspace_bytes. Every term is per image: the loop runs as
 * two half-batch launch chains like dmn_ddpm_sample_loop (same bits). */
int dmn_bpd_loop(DmnUnet* u, const float* table, int nsteps, const float* x0, int B, int H, int W, int learned_variance,
                 uint64_t seed, int64_t img_offset, float* terms, int terms_stride, float* prior, float prior_sqrt_acp,
                 float prior_log_1m_acp, void* workspace, size_t workspace_bytes, void* stream, int use_graph);
int dmn_sample_workspace_bytes(DmnUnet* u, int B, int H, int W, size_t* bytes);
/* Device time (HIP events on the loop's stream) of the last dmn_*_sample_loop call on this handle; waits for it.
 * No counterpart in the reference: measurement hook used by bench.py. */
int dmn_loop_last_ms(DmnUnet* u, float* ms);
/* Average duration (HIP events on `stream`) of `iters` back-to-back launches of the kernel that dominates a
 * denoising step: the level-0 3x3 convolution of a ResnetBlock with its fused GroupNorm + SiLU + time-bias prologue
 * and statistics epilogue (parts/convnext.py:35-41,81-85), on a [B, dim, H, W] activation. alg_bytes receives the
 * algorithmic traffic of one launch (input + output + weights). workspace: >= dmn_unet_workspace_bytes(B, H, W).
 * No counterpart in the reference: measurement hook used by bench.py. */
int dmn_bench_dominant_kernel(DmnUnet* u, int B, int H, int W, int iters, void* workspace, size_t workspace_bytes,
                              void* stream, float* avg_ms, double* alg_bytes);

/* The same kernel timed where it runs: HIP event pairs around its launch inside eager (use_graph = 0) network
 * evaluations or sampling loops on one stream, so that it follows and precedes the same kernels as in a captured step
 * and finds its operands in the same cache state. dmn_unet_probe_dominant(u, 1) starts collecting (at most 256
 * launches), (u, 0) stops; dmn_unet_probe_result waits for the recorded markers and returns avg_ms = the average
 * distance of the two markers around the kernel (the kernel plus one marker packet), marker_ms = the average distance of
 * two markers with nothing between them (recorded just before each probed launch), the number of probed launches and
 * the algorithmic bytes of one launch. avg_ms - marker_ms estimates the kernel's own duration.
 * No counterpart in the reference: measurement hooks used by bench.py. */
int dmn_unet_probe_dominant(DmnUnet* u, int enable);
int dmn_unet_probe_result(DmnUnet* u, float* avg_ms, float* marker_ms, int* launches, double* alg_bytes);

/* Measured ceilings of the device: `iters` back-to-back launches timed with HIP events on `stream`.
 * dmn_bench_stream: mode 0 copies `bytes` from src to dst (HBM traffic 2 x bytes per launch), mode 1 only reads src
 * (dst: any 4 writable bytes). Buffers 16-byte aligned, bytes a multiple of 16.
 * dmn_bench_mfma: register-resident matrix-core loop, kind 0 v_mfma_f32_16x16x32_bf16, kind 1 v_mfma_f32_16x16x4_f32;
 * 256 x wgs_per_cu workgroups of four waves, 32 x n_outer instructions per wave; sink: >= 65536 * wgs_per_cu floats;
 * flops receives the operations of one launch.
 * No counterpart in the reference: measurement hooks used by bench.py. */
int dmn_bench_stream(int mode, void* dst, const void* src, size_t bytes, int iters, void* stream, float* avg_ms);
int dmn_bench_mfma(int kind, int wgs_per_cu, int n_outer, int iters, float* sink, void* stream, float* avg_ms, double* flops);
/* The floor under a launch chain (no counterpart in the reference): `launches` dependent launches of a kernel that does
 * nothing (wgs workgroups, one store each) captured into one graph and replayed `iters` times on `stream` (not the
 * legacy stream); us_per_launch = average distance between two launches. sink: >= wgs floats. */
int dmn_bench_launch_floor(int launches, int wgs, int iters, float* sink, void* stream, float* us_per_launch);

/* Whether kernels launched on streams a and b run in turn because the runtime mapped both onto one hardware queue
 * (the device exposes few: four by default). A timing probe: two 300 us single-wave waits, twice; synchronises both
 * streams. No counterpart in the reference: lets a caller that wants concurrent chains pick streams that can overlap. */
int dmn_streams_share_queue(void* a, void* b, int* shared);

/* ---- training step ------------------------------------------------------------------------------------------------
 * DDPM.training_step (models/ddpm.py:43-57) and what its Lightning trainer does with the returned loss: backward
 * through Unet.forward (modules/unet.py:131-168), gradient clipping (trainer.gradient_clip_val) and AdamW
 * (examples/configs/ddpm/unet_small.yaml:65-78). Parameters and gradients are ONE flat float32 device buffer each,
 * every tensor in the reference's own layout at the offset dmn_train_tensor_info reports (offsets are multiples of
 * 4 floats), so a state_dict maps in by name and one all-reduce covers every gradient. float32 activations.
 * Networks with out_dim == channels only. */
typedef struct DmnTrainer DmnTrainer;
/* loss_kind: 0 'l1', 1 'l2', 2 'huber' with reduction 'mean' (loss/simple_loss.py:37-45).
 * cfg->dtype: 0 exact float32 matrix cores and float32 activations (parity mode); 1 matrix-core operands rounded to
 * bfloat16 (float32 accumulate), activations float32; 2 as 1 with the saved activations and their gradients STORED as
 * bfloat16 (half the memory traffic). Parameters, parameter gradients, statistics, softmax and the optimiser are
 * float32 in every mode. */
int dmn_train_create(const DmnUnetCfg* cfg, int loss_kind, DmnTrainer** out);
void dmn_train_destroy(DmnTrainer* tr);
/* Class labels of the next steps (unet.py:134-141; ConditionalDDPM.get_diffusion_model, models/conditional_ddpm.py:51-67,
 * applies the label drop-out before): DEVICE int64 [B], borrowed until replaced; NULL = the "no class" row for every
 * image. The gradient of that row stays zero (padding_idx of the reference's nn.Embedding). */
int dmn_train_set_class_labels(DmnTrainer* tr, const int64_t* labels);
int dmn_train_num_tensors(const DmnTrainer* tr);
int64_t dmn_train_flat_size(const DmnTrainer* tr);           /* floats in the flat parameter / gradient buffers */
/* name as in Unet.state_dict(); shape in PyTorch order (shape4 receives 4 entries, unused ones 0). */
int dmn_train_tensor_info(const DmnTrainer* tr, int idx, const char** name, int64_t* offset, int* ndim, int64_t* shape4);
int dmn_train_workspace_bytes(DmnTrainer* tr, int B, int H, int W, size_t* bytes);
/* x_t = sqrt_acp[t] x0 + sqrt_1m_acp[t] noise (gaussian_diffusion.py:104-116; both tables NULL: x_t = x0),
 * out = Unet(x_t, t), *loss = DiffusionLoss(out, noise), grads = d loss / d params (overwritten, not accumulated).
 * x0, noise: device [B][C][H][W]; t: device int64 [B] (clamped to [0, n_timesteps) for the table look-up; the network
 * sees it as given); tables: device float32 [n_timesteps]; loss: device scalar;
 * model_out: optional device [B][C][H][W]. workspace: 256-byte aligned, >= dmn_train_workspace_bytes. */
int dmn_train_loss_and_grads(DmnTrainer* tr, const float* params, float* grads, const float* x0, const float* noise,
                             const int64_t* t, const float* sqrt_acp, const float* sqrt_1m_acp, int n_timesteps, int B, int H,
                             int W, float* loss, float* model_out, void* workspace, size_t workspace_bytes, void* stream);
/* torch.nn.utils.clip_grad_norm_(max_norm) (max_norm <= 0: off) followed by one torch.optim.AdamW step on flat
 * buffers, all on the device (no host read of the norm). grads are multiplied by grad_scale first (1/world_size after
 * a summing all-reduce). step counts from 1. scratch: device, >= 4096 bytes (partial sums of the squared norm, added
 * in a fixed order: replicas that hold the same gradients compute the same clip factor bit for bit). */
int dmn_adamw_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1,
                   float beta2, float eps, float weight_decay, int step, float max_norm, float grad_scale, float* scratch,
                   void* stream);

/* N(0,1) * scale into x[B][per] with the library's generator (prior draws; tests). */
int dmn_fill_normal(float* x, int B, int per, uint64_t seed, uint32_t step, int64_t img_offset, float scale, void* stream);

/* ---- single kernels, exported for unit parity tests -------------------------------------------------------------
 * Activations NHWC in the internal dtype (float32 or bfloat16 bit patterns). Weights HOST float32 in PyTorch layout;
 * these helpers pack + upload them per call (synchronous, test use only). */
/* Fused LinearAttention core (parts/mha.py:44-58 up to and including to_out.0) on x [B][n][C] normalised with
 * GroupNorm(1) from x_stats = one partial {sum, sum of squared deviations from the mean, element count, 0} per image:
 * y [B][n][C] = to_out.0(attention) + bias, and the partials of y in the same format in y_stats
 * [B][*slices_out][4] (*slices_out = partials per image: one per wave of every pixel slice). wqkv [384][C], wout [C][128]
 * HOST float32.
 * scratch: device, >= B * 4 * ceil(n / 32) * 1088 * 4 bytes. */
int dmn_test_linattn(int dtype, const void* x, const float* x_stats, int B, int n, int C, const float* gamma_host,
                     const float* beta_host, const float* wqkv_host, const float* wout_host, const float* bout_host,
                     void* y, float* y_stats, int* slices_out, void* scratch, size_t scratch_bytes, void* stream);
/* The whole Residual(PreNorm(LinearAttention)) block (utils.py:68-93, parts/mha.py:44-59): out = GroupNorm(1)(y) + x
 * behind the fused attention; fused = 1 inside the attention's second launch (partial sums exchanged within it),
 * fused = 0 as a separate pass over y; + 2: the context pass always as a launch of its own (an image that one
 * workgroup holds is otherwise one launch for the whole attention). All forms give the same bits in `out`. plan_B: batch the pixel slices are
 * planned for (0: B). Shapes outside the exchange's limits (more than 8 slices per image) return an error for fused = 1. */
int dmn_test_linattn_block(int dtype, const void* x, const float* x_stats, int B, int n, int C, const float* gamma_host,
                           const float* beta_host, const float* wqkv_host, const float* wout_host, const float* bout_host,
                           const float* out_gamma_host, const float* out_beta_host, void* y, float* y_stats, void* out,
                           void* scratch, size_t scratch_bytes, int fused, int plan_B, unsigned* giveups, void* stream);
/* Softmax attention core (parts/mha.py:21-28: scale, q k^T, row maximum, softmax, attn v) on qkv [B][n][3 * heads * 32]
 * (channels q | k | v, each head-major) -> out [B][n][heads * 32], both in the internal dtype. 64 <= n <= 256 with
 * n % 64 == 0 runs on the matrix cores, other sizes on the vector units. */
int dmn_test_attention(int dtype, const void* qkv, void* out, int B, int n, int heads, void* stream);
/* Unit-test hook: forward and backward of the training step's softmax attention core (parts/mha.py:20-29) on
 * caller tensors. adt 0 = fp32, 1 = bf16 tensors; lse = [2][B*heads][n] floats of scratch; bf16_operands != 0 and
 * 64 <= n <= 256 (n % 64 == 0) runs on the matrix cores. */
int dmn_test_attn_train(int adt, const void* qkv, const void* dout, void* out, void* dqkv, float* lse, int B, int n, int heads,
                        int bf16_operands, void* stream);

/* Diagnosis: after every launch of a forward, a checksum of the whole workspace goes to slot k of a device array. */
int dmn_test_unet_trace(DmnUnet* u, unsigned long long* slots, int max_slots);
int dmn_test_unet_trace_count(const DmnUnet* u);
int dmn_test_unet_trace_poison(DmnUnet* u, unsigned pattern);   /* fill every CU's LDS with pattern before each launch; 0: off */
int dmn_test_unet_trace_stop(DmnUnet* u, int k);   /* following forwards end after launch k; -1: run all */
const char* dmn_test_unet_trace_name(const DmnUnet* u, int k);
/* Tuning aid: every later convolution / fused linear-attention launch writes per-workgroup phase timestamps (int64,
 * 8 per workgroup; the second linear-attention kernel writes at workgroup offset 65536) to `dev`, device memory of
 * at least 2 * 65536 * 8 int64; NULL switches it off. */
int dmn_test_set_conv_debug(long long* dev);
/* One convolution launch on caller tensors. pro_stats_dev / epi_stats_dev: GroupNorm statistics as partial moments
 * [B][G][slots][4] float32 = {sum, sum of squared deviations from the partial's own mean, element count, 0} per slot
 * (what the kernels hand each other; a consumer combines the slots of a group without the cancellation of
 * sum / sum-of-squares pairs). */
int dmn_test_conv2d(int dtype, const void* in0, const void* in1, int C0, int C1, int B, int H, int W, const float* w_host,
                    const float* bias_host, int Cout, int kh, int kw, int stride, int pad, int transposed, void* out,
                    const void* res, const float* pro_gamma_host, const float* pro_beta_host, const float* pro_tbias_dev,
                    const float* pro_stats_dev, int pro_G, int pro_slots, int pro_silu, float* epi_stats_dev, int epi_G,
                    int* epi_slots_out, int cfg_tile, int cfg_bn, void* stream);
/* block2.proj of a ResnetBlock with the block's closing SiLU(GroupNorm(h)) + res (parts/convnext.py:81-86) on
 * caller-provided device tensors: fused = 1 in one launch (partial sums exchanged inside it), fused = 0 as the
 * convolution followed by the separate closing pass (h2: its intermediate). Both forms give the same bits in `out`.
 * Shapes outside the exchange's limits (more than 8 workgroups per image) return an error for fused = 1; + 2 (either
 * form) lets 32-channel images of up to 8 tiles of 32 x 16 pixels (64x64) take those tiles, as the network does. */
int dmn_test_conv_block_tail(int dtype, const void* in, int Cin, int B, int H, int W, const float* w_host,
                             const float* bias_host, int Cout, const float* pro_gamma_host, const float* pro_beta_host,
                             const float* pro_stats_dev, int pro_G, int pro_slots, const float* tail_gamma_host,
                             const float* tail_beta_host, int G, const void* res, void* h2, void* out, float* out_stats_dev,
                             int out_G, int* out_slots, int fused, int plan_B, int cfg_tile, int cfg_bn, unsigned* giveups,
                             void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DMN_H */
