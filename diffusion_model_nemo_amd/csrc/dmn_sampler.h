// Internal launch interface of the sampler update kernels (NCHW fp32 state).
#pragma once
#include "dmn_common.h"

// Per-step schedule table: one row of DMN_TABLE_COLS floats per executed step, in execution order.
//   col 0  network time label (DDPM: t; VE: sigma(t); VP: 999 t)
//   DDPM rows:  1 sqrt(1/acp)  2 sqrt(1/acp - 1)  3 posterior_mean_coef1  4 posterior_mean_coef2
//               5 posterior_log_variance_clipped  6 log(beta)  7 nonzero mask (t != 0)
//   DDIM rows:  1 sqrt(1 - a_t)  2 sqrt(a_t)  3 sqrt(a_next)  4 c1 (noise)  5 c2 (direction)
//   SDE rows:   1 a  2 b  3 c  (x_mean = a x + b net, x = x_mean + c z)
//               4 score scale (score = scale * net)  5 alpha (Langevin)  6 fixed corrector step (ALD)
//   BPD rows:   1 sqrt(a_t)  2 sqrt(1 - a_t)  3 sqrt(1/a_t)  4 sqrt(1/a_t - 1)  5 posterior_mean_coef1
//               6 posterior_mean_coef2  7 posterior_log_variance_clipped  9 log(beta)  10 [t == 0]
//   col 8  step id used to key the noise generator
#define DMN_TABLE_COLS 12

struct StepCommon {
    const float* x;       // [B][C][HW]
    const float* net;     // network output
    float* x_out;
    const float* table;
    const int* step_idx;  // device step counter, or nullptr to use `step`
    int step;
    int B, C, HW;
    unsigned long long seed;
    long long img_offset; // global index of image 0 of this batch
    const float* noise;   // optional injected N(0,1) tensor [B][C][HW]; nullptr: generate
    float* keep;          // optional history buffer [nsteps][keep_stride]
    long long keep_stride;
};
struct DdpmStepArgs {
    StepCommon c;
    int* step_next;       // optional: receives (row + 1), the counter the NEXT network evaluation reads. Must not be
                          // the location c.step_idx points to (other workgroups of this launch still read that).
    int learned_variance; // net has 2C channels: eps | v
    int pred_x0;          // objective == 'pred_x0'
    int ddim;             // generalized (DDIM) update with the DDIM row layout
};
struct AffineStepArgs {
    StepCommon c;
    float* x_mean;        // optional
    int keep_mean;        // history stores x_mean (denoise=True) instead of x
};
struct LangevinArgs {
    StepCommon c;
    float* partials;      // [B][slots][2]
    int slots;
    int inner;            // corrector inner step index (noise stream 2 + inner)
    int use_norms;        // 1: Langevin, 0: annealed Langevin dynamics (fixed step)
    float snr;
    // A batch sampled as several launch chains (each with c.B of the B_all images, the first one b_first): the step size
    // is a mean over ALL images, so the chains meet here. The norms of a chain's images go to `gran` (shared by the
    // chains) as tagged 8-byte granules, [2 buffers][B_all][slots][2]; a one-workgroup launch per chain then waits until
    // the granules of every image carry the tag of this (step, inner step), forms the step size with the arithmetic and
    // summation order of the single-chain kernel (same bits) and leaves it in `step_io` for the chain's update launch.
    // gran == nullptr: one chain, plain partials, the update launch forms the step size itself.
    unsigned long long* gran;
    float* step_io;       // one float per chain
    unsigned* giveups;    // counts waits that ran into their time limit (the handle's counter)
    int B_all, b_first;
    int n_inner;          // corrector steps per iteration (buffer parity of a meeting = its running number & 1)
};
// One element of the DDPM / improved-DDPM / DDIM update (gaussian_diffusion.py:118-123,91-101,157-167;
// learned_gaussian_diffusion.py:35-48; generalized_gaussian_diffusion.py:43-46,84-97): x, eps (and v, the variance
// channel of a learned-variance network) -> x_{t-1}, with z ~ N(0,1). `tr` is the step's table row. Shared by the
// stand-alone update kernel and the network's last kernel when it applies the update itself, so both give the same bits.
#ifdef __HIPCC__
#include "dmn_rng.h"
// Four N(0,1) values for elements 4q .. 4q+3 of image b: the injected tensor when there is one, else Philox.
__device__ __forceinline__ void load_noise(const StepCommon& c, unsigned pstep, unsigned stream, int b, int q, int per,
                                            float z[4]) {
    if (c.noise) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int e = 4 * q + r;
            z[r] = e < per ? c.noise[(size_t)b * per + e] : 0.f;
        }
    } else {
        normal4(c.seed, pstep, stream, (unsigned)(c.img_offset + b), (unsigned)q, z);
    }
}
__device__ __forceinline__ float ddpm_update_element(const DdpmStepArgs& a, const float* tr, float x, float eps, float v, float z) {
    if (a.ddim) {
        // row: 1 sqrt(1-at) 2 sqrt(at) 3 sqrt(at_next) 4 c1 5 c2. With the pred_x0 objective the reference clamps the
        // network output in place, so the direction term uses the clamped value too.
        float x0 = a.pred_x0 ? eps : (x - eps * tr[1]) / tr[2];
        x0 = fminf(fmaxf(x0, -1.0f), 1.0f);
        return tr[3] * x0 + tr[4] * z + tr[5] * (a.pred_x0 ? x0 : eps);
    }
    const float sra = tr[1], srm1 = tr[2], c1 = tr[3], c2 = tr[4], plv = tr[5], logb = tr[6], mask = tr[7];
    float x0 = a.pred_x0 ? eps : sra * x - srm1 * eps;
    x0 = fminf(fmaxf(x0, -1.0f), 1.0f);
    const float mean = c1 * x0 + c2 * x;
    float logvar = plv;
    if (a.learned_variance) {
        const float frac = (v + 1.0f) * 0.5f;
        logvar = frac * logb + (1.0f - frac) * plv;
    }
    return mean + mask * expf(0.5f * logvar) * z;
}
#endif
int launch_ddpm_step(const DdpmStepArgs& a, hipStream_t s);
int launch_affine_step(const AffineStepArgs& a, hipStream_t s);
int launch_langevin(LangevinArgs& a, hipStream_t s);
int langevin_slots(int per);
int launch_prior(float* x, int B, int per, unsigned long long seed, unsigned pstep, long long img_offset, float scale,
                 hipStream_t s);
int launch_advance(int* step_idx, hipStream_t s);
int launch_delay(int us, hipStream_t s);   // holds the stream for `us` microseconds
// *seen = 1 if the flag was raised (launch_probe_set on another stream) within `us` microseconds of the waiter's start
int launch_probe_wait(unsigned* flag, unsigned* seen, int us, hipStream_t s);
int launch_probe_set(unsigned* flag, hipStream_t s);

// Bits-per-dimension evaluation (models/abstract_diffusion_model.py:137-197), one timestep per table row.
struct BpdArgs {
    StepCommon c;          // x = x_start, x_out = x_t buffer, net = network output at (x_t, t); noise stream 3
    float* partials;       // [B][slots]
    float* terms;          // [B][terms_stride], column = the row's timestep (table col 0)
    int terms_stride;
    int slots;
    int learned_variance;
};
int bpd_slots(int per);
int launch_bpd_qsample(const BpdArgs& a, hipStream_t s);   // x_t = sqrt(a_t) x_0 + sqrt(1 - a_t) z
int launch_bpd_term(BpdArgs& a, hipStream_t s);            // per-sample KL / decoder NLL in bits -> terms
int launch_bpd_prior(const float* x0, float* prior, int B, int per, float sqrt_acp_T, float log_1m_acp_T, hipStream_t s);
