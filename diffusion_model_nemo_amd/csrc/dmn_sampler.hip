// Per-step state updates of the samplers, on NCHW fp32 images (the reference's layout):
//   * DDPM / improved-DDPM ancestral step
//       diffusion_model_nemo/modules/gaussian_diffusion.py:118-123,91-101,157-167
//       diffusion_model_nemo/modules/learned_gaussian_diffusion.py:35-48
//   * score-SDE predictor step as an affine update  x_mean = a x + b net ; x = x_mean + c z
//       sde_predictors/{euler_maruyama,reverse_diffusion,ancestral_sampling}_predictor.py
//   * Langevin corrector (batch-mean of per-sample norms) sde_correctors/langevin_corrector.py:26-33
// All per-step scalars come from a device table indexed by a device-resident step counter, so one
// captured graph replays for every step. Noise is Philox (dmn_rng.h) or an injected tensor.
#include <string.h>

#include <mutex>

#include "../../include/dmn.h"
#include "dmn_handoff.h"
#include "dmn_sampler.h"

namespace {

// grid (ceil(per/4/256), B)
__global__ void __launch_bounds__(256) ddpm_step_kernel(const DdpmStepArgs a) {
    const StepCommon& c = a.c;
    const int b = blockIdx.y;
    const int per = c.C * c.HW;
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (4 * q >= per) return;
    const int row = c.step_idx ? *c.step_idx : c.step;
    if (a.step_next && q == 0 && b == 0) *a.step_next = row + 1;
    const float* tr = c.table + (size_t)row * DMN_TABLE_COLS;
    float z[4];
    load_noise(c, (unsigned)tr[8], 1u, b, q, per, z);
    const int mo_c = a.learned_variance ? 2 * c.C : c.C;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int e = 4 * q + r;
        if (e >= per) break;
        const size_t xi = (size_t)b * per + e;
        const size_t mi = (size_t)b * mo_c * c.HW + e;
        const float v = a.learned_variance ? c.net[mi + (size_t)c.C * c.HW] : 0.f;
        const float xn = ddpm_update_element(a, tr, c.x[xi], c.net[mi], v, z[r]);
        c.x_out[xi] = xn;
        if (c.keep) c.keep[(size_t)row * c.keep_stride + xi] = xn;
    }
}

__global__ void __launch_bounds__(256) affine_step_kernel(const AffineStepArgs a) {
    const StepCommon& c = a.c;
    const int b = blockIdx.y;
    const int per = c.C * c.HW;
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (4 * q >= per) return;
    const int row = c.step_idx ? *c.step_idx : c.step;
    const float* tr = c.table + (size_t)row * DMN_TABLE_COLS;
    const float ca = tr[1], cb = tr[2], cc = tr[3];
    float z[4];
    load_noise(c, (unsigned)tr[8], 1u, b, q, per, z);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int e = 4 * q + r;
        if (e >= per) break;
        const size_t xi = (size_t)b * per + e;
        const float mean = ca * c.x[xi] + cb * c.net[xi];
        c.x_out[xi] = mean + cc * z[r];
        if (a.x_mean) a.x_mean[xi] = mean;
        if (c.keep) c.keep[(size_t)row * c.keep_stride + xi] = a.keep_mean ? mean : mean + cc * z[r];
    }
}

// Tag and buffer of the meeting of the launch chains at (table row, corrector inner step): tags grow along a loop (the
// area is zeroed when the loop starts), and consecutive meetings alternate between two buffers, so a chain that has
// passed a meeting can publish for the next one while the other chain still reads this one's granules (it cannot get
// two ahead: the next meeting needs the other chain's granules too).
__device__ __forceinline__ unsigned langevin_tag(int row, int inner) { return (unsigned)row * 65u + (unsigned)inner + 1u; }
__device__ __forceinline__ int langevin_buffer(int row, int inner, int n_inner) { return (row * n_inner + inner) & 1; }

// Langevin pass 1: per-sample partial sums of net^2 and z^2. grid (slots, B); partials [B][slots][2]
__global__ void __launch_bounds__(256) langevin_norms_kernel(const LangevinArgs a) {
    __shared__ float sred[8];
    const StepCommon& c = a.c;
    const int b = blockIdx.y;
    const int per = c.C * c.HW;
    const int row = c.step_idx ? *c.step_idx : c.step;
    const float* tr = c.table + (size_t)row * DMN_TABLE_COLS;
    float sg = 0.f, sz = 0.f;
    const int nq = (per + 3) / 4;
    for (int q = blockIdx.x * 256 + threadIdx.x; q < nq; q += gridDim.x * 256) {
        float z[4];
        load_noise(c, (unsigned)tr[8], 2u + a.inner, b, q, per, z);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int e = 4 * q + r;
            if (e >= per) break;
            const float g = c.net[(size_t)b * per + e];
            sg = fmaf(g, g, sg);
            sz = fmaf(z[r], z[r], sz);
        }
    }
    sg = wave_sum(sg);
    sz = wave_sum(sz);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) { sred[2 * wave] = sg; sred[2 * wave + 1] = sz; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float tg = 0.f, tz = 0.f;
        for (int w = 0; w < 4; ++w) { tg += sred[2 * w]; tz += sred[2 * w + 1]; }
        if (a.gran) {   // several chains: the partial travels as tagged granules to the meeting of the chains
            const unsigned tag = langevin_tag(row, a.inner);
            unsigned long long* dst = a.gran + ((((size_t)langevin_buffer(row, a.inner, a.n_inner) * a.B_all + a.b_first + b) * gridDim.x) + blockIdx.x) * 2;
            granule_store(dst, tag, tg);
            granule_store(dst + 1, tag, tz);
        } else {
            float* dst = a.partials + ((size_t)b * gridDim.x + blockIdx.x) * 2;
            dst[0] = tg;
            dst[1] = tz;
        }
    }
}

// The meeting of the launch chains of one batch: grid (1), 64 threads. Lane i takes images i, i + 64, ...: it waits for
// the image's granules, sums its slots in slot order and takes the roots; thread 0 then adds the images in image order
// -- the arithmetic and order of the single-chain update kernel below, so the step size has the same bits.
__global__ void __launch_bounds__(64) langevin_meet_kernel(const LangevinArgs a) {
    extern __shared__ float sN[];   // [B_all][2]
    const StepCommon& c = a.c;
    const int row = c.step_idx ? *c.step_idx : c.step;
    const float* tr = c.table + (size_t)row * DMN_TABLE_COLS;
    const float sscale = tr[4];
    const unsigned tag = langevin_tag(row, a.inner);
    const unsigned long long* base = a.gran + (size_t)langevin_buffer(row, a.inner, a.n_inner) * a.B_all * a.slots * 2;
    const long long t0 = wall_clock64();
    bool gave_up = false;
    for (int i = threadIdx.x; i < a.B_all; i += 64) {
        float tg = 0.f, tz = 0.f;
        for (int s = 0; s < a.slots; ++s) {
            const unsigned long long* p = base + ((size_t)i * a.slots + s) * 2;
            unsigned long long g, z;
            for (;;) {
                g = granule_load(p);
                z = granule_load(p + 1);
                if (((unsigned)(g >> 32) == tag && (unsigned)(z >> 32) == tag) || gave_up) break;
                if (wall_clock64() - t0 > kHandoffSpinTicks) {
                    gave_up = true;
                    if (a.giveups) {
                        atomicAdd(a.giveups, 1u);
                        // diagnosis, as the in-launch exchanges leave it: the first lane that misses a granule describes it
                        if (atomicCAS(a.giveups + 1, 0u, 1u) == 0u) {
                            unsigned* d = a.giveups + 2;
                            d[0] = tag;
                            d[1] = (unsigned)(g >> 32);
                            d[2] = (unsigned)(z >> 32);
                            d[3] = (unsigned)i;
                            d[4] = (unsigned)a.b_first;
                            d[5] = (unsigned)s;
                            d[6] = (unsigned)a.slots;
                            d[7] = (unsigned)row;
                            d[8] = (unsigned)a.inner;
                            d[9] = (unsigned)a.n_inner;
                        }
                    }
                    break;
                }
                __builtin_amdgcn_s_sleep(32);
            }
            tg += __uint_as_float((unsigned)g);
            tz += __uint_as_float((unsigned)z);
        }
        sN[2 * i] = sqrtf(tg) * fabsf(sscale);
        sN[2 * i + 1] = sqrtf(tz);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float gn = 0.f, zn = 0.f;
        for (int i = 0; i < a.B_all; ++i) {
            gn += sN[2 * i];
            zn += sN[2 * i + 1];
        }
        gn /= (float)a.B_all;
        zn /= (float)a.B_all;
        const float ratio = a.snr * zn / gn;
        *a.step_io = ratio * ratio * 2.0f * tr[5];
    }
}

// Langevin pass 2: step = k * (mean_b|z| / mean_b|score|)^2 ; x_mean = x + step*score ; x = x_mean + sqrt(2 step) z
// Annealed Langevin (fixed step from the table) uses the same kernel with use_norms = 0.
__global__ void __launch_bounds__(256) langevin_apply_kernel(const LangevinArgs a) {
    __shared__ float sStep;
    const StepCommon& c = a.c;
    const int b = blockIdx.y;
    const int per = c.C * c.HW;
    const int row = c.step_idx ? *c.step_idx : c.step;
    const float* tr = c.table + (size_t)row * DMN_TABLE_COLS;
    const float sscale = tr[4];   // score = sscale * net
    if (threadIdx.x == 0) {
        float step;
        if (a.use_norms && a.gran) {
            step = *a.step_io;   // formed by this chain's meeting launch
        } else if (a.use_norms) {
            float gn = 0.f, zn = 0.f;
            for (int i = 0; i < c.B; ++i) {
                float tg = 0.f, tz = 0.f;
                for (int s = 0; s < a.slots; ++s) {
                    tg += a.partials[((size_t)i * a.slots + s) * 2];
                    tz += a.partials[((size_t)i * a.slots + s) * 2 + 1];
                }
                gn += sqrtf(tg) * fabsf(sscale);
                zn += sqrtf(tz);
            }
            gn /= (float)c.B;
            zn /= (float)c.B;
            const float ratio = a.snr * zn / gn;
            step = ratio * ratio * 2.0f * tr[5];
        } else {
            step = tr[6];
        }
        sStep = step;
    }
    __syncthreads();
    const float step = sStep;
    const float nz = sqrtf(step * 2.0f);
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (4 * q >= per) return;
    float z[4];
    load_noise(c, (unsigned)tr[8], 2u + a.inner, b, q, per, z);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int e = 4 * q + r;
        if (e >= per) break;
        const size_t xi = (size_t)b * per + e;
        const float mean = c.x[xi] + step * (sscale * c.net[xi]);
        const float xn = mean + nz * z[r];
        c.x_out[xi] = xn;
        if (c.keep) c.keep[(size_t)row * c.keep_stride + xi] = xn;   // sampler without a predictor: the kept image
    }
}

// x = scale * N(0,1) (prior draw). grid (ceil(per/4/256), B)
__global__ void __launch_bounds__(256) prior_kernel(float* x, int B, int per, unsigned long long seed, unsigned pstep,
                                                     long long img_offset, float scale) {
    const int b = blockIdx.y;
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (4 * q >= per) return;
    float z[4];
    normal4(seed, pstep, 0u, (unsigned)(img_offset + b), (unsigned)q, z);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int e = 4 * q + r;
        if (e < per) x[(size_t)b * per + e] = scale * z[r];
    }
}

__global__ void advance_kernel(int* step_idx) { *step_idx += 1; }

// ---- bits per dimension ---------------------------------------------------------------------------------------
// q_sample (gaussian_diffusion.py:104-116) with the generator's stream 3. grid (ceil(per/4/256), B)
__global__ void __launch_bounds__(256) bpd_qsample_kernel(const BpdArgs a) {
    const StepCommon& c = a.c;
    const int b = blockIdx.y, per = c.C * c.HW;
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (4 * q >= per) return;
    const float* tr = c.table + (size_t)(c.step_idx ? *c.step_idx : c.step) * DMN_TABLE_COLS;
    float z[4];
    load_noise(c, (unsigned)tr[8], 3u, b, q, per, z);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int e = 4 * q + r;
        if (e < per) c.x_out[(size_t)b * per + e] = tr[1] * c.x[(size_t)b * per + e] + tr[2] * z[r];
    }
}

__device__ __forceinline__ float approx_normal_cdf(float v) {   // utils.py:37-38
    return 0.5f * (1.0f + tanhf(0.7978845608028654f * (v + 0.044715f * (v * v * v))));
}

// Per-element variational-bound term (variational_bound_loss.py:31-52, utils.py:28-56) summed per workgroup.
// grid (slots, B); partials [B][slots]
__global__ void __launch_bounds__(256) bpd_term_kernel(const BpdArgs a) {
    __shared__ float sred[4];
    const StepCommon& c = a.c;
    const int b = blockIdx.y, per = c.C * c.HW;
    const float* tr = c.table + (size_t)(c.step_idx ? *c.step_idx : c.step) * DMN_TABLE_COLS;
    const float sra = tr[3], srm1 = tr[4], c1 = tr[5], c2 = tr[6], plv = tr[7], logb = tr[9];
    const bool t0 = tr[10] != 0.f;
    const int mo_c = a.learned_variance ? 2 * c.C : c.C;
    float sum = 0.f;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < per; e += gridDim.x * 256) {
        const float x0 = c.x[(size_t)b * per + e];
        const float xt = c.x_out[(size_t)b * per + e];
        const float eps = c.net[(size_t)b * mo_c * c.HW + e];
        float mlv = plv;
        if (a.learned_variance) {
            const float frac = (c.net[(size_t)b * mo_c * c.HW + (size_t)c.C * c.HW + e] + 1.0f) * 0.5f;
            mlv = frac * logb + (1.0f - frac) * plv;
        }
        const float true_mean = c1 * x0 + c2 * xt;
        const float px0 = fminf(fmaxf(sra * xt - srm1 * eps, -1.0f), 1.0f);
        const float model_mean = c1 * px0 + c2 * xt;
        float term;
        if (!t0) {
            const float d = true_mean - model_mean;
            term = 0.5f * (-1.0f + mlv - plv + expf(plv - mlv) + d * d * expf(-mlv));
        } else {
            const float cen = x0 - model_mean;
            const float inv = expf(-0.5f * mlv);
            const float cp = approx_normal_cdf(inv * (cen + 1.0f / 255.0f));
            const float cm = approx_normal_cdf(inv * (cen - 1.0f / 255.0f));
            const float lp = x0 < -0.999f ? logf(fmaxf(cp, 1e-12f))
                             : x0 > 0.999f ? logf(fmaxf(1.0f - cm, 1e-12f)) : logf(fmaxf(cp - cm, 1e-12f));
            term = -lp;
        }
        sum += term;
    }
    sum = wave_sum(sum);
    if ((threadIdx.x & 63) == 0) sred[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) a.partials[(size_t)b * a.slots + blockIdx.x] = (sred[0] + sred[1]) + (sred[2] + sred[3]);
}

// terms[b][t] = sum of the slots / (elements * ln 2). grid (ceil(B/256))
__global__ void __launch_bounds__(256) bpd_reduce_kernel(const BpdArgs a) {
    const StepCommon& c = a.c;
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= c.B) return;
    const float* tr = c.table + (size_t)(c.step_idx ? *c.step_idx : c.step) * DMN_TABLE_COLS;
    float s = 0.f;
    for (int i = 0; i < a.slots; ++i) s += a.partials[(size_t)b * a.slots + i];
    a.terms[(size_t)b * a.terms_stride + (int)tr[0]] = s / (float)(c.C * c.HW) * 1.4426950408889634f;
}

// prior_bpd[b] = mean_e KL(N(sqrt(a_T) x0, 1 - a_T) || N(0, 1)) / ln 2 (abstract_diffusion_model.py:181-184). grid (B)
__global__ void __launch_bounds__(256) bpd_prior_kernel(const float* x0, float* prior, int per, float m, float lv) {
    __shared__ float sred[4];
    const int b = blockIdx.x;
    float sum = 0.f;
    for (int e = threadIdx.x; e < per; e += 256) {
        const float mu = m * x0[(size_t)b * per + e];
        sum += 0.5f * (-1.0f + 0.0f - lv + expf(lv - 0.0f) + mu * mu * expf(-0.0f));
    }
    sum = wave_sum(sum);
    if ((threadIdx.x & 63) == 0) sred[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) prior[b] = ((sred[0] + sred[1]) + (sred[2] + sred[3])) / (float)per * 1.4426950408889634f;
}

}  // namespace

static dim3 step_grid(const StepCommon& c) { return dim3(((c.C * c.HW + 3) / 4 + 255) / 256, c.B); }

int launch_ddpm_step(const DdpmStepArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(ddpm_step_kernel, step_grid(a.c), dim3(256), 0, s, a);
    DMN_HIP_CHECK(hipGetLastError());
    return DMN_OK;
}
int launch_affine_step(const AffineStepArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(affine_step_kernel, step_grid(a.c), dim3(256), 0, s, a);
    DMN_HIP_CHECK(hipGetLastError());
    return DMN_OK;
}
int langevin_slots(int per) {
    int s = ((per + 3) / 4 + 255) / 256;
    return s > 16 ? 16 : s;
}
int launch_langevin(LangevinArgs& a, hipStream_t s) {
    const int per = a.c.C * a.c.HW;
    a.slots = langevin_slots(per);
    if (a.use_norms) {
        hipLaunchKernelGGL(langevin_norms_kernel, dim3(a.slots, a.c.B), dim3(256), 0, s, a);
        DMN_HIP_CHECK(hipGetLastError());
        if (a.gran) {
            DMN_REQUIRE(a.step_io && a.B_all >= a.c.B && a.b_first >= 0 && a.b_first + a.c.B <= a.B_all && a.n_inner >= 1 &&
                            a.inner < a.n_inner && a.B_all <= 4096,
                        DMN_ERR_INVALID, "Langevin corrector: inconsistent description of the launch chains");
            hipLaunchKernelGGL(langevin_meet_kernel, dim3(1), dim3(64), (size_t)a.B_all * 2 * sizeof(float), s, a);
            DMN_HIP_CHECK(hipGetLastError());
        }
    }
    hipLaunchKernelGGL(langevin_apply_kernel, step_grid(a.c), dim3(256), 0, s, a);
    DMN_HIP_CHECK(hipGetLastError());
    return DMN_OK;
}
int launch_prior(float* x, int B, int per, unsigned long long seed, unsigned pstep, long long img_offset, float scale,
                 hipStream_t s) {
    hipLaunchKernelGGL(prior_kernel, dim3(((per + 3) / 4 + 255) / 256, B), dim3(256), 0, s, x, B, per, seed, pstep,
                       img_offset, scale);
    DMN_HIP_CHECK(hipGetLastError());
    return DMN_OK;
}
int bpd_slots(int per) {
    int s = (per + 1023) / 1024;
    return s > 16 ? 16 : s;
}
int launch_bpd_qsample(const BpdArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(bpd_qsample_kernel, step_grid(a.c), dim3(256), 0, s, a);
    DMN_HIP_CHECK(hipGetLastError());
    return DMN_OK;
}
int launch_bpd_term(BpdArgs& a, hipStream_t s) {
    a.slots = bpd_slots(a.c.C * a.c.HW);
    hipLaunchKernelGGL(bpd_term_kernel, dim3(a.slots, a.c.B), dim3(256), 0, s, a);
    DMN_HIP_CHECK(hipGetLastError());
    hipLaunchKernelGGL(bpd_reduce_kernel, dim3((a.c.B + 255) / 256), dim3(256), 0, s, a);
    DMN_HIP_CHECK(hipGetLastError());
    return DMN_OK;
}
int launch_bpd_prior(const float* x0, float* prior, int B, int per, float sqrt_acp_T, float log_1m_acp_T, hipStream_t s) {
    hipLaunchKernelGGL(bpd_prior_kernel, dim3(B), dim3(256), 0, s, x0, prior, per, sqrt_acp_T, log_1m_acp_T);
    DMN_HIP_CHECK(hipGetLastError());
    return DMN_OK;
}
int launch_advance(int* step_idx, hipStream_t s) {
    hipLaunchKernelGGL(advance_kernel, dim3(1), dim3(1), 0, s, step_idx);
    DMN_HIP_CHECK(hipGetLastError());
    return DMN_OK;
}

// One wave that holds its stream for `us` microseconds of the 100 MHz wall clock (at most 5 ms; the loop also ends after
// a fixed number of sleeps, whatever the clock does).
__global__ void delay_kernel(unsigned ticks) {
    const unsigned long long t0 = wall_clock64();
    for (int i = 0; i < 200000 && wall_clock64() - t0 < ticks; ++i) __builtin_amdgcn_s_sleep(32);
}
// Do two streams run side by side? `wait` (on one stream) spins until `set` (on the other) has raised the flag or `us`
// microseconds have passed, and reports which: streams that share a hardware queue run their kernels one behind the
// other, so the flag cannot rise while the waiter is still on the queue.
__global__ void probe_wait_kernel(unsigned* flag, unsigned* seen, unsigned ticks) {
    if (threadIdx.x) return;
    const unsigned long long t0 = wall_clock64();
    unsigned v = 0;
    for (int i = 0; i < 2000000; ++i) {
        v = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (v || wall_clock64() - t0 > ticks) break;
        __builtin_amdgcn_s_sleep(16);
    }
    *seen = v;
}
__global__ void probe_set_kernel(unsigned* flag) {
    if (threadIdx.x == 0) __hip_atomic_store(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
int launch_probe_wait(unsigned* flag, unsigned* seen, int us, hipStream_t s) {
    hipLaunchKernelGGL(probe_wait_kernel, dim3(1), dim3(64), 0, s, flag, seen, (unsigned)us * 100u);
    DMN_HIP_CHECK(hipGetLastError());
    return DMN_OK;
}
int launch_probe_set(unsigned* flag, hipStream_t s) {
    hipLaunchKernelGGL(probe_set_kernel, dim3(1), dim3(64), 0, s, flag);
    DMN_HIP_CHECK(hipGetLastError());
    return DMN_OK;
}

int launch_delay(int us, hipStream_t s) {
    if (us <= 0) return DMN_OK;
    if (us > 5000) us = 5000;
    hipLaunchKernelGGL(delay_kernel, dim3(1), dim3(64), 0, s, (unsigned)us * 100u);
    DMN_HIP_CHECK(hipGetLastError());
    return DMN_OK;
}

// ---- Langevin corrector step for callers that cannot allocate -------------------------------------------------------
// Same kernels as dmn_sde_corrector_step; the per-sample partial sums live in a device array of the library instead
// of a caller's workspace. `scratch` is a host-side stand-in for the array: its device address is resolved once per
// device with hipGetSymbolAddress and only that address is ever handed to a kernel.
namespace {
constexpr size_t kStaticScratchBytes = 64 * 1024;
constexpr int kMaxDevices = 64;
}  // namespace
static __device__ __align__(256) unsigned char scratch[kStaticScratchBytes];
static float* g_scratch_ptr[kMaxDevices] = {};
static std::mutex g_scratch_mu;

extern "C" int dmn_sde_corrector_step_static(const float* table, int row, const float* x, const float* model_out,
                                             const float* noise, float* x_out, int B, int C, int H, int W, float snr,
                                             int inner, int use_norms, uint64_t seed, int64_t img_offset, void* stream) {
    DMN_REQUIRE(table && x && model_out && x_out && row >= 0 && B > 0 && C > 0 && H > 0 && W > 0, DMN_ERR_INVALID,
                "dmn_sde_corrector_step_static: bad argument");
    const size_t need = (size_t)B * langevin_slots(C * H * W) * 2 * sizeof(float);
    DMN_REQUIRE(need <= kStaticScratchBytes, DMN_ERR_LIMIT,
                "dmn_sde_corrector_step_static: %d images of %d elements need %zu bytes of partial sums, the built-in "
                "scratch holds %zu (use dmn_sde_corrector_step with a workspace)", B, C * H * W, need, kStaticScratchBytes);
    int dev = -1;
    DMN_HIP_CHECK(hipGetDevice(&dev));
    DMN_REQUIRE(dev >= 0 && dev < kMaxDevices, DMN_ERR_LIMIT, "dmn_sde_corrector_step_static: device ordinal %d", dev);
    float* part;
    {
        std::lock_guard<std::mutex> lk(g_scratch_mu);
        if (!g_scratch_ptr[dev]) {   // every device holds its own copy of the array at its own address
            void* p = nullptr;
            DMN_HIP_CHECK(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
            DMN_REQUIRE(p != nullptr, DMN_ERR_HIP, "dmn_sde_corrector_step_static: no device address for the scratch array");
            g_scratch_ptr[dev] = static_cast<float*>(p);
        }
        part = g_scratch_ptr[dev];
    }
    LangevinArgs a;
    memset(&a, 0, sizeof(a));
    a.c.x = x;
    a.c.net = model_out;
    a.c.x_out = x_out;
    a.c.table = table;
    a.c.step = row;
    a.c.B = B;
    a.c.C = C;
    a.c.HW = H * W;
    a.c.seed = seed;
    a.c.img_offset = img_offset;
    a.c.noise = noise;
    a.c.keep_stride = (long long)B * C * H * W;
    a.partials = part;   // every slot the second kernel reads is written by the first: nothing to clear
    a.inner = inner;
    a.use_norms = use_norms;
    a.snr = snr;
    return launch_langevin(a, (hipStream_t)stream);
}
