// Measured ceilings of the device the library runs on: a streaming copy (HBM read + write) and register-resident
// matrix-core loops. They put the nominal 8 TB/s / 2.5 PFLOP/s figures next to what this part sustains, so that a
// roofline fraction can be quoted against either. No counterpart in the reference: measurement hooks used by bench.py.
#include <cstring>
#include "dmn_common.h"

namespace {

// Every thread moves 16-byte pieces `stride` apart: consecutive lanes touch consecutive pieces (fully coalesced),
// four loads are in flight per thread before the first store.
__global__ __launch_bounds__(256) void stream_copy_kernel(f32x4* __restrict__ dst, const f32x4* __restrict__ src, size_t n16) {
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < n16; i += 4 * stride) {
        const f32x4 a = __builtin_nontemporal_load(src + i);
        const f32x4 b = __builtin_nontemporal_load(src + i + stride);
        const f32x4 c = __builtin_nontemporal_load(src + i + 2 * stride);
        const f32x4 d = __builtin_nontemporal_load(src + i + 3 * stride);
        __builtin_nontemporal_store(a, dst + i);
        __builtin_nontemporal_store(b, dst + i + stride);
        __builtin_nontemporal_store(c, dst + i + 2 * stride);
        __builtin_nontemporal_store(d, dst + i + 3 * stride);
    }
    for (; i < n16; i += stride) dst[i] = src[i];
}

// Read-only sweep (sum into one value per thread): the ceiling for kernels whose output is small.
__global__ __launch_bounds__(256) void stream_read_kernel(float* __restrict__ sink, const f32x4* __restrict__ src, size_t n16) {
    const size_t stride = (size_t)gridDim.x * 256;
    f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < n16; i += 4 * stride) {
        const f32x4 a = __builtin_nontemporal_load(src + i);
        const f32x4 b = __builtin_nontemporal_load(src + i + stride);
        const f32x4 c = __builtin_nontemporal_load(src + i + 2 * stride);
        const f32x4 d = __builtin_nontemporal_load(src + i + 3 * stride);
        s += (a + b) + (c + d);
    }
    for (; i < n16; i += stride) s += src[i];
    const float t = (s[0] + s[1]) + (s[2] + s[3]);
    if (t == 1.2345e30f) sink[0] = t;   // keeps the loads alive; never true for finite inputs of ordinary size
}

// Operands live in registers, eight independent accumulators per wave; operand values are small, non-trivial and
// differ per lane (all-zero operands let the part hold a higher clock than real data does).
template <int KIND>
__global__ __launch_bounds__(256) void mfma_loop_kernel(float* __restrict__ sink, int n_outer) {
    const int lane = threadIdx.x & 63;
    f32x4 acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (KIND == 0) {
        bf16x8 a, b;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            a[i] = (__bf16)(0.01f * (float)((lane * 7 + i * 3) % 23 - 11));
            b[i] = (__bf16)(0.02f * (float)((lane * 5 + i * 11) % 19 - 9));
        }
        for (int it = 0; it < n_outer; ++it) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[j], 0, 0, 0);
        }
    } else {
        const float a = 0.01f * (float)((lane * 7) % 23 - 11), b = 0.02f * (float)((lane * 5) % 19 - 9);
        for (int it = 0; it < n_outer; ++it) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j], 0, 0, 0);
        }
    }
    f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 8; ++j) s += acc[j];
    sink[(size_t)blockIdx.x * 256 + threadIdx.x] = (s[0] + s[1]) + (s[2] + s[3]);
}

template <class F>
int timed(hipStream_t s, int iters, float* avg_ms, const F& launch) {
    hipEvent_t e0, e1;
    DMN_HIP_CHECK(hipEventCreate(&e0));
    DMN_HIP_CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 2 + iters; ++i) {
        if (i == 2) (void)hipEventRecord(e0, s);
        launch();
    }
    (void)hipEventRecord(e1, s);
    const hipError_t err = hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    DMN_HIP_CHECK(err);
    DMN_HIP_CHECK(hipGetLastError());
    *avg_ms = ms / iters;
    return DMN_OK;
}

// one store per workgroup: what a dependent launch costs when it does nothing
__global__ void __launch_bounds__(256) touch_kernel(float* sink) {
    if (threadIdx.x == 0) sink[blockIdx.x] = (float)blockIdx.x;
}

// the same with 512 bytes of kernel arguments, of which the LAST word is needed before the store: what the size of an
// argument block adds to a launch
struct BigArgs {
    float* sink;
    int pad[125];
    int last;
};
__global__ void __launch_bounds__(256) touch_big_kernel(const BigArgs a) {
    if (threadIdx.x == 0) a.sink[blockIdx.x] = (float)(blockIdx.x + a.last);
}

}  // namespace

// The floor under a launch chain: `launches` dependent launches of a kernel that does nothing (wgs workgroups of 256
// threads, one store each), captured into one graph and replayed `iters` times on `stream` (not the legacy stream);
// us_per_launch receives the average distance between two launches. sink: >= wgs floats.
extern "C" int dmn_bench_launch_floor(int launches, int wgs, int iters, float* sink, void* stream, float* us_per_launch) {
    DMN_REQUIRE(sink && us_per_launch && stream && launches > 0 && launches <= 4096 && wgs > 0 && wgs <= 65535 && iters > 0, DMN_ERR_INVALID,
                "dmn_bench_launch_floor: bad argument");
    hipStream_t s = (hipStream_t)stream;
    hipGraph_t g = nullptr;
    hipGraphExec_t ge = nullptr;
    DMN_HIP_CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed));
    BigArgs big;
    memset(&big, 0, sizeof(big));
    big.sink = sink;
    const bool use_big = getenv("DMN_FLOOR_BIG_ARGS") != nullptr;   // 512-byte argument block instead of one pointer
    for (int i = 0; i < launches; ++i) {
        big.last = i;
        if (use_big) touch_big_kernel<<<wgs, 256, 0, s>>>(big);
        else touch_kernel<<<wgs, 256, 0, s>>>(sink);
    }
    DMN_HIP_CHECK(hipStreamEndCapture(s, &g));
    DMN_HIP_CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    (void)hipGraphDestroy(g);
    float ms = 0.f;
    int rc = timed(s, iters, &ms, [&] { (void)hipGraphLaunch(ge, s); });
    (void)hipGraphExecDestroy(ge);
    *us_per_launch = ms * 1e3f / launches;
    return rc;
}

// mode 0: copy `bytes` from src to dst (traffic 2 x bytes per launch); mode 1: read `bytes` from src (dst: >= 4 bytes).
extern "C" int dmn_bench_stream(int mode, void* dst, const void* src, size_t bytes, int iters, void* stream, float* avg_ms) {
    DMN_REQUIRE(dst && src && avg_ms && iters > 0 && bytes >= 16 && bytes % 16 == 0 && (mode == 0 || mode == 1), DMN_ERR_INVALID,
                "dmn_bench_stream: bad argument");
    DMN_REQUIRE(((uintptr_t)dst | (uintptr_t)src) % 16 == 0, DMN_ERR_INVALID, "dmn_bench_stream: buffers must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    const size_t n16 = bytes / 16;
    const int grid = (int)std::min<size_t>((n16 + 255) / 256, 256 * 16);
    return timed(s, iters, avg_ms, [&] {
        if (mode == 0)
            stream_copy_kernel<<<grid, 256, 0, s>>>((f32x4*)dst, (const f32x4*)src, n16);
        else
            stream_read_kernel<<<grid, 256, 0, s>>>((float*)dst, (const f32x4*)src, n16);
    });
}

// kind 0: v_mfma_f32_16x16x32_bf16, kind 1: v_mfma_f32_16x16x4_f32. Launches 256 x `wgs_per_cu` workgroups of four
// waves, each wave issuing 32 x n_outer matrix instructions. sink: >= 256 * wgs_per_cu * 256 floats. flops receives
// the floating-point operations of one launch.
extern "C" int dmn_bench_mfma(int kind, int wgs_per_cu, int n_outer, int iters, float* sink, void* stream, float* avg_ms,
                              double* flops) {
    DMN_REQUIRE(sink && avg_ms && iters > 0 && n_outer > 0 && wgs_per_cu > 0 && wgs_per_cu <= 8 && (kind == 0 || kind == 1),
                DMN_ERR_INVALID, "dmn_bench_mfma: bad argument");
    hipStream_t s = (hipStream_t)stream;
    const int grid = 256 * wgs_per_cu;
    if (flops) *flops = (double)grid * 4 * 32.0 * n_outer * 2.0 * 16 * 16 * (kind == 0 ? 32 : 4);
    return timed(s, iters, avg_ms, [&] {
        if (kind == 0)
            mfma_loop_kernel<0><<<grid, 256, 0, s>>>(sink, n_outer);
        else
            mfma_loop_kernel<1><<<grid, 256, 0, s>>>(sink, n_outer);
    });
}
