// Hand-off of GroupNorm partial sums between the workgroups of ONE launch (gfx950: eight XCDs with private L2s,
// per-CU L1s that other CUs' stores never refresh).
//
// A kernel whose epilogue needs statistics of its own output (the closing SiLU(GroupNorm(h)) + res of a ResnetBlock,
// parts/convnext.py:85-86, or the GroupNorm(1) + residual behind a linear attention, parts/mha.py:41-42) would
// otherwise end at a kernel boundary and hand its tile to a second pass. Here every workgroup of an image publishes its
// partial moments {sum, squared deviations, count} (dmn_norm.h) as 8-byte GRANULES {tag = epoch, value = float bits}, each written by one agent-scope
// relaxed atomic store (a write-through store that every XCD observes) and read by agent-scope relaxed atomic loads
// (which bypass the reader's L1): the data is its own flag, so no fence, no counter and no ordering between granules
// is needed. A reader polls its granules until all carry the current epoch.
//
// Epoch: one device word per launch chain, incremented by the first kernel of every network evaluation. The granule
// area belongs to the handle (never to the caller's workspace), is zeroed when allocated and only ever written with
// epochs that grow, so a tag equal to the current epoch can only have been written by the current evaluation; every
// fused launch of an evaluation owns a distinct range of the area.
//
// Progress: a waiting workgroup keeps its place on the CU, so a wait can only end if the workgroups it waits for get a
// place too. One launch alone is safe (the workgroups of an image are consecutive in launch order and publish BEFORE
// they wait), but two such launches on two streams can starve each other once waiters of both fill the machine
// (measured: about one stalled step in 3000 with two 512-workgroup launches in flight). Therefore every launch that
// exchanges is small enough for ALL its workgroups to be resident beside all those of a second such launch: at most
// kHandoffMaxWgsPerLaunch workgroups, each needing at most half a CU: 80 KiB of LDS (checked by the launchers) and 256
// of a SIMD's 512 registers per wave (the exchanging kernels are compiled with that bound -- __launch_bounds__(256, 2)
// -- and the build reads the register counts of the finished library back: isa_check.py). Hosts split the batch over several launches to stay below that, and leave shapes with more than
// kHandoffMaxWgsPerImage workgroups per image to the multi-launch forms. More than two exchanging launches at once
// (several handles sampling concurrently on one GPU) are outside this guarantee; the spins are bounded all the same
// (kHandoffSpinTicks of the 100 MHz clock), a give-up is counted in the area's header and surfaces as an error at the
// next host check, never as a hang.
#pragma once
#include "dmn_common.h"

typedef __attribute__((address_space(1))) unsigned long long dmn_gu64;

constexpr int kHandoffHeaderBytes = 256;           // word 0: epoch, word 1: spins given up, words 2..12: diagnosis of the first
constexpr long long kHandoffSpinTicks = 50000000;  // 0.5 s of the 100 MHz wall clock
constexpr int kHandoffMaxSlots = 64;               // partial sums per (image, group): one per lane of a wave
constexpr int kHandoffMaxGroups = 32;
constexpr int kHandoffMaxWgsPerLaunch = 256;       // one per CU: two such launches are resident together
constexpr int kHandoffMaxWgsPerImage = 8;          // 32 images (half of the planning batch) per launch
constexpr int kHandoffMaxLdsBytes = 80 * 1024;     // half a CU

// Host view of a fused tail's statistics hand-off (kernel argument).
struct HandoffArgs {
    unsigned long long* gran;   // [B][G][slots][kHandoffGranulesPerSlot] granules; nullptr: the launch has no fused tail
    const unsigned* epoch;      // current epoch of the chain
    unsigned* giveups;          // incremented when a bounded spin runs out
};

__device__ __forceinline__ void granule_store(unsigned long long* g, unsigned epoch, float v) {
    __hip_atomic_store((dmn_gu64*)g, ((unsigned long long)epoch << 32) | (unsigned long long)__float_as_uint(v), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long granule_load(const unsigned long long* g) {
    return __hip_atomic_load((const dmn_gu64*)g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Granules per partial: {sum, sum of squared deviations from the partial's own mean, element count} (dmn_norm.h); the
// fourth keeps the stride a power of two.
constexpr int kHandoffGranulesPerSlot = 4;

// Mean / rstd of `ng` consecutive groups (first group g_first) of image `b` from the granules of all its workgroups;
// result pairs in sMR[2 * local group]. Called by every wave of a 256-thread workgroup; no barrier inside. The partials
// of a group are combined exactly as combine_groups (dmn_norm.h) combines the slots of a stats buffer -- 16 lanes per
// group, lane l holds slots l, l + 16, l + 32, l + 48, a fixed butterfly for the sum, then one for the squared
// deviations -- so the coefficients equal those of the two-launch form bit for bit. slots <= kHandoffMaxSlots.
__device__ __forceinline__ void handoff_group_stats(const HandoffArgs& h, unsigned epoch, int b, int G, int slots, int g_first,
                                                    int ng, float inv_count, float* sMR) {
    constexpr int KEEP = kHandoffMaxSlots / 16;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, nwaves = blockDim.x >> 6;
    const int sl = lane & 15;
    for (int g0 = wave * 4; g0 < ng; g0 += nwaves * 4) {
        const int gl = g0 + (lane >> 4);
        const bool live = gl < ng;
        const unsigned long long* p = h.gran + ((size_t)b * G + g_first + (live ? gl : 0)) * slots * kHandoffGranulesPerSlot;
        unsigned long long va[KEEP], vb[KEEP], vc[KEEP];
#pragma unroll
        for (int k = 0; k < KEEP; ++k) va[k] = vb[k] = vc[k] = 0;
        unsigned bad_a = 0, bad_b = 0;
        const long long t0 = wall_clock64();
        for (;;) {
            bool ok = true;
#pragma unroll
            for (int k = 0; k < KEEP; ++k) {
                const int i = sl + 16 * k;
                if (live && i < slots) {
                    va[k] = granule_load(p + (size_t)i * kHandoffGranulesPerSlot);
                    vb[k] = granule_load(p + (size_t)i * kHandoffGranulesPerSlot + 1);
                    vc[k] = granule_load(p + (size_t)i * kHandoffGranulesPerSlot + 2);
                    const bool here = (unsigned)(va[k] >> 32) == epoch && (unsigned)(vb[k] >> 32) == epoch && (unsigned)(vc[k] >> 32) == epoch;
                    if (!here) {
                        bad_a = (unsigned)(va[k] >> 32);
                        bad_b = (unsigned)(vb[k] >> 32);
                    }
                    ok = ok && here;
                }
            }
            if (__all(ok)) break;
            if (wall_clock64() - t0 > kHandoffSpinTicks) {   // wave-uniform: every lane reads the same clock region
                if (lane == 0) atomicAdd(h.giveups, 1u);
                // diagnosis: word 2 of the header is claimed by the first lane that still misses a granule, which then
                // describes it in words 3..12
                if (!ok && atomicCAS(h.giveups + 1, 0u, 1u) == 0u) {
                    unsigned* d = h.giveups + 2;
                    d[0] = epoch;
                    d[1] = bad_a;
                    d[2] = bad_b;
                    d[3] = (unsigned)b;
                    d[4] = (unsigned)(g_first + gl);
                    d[5] = (unsigned)sl;
                    d[6] = (unsigned)slots;
                    d[7] = (blockIdx.x << 16) | (blockIdx.y << 8) | (blockIdx.z & 255);
                    d[8] = *h.epoch;
                    d[9] = (unsigned)G;
                }
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < KEEP; ++k) s += __uint_as_float((unsigned)va[k]);   // slots not held are zero
        const float mean = seg_sum(s, 16) * inv_count;
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < KEEP; ++k)
            t += moments_term(__uint_as_float((unsigned)va[k]), __uint_as_float((unsigned)vb[k]), __uint_as_float((unsigned)vc[k]), mean);
        const float M2 = seg_sum(t, 16);
        if (sl == 0 && live) {
            float rstd;
            moments_finish(M2, inv_count, rstd);
            sMR[2 * gl] = mean;
            sMR[2 * gl + 1] = rstd;
        }
    }
}

// One workgroup's partial of (image b, group g) into its granules.
__device__ __forceinline__ void handoff_publish(const HandoffArgs& h, unsigned epoch, int b, int G, int slots, int g, int slot,
                                                float s, float m2, float n) {
    unsigned long long* dst = h.gran + (((size_t)b * G + g) * slots + slot) * kHandoffGranulesPerSlot;
    granule_store(dst, epoch, s);
    granule_store(dst + 1, epoch, m2);
    granule_store(dst + 2, epoch, n);
}
