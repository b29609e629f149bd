// Host-side launch interface of the SHA-1 kernels (sha1_kernels.hip) and of
// the device-side planner (sha1_plan.hip).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sha1_device.hpp"

namespace snappy {

constexpr int kPlanBuckets = 512;

// Planner output and the kernels' work queues, one per batch in device memory.
// Zeroed by launch_sha1_plan, filled by its kernels, read by the SHA-1 kernels.
struct Sha1PlanState {
    uint32_t n_wave;                  // tasks [0, n_wave) go to the wave kernel
    uint32_t reserved0;
    unsigned long long total_blocks;  // padded 64-byte blocks over all messages
    unsigned long long max_blocks;    // of the largest message
    uint32_t reserved1[26];
    uint32_t queue_wave;              // next task of the wave kernel (own 128-byte line)
    uint32_t reserved2[31];
    uint32_t queue_lane;              // next task of the lane kernel, counted from n_wave
    uint32_t reserved3[31];
    uint32_t hist[kPlanBuckets];      // messages per size bucket, largest bucket first
    uint32_t cursor[kPlanBuckets];    // running write position per bucket
};

// Sorts the n messages by descending padded size into d_tasks and decides the
// split between the kernels (see sha1_plan.hip).  forced_min_blocks >= 0
// overrides the split: messages of more blocks than that go to the wave kernel.
hipError_t launch_sha1_plan(const uint64_t *d_offsets, const uint64_t *d_lengths, uint32_t n,
                            Sha1PlanState *d_state, Sha1Task *d_tasks, uint32_t n_simd,
                            double ns_wave, double ns_lane, double simd_ns_lane,
                            long long forced_min_blocks, hipStream_t stream);

// Tasks [state->n_wave, n) of d_tasks, one message per lane; `blocks` workgroups of 256.
hipError_t launch_sha1_lanes(const uint8_t *d_base, const Sha1Task *d_tasks, uint32_t n,
                             Sha1PlanState *d_state, uint8_t *d_digests, int blocks,
                             hipStream_t stream);

// Tasks [0, state->n_wave) of d_tasks, one message per wavefront; `waves`
// single-wave workgroups.
hipError_t launch_sha1_wave(const uint8_t *d_base, const Sha1Task *d_tasks,
                            Sha1PlanState *d_state, uint8_t *d_digests, int waves,
                            hipStream_t stream);

}  // namespace snappy
