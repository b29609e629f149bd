// Host-side launch interface of the SHA-1 kernels (sha1_kernels.hip).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sha1_device.hpp"

namespace snappy {

// Tasks [first, end) of d_tasks, one message per lane.  *d_queue must be zero
// at launch; lanes draw task indices from it.  `blocks` workgroups of 256.
hipError_t launch_sha1_lanes(const uint8_t *d_base, const Sha1Task *d_tasks, uint32_t first,
                             uint32_t end, uint32_t *d_queue, uint8_t *d_digests, int blocks,
                             hipStream_t stream);

// Tasks [0, n_tasks) of d_tasks, one message per wavefront.  *d_queue must be
// zero at launch.  `waves` single-wave workgroups (20 KiB of LDS each).
hipError_t launch_sha1_wave(const uint8_t *d_base, const Sha1Task *d_tasks, uint32_t n_tasks,
                            uint32_t *d_queue, uint8_t *d_digests, int waves, hipStream_t stream);

}  // namespace snappy
