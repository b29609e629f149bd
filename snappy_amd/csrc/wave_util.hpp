// Wavefront-level helpers shared by the kernels.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace snappy {

// Claims `n` consecutive tickets from a device counter for the whole wave and
// returns the first one in every lane.  Every lane executes the atomic (only
// lane 0 adds a non-zero amount), so there is no divergent branch in front of
// the cross-lane read: a lane-0-only branch here can be merged by the compiler
// with a lane-0-only branch at the end of the caller's loop, which splits the
// wave across loop iterations and breaks the broadcast.
__device__ __forceinline__ uint32_t wave_claim(uint32_t *counter, uint32_t n, uint32_t lane)
{
    const uint32_t old = atomicAdd(counter, lane == 0 ? n : 0u);
    return __builtin_amdgcn_readfirstlane(old);
}

}  // namespace snappy
