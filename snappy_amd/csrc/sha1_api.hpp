// Internal interface between the SHA-1 entry points and other parts of the
// library (the MCDF import hashes file bodies where they lie in HBM).
#pragma once

#include <vector>

#include "runtime.hpp"
#include "sha1_launch.hpp"

namespace snappy {

struct Sha1Plan {
    std::vector<Sha1Task> tasks;  // descending size; [0, n_wave) go to the wave kernel
    uint32_t n_wave = 0;
    uint64_t total_blocks = 0;  // padded 64-byte blocks over all messages
};

// Orders the messages by size and splits them between the two kernels.
// offsets are relative to the device base pointer the plan will run against.
void build_sha1_plan(const uint64_t *offsets, const uint64_t *lengths, uint64_t n, int n_cu,
                     Sha1Plan &plan);

// Splits n files into n_shards contiguous groups of near-equal byte count
// (cuts has n_shards + 1 entries).
void plan_shards(const uint64_t *lengths, uint64_t n, uint32_t n_shards, uint64_t *cuts);

// Uploads the plan, runs the kernels on the lease's device and returns the
// digests (20 bytes per message, in the caller's message order) in host memory.
int run_sha1_plan(SlotLease &lease, const uint8_t *d_base, const Sha1Plan &plan,
                  uint8_t *digests_host, hipStream_t user_stream);

}  // namespace snappy
