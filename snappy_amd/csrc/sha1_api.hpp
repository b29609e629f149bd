// Internal interface between the SHA-1 entry points and other parts of the
// library (the MCDF import hashes file bodies where they lie in HBM).
#pragma once

#include <vector>

#include "runtime.hpp"
#include "sha1_launch.hpp"
#include "transfer.hpp"

namespace snappy {

// The planner's cost table for a device (its own measured chain times once it has
// hashed, one MI355X's otherwise), with the tuning variables applied.
// waits_for_transfer: the kernel will poll an arrival counter, so only one wave
// per SIMD may stay.
Sha1CostModel sha1_cost_model(const DeviceCtx &dev, bool waits_for_transfer);

// Looks up each bound device's copy of the file-scope planning area used by
// snappy_sha1_batch_resident_noalloc (sha1_noalloc.hip).  Called by the
// library's initialisation with hip_device current.
int noalloc_setup_device(int hip_device);

// Plans and launches the SHA-1 kernels for n messages whose (offset, length)
// tables and data are all in device memory; digests (20 bytes per message, in
// the caller's message order) are written to d_digests.  Enqueues only.
// d_epochs / d_arrival (both null, or both set): the data is still being copied
// to the device; message i is complete once d_arrival->arrived >= d_epochs[i].
// group_ends (host, n_groups entries, ascending, the last one = n; may be null): the
// messages arrive in that many groups, in table order; each group is planned on its
// own and the kernel works them off in order, so that messages that have arrived are
// not queued behind larger ones that have not.
int enqueue_sha1_resident(SlotLease &lease, const uint8_t *d_base, const uint64_t *d_offsets,
                          const uint64_t *d_lengths, uint64_t n, uint8_t *d_digests,
                          hipStream_t stream, const uint32_t *d_epochs = nullptr,
                          Sha1Arrival *d_arrival = nullptr, const uint32_t *group_ends = nullptr,
                          uint32_t n_groups = 1);

// A kernel that waits for its input (an arrival counter raised by a transfer or by
// another kernel) keeps one wave on every SIMD for as long as it runs.  Two of them
// on one GPU would leave no room for the copy kernels and decoders they wait for,
// so only one may run per physical GPU at a time: whoever does not get the turn
// copies first and hashes afterwards.
class WaitingKernelTurn {
public:
    explicit WaitingKernelTurn(int hip_device);
    ~WaitingKernelTurn();
    WaitingKernelTurn(const WaitingKernelTurn &) = delete;
    WaitingKernelTurn &operator=(const WaitingKernelTurn &) = delete;
    bool mine() const { return mine_; }

private:
    std::atomic<bool> *flag_;
    bool mine_;
};

// After the stream has been synchronised: stores the kernels' event time for
// snappy_profile_last_ms(0).
void finish_sha1_profile(SlotLease &lease);

// Same for host tables and host digests: uploads the tables, runs the kernels on
// the lease's device, downloads the digests and waits.
int run_sha1_tables(SlotLease &lease, const uint8_t *d_base, const uint64_t *offsets,
                    const uint64_t *lengths, uint64_t n, uint8_t *digests_host,
                    hipStream_t user_stream);

// snappy_sha1_batch for sources that other threads are still filling: the gate is
// asked before any byte of a source is read.
int sha1_batch_gated(const uint8_t *base, const uint64_t *offsets, const uint64_t *lengths, uint64_t n,
                     uint8_t *digests, SourceGate *gate);

// Splits n files into n_shards contiguous groups of near-equal byte count
// (cuts has n_shards + 1 entries).
void plan_shards(const uint64_t *lengths, uint64_t n, uint32_t n_shards, uint64_t *cuts);

// The same by longest-processing-time-first over padded 64-byte blocks: shard_of[i]
// is the shard of file i, shard_blocks (may be null) the blocks each shard ends with.
void plan_shards_lpt(const uint64_t *lengths, uint64_t n, uint32_t n_shards, uint32_t *shard_of,
                     uint64_t *shard_blocks);

}  // namespace snappy
