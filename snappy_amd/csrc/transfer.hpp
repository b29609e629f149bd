// Host -> device transfers of scattered host ranges ("pieces") into one device
// arena, fast for both kinds of host memory: page-locked sources go by direct
// DMA; ordinary (pageable) ones, which no DMA engine can read, through a ring of
// page-locked buffers filled by several memcpy threads.  Optionally a progress
// counter in device memory is raised after every slice of the arena, for
// kernels that start before their input has arrived.
#pragma once

#include <functional>
#include <vector>

#include "runtime.hpp"

namespace snappy {

// len bytes at src go to arena offset dst.  Pieces are given in arena order.
struct Piece {
    const uint8_t *src;
    uint64_t dst, len;
};

constexpr uint64_t kTransferSliceBytes = 32ull << 20;

// Sources that are still being produced (files being read into their buffer by
// other threads): a transfer asks the gate before it touches source bytes.
struct SourceGate {
    virtual ~SourceGate() = default;
    // Returns once the len bytes at src hold their final content.
    virtual void wait_for(const uint8_t *src, uint64_t len) = 0;
};

// Moves the pieces to d_base on the slot's copy stream and waits for them.
// slice_bytes: granularity (a multiple of 256).  d_progress / slice_numbers (both
// null, or both set): after slice k of the arena has landed, the 32-bit word
// *d_progress is set to slice_numbers[k] == k + 1 (slice_numbers: page-locked
// host memory with one entry per slice).  Small batches of pageable memory are
// sent with plain copies.  gate: asked before source bytes are read.
hipError_t transfer_to_device(Slot &s, int hip_device, const std::vector<Piece> &pieces,
                              uint64_t slice_bytes, uint8_t *d_base, uint32_t *d_progress,
                              const uint32_t *slice_numbers, SourceGate *gate = nullptr);

// The same for pieces that form groups which may lie anywhere in the arena and are sent in the
// order given (an import's archives in the order their files are wanted): group g is
// pieces [group_begin[g], group_begin[g + 1]) (the last group ends with the last piece), in
// arena order inside the group.  The groups follow each other through the same ring without
// a pause; after_group(g) is called on the calling thread as soon as the last bytes of group
// g have been handed to the copy stream (what it enqueues there or on a stream that waits for
// an event recorded there runs behind them).  A failure it returns ends the transfer.
using AfterGroup = std::function<hipError_t(size_t)>;
hipError_t transfer_groups_to_device(Slot &s, int hip_device, const std::vector<Piece> &pieces,
                                     const std::vector<size_t> &group_begin, uint64_t slice_bytes,
                                     uint8_t *d_base, const AfterGroup &after_group);

// The other direction: bytes of device memory to a host buffer, after all work on
// the slot's main stream has finished (the caller has synchronised it).  Ordinary
// (pageable) destinations are served through the ring as well: the copy engine
// fills ring buffers, several threads empty them into the caller's memory.
hipError_t transfer_to_host(Slot &s, int hip_device, const uint8_t *d_src, uint8_t *dst,
                            uint64_t bytes);

// Makes sure the slot's page-locked ring can carry a transfer of n_slices slices of
// slice_bytes.  A caller that launches a kernel which waits for the transfer must
// call this BEFORE the launch: growing the ring frees and allocates page-locked
// memory, and those calls wait for every kernel on the device -- including the
// one that is waiting for the transfer.
int reserve_transfer_ring(Slot &s, uint64_t slice_bytes, uint64_t n_slices);

}  // namespace snappy
