// lz4net LZ4Stream framing (host side) and the device decode driver.
#pragma once

#include <stdint.h>

#include <vector>

#include "lz4_launch.hpp"
#include "runtime.hpp"

namespace snappy {

struct Lz4netLayout {
    std::vector<Lz4Chunk> chunks;  // chunks that are structurally sound, in stream order
    uint64_t decoded_size = 0;     // sum of their original lengths
    int stop_status = SNAPPY_OK;   // what ends the walk after those chunks: OK = clean EOF
};

// Walks the chunk headers of the stream.  Chunks before the first structural
// error are kept so that the caller can still report an earlier corrupt block
// first, as the managed reader would (it fails at the first bad chunk it meets).
void parse_lz4net_stream(const uint8_t *in, uint64_t in_len, Lz4netLayout &layout);

// Device-resident result of decoding a stream: the payload lies in
// slot.d_data at [payload_off, payload_off + decoded_size).
struct DecodedOnDevice {
    uint8_t *d_payload = nullptr;
    uint64_t size = 0;
};

// One lz4net stream of a batch: the caller fills in/in_len/layout, the decode
// fills the rest.  status = SNAPPY_E_LZ4_CORRUPT when a block is malformed, else
// layout.stop_status.
struct StreamJob {
    const uint8_t *in = nullptr;
    uint64_t in_len = 0;
    Lz4netLayout layout;
    uint64_t in_off = 0;   // of the stream in the slot's device arena
    uint64_t out_off = 0;  // of its payload in the arena
    uint8_t *d_payload = nullptr;
    int status = SNAPPY_OK;
    bool dropped = false;  // given up for lack of device memory (status SNAPPY_E_NOMEM)
};

// Where a batch of streams lies in the slot's device arena and in its chunk
// table, once prepare_streams has laid it out.
struct StreamPlan {
    uint64_t in_alloc = 0;               // bytes of the arena's input part (streams + slack)
    uint64_t n_chunks = 0;               // rows of the chunk table in slot.d_tasks
    std::vector<uint64_t> first_chunk;   // job a owns rows first_chunk[a] .. first_chunk[a + 1]
    size_t queue_words = 0;              // zeroed 32-bit words in slot.d_misc: word 16 * k is launch k's queue
};

// Lays the streams and their payloads out in the slot's device arena, sizes every
// buffer a decode needs (with room for extra_chunks more table rows and status
// words behind the batch's own), uploads the chunk table and clears the queue
// words.  Moves no stream.  Per-stream outcomes so far are in jobs[i].status.
int prepare_streams(SlotLease &lease, std::vector<StreamJob> &jobs, StreamPlan &plan, size_t extra_chunks);

// After the decode launches have finished: marks the streams with a malformed
// block SNAPPY_E_LZ4_CORRUPT.
int collect_decode_status(SlotLease &lease, std::vector<StreamJob> &jobs, const StreamPlan &plan);

// Uploads all streams and decodes all their chunks in one launch, so that a
// batch of archives fills the GPU (a 1 MiB chunk is one wavefront's work).
// Returns an infrastructure status; per-stream results are in jobs[i].status.
int decode_streams_on_device(SlotLease &lease, std::vector<StreamJob> &jobs);

// Uploads the stream, decodes every chunk of `layout` on the lease's device and
// leaves the payload in HBM.  Returns SNAPPY_E_LZ4_CORRUPT when a block is
// malformed, else layout.stop_status.
int decode_lz4net_on_device(SlotLease &lease, const uint8_t *in, uint64_t in_len,
                            const Lz4netLayout &layout, DecodedOnDevice &out);

}  // namespace snappy
