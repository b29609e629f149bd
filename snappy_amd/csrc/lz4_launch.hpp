// Host-side launch interface of the LZ4 decode kernel (lz4_kernels.hip) and of
// the LZ4 compressor (lz4_encode.hip).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace snappy {

// One chunk of an lz4net LZ4Stream, located in the device copies of the
// compressed stream (in_off) and of the decoded payload (out_off).
struct Lz4Chunk {
    uint64_t in_off;
    uint64_t out_off;
    uint32_t comp_len;    // payload bytes in the stream
    uint32_t orig_len;    // decoded bytes
    uint32_t compressed;  // 0 = stored
    uint32_t prefix_len;  // != 0: only the first prefix_len decoded bytes are wanted (at least; compressed chunks)
};

// Decodes n_chunks chunks.  d_in holds the stream and at least in_alloc
// readable bytes (in_alloc a multiple of 4, covering the stream plus slack);
// *d_queue must be zero at launch; d_status[i] receives 0 (ok) or 1 (corrupt).
hipError_t launch_lz4_decode(const uint8_t *d_in, uint64_t in_alloc, uint8_t *d_out,
                             const Lz4Chunk *d_chunks, uint32_t n_chunks, uint32_t *d_queue,
                             uint32_t *d_status, int n_cu, hipStream_t stream);

// One chunk to compress: in_len bytes at d_in + in_off; the block is written at
// d_out + out_off, where in_len + in_len / 255 + 16 bytes must be available.
struct Lz4EncChunk {
    uint64_t in_off;
    uint64_t out_off;
    uint32_t in_len;
    uint32_t reserved;
};

// Compresses n_chunks chunks, one per wavefront; d_out_sizes[i] receives the
// size of block i.
hipError_t launch_lz4_encode(const uint8_t *d_in, uint8_t *d_out, const Lz4EncChunk *d_chunks,
                             uint32_t n_chunks, uint32_t *d_out_sizes, hipStream_t stream);

// One piece of the finished stream: len bytes at d_base + src go to d_base + dst.
struct Lz4PackItem {
    uint64_t src;
    uint64_t dst;
    uint64_t len;
};

// Gathers the chunks' payloads (compressed blocks, or the input bytes of chunks
// that did not shrink) at their places in the stream, one workgroup per item, so
// that the stream leaves the device in one transfer.  Ranges do not overlap.
hipError_t launch_lz4_pack(uint8_t *d_base, const Lz4PackItem *d_items, uint32_t n_items,
                           hipStream_t stream);

}  // namespace snappy
