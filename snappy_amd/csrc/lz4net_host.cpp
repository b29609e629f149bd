// lz4net LZ4Stream read side: chunk-header walk on the host, block decode on
// the GPU.  Replaces `new LZ4Stream(fs, LZ4StreamMode.Decompress, ...)
// .CopyTo(memoryStream)` of /root/reference/Snappy/Features/Mcdf/McdfManager.cs:
// 72-80.  lz4net (NuGet 1.0.15.93, Snappy/Snappy.csproj:56) is not in the
// reference tree; the framing below follows its LZ4Stream.AcquireNextChunk:
//
//   chunk   := varint(flags) varint(originalLength)
//              [varint(compressedLength) if flags & 1] payload
//   flags   : bit0 compressed, bit1 high-compression (ignored when reading),
//             bits 2.. passes (must be 0 for a compressed chunk)
//   varint  : little-endian base-128, at most 64 bits
//   payload : one raw LZ4 block of compressedLength bytes, or originalLength
//             stored bytes
//   end     : EOF exactly at a chunk boundary
#include "lz4net_host.hpp"
#include "transfer.hpp"

#include <algorithm>
#include <cstdlib>
#include <cstring>

namespace snappy {

namespace {

// 1 = value read, 0 = EOF before the first byte, -1 = EOF inside the varint.
int read_varint(const uint8_t *in, uint64_t in_len, uint64_t &pos, uint64_t &value)
{
    uint64_t result = 0;
    unsigned count = 0;
    for (;;) {
        if (pos >= in_len)
            return count == 0 ? 0 : -1;
        const unsigned b = in[pos++];
        result += static_cast<uint64_t>(b & 0x7F) << count;
        count += 7;
        if ((b & 0x80) == 0 || count >= 64)
            break;
    }
    value = result;
    return 1;
}

}  // namespace

void parse_lz4net_stream(const uint8_t *in, uint64_t in_len, Lz4netLayout &layout)
{
    layout.chunks.clear();
    layout.decoded_size = 0;
    layout.stop_status = SNAPPY_OK;
    uint64_t pos = 0;
    for (;;) {
        uint64_t v = 0;
        int r = read_varint(in, in_len, pos, v);
        if (r == 0)
            return;  // clean end of stream
        if (r < 0) {
            layout.stop_status = SNAPPY_E_LZ4_TRUNCATED;
            return;
        }
        // The managed reader narrows all three fields to 32-bit ints.
        const int32_t flags = static_cast<int32_t>(static_cast<uint32_t>(v));
        const bool compressed = (flags & 1) != 0;
        if (read_varint(in, in_len, pos, v) != 1) {
            layout.stop_status = SNAPPY_E_LZ4_TRUNCATED;
            return;
        }
        const int32_t original = static_cast<int32_t>(static_cast<uint32_t>(v));
        int32_t comp = original;
        if (compressed) {
            if (read_varint(in, in_len, pos, v) != 1) {
                layout.stop_status = SNAPPY_E_LZ4_TRUNCATED;
                return;
            }
            comp = static_cast<int32_t>(static_cast<uint32_t>(v));
        }
        if (original < 0 || comp < 0 || comp > original) {
            layout.stop_status = SNAPPY_E_LZ4_BAD_LENGTHS;
            return;
        }
        if (pos + static_cast<uint64_t>(comp) > in_len) {
            layout.stop_status = SNAPPY_E_LZ4_TRUNCATED;
            return;
        }
        if (compressed && (flags >> 2) != 0) {
            layout.stop_status = SNAPPY_E_LZ4_PASSES;
            return;
        }
        Lz4Chunk c{};
        c.in_off = pos;
        c.out_off = layout.decoded_size;
        c.comp_len = static_cast<uint32_t>(comp);
        c.orig_len = static_cast<uint32_t>(original);
        c.compressed = compressed ? 1u : 0u;
        pos += static_cast<uint64_t>(comp);
        layout.decoded_size += static_cast<uint64_t>(original);
        if (original != 0)  // empty chunks are skipped by the reader
            layout.chunks.push_back(c);
    }
}

int prepare_streams(SlotLease &lease, std::vector<StreamJob> &jobs, StreamPlan &plan, size_t extra_chunks)
{
    Slot &s = *lease.slot;
    // A chunk header is a dozen bytes and may declare 2 GiB of output.  An LZ4 block
    // cannot expand more than 255-fold (one length-extension byte stands for 255 output
    // bytes), so a chunk that claims more can only fail to decode: it is reported as
    // the corrupt block it is without reserving memory for its claim, and the walk ends
    // there as the reader's would.
    for (StreamJob &j : jobs) {
        std::vector<Lz4Chunk> &chunks = j.layout.chunks;
        for (size_t i = 0; i < chunks.size(); ++i)
            if (chunks[i].compressed && chunks[i].orig_len > 255ull * chunks[i].comp_len) {
                j.layout.decoded_size = chunks[i].out_off;
                j.layout.stop_status = SNAPPY_E_LZ4_CORRUPT;
                chunks.resize(i);
                break;
            }
    }
    // Device arena: [stream 0 | slack][stream 1 | slack]...[payload 0][payload 1]...
    // The slack lets the decoder's input window read whole dwords past a stream's
    // last byte; every region starts on a 256-byte boundary.  When the arena cannot
    // be had, the stream that asks for the most is given up (SNAPPY_E_NOMEM for it
    // alone) and the others are tried again: one oversized archive does not take the
    // batch down with it.
    uint64_t cursor = 0, n_chunks = 0, in_alloc = 0;
    // SNAPPY_TEST_ARENA_LIMIT_BYTES: tests of that path (an arena above it counts as refused).
    const char *limit_text = std::getenv("SNAPPY_TEST_ARENA_LIMIT_BYTES");
    const uint64_t test_limit = limit_text ? std::strtoull(limit_text, nullptr, 10) : 0;
    for (;;) {
        cursor = 0;
        n_chunks = 0;
        for (StreamJob &j : jobs) {
            j.in_off = cursor;
            if (j.dropped)
                continue;
            cursor += (j.in_len + 256 + 255) & ~uint64_t(255);
            n_chunks += j.layout.chunks.size();
        }
        in_alloc = cursor;
        for (StreamJob &j : jobs) {
            j.out_off = cursor;
            if (!j.dropped)
                cursor += (j.layout.decoded_size + 255) & ~uint64_t(255);
        }
        int st = (test_limit && cursor > test_limit) ? SNAPPY_E_NOMEM
                                                     : s.d_data.ensure(static_cast<size_t>(cursor + 64));
        if (st == SNAPPY_OK)
            break;
        StreamJob *largest = nullptr;
        size_t alive = 0;
        for (StreamJob &j : jobs)
            if (!j.dropped) {
                ++alive;
                if (!largest || j.in_len + j.layout.decoded_size > largest->in_len + largest->layout.decoded_size)
                    largest = &j;
            }
        if (st != SNAPPY_E_NOMEM || alive <= 1)
            return st;
        largest->dropped = true;
        largest->layout.chunks.clear();
        largest->layout.decoded_size = 0;
        largest->layout.stop_status = SNAPPY_E_NOMEM;
    }
    int st;
    uint8_t *arena = static_cast<uint8_t *>(s.d_data.ptr);
    for (StreamJob &j : jobs) {
        j.d_payload = arena + j.out_off;
        j.status = j.layout.stop_status;
    }
    plan.in_alloc = in_alloc;
    plan.n_chunks = n_chunks;
    plan.first_chunk.assign(jobs.size() + 1, 0);
    if (n_chunks > 0xFFFFFFF0ull)
        return SNAPPY_E_INVALID_ARG;

    const size_t rows = static_cast<size_t>(n_chunks) + extra_chunks;
    const size_t table_bytes = rows * sizeof(Lz4Chunk);
    const size_t status_bytes = rows * sizeof(uint32_t);
    plan.queue_words = 16u * (jobs.size() + 4u);  // a launch's queue counter every 64 bytes (the import: one per arrival, at most archives + 2, and one for the first chunks)
    if ((st = s.d_chunks.ensure(std::max<size_t>(table_bytes, 256))) != SNAPPY_OK) return st;
    if ((st = s.d_status.ensure(std::max<size_t>(status_bytes, 256))) != SNAPPY_OK) return st;
    if ((st = s.d_queues.ensure(plan.queue_words * sizeof(uint32_t))) != SNAPPY_OK) return st;
    if ((st = s.h_stage.ensure(std::max<size_t>(std::max(table_bytes, status_bytes), 256) +
                               (jobs.size() + 4u) * sizeof(uint32_t))) != SNAPPY_OK) return st;

    hipStream_t stream = s.stream;
    Lz4Chunk *table = static_cast<Lz4Chunk *>(s.h_stage.ptr);
    uint64_t k = 0;
    for (size_t a = 0; a < jobs.size(); ++a) {
        StreamJob &j = jobs[a];
        plan.first_chunk[a] = k;
        if (j.dropped)
            continue;
        const uint64_t region = (j.in_len + 256 + 255) & ~uint64_t(255);
        SNAPPY_HIP_TRY(hipMemsetAsync(arena + j.in_off + j.in_len, 0, region - j.in_len, stream));
        for (const Lz4Chunk &c : j.layout.chunks) {
            table[k] = c;
            table[k].in_off += j.in_off;
            table[k].out_off += j.out_off;
            ++k;
        }
    }
    plan.first_chunk[jobs.size()] = k;
    if (n_chunks)
        SNAPPY_HIP_TRY(hipMemcpyAsync(s.d_chunks.ptr, s.h_stage.ptr, static_cast<size_t>(n_chunks) * sizeof(Lz4Chunk),
                                      hipMemcpyHostToDevice, stream));
    SNAPPY_HIP_TRY(hipMemsetAsync(s.d_queues.ptr, 0, plan.queue_words * sizeof(uint32_t), stream));
    SNAPPY_HIP_TRY(hipStreamSynchronize(stream));  // the staging buffer may be reused from here on
    return SNAPPY_OK;
}

int collect_decode_status(SlotLease &lease, std::vector<StreamJob> &jobs, const StreamPlan &plan)
{
    Slot &s = *lease.slot;
    if (plan.n_chunks == 0)
        return SNAPPY_OK;
    SNAPPY_HIP_TRY(hipMemcpy(s.h_stage.ptr, s.d_status.ptr, static_cast<size_t>(plan.n_chunks) * sizeof(uint32_t),
                             hipMemcpyDeviceToHost));
    // A malformed block comes before whatever ended the header walk after it.
    const uint32_t *status = static_cast<const uint32_t *>(s.h_stage.ptr);
    uint64_t k = 0;
    for (StreamJob &j : jobs) {
        for (size_t c = 0; c < j.layout.chunks.size(); ++c, ++k)
            if (status[k] != 0)
                j.status = SNAPPY_E_LZ4_CORRUPT;
    }
    return SNAPPY_OK;
}

int decode_streams_on_device(SlotLease &lease, std::vector<StreamJob> &jobs)
{
    Slot &s = *lease.slot;
    StreamPlan plan;
    int st = prepare_streams(lease, jobs, plan, 0);
    if (st != SNAPPY_OK || plan.n_chunks == 0)
        return st;
    uint8_t *arena = static_cast<uint8_t *>(s.d_data.ptr);
    hipStream_t stream = s.stream;
    std::vector<Piece> pieces;
    for (StreamJob &j : jobs)
        if (!j.dropped && j.in_len)
            pieces.push_back({j.in, j.in_off, j.in_len});
    // The compressed streams travel on the copy stream (through the page-locked ring
    // when they lie in ordinary memory); the call returns when they have landed.
    SNAPPY_HIP_TRY(transfer_to_device(s, lease.dev->hip_device, pieces, kTransferSliceBytes, arena,
                                      nullptr, nullptr));

    const bool prof = profiling_enabled();
    if (prof)
        SNAPPY_HIP_TRY(hipEventRecord(s.ev_prof[1][0], stream));
    SNAPPY_HIP_TRY(launch_lz4_decode(arena, plan.in_alloc, arena,
                                     static_cast<const Lz4Chunk *>(s.d_chunks.ptr),
                                     static_cast<uint32_t>(plan.n_chunks),
                                     static_cast<uint32_t *>(s.d_queues.ptr),
                                     static_cast<uint32_t *>(s.d_status.ptr), lease.dev->n_cu, stream));
    if (prof)
        SNAPPY_HIP_TRY(hipEventRecord(s.ev_prof[1][1], stream));
    SNAPPY_HIP_TRY(hipStreamSynchronize(stream));
    if (prof) {
        float ms = -1.f;
        if (hipEventElapsedTime(&ms, s.ev_prof[1][0], s.ev_prof[1][1]) == hipSuccess)
            profile_store(1, ms);
    }
    return collect_decode_status(lease, jobs, plan);
}

int decode_lz4net_on_device(SlotLease &lease, const uint8_t *in, uint64_t in_len,
                            const Lz4netLayout &layout, DecodedOnDevice &out)
{
    std::vector<StreamJob> jobs(1);
    jobs[0].in = in;
    jobs[0].in_len = in_len;
    jobs[0].layout = layout;
    int st = decode_streams_on_device(lease, jobs);
    if (st != SNAPPY_OK)
        return st;
    out.d_payload = jobs[0].d_payload;
    out.size = layout.decoded_size;
    return jobs[0].status;
}

}  // namespace snappy

using namespace snappy;

extern "C" int snappy_lz4net_decoded_size(const uint8_t *in, uint64_t in_len, uint64_t *out_len)
{
    return abi_guard([&]() -> int {
        if (!out_len || (!in && in_len))
            return SNAPPY_E_INVALID_ARG;
        Lz4netLayout layout;
        parse_lz4net_stream(in, in_len, layout);
        if (layout.stop_status != SNAPPY_OK)
            return layout.stop_status;
        *out_len = layout.decoded_size;
        return SNAPPY_OK;
    });
}

extern "C" int snappy_lz4net_decode(const uint8_t *in, uint64_t in_len, uint8_t *out,
                                    uint64_t out_cap, uint64_t *out_len)
{
    return abi_guard([&]() -> int {
        if (!out_len || (!in && in_len) || (!out && out_cap))
            return SNAPPY_E_INVALID_ARG;
        Lz4netLayout layout;
        parse_lz4net_stream(in, in_len, layout);
        // The reader meets chunks in order: a chunk that does not fit the caller's
        // buffer ends the walk there, before any later structural error.
        uint64_t total = 0;
        for (size_t i = 0; i < layout.chunks.size(); ++i) {
            if (total + layout.chunks[i].orig_len > out_cap) {
                layout.chunks.resize(i);
                layout.stop_status = SNAPPY_E_CAPACITY;
                break;
            }
            total += layout.chunks[i].orig_len;
        }
        layout.decoded_size = total;

        SlotLease lease;
        int st = lease.acquire(pick_device());
        if (st != SNAPPY_OK)
            return st;
        profile_reset();
        DecodedOnDevice dec;
        st = decode_lz4net_on_device(lease, in, in_len, layout, dec);
        if (st != SNAPPY_OK)
            return st;
        if (dec.size)
            SNAPPY_HIP_TRY(transfer_to_host(*lease.slot, lease.dev->hip_device, dec.d_payload, out, dec.size));
        *out_len = dec.size;
        return SNAPPY_OK;
    });
}

// ---------------------------------------------------------------------------
// The write side: lz4net LZ4Stream framing around blocks compressed on the GPU.
// ---------------------------------------------------------------------------
namespace {

uint64_t put_varint(uint8_t *out, uint64_t o, uint64_t v)
{
    while (v >= 0x80) {
        out[o++] = static_cast<uint8_t>(v | 0x80);
        v >>= 7;
    }
    out[o++] = static_cast<uint8_t>(v);
    return o;
}

uint64_t varint_size(uint64_t v)
{
    uint64_t n = 1;
    while (v >= 0x80) {
        ++n;
        v >>= 7;
    }
    return n;
}

constexpr uint32_t kDefaultBlockSize = 1u << 20;  // LZ4Stream's default
constexpr uint32_t kMaxBlockSize = 1u << 30;

}  // namespace

extern "C" int snappy_lz4net_encode_bound(uint64_t in_len, uint32_t block_size, uint64_t *bound)
{
    return abi_guard([&]() -> int {
        if (!bound || block_size > kMaxBlockSize)
            return SNAPPY_E_INVALID_ARG;
        const uint64_t bs = block_size ? block_size : kDefaultBlockSize;
        const uint64_t chunks = (in_len + bs - 1) / bs;
        *bound = in_len + chunks * 16 + 16;
        return SNAPPY_OK;
    });
}

extern "C" int snappy_lz4net_encode(const uint8_t *in, uint64_t in_len, uint32_t block_size,
                                    uint8_t *out, uint64_t out_cap, uint64_t *out_len)
{
    return abi_guard([&]() -> int {
        if ((!in && in_len) || !out_len || (!out && out_cap) || block_size > kMaxBlockSize)
            return SNAPPY_E_INVALID_ARG;
        *out_len = 0;
        if (in_len == 0)
            return SNAPPY_OK;  // LZ4Stream writes nothing for an empty input
        const uint64_t bs = block_size ? block_size : kDefaultBlockSize;
        const uint64_t n_chunks = (in_len + bs - 1) / bs;
        if (n_chunks >= 0x7FFFFFFFull)
            return SNAPPY_E_INVALID_ARG;

        SlotLease lease;
        int st = lease.acquire(pick_device());
        if (st != SNAPPY_OK)
            return st;
        Slot &s = *lease.slot;
        profile_reset();

        // Arena: [input][one worst-case output area per chunk][the finished stream].
        const uint64_t area = (bs + bs / 255 + 16 + 15) & ~uint64_t(15);
        const uint64_t in_area = (in_len + 15) & ~uint64_t(15);
        const uint64_t stream_at = in_area + n_chunks * area;
        const uint64_t stream_room = in_len + 16 * n_chunks + 256;  // headers are at most 11 bytes per chunk
        std::vector<Lz4EncChunk> chunks(n_chunks);
        for (uint64_t k = 0; k < n_chunks; ++k) {
            chunks[k].in_off = k * bs;
            chunks[k].out_off = in_area + k * area;
            chunks[k].in_len = static_cast<uint32_t>(std::min<uint64_t>(bs, in_len - k * bs));
            chunks[k].reserved = 0;
        }
        if ((st = s.d_data.ensure(stream_at + stream_room)) != SNAPPY_OK) return st;
        if ((st = s.d_tasks.ensure(n_chunks * sizeof(Lz4EncChunk))) != SNAPPY_OK) return st;
        if ((st = s.d_tables.ensure(n_chunks * sizeof(Lz4PackItem))) != SNAPPY_OK) return st;
        if ((st = s.d_out.ensure(n_chunks * sizeof(uint32_t))) != SNAPPY_OK) return st;
        uint8_t *d_base = static_cast<uint8_t *>(s.d_data.ptr);
        Lz4EncChunk *d_chunks = static_cast<Lz4EncChunk *>(s.d_tasks.ptr);
        uint32_t *d_sizes = static_cast<uint32_t *>(s.d_out.ptr);
        {
            std::vector<Piece> pieces{{in, 0, in_len}};
            SNAPPY_HIP_TRY(transfer_to_device(s, lease.dev->hip_device, pieces, kTransferSliceBytes, d_base,
                                              nullptr, nullptr));
        }
        SNAPPY_HIP_TRY(hipMemcpyAsync(d_chunks, chunks.data(), n_chunks * sizeof(Lz4EncChunk),
                                      hipMemcpyHostToDevice, s.stream));
        const bool prof = profiling_enabled();
        if (prof)
            SNAPPY_HIP_TRY(hipEventRecord(s.ev_prof[1][0], s.stream));
        SNAPPY_HIP_TRY(launch_lz4_encode(d_base, d_base, d_chunks, static_cast<uint32_t>(n_chunks),
                                         d_sizes, s.stream));
        if (prof)
            SNAPPY_HIP_TRY(hipEventRecord(s.ev_prof[1][1], s.stream));
        std::vector<uint32_t> sizes(n_chunks);
        SNAPPY_HIP_TRY(hipMemcpyAsync(sizes.data(), d_sizes, n_chunks * sizeof(uint32_t),
                                      hipMemcpyDeviceToHost, s.stream));
        SNAPPY_HIP_TRY(hipStreamSynchronize(s.stream));
        if (prof) {
            float ms = -1.f;
            if (hipEventElapsedTime(&ms, s.ev_prof[1][0], s.ev_prof[1][1]) == hipSuccess)
                profile_store(1, ms);
            else
                (void)hipGetLastError();
        }

        // FlushCurrentChunk: flags (bit 0 = compressed), original length, compressed
        // length when compressed, payload; a block that does not shrink is stored.
        uint64_t need = 0;
        for (uint64_t k = 0; k < n_chunks; ++k) {
            const uint64_t n = chunks[k].in_len, c = sizes[k];
            need += 1 + varint_size(n) + (c < n ? varint_size(c) + c : n);
        }
        *out_len = need;
        if (need > out_cap)
            return SNAPPY_E_CAPACITY;
        // The payloads are gathered at their places in the stream on the device, the
        // stream comes back in one transfer, and the headers are written into the gaps.
        std::vector<Lz4PackItem> items(n_chunks);
        uint64_t o = 0;
        for (uint64_t k = 0; k < n_chunks; ++k) {
            const uint64_t n = chunks[k].in_len, c = sizes[k];
            const bool compressed = c < n;
            o += 1 + varint_size(n) + (compressed ? varint_size(c) : 0);
            items[k] = {compressed ? chunks[k].out_off : chunks[k].in_off, stream_at + o, compressed ? c : n};
            o += items[k].len;
        }
        Lz4PackItem *d_items = static_cast<Lz4PackItem *>(s.d_tables.ptr);
        SNAPPY_HIP_TRY(hipMemcpyAsync(d_items, items.data(), n_chunks * sizeof(Lz4PackItem),
                                      hipMemcpyHostToDevice, s.stream));
        SNAPPY_HIP_TRY(launch_lz4_pack(d_base, d_items, static_cast<uint32_t>(n_chunks), s.stream));
        SNAPPY_HIP_TRY(hipStreamSynchronize(s.stream));
        SNAPPY_HIP_TRY(transfer_to_host(s, lease.dev->hip_device, d_base + stream_at, out, need));
        o = 0;
        for (uint64_t k = 0; k < n_chunks; ++k) {
            const uint64_t n = chunks[k].in_len, c = sizes[k];
            const bool compressed = c < n;
            o = put_varint(out, o, compressed ? 1u : 0u);
            o = put_varint(out, o, n);
            if (compressed)
                o = put_varint(out, o, c);
            o += items[k].len;
        }
        SNAPPY_HIP_TRY(hipStreamSynchronize(s.stream));
        return SNAPPY_OK;
    });
}
