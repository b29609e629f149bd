// Fused MCDF import data path: lz4net decode on the GPU, header parse on the
// host, SHA-1 of every file body where it lies in HBM.  Replaces the data flow
// of ImportMcdf (/root/reference/Snappy/Features/Mcdf/McdfManager.cs:72-84) and
// ExtractAndHashMapFiles (:28-66); the decoded payload never returns to the host
// unless the caller asks for it.  The batch form runs several archives through
// one decode launch and one hash launch: an archive is only 64-512 chunks, too
// few to occupy the GPU on its own.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>

#include "lz4net_host.hpp"
#include "mcdf_host.hpp"
#include "transfer.hpp"
#include "sha1_api.hpp"

namespace snappy {

namespace {

struct ArchiveState {
    int status = SNAPPY_OK;
    McdfHeaderInfo info;
    std::vector<McdfSlice> slices;
};

// Parses the header of archive a out of its decoded payload on the device and
// slices its files; appends the bodies to hash to (offsets, lengths, owner).
int header_and_slices(uint8_t *arena, const StreamJob &job, size_t a, uint64_t decoded_avail, ArchiveState &as,
                      std::vector<uint8_t> &digests_out, std::vector<uint64_t> &offsets,
                      std::vector<uint64_t> &lengths, std::vector<std::pair<size_t, size_t>> &owner,
                      bool *header_beyond_avail, const uint8_t *host_copy = nullptr, uint64_t host_len = 0)
{
    // host_copy (may be null): the payload's first host_len bytes, already on the host.
    const uint64_t size = job.layout.decoded_size;
    // Header: 9 bytes tell how much JSON follows.
    uint8_t head[9] = {0};
    const uint64_t head_len = size < 9 ? size : 9;
    if (head_len > decoded_avail) {
        *header_beyond_avail = true;
        return SNAPPY_OK;
    }
    if (head_len && host_copy && head_len <= host_len)
        std::memcpy(head, host_copy, head_len);
    else if (head_len)
        SNAPPY_HIP_TRY(hipMemcpy(head, job.d_payload, head_len, hipMemcpyDeviceToHost));
    uint64_t need = 0;
    as.status = mcdf_header_extent(head, head_len, size, &need);
    if (as.status != SNAPPY_OK)
        return SNAPPY_OK;
    if (need > decoded_avail) {
        *header_beyond_avail = true;
        return SNAPPY_OK;
    }
    std::vector<uint8_t> header;
    const uint8_t *text = host_copy;
    if (!host_copy || need > host_len) {
        header.resize(static_cast<size_t>(need));
        SNAPPY_HIP_TRY(hipMemcpy(header.data(), job.d_payload, need, hipMemcpyDeviceToHost));
        text = header.data();
    }
    as.status = parse_mcdf_header(text, need, as.info);
    if (as.status != SNAPPY_OK)
        return SNAPPY_OK;
    as.status = mcdf_slice_files(as.info.files, as.info.body_offset, size, as.slices);
    if (as.status != SNAPPY_OK)
        return SNAPPY_OK;
    digests_out.assign(as.slices.size() * 20, 0);
    for (size_t f = 0; f < as.slices.size(); ++f) {
        if (as.slices[f].status != SNAPPY_FILE_HASHED)
            continue;
        offsets.push_back(static_cast<uint64_t>(job.d_payload - arena) + as.slices[f].offset);
        lengths.push_back(as.slices[f].length);
        owner.emplace_back(a, f);
    }
    return SNAPPY_OK;
}

// The end of the overlapped import: which streams had a malformed block, and the
// digests to their archives' places.
int finish_overlapped(Slot &s, std::vector<StreamJob> &jobs, std::vector<ArchiveState> &states,
                      std::vector<std::vector<uint8_t>> &digests_out, const std::vector<size_t> &order,
                      const std::vector<uint64_t> &first_row, size_t n_rows,
                      const std::vector<std::pair<size_t, size_t>> &owner, const std::vector<uint8_t> &packed)
{
    const size_t n = jobs.size();
    const uint32_t *d_status = static_cast<const uint32_t *>(s.d_status.ptr);
    // A malformed block comes before whatever ended the header walk after it.
    if (n_rows) {
        std::vector<uint32_t> chunk_status(n_rows);
        SNAPPY_HIP_TRY(hipMemcpy(chunk_status.data(), d_status, n_rows * sizeof(uint32_t), hipMemcpyDeviceToHost));
        for (size_t k = 0; k < order.size(); ++k)
            for (uint64_t r = first_row[k]; r < first_row[k + 1]; ++r)
                if (chunk_status[r] != 0)
                    jobs[order[k]].status = SNAPPY_E_LZ4_CORRUPT;
    }
    // The stream is read to its end before anything looks at the header, so a stream error comes
    // before a header error, and among stream errors the malformed block (every decoded chunk lies
    // before the point where the walk over the chunk headers stopped) before whatever stopped the walk.
    for (size_t a = 0; a < n; ++a)
        if (jobs[a].status != SNAPPY_OK) {
            states[a].status = jobs[a].status;
            digests_out[a].clear();
        }
    for (size_t k = 0; k < owner.size(); ++k)
        if (states[owner[k].first].status == SNAPPY_OK)
            std::memcpy(digests_out[owner[k].first].data() + 20 * owner[k].second, packed.data() + 20 * k, 20);
    return SNAPPY_OK;
}

// Decoded bytes of an archive's first chunk that are asked for ahead of the rest:
// room for the header of some 1500 files.
constexpr uint32_t kHeaderPrefix = 128u << 10;

// The import with its three stages overlapped.  Hashing can only be planned once
// every archive's header is known, and the headers are inside the compressed
// streams; but a header sits in an archive's first chunk.  So: the first chunk of
// every archive is sent and decoded ahead (a few MB), the headers are parsed, and
// the SHA-1 kernel is launched at once, every file waiting for its archive's
// number on the arrival counter.  The archives then follow whole, the one with
// the largest file first: copy, decode, raise the counter.  The largest file's
// chain, which decides the finishing time, starts after one archive's transfer
// instead of after all of them.
// Returns SNAPPY_OK with *handled = false when the batch is not for this path.
int import_batch_overlapped(SlotLease &lease, std::vector<StreamJob> &jobs, std::vector<ArchiveState> &states,
                            std::vector<std::vector<uint8_t>> &digests_out, bool *handled)
{
    *handled = false;
    Slot &s = *lease.slot;
    const size_t n = jobs.size();
    // SNAPPY_TRACE_IMPORT=1: host-clock milestones of the overlapped import on stderr (tuning).
    const bool trace = std::getenv("SNAPPY_TRACE_IMPORT") != nullptr;
    const auto t_begin = std::chrono::steady_clock::now();
    auto since = [&] {
        return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    };
    uint64_t stream_bytes = 0;
    for (const StreamJob &j : jobs)
        stream_bytes += j.in_len;
    // SNAPPY_OVERLAP_MIN_BYTES: tests and tuning (the same switch as the SHA-1 entry points').
    const char *min_text = std::getenv("SNAPPY_OVERLAP_MIN_BYTES");
    const uint64_t min_bytes = min_text ? std::strtoull(min_text, nullptr, 10) : (64ull << 20);
    if (s.d_flags == nullptr || stream_bytes < min_bytes || n >= 0x7FFFFFFFull)
        return SNAPPY_OK;
    WaitingKernelTurn turn(lease.dev->hip_device);
    if (!turn.mine())
        return SNAPPY_OK;

    StreamPlan plan;
    int st = prepare_streams(lease, jobs, plan, n);
    if (st != SNAPPY_OK)
        return st;
    *handled = true;  // from here on the batch is laid out for this path
    const double t_laid_out = since();
    uint8_t *arena = static_cast<uint8_t *>(s.d_data.ptr);
    const int hip_device = lease.dev->hip_device;
    Lz4Chunk *d_table = static_cast<Lz4Chunk *>(s.d_chunks.ptr);
    uint32_t *d_queue = static_cast<uint32_t *>(s.d_queues.ptr);
    uint32_t *d_status = static_cast<uint32_t *>(s.d_status.ptr);

    // 1. The first chunk of every archive: send, decode, read the headers.
    float first_chunks_ms = 0.f;  // how long one chunk of this content takes to decode
    double t_first_sent = 0.0;
    std::vector<Lz4Chunk> firsts;
    std::vector<size_t> first_of(n, SIZE_MAX);
    std::vector<Piece> pieces;
    for (size_t a = 0; a < n; ++a) {
        states[a].status = jobs[a].status;
        if (jobs[a].dropped || jobs[a].layout.chunks.empty())
            continue;
        Lz4Chunk c = jobs[a].layout.chunks[0];
        pieces.push_back({jobs[a].in + c.in_off, jobs[a].in_off + c.in_off, c.comp_len});
        c.in_off += jobs[a].in_off;
        c.out_off += jobs[a].out_off;
        c.prefix_len = kHeaderPrefix < c.orig_len ? kHeaderPrefix : 0u;
        first_of[a] = firsts.size();
        firsts.push_back(c);
    }
    if (!firsts.empty()) {
        // Every piece a group of its own: slices are then cut per piece (the pieces lie all over the
        // arena; slices over the arena's whole span were several hundred, nearly all of them empty).
        std::vector<size_t> one_each(pieces.size());
        for (size_t k = 0; k < one_each.size(); ++k)
            one_each[k] = k;
        SNAPPY_HIP_TRY(transfer_groups_to_device(s, hip_device, pieces, one_each, kTransferSliceBytes, arena,
                                                 [](size_t) { return hipSuccess; }));
        t_first_sent = since();
        SNAPPY_HIP_TRY(hipMemcpyAsync(d_table + plan.n_chunks, firsts.data(), firsts.size() * sizeof(Lz4Chunk),
                                      hipMemcpyHostToDevice, s.stream));
        SNAPPY_HIP_TRY(hipEventRecord(s.ev_prof[1][0], s.stream));
        SNAPPY_HIP_TRY(launch_lz4_decode(arena, plan.in_alloc, arena, d_table + plan.n_chunks,
                                         static_cast<uint32_t>(firsts.size()), d_queue + 16u * (n + 3u),
                                         d_status + plan.n_chunks, lease.dev->n_cu, s.stream));
        SNAPPY_HIP_TRY(hipEventRecord(s.ev_prof[1][1], s.stream));
        SNAPPY_HIP_TRY(hipStreamSynchronize(s.stream));
        if (hipEventElapsedTime(&first_chunks_ms, s.ev_prof[1][0], s.ev_prof[1][1]) != hipSuccess) {
            (void)hipGetLastError();
            first_chunks_ms = 0.f;
        }
    }
    const double t_first_decoded = since();
    std::vector<uint32_t> first_status(firsts.size(), 0);
    if (!firsts.empty())
        SNAPPY_HIP_TRY(hipMemcpy(first_status.data(), d_status + plan.n_chunks, firsts.size() * sizeof(uint32_t),
                                 hipMemcpyDeviceToHost));
    // The decoded starts of all first chunks come to the host in one go, into the page-locked ring
    // (idle between the transfers): two small blocking copies per archive took 5 ms for 64 archives.
    const uint8_t *prefixes = nullptr;
    if (!firsts.empty() && firsts.size() * uint64_t(kHeaderPrefix) <= kTransferSliceBytes &&
        reserve_transfer_ring(s, kTransferSliceBytes, 1) == SNAPPY_OK) {
        uint8_t *ring = static_cast<uint8_t *>(s.h_ring.ptr);
        for (size_t a = 0; a < n; ++a) {
            if (first_of[a] == SIZE_MAX)
                continue;
            const uint64_t take = std::min<uint64_t>(firsts[first_of[a]].orig_len, kHeaderPrefix);
            if (take)
                SNAPPY_HIP_TRY(hipMemcpyAsync(ring + first_of[a] * size_t(kHeaderPrefix), jobs[a].d_payload, take,
                                              hipMemcpyDeviceToHost, s.stream));
        }
        SNAPPY_HIP_TRY(hipStreamSynchronize(s.stream));
        prefixes = ring;
    }
    std::vector<uint64_t> offsets, lengths;
    std::vector<std::pair<size_t, size_t>> owner;  // (archive, file)
    bool header_beyond = false;
    // Header by header (JSON text of some tens of KB each), on a few threads when the headers are
    // all on the host already: 64 archives' headers took 3.4 ms on one.
    struct Parsed {
        std::vector<uint64_t> offsets, lengths;
        std::vector<std::pair<size_t, size_t>> owner;
        bool beyond = false;
        int st = SNAPPY_OK;
    };
    std::vector<Parsed> parsed(n);
    auto parse_one = [&](size_t a) {
        ArchiveState &as = states[a];
        // A stream error ends the import before anything looks at the header; a first
        // chunk that does not decode is the stream's first error.
        if (first_of[a] != SIZE_MAX && first_status[first_of[a]] != 0)
            jobs[a].status = as.status = SNAPPY_E_LZ4_CORRUPT;
        if (as.status != SNAPPY_OK)
            return;
        const Lz4Chunk *fc = first_of[a] == SIZE_MAX ? nullptr : &firsts[first_of[a]];
        const uint64_t avail = !fc ? 0 : (fc->prefix_len ? fc->prefix_len : fc->orig_len);
        const bool on_host = prefixes && fc;
        Parsed &out = parsed[a];
        try {
            out.st = header_and_slices(arena, jobs[a], a, avail, as, digests_out[a], out.offsets, out.lengths, out.owner,
                                       &out.beyond, on_host ? prefixes + first_of[a] * size_t(kHeaderPrefix) : nullptr,
                                       on_host ? std::min<uint64_t>(fc->orig_len, kHeaderPrefix) : 0);
        } catch (const std::bad_alloc &) {
            out.st = SNAPPY_E_NOMEM;
        } catch (...) {
            out.st = SNAPPY_E_INVALID_ARG;
        }
    };
    {
        std::atomic<size_t> next{0};
        auto worker = [&] {
            for (size_t a; (a = next.fetch_add(1)) < n;)
                parse_one(a);
        };
        std::vector<std::thread> helpers;
        if (prefixes && n >= 8)   // (without the host copies a header is fetched by blocking copies: one thread)
            for (unsigned t = 1; t < 8 && t < n / 2; ++t)
                if (!try_spawn(helpers, worker))
                    break;
        worker();
        for (auto &h : helpers)
            h.join();
    }
    for (size_t a = 0; a < n; ++a) {
        Parsed &in = parsed[a];
        if (in.st != SNAPPY_OK)
            return in.st;
        header_beyond = header_beyond || in.beyond;
        offsets.insert(offsets.end(), in.offsets.begin(), in.offsets.end());
        lengths.insert(lengths.end(), in.lengths.begin(), in.lengths.end());
        owner.insert(owner.end(), in.owner.begin(), in.owner.end());
    }
    parsed.clear();
    if (trace)
        std::fprintf(stderr, "[import] streams laid out at %.1f ms, first chunks sent at %.1f, decoded at %.1f (kernel %.2f ms), "
                             "headers of %zu archives parsed at %.1f ms\n",
                     t_laid_out, t_first_sent, t_first_decoded, first_chunks_ms, n, since());
    if (header_beyond || owner.size() >= 0xFFFFFFF0ull) {
        // A header that runs past its archive's first chunk (many thousands of files):
        // the plain order of things handles it.
        *handled = false;
        return SNAPPY_OK;
    }

    // 2. The archives in the order they will arrive: largest file first.
    std::vector<uint64_t> largest(n, 0);
    for (size_t k = 0; k < owner.size(); ++k)
        largest[owner[k].first] = std::max(largest[owner[k].first], lengths[k]);
    std::vector<size_t> order;
    for (size_t a = 0; a < n; ++a)
        if (!jobs[a].dropped && jobs[a].in_len)
            order.push_back(a);
    std::stable_sort(order.begin(), order.end(), [&](size_t x, size_t y) { return largest[x] > largest[y]; });
    // What arrives as one piece of work: a contiguous range of an archive's stream and the chunks
    // that can be decoded once it is there.  Normally a whole archive; the first archive, which
    // holds the batch's largest file, comes in three: the chunks that file lies in, then the
    // stream in front of them, then the rest, so that the chain which decides the import's
    // finishing time starts after some tens of chunks' transfer and decode instead of after a
    // whole archive's.  SNAPPY_MCDF_NO_SPLIT=1: whole archives only (comparisons).
    struct Arrival {
        size_t archive;
        Piece piece;
        std::vector<Lz4Chunk> rows;  // arena coordinates
    };
    std::vector<Arrival> arrivals;
    size_t big = SIZE_MAX;  // the largest file, as an index into owner / lengths
    for (size_t k = 0; k < owner.size(); ++k)
        if (big == SIZE_MAX || lengths[k] > lengths[big])
            big = k;
    bool big_alone = false;
    for (size_t pos = 0; pos < order.size(); ++pos) {
        const size_t a = order[pos];
        const StreamJob &j = jobs[a];
        const std::vector<Lz4Chunk> &chunks = j.layout.chunks;
        auto rows_of = [&](size_t from, size_t to) {
            std::vector<Lz4Chunk> rows(chunks.begin() + static_cast<ptrdiff_t>(from), chunks.begin() + static_cast<ptrdiff_t>(to));
            for (Lz4Chunk &c : rows) {
                c.in_off += j.in_off;
                c.out_off += j.out_off;
            }
            return rows;
        };
        size_t c0 = 0, c1 = 0;
        if (pos == 0 && big != SIZE_MAX && owner[big].first == a && lengths[big] > 0 &&
            std::getenv("SNAPPY_MCDF_NO_SPLIT") == nullptr) {
            const uint64_t lo = offsets[big] - j.out_off, hi = lo + lengths[big];  // the file's place in the payload
            while (c0 < chunks.size() && chunks[c0].out_off + chunks[c0].orig_len <= lo)
                ++c0;
            c1 = c0;
            while (c1 < chunks.size() && chunks[c1].out_off < hi)
                ++c1;
        }
        if (c1 > c0 && chunks.size() - (c1 - c0) >= 4) {
            const uint64_t from = chunks[c0].in_off, to = chunks[c1 - 1].in_off + chunks[c1 - 1].comp_len;
            arrivals.push_back({a, {j.in + from, j.in_off + from, to - from}, rows_of(c0, c1)});
            arrivals.push_back({a, {j.in, j.in_off, from}, rows_of(0, c0)});
            arrivals.push_back({a, {j.in + to, j.in_off + to, j.in_len - to}, rows_of(c1, chunks.size())});
            big_alone = true;
        } else {
            arrivals.push_back({a, {j.in, j.in_off, j.in_len}, rows_of(0, chunks.size())});
        }
    }
    std::vector<uint32_t> epoch_of(n, 0);  // an archive is there with its last arrival
    std::vector<size_t> arrival_archive(arrivals.size());
    uint32_t *numbers = reinterpret_cast<uint32_t *>(static_cast<uint8_t *>(s.h_stage.ptr) + s.h_stage.cap) -
                        (n + 4u);  // page-locked words the counter is raised from
    for (size_t k = 0; k < arrivals.size(); ++k) {
        epoch_of[arrivals[k].archive] = static_cast<uint32_t>(k + 1);
        arrival_archive[k] = arrivals[k].archive;
        numbers[k] = static_cast<uint32_t>(k + 1);
    }
    // The chunk table once more, in arrival order, so that any run of consecutive
    // arrivals can be decoded by one launch.
    std::vector<Lz4Chunk> ordered;
    std::vector<uint64_t> first_row(arrivals.size() + 1, 0);
    for (size_t k = 0; k < arrivals.size(); ++k) {
        first_row[k] = ordered.size();
        ordered.insert(ordered.end(), arrivals[k].rows.begin(), arrivals[k].rows.end());
    }
    first_row[arrivals.size()] = ordered.size();
    if (!ordered.empty())
        SNAPPY_HIP_TRY(hipMemcpy(d_table, ordered.data(), ordered.size() * sizeof(Lz4Chunk), hipMemcpyHostToDevice));
    // Every file waits for its archive's last arrival; the largest file, sent ahead, for the first.
    std::vector<uint32_t> file_epoch(owner.size());
    for (size_t k = 0; k < owner.size(); ++k)
        file_epoch[k] = (big_alone && k == big) ? 1u : epoch_of[owner[k].first];

    // The files in the order their archives arrive (stable: an archive's files keep their order).
    {
        std::vector<size_t> by_arrival(owner.size());
        for (size_t k = 0; k < by_arrival.size(); ++k)
            by_arrival[k] = k;
        std::stable_sort(by_arrival.begin(), by_arrival.end(),
                         [&](size_t x, size_t y) { return file_epoch[x] < file_epoch[y]; });
        std::vector<uint64_t> off2(owner.size()), len2(owner.size());
        std::vector<std::pair<size_t, size_t>> own2(owner.size());
        std::vector<uint32_t> epoch2(owner.size());
        for (size_t k = 0; k < by_arrival.size(); ++k) {
            off2[k] = offsets[by_arrival[k]];
            len2[k] = lengths[by_arrival[k]];
            own2[k] = owner[by_arrival[k]];
            epoch2[k] = file_epoch[by_arrival[k]];
        }
        offsets.swap(off2);
        lengths.swap(len2);
        owner.swap(own2);
        file_epoch.swap(epoch2);
    }

    // A chunk is decoded by one wave, and on streams of short sequences its chain of
    // sequences takes milliseconds (long copies: 0.3 ms).  Next to a resident SHA-1
    // kernel only half as many decoder waves fit on a CU, so decoding such archives
    // behind their transfers one by one would hold the later ones back by more than
    // the overlap gains.  When one chunk takes much longer than an archive's transfer, the
    // order stays the plain one: send everything, decode everything in one launch with
    // the GPU to itself, then hash (the headers read above are not read again).
    uint64_t mean_stream = 0;
    for (size_t a : order)
        mean_stream += jobs[a].in_len / order.size();
    // (The first chunks were decoded up to kHeaderPrefix bytes: a whole chunk takes in proportion.)
    uint64_t mean_chunk = 0;
    for (const Lz4Chunk &c : firsts)
        mean_chunk += c.orig_len / firsts.size();
    const float whole_chunk_ms =
        first_chunks_ms * (mean_chunk > kHeaderPrefix ? static_cast<float>(mean_chunk) / kHeaderPrefix : 1.f);
    // (Decodes run on a stream of their own, beside the following transfers, in runs of up to
    // eight archives: only a chunk that takes several archives' transfers puts them behind.)
    // The waiting kernel takes the files group by group in the order of arrival (one group per
    // decode launch, below), largest first inside a group, so a wave never sits on a file of an
    // archive that is still to come while files that have arrived go unclaimed.  The import then
    // takes about the longer of "all transfers" and "the largest file's chain after its
    // archive's transfer" instead of their sum.  SNAPPY_MCDF_PLAIN_ORDER=1: send all, decode all,
    // hash (comparisons).
    const bool overlap_pays = std::getenv("SNAPPY_MCDF_PLAIN_ORDER") == nullptr ||
                              std::getenv("SNAPPY_MCDF_FORCE_OVERLAP") != nullptr;
    if (!order.empty() && (!overlap_pays || whole_chunk_ms > 4.f * static_cast<float>(mean_stream) / 50e6f)) {
        std::vector<Piece> all;
        for (size_t a = 0; a < n; ++a)  // arena order
            if (!jobs[a].dropped && jobs[a].in_len)
                all.push_back({jobs[a].in, jobs[a].in_off, jobs[a].in_len});
        SNAPPY_HIP_TRY(transfer_to_device(s, hip_device, all, kTransferSliceBytes, arena, nullptr, nullptr));
        const bool timed = profiling_enabled();
        if (timed)
            SNAPPY_HIP_TRY(hipEventRecord(s.ev_prof[1][0], s.stream));
        if (!ordered.empty())
            SNAPPY_HIP_TRY(launch_lz4_decode(arena, plan.in_alloc, arena, d_table, static_cast<uint32_t>(ordered.size()),
                                             d_queue, d_status, lease.dev->n_cu, s.stream));
        if (timed)
            SNAPPY_HIP_TRY(hipEventRecord(s.ev_prof[1][1], s.stream));
        SNAPPY_HIP_TRY(hipStreamSynchronize(s.stream));
        if (timed) {
            float ms = -1.f;
            if (hipEventElapsedTime(&ms, s.ev_prof[1][0], s.ev_prof[1][1]) == hipSuccess)
                profile_store(1, ms);
            else
                (void)hipGetLastError();
        }
        std::vector<uint8_t> hashed(owner.size() * 20);
        if (!owner.empty() &&
            (st = run_sha1_tables(lease, arena, offsets.data(), lengths.data(), owner.size(), hashed.data(), nullptr)) != SNAPPY_OK)
            return st;
        return finish_overlapped(s, jobs, states, digests_out, arrival_archive, first_row, ordered.size(), owner, hashed);
    }

    // The decode launches: the first archive alone (it holds the largest file, whose SHA-1 chain
    // decides when the import ends), the others in runs of `run_len`, larger where a chunk takes
    // long next to an archive's transfer (a decode launch lasts as long as one chunk's chain of
    // sequences, however few chunks it has).  run_end_at[k]: a run ends with the k-th arrival.
    const float transfer_ms = static_cast<float>(mean_stream) / 50e6f;  // one archive at PCIe rate
    size_t run_len = 1;
    if (whole_chunk_ms > 0.25f * transfer_ms)
        run_len = std::min<size_t>(8, static_cast<size_t>(4.f * whole_chunk_ms / (transfer_ms > 0.f ? transfer_ms : 1.f)) + 1);
    if (const char *v = std::getenv("SNAPPY_MCDF_DECODE_RUN"))  // tuning
        run_len = std::max<size_t>(1, std::strtoul(v, nullptr, 10));
    std::vector<bool> run_end_at(arrivals.size(), false);
    {
        size_t from = 0;
        for (size_t k = 0; k < arrivals.size(); ++k)
            if (k == 0 || k + 1 == arrivals.size() || k + 1 - from >= run_len) {
                run_end_at[k] = true;
                from = k + 1;
            }
    }
    // The hashing groups: the files of one run each (of several runs once there are more than
    // kMaxHashGroups runs; every file still waits for its own archive's number).
    constexpr size_t kMaxHashGroups = 64;
    std::vector<uint32_t> group_ends;
    {
        size_t runs = 0;
        for (size_t k = 0; k < arrivals.size(); ++k)
            runs += run_end_at[k] ? 1 : 0;
        const size_t runs_per_group = (runs + kMaxHashGroups - 1) / kMaxHashGroups;
        size_t file = 0, seen_runs = 0;
        for (size_t k = 0; k < arrivals.size(); ++k) {
            while (file < owner.size() && file_epoch[file] <= k + 1)
                ++file;
            if (run_end_at[k] && (++seen_runs % std::max<size_t>(1, runs_per_group) == 0 || k + 1 == arrivals.size() || k == 0))
                group_ends.push_back(static_cast<uint32_t>(file));
        }
        if (group_ends.empty() || group_ends.back() != owner.size())
            group_ends.push_back(static_cast<uint32_t>(owner.size()));
    }

    // 3. The SHA-1 kernel, waiting.  Everything it or the transfers need is allocated first.
    const size_t n_files = owner.size();
    const size_t table_bytes = n_files * sizeof(uint64_t);
    uint8_t *d_digests = nullptr;
    Sha1Arrival *d_arrival = static_cast<Sha1Arrival *>(s.d_flags);
    const bool prof = profiling_enabled();
    std::vector<hipEvent_t> marks;
    struct DestroyEvents {
        std::vector<hipEvent_t> &ev;
        ~DestroyEvents()
        {
            for (hipEvent_t e : ev)
                (void)hipEventDestroy(e);
        }
    } destroy_events{marks};
    for (size_t k = 0; k < 3 * arrivals.size(); ++k) {  // per arrival: transfers enqueued, decode begins, decode ends
        hipEvent_t e;
        SNAPPY_HIP_TRY(hipEventCreate(&e));
        marks.push_back(e);
    }
    if ((st = reserve_transfer_ring(s, kTransferSliceBytes, 2 * arrivals.size() + stream_bytes / kTransferSliceBytes + 1)) != SNAPPY_OK)
        return st;
    if (n_files) {
        // The tables, the arrival numbers and the digests share one buffer.
        if ((st = s.d_tables.ensure(2 * table_bytes + n_files * sizeof(uint32_t) + n_files * 20 + 64)) != SNAPPY_OK)
            return st;
        uint64_t *d_off = static_cast<uint64_t *>(s.d_tables.ptr);
        uint64_t *d_len = d_off + n_files;
        uint32_t *d_epochs = reinterpret_cast<uint32_t *>(d_len + n_files);
        d_digests = reinterpret_cast<uint8_t *>(d_epochs + n_files);
        std::vector<uint32_t> epochs(n_files);
        for (size_t k = 0; k < n_files; ++k)
            epochs[k] = file_epoch[k];
        Sha1Arrival zero = {};
        // ~1 s without progress before the kernel stops waiting.  Progress here is an archive's
        // transfer and decode, tens of milliseconds; where kernels cannot run side by side (a
        // profiler that serialises them) the decoder never starts, and the import should fall
        // back to the plain order soon rather than after the transfers' own 10 s.
        zero.max_idle_polls = 20000;
        if (const char *v = std::getenv("SNAPPY_ARRIVAL_MAX_POLLS"))
            zero.max_idle_polls = static_cast<uint32_t>(std::strtoul(v, nullptr, 10));
        SNAPPY_HIP_TRY(hipMemcpy(d_arrival, &zero, sizeof(zero), hipMemcpyHostToDevice));
        SNAPPY_HIP_TRY(hipMemcpyAsync(d_off, offsets.data(), table_bytes, hipMemcpyHostToDevice, s.stream));
        SNAPPY_HIP_TRY(hipMemcpyAsync(d_len, lengths.data(), table_bytes, hipMemcpyHostToDevice, s.stream));
        SNAPPY_HIP_TRY(hipMemcpyAsync(d_epochs, epochs.data(), n_files * sizeof(uint32_t), hipMemcpyHostToDevice,
                                      s.stream));
        st = enqueue_sha1_resident(lease, arena, d_off, d_len, n_files, d_digests, s.stream, d_epochs, d_arrival,
                                   group_ends.data(), static_cast<uint32_t>(group_ends.size()));
        if (st != SNAPPY_OK)
            return st;  // nothing was launched that waits
    }

    // 4. The archives: copy, decode, raise the counter.  From here on a kernel may be
    // waiting: whatever goes wrong, it has to be told.
    // The transfers follow each other on the copy stream.  The decodes go to a stream of
    // their own, each behind the transfers it needs (an event), so that a decode runs
    // beside the next archives' transfers instead of between them, and the host thread,
    // which waits for the copy stream when it feeds pageable memory through the ring,
    // does not wait for decodes.  A decode launch lasts as long as one chunk's chain of
    // sequences, however few chunks it has (0.3 ms on long copies, 3 ms on dense short
    // sequences), so archives are decoded in runs: the first archive alone (it holds the
    // largest file, whose SHA-1 chain decides when the import ends), the others in runs
    // of `run_len`, larger where a chunk takes long next to an archive's transfer.
    hipError_t failure = hipSuccess;
    std::vector<std::pair<size_t, size_t>> launches;  // (first, last + 1) arrivals per decode launch
    if (trace)
        std::fprintf(stderr, "[import] waiting kernel enqueued at %.1f ms\n", since());
    auto decode_run = [&](size_t from, size_t to) -> hipError_t {
        const uint32_t rows = static_cast<uint32_t>(first_row[to] - first_row[from]);
        // The run's last transfer has been enqueued: everything up to here on the copy stream.
        hipError_t e = hipEventRecord(marks[3 * from + 2], s.copy_stream);
        if (e == hipSuccess)
            e = hipStreamWaitEvent(s.decode_stream, marks[3 * from + 2], 0);
        if (e == hipSuccess && rows) {
            e = hipEventRecord(marks[3 * from], s.decode_stream);
            if (e == hipSuccess)
                e = launch_lz4_decode(arena, plan.in_alloc, arena, d_table + first_row[from], rows,
                                      d_queue + 16u * from, d_status + first_row[from], lease.dev->n_cu,
                                      s.decode_stream);
            if (e == hipSuccess)
                e = hipEventRecord(marks[3 * from + 1], s.decode_stream);
            if (e == hipSuccess)
                launches.emplace_back(from, to);
        }
        if (e == hipSuccess && n_files)
            e = hipMemcpyAsync(&d_arrival->arrived, &numbers[to - 1], sizeof(uint32_t), hipMemcpyHostToDevice,
                               s.decode_stream);
        return e;
    };
    // One transfer for all archives, in arrival order, through the same ring without a pause
    // between archives (a transfer per archive cost the pipeline's fill and drain 64 times:
    // 27 GB/s instead of 52); after each archive's last bytes the hook starts its run's decode.
    {
        std::vector<Piece> in_order;
        std::vector<size_t> group_begin;
        for (const Arrival &ar : arrivals) {
            group_begin.push_back(in_order.size());
            in_order.push_back(ar.piece);
        }
        size_t run_from = 0;
        const AfterGroup after_archive = [&](size_t k) -> hipError_t {
            hipError_t e = hipSuccess;
            if (run_end_at[k]) {
                e = decode_run(run_from, k + 1);
                run_from = k + 1;
            }
            if (trace)
                std::fprintf(stderr, "[import] arrival %zu (%.1f MB of archive %zu's stream, whose largest file has %.1f MB) handed to the copy stream at %.1f ms%s\n",
                             k, arrivals[k].piece.len / 1e6, arrivals[k].archive, largest[arrivals[k].archive] / 1e6, since(),
                             run_end_at[k] ? ", run ends" : "");
            return e;
        };
        if (!in_order.empty())
            failure = transfer_groups_to_device(s, hip_device, in_order, group_begin, kTransferSliceBytes, arena,
                                                after_archive);
    }
    if (failure == hipSuccess)
        failure = hipStreamSynchronize(s.copy_stream);
    if (trace)
        std::fprintf(stderr, "[import] copy stream drained at %.1f ms\n", since());
    if (failure == hipSuccess)
        failure = hipStreamSynchronize(s.decode_stream);
    if (trace)
        std::fprintf(stderr, "[import] decode stream drained at %.1f ms\n", since());
    if (failure != hipSuccess) {
        set_error_detail(std::string("MCDF import, transfer or decode: ") + hipGetErrorString(failure));
        (void)hipGetLastError();
        const uint32_t one = 1;
        (void)hipMemcpy(&d_arrival->abort, &one, sizeof(one), hipMemcpyHostToDevice);
        (void)hipStreamSynchronize(s.decode_stream);
        (void)hipStreamSynchronize(s.stream);  // the kernel leaves on abort (or on its own time limit)
        return SNAPPY_E_HIP;
    }
    std::vector<uint8_t> packed(n_files * 20);
    if (n_files) {
        uint64_t *d_off = static_cast<uint64_t *>(s.d_tables.ptr);
        uint64_t *d_len = d_off + n_files;
        SNAPPY_HIP_TRY(hipMemcpyAsync(packed.data(), d_digests, n_files * 20, hipMemcpyDeviceToHost, s.stream));
        SNAPPY_HIP_TRY(hipStreamSynchronize(s.stream));
        if (trace)
            std::fprintf(stderr, "[import] hashing done at %.1f ms (%zu files in %zu groups)\n", since(), n_files,
                         group_ends.size());
        Sha1Arrival seen;
        SNAPPY_HIP_TRY(hipMemcpy(&seen, d_arrival, sizeof(seen), hipMemcpyDeviceToHost));
        if (seen.error) {
            // The kernel stopped waiting; everything has been decoded since, so hash it in place.
            st = enqueue_sha1_resident(lease, arena, d_off, d_len, n_files, d_digests, s.stream);
            if (st != SNAPPY_OK)
                return st;
            SNAPPY_HIP_TRY(hipMemcpyAsync(packed.data(), d_digests, n_files * 20, hipMemcpyDeviceToHost, s.stream));
            SNAPPY_HIP_TRY(hipStreamSynchronize(s.stream));
        }
        finish_sha1_profile(lease);
    }
    if (prof) {
        // Decode kernel time of the call: the launches' own durations, added up.
        float total_ms = 0.f;
        for (const auto &run : launches) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, marks[3 * run.first], marks[3 * run.first + 1]) == hipSuccess)
                total_ms += ms;
            else
                (void)hipGetLastError();
        }
        if (!launches.empty())
            profile_store(1, total_ms);
    }
    st = finish_overlapped(s, jobs, states, digests_out, arrival_archive, first_row, ordered.size(), owner, packed);
    if (trace)
        std::fprintf(stderr, "[import] results sorted out at %.1f ms\n", since());
    return st;
}

// Decodes all archives, parses their headers and hashes every file body.
// digests_out[a] holds 20 bytes per Files entry of archive a (zero when the
// entry was skipped).  Returns an infrastructure status; per-archive results
// are in states[a].status.
int import_batch(SlotLease &lease, std::vector<StreamJob> &jobs, std::vector<ArchiveState> &states,
                 std::vector<std::vector<uint8_t>> &digests_out)
{
    const size_t n = jobs.size();
    states.assign(n, ArchiveState());
    digests_out.assign(n, std::vector<uint8_t>());
    // The callers have walked the chunk headers (parse_lz4net_stream): CopyTo runs to
    // the end of a stream before anything looks at the header, so stream errors come
    // first.
    uint64_t batch_bytes = 0;
    for (const StreamJob &j : jobs)
        batch_bytes += j.layout.decoded_size;
    lease.dev->work_units += n;
    lease.dev->work_bytes += batch_bytes;
    bool handled = false;
    int st = import_batch_overlapped(lease, jobs, states, digests_out, &handled);
    if (st != SNAPPY_OK || handled)
        return st;
    states.assign(n, ArchiveState());
    digests_out.assign(n, std::vector<uint8_t>());
    st = decode_streams_on_device(lease, jobs);
    if (st != SNAPPY_OK)
        return st;

    uint8_t *arena = static_cast<uint8_t *>(lease.slot->d_data.ptr);
    std::vector<uint64_t> offsets, lengths;
    std::vector<std::pair<size_t, size_t>> owner;  // (archive, file)
    for (size_t a = 0; a < n; ++a) {
        ArchiveState &as = states[a];
        as.status = jobs[a].status;
        if (as.status != SNAPPY_OK)
            continue;
        bool beyond = false;
        st = header_and_slices(arena, jobs[a], a, jobs[a].layout.decoded_size, as, digests_out[a], offsets, lengths,
                               owner, &beyond);
        if (st != SNAPPY_OK)
            return st;
    }
    if (!owner.empty()) {
        std::vector<uint8_t> packed(owner.size() * 20);
        st = run_sha1_tables(lease, arena, offsets.data(), lengths.data(), owner.size(),
                             packed.data(), nullptr);
        if (st != SNAPPY_OK)
            return st;
        for (size_t k = 0; k < owner.size(); ++k)
            std::memcpy(digests_out[owner[k].first].data() + 20 * owner[k].second,
                        packed.data() + 20 * k, 20);
    }
    return SNAPPY_OK;
}

}  // namespace
}  // namespace snappy

using namespace snappy;

extern "C" int snappy_mcdf_decode_and_hash(const uint8_t *in, uint64_t in_len, uint8_t *digests,
                                           uint8_t *statuses, uint64_t n_files_cap,
                                           uint64_t *n_files, uint64_t *body_offset,
                                           uint8_t *out_payload, uint64_t out_cap,
                                           uint64_t *payload_len)
{
    return abi_guard([&]() -> int {
        if ((!in && in_len) || !n_files || (n_files_cap && (!digests || !statuses)))
            return SNAPPY_E_INVALID_ARG;
        SlotLease lease;
        int st = lease.acquire(pick_device());
        if (st != SNAPPY_OK)
            return st;
        profile_reset();
        std::vector<StreamJob> jobs(1);
        jobs[0].in = in;
        jobs[0].in_len = in_len;
        parse_lz4net_stream(in, in_len, jobs[0].layout);
        std::vector<ArchiveState> states;
        std::vector<std::vector<uint8_t>> dig;
        st = import_batch(lease, jobs, states, dig);
        if (st != SNAPPY_OK)
            return st;
        if (jobs[0].status != SNAPPY_OK)
            return jobs[0].status;  // stream error: nothing else is defined
        const uint64_t size = jobs[0].layout.decoded_size;
        if (payload_len)
            *payload_len = size;
        if (states[0].status != SNAPPY_OK)
            return states[0].status;
        const size_t nf = states[0].slices.size();
        *n_files = nf;
        if (body_offset)
            *body_offset = states[0].info.body_offset;
        if (nf > n_files_cap)
            return SNAPPY_E_CAPACITY;
        for (size_t f = 0; f < nf; ++f)
            statuses[f] = states[0].slices[f].status;
        if (nf)
            std::memcpy(digests, dig[0].data(), nf * 20);
        if (out_payload) {
            if (out_cap < size)
                return SNAPPY_E_CAPACITY;
            if (size)
                SNAPPY_HIP_TRY(transfer_to_host(*lease.slot, lease.dev->hip_device, jobs[0].d_payload,
                                                out_payload, size));
        }
        return SNAPPY_OK;
    });
}

extern "C" int snappy_mcdf_decode_and_hash_batch(uint64_t n_archives, const uint8_t *const *ins,
                                                 const uint64_t *in_lens, uint8_t *digests,
                                                 uint8_t *statuses, uint64_t n_files_cap,
                                                 uint64_t *n_files, uint64_t *body_offsets,
                                                 uint64_t *payload_lens, int *archive_status)
{
    return abi_guard([&]() -> int {
        if (n_archives && (!ins || !in_lens || !n_files || !archive_status))
            return SNAPPY_E_INVALID_ARG;
        if (n_files_cap && (!digests || !statuses))
            return SNAPPY_E_INVALID_ARG;
        if (n_archives == 0)
            return SNAPPY_OK;
        std::vector<StreamJob> jobs(static_cast<size_t>(n_archives));
        for (uint64_t a = 0; a < n_archives; ++a) {
            if (!ins[a] && in_lens[a])
                return SNAPPY_E_INVALID_ARG;
            jobs[a].in = ins[a];
            jobs[a].in_len = in_lens[a];
        }
        int st = ensure_initialized();
        if (st != SNAPPY_OK)
            return st;
        profile_reset();
        // Shard by archive: an archive's payload stays on one device, so that decode and
        // hash never leave its HBM.  Longest first by decoded size, each archive to the
        // device with the least so far; one host thread and one slot per device.
        std::vector<uint64_t> weight(jobs.size());
        for (size_t a = 0; a < jobs.size(); ++a) {
            parse_lz4net_stream(jobs[a].in, jobs[a].in_len, jobs[a].layout);
            weight[a] = jobs[a].in_len + jobs[a].layout.decoded_size;
        }
        const int shards = static_cast<int>(
            std::min<uint64_t>(n_archives, static_cast<uint64_t>(std::max(1, device_count()))));
        std::vector<std::vector<size_t>> members(static_cast<size_t>(shards));
        {
            std::vector<size_t> order(jobs.size());
            for (size_t a = 0; a < order.size(); ++a)
                order[a] = a;
            std::stable_sort(order.begin(), order.end(),
                             [&](size_t x, size_t y) { return weight[x] > weight[y]; });
            std::vector<uint64_t> load(static_cast<size_t>(shards), 0);
            for (size_t a : order) {
                size_t best = 0;
                for (size_t k = 1; k < load.size(); ++k)
                    if (load[k] < load[best])
                        best = k;
                members[best].push_back(a);
                load[best] += weight[a];
            }
            for (auto &m : members)
                std::sort(m.begin(), m.end());  // archives of a device keep the caller's order
        }
        std::vector<ArchiveState> states(jobs.size());
        std::vector<std::vector<uint8_t>> dig(jobs.size());
        std::vector<int> shard_status(static_cast<size_t>(shards), SNAPPY_OK);
        std::vector<float> shard_ms(static_cast<size_t>(shards) * kProfileKernels, -1.f);
        const int first_device = shards == 1 ? pick_device() : 0;
        auto run_shard = [&](int d) {
            const std::vector<size_t> &mine = members[static_cast<size_t>(d)];
            if (mine.empty())
                return;
            shard_status[static_cast<size_t>(d)] = abi_guard([&]() -> int {
                SlotLease lease;
                int rc = lease.acquire(first_device + d);
                if (rc != SNAPPY_OK)
                    return rc;
                profile_reset();
                std::vector<StreamJob> sub(mine.size());
                for (size_t k = 0; k < mine.size(); ++k)
                    sub[k] = std::move(jobs[mine[k]]);
                std::vector<ArchiveState> sub_states;
                std::vector<std::vector<uint8_t>> sub_dig;
                rc = import_batch(lease, sub, sub_states, sub_dig);
                for (size_t k = 0; k < mine.size(); ++k)
                    jobs[mine[k]] = std::move(sub[k]);
                if (rc != SNAPPY_OK)
                    return rc;
                for (size_t k = 0; k < mine.size(); ++k) {
                    states[mine[k]] = std::move(sub_states[k]);
                    dig[mine[k]] = std::move(sub_dig[k]);
                }
                for (int q = 0; q < kProfileKernels; ++q)
                    shard_ms[static_cast<size_t>(d) * kProfileKernels + q] = snappy_profile_last_ms(q);
                return SNAPPY_OK;
            });
        };
        {
            std::vector<std::thread> workers;
            std::vector<int> inline_shards;
            for (int d = 1; d < shards; ++d)
                if (!try_spawn(workers, [&run_shard, d] { run_shard(d); }))
                    inline_shards.push_back(d);
            run_shard(0);
            for (int d : inline_shards)
                run_shard(d);
            for (auto &w : workers)
                w.join();
        }
        // Kernel times of the call: those of its slowest device.
        profile_reset();
        for (int q = 0; q < kProfileKernels; ++q) {
            float slowest = -1.f;
            for (int d = 0; d < shards; ++d)
                slowest = std::max(slowest, shard_ms[static_cast<size_t>(d) * kProfileKernels + q]);
            if (slowest >= 0.f)
                profile_store(q, slowest);
        }
        for (int v : shard_status)
            if (v != SNAPPY_OK)
                return v;
        uint64_t total = 0;
        for (uint64_t a = 0; a < n_archives; ++a) {
            archive_status[a] = states[a].status;
            const bool ok = states[a].status == SNAPPY_OK;
            n_files[a] = ok ? states[a].slices.size() : 0;
            if (body_offsets)
                body_offsets[a] = ok ? states[a].info.body_offset : 0;
            if (payload_lens)
                payload_lens[a] = jobs[a].status == SNAPPY_OK ? jobs[a].layout.decoded_size : 0;
            total += n_files[a];
        }
        if (total > n_files_cap)
            return SNAPPY_E_CAPACITY;
        uint64_t at = 0;
        for (uint64_t a = 0; a < n_archives; ++a) {
            for (uint64_t f = 0; f < n_files[a]; ++f)
                statuses[at + f] = states[a].slices[f].status;
            if (n_files[a])
                std::memcpy(digests + 20 * at, dig[a].data(), n_files[a] * 20);
            at += n_files[a];
        }
        return SNAPPY_OK;
    });
}
