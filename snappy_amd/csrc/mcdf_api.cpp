// Fused MCDF import data path: lz4net decode on the GPU, header parse on the
// host, SHA-1 of every file body where it lies in HBM.  Replaces the data flow
// of ImportMcdf (/root/reference/Snappy/Features/Mcdf/McdfManager.cs:72-84) and
// ExtractAndHashMapFiles (:28-66); the decoded payload never returns to the host
// unless the caller asks for it.  The batch form runs several archives through
// one decode launch and one hash launch: an archive is only 64-512 chunks, too
// few to occupy the GPU on its own.
#include <algorithm>
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
    int st = decode_streams_on_device(lease, jobs);
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
        const uint64_t size = jobs[a].layout.decoded_size;
        // Header: 9 bytes tell how much JSON follows.
        uint8_t head[9] = {0};
        const uint64_t head_len = size < 9 ? size : 9;
        if (head_len)
            SNAPPY_HIP_TRY(hipMemcpy(head, jobs[a].d_payload, head_len, hipMemcpyDeviceToHost));
        uint64_t need = 0;
        as.status = mcdf_header_extent(head, head_len, size, &need);
        if (as.status != SNAPPY_OK)
            continue;
        std::vector<uint8_t> header(static_cast<size_t>(need));
        SNAPPY_HIP_TRY(hipMemcpy(header.data(), jobs[a].d_payload, need, hipMemcpyDeviceToHost));
        as.status = parse_mcdf_header(header.data(), need, as.info);
        if (as.status != SNAPPY_OK)
            continue;
        as.status = mcdf_slice_files(as.info.files, as.info.body_offset, size, as.slices);
        if (as.status != SNAPPY_OK)
            continue;
        digests_out[a].assign(as.slices.size() * 20, 0);
        for (size_t f = 0; f < as.slices.size(); ++f) {
            if (as.slices[f].status != SNAPPY_FILE_HASHED)
                continue;
            offsets.push_back(static_cast<uint64_t>(jobs[a].d_payload - arena) + as.slices[f].offset);
            lengths.push_back(as.slices[f].length);
            owner.emplace_back(a, f);
        }
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
