// Fused MCDF import data path: lz4net decode on the GPU, header parse on the
// host, SHA-1 of every file body where it lies in HBM.  Replaces the data flow
// of ImportMcdf (/root/reference/Snappy/Features/Mcdf/McdfManager.cs:72-84) and
// ExtractAndHashMapFiles (:28-66); the decoded payload never returns to the host
// unless the caller asks for it.  The batch form runs several archives through
// one decode launch and one hash launch: an archive is only 64-512 chunks, too
// few to occupy the GPU on its own.
#include <cstring>

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
    // CopyTo runs to the end of a stream before anything looks at the header, so
    // stream errors come first.
    for (StreamJob &j : jobs)
        parse_lz4net_stream(j.in, j.in_len, j.layout);
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
}

extern "C" int snappy_mcdf_decode_and_hash_batch(uint64_t n_archives, const uint8_t *const *ins,
                                                 const uint64_t *in_lens, uint8_t *digests,
                                                 uint8_t *statuses, uint64_t n_files_cap,
                                                 uint64_t *n_files, uint64_t *body_offsets,
                                                 uint64_t *payload_lens, int *archive_status)
{
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
    SlotLease lease;
    int st = lease.acquire(pick_device());
    if (st != SNAPPY_OK)
        return st;
    profile_reset();
    std::vector<ArchiveState> states;
    std::vector<std::vector<uint8_t>> dig;
    st = import_batch(lease, jobs, states, dig);
    if (st != SNAPPY_OK)
        return st;
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
}
