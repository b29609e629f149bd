// Fused MCDF import data path: lz4net decode on the GPU, header parse on the
// host, SHA-1 of every file body where it lies in HBM.  Replaces the data flow
// of ImportMcdf (/root/reference/Snappy/Features/Mcdf/McdfManager.cs:72-84) and
// ExtractAndHashMapFiles (:28-66); the decoded payload never returns to the host
// unless the caller asks for it.
#include <cstring>

#include "lz4net_host.hpp"
#include "mcdf_host.hpp"
#include "sha1_api.hpp"

using namespace snappy;

extern "C" int snappy_mcdf_decode_and_hash(const uint8_t *in, uint64_t in_len, uint8_t *digests,
                                           uint8_t *statuses, uint64_t n_files_cap,
                                           uint64_t *n_files, uint64_t *body_offset,
                                           uint8_t *out_payload, uint64_t out_cap,
                                           uint64_t *payload_len)
{
    if ((!in && in_len) || !n_files || (n_files_cap && (!digests || !statuses)))
        return SNAPPY_E_INVALID_ARG;
    // CopyTo runs to the end of the stream before anything looks at the header,
    // so stream errors come first.
    Lz4netLayout layout;
    parse_lz4net_stream(in, in_len, layout);

    SlotLease lease;
    int st = lease.acquire(pick_device());
    if (st != SNAPPY_OK)
        return st;
    profile_reset();
    DecodedOnDevice dec;
    st = decode_lz4net_on_device(lease, in, in_len, layout, dec);
    if (st != SNAPPY_OK)
        return st;
    if (payload_len)
        *payload_len = dec.size;

    // Header: 9 bytes tell how much JSON follows.
    uint8_t head[9] = {0};
    const uint64_t head_len = dec.size < 9 ? dec.size : 9;
    if (head_len)
        SNAPPY_HIP_TRY(hipMemcpy(head, dec.d_payload, head_len, hipMemcpyDeviceToHost));
    uint64_t need = 0;
    st = mcdf_header_extent(head, head_len, dec.size, &need);
    if (st != SNAPPY_OK)
        return st;
    std::vector<uint8_t> header(static_cast<size_t>(need));
    SNAPPY_HIP_TRY(hipMemcpy(header.data(), dec.d_payload, need, hipMemcpyDeviceToHost));
    McdfHeaderInfo info;
    st = parse_mcdf_header(header.data(), need, info);
    if (st != SNAPPY_OK)
        return st;
    *n_files = info.files.size();
    if (body_offset)
        *body_offset = info.body_offset;
    if (info.files.size() > n_files_cap)
        return SNAPPY_E_CAPACITY;

    std::vector<McdfSlice> slices;
    st = mcdf_slice_files(info.files, info.body_offset, dec.size, slices);
    if (st != SNAPPY_OK)
        return st;

    std::vector<uint64_t> offsets, lengths;
    std::vector<size_t> owner;
    for (size_t i = 0; i < slices.size(); ++i) {
        statuses[i] = slices[i].status;
        std::memset(digests + 20 * i, 0, 20);
        if (slices[i].status == SNAPPY_FILE_HASHED) {
            offsets.push_back(slices[i].offset);
            lengths.push_back(slices[i].length);
            owner.push_back(i);
        }
    }
    if (!owner.empty()) {
        Sha1Plan plan;
        build_sha1_plan(offsets.data(), lengths.data(), owner.size(), lease.dev->n_cu, plan);
        std::vector<uint8_t> packed(owner.size() * 20);
        st = run_sha1_plan(lease, dec.d_payload, plan, packed.data(), nullptr);
        if (st != SNAPPY_OK)
            return st;
        for (size_t k = 0; k < owner.size(); ++k)
            std::memcpy(digests + 20 * owner[k], packed.data() + 20 * k, 20);
    }
    if (out_payload) {
        if (out_cap < dec.size)
            return SNAPPY_E_CAPACITY;
        if (dec.size)
            SNAPPY_HIP_TRY(hipMemcpy(out_payload, dec.d_payload, dec.size, hipMemcpyDeviceToHost));
    }
    return SNAPPY_OK;
}
