// SHA-1 entry points of libsnappyhash: batch planning, staging, dispatch.
//
// Mirrors /root/reference/Snappy/Utils/Crypto.cs:8-22 (GetHash, ToHashedString)
// and the three loops that call it (Snappy/Core/SnapshotManager.cs:312-324,
// Snappy/Features/Mcdf/McdfManager.cs:36-63, Snappy/Data/SnapshotMigrator.cs:
// 62-86).  All hashing runs in the HIP kernels of sha1_kernels.hip; there is no
// host implementation of SHA-1 in this library.
#include "sha1_api.hpp"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <thread>

namespace snappy {

namespace {

// Measured cost model, used only to split a batch between the two kernels:
// nanoseconds one message spends per 64-byte block when its wavefront has a
// SIMD to itself (both paths issue one instruction per ~4.2 cycles; the wave
// path has 412 instructions per block on the chain, the lane path ~650 plus
// what its 63 neighbours' tails cost).
constexpr double kNsPerBlockWave = 714.0;
constexpr double kNsPerBlockLane = 1300.0;
// SIMD time one block costs on the lane path when a SIMD is saturated with
// waves (64 messages share an instruction stream at ~2.5 cycles/instruction).
constexpr double kSimdNsPerBlockLane = 650.0 * 2.5 / 64.0 / 2.4;

inline uint64_t padded_blocks(uint64_t len) { return (len + 8) / 64 + 1; }

long env_long(const char *name, long fallback)
{
    const char *v = std::getenv(name);
    return (v && *v) ? std::atol(v) : fallback;
}

// Descending size order without a comparison sort: 8 buckets per octave of the
// padded block count (order inside a bucket does not matter to the schedule).
inline uint32_t size_bucket(uint64_t blocks)
{
    const int e = 63 - __builtin_clzll(blocks | 1);
    const uint32_t mant = (e >= 3) ? static_cast<uint32_t>((blocks >> (e - 3)) & 7)
                                   : static_cast<uint32_t>((blocks << (3 - e)) & 7);
    return static_cast<uint32_t>(e) * 8u + mant;
}

}  // namespace

void build_sha1_plan(const uint64_t *offsets, const uint64_t *lengths, uint64_t n, int n_cu,
                     Sha1Plan &plan)
{
    constexpr uint32_t kBuckets = 64 * 8;
    std::vector<uint32_t> count(kBuckets + 1, 0);
    uint64_t total_blocks = 0, max_blocks = 0;
    for (uint64_t i = 0; i < n; ++i) {
        const uint64_t b = padded_blocks(lengths[i]);
        total_blocks += b;
        max_blocks = std::max(max_blocks, b);
        ++count[kBuckets - 1 - size_bucket(b)];  // reversed: big buckets first
    }
    uint32_t run = 0;
    for (uint32_t k = 0; k <= kBuckets; ++k) {
        const uint32_t c = count[k];
        count[k] = run;
        run += c;
    }
    plan.tasks.resize(n);
    for (uint64_t i = 0; i < n; ++i) {
        const uint64_t b = padded_blocks(lengths[i]);
        Sha1Task &t = plan.tasks[count[kBuckets - 1 - size_bucket(b)]++];
        t.offset = offsets[i];
        t.len = lengths[i];
        t.out_index = static_cast<uint32_t>(i);
        t.reserved = 0;
    }
    plan.total_blocks = total_blocks;

    // Split.  The batch cannot finish before its largest message has run its
    // serial chain on the faster (wave) path, nor before the lane path has issued
    // all its blocks.  A message goes to the wave path when the lane path would
    // keep it past ~80 % of that bound; when there are no more messages than
    // SIMDs every message gets a wave (and a SIMD) of its own.
    const uint64_t n_simd = 4ull * static_cast<uint64_t>(n_cu);
    const double t_star = std::max(static_cast<double>(max_blocks) * kNsPerBlockWave,
                                   static_cast<double>(total_blocks) * kSimdNsPerBlockLane /
                                       static_cast<double>(n_simd));
    double thr_blocks = (n <= n_simd) ? 0.0 : 0.8 * t_star / kNsPerBlockLane;
    const long forced = env_long("SNAPPY_WAVE_MIN_BLOCKS", -1);  // tuning/testing override
    if (forced >= 0)
        thr_blocks = static_cast<double>(forced);
    const uint64_t max_wave_tasks = 2 * n_simd;
    uint32_t n_wave = 0;
    while (n_wave < n && n_wave < max_wave_tasks &&
           static_cast<double>(padded_blocks(plan.tasks[n_wave].len)) > thr_blocks)
        ++n_wave;
    plan.n_wave = n_wave;
}

int run_sha1_plan(SlotLease &lease, const uint8_t *d_base, const Sha1Plan &plan,
                  uint8_t *digests_host, hipStream_t user_stream)
{
    Slot &s = *lease.slot;
    const uint64_t n = plan.tasks.size();
    if (n == 0)
        return SNAPPY_OK;
    hipStream_t stream = user_stream ? user_stream : s.stream;

    const size_t task_bytes = n * sizeof(Sha1Task);
    const size_t digest_bytes = n * 20;
    int st;
    if ((st = s.d_tasks.ensure(task_bytes)) != SNAPPY_OK) return st;
    if ((st = s.d_out.ensure(digest_bytes)) != SNAPPY_OK) return st;
    if ((st = s.d_misc.ensure(256)) != SNAPPY_OK) return st;
    if ((st = s.h_stage.ensure(std::max(task_bytes, digest_bytes))) != SNAPPY_OK) return st;

    std::memcpy(s.h_stage.ptr, plan.tasks.data(), task_bytes);
    SNAPPY_HIP_TRY(hipMemcpyAsync(s.d_tasks.ptr, s.h_stage.ptr, task_bytes, hipMemcpyHostToDevice,
                                  stream));
    SNAPPY_HIP_TRY(hipMemsetAsync(s.d_misc.ptr, 0, 256, stream));

    const uint32_t n_wave = plan.n_wave;
    const uint32_t n_all = static_cast<uint32_t>(n);
    uint32_t *q_wave = static_cast<uint32_t *>(s.d_misc.ptr);
    uint32_t *q_lane = q_wave + 32;  // separate 128-byte line
    const Sha1Task *d_tasks = static_cast<const Sha1Task *>(s.d_tasks.ptr);
    uint8_t *d_digests = static_cast<uint8_t *>(s.d_out.ptr);

    const bool prof = profiling_enabled();
    if (prof)
        SNAPPY_HIP_TRY(hipEventRecord(s.ev_prof[0][0], stream));

    const int wave_groups =
        static_cast<int>(std::min<uint64_t>(n_wave, static_cast<uint64_t>(lease.dev->n_cu) * 8u));
    const uint32_t n_lane = n_all - n_wave;
    const long blocks_per_cu = std::max(1L, env_long("SNAPPY_LANE_BLOCKS_PER_CU", 8));
    const int lane_blocks = static_cast<int>(std::min<uint64_t>(
        (static_cast<uint64_t>(n_lane) + 255) / 256,
        static_cast<uint64_t>(lease.dev->n_cu) * static_cast<uint64_t>(blocks_per_cu)));

    if (n_wave > 0 && n_lane > 0) {
        // Both kernels run side by side: the wave kernel on the side stream.
        SNAPPY_HIP_TRY(hipEventRecord(s.ev_fork, stream));
        SNAPPY_HIP_TRY(hipStreamWaitEvent(s.side_stream, s.ev_fork, 0));
        SNAPPY_HIP_TRY(launch_sha1_wave(d_base, d_tasks, n_wave, q_wave, d_digests, wave_groups,
                                        s.side_stream));
        SNAPPY_HIP_TRY(launch_sha1_lanes(d_base, d_tasks, n_wave, n_all, q_lane, d_digests,
                                         lane_blocks, stream));
        SNAPPY_HIP_TRY(hipEventRecord(s.ev_join, s.side_stream));
        SNAPPY_HIP_TRY(hipStreamWaitEvent(stream, s.ev_join, 0));
    } else if (n_wave > 0) {
        SNAPPY_HIP_TRY(launch_sha1_wave(d_base, d_tasks, n_wave, q_wave, d_digests, wave_groups,
                                        stream));
    } else {
        SNAPPY_HIP_TRY(launch_sha1_lanes(d_base, d_tasks, 0, n_all, q_lane, d_digests, lane_blocks,
                                         stream));
    }
    if (prof)
        SNAPPY_HIP_TRY(hipEventRecord(s.ev_prof[0][1], stream));

    SNAPPY_HIP_TRY(hipMemcpyAsync(s.h_stage.ptr, d_digests, digest_bytes, hipMemcpyDeviceToHost,
                                  stream));
    SNAPPY_HIP_TRY(hipStreamSynchronize(stream));
    std::memcpy(digests_host, s.h_stage.ptr, digest_bytes);
    if (prof) {
        float ms = -1.f;
        if (hipEventElapsedTime(&ms, s.ev_prof[0][0], s.ev_prof[0][1]) == hipSuccess)
            profile_store(0, ms);
    }
    return SNAPPY_OK;
}

namespace {

struct HostRange {
    const uint8_t *ptr;
    uint64_t len;
};

// Copies the ranges [first, last) into the slot's device arena and hashes them
// on the slot's device.  Ranges that touch in host memory are moved by one copy
// and keep their relative layout; others are packed at 16-byte aligned offsets.
int hash_ranges_on_device(int device, const HostRange *ranges, uint64_t first, uint64_t last,
                          uint8_t *digests)
{
    SlotLease lease;
    int st = lease.acquire(device);
    if (st != SNAPPY_OK)
        return st;
    Slot &s = *lease.slot;
    const uint64_t n = last - first;

    std::vector<uint64_t> dev_off(n), lens(n);
    struct Copy {
        const uint8_t *src;
        uint64_t dst, len;
    };
    std::vector<Copy> copies;
    uint64_t cursor = 0;
    for (uint64_t i = 0; i < n; ++i) {
        const HostRange &r = ranges[first + i];
        lens[i] = r.len;
        if (r.len == 0) {
            dev_off[i] = cursor;
            continue;
        }
        if (!copies.empty() && copies.back().src + copies.back().len == r.ptr) {
            dev_off[i] = copies.back().dst + copies.back().len;
            copies.back().len += r.len;
        } else {
            cursor = (cursor + 15) & ~uint64_t(15);
            dev_off[i] = cursor;
            copies.push_back({r.ptr, cursor, r.len});
        }
        cursor = dev_off[i] + r.len;
    }
    if ((st = s.d_data.ensure(static_cast<size_t>(cursor) + 64)) != SNAPPY_OK)
        return st;
    uint8_t *d_base = static_cast<uint8_t *>(s.d_data.ptr);
    for (const Copy &c : copies)
        SNAPPY_HIP_TRY(hipMemcpyAsync(d_base + c.dst, c.src, c.len, hipMemcpyHostToDevice,
                                      s.stream));
    Sha1Plan plan;
    build_sha1_plan(dev_off.data(), lens.data(), n, lease.dev->n_cu, plan);
    return run_sha1_plan(lease, d_base, plan, digests + 20 * first, nullptr);
}

}  // namespace

// Contiguous groups of near-equal byte count: cuts[k]..cuts[k+1] is shard k.
void plan_shards(const uint64_t *lengths, uint64_t n, uint32_t n_shards, uint64_t *cuts)
{
    uint64_t total = 0;
    for (uint64_t i = 0; i < n; ++i)
        total += lengths[i];
    for (uint32_t k = 0; k <= n_shards; ++k)
        cuts[k] = n;
    cuts[0] = 0;
    uint64_t acc = 0;
    uint32_t k = 1;
    for (uint64_t i = 0; i < n && k < n_shards; ++i) {
        acc += lengths[i];
        while (k < n_shards && acc * n_shards >= total * k)
            cuts[k++] = i + 1;
    }
}

namespace {

// Files are independent, so a batch shards across devices with no exchange:
// contiguous groups of near-equal byte count, one host thread per device.
int hash_ranges(const std::vector<HostRange> &ranges, uint8_t *digests)
{
    const uint64_t n = ranges.size();
    if (n == 0)
        return SNAPPY_OK;
    if (n >= 0xFFFFFFF0ull)
        return SNAPPY_E_INVALID_ARG;
    int st = ensure_initialized();
    if (st != SNAPPY_OK)
        return st;
    profile_reset();
    const int n_dev = device_count();
    uint64_t total = 0;
    for (const HostRange &r : ranges)
        total += r.len;
    const uint64_t min_shard_bytes = 8ull << 20;
    int shards = static_cast<int>(
        std::min<uint64_t>(std::min<uint64_t>(n, static_cast<uint64_t>(n_dev)),
                           std::max<uint64_t>(1, total / min_shard_bytes)));
    if (shards <= 1)
        return hash_ranges_on_device(pick_device(), ranges.data(), 0, n, digests);

    std::vector<uint64_t> lens(n);
    for (uint64_t i = 0; i < n; ++i)
        lens[i] = ranges[i].len;
    std::vector<uint64_t> cut(static_cast<size_t>(shards) + 1, n);
    plan_shards(lens.data(), n, static_cast<uint32_t>(shards), cut.data());
    std::vector<int> status(static_cast<size_t>(shards), SNAPPY_OK);
    std::vector<std::thread> workers;
    for (int d = 0; d < shards; ++d) {
        workers.emplace_back([&, d] {
            const uint64_t a = cut[static_cast<size_t>(d)], b = cut[static_cast<size_t>(d) + 1];
            if (b > a)
                status[static_cast<size_t>(d)] =
                    hash_ranges_on_device(d, ranges.data(), a, b, digests);
        });
    }
    for (auto &w : workers)
        w.join();
    for (int v : status)
        if (v != SNAPPY_OK)
            return v;
    return SNAPPY_OK;
}

}  // namespace
}  // namespace snappy

using namespace snappy;

extern "C" int snappy_plan_shards(const uint64_t *lengths, uint64_t n, uint32_t n_shards,
                                  uint64_t *cuts)
{
    if ((n && !lengths) || !cuts || n_shards == 0)
        return SNAPPY_E_INVALID_ARG;
    plan_shards(lengths, n, n_shards, cuts);
    return SNAPPY_OK;
}

extern "C" int snappy_sha1(const uint8_t *data, uint64_t len, uint8_t out20[20])
{
    if (!out20 || (!data && len))
        return SNAPPY_E_INVALID_ARG;
    std::vector<HostRange> r{{data, len}};
    return hash_ranges(r, out20);
}

extern "C" int snappy_digests_to_hex(const uint8_t *digests, uint64_t n, char *hex)
{
    if ((!digests || !hex) && n)
        return SNAPPY_E_INVALID_ARG;
    static const char kDigits[] = "0123456789ABCDEF";  // "X2": uppercase
    for (uint64_t i = 0; i < n; ++i) {
        for (int b = 0; b < 20; ++b) {
            hex[41 * i + 2 * b] = kDigits[digests[20 * i + b] >> 4];
            hex[41 * i + 2 * b + 1] = kDigits[digests[20 * i + b] & 15];
        }
        hex[41 * i + 40] = '\0';
    }
    return SNAPPY_OK;
}

extern "C" int snappy_sha1_hex(const uint8_t *data, uint64_t len, char out41[41])
{
    if (!out41)
        return SNAPPY_E_INVALID_ARG;
    uint8_t d[20];
    int st = snappy_sha1(data, len, d);
    if (st != SNAPPY_OK)
        return st;
    return snappy_digests_to_hex(d, 1, out41);
}

extern "C" int snappy_sha1_batch(const uint8_t *base, const uint64_t *offsets,
                                 const uint64_t *lengths, uint64_t n, uint8_t *digests)
{
    if (n && (!offsets || !lengths || !digests))
        return SNAPPY_E_INVALID_ARG;
    std::vector<HostRange> r(n);
    for (uint64_t i = 0; i < n; ++i) {
        if (lengths[i] && !base)
            return SNAPPY_E_INVALID_ARG;
        r[i] = {base ? base + offsets[i] : nullptr, lengths[i]};
    }
    return hash_ranges(r, digests);
}

extern "C" int snappy_sha1_files(const uint8_t *const *datas, const uint64_t *lengths, uint64_t n,
                                 uint8_t *digests)
{
    if (n && (!datas || !lengths || !digests))
        return SNAPPY_E_INVALID_ARG;
    std::vector<HostRange> r(n);
    for (uint64_t i = 0; i < n; ++i) {
        if (lengths[i] && !datas[i])
            return SNAPPY_E_INVALID_ARG;
        r[i] = {datas[i], lengths[i]};
    }
    return hash_ranges(r, digests);
}

extern "C" int snappy_sha1_batch_device(int device, const void *d_base, const uint64_t *offsets,
                                        const uint64_t *lengths, uint64_t n, uint8_t *digests,
                                        void *stream)
{
    if (n && (!offsets || !lengths || !digests))
        return SNAPPY_E_INVALID_ARG;
    if (n >= 0xFFFFFFF0ull)
        return SNAPPY_E_INVALID_ARG;
    if (n == 0)
        return SNAPPY_OK;
    for (uint64_t i = 0; i < n; ++i)
        if (lengths[i] && !d_base)
            return SNAPPY_E_INVALID_ARG;
    SlotLease lease;
    int st = lease.acquire(device);
    if (st != SNAPPY_OK)
        return st;
    profile_reset();
    Sha1Plan plan;
    build_sha1_plan(offsets, lengths, n, lease.dev->n_cu, plan);
    return run_sha1_plan(lease, static_cast<const uint8_t *>(d_base), plan, digests,
                         static_cast<hipStream_t>(stream));
}
