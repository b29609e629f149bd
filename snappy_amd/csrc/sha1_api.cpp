// SHA-1 entry points of libsnappyhash: staging, device-side planning, dispatch.
//
// Mirrors /root/reference/Snappy/Utils/Crypto.cs:8-22 (GetHash, ToHashedString)
// and the three loops that call it (Snappy/Core/SnapshotManager.cs:312-324,
// Snappy/Features/Mcdf/McdfManager.cs:36-63, Snappy/Data/SnapshotMigrator.cs:
// 62-86).  All hashing runs in the HIP kernels of sha1_kernels.hip; there is no
// host implementation of SHA-1 in this library.
#include "sha1_api.hpp"
#include "transfer.hpp"

#include <algorithm>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <thread>

namespace snappy {

namespace {

// Cost model, used only to cut a batch into the kernel's paths (groups of 64 / 16
// / 4 / 1 lanes per message).  chain: nanoseconds a message spends per 64-byte
// block when its wavefront has a SIMD to itself (one instruction issues per ~4.1
// cycles; 412 instructions per block on the chain plus the schedule expansion
// shared by the group).  The SIMD time a block costs is the chain time divided by
// the messages a wave carries.
//
// The table is what one MI355X gave with the other SIMDs busy (tools/probe_planner.py,
// tools/probe_one_lane_chain.py; alone on the chip the chains run at 705 / 775 / 900 /
// 1190 ns).  What differs from box to box is the level, so that is what is measured: the
// 64-lane chain, one message alone on the GPU at two sizes, the first time a device hashes
// (about 7 ms once per device and process), and the whole table is scaled by what comes
// out against kIdleChainNs.  SNAPPY_SHA1_CALIBRATE=0 switches the measurement off; a
// result outside 0.6-1.6 times the expected one (a busy GPU) is not used.
constexpr double kChainNs[kSha1Paths] = {716.0, 795.0, 960.0, 1300.0};
constexpr double kIdleChainNs = 709.0;  // the 64-lane chain on the same MI355X with nothing else running
// The one-lane path streams 64 files per wave, 64 bytes per lane and block.  With a wave's
// files next to each other in memory its chain runs at 1190-1300 ns per block whatever the
// file size; with the files of a wave scattered over the batch (the planner orders by size,
// not by address) it ran at 1490 ns per block over 28 GB, 1880 over 112 GB and 2070 over
// 140 GB, unsteadily (tools/probe_one_lane_big.py), and at up to 2800 next to wide-group
// waves on a 100 GB corpus, where such waves then decided the finishing time.  So the chain
// is priced by the batch's address span, up to a cap that sends multi-megabyte files of a
// large corpus to the 4-lane path (900-915 ns per block wherever its files lie).
constexpr double kOneLaneNsPerGb = 17.0;
constexpr double kOneLaneCapNs = 3000.0;
constexpr double kOneLaneOrderNs = 1800.0;

// Chains measured next to a chip full of other work are 1-3 % slower than in the table's
// conditions; no further allowance is made (a factor of up to 1.45 was used before the
// launches were made one wave per SIMD for certain: it had absorbed the launches in
// which two waves shared a SIMD).
constexpr double kLoadSlowdown = 0.0;

long env_long(const char *name, long fallback)
{
    const char *v = std::getenv(name);
    return (v && *v) ? std::atol(v) : fallback;
}

}  // namespace

Sha1CostModel sha1_cost_model(const DeviceCtx &dev, bool waits_for_transfer)
{
    const uint32_t n_simd = 4u * static_cast<uint32_t>(dev.n_cu);
    Sha1CostModel model;
    const double level = dev.cost_measured.load(std::memory_order_acquire) ? dev.chain_level : 1.0;
    for (int k = 0; k < kSha1Paths; ++k)
        model.chain_ns[k] = kChainNs[k] * level;
    model.one_lane_ns_per_gb = kOneLaneNsPerGb * level;
    model.one_lane_cap_ns = kOneLaneCapNs * level;
    // For the hand-out order the one-lane chain counts with what it usually takes, not with the cap.
    model.one_lane_order_ns =
        static_cast<double>(env_long("SNAPPY_SHA1_ONE_LANE_ORDER_NS", static_cast<long>(kOneLaneOrderNs))) * level;
    if (const char *v = std::getenv("SNAPPY_SHA1_CHAIN_NS")) {  // tuning: "c0,c1,c2,c3[,per GB,cap]"
        const int got = std::sscanf(v, "%lf,%lf,%lf,%lf,%lf,%lf", &model.chain_ns[0], &model.chain_ns[1],
                                    &model.chain_ns[2], &model.chain_ns[3], &model.one_lane_ns_per_gb,
                                    &model.one_lane_cap_ns);
        if (got < 6) {  // a fixed one-lane figure
            model.one_lane_ns_per_gb = 0.0;
            model.one_lane_cap_ns = model.chain_ns[3];
        }
    }
    for (int k = 0; k < kSha1Paths; ++k)
        model.simd_ns[k] = model.chain_ns[k] * kHopGroupLanes[k] / 64.0;
    model.n_simd = n_simd;
    model.load_slowdown = kLoadSlowdown;
    if (const char *v = std::getenv("SNAPPY_SHA1_LOAD_SLOWDOWN"))  // tuning
        model.load_slowdown = std::atof(v);
    // SNAPPY_SHA1_FORCE_PATH = 0..3 sends every message down one path (tests, tuning).
    model.forced_path = static_cast<int>(env_long("SNAPPY_SHA1_FORCE_PATH", -1));
    if (model.forced_path >= kSha1Paths)
        model.forced_path = -1;
    model.forced_waves = static_cast<int>(env_long("SNAPPY_SHA1_WAVES_PER_SIMD", 0));
    // Waves that wait for a transfer must leave registers free on every SIMD: small
    // copies are carried out by copy kernels, which would otherwise never be placed
    // (two of these waves fill a SIMD's register file).
    if (waits_for_transfer)
        model.forced_waves = 1;
    return model;
}

namespace {

// One message on 64 lanes, alone on the GPU, at two sizes: the difference of the
// two times over the difference in blocks is the chain time per block, free of
// launch and planning overhead.  Runs on the slot's own stream with a temporary
// buffer; failures leave the table in charge.
void measure_chain_times(SlotLease &lease)
{
    DeviceCtx &dev = *lease.dev;
    if (env_long("SNAPPY_SHA1_CALIBRATE", 1) == 0)
        return;
    constexpr uint64_t kBlocks[2] = {512, 2560};
    constexpr size_t kDataBytes = kBlocks[1] * 64;
    constexpr size_t kStateAt = (kDataBytes + 255) & ~size_t(255);
    constexpr size_t kTaskAt = kStateAt + ((sizeof(Sha1PlanState) + 255) & ~size_t(255));
    constexpr size_t kTablesAt = kTaskAt + 256;  // offset, length, digest
    uint8_t *area = nullptr;
    hipEvent_t ev[2] = {nullptr, nullptr};
    auto done = [&] {
        for (hipEvent_t e : ev)
            if (e)
                (void)hipEventDestroy(e);
        if (area)
            (void)hipFree(area);
        (void)hipGetLastError();
    };
    if (hipMalloc(&area, kTablesAt + 256) != hipSuccess || hipEventCreate(&ev[0]) != hipSuccess ||
        hipEventCreate(&ev[1]) != hipSuccess)
        return done();
    hipStream_t stream = lease.slot->stream;
    Sha1PlanState *d_state = reinterpret_cast<Sha1PlanState *>(area + kStateAt);
    Sha1Task *d_task = reinterpret_cast<Sha1Task *>(area + kTaskAt);
    uint64_t *d_off = reinterpret_cast<uint64_t *>(area + kTablesAt);
    uint64_t *d_len = d_off + 1;
    uint8_t *d_digest = reinterpret_cast<uint8_t *>(d_off + 2);
    if (hipMemsetAsync(area, 0x5A, kTablesAt + 256, stream) != hipSuccess ||
        hipMemsetAsync(d_off, 0, 8, stream) != hipSuccess)
        return done();
    const uint32_t n_simd = 4u * static_cast<uint32_t>(dev.n_cu);
    Sha1CostModel model = {};
    for (int k = 0; k < kSha1Paths; ++k)
        model.chain_ns[k] = model.simd_ns[k] = kChainNs[k];
    model.one_lane_cap_ns = kChainNs[kSha1Paths - 1];
    model.n_simd = n_simd;
    model.forced_waves = 1;
    double measured = 0.0;
    model.forced_path = 0;
    for (int pass = 0; pass < 2; ++pass) {  // the first pass only brings the clock up
        float ms[2] = {0.f, 0.f};
        for (int size = 0; size < 2; ++size) {
            const uint64_t len = kBlocks[size] * 64 - 9;  // pads to exactly kBlocks[size] blocks
            if (hipMemcpyAsync(d_len, &len, sizeof len, hipMemcpyHostToDevice, stream) != hipSuccess ||
                hipStreamSynchronize(stream) != hipSuccess || hipEventRecord(ev[0], stream) != hipSuccess ||
                launch_sha1_plan(d_off, d_len, nullptr, 1, d_state, d_task, model, stream) != hipSuccess ||
                launch_sha1_hop(area, d_task, d_state, d_digest, nullptr, 1, n_simd, 1, stream) != hipSuccess ||
                hipEventRecord(ev[1], stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess ||
                hipEventElapsedTime(&ms[size], ev[0], ev[1]) != hipSuccess)
                return done();
        }
        measured = static_cast<double>(ms[1] - ms[0]) * 1e6 / static_cast<double>(kBlocks[1] - kBlocks[0]);
    }
    done();
    if (!(measured > 0.6 * kIdleChainNs && measured < 1.6 * kIdleChainNs))
        return;  // not a figure to plan with
    dev.chain_level = measured / kIdleChainNs;
    dev.cost_measured.store(true, std::memory_order_release);
}

}  // namespace

int enqueue_sha1_resident(SlotLease &lease, const uint8_t *d_base, const uint64_t *d_offsets,
                          const uint64_t *d_lengths, uint64_t n, uint8_t *d_digests,
                          hipStream_t stream, const uint32_t *d_epochs, Sha1Arrival *d_arrival,
                          const uint32_t *group_ends, uint32_t n_groups)
{
    Slot &s = *lease.slot;
    if (n == 0)
        return SNAPPY_OK;
    if (n >= 0xFFFFFFF0ull)
        return SNAPPY_E_INVALID_ARG;
    if (group_ends == nullptr)
        n_groups = 1;
    if (n_groups == 0 || (group_ends && group_ends[n_groups - 1] != n))
        return SNAPPY_E_INVALID_ARG;
    int st;
    if ((st = s.d_tasks.ensure(n * sizeof(Sha1Task))) != SNAPPY_OK) return st;
    if ((st = s.d_misc.ensure(n_groups * sizeof(Sha1PlanState))) != SNAPPY_OK) return st;
    Sha1PlanState *d_state = static_cast<Sha1PlanState *>(s.d_misc.ptr);
    Sha1Task *d_tasks = static_cast<Sha1Task *>(s.d_tasks.ptr);
    const uint32_t n32 = static_cast<uint32_t>(n);
    const uint32_t n_simd = 4u * static_cast<uint32_t>(lease.dev->n_cu);

    std::call_once(lease.dev->cost_once, [&] { measure_chain_times(lease); });
    const Sha1CostModel model = sha1_cost_model(*lease.dev, d_arrival != nullptr);

    const bool prof = profiling_enabled();
    if (prof)
        SNAPPY_HIP_TRY(hipEventRecord(s.ev_prof[0][0], stream));
    if (group_ends == nullptr) {
        SNAPPY_HIP_TRY(launch_sha1_plan(d_offsets, d_lengths, d_epochs, n32, d_state, d_tasks, model, stream));
    } else {
        uint32_t first = 0, used = 0;
        for (uint32_t g = 0; g < n_groups; ++g) {
            if (group_ends[g] < first)
                return SNAPPY_E_INVALID_ARG;
            if (group_ends[g] > first)   // empty groups take no plan
                SNAPPY_HIP_TRY(launch_sha1_plan(d_offsets, d_lengths, d_epochs, group_ends[g] - first,
                                                d_state + used++, d_tasks, model, stream, first));
            first = group_ends[g];
        }
        n_groups = used;
    }
    // Only a batch with at least two full waves of one-lane work per SIMD can be
    // given a second wave per SIMD by the planner: do not launch what must leave.
    const int launch_rounds =
        (d_arrival == nullptr && n >= 128ull * n_simd) ? kMaxWavesPerSimd : 1;
    SNAPPY_HIP_TRY(launch_sha1_hop(d_base, d_tasks, d_state, d_digests, d_arrival, n32, n_simd,
                                   static_cast<int>(env_long("SNAPPY_HOP_GRID_ROUNDS", launch_rounds)),
                                   stream, n_groups));
    if (prof)
        SNAPPY_HIP_TRY(hipEventRecord(s.ev_prof[0][1], stream));
    return SNAPPY_OK;
}

namespace {
thread_local bool t_plan_valid = false;
thread_local uint32_t t_plan_end[kSha1Paths];
thread_local double t_plan_ms = 0.0;
}  // namespace

void finish_sha1_profile(SlotLease &lease)
{
    t_plan_valid = false;
    if (!profiling_enabled())
        return;
    // Called after the stream has been synchronised: the planner's header is final.
    Sha1PlanState head;
    const size_t head_bytes = offsetof(Sha1PlanState, reserved1);
    if (lease.slot->d_misc.ptr &&
        hipMemcpy(&head, lease.slot->d_misc.ptr, head_bytes, hipMemcpyDeviceToHost) == hipSuccess) {
        for (int k = 0; k < kSha1Paths; ++k)
            t_plan_end[k] = head.path_end[k];
        t_plan_ms = head.planned_ns * 1e-6;
        t_plan_valid = true;
    } else {
        (void)hipGetLastError();
    }
    float ms = -1.f;
    if (hipEventElapsedTime(&ms, lease.slot->ev_prof[0][0], lease.slot->ev_prof[0][1]) ==
        hipSuccess)
        profile_store(0, ms);
    else
        (void)hipGetLastError();
}

int run_sha1_tables(SlotLease &lease, const uint8_t *d_base, const uint64_t *offsets,
                    const uint64_t *lengths, uint64_t n, uint8_t *digests_host,
                    hipStream_t user_stream)
{
    Slot &s = *lease.slot;
    if (n == 0)
        return SNAPPY_OK;
    hipStream_t stream = user_stream ? user_stream : s.stream;
    const size_t table_bytes = n * sizeof(uint64_t);
    int st;
    if ((st = s.d_tables.ensure(2 * table_bytes)) != SNAPPY_OK) return st;
    if ((st = s.d_out.ensure(n * 20)) != SNAPPY_OK) return st;
    uint64_t *d_off = static_cast<uint64_t *>(s.d_tables.ptr);
    uint64_t *d_len = d_off + n;
    SNAPPY_HIP_TRY(hipMemcpyAsync(d_off, offsets, table_bytes, hipMemcpyHostToDevice, stream));
    SNAPPY_HIP_TRY(hipMemcpyAsync(d_len, lengths, table_bytes, hipMemcpyHostToDevice, stream));
    uint8_t *d_digests = static_cast<uint8_t *>(s.d_out.ptr);
    st = enqueue_sha1_resident(lease, d_base, d_off, d_len, n, d_digests, stream);
    if (st != SNAPPY_OK)
        return st;
    SNAPPY_HIP_TRY(
        hipMemcpyAsync(digests_host, d_digests, n * 20, hipMemcpyDeviceToHost, stream));
    SNAPPY_HIP_TRY(hipStreamSynchronize(stream));
    finish_sha1_profile(lease);
    return SNAPPY_OK;
}

namespace {

struct HostRange {
    const uint8_t *ptr;
    uint64_t len;
};

// Lays the ranges [first, first + n) out in the device arena in the order
// `order` (indices relative to first).  Ranges that follow each other both in
// that order and in host memory keep their relative layout and move as one
// piece; others start on 64-byte boundaries, where a block is one aligned line per
// lane (measured +4 % on the one-lane path against arbitrary offsets).  Returns the
// arena size.
uint64_t lay_out(const HostRange *ranges, uint64_t first, const std::vector<uint32_t> &order,
                 std::vector<uint64_t> &dev_off, std::vector<uint64_t> &lens,
                 std::vector<Piece> &pieces)
{
    uint64_t cursor = 0;
    for (uint32_t i : order) {
        const HostRange &r = ranges[first + i];
        lens[i] = r.len;
        if (r.len == 0) {
            dev_off[i] = cursor;
            continue;
        }
        if (!pieces.empty() && pieces.back().src + pieces.back().len == r.ptr &&
            pieces.back().dst + pieces.back().len == cursor) {
            dev_off[i] = cursor;
            pieces.back().len += r.len;
        } else {
            cursor = (cursor + 63) & ~uint64_t(63);
            dev_off[i] = cursor;
            pieces.push_back({r.ptr, cursor, r.len});
        }
        cursor = dev_off[i] + r.len;
    }
    return cursor;
}

// Copy first, then hash: small batches, or no fine-grained memory for the
// progress words.
int hash_ranges_serial(SlotLease &lease, const HostRange *ranges, uint64_t first, uint64_t n,
                       uint8_t *digests, SourceGate *gate)
{
    Slot &s = *lease.slot;
    std::vector<uint64_t> dev_off(n), lens(n);
    std::vector<uint32_t> order(n);
    for (uint64_t i = 0; i < n; ++i)
        order[i] = static_cast<uint32_t>(i);
    std::vector<Piece> pieces;
    const uint64_t bytes = lay_out(ranges, first, order, dev_off, lens, pieces);
    int st;
    if ((st = s.d_data.ensure(static_cast<size_t>(bytes) + 64)) != SNAPPY_OK)
        return st;
    uint8_t *d_base = static_cast<uint8_t *>(s.d_data.ptr);
    SNAPPY_HIP_TRY(transfer_to_device(s, lease.dev->hip_device, pieces, kTransferSliceBytes, d_base,
                                      nullptr, nullptr, gate));
    return run_sha1_tables(lease, d_base, dev_off.data(), lens.data(), n, digests + 20 * first,
                           nullptr);
}

// Hash while copying.  The kernel is launched first; the data then follows on
// the copy stream, largest files first, in slices of kSliceBytes.  After every
// slice the copy stream raises the batch's arrival counter (a 4-byte copy of
// the slice number into uncached device memory), and a wave that has claimed a
// file waits until the slice holding the file's last byte has landed.  The
// files whose chains decide the finishing time therefore start within
// milliseconds, and the rest of the transfer hides behind them.
int hash_ranges_overlapped(SlotLease &lease, const HostRange *ranges, uint64_t first, uint64_t n,
                           uint64_t slice_bytes, uint8_t *digests, SourceGate *gate)
{
    Slot &s = *lease.slot;
    std::vector<uint64_t> dev_off(n), lens(n);
    std::vector<uint32_t> order(n);
    for (uint64_t i = 0; i < n; ++i)
        order[i] = static_cast<uint32_t>(i);
    // Sorting 10^6 tiny files would cost more than their transfer order can win.
    if (n <= 65536)
        std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) {
            return ranges[first + a].len > ranges[first + b].len;
        });
    std::vector<Piece> pieces;
    const uint64_t bytes = lay_out(ranges, first, order, dev_off, lens, pieces);
    const uint64_t n_slices = (bytes + slice_bytes - 1) / slice_bytes;
    if (n_slices == 0 || n_slices >= 0x7FFFFFFFull)
        return hash_ranges_serial(lease, ranges, first, n, digests, gate);

    int st;
    const size_t table_bytes = n * sizeof(uint64_t);
    if ((st = s.d_data.ensure(static_cast<size_t>(bytes) + 64)) != SNAPPY_OK) return st;
    if ((st = s.d_tables.ensure(2 * table_bytes + n * sizeof(uint32_t))) != SNAPPY_OK) return st;
    if ((st = s.d_out.ensure(n * 20)) != SNAPPY_OK) return st;
    if ((st = s.h_stage.ensure(n_slices * sizeof(uint32_t))) != SNAPPY_OK) return st;
    // Every allocation happens before the kernel is launched: freeing or allocating
    // device or page-locked memory waits for running kernels, and this one will be
    // waiting for the transfer.
    if ((st = reserve_transfer_ring(s, slice_bytes, n_slices)) != SNAPPY_OK) return st;
    uint8_t *d_base = static_cast<uint8_t *>(s.d_data.ptr);
    uint64_t *d_off = static_cast<uint64_t *>(s.d_tables.ptr);
    uint64_t *d_len = d_off + n;
    uint32_t *d_epochs = reinterpret_cast<uint32_t *>(d_len + n);
    uint8_t *d_digests = static_cast<uint8_t *>(s.d_out.ptr);
    Sha1Arrival *d_arrival = static_cast<Sha1Arrival *>(s.d_flags);
    uint32_t *slice_numbers = static_cast<uint32_t *>(s.h_stage.ptr);
    for (uint64_t k = 0; k < n_slices; ++k)
        slice_numbers[k] = static_cast<uint32_t>(k + 1);

    // A file is complete when the slice holding its last byte has landed.
    std::vector<uint32_t> epochs(n);
    for (uint64_t i = 0; i < n; ++i)
        epochs[i] = lens[i] ? static_cast<uint32_t>((dev_off[i] + lens[i] - 1) / slice_bytes + 1) : 0u;

    Sha1Arrival zero = {};
    // ~10 s without a new slice before the kernel stops waiting (SNAPPY_ARRIVAL_MAX_POLLS: tests).
    zero.max_idle_polls = static_cast<uint32_t>(env_long("SNAPPY_ARRIVAL_MAX_POLLS", 200000));
    SNAPPY_HIP_TRY(hipMemcpy(d_arrival, &zero, sizeof(zero), hipMemcpyHostToDevice));
    SNAPPY_HIP_TRY(hipMemcpyAsync(d_off, dev_off.data(), table_bytes, hipMemcpyHostToDevice, s.stream));
    SNAPPY_HIP_TRY(hipMemcpyAsync(d_len, lens.data(), table_bytes, hipMemcpyHostToDevice, s.stream));
    SNAPPY_HIP_TRY(hipMemcpyAsync(d_epochs, epochs.data(), n * sizeof(uint32_t),
                                  hipMemcpyHostToDevice, s.stream));
    st = enqueue_sha1_resident(lease, d_base, d_off, d_len, n, d_digests, s.stream, d_epochs,
                               d_arrival);
    if (st != SNAPPY_OK)
        return st;  // nothing was launched that waits

    // From here on a kernel may be waiting: whatever goes wrong, it has to be told.
    if (const long delay_ms = env_long("SNAPPY_TEST_TRANSFER_DELAY_MS", 0))  // tests: a held-up transfer
        std::this_thread::sleep_for(std::chrono::milliseconds(delay_ms));
    hipError_t copy_error = transfer_to_device(s, lease.dev->hip_device, pieces, slice_bytes, d_base,
                                               &d_arrival->arrived, slice_numbers, gate);
    if (copy_error != hipSuccess) {
        set_error_detail(std::string("host to device transfer: ") + hipGetErrorString(copy_error));
        (void)hipGetLastError();
        const uint32_t one = 1;
        (void)hipMemcpy(&d_arrival->abort, &one, sizeof(one), hipMemcpyHostToDevice);
        (void)hipStreamSynchronize(s.stream);  // the kernel leaves on abort (or on its own time limit)
        return SNAPPY_E_HIP;
    }
    SNAPPY_HIP_TRY(hipMemcpyAsync(digests + 20 * first, d_digests, n * 20, hipMemcpyDeviceToHost,
                                  s.stream));
    SNAPPY_HIP_TRY(hipStreamSynchronize(s.stream));
    Sha1Arrival seen;
    SNAPPY_HIP_TRY(hipMemcpy(&seen, d_arrival, sizeof(seen), hipMemcpyDeviceToHost));
    if (seen.error) {
        // The kernel stopped waiting (nothing arrived for its time limit, e.g. because the
        // transfer was held up behind other work on the device), but the transfer itself
        // has completed since: the data is in the arena, so hash it there in the plain way.
        st = enqueue_sha1_resident(lease, d_base, d_off, d_len, n, d_digests, s.stream);
        if (st != SNAPPY_OK)
            return st;
        SNAPPY_HIP_TRY(hipMemcpyAsync(digests + 20 * first, d_digests, n * 20, hipMemcpyDeviceToHost,
                                      s.stream));
        SNAPPY_HIP_TRY(hipStreamSynchronize(s.stream));
    }
    finish_sha1_profile(lease);
    return SNAPPY_OK;
}

}  // namespace

namespace {
constexpr int kMaxHipDevices = 64;
std::atomic<bool> g_waiting_kernel[kMaxHipDevices];
}  // namespace

WaitingKernelTurn::WaitingKernelTurn(int hip_device)
    : flag_(&g_waiting_kernel[static_cast<unsigned>(hip_device) % kMaxHipDevices]),
      mine_(!flag_->exchange(true))
{
}

WaitingKernelTurn::~WaitingKernelTurn()
{
    if (mine_)
        flag_->store(false);
}

namespace {

// Copies the ranges [first, last) into the slot's device arena and hashes them
// on the slot's device.
int hash_ranges_on_device(int device, const HostRange *ranges, uint64_t first, uint64_t last,
                          uint8_t *digests, SourceGate *gate)
{
    SlotLease lease;
    int st = lease.acquire(device);
    if (st != SNAPPY_OK)
        return st;
    const uint64_t n = last - first;
    uint64_t total = 0;
    for (uint64_t i = 0; i < n; ++i)
        total += ranges[first + i].len;
    lease.dev->work_units += n;
    lease.dev->work_bytes += total;
    // SNAPPY_OVERLAP_MIN_BYTES / SNAPPY_OVERLAP_SLICE_BYTES: tests and tuning.
    const uint64_t min_bytes =
        static_cast<uint64_t>(env_long("SNAPPY_OVERLAP_MIN_BYTES", 64L << 20));
    // Slices end on 256-byte boundaries, so no cache line holds bytes of two slices: a
    // line is first read only after everything in it has landed.
    const uint64_t slice = std::max<uint64_t>(
        256, static_cast<uint64_t>(env_long("SNAPPY_OVERLAP_SLICE_BYTES", 32L << 20)) & ~uint64_t(255));
    if (lease.slot->d_flags == nullptr || total < min_bytes)
        return hash_ranges_serial(lease, ranges, first, n, digests, gate);
    // One waiting kernel per physical GPU at a time, for the same reason: a second
    // caller copies first and hashes afterwards.
    WaitingKernelTurn turn(lease.dev->hip_device);
    if (!turn.mine())
        return hash_ranges_serial(lease, ranges, first, n, digests, gate);
    return hash_ranges_overlapped(lease, ranges, first, n, slice, digests, gate);
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

// Longest processing time first, by padded 64-byte blocks (what a file costs the
// kernel): files at least 1/64 of a device's share are sorted by size and each goes
// to the device with the fewest blocks so far; the many small ones that remain are
// then dealt out in caller order as contiguous runs that fill each device up to the
// average, so that neighbours in host memory still travel as one transfer.  A
// device ends within one such small file (under 1/64 of the average) of the
// sorted-list bound.
void plan_shards_lpt(const uint64_t *lengths, uint64_t n, uint32_t n_shards, uint32_t *shard_of,
                     uint64_t *shard_blocks)
{
    std::vector<uint64_t> load(n_shards, 0);
    auto blocks_of = [&](uint64_t i) { return (lengths[i] + 8) / 64 + 1; };
    uint64_t total = 0;
    for (uint64_t i = 0; i < n; ++i)
        total += blocks_of(i);
    const uint64_t average = (total + n_shards - 1) / n_shards;
    const uint64_t large_from = std::max<uint64_t>(1, average / 64);
    std::vector<uint64_t> large;
    for (uint64_t i = 0; i < n; ++i)
        if (blocks_of(i) >= large_from)
            large.push_back(i);
    std::stable_sort(large.begin(), large.end(),
                     [&](uint64_t x, uint64_t y) { return blocks_of(x) > blocks_of(y); });
    for (uint64_t i : large) {
        uint32_t best = 0;
        for (uint32_t k = 1; k < n_shards; ++k)
            if (load[k] < load[best])
                best = k;
        shard_of[i] = best;
        load[best] += blocks_of(i);
    }
    uint32_t k = 0;
    for (uint64_t i = 0; i < n; ++i) {
        if (blocks_of(i) >= large_from)
            continue;
        while (k + 1 < n_shards && load[k] >= average)
            ++k;
        shard_of[i] = k;
        load[k] += blocks_of(i);
    }
    if (shard_blocks)
        for (uint32_t q = 0; q < n_shards; ++q)
            shard_blocks[q] = load[q];
}

namespace {

// Files are independent, so a batch shards across devices with no exchange: the
// files are dealt out by plan_shards_lpt, one host thread per device hashes its
// share on that device, and the digests return to the caller's order.
int hash_ranges(const std::vector<HostRange> &ranges, uint8_t *digests, SourceGate *gate = nullptr)
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
    const int shards = static_cast<int>(
        std::min<uint64_t>(std::min<uint64_t>(n, static_cast<uint64_t>(n_dev)),
                           std::max<uint64_t>(1, total / min_shard_bytes)));
    if (shards <= 1) {
        return hash_ranges_on_device(pick_device(), ranges.data(), 0, n, digests, gate);
    }

    std::vector<uint64_t> lens(n);
    for (uint64_t i = 0; i < n; ++i)
        lens[i] = ranges[i].len;
    std::vector<uint32_t> shard_of(n);
    plan_shards_lpt(lens.data(), n, static_cast<uint32_t>(shards), shard_of.data(), nullptr);
    // Each device's files, in caller order.
    std::vector<std::vector<uint32_t>> members(static_cast<size_t>(shards));
    for (uint64_t i = 0; i < n; ++i)
        members[shard_of[i]].push_back(static_cast<uint32_t>(i));

    std::vector<int> status(static_cast<size_t>(shards), SNAPPY_OK);
    std::vector<float> shard_ms(static_cast<size_t>(shards), -1.f);
    auto run_shard = [&](int d) {
        const std::vector<uint32_t> &mine = members[static_cast<size_t>(d)];
        if (mine.empty())
            return;
        status[static_cast<size_t>(d)] = abi_guard([&]() -> int {
            std::vector<HostRange> sub(mine.size());
            for (size_t k = 0; k < mine.size(); ++k)
                sub[k] = ranges[mine[k]];
            std::vector<uint8_t> local(mine.size() * 20);
            profile_reset();
            const int rc = hash_ranges_on_device(d, sub.data(), 0, sub.size(), local.data(), gate);
            if (rc != SNAPPY_OK)
                return rc;
            for (size_t k = 0; k < mine.size(); ++k)
                std::memcpy(digests + 20ull * mine[k], local.data() + 20 * k, 20);
            shard_ms[static_cast<size_t>(d)] = snappy_profile_last_ms(0);
            return SNAPPY_OK;
        });
    };
    std::vector<std::thread> workers;
    std::vector<int> inline_shards;  // no thread to be had: the calling thread takes them afterwards
    for (int d = 1; d < shards; ++d)
        if (!try_spawn(workers, [&run_shard, d] { run_shard(d); }))
            inline_shards.push_back(d);
    run_shard(0);
    for (int d : inline_shards)
        run_shard(d);
    for (auto &w : workers)
        w.join();
    // The call's kernel time is that of its slowest device.
    profile_reset();
    float slowest = -1.f;
    for (float ms : shard_ms)
        slowest = std::max(slowest, ms);
    if (slowest >= 0.f)
        profile_store(0, slowest);
    for (int v : status)
        if (v != SNAPPY_OK)
            return v;
    return SNAPPY_OK;
}

}  // namespace
}  // namespace snappy

namespace snappy {

int sha1_batch_gated(const uint8_t *base, const uint64_t *offsets, const uint64_t *lengths, uint64_t n,
                     uint8_t *digests, SourceGate *gate)
{
    std::vector<HostRange> r(n);
    for (uint64_t i = 0; i < n; ++i)
        r[i] = {base + offsets[i], lengths[i]};
    return hash_ranges(r, digests, gate);
}

}  // namespace snappy

using namespace snappy;

extern "C" int snappy_plan_shards(const uint64_t *lengths, uint64_t n, uint32_t n_shards,
                                  uint64_t *cuts)
{
    return abi_guard([&]() -> int {
        if ((n && !lengths) || !cuts || n_shards == 0)
            return SNAPPY_E_INVALID_ARG;
        plan_shards(lengths, n, n_shards, cuts);
        return SNAPPY_OK;
    });
}

extern "C" int snappy_plan_shards_lpt(const uint64_t *lengths, uint64_t n, uint32_t n_shards,
                                      uint32_t *shard_of, uint64_t *shard_blocks)
{
    return abi_guard([&]() -> int {
        if ((n && (!lengths || !shard_of)) || n_shards == 0)
            return SNAPPY_E_INVALID_ARG;
        plan_shards_lpt(lengths, n, n_shards, shard_of, shard_blocks);
        return SNAPPY_OK;
    });
}

extern "C" int snappy_sha1_cost_model(int device, double chain_ns[4], double simd_ns[4], int *measured)
{
    return abi_guard([&]() -> int {
        DeviceCtx *dev = bound_device(device);
        if (!dev || !chain_ns || !simd_ns)
            return SNAPPY_E_INVALID_ARG;
        const Sha1CostModel model = sha1_cost_model(*dev, false);
        for (int k = 0; k < kSha1Paths; ++k) {
            chain_ns[k] = model.chain_ns[k];
            simd_ns[k] = model.simd_ns[k];
        }
        if (measured)
            *measured = dev->cost_measured.load() ? 1 : 0;
        return SNAPPY_OK;
    });
}

extern "C" int snappy_device_work(int device, uint64_t *units, uint64_t *bytes)
{
    return abi_guard([&]() -> int {
        DeviceCtx *dev = bound_device(device);
        if (!dev || !units || !bytes)
            return SNAPPY_E_INVALID_ARG;
        *units = dev->work_units.load();
        *bytes = dev->work_bytes.load();
        return SNAPPY_OK;
    });
}

extern "C" int snappy_sha1(const uint8_t *data, uint64_t len, uint8_t out20[20])
{
    return abi_guard([&]() -> int {
        if (!out20 || (!data && len))
            return SNAPPY_E_INVALID_ARG;
        std::vector<HostRange> r{{data, len}};
        return hash_ranges(r, out20);
    });
}

extern "C" int snappy_digests_to_hex(const uint8_t *digests, uint64_t n, char *hex)
{
    return abi_guard([&]() -> int {
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
    });
}

extern "C" int snappy_sha1_hex(const uint8_t *data, uint64_t len, char out41[41])
{
    return abi_guard([&]() -> int {
        if (!out41)
            return SNAPPY_E_INVALID_ARG;
        uint8_t d[20];
        int st = snappy_sha1(data, len, d);
        if (st != SNAPPY_OK)
            return st;
        return snappy_digests_to_hex(d, 1, out41);
    });
}

extern "C" int snappy_sha1_batch(const uint8_t *base, const uint64_t *offsets,
                                 const uint64_t *lengths, uint64_t n, uint8_t *digests)
{
    return abi_guard([&]() -> int {
        if (n && (!offsets || !lengths || !digests))
            return SNAPPY_E_INVALID_ARG;
        std::vector<HostRange> r(n);
        for (uint64_t i = 0; i < n; ++i) {
            if (lengths[i] && !base)
                return SNAPPY_E_INVALID_ARG;
            r[i] = {base ? base + offsets[i] : nullptr, lengths[i]};
        }
        return hash_ranges(r, digests);
    });
}

extern "C" int snappy_sha1_files(const uint8_t *const *datas, const uint64_t *lengths, uint64_t n,
                                 uint8_t *digests)
{
    return abi_guard([&]() -> int {
        if (n && (!datas || !lengths || !digests))
            return SNAPPY_E_INVALID_ARG;
        std::vector<HostRange> r(n);
        for (uint64_t i = 0; i < n; ++i) {
            if (lengths[i] && !datas[i])
                return SNAPPY_E_INVALID_ARG;
            r[i] = {datas[i], lengths[i]};
        }
        return hash_ranges(r, digests);
    });
}

extern "C" int snappy_sha1_batch_device(int device, const void *d_base, const uint64_t *offsets,
                                        const uint64_t *lengths, uint64_t n, uint8_t *digests,
                                        void *stream)
{
    return abi_guard([&]() -> int {
        if (n && (!offsets || !lengths || !digests))
            return SNAPPY_E_INVALID_ARG;
        if (n >= 0xFFFFFFF0ull)
            return SNAPPY_E_INVALID_ARG;
        if (n == 0)
            return SNAPPY_OK;
        if (!d_base)
            for (uint64_t i = 0; i < n; ++i)
                if (lengths[i])
                    return SNAPPY_E_INVALID_ARG;
        SlotLease lease;
        int st = lease.acquire(device);
        if (st != SNAPPY_OK)
            return st;
        profile_reset();
        return run_sha1_tables(lease, static_cast<const uint8_t *>(d_base), offsets, lengths, n,
                               digests, static_cast<hipStream_t>(stream));
    });
}

extern "C" int snappy_sha1_batch_resident(int device, const void *d_base,
                                          const uint64_t *d_offsets, const uint64_t *d_lengths,
                                          uint64_t n, uint8_t *d_digests, void *stream)
{
    return abi_guard([&]() -> int {
        if (n && (!d_offsets || !d_lengths || !d_digests))
            return SNAPPY_E_INVALID_ARG;
        if (n == 0)
            return SNAPPY_OK;
        SlotLease lease;
        int st = lease.acquire(device);
        if (st != SNAPPY_OK)
            return st;
        profile_reset();
        hipStream_t s = stream ? static_cast<hipStream_t>(stream) : lease.slot->stream;
        st = enqueue_sha1_resident(lease, static_cast<const uint8_t *>(d_base), d_offsets, d_lengths,
                                   n, d_digests, s);
        if (st != SNAPPY_OK)
            return st;
        SNAPPY_HIP_TRY(hipStreamSynchronize(s));
        finish_sha1_profile(lease);
        return SNAPPY_OK;
    });
}

extern "C" int snappy_sha1_last_plan(uint32_t path_end[4], double *planned_ms)
{
    return abi_guard([&]() -> int {
        if (!t_plan_valid || !path_end)
            return SNAPPY_E_INVALID_ARG;
        for (int k = 0; k < kSha1Paths; ++k)
            path_end[k] = t_plan_end[k];
        if (planned_ms)
            *planned_ms = t_plan_ms;
        return SNAPPY_OK;
    });
}
