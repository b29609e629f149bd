// See transfer.hpp.
#include "transfer.hpp"

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>

namespace snappy {

namespace {

constexpr unsigned kRingSlots = 6;
constexpr unsigned kStageThreads = 6;

// Moves the pieces to the device in arena order on the slot's copy stream, cut at
// slice boundaries, and raises the arrival counter after every complete slice.
// Direct form: the sources are page-locked, every copy is a DMA from where the
// bytes lie.
hipError_t transfer_direct(Slot &s, const std::vector<Piece> &pieces, uint64_t slice_bytes,
                           uint64_t n_slices, uint8_t *d_base, uint32_t *d_progress,
                           const uint32_t *slice_numbers, SourceGate *gate,
                           const std::vector<size_t> *group_begin = nullptr, const AfterGroup *after_group = nullptr)
{
    hipError_t e = hipSuccess;
    uint64_t next_mark = 1;  // slice whose end has not been marked yet
    size_t group = 0;
    // Fires the hook of every group whose pieces all lie before `sent` (groups without pieces included).
    auto groups_done_through = [&](size_t sent) -> hipError_t {
        while (group_begin && after_group && group < group_begin->size()) {
            const size_t group_end = group + 1 < group_begin->size() ? (*group_begin)[group + 1] : pieces.size();
            if (group_end > sent)
                break;
            const hipError_t he = (*after_group)(group++);
            if (he != hipSuccess)
                return he;
        }
        return hipSuccess;
    };
    for (size_t ci = 0; ci < pieces.size(); ++ci) {
        const Piece &c = pieces[ci];
        uint64_t done = 0;
        while (done < c.len && e == hipSuccess) {
            const uint64_t pos = c.dst + done;
            const uint64_t slice_end = (pos / slice_bytes + 1) * slice_bytes;
            const uint64_t take = std::min<uint64_t>(c.len - done, slice_end - pos);
            if (gate)
                gate->wait_for(c.src + done, take);
            e = hipMemcpyAsync(d_base + pos, c.src + done, take, hipMemcpyHostToDevice,
                               s.copy_stream);
            done += take;
            // Every slice that ends at or before this position is complete.
            const uint64_t complete = (c.dst + done) / slice_bytes;
            while (e == hipSuccess && d_progress && next_mark <= complete) {
                e = hipMemcpyAsync(d_progress, &slice_numbers[next_mark - 1],
                                   sizeof(uint32_t), hipMemcpyHostToDevice, s.copy_stream);
                ++next_mark;
            }
        }
        if (e != hipSuccess)
            return e;
        if ((e = groups_done_through(ci + 1)) != hipSuccess)
            return e;
    }
    if ((e = groups_done_through(pieces.size())) != hipSuccess)   // a batch without pieces still has its groups
        return e;
    if (!d_progress)
        return hipSuccess;
    // The last, partial slice (and any gap slices before it).
    return hipMemcpyAsync(d_progress, &slice_numbers[n_slices - 1], sizeof(uint32_t),
                          hipMemcpyHostToDevice, s.copy_stream);
}

// Staged form for ordinary (pageable) sources, which no DMA engine can read:
// worker threads copy slice after slice into a ring of page-locked buffers, the
// calling thread sends each filled buffer on in slice order and raises the
// counter behind it.  Several memcpy threads are what it takes to keep up with
// the PCIe link.
hipError_t transfer_staged(Slot &s, int hip_device, const std::vector<Piece> &pieces,
                           uint64_t slice_bytes, uint64_t n_slices, uint8_t *d_base,
                           uint32_t *d_progress, const uint32_t *slice_numbers, SourceGate *gate,
                           const std::vector<size_t> *group_begin = nullptr, const AfterGroup *after_group = nullptr)
{
    struct Segment {
        const uint8_t *src;
        uint64_t dst, len;
    };
    // Slices are cut in arena coordinates, so that a slice is one contiguous range of the arena
    // and leaves the ring in one copy however many pieces it holds.  Groups (see
    // transfer_groups_to_device) lie anywhere in the arena: each is sliced on its own, from
    // the slice boundary below its first byte, and the groups' slices are numbered one after the
    // other in the order the groups were given.
    std::vector<Segment> segs;
    std::vector<uint64_t> slice_start;   // arena offset a slice's ring buffer starts at
    std::vector<size_t> seg_count;       // segments per slice
    std::vector<size_t> group_last_slice;  // per group: its last slice (SIZE_MAX: no bytes)
    {
        const size_t n_groups = group_begin ? group_begin->size() : 1;
        for (size_t g = 0; g < n_groups; ++g) {
            const size_t lo = group_begin ? (*group_begin)[g] : 0;
            const size_t hi = (group_begin && g + 1 < n_groups) ? (*group_begin)[g + 1] : pieces.size();
            const size_t first_slice = slice_start.size();
            uint64_t base = 0;
            bool have_base = false;
            for (size_t ci = lo; ci < hi; ++ci) {
                const Piece &c = pieces[ci];
                uint64_t done = 0;
                while (done < c.len) {
                    const uint64_t pos = c.dst + done;
                    if (!have_base) {
                        base = group_begin ? pos / slice_bytes * slice_bytes : 0;
                        have_base = true;
                    }
                    const uint64_t local = (pos - base) / slice_bytes;
                    while (slice_start.size() <= first_slice + local) {
                        slice_start.push_back(base + (slice_start.size() - first_slice) * slice_bytes);
                        seg_count.push_back(0);
                    }
                    const uint64_t take = std::min<uint64_t>(c.len - done, base + (local + 1) * slice_bytes - pos);
                    segs.push_back({c.src + done, pos, take});
                    ++seg_count[first_slice + local];
                    done += take;
                }
            }
            group_last_slice.push_back(slice_start.size() > first_slice ? slice_start.size() - 1 : SIZE_MAX);
        }
    }
    if (!group_begin)   // one group from arena offset 0: the caller's slice numbering (the progress counter's)
        while (slice_start.size() < n_slices) {
            slice_start.push_back(slice_start.size() * slice_bytes);
            seg_count.push_back(0);
        }
    n_slices = slice_start.size();
    if (n_slices == 0) {   // nothing to move: the groups are "sent"
        hipError_t e0 = hipSuccess;
        for (size_t g = 0; after_group && g < group_last_slice.size() && e0 == hipSuccess; ++g)
            e0 = (*after_group)(g);
        return e0;
    }
    std::vector<size_t> seg_begin(n_slices + 1, 0);
    for (uint64_t k = 0; k < n_slices; ++k)
        seg_begin[k + 1] = seg_begin[k] + seg_count[k];

    // No more ring than the batch has slices (a small batch must not pin 192 MiB).
    const unsigned ring_slots = static_cast<unsigned>(std::min<uint64_t>(kRingSlots, n_slices));
    if (reserve_transfer_ring(s, slice_bytes, n_slices) != SNAPPY_OK)
        return hipErrorOutOfMemory;
    uint8_t *ring = static_cast<uint8_t *>(s.h_ring.ptr);
    hipEvent_t sent[kRingSlots] = {};  // the first ring_slots are used
    for (unsigned r = 0; r < ring_slots; ++r)
        if (hipEventCreateWithFlags(&sent[r], hipEventDisableTiming) != hipSuccess) {
            for (unsigned q = 0; q < r; ++q)
                (void)hipEventDestroy(sent[q]);
            return hipErrorOutOfMemory;
        }

    std::mutex mu;
    std::condition_variable cv;
    std::vector<char> filled(n_slices, 0);
    uint64_t issued = 0;  // slices handed to the copy stream (their ring event is recorded)
    bool failed = false;
    std::atomic<uint64_t> next_fill{0};

    auto fill = [&] {
        (void)hipSetDevice(hip_device);
        for (;;) {
            const uint64_t k = next_fill.fetch_add(1);
            if (k >= n_slices)
                return;
            if (k >= ring_slots) {  // the ring slot's previous content must have left
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return failed || issued > k - ring_slots; });
                if (failed)
                    return;
                lk.unlock();
                if (hipEventSynchronize(sent[k % ring_slots]) != hipSuccess) {
                    std::lock_guard<std::mutex> g(mu);
                    failed = true;
                    cv.notify_all();
                    return;
                }
            }
            uint8_t *slot = ring + (k % ring_slots) * slice_bytes;
            for (size_t q = seg_begin[k]; q < seg_begin[k + 1]; ++q) {
                if (gate)
                    gate->wait_for(segs[q].src, segs[q].len);
                std::memcpy(slot + (segs[q].dst - slice_start[k]), segs[q].src, segs[q].len);
            }
            {
                std::lock_guard<std::mutex> g(mu);
                filled[k] = 1;
            }
            cv.notify_all();
        }
    };
    const unsigned n_threads = static_cast<unsigned>(std::min<uint64_t>(
        n_slices, std::min<unsigned>(kStageThreads, std::max(1u, std::thread::hardware_concurrency()))));
    std::vector<std::thread> workers;
    workers.reserve(n_threads);
    for (unsigned t = 0; t < n_threads; ++t)
        if (!try_spawn(workers, fill))
            break;
    if (workers.empty()) {
        for (unsigned r = 0; r < ring_slots; ++r)
            (void)hipEventDestroy(sent[r]);
        return hipErrorOutOfMemory;  // not one thread to stage with
    }

    hipError_t e = hipSuccess;
    size_t group = 0;
    auto groups_done_through = [&](uint64_t k) {   // groups whose last slice is k (or that hold no bytes)
        while (e == hipSuccess && after_group && group < group_last_slice.size() &&
               (group_last_slice[group] == SIZE_MAX || group_last_slice[group] <= k))
            e = (*after_group)(group++);
    };
    for (uint64_t k = 0; k < n_slices && e == hipSuccess; ++k) {
        {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return failed || filled[k]; });
            if (failed) {
                e = hipErrorUnknown;
                break;
            }
        }
        if (seg_begin[k + 1] > seg_begin[k]) {
            const uint64_t lo = segs[seg_begin[k]].dst;
            const Segment &last = segs[seg_begin[k + 1] - 1];
            const uint64_t hi = last.dst + last.len;
            e = hipMemcpyAsync(d_base + lo, ring + (k % ring_slots) * slice_bytes + (lo - slice_start[k]),
                               hi - lo, hipMemcpyHostToDevice, s.copy_stream);
        }
        if (e == hipSuccess && d_progress)
            e = hipMemcpyAsync(d_progress, &slice_numbers[k], sizeof(uint32_t),
                               hipMemcpyHostToDevice, s.copy_stream);
        if (e == hipSuccess)
            e = hipEventRecord(sent[k % ring_slots], s.copy_stream);
        groups_done_through(k);
        {
            std::lock_guard<std::mutex> g(mu);
            if (e == hipSuccess)
                issued = k + 1;
            else
                failed = true;
        }
        cv.notify_all();
    }
    {
        std::lock_guard<std::mutex> g(mu);
        if (e != hipSuccess)
            failed = true;
    }
    cv.notify_all();
    for (auto &w : workers)
        w.join();
    if (e == hipSuccess)
        e = hipStreamSynchronize(s.copy_stream);  // the ring must not be reused before it has left
    for (unsigned r = 0; r < ring_slots; ++r)
        (void)hipEventDestroy(sent[r]);
    return e;
}

// True when every piece starts in memory that HIP knows as page-locked host
// memory (snappy_host_alloc, hipHostMalloc, hipHostRegister).
bool sources_are_page_locked(const std::vector<Piece> &pieces)
{
    if (pieces.size() > 4096)
        return false;  // not worth a query per piece: staging is correct for any source
    for (const Piece &c : pieces) {
        hipPointerAttribute_t attr;
        if (hipPointerGetAttributes(&attr, c.src) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        if (attr.type != hipMemoryTypeHost)
            return false;
    }
    return true;
}

}  // namespace

hipError_t transfer_to_host(Slot &s, int hip_device, const uint8_t *d_src, uint8_t *dst,
                            uint64_t bytes)
{
    if (bytes == 0)
        return hipSuccess;
    bool locked = false;
    {
        hipPointerAttribute_t attr;
        if (hipPointerGetAttributes(&attr, dst) == hipSuccess)
            locked = attr.type == hipMemoryTypeHost;
        else
            (void)hipGetLastError();
    }
    const char *env = std::getenv("SNAPPY_STAGE_PAGEABLE");
    if (locked || bytes < (8ull << 20) || (env && env[0] == '0')) {
        hipError_t e = hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, s.copy_stream);
        return e == hipSuccess ? hipStreamSynchronize(s.copy_stream) : e;
    }
    const uint64_t want = (bytes / (2 * kStageThreads) + 255) & ~uint64_t(255);
    const uint64_t slice = std::min<uint64_t>(kTransferSliceBytes, std::max<uint64_t>(want, 1ull << 20));
    const uint64_t n_slices = (bytes + slice - 1) / slice;
    const unsigned ring_slots = static_cast<unsigned>(std::min<uint64_t>(kRingSlots, n_slices));
    if (reserve_transfer_ring(s, slice, n_slices) != SNAPPY_OK)
        return hipErrorOutOfMemory;
    uint8_t *ring = static_cast<uint8_t *>(s.h_ring.ptr);
    hipEvent_t landed[kRingSlots] = {};  // the first ring_slots are used
    for (unsigned r = 0; r < ring_slots; ++r)
        if (hipEventCreateWithFlags(&landed[r], hipEventDisableTiming) != hipSuccess) {
            for (unsigned q = 0; q < r; ++q)
                (void)hipEventDestroy(landed[q]);
            return hipErrorOutOfMemory;
        }

    std::mutex mu;
    std::condition_variable cv;
    uint64_t issued = 0;                     // slices whose copy into the ring has been enqueued
    std::vector<char> drained(n_slices, 0);  // slices copied out of the ring
    bool failed = false;
    std::atomic<uint64_t> next_drain{0};

    auto drain = [&] {
        (void)hipSetDevice(hip_device);
        for (;;) {
            const uint64_t k = next_drain.fetch_add(1);
            if (k >= n_slices)
                return;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return failed || issued > k; });
                if (failed)
                    return;
            }
            if (hipEventSynchronize(landed[k % ring_slots]) != hipSuccess) {
                std::lock_guard<std::mutex> g(mu);
                failed = true;
                cv.notify_all();
                return;
            }
            const uint64_t len = std::min<uint64_t>(slice, bytes - k * slice);
            std::memcpy(dst + k * slice, ring + (k % ring_slots) * slice, len);
            {
                std::lock_guard<std::mutex> g(mu);
                drained[k] = 1;
            }
            cv.notify_all();
        }
    };
    const unsigned n_threads = static_cast<unsigned>(std::min<uint64_t>(
        n_slices, std::min<unsigned>(kStageThreads, std::max(1u, std::thread::hardware_concurrency()))));
    std::vector<std::thread> workers;
    workers.reserve(n_threads);
    for (unsigned t = 0; t < n_threads; ++t)
        if (!try_spawn(workers, drain))
            break;
    if (workers.empty()) {
        for (unsigned r = 0; r < ring_slots; ++r)
            (void)hipEventDestroy(landed[r]);
        return hipErrorOutOfMemory;  // not one thread to drain with
    }

    hipError_t e = hipSuccess;
    for (uint64_t k = 0; k < n_slices && e == hipSuccess; ++k) {
        if (k >= ring_slots) {  // the ring slot must have been emptied
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return failed || drained[k - ring_slots]; });
            if (failed) {
                e = hipErrorUnknown;
                break;
            }
        }
        const uint64_t len = std::min<uint64_t>(slice, bytes - k * slice);
        e = hipMemcpyAsync(ring + (k % ring_slots) * slice, d_src + k * slice, len, hipMemcpyDeviceToHost,
                           s.copy_stream);
        if (e == hipSuccess)
            e = hipEventRecord(landed[k % ring_slots], s.copy_stream);
        {
            std::lock_guard<std::mutex> g(mu);
            if (e == hipSuccess)
                issued = k + 1;
            else
                failed = true;
        }
        cv.notify_all();
    }
    {
        std::lock_guard<std::mutex> g(mu);
        if (e != hipSuccess)
            failed = true;
    }
    cv.notify_all();
    for (auto &w : workers)
        w.join();
    if (e == hipSuccess)
        e = hipStreamSynchronize(s.copy_stream);
    for (unsigned r = 0; r < ring_slots; ++r)
        (void)hipEventDestroy(landed[r]);
    {
        std::lock_guard<std::mutex> g(mu);
        if (e == hipSuccess && failed)
            e = hipErrorUnknown;
    }
    return e;
}

int reserve_transfer_ring(Slot &s, uint64_t slice_bytes, uint64_t n_slices)
{
    return s.h_ring.ensure(static_cast<size_t>(std::min<uint64_t>(kRingSlots, n_slices) * slice_bytes));
}

hipError_t transfer_to_device(Slot &s, int hip_device, const std::vector<Piece> &pieces,
                              uint64_t slice_bytes, uint8_t *d_base, uint32_t *d_progress,
                              const uint32_t *slice_numbers, SourceGate *gate)
{
    uint64_t end = 0, bytes = 0;
    for (const Piece &c : pieces) {
        end = std::max(end, c.dst + c.len);
        bytes += c.len;
    }
    if (end == 0)
        return hipSuccess;
    if (d_progress == nullptr) {
        // Nobody watches the slices: size them so that all staging threads get work
        // even for a batch of a few tens of megabytes.
        const uint64_t want = (end / (2 * kStageThreads) + 255) & ~uint64_t(255);
        slice_bytes = std::min<uint64_t>(slice_bytes, std::max<uint64_t>(want, 1ull << 20));
    }
    const uint64_t n_slices = (end + slice_bytes - 1) / slice_bytes;
    const char *env = std::getenv("SNAPPY_STAGE_PAGEABLE");  // "0": never stage (tuning)
    const bool may_stage = !(env && env[0] == '0');
    // Staging pays off from a few slices on; below that the runtime's own path is as good.
    const bool stage = may_stage && (d_progress != nullptr || bytes >= (8ull << 20)) &&
                       !sources_are_page_locked(pieces);
    hipError_t e = stage ? transfer_staged(s, hip_device, pieces, slice_bytes, n_slices, d_base,
                                           d_progress, slice_numbers, gate)
                         : transfer_direct(s, pieces, slice_bytes, n_slices, d_base, d_progress,
                                           slice_numbers, gate);
    if (e == hipSuccess)
        e = hipStreamSynchronize(s.copy_stream);
    return e;
}

hipError_t transfer_groups_to_device(Slot &s, int hip_device, const std::vector<Piece> &pieces,
                                     const std::vector<size_t> &group_begin, uint64_t slice_bytes,
                                     uint8_t *d_base, const AfterGroup &after_group)
{
    uint64_t bytes = 0;
    for (const Piece &c : pieces)
        bytes += c.len;
    const char *env = std::getenv("SNAPPY_STAGE_PAGEABLE");  // "0": never stage (tuning)
    const bool stage = !(env && env[0] == '0') && bytes >= (8ull << 20) && !sources_are_page_locked(pieces);
    hipError_t e = stage ? transfer_staged(s, hip_device, pieces, slice_bytes, 0, d_base, nullptr, nullptr, nullptr,
                                           &group_begin, &after_group)
                         : transfer_direct(s, pieces, slice_bytes, 0, d_base, nullptr, nullptr, nullptr,
                                           &group_begin, &after_group);
    if (e == hipSuccess)
        e = hipStreamSynchronize(s.copy_stream);
    return e;
}

}  // namespace snappy
