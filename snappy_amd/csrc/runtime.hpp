// Host runtime shared by the C-ABI entry points: device contexts, per-call
// slots (stream + pooled buffers), error plumbing.
//
// The reference's callers are a UI thread and a thread-pool task that may
// overlap (/root/reference/Snappy/Plugin.cs:257-278 runs the migrator's hash
// loop in Task.Run while Snappy/Core/SnapshotManager.cs:312-324 may run from a
// button press), so every entry point takes a private slot for the duration of
// the call and shares no mutable state outside the slot pool's lock.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <new>
#include <stdexcept>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "../../include/snappyhash.h"
#include "abi_guard.hpp"

namespace snappy {

void set_error_detail(const std::string &s);

#define SNAPPY_HIP_TRY(expr)                                                              \
    do {                                                                                  \
        hipError_t _e = (expr);                                                           \
        if (_e != hipSuccess) {                                                           \
            ::snappy::set_error_detail(std::string(#expr) + ": " + hipGetErrorString(_e)); \
            return (_e == hipErrorOutOfMemory) ? SNAPPY_E_NOMEM : SNAPPY_E_HIP;           \
        }                                                                                 \
    } while (0)

// Starts fn on a new thread of `pool`.  False when the system has no thread to
// give: the caller then does that work itself, or with the threads it has (a
// pool that went out of scope with running threads would end the process).
template <typename Fn>
bool try_spawn(std::vector<std::thread> &pool, Fn &&fn) noexcept
{
    try {
        pool.emplace_back(std::forward<Fn>(fn));
        return true;
    } catch (...) {
        return false;
    }
}

// Grow-only device or pinned-host buffer owned by a slot.
struct Buffer {
    void *ptr = nullptr;
    size_t cap = 0;
    bool pinned_host = false;
    int ensure(size_t bytes);  // SNAPPY_* status; keeps contents only when no growth is needed
    void release();
};

constexpr int kProfileKernels = 2;  // 0 = SHA-1, 1 = LZ4 decode

// Everything one in-flight call needs on one device.
struct Slot {
    hipStream_t stream = nullptr;      // main stream of the call
    hipStream_t copy_stream = nullptr; // host -> device transfers that overlap the kernels
    hipStream_t decode_stream = nullptr; // MCDF import: decode launches behind the transfers, beside the next ones
    void *d_flags = nullptr;           // 256 bytes of fine-grained device memory (progress words)
    hipEvent_t ev_prof[kProfileKernels][2] = {};
    Buffer d_data;     // device arena for host-pointer calls
    Buffer d_tasks;    // Sha1Task[] / chunk tables
    Buffer d_tables;   // (offset, length) tables of a batch
    Buffer d_out;      // digests / statuses
    Buffer d_misc;     // queue counters, error words
    // The LZ4 decoder's tables have buffers of their own: in an MCDF import the
    // decoder and the SHA-1 kernels (which use the buffers above) run side by side.
    Buffer d_chunks;   // Lz4Chunk[]
    Buffer d_queues;   // one queue counter per decode launch
    Buffer d_status;   // one status word per chunk
    Buffer h_stage;    // pinned host staging (tasks up, digests down)
    Buffer h_ring;     // pinned ring through which pageable sources travel to the device
    bool busy = false;
};

struct DeviceCtx {
    int hip_device = 0;
    int n_cu = 0;
    // Host-pointer work this device has been given since it was bound (files or
    // archives, and their bytes): what snappy_device_work reports.
    std::atomic<uint64_t> work_units{0}, work_bytes{0};
    // How this device's SHA-1 chain compares with the planner's table, measured the
    // first time the device hashes (sha1_api.cpp); unused until cost_measured is set.
    std::once_flag cost_once;
    std::atomic<bool> cost_measured{false};
    double chain_level = 1.0;
    std::mutex mu;
    std::condition_variable cv;
    std::vector<std::unique_ptr<Slot>> slots;
};

// RAII lease of a slot; sets the HIP device for the calling thread.
class SlotLease {
public:
    SlotLease() = default;
    ~SlotLease();
    SlotLease(const SlotLease &) = delete;
    SlotLease &operator=(const SlotLease &) = delete;
    int acquire(int device);  // SNAPPY_* status
    Slot *slot = nullptr;
    DeviceCtx *dev = nullptr;
};

int ensure_initialized();          // SNAPPY_OK or SNAPPY_E_NO_DEVICE / SNAPPY_E_HIP
int device_count();                // after ensure_initialized
int pick_device();                 // round-robin choice for single-buffer calls
// The bound device `device`, or null when the library is not initialised or the
// index is out of range.  Initialises nothing and allocates nothing.
DeviceCtx *bound_device(int device);
void release_file_arena();         // host_mirror.cpp: the page-locked arena file contents are read into
bool profiling_enabled();
void profile_reset();
void profile_store(int kernel, float ms);

}  // namespace snappy
