// snappy_sha1_batch_resident for callers that cannot allocate: code under HIP
// graph capture, tight serving loops, harnesses that forbid hipMalloc.  Unlike the
// entry point it is a variant of, it only ENQUEUES: a memset and four kernel
// launches on the caller's stream, no allocation, no synchronisation, no wait.  The
// digests are there when the stream reaches that point; the caller synchronises (or
// captures the call into a graph and replays it).
//
// snappy_sha1_batch_resident keeps its planning area (the planner's header and
// the size-sorted task list, sha1_plan.hip) in pooled device buffers that grow
// with hipMalloc.  Here the same area is a file-scope device array of fixed
// size.  Every GPU holds its own copy of the array at its own address until the
// process ends; the address is looked up once per device when the library is
// initialised and only that address is ever handed to kernels or HIP calls (the
// array's name in host code is a stand-in, not a device address).
//
// The kernels are the ones the allocating entry point runs; the hashing is that
// of /root/reference/Snappy/Utils/Crypto.cs:8-12 over the loop of
// Snappy/Core/SnapshotManager.cs:312-324.  This entry point itself has no
// counterpart in the reference.
#include <cstdio>
#include <mutex>

#include "sha1_api.hpp"

namespace snappy {

namespace {

// 2^20 files: the planner's header (~9 KiB) plus 24 bytes per file, 24 MiB + 16 KiB
// per GPU.
constexpr uint64_t kNoAllocMaxFiles = SNAPPY_NOALLOC_MAX_FILES;
constexpr size_t kStateBytes = (sizeof(Sha1PlanState) + 255) & ~size_t(255);
constexpr size_t kScratchBytes = kStateBytes + kNoAllocMaxFiles * sizeof(Sha1Task);

static __device__ __align__(256) unsigned char scratch[kScratchBytes];

constexpr int kMaxHipDevices = 64;
// Device address of `scratch` on each HIP device (null until set up).  Calls on one
// GPU share the array, so their work must not overlap on the device: the caller
// issues them on one stream (or orders them by events).  The lock below only keeps
// two host threads from interleaving their launches; it is not held while the GPU works.
unsigned char *g_scratch_ptr[kMaxHipDevices] = {};
std::mutex g_scratch_mu[kMaxHipDevices];

}  // namespace

int noalloc_setup_device(int hip_device)
{
    if (hip_device < 0 || hip_device >= kMaxHipDevices)
        return SNAPPY_E_INVALID_ARG;
    if (g_scratch_ptr[hip_device] != nullptr)
        return SNAPPY_OK;
    void *ptr = nullptr;
    SNAPPY_HIP_TRY(hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));
    if (ptr == nullptr) {
        set_error_detail("hipGetSymbolAddress returned a null address for the planning area");
        return SNAPPY_E_HIP;
    }
    g_scratch_ptr[hip_device] = static_cast<unsigned char *>(ptr);
    return SNAPPY_OK;
}

}  // namespace snappy

using namespace snappy;

extern "C" int snappy_sha1_batch_resident_noalloc(int device, const void *d_base,
                                                  const uint64_t *d_offsets,
                                                  const uint64_t *d_lengths, uint64_t n,
                                                  uint8_t *d_digests, void *stream)
{
    return abi_guard([&]() -> int {
        if (n && (!d_offsets || !d_lengths || !d_digests))
            return SNAPPY_E_INVALID_ARG;
        if (n > kNoAllocMaxFiles) {
            char msg[160];
            std::snprintf(msg, sizeof msg,
                          "snappy_sha1_batch_resident_noalloc takes at most %llu files per call, got %llu",
                          static_cast<unsigned long long>(kNoAllocMaxFiles),
                          static_cast<unsigned long long>(n));
            set_error_detail(msg);
            return SNAPPY_E_CAPACITY;
        }
        // Initialisation creates host-side tables: it is the caller's job (snappy_init),
        // not something this entry point does on the side.
        DeviceCtx *dev = bound_device(device);
        if (dev == nullptr) {
            set_error_detail("snappy_init has not run, or the device index is out of range");
            return snappy_device_count() == 0 ? SNAPPY_E_NO_DEVICE : SNAPPY_E_INVALID_ARG;
        }
        if (n == 0)
            return SNAPPY_OK;
        const int hip_device = dev->hip_device;
        if (hip_device < 0 || hip_device >= kMaxHipDevices || g_scratch_ptr[hip_device] == nullptr) {
            set_error_detail("the planning area of this device was not set up");
            return SNAPPY_E_HIP;
        }
        // No hipSetDevice unless it changes something (the call may sit inside a capture).
        int current = -1;
        SNAPPY_HIP_TRY(hipGetDevice(&current));
        if (current != hip_device)
            SNAPPY_HIP_TRY(hipSetDevice(hip_device));

        unsigned char *area = g_scratch_ptr[hip_device];
        Sha1PlanState *d_state = reinterpret_cast<Sha1PlanState *>(area);
        Sha1Task *d_tasks = reinterpret_cast<Sha1Task *>(area + kStateBytes);
        hipStream_t s = static_cast<hipStream_t>(stream);  // null: the device's default stream
        const uint32_t n32 = static_cast<uint32_t>(n);
        const uint32_t n_simd = 4u * static_cast<uint32_t>(dev->n_cu);
        const Sha1CostModel model = sha1_cost_model(*dev, false);

        std::lock_guard<std::mutex> turn(g_scratch_mu[hip_device]);
        // The planner clears its header itself (the array still holds the last call's);
        // the task list is rewritten for entries [0, n) and read no further.
        SNAPPY_HIP_TRY(launch_sha1_plan(d_offsets, d_lengths, nullptr, n32, d_state, d_tasks, model, s));
        const int launch_rounds = (n >= 128ull * n_simd) ? kMaxWavesPerSimd : 1;
        SNAPPY_HIP_TRY(launch_sha1_hop(static_cast<const uint8_t *>(d_base), d_tasks, d_state, d_digests,
                                       nullptr, n32, n_simd, launch_rounds, s));
        return SNAPPY_OK;  // enqueued, not finished
    });
}
