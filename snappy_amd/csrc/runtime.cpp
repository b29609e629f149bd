// Device contexts, slot pool and lifecycle entry points of libsnappyhash.
#include "runtime.hpp"
#include "sha1_api.hpp"

#include <atomic>
#include <cstdlib>
#include <cstring>

namespace snappy {

namespace {

constexpr int kSlotsPerDevice = 2;

std::mutex g_init_mu;
bool g_initialized = false;
int g_init_status = SNAPPY_E_NO_DEVICE;
std::vector<std::unique_ptr<DeviceCtx>> g_devices;
std::atomic<unsigned> g_round_robin{0};
std::atomic<int> g_profile{0};

thread_local std::string t_error_detail;
thread_local float t_profile_ms[kProfileKernels] = {-1.f, -1.f};

int create_slot(Slot &s)
{
    SNAPPY_HIP_TRY(hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking));
    SNAPPY_HIP_TRY(hipStreamCreateWithFlags(&s.copy_stream, hipStreamNonBlocking));
    SNAPPY_HIP_TRY(hipStreamCreateWithFlags(&s.decode_stream, hipStreamNonBlocking));
    // Uncached on the device, so that a running kernel sees what a copy engine wrote.
    if (hipExtMallocWithFlags(&s.d_flags, 256, hipDeviceMallocFinegrained) != hipSuccess) {
        (void)hipGetLastError();
        s.d_flags = nullptr;  // callers fall back to copy-then-hash
    }
    for (int k = 0; k < kProfileKernels; ++k)
        for (int j = 0; j < 2; ++j)
            SNAPPY_HIP_TRY(hipEventCreate(&s.ev_prof[k][j]));
    s.h_stage.pinned_host = true;
    s.h_ring.pinned_host = true;
    return SNAPPY_OK;
}

void destroy_slot(Slot &s)
{
    s.d_data.release();
    s.d_tasks.release();
    s.d_tables.release();
    s.d_out.release();
    s.d_misc.release();
    s.d_chunks.release();
    s.d_queues.release();
    s.d_status.release();
    s.h_stage.release();
    s.h_ring.release();
    for (int k = 0; k < kProfileKernels; ++k)
        for (int j = 0; j < 2; ++j)
            if (s.ev_prof[k][j])
                (void)hipEventDestroy(s.ev_prof[k][j]);
    if (s.d_flags)
        (void)hipFree(s.d_flags);
    if (s.decode_stream)
        (void)hipStreamDestroy(s.decode_stream);
    if (s.copy_stream)
        (void)hipStreamDestroy(s.copy_stream);
    if (s.stream)
        (void)hipStreamDestroy(s.stream);
}

// Caller holds g_init_mu.
int init_locked(int n_devices)
{
    if (g_initialized)
        return g_init_status;
    int visible = 0;
    hipError_t e = hipGetDeviceCount(&visible);
    if (e != hipSuccess || visible <= 0) {
        set_error_detail(std::string("hipGetDeviceCount: ") +
                         (e != hipSuccess ? hipGetErrorString(e) : "no devices"));
        (void)hipGetLastError();
        g_init_status = SNAPPY_E_NO_DEVICE;
        return g_init_status;
    }
    // The kernels are built for gfx950 only: bind the gfx950 devices, in order.
    std::vector<int> usable;
    for (int d = 0; d < visible; ++d) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, d) != hipSuccess)
            continue;
        if (std::strncmp(prop.gcnArchName, "gfx950", 6) == 0)
            usable.push_back(d);
    }
    if (usable.empty()) {
        set_error_detail("no gfx950 device among the visible HIP devices");
        g_init_status = SNAPPY_E_NO_DEVICE;
        return g_init_status;
    }
    // Test hook: present one physical GPU as several logical devices so the
    // multi-device dispatcher can be exercised on a one-GPU box.
    int logical = 0;
    if (const char *env = std::getenv("SNAPPY_LOGICAL_DEVICES"))
        logical = std::atoi(env);
    if (logical > 0) {
        std::vector<int> rep;
        for (int i = 0; i < logical; ++i)
            rep.push_back(usable[static_cast<size_t>(i) % usable.size()]);
        usable.swap(rep);
    }
    if (n_devices > 0 && static_cast<size_t>(n_devices) < usable.size())
        usable.resize(static_cast<size_t>(n_devices));

    // Binding makes each device current in turn (the no-allocation entry point's planning
    // area is looked up per device): the caller's current device is put back afterwards, and
    // a failure on the way leaves nothing half bound.
    int callers_device = -1;
    if (hipGetDevice(&callers_device) != hipSuccess) {
        (void)hipGetLastError();
        callers_device = -1;
    }
    int status = SNAPPY_OK;
    for (int d : usable) {
        auto ctx = std::make_unique<DeviceCtx>();
        ctx->hip_device = d;
        hipDeviceProp_t prop;
        hipError_t e = hipGetDeviceProperties(&prop, d);
        if (e == hipSuccess)
            e = hipSetDevice(d);
        if (e != hipSuccess) {
            set_error_detail(std::string("binding device: ") + hipGetErrorString(e));
            (void)hipGetLastError();
            status = SNAPPY_E_HIP;
            break;
        }
        ctx->n_cu = prop.multiProcessorCount;
        if ((status = noalloc_setup_device(d)) != SNAPPY_OK)
            break;
        g_devices.push_back(std::move(ctx));
    }
    if (callers_device >= 0)
        (void)hipSetDevice(callers_device);
    if (status != SNAPPY_OK) {
        g_devices.clear();
        g_init_status = status;
        return status;
    }
    g_initialized = true;
    g_init_status = SNAPPY_OK;
    return g_init_status;
}

}  // namespace

void set_error_detail(const std::string &s) { t_error_detail = s; }

int Buffer::ensure(size_t bytes)
{
    if (bytes <= cap)
        return SNAPPY_OK;
    release();
    // Grow geometrically so repeated calls of rising size settle quickly.
    size_t want = bytes + bytes / 4;
    want = (want + 4095) & ~size_t(4095);
    hipError_t e = pinned_host ? hipHostMalloc(&ptr, want, hipHostMallocDefault)
                               : hipMalloc(&ptr, want);
    if (e != hipSuccess && want != bytes) {
        (void)hipGetLastError();
        want = (bytes + 4095) & ~size_t(4095);
        e = pinned_host ? hipHostMalloc(&ptr, want, hipHostMallocDefault) : hipMalloc(&ptr, want);
    }
    if (e != hipSuccess) {
        (void)hipGetLastError();
        ptr = nullptr;
        set_error_detail(std::string(pinned_host ? "hipHostMalloc: " : "hipMalloc: ") +
                         hipGetErrorString(e));
        return SNAPPY_E_NOMEM;
    }
    cap = want;
    return SNAPPY_OK;
}

void Buffer::release()
{
    if (ptr) {
        if (pinned_host)
            (void)hipHostFree(ptr);
        else
            (void)hipFree(ptr);
    }
    ptr = nullptr;
    cap = 0;
}

int ensure_initialized()
{
    std::lock_guard<std::mutex> lk(g_init_mu);
    return init_locked(0);
}

int device_count()
{
    std::lock_guard<std::mutex> lk(g_init_mu);
    return g_initialized ? static_cast<int>(g_devices.size()) : 0;
}

int pick_device()
{
    const int n = device_count();
    return n > 0 ? static_cast<int>(g_round_robin.fetch_add(1) % static_cast<unsigned>(n)) : 0;
}

DeviceCtx *bound_device(int device)
{
    std::lock_guard<std::mutex> lk(g_init_mu);
    if (!g_initialized || device < 0 || static_cast<size_t>(device) >= g_devices.size())
        return nullptr;
    return g_devices[static_cast<size_t>(device)].get();
}

bool profiling_enabled() { return g_profile.load(std::memory_order_relaxed) != 0; }

void profile_reset()
{
    for (float &v : t_profile_ms)
        v = -1.f;
}

void profile_store(int kernel, float ms)
{
    if (kernel >= 0 && kernel < kProfileKernels)
        t_profile_ms[kernel] = ms;
}

SlotLease::~SlotLease()
{
    if (slot && dev) {
        {
            std::lock_guard<std::mutex> lk(dev->mu);
            slot->busy = false;
        }
        dev->cv.notify_one();
    }
}

int SlotLease::acquire(int device)
{
    int st = ensure_initialized();
    if (st != SNAPPY_OK)
        return st;
    {
        std::lock_guard<std::mutex> lk(g_init_mu);
        if (device < 0 || static_cast<size_t>(device) >= g_devices.size())
            return SNAPPY_E_INVALID_ARG;
        dev = g_devices[static_cast<size_t>(device)].get();
    }
    SNAPPY_HIP_TRY(hipSetDevice(dev->hip_device));
    std::unique_lock<std::mutex> lk(dev->mu);
    for (;;) {
        for (auto &s : dev->slots) {
            if (!s->busy) {
                s->busy = true;
                slot = s.get();
                return SNAPPY_OK;
            }
        }
        if (dev->slots.size() < static_cast<size_t>(kSlotsPerDevice)) {
            auto s = std::make_unique<Slot>();
            st = create_slot(*s);
            if (st != SNAPPY_OK) {
                destroy_slot(*s);
                dev = nullptr;
                return st;
            }
            s->busy = true;
            slot = s.get();
            dev->slots.push_back(std::move(s));
            return SNAPPY_OK;
        }
        dev->cv.wait(lk);
    }
}

}  // namespace snappy

// ---------------------------------------------------------------------------
// C ABI: lifecycle, status text, profiling switches.
// ---------------------------------------------------------------------------
using namespace snappy;

extern "C" int snappy_init(int n_devices)
{
    return abi_guard([&]() -> int {
        std::lock_guard<std::mutex> lk(g_init_mu);
        return init_locked(n_devices);
    });
}

extern "C" void snappy_shutdown(void)
{
    std::lock_guard<std::mutex> lk(g_init_mu);
    release_file_arena();
    for (auto &d : g_devices) {
        (void)hipSetDevice(d->hip_device);
        for (auto &s : d->slots)
            destroy_slot(*s);
        d->slots.clear();
    }
    g_devices.clear();
    g_initialized = false;
    g_init_status = SNAPPY_E_NO_DEVICE;
}

extern "C" int snappy_device_count(void) { return device_count(); }

extern "C" int snappy_host_alloc(uint64_t bytes, void **out)
{
    return abi_guard([&]() -> int {
        if (!out)
            return SNAPPY_E_INVALID_ARG;
        *out = nullptr;
        int st = ensure_initialized();
        if (st != SNAPPY_OK)
            return st;
        if (bytes == 0)
            return SNAPPY_OK;
        // Portable: page-locked for every device, whichever one the data ends up on.
        hipError_t e = hipHostMalloc(out, static_cast<size_t>(bytes), hipHostMallocPortable);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            *out = nullptr;
            set_error_detail(std::string("hipHostMalloc: ") + hipGetErrorString(e));
            return SNAPPY_E_NOMEM;
        }
        return SNAPPY_OK;
    });
}

extern "C" int snappy_host_free(void *ptr)
{
    return abi_guard([&]() -> int {
        if (!ptr)
            return SNAPPY_OK;
        SNAPPY_HIP_TRY(hipHostFree(ptr));
        return SNAPPY_OK;
    });
}

extern "C" const char *snappy_strerror(int code)
{
    switch (code) {
    case SNAPPY_OK: return "ok";
    case SNAPPY_E_INVALID_ARG: return "invalid argument";
    case SNAPPY_E_NO_DEVICE: return "no usable gfx950 device";
    case SNAPPY_E_HIP: return "HIP runtime error";
    case SNAPPY_E_NOMEM: return "out of memory";
    case SNAPPY_E_LZ4_TRUNCATED: return "lz4net stream ends inside a chunk header or payload";
    case SNAPPY_E_LZ4_BAD_LENGTHS: return "lz4net chunk lengths are inconsistent";
    case SNAPPY_E_LZ4_PASSES: return "lz4net chunks with multiple passes are not supported";
    case SNAPPY_E_LZ4_CORRUPT: return "LZ4 block is corrupted, or invalid length has been given";
    case SNAPPY_E_CAPACITY: return "output buffer too small";
    case SNAPPY_E_MCDF_MAGIC: return "Not a Mare Chara File";
    case SNAPPY_E_MCDF_VERSION: return "Unsupported MCDF version";
    case SNAPPY_E_MCDF_TRUNCATED: return "MCDF payload ends inside the header";
    case SNAPPY_E_MCDF_JSON: return "MCDF header JSON is malformed";
    case SNAPPY_E_SNAPSHOT_JSON: return "an existing snapshot JSON file does not deserialize";
    case SNAPPY_E_MCDF_BASE64: return "The input is not a valid Base-64 string";
    default: return "unknown status";
    }
}

extern "C" const char *snappy_last_error_detail(void) { return t_error_detail.c_str(); }

extern "C" void snappy_profile_enable(int on) { g_profile.store(on ? 1 : 0); }

extern "C" float snappy_profile_last_ms(int kernel)
{
    return (kernel >= 0 && kernel < kProfileKernels) ? t_profile_ms[kernel] : -1.f;
}
