// Micro-benchmark: what does one LDS wave-instruction cost the CU's LDS pipeline on gfx950,
// by access shape?  The LZ4 decoder's lanes each copy their own short sequence through LDS,
// so its accesses are per-lane random and, copied naively, at any byte alignment.
//   shapes: aligned / misaligned addresses; every lane its own random address / all lanes one
//   address (the token walk's broadcast read); 1-, 2-, 4- and 8-byte reads and writes.
// Every CU runs 16 waves (as the decoder does) that issue nothing but the instruction under
// test, eight at a time between waits; reported: LDS-pipeline cycles per wave-instruction
// (kernel time x 2.4 GHz / instructions per CU) and, with one wave per CU, the round trip of
// a dependent chain (s_memtime ticks per instruction).
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_lds_access.hip -o tools/ubench_lds_access
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

enum Op { kRead32, kRead2x32, kRead8, kRead16, kWrite32, kWrite8, kWrite16 };

template <int OP>
__device__ __forceinline__ void issue(uint32_t addr, uint32_t &acc)
{
    uint32_t r0 = 0;
    if (OP == kRead32)
        asm volatile("ds_read_b32 %0, %1" : "=v"(r0) : "v"(addr));
    else if (OP == kRead2x32) {
        uint64_t r;
        asm volatile("ds_read2_b32 %0, %1 offset1:1" : "=v"(r) : "v"(addr));
        r0 = uint32_t(r) ^ uint32_t(r >> 32);
    }
    else if (OP == kRead8)
        asm volatile("ds_read_u8 %0, %1" : "=v"(r0) : "v"(addr));
    else if (OP == kRead16)
        asm volatile("ds_read_u16 %0, %1" : "=v"(r0) : "v"(addr));
    else if (OP == kWrite32)
        asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(acc));
    else if (OP == kWrite8)
        asm volatile("ds_write_b8 %0, %1" ::"v"(addr), "v"(acc));
    else if (OP == kWrite16)
        asm volatile("ds_write_b16 %0, %1" ::"v"(addr), "v"(acc));
    acc ^= r0;
}

template <int OP>
__global__ void __launch_bounds__(1024)
bench(uint32_t *sink, uint64_t *ticks, int iters, uint32_t misalign, int same_address, int dependent)
{
    __shared__ uint32_t lds[16384];  // 64 KiB
    for (uint32_t i = threadIdx.x; i < 16384; i += blockDim.x)
        lds[i] = i * 2654435761u;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    // Eight addresses per lane inside this wave's 4 KiB: random dwords, plus the misalignment.
    uint32_t addr[8];
    uint32_t x = (threadIdx.x + 1u) * 2246822519u + blockIdx.x;
    for (int k = 0; k < 8; ++k) {
        x = x * 1664525u + 1013904223u;
        const uint32_t dword = same_address ? uint32_t(k * 37 + 5) : (x >> 9) % 1000u;
        addr[k] = wave * 4096u + dword * 4u + misalign;
    }
    uint32_t acc = lane;
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    if (dependent) {
        // one chain: the next address comes from the value read (kept inside the wave's 4 KiB)
        uint32_t a = addr[0];
        for (int i = 0; i < iters * 8; ++i) {
            uint32_t r;
            asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(a));
            a = wave * 4096u + ((r >> 7) % 1000u) * 4u + misalign;
            a = same_address ? __builtin_amdgcn_readfirstlane(a) : a;
            acc ^= r;
        }
    } else {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int k = 0; k < 8; ++k)
                issue<OP>(addr[k], acc);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    sink[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (threadIdx.x == 0)
        ticks[blockIdx.x] = t1 - t0;
}

template <int OP>
static void run(const char *name, uint32_t misalign, int same, int waves_per_cu, int dependent = 0)
{
    const int iters = dependent ? 500 : 2000, grid = 256, threads = 64 * waves_per_cu;
    uint32_t *d_sink;
    uint64_t *d_ticks;
    (void)hipMalloc(&d_sink, size_t(grid) * threads * 4);
    (void)hipMalloc(&d_ticks, grid * 8);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(bench<OP>, dim3(grid), dim3(threads), 0, 0, d_sink, d_ticks, iters, misalign, same, dependent);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(bench<OP>, dim3(grid), dim3(threads), 0, 0, d_sink, d_ticks, iters, misalign, same, dependent);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<uint64_t> t(grid);
    (void)hipMemcpy(t.data(), d_ticks, grid * 8, hipMemcpyDeviceToHost);
    double sum = 0;
    for (auto v : t) sum += double(v);
    const double instr_per_cu = double(iters) * 8 * waves_per_cu;
    printf("%-12s misalign=%u %-14s waves/CU=%2d %s  %7.1f ticks per instruction per CU (wave 0: %7.1f ticks per own instruction)\n",
           name, misalign, same ? "one address" : "random address", waves_per_cu, dependent ? "dependent chain" : "8 in flight     ",
           sum / grid / instr_per_cu, sum / grid / (double(iters) * 8));
    (void)ms;
    (void)hipFree(d_sink);
    (void)hipFree(d_ticks);
}

int main()
{
    for (int waves : {1, 16}) {
        for (uint32_t mis : {0u, 1u, 2u}) {
            run<kRead32>("ds_read_b32", mis, 0, waves);
            run<kRead32>("ds_read_b32", mis, 1, waves);
            run<kWrite32>("ds_write_b32", mis, 0, waves);
        }
        run<kRead2x32>("ds_read2_b32", 0, 0, waves);
        run<kRead2x32>("ds_read2_b32", 0, 1, waves);
        run<kRead8>("ds_read_u8", 1, 0, waves);
        run<kRead8>("ds_read_u8", 1, 1, waves);
        run<kRead16>("ds_read_u16", 0, 0, waves);
        run<kRead16>("ds_read_u16", 1, 0, waves);
        run<kRead16>("ds_read_u16", 1, 1, waves);
        run<kWrite8>("ds_write_b8", 1, 0, waves);
        run<kWrite16>("ds_write_b16", 0, 0, waves);
        run<kWrite16>("ds_write_b16", 1, 0, waves);
    }
    // the token walk's shape: one wave's dependent chain, aligned and misaligned, alone on the CU and beside 15 others
    for (int waves : {1, 16}) {
        run<kRead32>("ds_read_b32", 0, 1, waves, 1);
        run<kRead32>("ds_read_b32", 1, 1, waves, 1);
        run<kRead32>("ds_read_b32", 0, 0, waves, 1);
        run<kRead32>("ds_read_b32", 1, 0, waves, 1);
    }
    return 0;
}
