// Micro-benchmark: SHA-1 compression ceiling on gfx950.
//   mode 0: compress from registers, no memory traffic (pure VALU ceiling)
//   mode 1: each lane streams its own contiguous message from HBM (64 B per block)
// Reports GB/s of message bytes, wave-instructions per second and the shader
// clock implied by s_memtime.  Build:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I snappy_amd/csrc tools/ubench_sha1.hip -o tools/ubench_sha1
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

#include "sha1_device.hpp"

using namespace snappy;

template <int MODE>
__global__ void __launch_bounds__(256)
bench(const uint8_t *base, uint64_t stride, uint32_t blocks_per_lane, uint32_t *sink, uint64_t *clk)
{
    const uint64_t gid = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    uint32_t h[5];
    for (int k = 0; k < 5; ++k)
        h[k] = kSha1Init[k] + uint32_t(gid);
    uint32_t w[16];
    for (int k = 0; k < 16; ++k)
        w[k] = uint32_t(gid) * 2654435761u + k;
    const uint8_t *p = base + gid * stride;
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (uint32_t b = 0; b < blocks_per_lane; ++b) {
        if (MODE == 1) {
            load_block_be(p, w);
            p += 64;
        }
        sha1_compress(h, w);
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    sink[gid] = h[0] ^ h[1] ^ h[2] ^ h[3] ^ h[4];
    if (gid == 0)
        *clk = t1 - t0;
}

template <int MODE>
static void run(int blocks_per_cu, uint32_t blocks_per_lane, const uint8_t *d_data, uint64_t stride)
{
    const int grid = 256 * blocks_per_cu;
    uint32_t *d_sink;
    uint64_t *d_clk;
    (void)hipMalloc(&d_sink, size_t(grid) * 256 * 4);
    (void)hipMalloc(&d_clk, 8);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(bench<MODE>, dim3(grid), dim3(256), 0, 0, d_data, stride, blocks_per_lane, d_sink, d_clk);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(bench<MODE>, dim3(grid), dim3(256), 0, 0, d_data, stride, blocks_per_lane, d_sink, d_clk);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    uint64_t clk = 0;
    (void)hipMemcpy(&clk, d_clk, 8, hipMemcpyDeviceToHost);
    const double bytes = double(grid) * 256 * blocks_per_lane * 64.0;
    const double waves = double(grid) * 4;
    printf("mode %d  waves/SIMD=%d  blocks/lane=%u  %.3f ms  %.1f GB/s  %.2f ns/block/wave  memtime ticks=%llu (%.3f GHz if ticks are cycles)\n",
           MODE, blocks_per_cu, blocks_per_lane, ms, bytes / ms / 1e6, ms * 1e6 / blocks_per_lane, (unsigned long long)clk,
           double(clk) / (ms * 1e6));
    (void)waves;
    (void)hipFree(d_sink);
    (void)hipFree(d_clk);
}

int main()
{
    const uint64_t stride = 20000;  // bytes between lanes' messages (mode 1)
    const uint32_t bpl = 256;
    uint8_t *d_data;
    const size_t bytes = size_t(256) * 8 * 256 * stride + 4096;
    (void)hipMalloc(&d_data, bytes);
    (void)hipMemset(d_data, 0x5A, bytes);
    for (int w : {1, 2, 4, 8})
        run<0>(w, 2048, d_data, stride);
    for (int w : {1, 2, 4, 8})
        run<1>(w, bpl, d_data, stride);
    return 0;
}
