// Micro-benchmark: cycles per token of the LZ4 decoder's token walk (one wave, the
// 256-byte window in registers, no memory traffic), in a few code shapes.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_lz4_walk.hip -o tools/ubench_lz4_walk
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

__device__ __forceinline__ uint32_t uni(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }

// Every token is `token` (one byte), followed by its literals and a 2-byte offset.
// MODE 0: the walk as the decoder has it (peek of 4 bytes, rare paths as branches, record per lane)
// MODE 1: one readlane, no 64-bit shift
// MODE 2: MODE 1 without the per-lane record
// MODE 3: table walk: lane l holds the next position for a token starting at byte l of a 64-byte span
template <int MODE>
__global__ void walk(uint64_t *cycles, uint32_t *sink, int iters, uint32_t token)
{
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t a = token * 0x01010101u;  // the window: every byte is the token
    uint32_t my_ip = 0, total = 0;
    // table for MODE 3
    const uint32_t lit_t = token >> 4;
    const uint32_t next_tbl = lane + 3u + lit_t;
    uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        uint32_t ip = uni(it & 3), n = 0;
        if (MODE == 3) {
            while (ip < 64u) {
                my_ip = (lane == n) ? ip : my_ip;
                ++n;
                ip = __builtin_amdgcn_readlane(next_tbl, ip);
            }
        } else {
            while (ip <= 240u) {
                uint32_t tok;
                if (MODE == 0) {
                    const uint32_t di = ip >> 2;
                    const uint32_t lo = __builtin_amdgcn_readlane(a, di);
                    const uint32_t hi = __builtin_amdgcn_readlane(a, di + 1u);
                    tok = static_cast<uint32_t>(((static_cast<uint64_t>(hi) << 32) | lo) >> ((ip & 3u) * 8u));
                } else {
                    tok = __builtin_amdgcn_readlane(a, ip >> 2) >> ((ip & 3u) * 8u);
                }
                const uint32_t lit = (tok >> 4) & 15u;
                uint32_t len = 3u + lit;
                if (__builtin_expect(lit == 15u, 0)) {
                    const uint32_t more = (tok >> 8) & 255u;
                    if (more > 225u)
                        break;
                    len += 1u + more;
                }
                if (__builtin_expect((tok & 15u) == 15u, 0)) {
                    const uint32_t more = __builtin_amdgcn_readlane(a, (ip + len) >> 2) & 255u;
                    if (more > 236u)
                        break;
                    len += 1u;
                }
                if (MODE != 2)
                    my_ip = (lane == n) ? ip : my_ip;
                ++n;
                ip += len;
                if (__builtin_expect(n >= 64u, 0))
                    break;
            }
        }
        total += n;
    }
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) {
        cycles[(blockIdx.x * blockDim.x + threadIdx.x) / 64 * 2] = t1 - t0;
        cycles[(blockIdx.x * blockDim.x + threadIdx.x) / 64 * 2 + 1] = total;
    }
    sink[blockIdx.x * blockDim.x + threadIdx.x] = my_ip + total;
}

template <int MODE>
static void run(const char *name, int threads, int blocks, uint32_t token)
{
    const int iters = 4000;
    const int waves = blocks * threads / 64;
    uint64_t *d_cycles;
    uint32_t *d_sink;
    (void)hipMalloc(&d_cycles, waves * 2 * sizeof(uint64_t));
    (void)hipMalloc(&d_sink, blocks * threads * sizeof(uint32_t));
    for (int rep = 0; rep < 2; ++rep)
        hipLaunchKernelGGL(walk<MODE>, dim3(blocks), dim3(threads), 0, 0, d_cycles, d_sink, iters, token);
    (void)hipDeviceSynchronize();
    std::vector<uint64_t> h(waves * 2);
    (void)hipMemcpy(h.data(), d_cycles, waves * 2 * sizeof(uint64_t), hipMemcpyDeviceToHost);
    double cyc = 0, tok = 0;
    for (int w = 0; w < waves; ++w) {
        cyc += double(h[2 * w]);
        tok += double(h[2 * w + 1]);
    }
    printf("%-44s token=0x%02x threads/block=%4d blocks=%3d  cycles/token/wave=%.1f\n", name, token, threads, blocks,
           cyc / tok);
    (void)hipFree(d_cycles);
    (void)hipFree(d_sink);
}

int main()
{
    for (int threads : {64, 256, 512}) {
        for (uint32_t token : {0x44u, 0xF4u}) {
            run<0>("walk as in the decoder", threads, 1, token);
            run<1>("one readlane per token", threads, 1, token);
            run<2>("one readlane, no per-lane record", threads, 1, token);
        }
        run<3>("table walk (readlane of next position)", threads, 1, 0x44u);
    }
    run<0>("walk as in the decoder, chip full", 512, 256, 0x44u);
    run<3>("table walk, chip full", 512, 256, 0x44u);
    return 0;
}
