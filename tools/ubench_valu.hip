// Micro-benchmark: how fast does one wavefront issue the SHA-1 round's
// instruction kinds on gfx950?  Prints shader cycles per instruction for
// dependent and independent streams, full and partial EXEC masks, and one or two
// waves per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o /tmp/ubench_valu
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

#define REP8(x) x x x x x x x x
#define REP32(x) REP8(x) REP8(x) REP8(x) REP8(x)

// mode 0: dependent chain alignbit -> add (64 instr per iteration)
// mode 1: four independent chains interleaved (64 instr per iteration)
// mode 2: dependent chain alignbit -> add3 -> bitop3 (96 instr per iteration)
template <int MODE>
__global__ void bench(uint64_t *cycles, uint32_t *sink, int iters, int active_lanes)
{
    uint32_t a = threadIdx.x * 2654435761u + 1u, b = 0x9E3779B9u, c = a ^ 0x55u, d = a + 7u, e = a * 3u;
    uint64_t t0 = 0, t1 = 0;
    if ((threadIdx.x & 63) < active_lanes) {
        t0 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < iters; ++i) {
            if (MODE == 0) {
                asm volatile(REP32("v_alignbit_b32 %0, %0, %0, 27\n v_add_u32 %0, %0, %1\n")
                             : "+v"(a) : "v"(b));
            } else if (MODE == 1) {
                asm volatile(REP8("v_alignbit_b32 %0, %0, %0, 27\n v_alignbit_b32 %1, %1, %1, 27\n"
                                  "v_alignbit_b32 %2, %2, %2, 27\n v_alignbit_b32 %3, %3, %3, 27\n"
                                  "v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n"
                                  "v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n")
                             : "+v"(a), "+v"(c), "+v"(d), "+v"(e) : "v"(b));
            } else {
                asm volatile(REP32("v_alignbit_b32 %0, %0, %0, 27\n v_add3_u32 %0, %0, %1, %2\n"
                                   "v_bitop3_b32 %0, %0, %1, %2 bitop3:0xca\n")
                             : "+v"(a) : "v"(b), "v"(c));
            }
        }
        t1 = __builtin_amdgcn_s_memtime();
    }
    if ((threadIdx.x & 63) == 0)
        cycles[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
    sink[blockIdx.x * blockDim.x + threadIdx.x] = a + c + d + e;
}

template <int MODE>
static void run(const char *name, int instr_per_iter, int threads, int active, int blocks)
{
    const int iters = 20000;
    uint64_t *d_cycles;
    uint32_t *d_sink;
    const int waves = blocks * threads / 64;
    (void)hipMalloc(&d_cycles, waves * sizeof(uint64_t));
    (void)hipMalloc(&d_sink, blocks * threads * sizeof(uint32_t));
    for (int rep = 0; rep < 2; ++rep)
        hipLaunchKernelGGL(bench<MODE>, dim3(blocks), dim3(threads), 0, 0, d_cycles, d_sink, iters, active);
    (void)hipDeviceSynchronize();
    std::vector<uint64_t> h(waves);
    (void)hipMemcpy(h.data(), d_cycles, waves * sizeof(uint64_t), hipMemcpyDeviceToHost);
    double sum = 0;
    for (auto v : h) sum += double(v);
    printf("%-28s threads/block=%4d blocks=%4d active_lanes=%2d  cycles/instr/wave=%.3f\n", name, threads,
           blocks, active, sum / waves / iters / instr_per_iter);
    (void)hipFree(d_cycles);
    (void)hipFree(d_sink);
}

int main()
{
    for (int threads : {64, 256, 512, 1024}) {
        for (int active : {64, 32, 1}) {
            run<0>("dependent alignbit->add", 64, threads, active, 1);
            run<1>("4 independent chains", 64, threads, active, 1);
            run<2>("dep alignbit->add3->bitop3", 96, threads, active, 1);
        }
    }
    // every CU busy: 256 blocks of 256 and of 512 threads
    run<0>("dependent, chip full", 64, 256, 64, 256);
    run<0>("dependent, chip full", 64, 512, 64, 256);
    run<1>("independent, chip full", 64, 512, 64, 256);
    return 0;
}
