// Micro-benchmark: scalar-ALU issue rate of one wavefront on gfx950, alone and
// interleaved with vector instructions.  Prints shader cycles per instruction.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_salu.hip -o tools/ubench_salu
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

#define REP8(x) x x x x x x x x
#define REP32(x) REP8(x) REP8(x) REP8(x) REP8(x)

// mode 0: dependent SALU chain (s_add_u32, s_xor_b32 alternating), 64/iter
// mode 1: two independent SALU chains, 64/iter
// mode 2: alternating dependent VALU and dependent SALU, 32 + 32 per iter
// mode 3: 1 VALU followed by 3 SALU, 16 + 48 per iter
// mode 4: dependent VALU chain with v_readfirstlane each 2nd (VALU->SGPR hop), 64/iter
// mode 5: s_lshr_b64 rotate emulation chain, 64/iter
template <int MODE>
__global__ void bench(uint64_t *cycles, uint32_t *sink, int iters, uint32_t seed)
{
    uint32_t a = threadIdx.x * 2654435761u + 1u, vb = 0x9E3779B9u;
    uint32_t s0 = seed, s1 = seed * 3u + 1u, s2 = 0x5A827999u;
    uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {
            asm volatile(REP32("s_add_u32 %0, %0, %1\n s_xor_b32 %0, %0, %1\n") : "+s"(s0) : "s"(s2) : "scc");
        } else if (MODE == 1) {
            asm volatile(REP32("s_add_u32 %0, %0, %2\n s_add_u32 %1, %1, %2\n") : "+s"(s0), "+s"(s1) : "s"(s2) : "scc");
        } else if (MODE == 2) {
            asm volatile(REP32("v_add_u32 %0, %0, %2\n s_add_u32 %1, %1, %3\n")
                         : "+v"(a), "+s"(s0) : "v"(vb), "s"(s2) : "scc");
        } else if (MODE == 3) {
            asm volatile(REP8("v_add_u32 %0, %0, %2\n s_add_u32 %1, %1, %3\n s_xor_b32 %1, %1, %3\n s_add_u32 %1, %1, %3\n"
                              "v_add_u32 %0, %0, %2\n s_add_u32 %1, %1, %3\n s_xor_b32 %1, %1, %3\n s_add_u32 %1, %1, %3\n")
                         : "+v"(a), "+s"(s0) : "v"(vb), "s"(s2) : "scc");
        } else if (MODE == 4) {
            asm volatile(REP32("v_add_u32 %0, %0, %2\n v_readfirstlane_b32 %1, %0\n")
                         : "+v"(a), "+s"(s0) : "v"(vb));
        } else {
            uint64_t pair = (uint64_t(s0) << 32) | s0;
            asm volatile(REP32("s_lshr_b64 %0, %0, 27\n s_add_u32 %1, %1, %2\n") : "+s"(pair), "+s"(s1) : "s"(s2) : "scc");
            s0 = uint32_t(pair);
        }
    }
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0)
        cycles[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
    sink[blockIdx.x * blockDim.x + threadIdx.x] = a + s0 + s1;
}

template <int MODE>
static void run(const char *name, int instr_per_iter, int threads, int blocks)
{
    const int iters = 20000;
    uint64_t *d_cycles;
    uint32_t *d_sink;
    const int waves = blocks * threads / 64;
    (void)hipMalloc(&d_cycles, waves * sizeof(uint64_t));
    (void)hipMalloc(&d_sink, blocks * threads * sizeof(uint32_t));
    for (int rep = 0; rep < 2; ++rep)
        hipLaunchKernelGGL(bench<MODE>, dim3(blocks), dim3(threads), 0, 0, d_cycles, d_sink, iters, 12345u);
    (void)hipDeviceSynchronize();
    std::vector<uint64_t> h(waves);
    (void)hipMemcpy(h.data(), d_cycles, waves * sizeof(uint64_t), hipMemcpyDeviceToHost);
    double sum = 0;
    for (auto v : h) sum += double(v);
    printf("%-36s threads/block=%4d blocks=%3d  cycles/instr/wave=%.3f\n", name, threads, blocks,
           sum / waves / iters / instr_per_iter);
    (void)hipFree(d_cycles);
    (void)hipFree(d_sink);
}

int main()
{
    for (int threads : {64, 256, 512}) {
        run<0>("SALU dependent add/xor", 64, threads, 1);
        run<1>("SALU two chains", 64, threads, 1);
        run<2>("VALU+SALU alternating", 64, threads, 1);
        run<3>("1 VALU : 3 SALU", 64, threads, 1);
        run<4>("VALU + v_readfirstlane", 64, threads, 1);
        run<5>("s_lshr_b64 + s_add", 64, threads, 1);
    }
    run<0>("SALU dependent, chip full", 64, 256, 256);
    run<3>("1 VALU : 3 SALU, chip full", 64, 256, 256);
    return 0;
}
