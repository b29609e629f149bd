// snappy_sha1_batch_resident_noalloc inside a HIP stream capture: captured, instantiated and
// replayed twice, the second time over different bytes in the same buffers.  Plain HIP, no
// framework around it: every HIP call's status is printed as it happens and nothing is replayed
// unless every step before it succeeded.  The digests of both replays are written to the file
// named by argv[1] (2 x n x 20 bytes) for tests/test_zz_no_alloc.py to compare with the CPU
// oracle; the program itself compares them with the allocating entry point's.
// Built by __graft_entry__.build():
//   hipcc --offload-arch=gfx950 -O2 tests/c_driver/capture_probe.cpp -Iinclude -Lsnappy_amd -lsnappyhash \
//         -Wl,-rpath,'$ORIGIN/../../snappy_amd' -o tests/c_driver/capture_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "snappyhash.h"

#define STEP(expr)                                                                       \
    do {                                                                                 \
        hipError_t e_ = (expr);                                                          \
        printf("%-70s -> %s\n", #expr, hipGetErrorString(e_));                           \
        fflush(stdout);                                                                  \
        if (e_ != hipSuccess)                                                            \
            return 2;                                                                    \
    } while (0)

int main(int argc, char **argv)
{
    printf("snappy_init -> %d\n", snappy_init(0));
    // The same figures in tests/test_zz_no_alloc.py.
    const uint64_t n = 300, bytes = 4u << 20;
    std::vector<uint8_t> host(bytes);
    for (uint64_t i = 0; i < bytes; ++i)
        host[i] = uint8_t((i * 2654435761ull) >> 13);
    std::vector<uint64_t> off(n), len(n);
    for (uint64_t i = 0; i < n; ++i) {
        len[i] = (i * 7919u) % 30000u;
        off[i] = (i * 104729u) % (bytes - 1600000u);
    }
    len[0] = 1500001;
    uint8_t *d_data, *d_dig;
    uint64_t *d_off, *d_len;
    STEP(hipMalloc(&d_data, bytes));
    STEP(hipMalloc(&d_off, n * 8));
    STEP(hipMalloc(&d_len, n * 8));
    STEP(hipMalloc(&d_dig, n * 20));
    STEP(hipMemcpy(d_off, off.data(), n * 8, hipMemcpyHostToDevice));
    STEP(hipMemcpy(d_len, len.data(), n * 8, hipMemcpyHostToDevice));
    hipStream_t s;
    STEP(hipStreamCreate(&s));

    hipGraph_t graph;
    hipGraphExec_t exec;
    STEP(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    int st = snappy_sha1_batch_resident_noalloc(0, d_data, d_off, d_len, n, d_dig, s);
    printf("noalloc (under capture) -> %d %s\n", st, st ? snappy_last_error_detail() : "");
    fflush(stdout);
    hipError_t end = hipStreamEndCapture(s, &graph);
    printf("hipStreamEndCapture -> %s\n", hipGetErrorString(end));
    fflush(stdout);
    if (st != 0 || end != hipSuccess)
        return 3;
    size_t nodes = 0;
    STEP(hipGraphGetNodes(graph, nullptr, &nodes));
    printf("graph nodes: %zu\n", nodes);
    STEP(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));

    std::vector<uint8_t> want(n * 20), got(2 * n * 20);
    int bad = 0;
    for (int replay = 0; replay < 2; ++replay) {
        if (replay)
            for (uint64_t i = 0; i < bytes; ++i)
                host[i] ^= 0x5A;
        STEP(hipMemcpy(d_data, host.data(), bytes, hipMemcpyHostToDevice));
        st = snappy_sha1_batch_resident(0, d_data, d_off, d_len, n, d_dig, nullptr);
        printf("snappy_sha1_batch_resident -> %d\n", st);
        if (st != 0)
            return 4;
        STEP(hipMemcpy(want.data(), d_dig, n * 20, hipMemcpyDeviceToHost));
        STEP(hipMemsetAsync(d_dig, 0xEE, n * 20, s));
        STEP(hipGraphLaunch(exec, s));
        STEP(hipStreamSynchronize(s));
        STEP(hipMemcpy(got.data() + replay * n * 20, d_dig, n * 20, hipMemcpyDeviceToHost));
        const bool same = memcmp(got.data() + replay * n * 20, want.data(), n * 20) == 0;
        printf("replay %d digests %s\n", replay, same ? "equal" : "DIFFER");
        bad += same ? 0 : 1;
    }
    if (argc > 1) {
        FILE *f = fopen(argv[1], "wb");
        if (!f || fwrite(got.data(), 1, got.size(), f) != got.size())
            return 5;
        fclose(f);
    }
    STEP(hipGraphExecDestroy(exec));
    STEP(hipGraphDestroy(graph));
    return bad ? 6 : 0;
}
