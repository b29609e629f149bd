// SHA-1 batch kernels for gfx950.
//
// SHA-1 is serial inside one message (block i+1 needs the state after block i),
// so all parallelism is across messages.  Two kernels cover the two regimes of
// the reference's hash loops (/root/reference/Snappy/Core/SnapshotManager.cs:
// 312-324, Snappy/Features/Mcdf/McdfManager.cs:36-63, Snappy/Data/
// SnapshotMigrator.cs:62-86):
//
//   sha1_lanes_kernel  one message per lane, 64 messages per wavefront.  Lanes
//                      that finish a message pull the next one from a device
//                      queue, so a wave stays full whatever the size mix.
//                      Throughput path (many small files).
//   sha1_wave_kernel   one message per wavefront.  The 64 lanes expand and
//                      K-bias the message schedules of 64 consecutive blocks in
//                      parallel (state-independent work) into their own
//                      registers; the serial 80-round chain then hops from lane
//                      to lane, one block per lane.  Latency path (few large
//                      files).
#include "sha1_device.hpp"
#include "sha1_launch.hpp"
#include "wave_util.hpp"

namespace snappy {

namespace {

constexpr int kWave = 64;
constexpr uint32_t kLaneBatch = 64;  // tasks a wave claims from the queue at a time

__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & (kWave - 1); }

// ---------------------------------------------------------------------------
// One message per lane.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
sha1_lanes_kernel(const uint8_t *__restrict__ base, const Sha1Task *__restrict__ tasks,
                  uint32_t end, Sha1PlanState *__restrict__ state, uint8_t *__restrict__ digests)
{
    const uint32_t lane = lane_id();
    const uint32_t first = __builtin_amdgcn_readfirstlane(state->n_wave);
    uint32_t *queue = &state->queue_lane;
    uint32_t h[5] = {0, 0, 0, 0, 0};
    const uint8_t *p = nullptr;
    uint64_t rem = 0, total = 0;
    uint32_t out_index = 0;
    // 0 = idle, 1 = message bytes (or its 0x80 block) next, 2 = length-only block next
    uint32_t mode = 0;
    // The wave claims task indices from the device queue kLaneBatch at a time and
    // hands them to its idle lanes from this pool (all three values are uniform).
    uint32_t pool_next = 0, pool_end = 0;
    bool queue_empty = false;

    for (;;) {
        for (;;) {
            const uint64_t need_mask = __ballot(mode == 0);
            if (need_mask == 0)
                break;
            if (pool_next >= pool_end) {
                if (queue_empty)
                    break;
                const uint32_t start = first + wave_claim(queue, kLaneBatch, lane);
                if (start >= end) {
                    queue_empty = true;
                    break;
                }
                pool_next = start;
                pool_end = (end - start < kLaneBatch) ? end : start + kLaneBatch;
            }
            const uint32_t rank =
                static_cast<uint32_t>(__popcll(need_mask & ((1ull << lane) - 1ull)));
            const uint32_t i = pool_next + rank;
            if (mode == 0 && i < pool_end) {
                const Sha1Task t = tasks[i];
                p = base + t.offset;
                rem = total = t.len;
                out_index = t.out_index;
#pragma unroll
                for (int k = 0; k < 5; ++k)
                    h[k] = kSha1Init[k];
                mode = 1;
            }
            const uint32_t taken = static_cast<uint32_t>(__popcll(need_mask));
            pool_next = (pool_end - pool_next < taken) ? pool_end : pool_next + taken;
        }
        if (__ballot(mode != 0) == 0)
            break;
        if (mode != 0) {
            uint32_t w[16];
            bool last = false;
            if (mode == 1) {
                if (rem >= 64) {
                    // Memory latency is covered by the other waves on the SIMD (the
                    // kernel runs at eight waves per SIMD), not by prefetching.
                    load_block_be(p, w);
                    p += 64;
                    rem -= 64;
                } else {
                    last = load_tail_block_be(p, static_cast<uint32_t>(rem), total, w);
                    if (!last)
                        mode = 2;
                }
            } else {
                make_length_block(total, w);
                last = true;
            }
            sha1_compress(h, w);
            if (last) {
                store_digest(digests + 20ull * out_index, h);
                mode = 0;
            }
        }
    }
}

// ---------------------------------------------------------------------------
// One message per wavefront.
// ---------------------------------------------------------------------------

// A superblock is 64 consecutive 64-byte blocks of one message: lane j expands
// block j's schedule (K already added) into 80 registers of its own.  The chain
// state then visits the lanes in order: all lanes execute the 80 rounds, the lane
// that currently holds the state does so on its own schedule registers, and one
// DPP rotate per state word (wave_ror:1) hands the state to the next lane.  The
// serial part of a block is therefore 400 round instructions + 5 adds + 5
// rotates, with no memory or LDS access on the chain.
constexpr int kSuper = 64;
constexpr uint32_t kDppWaveRor1 = 0x13C;

__device__ __forceinline__ uint32_t rotate_to_next_lane(uint32_t v)
{
    return __builtin_amdgcn_update_dpp(v, v, kDppWaveRor1, 0xF, 0xF, false);
}

// Schedule words K + W[0..79] of the block whose 16 big-endian words are in w.
__device__ __forceinline__ void expand_biased(uint32_t (&w)[16], uint32_t (&wk)[80])
{
#pragma unroll
    for (int t = 0; t < 80; ++t) {
        const uint32_t wt = (t < 16) ? w[t] : sha1_schedule(w, t);
        wk[t] = wt + kSha1K[t / 20];
    }
}

__global__ void __launch_bounds__(kWave)
sha1_wave_kernel(const uint8_t *__restrict__ base, const Sha1Task *__restrict__ tasks,
                 Sha1PlanState *__restrict__ state, uint8_t *__restrict__ digests)
{
    const uint32_t lane = lane_id();
    const uint32_t n_tasks = __builtin_amdgcn_readfirstlane(state->n_wave);
    uint32_t *queue = &state->queue_wave;

    for (;;) {
        const uint32_t i = wave_claim(queue, 1u, lane);
        if (i >= n_tasks)
            break;
        const Sha1Task task = tasks[i];
        const uint8_t *msg = base + task.offset;
        const uint64_t len = task.len;
        const uint64_t n_full = len >> 6;
        const uint32_t rem = static_cast<uint32_t>(len & 63u);
        const uint64_t n_blocks = n_full + (rem < 56 ? 1u : 2u);
        const uint32_t shift = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(msg) & 3u);

        // The state lives in one lane at a time, lane 0 first.
        uint32_t h[5];
#pragma unroll
        for (int k = 0; k < 5; ++k)
            h[k] = kSha1Init[k];

        // Raw dwords of this lane's block in the coming superblock, loaded one
        // superblock ahead so the chain never waits for memory.
        uint32_t raw[17];
#pragma unroll
        for (int k = 0; k < 17; ++k)
            raw[k] = 0;
        if (lane < n_full)
            load_block_raw(msg + (static_cast<uint64_t>(lane) << 6), shift, raw);

        uint32_t hops = 0;
        for (uint64_t blk0 = 0; blk0 < n_blocks; blk0 += kSuper) {
            const uint32_t nb =
                static_cast<uint32_t>(n_blocks - blk0 < kSuper ? n_blocks - blk0 : kSuper);

            // Phase 1: lane j expands block blk0+j (lanes past the end idle).
            uint32_t wk[80];
            {
                const uint64_t b = blk0 + lane;
                uint32_t w[16];
                if (b < n_full)
                    raw_block_to_be(raw, shift, w);
                else if (b == n_full)
                    load_tail_block_be(msg + (n_full << 6), rem, len, w);
                else
                    make_length_block(len, w);
                expand_biased(w, wk);
                __builtin_amdgcn_sched_barrier(0);
                const uint64_t nxt = b + kSuper;
                if (nxt < n_full)
                    load_block_raw(msg + (nxt << 6), shift, raw);
            }

            // Phase 2: the chain hops through the lanes.
            for (uint32_t b = 0; b < nb; ++b) {
                uint32_t a = h[0], bb = h[1], c = h[2], d = h[3], e = h[4];
#pragma unroll
                for (int t = 0; t < 80; ++t)
                    sha1_round<true>(t, a, bb, c, d, e, wk[t]);
                h[0] = rotate_to_next_lane(h[0] + a);
                h[1] = rotate_to_next_lane(h[1] + bb);
                h[2] = rotate_to_next_lane(h[2] + c);
                h[3] = rotate_to_next_lane(h[3] + d);
                h[4] = rotate_to_next_lane(h[4] + e);
            }
            hops += nb;
        }
        // The state ended in lane (hops mod 64); every lane stores the same bytes.
        const uint32_t holder = hops & (kWave - 1);
        uint32_t out[5];
#pragma unroll
        for (int k = 0; k < 5; ++k)
            out[k] = __builtin_amdgcn_readlane(h[k], holder);
        store_digest(digests + 20ull * task.out_index, out);
    }
}

}  // namespace

hipError_t launch_sha1_lanes(const uint8_t *d_base, const Sha1Task *d_tasks, uint32_t n,
                             Sha1PlanState *d_state, uint8_t *d_digests, int blocks,
                             hipStream_t stream)
{
    if (n == 0 || blocks <= 0)
        return hipSuccess;
    hipLaunchKernelGGL(sha1_lanes_kernel, dim3(blocks), dim3(256), 0, stream, d_base, d_tasks, n,
                       d_state, d_digests);
    return hipGetLastError();
}

hipError_t launch_sha1_wave(const uint8_t *d_base, const Sha1Task *d_tasks,
                            Sha1PlanState *d_state, uint8_t *d_digests, int waves,
                            hipStream_t stream)
{
    if (waves <= 0)
        return hipSuccess;
    hipLaunchKernelGGL(sha1_wave_kernel, dim3(waves), dim3(kWave), 0, stream, d_base, d_tasks,
                       d_state, d_digests);
    return hipGetLastError();
}

}  // namespace snappy
