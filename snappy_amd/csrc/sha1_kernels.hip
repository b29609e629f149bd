// SHA-1 batch kernel for gfx950.
//
// SHA-1 is serial inside one message (block i+1 needs the state after block i),
// so all parallelism is across messages, and the time of a batch is bounded
// below both by its largest message's chain and by its total work.  One kernel
// covers the regimes of the reference's hash loops (/root/reference/Snappy/Core/
// SnapshotManager.cs:312-324, Snappy/Features/Mcdf/McdfManager.cs:36-63,
// Snappy/Data/SnapshotMigrator.cs:62-86) by spreading a message over a group of
// G = 64, 16, 4 or 1 lanes of a wavefront:
//
//   G > 1   each lane expands and K-biases the message schedule of one block
//           (state-independent work, G blocks of a message in parallel) into
//           its own registers; the serial 80-round chain then hops from lane to
//           lane inside the group.  G = 64 gives the fastest chain for one
//           message (412 instructions per block on the chain, against ~650 when
//           one lane does everything); smaller groups trade chain speed for
//           4x / 16x more messages per wave.
//   G = 1   one message per lane, 64 per wave, 613 instructions per block and
//           lane: bulk throughput (10^5 and more small files).
#include "sha1_device.hpp"
#include "sha1_launch.hpp"
#include "wave_util.hpp"

namespace snappy {

namespace {

constexpr int kWave = 64;

__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & (kWave - 1); }

// ---------------------------------------------------------------------------
// One message per group of G lanes.
// ---------------------------------------------------------------------------

// A superblock is G consecutive 64-byte blocks of one message: lane j of the
// group expands block j's schedule (K already added) into 80 registers of its
// own.  The chain state then visits the group's lanes in order: all lanes
// execute the 80 rounds, the lane that currently holds the state does so on its
// own schedule registers, and one DPP rotate per state word hands the state to
// the next lane of the group.  The serial part of a block is therefore 400 round
// instructions + 5 adds + 5 rotates, with no memory or LDS access on the chain;
// the expansion (~320 instructions) is paid once per G blocks.
template <int G>
__device__ __forceinline__ uint32_t rotate_in_group(uint32_t v)
{
    if constexpr (G == 64)
        return __builtin_amdgcn_update_dpp(v, v, 0x13C, 0xF, 0xF, false);  // wave_ror:1
    else if constexpr (G == 16)
        return __builtin_amdgcn_update_dpp(v, v, 0x121, 0xF, 0xF, false);  // row_ror:1
    else
        return __builtin_amdgcn_update_dpp(v, v, 0x93, 0xF, 0xF, false);   // quad_perm:[3,0,1,2]
}

// Waits until the transfer of the batch has passed slice `need` (per lane; 0 =
// nothing to wait for).  Returns false when the wave should give up: the host
// raised `abort`, another wave reported an error, or nothing arrived for
// max_idle_polls polls (the host sets ~10 s).  The words are in uncached memory;
// one coalesced load per poll, ~50 us apart.

__device__ __forceinline__ bool wait_for_arrival(Sha1Arrival *arrival, uint32_t need)
{
    if (arrival == nullptr || __ballot(need != 0) == 0)
        return true;
    uint32_t last_seen = 0, idle_polls = 0;
    const uint32_t max_idle = __builtin_amdgcn_readfirstlane(
        __hip_atomic_load(&arrival->max_idle_polls, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
    for (;;) {
        const uint32_t seen = __builtin_amdgcn_readfirstlane(
            __hip_atomic_load(&arrival->arrived, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
        if (__ballot(seen < need) == 0) {
            // Order the message loads after the observation of the counter.
            __atomic_thread_fence(__ATOMIC_ACQUIRE);
            return true;
        }
        const uint32_t stop = __builtin_amdgcn_readfirstlane(
            __hip_atomic_load(&arrival->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) |
            __hip_atomic_load(&arrival->error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
        idle_polls = (seen != last_seen) ? 0u : idle_polls + 1u;
        last_seen = seen;
        if (stop != 0 || idle_polls > max_idle) {
            if (stop == 0 && seen < need)  // what this wave was waiting for, for the error report
                __hip_atomic_store(&arrival->reserved, (need << 16) | (seen & 0xFFFFu), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(&arrival->error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            return false;
        }
#pragma unroll
        for (int k = 0; k < 16; ++k)
            __builtin_amdgcn_s_sleep(127);
    }
}

enum HopResult { kHopEmpty = 0, kHopDone = 1, kHopGiveUp = 2 };

// Claims the next 64 / G messages of tasks [first, end) and hashes them, one per
// group.  kHopEmpty: the range was already exhausted.
template <int G>
__device__ __forceinline__ HopResult hop_once(const uint8_t *__restrict__ base,
                                              const Sha1Task *__restrict__ tasks, uint32_t first,
                                              uint32_t end, uint32_t *queue,
                                              uint8_t *__restrict__ digests,
                                              Sha1Arrival *arrival, uint32_t lane)
{
    static_assert(G == 64 || G == 16 || G == 4, "group sizes with a DPP rotate");
    constexpr uint32_t kGroups = kWave / G;
    const uint32_t j = lane & (G - 1);  // position in the group
    const uint32_t grp = lane / G;

    const uint32_t i0 = first + wave_claim(queue, kGroups, lane);
    if (i0 >= end)
        return kHopEmpty;
    // Groups past the end of the range idle through this round (no blocks).
    const bool have = (G == 64) || i0 + grp < end;
    const Sha1Task task = tasks[have ? i0 + grp : i0];
    if (!wait_for_arrival(arrival, have ? task.ready_epoch : 0u))
        return kHopGiveUp;
    const uint8_t *msg = base + task.offset;
    const uint64_t len = have ? task.len : 0;
    const uint64_t n_full = len >> 6;
    const uint32_t rem = static_cast<uint32_t>(len & 63u);
    const uint64_t n_blocks = have ? n_full + (rem < 56 ? 1u : 2u) : 0;
    const uint32_t shift = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(msg) & 3u);

    // The state lives in one lane of the group at a time, lane 0 first.
    uint32_t h[5];
#pragma unroll
    for (int k = 0; k < 5; ++k)
        h[k] = kSha1Init[k];

    // Raw dwords of this lane's block in the coming superblock, loaded one
    // superblock ahead so the chain does not wait for memory.
    uint32_t raw[17];
#pragma unroll
    for (int k = 0; k < 17; ++k)
        raw[k] = 0;
    if (j < n_full)
        load_block_raw(msg + (static_cast<uint64_t>(j) << 6), shift, raw);

    for (uint64_t blk0 = 0;; blk0 += G) {
        const uint64_t left = blk0 < n_blocks ? n_blocks - blk0 : 0;
        if (G == 64 ? left == 0 : __ballot(left != 0) == 0)
            break;
        const uint32_t nb = left < G ? static_cast<uint32_t>(left) : static_cast<uint32_t>(G);

        // Phase 1: lane j prepares block blk0+j (lanes past the end idle).
        uint32_t w[16];
        {
            const uint64_t b = blk0 + j;
            if (b < n_full)
                raw_block_to_be(raw, shift, w);
            else if (b == n_full)
                load_tail_block_be(msg + (n_full << 6), rem, len, w);
            else
                make_length_block(len, w);
        }
        {
            // K is added to the schedule words here, off the chain.
            uint32_t wk[80];
#pragma unroll
            for (int t = 0; t < 80; ++t) {
                const uint32_t wt = (t < 16) ? w[t] : sha1_schedule(w, t);
                wk[t] = wt + kSha1K[t / 20];
            }
            __builtin_amdgcn_sched_barrier(0);
            {
                const uint64_t nxt = blk0 + j + G;
                if (nxt < n_full)
                    load_block_raw(msg + (nxt << 6), shift, raw);
            }

            // Phase 2: the chain hops through the group's lanes.  Groups whose
            // message has fewer blocks left than the wave's longest keep their
            // state where it is.
            auto hop = [&](bool on) {
                uint32_t a = h[0], bb = h[1], c = h[2], d = h[3], e = h[4];
#pragma unroll
                for (int t = 0; t < 80; ++t)
                    sha1_round<true>(t, a, bb, c, d, e, wk[t]);
                const uint32_t moved[5] = {
                    rotate_in_group<G>(h[0] + a), rotate_in_group<G>(h[1] + bb),
                    rotate_in_group<G>(h[2] + c), rotate_in_group<G>(h[3] + d),
                    rotate_in_group<G>(h[4] + e)};
#pragma unroll
                for (int k = 0; k < 5; ++k)
                    h[k] = on ? moved[k] : h[k];
            };
            if (G == 64) {
                // One message for the whole wave: the count is the same in every lane, so the
                // loop runs on a scalar counter, two blocks per turn (the jump back costs
                // about as much as three instructions of the chain).
                const uint32_t count = __builtin_amdgcn_readfirstlane(nb);
                uint32_t s = 0;
                for (; s + 2u <= count; s += 2u) {
                    hop(true);
                    hop(true);
                }
                if (s < count)
                    hop(true);
            } else {
                for (uint32_t s = 0; __ballot(s < nb) != 0; ++s)
                    hop(s < nb);
            }
        }
    }
    // After n_blocks hops the state sits in lane (n_blocks mod G) of the group.
    if (have && j == static_cast<uint32_t>(n_blocks & (G - 1)))
        store_digest(digests + 20ull * task.out_index, h);
    return kHopDone;
}

// One message per lane: claims the next 64 messages of tasks [first, end).  There
// is nothing to share between lanes, so the schedule is expanded inside the
// rounds from a 16-word ring (613 instructions per block and lane).  The loop
// runs over full blocks only, two per iteration with the raw words of the two
// after them already on their way; the padding blocks follow once for the whole
// wave (the messages of a wave are of similar size, so little is lost to lanes
// that wait for the longest).  The loads of the loop are unconditional: a lane
// with no block left reads `idle` (68 readable bytes that nobody writes) instead,
// which keeps the raw words out of conditional code, where every iteration would
// pay a register copy per word to carry the old values past the skipped load.
// Smallest value of v over the wave's 64 lanes, in every lane.
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const uint32_t o = static_cast<uint32_t>(__shfl_xor(static_cast<int>(v), m, kWave));
        v = o < v ? o : v;
    }
    return __builtin_amdgcn_readfirstlane(v);
}

__device__ __forceinline__ HopResult lanes_once(const uint8_t *__restrict__ base,
                                                const Sha1Task *__restrict__ tasks, uint32_t first,
                                                uint32_t end, uint32_t *queue,
                                                uint8_t *__restrict__ digests, Sha1Arrival *arrival,
                                                const uint8_t *idle, uint32_t lane)
{
    const uint32_t i0 = first + wave_claim(queue, kWave, lane);
    if (i0 >= end)
        return kHopEmpty;
    const bool have = i0 + lane < end;
    const Sha1Task task = tasks[have ? i0 + lane : i0];
    if (!wait_for_arrival(arrival, have ? task.ready_epoch : 0u))
        return kHopGiveUp;
    const uint8_t *msg = base + task.offset;
    const uint64_t len = have ? task.len : 0;
    const uint64_t n_full = len >> 6;
    const uint32_t rem = static_cast<uint32_t>(len & 63u);
    const uint32_t shift = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(msg) & 3u);
    const uint8_t *rest = idle + shift;  // load_block_raw steps back by `shift` to the aligned dword

    uint32_t h[5];
#pragma unroll
    for (int k = 0; k < 5; ++k)
        h[k] = kSha1Init[k];
    uint32_t even[17], odd[17];  // raw words of the next even-numbered and odd-numbered block
    load_block_raw(n_full > 0 ? msg : rest, shift, even);
    load_block_raw(n_full > 1 ? msg + 64 : rest, shift, odd);

    uint32_t w[16];
    uint64_t blk = 0;
    // While every lane has this pair of blocks and the next, nothing in the loop
    // depends on the lane: no masks, no selects, no copies of the state.  A block
    // that does not start on a dword spills into a 17th dword, which is the first
    // dword of the block after it: it is taken from there (for the odd block, once
    // the next pair has arrived) rather than loaded, because what bounds this loop
    // next to the arithmetic is the number of load instructions, not their size.
    const uint64_t all_have = wave_min_u32(
        have ? static_cast<uint32_t>(n_full < 0xFFFFFFFFull ? n_full : 0xFFFFFFFFull) : 0u);
    if (blk + 3 < all_have) {
        const uint32_t sel = be_selector(shift);
        do {
            uint32_t w2[16];
            even[16] = odd[0];
            raw_block_to_be(even, shift, w);
#pragma unroll
            for (int t = 0; t < 15; ++t)
                w2[t] = __builtin_amdgcn_perm(odd[t + 1], odd[t], sel);
            const uint32_t last = odd[15];
            __builtin_amdgcn_sched_barrier(0);
            load_block_raw16(msg + ((blk + 2) << 6), shift, even);
            load_block_raw16(msg + ((blk + 3) << 6), shift, odd);
            sha1_compress(h, w);
            w2[15] = __builtin_amdgcn_perm(even[0], last, sel);
            sha1_compress(h, w2);
            blk += 2;
        } while (blk + 3 < all_have);
        // Back to the form the ragged loop keeps: 17 dwords per block.
        even[16] = odd[0];
        odd[16] = load_dword_after_block(blk + 1 < n_full ? msg + ((blk + 1) << 6) : rest, shift);
    }
    // The ragged rest, lane by lane.
    for (; __ballot(blk < n_full) != 0; blk += 2) {
        // Both blocks of the next pair are asked for at once: the two halves of a
        // 128-byte line then arrive together instead of a block's time apart.
        uint32_t w2[16];
        raw_block_to_be(even, shift, w);
        raw_block_to_be(odd, shift, w2);
        // The loads stay behind the byte swaps, so they can land in the registers
        // the swaps have just read instead of in a second set that is copied over.
        __builtin_amdgcn_sched_barrier(0);
        load_block_raw(blk + 2 < n_full ? msg + ((blk + 2) << 6) : rest, shift, even);
        load_block_raw(blk + 3 < n_full ? msg + ((blk + 3) << 6) : rest, shift, odd);
        if (blk < n_full)
            sha1_compress(h, w);
        if (blk + 1 < n_full)
            sha1_compress(h, w2);
    }
    const bool closed = load_tail_block_be(msg + (n_full << 6), rem, len, w);
    if (have)
        sha1_compress(h, w);
    if (have && !closed) {
        make_length_block(len, w);
        sha1_compress(h, w);
    }
    if (have)
        store_digest(digests + 20ull * task.out_index, h);
    return kHopDone;
}

// Every wave repeatedly takes the most urgent piece of work left: among the four
// paths' next unclaimed messages, the one whose chain will run longest
// (blocks x the path's time per block).  Within a path the messages are sorted
// by size, so this hands out wave-sized pieces of work longest first across all
// paths, which keeps the batch's finishing time close to max(longest chain,
// total SIMD time / SIMDs).
//
// kAlone: the launch is one wave per SIMD and a chain must have its SIMD to itself
// (a second wave on a SIMD halves the speed of the first).  The dispatcher does not
// promise that: of 1024 single-wave workgroups on 1024 SIMDs it now and then puts
// two on one SIMD and leaves another empty, and a batch bound by its largest file's
// chain then takes twice as long (seen in one launch out of 6 to 20).  So this form
// also reserves 16 accumulation registers it never touches, which brings a wave to
// 264 of a SIMD's 512 registers: two of them cannot share a SIMD, and the 248
// that remain still hold two decoder waves or a copy kernel's.
template <bool kAlone>
__global__ void __launch_bounds__(kWave)
sha1_hop_kernel(const uint8_t *__restrict__ base, const Sha1Task *__restrict__ all_tasks,
                Sha1PlanState *__restrict__ states, uint8_t *__restrict__ digests,
                Sha1Arrival *arrival, uint32_t per_simd_round, uint32_t n_groups)
{
    if constexpr (kAlone)
        asm volatile("" ::: "a15");
    const uint32_t lane = lane_id();
    if (blockIdx.x >= per_simd_round * __builtin_amdgcn_readfirstlane(states->waves_per_simd))
        return;
    // The groups of a batch that arrives piecewise, in the order of their arrival: a wave
    // that finds a group handed out moves on to the next (and waits there for its data).
    for (uint32_t g = 0; g < n_groups; ++g) {
        Sha1PlanState *state = states + g;
        const Sha1Task *tasks = all_tasks + __builtin_amdgcn_readfirstlane(state->task_base);
        uint32_t lo[kSha1Paths], hi[kSha1Paths];
        float chain[kSha1Paths];
        bool open[kSha1Paths];
#pragma unroll
        for (int k = 0; k < kSha1Paths; ++k) {
            lo[k] = k ? __builtin_amdgcn_readfirstlane(state->path_end[k - 1]) : 0u;
            hi[k] = __builtin_amdgcn_readfirstlane(state->path_end[k]);
            chain[k] = __builtin_amdgcn_readfirstlane(state->chain_ns[k]);
            open[k] = lo[k] < hi[k];
        }
        for (;;) {
            int best = -1;
            float best_time = -1.f;
#pragma unroll
            for (int k = 0; k < kSha1Paths; ++k) {
                if (!open[k])
                    continue;
                const uint32_t next = __builtin_amdgcn_readfirstlane(
                    __hip_atomic_load(&state->queue[k].next, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                if (next >= hi[k] - lo[k]) {
                    open[k] = false;
                    continue;
                }
                const uint64_t len = tasks[lo[k] + next].len;
                const float t = static_cast<float>((len + 8) / 64 + 1) * chain[k];
                if (t > best_time) {
                    best_time = t;
                    best = k;
                }
            }
            if (best < 0)
                break;
            // A failed claim only means another wave was quicker: look again.
            HopResult r;
            switch (best) {
            case 0: r = hop_once<64>(base, tasks, lo[0], hi[0], &state->queue[0].next, digests, arrival, lane); break;
            case 1: r = hop_once<16>(base, tasks, lo[1], hi[1], &state->queue[1].next, digests, arrival, lane); break;
            case 2: r = hop_once<4>(base, tasks, lo[2], hi[2], &state->queue[2].next, digests, arrival, lane); break;
            default:
                r = lanes_once(base, tasks, lo[3], hi[3], &state->queue[3].next, digests, arrival,
                               reinterpret_cast<const uint8_t *>(state->blocks), lane);
                break;
            }
            if (r == kHopGiveUp)
                return;
        }
    }
}

}  // namespace

hipError_t launch_sha1_hop(const uint8_t *d_base, const Sha1Task *d_tasks, Sha1PlanState *d_state,
                           uint8_t *d_digests, Sha1Arrival *d_arrival, uint32_t n, uint32_t n_simd,
                           int waves_per_simd, hipStream_t stream, uint32_t n_groups)
{
    if (n == 0)
        return hipSuccess;
    // `per` workgroups make one wave per SIMD; the planner's waves_per_simd decides
    // how many of the launched rounds stay.
    const uint32_t per = n < n_simd ? n : n_simd;
    if (waves_per_simd <= 1)
        hipLaunchKernelGGL(sha1_hop_kernel<true>, dim3(per), dim3(kWave), 0, stream, d_base, d_tasks, d_state,
                           d_digests, d_arrival, per, n_groups);
    else
        hipLaunchKernelGGL(sha1_hop_kernel<false>, dim3(per * static_cast<uint32_t>(waves_per_simd)),
                           dim3(kWave), 0, stream, d_base, d_tasks, d_state, d_digests, d_arrival, per, n_groups);
    return hipGetLastError();
}

}  // namespace snappy
