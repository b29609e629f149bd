// Host-side launch interface of the SHA-1 kernels (sha1_kernels.hip) and of
// the device-side planner (sha1_plan.hip).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sha1_device.hpp"

namespace snappy {

constexpr int kPlanBuckets = 512;

// A batch is sorted by descending size and cut into four consecutive ranges, one
// per execution path: a message is spread over a group of 64, 16, 4 or 1 lanes
// of a wavefront (so a wave carries 1, 4, 16 or 64 messages).  The wider the
// group, the faster one message's chain and the more SIMD time a byte costs.
constexpr int kSha1Paths = 4;
constexpr int kHopGroupLanes[kSha1Paths] = {64, 16, 4, 1};

// What the planner needs to know about the paths (all times in nanoseconds per
// 64-byte block): chain_ns = how fast one message advances when its wave has a
// SIMD to itself; simd_ns = SIMD time one block costs (chain_ns / messages per
// wave).
struct Sha1CostModel {
    double chain_ns[kSha1Paths];
    double simd_ns[kSha1Paths];
    // The one-lane chain depends on how far apart in memory the 64 files of a wave lie: a batch
    // spread over `span` bytes is priced at min(chain_ns[3] + one_lane_ns_per_gb * span / 2^30,
    // one_lane_cap_ns); simd_ns[3] follows (1/64 of it).
    double one_lane_ns_per_gb;
    double one_lane_cap_ns;
    // The kernel hands work out longest first by blocks x chain time.  For that ORDER the one-lane
    // chain counts with at most this figure: priced for the worst case it would put one-lane pieces
    // ahead of wide-group pieces that really run longer, and those would then start late.
    double one_lane_order_ns;
    double load_slowdown;     // chains run up to (1 + this) times slower when every SIMD is busy
    uint32_t n_simd;
    int forced_path;          // >= 0: everything takes this path (testing / tuning)
    int forced_waves;         // > 0: waves per SIMD (testing / tuning)
};

// Planner output and the kernel's work queues, one per batch in device memory.
// Zeroed by launch_sha1_plan, filled by its kernels, read by the SHA-1 kernel.
struct Sha1PlanState {
    uint32_t path_end[kSha1Paths];    // path k owns tasks [path_end[k-1], path_end[k]); path_end[-1] = 0
    unsigned long long total_blocks;  // padded 64-byte blocks over all messages
    unsigned long long max_blocks;    // of the largest message
    double planned_ns;                // the model's finishing time for the chosen cut
    float chain_ns[kSha1Paths];       // the model's chain times, for the kernel's work order
    uint32_t waves_per_simd;          // 1: chains decide the finishing time; 2: bulk throughput
    uint32_t task_base;               // index of this plan's first task in the batch's task list
    // Address range of the batch: the largest ~offset (i.e. the smallest offset, kept this way so
    // that the zeroed header is the neutral start) and the largest end.
    unsigned long long span_lo_inverted;
    unsigned long long span_hi;
    uint32_t reserved1[11];
    struct alignas(128) Queue {       // next task of a path, counted from its range's start
        uint32_t next;
    } queue[kSha1Paths];
    uint32_t hist[kPlanBuckets];      // messages per size bucket, largest bucket first
    uint32_t cursor[kPlanBuckets];    // running write position per bucket
    unsigned long long blocks[kPlanBuckets];  // padded blocks per bucket
};

// Progress words of a batch whose data is copied to the device while the kernel
// already runs.  They live in fine-grained (uncached) device memory: the copy
// stream raises `arrived` after every slice of the transfer, the kernel's waves
// poll it before they touch a message.
struct Sha1Arrival {
    uint32_t arrived;  // slices of the transfer that have landed
    uint32_t abort;    // host: the transfer failed, give up
    uint32_t error;    // kernel: gave up (abort seen, or no progress for ~10 s)
    uint32_t reserved;  // kernel, on giving up: (slice waited for << 16) | slices seen
    uint32_t max_idle_polls;  // host: polls (~50 us each) without progress before a wave gives up
    uint32_t pad[3];
};

// Sorts the n messages by descending padded size into d_tasks and cuts the
// sorted list into the paths' ranges (see sha1_plan.hip).  d_epochs (may be
// null) gives each message's Sha1Task::ready_epoch.  `first`: the messages are
// entries [first, first + n) of the batch's tables (the pointers are the tables'
// starts); their tasks go to d_tasks[first ..] and report to digest slots first + i.
// A batch planned in several such groups takes one Sha1PlanState per group.
hipError_t launch_sha1_plan(const uint64_t *d_offsets, const uint64_t *d_lengths,
                            const uint32_t *d_epochs, uint32_t n, Sha1PlanState *d_state,
                            Sha1Task *d_tasks, const Sha1CostModel &model, hipStream_t stream,
                            uint32_t first = 0);

// Hashes the planned batch.  n_simd * kMaxWavesPerSimd single-wave workgroups are
// launched (fewer for small batches); those beyond the planner's waves_per_simd
// leave at once, the others take work, longest chains first, until none is left.
constexpr int kMaxWavesPerSimd = 2;
// d_arrival (may be null): see Sha1Arrival.  n_groups plans (d_state[0 .. n_groups)) are
// worked off one after the other by the same waves: a batch whose data arrives in
// groups is planned group by group, so that what has arrived is never queued behind
// what has not.
hipError_t launch_sha1_hop(const uint8_t *d_base, const Sha1Task *d_tasks,
                           Sha1PlanState *d_state, uint8_t *d_digests, Sha1Arrival *d_arrival,
                           uint32_t n, uint32_t n_simd, int waves_per_simd, hipStream_t stream,
                           uint32_t n_groups = 1);

}  // namespace snappy
