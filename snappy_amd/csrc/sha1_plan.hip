// Device-side batch planning for the SHA-1 kernels.
//
// The reference hashes its files one after another in list order
// (/root/reference/Snappy/Core/SnapshotManager.cs:312-324).  On the GPU the order
// matters for the makespan: messages are handed out largest first, and the few
// that would outlast the batch on the one-message-per-lane kernel go to the
// one-message-per-wave kernel instead.  With 10^6 files that ordering costs more
// on the host than the hashing costs on the GPU, so it is done here: a bucket
// sort by padded block count (8 buckets per octave, descending) in three small
// kernels, and the split point written to a header the SHA-1 kernels read.
#include "sha1_launch.hpp"

namespace snappy {

namespace {

constexpr int kPlanThreads = 1024;
constexpr int kFilesPerThread = 4;

__device__ __forceinline__ uint64_t padded_blocks(uint64_t len) { return (len + 8) / 64 + 1; }

// Bucket of a block count: 8 per octave.  Bucket b holds counts in
// [bucket_floor(b), bucket_floor(b + 1)).
__device__ __forceinline__ uint32_t size_bucket(uint64_t blocks)
{
    const int e = 63 - __builtin_clzll(blocks | 1);
    const uint32_t mant = (e >= 3) ? static_cast<uint32_t>((blocks >> (e - 3)) & 7)
                                   : static_cast<uint32_t>((blocks << (3 - e)) & 7);
    return static_cast<uint32_t>(e) * 8u + mant;
}

__device__ __forceinline__ double bucket_floor(uint32_t b)
{
    const int e = static_cast<int>(b >> 3);
    const double m = 8.0 + static_cast<double>(b & 7);
    return (e >= 3) ? m * static_cast<double>(1ull << (e - 3)) : m / static_cast<double>(1 << (3 - e));
}

// Pass 1: histogram of buckets (reversed so that big buckets come first), total
// and largest block count.
__global__ void __launch_bounds__(kPlanThreads)
plan_histogram(const uint64_t *__restrict__ lengths, uint32_t n, Sha1PlanState *__restrict__ st)
{
    __shared__ uint32_t hist[kPlanBuckets];
    __shared__ unsigned long long s_total, s_max;
    for (int k = threadIdx.x; k < kPlanBuckets; k += kPlanThreads)
        hist[k] = 0;
    if (threadIdx.x == 0) {
        s_total = 0;
        s_max = 0;
    }
    __syncthreads();
    unsigned long long total = 0, mx = 0;
    const uint32_t base = blockIdx.x * (kPlanThreads * kFilesPerThread);
#pragma unroll
    for (int j = 0; j < kFilesPerThread; ++j) {
        const uint32_t i = base + j * kPlanThreads + threadIdx.x;
        if (i < n) {
            const uint64_t b = padded_blocks(lengths[i]);
            total += b;
            mx = b > mx ? b : mx;
            atomicAdd(&hist[kPlanBuckets - 1 - size_bucket(b)], 1u);
        }
    }
    atomicAdd(&s_total, total);
    atomicMax(&s_max, mx);
    __syncthreads();
    for (int k = threadIdx.x; k < kPlanBuckets; k += kPlanThreads)
        if (hist[k])
            atomicAdd(&st->hist[k], hist[k]);
    if (threadIdx.x == 0) {
        atomicAdd(&st->total_blocks, s_total);
        atomicMax(&st->max_blocks, s_max);
    }
}

// Pass 2 (one workgroup): bucket start positions and the split between kernels.
// A message goes to the wave kernel when the lane kernel would keep it past
// ~80 % of the batch's lower bound max(longest chain, lane-kernel issue time);
// with no more messages than SIMDs every message gets a wave of its own.
__global__ void __launch_bounds__(kPlanBuckets)
plan_scan(Sha1PlanState *__restrict__ st, uint32_t n, uint32_t n_simd, double ns_wave,
          double ns_lane, double simd_ns_lane, long long forced_min_blocks)
{
    __shared__ uint32_t scan[kPlanBuckets];
    const uint32_t k = threadIdx.x;
    const uint32_t count = st->hist[k];
    scan[k] = count;
    __syncthreads();
    for (uint32_t d = 1; d < kPlanBuckets; d <<= 1) {
        const uint32_t v = (k >= d) ? scan[k - d] : 0u;
        __syncthreads();
        scan[k] += v;
        __syncthreads();
    }
    const uint32_t start = scan[k] - count;
    st->cursor[k] = start;

    const double t_star = fmax(static_cast<double>(st->max_blocks) * ns_wave,
                               static_cast<double>(st->total_blocks) * simd_ns_lane /
                                   static_cast<double>(n_simd));
    double thr = (n <= n_simd) ? 0.0 : 0.8 * t_star / ns_lane;
    if (forced_min_blocks >= 0)
        thr = static_cast<double>(forced_min_blocks);
    // Position k holds bucket (kPlanBuckets-1-k); whole buckets above the threshold
    // go to the wave kernel.  They are a prefix of the sorted order.
    const bool above = bucket_floor(kPlanBuckets - 1 - k) > thr || thr == 0.0;
    __shared__ uint32_t n_wave;
    if (k == 0)
        n_wave = 0;
    __syncthreads();
    if (above && count)
        atomicMax(&n_wave, start + count);
    __syncthreads();
    if (k == 0) {
        const uint32_t cap = 2u * n_simd;
        st->n_wave = n_wave < cap ? n_wave : cap;
    }
}

// Pass 3: scatter the tasks to their bucket ranges.  A workgroup counts its files
// per bucket in LDS, reserves its ranges with one global atomic per bucket, then
// places its files.
__global__ void __launch_bounds__(kPlanThreads)
plan_scatter(const uint64_t *__restrict__ offsets, const uint64_t *__restrict__ lengths, uint32_t n,
             Sha1PlanState *__restrict__ st, Sha1Task *__restrict__ tasks)
{
    __shared__ uint32_t local[kPlanBuckets];
    __shared__ uint32_t base_of[kPlanBuckets];
    for (int k = threadIdx.x; k < kPlanBuckets; k += kPlanThreads)
        local[k] = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * (kPlanThreads * kFilesPerThread);
    uint32_t bucket[kFilesPerThread], slot[kFilesPerThread];
#pragma unroll
    for (int j = 0; j < kFilesPerThread; ++j) {
        const uint32_t i = base + j * kPlanThreads + threadIdx.x;
        bucket[j] = 0;
        slot[j] = 0;
        if (i < n) {
            bucket[j] = kPlanBuckets - 1 - size_bucket(padded_blocks(lengths[i]));
            slot[j] = atomicAdd(&local[bucket[j]], 1u);
        }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < kPlanBuckets; k += kPlanThreads)
        base_of[k] = local[k] ? atomicAdd(&st->cursor[k], local[k]) : 0u;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < kFilesPerThread; ++j) {
        const uint32_t i = base + j * kPlanThreads + threadIdx.x;
        if (i < n) {
            Sha1Task t;
            t.offset = offsets[i];
            t.len = lengths[i];
            t.out_index = i;
            t.reserved = 0;
            tasks[base_of[bucket[j]] + slot[j]] = t;
        }
    }
}

}  // namespace

hipError_t launch_sha1_plan(const uint64_t *d_offsets, const uint64_t *d_lengths, uint32_t n,
                            Sha1PlanState *d_state, Sha1Task *d_tasks, uint32_t n_simd,
                            double ns_wave, double ns_lane, double simd_ns_lane,
                            long long forced_min_blocks, hipStream_t stream)
{
    if (n == 0)
        return hipSuccess;
    hipError_t e = hipMemsetAsync(d_state, 0, sizeof(Sha1PlanState), stream);
    if (e != hipSuccess)
        return e;
    const uint32_t per_group = kPlanThreads * kFilesPerThread;
    const uint32_t groups = (n + per_group - 1) / per_group;
    hipLaunchKernelGGL(plan_histogram, dim3(groups), dim3(kPlanThreads), 0, stream, d_lengths, n,
                       d_state);
    hipLaunchKernelGGL(plan_scan, dim3(1), dim3(kPlanBuckets), 0, stream, d_state, n, n_simd,
                       ns_wave, ns_lane, simd_ns_lane, forced_min_blocks);
    hipLaunchKernelGGL(plan_scatter, dim3(groups), dim3(kPlanThreads), 0, stream, d_offsets,
                       d_lengths, n, d_state, d_tasks);
    return hipGetLastError();
}

}  // namespace snappy
