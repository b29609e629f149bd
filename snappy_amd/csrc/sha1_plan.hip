// Device-side batch planning for the SHA-1 kernels.
//
// The reference hashes its files one after another in list order
// (/root/reference/Snappy/Core/SnapshotManager.cs:312-324).  On the GPU the order
// matters for the makespan: messages are handed out largest first, and the larger
// a message, the more lanes of a wavefront work on it (see sha1_kernels.hip).
// With 10^6 files that ordering costs more on the host than the hashing costs on
// the GPU, so it is done here: a bucket sort by padded block count (8 buckets
// per octave, descending) in three small kernels, and the cuts between the
// execution paths written to a header the SHA-1 kernels read.
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
plan_histogram(const uint64_t *__restrict__ offsets, const uint64_t *__restrict__ lengths, uint32_t n,
               Sha1PlanState *__restrict__ st)
{
    __shared__ uint32_t hist[kPlanBuckets];
    __shared__ unsigned long long sums[kPlanBuckets];
    __shared__ unsigned long long s_total, s_max, s_lo_inv, s_hi;
    for (int k = threadIdx.x; k < kPlanBuckets; k += kPlanThreads) {
        hist[k] = 0;
        sums[k] = 0;
    }
    if (threadIdx.x == 0) {
        s_total = 0;
        s_max = 0;
        s_lo_inv = 0;
        s_hi = 0;
    }
    __syncthreads();
    unsigned long long total = 0, mx = 0, lo_inv = 0, hi = 0;
    const uint32_t base = blockIdx.x * (kPlanThreads * kFilesPerThread);
#pragma unroll
    for (int j = 0; j < kFilesPerThread; ++j) {
        const uint32_t i = base + j * kPlanThreads + threadIdx.x;
        if (i < n) {
            const uint64_t len = lengths[i];
            const uint64_t b = padded_blocks(len);
            total += b;
            mx = b > mx ? b : mx;
            if (len) {
                const uint64_t off = offsets[i];
                lo_inv = ~off > lo_inv ? ~off : lo_inv;
                hi = off + len > hi ? off + len : hi;
            }
            const uint32_t pos = kPlanBuckets - 1 - size_bucket(b);
            atomicAdd(&hist[pos], 1u);
            atomicAdd(&sums[pos], static_cast<unsigned long long>(b));
        }
    }
    atomicAdd(&s_total, total);
    atomicMax(&s_max, mx);
    atomicMax(&s_lo_inv, lo_inv);
    atomicMax(&s_hi, hi);
    __syncthreads();
    for (int k = threadIdx.x; k < kPlanBuckets; k += kPlanThreads)
        if (hist[k]) {
            atomicAdd(&st->hist[k], hist[k]);
            atomicAdd(&st->blocks[k], sums[k]);
        }
    if (threadIdx.x == 0) {
        atomicAdd(&st->total_blocks, s_total);
        atomicMax(&st->max_blocks, s_max);
        atomicMax(&st->span_lo_inverted, s_lo_inv);
        atomicMax(&st->span_hi, s_hi);
    }
}

// Pass 2 (one workgroup): bucket start positions and the cuts between paths.
//
// The kernel hands out wave-sized pieces of work longest first (a piece = the
// 64 / G files one wave takes together; its duration = its largest file's
// blocks x chain_ns of the path).  For m SIMDs such a schedule finishes no later
// than max(d_1, W + d_(m+1)): the longest piece, or the average load W (total
// SIMD time / m) plus the longest piece that does not start at once.  The cut is
// a one-parameter family: with D the longest duration allowed on paths 1..3, a
// size bucket takes the cheapest of those paths on which it stays within D, and
// path 0 (the fastest chain, the most SIMD time) otherwise.  A large D means
// little work but long late pieces, a small D the opposite; the D with the
// smallest bound is searched on a geometric grid.
constexpr int kCoarseCandidates = 22;  // 11 octaves
constexpr int kFineCandidates = 14;    // one octave around the coarse optimum
constexpr int kOwnCandidates = kSha1Paths;  // the largest file's own chain times
constexpr int kPlanCandidates = kCoarseCandidates + kOwnCandidates + kFineCandidates;

__device__ __forceinline__ int path_for(double edge_blocks, double d_max, const Sha1CostModel &m)
{
    for (int k = kSha1Paths - 1; k > 0; --k)
        if (edge_blocks * m.chain_ns[k] <= d_max)
            return k;
    return 0;
}

constexpr int kPlanWaves = kPlanBuckets / 64;

__device__ __forceinline__ double candidate_d(int c, double d_top, double max_blocks, double best_d,
                                              const Sha1CostModel &m)
{
    if (c < kCoarseCandidates)
        return d_top * exp2(-0.5 * static_cast<double>(c));
    if (c == kCoarseCandidates)
        return max_blocks * m.chain_ns[0] * 1.0001;
    if (c < kCoarseCandidates + kOwnCandidates)
        return max_blocks * m.chain_ns[c - kCoarseCandidates] * 0.9999;
    return best_d * exp2((static_cast<double>(c - kCoarseCandidates - kOwnCandidates) -
                          0.5 * (kFineCandidates - 1)) / kFineCandidates);
}

__global__ void __launch_bounds__(kPlanBuckets)
plan_scan(Sha1PlanState *__restrict__ st, uint32_t n, Sha1CostModel m, uint32_t first)
{
    __shared__ uint32_t scan[kPlanBuckets];
    __shared__ uint32_t used[kPlanBuckets];   // positions of the non-empty buckets
    __shared__ double u_edge[kPlanBuckets];   // per used bucket: largest file possible,
    __shared__ double u_blocks[kPlanBuckets]; //   blocks in it,
    __shared__ uint32_t u_files[kPlanBuckets];  // files in it
    // Per wave, for the candidate the wave is working on: duration of a piece and
    // number of pieces of every used bucket.
    __shared__ float dur[kPlanWaves][kPlanBuckets];
    __shared__ uint32_t pieces[kPlanWaves][kPlanBuckets];
    __shared__ double cand_bound[kPlanCandidates], cand_d[kPlanCandidates];
    __shared__ double cand_longest[kPlanCandidates], cand_load[kPlanCandidates], cand_late[kPlanCandidates];
    __shared__ uint32_t n_used;
    __shared__ double best_bound, best_d, best_longest, best_load, best_late;
    // The one-lane chain as this batch's spread in memory prices it (every thread computes the same).
    {
        const unsigned long long lo = ~st->span_lo_inverted, hi = st->span_hi;
        const double span_gb = hi > lo ? static_cast<double>(hi - lo) / 1073741824.0 : 0.0;
        m.chain_ns[kSha1Paths - 1] = fmin(m.chain_ns[kSha1Paths - 1] + m.one_lane_ns_per_gb * span_gb,
                                          fmax(m.one_lane_cap_ns, m.chain_ns[kSha1Paths - 1]));
        m.simd_ns[kSha1Paths - 1] = m.chain_ns[kSha1Paths - 1] / 64.0;
    }
    const uint32_t k = threadIdx.x;
    const uint32_t lane = k & 63u, wave = k >> 6;
    const uint32_t count = st->hist[k];
    scan[k] = count;
    if (k == 0)
        n_used = 0;
    __syncthreads();
    for (uint32_t d = 1; d < kPlanBuckets; d <<= 1) {
        const uint32_t v = (k >= d) ? scan[k - d] : 0u;
        __syncthreads();
        scan[k] += v;
        __syncthreads();
    }
    st->cursor[k] = scan[k] - count;
    if (count)
        used[atomicAdd(&n_used, 1u)] = k;
    __syncthreads();

    const double max_blocks = static_cast<double>(st->max_blocks);
    const double simds = static_cast<double>(m.n_simd);
    const uint32_t u_count = n_used;
    if (k < u_count) {
        const uint32_t pos = used[k];
        u_edge[k] = fmin(bucket_floor(kPlanBuckets - pos), max_blocks);
        u_blocks[k] = static_cast<double>(st->blocks[pos]);
        u_files[k] = st->hist[pos];
    }
    if (k == 0) {
        best_bound = -1.0;
        best_d = 0.0;
    }
    __syncthreads();

    if (m.forced_path < 0) {
        // Coarse pass: from "everything may take the slowest path" down by sqrt(2)
        // per step; then the largest file's own thresholds (just below them it moves
        // to the next faster path while everything else keeps the most room); then a
        // fine pass around the best so far.  Each wave of the workgroup weighs one
        // candidate at a time; the choice among them is made in candidate order.
        const double d_top = max_blocks * m.chain_ns[kSha1Paths - 1] * 1.0001;
        // With no more files than SIMDs every piece starts at once, the batch lasts as
        // long as its largest file on the fastest path, and the cheapest plan is to
        // let everything else take that long too: one candidate settles it.
        const bool one_round = n <= m.n_simd;
        for (int stage = 0; stage < (one_round ? 1 : 2); ++stage) {
            const int c_lo = one_round ? kCoarseCandidates : (stage == 0 ? 0 : kCoarseCandidates + kOwnCandidates);
            const int c_hi = one_round ? kCoarseCandidates + 1
                                       : (stage == 0 ? kCoarseCandidates + kOwnCandidates : kPlanCandidates);
            for (int c0 = c_lo; c0 < c_hi; c0 += kPlanWaves) {
                const int c = c0 + static_cast<int>(wave);
                const bool on = c < c_hi;
                const double d_max = candidate_d(on ? c : c_lo, d_top, max_blocks, best_d, m);
                double total = 0.0;
                float longest = 0.f;
                for (uint32_t v = lane; on && v < u_count; v += 64u) {
                    const double edge = u_edge[v];
                    const int path = path_for(edge, d_max, m);
                    const uint32_t per_wave = 64u / static_cast<uint32_t>(kHopGroupLanes[path]);
                    const float d = static_cast<float>(edge * m.chain_ns[path]);
                    dur[wave][v] = d;
                    pieces[wave][v] = (u_files[v] + per_wave - 1) / per_wave;
                    total += u_blocks[v] * m.simd_ns[path];
                    longest = fmaxf(longest, d);
                }
                __syncthreads();
                // Duration of the first piece that has to wait for a SIMD.
                float late = 0.f;
                for (uint32_t v = lane; on && v < u_count; v += 64u) {
                    const float mine = dur[wave][v];
                    uint32_t before = 0;  // pieces handed out before this bucket's
                    for (uint32_t x = 0; x < u_count; ++x) {
                        const float dx = dur[wave][x];
                        if (dx > mine || (dx == mine && x < v))
                            before += pieces[wave][x];
                    }
                    if (before <= m.n_simd && m.n_simd < before + pieces[wave][v])
                        late = mine;
                }
                for (int sh = 32; sh > 0; sh >>= 1) {
                    total += __shfl_xor(total, sh);
                    longest = fmaxf(longest, __shfl_xor(longest, sh));
                    late = fmaxf(late, __shfl_xor(late, sh));
                }
                if (on && lane == 0) {
                    // The busier the chip, the lower its clock: scale the chains by the
                    // share of SIMD time in use.
                    const double load = total / simds;
                    const double first = fmax(static_cast<double>(longest), load + static_cast<double>(late));
                    const double slow = 1.0 + m.load_slowdown * fmin(1.0, load / first);
                    cand_bound[c] = fmax(static_cast<double>(longest) * slow, load + static_cast<double>(late) * slow);
                    cand_d[c] = d_max;
                    cand_longest[c] = static_cast<double>(longest) * slow;
                    cand_load[c] = load;
                    cand_late[c] = static_cast<double>(late) * slow;
                }
                __syncthreads();
            }
            if (k == 0) {
                for (int c = c_lo; c < c_hi; ++c)
                    if (best_bound < 0.0 || cand_bound[c] < 0.98 * best_bound) {  // move on only for a clear gain
                        best_bound = cand_bound[c];
                        best_d = cand_d[c];
                        best_longest = cand_longest[c];
                        best_load = cand_load[c];
                        best_late = cand_late[c];
                    }
            }
            __syncthreads();
        }
    }
    if (k != 0)
        return;

    uint32_t cut[kSha1Paths];
    for (int q = 0; q < kSha1Paths; ++q)
        cut[q] = 0;
    uint32_t waves = 1;
    if (m.forced_path >= 0) {
        for (int q = m.forced_path; q < kSha1Paths; ++q)
            cut[q] = n;
    } else {
        // When nearly all the SIMD time goes into one-lane groups the batch is a
        // throughput job: a second wave per SIMD then hides what the first one
        // waits for.  Otherwise chains matter and a shared SIMD would halve their speed.
        double bulk = 0.0, all = 0.0;
        for (uint32_t u = 0; u < u_count; ++u) {
            const uint32_t p = used[u];
            const double e = fmin(bucket_floor(kPlanBuckets - p), max_blocks);
            const int path = path_for(e, best_d, m);
            if (scan[p] > cut[path])
                cut[path] = scan[p];
            const double w = static_cast<double>(st->blocks[p]) * m.simd_ns[path];
            all += w;
            if (path == kSha1Paths - 1)
                bulk += w;
        }
        // Ends of empty paths follow their predecessor; the last path ends the list.
        for (int q = 1; q < kSha1Paths; ++q)
            if (cut[q] < cut[q - 1])
                cut[q] = cut[q - 1];
        cut[kSha1Paths - 1] = n;
        waves = (bulk >= 0.9 * all && n >= 128u * m.n_simd) ? 2u : 1u;
    }
    for (int q = 0; q < kSha1Paths; ++q)
        st->path_end[q] = cut[q];
    for (int q = 0; q < kSha1Paths; ++q)
        st->chain_ns[q] = static_cast<float>(m.chain_ns[q]);
    if (m.one_lane_order_ns > 0.0)
        st->chain_ns[kSha1Paths - 1] = static_cast<float>(fmin(m.chain_ns[kSha1Paths - 1], m.one_lane_order_ns));
    if (m.forced_waves > 0)
        waves = m.forced_waves < kMaxWavesPerSimd ? m.forced_waves : kMaxWavesPerSimd;
    st->waves_per_simd = waves;
    st->task_base = first;
    // The cut minimises the list-scheduling BOUND max(longest, load + late).  What is reported is the
    // expected finishing time under it: the pieces that wait for a SIMD are handed out longest first,
    // so what sticks out over the average load at the end is on average half a late piece, and a
    // second wave per SIMD hides a fifth of the one-lane path's SIMD time.
    const double load = best_load * (waves == 2u ? 0.8 : 1.0);
    st->planned_ns = m.forced_path >= 0 ? 0.0 : fmax(best_longest, load + 0.5 * best_late);
}

// Pass 3: scatter the tasks to their bucket ranges.  A workgroup counts its files
// per bucket in LDS, reserves its ranges with one global atomic per bucket, then
// places its files.
__global__ void __launch_bounds__(kPlanThreads)
plan_scatter(const uint64_t *__restrict__ offsets, const uint64_t *__restrict__ lengths,
             const uint32_t *__restrict__ epochs, uint32_t n, Sha1PlanState *__restrict__ st,
             Sha1Task *__restrict__ tasks, uint32_t first)
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
            t.out_index = first + i;
            t.ready_epoch = epochs ? epochs[i] : 0u;
            tasks[base_of[bucket[j]] + slot[j]] = t;
        }
    }
}

// A batch of one message needs no sorting and no search: one thread writes the
// task and a header that gives the message the fastest chain (or the forced
// path).  A single GetHash call is three kernel launches shorter for it.
__global__ void plan_single(const uint64_t *__restrict__ offsets, const uint64_t *__restrict__ lengths,
                            const uint32_t *__restrict__ epochs, Sha1PlanState *__restrict__ st,
                            Sha1Task *__restrict__ tasks, Sha1CostModel m, uint32_t first)
{
    if (threadIdx.x != 0)
        return;
    const uint64_t blocks = padded_blocks(lengths[0]);
    const int path = m.forced_path >= 0 ? m.forced_path : 0;
    for (int q = 0; q < kSha1Paths; ++q) {
        st->path_end[q] = q >= path ? 1u : 0u;
        st->chain_ns[q] = static_cast<float>(m.chain_ns[q]);
        st->queue[q].next = 0;
    }
    st->total_blocks = blocks;
    st->max_blocks = blocks;
    st->planned_ns = m.forced_path >= 0 ? 0.0 : static_cast<double>(blocks) * m.chain_ns[0];
    st->waves_per_simd = 1;
    st->task_base = first;
    Sha1Task t;
    t.offset = offsets[0];
    t.len = lengths[0];
    t.out_index = first;
    t.ready_epoch = epochs ? epochs[0] : 0u;
    tasks[0] = t;
}

}  // namespace

hipError_t launch_sha1_plan(const uint64_t *d_offsets, const uint64_t *d_lengths,
                            const uint32_t *d_epochs, uint32_t n, Sha1PlanState *d_state,
                            Sha1Task *d_tasks, const Sha1CostModel &model, hipStream_t stream, uint32_t first)
{
    if (n == 0)
        return hipSuccess;
    d_offsets += first;
    d_lengths += first;
    if (d_epochs)
        d_epochs += first;
    d_tasks += first;
    if (n == 1) {
        hipLaunchKernelGGL(plan_single, dim3(1), dim3(64), 0, stream, d_offsets, d_lengths, d_epochs,
                           d_state, d_tasks, model, first);
        return hipGetLastError();
    }
    hipError_t e = hipMemsetAsync(d_state, 0, sizeof(Sha1PlanState), stream);
    if (e != hipSuccess)
        return e;
    const uint32_t per_group = kPlanThreads * kFilesPerThread;
    const uint32_t groups = (n + per_group - 1) / per_group;
    hipLaunchKernelGGL(plan_histogram, dim3(groups), dim3(kPlanThreads), 0, stream, d_offsets, d_lengths, n,
                       d_state);
    hipLaunchKernelGGL(plan_scan, dim3(1), dim3(kPlanBuckets), 0, stream, d_state, n, model, first);
    hipLaunchKernelGGL(plan_scatter, dim3(groups), dim3(kPlanThreads), 0, stream, d_offsets,
                       d_lengths, d_epochs, n, d_state, d_tasks, first);
    return hipGetLastError();
}

}  // namespace snappy
