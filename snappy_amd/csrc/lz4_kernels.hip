// Wave-cooperative LZ4 block decoder for gfx950.
//
// Decodes the chunks of an lz4net LZ4Stream -- what
// /root/reference/Snappy/Features/Mcdf/McdfManager.cs:73-79 obtains from
// `new LZ4Stream(fs, Decompress, ...).CopyTo(ms)`.  Chunks are independent raw
// LZ4 blocks (or stored bytes), so one wavefront owns one chunk at a time and
// draws chunks from a device queue.
//
// Inside a chunk the token stream is serial.  Compressed bytes are staged in
// LDS and parsed out of a 256-byte window spread over the wave's 64 lanes (one
// dword per lane, v_readlane), so parsing waits on memory once per 8 KiB.
//
// Long sequences are copied one at a time by all 64 lanes, consecutive lanes on
// consecutive bytes.  A match whose offset is smaller than its length repeats a
// pattern of `offset` bytes; it is copied as out[op+i] = out[op-offset + i mod
// offset], which reads only bytes written by earlier sequences, so every lane
// can go at once.  Runs of short sequences are taken up to 64 at a time, one
// sequence per lane (take_batch below).
//
// Acceptance rules are those of the decoder lz4net ports (LZ4 r7x
// LZ4_uncompress with a known output length, as recalled): a literal run that
// ends within 8 bytes of the end must end the block exactly; a match must end at
// least 5 bytes before the end of the block; consumed input must equal the chunk's
// compressed length.  Every read and write is bounds-checked against the chunk, so a
// corrupt stream yields a status word, never a stray access.
#include "lz4_launch.hpp"
#include "wave_util.hpp"

#include <algorithm>
#include <cstdio>
#include <cstdlib>

namespace snappy {

namespace {

constexpr int kWave = 64;
constexpr uint32_t kCopyLength = 8;

typedef __attribute__((address_space(3))) uint8_t LdsByte;  // an LDS address is 32 bits
typedef __attribute__((address_space(3))) uint32_t LdsWord;

__device__ __forceinline__ uint32_t uni(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }

// A pointer that went through LDS (the pair's command slots) comes back as a generic one, and
// the compiler then uses flat instructions for it: those count as LDS operations too, so every
// wait for an LDS result also waits for the loads from memory in flight, and before each flat
// load everything is drained because it might read LDS.  This says what the pointer is (device
// memory, the same address in every lane), and the accesses become global ones again.
template <typename T>
__device__ __forceinline__ T *known_global(T *p)
{
    const uint64_t v = reinterpret_cast<uintptr_t>(p);
    const uint64_t same = (static_cast<uint64_t>(uni(static_cast<uint32_t>(v >> 32))) << 32) | uni(static_cast<uint32_t>(v));
    typedef __attribute__((address_space(1))) T GlobalT;
    return (T *)(GlobalT *)same;
}

// All vector memory operations issued so far have completed (s_waitcnt vmcnt(0);
// the other two counters are left at their maximum).
__device__ __forceinline__ void wait_for_memory() { __builtin_amdgcn_s_waitcnt(0x0F70); }

// The LDS counter alone: all LDS reads issued so far have returned.
__device__ __forceinline__ void wait_for_lds() { __builtin_amdgcn_s_waitcnt(0xC07F); }

// The chunk's compressed bytes on their way to the parser.  A memory round trip
// takes as long as parsing some twenty tokens, so input is staged in LDS, up to
// kStageBlocks blocks of 256 bytes per trip.  Tokens are parsed out of a
// 256-byte window in registers, one dword per lane, which starts at any dword
// and is moved on with one LDS read.
constexpr uint32_t kStageBlocks = 24;  // 6 KiB per wave (and behind them one dword per lane: see produce_batch)
constexpr uint32_t kJumpBlocks = 4;    // staged after a jump (a long literal run was skipped)

struct InputWindow {
    const uint32_t *aligned;  // dword-aligned pointer at or below the chunk's first byte
    uint32_t limit_dwords;    // dwords that may be read from `aligned`
    uint32_t *stage;          // LDS, kStageBlocks * 64 dwords
    uint32_t stage_block;     // uniform: number of the first staged block
    uint32_t stage_count;     // uniform: blocks staged
    uint32_t first;           // uniform: number of the dword held by lane 0
    uint32_t a;               // per lane: dword first + lane

    // Every wait is placed here, on the paths that move the window.  A wait left
    // to the compiler lands where the paths join again, is executed by every
    // token, and waits for whatever stores are in flight by then as well (loads
    // and stores share one in-order counter).
    //
    // The bytes go from memory straight into LDS (global_load_lds_dwordx4: 16 bytes per lane,
    // 1 KiB = four blocks per instruction, no vector registers on the way), so all of a trip's
    // loads are in flight together and one wait covers them; through registers a trip of 6 KiB
    // was three round trips one after the other, a sixth of the parsing wave's time on streams
    // of short sequences.  A piece that would reach past the readable bytes is taken from the
    // chunk's first bytes instead: no position at or beyond the chunk's end is ever parsed, so
    // what the stage holds there does not matter.
    __device__ __forceinline__ void load_kib(uint32_t blk, uint32_t slot, uint32_t n_kib, uint32_t lane)
    {
        for (uint32_t g = 0; g < n_kib; ++g) {
            const uint32_t idx = (blk + 4u * g) * 64u + 4u * lane;   // first dword of this lane's 16 bytes
            const uint32_t *from = aligned + (idx + 4u <= limit_dwords && idx + 4u > idx ? idx : 0u);
            __builtin_amdgcn_global_load_lds(from, stage + (slot + 4u * g) * 64u, 16, 0, 0);
        }
    }

    // Stages the blocks from `blk` on: all of them when the parser ran off the end
    // of what was staged (tokens are dense here), a few after a jump.
    __device__ __forceinline__ void restage(uint32_t blk, bool all, uint32_t lane)
    {
        static_assert(kJumpBlocks % 4u == 0 && kStageBlocks % 4u == 0, "staged 1 KiB at a time");
        // The LDS reads of the old content have returned before it is overwritten.
        wait_for_lds();
        load_kib(blk, 0, (all ? kStageBlocks : kJumpBlocks) / 4u, lane);
        wait_for_memory();
        asm volatile("" ::: "memory");
        stage_count = all ? kStageBlocks : kJumpBlocks;
        stage_block = blk;
    }

    // Position one past the staged bytes.
    __device__ __forceinline__ uint32_t stage_end() const { return (stage_block + stage_count) * 256u; }

    // Moves the window so that it starts with the dword of byte position `pos`.
    __device__ __forceinline__ void slide(uint32_t pos, uint32_t lane)
    {
        const uint32_t dw = pos >> 2;
        uint32_t at = dw - stage_block * 64u;  // huge when dw lies before the stage
        if (at + 64u > stage_count * 64u) {
            // Running off the end of a full stage means dense tokens; anything else is a jump.
            const bool dense = stage_count == kStageBlocks && at < (kStageBlocks + 1u) * 64u;
            restage(pos >> 8, dense, lane);
            at = dw - stage_block * 64u;
        }
        a = stage[at + lane];
        first = dw;
        wait_for_lds();
    }

    __device__ __forceinline__ void reset(uint32_t lane)
    {
        restage(0, false, lane);
        slide(0, lane);
    }

    // The four bytes at byte position `pos` (counted from `aligned`), little-endian.
    __device__ __forceinline__ uint32_t peek32(uint32_t pos, uint32_t lane)
    {
        uint32_t di = (pos >> 2) - first;
        if (__builtin_expect(di >= 63u, 0)) {  // also catches positions before the window
            slide(pos, lane);
            di = 0;
        }
        const uint32_t lo = __builtin_amdgcn_readlane(a, di);
        const uint32_t hi = __builtin_amdgcn_readlane(a, di + 1u);
        const uint64_t both = (static_cast<uint64_t>(hi) << 32) | lo;
        return static_cast<uint32_t>(both >> ((pos & 3u) * 8u));
    }

    // Makes the window hold the n bytes from `pos` on (n <= 252).
    __device__ __forceinline__ void cover(uint32_t pos, uint32_t n, uint32_t lane)
    {
        if ((pos >> 2) < first || ((pos + n + 3u) >> 2) > first + 64u)
            slide(pos, lane);
    }

    // Four staged bytes at `pos`, which any lane may ask for; [pos, pos + 4) is staged.
    // Two aligned dwords and a funnel shift: an LDS access that is not naturally aligned
    // occupies the CU's LDS pipeline for 24 clocks instead of 2-3 (tools/ubench_lds_access.hip),
    // and eight chunks share that pipeline.
    __device__ __forceinline__ uint32_t staged32(uint32_t pos) const
    {
        const uint32_t i = (pos >> 2) - stage_block * 64u;
        return __builtin_amdgcn_alignbit(stage[i + 1u], stage[i], pos << 3);  // shift = 8 (pos & 3)
    }
};

// Lanes copy n bytes, consecutive lanes on consecutive bytes.  src and dst do
// not overlap within the copy.
__device__ __forceinline__ void copy_bytes(uint8_t *dst, const uint8_t *src, uint32_t n,
                                           uint32_t lane)
{
    uint32_t i = lane;
    // Four loads in flight per lane before the first store.
#pragma clang loop vectorize(disable) unroll(disable)
    for (; i + 3u * kWave < n; i += 4u * kWave) {
        const uint8_t a = src[i], b = src[i + kWave], c = src[i + 2 * kWave], d = src[i + 3 * kWave];
        dst[i] = a;
        dst[i + kWave] = b;
        dst[i + 2 * kWave] = c;
        dst[i + 3 * kWave] = d;
    }
#pragma clang loop vectorize(disable) unroll(disable)
    for (; i < n; i += kWave)
        dst[i] = src[i];
}

// Same for long runs, 16 bytes per lane and four such loads in flight per lane
// (4 KiB per wave per iteration), so the copy is bound by bandwidth rather than
// by one memory round trip per 64 bytes.  The destination is brought to a
// 16-byte boundary; the source is read as dword-aligned 16-byte pieces plus one
// more dword and funnel-shifted (v_alignbyte_b32) by its misalignment.
struct Piece {
    uint32_t v[4];
    uint32_t extra;
};

__device__ __forceinline__ Piece load_piece(const uint32_t *s, uint32_t vec, uint32_t shift)
{
    Piece p;
    __builtin_memcpy(p.v, s + 4u * vec, 16);  // dword-aligned 16-byte load
    p.extra = shift ? s[4u * vec + 4u] : 0u;  // holds source bytes only when misaligned
    return p;
}

__device__ __forceinline__ void store_piece(uint8_t *dst, uint32_t vec, const Piece &p, uint32_t shift)
{
    uint4 o;
    o.x = __builtin_amdgcn_alignbyte(p.v[1], p.v[0], shift);
    o.y = __builtin_amdgcn_alignbyte(p.v[2], p.v[1], shift);
    o.z = __builtin_amdgcn_alignbyte(p.v[3], p.v[2], shift);
    o.w = __builtin_amdgcn_alignbyte(p.extra, p.v[3], shift);
    reinterpret_cast<uint4 *>(dst)[vec] = o;
}

template <uint32_t kDepth>
__device__ __forceinline__ void copy_long(uint8_t *__restrict__ dst, const uint8_t *__restrict__ src,
                                          uint32_t n, uint32_t lane)
{
    const uint32_t head = (16u - static_cast<uint32_t>(reinterpret_cast<uintptr_t>(dst) & 15u)) & 15u;
    if (head) {
        if (lane < head)
            dst[lane] = src[lane];
        dst += head;
        src += head;
        n -= head;
    }
    const uint32_t shift = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(src) & 3u);
    const uint32_t *s = reinterpret_cast<const uint32_t *>(src - shift);
    const uint32_t n_vec = n >> 4;
    uint32_t k = lane;
    // Software pipeline, kDepth KiB per step: the loads of step i+1 are issued
    // before the stores of step i.  Loads and stores share one in-order counter on
    // this hardware, so a load issued behind a store cannot be waited for without
    // waiting for the store as well; issued ahead of it, it can.  kDepth = 8 where one
    // wave does a chunk's copies, 4 where two share them (the pieces in flight are what
    // the kernel's register count hangs on).
    if (k + (kDepth - 1u) * kWave < n_vec) {
        Piece cur[kDepth];
#pragma unroll
        for (uint32_t j = 0; j < kDepth; ++j)
            cur[j] = load_piece(s, k + j * kWave, shift);
#pragma clang loop vectorize(disable) unroll(disable)
        for (; k + (2u * kDepth - 1u) * kWave < n_vec; k += kDepth * kWave) {
            Piece nxt[kDepth];
#pragma unroll
            for (uint32_t j = 0; j < kDepth; ++j)
                nxt[j] = load_piece(s, k + (kDepth + j) * kWave, shift);
#pragma unroll
            for (uint32_t j = 0; j < kDepth; ++j)
                store_piece(dst, k + j * kWave, cur[j], shift);
#pragma unroll
            for (uint32_t j = 0; j < kDepth; ++j)
                cur[j] = nxt[j];
        }
#pragma unroll
        for (uint32_t j = 0; j < kDepth; ++j)
            store_piece(dst, k + j * kWave, cur[j], shift);
        k += kDepth * kWave;
    }
#pragma clang loop vectorize(disable) unroll(disable)
    for (; k < n_vec; k += kWave)
        store_piece(dst, k, load_piece(s, k, shift), shift);
    const uint32_t done = n_vec << 4;
    if (lane < (n & 15u))
        dst[done + lane] = src[done + lane];
}

constexpr uint32_t kLongCopy = 512;

template <uint32_t kDepth = 8>
__device__ __forceinline__ void copy_run(uint8_t *dst, const uint8_t *src, uint32_t n,
                                         uint32_t lane)
{
    if (n >= kLongCopy)
        copy_long<kDepth>(dst, src, n, lane);
    else
        copy_bytes(dst, src, n, lane);
}

// Overlapping match: the output repeats the `period` bytes that end at dst.
// The first `span` bytes (a multiple of the period, at least kPeriodSpan) are
// written as dst[i] = dst[i mod period - period], which reads only bytes that
// existed before the match; the rest is filled by doubling.
constexpr uint32_t kPeriodSpan = 4096;
constexpr uint32_t kRingBytes = 8192;   // = kHist, declared below with the ring's other figures

// ring (may be null): the LDS ring when it holds the pattern, whose first byte is output position ring_pos.
template <typename Team>
__device__ __forceinline__ void copy_periodic(Team &team, uint8_t *dst, uint32_t period, uint32_t n,
                                              uint32_t lane, const uint8_t *ring = nullptr, uint32_t ring_pos = 0)
{
    if (ring != nullptr && period <= 16u && (16u % period) == 0u && n >= 4u * kWave) {
        // A run of one byte, or of a pattern of 2, 4, 8 or 16 bytes (zero padding, a repeated
        // value): every group of 16 bytes of the run is the same, so it is put together once,
        // out of the ring, and stored over and over.  Nothing is read back from memory, where
        // the general way below doubles what it has written.
        uint32_t w[4];
#pragma unroll
        for (uint32_t q = 0; q < 4u; ++q) {
            uint32_t word = 0;
#pragma unroll
            for (uint32_t j = 0; j < 4u; ++j)
                word |= static_cast<uint32_t>(ring[(ring_pos + ((4u * q + j) & (period - 1u))) & (kRingBytes - 1u)]) << (8u * j);
            w[q] = word;
        }
        const uint4 v = make_uint4(w[0], w[1], w[2], w[3]);
        const uint32_t groups = n >> 4;
        for (uint32_t g = lane; g < groups; g += kWave)
            __builtin_memcpy(dst + 16u * g, &v, 16);
        if (lane < (n & 15u))
            dst[16u * groups + lane] = ring[(ring_pos + (lane & (period - 1u))) & (kRingBytes - 1u)];
        return;
    }
    const uint8_t *pat = dst - period;
    const uint32_t reps = (kPeriodSpan + period - 1u) / period;
    const uint32_t span = reps * period;
    const uint32_t head = n < span ? n : span;
    if (period >= 4u * kWave) {
        // Long pattern: each repetition is a plain copy of the pattern itself.
        for (uint32_t at = 0; at < head; at += period)
            team.copy(dst + at, pat, (head - at < period) ? head - at : period, lane);
    } else if (ring != nullptr) {
        // The pattern is in the ring (it is what the sequence's literals and the matches before
        // them just wrote): read from there, byte r of the pattern at ring position ring_pos + r.
        // From memory every turn of this loop waited for a round trip to the L2, a microsecond,
        // and a pattern of 40 bytes repeated 40 times is six turns.
        uint32_t r = lane % period;
        const uint32_t step = kWave % period;
        uint32_t i = lane;
        // Eight rows of 64 bytes per turn: their reads are in flight together.  (The rows left
        // over go one by one: a last turn that stores only the rows it has, each store under its
        // own test, measured slower than that.)
#pragma clang loop vectorize(disable) unroll(disable)
        for (; i + 7u * kWave < head; i += 8u * kWave) {
            uint8_t v[8];
#pragma unroll
            for (uint32_t k = 0; k < 8u; ++k) {
                v[k] = ring[(ring_pos + r) & (kRingBytes - 1u)];
                r += step;
                r -= (r >= period) ? period : 0u;
            }
#pragma unroll
            for (uint32_t k = 0; k < 8u; ++k)
                dst[i + k * kWave] = v[k];
        }
#pragma clang loop vectorize(disable) unroll(disable)
        for (; i < head; i += kWave) {
            dst[i] = ring[(ring_pos + r) & (kRingBytes - 1u)];
            r += step;
            r -= (r >= period) ? period : 0u;
        }
    } else {
        uint32_t r = lane % period;
        const uint32_t step = kWave % period;
        uint32_t i = lane;
#pragma clang loop vectorize(disable) unroll(disable)
        for (; i + 3u * kWave < head; i += 4u * kWave) {
            uint32_t r1 = r + step;
            r1 -= (r1 >= period) ? period : 0u;
            uint32_t r2 = r1 + step;
            r2 -= (r2 >= period) ? period : 0u;
            uint32_t r3 = r2 + step;
            r3 -= (r3 >= period) ? period : 0u;
            const uint8_t a = pat[r], b = pat[r1], c = pat[r2], d = pat[r3];
            dst[i] = a;
            dst[i + kWave] = b;
            dst[i + 2 * kWave] = c;
            dst[i + 3 * kWave] = d;
            r = r3 + step;
            r -= (r >= period) ? period : 0u;
        }
#pragma clang loop vectorize(disable) unroll(disable)
        for (; i < head; i += kWave) {
            dst[i] = pat[r];
            r += step;
            r -= (r >= period) ? period : 0u;
        }
    }
    // What has been written so far is a whole number of periods, so the next
    // `done` bytes are a copy of the first `done`: the written part doubles with
    // every step, and each step is one plain copy between disjoint ranges.
    for (uint32_t done = head; done < n;) {
        const uint32_t piece = (n - done < done) ? n - done : done;
        team.copy(dst + done, dst, piece, lane);
        done += piece;
    }
}

// Short sequences.  A copy through global memory costs a memory round trip (and a
// match that reads what the previous sequence wrote also waits for that write),
// about a microsecond per sequence whatever its length; real LZ4 streams are
// mostly sequences of tens of bytes.  So short literal runs are taken from the
// input window the wave already holds in registers, and every short piece of
// output is also written to a ring of the wave's most recent kHist output bytes
// in LDS, from which short matches read.  The ring holds the output positions
// [valid_from, op); long copies bypass it and reset valid_from.
constexpr uint32_t kHist = 8192;       // bytes of recent output per wave in LDS
static_assert(kHist == kRingBytes, "one ring");
constexpr uint32_t kShortLit = 192;    // longest literal run served from the window registers
constexpr uint32_t kShortMatch = 512;  // longest match served from the ring

__device__ __forceinline__ void emit_short_literals(const InputWindow &win, uint32_t ip,
                                                    uint32_t lit, uint8_t *out, uint8_t *ring,
                                                    uint32_t op, uint32_t lane)
{
    // The caller has made sure the window covers [ip, ip + lit).
    for (uint32_t base = 0; base < lit; base += kWave) {
        const uint32_t i = base + lane;
        const bool on = i < lit;
        const uint32_t pos = ip + (on ? i : 0u);
        const uint32_t from_lane = (pos >> 2) - win.first;
        const uint32_t word = static_cast<uint32_t>(
            __builtin_amdgcn_ds_bpermute(static_cast<int>(from_lane << 2), static_cast<int>(win.a)));
        const uint8_t byte = static_cast<uint8_t>(word >> ((pos & 3u) * 8u));
        if (on) {
            out[op + i] = byte;
            ring[(op + i) & (kHist - 1u)] = byte;
        }
    }
}

__device__ __forceinline__ void emit_short_match(uint8_t *out, uint8_t *ring, uint32_t op,
                                                 uint32_t offset, uint32_t ml, uint32_t lane)
{
    const uint32_t src0 = op - offset;
    const bool overlapping = offset < ml;
    // i mod offset for i < 1024: exact in float up to one step either way.
    const float inv = 1.0f / static_cast<float>(offset);
    for (uint32_t base = 0; base < ml; base += kWave) {
        const uint32_t i = base + lane;
        uint32_t k = i;
        if (overlapping) {
            const uint32_t q = static_cast<uint32_t>(static_cast<float>(i) * inv);
            k = i - q * offset;
            k += (static_cast<int32_t>(k) < 0) ? offset : 0u;
            k -= (k >= offset) ? offset : 0u;
        }
        // The pattern [src0, src0 + offset) lies before op and, offset + ml being
        // at most kHist, is not overwritten by this match's own ring writes.
        const uint8_t byte = ring[(src0 + k) & (kHist - 1u)];
        if (i < ml) {
            out[op + i] = byte;
            ring[(op + i) & (kHist - 1u)] = byte;
        }
    }
}

// Runs of plain short sequences are taken up to 64 at a time.  The only serial
// part is finding where each token starts, which takes the token and the byte
// behind it and touches no memory.  Lane j then reads sequence j's lengths and
// offset out of the staged input, a prefix sum gives every sequence its place in
// the output, and the batch is cut in front of the first sequence it must not
// take.  Every lane copies its own sequence's literals from the stage.  The
// matches follow in rounds: a match is copied once everything it reads is
// final, that is, once its source ends at or before the output of the first
// match still waiting.  Sources in reach of the ring are read from LDS; sources
// that end before the batch's output begins are read from memory.
//
// A sequence is taken only if the one-at-a-time code below would accept it with
// room to spare, so every rule about the end of a block stays with that code:
// at most one length-extension byte each, a match that does not overlap its own
// output within one copy step, 4 more input bytes and 16 more output bytes
// behind it, and a source that is either in the ring or entirely older than the
// batch.
constexpr uint32_t kBatchLit = 240;
// Measurement builds (-DSNAPPY_LZ4_PHASE_SWITCHES) can leave out one kind of copy
// (tuning bits 1-3; the output is then wrong) to see what each costs.
#ifdef SNAPPY_LZ4_PHASE_SWITCHES
constexpr uint32_t kPhaseSwitches = 0xEu;
__device__ unsigned long long g_phase_cycles[12];
struct PhaseClock {
    uint64_t last;
    uint64_t acc[12];
    __device__ void start() { last = __builtin_amdgcn_s_memtime(); for (int i = 0; i < 12; ++i) acc[i] = 0; }
    __device__ void mark(int phase) { const uint64_t now = __builtin_amdgcn_s_memtime(); acc[phase] += now - last; last = now; }
    __device__ void flush(uint32_t lane) { if (lane == 0) for (int i = 0; i < 12; ++i) atomicAdd(&g_phase_cycles[i], acc[i]); }
};
#define PHASE_MARK(clock, phase) (clock).mark(phase)
#else
constexpr uint32_t kPhaseSwitches = 0u;
struct PhaseClock {
    __device__ void start() {}
    __device__ void flush(uint32_t) {}
};
#define PHASE_MARK(clock, phase) ((void)0)
#endif
constexpr uint32_t kBatchMatch = 255;
// Most output one batch produces, and how far behind a batch's start the ring is still read
// (kReach).  One wave doing everything: the ring must hold a batch and its reach.  Two
// waves: the parser already writes the next batch's literals into the ring while the helper
// copies this batch's matches, so the ring must hold two batches and the reach.
constexpr uint32_t kSoloSpan = 6144, kSoloReach = 2048;
constexpr uint32_t kPairSpan = 3072, kPairReach = 2048;
constexpr uint32_t kBatchInput = 512;   // staged bytes a sequence may touch from its token on
constexpr uint32_t kLitStep = 8;        // dwords a lane has in flight per step when copying literals
constexpr uint32_t kMatchStep = 4;      // and matches: a match may repeat a pattern of 16 bytes or more
constexpr uint32_t kFarStep = 10;       // and matches read from memory: 37 bytes or more per round trip
#ifndef SNAPPY_LZ4_FAR_DEPTH
#define SNAPPY_LZ4_FAR_DEPTH 3
#endif
constexpr uint32_t kFarDepth = SNAPPY_LZ4_FAR_DEPTH;  // steps of such a match in flight

// A lane's copy of n bytes to output position dst of the ring, done with naturally
// aligned LDS accesses only.  The copy is cut along the dwords of the destination:
// dword k of the copy is ring dword (dst >> 2) + k, and its bytes come from the source
// bytes B0 + 4 k .. B0 + 4 k + 3, B0 = src - (dst & 3): the funnel shift (v_alignbyte)
// of two neighbouring aligned source dwords, S[k] and S[k + 1], by m = B0 & 3 bytes.
// Dwords that lie wholly inside the copy are stored whole; the first and the last, when
// they are shared with a neighbouring copy (of another lane, or of another wave), are
// stored byte by byte, so nothing outside [dst, dst + n) is ever written and nothing is
// read, modified and written back.  The source may be the staged input, the ring itself
// (each step reads what the steps before it wrote) or dwords fetched from memory.
struct LaneCopy {
    uint32_t d0;        // ring dword number of dword 0 (not yet reduced modulo the ring)
    uint32_t lead;      // bytes of dword 0 in front of dst
    uint32_t total;     // lead + n: bytes from dword 0's start to the copy's end (0: nothing to copy)
    uint32_t count;     // dwords touched
    uint32_t full_beg;  // dwords [full_beg, full_end) are stored whole
    uint32_t full_end;
    int32_t sbase;      // number of the source dword S[0], in the source's own dword numbering
    uint32_t m;         // bytes to shift

    __device__ __forceinline__ LaneCopy(uint32_t dst, uint32_t n, int32_t src)
    {
        d0 = dst >> 2;
        lead = dst & 3u;
        total = n ? lead + n : 0u;
        count = (total + 3u) >> 2;
        full_beg = lead ? 1u : 0u;
        full_end = (total & 3u) ? count - 1u : count;
        const int32_t b0 = src - static_cast<int32_t>(lead);
        sbase = b0 >> 2;
        m = static_cast<uint32_t>(b0) & 3u;
    }
    // Source dword j is wanted by this lane in a step that starts at dword k0.
    __device__ __forceinline__ bool wants(uint32_t j) const { return j <= count && count != 0u; }
};

// The stores of one step, dwords 1 .. kStep - 1 of it: lane by lane the dwords to store
// whole form a run that ends at `rem` (the step's dwords below full_end), so EXEC only
// ever narrows: one v_cmpx and one ds_write per dword, where the compiler's code for
// "if (i < rem) at[i] = w[i]" is a compare, a save and a restore of EXEC and a branch
// around each store.  EXEC is back to what it was when the statement ends.
#define SNAPPY_LZ4_ST(i, off) "v_cmpx_lt_u32_e32 vcc, " #i ", %[rem]\n\tds_write_b32 %[at], %[w" #i "] offset:" #off "\n\t"
template <uint32_t kStep>
__device__ __forceinline__ void store_run(uint32_t at, uint32_t rem, const uint32_t (&w)[kStep])
{
    static_assert(kStep == 4u || kStep == 8u || kStep == 10u, "one asm statement per step size");
    uint64_t save;
    if constexpr (kStep == 4u) {
        asm volatile("s_mov_b64 %[save], exec\n\t" SNAPPY_LZ4_ST(1, 4) SNAPPY_LZ4_ST(2, 8) SNAPPY_LZ4_ST(3, 12)
                     "s_mov_b64 exec, %[save]"
                     : [save] "=&s"(save)
                     : [rem] "v"(rem), [at] "v"(at), [w1] "v"(w[1]), [w2] "v"(w[2]), [w3] "v"(w[3])
                     : "vcc", "memory");
    } else if constexpr (kStep == 8u) {
        asm volatile("s_mov_b64 %[save], exec\n\t" SNAPPY_LZ4_ST(1, 4) SNAPPY_LZ4_ST(2, 8) SNAPPY_LZ4_ST(3, 12)
                     SNAPPY_LZ4_ST(4, 16) SNAPPY_LZ4_ST(5, 20) SNAPPY_LZ4_ST(6, 24) SNAPPY_LZ4_ST(7, 28)
                     "s_mov_b64 exec, %[save]"
                     : [save] "=&s"(save)
                     : [rem] "v"(rem), [at] "v"(at), [w1] "v"(w[1]), [w2] "v"(w[2]), [w3] "v"(w[3]), [w4] "v"(w[4]),
                       [w5] "v"(w[5]), [w6] "v"(w[6]), [w7] "v"(w[7])
                     : "vcc", "memory");
    } else {
        asm volatile("s_mov_b64 %[save], exec\n\t" SNAPPY_LZ4_ST(1, 4) SNAPPY_LZ4_ST(2, 8) SNAPPY_LZ4_ST(3, 12)
                     SNAPPY_LZ4_ST(4, 16) SNAPPY_LZ4_ST(5, 20) SNAPPY_LZ4_ST(6, 24) SNAPPY_LZ4_ST(7, 28)
                     SNAPPY_LZ4_ST(8, 32) SNAPPY_LZ4_ST(9, 36)
                     "s_mov_b64 exec, %[save]"
                     : [save] "=&s"(save)
                     : [rem] "v"(rem), [at] "v"(at), [w1] "v"(w[1]), [w2] "v"(w[2]), [w3] "v"(w[3]), [w4] "v"(w[4]),
                       [w5] "v"(w[5]), [w6] "v"(w[6]), [w7] "v"(w[7]), [w8] "v"(w[8]), [w9] "v"(w[9])
                     : "vcc", "memory");
    }
}
#undef SNAPPY_LZ4_ST

// The LDS byte address of a ring position's dword.
__device__ __forceinline__ uint32_t ring_address(const uint8_t *ring, uint32_t dword)
{
    return static_cast<uint32_t>(reinterpret_cast<uintptr_t>((LdsByte *)ring)) + ((dword & (kHist / 4u - 1u)) << 2);
}

// One step: dwords k0 .. k0 + kStep - 1 from S[0 .. kStep] = source dwords k0 .. k0 + kStep.
// Most of a batch's instructions are spent here, so: stores go through one base address
// and constant offsets unless some lane's dwords wrap around the ring's end in this step;
// the caller's source reads are unconditional (an LDS read of an address the lane has no
// business with is harmless and cheaper than the exec-mask dance around it).
template <uint32_t kStep>
__device__ __forceinline__ void lane_copy_step(uint8_t *ring, const LaneCopy &c, uint32_t k0,
                                               const uint32_t (&S)[kStep + 1u])
{
    uint32_t *ring32 = reinterpret_cast<uint32_t *>(ring);
    const uint32_t first = (c.d0 + k0) & (kHist / 4u - 1u);
    uint32_t w[kStep];
#pragma unroll
    for (uint32_t i = 0; i < kStep; ++i)
        w[i] = __builtin_amdgcn_alignbyte(S[i + 1u], S[i], c.m);
    const int32_t left = static_cast<int32_t>(c.full_end - k0);  // the step's dwords below full_end
    const uint32_t rem = left > 0 ? static_cast<uint32_t>(left) : 0u;
    if (__builtin_amdgcn_ballot_w64(first + kStep > kHist / 4u && rem != 0u) == 0) {
        if (rem != 0u && k0 >= c.full_beg)
            ring32[first] = w[0];
        store_run<kStep>(ring_address(ring, first), rem, w);
    } else {
#pragma unroll
        for (uint32_t i = 0; i < kStep; ++i)
            if (i < rem && (i != 0u || k0 >= c.full_beg))
                ring32[(first + i) & (kHist / 4u - 1u)] = w[i];
    }
    if (k0 == 0u) {
        // Dword 0's bytes lead .. min(4, total) - 1, when it is not stored whole.  EXEC
        // narrows from byte 3 down (byte 3 for every lead, byte 2 for leads 1 and 2,
        // byte 1 for lead 1) unless some lane's copy ends inside dword 0.
        const uint32_t lead = c.total ? c.lead : 0u;
        if (__builtin_amdgcn_ballot_w64(lead != 0u && c.total < 4u) == 0) {
            uint64_t save;
            uint32_t t;
            asm volatile("s_mov_b64 %[save], exec\n\t"
                         "v_cmpx_lt_u32_e32 vcc, 0, %[lead]\n\t"
                         "v_lshrrev_b32_e32 %[t], 24, %[w]\n\t"
                         "ds_write_b8 %[at], %[t] offset:3\n\t"
                         "v_cmpx_gt_u32_e32 vcc, 3, %[lead]\n\t"
                         "ds_write_b8_d16_hi %[at], %[w] offset:2\n\t"
                         "v_cmpx_gt_u32_e32 vcc, 2, %[lead]\n\t"
                         "v_lshrrev_b32_e32 %[t], 8, %[w]\n\t"
                         "ds_write_b8 %[at], %[t] offset:1\n\t"
                         "s_mov_b64 exec, %[save]"
                         : [save] "=&s"(save), [t] "=&v"(t)
                         : [lead] "v"(lead), [at] "v"(ring_address(ring, c.d0)), [w] "v"(w[0])
                         : "vcc", "memory");
        } else {
            const uint32_t at = (c.d0 * 4u) & (kHist - 1u);
#pragma unroll
            for (uint32_t q = 1; q < 4u; ++q)
                if (lead != 0u && q >= lead && q < c.total)
                    ring[at + q] = static_cast<uint8_t>(w[0] >> (8u * q));
        }
    }
}

// The last dword's bytes 0 .. (total & 3) - 1, unless dword 0 was the last and has been done.
// S0, S1: source dwords count - 1 and count.
__device__ __forceinline__ void lane_copy_tail(uint8_t *ring, const LaneCopy &c, uint32_t S0, uint32_t S1)
{
    const uint32_t w_tail = __builtin_amdgcn_alignbyte(S1, S0, c.m);
    const bool separate = (c.total & 3u) != 0u && c.count - 1u >= c.full_beg;
    const uint32_t tail = separate ? c.total & 3u : 0u;
    uint64_t save;
    uint32_t t;
    asm volatile("s_mov_b64 %[save], exec\n\t"
                 "v_cmpx_lt_u32_e32 vcc, 0, %[tail]\n\t"
                 "ds_write_b8 %[at], %[w]\n\t"
                 "v_cmpx_lt_u32_e32 vcc, 1, %[tail]\n\t"
                 "v_lshrrev_b32_e32 %[t], 8, %[w]\n\t"
                 "ds_write_b8 %[at], %[t] offset:1\n\t"
                 "v_cmpx_lt_u32_e32 vcc, 2, %[tail]\n\t"
                 "ds_write_b8_d16_hi %[at], %[w] offset:2\n\t"
                 "s_mov_b64 exec, %[save]"
                 : [save] "=&s"(save), [t] "=&v"(t)
                 : [tail] "v"(tail), [at] "v"(ring_address(ring, c.d0 + c.count - 1u)), [w] "v"(w_tail)
                 : "vcc", "memory");
}

// Output positions [beg, end), which the ring holds, go to memory: 16 bytes per lane
// and step, consecutive lanes on consecutive pieces, instead of the scattered
// dword stores a sequence per lane would make.  Pieces lie on multiples of 16
// in output position (and so in the ring); the bytes before the first whole piece
// and behind the last are stored one per lane.
__device__ __forceinline__ void flush_ring(uint8_t *dst, const uint8_t *ring, uint32_t beg, uint32_t end,
                                           uint32_t lane)
{
    const uint32_t up = (beg + 15u) & ~15u;
    const uint32_t head_end = up < end ? up : end;
    const uint32_t down = end & ~15u;
    const uint32_t tail_beg = down > head_end ? down : head_end;
    if (lane < head_end - beg)
        dst[beg + lane] = ring[(beg + lane) & (kHist - 1u)];
    if (lane < end - tail_beg)
        dst[tail_beg + lane] = ring[(tail_beg + lane) & (kHist - 1u)];
    for (uint32_t p = head_end + 16u * lane; p < tail_beg; p += 16u * kWave) {
        const uint4 v = *reinterpret_cast<const uint4 *>(ring + (p & (kHist - 1u)));
        __builtin_memcpy(dst + p, &v, 16);
    }
}

// Sum over the lanes up to and including this one.
__device__ __forceinline__ uint32_t wave_inclusive_sum(uint32_t v)
{
    // Within rows of 16 lanes: shifts by 1, 2, 4, 8 (lanes shifted in from outside the row add 0).
    v += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x111, 0xF, 0xF, true));
    v += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x112, 0xF, 0xF, true));
    v += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x114, 0xF, 0xF, true));
    v += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x118, 0xF, 0xF, true));
    // Lane 15 of a row into the next row (rows 1 and 3), then lane 31 into rows 2 and 3.
    v += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x142, 0xA, 0xF, false));
    v += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x143, 0xC, 0xF, false));
    return v;
}

// A batch as its first half (the walk and the classification) hands it to its
// second (the copies): what lane j needs to know about sequence j, and the
// figures all lanes share.
struct BatchLane {
    uint32_t lip;      // input position of the literals
    uint32_t lit;      // literal bytes
    uint32_t ml;       // match bytes
    uint32_t off;      // match offset
    uint32_t out_end;  // output position behind the sequence
};

struct BatchInfo {
    uint32_t take;        // sequences in the batch: lanes 0 .. take - 1
    uint32_t start;       // output position of the first
    uint32_t end;         // and behind the last
    uint32_t valid_from;  // the ring holds output positions [valid_from, start)
    uint32_t reach;       // sources down to start - reach are read from the ring, older ones from memory
    uint32_t stage_block; // number of the first staged block (the stage as the batch's literals are read from it)
    uint32_t literals;    // != 0: the literals are still to be copied (by whoever takes the batch)
};

// How the ring is shared out between a batch's output and the older output its matches may
// still read from it (the reach).  With two waves the parser already writes the next batch's
// literals while the helper copies this batch's matches, so reach + this batch + the next must
// fit the ring; where a stream's matches read the ring behind a batch (offsets of a few hundred
// bytes) the reach is worth 2 KiB, where they do not (offsets anywhere in 64 KiB) it is worth
// nothing and the batches can be larger.  Decided batch by batch from where the last batch's
// sources lay.
struct RingShare {
    uint32_t reach;       // of the next batch
    uint32_t prev_reach;  // of the batch the helper may still be copying (0: none)
    uint32_t prev_span;   // and its output bytes
};

// First half.  Returns the number of sequences taken (0: none, nothing changed but
// `walk`).  `walk` is how many tokens the next call should look at: a stream of
// sequences that are not for this code should not pay for 64 tokens walked in vain
// each time.
template <typename Team>
__device__ __forceinline__ uint32_t produce_batch(InputWindow &win, uint32_t &ip_io, uint32_t &op_io,
                                                  uint32_t valid_from, uint32_t iend, uint32_t oend,
                                                  uint32_t &walk, uint32_t lane, PhaseClock &clk,
                                                  BatchLane &desc, BatchInfo &info, RingShare &share, Team &team)
{
    PHASE_MARK(clk, 5);
    const uint32_t ip0 = ip_io;  // uniform
    const uint32_t start = op_io;
    const bool likely = walk >= 16u;  // recent batches were worth their while
    const uint32_t ahead = likely ? 3u * kBatchInput : 0u;
    if (ip0 < win.stage_block * 256u || ip0 + kBatchInput + ahead > win.stage_end()) {
        // Not enough staged for a batch.  Staging more pays when batches have been
        // taken lately; otherwise the one-at-a-time code stages what it needs.
        if (!likely || ip0 + kBatchInput > iend)
            return 0;
        team.before_restage();
        win.restage(ip0 >> 8, true, lane);
    }
    const uint32_t stage_end = win.stage_end();

    PHASE_MARK(clk, 6);
    // Where the tokens start: the decoder's serial part.  Every lane follows the same
    // chain through the staged bytes in LDS: one read of the token and the byte behind
    // it (all lanes the same address: a broadcast), the distance to the next token from
    // those two bytes, the next read.  Nothing else is looked at here: a literal run or
    // a match with more than one length-extension byte makes the positions from there
    // on meaningless; the lanes find that out below and the batch ends in front of such
    // a sequence.  The chain is one LDS round trip and a handful of vector instructions
    // per token, with no hop through the scalar unit (a walk over a 256-byte register
    // window with v_readlane took 230 clocks per token plus 620 per window move).
    const uint32_t ip_limit = stage_end - kBatchInput;  // at least kJumpBlocks blocks are staged
    const uint32_t quota = (walk + 7u) & ~7u;           // whole turns; at most 64
    uint32_t n = 0;
    uint32_t my_ip = 0xFFFFFFFFu;  // lane j: sequence j
    // The chain stays in vector registers: the value is the same in every lane, which
    // the compiler must not see (the empty asm), or it moves the arithmetic to the scalar
    // unit and pays a v_readfirstlane, a v_mov and twenty scalar instructions per token.
    // q is the LDS byte address of the staged byte at position p; bias is a multiple of 4.
    const uint32_t bias = static_cast<uint32_t>(reinterpret_cast<uintptr_t>((LdsByte *)win.stage)) -
                          win.stage_block * 256u;
    const uint32_t q_limit = ip_limit + bias;
    uint32_t q = ip0 + bias;
    asm volatile("" : "+v"(q));
    // Every instruction of a wave takes an issue slot whether the chain needs its result or not,
    // so the step is written out with as few as it takes: five vector instructions.  With t the
    // token and nb the byte behind it, the next token lies 3 + lit + [ml nibble = 15] +
    // [lit = 15] (nb + 1) bytes on: (t + 1) >> 4 is lit + [ml nibble = 15], and
    // e = t >= 0xF0 ? nb : -1 carries the extension byte or takes one off, which leaves 4 per
    // token to add; those go into the reads' offset fields (w is the address less 4 per token
    // of this turn).  The compare comes first and two instructions that do not need it follow,
    // which is the distance a v_cndmask wants from the instruction that wrote its mask.  Where
    // token n starts is not kept by a compare and a select per token either: it is written to
    // LDS (one store, the same word from every lane) and lane n reads it after the walk.
    // The step's LDS instructions are inside the statement as well, with its one wait (for the two
    // reads; the store behind them may still be on its way): left to the compiler, the reads'
    // destination registers are "maybe still being loaded" where the loop's paths join, and the
    // wait it puts there lands behind the turn's first read and makes two round trips of one.
    uint32_t *starts = win.stage + kStageBlocks * 64u;
    const uint32_t starts_at = static_cast<uint32_t>(reinterpret_cast<uintptr_t>((LdsByte *)starts));
    wait_for_lds();
    while (n < quota && uni(q) <= q_limit) {
        uint32_t w = q;
        const uint32_t slot_at = starts_at + 4u * n;
#define SNAPPY_LZ4_WALK_STEP(at, at1)                                                              \
        {                                                                                            \
            uint32_t t, nb, step, e;                                                                 \
            asm volatile("ds_read_u8 %[t], %[w] offset:" #at "\n\t"                                 \
                         "ds_read_u8 %[nb], %[w] offset:" #at1 "\n\t"                               \
                         "ds_write_b32 %[slot], %[w] offset:" #at "\n\t"                            \
                         "s_waitcnt lgkmcnt(1)\n\t"                                                 \
                         "v_cmp_lt_u32_e32 vcc, %[top], %[t]\n\t"                                   \
                         "v_add_u32_e32 %[step], 1, %[t]\n\t"                                       \
                         "v_lshrrev_b32_e32 %[step], 4, %[step]\n\t"                                \
                         "v_cndmask_b32_e32 %[e], -1, %[nb], vcc\n\t"                               \
                         "v_add3_u32 %[w], %[step], %[w], %[e]"                                      \
                         : [w] "+v"(w), [t] "=&v"(t), [nb] "=&v"(nb), [step] "=&v"(step), [e] "=&v"(e) \
                         : [slot] "v"(slot_at), [top] "s"(0xEFu)                                     \
                         : "vcc", "memory");                                                         \
        }
        // Token k of the turn is read at w + 4 k and noted in slot k as w (less 4 k: added by the lane that reads it).
        SNAPPY_LZ4_WALK_STEP(0, 1)
        SNAPPY_LZ4_WALK_STEP(4, 5)
        SNAPPY_LZ4_WALK_STEP(8, 9)
        SNAPPY_LZ4_WALK_STEP(12, 13)
        SNAPPY_LZ4_WALK_STEP(16, 17)
        SNAPPY_LZ4_WALK_STEP(20, 21)
        SNAPPY_LZ4_WALK_STEP(24, 25)
        SNAPPY_LZ4_WALK_STEP(28, 29)
#undef SNAPPY_LZ4_WALK_STEP
        n += 8u;
        q = w + 32u;
    }
    const uint32_t p = q - bias;
    // Lanes that have no token stay above ip_limit (bias is far below 2^31).
    if (lane < n)
        my_ip = starts[lane] + 4u * (lane & 7u) - bias;
    PHASE_MARK(clk, 0);
    if (n == 0) {
        walk = 4;
        return 0;
    }

    // What each sequence is.  A turn's later steps may have left the staged bytes a
    // sequence may touch: such a position is not taken.
    const bool walked = lane < n && my_ip <= ip_limit;  // my_ip: 0xFFFFFFFF - bias when nothing was recorded
    const uint32_t at = walked ? my_ip : ip0;
    const uint32_t r1 = win.staged32(at);
    uint32_t lit = (r1 >> 4) & 15u;
    uint32_t ml = r1 & 15u;
    uint32_t lip = at + 1u;
    if (lit == 15u) {
        lit += (r1 >> 8) & 255u;
        lip += 1u;
    }
    uint32_t after = lip + lit;  // the offset; then the next token
    const uint32_t r2 = win.staged32(walked && lit <= kBatchLit ? after : ip0);
    const uint32_t off = r2 & 0xFFFFu;
    after += 2u;
    if (ml == 15u) {
        ml += (r2 >> 16) & 255u;
        after += 1u;
    }
    ml += 4u;
    const uint32_t out_len = walked ? lit + ml : 0u;
    const uint32_t out_end = start + wave_inclusive_sum(out_len);
    const uint32_t my_op = out_end - out_len;
    const uint32_t mo = my_op + lit;
    const uint32_t s = mo - off;
    // This batch's share of the ring (see RingShare).
    const uint32_t reach = Team::kAdaptive ? share.reach : Team::kReach;
    uint32_t span = Team::kSpan;
    if (Team::kAdaptive) {
        span = (kHist - reach) / 2u;   // what two batches of this kind leave each other
        const uint32_t room = kHist - share.prev_reach - share.prev_span;   // beside the batch in the helper's hands
        span = span < room ? span : room;
    }
    const bool in_ring = s >= valid_from && s + reach >= start;
    const bool bad = !walked || lit > kBatchLit || ml > kBatchMatch || (off < ml && off < 4u * kMatchStep) ||
                     off > mo || off == 0u || after + 4u > iend || out_end + 16u > oend ||
                     out_end - start > span || (!in_ring && s + ml > start) ||
                     (off < ml && !in_ring);
    const uint64_t stop = __builtin_amdgcn_ballot_w64(bad);
    const uint32_t take = stop ? static_cast<uint32_t>(__builtin_ctzll(stop)) : kWave;
    walk = take >= walk ? (2u * walk < kWave ? 2u * walk : kWave) : (take + 4u < kWave ? take + 4u : kWave);
    if (take == 0)
        return 0;
    ip_io = take < n ? __builtin_amdgcn_readlane(my_ip, take) : uni(p);
    op_io = __builtin_amdgcn_readlane(out_end, take - 1u);
    desc.lip = lip;
    desc.lit = lit;
    desc.ml = ml;
    desc.off = off;
    desc.out_end = out_end;
    info.take = take;
    info.start = start;
    info.end = op_io;
    info.valid_from = valid_from;
    info.reach = reach;
    info.stage_block = win.stage_block;
    info.literals = 1u;
    if (Team::kAdaptive) {
        // Where this batch's older sources lay decides the next batch's reach: 2 KiB when a good
        // part of them lies between 512 bytes and 2 KiB behind the batch, else 512 bytes (batches
        // of up to 3840 bytes instead of 3072).  That is the choice when nearly all lie within
        // 512 bytes, and also when most lie further back than any reach (offsets anywhere in
        // 64 KiB): the 512 bytes then still keep the ring for a source that begins in front of the
        // batch and ends inside it (no reach at all measured 3 % slower on such streams: such a
        // source ends its batch).
        const bool older = lane < take && s < start;
        const uint32_t n_older = __builtin_popcountll(__builtin_amdgcn_ballot_w64(older));
        const uint32_t n_2k = __builtin_popcountll(__builtin_amdgcn_ballot_w64(older && s + 2048u >= start));
        const uint32_t n_512 = __builtin_popcountll(__builtin_amdgcn_ballot_w64(older && s + 512u >= start));
        share.reach = (4u * n_2k < n_older || 8u * n_512 >= 7u * n_2k) ? 512u : 2048u;
        share.prev_reach = reach;
        share.prev_span = op_io - start;
    }
    PHASE_MARK(clk, 1);
#ifdef SNAPPY_LZ4_PHASE_SWITCHES
    clk.acc[7] += (uint64_t(1) << 32) | take;  // batches taken, sequences in them
#endif
    return take;
}

// Second half: the copies.  First the literals, every lane its own sequence's, from
// the staged input into the ring: the wave that walked does this, because it holds the
// stage (a second wave would have to fetch the same bytes from memory again, a round
// trip per 32 bytes of literals and twice the traffic).
__device__ __forceinline__ void batch_literals(const InputWindow &win, uint8_t *ring, const BatchLane &desc,
                                               const BatchInfo &info, uint32_t lane, uint32_t tuning)
{
    if (tuning & kPhaseSwitches & 2u)
        return;
    const uint32_t my_lit = lane < info.take ? desc.lit : 0u;
    const uint32_t my_op = desc.out_end - (desc.lit + desc.ml);
    // Source positions count bytes from win.aligned, as the stage's dwords do.
    LaneCopy c(my_op, my_lit, static_cast<int32_t>(desc.lip));
    // The last dword is stored whole even when the literals end inside it: the one to three
    // bytes too many lie where the sequence's own match (four bytes at least) is written
    // afterwards, by whoever does the matches, and nobody reads them before that.  So no tail.
    // (The first dword stays exact: the bytes in front of a batch's first literals may be the
    // end of a match the helper is still copying.)
    c.full_end = c.count;
    // Lanes with nothing to copy read the stage's first dwords.
    const uint32_t *src = win.stage + (c.count ? c.sbase - static_cast<int32_t>(win.stage_block * 64u) : 0);
    for (uint32_t k0 = 0; __builtin_amdgcn_ballot_w64(k0 < c.count) != 0; k0 += kLitStep) {
        uint32_t S[kLitStep + 1u];
#pragma unroll
        for (uint32_t i = 0; i <= kLitStep; ++i)
            S[i] = src[k0 + i];
        lane_copy_step<kLitStep>(ring, c, k0, S);
    }
}

// Then the matches, and the batch's output from the ring to memory.  In a pair the
// helper wave does this while the parser walks on, and the parser may already be
// writing the next batch's literals into the ring behind info.end.
// lit_from (may be null): the stage, when the batch's literals are still to be copied
// (info.literals); *lit_done is then set to lit_done_value once they have been read.
template <typename Team>
__device__ __forceinline__ void batch_matches(uint8_t *dst, uint8_t *ring, const BatchLane &desc,
                                              const BatchInfo &info, uint32_t lane, uint32_t tuning,
                                              PhaseClock &clk, const InputWindow *lit_from = nullptr,
                                              uint32_t *lit_done = nullptr, uint32_t lit_done_value = 0)
{
    const uint32_t take = info.take, start = info.start;
    const bool mine = lane < take;
    const uint32_t lit = desc.lit, ml = desc.ml, off = desc.off;
    const uint32_t out_end = desc.out_end;
    const uint32_t my_ml = mine ? ml : 0u;
    const uint32_t my_op = out_end - (lit + ml);
    const uint32_t mo = my_op + lit;
    const uint32_t s = mo - off;
    const bool in_ring = s >= info.valid_from && s + info.reach >= start;

    // Everything a batch produces is written to the ring only and goes to memory in
    // one piece at the end (flush_ring).  Matches whose source is older than the
    // batch and out of the ring's reach read memory, as aligned dwords from the one
    // that holds the byte that goes to the first ring dword's first byte: their loads
    // are issued here, ahead of the dependency search below, which touches LDS alone,
    // so the round trip runs beside that work.  Every lane loads, from a harmless
    // address when it has nothing to copy, so that one wait covers a step's loads.
    const bool far = mine && !in_ring;
    const bool any_far = !(tuning & kPhaseSwitches & 4u) && __builtin_amdgcn_ballot_w64(far) != 0;
    const uint32_t far_len = far ? ml : 0u;
    // The source dwords are numbered from the dword that holds the chunk's first output byte:
    // dwords below it hold nothing of the chunk's output (and may lie outside the allocation), so
    // a number below 0 (only -1 occurs) is read as dword 0.  One base address, the same in every
    // lane, and a 32-bit number per lane: the loads are global ones with a scalar base.
    const uint32_t lead = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(dst)) & 3u;
    const uint32_t *floor32 = reinterpret_cast<const uint32_t *>(dst - lead);
    const int32_t from0 = static_cast<int32_t>(lead + (far ? s : 3u) - (mo & 3u));  // byte offset from floor32, >= -3
    const int32_t from_dw = from0 >> 2;
    LaneCopy fc(mo, far_len, 0);
    fc.m = static_cast<uint32_t>(from0) & 3u;
    // Several steps of source dwords are kept in flight (a step is 40 bytes): one round trip after
    // the other, the steps of long matches (200 bytes: five) took more of the helper's time on
    // such streams than everything else it does.  kFarDepth register sets; the one a copy step
    // has used up is asked for again at once, for the step kFarDepth further on.
    uint32_t ff[kFarDepth][kFarStep + 1u];
    auto far_loads = [&](uint32_t k0, uint32_t (&to)[kFarStep + 1u]) {
#pragma unroll
        for (uint32_t i = 0; i <= kFarStep; ++i) {
            const int32_t at = from_dw + static_cast<int32_t>(fc.wants(k0 + i) ? k0 + i : 0u);
            to[i] = floor32[at < 0 ? 0u : static_cast<uint32_t>(at)];
        }
    };
    // Steps some lane has dwords in (uniform).
    uint32_t far_steps = 0;
    if (any_far)
        while (__builtin_amdgcn_ballot_w64(far_steps * kFarStep < fc.count) != 0)
            ++far_steps;
#pragma unroll
    for (uint32_t d = 0; d < kFarDepth; ++d)
        if (d < far_steps)
            far_loads(d * kFarStep, ff[d]);

    // The literals, when they came over with the batch: behind the far loads, whose round trip
    // they then run beside (they touch LDS alone), and in front of everything that reads the ring.
    if (lit_from != nullptr && info.literals != 0u) {
        batch_literals(*lit_from, ring, desc, info, lane, tuning);
        // The literals' reads have been carried out when this store is (one wave's LDS operations
        // take effect in order): the parser may refill the stage.
        __hip_atomic_store(lit_done, lit_done_value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        PHASE_MARK(clk, 2);
    }

    // Matches that read the ring are copied in rounds: a match is copied once every
    // match whose output it reads has been.  These are the sequences from the one that
    // holds the source's first byte to the last one whose match begins before the
    // source ends (output positions grow with the lane number); a source inside
    // literals alone depends on nothing.  A match that repeats a pattern (offset
    // < length) needs the pattern only; each copy step reads what the steps before
    // it wrote, LDS operations of one wave taking effect in order.
    const uint32_t need_end = s + (off < my_ml ? off : my_ml);
    const uint32_t sorted_end = mine ? out_end : 0xFFFFFFFFu;
    const uint32_t sorted_mo = mine ? mo : 0xFFFFFFFFu;
    uint32_t dep_first = 0, dep_count = 0;  // lanes with out_end <= s; lanes with mo < need_end
#pragma unroll
    for (uint32_t step = 32; step != 0; step >>= 1) {
        const uint32_t e = static_cast<uint32_t>(__builtin_amdgcn_ds_bpermute(
            static_cast<int>((dep_first + step - 1u) << 2), static_cast<int>(sorted_end)));
        const uint32_t m = static_cast<uint32_t>(__builtin_amdgcn_ds_bpermute(
            static_cast<int>((dep_count + step - 1u) << 2), static_cast<int>(sorted_mo)));
        dep_first += (e <= s) ? step : 0u;
        dep_count += (m < need_end) ? step : 0u;
    }
    // Bits dep_first .. dep_count - 1 (both at most 63: the lane's own output lies behind its source).
    const uint64_t below_count = (uint64_t(1) << dep_count) - 1u;
    const uint64_t below_first = (uint64_t(1) << dep_first) - 1u;
    const uint64_t depends_on = below_count & ~below_first;
    PHASE_MARK(clk, 9);

    // Matches are written to the byte: what follows a match is already in place.
    bool waiting = mine;
    if (any_far) {
        // The last dword's source dwords, for its bytes when it is not stored whole: the
        // step that holds them keeps them (a step's dwords are registers, not memory).
        uint32_t t0 = 0, t1 = 0;
        const uint32_t last = fc.count ? fc.count - 1u : 0u;
        // Loads return in order and the compiler counts them: the wait in front of a step is
        // for that step's loads alone, the later sets stay in flight.
        for (uint32_t base = 0; base < far_steps; base += kFarDepth) {
#pragma unroll
            for (uint32_t d = 0; d < kFarDepth; ++d) {
                const uint32_t step = base + d;
                if (step < far_steps) {
                    const uint32_t k0 = step * kFarStep;
#pragma unroll
                    for (uint32_t i = 0; i < kFarStep; ++i) {
                        t0 = (k0 + i == last) ? ff[d][i] : t0;
                        t1 = (k0 + i == last) ? ff[d][i + 1u] : t1;
                    }
                    lane_copy_step<kFarStep>(ring, fc, k0, ff[d]);
                    if (step + kFarDepth < far_steps)
                        far_loads(k0 + kFarDepth * kFarStep, ff[d]);
                }
            }
        }
        lane_copy_tail(ring, fc, t0, t1);
        waiting = waiting && !far;
    }

    PHASE_MARK(clk, 10);
    const uint32_t *ring32 = reinterpret_cast<const uint32_t *>(ring);
    for (;;) {
        const uint64_t left = __builtin_amdgcn_ballot_w64(waiting);
        if (left == 0 || (tuning & kPhaseSwitches & 8u))
            break;
        const bool go = waiting && (depends_on & left) == 0;
        const LaneCopy c(mo, go ? my_ml : 0u, static_cast<int32_t>(s));
        for (uint32_t k0 = 0; __builtin_amdgcn_ballot_w64(k0 < c.count) != 0; k0 += kMatchStep) {
            // Source dwords through one base address and constant offsets unless some lane's
            // wrap around the ring's end in this step.
            uint32_t S[kMatchStep + 1u];
            const uint32_t first = static_cast<uint32_t>(c.sbase + static_cast<int32_t>(k0)) & (kHist / 4u - 1u);
            if (__builtin_amdgcn_ballot_w64(first + kMatchStep + 1u > kHist / 4u) == 0) {
#pragma unroll
                for (uint32_t i = 0; i <= kMatchStep; ++i)
                    S[i] = ring32[first + i];
            } else {
#pragma unroll
                for (uint32_t i = 0; i <= kMatchStep; ++i)
                    S[i] = ring32[(first + i) & (kHist / 4u - 1u)];
            }
            lane_copy_step<kMatchStep>(ring, c, k0, S);
        }
        {
            const uint32_t last = static_cast<uint32_t>(c.sbase + static_cast<int32_t>(c.count ? c.count - 1u : 0u));
            lane_copy_tail(ring, c, ring32[last & (kHist / 4u - 1u)], ring32[(last + 1u) & (kHist / 4u - 1u)]);
        }
        waiting = waiting && !go;
    }
    PHASE_MARK(clk, 4);
    flush_ring(dst, ring, start, info.end, lane);
    PHASE_MARK(clk, 11);
}

// Who does a chunk's copies.  Solo: the wave that parses, as it goes.  Pair: a
// second wave of the workgroup takes the batches' copies (so that they run beside
// the walk to the next batch) and half of every long copy.  The parsing wave
// publishes commands in LDS; the helper sits at a barrier between commands, so one
// barrier hands a command over and, at the same time, tells the parser that the
// command before it is finished.
constexpr uint32_t kCmdExit = 0, kCmdSync = 1, kCmdBatch = 2, kCmdCopy = 3;
constexpr uint32_t kTeamCopyMin = 4096;  // shorter copies are not worth two barriers

struct TeamShared {
    // Two of everything: command k uses slot k & 1, and the parser cannot get two
    // commands ahead of the helper (every command is a barrier both must reach).
    uint32_t cmd[2];
    uint32_t len[2];
    uint8_t *dst[2];
    const uint8_t *src[2];
    BatchInfo info[2];
    uint32_t lanes[2][5][kWave];
    // Written by the helper: commands it has finished (written just before it waits for the
    // next one), and batches whose literals it has finished reading out of the stage.
    uint32_t helper_done;
    uint32_t stage_read;
};

struct SoloTeam {
    static constexpr uint32_t kSpan = kSoloSpan, kReach = kSoloReach;
    static constexpr bool kAdaptive = false;
    __device__ __forceinline__ void sync() {}
    __device__ __forceinline__ void before_restage() {}
    __device__ __forceinline__ void copy(uint8_t *dst, const uint8_t *src, uint32_t n, uint32_t lane)
    {
        copy_run(dst, src, n, lane);
    }
    __device__ __forceinline__ void batch(const InputWindow &win, const uint8_t *, uint8_t *dst, uint8_t *ring,
                                          const BatchLane &desc, const BatchInfo &info, uint32_t lane,
                                          uint32_t tuning, PhaseClock &clk)
    {
        batch_literals(win, ring, desc, info, lane, tuning);
        batch_matches<SoloTeam>(dst, ring, desc, info, lane, tuning, clk);
    }
};

struct PairTeam {
    static constexpr uint32_t kCopyDepth = 4;
    static constexpr uint32_t kSpan = kPairSpan, kReach = kPairReach;   // the shares a chunk starts with
    static constexpr bool kAdaptive = true;
    TeamShared *sh;
    uint32_t seq;   // commands published so far
    bool pending;   // the helper may still be busy with a batch
    uint32_t stage_lent = 0;   // != 0: the helper reads the stage until sh->stage_read reaches this

    // The helper copies a batch's literals out of the stage when it is idle at the hand-over:
    // the stage must then stay as it is until it has.
    __device__ __forceinline__ void before_restage()
    {
        if (stage_lent == 0u)
            return;
        while (uni(__hip_atomic_load(&sh->stage_read, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) < stage_lent)
            __builtin_amdgcn_s_sleep(1);
        stage_lent = 0u;
    }

    // Every lane stores the same words.  The parser's own stores to memory are
    // complete before the helper may read what they wrote.
    __device__ __forceinline__ void publish(uint32_t cmd, uint8_t *dst, const uint8_t *src, uint32_t n)
    {
        const uint32_t slot = seq & 1u;
        sh->cmd[slot] = cmd;
        sh->len[slot] = n;
        sh->dst[slot] = dst;
        sh->src[slot] = src;
        wait_for_memory();
        __syncthreads();
        ++seq;
        pending = cmd == kCmdBatch;
    }
    __device__ __forceinline__ void sync()
    {
        if (pending)
            publish(kCmdSync, nullptr, nullptr, 0);
        stage_lent = 0u;   // the helper has finished everything
    }
    __device__ __forceinline__ void copy(uint8_t *dst, const uint8_t *src, uint32_t n, uint32_t lane)
    {
        if (n < kTeamCopyMin) {
            sync();
            copy_run<kCopyDepth>(dst, src, n, lane);
            return;
        }
        publish(kCmdCopy, dst, src, n);  // the helper takes [half, n)
        const uint32_t half = (n >> 1) & ~15u;
        copy_run<kCopyDepth>(dst, src, half, lane);
        wait_for_memory();
        __syncthreads();
    }
    __device__ __forceinline__ void batch(const InputWindow &win, const uint8_t *src_al, uint8_t *dst, uint8_t *ring,
                                          const BatchLane &desc, const BatchInfo &info, uint32_t lane,
                                          uint32_t tuning, PhaseClock &clk)
    {
        // Who copies the literals is decided at the hand-over: the helper when it sits idle, waiting
        // for this command (the walk is the chunk's critical path, and on streams of short sequences
        // the helper has time to spare); the parser when the helper is still busy with the batch
        // before, so that the copy runs beside that work.  The parser's copy goes into the ring
        // behind the output of the batch the helper may still be copying (RingShare keeps the two
        // apart); the barrier in publish() comes after these LDS writes have completed.
        const bool helper_idle =
            uni(__hip_atomic_load(&sh->helper_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) == seq;
        BatchInfo handed = info;
        if (helper_idle && !(tuning & 64u)) {
            stage_lent = seq + 1u;
        } else {
            batch_literals(win, ring, desc, info, lane, tuning);
            handed.literals = 0u;
        }
        PHASE_MARK(clk, 2);
        const uint32_t slot = seq & 1u;
        sh->lanes[slot][0][lane] = desc.lip;
        sh->lanes[slot][1][lane] = desc.lit;
        sh->lanes[slot][2][lane] = desc.ml;
        sh->lanes[slot][3][lane] = desc.off;
        sh->lanes[slot][4][lane] = desc.out_end;
        sh->info[slot] = handed;
        publish(kCmdBatch, dst, src_al, 0);
    }
};

// The helper wave of a pair: carries out the parser's commands until told to leave.
__device__ void helper_loop(TeamShared *sh, uint8_t *ring, uint32_t *stage, uint32_t lane, uint32_t tuning)
{
    PhaseClock clk;
    clk.start();
    for (uint32_t seq = 0;; ++seq) {
        __hip_atomic_store(&sh->helper_done, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __syncthreads();
        const uint32_t slot = seq & 1u;
        const uint32_t cmd = uni(sh->cmd[slot]);
        PHASE_MARK(clk, 8);  // measurement builds: the helper's time at the barrier
        if (cmd == kCmdExit) {
            clk.flush(lane);
            return;
        }
        if (cmd == kCmdBatch) {
            BatchLane desc;
            desc.lip = sh->lanes[slot][0][lane];
            desc.lit = sh->lanes[slot][1][lane];
            desc.ml = sh->lanes[slot][2][lane];
            desc.off = sh->lanes[slot][3][lane];
            desc.out_end = sh->lanes[slot][4][lane];
            BatchInfo info;
            info.take = uni(sh->info[slot].take);
            info.start = uni(sh->info[slot].start);
            info.end = uni(sh->info[slot].end);
            info.valid_from = uni(sh->info[slot].valid_from);
            info.reach = uni(sh->info[slot].reach);
            info.literals = uni(sh->info[slot].literals);
            InputWindow from;
            from.stage = stage;
            from.stage_block = uni(sh->info[slot].stage_block);
            batch_matches<PairTeam>(known_global(sh->dst[slot]), ring, desc, info, lane, tuning, clk, &from,
                                    &sh->stage_read, seq + 1u);
            wait_for_memory();  // the flush is in memory before the parser is told so
        } else if (cmd == kCmdCopy) {
            const uint32_t n = uni(sh->len[slot]);
            const uint32_t half = (n >> 1) & ~15u;
            copy_run<PairTeam::kCopyDepth>(known_global(sh->dst[slot]) + half, known_global(sh->src[slot]) + half, n - half,
                                           lane);
            wait_for_memory();
            __syncthreads();
        }
    }
}

// Decodes one compressed chunk; returns true when the block is well-formed.
template <typename Team>
__device__ bool decode_sequences(const uint8_t *__restrict__ in_base, uint64_t in_alloc,
                                 const Lz4Chunk &ck, uint8_t *out_base, uint8_t *ring, uint32_t *stage,
                                 uint32_t lane, uint32_t tuning, PhaseClock &clk, Team &team)
{
    const uint8_t *src = in_base + ck.in_off;
    uint8_t *dst = out_base + ck.out_off;
    const uint32_t in_shift = static_cast<uint32_t>(ck.in_off & 3u);
    const uint32_t iend = ck.comp_len + in_shift;  // positions are counted from the aligned base
    const uint32_t oend = ck.orig_len;

    InputWindow win;
    win.aligned = reinterpret_cast<const uint32_t *>(src - in_shift);
    const uint64_t avail = in_alloc - (ck.in_off - in_shift);
    win.limit_dwords = static_cast<uint32_t>(avail >> 2 > 0xFFFFFFFFull ? 0xFFFFFFFFull : avail >> 2);
    win.stage = stage;
    win.reset(lane);

    uint32_t ip = in_shift;  // uniform
    uint32_t op = 0;         // uniform
    uint32_t valid_from = 0; // uniform: the ring holds output positions [valid_from, op)
    uint32_t walk = kWave;   // uniform: tokens the next batch looks at
    uint32_t misses = 0, skip = 0;  // uniform: back-off of the batch attempts on streams of long sequences
    RingShare share = {Team::kReach, 0u, 0u};
    const uint8_t *src_al = src - in_shift;

    // A chunk may be asked for in part (its first bytes hold an archive's header): the
    // decode then ends, well-formed so far, once that much output exists.
    const uint32_t stop_at = ck.prefix_len ? ck.prefix_len : 0xFFFFFFFFu;
    for (;;) {
        if (op >= stop_at)
            return true;
        if (skip != 0u) {
            --skip;
        } else if (!(tuning & 1u)) {
            BatchLane desc;
            BatchInfo info;
            const uint32_t taken = produce_batch<Team>(win, ip, op, valid_from, iend, oend, walk, lane, clk, desc, info,
                                                       share, team);
            if (taken != 0u)
                team.batch(win, src_al, dst, ring, desc, info, lane, tuning, clk);
            if (taken >= 16u)
                continue;  // more of the same is likely
            // Attempts that take nothing back off: 1, 3, 7, 15 sequences go by before the next one.
            misses = taken != 0u ? 0u : (2u * misses + 1u < 15u ? 2u * misses + 1u : 15u);
            skip = misses;
        }
        // What follows writes the ring and memory itself: the batches before it are finished first.
        team.sync();
        share.prev_reach = share.prev_span = 0;
        if (ip >= iend)
            return false;
        uint32_t bytes = uni(win.peek32(ip, lane));
        const uint32_t token = bytes & 0xFFu;
        ip += 1;

        // Literal length.
        uint32_t lit = token >> 4;
        if (lit == 15u) {
            for (;;) {
                if (ip + 4u <= iend) {
                    bytes = uni(win.peek32(ip, lane));
                    if (bytes == 0xFFFFFFFFu) {
                        lit += 4u * 255u;
                        ip += 4;
                        if (lit > oend)
                            return false;
                        continue;
                    }
                } else if (ip >= iend) {
                    return false;
                } else {
                    bytes = uni(win.peek32(ip, lane));
                }
                const uint32_t b = bytes & 0xFFu;
                lit += b;
                ip += 1;
                if (b != 255u)
                    break;
            }
        }
        if (lit > oend - op || lit > iend - ip)
            return false;
        const bool last = (op + lit + kCopyLength > oend);
        if (last && op + lit != oend)
            return false;
        if (lit != 0u && lit <= kShortLit) {
            win.cover(ip, lit, lane);
            emit_short_literals(win, ip, lit, dst, ring, op, lane);
        } else if (lit != 0u) {
            team.copy(dst + op, src_al + ip, lit, lane);
            valid_from = op + lit;
        }
        ip += lit;
        op += lit;
        if (last)
            return ip == iend;

        // Match offset.
        if (ip + 2u > iend)
            return false;
        bytes = uni(win.peek32(ip, lane));
        const uint32_t offset = bytes & 0xFFFFu;
        ip += 2;
        if (offset == 0u || offset > op)
            return false;

        // Match length.
        uint32_t ml = token & 15u;
        if (ml == 15u) {
            for (;;) {
                if (ip + 4u <= iend) {
                    bytes = uni(win.peek32(ip, lane));
                    if (bytes == 0xFFFFFFFFu) {
                        ml += 4u * 255u;
                        ip += 4;
                        if (ml > oend)
                            return false;
                        continue;
                    }
                } else if (ip >= iend) {
                    return false;
                } else {
                    bytes = uni(win.peek32(ip, lane));
                }
                const uint32_t b = bytes & 0xFFu;
                ml += b;
                ip += 1;
                if (b != 255u)
                    break;
            }
        }
        ml += 4u;
        // The last five bytes of a block are literals: a match that reaches into them is
        // refused (LZ4's LASTLITERALS rule, in the r7x decoder lz4net ports as in today's).
        if (ml + 5u > oend - op)
            return false;

        if (ml <= kShortMatch && offset <= kHist - kShortMatch && op - offset >= valid_from) {
            emit_short_match(dst, ring, op, offset, ml, lane);
        } else {
            uint8_t *mdst = dst + op;
            const uint8_t *msrc = mdst - offset;
            if (offset >= ml) {
                team.copy(mdst, msrc, ml, lane);
            } else {
                const bool in_ring = op - offset >= valid_from && offset <= kHist;
                copy_periodic(team, mdst, offset, ml, lane, in_ring ? ring : nullptr, op - offset);
            }
            valid_from = op + ml;
        }
        op += ml;
    }
}

// The same, and every copy handed out on the way is finished when it returns.
template <typename Team>
__device__ bool decode_block(const uint8_t *__restrict__ in_base, uint64_t in_alloc, const Lz4Chunk &ck,
                             uint8_t *out_base, uint8_t *ring, uint32_t *stage, uint32_t lane, uint32_t tuning,
                             PhaseClock &clk, Team &team)
{
    const bool ok = decode_sequences(in_base, in_alloc, ck, out_base, ring, stage, lane, tuning, clk, team);
    team.sync();
    return ok;
}

__global__ void __launch_bounds__(kWave)
lz4_decode_kernel(const uint8_t *__restrict__ in_base, uint64_t in_alloc, uint8_t *out_base,
                  const Lz4Chunk *__restrict__ chunks, uint32_t n_chunks,
                  uint32_t *__restrict__ queue, uint32_t *__restrict__ status, uint32_t tuning)
{
    __shared__ __attribute__((aligned(16))) uint8_t ring[kHist];
    const uint32_t lane = threadIdx.x & (kWave - 1);
    __shared__ __attribute__((aligned(16))) uint32_t stage[kStageBlocks * 64u + kWave];
    // In an MCDF import the decoder runs next to a SHA-1 kernel whose waves issue an
    // instruction in nearly every slot of their SIMD.  A decoder wave spends most of its
    // time waiting for memory and LDS; when it is ready it should not queue behind that.
    __builtin_amdgcn_s_setprio(3);
    SoloTeam team;
    for (;;) {
        const uint32_t i = wave_claim(queue, 1u, lane);
        if (i >= n_chunks)
            break;
        const Lz4Chunk ck = chunks[i];
        bool ok = true;
        if (ck.compressed) {
            PhaseClock clk;
            clk.start();
            ok = decode_block(in_base, in_alloc, ck, out_base, ring, stage, lane, tuning, clk, team);
            PHASE_MARK(clk, 5);
            clk.flush(lane);
        } else {
            copy_run(out_base + ck.out_off, in_base + ck.in_off, ck.orig_len, lane);
        }
        // Every lane stores the same word: no lane-0-only tail (see wave_claim).
        status[i] = ok ? 0u : 1u;
    }
}

// Two waves per chunk: wave 0 parses, wave 1 copies (PairTeam above).  Both leave
// through one door: the helper returns on the parser's last command, which the
// parser sends when the queue is empty, whatever happened to its chunks.
__global__ void __launch_bounds__(2 * kWave)
lz4_decode_pair_kernel(const uint8_t *__restrict__ in_base, uint64_t in_alloc, uint8_t *out_base,
                       const Lz4Chunk *__restrict__ chunks, uint32_t n_chunks,
                       uint32_t *__restrict__ queue, uint32_t *__restrict__ status, uint32_t tuning)
{
    __shared__ __attribute__((aligned(16))) uint8_t ring[kHist];
    __shared__ __attribute__((aligned(16))) uint32_t stage[kStageBlocks * 64u + kWave];
    __shared__ TeamShared shared;
    const uint32_t lane = threadIdx.x & (kWave - 1);
    // The parser's chain of dependent steps is what a chunk's time hangs on: when its
    // next instruction is ready it goes before the helper's (and both before a SHA-1
    // kernel's waves in an import).
    if (threadIdx.x >= kWave) {
        __builtin_amdgcn_s_setprio(2);
        helper_loop(&shared, ring, stage, lane, tuning);
        return;
    }
    __builtin_amdgcn_s_setprio(3);
    shared.stage_read = 0;   // before the first command (the helper only ever raises it)
    PairTeam team{&shared, 0u, false};
    for (;;) {
        const uint32_t i = wave_claim(queue, 1u, lane);
        if (i >= n_chunks)
            break;
        const Lz4Chunk ck = chunks[i];
        bool ok = true;
        if (ck.compressed) {
            PhaseClock clk;
            clk.start();
            ok = decode_block(in_base, in_alloc, ck, out_base, ring, stage, lane, tuning, clk, team);
            PHASE_MARK(clk, 5);
            clk.flush(lane);
        } else {
            team.copy(out_base + ck.out_off, in_base + ck.in_off, ck.orig_len, lane);
        }
        status[i] = ok ? 0u : 1u;
    }
    team.publish(kCmdExit, nullptr, nullptr, 0);
}

}  // namespace

hipError_t launch_lz4_decode(const uint8_t *d_in, uint64_t in_alloc, uint8_t *d_out,
                             const Lz4Chunk *d_chunks, uint32_t n_chunks, uint32_t *d_queue,
                             uint32_t *d_status, int n_cu, hipStream_t stream)
{
    if (n_chunks == 0 || n_cu <= 0)
        return hipSuccess;
    // One wave per workgroup, one chunk per wave at a time.  The dispatcher fills a
    // CU with as many workgroups as fit before it moves on, so a launch of fewer
    // waves than the chip holds would crowd some CUs and leave others idle (1024
    // chunks: 11 % slower).  Each workgroup therefore asks for so much LDS that
    // only its share of a CU's 160 KiB fits: the waves spread evenly.
    constexpr uint32_t kWavesPerCu = 8, kLdsPerCu = 156u << 10, kLdsUsed = kHist + kStageBlocks * 256u;
    const uint32_t blocks = static_cast<uint32_t>(
        std::min<uint64_t>(n_chunks, static_cast<uint64_t>(n_cu) * kWavesPerCu));
    const uint32_t per_cu = (blocks + static_cast<uint32_t>(n_cu) - 1u) / static_cast<uint32_t>(n_cu);
    const uint32_t share = std::min<uint32_t>(kLdsPerCu / per_cu, 96u << 10) & ~1023u;
    uint32_t pad_lds = share > kLdsUsed ? share - kLdsUsed : 0u;
    // More than 64 KiB per workgroup has to be asked for, on every device the kernel
    // runs on; where that is refused the launch is merely spread less well.
    if (kLdsUsed + pad_lds > (64u << 10) &&
        hipFuncSetAttribute(reinterpret_cast<const void *>(lz4_decode_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 96 << 10) != hipSuccess) {
        (void)hipGetLastError();
        pad_lds = (64u << 10) - kLdsUsed;
    }
    // SNAPPY_LZ4_TUNING, bit 0: every sequence takes the one-at-a-time path (for comparisons);
    // bit 5 (32): one wave per chunk instead of two; bit 6 (64): the parsing wave copies every
    // batch's literals itself (the helper takes none).
    static const uint32_t tuning = [] {
        const char *env = std::getenv("SNAPPY_LZ4_TUNING");
        return env ? static_cast<uint32_t>(std::strtoul(env, nullptr, 0)) : 0u;
    }();
#ifdef SNAPPY_LZ4_PHASE_SWITCHES
    unsigned long long zero[12] = {};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_phase_cycles), zero, sizeof zero);
#endif
    if (!(tuning & 32u)) {
        constexpr uint32_t kPairLds = kLdsUsed + static_cast<uint32_t>(sizeof(TeamShared));
        uint32_t pad = share > kPairLds ? share - kPairLds : 0u;
        if (kPairLds + pad > (64u << 10) &&
            hipFuncSetAttribute(reinterpret_cast<const void *>(lz4_decode_pair_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 96 << 10) != hipSuccess) {
            (void)hipGetLastError();
            pad = (64u << 10) - kPairLds;
        }
        hipLaunchKernelGGL(lz4_decode_pair_kernel, dim3(blocks), dim3(2 * kWave), pad, stream, d_in, in_alloc, d_out,
                           d_chunks, n_chunks, d_queue, d_status, tuning);
    } else {
        hipLaunchKernelGGL(lz4_decode_kernel, dim3(blocks), dim3(kWave), pad_lds, stream, d_in, in_alloc, d_out,
                           d_chunks, n_chunks, d_queue, d_status, tuning);
    }
#ifdef SNAPPY_LZ4_PHASE_SWITCHES
    (void)hipStreamSynchronize(stream);
    unsigned long long c[12];
    (void)hipMemcpyFromSymbol(c, HIP_SYMBOL(g_phase_cycles), sizeof c);
    fprintf(stderr, "lz4 phases (s_memtime ticks summed over %u chunks): parser: walk %llu classify %llu literals %llu "
                    "between batches and one-at-a-time %llu staging %llu; helper: waiting %llu search and far loads issued %llu "
                    "far sources %llu rounds %llu flush %llu; %llu batches of %.1f sequences\n", n_chunks, c[0], c[1], c[2], c[5], c[6], c[8], c[9], c[10], c[4], c[11],
            c[7] >> 32, static_cast<double>(c[7] & 0xFFFFFFFFull) / static_cast<double>((c[7] >> 32) ? (c[7] >> 32) : 1));
#endif
    return hipGetLastError();
}

}  // namespace snappy
