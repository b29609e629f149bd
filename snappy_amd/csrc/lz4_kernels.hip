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
// ends within 8 bytes of the end must end the block exactly; a match must end
// before the end of the block (not at it); consumed input must equal the chunk's
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

__device__ __forceinline__ uint32_t uni(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }

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
constexpr uint32_t kStageBlocks = 24;  // 6 KiB per wave
constexpr uint32_t kJumpBlocks = 4;    // staged after a jump (a long literal run was skipped)

struct InputWindow {
    const uint32_t *aligned;  // dword-aligned pointer at or below the chunk's first byte
    uint32_t limit_dwords;    // dwords that may be read from `aligned`
    uint32_t *stage;          // LDS, kStageBlocks * 64 dwords
    uint32_t stage_block;     // uniform: number of the first staged block
    uint32_t stage_count;     // uniform: blocks staged
    uint32_t first;           // uniform: number of the dword held by lane 0
    uint32_t a;               // per lane: dword first + lane
    uint32_t lens;            // per lane: see token_lengths(); belongs to the window at lens_first
    uint32_t lens_first;      // uniform
    uint32_t lens_limit;      // uniform: the limit the table was made for

    // Every wait is placed here, on the paths that move the window.  A wait left
    // to the compiler lands where the paths join again, is executed by every
    // token, and waits for whatever stores are in flight by then as well (loads
    // and stores share one in-order counter).
    template <uint32_t kBlocks>
    __device__ __forceinline__ void load_blocks(uint32_t blk, uint32_t slot, uint32_t lane)
    {
        uint32_t w[kBlocks];
#pragma unroll
        for (uint32_t q = 0; q < kBlocks; ++q) {
            const uint32_t idx = (blk + q) * 64u + lane;
            w[q] = aligned[idx < limit_dwords ? idx : 0u];
            w[q] = idx < limit_dwords ? w[q] : 0u;
        }
        wait_for_memory();
#pragma unroll
        for (uint32_t q = 0; q < kBlocks; ++q)
            stage[(slot + q) * 64u + lane] = w[q];
    }

    // Stages the blocks from `blk` on: all of them when the parser ran off the end
    // of what was staged (tokens are dense here), a few after a jump.
    __device__ __forceinline__ void restage(uint32_t blk, bool all, uint32_t lane)
    {
        load_blocks<kJumpBlocks>(blk, 0, lane);
        stage_count = kJumpBlocks;
        if (all) {
            load_blocks<(kStageBlocks - kJumpBlocks) / 2u>(blk + kJumpBlocks, kJumpBlocks, lane);
            load_blocks<(kStageBlocks - kJumpBlocks) / 2u>(blk + (kStageBlocks + kJumpBlocks) / 2u,
                                                           (kStageBlocks + kJumpBlocks) / 2u, lane);
            stage_count = kStageBlocks;
        }
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
        lens_first = 0xFFFFFFFFu;
    }

    // For each of the lane's four byte positions: how far the next token is if a
    // token starts there and its sequence is one the batches take (a literal run
    // with at most one length-extension byte, a match with at most one), one
    // byte per position; 0 where that cannot be said (a longer extension, or a
    // distance that does not fit a byte) and for positions past `limit`, the last
    // one a batch's sequence may start at.  It takes the token and the byte behind
    // it, nothing else, so all 256 positions of the window are done side by side
    // and the serial walk from token to token is left with one v_readlane and a
    // few scalar operations per token.  Lane 63's last position looks at a byte
    // outside the window; the walk moves the window before it gets there.
    __device__ __forceinline__ void token_lengths(uint32_t limit, uint32_t lane)
    {
        // The dword of the lane above (wave_shl:1; lane 63 keeps 0).
        const uint32_t next =
            static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(a), 0x130, 0xF, 0xF, false));
        const uint32_t p0 = (first + lane) * 4u;
        uint32_t packed = 0;
#pragma unroll
        for (uint32_t k = 0; k < 4u; ++k) {
            const uint32_t t = (a >> (8u * k)) & 255u;
            const uint32_t nb = (k < 3u ? a >> (8u * k + 8u) : next) & 255u;
            const uint32_t lit = t >> 4;
            uint32_t len = 3u + lit + ((t & 15u) == 15u ? 1u : 0u);
            len += lit == 15u ? 1u + nb : 0u;
            len = (lit == 15u && nb >= 236u) || p0 + k > limit ? 0u : len;  // 3 + 15 + 1 + 1 + 236 = 256
            packed |= len << (8u * k);
        }
        lens = packed;
        lens_first = first;
        lens_limit = limit;
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
    __device__ __forceinline__ uint32_t staged32(uint32_t pos) const
    {
        uint32_t w;
        __builtin_memcpy(&w, reinterpret_cast<const uint8_t *>(stage) + (pos - stage_block * 256u), 4);
        return w;
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

template <typename Team>
__device__ __forceinline__ void copy_periodic(Team &team, uint8_t *dst, uint32_t period, uint32_t n,
                                              uint32_t lane)
{
    const uint8_t *pat = dst - period;
    const uint32_t reps = (kPeriodSpan + period - 1u) / period;
    const uint32_t span = reps * period;
    const uint32_t head = n < span ? n : span;
    if (period >= 4u * kWave) {
        // Long pattern: each repetition is a plain copy of the pattern itself.
        for (uint32_t at = 0; at < head; at += period)
            team.copy(dst + at, pat, (head - at < period) ? head - at : period, lane);
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
constexpr uint32_t kShortLit = 192;    // longest literal run served from the window registers
constexpr uint32_t kShortMatch = 512;  // longest match served from the ring

__device__ __forceinline__ uint32_t load32(const uint8_t *p)
{
    uint32_t w;
    __builtin_memcpy(&w, p, 4);  // a dword access at any byte address
    return w;
}

__device__ __forceinline__ void store32(uint8_t *p, uint32_t w) { __builtin_memcpy(p, &w, 4); }

// Four ring bytes from output position `pos` on, and the same as a write.
__device__ __forceinline__ uint32_t ring_read32(const uint8_t *ring, uint32_t pos)
{
    const uint32_t i = pos & (kHist - 1u);
    if (i <= kHist - 4u)
        return load32(ring + i);
    uint32_t w = 0;
    for (uint32_t q = 0; q < 4u; ++q)
        w |= static_cast<uint32_t>(ring[(i + q) & (kHist - 1u)]) << (8u * q);
    return w;
}

__device__ __forceinline__ void ring_write32(uint8_t *ring, uint32_t pos, uint32_t w)
{
    const uint32_t i = pos & (kHist - 1u);
    if (i <= kHist - 4u) {
        store32(ring + i, w);
    } else {
        for (uint32_t q = 0; q < 4u; ++q)
            ring[(i + q) & (kHist - 1u)] = static_cast<uint8_t>(w >> (8u * q));
    }
}

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
__device__ unsigned long long g_phase_cycles[9];
struct PhaseClock {
    uint64_t last;
    uint64_t acc[9];
    __device__ void start() { last = __builtin_amdgcn_s_memtime(); for (int i = 0; i < 9; ++i) acc[i] = 0; }
    __device__ void mark(int phase) { const uint64_t now = __builtin_amdgcn_s_memtime(); acc[phase] += now - last; last = now; }
    __device__ void flush(uint32_t lane) { if (lane == 0) for (int i = 0; i < 9; ++i) atomicAdd(&g_phase_cycles[i], acc[i]); }
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
constexpr uint32_t kBatchSpan = 6144;   // most output one batch produces; the ring keeps kHist - kBatchSpan bytes from before it
constexpr uint32_t kBatchInput = 512;   // staged bytes a sequence may touch from its token on
constexpr uint32_t kLitStep = 8;        // dwords a lane has in flight per step when copying literals
constexpr uint32_t kMatchStep = 4;      // and matches: a match may repeat a pattern of 16 bytes or more

// The first `count` (at most 4 count) bytes of `w` go to output position `pos` of
// the ring: a batch's output goes to memory in one piece at its end.  A match's
// last, partial dword is merged into the four bytes that are there and written
// back whole (one read, one write, instead of a loop over bytes that the whole
// wave walks through whenever one lane has such a tail, which is always).  The
// bytes behind the match that are written back unchanged belong to the next
// sequence: its literals, which are final by now, or the first dword of its match,
// which that lane writes whole, before this read or after this write (the two
// kinds of store are different instructions of the same wave).
__device__ __forceinline__ void put_ring(uint8_t *ring, uint32_t pos, uint32_t w, uint32_t count)
{
    if (count >= 4u) {
        ring_write32(ring, pos, w);
    } else if (count != 0u) {
        const uint32_t mine = 0xFFFFFFFFu >> (8u * (4u - count));
        const uint32_t old = ring_read32(ring, pos);
        ring_write32(ring, pos, (w & mine) | (old & ~mine));
    }
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
};

// First half.  Returns the number of sequences taken (0: none, nothing changed but
// `walk`).  `walk` is how many tokens the next call should look at: a stream of
// sequences that are not for this code should not pay for 64 tokens walked in vain
// each time.
__device__ __forceinline__ uint32_t produce_batch(InputWindow &win, uint32_t &ip_io, uint32_t &op_io,
                                                  uint32_t valid_from, uint32_t iend, uint32_t oend,
                                                  uint32_t &walk, uint32_t lane, PhaseClock &clk,
                                                  BatchLane &desc, BatchInfo &info)
{
    PHASE_MARK(clk, 5);
    uint32_t ip = ip_io;  // uniform
    const uint32_t start = op_io;
    const bool likely = walk >= 16u;  // recent batches were worth their while
    const uint32_t ahead = likely ? 3u * kBatchInput : 0u;
    if (ip < win.stage_block * 256u || ip + kBatchInput + ahead > win.stage_end()) {
        // Not enough staged for a batch.  Staging more pays when batches have been
        // taken lately; otherwise the one-at-a-time code stages what it needs.
        if (!likely || ip + kBatchInput > iend)
            return 0;
        win.restage(ip >> 8, true, lane);
        win.slide(ip, lane);
    }
    const uint32_t stage_end = win.stage_end();

    PHASE_MARK(clk, 6);
    // Where the tokens start.
    uint32_t n = 0;
    uint32_t my_ip = 0;  // lane j: sequence j
    // Only what the next token's position depends on is looked at here: the token
    // and the literal run's extension byte (token_lengths above).  A match length
    // with more than one extension byte makes the positions from there on
    // meaningless; the lanes find that out below and the batch ends in front of
    // such a sequence.  The loop is the decoder's serial part: one v_readlane and
    // some fifteen scalar instructions per token, four tokens per turn.  The table
    // says 0 for positions past ip_limit, so the only tests per token are "still
    // inside the window" and "not 0".  (The same walk in vector registers, with
    // ds_bpermute for the table look-up, measured 4 % slower.)
    const uint32_t ip_limit = stage_end - kBatchInput;  // at least kJumpBlocks blocks are staged
    const uint32_t quota = (walk + 3u) & ~3u;           // whole turns; at most 64
    bool more = ip <= ip_limit;
    if (more && (win.lens_first != win.first || win.lens_limit != ip_limit))
        win.token_lengths(ip_limit, lane);
    while (more) {
        // A length of 0 (a sequence for the one-at-a-time code starts here, or the staged
        // input ends) stops the walk where it is: the steps left in the turn find the
        // same 0 at the same place and change nothing, so one test per turn will do.
        uint32_t len = 0;
#pragma unroll
        for (uint32_t u = 0; u < 4u; ++u) {
            uint32_t di = (ip >> 2) - win.first;
            if (__builtin_expect(di >= 63u, 0)) {  // also catches positions before the window
                win.slide(ip, lane);
                win.token_lengths(ip_limit, lane);
                di = 0;
            }
            len = (__builtin_amdgcn_readlane(win.lens, di) >> ((ip & 3u) * 8u)) & 255u;
            my_ip = (lane == n) ? ip : my_ip;
            n += len != 0u ? 1u : 0u;
            ip += len;
        }
        more = len != 0u && n < quota;
    }
    PHASE_MARK(clk, 0);
    if (n == 0) {
        walk = 4;
        return 0;
    }

    // What each sequence is.
    const bool walked = lane < n;
    const uint32_t at = walked ? my_ip : ip_io;
    const uint32_t r1 = win.staged32(at);
    uint32_t lit = (r1 >> 4) & 15u;
    uint32_t ml = r1 & 15u;
    uint32_t lip = at + 1u;
    if (lit == 15u) {
        lit += (r1 >> 8) & 255u;
        lip += 1u;
    }
    uint32_t after = lip + lit;  // the offset; then the next token
    const uint32_t r2 = win.staged32(after);
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
    const bool in_ring = s >= valid_from && s + kHist >= start + kBatchSpan;
    const bool bad = !walked || lit > kBatchLit || ml > kBatchMatch || (off < ml && off < 4u * kMatchStep) ||
                     off > mo || off == 0u || after + 4u > iend || out_end + 16u > oend ||
                     out_end - start > kBatchSpan || (!in_ring && s + ml > start) ||
                     (off < ml && !in_ring);
    const uint64_t stop = __builtin_amdgcn_ballot_w64(bad);
    const uint32_t take = stop ? static_cast<uint32_t>(__builtin_ctzll(stop)) : kWave;
    walk = take >= walk ? (2u * walk < kWave ? 2u * walk : kWave) : (take + 4u < kWave ? take + 4u : kWave);
    if (take == 0)
        return 0;
    ip_io = take < n ? __builtin_amdgcn_readlane(my_ip, take) : ip;
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
    PHASE_MARK(clk, 1);
#ifdef SNAPPY_LZ4_PHASE_SWITCHES
    clk.acc[7] += (uint64_t(1) << 32) | take;  // batches taken, sequences in them
#endif
    return take;
}

// Second half: the copies.  The literals come from the staged input (kFromStage:
// the wave that walked is the one that copies) or from the chunk's input in
// memory at src_al (another wave copies while the first walks on and may stage
// something else).
template <bool kFromStage>
__device__ __forceinline__ void consume_batch(const InputWindow *win, const uint8_t *src_al, uint8_t *dst,
                                              uint8_t *ring, const BatchLane &desc, const BatchInfo &info,
                                              uint32_t lane, uint32_t tuning, PhaseClock &clk)
{
    const uint32_t take = info.take, start = info.start;
    const bool mine = lane < take;
    const uint32_t lip = desc.lip, lit = desc.lit, ml = desc.ml, off = desc.off;
    const uint32_t out_end = desc.out_end;
    const uint32_t my_lit = mine ? lit : 0u;
    const uint32_t my_ml = mine ? ml : 0u;
    const uint32_t my_op = out_end - (lit + ml);
    const uint32_t mo = my_op + lit;
    const uint32_t s = mo - off;
    const bool in_ring = s >= info.valid_from && s + kHist >= start + kBatchSpan;

    // Everything a batch produces is written to the ring only and goes to memory in
    // one piece at the end (flush_ring).  Matches whose source is older than the
    // batch and out of the ring's reach read memory: their first loads are issued
    // here, ahead of the literals, which touch LDS alone, so the round trip runs
    // beside that work.  Every lane loads, from a harmless address when it has
    // nothing to copy, so that one wait covers a step's loads.
    const bool far = mine && !in_ring;
    const bool any_far = !(tuning & kPhaseSwitches & 4u) && __builtin_amdgcn_ballot_w64(far) != 0;
    const uint32_t far_len = far ? ml : 0u;
    const uint8_t *from = dst + (far ? s : 0u);
    uint32_t fw[kLitStep];
    if (any_far) {
#pragma unroll
        for (uint32_t k = 0; k < kLitStep; ++k)
            fw[k] = load32(from + ((4u * k < far_len) ? 4u * k : 0u));
    }

    // Literals, whole dwords: up to 3 bytes too many land where the sequence's own
    // match (4 bytes at least) is written afterwards.
    bool far_arrived = false;
    if (!(tuning & kPhaseSwitches & 2u)) {
        for (uint32_t t = 0; __builtin_amdgcn_ballot_w64(t < my_lit) != 0; t += 4u * kLitStep) {
            uint32_t w[kLitStep];
            if (kFromStage) {
#pragma unroll
                for (uint32_t k = 0; k < kLitStep; ++k)
                    if (t + 4u * k < my_lit)
                        w[k] = win->staged32(lip + t + 4u * k);
            } else {
                // From memory, every lane loading (its first literals again when it has none
                // left), and one wait for the step, which covers the far matches' loads too.
#pragma unroll
                for (uint32_t k = 0; k < kLitStep; ++k)
                    w[k] = load32(src_al + (mine ? lip : 0u) + ((t + 4u * k < my_lit) ? t + 4u * k : 0u));
                wait_for_memory();
                far_arrived = true;
            }
#pragma unroll
            for (uint32_t k = 0; k < kLitStep; ++k)
                if (t + 4u * k < my_lit)
                    ring_write32(ring, my_op + t + 4u * k, w[k]);
        }
    }

    PHASE_MARK(clk, 2);
    // Matches are written to the byte: what follows a match is already in place.
    bool waiting = mine;
    if (any_far) {
        if (!far_arrived)
            wait_for_memory();
#pragma unroll
        for (uint32_t k = 0; k < kLitStep; ++k)
            put_ring(ring, mo + 4u * k, fw[k], far_len > 4u * k ? far_len - 4u * k : 0u);
        for (uint32_t t = 4u * kLitStep; __builtin_amdgcn_ballot_w64(t < far_len) != 0; t += 4u * kLitStep) {
#pragma unroll
            for (uint32_t k = 0; k < kLitStep; ++k)
                fw[k] = load32(from + ((t + 4u * k < far_len) ? t + 4u * k : 0u));
            wait_for_memory();
#pragma unroll
            for (uint32_t k = 0; k < kLitStep; ++k)
                put_ring(ring, mo + t + 4u * k, fw[k], far_len > t + 4u * k ? far_len - (t + 4u * k) : 0u);
        }
        waiting = waiting && !far;
    }

    PHASE_MARK(clk, 3);
    // Then those that read the ring, in rounds: a match is copied once every match
    // whose output it reads has been.  These are the sequences from the one that
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
    for (;;) {
        const uint64_t left = __builtin_amdgcn_ballot_w64(waiting);
        if (left == 0 || (tuning & kPhaseSwitches & 8u))
            break;
        const bool go = waiting && (depends_on & left) == 0;
        const uint32_t len = go ? my_ml : 0u;
        for (uint32_t t = 0; __builtin_amdgcn_ballot_w64(t < len) != 0; t += 4u * kMatchStep) {
            uint32_t w[kMatchStep];
#pragma unroll
            for (uint32_t k = 0; k < kMatchStep; ++k)
                if (t + 4u * k < len)
                    w[k] = ring_read32(ring, s + t + 4u * k);
#pragma unroll
            for (uint32_t k = 0; k < kMatchStep; ++k)
                put_ring(ring, mo + t + 4u * k, w[k], len > t + 4u * k ? len - (t + 4u * k) : 0u);
        }
        waiting = waiting && !go;
    }
    PHASE_MARK(clk, 4);
    flush_ring(dst, ring, start, info.end, lane);
    PHASE_MARK(clk, 3);
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
};

struct SoloTeam {
    __device__ __forceinline__ void sync() {}
    __device__ __forceinline__ void copy(uint8_t *dst, const uint8_t *src, uint32_t n, uint32_t lane)
    {
        copy_run(dst, src, n, lane);
    }
    __device__ __forceinline__ void batch(const InputWindow &win, const uint8_t *, uint8_t *dst, uint8_t *ring,
                                          const BatchLane &desc, const BatchInfo &info, uint32_t lane,
                                          uint32_t tuning, PhaseClock &clk)
    {
        consume_batch<true>(&win, nullptr, dst, ring, desc, info, lane, tuning, clk);
    }
};

struct PairTeam {
    static constexpr uint32_t kCopyDepth = 4;
    TeamShared *sh;
    uint32_t seq;   // commands published so far
    bool pending;   // the helper may still be busy with a batch

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
    __device__ __forceinline__ void batch(const InputWindow &, const uint8_t *src_al, uint8_t *dst, uint8_t *,
                                          const BatchLane &desc, const BatchInfo &info, uint32_t lane, uint32_t,
                                          PhaseClock &)
    {
        const uint32_t slot = seq & 1u;
        sh->lanes[slot][0][lane] = desc.lip;
        sh->lanes[slot][1][lane] = desc.lit;
        sh->lanes[slot][2][lane] = desc.ml;
        sh->lanes[slot][3][lane] = desc.off;
        sh->lanes[slot][4][lane] = desc.out_end;
        sh->info[slot] = info;
        publish(kCmdBatch, dst, src_al, 0);
    }
};

// The helper wave of a pair: carries out the parser's commands until told to leave.
__device__ void helper_loop(TeamShared *sh, uint8_t *ring, uint32_t lane, uint32_t tuning)
{
    PhaseClock clk;
    clk.start();
    for (uint32_t seq = 0;; ++seq) {
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
            consume_batch<false>(nullptr, sh->src[slot], sh->dst[slot], ring, desc, info, lane, tuning, clk);
            wait_for_memory();  // the flush is in memory before the parser is told so
        } else if (cmd == kCmdCopy) {
            const uint32_t n = uni(sh->len[slot]);
            const uint32_t half = (n >> 1) & ~15u;
            copy_run<PairTeam::kCopyDepth>(sh->dst[slot] + half, sh->src[slot] + half, n - half, lane);
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
            const uint32_t taken = produce_batch(win, ip, op, valid_from, iend, oend, walk, lane, clk, desc, info);
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
        // r7x refuses a match that would write past the end or end exactly at it (the
        // last bytes are supposed to be literals); one that ends 1-4 bytes short is taken,
        // and the literal rule above makes the next sequence end the block.
        if (ml >= oend - op)
            return false;

        if (ml <= kShortMatch && offset <= kHist - kShortMatch && op - offset >= valid_from) {
            emit_short_match(dst, ring, op, offset, ml, lane);
        } else {
            uint8_t *mdst = dst + op;
            const uint8_t *msrc = mdst - offset;
            if (offset >= ml) {
                team.copy(mdst, msrc, ml, lane);
            } else {
                copy_periodic(team, mdst, offset, ml, lane);
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
    __shared__ uint32_t stage[kStageBlocks * 64u];
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
    __shared__ uint32_t stage[kStageBlocks * 64u];
    __shared__ TeamShared shared;
    const uint32_t lane = threadIdx.x & (kWave - 1);
    __builtin_amdgcn_s_setprio(3);
    if (threadIdx.x >= kWave) {
        helper_loop(&shared, ring, lane, tuning);
        return;
    }
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
    // bit 5 (32): one wave per chunk instead of two.
    static const uint32_t tuning = [] {
        const char *env = std::getenv("SNAPPY_LZ4_TUNING");
        return env ? static_cast<uint32_t>(std::strtoul(env, nullptr, 0)) : 0u;
    }();
#ifdef SNAPPY_LZ4_PHASE_SWITCHES
    unsigned long long zero[9] = {};
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
    unsigned long long c[9];
    (void)hipMemcpyFromSymbol(c, HIP_SYMBOL(g_phase_cycles), sizeof c);
    fprintf(stderr, "lz4 phases (s_memtime ticks summed over %u chunks): walk %llu classify %llu literals %llu "
                    "old sources %llu rounds %llu one-at-a-time %llu staging %llu helper waiting %llu; %llu batches of %.1f sequences\n", n_chunks, c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[8],
            c[7] >> 32, static_cast<double>(c[7] & 0xFFFFFFFFull) / static_cast<double>((c[7] >> 32) ? (c[7] >> 32) : 1));
#endif
    return hipGetLastError();
}

}  // namespace snappy
