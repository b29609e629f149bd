// Wave-cooperative LZ4 block decoder for gfx950.
//
// Decodes the chunks of an lz4net LZ4Stream -- what
// /root/reference/Snappy/Features/Mcdf/McdfManager.cs:73-79 obtains from
// `new LZ4Stream(fs, Decompress, ...).CopyTo(ms)`.  Chunks are independent raw
// LZ4 blocks (or stored bytes), so one wavefront owns one chunk at a time and
// draws chunks from a device queue.
//
// Inside a chunk the token stream is serial.  The wave keeps a 256-byte window
// of the compressed input spread over its 64 lanes (one dword per lane) and
// parses tokens on the scalar unit with v_readlane, so parsing never waits on
// memory except to slide the window.  Literal and match copies are done by all
// 64 lanes with consecutive lanes on consecutive bytes.  A match whose offset
// is smaller than its length repeats a pattern of `offset` bytes; it is copied
// as out[op+i] = out[op-offset + i mod offset], which reads only bytes written
// by earlier sequences, so every lane can go at once.
//
// Acceptance rules are those of the decoder lz4net ports (LZ4 r7x
// LZ4_uncompress with a known output length): a literal run that ends within 8
// bytes of the end must end the block exactly; a match must end at least 5
// bytes before the end; consumed input must equal the chunk's compressed
// length.  Every read and write is bounds-checked against the chunk, so a
// corrupt stream yields a status word, never a stray access.
#include "lz4_launch.hpp"
#include "wave_util.hpp"

namespace snappy {

namespace {

constexpr int kWave = 64;
constexpr uint32_t kCopyLength = 8;
constexpr uint32_t kLastLiterals = 5;

__device__ __forceinline__ uint32_t uni(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }

// 256-byte window of the chunk's compressed bytes: lane l holds the aligned
// dword number (first_dword + l) counted from `aligned`.
struct InputWindow {
    const uint32_t *aligned;  // dword-aligned pointer at or below the chunk's first byte
    uint32_t limit_dwords;    // dwords that may be read from `aligned`
    uint32_t first_dword;     // uniform
    uint32_t dword;           // per lane

    __device__ __forceinline__ void fill(uint32_t first, uint32_t lane)
    {
        first_dword = first;
        const uint32_t idx = first + lane;
        dword = (idx < limit_dwords) ? aligned[idx] : 0u;
    }

    // The four bytes at byte position `pos` (counted from `aligned`), little-endian.
    __device__ __forceinline__ uint32_t peek32(uint32_t pos, uint32_t lane)
    {
        uint32_t di = (pos >> 2) - first_dword;
        if (di >= 63u) {  // also catches positions before the window
            fill(pos >> 2, lane);
            di = 0;
        }
        const uint32_t lo = __builtin_amdgcn_readlane(dword, di);
        const uint32_t hi = __builtin_amdgcn_readlane(dword, di + 1u);
        const uint64_t both = (static_cast<uint64_t>(hi) << 32) | lo;
        return static_cast<uint32_t>(both >> ((pos & 3u) * 8u));
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
    // waiting for the store as well; issued ahead of it, it can.
    constexpr uint32_t kDepth = 8;
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

__device__ __forceinline__ void copy_run(uint8_t *dst, const uint8_t *src, uint32_t n,
                                         uint32_t lane)
{
    if (n >= kLongCopy)
        copy_long(dst, src, n, lane);
    else
        copy_bytes(dst, src, n, lane);
}

// Overlapping match: the output repeats the `period` bytes that end at dst.
// The first `span` bytes (a multiple of the period, at least kPeriodSpan) are
// written as dst[i] = dst[i mod period - period], which reads only bytes that
// existed before the match; the rest is filled by doubling.
constexpr uint32_t kPeriodSpan = 4096;

__device__ __forceinline__ void copy_periodic(uint8_t *dst, uint32_t period, uint32_t n,
                                              uint32_t lane)
{
    const uint8_t *pat = dst - period;
    const uint32_t reps = (kPeriodSpan + period - 1u) / period;
    const uint32_t span = reps * period;
    const uint32_t head = n < span ? n : span;
    if (period >= 4u * kWave) {
        // Long pattern: each repetition is a plain copy of the pattern itself.
        for (uint32_t at = 0; at < head; at += period)
            copy_run(dst + at, pat, (head - at < period) ? head - at : period, lane);
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
        copy_run(dst + done, dst, piece, lane);
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

__device__ __forceinline__ void emit_short_literals(const InputWindow &win, uint32_t ip,
                                                    uint32_t lit, uint8_t *out, uint8_t *ring,
                                                    uint32_t op, uint32_t lane)
{
    // The caller has made sure the window covers [ip, ip + lit).
    for (uint32_t base = 0; base < lit; base += kWave) {
        const uint32_t i = base + lane;
        const bool on = i < lit;
        const uint32_t pos = ip + (on ? i : 0u);
        const uint32_t from_lane = (pos >> 2) - win.first_dword;
        const uint32_t word = static_cast<uint32_t>(
            __builtin_amdgcn_ds_bpermute(static_cast<int>(from_lane << 2), static_cast<int>(win.dword)));
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

// Decodes one compressed chunk; returns true when the block is well-formed.
__device__ bool decode_block(const uint8_t *__restrict__ in_base, uint64_t in_alloc,
                             const Lz4Chunk &ck, uint8_t *out_base, uint8_t *ring, uint32_t lane)
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
    win.fill(0, lane);

    uint32_t ip = in_shift;  // uniform
    uint32_t op = 0;         // uniform
    uint32_t valid_from = 0; // uniform: the ring holds output positions [valid_from, op)
    const uint8_t *src_al = src - in_shift;

    for (;;) {
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
            if ((ip >> 2) < win.first_dword || ((ip + lit + 3u) >> 2) > win.first_dword + 64u)
                win.fill(ip >> 2, lane);
            emit_short_literals(win, ip, lit, dst, ring, op, lane);
        } else if (lit != 0u) {
            copy_run(dst + op, src_al + ip, lit, lane);
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
        if (ml > oend - op || op + ml + kLastLiterals > oend)
            return false;

        if (ml <= kShortMatch && offset <= kHist - kShortMatch && op - offset >= valid_from) {
            emit_short_match(dst, ring, op, offset, ml, lane);
        } else {
            uint8_t *mdst = dst + op;
            const uint8_t *msrc = mdst - offset;
            if (offset >= ml) {
                copy_run(mdst, msrc, ml, lane);
            } else {
                copy_periodic(mdst, offset, ml, lane);
            }
            valid_from = op + ml;
        }
        op += ml;
    }
}

__global__ void __launch_bounds__(256)
lz4_decode_kernel(const uint8_t *__restrict__ in_base, uint64_t in_alloc, uint8_t *out_base,
                  const Lz4Chunk *__restrict__ chunks, uint32_t n_chunks,
                  uint32_t *__restrict__ queue, uint32_t *__restrict__ status)
{
    __shared__ uint8_t rings[4][kHist];
    const uint32_t lane = threadIdx.x & (kWave - 1);
    uint8_t *ring = rings[__builtin_amdgcn_readfirstlane(threadIdx.x >> 6)];
    for (;;) {
        const uint32_t i = wave_claim(queue, 1u, lane);
        if (i >= n_chunks)
            break;
        const Lz4Chunk ck = chunks[i];
        bool ok = true;
        if (ck.compressed) {
            ok = decode_block(in_base, in_alloc, ck, out_base, ring, lane);
        } else {
            copy_run(out_base + ck.out_off, in_base + ck.in_off, ck.orig_len, lane);
        }
        // Every lane stores the same word: no lane-0-only tail (see wave_claim).
        status[i] = ok ? 0u : 1u;
    }
}

}  // namespace

hipError_t launch_lz4_decode(const uint8_t *d_in, uint64_t in_alloc, uint8_t *d_out,
                             const Lz4Chunk *d_chunks, uint32_t n_chunks, uint32_t *d_queue,
                             uint32_t *d_status, int blocks, hipStream_t stream)
{
    if (n_chunks == 0 || blocks <= 0)
        return hipSuccess;
    hipLaunchKernelGGL(lz4_decode_kernel, dim3(blocks), dim3(256), 0, stream, d_in, in_alloc, d_out,
                       d_chunks, n_chunks, d_queue, d_status);
    return hipGetLastError();
}

}  // namespace snappy
