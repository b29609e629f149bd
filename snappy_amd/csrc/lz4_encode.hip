// LZ4 block compressor for gfx950: one lz4net chunk per wavefront.
//
// The reference never writes an archive (its McdfHeader.WriteToStream,
// /root/reference/Snappy/Features/Mcdf/McdfHeader.cs:14-24, has no caller); this
// is the inverse of the import path (McdfManager.cs:72-80), so that archives can
// be produced where the assets already are.  Any valid LZ4 block decodes to the
// same bytes, so the output need not match another compressor's byte for byte;
// it must only obey the block format and the end-of-block rules the decoder
// lz4net ports (LZ4 r7x) enforces: the last 5 bytes are literals and no match
// starts within the last 12 bytes.
//
// Greedy matching, 64 positions at a time: lane i hashes the 4 bytes at
// position ip + i, looks its candidate up in an LDS table of 8192 earlier
// positions and enters its own position; the first lane with a verified match
// wins, the wave extends the match (64 bytes, then up to 1 KiB per step), emits "literals so far +
// match" cooperatively and continues behind it.  Without a match the window
// moves on by 64.
//
// What a chunk's time is made of is memory round trips, one after the other, so
// kWindows windows are looked at per trip: their table look-ups and entries are
// made in window order, all their candidates are fetched together (with the next
// trip's positions already on their way), and when a window holds a match the
// entries of the windows behind it are taken back, youngest first, by storing what
// their look-ups had found.  The table, and with it the stream that comes out, is
// byte for byte what one window per trip gives.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lz4_launch.hpp"

namespace snappy {

namespace {

constexpr uint32_t kHashBits = 13;
constexpr uint32_t kNoPos = 0xFFFFFFFFu;
constexpr uint32_t kMinMatch = 4;
constexpr uint32_t kMfLimit = 12;      // no match may start in the last 12 bytes
constexpr uint32_t kLastLiterals = 5;  // the last 5 bytes are always literals
#ifndef SNAPPY_LZ4_ENC_WINDOWS
#define SNAPPY_LZ4_ENC_WINDOWS 8
#endif
constexpr uint32_t kWindows = SNAPPY_LZ4_ENC_WINDOWS;  // windows of 64 positions per memory round trip

__device__ __forceinline__ uint32_t load32(const uint8_t *p)
{
    uint32_t v;
    __builtin_memcpy(&v, p, 4);
    return v;
}

__device__ __forceinline__ void store32(uint8_t *p, uint32_t v) { __builtin_memcpy(p, &v, 4); }

// Writes a length's extension bytes (value = length - 15): 255s, then the rest.
__device__ __forceinline__ uint32_t put_length(uint8_t *out, uint32_t o, uint32_t value, uint32_t lane)
{
    const uint32_t full = value / 255u;
    for (uint32_t i = lane; i < full; i += 64)
        out[o + i] = 255;
    out[o + full] = static_cast<uint8_t>(value - full * 255u);
    return o + full + 1;
}

// Token, literal length, literals; returns the position after them and leaves the
// token's address in *token_at so that the match length can be added.
__device__ __forceinline__ uint32_t put_literals(uint8_t *__restrict__ out, uint32_t o,
                                                 const uint8_t *__restrict__ src, uint32_t lit, uint32_t lane,
                                                 uint32_t *token_at)
{
    *token_at = o;
    out[o] = static_cast<uint8_t>((lit >= 15 ? 15u : lit) << 4);
    ++o;
    if (lit >= 15)
        o = put_length(out, o, lit - 15, lane);
    if (lit <= 64u) {  // a short run is one byte per lane: a single load and store, nothing to piece together
        if (lane < lit)
            out[o + lane] = src[lane];
        return o + lit;
    }
    // Longer runs in dwords at any byte address, the last lit & 3 bytes one per lane; every load
    // of a pass is issued before its first store (a load behind a store would wait for a trip of
    // its own).
    const uint32_t whole = lit & ~3u;
    const bool tail = lane < (lit & 3u);
    const uint8_t last = tail ? src[whole + lane] : uint8_t(0);
    // Passes of 4 KiB (16 bytes per lane, four of them in flight) while that much is left, ...
    uint32_t done = 0;
    for (; done + 4096u <= whole; done += 4096u) {
        uint4 v[4];
#pragma unroll
        for (uint32_t k = 0; k < 4u; ++k)
            __builtin_memcpy(&v[k], src + done + 16u * lane + 1024u * k, 16);
#pragma unroll
        for (uint32_t k = 0; k < 4u; ++k)
            __builtin_memcpy(out + o + done + 16u * lane + 1024u * k, &v[k], 16);
    }
    // ... then of 1 KiB, a dword per lane.
    for (uint32_t i = done + 4u * lane; i < whole; i += 1024u) {
        uint32_t v[4];
#pragma unroll
        for (uint32_t k = 0; k < 4u; ++k)
            v[k] = i + 256u * k < whole ? load32(src + i + 256u * k) : 0u;
#pragma unroll
        for (uint32_t k = 0; k < 4u; ++k)
            if (i + 256u * k < whole)
                store32(out + o + i + 256u * k, v[k]);
    }
    if (tail)
        out[o + whole + lane] = last;
    return o + lit;
}

// One trip: W windows of 64 positions from ip on; seq[j] holds what was loaded for window j's
// positions (anything for a position at or beyond mflimit).  Their look-ups and entries are made in
// window order, all candidates are fetched together with the next trip's positions behind
// them, and when window w holds a match the entries of the windows behind it are taken back,
// youngest first, by storing what their look-ups had found: the table is then what it would
// be had the windows been visited one per trip, up to and including w.
template <uint32_t W, uint32_t kMax>
__device__ __forceinline__ uint64_t look(const uint8_t *__restrict__ in, uint32_t *table, uint32_t ip,
                                         uint32_t mflimit, uint32_t lane, const uint32_t (&seq)[kMax],
                                         uint32_t (&next)[kMax], uint32_t &w, uint32_t &ref_all)
{
    uint32_t h[W], cand[W], there[W], sq[W];
    bool here[W], maybe[W];
#pragma unroll
    for (uint32_t j = 0; j < W; ++j) {
        const uint32_t p = ip + 64u * j + lane;
        here[j] = p < mflimit;
        sq[j] = here[j] ? seq[j] : 0u;
        h[j] = (sq[j] * 2654435761u) >> (32 - kHashBits);
        cand[j] = here[j] ? table[h[j]] : kNoPos;
        if (here[j])
            table[h[j]] = p;  // after every lane's read: LDS serves a wave's accesses in order
        maybe[j] = cand[j] != kNoPos && cand[j] < p && p - cand[j] <= 65535u;
    }
    // A lane without a candidate reads its own position (or the chunk's first bytes).
#pragma unroll
    for (uint32_t j = 0; j < W; ++j)
        there[j] = load32(in + (maybe[j] ? cand[j] : (here[j] ? ip + 64u * j + lane : 0u)));
    // (One window at a time is for streams where nearly every trip finds a match: nothing is
    // asked for in advance there.)
    if constexpr (W > 1) {
#pragma unroll
        for (uint32_t j = 0; j < W; ++j) {
            const uint32_t p = ip + 64u * (W + j) + lane;
            next[j] = load32(in + (p < mflimit ? p : 0u));
        }
    }
    uint64_t found = 0;
    w = 0;
    ref_all = 0;
#pragma unroll
    for (uint32_t j = W; j-- > 0;) {
        const uint64_t f = __ballot(maybe[j] && there[j] == sq[j]);
        if (f != 0) {
            found = f;
            w = j;
            ref_all = cand[j];
        }
    }
    if (found != 0) {
#pragma unroll
        for (uint32_t j = W; j-- > 1;)
            if (j > w && here[j])
                table[h[j]] = cand[j];
    }
    return found;
}

__global__ void __launch_bounds__(64)
lz4_encode_kernel(const uint8_t *__restrict__ base_in, uint8_t *__restrict__ base_out,
                  const Lz4EncChunk *__restrict__ chunks, uint32_t *__restrict__ out_sizes)
{
    __shared__ uint32_t table[1u << kHashBits];
    const uint32_t lane = threadIdx.x;
    const Lz4EncChunk c = chunks[blockIdx.x];
    const uint8_t *in = base_in + c.in_off;
    uint8_t *out = base_out + c.out_off;
    const uint32_t len = c.in_len;

    for (uint32_t i = lane; i < (1u << kHashBits); i += 64)
        table[i] = kNoPos;
    __syncthreads();

    uint32_t o = 0, anchor = 0, ip = 0;
    if (len > kMfLimit) {
        const uint32_t mflimit = len - kMfLimit;
        const uint32_t matchlimit = len - kLastLiterals;
        // The 4 bytes at each position of windows [j0, j1) counted from `from` (the chunk's first
        // bytes for a position at or beyond mflimit: look() does not use them).
        auto fetch = [&](uint32_t from, uint32_t (&seq)[kWindows], uint32_t j0, uint32_t j1) {
#pragma unroll
            for (uint32_t j = 0; j < kWindows; ++j) {
                const uint32_t p = from + 64u * j + lane;
                if (j >= j0 && j < j1)
                    seq[j] = load32(in + (p < mflimit ? p : 0u));
            }
        };
        // Where matches follow each other closely the windows behind the first are looked at in
        // vain, so a match in a trip's first window makes the next trips narrow (one window)
        // until one of them finds nothing.  The stream does not depend on this choice.
        uint32_t wide = 1;  // the same in every lane, and said so to the compiler where it decides a branch
        uint32_t seq[kWindows];
        fetch(ip, seq, 0, kWindows);
        while (ip < mflimit) {
            uint32_t next[kWindows], w, ref_all;
            uint64_t found;
            if (__builtin_amdgcn_readfirstlane(wide)) {
                found = look<kWindows>(in, table, ip, mflimit, lane, seq, next, w, ref_all);
                if (found == 0) {
                    ip += 64u * kWindows;
#pragma unroll
                    for (uint32_t j = 0; j < kWindows; ++j)
                        seq[j] = next[j];
                    continue;
                }
            } else {
                found = look<1>(in, table, ip, mflimit, lane, seq, next, w, ref_all);
                if (found == 0) {
                    ip += 64u;
                    wide = 1;
                    fetch(ip, seq, 0, kWindows);
                    continue;
                }
            }
            wide = __builtin_amdgcn_readfirstlane(wide != 0u && w != 0u ? 1u : 0u);
            const uint32_t first = static_cast<uint32_t>(__builtin_ctzll(found));
            const uint32_t m = ip + 64u * w + first;
            const uint32_t ref = __builtin_amdgcn_readlane(ref_all, first);
            // The first 64 bytes one per lane (most matches end there, and a byte load has nothing
            // to take apart), then 16 or 4 bytes per lane and step; a lane's dword may reach past
            // matchlimit (never past the chunk's end: matchlimit is 5 bytes short of it), its
            // bytes beyond do not count.
            uint32_t mlen = kMinMatch;
            bool open_end;
            {
                const uint32_t q = m + mlen + lane;
                const bool same = q < matchlimit && in[q] == in[ref + mlen + lane];
                const uint64_t differ = __ballot(!same);
                open_end = differ == 0;
                mlen += open_end ? 64u : static_cast<uint32_t>(__builtin_ctzll(differ));
            }
            // (1 KiB per step, 16 bytes per lane, while that much may still be covered, ...)
            while (open_end && m + mlen + 1024u <= matchlimit) {
                uint4 a, b;
                __builtin_memcpy(&a, in + m + mlen + 16u * lane, 16);
                __builtin_memcpy(&b, in + ref + mlen + 16u * lane, 16);
                const uint32_t x[4] = {a.x ^ b.x, a.y ^ b.y, a.z ^ b.z, a.w ^ b.w};
                uint32_t same = 16u;
#pragma unroll
                for (uint32_t k = 4; k-- > 0;)
                    if (x[k] != 0u)
                        same = 4u * k + (static_cast<uint32_t>(__builtin_ctz(x[k])) >> 3);
                const uint64_t differ = __ballot(same < 16u);
                if (differ) {
                    const uint32_t at = static_cast<uint32_t>(__builtin_ctzll(differ));
                    mlen += 16u * at + __builtin_amdgcn_readlane(same, at);
                    open_end = false;
                    break;
                }
                mlen += 1024u;
            }
            // (... then a dword per lane.)
            while (open_end) {
                const uint32_t q = m + mlen + 4u * lane;
                const uint32_t valid = q + 4u <= matchlimit ? 4u : (q < matchlimit ? matchlimit - q : 0u);
                uint32_t same = 0;
                if (valid != 0u) {
                    const uint32_t x = load32(in + q) ^ load32(in + ref + mlen + 4u * lane);
                    same = x == 0u ? 4u : static_cast<uint32_t>(__builtin_ctz(x)) >> 3;
                    same = same < valid ? same : valid;
                }
                const uint64_t differ = __ballot(same < 4u);
                if (differ) {
                    const uint32_t at = static_cast<uint32_t>(__builtin_ctzll(differ));
                    mlen += 4u * at + __builtin_amdgcn_readlane(same, at);
                    break;
                }
                mlen += 256u;
            }
            // The next trip's positions are on their way while the sequence is written.
            if (wide)
                fetch(m + mlen, seq, 0, kWindows);
            else
                fetch(m + mlen, seq, 0, 1);
            uint32_t token_at;
            o = put_literals(out, o, in + anchor, m - anchor, lane, &token_at);
            const uint32_t offset = m - ref;
            out[o] = static_cast<uint8_t>(offset);
            out[o + 1] = static_cast<uint8_t>(offset >> 8);
            o += 2;
            const uint32_t mcode = mlen - kMinMatch;
            // The token was written by this wave a moment ago: add the match nibble
            // in a second store of the whole byte (same value from every lane).
            const uint32_t lit = m - anchor;
            out[token_at] = static_cast<uint8_t>(((lit >= 15 ? 15u : lit) << 4) | (mcode >= 15 ? 15u : mcode));
            if (mcode >= 15)
                o = put_length(out, o, mcode - 15, lane);
            ip = anchor = m + mlen;
        }
    }
    uint32_t token_at;
    o = put_literals(out, o, in + anchor, len - anchor, lane, &token_at);
    if (lane == 0)
        out_sizes[blockIdx.x] = o;
}

// Source and destination start at any byte; dword accesses at any byte address
// are fine on this hardware, so the copy runs in dwords with a byte tail.
__global__ void __launch_bounds__(256)
lz4_pack_kernel(uint8_t *__restrict__ base, const Lz4PackItem *__restrict__ items)
{
    const Lz4PackItem it = items[blockIdx.x];
    const uint8_t *src = base + it.src;
    uint8_t *dst = base + it.dst;
    const uint64_t words = it.len >> 2;
    for (uint64_t i = threadIdx.x; i < words; i += 256) {
        uint32_t w;
        __builtin_memcpy(&w, src + 4 * i, 4);
        __builtin_memcpy(dst + 4 * i, &w, 4);
    }
    const uint64_t tail = words << 2;
    if (threadIdx.x < (it.len & 3u))
        dst[tail + threadIdx.x] = src[tail + threadIdx.x];
}

}  // namespace

hipError_t launch_lz4_pack(uint8_t *d_base, const Lz4PackItem *d_items, uint32_t n_items,
                           hipStream_t stream)
{
    if (n_items == 0)
        return hipSuccess;
    hipLaunchKernelGGL(lz4_pack_kernel, dim3(n_items), dim3(256), 0, stream, d_base, d_items);
    return hipGetLastError();
}

hipError_t launch_lz4_encode(const uint8_t *d_in, uint8_t *d_out, const Lz4EncChunk *d_chunks,
                             uint32_t n_chunks, uint32_t *d_out_sizes, hipStream_t stream)
{
    if (n_chunks == 0)
        return hipSuccess;
    hipLaunchKernelGGL(lz4_encode_kernel, dim3(n_chunks), dim3(64), 0, stream, d_in, d_out,
                       d_chunks, d_out_sizes);
    return hipGetLastError();
}

}  // namespace snappy
