// SHA-1 building blocks for gfx950 (CDNA4) device code.
//
// SHA-1 (FIPS 180-4) is what /root/reference/Snappy/Utils/Crypto.cs:10-11 gets
// from SHA1.Create().ComputeHash.  It is 32-bit integer work: rotates map to
// v_alignbit_b32, the three-input sums to v_add3_u32, Ch / Maj / Parity and the
// schedule's three-way xor to one v_bitop3_b32 each (gfx950's 3-input truth-
// table op), and the big-endian load plus any sub-dword misalignment of the
// message to one v_perm_b32 per word: 613 vector instructions per 64-byte
// block (16 perm, 64 x 3 schedule, 80 x 5 round, 5 feed-forward adds).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace snappy {

// One message to hash: bytes [offset, offset+len) of the batch's base pointer;
// the digest goes to digests + 20*out_index.  ready_epoch != 0: the bytes are
// still on their way from the host and are complete once the batch's arrival
// counter has reached this value.
struct Sha1Task {
    uint64_t offset;
    uint64_t len;
    uint32_t out_index;
    uint32_t ready_epoch;
};

constexpr uint32_t kSha1Init[5] = {0x67452301u, 0xEFCDAB89u, 0x98BADCFEu, 0x10325476u,
                                   0xC3D2E1F0u};
constexpr uint32_t kSha1K[4] = {0x5A827999u, 0x6ED9EBA1u, 0x8F1BBCDCu, 0xCA62C1D6u};

__device__ __forceinline__ uint32_t rotl32(uint32_t x, uint32_t n)
{
    return __builtin_amdgcn_alignbit(x, x, 32u - n);
}

// v_bitop3_b32 truth tables with the operands weighted 0xF0, 0xCC, 0xAA.
constexpr uint32_t kLutCh = 0xCA;      // (b & c) | (~b & d)
constexpr uint32_t kLutParity = 0x96;  // b ^ c ^ d
constexpr uint32_t kLutMaj = 0xE8;     // (b & c) | (b & d) | (c & d)

__device__ __forceinline__ uint32_t xor3(uint32_t a, uint32_t b, uint32_t c)
{
    return __builtin_amdgcn_bitop3_b32(a, b, c, kLutParity);
}

__device__ __forceinline__ uint32_t sha1_f(int t, uint32_t b, uint32_t c, uint32_t d)
{
    if (t < 20)
        return __builtin_amdgcn_bitop3_b32(b, c, d, kLutCh);
    if (t < 40 || t >= 60)
        return __builtin_amdgcn_bitop3_b32(b, c, d, kLutParity);
    return __builtin_amdgcn_bitop3_b32(b, c, d, kLutMaj);
}

// Next schedule word from the 16-word ring (w[t] for t >= 16), stored in place.
__device__ __forceinline__ uint32_t sha1_schedule(uint32_t (&w)[16], int t)
{
    const uint32_t wt =
        rotl32(xor3(w[(t - 3) & 15], w[(t - 8) & 15], w[(t - 14) & 15]) ^ w[t & 15], 1);
    w[t & 15] = wt;
    return wt;
}

// One round.  `g` = f + e + W[t] does not depend on the newest state word, so
// the serial chain through `a` is one rotate and one add; the empty asm keeps
// the compiler from re-associating the sum onto that chain.  kBiased: the
// caller's word already holds K + W[t].
template <bool kBiased>
__device__ __forceinline__ void sha1_round(int t, uint32_t &a, uint32_t &b, uint32_t &c,
                                           uint32_t &d, uint32_t &e, uint32_t wt)
{
    uint32_t g = sha1_f(t, b, c, d) + e + wt;
    asm volatile("" : "+v"(g));
    const uint32_t tmp = kBiased ? rotl32(a, 5) + g : rotl32(a, 5) + g + kSha1K[t / 20];
    e = d;
    d = c;
    c = rotl32(b, 30);
    b = a;
    a = tmp;
}

// Expands w[0..15] in place as a 16-word ring and runs the 80 rounds.
__device__ __forceinline__ void sha1_compress(uint32_t (&h)[5], uint32_t (&w)[16])
{
    uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4];
#pragma unroll
    for (int t = 0; t < 80; ++t) {
        const uint32_t wt = (t < 16) ? w[t] : sha1_schedule(w, t);
        sha1_round<false>(t, a, b, c, d, e, wt);
    }
    h[0] += a;
    h[1] += b;
    h[2] += c;
    h[3] += d;
    h[4] += e;
}

// Selector for v_perm_b32 that reads a big-endian 32-bit word starting at byte
// `shift` (0..3) of the little-endian dword pair {lo, hi}.
__device__ __forceinline__ uint32_t be_selector(uint32_t shift)
{
    return 0x00010203u + shift * 0x01010101u;
}

// Loads the aligned dwords that overlap the 64 message bytes at p: 16 of them
// when p is dword-aligned (shift == 0), 17 otherwise, so no byte outside a dword
// that holds message bytes is touched.  shift = address of p modulo 4.
__device__ __forceinline__ void load_block_raw(const uint8_t *p, uint32_t shift, uint32_t (&d)[17])
{
    const uint32_t *q = reinterpret_cast<const uint32_t *>(p - shift);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        uint32_t v[4];
        __builtin_memcpy(v, q + 4 * i, 16);  // dword-aligned 16-byte load
        d[4 * i] = v[0];
        d[4 * i + 1] = v[1];
        d[4 * i + 2] = v[2];
        d[4 * i + 3] = v[3];
    }
    d[16] = q[shift ? 16 : 15];
}

// The first 16 of those dwords (four 16-byte loads), and the 17th alone.
__device__ __forceinline__ void load_block_raw16(const uint8_t *p, uint32_t shift, uint32_t (&d)[17])
{
    const uint32_t *q = reinterpret_cast<const uint32_t *>(p - shift);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        uint32_t v[4];
        __builtin_memcpy(v, q + 4 * i, 16);
        d[4 * i] = v[0];
        d[4 * i + 1] = v[1];
        d[4 * i + 2] = v[2];
        d[4 * i + 3] = v[3];
    }
}

__device__ __forceinline__ uint32_t load_dword_after_block(const uint8_t *p, uint32_t shift)
{
    const uint32_t *q = reinterpret_cast<const uint32_t *>(p - shift);
    return q[shift ? 16 : 15];
}

// Turns those dwords into the block's 16 big-endian words: one v_perm_b32 per
// word does the byte swap and the sub-dword shift together.
__device__ __forceinline__ void raw_block_to_be(const uint32_t (&d)[17], uint32_t shift,
                                                uint32_t (&w)[16])
{
    const uint32_t sel = be_selector(shift);
#pragma unroll
    for (int t = 0; t < 16; ++t)
        w[t] = __builtin_amdgcn_perm(d[t + 1], d[t], sel);
}

// The 64 message bytes at p (any byte alignment) as 16 big-endian words.
__device__ __forceinline__ void load_block_be(const uint8_t *p, uint32_t (&w)[16])
{
    const uint32_t shift = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(p) & 3u);
    uint32_t d[17];
    load_block_raw(p, shift, d);
    raw_block_to_be(d, shift, w);
}

// Builds a padding block from the last `rem` (0..63) message bytes at p.
// The block holds those bytes, 0x80, zeros, and -- when rem < 56 -- the 64-bit
// big-endian bit count.  Returns true when the length was written (the block is
// the message's last); otherwise one more block from make_length_block follows.
// Only dwords that overlap [p, p+rem) are read.
__device__ __forceinline__ bool load_tail_block_be(const uint8_t *p, uint32_t rem,
                                                   uint64_t total_len, uint32_t (&w)[16])
{
    const uint32_t shift = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(p) & 3u);
    const uint32_t *q = reinterpret_cast<const uint32_t *>(p - shift);
    const uint32_t sel = be_selector(shift);
    // Dword i (from q) overlaps the message iff rem > 0 and 4*i < shift + rem.
    // All 17 loads are issued together; indices past the last overlapping dword
    // are clamped onto it (their bytes are masked off below).
    uint32_t d[17];
    if (rem) {
        const uint32_t last_dword = (shift + rem - 1u) >> 2;
#pragma unroll
        for (int i = 0; i < 17; ++i)
            d[i] = q[static_cast<uint32_t>(i) < last_dword ? static_cast<uint32_t>(i) : last_dword];
    } else {
#pragma unroll
        for (int i = 0; i < 17; ++i)
            d[i] = 0u;
    }
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        uint32_t word = __builtin_amdgcn_perm(d[t + 1], d[t], sel);
        // Valid message bytes in this word: rem - 4t clamped to [0, 4].
        const int valid = static_cast<int>(rem) - 4 * t;
        if (valid <= 0) {
            word = (valid == 0) ? 0x80000000u : 0u;
        } else if (valid < 4) {
            const uint32_t keep = ~(0xFFFFFFFFu >> (8 * valid));
            word = (word & keep) | (0x80000000u >> (8 * valid));
        }
        w[t] = word;
    }
    if (rem < 56) {
        const uint64_t bits = total_len << 3;
        w[14] = static_cast<uint32_t>(bits >> 32);
        w[15] = static_cast<uint32_t>(bits);
        return true;
    }
    return false;
}

__device__ __forceinline__ void make_length_block(uint64_t total_len, uint32_t (&w)[16])
{
#pragma unroll
    for (int t = 0; t < 14; ++t)
        w[t] = 0u;
    const uint64_t bits = total_len << 3;
    w[14] = static_cast<uint32_t>(bits >> 32);
    w[15] = static_cast<uint32_t>(bits);
}

// Stores the state as the 20-byte big-endian digest (dst has no alignment).
__device__ __forceinline__ void store_digest(uint8_t *dst, const uint32_t (&h)[5])
{
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        dst[4 * i] = static_cast<uint8_t>(h[i] >> 24);
        dst[4 * i + 1] = static_cast<uint8_t>(h[i] >> 16);
        dst[4 * i + 2] = static_cast<uint8_t>(h[i] >> 8);
        dst[4 * i + 3] = static_cast<uint8_t>(h[i]);
    }
}

}  // namespace snappy
