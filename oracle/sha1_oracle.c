/*
 * CPU oracle, SHA-1 part -- TEST INFRASTRUCTURE ONLY (see snappy_oracle.h).
 *
 * Scalar FIPS 180-4 SHA-1, written for clarity rather than speed.  It restates
 * what /root/reference/Snappy/Utils/Crypto.cs:8-22 obtains from
 * SHA1.Create().ComputeHash(bytes) followed by the "X2" formatting loop.
 */
#include "snappy_oracle.h"

#include <string.h>

static uint32_t rotl32(uint32_t x, unsigned n) { return (x << n) | (x >> (32 - n)); }

static void sha1_compress(uint32_t h[5], const uint8_t block[64])
{
    uint32_t w[80];
    for (int t = 0; t < 16; ++t)
        w[t] = ((uint32_t)block[4 * t] << 24) | ((uint32_t)block[4 * t + 1] << 16) |
               ((uint32_t)block[4 * t + 2] << 8) | (uint32_t)block[4 * t + 3];
    for (int t = 16; t < 80; ++t)
        w[t] = rotl32(w[t - 3] ^ w[t - 8] ^ w[t - 14] ^ w[t - 16], 1);

    uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4];
    for (int t = 0; t < 80; ++t) {
        uint32_t f, k;
        if (t < 20) {
            f = (b & c) | (~b & d);
            k = 0x5A827999u;
        } else if (t < 40) {
            f = b ^ c ^ d;
            k = 0x6ED9EBA1u;
        } else if (t < 60) {
            f = (b & c) | (b & d) | (c & d);
            k = 0x8F1BBCDCu;
        } else {
            f = b ^ c ^ d;
            k = 0xCA62C1D6u;
        }
        uint32_t tmp = rotl32(a, 5) + f + e + k + w[t];
        e = d;
        d = c;
        c = rotl32(b, 30);
        b = a;
        a = tmp;
    }
    h[0] += a;
    h[1] += b;
    h[2] += c;
    h[3] += d;
    h[4] += e;
}

/* Crypto.cs:10-11: one-shot hash of the whole array. */
void oracle_sha1(const uint8_t *data, uint64_t len, uint8_t out20[20])
{
    uint32_t h[5] = {0x67452301u, 0xEFCDAB89u, 0x98BADCFEu, 0x10325476u, 0xC3D2E1F0u};
    uint64_t full = len / 64;
    for (uint64_t i = 0; i < full; ++i)
        sha1_compress(h, data + 64 * i);

    /* Padding: 0x80, zeros, 64-bit big-endian bit length. */
    uint8_t tail[128];
    uint64_t rem = len - 64 * full;
    memset(tail, 0, sizeof tail);
    if (rem)
        memcpy(tail, data + 64 * full, rem);
    tail[rem] = 0x80;
    unsigned tail_len = (rem < 56) ? 64 : 128;
    uint64_t bits = len * 8;
    for (int i = 0; i < 8; ++i)
        tail[tail_len - 1 - i] = (uint8_t)(bits >> (8 * i));
    sha1_compress(h, tail);
    if (tail_len == 128)
        sha1_compress(h, tail + 64);

    for (int i = 0; i < 5; ++i) {
        out20[4 * i] = (uint8_t)(h[i] >> 24);
        out20[4 * i + 1] = (uint8_t)(h[i] >> 16);
        out20[4 * i + 2] = (uint8_t)(h[i] >> 8);
        out20[4 * i + 3] = (uint8_t)h[i];
    }
}

/* Crypto.cs:14-22: every byte as two UPPERCASE hex digits, no separators. */
void oracle_sha1_hex(const uint8_t *data, uint64_t len, char out41[41])
{
    static const char digits[] = "0123456789ABCDEF";
    uint8_t d[20];
    oracle_sha1(data, len, d);
    for (int i = 0; i < 20; ++i) {
        out41[2 * i] = digits[d[i] >> 4];
        out41[2 * i + 1] = digits[d[i] & 15];
    }
    out41[40] = '\0';
}

void oracle_sha1_batch(const uint8_t *base, const uint64_t *offsets,
                       const uint64_t *lengths, uint64_t n, uint8_t *digests)
{
    for (uint64_t i = 0; i < n; ++i)
        oracle_sha1(base + offsets[i], lengths[i], digests + 20 * i);
}

/* splitmix64 byte stream; little-endian words, truncated at len. */
void oracle_fill_splitmix(uint8_t *dst, uint64_t len, uint64_t seed)
{
    uint64_t x = seed;
    uint64_t i = 0;
    while (i < len) {
        x += 0x9E3779B97F4A7C15ull;
        uint64_t z = x;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        for (int b = 0; b < 8 && i < len; ++b, ++i)
            dst[i] = (uint8_t)(z >> (8 * b));
    }
}
