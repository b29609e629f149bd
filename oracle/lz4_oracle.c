/*
 * CPU oracle, LZ4 part -- TEST INFRASTRUCTURE ONLY (see snappy_oracle.h).
 *
 * Restates what /root/reference/Snappy/Features/Mcdf/McdfManager.cs:73-79
 * obtains from `new LZ4Stream(fs, LZ4StreamMode.Decompress, ...).CopyTo(ms)`:
 * lz4net 1.0.15.93's chunked stream framing around raw LZ4 blocks.  lz4net is
 * a NuGet dependency (Snappy/Snappy.csproj:56) whose source is not in the
 * reference tree; the block rules below are the published LZ4 block format
 * with the acceptance rules of LZ4 r7x LZ4_uncompress (the decoder lz4net
 * ports), and the framing is restated from the lz4net LZ4Stream source as
 * recalled (PARITY UNPINNED for the framing, see snappy_oracle.h).
 */
#include "snappy_oracle.h"

#include <stdlib.h>
#include <string.h>

#define COPYLENGTH 8   /* r7x: a literal run ending in the last 8 bytes must end the block */
#define LASTLITERALS 5 /* encoder: the last 5 bytes are literals */
#define MFLIMIT 12     /* encoder: no match starts in the last 12 bytes */
#define MINMATCH 4

int oracle_lz4_block_decode(const uint8_t *in, uint64_t in_len, uint8_t *out, uint64_t out_len)
{
    uint64_t ip = 0, op = 0;
    for (;;) {
        if (ip >= in_len)
            return ORACLE_E_LZ4_CORRUPT;
        unsigned token = in[ip++];

        uint64_t length = token >> 4;
        if (length == 15) {
            unsigned b;
            do {
                if (ip >= in_len)
                    return ORACLE_E_LZ4_CORRUPT;
                b = in[ip++];
                length += b;
            } while (b == 255);
        }
        uint64_t cpy = op + length;
        if (cpy > out_len || ip + length > in_len)
            return ORACLE_E_LZ4_CORRUPT;
        if (cpy + COPYLENGTH > out_len) {
            /* Last sequence: literals only, and they must fill the output exactly. */
            if (cpy != out_len)
                return ORACLE_E_LZ4_CORRUPT;
            memcpy(out + op, in + ip, length);
            ip += length;
            op = cpy;
            break;
        }
        memcpy(out + op, in + ip, length);
        ip += length;
        op = cpy;

        if (ip + 2 > in_len)
            return ORACLE_E_LZ4_CORRUPT;
        uint64_t offset = (uint64_t)in[ip] | ((uint64_t)in[ip + 1] << 8);
        ip += 2;
        /* Offset 0 is undefined in the format (r7x would copy unwritten output);
         * an offset reaching before the block start is r7x's "ref < dest" error. */
        if (offset == 0 || offset > op)
            return ORACLE_E_LZ4_CORRUPT;

        uint64_t mlen = token & 15;
        if (mlen == 15) {
            unsigned b;
            do {
                if (ip >= in_len)
                    return ORACLE_E_LZ4_CORRUPT;
                b = in[ip++];
                mlen += b;
            } while (b == 255);
        }
        mlen += MINMATCH;
        cpy = op + mlen;
        /* LASTLITERALS: the last 5 bytes of a block are literals, so a match that reaches
         * into them is refused ("cpy > oend - LASTLITERALS" in the r7x decoder lz4net ports,
         * as recalled, and in today's liblz4).  No compressor emits such a match. */
        if (cpy + 5 > out_len)
            return ORACLE_E_LZ4_CORRUPT;
        /* Byte-serial copy: offset < mlen repeats the pattern. */
        for (uint64_t i = 0; i < mlen; ++i)
            out[op + i] = out[op - offset + i];
        op = cpy;
    }
    /* LZ4Codec.Decode(knownOutputLength): consumed input must equal the chunk's compressedLength. */
    return (ip == in_len && op == out_len) ? ORACLE_OK : ORACLE_E_LZ4_CORRUPT;
}

/* lz4net TryReadVarInt/ReadVarInt: base-128 little-endian, stops after 64 bits.
 * Returns 1 = value read, 0 = clean EOF before the first byte, -1 = EOF inside. */
static int read_varint(const uint8_t *in, uint64_t in_len, uint64_t *pos, uint64_t *value)
{
    uint64_t result = 0;
    unsigned count = 0;
    for (;;) {
        if (*pos >= in_len)
            return count == 0 ? 0 : -1;
        unsigned b = in[(*pos)++];
        result += (uint64_t)(b & 0x7F) << count;
        count += 7;
        if ((b & 0x80) == 0 || count >= 64)
            break;
    }
    *value = result;
    return 1;
}

typedef struct {
    int compressed;
    int passes_nonzero;
    uint64_t original_len;
    uint64_t compressed_len;
    uint64_t payload_pos;
} chunk_header;

/* lz4net LZ4Stream.AcquireNextChunk header part.
 * Returns 1 = header read, 0 = clean end of stream, <0 = error code. */
static int read_chunk_header(const uint8_t *in, uint64_t in_len, uint64_t *pos, chunk_header *h)
{
    uint64_t v;
    int r = read_varint(in, in_len, pos, &v);
    if (r == 0)
        return 0;
    if (r < 0)
        return ORACLE_E_LZ4_TRUNCATED;
    int32_t flags = (int32_t)(uint32_t)v; /* (ChunkFlags)varint */
    h->compressed = flags & 1;
    h->passes_nonzero = (flags >> 2) != 0;

    if (read_varint(in, in_len, pos, &v) != 1)
        return ORACLE_E_LZ4_TRUNCATED;
    int32_t original = (int32_t)(uint32_t)v; /* (int)ReadVarInt() */
    int32_t compressed = original;
    if (h->compressed) {
        if (read_varint(in, in_len, pos, &v) != 1)
            return ORACLE_E_LZ4_TRUNCATED;
        compressed = (int32_t)(uint32_t)v;
    }
    if (original < 0 || compressed < 0 || compressed > original)
        return ORACLE_E_LZ4_BAD_LENGTHS;
    h->original_len = (uint64_t)original;
    h->compressed_len = (uint64_t)compressed;
    h->payload_pos = *pos;
    if (*pos + h->compressed_len > in_len)
        return ORACLE_E_LZ4_TRUNCATED;
    *pos += h->compressed_len;
    return 1;
}

int oracle_lz4net_decoded_size(const uint8_t *in, uint64_t in_len, uint64_t *out_len)
{
    uint64_t pos = 0, total = 0;
    chunk_header h;
    int r;
    while ((r = read_chunk_header(in, in_len, &pos, &h)) == 1) {
        if (h.compressed && h.passes_nonzero)
            return ORACLE_E_LZ4_PASSES;
        total += h.original_len;
    }
    if (r < 0)
        return r;
    *out_len = total;
    return ORACLE_OK;
}

int oracle_lz4net_decode(const uint8_t *in, uint64_t in_len, uint8_t *out, uint64_t out_cap,
                         uint64_t *out_len)
{
    uint64_t pos = 0, total = 0;
    chunk_header h;
    int r;
    while ((r = read_chunk_header(in, in_len, &pos, &h)) == 1) {
        if (h.compressed && h.passes_nonzero)
            return ORACLE_E_LZ4_PASSES;
        if (total + h.original_len > out_cap)
            return ORACLE_E_CAPACITY;
        if (h.original_len == 0)
            continue; /* LZ4Codec.Decode returns 0 for an empty output; the reader skips the chunk */
        if (h.compressed) {
            int d = oracle_lz4_block_decode(in + h.payload_pos, h.compressed_len, out + total,
                                            h.original_len);
            if (d != ORACLE_OK)
                return d;
        } else {
            memcpy(out + total, in + h.payload_pos, h.original_len);
        }
        total += h.original_len;
    }
    if (r < 0)
        return r;
    *out_len = total;
    return ORACLE_OK;
}

/* ---------------------------------------------------------------------------
 * Test-data generators.
 * ------------------------------------------------------------------------- */

static uint32_t read32(const uint8_t *p)
{
    uint32_t v;
    memcpy(&v, p, 4);
    return v;
}

static uint64_t put_length(uint8_t *out, uint64_t o, uint64_t extra)
{
    while (extra >= 255) {
        out[o++] = 255;
        extra -= 255;
    }
    out[o++] = (uint8_t)extra;
    return o;
}

uint64_t oracle_lz4_block_encode(const uint8_t *in, uint64_t in_len, uint8_t *out,
                                 uint64_t out_cap)
{
    /* Worst case: all literals plus length bytes. */
    if (out_cap < in_len + in_len / 255 + 16)
        return 0;
    enum { HASH_BITS = 16 };
    uint32_t *table = (uint32_t *)calloc((size_t)1 << HASH_BITS, sizeof(uint32_t));
    if (!table)
        return 0;
    uint64_t o = 0, anchor = 0, ip = 0;
    if (in_len > MFLIMIT) {
        const uint64_t mflimit = in_len - MFLIMIT;
        const uint64_t matchlimit = in_len - LASTLITERALS;
        while (ip < mflimit) {
            uint32_t seq = read32(in + ip);
            uint32_t hsh = (seq * 2654435761u) >> (32 - HASH_BITS);
            uint64_t ref = table[hsh];
            table[hsh] = (uint32_t)(ip + 1);
            if (ref != 0 && ip - (ref - 1) <= 65535 && read32(in + ref - 1) == seq) {
                ref -= 1;
                uint64_t mlen = MINMATCH;
                while (ip + mlen < matchlimit && in[ref + mlen] == in[ip + mlen])
                    ++mlen;
                uint64_t lit = ip - anchor;
                uint8_t *token = out + o++;
                if (lit >= 15) {
                    *token = 15 << 4;
                    o = put_length(out, o, lit - 15);
                } else {
                    *token = (uint8_t)(lit << 4);
                }
                memcpy(out + o, in + anchor, lit);
                o += lit;
                uint64_t off = ip - ref;
                out[o++] = (uint8_t)off;
                out[o++] = (uint8_t)(off >> 8);
                uint64_t mcode = mlen - MINMATCH;
                if (mcode >= 15) {
                    *token |= 15;
                    o = put_length(out, o, mcode - 15);
                } else {
                    *token |= (uint8_t)mcode;
                }
                ip += mlen;
                anchor = ip;
            } else {
                ++ip;
            }
        }
    }
    uint64_t lit = in_len - anchor;
    uint8_t *token = out + o++;
    if (lit >= 15) {
        *token = 15 << 4;
        o = put_length(out, o, lit - 15);
    } else {
        *token = (uint8_t)(lit << 4);
    }
    memcpy(out + o, in + anchor, lit);
    o += lit;
    free(table);
    return o;
}

static uint64_t put_varint(uint8_t *out, uint64_t o, uint64_t v)
{
    while (v >= 0x80) {
        out[o++] = (uint8_t)(v | 0x80);
        v >>= 7;
    }
    out[o++] = (uint8_t)v;
    return o;
}

uint64_t oracle_lz4net_encode_bound(uint64_t in_len, uint64_t chunk_size)
{
    uint64_t chunks = chunk_size ? (in_len + chunk_size - 1) / chunk_size : 0;
    return in_len + chunks * 16 + 16;
}

/* lz4net LZ4Stream write side (FlushCurrentChunk): one chunk per chunk_size
 * input bytes; stored when the block does not come out smaller. */
uint64_t oracle_lz4net_encode(const uint8_t *in, uint64_t in_len, uint8_t *out,
                              uint64_t out_cap, uint64_t chunk_size, int hc_flag)
{
    if (chunk_size == 0 || out_cap < oracle_lz4net_encode_bound(in_len, chunk_size))
        return 0;
    uint64_t tmp_cap = chunk_size + chunk_size / 255 + 16;
    uint8_t *tmp = (uint8_t *)malloc(tmp_cap);
    if (!tmp)
        return 0;
    uint64_t o = 0;
    for (uint64_t pos = 0; pos < in_len; pos += chunk_size) {
        uint64_t n = in_len - pos < chunk_size ? in_len - pos : chunk_size;
        uint64_t c = oracle_lz4_block_encode(in + pos, n, tmp, tmp_cap);
        int compressed = (c > 0 && c < n);
        uint64_t flags = (compressed ? 1u : 0u) | (hc_flag ? 2u : 0u);
        o = put_varint(out, o, flags);
        o = put_varint(out, o, n);
        if (compressed) {
            o = put_varint(out, o, c);
            memcpy(out + o, tmp, c);
            o += c;
        } else {
            memcpy(out + o, in + pos, n);
            o += n;
        }
    }
    free(tmp);
    return o;
}
