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
// wins, the wave extends the match 64 bytes per step, emits "literals so far +
// match" cooperatively and continues behind it.  Without a match the window
// moves on by 64.
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

__device__ __forceinline__ uint32_t load32(const uint8_t *p)
{
    uint32_t v;
    __builtin_memcpy(&v, p, 4);
    return v;
}

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
__device__ __forceinline__ uint32_t put_literals(uint8_t *out, uint32_t o, const uint8_t *src,
                                                 uint32_t lit, uint32_t lane, uint32_t *token_at)
{
    *token_at = o;
    out[o] = static_cast<uint8_t>((lit >= 15 ? 15u : lit) << 4);
    ++o;
    if (lit >= 15)
        o = put_length(out, o, lit - 15, lane);
    for (uint32_t i = lane; i < lit; i += 64)
        out[o + i] = src[i];
    return o + lit;
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
        while (ip < mflimit) {
            const uint32_t p = ip + lane;
            const bool here = p < mflimit;
            const uint32_t seq = here ? load32(in + p) : 0u;
            const uint32_t h = (seq * 2654435761u) >> (32 - kHashBits);
            const uint32_t cand = here ? table[h] : kNoPos;
            if (here)
                table[h] = p;  // after every lane's read: LDS serves a wave's accesses in order
            const bool ok = cand != kNoPos && cand < p && p - cand <= 65535u && load32(in + cand) == seq;
            const uint64_t found = __ballot(ok);
            if (found == 0) {
                ip += 64;
                continue;
            }
            const uint32_t first = static_cast<uint32_t>(__builtin_ctzll(found));
            const uint32_t m = ip + first;
            const uint32_t ref = __builtin_amdgcn_readlane(cand, first);
            uint32_t mlen = kMinMatch;
            for (;;) {
                const uint32_t q = m + mlen + lane;
                const bool same = q < matchlimit && in[q] == in[ref + mlen + lane];
                const uint64_t differ = __ballot(!same);
                if (differ) {
                    mlen += static_cast<uint32_t>(__builtin_ctzll(differ));
                    break;
                }
                mlen += 64;
            }
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
