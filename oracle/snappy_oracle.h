/*
 * CPU oracle for the Snappy asset hot path -- TEST INFRASTRUCTURE ONLY.
 *
 * This is a plain-C restatement of the arithmetic the reference performs when
 * it hashes mod asset files and imports MCDF archives.  Only tests/, the smoke
 * check and the cpu_baseline leg of bench.py may use it, and only as the
 * checker; the shipped library (libsnappyhash.so) never links or calls it.
 *
 * What it follows (paths relative to /root/reference):
 *   Snappy/Utils/Crypto.cs:8-12      GetHash      = SHA-1 of a whole buffer
 *   Snappy/Utils/Crypto.cs:14-22     ToHashedString = 40 UPPERCASE hex chars
 *   Snappy/Features/Mcdf/McdfManager.cs:72-80  LZ4Stream(...Decompress...).CopyTo
 *   Snappy/Features/Mcdf/McdfManager.cs:36-63  slice / skip / short-read loop
 *   Snappy/Features/Mcdf/McdfHeader.cs:26-40   magic, version, i32 length
 *
 * Pinning status.  The reference has no tests, fixtures or golden vectors, and
 * it cannot be built in this image (C#, no dotnet).  The arithmetic lives in
 * dependencies that are not in the tree:
 *   - SHA-1: .NET System.Security.Cryptography.SHA1 (FIPS 180-4).  Pinned here
 *     by the FIPS 180-4 known answers and by digests produced with Python's
 *     hashlib in the build container (tests/golden/, generator committed).
 *   - LZ4 block format: lz4net 1.0.15.93 (NuGet, Snappy/Snappy.csproj:56), a
 *     port of LZ4 r7x LZ4_uncompress.  The block decoder is pinned by
 *     hand-assembled vectors and by cross-decoding with the system liblz4
 *     1.9.3 in the build container (fixtures committed).
 *   - lz4net LZ4Stream chunk framing: restated from the published lz4net
 *     source as recalled; no copy of it is available offline, so for the
 *     FRAMING alone: PARITY UNPINNED.
 *
 * The assumptions nothing here can verify, in one place:
 *   1. The chunk framing (varint flags, original length, compressed length when
 *      flag bit 0 is set; passes in bits 2..; compressedLength > originalLength is
 *      an error; a clean end only at a chunk boundary) is lz4net's as recalled.
 *   2. The block decoder's acceptance of malformed input follows LZ4 r7x
 *      LZ4_uncompress as recalled: a literal run ending within 8 bytes of the end
 *      must end the block; a match that reaches into the last 5 bytes of the
 *      block is refused (LASTLITERALS, as in today's liblz4).  Valid compressor
 *      output never meets either rule.
 *   3. The MCDF header is read with a strict JSON reader.  Newtonsoft also takes
 *      comments, single-quoted strings and unquoted names, and coerces a number or
 *      bool given for a string property; those inputs are refused here
 *      (E_MCDF_JSON).  A Files entry without GamePaths fails the header up front
 *      (the reference fails when that file is reached); a JSON length beyond the
 *      payload is E_MCDF_TRUNCATED (the reference's short ReadBytes then fails as
 *      a JSON error).  Well-formed archives are unaffected.
 */
#ifndef SNAPPY_ORACLE_H
#define SNAPPY_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Same numeric values as include/snappyhash.h so tests can compare statuses. */
#define ORACLE_OK                 0
#define ORACLE_E_INVALID_ARG     (-1)
#define ORACLE_E_LZ4_TRUNCATED   (-10)
#define ORACLE_E_LZ4_BAD_LENGTHS (-11)
#define ORACLE_E_LZ4_PASSES      (-12)
#define ORACLE_E_LZ4_CORRUPT     (-13)
#define ORACLE_E_CAPACITY        (-14)

/* Crypto.cs:8-12 -- SHA-1 of data[0..len). */
void oracle_sha1(const uint8_t *data, uint64_t len, uint8_t out20[20]);
/* Crypto.cs:8-22 -- GetHash: 40 uppercase hex chars + NUL. */
void oracle_sha1_hex(const uint8_t *data, uint64_t len, char out41[41]);
/* The three hash loops (SnapshotManager.cs:312-324, McdfManager.cs:36-63,
 * SnapshotMigrator.cs:62-86) collapsed: digest i = SHA-1(base+offsets[i], lengths[i]). */
void oracle_sha1_batch(const uint8_t *base, const uint64_t *offsets,
                       const uint64_t *lengths, uint64_t n, uint8_t *digests);

/* One raw LZ4 block with known output length (lz4net LZ4Codec.Decode with
 * knownOutputLength=true, i.e. LZ4 r7x LZ4_uncompress acceptance rules).
 * Returns ORACLE_OK when exactly in_len input bytes produce exactly out_len
 * output bytes, ORACLE_E_LZ4_CORRUPT otherwise. */
int oracle_lz4_block_decode(const uint8_t *in, uint64_t in_len,
                            uint8_t *out, uint64_t out_len);

/* lz4net LZ4Stream read side (McdfManager.cs:73-79): walk chunk headers. */
int oracle_lz4net_decoded_size(const uint8_t *in, uint64_t in_len, uint64_t *out_len);
/* lz4net LZ4Stream.CopyTo(MemoryStream): decode every chunk back to back. */
int oracle_lz4net_decode(const uint8_t *in, uint64_t in_len, uint8_t *out,
                         uint64_t out_cap, uint64_t *out_len);

/* Test-data generators (the reference never compresses).  Greedy single-probe
 * hash-table LZ4 block encoder that honours the format's end rules, and the
 * lz4net chunk writer (LZ4Stream write side: stored chunk when compression
 * does not shrink it). Return bytes written, 0 when out_cap is too small. */
uint64_t oracle_lz4_block_encode(const uint8_t *in, uint64_t in_len,
                                 uint8_t *out, uint64_t out_cap);
uint64_t oracle_lz4net_encode(const uint8_t *in, uint64_t in_len, uint8_t *out,
                              uint64_t out_cap, uint64_t chunk_size, int hc_flag);
uint64_t oracle_lz4net_encode_bound(uint64_t in_len, uint64_t chunk_size);

/* Deterministic synthetic data: splitmix64 stream seeded per file. */
void oracle_fill_splitmix(uint8_t *dst, uint64_t len, uint64_t seed);

#ifdef __cplusplus
}
#endif
#endif
