/* Plain-C consumer of libsnappyhash: proves that include/snappyhash.h is C (not
 * C++) and that the entry points link and behave through the C ABI alone.
 * Usage: driver            host-only entry points (no GPU needed)
 *        driver --gpu      also hashes and decodes on the GPU */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "snappyhash.h"

#define CHECK(cond)                                                        \
    do {                                                                   \
        if (!(cond)) {                                                     \
            fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); \
            return 1;                                                      \
        }                                                                  \
    } while (0)

int main(int argc, char **argv)
{
    /* Crypto.ToHashedString: uppercase hex, NUL-terminated. */
    const uint8_t digest[20] = {0xDA, 0x39, 0xA3, 0xEE, 0x5E, 0x6B, 0x4B, 0x0D, 0x32, 0x55,
                                0xBF, 0xEF, 0x95, 0x60, 0x18, 0x90, 0xAF, 0xD8, 0x07, 0x09};
    char hex[41];
    CHECK(snappy_digests_to_hex(digest, 1, hex) == SNAPPY_OK);
    CHECK(strcmp(hex, "DA39A3EE5E6B4B0D3255BFEF95601890AFD80709") == 0);
    CHECK(strcmp(snappy_strerror(SNAPPY_E_MCDF_MAGIC), "Not a Mare Chara File") == 0);

    /* Header writer -> header reader. */
    const char *paths[3] = {"chara/a.tex", "chara/b.tex", "chara/c.mdl"};
    const uint32_t per_file[2] = {2, 1};
    const int64_t lengths[2] = {1234, 0};
    uint8_t head[512];
    uint64_t head_len = 0;
    CHECK(snappy_mcdf_write_header("from C", "", "", "", paths, per_file, lengths, 2, head, sizeof head,
                                   &head_len) == SNAPPY_OK);
    CHECK(head_len > 9 && memcmp(head, "MCDF\x01", 5) == 0);
    uint64_t n_files = 0, body = 0;
    int64_t got[2] = {-1, -1};
    CHECK(snappy_mcdf_parse_header(head, head_len, &n_files, &body, got, 2) == SNAPPY_OK);
    CHECK(n_files == 2 && body == head_len && got[0] == 1234 && got[1] == 0);
    head[0] = 'X';
    CHECK(snappy_mcdf_parse_header(head, head_len, &n_files, &body, got, 2) == SNAPPY_E_MCDF_MAGIC);

    /* lz4net framing walk on a hand-made stored chunk: flags 0, length 3, "abc". */
    const uint8_t stored[5] = {0x00, 0x03, 'a', 'b', 'c'};
    uint64_t decoded = 0;
    CHECK(snappy_lz4net_decoded_size(stored, sizeof stored, &decoded) == SNAPPY_OK && decoded == 3);
    uint64_t cuts[3];
    const uint64_t sizes[4] = {10, 10, 10, 10};
    CHECK(snappy_plan_shards(sizes, 4, 2, cuts) == SNAPPY_OK && cuts[0] == 0 && cuts[1] == 2 && cuts[2] == 4);

    if (argc > 1 && strcmp(argv[1], "--gpu") == 0) {
        CHECK(snappy_init(0) == SNAPPY_OK);
        CHECK(snappy_sha1_hex((const uint8_t *)"abc", 3, hex) == SNAPPY_OK);
        CHECK(strcmp(hex, "A9993E364706816ABA3E25717850C26C9CD0D89D") == 0);
        uint8_t out[8];
        uint64_t out_len = 0;
        CHECK(snappy_lz4net_decode(stored, sizeof stored, out, sizeof out, &out_len) == SNAPPY_OK);
        CHECK(out_len == 3 && memcmp(out, "abc", 3) == 0);
        /* compress -> decompress */
        enum { N = 300000 };
        uint8_t *data = (uint8_t *)malloc(N), *back = (uint8_t *)malloc(N);
        for (int i = 0; i < N; ++i)
            data[i] = (uint8_t)((i / 7) * 13 + (i % 5));
        uint64_t bound = 0, packed_len = 0;
        CHECK(snappy_lz4net_encode_bound(N, 0, &bound) == SNAPPY_OK);
        uint8_t *packed = (uint8_t *)malloc(bound);
        CHECK(snappy_lz4net_encode(data, N, 0, packed, bound, &packed_len) == SNAPPY_OK && packed_len < N);
        CHECK(snappy_lz4net_decode(packed, packed_len, back, N, &out_len) == SNAPPY_OK);
        CHECK(out_len == N && memcmp(data, back, N) == 0);
        free(data);
        free(back);
        free(packed);
        snappy_shutdown();
        puts("c driver ok (gpu)");
        return 0;
    }
    puts("c driver ok (host)");
    return 0;
}
