// Sanitizer-built fuzz driver for the MCDF header reader (untrusted input: a decoded
// archive's first bytes).  Compiled with -fsanitize=address,undefined against
// snappy_amd/csrc/mcdf_host.cpp alone; mutates valid headers and random bytes and
// checks that the parser either accepts or rejects, and that what it accepts is
// consistent (body offset inside the payload, one length per file).
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "mcdf_host.hpp"
#include "snappyhash.h"

namespace snappy {
void set_error_detail(const std::string &) {}  // the library's thread-local error text is not under test
}

static std::vector<uint8_t> make_header(std::mt19937_64 &rng)
{
    std::string js = "{\"Description\":\"d\\u00e9 \\\"q\\\" \\\\\",\"GlamourerData\":\"Zm9v\",\"CustomizePlusData\":\"\","
                     "\"ManipulationData\":\"bWV0YQ==\",\"Files\":[";
    const int n = static_cast<int>(rng() % 6);
    for (int i = 0; i < n; ++i) {
        js += i ? ",{" : "{";
        js += "\"GamePaths\":[";
        const int p = static_cast<int>(rng() % 3);
        for (int k = 0; k < p; ++k)
            js += std::string(k ? "," : "") + "\"chara/x/" + std::to_string(rng() % 1000) + ".tex\"";
        js += "],\"Length\":" + std::to_string(rng() % 100000) + "}";
    }
    js += "]}";
    std::vector<uint8_t> out = {'M', 'C', 'D', 'F', 1};
    const uint32_t len = static_cast<uint32_t>(js.size());
    for (int b = 0; b < 4; ++b)
        out.push_back(static_cast<uint8_t>(len >> (8 * b)));
    out.insert(out.end(), js.begin(), js.end());
    for (int i = 0; i < 40; ++i)
        out.push_back(static_cast<uint8_t>(rng()));
    return out;
}

int main(int argc, char **argv)
{
    const long rounds = argc > 1 ? std::atol(argv[1]) : 200000;
    std::mt19937_64 rng(12345);
    long accepted = 0, rejected = 0;
    for (long r = 0; r < rounds; ++r) {
        std::vector<uint8_t> buf;
        const unsigned kind = static_cast<unsigned>(rng() % 8);
        if (kind == 0) {
            buf.resize(rng() % 200);
            for (auto &b : buf)
                b = static_cast<uint8_t>(rng());
        } else {
            buf = make_header(rng);
            const int edits = kind == 1 ? 0 : static_cast<int>(rng() % 4) + 1;
            for (int e = 0; e < edits && !buf.empty(); ++e) {
                const size_t at = rng() % buf.size();
                switch (rng() % 4) {
                case 0: buf[at] = static_cast<uint8_t>(rng()); break;
                case 1: buf.erase(buf.begin() + static_cast<long>(at)); break;
                case 2: buf.insert(buf.begin() + static_cast<long>(at), static_cast<uint8_t>(rng())); break;
                default: buf.resize(at); break;
                }
            }
        }
        // Exactly-sized heap copy, so that any read past the end is caught.
        std::vector<uint8_t> exact(buf);
        exact.shrink_to_fit();
        snappy::McdfHeaderInfo info;
        const int st = snappy::parse_mcdf_header(exact.data(), exact.size(), info);
        if (st == SNAPPY_OK) {
            ++accepted;
            if (info.body_offset > exact.size()) {
                std::fprintf(stderr, "accepted header with body offset past the payload\n");
                return 1;
            }
            std::vector<snappy::McdfSlice> slices;
            if (snappy::mcdf_slice_files(info.files, info.body_offset, exact.size(), slices) == SNAPPY_OK &&
                slices.size() != info.files.size()) {
                std::fprintf(stderr, "slice count differs from file count\n");
                return 1;
            }
        } else {
            ++rejected;
        }
    }
    std::printf("fuzz ok: %ld accepted, %ld rejected\n", accepted, rejected);
    return 0;
}
