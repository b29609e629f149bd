// MCDF container header (host side).
//
// Layout written by /root/reference/Snappy/Features/Mcdf/McdfHeader.cs:14-24 and
// read by :26-62: bytes "MCDF", u8 version (== 1), i32 little-endian JSON
// length, UTF-8 JSON of McdfData (McdfData.cs:9-13,25), then the file bodies
// back to back in Files order.
#pragma once

#include <stdint.h>

#include <string>
#include <utility>
#include <vector>

namespace snappy {

// The small JSON document model the header parser uses (strict RFC 8259; at most
// 64 levels and 2^20 values).  Also read by the snapshot-folder code, which takes
// an existing snapshot.json and the two history files back in.
struct JsonValue {
    enum Kind { Null, Bool, Number, String, Array, Object } kind = Null;
    bool boolean = false;
    std::string text;  // String: decoded UTF-8; Number: the literal
    std::vector<JsonValue> items;
    std::vector<std::pair<std::string, JsonValue>> members;
};
bool parse_json(const char *text, size_t n, JsonValue &out);
// Property binding the way the managed deserializer does it: exact name first,
// then case-insensitive; the last duplicate wins.  Null when absent.
const JsonValue *json_member(const JsonValue &obj, const char *name);

struct McdfFileEntry {
    std::vector<std::string> game_paths;  // FileData.GamePaths
    int64_t length = 0;                   // FileData.Length
};

struct McdfHeaderInfo {
    uint8_t version = 0;
    std::string description;         // McdfData.Description
    std::string glamourer_data;      // McdfData.GlamourerData
    std::string customize_plus_data; // McdfData.CustomizePlusData
    std::string manipulation_data;   // McdfData.ManipulationData
    std::vector<McdfFileEntry> files;
    uint64_t body_offset = 0;        // first file body in the payload
};

// Bytes of the payload needed to parse the header: 9 to learn the JSON length,
// then 9 + that length.  Returns SNAPPY_OK and sets *need, or an MCDF error.
int mcdf_header_extent(const uint8_t *payload, uint64_t avail, uint64_t payload_len,
                       uint64_t *need);

// McdfHeader.FromBinaryReader: parses payload[0..avail) (avail >= extent).
int parse_mcdf_header(const uint8_t *payload, uint64_t avail, McdfHeaderInfo &out);

// The slice loop of McdfManager.cs:36-49 without the hashing: for every Files
// entry, either the (offset, length) of its body or the reason it is skipped.
struct McdfSlice {
    uint64_t offset = 0;
    uint64_t length = 0;
    uint8_t status = 0;  // SNAPPY_FILE_*
};
int mcdf_slice_files(const std::vector<McdfFileEntry> &files, uint64_t body_offset,
                     uint64_t payload_len, std::vector<McdfSlice> &slices);

}  // namespace snappy
