// Headless host-side mirror of the reference's callers of the hot path: the
// hash-and-store loops and the MCDF import, file system included.
//
//   snappy_store_files        /root/reference/Snappy/Core/SnapshotManager.cs:312-324 (read + hash)
//                             and :453-475 (copy to _files/<HASH>.dat unless present);
//                             Snappy/Data/SnapshotMigrator.cs:62-86 is the same loop.
//   snappy_mcdf_import_to_dir Snappy/Features/Mcdf/McdfManager.cs:68-156 without the UI and
//                             IPC: decode + hash on the GPU, then the snapshot folder
//                             (_files/<HASH>.dat, snapshot.json, glamourer_history.json,
//                             customize_history.json as SnapshotManager.cs:672-695 writes them).
//
// All hashing and decoding goes through the same GPU entry points as the rest of
// the library; this file only moves bytes between the file system and them.
#include <chrono>
#include <cstdio>
#include <cstring>
#include <ctime>
#include <filesystem>
#include <fstream>
#include <map>
#include <string>
#include <system_error>
#include <vector>

#include "../../include/snappyhash.h"
#include "mcdf_host.hpp"
#include "runtime.hpp"

namespace fs = std::filesystem;

namespace snappy {

namespace {

bool read_all_bytes(const std::string &path, std::vector<uint8_t> &out)
{
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    if (!f)
        return false;
    const std::streamoff size = f.tellg();
    if (size < 0)
        return false;
    out.resize(static_cast<size_t>(size));
    f.seekg(0);
    if (size && !f.read(reinterpret_cast<char *>(out.data()), size))
        return false;
    return true;
}

bool write_all_bytes(const std::string &path, const uint8_t *data, size_t n)
{
    std::ofstream f(path, std::ios::binary | std::ios::trunc);
    if (!f)
        return false;
    if (n)
        f.write(reinterpret_cast<const char *>(data), static_cast<std::streamsize>(n));
    return static_cast<bool>(f);
}

// Key folding of StringComparer.OrdinalIgnoreCase, ASCII range (game paths are ASCII).
std::string fold_key(const std::string &s)
{
    std::string k = s;
    for (char &c : k)
        if (c >= 'a' && c <= 'z')
            c = static_cast<char>(c - 'a' + 'A');
    return k;
}

// Dictionary<string,string>(OrdinalIgnoreCase): first spelling of a key is kept,
// the last value wins, enumeration follows first insertion.
class IgnoreCaseMap {
public:
    void set(const std::string &key, const std::string &value)
    {
        const std::string k = fold_key(key);
        auto it = index_.find(k);
        if (it == index_.end()) {
            index_[k] = items_.size();
            items_.emplace_back(key, value);
        } else {
            items_[it->second].second = value;
        }
    }
    const std::vector<std::pair<std::string, std::string>> &items() const { return items_; }

private:
    std::map<std::string, size_t> index_;
    std::vector<std::pair<std::string, std::string>> items_;
};

// JsonConvert string escaping (default handling: quotes, backslash, control characters).
std::string json_string(const std::string &s)
{
    std::string o = "\"";
    for (unsigned char c : s) {
        switch (c) {
        case '"': o += "\\\""; break;
        case '\\': o += "\\\\"; break;
        case '\b': o += "\\b"; break;
        case '\f': o += "\\f"; break;
        case '\n': o += "\\n"; break;
        case '\r': o += "\\r"; break;
        case '\t': o += "\\t"; break;
        default:
            if (c < 0x20) {
                char buf[8];
                std::snprintf(buf, sizeof buf, "\\u%04x", c);
                o += buf;
            } else {
                o.push_back(static_cast<char>(c));
            }
        }
    }
    o += "\"";
    return o;
}

// DateTime.UtcNow.ToString("o"): 2025-08-15T12:34:56.1234567Z
std::string utc_now_roundtrip()
{
    using namespace std::chrono;
    const auto now = system_clock::now();
    const std::time_t t = system_clock::to_time_t(now);
    const auto ticks = duration_cast<duration<long long, std::ratio<1, 10000000>>>(
                           now.time_since_epoch()).count() % 10000000;
    std::tm tm{};
    gmtime_r(&t, &tm);
    char buf[64];
    std::snprintf(buf, sizeof buf, "%04d-%02d-%02dT%02d:%02d:%02d.%07lldZ", tm.tm_year + 1900,
                  tm.tm_mon + 1, tm.tm_mday, tm.tm_hour, tm.tm_min, tm.tm_sec, ticks);
    return buf;
}

std::string local_now_compact()  // DateTime.Now:yyyyMMddHHmmss
{
    const std::time_t t = std::time(nullptr);
    std::tm tm{};
    localtime_r(&t, &tm);
    char buf[32];
    std::snprintf(buf, sizeof buf, "%04d%02d%02d%02d%02d%02d", tm.tm_year + 1900, tm.tm_mon + 1,
                  tm.tm_mday, tm.tm_hour, tm.tm_min, tm.tm_sec);
    return buf;
}

// SnapshotInfo (Snappy/Models/SnapshotInfo.cs:6-14) as JsonConvert writes it with Formatting.Indented.
std::string snapshot_info_json(const std::string &source_actor, const std::string &last_update,
                               const IgnoreCaseMap &replacements, const std::string &manipulation)
{
    std::string o = "{\n  \"FormatVersion\": 1,\n  \"SourceActor\": " + json_string(source_actor) +
                    ",\n  \"LastUpdate\": " + json_string(last_update) + ",\n  \"FileReplacements\": {";
    const auto &items = replacements.items();
    for (size_t i = 0; i < items.size(); ++i) {
        o += (i ? ",\n    " : "\n    ");
        o += json_string(items[i].first) + ": " + json_string(items[i].second);
    }
    o += items.empty() ? "}" : "\n  }";
    o += ",\n  \"ManipulationString\": " + json_string(manipulation) + "\n}";
    return o;
}

std::string history_json(const std::vector<std::vector<std::pair<std::string, std::string>>> &entries)
{
    if (entries.empty())
        return "{\n  \"Entries\": []\n}";
    std::string o = "{\n  \"Entries\": [";
    for (size_t e = 0; e < entries.size(); ++e) {
        o += (e ? ",\n    {" : "\n    {");
        for (size_t k = 0; k < entries[e].size(); ++k) {
            o += (k ? ",\n      " : "\n      ");
            o += json_string(entries[e][k].first) + ": " + json_string(entries[e][k].second);
        }
        o += "\n    }";
    }
    o += "\n  ]\n}";
    return o;
}

bool write_text(const std::string &path, const std::string &text)
{
    return write_all_bytes(path, reinterpret_cast<const uint8_t *>(text.data()), text.size());
}

void hex40(const uint8_t *digest, char *out41)
{
    (void)snappy_digests_to_hex(digest, 1, out41);
}

}  // namespace
}  // namespace snappy

using namespace snappy;

extern "C" int snappy_store_files(const char *const *paths, uint64_t n, const char *files_dir,
                                  char *hex, uint8_t *status)
{
    if (n && (!paths || !hex || !status))
        return SNAPPY_E_INVALID_ARG;
    // File.Exists + File.ReadAllBytes per entry (SnapshotManager.cs:314-317).
    std::vector<std::vector<uint8_t>> blobs;
    std::vector<uint64_t> owner;
    blobs.reserve(n);
    for (uint64_t i = 0; i < n; ++i) {
        hex[41 * i] = '\0';
        status[i] = 1;  // source missing: `continue`
        if (!paths[i])
            continue;
        std::error_code ec;
        if (!fs::is_regular_file(paths[i], ec))
            continue;
        std::vector<uint8_t> bytes;
        if (!read_all_bytes(paths[i], bytes))
            continue;
        blobs.push_back(std::move(bytes));
        owner.push_back(i);
        status[i] = 0;
    }
    if (blobs.empty())
        return SNAPPY_OK;
    // Crypto.GetHash for all of them in one GPU batch (:318).
    std::vector<const uint8_t *> ptrs(blobs.size());
    std::vector<uint64_t> lens(blobs.size());
    for (size_t k = 0; k < blobs.size(); ++k) {
        ptrs[k] = blobs[k].empty() ? nullptr : blobs[k].data();
        lens[k] = blobs[k].size();
    }
    std::vector<uint8_t> digests(blobs.size() * 20);
    int st = snappy_sha1_files(ptrs.data(), lens.data(), blobs.size(), digests.data());
    if (st != SNAPPY_OK)
        return st;
    if (files_dir) {
        std::error_code ec;
        fs::create_directories(files_dir, ec);
    }
    for (size_t k = 0; k < blobs.size(); ++k) {
        char *h = hex + 41 * owner[k];
        hex40(digests.data() + 20 * k, h);
        if (!files_dir)
            continue;
        // `if (!File.Exists(hashedFilePath)) File.Copy(...)` (:457-466).
        const std::string target = (fs::path(files_dir) / (std::string(h) + ".dat")).string();
        std::error_code ec;
        if (!fs::exists(target, ec) && !write_all_bytes(target, blobs[k].data(), blobs[k].size())) {
            set_error_detail("cannot write " + target);
            return SNAPPY_E_INVALID_ARG;
        }
    }
    return SNAPPY_OK;
}

extern "C" int snappy_mcdf_import_to_dir(const char *mcdf_path, const char *working_dir,
                                         char *snapshot_dir_out, uint64_t out_cap)
{
    if (!mcdf_path || !working_dir)
        return SNAPPY_E_INVALID_ARG;
    std::vector<uint8_t> stream;
    if (!read_all_bytes(mcdf_path, stream)) {
        set_error_detail(std::string("cannot read ") + mcdf_path);
        return SNAPPY_E_INVALID_ARG;
    }
    // lz4Stream.CopyTo(memoryStream) + header + ExtractAndHashMapFiles, fused on the GPU.
    uint64_t decoded = 0;
    int st = snappy_lz4net_decoded_size(stream.data(), stream.size(), &decoded);
    if (st != SNAPPY_OK)
        decoded = 0;  // the fused call reports stream errors in stream order
    std::vector<uint8_t> payload(static_cast<size_t>(decoded));
    uint64_t n_files = 0, body = 0, payload_len = 0, cap = 1024;
    std::vector<uint8_t> digests, statuses;
    for (;;) {
        digests.assign(cap * 20, 0);
        statuses.assign(cap, 0);
        st = snappy_mcdf_decode_and_hash(stream.data(), stream.size(), digests.data(),
                                         statuses.data(), cap, &n_files, &body,
                                         payload.empty() ? nullptr : payload.data(), payload.size(),
                                         &payload_len);
        if (st == SNAPPY_E_CAPACITY && n_files > cap) {
            cap = n_files;
            continue;
        }
        break;
    }
    if (st != SNAPPY_OK)
        return st;
    McdfHeaderInfo info;
    st = parse_mcdf_header(payload.data(), payload_len, info);
    if (st != SNAPPY_OK)
        return st;

    // Snapshot folder (McdfManager.cs:93-108): named after the description, replaced when present.
    const std::string name =
        info.description.empty() ? "MCDF_Import_" + local_now_compact() : info.description;
    const fs::path snapshot = fs::path(working_dir) / name;
    const fs::path files = snapshot / "_files";
    std::error_code ec;
    if (fs::exists(snapshot, ec))
        fs::remove_all(snapshot, ec);
    fs::create_directories(files, ec);
    if (ec) {
        set_error_detail("cannot create " + files.string());
        return SNAPPY_E_INVALID_ARG;
    }

    // ExtractAndHashMapFiles (McdfManager.cs:36-63) with the digests already computed.
    std::vector<McdfSlice> slices;
    st = mcdf_slice_files(info.files, info.body_offset, payload_len, slices);
    if (st != SNAPPY_OK)
        return st;
    IgnoreCaseMap game_path_to_hash;
    for (size_t i = 0; i < slices.size(); ++i) {
        if (slices[i].status != SNAPPY_FILE_HASHED)
            continue;  // Length == 0, or a short read (logged and skipped by the reference)
        char h[41];
        hex40(digests.data() + 20 * i, h);
        const std::string target = (files / (std::string(h) + ".dat")).string();
        if (!fs::exists(target, ec) &&
            !write_all_bytes(target, payload.data() + slices[i].offset, slices[i].length)) {
            set_error_detail("cannot write " + target);
            return SNAPPY_E_INVALID_ARG;
        }
        for (const std::string &p : info.files[i].game_paths)
            game_path_to_hash.set(p, h);
    }

    // SaveSnapshotToDisk (SnapshotManager.cs:672-695).
    const std::string stamp = utc_now_roundtrip();
    const std::string info_json =
        snapshot_info_json(name, stamp, game_path_to_hash, info.manipulation_data);
    std::vector<std::vector<std::pair<std::string, std::string>>> glam = {
        {{"Timestamp", stamp}, {"Description", "Imported from MCDF"},
         {"GlamourerString", info.glamourer_data}}};
    std::vector<std::vector<std::pair<std::string, std::string>>> cust;
    if (!info.customize_plus_data.empty())
        // The importable Customize+ template (CustomizePlusUtil, outside the hot path) is left empty.
        cust.push_back({{"Timestamp", stamp}, {"Description", "Imported from MCDF"},
                        {"CustomizeData", info.customize_plus_data}, {"CustomizeTemplate", ""}});
    if (!write_text((snapshot / "snapshot.json").string(), info_json) ||
        !write_text((snapshot / "glamourer_history.json").string(), history_json(glam)) ||
        !write_text((snapshot / "customize_history.json").string(), history_json(cust))) {
        set_error_detail("cannot write snapshot JSON under " + snapshot.string());
        return SNAPPY_E_INVALID_ARG;
    }
    if (snapshot_dir_out) {
        const std::string s = snapshot.string();
        if (s.size() + 1 > out_cap)
            return SNAPPY_E_CAPACITY;
        std::memcpy(snapshot_dir_out, s.c_str(), s.size() + 1);
    }
    return SNAPPY_OK;
}
