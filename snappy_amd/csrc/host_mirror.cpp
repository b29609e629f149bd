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
#include <fcntl.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <ctime>
#include <filesystem>
#include <fstream>
#include <map>
#include <mutex>
#include <string>
#include <system_error>
#include <thread>
#include <vector>

#include "../../include/snappyhash.h"
#include "mcdf_host.hpp"
#include "runtime.hpp"
#include "sha1_api.hpp"

namespace fs = std::filesystem;

namespace snappy {

namespace {

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

namespace snappy {
namespace {

// Process-wide page-locked arena for file contents on their way to the GPU
// (grow-only; a second concurrent caller gets a temporary one of its own).
struct FileArena {
    std::mutex mu;
    bool busy = false;
    void *ptr = nullptr;
    uint64_t cap = 0;
};
FileArena g_file_arena;

class ArenaLease {
public:
    ~ArenaLease()
    {
        if (shared_) {
            std::lock_guard<std::mutex> lk(g_file_arena.mu);
            g_file_arena.busy = false;
        } else if (ptr_) {
            (void)snappy_host_free(ptr_);
        }
    }
    int acquire(uint64_t bytes)
    {
        {
            std::lock_guard<std::mutex> lk(g_file_arena.mu);
            if (!g_file_arena.busy) {
                if (g_file_arena.cap < bytes) {
                    if (g_file_arena.ptr)
                        (void)snappy_host_free(g_file_arena.ptr);
                    g_file_arena.ptr = nullptr;
                    g_file_arena.cap = 0;
                    const uint64_t want = bytes + bytes / 4;
                    int st = snappy_host_alloc(want, &g_file_arena.ptr);
                    if (st != SNAPPY_OK)
                        return st;
                    g_file_arena.cap = want;
                }
                g_file_arena.busy = true;
                shared_ = true;
                ptr_ = g_file_arena.ptr;
                return SNAPPY_OK;
            }
        }
        own_cap_ = bytes;
        return snappy_host_alloc(bytes, &ptr_);
    }
    // Makes room for `bytes`, keeping the first `keep` bytes.
    int grow(uint64_t bytes, uint64_t keep)
    {
        if (shared_) {
            std::lock_guard<std::mutex> lk(g_file_arena.mu);
            if (g_file_arena.cap >= bytes)
                return SNAPPY_OK;
        } else if (own_cap_ >= bytes) {
            return SNAPPY_OK;
        }
        void *bigger = nullptr;
        const uint64_t want = bytes + bytes / 4;
        int st = snappy_host_alloc(want, &bigger);
        if (st != SNAPPY_OK)
            return st;
        if (keep)
            std::memcpy(bigger, ptr_, static_cast<size_t>(keep));
        (void)snappy_host_free(ptr_);
        ptr_ = bigger;
        if (shared_) {
            std::lock_guard<std::mutex> lk(g_file_arena.mu);
            g_file_arena.ptr = bigger;
            g_file_arena.cap = want;
        } else {
            own_cap_ = want;
        }
        return SNAPPY_OK;
    }
    uint8_t *data() const { return static_cast<uint8_t *>(ptr_); }

private:
    void *ptr_ = nullptr;
    bool shared_ = false;
    uint64_t own_cap_ = 0;
};

// Runs fn(k) for k in [0, n) on up to kIoThreads threads (file reads and writes
// are latency-bound, so several are kept in flight).
constexpr unsigned kIoThreads = 8;

template <typename Fn>
void parallel_for(uint64_t n, Fn fn)
{
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    const unsigned threads = static_cast<unsigned>(std::min<uint64_t>(n, std::min(hw, kIoThreads)));
    if (threads <= 1) {
        for (uint64_t k = 0; k < n; ++k)
            fn(k);
        return;
    }
    std::atomic<uint64_t> next{0};
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < threads; ++t)
        pool.emplace_back([&] {
            for (uint64_t k = next.fetch_add(1); k < n; k = next.fetch_add(1))
                fn(k);
        });
    for (auto &t : pool)
        t.join();
}

// Reads exactly `size` bytes of the file into dst; false if it is shorter or unreadable.
bool read_exact(const char *path, uint8_t *dst, uint64_t size)
{
    const int fd = ::open(path, O_RDONLY | O_CLOEXEC);
    if (fd < 0)
        return false;
    uint64_t done = 0;
    while (done < size) {
        const ssize_t r = ::pread(fd, dst + done, static_cast<size_t>(std::min<uint64_t>(size - done, 1u << 30)),
                                  static_cast<off_t>(done));
        if (r <= 0)
            break;
        done += static_cast<uint64_t>(r);
    }
    // A file that grew since it was measured is read to its new end by ReadAllBytes;
    // keep it simple and exact: such a file is reported as unreadable.
    uint8_t probe;
    const bool grew = done == size && ::pread(fd, &probe, 1, static_cast<off_t>(size)) == 1;
    ::close(fd);
    return done == size && !grew;
}

// The same with several threads, each reading a range of the file: one thread does
// not reach the page cache's bandwidth, let alone a fast disk's.
bool read_exact_parallel(const char *path, uint8_t *dst, uint64_t size)
{
    constexpr uint64_t kPart = 8ull << 20;
    if (size <= 2 * kPart)
        return read_exact(path, dst, size);
    const int fd = ::open(path, O_RDONLY | O_CLOEXEC);
    if (fd < 0)
        return false;
    std::atomic<bool> ok{true};
    parallel_for((size + kPart - 1) / kPart, [&](uint64_t k) {
        uint64_t done = k * kPart;
        const uint64_t end = std::min<uint64_t>(size, done + kPart);
        while (done < end) {
            const ssize_t r = ::pread(fd, dst + done, static_cast<size_t>(end - done), static_cast<off_t>(done));
            if (r <= 0) {
                ok.store(false);
                return;
            }
            done += static_cast<uint64_t>(r);
        }
    });
    ::close(fd);
    return ok.load();
}

}  // namespace
}  // namespace snappy

// The hash-and-store loop with the file system taken off the critical path: the
// sources are read by several threads straight into page-locked memory, hashed
// in one batch whose transfer overlaps the kernel, and the new <HASH>.dat files
// are written by several threads.  Per entry the outcome is what the reference's
// sequential loop produces.
namespace {

// Files being read into one arena by other threads: a transfer out of the arena
// waits until the files it is about to send have been read.
class ReadGate : public snappy::SourceGate {
public:
    ReadGate(const uint8_t *arena, const std::vector<uint64_t> &offsets, const std::vector<uint64_t> &sizes)
        : arena_(arena), offsets_(offsets), sizes_(sizes), state_(offsets.size(), 0)
    {
    }

    void finished(uint64_t k, bool ok)
    {
        {
            std::lock_guard<std::mutex> lk(mu_);
            state_[k] = ok ? 1 : 2;
        }
        cv_.notify_all();
    }

    bool readable(uint64_t k) const { return state_[k] == 1; }

    void wait_for(const uint8_t *src, uint64_t len) override
    {
        if (len == 0 || src < arena_)
            return;
        const uint64_t lo = static_cast<uint64_t>(src - arena_), hi = lo + len;
        // Offsets ascend: the first file that ends behind lo, up to the last that starts before hi.
        size_t k = static_cast<size_t>(std::upper_bound(offsets_.begin(), offsets_.end(), lo) - offsets_.begin());
        k = k ? k - 1 : 0;
        std::unique_lock<std::mutex> lk(mu_);
        for (; k < offsets_.size() && offsets_[k] < hi; ++k)
            if (offsets_[k] + sizes_[k] > lo)
                cv_.wait(lk, [&] { return state_[k] != 0; });
    }

private:
    const uint8_t *arena_;
    const std::vector<uint64_t> &offsets_, &sizes_;
    std::vector<uint8_t> state_;  // 0 being read, 1 read, 2 unreadable
    std::mutex mu_;
    std::condition_variable cv_;
};

}  // namespace

extern "C" int snappy_store_files(const char *const *paths, uint64_t n, const char *files_dir,
                                  char *hex, uint8_t *status)
{
    if (n && (!paths || !hex || !status))
        return SNAPPY_E_INVALID_ARG;
    // File.Exists per entry (SnapshotManager.cs:314-315): missing sources are skipped.
    std::vector<uint64_t> owner, sizes, offs;
    uint64_t total = 0;
    for (uint64_t i = 0; i < n; ++i) {
        hex[41 * i] = '\0';
        status[i] = 1;
        if (!paths[i])
            continue;
        std::error_code ec;
        if (!fs::is_regular_file(paths[i], ec))
            continue;
        const uint64_t size = static_cast<uint64_t>(fs::file_size(paths[i], ec));
        if (ec)
            continue;
        owner.push_back(i);
        sizes.push_back(size);
        offs.push_back(total);
        total = (total + size + 255) & ~uint64_t(255);
    }
    if (owner.empty())
        return SNAPPY_OK;
    ArenaLease arena;
    int st = arena.acquire(total + 256);
    if (st != SNAPPY_OK)
        return st;
    // File.ReadAllBytes (:317), many files in flight, largest first: that is the order
    // in which the batch below sends them to the GPU, and it starts sending (and
    // hashing) while the rest is still being read.
    std::vector<uint64_t> by_size(owner.size());
    for (size_t k = 0; k < owner.size(); ++k)
        by_size[k] = k;
    std::stable_sort(by_size.begin(), by_size.end(),
                     [&](uint64_t x, uint64_t y) { return sizes[x] > sizes[y]; });
    ReadGate gate(arena.data(), offs, sizes);
    std::thread reader([&] {
        parallel_for(owner.size(), [&](uint64_t j) {
            const uint64_t k = by_size[j];
            gate.finished(k, read_exact(paths[owner[k]], arena.data() + offs[k], sizes[k]));
        });
    });
    // Crypto.GetHash for all of them in one GPU batch (:318).
    std::vector<uint8_t> all_digests(owner.size() * 20);
    st = sha1_batch_gated(arena.data(), offs.data(), sizes.data(), owner.size(), all_digests.data(), &gate);
    reader.join();
    if (st != SNAPPY_OK)
        return st;
    // A file that could not be read in full is skipped like a missing one (its digest is of no use).
    std::vector<uint64_t> b_off, b_len, b_owner;
    std::vector<uint8_t> digests;
    for (size_t k = 0; k < owner.size(); ++k)
        if (gate.readable(k)) {
            b_off.push_back(offs[k]);
            b_len.push_back(sizes[k]);
            b_owner.push_back(owner[k]);
            digests.insert(digests.end(), all_digests.begin() + 20 * k, all_digests.begin() + 20 * (k + 1));
            status[owner[k]] = 0;
        }
    if (b_owner.empty())
        return SNAPPY_OK;
    // `if (!File.Exists(hashedFilePath)) File.Copy(...)` (:457-466).  Entries with the
    // same digest have the same bytes: the first of them writes the file.
    std::map<std::string, size_t> first_with_hash;
    std::vector<size_t> writers;
    for (size_t k = 0; k < b_owner.size(); ++k) {
        char *h = hex + 41 * b_owner[k];
        hex40(digests.data() + 20 * k, h);
        if (files_dir && first_with_hash.emplace(std::string(h), k).second)
            writers.push_back(k);
    }
    if (!files_dir)
        return SNAPPY_OK;
    std::error_code ec;
    fs::create_directories(files_dir, ec);
    std::atomic<bool> write_failed{false};
    std::mutex fail_mu;
    std::string failed_target;
    parallel_for(writers.size(), [&](uint64_t w) {
        const size_t k = writers[w];
        const std::string target =
            (fs::path(files_dir) / (std::string(hex + 41 * b_owner[k]) + ".dat")).string();
        std::error_code e2;
        if (!fs::exists(target, e2) &&
            !write_all_bytes(target, arena.data() + b_off[k], static_cast<size_t>(b_len[k]))) {
            write_failed.store(true);
            std::lock_guard<std::mutex> lk(fail_mu);
            failed_target = target;
        }
    });
    if (write_failed.load()) {
        set_error_detail("cannot write " + failed_target);
        return SNAPPY_E_INVALID_ARG;
    }
    return SNAPPY_OK;
}

// McdfHeader.WriteToStream (McdfHeader.cs:14-24) over McdfData.ToByteArray
// (McdfData.cs:15-18, JsonConvert.SerializeObject without formatting: properties
// in declaration order, no spaces).
extern "C" int snappy_mcdf_write_header(const char *description, const char *glamourer_data,
                                        const char *customize_plus_data,
                                        const char *manipulation_data,
                                        const char *const *game_paths,
                                        const uint32_t *paths_per_file, const int64_t *lengths,
                                        uint64_t n_files, uint8_t *out, uint64_t out_cap,
                                        uint64_t *out_len)
{
    if (!out_len || (!out && out_cap) || (n_files && (!paths_per_file || !lengths)))
        return SNAPPY_E_INVALID_ARG;
    auto text = [](const char *s) { return json_string(s ? s : ""); };
    std::string js = "{\"Description\":" + text(description) + ",\"GlamourerData\":" +
                     text(glamourer_data) + ",\"CustomizePlusData\":" + text(customize_plus_data) +
                     ",\"ManipulationData\":" + text(manipulation_data) + ",\"Files\":[";
    uint64_t next_path = 0;
    for (uint64_t i = 0; i < n_files; ++i) {
        js += i ? ",{\"GamePaths\":[" : "{\"GamePaths\":[";
        for (uint32_t k = 0; k < paths_per_file[i]; ++k) {
            if (!game_paths || !game_paths[next_path])
                return SNAPPY_E_INVALID_ARG;
            js += (k ? "," : "") + json_string(game_paths[next_path++]);
        }
        js += "],\"Length\":" + std::to_string(static_cast<long long>(lengths[i])) + "}";
    }
    js += "]}";
    if (js.size() > 0x7FFFFFFFull)
        return SNAPPY_E_INVALID_ARG;  // BinaryWriter.Write(int)
    *out_len = 9 + js.size();
    if (*out_len > out_cap)
        return SNAPPY_E_CAPACITY;
    std::memcpy(out, "MCDF", 4);
    out[4] = 1;  // McdfHeader.CurrentVersion
    const uint32_t n = static_cast<uint32_t>(js.size());
    out[5] = static_cast<uint8_t>(n);
    out[6] = static_cast<uint8_t>(n >> 8);
    out[7] = static_cast<uint8_t>(n >> 16);
    out[8] = static_cast<uint8_t>(n >> 24);
    std::memcpy(out + 9, js.data(), js.size());
    return SNAPPY_OK;
}

extern "C" int snappy_mcdf_import_to_dir(const char *mcdf_path, const char *working_dir,
                                         char *snapshot_dir_out, uint64_t out_cap)
{
    if (!mcdf_path || !working_dir)
        return SNAPPY_E_INVALID_ARG;
    // The file and the decoded payload share one page-locked arena: both cross PCIe by
    // direct DMA, and neither costs a fresh allocation's page faults.
    std::error_code size_ec;
    const uint64_t stream_len = static_cast<uint64_t>(fs::file_size(mcdf_path, size_ec));
    if (size_ec) {
        set_error_detail(std::string("cannot read ") + mcdf_path);
        return SNAPPY_E_INVALID_ARG;
    }
    const uint64_t payload_at = (stream_len + 511) & ~uint64_t(255);
    ArenaLease arena;
    int st = arena.acquire(payload_at + 2 * stream_len + 256);
    if (st != SNAPPY_OK)
        return st;
    if (!read_exact_parallel(mcdf_path, arena.data(), stream_len)) {
        set_error_detail(std::string("cannot read ") + mcdf_path);
        return SNAPPY_E_INVALID_ARG;
    }
    // lz4Stream.CopyTo(memoryStream) + header + ExtractAndHashMapFiles, fused on the GPU.
    uint64_t decoded = 0;
    st = snappy_lz4net_decoded_size(arena.data(), stream_len, &decoded);
    if (st != SNAPPY_OK)
        decoded = 0;  // the fused call reports stream errors in stream order
    if ((st = arena.grow(payload_at + decoded + 256, stream_len)) != SNAPPY_OK)
        return st;
    const uint8_t *stream = arena.data();
    uint8_t *payload = arena.data() + payload_at;
    uint64_t n_files = 0, body = 0, payload_len = 0, cap = 1024;
    std::vector<uint8_t> digests, statuses;
    for (;;) {
        digests.assign(cap * 20, 0);
        statuses.assign(cap, 0);
        st = snappy_mcdf_decode_and_hash(stream, stream_len, digests.data(), statuses.data(), cap, &n_files,
                                         &body, decoded ? payload : nullptr, decoded, &payload_len);
        if (st == SNAPPY_E_CAPACITY && n_files > cap) {
            cap = n_files;
            continue;
        }
        break;
    }
    if (st != SNAPPY_OK)
        return st;
    McdfHeaderInfo info;
    st = parse_mcdf_header(payload, payload_len, info);
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
    // The map is filled in file order (last writer wins, McdfManager.cs:59-62); the file
    // bodies go to disk from several threads, the first entry with a given hash writing it
    // (the reference's File.Exists guard makes later ones no-ops).
    IgnoreCaseMap game_path_to_hash;
    std::vector<std::string> hashes(slices.size());
    std::map<std::string, size_t> first_with_hash;
    std::vector<size_t> writers;
    for (size_t i = 0; i < slices.size(); ++i) {
        if (slices[i].status != SNAPPY_FILE_HASHED)
            continue;  // Length == 0, or a short read (logged and skipped by the reference)
        char h[41];
        hex40(digests.data() + 20 * i, h);
        hashes[i] = h;
        if (first_with_hash.emplace(hashes[i], i).second)
            writers.push_back(i);
        for (const std::string &p : info.files[i].game_paths)
            game_path_to_hash.set(p, h);
    }
    std::atomic<bool> write_failed{false};
    std::mutex fail_mu;
    std::string failed_target;
    parallel_for(writers.size(), [&](uint64_t w) {
        const size_t i = writers[w];
        const std::string target = (files / (hashes[i] + ".dat")).string();
        std::error_code e2;
        if (!fs::exists(target, e2) &&
            !write_all_bytes(target, payload + slices[i].offset, slices[i].length)) {
            write_failed.store(true);
            std::lock_guard<std::mutex> lk(fail_mu);
            failed_target = target;
        }
    });
    if (write_failed.load()) {
        set_error_detail("cannot write " + failed_target);
        return SNAPPY_E_INVALID_ARG;
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
