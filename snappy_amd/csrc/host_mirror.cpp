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
#include <exception>
#include <filesystem>
#include <fstream>
#include <map>
#include <mutex>
#include <string>
#include <system_error>
#include <thread>
#include <vector>

#include "unicode_upper_table.hpp"
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

// Simple upper-case mapping of one code point (unicode_upper_table.hpp; ASCII done here).
uint32_t simple_upper(uint32_t cp)
{
    if (cp < 0x80u)
        return (cp >= 'a' && cp <= 'z') ? cp - 'a' + 'A' : cp;
    const UpperPair *lo = kSimpleUpper, *hi = kSimpleUpper + sizeof(kSimpleUpper) / sizeof(kSimpleUpper[0]);
    while (lo < hi) {
        const UpperPair *mid = lo + (hi - lo) / 2;
        if (mid->from < cp)
            lo = mid + 1;
        else
            hi = mid;
    }
    const UpperPair *end = kSimpleUpper + sizeof(kSimpleUpper) / sizeof(kSimpleUpper[0]);
    return (lo < end && lo->from == cp) ? lo->to : cp;
}

void append_utf8(std::string &out, uint32_t cp)
{
    if (cp < 0x80u) {
        out.push_back(static_cast<char>(cp));
    } else if (cp < 0x800u) {
        out.push_back(static_cast<char>(0xC0u | (cp >> 6)));
        out.push_back(static_cast<char>(0x80u | (cp & 0x3Fu)));
    } else if (cp < 0x10000u) {
        out.push_back(static_cast<char>(0xE0u | (cp >> 12)));
        out.push_back(static_cast<char>(0x80u | ((cp >> 6) & 0x3Fu)));
        out.push_back(static_cast<char>(0x80u | (cp & 0x3Fu)));
    } else {
        out.push_back(static_cast<char>(0xF0u | (cp >> 18)));
        out.push_back(static_cast<char>(0x80u | ((cp >> 12) & 0x3Fu)));
        out.push_back(static_cast<char>(0x80u | ((cp >> 6) & 0x3Fu)));
        out.push_back(static_cast<char>(0x80u | (cp & 0x3Fu)));
    }
}

// Key folding of StringComparer.OrdinalIgnoreCase (Snappy/Models/SnapshotInfo.cs:11-12,
// Snappy/Features/Mcdf/McdfManager.cs:34): every character goes to its simple upper-case
// mapping, letters outside ASCII and outside the BMP included ("chara/\u00e9.tex" and
// "CHARA/\u00c9.TEX" are one key).  Keys are UTF-8 here; bytes that are not well-formed
// UTF-8 are kept as they are.
std::string fold_key(const std::string &s)
{
    std::string k;
    k.reserve(s.size());
    const size_t n = s.size();
    for (size_t i = 0; i < n;) {
        const unsigned char c = static_cast<unsigned char>(s[i]);
        if (c < 0x80u) {
            k.push_back((c >= 'a' && c <= 'z') ? static_cast<char>(c - 'a' + 'A') : static_cast<char>(c));
            ++i;
            continue;
        }
        const size_t len = c >= 0xF0u ? 4 : c >= 0xE0u ? 3 : c >= 0xC0u ? 2 : 1;
        uint32_t cp = len == 4 ? c & 0x07u : len == 3 ? c & 0x0Fu : c & 0x1Fu;
        bool ok = len > 1 && i + len <= n && c < 0xF8u;
        for (size_t q = 1; ok && q < len; ++q) {
            const unsigned char t = static_cast<unsigned char>(s[i + q]);
            ok = (t & 0xC0u) == 0x80u;
            cp = (cp << 6) | (t & 0x3Fu);
        }
        if (!ok) {
            k.push_back(static_cast<char>(c));
            ++i;
            continue;
        }
        append_utf8(k, simple_upper(cp));
        i += len;
    }
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

// JsonConvert string escaping (StringEscapeHandling.Default: quotes, backslash, control
// characters, and U+0085, U+2028, U+2029, which it writes as \uXXXX).
std::string json_string(const std::string &s)
{
    std::string o = "\"";
    for (size_t i = 0; i < s.size(); ++i) {
        const unsigned char c = static_cast<unsigned char>(s[i]);
        const unsigned char c1 = i + 1 < s.size() ? static_cast<unsigned char>(s[i + 1]) : 0;
        const unsigned char c2 = i + 2 < s.size() ? static_cast<unsigned char>(s[i + 2]) : 0;
        if (c == 0xC2 && c1 == 0x85) {
            o += "\\u0085";
            i += 1;
            continue;
        }
        if (c == 0xE2 && c1 == 0x80 && (c2 == 0xA8 || c2 == 0xA9)) {
            o += c2 == 0xA8 ? "\\u2028" : "\\u2029";
            i += 2;
            continue;
        }
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
std::string snapshot_info_json(const std::string &format_version, const std::string &source_actor_json,
                               const std::string &last_update, const IgnoreCaseMap &replacements,
                               const std::string &manipulation)
{
    std::string o = "{\n  \"FormatVersion\": " + format_version + ",\n  \"SourceActor\": " +
                    source_actor_json +
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

// A string property as Newtonsoft holds it: a JSON null read from a file is assigned to the
// property as null and written back as null (absent properties keep the class's default "").
struct JsonText {
    std::string text;
    bool is_null = false;
    JsonText() = default;
    JsonText(std::string s) : text(std::move(s)) {}
    JsonText(const char *s) : text(s) {}
    bool differs_from(const std::string &other) const { return is_null || text != other; }
    std::string json() const { return is_null ? "null" : json_string(text); }
};

std::string history_json(const std::vector<std::vector<std::pair<std::string, JsonText>>> &entries)
{
    if (entries.empty())
        return "{\n  \"Entries\": []\n}";
    std::string o = "{\n  \"Entries\": [";
    for (size_t e = 0; e < entries.size(); ++e) {
        o += (e ? ",\n    {" : "\n    {");
        for (size_t k = 0; k < entries[e].size(); ++k) {
            o += (k ? ",\n      " : "\n      ");
            o += json_string(entries[e][k].first) + ": " + entries[e][k].second.json();
        }
        o += "\n    }";
    }
    o += "\n  ]\n}";
    return o;
}

// A snapshot folder's name comes out of an archive that another user made.  The
// reference hands it to Path.Combine and Directory.Delete(recursive) as it is
// (McdfManager.cs:93-107), so "..", an absolute path or "a/b" would delete or write
// outside the working directory.  Here such a name is refused before any file
// system call: it must be one path component.
bool is_single_component(const std::string &name)
{
    if (name.empty() || name == "." || name == "..")
        return false;
    bool only_blank = true;
    for (unsigned char c : name) {
        if (c == '/' || c == '\\' || c == '\0')
            return false;
        if (c != ' ' && c != '\t')
            only_blank = false;
    }
    return !only_blank;
}

// Convert.FromBase64String (McdfManager.cs:136): white space anywhere is ignored,
// the rest must be groups of four characters of the Base64 alphabet with '='
// padding only in the last one or two places; anything else is a FormatException.
bool is_dotnet_base64(const std::string &s)
{
    size_t chars = 0, pads = 0;
    for (unsigned char c : s) {
        if (c == ' ' || c == '\t' || c == '\r' || c == '\n')
            continue;
        if (c == '=') {
            ++pads;
        } else {
            const bool digit = (c >= 'A' && c <= 'Z') || (c >= 'a' && c <= 'z') ||
                               (c >= '0' && c <= '9') || c == '+' || c == '/';
            if (!digit || pads)
                return false;  // foreign character, or data after padding
        }
        ++chars;
    }
    return chars % 4 == 0 && pads <= 2;
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

}  // namespace

// snappy_shutdown: gives the page-locked file arena back (unless a call still holds it).
void release_file_arena()
{
    std::lock_guard<std::mutex> lk(g_file_arena.mu);
    if (g_file_arena.busy || !g_file_arena.ptr)
        return;
    (void)hipHostFree(g_file_arena.ptr);
    g_file_arena.ptr = nullptr;
    g_file_arena.cap = 0;
}

namespace {

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
        // Page-locked memory is allocated and freed with the arena's lock released: those calls
        // go through the library's initialisation lock, which snappy_shutdown holds while it takes
        // the arena's (release_file_arena).  The arena is marked busy first, so nobody else
        // touches it meanwhile.
        void *stale = nullptr;
        bool too_small = false;
        {
            std::lock_guard<std::mutex> lk(g_file_arena.mu);
            if (!g_file_arena.busy) {
                g_file_arena.busy = true;
                shared_ = true;
                if (g_file_arena.cap < bytes) {
                    too_small = true;
                    stale = g_file_arena.ptr;
                    g_file_arena.ptr = nullptr;
                    g_file_arena.cap = 0;
                } else {
                    ptr_ = g_file_arena.ptr;
                }
            }
        }
        if (!shared_) {
            own_cap_ = bytes;
            return snappy_host_alloc(bytes, &ptr_);
        }
        if (!too_small)
            return SNAPPY_OK;
        if (stale)
            (void)snappy_host_free(stale);
        const uint64_t want = bytes + bytes / 4;
        void *fresh = nullptr;
        const int st = snappy_host_alloc(want, &fresh);
        std::lock_guard<std::mutex> lk(g_file_arena.mu);
        if (st != SNAPPY_OK) {
            g_file_arena.busy = false;
            shared_ = false;
            return st;
        }
        g_file_arena.ptr = fresh;
        g_file_arena.cap = want;
        ptr_ = fresh;
        return SNAPPY_OK;
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
    std::mutex err_mu;
    std::exception_ptr first_error;
    auto work = [&] {
        try {
            for (uint64_t k = next.fetch_add(1); k < n; k = next.fetch_add(1))
                fn(k);
        } catch (...) {
            next.store(n);  // nobody starts another item
            std::lock_guard<std::mutex> lk(err_mu);
            if (!first_error)
                first_error = std::current_exception();
        }
    };
    std::vector<std::thread> pool;
    for (unsigned t = 0; t + 1 < threads; ++t)
        if (!try_spawn(pool, work))
            break;
    work();  // the calling thread is one of the workers
    for (auto &t : pool)
        t.join();
    if (first_error)
        std::rethrow_exception(first_error);  // surfaces in the caller's thread, under its guard
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

namespace {

// The read-and-hash half of the loops (SnapshotManager.cs:312-318): the files of
// `paths` that exist, read whole into one page-locked arena and hashed in one
// GPU batch while they are being read.
struct HashedFiles {
    ArenaLease arena;
    std::vector<uint64_t> owner;    // index into the caller's list, per file that was read
    std::vector<uint64_t> offs, sizes;  // where its bytes lie in the arena
    std::vector<uint8_t> digests;   // 20 bytes each
};

int read_and_hash_files(const char *const *paths, uint64_t n, HashedFiles &out)
{
    // File.Exists per entry (SnapshotManager.cs:314-315): missing sources are skipped.
    std::vector<uint64_t> owner, sizes, offs;
    uint64_t total = 0;
    for (uint64_t i = 0; i < n; ++i) {
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
    int st = out.arena.acquire(total + 256);
    if (st != SNAPPY_OK)
        return st;
    uint8_t *arena = out.arena.data();
    // File.ReadAllBytes (:317), many files in flight, largest first: that is the order
    // in which the batch below sends them to the GPU, and it starts sending (and
    // hashing) while the rest is still being read.
    std::vector<uint64_t> by_size(owner.size());
    for (size_t k = 0; k < owner.size(); ++k)
        by_size[k] = k;
    std::stable_sort(by_size.begin(), by_size.end(),
                     [&](uint64_t x, uint64_t y) { return sizes[x] > sizes[y]; });
    ReadGate gate(arena, offs, sizes);
    // Whatever happens to the reader, every file is reported to the gate: the transfer waits on it.
    std::vector<uint8_t> reported(owner.size(), 0);
    auto read_all = [&] {
        try {
            parallel_for(owner.size(), [&](uint64_t j) {
                const uint64_t k = by_size[j];
                const bool ok = read_exact(paths[owner[k]], arena + offs[k], sizes[k]);
                reported[k] = 1;
                gate.finished(k, ok);
            });
        } catch (...) {
        }
        for (size_t k = 0; k < owner.size(); ++k)
            if (!reported[k])
                gate.finished(k, false);
    };
    std::vector<uint8_t> all_digests(owner.size() * 20);
    {
        std::vector<std::thread> reader;
        struct Join {
            std::vector<std::thread> &t;
            ~Join()
            {
                for (auto &x : t)
                    if (x.joinable())
                        x.join();
            }
        } join_reader{reader};
        if (!try_spawn(reader, read_all))
            read_all();  // no thread to be had: read first, hash afterwards
        // Crypto.GetHash for all of them in one GPU batch (:318).
        st = sha1_batch_gated(arena, offs.data(), sizes.data(), owner.size(), all_digests.data(), &gate);
    }
    if (st != SNAPPY_OK)
        return st;
    // A file that could not be read in full is skipped like a missing one (its digest is of no use).
    for (size_t k = 0; k < owner.size(); ++k)
        if (gate.readable(k)) {
            out.owner.push_back(owner[k]);
            out.offs.push_back(offs[k]);
            out.sizes.push_back(sizes[k]);
            out.digests.insert(out.digests.end(), all_digests.begin() + 20 * k,
                               all_digests.begin() + 20 * (k + 1));
        }
    return SNAPPY_OK;
}

// One content-addressed file to store: dir/<hash>.dat = data[0..len) unless it exists
// (`if (!File.Exists(hashedFilePath))`, SnapshotManager.cs:457-466, McdfManager.cs:54-57).
struct Blob {
    std::string hash;
    const uint8_t *data;
    uint64_t len;
};

// Writes the blobs from several threads.  The caller passes every hash once.
int write_blobs(const fs::path &dir, const std::vector<Blob> &blobs)
{
    std::atomic<bool> write_failed{false};
    std::mutex fail_mu;
    std::string failed_target;
    parallel_for(blobs.size(), [&](uint64_t w) {
        const std::string target = (dir / (blobs[w].hash + ".dat")).string();
        std::error_code ec;
        if (!fs::exists(target, ec) &&
            !write_all_bytes(target, blobs[w].data, static_cast<size_t>(blobs[w].len))) {
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

// Digest -> index, for finding duplicates among 20-byte keys without formatting them.
struct DigestKey {
    uint8_t b[20];
    bool operator<(const DigestKey &o) const { return std::memcmp(b, o.b, 20) < 0; }
};

}  // namespace

extern "C" int snappy_store_files(const char *const *paths, uint64_t n, const char *files_dir,
                                  char *hex, uint8_t *status)
{
    return abi_guard([&]() -> int {
        if (n && (!paths || !hex || !status))
            return SNAPPY_E_INVALID_ARG;
        for (uint64_t i = 0; i < n; ++i) {
            hex[41 * i] = '\0';
            status[i] = 1;
        }
        HashedFiles files;
        int st = read_and_hash_files(paths, n, files);
        if (st != SNAPPY_OK)
            return st;
        // Entries with the same digest have the same bytes: the first of them writes the file.
        std::map<DigestKey, size_t> first_with_hash;
        std::vector<Blob> blobs;
        for (size_t k = 0; k < files.owner.size(); ++k) {
            char *h = hex + 41 * files.owner[k];
            hex40(files.digests.data() + 20 * k, h);
            status[files.owner[k]] = 0;
            DigestKey key;
            std::memcpy(key.b, files.digests.data() + 20 * k, 20);
            if (files_dir && first_with_hash.emplace(key, k).second)
                blobs.push_back({h, files.arena.data() + files.offs[k], files.sizes[k]});
        }
        if (!files_dir || blobs.empty())
            return SNAPPY_OK;
        std::error_code ec;
        fs::create_directories(files_dir, ec);
        return write_blobs(files_dir, blobs);
    });
}

namespace {

// "yyyy-MM-dd HH:mm:ss" of DateTime.UtcNow (the history entries' descriptions).
std::string utc_now_plain()
{
    const std::time_t t = std::time(nullptr);
    std::tm tm{};
    gmtime_r(&t, &tm);
    char buf[32];
    std::snprintf(buf, sizeof buf, "%04d-%02d-%02d %02d:%02d:%02d", tm.tm_year + 1900, tm.tm_mon + 1,
                  tm.tm_mday, tm.tm_hour, tm.tm_min, tm.tm_sec);
    return buf;
}

std::string base64_of(const std::string &s)  // Convert.ToBase64String
{
    static const char kAlphabet[] = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789+/";
    std::string o;
    size_t i = 0;
    for (; i + 2 < s.size(); i += 3) {
        const uint32_t v = (static_cast<uint8_t>(s[i]) << 16) | (static_cast<uint8_t>(s[i + 1]) << 8) |
                           static_cast<uint8_t>(s[i + 2]);
        o += kAlphabet[v >> 18];
        o += kAlphabet[(v >> 12) & 63];
        o += kAlphabet[(v >> 6) & 63];
        o += kAlphabet[v & 63];
    }
    if (i + 1 == s.size()) {
        const uint32_t v = static_cast<uint8_t>(s[i]) << 16;
        o += kAlphabet[v >> 18];
        o += kAlphabet[(v >> 12) & 63];
        o += "==";
    } else if (i + 2 == s.size()) {
        const uint32_t v = (static_cast<uint8_t>(s[i]) << 16) | (static_cast<uint8_t>(s[i + 1]) << 8);
        o += kAlphabet[v >> 18];
        o += kAlphabet[(v >> 12) & 63];
        o += kAlphabet[(v >> 6) & 63];
        o += '=';
    }
    return o;
}

bool read_text_file(const fs::path &path, std::string &out)
{
    std::ifstream f(path, std::ios::binary);
    if (!f)
        return false;
    out.assign(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
    if (out.size() >= 3 && std::memcmp(out.data(), "\xEF\xBB\xBF", 3) == 0)
        out.erase(0, 3);  // File.ReadAllText drops a byte-order mark
    return true;
}

// A string property of a deserialized object: absent keeps the default (""), a JSON
// null is assigned as null.
bool string_field(const JsonValue &obj, const char *name, JsonText &out)
{
    const JsonValue *v = json_member(obj, name);
    if (!v)
        return true;
    if (v->kind == JsonValue::Null) {
        out.text.clear();
        out.is_null = true;
        return true;
    }
    if (v->kind != JsonValue::String)
        return false;
    out.text = v->text;
    out.is_null = false;
    return true;
}

using HistoryEntries = std::vector<std::vector<std::pair<std::string, JsonText>>>;

// GlamourerHistory / CustomizeHistory as JsonConvert.DeserializeObject reads them
// (Snappy/Models/GlamourerHistory.cs, CustomizeHistory.cs): {"Entries":[{...}]},
// unknown properties ignored, the named string properties in declaration order.
bool load_history(const fs::path &path, const std::vector<const char *> &fields, HistoryEntries &out)
{
    std::string text;
    JsonValue root;
    if (!read_text_file(path, text) || !parse_json(text.data(), text.size(), root) ||
        root.kind != JsonValue::Object)
        return false;
    const JsonValue *entries = json_member(root, "Entries");
    if (!entries || entries->kind == JsonValue::Null)
        return true;
    if (entries->kind != JsonValue::Array)
        return false;
    for (const JsonValue &e : entries->items) {
        if (e.kind != JsonValue::Object)
            return false;
        std::vector<std::pair<std::string, JsonText>> entry;
        for (const char *f : fields) {
            JsonText v;
            if (!string_field(e, f, v))
                return false;
            entry.emplace_back(f, v);
        }
        out.push_back(std::move(entry));
    }
    return true;
}

}  // namespace

// SnapshotManager.UpdateSnapshot for the local player without the game and the
// plugin IPC (SnapshotManager.cs:292-332 builds the data, :408-527 stores it).
extern "C" int snappy_update_snapshot(const char *snapshot_dir, const char *source_actor,
                                      const char *const *resolved_paths,
                                      const char *const *game_paths, const uint32_t *paths_per_file,
                                      uint64_t n_files, const char *glamourer, const char *customize,
                                      const char *manipulation, char *hex, uint8_t *status)
{
    return abi_guard([&]() -> int {
        if (!snapshot_dir || !*snapshot_dir || (n_files && (!resolved_paths || !paths_per_file)))
            return SNAPPY_E_INVALID_ARG;
        uint64_t n_game_paths = 0;
        for (uint64_t i = 0; i < n_files; ++i)
            n_game_paths += paths_per_file[i];
        if (n_game_paths && !game_paths)
            return SNAPPY_E_INVALID_ARG;
        for (uint64_t g = 0; g < n_game_paths; ++g)
            if (!game_paths[g])
                return SNAPPY_E_INVALID_ARG;
        for (uint64_t i = 0; i < n_files; ++i) {
            if (hex)
                hex[41 * i] = '\0';
            if (status)
                status[i] = 1;
        }

        // BuildSnapshotFromLocalPlayer (:312-324): read and hash what exists, then
        //   resolvedPaths[hash] = path        (last writer wins)
        //   newFileReplacements[gamePath] = hash   (ordinary, case-sensitive dictionary)
        HashedFiles files;
        int st = read_and_hash_files(resolved_paths, n_files, files);
        if (st != SNAPPY_OK)
            return st;
        std::vector<uint64_t> first_game_path(n_files + 1, 0);
        for (uint64_t i = 0; i < n_files; ++i)
            first_game_path[i + 1] = first_game_path[i] + paths_per_file[i];
        std::map<DigestKey, size_t> source_of;  // hash -> the file whose bytes stand for it
        std::vector<std::string> hash_of(files.owner.size());
        std::map<std::string, size_t> slot_of;  // gamePath -> position in `fresh`
        std::vector<std::pair<std::string, size_t>> fresh;  // (gamePath, file) in first-insertion order
        for (size_t k = 0; k < files.owner.size(); ++k) {
            const uint64_t i = files.owner[k];
            char h[41];
            hex40(files.digests.data() + 20 * k, h);
            hash_of[k] = h;
            if (hex)
                std::memcpy(hex + 41 * i, h, 41);
            if (status)
                status[i] = 0;
            DigestKey key;
            std::memcpy(key.b, files.digests.data() + 20 * k, 20);
            source_of[key] = k;
            for (uint64_t g = first_game_path[i]; g < first_game_path[i + 1]; ++g) {
                auto it = slot_of.find(game_paths[g]);
                if (it == slot_of.end()) {
                    slot_of.emplace(game_paths[g], fresh.size());
                    fresh.emplace_back(game_paths[g], k);
                } else {
                    fresh[it->second].second = k;
                }
            }
        }

        // UpdateSnapshot (:424-451): the folder, and what it already holds.
        const fs::path snapshot = snapshot_dir;
        const fs::path files_dir = snapshot / "_files";
        std::error_code ec;
        fs::create_directories(files_dir, ec);
        if (ec) {
            set_error_detail("cannot create " + files_dir.string());
            return SNAPPY_E_INVALID_ARG;
        }
        const fs::path info_path = snapshot / "snapshot.json";
        const fs::path glam_path = snapshot / "glamourer_history.json";
        const fs::path cust_path = snapshot / "customize_history.json";
        std::string format_version = "1";
        JsonText actor(source_actor ? source_actor : ""), old_manipulation;
        IgnoreCaseMap replacements;  // SnapshotInfo.FileReplacements: OrdinalIgnoreCase
        if (fs::exists(info_path, ec)) {
            std::string text;
            JsonValue root;
            if (!read_text_file(info_path, text) || !parse_json(text.data(), text.size(), root) ||
                root.kind != JsonValue::Object) {
                set_error_detail("cannot parse " + info_path.string());
                return SNAPPY_E_SNAPSHOT_JSON;
            }
            if (const JsonValue *v = json_member(root, "FormatVersion")) {
                if (v->kind != JsonValue::Number || v->text.find_first_not_of("-0123456789") != std::string::npos) {
                    set_error_detail("FormatVersion is not an integer in " + info_path.string());
                    return SNAPPY_E_SNAPSHOT_JSON;
                }
                format_version = v->text;
            }
            actor = JsonText();
            JsonText ignored;
            bool ok = string_field(root, "SourceActor", actor) && string_field(root, "LastUpdate", ignored) &&
                      string_field(root, "ManipulationString", old_manipulation);
            if (const JsonValue *fr = json_member(root, "FileReplacements")) {
                if (fr->kind == JsonValue::Object) {
                    for (const auto &kv : fr->members) {
                        if (kv.second.kind != JsonValue::String) {
                            ok = false;
                            break;
                        }
                        replacements.set(kv.first, kv.second.text);
                    }
                } else if (fr->kind != JsonValue::Null) {
                    ok = false;
                }
            }
            if (!ok) {
                set_error_detail("unexpected value types in " + info_path.string());
                return SNAPPY_E_SNAPSHOT_JSON;
            }
        }
        HistoryEntries glam, cust;
        if (fs::exists(glam_path, ec) &&
            !load_history(glam_path, {"Timestamp", "Description", "GlamourerString"}, glam)) {
            set_error_detail("cannot parse " + glam_path.string());
            return SNAPPY_E_SNAPSHOT_JSON;
        }
        if (fs::exists(cust_path, ec) &&
            !load_history(cust_path, {"Timestamp", "Description", "CustomizeData", "CustomizeTemplate"}, cust)) {
            set_error_detail("cannot parse " + cust_path.string());
            return SNAPPY_E_SNAPSHOT_JSON;
        }

        // :453-475: merge the map (case-insensitive keys, last value wins) and store
        // each new hash once.  The bytes are the ones that were hashed: they are still
        // in the arena, so no source is read a second time (the reference's File.Copy
        // reads it again) and a hash that occurs many times is looked at once.
        std::vector<Blob> blobs;
        std::map<DigestKey, bool> stored;
        for (const auto &entry : fresh) {
            const size_t k = entry.second;
            replacements.set(entry.first, hash_of[k]);
            DigestKey key;
            std::memcpy(key.b, files.digests.data() + 20 * k, 20);
            if (!stored.emplace(key, true).second)
                continue;
            const size_t src = source_of[key];
            blobs.push_back({hash_of[k], files.arena.data() + files.offs[src], files.sizes[src]});
        }
        if ((st = write_blobs(files_dir, blobs)) != SNAPPY_OK)
            return st;

        // :477-515: manipulation string, the two histories (a new entry only when the
        // value differs from the last one), the time stamp.
        const std::string glamourer_now = glamourer ? glamourer : "";
        const std::string customize_now = customize ? customize : "";
        auto last_value = [](const HistoryEntries &h, const char *field) -> const JsonText * {
            if (h.empty())
                return nullptr;
            for (const auto &kv : h.back())
                if (kv.first == field)
                    return &kv.second;
            return nullptr;
        };
        const JsonText *last_glam = last_value(glam, "GlamourerString");
        if (!last_glam || last_glam->differs_from(glamourer_now))
            glam.push_back({{"Timestamp", JsonText(utc_now_roundtrip())},
                            {"Description", JsonText("Glamourer Update - " + utc_now_plain() + " UTC")},
                            {"GlamourerString", JsonText(glamourer_now)}});
        const std::string b64 = customize_now.empty() ? "" : base64_of(customize_now);
        const JsonText *last_cust = last_value(cust, "CustomizeData");
        if ((!last_cust || last_cust->differs_from(b64)) && !b64.empty())
            // CustomizeTemplate stays empty: see snappy_mcdf_import_to_dir.
            cust.push_back({{"Timestamp", JsonText(utc_now_roundtrip())},
                            {"Description", JsonText("Customize+ Update - " + utc_now_plain() + " UTC")},
                            {"CustomizeData", JsonText(b64)}, {"CustomizeTemplate", JsonText("")}});
        const std::string info_json = snapshot_info_json(
            format_version, actor.json(), utc_now_roundtrip(), replacements, manipulation ? manipulation : "");
        if (!write_text(info_path.string(), info_json) || !write_text(glam_path.string(), history_json(glam)) ||
            !write_text(cust_path.string(), history_json(cust))) {
            set_error_detail("cannot write snapshot JSON under " + snapshot.string());
            return SNAPPY_E_INVALID_ARG;
        }
        return SNAPPY_OK;
    });
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
    return abi_guard([&]() -> int {
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
    });
}

extern "C" int snappy_mcdf_import_to_dir(const char *mcdf_path, const char *working_dir,
                                         char *snapshot_dir_out, uint64_t out_cap)
{
    return abi_guard([&]() -> int {
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
        if (!is_single_component(name)) {
            set_error_detail("the archive's Description is not usable as a folder name: " + name);
            return SNAPPY_E_INVALID_ARG;
        }
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
            snapshot_info_json("1", json_string(name), stamp, game_path_to_hash, info.manipulation_data);
        HistoryEntries glam = {{{"Timestamp", JsonText(stamp)}, {"Description", JsonText("Imported from MCDF")},
                                {"GlamourerString", JsonText(info.glamourer_data)}}};
        HistoryEntries cust;
        // Convert.FromBase64String throws on malformed data (McdfManager.cs:136), after the
        // files have been extracted and before anything is saved: the import fails there.
        if (!info.customize_plus_data.empty() && !is_dotnet_base64(info.customize_plus_data)) {
            set_error_detail("CustomizePlusData is not valid Base64");
            return SNAPPY_E_MCDF_BASE64;
        }
        if (!info.customize_plus_data.empty())
            // CustomizeTemplate stays empty: the reference fills it with Base64(gzip(JSON))
            // (Snappy/Utils/CustomizePlusUtil.cs:26-76), whose bytes depend on the deflate
            // encoder and on Newtonsoft's float formatting, so no byte-exact mirror exists;
            // CustomizeHistoryEntry leaves it empty itself whenever generation fails
            // (Snappy/Models/CustomizeHistory.cs:34-48).
            cust.push_back({{"Timestamp", JsonText(stamp)}, {"Description", JsonText("Imported from MCDF")},
                            {"CustomizeData", JsonText(info.customize_plus_data)}, {"CustomizeTemplate", JsonText("")}});
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
    });
}
