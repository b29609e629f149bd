// MCDF header parse and file slicing on the host
// (/root/reference/Snappy/Features/Mcdf/McdfHeader.cs:26-62, McdfData.cs:7-26,
// McdfManager.cs:36-49).  The header is a few KB of JSON; it stays on the CPU.
#include "mcdf_host.hpp"

#include <cstring>
#include <memory>
#include <utility>

#include "../../include/snappyhash.h"
#include "abi_guard.hpp"

namespace snappy {

namespace {

// ---------------------------------------------------------------- tiny JSON DOM
class JsonParser {
public:
    JsonParser(const char *p, size_t n) : p_(p), end_(p + n) {}

    bool parse_document(JsonValue &out)
    {
        skip_ws();
        if (!parse_value(out, 0))
            return false;
        skip_ws();
        return p_ == end_;
    }

private:
    const char *p_;
    const char *end_;
    static constexpr int kMaxDepth = 64;
    // A value costs ~100 bytes of DOM however small its text: "[[],[],..." would turn
    // a 2 GiB header into tens of GiB.  A real header has a few thousand values.
    static constexpr size_t kMaxValues = size_t(1) << 20;
    size_t n_values_ = 0;

    void skip_ws()
    {
        while (p_ < end_ && (*p_ == ' ' || *p_ == '\t' || *p_ == '\n' || *p_ == '\r'))
            ++p_;
    }

    bool literal(const char *word)
    {
        const size_t n = std::strlen(word);
        if (static_cast<size_t>(end_ - p_) < n || std::memcmp(p_, word, n) != 0)
            return false;
        p_ += n;
        return true;
    }

    static void append_utf8(std::string &s, uint32_t cp)
    {
        if (cp < 0x80) {
            s.push_back(static_cast<char>(cp));
        } else if (cp < 0x800) {
            s.push_back(static_cast<char>(0xC0 | (cp >> 6)));
            s.push_back(static_cast<char>(0x80 | (cp & 0x3F)));
        } else if (cp < 0x10000) {
            s.push_back(static_cast<char>(0xE0 | (cp >> 12)));
            s.push_back(static_cast<char>(0x80 | ((cp >> 6) & 0x3F)));
            s.push_back(static_cast<char>(0x80 | (cp & 0x3F)));
        } else {
            s.push_back(static_cast<char>(0xF0 | (cp >> 18)));
            s.push_back(static_cast<char>(0x80 | ((cp >> 12) & 0x3F)));
            s.push_back(static_cast<char>(0x80 | ((cp >> 6) & 0x3F)));
            s.push_back(static_cast<char>(0x80 | (cp & 0x3F)));
        }
    }

    bool hex4(uint32_t &v)
    {
        if (end_ - p_ < 4)
            return false;
        v = 0;
        for (int i = 0; i < 4; ++i) {
            const char c = *p_++;
            v <<= 4;
            if (c >= '0' && c <= '9') v |= static_cast<uint32_t>(c - '0');
            else if (c >= 'a' && c <= 'f') v |= static_cast<uint32_t>(c - 'a' + 10);
            else if (c >= 'A' && c <= 'F') v |= static_cast<uint32_t>(c - 'A' + 10);
            else return false;
        }
        return true;
    }

    bool parse_string(std::string &out)
    {
        if (p_ >= end_ || *p_ != '"')
            return false;
        ++p_;
        out.clear();
        while (p_ < end_) {
            const unsigned char c = static_cast<unsigned char>(*p_++);
            if (c == '"')
                return true;
            if (c < 0x20)
                return false;
            if (c != '\\') {
                out.push_back(static_cast<char>(c));
                continue;
            }
            if (p_ >= end_)
                return false;
            const char e = *p_++;
            switch (e) {
            case '"': out.push_back('"'); break;
            case '\\': out.push_back('\\'); break;
            case '/': out.push_back('/'); break;
            case 'b': out.push_back('\b'); break;
            case 'f': out.push_back('\f'); break;
            case 'n': out.push_back('\n'); break;
            case 'r': out.push_back('\r'); break;
            case 't': out.push_back('\t'); break;
            case 'u': {
                uint32_t cp;
                if (!hex4(cp))
                    return false;
                if (cp >= 0xD800 && cp < 0xDC00 && end_ - p_ >= 6 && p_[0] == '\\' && p_[1] == 'u') {
                    const char *save = p_;
                    p_ += 2;
                    uint32_t lo;
                    if (!hex4(lo))
                        return false;
                    if (lo >= 0xDC00 && lo < 0xE000)
                        cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
                    else
                        p_ = save;
                }
                append_utf8(out, cp);
                break;
            }
            default: return false;
            }
        }
        return false;
    }

    bool parse_number(JsonValue &out)
    {
        const char *s = p_;
        if (p_ < end_ && *p_ == '-')
            ++p_;
        if (p_ >= end_)
            return false;
        if (*p_ == '0') {
            ++p_;
        } else if (*p_ >= '1' && *p_ <= '9') {
            while (p_ < end_ && *p_ >= '0' && *p_ <= '9')
                ++p_;
        } else {
            return false;
        }
        if (p_ < end_ && *p_ == '.') {
            ++p_;
            if (p_ >= end_ || *p_ < '0' || *p_ > '9')
                return false;
            while (p_ < end_ && *p_ >= '0' && *p_ <= '9')
                ++p_;
        }
        if (p_ < end_ && (*p_ == 'e' || *p_ == 'E')) {
            ++p_;
            if (p_ < end_ && (*p_ == '+' || *p_ == '-'))
                ++p_;
            if (p_ >= end_ || *p_ < '0' || *p_ > '9')
                return false;
            while (p_ < end_ && *p_ >= '0' && *p_ <= '9')
                ++p_;
        }
        out.kind = JsonValue::Number;
        out.text.assign(s, static_cast<size_t>(p_ - s));
        return true;
    }

    bool parse_value(JsonValue &out, int depth)
    {
        if (depth > kMaxDepth || p_ >= end_ || ++n_values_ > kMaxValues)
            return false;
        const char c = *p_;
        if (c == '{') {
            ++p_;
            out.kind = JsonValue::Object;
            skip_ws();
            if (p_ < end_ && *p_ == '}') {
                ++p_;
                return true;
            }
            for (;;) {
                skip_ws();
                std::string key;
                if (!parse_string(key))
                    return false;
                skip_ws();
                if (p_ >= end_ || *p_ != ':')
                    return false;
                ++p_;
                skip_ws();
                JsonValue v;
                if (!parse_value(v, depth + 1))
                    return false;
                out.members.emplace_back(std::move(key), std::move(v));
                skip_ws();
                if (p_ >= end_)
                    return false;
                if (*p_ == ',') {
                    ++p_;
                    continue;
                }
                if (*p_ == '}') {
                    ++p_;
                    return true;
                }
                return false;
            }
        }
        if (c == '[') {
            ++p_;
            out.kind = JsonValue::Array;
            skip_ws();
            if (p_ < end_ && *p_ == ']') {
                ++p_;
                return true;
            }
            for (;;) {
                skip_ws();
                JsonValue v;
                if (!parse_value(v, depth + 1))
                    return false;
                out.items.push_back(std::move(v));
                skip_ws();
                if (p_ >= end_)
                    return false;
                if (*p_ == ',') {
                    ++p_;
                    continue;
                }
                if (*p_ == ']') {
                    ++p_;
                    return true;
                }
                return false;
            }
        }
        if (c == '"') {
            out.kind = JsonValue::String;
            return parse_string(out.text);
        }
        if (c == 't') {
            out.kind = JsonValue::Bool;
            out.boolean = true;
            return literal("true");
        }
        if (c == 'f') {
            out.kind = JsonValue::Bool;
            out.boolean = false;
            return literal("false");
        }
        if (c == 'n') {
            out.kind = JsonValue::Null;
            return literal("null");
        }
        return parse_number(out);
    }
};

bool ascii_iequals(const std::string &a, const char *b)
{
    const size_t n = std::strlen(b);
    if (a.size() != n)
        return false;
    for (size_t i = 0; i < n; ++i) {
        char x = a[i], y = b[i];
        if (x >= 'A' && x <= 'Z') x = static_cast<char>(x - 'A' + 'a');
        if (y >= 'A' && y <= 'Z') y = static_cast<char>(y - 'A' + 'a');
        if (x != y)
            return false;
    }
    return true;
}

const JsonValue *member(const JsonValue &obj, const char *name)
{
    const JsonValue *exact = nullptr, *loose = nullptr;
    for (const auto &kv : obj.members) {
        if (kv.first == name)
            exact = &kv.second;
        else if (ascii_iequals(kv.first, name))
            loose = &kv.second;
    }
    return exact ? exact : loose;
}

bool string_member(const JsonValue &obj, const char *name, std::string &out)
{
    const JsonValue *v = member(obj, name);
    out.clear();
    if (!v || v->kind == JsonValue::Null)
        return true;  // absent or null: the field keeps its empty default
    if (v->kind != JsonValue::String)
        return false;
    out = v->text;
    return true;
}

// Integer literal (or a string holding one) -> int64; false on anything else.
bool to_int64(const JsonValue &v, int64_t &out)
{
    if (v.kind != JsonValue::Number && v.kind != JsonValue::String)
        return false;
    const std::string &t = v.text;
    size_t i = 0;
    bool neg = false;
    if (i < t.size() && (t[i] == '-' || (v.kind == JsonValue::String && t[i] == '+'))) {
        neg = t[i] == '-';
        ++i;
    }
    if (i >= t.size())
        return false;
    uint64_t acc = 0;
    for (; i < t.size(); ++i) {
        if (t[i] < '0' || t[i] > '9')
            return false;
        const uint64_t d = static_cast<uint64_t>(t[i] - '0');
        if (acc > (0x8000000000000000ull - d) / 10)
            return false;
        acc = acc * 10 + d;
    }
    if (neg) {
        out = (acc == 0x8000000000000000ull) ? INT64_MIN : -static_cast<int64_t>(acc);
    } else {
        if (acc > 0x7FFFFFFFFFFFFFFFull)
            return false;
        out = static_cast<int64_t>(acc);
    }
    return true;
}

}  // namespace

bool parse_json(const char *text, size_t n, JsonValue &out)
{
    JsonParser parser(text, n);
    return parser.parse_document(out);
}

const JsonValue *json_member(const JsonValue &obj, const char *name) { return member(obj, name); }

int mcdf_header_extent(const uint8_t *payload, uint64_t avail, uint64_t payload_len, uint64_t *need)
{
    // McdfHeader.ReadHeader: ReadChars(4) must spell "MCDF" (McdfHeader.cs:28-30).
    if (payload_len < 4 || avail < 4 || std::memcmp(payload, "MCDF", 4) != 0)
        return SNAPPY_E_MCDF_MAGIC;
    if (payload_len < 5 || avail < 5)
        return SNAPPY_E_MCDF_TRUNCATED;
    if (payload[4] != 1)  // McdfHeader.cs:32-39
        return SNAPPY_E_MCDF_VERSION;
    if (payload_len < 9 || avail < 9)
        return SNAPPY_E_MCDF_TRUNCATED;
    const uint32_t raw = static_cast<uint32_t>(payload[5]) | (static_cast<uint32_t>(payload[6]) << 8) |
                         (static_cast<uint32_t>(payload[7]) << 16) |
                         (static_cast<uint32_t>(payload[8]) << 24);
    const int32_t json_len = static_cast<int32_t>(raw);  // reader.ReadInt32()
    if (json_len < 0 || 9ull + static_cast<uint64_t>(json_len) > payload_len)
        return SNAPPY_E_MCDF_TRUNCATED;
    *need = 9ull + static_cast<uint64_t>(json_len);
    return SNAPPY_OK;
}

int parse_mcdf_header(const uint8_t *payload, uint64_t avail, McdfHeaderInfo &out)
{
    uint64_t need = 0;
    int st = mcdf_header_extent(payload, avail, avail, &need);
    if (st != SNAPPY_OK)
        return st;
    out = McdfHeaderInfo();
    out.version = payload[4];
    out.body_offset = need;

    JsonValue root;
    JsonParser parser(reinterpret_cast<const char *>(payload) + 9, static_cast<size_t>(need - 9));
    if (!parser.parse_document(root) || root.kind != JsonValue::Object)
        return SNAPPY_E_MCDF_JSON;
    if (!string_member(root, "Description", out.description) ||
        !string_member(root, "GlamourerData", out.glamourer_data) ||
        !string_member(root, "CustomizePlusData", out.customize_plus_data) ||
        !string_member(root, "ManipulationData", out.manipulation_data))
        return SNAPPY_E_MCDF_JSON;

    const JsonValue *files = member(root, "Files");
    if (!files)
        return SNAPPY_OK;  // Files keeps its default: an empty list (McdfData.cs:13)
    if (files->kind != JsonValue::Array)
        return SNAPPY_E_MCDF_JSON;
    out.files.reserve(files->items.size());
    for (const JsonValue &f : files->items) {
        if (f.kind != JsonValue::Object)
            return SNAPPY_E_MCDF_JSON;
        McdfFileEntry e;
        const JsonValue *paths = member(f, "GamePaths");
        if (!paths || paths->kind != JsonValue::Array)
            return SNAPPY_E_MCDF_JSON;
        for (const JsonValue &p : paths->items) {
            if (p.kind != JsonValue::String)
                return SNAPPY_E_MCDF_JSON;
            e.game_paths.push_back(p.text);
        }
        const JsonValue *len = member(f, "Length");
        if (len && !to_int64(*len, e.length))
            return SNAPPY_E_MCDF_JSON;
        out.files.push_back(std::move(e));
    }
    return SNAPPY_OK;
}

int mcdf_slice_files(const std::vector<McdfFileEntry> &files, uint64_t body_offset,
                     uint64_t payload_len, std::vector<McdfSlice> &slices)
{
    slices.assign(files.size(), McdfSlice());
    uint64_t cursor = body_offset;
    for (size_t i = 0; i < files.size(); ++i) {
        const int64_t length = files[i].length;
        McdfSlice &s = slices[i];
        if (length == 0) {  // McdfManager.cs:39-40
            s.status = SNAPPY_FILE_SKIPPED_EMPTY;
            continue;
        }
        // reader.ReadBytes((int)length): the count is narrowed to int; a negative
        // count throws, which aborts the import.
        const int32_t count = static_cast<int32_t>(static_cast<uint32_t>(static_cast<uint64_t>(length)));
        if (count < 0)
            return SNAPPY_E_INVALID_ARG;
        const uint64_t left = payload_len - cursor;
        const uint64_t got = static_cast<uint64_t>(count) < left ? static_cast<uint64_t>(count) : left;
        s.offset = cursor;
        cursor += got;  // ReadBytes consumes what it returns, short or not
        if (static_cast<int64_t>(got) != length) {  // McdfManager.cs:43-49
            s.status = SNAPPY_FILE_SKIPPED_SHORT_READ;
            continue;
        }
        s.length = got;
        s.status = SNAPPY_FILE_HASHED;
    }
    return SNAPPY_OK;
}

}  // namespace snappy

using namespace snappy;

extern "C" int snappy_mcdf_parse_header(const uint8_t *payload, uint64_t payload_len,
                                        uint64_t *n_files, uint64_t *body_offset,
                                        int64_t *file_lengths, uint64_t lengths_cap)
{
    return abi_guard([&]() -> int {
        if (!n_files || !body_offset || (!payload && payload_len))
            return SNAPPY_E_INVALID_ARG;
        if (!payload)
            return SNAPPY_E_MCDF_MAGIC;
        McdfHeaderInfo info;
        int st = parse_mcdf_header(payload, payload_len, info);
        if (st != SNAPPY_OK)
            return st;
        *n_files = info.files.size();
        *body_offset = info.body_offset;
        if (file_lengths) {
            if (lengths_cap < info.files.size())
                return SNAPPY_E_CAPACITY;
            for (size_t i = 0; i < info.files.size(); ++i)
                file_lengths[i] = info.files[i].length;
        }
        return SNAPPY_OK;
    });
}
