"""CPU oracle, Python side -- TEST INFRASTRUCTURE ONLY.

Loads the C restatement (oracle/_build/libsnappy_oracle.so) through ctypes and
adds the small host-side parts of the MCDF import as pure Python: the header
parse (/root/reference/Snappy/Features/Mcdf/McdfHeader.cs:26-62, McdfData.cs:7-26)
and the slice/skip/short-read loop (McdfManager.cs:28-66).

Only tests/, the smoke check and the cpu_baseline leg of bench.py may import
this module, and only as the checker.  Nothing under snappy_amd/ imports it.
"""
from __future__ import annotations

import ctypes
import json
import os
import re
import struct
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libsnappy_oracle.so")

OK = 0
E_INVALID_ARG = -1
E_LZ4_TRUNCATED = -10
E_LZ4_BAD_LENGTHS = -11
E_LZ4_PASSES = -12
E_LZ4_CORRUPT = -13
E_CAPACITY = -14
E_MCDF_MAGIC = -20
E_MCDF_VERSION = -21
E_MCDF_TRUNCATED = -22
E_MCDF_JSON = -23

FILE_HASHED = 0
FILE_SKIPPED_EMPTY = 1
FILE_SKIPPED_SHORT_READ = 2

_u8p = ctypes.POINTER(ctypes.c_uint8)
_u64p = ctypes.POINTER(ctypes.c_uint64)


def build(force: bool = False) -> str:
    """Compile the C oracle with make (gcc); returns the library path."""
    if force or not os.path.exists(_LIB_PATH):
        subprocess.run(["make", "-s", "-C", _HERE, "all"], check=True)
    return _LIB_PATH


_lib = None


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH)
        L.oracle_sha1.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p]
        L.oracle_sha1.restype = None
        L.oracle_sha1_hex.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_char_p]
        L.oracle_sha1_hex.restype = None
        L.oracle_sha1_batch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                        ctypes.c_uint64, ctypes.c_void_p]
        L.oracle_sha1_batch.restype = None
        L.oracle_lz4_block_decode.argtypes = [ctypes.c_void_p, ctypes.c_uint64,
                                              ctypes.c_void_p, ctypes.c_uint64]
        L.oracle_lz4_block_decode.restype = ctypes.c_int
        L.oracle_lz4net_decoded_size.argtypes = [ctypes.c_void_p, ctypes.c_uint64, _u64p]
        L.oracle_lz4net_decoded_size.restype = ctypes.c_int
        L.oracle_lz4net_decode.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p,
                                           ctypes.c_uint64, _u64p]
        L.oracle_lz4net_decode.restype = ctypes.c_int
        L.oracle_lz4_block_encode.argtypes = [ctypes.c_void_p, ctypes.c_uint64,
                                              ctypes.c_void_p, ctypes.c_uint64]
        L.oracle_lz4_block_encode.restype = ctypes.c_uint64
        L.oracle_lz4net_encode.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p,
                                           ctypes.c_uint64, ctypes.c_uint64, ctypes.c_int]
        L.oracle_lz4net_encode.restype = ctypes.c_uint64
        L.oracle_lz4net_encode_bound.argtypes = [ctypes.c_uint64, ctypes.c_uint64]
        L.oracle_lz4net_encode_bound.restype = ctypes.c_uint64
        L.oracle_fill_splitmix.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64]
        L.oracle_fill_splitmix.restype = None
        _lib = L
    return _lib


def _buf(data) -> tuple:
    """Read-only view of bytes-like data as (address-holder, length)."""
    b = bytes(data) if not isinstance(data, (bytes, bytearray)) else data
    arr = (ctypes.c_uint8 * max(len(b), 1)).from_buffer_copy(b if len(b) else b"\0")
    return arr, len(b)


# --------------------------------------------------------------------------- SHA-1

def sha1(data) -> bytes:
    arr, n = _buf(data)
    out = (ctypes.c_uint8 * 20)()
    lib().oracle_sha1(arr, n, out)
    return bytes(out)


def get_hash(data) -> str:
    """Crypto.GetHash (Crypto.cs:8-22): 40 uppercase hex chars."""
    arr, n = _buf(data)
    out = ctypes.create_string_buffer(41)
    lib().oracle_sha1_hex(arr, n, out)
    return out.value.decode("ascii")


def sha1_batch_np(base, offsets, lengths):
    """base: numpy uint8 array; offsets/lengths: numpy uint64 arrays. Returns (n,20) uint8."""
    import numpy as np

    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    lengths = np.ascontiguousarray(lengths, dtype=np.uint64)
    base = np.ascontiguousarray(base, dtype=np.uint8)
    n = offsets.shape[0]
    out = np.zeros((n, 20), dtype=np.uint8)
    lib().oracle_sha1_batch(base.ctypes.data, offsets.ctypes.data, lengths.ctypes.data, n,
                            out.ctypes.data)
    return out


def fill_splitmix(length: int, seed: int) -> bytes:
    out = (ctypes.c_uint8 * max(length, 1))()
    lib().oracle_fill_splitmix(out, length, seed & 0xFFFFFFFFFFFFFFFF)
    return bytes(out)[:length]


# --------------------------------------------------------------------------- LZ4

def lz4_block_decode(block, out_len: int):
    """Returns (status, bytes)."""
    arr, n = _buf(block)
    out = (ctypes.c_uint8 * max(out_len, 1))()
    st = lib().oracle_lz4_block_decode(arr, n, out, out_len)
    return st, bytes(out)[:out_len]


def lz4_block_encode(data) -> bytes:
    arr, n = _buf(data)
    cap = n + n // 255 + 32
    out = (ctypes.c_uint8 * cap)()
    w = lib().oracle_lz4_block_encode(arr, n, out, cap)
    assert w > 0
    return bytes(out)[:w]


def lz4net_decoded_size(stream):
    arr, n = _buf(stream)
    v = ctypes.c_uint64(0)
    st = lib().oracle_lz4net_decoded_size(arr, n, ctypes.byref(v))
    return st, v.value


def lz4net_decode(stream, out_cap=None):
    """Returns (status, bytes)."""
    arr, n = _buf(stream)
    if out_cap is None:
        st, out_cap = lz4net_decoded_size(stream)
        if st != OK:
            # Still run the decoder (it reports errors in stream order) with room for
            # whatever precedes the bad chunk: LZ4 expands at most 255x.
            out_cap = min(255 * n + (1 << 16), 1 << 31)
    out = (ctypes.c_uint8 * max(out_cap, 1))()
    v = ctypes.c_uint64(0)
    st = lib().oracle_lz4net_decode(arr, n, out, out_cap, ctypes.byref(v))
    return st, (bytes(out)[: v.value] if st == OK else b"")


def lz4net_encode(data, chunk_size: int = 1 << 20, hc_flag: bool = True) -> bytes:
    arr, n = _buf(data)
    cap = lib().oracle_lz4net_encode_bound(n, chunk_size)
    out = (ctypes.c_uint8 * cap)()
    w = lib().oracle_lz4net_encode(arr, n, out, cap, chunk_size, int(hc_flag))
    assert w > 0 or n == 0
    return bytes(out)[:w]


# --------------------------------------------------------------------------- MCDF

class McdfError(Exception):
    def __init__(self, code: int, msg: str):
        super().__init__(msg)
        self.code = code


def _ci_get(obj: dict, name: str, default):
    """Newtonsoft binds JSON properties to members case-insensitively (exact match first)."""
    if name in obj:
        return obj[name]
    low = name.lower()
    for k, v in obj.items():
        if k.lower() == low:
            return v
    return default


def mcdf_parse_header(payload: bytes):
    """McdfHeader.ReadHeader + FromBinaryReader (McdfHeader.cs:26-62).

    Returns (version, data, body_offset) where data has the McdfData fields
    (McdfData.cs:9-13) and Files is a list of (game_paths, length).
    """
    if len(payload) < 4 or payload[:4] != b"MCDF":
        raise McdfError(E_MCDF_MAGIC, "Not a Mare Chara File")
    if len(payload) < 5:
        raise McdfError(E_MCDF_TRUNCATED, "end of stream reading version")
    version = payload[4]
    if version != 1:
        raise McdfError(E_MCDF_VERSION, f"Unsupported MCDF version: {version}")
    if len(payload) < 9:
        raise McdfError(E_MCDF_TRUNCATED, "end of stream reading data length")
    (data_len,) = struct.unpack_from("<i", payload, 5)
    if data_len < 0:
        raise McdfError(E_MCDF_TRUNCATED, "negative data length")
    if 9 + data_len > len(payload):
        raise McdfError(E_MCDF_TRUNCATED, "header JSON runs past end of payload")
    try:
        # Encoding.UTF8.GetString substitutes U+FFFD for bad bytes; it does not throw.
        text = payload[9 : 9 + data_len].decode("utf-8", errors="replace")
        obj = json.loads(text)
    except ValueError as e:
        raise McdfError(E_MCDF_JSON, f"bad header JSON: {e}")
    if not isinstance(obj, dict):
        raise McdfError(E_MCDF_JSON, "header JSON is not an object")
    files = []
    raw_files = _ci_get(obj, "Files", [])
    if not isinstance(raw_files, list):
        raise McdfError(E_MCDF_JSON, "Files is not an array")
    for f in raw_files:
        if not isinstance(f, dict):
            raise McdfError(E_MCDF_JSON, "Files entry is not an object")
        # A missing or null GamePaths leaves the record's IEnumerable null and the
        # import dies on it (McdfManager.cs:59); both count as a malformed header here.
        paths = _ci_get(f, "GamePaths", None)
        length = _ci_get(f, "Length", 0)
        if isinstance(length, str):
            if not re.fullmatch(r"[+-]?[0-9]+", length):
                raise McdfError(E_MCDF_JSON, "Length is not an integer")
            length = int(length)
        if (not isinstance(paths, list) or any(not isinstance(p, str) for p in paths)
                or isinstance(length, bool) or not isinstance(length, int)
                or not -(1 << 63) <= length < (1 << 63)):
            raise McdfError(E_MCDF_JSON, "bad Files entry")
        files.append((list(paths), length))
    data = {
        "Description": _ci_get(obj, "Description", "") or "",
        "GlamourerData": _ci_get(obj, "GlamourerData", "") or "",
        "CustomizePlusData": _ci_get(obj, "CustomizePlusData", "") or "",
        "ManipulationData": _ci_get(obj, "ManipulationData", "") or "",
        "Files": files,
    }
    return version, data, 9 + data_len


def simple_upper(ch: str) -> str:
    """Simple (one code point to one code point) upper-case mapping of one character, as .NET's
    ordinal casing uses it: str.upper() when that is a single character; the Greek letters with
    iota subscript, whose full mapping is two letters, go to their single titlecase letter (the
    UnicodeData.txt simple mapping); U+0131 (dotless i) stays, as in .NET's invariant casing."""
    cp = ord(ch)
    if cp == 0x0131:
        return ch
    if 0x1F80 <= cp <= 0x1FAF and (cp & 0xF) < 8:
        return chr(cp + 8)
    if cp in (0x1FB3, 0x1FC3, 0x1FF3):
        return chr(cp + 9)
    u = ch.upper()
    return u if len(u) == 1 else ch


def _ordinal_upper(s: str) -> str:
    """Key folding of StringComparer.OrdinalIgnoreCase (Models/SnapshotInfo.cs:11-12,
    Features/Mcdf/McdfManager.cs:34): simple per-character upper-casing, supplementary-plane
    letters included."""
    return "".join(simple_upper(ch) for ch in s)


def mcdf_extract_and_hash(files, payload: bytes, body_offset: int):
    """ExtractAndHashMapFiles (McdfManager.cs:28-66).

    Returns (game_path_to_hash, statuses, blobs) where game_path_to_hash keeps
    first-seen key spelling and insertion order (Dictionary with
    OrdinalIgnoreCase), statuses[i] is FILE_* for Files[i], and blobs maps
    HASH -> bytes for every `<HASH>.dat` the loop would write.
    """
    cursor = body_offset
    mapping = {}  # folded key -> [original key, hash]
    statuses = []
    blobs = {}
    for paths, length in files:
        if length == 0:  # McdfManager.cs:39-40
            statuses.append(FILE_SKIPPED_EMPTY)
            continue
        n = ((length + (1 << 31)) % (1 << 32)) - (1 << 31)  # (int)length
        if n < 0:
            raise McdfError(E_INVALID_ARG, "ReadBytes count is negative")
        avail = min(n, len(payload) - cursor)
        buf = payload[cursor : cursor + avail]  # ReadBytes consumes what is left
        cursor += avail
        if avail != length:  # McdfManager.cs:43-49: log and skip
            statuses.append(FILE_SKIPPED_SHORT_READ)
            continue
        h = get_hash(buf)
        blobs.setdefault(h, buf)  # File.Exists guard, McdfManager.cs:54-57
        for p in paths:
            key = _ordinal_upper(p)
            if key in mapping:
                mapping[key][1] = h
            else:
                mapping[key] = [p, h]
        statuses.append(FILE_HASHED)
    return {k: v for k, v in mapping.values()}, statuses, blobs


def mcdf_import(stream: bytes):
    """ImportMcdf data path (McdfManager.cs:72-114) without the file system or IPC."""
    st, payload = lz4net_decode(stream)
    if st != OK:
        raise McdfError(st, "lz4net stream error")
    version, data, body = mcdf_parse_header(payload)
    mapping, statuses, blobs = mcdf_extract_and_hash(data["Files"], payload, body)
    return {"version": version, "data": data, "map": mapping, "statuses": statuses,
            "blobs": blobs, "payload": payload, "body_offset": body}


def build_mcdf_payload(description: str, files, extra: dict | None = None) -> bytes:
    """Test-data generator: McdfHeader.WriteToStream layout (McdfHeader.cs:14-24)
    followed by the file bodies. files = [(game_paths, data_bytes_or_None, declared_len)]."""
    obj = {
        "Description": description,
        "GlamourerData": "",
        "CustomizePlusData": "",
        "ManipulationData": "",
        "Files": [{"GamePaths": list(p), "Length": int(ln)} for p, _, ln in files],
    }
    if extra:
        obj.update(extra)
    js = json.dumps(obj, separators=(",", ":")).encode("utf-8")
    out = bytearray(b"MCDF")
    out.append(1)
    out += struct.pack("<i", len(js))
    out += js
    for _, body, _ in files:
        if body:
            out += body
    return bytes(out)


# ------------------------------------------------------------------ snapshot folder

def _ignore_case_set(d: dict, key: str, value):
    """Indexer set of Dictionary<string,string>(OrdinalIgnoreCase) kept as {folded: [key, value]}."""
    k = _ordinal_upper(key)
    if k in d:
        d[k][1] = value
    else:
        d[k] = [key, value]


def update_snapshot(snapshot_dir: str, source_actor: str, replacements, glamourer: str = "",
                    customize: str = "", manipulation: str = "") -> str:
    """UpdateSnapshot for the local player (Snappy/Core/SnapshotManager.cs:292-332, 408-527) on the
    file system, sequential like the reference.  replacements = [(resolved_path, [game paths])].
    Time stamps are written as fixed markers; CustomizeTemplate is left empty."""
    import base64

    new_replacements, resolved_paths = {}, {}        # ordinary (case-sensitive) dictionaries
    for resolved, game_paths in replacements:        # :312-324
        if not os.path.isfile(resolved):
            continue
        with open(resolved, "rb") as f:
            h = get_hash(f.read())
        resolved_paths[h] = resolved
        for g in game_paths:
            new_replacements[g] = h
    files_dir = os.path.join(snapshot_dir, "_files")
    os.makedirs(files_dir, exist_ok=True)            # :424-425
    info_path = os.path.join(snapshot_dir, "snapshot.json")
    glam_path = os.path.join(snapshot_dir, "glamourer_history.json")
    cust_path = os.path.join(snapshot_dir, "customize_history.json")
    info = {"FormatVersion": 1, "SourceActor": source_actor, "LastUpdate": "", "FileReplacements": {},
            "ManipulationString": ""}
    folded = {}
    if os.path.exists(info_path):                    # :431-435
        old = json.load(open(info_path, encoding="utf-8-sig"))
        info["FormatVersion"] = _ci_get(old, "FormatVersion", 1)
        info["SourceActor"] = _ci_get(old, "SourceActor", "")   # a JSON null stays null (Newtonsoft assigns it)
        for k, v in (_ci_get(old, "FileReplacements", {}) or {}).items():
            _ignore_case_set(folded, k, v)
    glam = {"Entries": []}
    if os.path.exists(glam_path):
        for e in (_ci_get(json.load(open(glam_path, encoding="utf-8-sig")), "Entries", []) or []):
            glam["Entries"].append({k: _ci_get(e, k, "") for k in ("Timestamp", "Description", "GlamourerString")})
    cust = {"Entries": []}
    if os.path.exists(cust_path):
        for e in (_ci_get(json.load(open(cust_path, encoding="utf-8-sig")), "Entries", []) or []):
            cust["Entries"].append({k: _ci_get(e, k, "")
                                    for k in ("Timestamp", "Description", "CustomizeData", "CustomizeTemplate")})
    for game_path, h in new_replacements.items():    # :453-475
        _ignore_case_set(folded, game_path, h)
        target = os.path.join(files_dir, h + ".dat")
        if not os.path.exists(target):
            source = resolved_paths.get(h)
            if source and os.path.isfile(source):
                with open(source, "rb") as f, open(target, "wb") as g:
                    g.write(f.read())
    info["FileReplacements"] = {k: v for k, v in folded.values()}
    info["ManipulationString"] = manipulation        # :477
    if not glam["Entries"] or glam["Entries"][-1]["GlamourerString"] != glamourer:   # :479-492
        glam["Entries"].append({"Timestamp": "<now>", "Description": "Glamourer Update - <now> UTC",
                                "GlamourerString": glamourer})
    b64 = base64.b64encode(customize.encode("utf-8")).decode("ascii") if customize else ""   # :494-496
    if (not cust["Entries"] or cust["Entries"][-1]["CustomizeData"] != b64) and b64:         # :497-511
        cust["Entries"].append({"Timestamp": "<now>", "Description": "Customize+ Update - <now> UTC",
                                "CustomizeData": b64, "CustomizeTemplate": ""})
    info["LastUpdate"] = "<now>"
    for path, obj in ((info_path, info), (glam_path, glam), (cust_path, cust)):     # :672-695
        with open(path, "w", encoding="utf-8") as f:
            text = json.dumps(obj, indent=2, ensure_ascii=False)
            # JsonConvert also escapes these three (StringEscapeHandling.Default)
            for ch in ("\u0085", "\u2028", "\u2029"):
                text = text.replace(ch, "\\u%04x" % ord(ch))
            f.write(text)
    return snapshot_dir
