"""Python face of the MCDF import data path
(/root/reference/Snappy/Features/Mcdf/McdfManager.cs:28-84, McdfHeader.cs:26-62),
marshalled to the C ABI."""
from __future__ import annotations

import ctypes

import numpy as np

from . import native


def _as_u8(data) -> np.ndarray:
    if isinstance(data, np.ndarray):
        return np.ascontiguousarray(data, dtype=np.uint8).reshape(-1)
    return np.frombuffer(bytes(data), dtype=np.uint8)


def lz4net_decoded_size(stream) -> int:
    buf = _as_u8(stream)
    v = ctypes.c_uint64(0)
    native.check(native.lib().snappy_lz4net_decoded_size(buf.ctypes.data if buf.size else None,
                                                         buf.size, ctypes.byref(v)),
                 "snappy_lz4net_decoded_size")
    return v.value


def lz4net_decode(stream, out_cap=None) -> bytes:
    """`lz4Stream.CopyTo(memoryStream)`: the decoded bytes of an lz4net stream."""
    buf = _as_u8(stream)
    if out_cap is None:
        try:
            out_cap = lz4net_decoded_size(buf)
        except native.SnappyError:
            # the decode call reports the error in stream order; leave room for
            # whatever precedes the bad chunk (LZ4 expands at most 255x)
            out_cap = min(255 * buf.size + (1 << 16), 1 << 31)
    out = np.zeros(max(out_cap, 1), dtype=np.uint8)
    v = ctypes.c_uint64(0)
    native.check(native.lib().snappy_lz4net_decode(buf.ctypes.data if buf.size else None, buf.size,
                                                   out.ctypes.data, out_cap, ctypes.byref(v)),
                 "snappy_lz4net_decode")
    return out[: v.value].tobytes()


def mcdf_parse_header(payload):
    """Returns (lengths, body_offset) of a decoded MCDF payload."""
    buf = _as_u8(payload)
    n = ctypes.c_uint64(0)
    body = ctypes.c_uint64(0)
    ptr = buf.ctypes.data if buf.size else None
    native.check(native.lib().snappy_mcdf_parse_header(ptr, buf.size, ctypes.byref(n),
                                                       ctypes.byref(body), None, 0),
                 "snappy_mcdf_parse_header")
    lengths = np.zeros(max(n.value, 1), dtype=np.int64)
    native.check(native.lib().snappy_mcdf_parse_header(ptr, buf.size, ctypes.byref(n),
                                                       ctypes.byref(body), lengths.ctypes.data,
                                                       lengths.size),
                 "snappy_mcdf_parse_header")
    return lengths[: n.value].copy(), body.value


def mcdf_decode_and_hash(stream, want_payload: bool = False, n_files_cap: int = 1024):
    """Fused decode + slice + hash.  Returns dict(digests, statuses, body_offset, payload_len, payload)."""
    buf = _as_u8(stream)
    lib = native.lib()
    while True:
        digests = np.zeros((max(n_files_cap, 1), 20), dtype=np.uint8)
        statuses = np.zeros(max(n_files_cap, 1), dtype=np.uint8)
        n = ctypes.c_uint64(0)
        body = ctypes.c_uint64(0)
        plen = ctypes.c_uint64(0)
        payload = None
        cap = 0
        if want_payload:
            cap = lz4net_decoded_size(buf)
            payload = np.zeros(max(cap, 1), dtype=np.uint8)
        st = lib.snappy_mcdf_decode_and_hash(buf.ctypes.data if buf.size else None, buf.size,
                                             digests.ctypes.data, statuses.ctypes.data,
                                             n_files_cap, ctypes.byref(n), ctypes.byref(body),
                                             payload.ctypes.data if want_payload else None, cap,
                                             ctypes.byref(plen))
        if st == native.E_CAPACITY and n.value > n_files_cap:
            n_files_cap = n.value
            continue
        native.check(st, "snappy_mcdf_decode_and_hash")
        return {
            "digests": digests[: n.value].copy(),
            "statuses": statuses[: n.value].copy(),
            "body_offset": body.value,
            "payload_len": plen.value,
            "payload": payload[: plen.value].tobytes() if want_payload else None,
        }


def mcdf_decode_and_hash_batch(streams, n_files_cap: int = 4096):
    """Several archives through one decode launch and one hash launch.
    Returns a list of dicts (status, digests, statuses, body_offset, payload_len), one per archive."""
    bufs = [_as_u8(s) for s in streams]
    n = len(bufs)
    lib = native.lib()
    ptrs = (ctypes.c_void_p * max(n, 1))(*[b.ctypes.data if b.size else None for b in bufs])
    lens = np.array([b.size for b in bufs], dtype=np.uint64)
    while True:
        digests = np.zeros((max(n_files_cap, 1), 20), dtype=np.uint8)
        statuses = np.zeros(max(n_files_cap, 1), dtype=np.uint8)
        n_files = np.zeros(max(n, 1), dtype=np.uint64)
        body = np.zeros(max(n, 1), dtype=np.uint64)
        plen = np.zeros(max(n, 1), dtype=np.uint64)
        ast = np.zeros(max(n, 1), dtype=np.int32)
        st = lib.snappy_mcdf_decode_and_hash_batch(n, ptrs, lens.ctypes.data, digests.ctypes.data,
                                                   statuses.ctypes.data, n_files_cap,
                                                   n_files.ctypes.data, body.ctypes.data,
                                                   plen.ctypes.data, ast.ctypes.data)
        if st == native.E_CAPACITY and int(n_files.sum()) > n_files_cap:
            n_files_cap = int(n_files.sum())
            continue
        native.check(st, "snappy_mcdf_decode_and_hash_batch")
        break
    out = []
    at = 0
    for a in range(n):
        k = int(n_files[a])
        out.append({"status": int(ast[a]), "digests": digests[at: at + k].copy(),
                    "statuses": statuses[at: at + k].copy(), "body_offset": int(body[a]),
                    "payload_len": int(plen[a])})
        at += k
    return out


def lz4net_encode(data, block_size: int = 0) -> bytes:
    """The lz4net LZ4Stream (compress) framing of `data`, blocks compressed on the GPU
    (block_size 0 = lz4net's default of 1 MiB).  Inverse of lz4net_decode."""
    buf = _as_u8(data)
    lib = native.lib()
    bound = ctypes.c_uint64(0)
    native.check(lib.snappy_lz4net_encode_bound(buf.size, block_size, ctypes.byref(bound)),
                 "snappy_lz4net_encode_bound")
    out = np.zeros(max(bound.value, 1), dtype=np.uint8)
    n = ctypes.c_uint64(0)
    native.check(lib.snappy_lz4net_encode(buf.ctypes.data if buf.size else None, buf.size, block_size,
                                          out.ctypes.data, bound.value, ctypes.byref(n)),
                 "snappy_lz4net_encode")
    return out[: n.value].tobytes()


def mcdf_write_header(files, description="", glamourer_data="", customize_plus_data="",
                      manipulation_data="") -> bytes:
    """McdfHeader.WriteToStream: "MCDF", version, length, compact JSON.  `files` is a list of
    (game_paths, length) pairs."""
    flat = [p.encode("utf-8") for paths, _ in files for p in paths]
    arr = (ctypes.c_char_p * max(len(flat), 1))(*flat)
    per = np.array([len(paths) for paths, _ in files] or [0], dtype=np.uint32)
    lens = np.array([length for _, length in files] or [0], dtype=np.int64)
    lib = native.lib()
    n = ctypes.c_uint64(0)
    args = (description.encode("utf-8"), glamourer_data.encode("utf-8"), customize_plus_data.encode("utf-8"),
            manipulation_data.encode("utf-8"), arr, per.ctypes.data, lens.ctypes.data, len(files))
    st = lib.snappy_mcdf_write_header(*args, None, 0, ctypes.byref(n))
    if st not in (0, native.E_CAPACITY):
        native.check(st, "snappy_mcdf_write_header")
    out = np.zeros(max(n.value, 1), dtype=np.uint8)
    native.check(lib.snappy_mcdf_write_header(*args, out.ctypes.data, out.size, ctypes.byref(n)),
                 "snappy_mcdf_write_header")
    return out[: n.value].tobytes()


def write_mcdf(files, bodies, block_size: int = 0, **header_fields) -> bytes:
    """A whole archive: header (lengths taken from the bodies) + file bodies, lz4net-framed."""
    head = mcdf_write_header([(paths, len(body)) for paths, body in zip(files, bodies)], **header_fields)
    return lz4net_encode(head + b"".join(bytes(b) for b in bodies), block_size)
