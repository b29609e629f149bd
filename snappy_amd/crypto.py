"""Python face of the hashing helper: same names and results as
/root/reference/Snappy/Utils/Crypto.cs:8-22, marshalled to the C ABI."""
from __future__ import annotations

import ctypes

import numpy as np

from . import native


def _as_u8(data) -> np.ndarray:
    if isinstance(data, np.ndarray):
        return np.ascontiguousarray(data, dtype=np.uint8).reshape(-1)
    return np.frombuffer(bytes(data), dtype=np.uint8)


def sha1(data) -> bytes:
    """20-byte SHA-1 of a bytes-like object (Crypto.cs:10-11)."""
    buf = _as_u8(data)
    out = (ctypes.c_uint8 * 20)()
    native.check(native.lib().snappy_sha1(buf.ctypes.data if buf.size else None, buf.size, out),
                 "snappy_sha1")
    return bytes(out)


def get_hash(data) -> str:
    """Crypto.GetHash: 40 uppercase hex characters."""
    buf = _as_u8(data)
    out = ctypes.create_string_buffer(41)
    native.check(native.lib().snappy_sha1_hex(buf.ctypes.data if buf.size else None, buf.size, out),
                 "snappy_sha1_hex")
    return out.value.decode("ascii")


def digests_to_hex(digests: np.ndarray) -> list:
    d = np.ascontiguousarray(digests, dtype=np.uint8).reshape(-1, 20)
    n = d.shape[0]
    out = ctypes.create_string_buffer(41 * max(n, 1))
    native.check(native.lib().snappy_digests_to_hex(d.ctypes.data, n, out), "snappy_digests_to_hex")
    raw = out.raw
    return [raw[41 * i: 41 * i + 40].decode("ascii") for i in range(n)]


def sha1_batch(base, offsets, lengths) -> np.ndarray:
    """digests[i] = SHA-1(base[offsets[i] : offsets[i]+lengths[i]]); returns (n, 20) uint8."""
    buf = _as_u8(base)
    off = np.ascontiguousarray(offsets, dtype=np.uint64)
    ln = np.ascontiguousarray(lengths, dtype=np.uint64)
    n = off.shape[0]
    if ln.shape[0] != n:
        raise ValueError("offsets and lengths differ in length")
    if n and int((off + ln).max()) > buf.size:
        raise ValueError("a range runs past the end of base")
    out = np.zeros((n, 20), dtype=np.uint8)
    native.check(native.lib().snappy_sha1_batch(buf.ctypes.data if buf.size else None,
                                                off.ctypes.data, ln.ctypes.data, n,
                                                out.ctypes.data), "snappy_sha1_batch")
    return out


def sha1_files(blobs) -> np.ndarray:
    """One digest per bytes-like object, each in its own buffer (File.ReadAllBytes per file)."""
    arrays = [_as_u8(b) for b in blobs]
    n = len(arrays)
    ptrs = (ctypes.c_void_p * max(n, 1))(*[a.ctypes.data if a.size else None for a in arrays])
    lens = np.array([a.size for a in arrays], dtype=np.uint64)
    out = np.zeros((n, 20), dtype=np.uint8)
    native.check(native.lib().snappy_sha1_files(ptrs, lens.ctypes.data, n, out.ctypes.data),
                 "snappy_sha1_files")
    return out


def sha1_batch_device(device: int, d_base: int, offsets, lengths, stream: int = 0) -> np.ndarray:
    """Same as sha1_batch for data resident in HBM: d_base is a device address."""
    off = np.ascontiguousarray(offsets, dtype=np.uint64)
    ln = np.ascontiguousarray(lengths, dtype=np.uint64)
    n = off.shape[0]
    out = np.zeros((n, 20), dtype=np.uint8)
    native.check(native.lib().snappy_sha1_batch_device(device, d_base, off.ctypes.data,
                                                       ln.ctypes.data, n, out.ctypes.data,
                                                       stream or None),
                 "snappy_sha1_batch_device")
    return out


def plan_shards(lengths, n_shards: int) -> np.ndarray:
    """Contiguous shards of near-equal byte count; returns n_shards + 1 cut indices."""
    ln = np.ascontiguousarray(lengths, dtype=np.uint64)
    cuts = np.zeros(n_shards + 1, dtype=np.uint64)
    native.check(native.lib().snappy_plan_shards(ln.ctypes.data, ln.shape[0], n_shards,
                                                 cuts.ctypes.data), "snappy_plan_shards")
    return cuts


def plan_shards_lpt(lengths, n_shards: int):
    """Longest-first dealing by padded blocks; returns (shard_of[n] uint32, blocks[n_shards] uint64)."""
    ln = np.ascontiguousarray(lengths, dtype=np.uint64)
    shard_of = np.zeros(ln.shape[0], dtype=np.uint32)
    blocks = np.zeros(n_shards, dtype=np.uint64)
    native.check(native.lib().snappy_plan_shards_lpt(ln.ctypes.data, ln.shape[0], n_shards,
                                                     shard_of.ctypes.data, blocks.ctypes.data),
                 "snappy_plan_shards_lpt")
    return shard_of, blocks


def device_work(device: int):
    """(files or archives, bytes) the host-pointer entry points have given a device so far."""
    u, b = ctypes.c_uint64(0), ctypes.c_uint64(0)
    native.check(native.lib().snappy_device_work(device, ctypes.byref(u), ctypes.byref(b)),
                 "snappy_device_work")
    return u.value, b.value


def sha1_batch_resident(device: int, d_base: int, d_offsets: int, d_lengths: int, n: int,
                        d_digests: int, stream: int = 0) -> None:
    """Everything in HBM: device addresses of the data, the two uint64 tables and the 20*n digest bytes."""
    native.check(native.lib().snappy_sha1_batch_resident(device, d_base, d_offsets, d_lengths, n,
                                                         d_digests, stream or None),
                 "snappy_sha1_batch_resident")
