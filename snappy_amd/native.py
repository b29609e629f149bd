"""ctypes binding of libsnappyhash.so (the C ABI declared in include/snappyhash.h).

The binding only marshals pointers and sizes.  There is no Python or CPU
implementation behind it: if the shared library is missing, loading fails, and
if no gfx950 device is usable every compute call raises SnappyError with
SNAPPY_E_NO_DEVICE.
"""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SNAPPY_LIB_PATH: a variant build of the same library (measurement builds of the tools).
LIB_PATH = os.environ.get("SNAPPY_LIB_PATH") or os.path.join(_HERE, "libsnappyhash.so")

OK = 0
E_INVALID_ARG = -1
E_NO_DEVICE = -2
E_HIP = -3
E_NOMEM = -4
E_LZ4_TRUNCATED = -10
E_LZ4_BAD_LENGTHS = -11
E_LZ4_PASSES = -12
E_LZ4_CORRUPT = -13
E_CAPACITY = -14
E_MCDF_MAGIC = -20
E_MCDF_VERSION = -21
E_MCDF_TRUNCATED = -22
E_MCDF_JSON = -23
E_MCDF_BASE64 = -24
E_SNAPSHOT_JSON = -30

FILE_HASHED = 0
FILE_SKIPPED_EMPTY = 1
FILE_SKIPPED_SHORT_READ = 2

# Every symbol include/snappyhash.h declares: (name, restype, argtypes).
_vp, _u64, _i = ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int
_u64p = ctypes.POINTER(ctypes.c_uint64)
SYMBOLS = [
    ("snappy_init", _i, [_i]),
    ("snappy_shutdown", None, []),
    ("snappy_device_count", _i, []),
    ("snappy_host_alloc", _i, [_u64, ctypes.POINTER(ctypes.c_void_p)]),
    ("snappy_host_free", _i, [_vp]),
    ("snappy_strerror", ctypes.c_char_p, [_i]),
    ("snappy_last_error_detail", ctypes.c_char_p, []),
    ("snappy_sha1", _i, [_vp, _u64, _vp]),
    ("snappy_sha1_hex", _i, [_vp, _u64, _vp]),
    ("snappy_sha1_batch", _i, [_vp, _vp, _vp, _u64, _vp]),
    ("snappy_sha1_files", _i, [_vp, _vp, _u64, _vp]),
    ("snappy_digests_to_hex", _i, [_vp, _u64, _vp]),
    ("snappy_sha1_batch_device", _i, [_i, _vp, _vp, _vp, _u64, _vp, _vp]),
    ("snappy_sha1_batch_resident", _i, [_i, _vp, _vp, _vp, _u64, _vp, _vp]),
    ("snappy_sha1_batch_resident_noalloc", _i, [_i, _vp, _vp, _vp, _u64, _vp, _vp]),
    ("snappy_plan_shards_lpt", _i, [_vp, _u64, ctypes.c_uint32, _vp, _vp]),
    ("snappy_device_work", _i, [_i, _u64p, _u64p]),
    ("snappy_plan_shards", _i, [_vp, _u64, ctypes.c_uint32, _vp]),
    ("snappy_lz4net_decoded_size", _i, [_vp, _u64, _u64p]),
    ("snappy_lz4net_decode", _i, [_vp, _u64, _vp, _u64, _u64p]),
    ("snappy_mcdf_parse_header", _i, [_vp, _u64, _u64p, _u64p, _vp, _u64]),
    ("snappy_mcdf_decode_and_hash", _i, [_vp, _u64, _vp, _vp, _u64, _u64p, _u64p, _vp, _u64, _u64p]),
    ("snappy_mcdf_decode_and_hash_batch", _i, [_u64, _vp, _vp, _vp, _vp, _u64, _vp, _vp, _vp, _vp]),
    ("snappy_store_files", _i, [_vp, _u64, ctypes.c_char_p, _vp, _vp]),
    ("snappy_update_snapshot", _i, [ctypes.c_char_p, ctypes.c_char_p, _vp, _vp, _vp, _u64, ctypes.c_char_p,
                                    ctypes.c_char_p, ctypes.c_char_p, _vp, _vp]),
    ("snappy_mcdf_import_to_dir", _i, [ctypes.c_char_p, ctypes.c_char_p, _vp, _u64]),
    ("snappy_mcdf_write_header", _i, [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p,
                                      _vp, _vp, _vp, _u64, _vp, _u64, _u64p]),
    ("snappy_lz4net_encode_bound", _i, [_u64, ctypes.c_uint32, _u64p]),
    ("snappy_lz4net_encode", _i, [_vp, _u64, ctypes.c_uint32, _vp, _u64, _u64p]),
    ("snappy_profile_enable", None, [_i]),
    ("snappy_profile_last_ms", ctypes.c_float, [_i]),
    ("snappy_sha1_cost_model", _i, [_i, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double),
                                    ctypes.POINTER(ctypes.c_int)]),
    ("snappy_sha1_last_plan", _i, [ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_double)]),
]


class SnappyError(RuntimeError):
    def __init__(self, code: int, where: str, detail: str = ""):
        self.code = code
        msg = f"{where}: {strerror(code)} ({code})"
        if detail:
            msg += f" [{detail}]"
        super().__init__(msg)


_lib = None


def lib() -> ctypes.CDLL:
    """Loads the shared library; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -m snappy_amd.build` "
                "(there is no fallback implementation)")
        # PyTorch ships its own copy of the HIP runtime (same soname).  Two copies in
        # one process cannot both open the GPU, so when torch is importable it goes
        # first and this library binds to the runtime torch brought in; device
        # pointers and streams from torch are then valid here.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = ctypes.CDLL(LIB_PATH)
        for name, restype, argtypes in SYMBOLS:
            fn = getattr(L, name)  # AttributeError if the library lacks a declared symbol
            fn.restype = restype
            fn.argtypes = argtypes
        _lib = L
    return _lib


def strerror(code: int) -> str:
    return lib().snappy_strerror(code).decode("ascii")


def check(code: int, where: str) -> None:
    if code != OK:
        detail = lib().snappy_last_error_detail().decode("utf-8", "replace") if code == E_HIP else ""
        raise SnappyError(code, where, detail)
