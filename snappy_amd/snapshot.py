"""Python face of the headless callers: the hash-and-store loop
(/root/reference/Snappy/Core/SnapshotManager.cs:312-324, 453-475; Snappy/Data/SnapshotMigrator.cs:
62-86) and ImportMcdf (/root/reference/Snappy/Features/Mcdf/McdfManager.cs:68-156), marshalled to
the C ABI."""
from __future__ import annotations

import ctypes
import os

import numpy as np

from . import native


def store_files(paths, files_dir=None):
    """Hashes every existing file of `paths` (one GPU batch) and, when files_dir is given, stores
    each as files_dir/<HASH>.dat unless present.  Returns a list: HASH string, or None for a
    missing source file."""
    enc = [os.fsencode(p) for p in paths]
    n = len(enc)
    arr = (ctypes.c_char_p * max(n, 1))(*enc)
    hexbuf = ctypes.create_string_buffer(41 * max(n, 1))
    status = np.zeros(max(n, 1), dtype=np.uint8)
    native.check(native.lib().snappy_store_files(arr, n, os.fsencode(files_dir) if files_dir else None,
                                                 hexbuf, status.ctypes.data), "snappy_store_files")
    raw = hexbuf.raw
    return [raw[41 * i: 41 * i + 40].decode("ascii") if status[i] == 0 else None for i in range(n)]


def import_mcdf(mcdf_path, working_dir) -> str:
    """ImportMcdf: writes the snapshot folder and returns its path."""
    out = ctypes.create_string_buffer(4096)
    native.check(native.lib().snappy_mcdf_import_to_dir(os.fsencode(mcdf_path), os.fsencode(working_dir),
                                                        out, 4096), "snappy_mcdf_import_to_dir")
    return os.fsdecode(out.value)
