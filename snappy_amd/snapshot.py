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


def update_snapshot(snapshot_dir, source_actor, replacements, glamourer="", customize="", manipulation=""):
    """UpdateSnapshot for the local player: replacements = [(resolved_path, [game paths])].
    Returns a list with each file's HASH, or None where the resolved path names no file."""
    n = len(replacements)
    res = (ctypes.c_char_p * max(n, 1))(*[os.fsencode(r) for r, _ in replacements])
    flat = [g.encode("utf-8") for _, paths in replacements for g in paths]
    gps = (ctypes.c_char_p * max(len(flat), 1))(*flat)
    per = np.array([len(paths) for _, paths in replacements], dtype=np.uint32)
    hexbuf = ctypes.create_string_buffer(41 * max(n, 1))
    status = np.zeros(max(n, 1), dtype=np.uint8)
    native.check(native.lib().snappy_update_snapshot(
        os.fsencode(snapshot_dir), source_actor.encode("utf-8"), res, gps, per.ctypes.data, n,
        glamourer.encode("utf-8"), customize.encode("utf-8"), manipulation.encode("utf-8"),
        hexbuf, status.ctypes.data), "snappy_update_snapshot")
    raw = hexbuf.raw
    return [raw[41 * i: 41 * i + 40].decode("ascii") if status[i] == 0 else None for i in range(n)]


def import_mcdf(mcdf_path, working_dir) -> str:
    """ImportMcdf: writes the snapshot folder and returns its path."""
    out = ctypes.create_string_buffer(4096)
    native.check(native.lib().snappy_mcdf_import_to_dir(os.fsencode(mcdf_path), os.fsencode(working_dir),
                                                        out, 4096), "snappy_mcdf_import_to_dir")
    return os.fsdecode(out.value)
