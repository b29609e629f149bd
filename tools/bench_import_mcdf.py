"""Wall time of snappy_mcdf_import_to_dir (file -> snapshot folder) next to its parts, for one
synthetic archive.  Usage: python tools/bench_import_mcdf.py [MiB] [files]"""
import ctypes
import os
import shutil
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import bench
    import snappy_amd
    from snappy_amd import native

    mib = float(sys.argv[1]) if len(sys.argv) > 1 else 256.0
    n_files = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    lib = native.lib()
    native.check(lib.snappy_init(0), "init")
    lib.snappy_profile_enable(1)
    stream, size, _ = bench.make_mcdf_archive(int(mib * (1 << 20)), n_files, np.random.default_rng(4), "bench")
    root = tempfile.mkdtemp(prefix="snappy_import_")
    try:
        path = os.path.join(root, "a.mcdf")
        with open(path, "wb") as f:
            f.write(stream.tobytes())
        times = []
        for rep in range(3):
            work = os.path.join(root, f"w{rep}")
            os.makedirs(work)
            t0 = time.perf_counter()
            snappy_amd.import_mcdf(path, work)
            times.append(time.perf_counter() - t0)
        sha_ms, lz4_ms = lib.snappy_profile_last_ms(0), lib.snappy_profile_last_ms(1)
        # the same archive through the in-memory entry (no file system)
        t0 = time.perf_counter()
        snappy_amd.mcdf_decode_and_hash(stream.tobytes())
        mem = time.perf_counter() - t0
        print(f"archive {stream.size / 1e6:.0f} MB -> {size / 1e6:.0f} MB in {n_files} files: import_to_dir "
              f"{[round(t * 1e3) for t in times]} ms (SHA-1 kernel {sha_ms:.0f} ms, LZ4 kernel {lz4_ms:.2f} ms); "
              f"decode_and_hash from memory {mem * 1e3:.0f} ms", flush=True)
    finally:
        shutil.rmtree(root, ignore_errors=True)


if __name__ == "__main__":
    main()
