"""Hash-and-store of one snapshot folder from the file system (the reference's save loop,
Snappy/Core/SnapshotManager.cs:312-324 + 453-475): the library's snappy_store_files against a
sequential CPU loop of the same shape (read whole file -> SHA-1 (OpenSSL via hashlib) -> copy
to _files/<HASH>.dat unless present).  Files are written first, so both sides read from the
page cache.  Usage: python tools/bench_store_folder.py [--mb 200] [--files 160]"""
import argparse
import hashlib
import json
import os
import shutil
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mb", type=float, default=200.0)
    ap.add_argument("--files", type=int, default=160)
    args = ap.parse_args()
    import snappy_amd

    rng = np.random.default_rng(5)
    n_tex = args.files * 3 // 4
    tex = np.exp(rng.uniform(np.log(64 << 10), np.log(22 << 20), size=n_tex))
    mdl = np.exp(rng.uniform(np.log(32 << 10), np.log(6 << 20), size=args.files - n_tex))
    sizes = np.concatenate([tex, mdl])
    sizes = np.maximum((sizes * (args.mb * 1e6 / sizes.sum())).astype(np.int64), 1)
    root = tempfile.mkdtemp(prefix="snappy_folder_")
    try:
        src = os.path.join(root, "mods")
        os.mkdir(src)
        paths = []
        for i, n in enumerate(sizes):
            p = os.path.join(src, f"f{i:04d}.tex")
            with open(p, "wb") as f:
                f.write(rng.integers(0, 256, size=int(n), dtype=np.uint8).tobytes())
            paths.append(p)
        total = int(sizes.sum())

        def cpu_loop(dst):
            os.makedirs(dst)
            out = []
            for p in paths:
                with open(p, "rb") as f:
                    data = f.read()
                h = hashlib.sha1(data).hexdigest().upper()
                target = os.path.join(dst, h + ".dat")
                if not os.path.exists(target):
                    shutil.copyfile(p, target)
                out.append(h)
            return out

        t0 = time.perf_counter()
        want = cpu_loop(os.path.join(root, "cpu_files"))
        t_cpu = time.perf_counter() - t0
        snappy_amd.store_files(paths[:2], os.path.join(root, "warm"))   # library + device set-up
        times = []
        for rep in range(3):
            dst = os.path.join(root, f"gpu_files_{rep}")
            t0 = time.perf_counter()
            got = snappy_amd.store_files(paths, dst)
            times.append(time.perf_counter() - t0)
            assert got == want
        t0 = time.perf_counter()
        snappy_amd.store_files(paths, None)
        t_hash_only = time.perf_counter() - t0
        print(json.dumps({
            "workload": f"{len(paths)} files, {total} bytes, largest {int(sizes.max())}",
            "cpu_sequential_loop": {"seconds": t_cpu, "GBps": total / t_cpu / 1e9,
                                    "what": "read -> hashlib.sha1 (OpenSSL) -> copy if absent, one thread"},
            "snappy_store_files": {"seconds": min(times), "GBps": total / min(times) / 1e9,
                                   "all_runs_s": times},
            "snappy_store_files_hash_only": {"seconds": t_hash_only, "GBps": total / t_hash_only / 1e9},
            "digests_equal": True}))
    finally:
        shutil.rmtree(root, ignore_errors=True)


if __name__ == "__main__":
    main()
