"""The GPU LZ4 compressor by kind of content: kernel time, ratio and a digest of the stream it wrote
(the same build gives the same bytes every time; a change that is meant to keep the parse must keep
the digests).  Usage: python tools/probe_lz4_encode_kinds.py [MiB per kind, default 1024]"""
import ctypes
import hashlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import bench
    from snappy_amd import native

    mib = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    n = mib << 20
    lib = native.lib()
    native.check(lib.snappy_init(0), "init")
    lib.snappy_profile_enable(1)
    rng = np.random.default_rng(5)

    def motif(lo, hi, reps_hi):
        unit = min(n, 64 << 20)
        parts, total = [], 0
        while total < unit:
            m = rng.integers(0, 256, size=int(rng.integers(lo, hi)), dtype=np.uint8)
            seg = np.tile(m, int(rng.integers(2, reps_hi)))
            parts.append(seg)
            total += seg.size
        return np.tile(np.concatenate(parts)[:unit], n // unit)

    kinds = {
        "asset-like mix": lambda: bench.asset_like_bytes(n, rng),
        "lz-like lit<=24 match<=36": lambda: bench.dense_bytes(n, rng, 24, 36, 65535),
        "lz-like lit<=100 match<=200": lambda: bench.dense_bytes(n, rng, 100, 200, 65535),
        "random (stored chunks)": lambda: rng.integers(0, 256, size=n, dtype=np.uint8),
        "zeros": lambda: np.zeros(n, dtype=np.uint8),
        "motifs 16-64 B x <80": lambda: motif(16, 64, 80),
        "motifs 1-4 KiB x <80": lambda: motif(1024, 4096, 80),
        "ragged end (n - 12345 bytes of the mix)": lambda: bench.asset_like_bytes(n, rng)[: n - 12345],
    }
    for name, make in kinds.items():
        data = np.ascontiguousarray(make())
        bound = ctypes.c_uint64(0)
        lib.snappy_lz4net_encode_bound(data.size, 0, ctypes.byref(bound))
        stream = np.zeros(bound.value, dtype=np.uint8)
        w = ctypes.c_uint64(0)
        ms = []
        for _ in range(3):
            native.check(lib.snappy_lz4net_encode(data.ctypes.data, data.size, 0, stream.ctypes.data, stream.size,
                                                  ctypes.byref(w)), "encode")
            ms.append(lib.snappy_profile_last_ms(1))
        digest = hashlib.sha1(stream[: w.value].tobytes()).hexdigest()
        back = np.zeros(data.size + 4096, dtype=np.uint8)
        v = ctypes.c_uint64(0)
        native.check(lib.snappy_lz4net_decode(stream.ctypes.data, w.value, back.ctypes.data, back.size,
                                              ctypes.byref(v)), "decode")
        ok = v.value == data.size and np.array_equal(back[: data.size], data)
        k = min(ms)
        print(f"{name:42s} ratio {data.size / w.value:8.2f}  encode kernel {k:7.2f} ms  {data.size / k / 1e6:7.1f} GB/s in  "
              f"stream sha1 {digest[:16]}  roundtrip {'ok' if ok else 'WRONG'}", flush=True)


if __name__ == "__main__":
    main()
