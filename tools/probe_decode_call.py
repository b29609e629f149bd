"""Wall time of snappy_lz4net_decode (host stream in, host payload out) against its kernel time."""
import ctypes
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import bench
    from snappy_amd import native

    lib = native.lib()
    native.check(lib.snappy_init(0), "init")
    lib.snappy_profile_enable(1)
    mib = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    data = bench.asset_like_bytes(mib << 20, np.random.default_rng(3))
    bound = ctypes.c_uint64(0)
    lib.snappy_lz4net_encode_bound(data.size, 0, ctypes.byref(bound))
    stream = np.zeros(bound.value, dtype=np.uint8)
    w = ctypes.c_uint64(0)
    native.check(lib.snappy_lz4net_encode(data.ctypes.data, data.size, 0, stream.ctypes.data, stream.size,
                                          ctypes.byref(w)), "encode")
    out = np.zeros(data.size, dtype=np.uint8)
    v = ctypes.c_uint64(0)
    for rep in range(4):
        t0 = time.perf_counter()
        native.check(lib.snappy_lz4net_decode(stream.ctypes.data, w.value, out.ctypes.data, out.size,
                                              ctypes.byref(v)), "decode")
        dt = time.perf_counter() - t0
        print(f"decode {mib} MiB: call {dt * 1e3:7.1f} ms ({data.size / dt / 1e9:5.1f} GB/s out)  "
              f"kernel {lib.snappy_profile_last_ms(1):.2f} ms  ok {bool(np.array_equal(out, data))}", flush=True)


if __name__ == "__main__":
    main()
