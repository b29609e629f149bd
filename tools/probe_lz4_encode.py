"""Throughput and ratio of the GPU LZ4 compressor on asset-like data.
Usage: python tools/probe_lz4_encode.py [MiB]"""
import ctypes
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import bench
    from oracle import oracle
    from snappy_amd import native

    mib = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    lib = native.lib()
    native.check(lib.snappy_init(0), "init")
    lib.snappy_profile_enable(1)
    data = bench.asset_like_bytes(mib << 20, np.random.default_rng(3))
    bound = ctypes.c_uint64(0)
    lib.snappy_lz4net_encode_bound(data.size, 0, ctypes.byref(bound))
    out = np.zeros(bound.value, dtype=np.uint8)
    n = ctypes.c_uint64(0)
    for rep in range(3):
        t0 = time.perf_counter()
        native.check(lib.snappy_lz4net_encode(data.ctypes.data, data.size, 0, out.ctypes.data, out.size,
                                              ctypes.byref(n)), "encode")
        dt = time.perf_counter() - t0
        k = lib.snappy_profile_last_ms(1)
        print(f"GPU encode {mib} MiB: ratio {data.size / n.value:.3f}  kernel {k:.2f} ms "
              f"({data.size / k / 1e6:.1f} GB/s in)  call {dt * 1e3:.0f} ms", flush=True)
    sample = data[: 64 << 20]
    t0 = time.perf_counter()
    ref = oracle.lz4net_encode(sample.tobytes(), 1 << 20, False)
    dt = time.perf_counter() - t0
    print(f"CPU oracle encoder, 1 thread, 64 MiB: ratio {sample.size / len(ref):.3f}  {sample.size / dt / 1e6:.0f} MB/s")
    st, back = oracle.lz4net_decode(out[: n.value][: 1].tobytes() and out[: n.value].tobytes())
    print("oracle decodes GPU stream:", st == 0 and back == data.tobytes())


if __name__ == "__main__":
    main()
