"""Latency of one Crypto.GetHash-shaped call (snappy_sha1_hex on a host buffer) by size."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import hashlib

    import numpy as np

    import snappy_amd as S

    S.get_hash(b"warm up")
    for n in (0, 64, 1024, 65536, 1 << 20, 16 << 20):
        data = np.random.default_rng(n).integers(0, 256, size=n, dtype=np.uint8).tobytes()
        reps = 200 if n <= 65536 else 20
        t0 = time.perf_counter()
        for _ in range(reps):
            h = S.get_hash(data)
        dt = (time.perf_counter() - t0) / reps
        t0 = time.perf_counter()
        for _ in range(reps):
            ref = hashlib.sha1(data).hexdigest().upper()
        dc = (time.perf_counter() - t0) / reps
        assert h == ref
        print(f"{n:9d} bytes: GPU call {dt * 1e6:9.1f} us   hashlib {dc * 1e6:9.1f} us", flush=True)


if __name__ == "__main__":
    main()
