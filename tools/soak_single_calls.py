"""Random single GetHash calls and tiny batches from two threads against hashlib.
Usage: python tools/soak_single_calls.py [seconds]"""
import hashlib
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import snappy_amd
    from snappy_amd.crypto import digests_to_hex

    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
    stop = time.time() + seconds
    counts, failures = [0, 0], []

    def work(t):
        rng = np.random.default_rng(1000 + t)
        while time.time() < stop:
            k = int(rng.integers(1, 4))
            sizes = [int(rng.choice([0, 1, 55, 56, 64, int(rng.integers(0, 300)), int(rng.integers(0, 200_000))]))
                     for _ in range(k)]
            blobs = [rng.integers(0, 256, size=s, dtype=np.uint8).tobytes() for s in sizes]
            want = [hashlib.sha1(b).hexdigest().upper() for b in blobs]
            if k == 1:
                got = [snappy_amd.get_hash(blobs[0])]
            else:
                got = digests_to_hex(snappy_amd.sha1_files(blobs))
            if got != want:
                failures.append((t, sizes))
            counts[t] += 1

    threads = [threading.Thread(target=work, args=(t,)) for t in range(2)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    print(f"calls {counts} failures {failures[:3]}", flush=True)
    return 1 if failures else 0


if __name__ == "__main__":
    sys.exit(main())
