"""Randomised soak of the host-pointer entry points (hash-while-copying, staging ring, planner):
many batches of random shape from one or two threads, every digest checked against hashlib.
Prints a progress line every few batches.  Usage: python tools/soak_host_path.py [seconds]"""
import hashlib
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import snappy_amd as S

    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 90.0
    pool = np.random.default_rng(1).integers(0, 256, size=600 << 20, dtype=np.uint8)
    failures = []
    counts = [0, 0]
    t_end = time.time() + budget

    def worker(tid):
        rng = np.random.default_rng(100 + tid)
        while time.time() < t_end and not failures:
            kind = rng.integers(0, 4)
            if kind == 0:      # many small
                n = int(rng.integers(1, 40000))
                lens = rng.integers(0, 20000, size=n)
            elif kind == 1:    # few large
                n = int(rng.integers(1, 12))
                lens = rng.integers(1 << 20, 40 << 20, size=n)
            elif kind == 2:    # mixed
                n = int(rng.integers(2, 3000))
                lens = np.exp(rng.uniform(np.log(16), np.log(8 << 20), size=n)).astype(np.int64)
            else:              # tiny batch
                n = int(rng.integers(1, 4))
                lens = rng.integers(0, 300000, size=n)
            lens = lens.astype(np.uint64)
            while int(lens.sum()) > pool.size - 64:
                lens = lens[: max(1, len(lens) // 2)]
            n = len(lens)
            if rng.random() < 0.5:   # packed back to back
                offs = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.uint64)
            else:                    # anywhere, overlapping allowed
                offs = (rng.random(n) * (pool.size - lens.astype(np.float64) - 1)).astype(np.uint64)
            os.environ["SNAPPY_OVERLAP_MIN_BYTES"] = str(int(rng.choice([0, 1 << 20, 64 << 20])))
            os.environ["SNAPPY_OVERLAP_SLICE_BYTES"] = str(int(rng.choice([4096, 1 << 20, 32 << 20])))
            try:
                got = S.sha1_batch(pool, offs, lens)
            except Exception as e:  # noqa: BLE001
                failures.append((tid, "exception", repr(e), n, int(lens.sum())))
                return
            for i in rng.integers(0, n, size=min(n, 40)):
                want = hashlib.sha1(pool[int(offs[i]): int(offs[i] + lens[i])].tobytes()).digest()
                if got[i].tobytes() != want:
                    failures.append((tid, "digest", int(i), n, int(lens[i])))
                    return
            counts[tid] += 1
            if counts[tid] % 10 == 0:
                print(f"thread {tid}: {counts[tid]} batches ok, {time.time() - (t_end - budget):.0f} s", flush=True)

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    print("batches", counts, "failures", failures[:3])
    sys.exit(1 if failures else 0)


if __name__ == "__main__":
    main()
