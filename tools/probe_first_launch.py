"""Is the first launch on a batch slower than the following ones, and why: idle clocks or first touch of the
data?  Mixed corpus at a scale; launches back to back, after an idle pause, and on a second buffer.
Usage: python tools/probe_first_launch.py [scale]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch

    import bench
    from snappy_amd import native

    scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.25
    lib = native.lib()
    native.check(lib.snappy_init(0), "init")
    lib.snappy_profile_enable(1)
    lengths = bench.make_sizes("corpus", np.random.default_rng(0x534E4150), scale)
    offsets = np.concatenate([[0], np.cumsum(lengths)[:-1]]).astype(np.uint64)
    total = int(lengths.sum())
    n = len(lengths)
    d_off = torch.from_numpy(offsets.astype(np.int64)).cuda()
    d_len = torch.from_numpy(lengths.astype(np.int64)).cuda()
    d_dig = torch.zeros(n * 20, dtype=torch.uint8, device="cuda")

    def launch(buf):
        native.check(lib.snappy_sha1_batch_resident(0, buf.data_ptr(), d_off.data_ptr(), d_len.data_ptr(), n,
                                                    d_dig.data_ptr(), None), "resident")
        return round(lib.snappy_profile_last_ms(0), 1)

    a = bench.fill_random(torch, total, torch.device("cuda", 0), seed=1)
    torch.cuda.synchronize()
    print(f"{total / 1e9:.1f} GB in {n} files")
    print("buffer A, right after it was filled:", [launch(a) for _ in range(4)], flush=True)
    time.sleep(3.0)
    print("buffer A, after 3 s of idling:      ", [launch(a) for _ in range(3)], flush=True)
    b = bench.fill_random(torch, total, torch.device("cuda", 0), seed=2)
    torch.cuda.synchronize()
    print("buffer B, right after it was filled:", [launch(b) for _ in range(3)], flush=True)
    print("buffer A again:                     ", [launch(a) for _ in range(2)], flush=True)
    b.copy_(a)
    torch.cuda.synchronize()
    print("buffer B after a device copy A->B:  ", [launch(b) for _ in range(2)], flush=True)


if __name__ == "__main__":
    main()
