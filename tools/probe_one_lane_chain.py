"""Chain time per block on the one-lane path (one file per lane, 64 per wave) by file size and by
how many SIMDs are busy: n = 64 x W equal files, every file forced onto the one-lane path.
Prints kernel ms / blocks per file = ns per block of a lane's chain."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch

    import bench
    from snappy_amd import native

    lib = native.lib()
    native.check(lib.snappy_init(0), "init")
    lib.snappy_profile_enable(1)
    os.environ["SNAPPY_SHA1_FORCE_PATH"] = sys.argv[1] if len(sys.argv) > 1 else "3"
    per_wave = {"3": 64, "2": 16, "1": 4, "0": 1}[os.environ["SNAPPY_SHA1_FORCE_PATH"]]
    cap = 200 << 30
    buf = bench.fill_random(torch, cap, torch.device("cuda", 0), seed=1)
    for size in (8 << 10, 32 << 10, 128 << 10, 512 << 10, 1 << 20, 2 << 20, 4 << 20, 8 << 20, 16 << 20):
        row = []
        for waves in (64, 256, 1024, 2048):
            n = per_wave * waves
            if n * size > cap:
                continue
            lengths = np.full(n, size - 9, dtype=np.int64)   # pads to exactly size / 64 blocks
            offsets = np.arange(n, dtype=np.int64) * size
            d_off = torch.from_numpy(offsets).cuda()
            d_len = torch.from_numpy(lengths).cuda()
            d_dig = torch.zeros(n * 20, dtype=torch.uint8, device="cuda")
            ms = []
            for _ in range(3):
                native.check(lib.snappy_sha1_batch_resident(0, buf.data_ptr(), d_off.data_ptr(), d_len.data_ptr(), n,
                                                            d_dig.data_ptr(), None), "resident")
                ms.append(lib.snappy_profile_last_ms(0))
            row.append(f"{waves} waves: {min(ms) * 1e6 / (size // 64):7.0f} ns/block ({min(ms):.2f} ms)")
        print(f"path {os.environ['SNAPPY_SHA1_FORCE_PATH']}  {size >> 10:6d} KiB files   " + "   ".join(row), flush=True)


if __name__ == "__main__":
    main()
