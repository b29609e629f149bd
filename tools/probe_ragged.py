"""Cost of ragged wave ends on the one-lane path: 10^6 files whose sizes are (a) uniform in
4-32 KB, (b) the same total but quantised to one size per planning bucket (no ragged ends, same
order and locality), (c) the original sizes with every file starting on a 4/16/64/128-byte boundary."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(lib, native, torch, buf, lengths, label, align=1):
    n = len(lengths)
    padded = (lengths + align - 1) // align * align
    offsets = np.concatenate([[0], np.cumsum(padded)[:-1]]).astype(np.int64)
    d_off = torch.from_numpy(offsets).cuda()
    d_len = torch.from_numpy(lengths.astype(np.int64)).cuda()
    d_dig = torch.zeros(n * 20, dtype=torch.uint8, device="cuda")
    ms = []
    for _ in range(4):
        native.check(lib.snappy_sha1_batch_resident(0, buf.data_ptr(), d_off.data_ptr(), d_len.data_ptr(), n,
                                                    d_dig.data_ptr(), None), "resident")
        ms.append(lib.snappy_profile_last_ms(0))
    blocks = int(((lengths + 8) // 64 + 1).sum())
    print(f"{label:34s} {min(ms):6.3f} ms  {blocks * 64 / min(ms) / 1e6:7.1f} GB/s (padded bytes)", flush=True)


def main():
    import torch

    from snappy_amd import native

    lib = native.lib()
    native.check(lib.snappy_init(0), "init")
    lib.snappy_profile_enable(1)
    rng = np.random.default_rng(3)
    lengths = rng.integers(4096, 32769, size=1_000_000).astype(np.int64)
    buf = torch.randint(0, 256, (int(lengths.sum()) + 4096 + 128 * len(lengths),), dtype=torch.uint8, device="cuda")
    run(lib, native, torch, buf, lengths, "uniform sizes (ragged)")
    for a in (4, 16, 64, 128):
        run(lib, native, torch, buf, lengths, f"same, files aligned to {a} B", align=a)
    blocks = (lengths + 8) // 64 + 1
    e = np.floor(np.log2(blocks)).astype(np.int64)
    mant = (blocks >> (e - 3)) & 7
    floor_blocks = (8 + mant) << (e - 3)            # lower edge of the planning bucket
    quant = np.maximum((floor_blocks - 1) * 64 - 8, 1)
    run(lib, native, torch, buf, quant, "one size per bucket (no ragged ends)")
    run(lib, native, torch, buf, np.full(1_000_000, 18432, dtype=np.int64), "all 18 KiB")


if __name__ == "__main__":
    main()
