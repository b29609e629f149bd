"""Chain speed of the narrow SHA-1 paths on large files spread over a large footprint (every lane
streams from pages of its own): one wave per SIMD, all files equal.  Usage: python tools/probe_sha1_bigfiles.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch

    from snappy_amd import native

    lib = native.lib()
    native.check(lib.snappy_init(0), "init")
    lib.snappy_profile_enable(1)
    n_simd = 4 * torch.cuda.get_device_properties(0).multi_processor_count
    buf = torch.empty(96 << 30, dtype=torch.uint8, device="cuda")
    buf.zero_()
    for path, per_wave in ((3, 64), (2, 16), (1, 4)):
        for size_kib in (16, 256, 1024):
            n = n_simd * per_wave
            size = size_kib << 10
            stride = buf.numel() // n // 4096 * 4096      # spread the files over the whole buffer
            if size > stride:
                continue
            blocks = (size + 8) // 64 + 1
            lengths = np.full(n, size, dtype=np.int64)
            offsets = np.arange(n, dtype=np.int64) * stride
            d_off = torch.from_numpy(offsets).cuda()
            d_len = torch.from_numpy(lengths).cuda()
            d_dig = torch.zeros(n * 20, dtype=torch.uint8, device="cuda")
            os.environ["SNAPPY_SHA1_FORCE_PATH"] = str(path)
            ms = []
            for _ in range(3):
                native.check(lib.snappy_sha1_batch_resident(0, buf.data_ptr(), d_off.data_ptr(), d_len.data_ptr(),
                                                            n, d_dig.data_ptr(), None), "resident")
                ms.append(lib.snappy_profile_last_ms(0))
            print(f"path {path} {n} files x {size_kib} KiB, stride {stride >> 10} KiB: {[round(m, 2) for m in ms]} ms  "
                  f"chain {min(ms) * 1e6 / blocks:7.1f} ns/block  {n * size / min(ms) / 1e6:8.1f} GB/s", flush=True)


if __name__ == "__main__":
    main()
