"""Measures the cost figures of the SHA-1 execution paths on the GPU: equal-sized files,
every file forced down one path, one or two waves (and their fill of files) per SIMD.
chain_ns = kernel time / blocks per file (how fast one file advances);
simd_ns  = kernel time * SIMDs / total blocks (SIMD time a block costs).
Usage: python tools/probe_sha1_paths.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch

    from snappy_amd import native

    lib = native.lib()
    native.check(lib.snappy_init(0), "snappy_init")
    lib.snappy_profile_enable(1)
    n_simd = 4 * torch.cuda.get_device_properties(0).multi_processor_count
    per_wave = (1, 4, 16, 64)
    file_kib = (2048, 2048, 1024, 512)
    buf = torch.randint(0, 256, (1 << 30,), dtype=torch.uint8, device="cuda")
    print(f"n_simd {n_simd}")
    for path in range(4):
        for waves_per_simd in (1, 2):
            os.environ["SNAPPY_SHA1_WAVES_PER_SIMD"] = str(waves_per_simd)
            n = n_simd * per_wave[path] * waves_per_simd
            size = file_kib[path] << 10
            while n * size > buf.numel():
                size //= 2
            blocks = (size + 8) // 64 + 1
            lengths = np.full(n, size, dtype=np.int64)
            offsets = (np.arange(n, dtype=np.int64) * size)
            d_off = torch.from_numpy(offsets).cuda()
            d_len = torch.from_numpy(lengths).cuda()
            d_dig = torch.zeros(n * 20, dtype=torch.uint8, device="cuda")
            os.environ["SNAPPY_SHA1_FORCE_PATH"] = str(path)
            ms = []
            for _ in range(3):
                native.check(lib.snappy_sha1_batch_resident(0, buf.data_ptr(), d_off.data_ptr(), d_len.data_ptr(),
                                                            n, d_dig.data_ptr(), None), "resident")
                ms.append(lib.snappy_profile_last_ms(0))
            t = min(ms)
            print(f"path {path} waves/simd {waves_per_simd} files {n} x {size >> 10} KiB: {t:8.3f} ms  "
                  f"chain {t * 1e6 / blocks:7.1f} ns/block  simd {t * 1e6 * n_simd / (n * blocks):7.2f} ns/block  "
                  f"{n * size / t / 1e6:8.1f} GB/s", flush=True)


if __name__ == "__main__":
    main()
