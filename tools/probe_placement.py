"""Do 1024 single-wave workgroups land one per SIMD?  Equal files, 64 lanes per file, one file
per wave: if two waves share a SIMD the batch takes twice as long.  Prints every repetition."""
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
    buf = torch.randint(0, 256, (1 << 30,), dtype=torch.uint8, device="cuda")
    os.environ["SNAPPY_SHA1_FORCE_PATH"] = "0"
    for n in (n_simd // 4, n_simd // 2, n_simd * 3 // 4, n_simd - 64, n_simd):
        size = 512 << 10
        lengths = np.full(n, size, dtype=np.int64)
        offsets = np.arange(n, dtype=np.int64) * size
        d_off = torch.from_numpy(offsets).cuda()
        d_len = torch.from_numpy(lengths).cuda()
        d_dig = torch.zeros(n * 20, dtype=torch.uint8, device="cuda")
        ms = []
        for _ in range(10):
            native.check(lib.snappy_sha1_batch_resident(0, buf.data_ptr(), d_off.data_ptr(), d_len.data_ptr(), n,
                                                        d_dig.data_ptr(), None), "resident")
            ms.append(round(lib.snappy_profile_last_ms(0), 2))
        print(f"{n} waves: {ms}", flush=True)


if __name__ == "__main__":
    main()
