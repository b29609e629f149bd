"""Per-step kernel times of the mixed-corpus workload (run-to-run spread of the planner's cut).
Usage: python tools/probe_corpus_steps.py [scale] [steps]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch

    import bench
    from snappy_amd import native

    scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    lib = native.lib()
    native.check(lib.snappy_init(0), "init")
    lib.snappy_profile_enable(1)
    lengths = bench.make_sizes("corpus", np.random.default_rng(0x534E4150), scale)
    offsets = np.concatenate([[0], np.cumsum(lengths)[:-1]]).astype(np.uint64)
    total = int(lengths.sum())
    buf = bench.fill_random(torch, total, torch.device("cuda", 0), seed=1)
    n = len(lengths)
    d_off = torch.from_numpy(offsets.astype(np.int64)).cuda()
    d_len = torch.from_numpy(lengths.astype(np.int64)).cuda()
    d_dig = torch.zeros(n * 20, dtype=torch.uint8, device="cuda")
    ms = []
    for _ in range(steps):
        native.check(lib.snappy_sha1_batch_resident(0, buf.data_ptr(), d_off.data_ptr(), d_len.data_ptr(), n,
                                                    d_dig.data_ptr(), None), "resident")
        ms.append(round(lib.snappy_profile_last_ms(0), 1))
    print(f"kernel ms per step: {ms}  plan {bench.last_plan(lib)}", flush=True)


if __name__ == "__main__":
    main()
