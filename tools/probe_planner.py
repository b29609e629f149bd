"""The SHA-1 planner's prediction against the kernel's time, over the bench's corpora and a few
hostile size mixes, for one or several settings of the cost table.

  python tools/probe_planner.py [name=chain0,chain1,chain2,chain3:slowdown[:one-lane ns for the hand-out order] ...]
  (PROBE_ONLY=<part of a corpus label>, PROBE_REPS=<launches per setting>)

Without arguments the library's built-in table is used.  Prints, per corpus and setting: files per
path, planned ms, kernel ms of three launches, and the two floors a batch cannot beat: its
largest file's chain (blocks x 412 instructions x 4 clocks at 2.4 GHz) and its total work on the
one-lane path spread over all SIMDs."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def mixes(bench, rng_seed=0x534E4150):
    rng = np.random.default_rng
    yield "corpus 100 GB", bench.make_sizes("corpus", rng(rng_seed), 1.0)
    yield "corpus 200 GB", bench.make_sizes("corpus", rng(rng_seed), 2.0)
    yield "corpus 25 GB", bench.make_sizes("corpus", rng(rng_seed), 0.25)
    yield "blobs 8 GiB", bench.make_sizes("blobs", rng(rng_seed), 1.0)
    yield "small 10^6", bench.make_sizes("small", rng(rng_seed), 1.0)
    yield "folder 200 MB", bench.make_sizes("folder", rng(rng_seed), 1.0)
    yield "one 64 MB + 10^5 x 8 KB", np.concatenate([[64 << 20], np.full(100_000, 8 << 10)]).astype(np.uint64)
    yield "500 x 8 MB", np.full(500, 8 << 20, dtype=np.uint64)
    yield "3000 x 8 MB", np.full(3000, 8 << 20, dtype=np.uint64)
    yield "20000 x 1 MB", np.full(20000, 1 << 20, dtype=np.uint64)
    yield "2000 x 3 MB + 10^5 x 64 KB", np.concatenate([np.full(2000, 3 << 20), np.full(100_000, 64 << 10)]).astype(np.uint64)
    yield "1100 x 20 MB", np.full(1100, 20 << 20, dtype=np.uint64)


def main():
    import torch

    import bench
    from snappy_amd import native

    settings = [("built-in", None, None, None)]
    for arg in sys.argv[1:]:
        name, rest = arg.split("=", 1)
        parts = rest.split(":")
        settings.append((name, parts[0] or None, parts[1] or None, parts[2] if len(parts) > 2 else None))
    lib = native.lib()
    native.check(lib.snappy_init(0), "init")
    lib.snappy_profile_enable(1)
    n_simd = 4 * torch.cuda.get_device_properties(0).multi_processor_count
    buf = None
    only = os.environ.get("PROBE_ONLY")
    for label, lengths in mixes(bench):
        if only and only not in label:
            continue
        total = int(lengths.sum())
        if buf is None or buf.numel() < total:
            del buf
            torch.cuda.empty_cache()
            buf = bench.fill_random(torch, max(total, 25 << 30), torch.device("cuda", 0), seed=1)
        offsets = np.concatenate([[0], np.cumsum(lengths)[:-1]]).astype(np.uint64)
        n = len(lengths)
        blocks = (lengths.astype(np.int64) + 8) // 64 + 1
        chain_floor = float(blocks.max()) * 412 * 4 / 2.4e9 * 1e3
        work_floor = float(blocks.sum()) * 613 * 4 / 2.4e9 / 64 / n_simd * 1e3
        d_off = torch.from_numpy(offsets.astype(np.int64)).cuda()
        d_len = torch.from_numpy(lengths.astype(np.int64)).cuda()
        d_dig = torch.zeros(n * 20, dtype=torch.uint8, device="cuda")
        print(f"== {label}: {n} files, {total / 1e9:.2f} GB, largest {int(lengths.max()) / 1e6:.1f} MB; floors: chain "
              f"{chain_floor:.1f} ms, work {work_floor:.1f} ms", flush=True)
        for name, chain, slow, order in settings:
            for key, val in (("SNAPPY_SHA1_CHAIN_NS", chain), ("SNAPPY_SHA1_LOAD_SLOWDOWN", slow),
                             ("SNAPPY_SHA1_ONE_LANE_ORDER_NS", order)):
                if val is None:
                    os.environ.pop(key, None)
                else:
                    os.environ[key] = val
            ms = []
            for _ in range(int(os.environ.get("PROBE_REPS", "3"))):
                native.check(lib.snappy_sha1_batch_resident(0, buf.data_ptr(), d_off.data_ptr(), d_len.data_ptr(), n,
                                                            d_dig.data_ptr(), None), "resident")
                ms.append(round(lib.snappy_profile_last_ms(0), 2))
            plan = bench.last_plan(lib)
            files = list(plan["files"].values())
            best = min(ms)
            print(f"   {name:>10}: paths {files}  planned {plan['planned_ms']:.1f} ms  kernel {ms}  planned/kernel "
                  f"{plan['planned_ms'] / best:.2f}  kernel/floor {best / max(chain_floor, work_floor):.2f}", flush=True)


if __name__ == "__main__":
    main()
