"""The one-lane SHA-1 path (one file per lane, 64 per wave) on files of 4-10 MB: how much does it
matter where the 64 files of a wave lie in memory?  Every file is forced onto the one-lane path; the
same sizes are laid out shuffled (a wave's files, neighbours in size order, scattered over the
buffer) and sorted by size (a wave's files side by side).  Prints the kernel time and the ns per
block of the largest file's chain.  Usage: python tools/probe_one_lane_big.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from snappy_amd import native
lib = native.lib(); native.check(lib.snappy_init(0), "init"); lib.snappy_profile_enable(1)
os.environ["SNAPPY_SHA1_FORCE_PATH"] = "3"
buf = bench.fill_random(torch, 200 << 30, torch.device("cuda", 0), seed=1)
rng = np.random.default_rng(5)
def run(label, lengths, shuffle):
    lengths = lengths.astype(np.int64)
    order = rng.permutation(len(lengths)) if shuffle else np.arange(len(lengths))
    lengths = lengths[order]
    offsets = np.concatenate([[0], np.cumsum(lengths)[:-1]]).astype(np.int64)
    n = len(lengths)
    d_off = torch.from_numpy(offsets).cuda(); d_len = torch.from_numpy(lengths).cuda()
    d_dig = torch.zeros(n * 20, dtype=torch.uint8, device="cuda")
    ms = []
    for _ in range(3):
        native.check(lib.snappy_sha1_batch_resident(0, buf.data_ptr(), d_off.data_ptr(), d_len.data_ptr(), n, d_dig.data_ptr(), None), "r")
        ms.append(lib.snappy_profile_last_ms(0))
    blocks = (lengths.max() + 8) // 64 + 1
    print(f"{label}: n={n} total={lengths.sum()/1e9:.1f} GB  kernel {min(ms):.1f}..{max(ms):.1f} ms  = {min(ms)*1e6/blocks:.0f} ns per block of the largest file", flush=True)
for waves in (64, 256, 320):
    n = 64 * waves
    ln = np.exp(rng.uniform(np.log(4 << 20), np.log(10 << 20), size=n))
    run(f"log-uniform 4-10 MB, {waves} waves, shuffled placement", ln, True)
    run(f"log-uniform 4-10 MB, {waves} waves, sorted placement", np.sort(ln)[::-1], False)
n = 64 * 256
run("equal 8 MB - 9, 256 waves", np.full(n, (8 << 20) - 9), False)
run("equal 8 MB + 4097 (odd stride), 256 waves", np.full(n, (8 << 20) + 4097), False)
