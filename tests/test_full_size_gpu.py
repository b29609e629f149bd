"""Properties at the benchmark's full sizes, where the CPU oracle would take minutes:
  * 10^6 files of 4-32 KB: a 3000-file sample against hashlib, and order independence (the same
    files handed over in reverse order give the same digests, un-permuted);
  * a 66 MB file's digest does not depend on the execution path or on what else is in the batch;
  * 1 GiB of asset-like data: compress on the GPU, decode on the GPU, byte-identical;
  * a batch import at the import configuration's own shape (64 archives of 64-512 MiB, 50-300 files
    each, long-run and short-sequence content alternating): EVERY file's digest against hashlib."""
import hashlib
import json
import os
import struct
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu


def test_million_small_files_sample_and_order_independence(snappy):
    import torch

    from snappy_amd.crypto import sha1_batch_resident

    n = 1_000_000
    rng = np.random.default_rng(123)
    lengths = rng.integers(4096, 32769, size=n).astype(np.int64)
    offsets = np.concatenate([[0], np.cumsum(lengths)[:-1]]).astype(np.int64)
    total = int(lengths.sum())
    dev = torch.randint(0, 256, (total + 64,), dtype=torch.uint8, device="cuda")
    d_dig = torch.zeros(n * 20, dtype=torch.uint8, device="cuda")
    d_off = torch.from_numpy(offsets).cuda()      # kept alive across the call
    d_len = torch.from_numpy(lengths).cuda()
    sha1_batch_resident(0, dev.data_ptr(), d_off.data_ptr(), d_len.data_ptr(), n, d_dig.data_ptr())
    forward = d_dig.cpu().numpy().reshape(n, 20).copy()
    for i in rng.integers(0, n, size=3000):
        blob = dev[int(offsets[i]): int(offsets[i] + lengths[i])].cpu().numpy().tobytes()
        assert forward[i].tobytes() == hashlib.sha1(blob).digest(), int(i)
    rev_off = torch.from_numpy(offsets[::-1].copy()).cuda()
    rev_len = torch.from_numpy(lengths[::-1].copy()).cuda()
    d_dig.zero_()
    sha1_batch_resident(0, dev.data_ptr(), rev_off.data_ptr(), rev_len.data_ptr(), n, d_dig.data_ptr())
    backward = d_dig.cpu().numpy().reshape(n, 20)
    assert np.array_equal(backward[::-1], forward)


def test_large_file_digest_independent_of_path_and_company(snappy):
    import torch

    size = 66_000_003
    dev = torch.randint(0, 256, (size + 5_000_000,), dtype=torch.uint8, device="cuda")
    want = hashlib.sha1(dev[7: 7 + size].cpu().numpy().tobytes()).digest()
    keep = os.environ.get("SNAPPY_SHA1_FORCE_PATH")
    try:
        for path in ("0", "1", None):
            if path is None:
                os.environ.pop("SNAPPY_SHA1_FORCE_PATH", None)
            else:
                os.environ["SNAPPY_SHA1_FORCE_PATH"] = path
            alone = snappy.sha1_batch_device(0, dev.data_ptr(), np.array([7], dtype=np.uint64),
                                             np.array([size], dtype=np.uint64))
            assert alone[0].tobytes() == want, path
        os.environ.pop("SNAPPY_SHA1_FORCE_PATH", None)
        rng = np.random.default_rng(9)
        lens = np.concatenate([[size], rng.integers(0, 400_000, size=2000)]).astype(np.uint64)
        offs = np.concatenate([[7], rng.integers(0, size, size=2000)]).astype(np.uint64)
        crowd = snappy.sha1_batch_device(0, dev.data_ptr(), offs, lens)
        assert crowd[0].tobytes() == want
    finally:
        if keep is not None:
            os.environ["SNAPPY_SHA1_FORCE_PATH"] = keep


def test_gib_compress_decompress_roundtrip(snappy):
    rng = np.random.default_rng(31)
    piece = np.frombuffer(H.asset_like(64 << 20, 77), dtype=np.uint8)
    data = np.concatenate([np.roll(piece, int(rng.integers(1, 1 << 20))) ^ np.uint8(k) for k in range(16)])
    assert data.size == 1 << 30
    stream = snappy.lz4net_encode(data)
    assert len(stream) < data.size
    back = np.frombuffer(snappy.lz4net_decode(stream), dtype=np.uint8)
    assert back.size == data.size and np.array_equal(back, data)


def test_import_batch_at_the_configurations_shape(snappy):
    import bench

    rng = np.random.default_rng(0x35)
    pools = {"asset": bench.asset_like_bytes(513 << 20, rng), "dense": bench.dense_bytes(513 << 20, rng)}
    streams, jobs, all_lengths = [], [], []
    with ThreadPoolExecutor(8) as pool:   # hashlib lets go of the interpreter lock on large buffers
        for a in range(64):
            size = int(rng.integers(64 << 20, (512 << 20) + 1))
            kind = "asset" if a % 2 == 0 else "dense"
            start = int(rng.integers(0, pools[kind].size - size + 1))
            body = pools[kind][start:start + size]
            stream, _, payload = bench.make_mcdf_archive(size, int(rng.integers(50, 301)), rng, f"full{a}", kind, body=body)
            json_len = struct.unpack_from("<i", payload[5:9].tobytes())[0]
            lengths = [f["Length"] for f in json.loads(payload[9:9 + json_len].tobytes())["Files"]]
            del payload
            assert sum(lengths) == size
            streams.append(stream)
            at = 0
            all_lengths += lengths
            for n in lengths:
                jobs.append(pool.submit(lambda b=body[at:at + n]: hashlib.sha1(b).digest()))
                at += n
        got = snappy.mcdf_decode_and_hash_batch(streams, n_files_cap=len(jobs))
        want = [j.result() for j in jobs]
    at = 0
    for a, r in enumerate(got):
        assert r["status"] == 0, a
        k = len(r["digests"])
        for i in range(k):
            if all_lengths[at + i] == 0:      # the importer skips empty files without hashing them
                assert r["statuses"][i] == 1, (a, i)
            else:
                assert r["statuses"][i] == 0 and r["digests"][i].tobytes() == want[at + i], (a, i)
        at += k
    assert at == len(want)
