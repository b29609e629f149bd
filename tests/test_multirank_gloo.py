"""The N > 1 path without GPUs: two gloo ranks run bench.py's control flow (barrier, MAX of
times, SUM of bytes, one JSON line), and a file list is sharded the way the library shards a
batch over devices, each rank hashing its shard (the CPU oracle stands in for the kernels)."""
import json
import os
import subprocess
import sys

import numpy as np

import helpers as H


def test_bench_two_ranks_gloo(tmp_path):
    env = dict(os.environ)
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29731", os.path.join(H.ROOT, "bench.py"),
           "--gpus", "2", "--steps", "2", "--warmup", "0", "--cpu-dry-run", "--workload", "folder"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=240, env=env, cwd=H.ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout          # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["dry_run"] is True
    # both ranks' bytes were summed: each rank has a folder of its own at this scale
    single = subprocess.run([sys.executable, os.path.join(H.ROOT, "bench.py"), "--steps", "1", "--warmup", "0",
                             "--cpu-dry-run", "--workload", "folder"], capture_output=True, text=True, timeout=240,
                            cwd=H.ROOT)
    assert single.returncode == 0, single.stderr[-2000:]
    one = json.loads([ln for ln in single.stdout.splitlines() if ln.startswith("{")][0])
    assert one["n_gpus"] == 1
    assert 1.5 * one["job_bytes"] < d["job_bytes"] < 2.5 * one["job_bytes"]
    assert one["digest0"] == d["digest0"]         # rank 0 hashes the same corpus either way


def test_bench_two_ranks_default_workload(tmp_path):
    """The default workload with two ranks: a corpus of its own on each rank (files are independent,
    so the bytes add up), and beside it one corpus dealt out over the ranks, complete."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29741", os.path.join(H.ROOT, "bench.py"),
           "--gpus", "2", "--steps", "1", "--warmup", "0", "--cpu-dry-run", "--scale", "2"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=240, cwd=H.ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])
    assert d["scaling"] == "weak" and d["n_gpus"] == 2 and d["corpus_bytes"] is None
    assert d["job_bytes"] > 1.3 * d["bytes_rank0"]
    sh = d["one_corpus_sharded"]
    assert sh["job_bytes"] == sh["corpus_bytes"] and sh["job_files"] == sh["corpus_files"] > 0


def test_bench_two_ranks_one_corpus(tmp_path):
    """--strong: ONE corpus for the job whatever the number of ranks, dealt out by file
    (longest first by padded blocks); the ranks' bytes and files add up to it."""
    def run(n):
        cmd = [sys.executable, os.path.join(H.ROOT, "bench.py")]
        if n > 1:
            cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
                   "--master-addr", "127.0.0.1", "--master-port", "29751", os.path.join(H.ROOT, "bench.py")]
        cmd += ["--gpus", str(n), "--steps", "1", "--warmup", "0", "--cpu-dry-run", "--scale", "2", "--strong"]
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=240, cwd=H.ROOT)
        assert out.returncode == 0, out.stderr[-2000:]
        return json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])

    one, two = run(1), run(2)
    assert two["scaling"] == "strong" and one["scaling"] == "strong"
    assert two["job_bytes"] == one["job_bytes"] == one["corpus_bytes"] == two["corpus_bytes"]
    assert two["job_files"] == one["job_files"] == one["files_rank0"]
    assert 0 < two["files_rank0"] < one["files_rank0"]
    # balance by blocks: rank 0 holds half the bytes, give or take its largest file
    assert abs(two["bytes_rank0"] - one["job_bytes"] / 2) <= one["largest_rank0"]


def _worker(rank, world, port, lengths, seed, ret):
    import torch.distributed as dist

    from oracle import oracle
    from snappy_amd.crypto import plan_shards_lpt

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lengths = np.asarray(lengths, dtype=np.uint64)
    offsets = np.concatenate([[0], np.cumsum(lengths)[:-1]]).astype(np.uint64)
    base = np.random.default_rng(seed).integers(0, 256, size=int(lengths.sum()) + 1, dtype=np.uint8)
    shard_of, _ = plan_shards_lpt(lengths, world)  # the library's own dealing, one shard per rank
    idx = np.flatnonzero(shard_of == rank)
    mine = oracle.sha1_batch_np(base, offsets[idx], lengths[idx])
    gathered = [None] * world
    dist.all_gather_object(gathered, (idx.tolist(), mine.tobytes()))   # host-side merge, no reduction
    if rank == 0:
        ret.put(gathered)
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_file_list_two_ranks(oracle, snappy):
    import torch.multiprocessing as mp

    rng = np.random.default_rng(4)
    lengths = np.concatenate([H.mixed_sizes(rng, 60, 1000, 400_000), [0, 1, 3_000_000]])
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, 29741, lengths.tolist(), 99, ret)) for r in range(2)]
    for p in procs:
        p.start()
    gathered = ret.get(timeout=200)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # coverage: every file is in exactly one shard
    all_idx = sorted(i for g in gathered for i in g[0])
    assert all_idx == list(range(len(lengths)))
    offsets = np.concatenate([[0], np.cumsum(lengths)[:-1]]).astype(np.uint64)
    base = np.random.default_rng(99).integers(0, 256, size=int(lengths.sum()) + 1, dtype=np.uint8)
    want = oracle.sha1_batch_np(base, offsets, lengths.astype(np.uint64))
    merged = np.zeros_like(want)
    for idx, raw in gathered:
        merged[idx] = np.frombuffer(raw, dtype=np.uint8).reshape(-1, 20)
    assert np.array_equal(merged, want)
    # balance: neither shard holds more than ~the largest file above half the bytes
    b0 = int(lengths[gathered[0][0]].sum())
    assert abs(b0 - lengths.sum() / 2) <= lengths.max()


def test_plan_shards_properties(snappy):
    from snappy_amd.crypto import plan_shards

    rng = np.random.default_rng(8)
    for n, k in [(0, 3), (1, 4), (5, 5), (1000, 8), (37, 2)]:
        ln = rng.integers(0, 1 << 20, size=n).astype(np.uint64)
        cuts = plan_shards(ln, k)
        assert cuts[0] == 0 and cuts[-1] == n and np.all(np.diff(cuts.astype(np.int64)) >= 0)
        sys.path.insert(0, H.ROOT)
        import bench

        assert np.array_equal(bench.host_shard_cuts(ln, k), cuts)   # bench's dry-run restatement
        if n >= 100:
            per = np.array([ln[int(cuts[i]): int(cuts[i + 1])].sum() for i in range(k)], dtype=np.float64)
            assert per.max() <= ln.sum() / k + ln.max()
