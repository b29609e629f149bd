#!/usr/bin/env python3
"""Benchmark of the asset-hashing hot path on MI355X.

One "step" = one pass of the hot path over one batch of synthetic assets that
already sit in HBM: `snappy_sha1_batch_device` over this rank's files (host
tables in, digests back on the host).

The default workload is the 100 GB mixed asset corpus (70 % of the bytes in
.tex-like files of 256 KB - 22 MB, 30 % in .mdl-like files of 64 KB - 6 MB),
ONE corpus for the whole job: with N ranks (launched by torch.distributed.run,
one rank per GPU) the files are dealt out to the ranks longest first by padded
64-byte blocks (the library's snappy_plan_shards_lpt), rank r hashes shard r, and
there is no data-path collective (files are independent).  N = 1 hashes the
whole corpus on one GPU.  The timed region is bracketed by a barrier and a device
synchronize; the slowest rank's time counts.  The same corpus is then hashed end
to end from page-locked host shards (one snappy_sha1_batch call per rank).

With N = 1 the line also carries the other configurations (`other_workloads`):
the 8 GiB corpus of 4-64 MiB blobs, 10^6 small files, one snapshot folder, and
the MCDF import on two kinds of archive content.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

GB = 1e9
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak
SHA1_INSTR_PER_BLOCK = 613     # gfx950: 16 v_perm, 64 x 3 schedule, 80 x 5 rounds, 5 adds
SHA1_CHAIN_INSTR = 412         # of them on one file's serial chain when 64 lanes share the file
PEAK_CLOCK_HZ = 2.4e9
METRIC = "GB/s of mod assets hashed (+LZ4-decoded) per GPU and per node, kernel & e2e"
CORPUS_SEED = 0x534E4150

WORKLOAD_TEXT = {
    "corpus": "mixed asset corpus (70 pct of the bytes .tex-like 256 KB-22 MB, 30 pct .mdl-like 64 KB-6 MB), "
              "{gb:.1f} GB per {per}, SHA-1, inputs resident in HBM",
    "blobs": "8 GiB corpus of 4-64 MiB blobs per {per}, SHA-1, inputs resident in HBM",
    "small": "10^6 files of 4-32 KB per {per}, SHA-1, data + (offset,len) table + digests resident in HBM",
    "folder": "one snapshot folder (160 files, ~200 MB) per {per}, SHA-1, inputs resident in HBM",
}


def log(text: str):
    """Progress on stderr (stdout carries the one JSON line)."""
    if os.environ.get("BENCH_VERBOSE") == "1":
        print(f"[bench rank {os.environ.get('RANK', '0')} +{time.perf_counter() - _T0:7.1f}s] {text}",
              file=sys.stderr, flush=True)


_T0 = time.perf_counter()


def padded_blocks(lengths: np.ndarray) -> np.ndarray:
    return (lengths.astype(np.uint64) + 8) // 64 + 1


def make_sizes(workload: str, rng: np.random.Generator, scale: float):
    """File sizes of one corpus."""
    if workload == "blobs":       # 4-64 MiB texture-sized blobs, 8 GiB in all
        target = int((8 << 30) * scale)
        sizes = []
        total = 0
        while total < target:
            s = int(rng.integers(4 << 20, (64 << 20) + 1))
            sizes.append(s)
            total += s
        return np.array(sizes, dtype=np.uint64)
    if workload == "small":       # 10^6 files of 4-32 KB, not multiples of 64
        n = int(1_000_000 * scale)
        return rng.integers(4096, 32769, size=n).astype(np.uint64)
    if workload == "folder":      # one saved snapshot folder, ~200 MB of .tex/.mdl sizes
        tex = np.exp(rng.uniform(np.log(64 << 10), np.log(22 << 20), size=120))
        mdl = np.exp(rng.uniform(np.log(32 << 10), np.log(6 << 20), size=40))
        s = np.concatenate([tex, mdl])
        s = s * (200e6 * scale / s.sum())
        return np.maximum(s.astype(np.uint64), 1)
    if workload == "corpus":      # mixed asset corpus: 70 % of the bytes .tex-like, 30 % .mdl-like
        target = 100e9 * scale
        tex, mdl = [], []
        t = m = 0.0
        while t < 0.7 * target:
            s = float(np.exp(rng.uniform(np.log(256 << 10), np.log(22 << 20))))
            tex.append(s)
            t += s
        while m < 0.3 * target:
            s = float(np.exp(rng.uniform(np.log(64 << 10), np.log(6 << 20))))
            mdl.append(s)
            m += s
        sizes = np.array(tex + mdl)
        rng.shuffle(sizes)
        return sizes.astype(np.uint64)
    raise SystemExit(f"unknown workload {workload}")


def is_strong(args, world):
    """One corpus for the whole job whatever the number of ranks (sharded by file over them: --strong),
    or one per rank (the default: files are independent, so a GPU's work does not depend on how many
    GPUs there are; with N > 1 the line also carries the one-corpus-over-N-ranks measurement)."""
    if args.weak:
        return False
    return bool(args.strong)


def shard_lpt(lib, lengths, k):
    """The library's dealing of a file list to k devices: (shard_of, blocks per shard)."""
    ln = np.ascontiguousarray(lengths, dtype=np.uint64)
    shard_of = np.zeros(len(ln), dtype=np.uint32)
    blocks = np.zeros(k, dtype=np.uint64)
    if lib.snappy_plan_shards_lpt(ln.ctypes.data, len(ln), k, shard_of.ctypes.data, blocks.ctypes.data) != 0:
        raise SystemExit("snappy_plan_shards_lpt failed")
    return shard_of, blocks


def host_shard_cuts(lengths, k):
    """snappy_plan_shards (the contiguous split) restated: cut j is the first file index at which
    the running byte count reaches j/k of the total."""
    total = int(lengths.sum())
    cuts = [0]
    acc, j = 0, 1
    for i, n in enumerate(lengths):
        acc += int(n)
        while j < k and acc * k >= total * j:
            cuts.append(i + 1)
            j += 1
    while len(cuts) < k + 1:
        cuts.append(len(lengths))
    return np.array(cuts, dtype=np.uint64)


def rank_corpus(args, lib, rank, world):
    """(file sizes this rank hashes, bytes of the whole job's corpus or None when every rank has its own)."""
    if not is_strong(args, world):
        seed = CORPUS_SEED + rank
        return make_sizes(args.workload, np.random.default_rng(seed), args.scale), None
    lengths = make_sizes(args.workload, np.random.default_rng(CORPUS_SEED), args.scale)
    shard_of, _ = shard_lpt(lib, lengths, world)
    return lengths[shard_of == rank], int(lengths.sum())


def fill_random(torch, n_bytes: int, device, seed: int):
    """Random bytes resident in HBM (SHA-1 timing does not depend on content)."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    buf = torch.empty(n_bytes + 256, dtype=torch.uint8, device=device)
    step = 1 << 30
    for lo in range(0, n_bytes + 256, step):
        hi = min(lo + step, n_bytes + 256)
        # int32 words: 4 random bytes each
        words = torch.randint(-(1 << 31), (1 << 31) - 1, ((hi - lo + 3) // 4,), dtype=torch.int32,
                              device=device, generator=g)
        buf[lo:hi] = words.view(torch.uint8)[: hi - lo]
        del words
    return buf


def asset_like_bytes(n: int, rng: np.random.Generator) -> np.ndarray:
    """Half random, half compressible: repeated motifs of 16-4096 bytes and zero runs."""
    parts = []
    total = 0
    while total < n:
        k = rng.random()
        if k < 0.4:
            seg = rng.integers(0, 256, size=int(rng.integers(4096, 400_000)), dtype=np.uint8)
        elif k < 0.6:
            seg = np.zeros(int(rng.integers(1024, 200_000)), dtype=np.uint8)
        else:
            motif = rng.integers(0, 256, size=int(rng.integers(16, 4097)), dtype=np.uint8)
            seg = np.tile(motif, int(rng.integers(2, 80)))
        parts.append(seg)
        total += seg.size
    return np.concatenate(parts)[:n]


_DENSE_UNITS = {}


def dense_bytes(n: int, rng: np.random.Generator, max_lit: int = 24, max_match: int = 36,
                max_off: int = 65535) -> np.ndarray:
    """What a compressor makes of textures and meshes: short literal runs between short copies of
    earlier bytes, i.e. LZ4 streams of short sequences.  16 different 1 MiB chunks, repeated (the
    1 MiB chunks of an lz4net stream are compressed independently)."""
    key = (max_lit, max_match, max_off)
    if key in _DENSE_UNITS:   # generated once per process (a Python loop per sequence)
        unit = _DENSE_UNITS[key]
        return np.tile(unit, -(-n // unit.size))[:n]
    chunks = []
    for _ in range(16):
        buf = bytearray(rng.integers(0, 256, size=64, dtype=np.uint8).tobytes())
        while len(buf) < (1 << 20):
            buf += rng.integers(0, 256, size=int(rng.integers(0, max_lit + 1)), dtype=np.uint8).tobytes()
            ml = int(rng.integers(4, min(max_match, len(buf)) + 1))
            off = int(rng.integers(ml, min(len(buf), max_off) + 1))
            buf += buf[len(buf) - off: len(buf) - off + ml]
        chunks.append(np.frombuffer(bytes(buf[: 1 << 20]), dtype=np.uint8))
    unit = _DENSE_UNITS[key] = np.concatenate(chunks)
    return np.tile(unit, -(-n // unit.size))[:n]


def make_mcdf_archive(payload_bytes: int, n_files: int, rng: np.random.Generator, tag: str,
                      content: str = "asset", body: np.ndarray = None):
    """One synthetic MCDF archive: header JSON + file bodies, lz4net-framed in 1 MiB chunks
    by the library's own compressor (snappy_lz4net_encode).  Returns (stream bytes as uint8
    array, decoded payload size, payload).  `body`: the file bodies, made by the caller."""
    import json as _json
    import struct

    from snappy_amd import native

    cuts = np.sort(rng.integers(0, payload_bytes, size=n_files - 1))
    sizes = np.diff(np.concatenate([[0], cuts, [payload_bytes]])).astype(np.int64)
    obj = {"Description": tag, "GlamourerData": "", "CustomizePlusData": "", "ManipulationData": "",
           "Files": [{"GamePaths": [f"chara/{tag}/f{i}.tex"], "Length": int(sz)} for i, sz in enumerate(sizes)]}
    js = _json.dumps(obj, separators=(",", ":")).encode()
    head = b"MCDF\x01" + struct.pack("<i", len(js)) + js
    if body is None:
        body = dense_bytes(payload_bytes, rng) if content == "dense" else asset_like_bytes(payload_bytes, rng)
    payload = np.concatenate([np.frombuffer(head, dtype=np.uint8), body])
    L = native.lib()
    cap = ctypes.c_uint64(0)
    native.check(L.snappy_lz4net_encode_bound(payload.size, 1 << 20, ctypes.byref(cap)), "encode bound")
    out = np.empty(cap.value, dtype=np.uint8)
    w = ctypes.c_uint64(0)
    t0 = time.perf_counter()
    native.check(L.snappy_lz4net_encode(payload.ctypes.data, payload.size, 1 << 20, out.ctypes.data, out.size,
                                        ctypes.byref(w)), "snappy_lz4net_encode")
    _ARCHIVES_WRITTEN["call_s"] += time.perf_counter() - t0
    _ARCHIVES_WRITTEN["bytes"] += int(payload.size)
    _ARCHIVES_WRITTEN["kernel_ms"] += max(0.0, float(L.snappy_profile_last_ms(1)))
    return out[: w.value].copy(), int(payload.size), payload


# The write side, measured where the bench makes its input archives (not part of any timed step).
_ARCHIVES_WRITTEN = {"bytes": 0, "call_s": 0.0, "kernel_ms": 0.0}


def archives_written_summary():
    a = _ARCHIVES_WRITTEN
    if not a["bytes"] or a["kernel_ms"] <= 0.0:
        return None
    return {"what": "snappy_lz4net_encode over the archives this workload imports (payload in host memory -> "
                    "lz4net stream in host memory, 1 MiB chunks), one call per archive, outside the timed steps",
            "decoded_bytes": a["bytes"], "kernel_ms": a["kernel_ms"],
            "kernel_GBps_in": a["bytes"] / a["kernel_ms"] / 1e6,
            "call_GBps_in": a["bytes"] / a["call_s"] / 1e9}


def run_mcdf(args, rank, world, local_rank, dist):
    """MCDF import workload: A archives per GPU through snappy_mcdf_decode_and_hash_batch
    (host streams in: the PCIe copy of the compressed bytes is inside the step)."""
    import hashlib

    import torch

    from snappy_amd import native

    lib = native.lib()
    native.check(lib.snappy_init(0), "snappy_init")
    lib.snappy_profile_enable(1)
    rng = np.random.default_rng(0x4D434446 + rank)
    n_arch = max(1, int(args.archives))
    _ARCHIVES_WRITTEN.update(bytes=0, call_s=0.0, kernel_ms=0.0)
    content = getattr(args, "mcdf_content", "asset")
    spec = getattr(args, "mcdf_shape", "uniform") == "spec"
    streams, sizes, first_payload = [], [], None
    files_per_archive = []
    if spec:
        # The configuration as written: archives of 64-512 MiB (uniform), 50-300 files each, half of
        # them long-run content and half dense short sequences.  The bodies are windows into two
        # pools of 513 MiB (generating 18 GB of fresh content would take longer than the run).
        content = "asset and dense, alternating"
        pools = {"asset": asset_like_bytes((513 << 20), rng), "dense": dense_bytes((513 << 20), rng)}
        for a in range(n_arch):
            size = int(rng.integers(64 << 20, (512 << 20) + 1))
            kind = "asset" if a % 2 == 0 else "dense"
            start = int(rng.integers(0, pools[kind].size - size + 1))
            n_files = int(rng.integers(50, 301))
            st, total, payload = make_mcdf_archive(size, n_files, rng, f"r{rank}a{a}", kind,
                                                   body=pools[kind][start:start + size])
            streams.append(st)
            sizes.append(total)
            files_per_archive.append(n_files)
            if a == 0:
                first_payload = payload
            del payload
        del pools
    else:
        for a in range(n_arch):
            st, size, payload = make_mcdf_archive(int(args.archive_mib * (1 << 20)), 100, rng, f"r{rank}a{a}", content)
            streams.append(st)
            sizes.append(size)
            files_per_archive.append(100)
            if a == 0:
                first_payload = payload
    written = archives_written_summary()
    comp_bytes = int(sum(s.size for s in streams))
    out_bytes = int(sum(sizes))
    ptrs = (ctypes.c_void_p * n_arch)(*[s.ctypes.data for s in streams])
    lens = np.array([s.size for s in streams], dtype=np.uint64)
    cap = int(sum(files_per_archive))
    digests = np.zeros((cap, 20), dtype=np.uint8)
    statuses = np.zeros(cap, dtype=np.uint8)
    n_files = np.zeros(n_arch, dtype=np.uint64)
    ast = np.zeros(n_arch, dtype=np.int32)

    def step():
        native.check(lib.snappy_mcdf_decode_and_hash_batch(n_arch, ptrs, lens.ctypes.data,
                                                           digests.ctypes.data, statuses.ctypes.data,
                                                           cap, n_files.ctypes.data, None, None,
                                                           ast.ctypes.data), "mcdf batch")

    if getattr(args, "mcdf_decode_only", False):   # counter passes: nothing but the decode launches
        k_lz4, decode_ok = decode_in_one_launch(lib, native, streams, sizes, first_payload)
        if rank != 0:
            return None
        return {"metric": METRIC, "decode_only": True, "kernel": "lz4_decode_pair_kernel", "kernel_ms": k_lz4,
                "algorithmic_bytes_per_launch": comp_bytes + out_bytes, "decoded_bytes_match": decode_ok,
                "content": content}
    for _ in range(args.warmup):
        step()
    lz4_ms, sha_ms = [], []
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        lz4_ms.append(lib.snappy_profile_last_ms(1))
        sha_ms.append(lib.snappy_profile_last_ms(0))
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    first_group_plan = last_plan(lib)   # of the archive that arrives first: the one with the largest file
    job_bytes = float(out_bytes)
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        tot = torch.tensor([job_bytes], dtype=torch.float64)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        job_bytes = float(tot.item())
    if rank != 0:
        return None
    assert int(ast.max()) == 0 and int(ast.min()) == 0
    # Spot check against hashlib: first archive's first files.
    import json as _json
    import struct
    jl = struct.unpack_from("<i", first_payload.tobytes()[:9], 5)[0]
    files = _json.loads(first_payload[9:9 + jl].tobytes())["Files"]
    cur = 9 + jl
    ok = True
    for i, f in enumerate(files[:5]):
        want = hashlib.sha1(first_payload[cur:cur + f["Length"]].tobytes()).digest()
        ok = ok and (f["Length"] == 0 or digests[i].tobytes() == want)
        cur += f["Length"]
    cpu = mcdf_cpu_baseline(streams, sizes) if not getattr(args, "no_cpu_baseline", False) else None
    k_lz4, decode_ok = decode_in_one_launch(lib, native, streams, sizes, first_payload)
    algo = comp_bytes + out_bytes
    achieved = algo / (k_lz4 * 1e-3) / GB
    return {
        "metric": METRIC,
        "value": job_bytes * args.steps / elapsed / GB,
        "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": (f"{n_arch} MCDF archives of 64-512 MiB (uniform), 50-300 files each, per GPU"
                                if spec else f"{n_arch} MCDF archives of {args.archive_mib} MiB per GPU") +
                               ": lz4net decode + slice + SHA-1 in one snappy_mcdf_decode_and_hash_batch call, "
                               "compressed streams start in host memory (PCIe copy inside the step)",
                   "files": cap,
                   "content": content,
                   "compressed_bytes_per_gpu": comp_bytes, "decoded_bytes_per_gpu": out_bytes,
                   "ratio": out_bytes / comp_bytes},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS,
                     "traffic": measured_traffic("mcdf" if content == "asset" else "mcdf_dense", 1.0)
                     if (n_arch, args.archive_mib) == (8, 256.0) and not spec else None,
                     "kernel": "lz4_decode_pair_kernel",
                     "kernel_ms": k_lz4, "algorithmic_bytes_per_launch": algo,
                     "launch": "all chunks of all archives in one launch (snappy_lz4net_decode of the archives' "
                               "streams back to back): the kernel with the GPU to itself",
                     "decoded_bytes_match": decode_ok,
                     # Inside the import the decode is split per archive and runs next to the waiting
                     # SHA-1 kernel, behind each archive's transfer; these are those launches' times added up.
                     "import_decode_launches_ms": float(np.mean(lz4_ms)),
                     "import_decode_launches_over_one_launch": float(np.mean(lz4_ms)) / k_lz4,
                     "sha1_kernel_ms": float(np.mean(sha_ms)),
                     # The hashing is planned group by group in arrival order; this is the first group's plan
                     # (the archive holding the batch's largest file): its planned_ms is that file's chain, the
                     # floor under the whole import next to one archive's transfer and decode.
                     "sha1_plan_first_group": first_group_plan,
                     "import_ms_over_first_group_planned_ms":
                         (elapsed / args.steps * 1e3 / first_group_plan["planned_ms"])
                         if first_group_plan and first_group_plan["planned_ms"] > 0 else None,
                     # What a plain device-to-device copy of the decoded size moves on this box (read + write):
                     # the practical ceiling next to the 8 TB/s of the data sheet.
                     "device_copy_GBps_measured": device_copy_rate(torch, min(out_bytes, 2 << 30))},
        "digests_match_hashlib_sample": bool(ok),
        **({"archives_written": written} if written else {}),
        **({"cpu_baseline": cpu} if cpu else {}),
    }


def device_copy_rate(torch, n):
    """GB/s (read + write) of a device-to-device copy of n bytes: what the memory system gives a kernel
    that does nothing but stream."""
    try:
        x = torch.empty(n, dtype=torch.uint8, device="cuda")
        y = torch.empty_like(x)
        for _ in range(2):
            y.copy_(x)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            y.copy_(x)
        e1.record()
        torch.cuda.synchronize()
        rate = 2.0 * n * 5 / (e0.elapsed_time(e1) * 1e-3) / GB
        del x, y
        return rate
    except Exception:   # out of memory next to the workload: the figure is an extra, not the result
        return None


def decode_in_one_launch(lib, native, streams, sizes, first_payload):
    """The decode kernel on its own: every chunk of every archive in one launch.  lz4net streams are
    bare runs of chunks, so the archives' streams back to back are one stream.  Returns the kernel
    time (HIP events around the launch, mean of three) and whether the first archive came out right."""
    whole = np.concatenate(streams)
    out = np.empty(int(sum(sizes)) + 64, dtype=np.uint8)
    got = ctypes.c_uint64(0)
    ms = []
    for _ in range(4):
        native.check(lib.snappy_lz4net_decode(whole.ctypes.data, whole.size, out.ctypes.data, out.size,
                                              ctypes.byref(got)), "snappy_lz4net_decode")
        ms.append(lib.snappy_profile_last_ms(1))
    ok = got.value == int(sum(sizes)) and bool(np.array_equal(out[: first_payload.size], first_payload))
    return float(np.mean(ms[1:])), ok


def mcdf_cpu_baseline(streams, sizes):
    """The import's CPU path on a bounded sample: lz4net decode (the oracle's scalar C decoder) and
    SHA-1 of the decoded bytes (OpenSSL via hashlib, as one message: the hashing rate does not
    depend on the slicing), one thread on the first archive, then one archive per thread."""
    import hashlib
    from concurrent.futures import ThreadPoolExecutor

    from oracle import oracle

    L = oracle.lib()

    def one(a):
        out = np.empty(sizes[a], dtype=np.uint8)
        v = ctypes.c_uint64(0)
        t0 = time.perf_counter()
        st = L.oracle_lz4net_decode(streams[a].ctypes.data, streams[a].size, out.ctypes.data, out.size,
                                    ctypes.byref(v))
        t1 = time.perf_counter()
        hashlib.sha1(out[: v.value]).digest()
        t2 = time.perf_counter()
        return st, v.value, t1 - t0, t2 - t1

    st, n, t_dec, t_sha = one(0)
    res = {"value": n / (t_dec + t_sha) / GB, "unit": "GB/s", "cores": 1, "kind": "port",
           "sample": f"first archive ({n} decoded bytes): oracle/lz4_oracle.c decode + OpenSSL SHA-1, one thread",
           "decode_only_value": n / t_dec / GB, "status": int(st)}
    k = min(len(streams), max(1, min(16, os.cpu_count() or 1)))
    t0 = time.perf_counter()
    with ThreadPoolExecutor(k) as pool:
        done = list(pool.map(one, range(k)))
    dt = time.perf_counter() - t0
    res["allcore_value"] = sum(d[1] for d in done) / dt / GB
    res["allcore_threads"] = k
    res.update(cpu_info())
    return res


def sha1_roofline(algo_bytes, kernel_ms, max_blocks, n_cu, plan, traffic):
    """SHA-1 is bound by the integer ALUs, not by HBM: 613 vector instructions per 64-byte block,
    64 int32 lanes per CU and clock.  The HBM view and the serial-chain floor are given beside it."""
    achieved = algo_bytes / (kernel_ms * 1e-3) / GB
    alu_peak = n_cu * 64 * PEAK_CLOCK_HZ / SHA1_INSTR_PER_BLOCK * 64 / GB
    # One wavefront issues one instruction per 4 clocks; a block of the largest file is 412 of them.
    chain_floor_ms = max_blocks * SHA1_CHAIN_INSTR * 4 / PEAK_CLOCK_HZ * 1e3
    return {
        "bound": "int-alu", "achieved": achieved, "peak": alu_peak, "unit": "GB/s",
        "frac": achieved / alu_peak, "traffic": traffic,
        "kernel": "sha1_hop_kernel", "kernel_ms": kernel_ms, "algorithmic_bytes_per_launch": algo_bytes,
        "peak_is": f"{n_cu} CUs x 64 lanes x 2.4 GHz / {SHA1_INSTR_PER_BLOCK} instructions per block x 64 B",
        "measured_alu_ceiling_GBps": 4200.0,   # register-only compression loop at full clock, tools/ubench_sha1.hip
        "hbm_peak_GBps": HBM_PEAK_GBS, "hbm_frac": achieved / HBM_PEAK_GBS,
        # SHA-1 is serial inside a file: a batch cannot finish before its largest file's chain.
        "serial_chain_blocks": max_blocks, "chain_floor_ms": chain_floor_ms,
        "chain_floor_frac": chain_floor_ms / kernel_ms,
        "chain_ns_per_block": kernel_ms * 1e6 / max_blocks,
        "plan": plan,
    }


def hash_resident(lib, native, torch, device, local_rank, buf, offsets, lengths, steps, warmup, tables_resident,
                  barrier=None):
    """`steps` timed passes of the hot path over data in HBM.  Returns (elapsed s, kernel ms list, digests)."""
    n = len(lengths)
    digests = np.zeros((n, 20), dtype=np.uint8)
    stream = torch.cuda.current_stream().cuda_stream
    if tables_resident:    # the (offset, length) table and the digests stay in HBM too
        d_off = torch.from_numpy(offsets.astype(np.int64)).to(device)
        d_len = torch.from_numpy(lengths.astype(np.int64)).to(device)
        d_dig = torch.zeros(n * 20, dtype=torch.uint8, device=device)

    def step():
        if tables_resident:
            native.check(lib.snappy_sha1_batch_resident(local_rank, buf.data_ptr(), d_off.data_ptr(),
                                                        d_len.data_ptr(), n, d_dig.data_ptr(), stream),
                         "snappy_sha1_batch_resident")
        else:
            native.check(lib.snappy_sha1_batch_device(local_rank, buf.data_ptr(), offsets.ctypes.data,
                                                      lengths.ctypes.data, n, digests.ctypes.data,
                                                      stream), "snappy_sha1_batch_device")

    for _ in range(warmup):
        step()
    kernel_ms = []
    torch.cuda.synchronize()
    if barrier is not None:
        barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
        kernel_ms.append(lib.snappy_profile_last_ms(0))
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if tables_resident:
        digests[:] = d_dig.cpu().numpy().reshape(n, 20)
    return elapsed, kernel_ms, digests


def extra_sha1_workload(name, lib, native, torch, device, local_rank, n_cu, want_e2e):
    """One of the other SHA-1 configurations, measured briefly on one GPU."""
    import hashlib

    lengths = make_sizes(name, np.random.default_rng(CORPUS_SEED), 1.0)   # the corpus `--workload name` hashes
    offsets = np.concatenate([[0], np.cumsum(lengths)[:-1]]).astype(np.uint64)
    total = int(lengths.sum())
    algo = int(padded_blocks(lengths).sum()) * 64
    buf = fill_random(torch, total, device, seed=99)
    n = len(lengths)
    elapsed, k_ms, digests = hash_resident(lib, native, torch, device, local_rank, buf, offsets, lengths, 2, 1,
                                           name == "small")
    ok = all(hashlib.sha1(buf[int(offsets[i]): int(offsets[i] + lengths[i])].cpu().numpy().tobytes()).digest()
             == digests[i].tobytes() for i in sorted({0, 1, n // 2, n - 1}))
    km = float(np.mean(k_ms))
    res = {"workload": WORKLOAD_TEXT[name].format(per="GPU", gb=total / GB),
           "files": n, "bytes": total, "largest_file_bytes": int(lengths.max()),
           "value": total * 2 / elapsed / GB, "unit": "GB/s", "ms_per_step": elapsed / 2 * 1e3,
           "roofline": sha1_roofline(algo, km, int(padded_blocks(lengths).max()), n_cu, last_plan(lib),
                                     measured_traffic(name, 1.0)),
           "digests_match_hashlib_sample": bool(ok)}
    if want_e2e:
        e2e = end_to_end(lib, native, torch, {"buf": buf}, offsets, lengths, total, digests, None, pageable=True)
        res["end_to_end"] = {"value": total / e2e["seconds"] / GB, "unit": "GB/s",
                             "ms": e2e["seconds"] * 1e3, "host_memory_page_locked": e2e["page_locked"],
                             "digests_match_resident_run": e2e["digests_match_resident_run"]}
        if "pageable_seconds" in e2e:
            res["end_to_end"]["pageable_value"] = total / e2e["pageable_seconds"] / GB
    del buf
    torch.cuda.empty_cache()
    return res


def concurrent_callers(lib, native, torch, device, local_rank):
    """Two callers at once on one GPU, as the reference can have them: the migration task hashing
    on a pool thread (Snappy/Plugin.cs:257-278, Data/SnapshotMigrator.cs:62-86) while the UI thread
    saves a snapshot (Core/SnapshotManager.cs:312-324).  Each caller is a thread with a batch of
    its own in HBM; the wall time of both against the time of each alone."""
    import threading

    def corpus(name, seed):
        lengths = make_sizes(name, np.random.default_rng(CORPUS_SEED), 1.0)
        offsets = np.concatenate([[0], np.cumsum(lengths)[:-1]]).astype(np.uint64)
        buf = fill_random(torch, int(lengths.sum()), device, seed=seed)
        return {"name": name, "lengths": lengths, "offsets": offsets, "buf": buf, "bytes": int(lengths.sum()),
                "digests": np.zeros((len(lengths), 20), dtype=np.uint8)}

    def call(c):
        native.check(lib.snappy_sha1_batch_device(local_rank, c["buf"].data_ptr(), c["offsets"].ctypes.data,
                                                  c["lengths"].ctypes.data, len(c["lengths"]),
                                                  c["digests"].ctypes.data, None), "snappy_sha1_batch_device")

    def alone(c, reps=3):
        call(c)
        t = []
        for _ in range(reps):
            t0 = time.perf_counter()
            call(c)
            t.append(time.perf_counter() - t0)
        return min(t)

    def together(cs, reps=3):
        best = None
        for _ in range(reps):
            gate = threading.Barrier(len(cs) + 1)
            lat = [0.0] * len(cs)

            def work(i):
                gate.wait()
                t0 = time.perf_counter()
                call(cs[i])
                lat[i] = time.perf_counter() - t0

            threads = [threading.Thread(target=work, args=(i,)) for i in range(len(cs))]
            for th in threads:
                th.start()
            gate.wait()
            t0 = time.perf_counter()
            for th in threads:
                th.join()
            wall = time.perf_counter() - t0
            if best is None or wall < best[0]:
                best = (wall, list(lat))
        return best

    res = {"what": "two threads, one snappy_sha1_batch_device call each at the same moment on one GPU, batches resident "
                   "in HBM; wall time of the pair against each call alone (best of 3)"}
    f1, f2 = corpus("folder", 501), corpus("folder", 502)
    t1 = alone(f1)
    wall, lat = together([f1, f2])
    keep = f1["digests"].copy()
    call(f1)
    res["folder_and_folder"] = {
        "alone_ms": t1 * 1e3, "alone_GBps": f1["bytes"] / t1 / GB, "pair_wall_ms": wall * 1e3,
        "pair_latency_ms": [x * 1e3 for x in lat], "aggregate_GBps": (f1["bytes"] + f2["bytes"]) / wall / GB,
        "aggregate_over_alone": (f1["bytes"] + f2["bytes"]) / wall / (f1["bytes"] / t1),
        "digests_same_as_alone": bool(np.array_equal(keep, f1["digests"]))}
    del f2
    sm = corpus("small", 503)
    ts = alone(sm)
    wall, lat = together([f1, sm])
    res["folder_and_small_files"] = {
        "folder_alone_ms": t1 * 1e3, "small_files_alone_ms": ts * 1e3, "pair_wall_ms": wall * 1e3,
        "pair_latency_ms": {"folder": lat[0] * 1e3, "small_files": lat[1] * 1e3},
        "aggregate_GBps": (f1["bytes"] + sm["bytes"]) / wall / GB,
        "one_after_the_other_ms": (t1 + ts) * 1e3}
    del f1, sm
    torch.cuda.empty_cache()
    return res


def extra_workloads(args, lib, native, torch, device, local_rank, n_cu):
    """The other single-GPU configurations, measured briefly after the headline run so that one
    bench line carries every configuration and every kernel's roofline."""
    res = {}
    res["blobs"] = extra_sha1_workload("blobs", lib, native, torch, device, local_rank, n_cu, True)
    res["small_files"] = extra_sha1_workload("small", lib, native, torch, device, local_rank, n_cu, True)
    res["folder"] = extra_sha1_workload("folder", lib, native, torch, device, local_rank, n_cu, True)
    res["concurrent_callers"] = concurrent_callers(lib, native, torch, device, local_rank)

    # MCDF import: lz4net decode + slice + hash, 8 archives of 256 MiB in one call, on both kinds of
    # content: long runs (random, zeros, repeated motifs) and dense short sequences (what an LZ4
    # compressor makes of mesh and texture data).
    class A:
        pass
    # "mcdf_import_spec" is the import configuration at its full shape: 64 archives of 64-512 MiB,
    # 50-300 files each, both kinds of content, one call; the two 8 x 256 MiB rows are kept beside
    # it (one kind of content each: the decode kernel's roofline per kind).  Each row carries the
    # import's CPU baseline: the oracle's scalar decoder + OpenSSL SHA-1, one thread and one
    # archive per thread.
    for key, content, shape, n_arch in (("mcdf_import_spec", "asset", "spec", 64), ("mcdf_import", "asset", "uniform", 8),
                                       ("mcdf_import_dense", "dense", "uniform", 8)):
        a = A()
        a.archives, a.archive_mib, a.steps, a.warmup, a.mcdf_content, a.mcdf_shape = n_arch, 256.0, 2, 1, content, shape
        a.no_cpu_baseline = getattr(args, "no_cpu_baseline", False)
        m = run_mcdf(a, 0, 1, local_rank, None)
        res[key] = {"workload": m["config"]["workload"], "content": m["config"]["content"], "value": m["value"],
                    "unit": "GB/s", "files": m["config"]["files"],
                    "ms_per_step": m["ms_per_step"], "roofline": m["roofline"],
                    "compressed_bytes": m["config"]["compressed_bytes_per_gpu"],
                    "decoded_bytes": m["config"]["decoded_bytes_per_gpu"],
                    "digests_match_hashlib_sample": m["digests_match_hashlib_sample"]}
        if m.get("archives_written"):
            res[key]["archives_written"] = m["archives_written"]
        if "cpu_baseline" in m:
            res[key]["cpu_baseline"] = m["cpu_baseline"]
    return res


def cpu_baseline(host_sample: np.ndarray, offsets, lengths):
    """The reference loop (read -> SHA-1 -> uppercase hex, sequential, one thread:
    Snappy/Core/SnapshotManager.cs:312-324) on a bounded sample of the same corpus.  `value` is
    OpenSSL's SHA1 on one core -- what .NET's SHA1.ComputeHash calls on Linux, and the stand-in
    for the reference's .NET path, which cannot be built here; the oracle's scalar C port and
    one-file-per-thread OpenSSL are given beside it."""
    from oracle import oracle

    L = oracle.lib()
    n = len(offsets)
    off = np.ascontiguousarray(offsets, dtype=np.uint64)
    ln = np.ascontiguousarray(lengths, dtype=np.uint64)
    out = np.zeros((n, 20), dtype=np.uint8)
    t0 = time.perf_counter()
    L.oracle_sha1_batch(host_sample.ctypes.data, off.ctypes.data, ln.ctypes.data, n, out.ctypes.data)
    port_value = float(ln.sum()) / (time.perf_counter() - t0) / GB
    sample = f"{n} files, {int(ln.sum())} bytes of the same corpus, sequential loop"
    res = {"value": port_value, "unit": "GB/s", "cores": 1, "kind": "port",
           "sample": sample + ", scalar C restatement of Crypto.GetHash (oracle/sha1_oracle.c)"}
    try:
        crypto = ctypes.CDLL("libcrypto.so.3")
        crypto.SHA1.restype = ctypes.c_void_p
        crypto.SHA1.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
        md = (ctypes.c_uint8 * 20)()
        base = host_sample.ctypes.data
        t0 = time.perf_counter()
        for i in range(n):
            crypto.SHA1(base + int(off[i]), int(ln[i]), md)
        dt2 = time.perf_counter() - t0
        res = {"value": float(ln.sum()) / dt2 / GB, "unit": "GB/s", "cores": 1, "kind": "openssl-stand-in",
               "sample": sample + ", OpenSSL libcrypto SHA1() (what .NET's SHA1 calls on Linux; uses SHA-NI "
                                  "when the CPU has it)",
               "port_1core_value": port_value,
               "openssl_agrees_with_port": bool(bytes(md) == out[n - 1].tobytes())}
        # All cores this job may use, one file per thread at a time (ctypes drops the GIL).
        from concurrent.futures import ThreadPoolExecutor

        cores = max(1, min(16, os.cpu_count() or 1))

        def one(i):
            buf20 = (ctypes.c_uint8 * 20)()
            crypto.SHA1(base + int(off[i]), int(ln[i]), buf20)

        t0 = time.perf_counter()
        with ThreadPoolExecutor(cores) as pool:
            list(pool.map(one, range(n)))
        dt3 = time.perf_counter() - t0
        res["openssl_allcore_value"] = float(ln.sum()) / dt3 / GB
        res["openssl_allcore_threads"] = cores
    except OSError:
        pass
    res.update(cpu_info())
    return res, out


def cpu_info():
    model, sha_ni = "unknown", None
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name") and model == "unknown":
                    model = line.split(":", 1)[1].strip()
                if line.startswith("flags") and sha_ni is None:
                    sha_ni = " sha_ni" in line
    except OSError:
        pass
    return {"cpu_model": model, "cpu_has_sha_ni": sha_ni}


def h2d_bandwidth(torch, device):
    """Measured pinned host -> HBM copy rate (GB/s), the ceiling of any end-to-end number."""
    n = 1 << 30
    src = torch.empty(n, dtype=torch.uint8, pin_memory=True)
    dst = torch.empty(n, dtype=torch.uint8, device=device)
    dst.copy_(src, non_blocking=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        dst.copy_(src, non_blocking=True)
    torch.cuda.synchronize()
    return 3 * n / (time.perf_counter() - t0) / GB


def end_to_end(lib, native, torch, holder, offsets, lengths, total_bytes, digests, dist, pageable=False,
               free_resident=False):
    """The same files starting in host memory: one snappy_sha1_batch call moves them over PCIe,
    hashes them and returns the digests.  The shard lies whole in page-locked host memory
    (snappy_host_alloc, no ring).  holder["buf"] is the resident copy; with free_resident it is
    dropped once the host has its own (the library's device arena needs the room)."""
    n = len(lengths)
    buf = holder["buf"]
    # A process gets only so much page-locked memory (about 64 GiB on the boxes this ran on): a
    # shard beyond that lies in ordinary memory and travels through the library's page-locked ring.
    ptr = ctypes.c_void_p()
    locked = lib.snappy_host_alloc(total_bytes + 64, ctypes.byref(ptr)) == 0
    if locked:
        raw = (ctypes.c_uint8 * (total_bytes + 64)).from_address(ptr.value)
        host = torch.frombuffer(raw, dtype=torch.uint8)   # HIP knows the range as page-locked: direct DMA
        base_address = ptr.value
    else:
        plain_host = np.empty(total_bytes + 64, dtype=np.uint8)
        host = torch.from_numpy(plain_host)
        base_address = plain_host.ctypes.data
    step = 1 << 30
    for lo in range(0, total_bytes, step):
        hi = min(lo + step, total_bytes)
        host[lo:hi].copy_(buf[lo:hi])
    torch.cuda.synchronize()
    del buf
    if free_resident:
        holder.clear()
        torch.cuda.empty_cache()
    d2 = np.zeros((n, 20), dtype=np.uint8)

    def run(base_ptr):
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        t0 = time.perf_counter()
        native.check(lib.snappy_sha1_batch(base_ptr, offsets.ctypes.data, lengths.ctypes.data, n,
                                           d2.ctypes.data), "snappy_sha1_batch")
        dt = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([dt], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    log(f"host copy ready (page-locked: {locked})")
    run(base_address)  # first call sizes the library's device arena
    log("first end-to-end call done")
    t_host = min(run(base_address) for _ in range(2))
    same = bool(np.array_equal(d2, digests))
    out = {"seconds": t_host, "digests_match_resident_run": same, "page_locked": locked}
    if pageable and locked and dist is None and total_bytes <= 20 * GB:
        plain = np.empty(total_bytes + 64, dtype=np.uint8)
        plain[:] = np.ctypeslib.as_array(raw)
        run(plain.ctypes.data)
        out["pageable_seconds"] = min(run(plain.ctypes.data) for _ in range(2))
        out["pageable_digests_match"] = bool(np.array_equal(d2, digests))
        del plain
    del host
    if locked:
        del raw
        native.check(lib.snappy_host_free(ptr), "snappy_host_free")
    return out


def dry_run(args, rank, world):
    """The N-rank control flow of main() without a GPU: this rank's share of the corpus (dealt out
    by the library's own host-side planner), barrier, timed steps, MAX of the times and SUM of the
    bytes over ranks, one JSON line from rank 0."""
    import hashlib

    import torch
    import torch.distributed as dist

    from snappy_amd import native

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    small = argparse.Namespace(**vars(args))
    small.workload = args.workload if args.workload != "mcdf" else "folder"
    small.scale = (2e-4 if small.workload == "corpus" else 2e-3) * args.scale
    lengths, corpus_bytes = rank_corpus(small, native.lib(), rank, world)
    offsets = np.concatenate([[0], np.cumsum(lengths)[:-1]]).astype(np.uint64)
    base = np.random.default_rng(rank).integers(0, 256, size=int(lengths.sum()) + 1, dtype=np.uint8)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        # hashlib stands in for the kernels: this mode only exercises the rank plumbing
        digests = [hashlib.sha1(base[int(o): int(o + n)].tobytes()).digest() for o, n in zip(offsets, lengths)]
    elapsed = time.perf_counter() - t0
    job_bytes = float(lengths.sum())
    job_files = float(len(lengths))
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        tot = torch.tensor([job_bytes, job_files], dtype=torch.float64)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        job_bytes, job_files = float(tot[0].item()), float(tot[1].item())
    sharded = None
    if world > 1 and not is_strong(small, world) and small.workload == "corpus":
        # main()'s second measurement with several ranks: one corpus dealt out over them.
        whole = make_sizes("corpus", np.random.default_rng(CORPUS_SEED), small.scale)
        shard_of, _ = shard_lpt(native.lib(), whole, world)
        mine = whole[shard_of == rank]
        tot = torch.tensor([float(mine.sum()), float(len(mine))], dtype=torch.float64)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        sharded = {"job_bytes": float(tot[0].item()), "job_files": float(tot[1].item()),
                   "corpus_bytes": float(whole.sum()), "corpus_files": float(len(whole))}
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": world, "steps": args.steps, "one_corpus_sharded": sharded,
                          "scaling": "strong" if is_strong(small, world) else "weak",
                          "value": job_bytes * args.steps / elapsed / GB, "unit": "GB/s",
                          "job_bytes": job_bytes, "job_files": job_files, "corpus_bytes": corpus_bytes,
                          "files_rank0": int(len(lengths)), "bytes_rank0": int(lengths.sum()),
                          "largest_rank0": int(lengths.max()) if len(lengths) else 0,
                          "digest0": digests[0].hex() if digests else ""}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def measured_traffic(workload, scale):
    """HBM bytes per launch from the committed rocprofv3 counter pass for this workload
    (profiles/hbm_traffic.json; collected and corrected as that file says), or None."""
    try:
        with open(os.path.join(ROOT, "profiles", "hbm_traffic.json")) as f:
            rows = json.load(f)
    except (OSError, ValueError):
        return None
    row = rows.get(workload) if scale == 1.0 else None
    return row.get("hbm_bytes_per_launch") if row else None


def last_plan(lib):
    """How the planner cut the last batch: files per execution path and its expected time."""
    ends = (ctypes.c_uint32 * 4)()
    ms = ctypes.c_double(0.0)
    if lib.snappy_sha1_last_plan(ends, ctypes.byref(ms)) != 0:
        return None
    e = [0] + list(ends)
    names = ("64_lanes_per_file", "16_lanes_per_file", "4_lanes_per_file", "1_lane_per_file")
    return {"files": {names[k]: e[k + 1] - e[k] for k in range(4)}, "planned_ms": ms.value}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="corpus", choices=["corpus", "blobs", "small", "folder", "mcdf"])
    ap.add_argument("--archives", type=int, default=8, help="mcdf workload: archives per GPU")
    ap.add_argument("--archive-mib", type=float, default=128.0, help="mcdf workload: decoded MiB per archive")
    ap.add_argument("--mcdf-content", default="asset", choices=["asset", "dense"],
                    help="mcdf workload: file bodies of long runs (random, zeros, repeated motifs) or of dense "
                         "short LZ4 sequences (short literal runs between short copies)")
    ap.add_argument("--mcdf-shape", default="uniform", choices=["uniform", "spec"],
                    help="mcdf workload: 'spec' = archives of 64-512 MiB (uniform), 50-300 files each, both kinds of "
                         "content (use with --archives 64)")
    ap.add_argument("--mcdf-decode-only", action="store_true",
                    help="mcdf workload: only the decode kernel, all archives' chunks in one launch (counter passes)")
    ap.add_argument("--scale", type=float, default=1.0, help="shrink or grow the corpus")
    ap.add_argument("--strong", action="store_true",
                    help="one corpus for the whole job, sharded by file over the ranks, as the timed workload")
    ap.add_argument("--weak", action="store_true",
                    help="every rank hashes a corpus of its own (the default; with N > 1 ranks the corpus workload "
                         "also measures ONE corpus dealt out over the ranks and reports it beside the value)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-dry-run", action="store_true",
                    help="no GPU: exercise the multi-rank plumbing with hashlib standing in for the "
                         "kernels on a tiny corpus (used by the gloo test; the numbers mean nothing)")
    ap.add_argument("--cpu-sample-gb", type=float, default=4.0)
    ap.add_argument("--only-concurrent", action="store_true",
                    help="measure only the two-callers-at-once case (other_workloads.concurrent_callers) and print it")
    ap.add_argument("--no-extra", action="store_true",
                    help="skip the brief measurements of the other configurations appended to the default run")
    ap.add_argument("--e2e-max-gb", type=float, default=120.0,
                    help="largest per-GPU shard for which the end-to-end leg (a page-locked host copy) is run")
    ap.add_argument("--no-e2e", action="store_true",
                    help="skip the end-to-end leg (same files from host memory through snappy_sha1_batch)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    if world > 1 and not args.cpu_dry_run:
        # One process per GPU: make this rank's GPU the only one its HIP runtime sees, so
        # that the host-pointer entry points (which use every visible device) stay on it.
        visible = os.environ.get("HIP_VISIBLE_DEVICES")
        ids = [v for v in visible.split(",") if v] if visible else None
        if os.environ.get("BENCH_REHEARSE_ON_ONE_GPU") == "1":
            # Rehearsal of the N-rank flow on a one-GPU box: every rank shares GPU 0 (the numbers mean nothing).
            os.environ["HIP_VISIBLE_DEVICES"] = ids[0] if ids else "0"
        else:
            os.environ["HIP_VISIBLE_DEVICES"] = ids[local_rank] if ids and local_rank < len(ids) else str(local_rank)
        local_rank = 0

    import torch

    if args.cpu_dry_run:
        return dry_run(args, rank, world)

    import snappy_amd  # noqa: F401
    from snappy_amd import native

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device is visible")
    dist = None
    if world > 1:
        import torch.distributed as dist_mod

        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # Control plane only (barrier, max of timings): no data-path collective exists.
        dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    lib = native.lib()
    native.check(lib.snappy_init(0), "snappy_init")
    lib.snappy_profile_enable(1)

    def finish():
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()

    if args.only_concurrent:
        print(json.dumps({"concurrent_callers": concurrent_callers(lib, native, torch, device, local_rank)}), flush=True)
        return finish()

    if args.workload == "mcdf":
        out = run_mcdf(args, rank, world, local_rank, dist)
        if out is not None:
            print(json.dumps(out), flush=True)
        return finish()

    strong = is_strong(args, world)
    lengths, corpus_bytes = rank_corpus(args, lib, rank, world)
    offsets = np.concatenate([[0], np.cumsum(lengths)[:-1]]).astype(np.uint64)
    total_bytes = int(lengths.sum())
    algo_bytes = int(padded_blocks(lengths).sum()) * 64
    log(f"corpus: {len(lengths)} files, {total_bytes} bytes; filling HBM")
    buf = fill_random(torch, total_bytes, device, seed=1234 + rank)
    torch.cuda.synchronize()
    log("resident data ready")
    n = len(lengths)
    n_cu = torch.cuda.get_device_properties(device).multi_processor_count

    def barrier():
        if dist is not None:
            dist.barrier()

    elapsed, kernel_ms, digests = hash_resident(lib, native, torch, device, local_rank, buf, offsets, lengths,
                                                args.steps, args.warmup, args.workload == "small", barrier)
    log(f"timed steps done: {elapsed:.3f} s")
    plan = last_plan(lib)
    k_ms = float(np.mean(kernel_ms))
    per_rank = [[float(total_bytes), float(n), k_ms, float(lengths.max()) if n else 0.0]]
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        rows = [torch.zeros(4, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(rows, torch.tensor(per_rank[0], dtype=torch.float64))
        per_rank = [r.tolist() for r in rows]
    job_bytes = float(sum(r[0] for r in per_rank))

    # With several ranks: ONE corpus of the same kind dealt out by file over the ranks (each rank's
    # share read out of the data it already holds; SHA-1 timing does not depend on content).
    sharded = None
    e2e_offsets, e2e_lengths, e2e_bytes, e2e_digests = offsets, lengths, total_bytes, digests
    if dist is not None and not strong and args.workload == "corpus" and not args.no_extra:
        whole = make_sizes(args.workload, np.random.default_rng(CORPUS_SEED), args.scale)
        shard_of, _ = shard_lpt(lib, whole, world)
        s_len = whole[shard_of == rank]
        s_off = np.concatenate([[0], np.cumsum(s_len)[:-1]]).astype(np.uint64)
        assert int(s_len.sum()) <= total_bytes, "a rank's share of one corpus is smaller than a corpus"
        s_elapsed, s_kernel_ms, s_digests = hash_resident(lib, native, torch, device, local_rank, buf, s_off, s_len,
                                                          min(args.steps, 3), 1, False, barrier)
        t = torch.tensor([s_elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        rows = [torch.zeros(4, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(rows, torch.tensor([float(s_len.sum()), float(len(s_len)), float(np.mean(s_kernel_ms)),
                                            float(s_len.max()) if len(s_len) else 0.0], dtype=torch.float64))
        sharded = {"elapsed": float(t.item()), "steps": min(args.steps, 3), "rows": [r.tolist() for r in rows],
                   "corpus_bytes": int(whole.sum())}
        # The end-to-end leg runs on this share: the host copies of all ranks together are one corpus.
        e2e_offsets, e2e_lengths, e2e_bytes, e2e_digests = s_off, s_len, int(s_len.sum()), s_digests
        log(f"one corpus over {world} ranks: {sharded['elapsed']:.3f} s")

    # Bounded host sample for the CPU baseline, taken while the data is still resident.
    sample = None
    if rank == 0 and not args.no_cpu_baseline:
        cap = int(args.cpu_sample_gb * GB)
        k = max(1, int(np.searchsorted(np.cumsum(lengths), cap, side="right")))
        k = min(k, n)
        sample_bytes = int(offsets[k - 1] + lengths[k - 1])
        sample = (buf[:sample_bytes].cpu().numpy(), k)

    e2e = None
    run_e2e = not args.no_e2e and e2e_bytes <= args.e2e_max_gb * GB
    if dist is not None:   # every rank or none: the leg has barriers of its own
        flag = torch.tensor([1.0 if run_e2e else 0.0], dtype=torch.float64)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        run_e2e = bool(flag.item() > 0)
    log(f"end-to-end leg: {run_e2e}")
    if run_e2e:
        holder = {"buf": buf}
        del buf
        e2e = end_to_end(lib, native, torch, holder, e2e_offsets, e2e_lengths, e2e_bytes, e2e_digests, dist,
                         pageable=False, free_resident=True)
    else:
        del buf
    torch.cuda.empty_cache()

    locked = bool(e2e and e2e["page_locked"])
    if dist is not None and run_e2e:   # page-locked on every rank, or said so
        flag = torch.tensor([1.0 if locked else 0.0], dtype=torch.float64)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        locked = bool(flag.item() > 0)
    log("end-to-end leg done")
    if rank != 0:
        return finish()

    ms_per_step = elapsed / args.steps * 1e3
    value = job_bytes * args.steps / elapsed / GB
    max_blocks = int(padded_blocks(lengths).max())
    per = "job" if strong else "GPU"
    out = {
        "metric": METRIC,
        "value": value,
        "unit": "GB/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "strong" if strong else "weak",
        "vs_baseline": None,
        "dtype": "u32",
        "data": "synthetic",
        "config": {
            "workload": WORKLOAD_TEXT[args.workload].format(per=per, gb=(corpus_bytes or total_bytes) / GB),
            "files_per_gpu": [int(r[1]) for r in per_rank],
            "bytes_per_gpu": [int(r[0]) for r in per_rank],
            "largest_file_bytes_per_gpu": [int(r[3]) for r in per_rank],
            "kernel_ms_per_gpu": [r[2] for r in per_rank],
            "kernel_ms_each_step_rank0": [round(float(x), 3) for x in kernel_ms],
            "parallelism": (f"one corpus dealt out by file over {world} GPU(s), longest first by padded blocks "
                            "(snappy_plan_shards_lpt)" if strong else
                            "the whole corpus on one GPU" if world == 1 else
                            f"a corpus of its own on each of {world} GPUs: files are independent, every GPU's "
                            "batch is hashed where it lies") + "; host dispatch, no collective",
        },
        "kernel_only": {
            "value": value, "unit": "GB/s", "ms": ms_per_step,
            "note": "a GPU's time is bounded below by the serial SHA-1 chain of its largest file (about "
                    f"{max(r[3] for r in per_rank) / 1e6:.1f} MB here): more files per GPU cost nothing until the "
                    "SIMDs are full, and more GPUs do not shorten a chain"},
        "roofline": sha1_roofline(algo_bytes, k_ms, max_blocks, n_cu, plan, measured_traffic(args.workload, args.scale)),
    }
    e2e_job_bytes = job_bytes
    if sharded is not None:
        # Configuration 4 as it is written: one corpus sharded by file over the ranks.
        rows = sharded["rows"]
        e2e_job_bytes = float(sum(r[0] for r in rows))
        out["one_corpus_sharded"] = {
            "workload": WORKLOAD_TEXT["corpus"].format(per="job", gb=sharded["corpus_bytes"] / GB) +
                        f", dealt out by file over {world} GPUs, longest first by padded blocks (snappy_plan_shards_lpt)",
            "scaling": "strong",
            "value": e2e_job_bytes * sharded["steps"] / sharded["elapsed"] / GB, "unit": "GB/s",
            "steps": sharded["steps"], "ms_per_step": sharded["elapsed"] / sharded["steps"] * 1e3,
            "files_per_gpu": [int(r[1]) for r in rows], "bytes_per_gpu": [int(r[0]) for r in rows],
            "largest_file_bytes_per_gpu": [int(r[3]) for r in rows], "kernel_ms_per_gpu": [r[2] for r in rows],
            "note": "flat in the number of GPUs by construction: one GPU already finishes the whole corpus in the "
                    "time of its largest file's chain, every share holds a file of about that size, and more GPUs do "
                    "not shorten a chain; end to end (below, bound by PCIe per GPU) is what shrinks with N"}
    if e2e is not None:
        bw = h2d_bandwidth(torch, device)
        e2e_rows = sharded["rows"] if sharded is not None else per_rank
        (out["one_corpus_sharded"] if sharded is not None else out)["end_to_end"] = {
            "value": e2e_job_bytes / e2e["seconds"] / GB, "unit": "GB/s",
            "ms": e2e["seconds"] * 1e3,
            "path": ("page-locked host shard (whole, no ring)" if locked else
                     "host shard in ordinary memory (more than a process may page-lock here), moved through the "
                     "library's ring of page-locked buffers")
                    + " -> HBM -> kernels -> digests on the host, one snappy_sha1_batch call per GPU; the "
                      "slowest GPU's time counts",
            "host_memory_page_locked": locked,
            "h2d_pinned_GBps_measured": bw,
            "frac_of_h2d": e2e_job_bytes / e2e["seconds"] / GB / (bw * world),
            "copy_only_ms_at_that_rate": max(r[0] for r in e2e_rows) / bw / GB * 1e3,
            "digests_match_resident_run": e2e["digests_match_resident_run"]}

    if world == 1 and args.workload == "corpus" and args.scale == 1.0 and not args.no_extra:
        out["other_workloads"] = extra_workloads(args, lib, native, torch, device, local_rank, n_cu)

    if sample is not None:
        host, k = sample
        base, ref = cpu_baseline(host, offsets[:k], lengths[:k])
        out["cpu_baseline"] = base
        out["digests_match_cpu_oracle"] = bool(np.array_equal(ref, digests[:k]))
        if not out["digests_match_cpu_oracle"]:
            print("ERROR: GPU digests differ from the CPU oracle", file=sys.stderr)
    print(json.dumps(out), flush=True)
    finish()
    if out.get("digests_match_cpu_oracle") is False:
        sys.exit(1)


if __name__ == "__main__":
    main()
