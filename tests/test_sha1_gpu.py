"""SHA-1 parity on the GPU: every call goes through the C ABI and is checked bit for
bit against the CPU oracle (and the committed hashlib golden digests)."""
import os

import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu

# Sends every message of a batch down one execution path: 64, 16, 4 or 1 lanes of a
# wavefront per message (0..3).  Unset: the planner cuts the batch itself.
FORCE = "SNAPPY_SHA1_FORCE_PATH"
ALL_PATHS = ("0", "1", "2", "3")


def force_path(path):
    if path is None:
        os.environ.pop(FORCE, None)
    else:
        os.environ[FORCE] = path


@pytest.fixture(autouse=True)
def _clean_env():
    keep = {k: os.environ.get(k) for k in (FORCE, "SNAPPY_OVERLAP_MIN_BYTES", "SNAPPY_OVERLAP_SLICE_BYTES",
                                           "SNAPPY_ARRIVAL_MAX_POLLS", "SNAPPY_TEST_TRANSFER_DELAY_MS")}
    yield
    for k, v in keep.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v


def test_device_is_bound(snappy):
    L = snappy.native.lib()
    assert L.snappy_init(0) == 0
    assert L.snappy_device_count() >= 1


def test_fips_and_padding_edges(snappy):
    g = H.load_golden("sha1_golden.json")
    for e in g["fips"]:
        data = e["ascii"].encode() if "ascii" in e else e["repeat"][0].encode() * e["repeat"][1]
        assert snappy.get_hash(data) == e["sha1"]
    for n, want in g["zeros"].items():
        assert snappy.get_hash(b"\0" * int(n)) == want


def test_golden_splitmix_single_calls(snappy):
    g = H.load_golden("sha1_golden.json")
    for row in g["splitmix"][:140]:
        data = H.splitmix_bytes(row["len"], H.file_seed(row["index"]))
        assert snappy.get_hash(data) == row["sha1"], row


def test_golden_splitmix_one_batch_every_path(snappy, oracle):
    """All golden rows in one packed batch, once per execution path."""
    g = H.load_golden("sha1_golden.json")
    blobs = [H.splitmix_bytes(r["len"], H.file_seed(r["index"])) for r in g["splitmix"]]
    want = [r["sha1"] for r in g["splitmix"]]
    base = np.frombuffer(b"".join(blobs), dtype=np.uint8)
    lengths = np.array([len(b) for b in blobs], dtype=np.uint64)
    offsets = np.concatenate([[0], np.cumsum(lengths)[:-1]]).astype(np.uint64)
    from snappy_amd.crypto import digests_to_hex

    for path in ALL_PATHS + (None,):
        force_path(path)
        got = digests_to_hex(snappy.sha1_batch(base, offsets, lengths))
        assert got == want, path


@pytest.mark.parametrize("path", ALL_PATHS)
def test_random_unaligned_slices_vs_oracle(snappy, oracle, path):
    """Slices at every byte alignment, overlapping, empty, and ending at the buffer's last byte."""
    force_path(path)
    rng = np.random.default_rng(11)
    base = rng.integers(0, 256, size=300_000, dtype=np.uint8)
    n = 700 if path == "0" else 3000
    lengths = rng.integers(0, 2500, size=n).astype(np.uint64)
    lengths[::97] = 0
    offsets = rng.integers(0, 300_000 - 2500, size=n).astype(np.uint64)
    # force tails at the very end of the allocation, all four alignments
    for k in range(8):
        lengths[k] = 55 + k * 9
        offsets[k] = base.size - int(lengths[k])
    got = snappy.sha1_batch(base, offsets, lengths)
    want = oracle.sha1_batch_np(base, offsets, lengths)
    assert np.array_equal(got, want)


def test_every_length_0_to_300_every_alignment(snappy, oracle):
    base = np.frombuffer(H.random_bytes(4096, 3), dtype=np.uint8)
    offs, lens = [], []
    for a in range(4):
        for n in range(0, 301):
            offs.append(a + (n % 7) * 4)
            lens.append(n)
    offs = np.array(offs, dtype=np.uint64)
    lens = np.array(lens, dtype=np.uint64)
    want = oracle.sha1_batch_np(base, offs, lens)
    for path in ALL_PATHS:
        force_path(path)
        assert np.array_equal(snappy.sha1_batch(base, offs, lens), want), path


def test_separate_buffers_and_single_calls(snappy, oracle):
    blobs = [H.random_bytes(n, n) for n in (0, 1, 63, 64, 65, 1000, 70_000, 1_500_001)]
    got = snappy.sha1_files(blobs)
    for i, b in enumerate(blobs):
        assert got[i].tobytes() == oracle.sha1(b)
        assert snappy.sha1(b) == oracle.sha1(b)
    assert snappy.sha1_files([]).shape == (0, 20)


def test_snapshot_folder_mix_planner_split(snappy, oracle):
    """A scaled-down snapshot folder: .tex-like and .mdl-like sizes, planner picks the paths."""
    rng = np.random.default_rng(21)
    sizes = np.concatenate([H.mixed_sizes(rng, 40, 64 << 10, 3 << 20),
                            H.mixed_sizes(rng, 20, 32 << 10, 1 << 20)])
    blobs = [H.random_bytes(int(s), 1000 + i) for i, s in enumerate(sizes)]
    got = snappy.sha1_files(blobs)
    for i, b in enumerate(blobs):
        assert got[i].tobytes() == oracle.sha1(b), i


def test_device_resident_batch(snappy, oracle):
    import torch

    rng = np.random.default_rng(31)
    host = rng.integers(0, 256, size=8 << 20, dtype=np.uint8)
    dev = torch.from_numpy(host).cuda()
    lengths = rng.integers(0, 40_000, size=500).astype(np.uint64)
    offsets = rng.integers(0, host.size - 40_000, size=500).astype(np.uint64)
    want = oracle.sha1_batch_np(host, offsets, lengths)
    got = snappy.sha1_batch_device(0, dev.data_ptr(), offsets, lengths)
    assert np.array_equal(got, want)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        got2 = snappy.sha1_batch_device(0, dev.data_ptr(), offsets, lengths, stream=s.cuda_stream)
    assert np.array_equal(got2, want)


def test_large_files_properties(snappy, oracle):
    """Full-size blobs: one 64 MiB file against the oracle, plus a prefix-extension
    property across files (digest of a prefix equals the digest of the same bytes hashed alone)."""
    import torch

    n = 64 << 20
    dev = torch.randint(0, 256, (n + 13,), dtype=torch.uint8, device="cuda")
    host = dev.cpu().numpy()
    offsets = np.array([0, 5, 13, 1, 7], dtype=np.uint64)
    lengths = np.array([n, 4 << 20, 16 << 20, (32 << 20) + 3, 64], dtype=np.uint64)
    got = snappy.sha1_batch_device(0, dev.data_ptr(), offsets, lengths)
    want = oracle.sha1_batch_np(host, offsets, lengths)
    assert np.array_equal(got, want)
    again = snappy.sha1_batch(host, offsets, lengths)
    assert np.array_equal(again, want)


def test_concurrent_callers(snappy, oracle):
    """Two threads hash at once, as the UI thread and the migration task may."""
    import threading

    blobs = [H.random_bytes(200_000 + 977 * i, i) for i in range(24)]
    want = [oracle.sha1(b) for b in blobs]
    errors = []

    def work(lo, hi):
        try:
            for _ in range(5):
                got = snappy.sha1_files(blobs[lo:hi])
                for k in range(hi - lo):
                    if got[k].tobytes() != want[lo + k]:
                        errors.append((lo, k))
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=work, args=(0, 12)), threading.Thread(target=work, args=(12, 24)),
               threading.Thread(target=work, args=(6, 18))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors


def test_fully_resident_batch(snappy, oracle):
    """Tables and digests in HBM as well: planned and hashed without host tables."""
    import torch

    from snappy_amd.crypto import sha1_batch_resident

    rng = np.random.default_rng(41)
    host = rng.integers(0, 256, size=24 << 20, dtype=np.uint8)
    dev = torch.from_numpy(host).cuda()
    for n, hi in ((1, 5000), (700, 30_000), (5000, 3000), (40, 2_000_000)):
        lengths = rng.integers(0, hi, size=n).astype(np.int64)
        offsets = rng.integers(0, host.size - hi, size=n).astype(np.int64)
        d_off = torch.from_numpy(offsets).cuda()
        d_len = torch.from_numpy(lengths).cuda()
        d_dig = torch.zeros(n * 20, dtype=torch.uint8, device="cuda")
        for forced in (None,) + ALL_PATHS:
            force_path(forced)
            d_dig.zero_()
            sha1_batch_resident(0, dev.data_ptr(), d_off.data_ptr(), d_len.data_ptr(), n, d_dig.data_ptr())
            got = d_dig.cpu().numpy().reshape(n, 20)
            want = oracle.sha1_batch_np(host, offsets.astype(np.uint64), lengths.astype(np.uint64))
            assert np.array_equal(got, want), (n, hi, forced)


def test_ragged_groups_and_partial_waves(snappy, oracle):
    """Groups of one wave holding messages of very different block counts (some empty,
    some ending mid-superblock), and message counts that leave groups of the last wave idle."""
    rng = np.random.default_rng(51)
    base = rng.integers(0, 256, size=2 << 20, dtype=np.uint8)
    for n in (1, 3, 5, 17, 63, 65, 130):
        lengths = rng.choice(np.array([0, 1, 55, 56, 63, 64, 119, 120, 128, 1000, 4096, 4159, 70_001,
                                       300_000], dtype=np.uint64), size=n)
        offsets = rng.integers(0, base.size - 300_000, size=n).astype(np.uint64)
        want = oracle.sha1_batch_np(base, offsets, lengths)
        for path in ALL_PATHS + (None,):
            force_path(path)
            assert np.array_equal(snappy.sha1_batch(base, offsets, lengths), want), (n, path)


@pytest.mark.parametrize("slice_bytes", ["4096", "1048576"])
def test_hash_while_copying(snappy, oracle, slice_bytes):
    """The host-pointer entry points start the kernel before the data has arrived and feed it
    slice by slice.  Forced on for small batches here, with slices far smaller than the files;
    the second and third calls reuse the same device arena with different bytes (a stale cache
    line would show up as a wrong digest)."""
    os.environ["SNAPPY_OVERLAP_MIN_BYTES"] = "0"
    os.environ["SNAPPY_OVERLAP_SLICE_BYTES"] = slice_bytes
    rng = np.random.default_rng(61)
    sizes = np.concatenate([rng.integers(0, 3000, size=300), rng.integers(10_000, 400_000, size=40),
                            np.array([0, 1, 63, 64, 2_000_003])]).astype(np.uint64)
    rng.shuffle(sizes)
    offsets = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.uint64)
    for rep in range(3):
        base = rng.integers(0, 256, size=int(sizes.sum()) + 1, dtype=np.uint8)
        want = oracle.sha1_batch_np(base, offsets, sizes)
        assert np.array_equal(snappy.sha1_batch(base, offsets, sizes), want), rep
    blobs = [H.random_bytes(int(n), 7000 + i) for i, n in enumerate(sizes[:60])]
    got = snappy.sha1_files(blobs)
    for i, b in enumerate(blobs):
        assert got[i].tobytes() == oracle.sha1(b), i
    for path in ALL_PATHS:
        force_path(path)
        base = rng.integers(0, 256, size=int(sizes.sum()) + 1, dtype=np.uint8)
        assert np.array_equal(snappy.sha1_batch(base, offsets, sizes),
                              oracle.sha1_batch_np(base, offsets, sizes)), path


def test_page_locked_sources(snappy, oracle):
    """Files read into memory from snappy_host_alloc travel by direct DMA (no staging ring)."""
    import ctypes

    os.environ["SNAPPY_OVERLAP_MIN_BYTES"] = "0"
    os.environ["SNAPPY_OVERLAP_SLICE_BYTES"] = "262144"
    L = snappy.native.lib()
    rng = np.random.default_rng(71)
    sizes = rng.integers(0, 600_000, size=50).astype(np.uint64)
    total = int(sizes.sum())
    ptr = ctypes.c_void_p()
    assert L.snappy_host_alloc(total + 1, ctypes.byref(ptr)) == 0
    try:
        pinned = np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(ctypes.c_uint8)), shape=(total + 1,))
        pinned[:] = rng.integers(0, 256, size=total + 1, dtype=np.uint8)
        offsets = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.uint64)
        want = oracle.sha1_batch_np(pinned, offsets, sizes)
        assert np.array_equal(snappy.sha1_batch(pinned, offsets, sizes), want)
        del pinned
    finally:
        assert L.snappy_host_free(ptr) == 0
    assert L.snappy_host_alloc(0, ctypes.byref(ptr)) == 0 and not ptr.value
    assert L.snappy_host_free(None) == 0


def test_offsets_beyond_4_gib(snappy, oracle):
    """64-bit offset arithmetic: files that lie past the 4 GiB and 8 GiB marks of a 9 GiB buffer."""
    import torch

    size = 9 << 30
    dev = torch.empty(size, dtype=torch.uint8, device="cuda")
    rng = np.random.default_rng(81)
    marks = [0, (4 << 30) - 1000, (4 << 30) + 2003, (8 << 30) - 7, size - 300_000]   # disjoint ranges
    lengths = np.array([5000, 2001, 70_000, 14, 300_000], dtype=np.uint64)
    offsets = np.array(marks, dtype=np.uint64)
    want = []
    for off, n in zip(marks, lengths):
        blob = rng.integers(0, 256, size=int(n), dtype=np.uint8)
        dev[off: off + int(n)] = torch.from_numpy(blob).cuda()
        want.append(oracle.sha1(blob.tobytes()))
    for path in ALL_PATHS + (None,):
        force_path(path)
        got = snappy.sha1_batch_device(0, dev.data_ptr(), offsets, lengths)
        assert [g.tobytes() for g in got] == want, path


def test_waiting_kernel_next_to_allocating_caller(snappy, oracle):
    """One thread runs hash-while-copying batches (a kernel that waits for its transfer), another
    thread keeps making calls whose growing sizes force the library to free and allocate device
    and page-locked memory meanwhile.  Nobody may stall: the waiting kernel's own time limit would
    turn a deadlock into an error here."""
    import threading
    import time

    os.environ["SNAPPY_OVERLAP_MIN_BYTES"] = str(1 << 20)
    os.environ["SNAPPY_OVERLAP_SLICE_BYTES"] = str(1 << 20)
    rng = np.random.default_rng(91)
    big = rng.integers(0, 256, size=48 << 20, dtype=np.uint8)
    big_len = np.array([20 << 20, 9 << 20, 5 << 20, 1 << 20, 777], dtype=np.uint64)
    big_off = np.concatenate([[0], np.cumsum(big_len)[:-1]]).astype(np.uint64)
    big_want = oracle.sha1_batch_np(big, big_off, big_len)
    errors = []

    def overlapped():
        try:
            for _ in range(6):
                if not np.array_equal(snappy.sha1_batch(big, big_off, big_len), big_want):
                    errors.append("overlapped digests")
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    def growing():
        try:
            for k in range(12):
                n = (3 + 5 * k) << 18          # below the overlap threshold at first, then above
                blob = rng.integers(0, 256, size=n, dtype=np.uint8)
                if snappy.sha1(blob.tobytes()) != oracle.sha1(blob.tobytes()):
                    errors.append(("growing", k))
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    t0 = time.time()
    threads = [threading.Thread(target=overlapped), threading.Thread(target=growing)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    assert time.time() - t0 < 9.0      # a stall would have lasted the kernel's 10 s limit


def test_held_up_transfer_still_gives_right_digests(snappy, oracle):
    """The kernel's patience is cut to a fraction of a second and the transfer is delayed beyond
    it: the waves give up, and the call then hashes the data where it has arrived meanwhile."""
    os.environ["SNAPPY_OVERLAP_MIN_BYTES"] = "0"
    os.environ["SNAPPY_OVERLAP_SLICE_BYTES"] = "65536"
    os.environ["SNAPPY_ARRIVAL_MAX_POLLS"] = "1000"
    os.environ["SNAPPY_TEST_TRANSFER_DELAY_MS"] = "700"
    rng = np.random.default_rng(95)
    sizes = rng.integers(0, 300_000, size=40).astype(np.uint64)
    offsets = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.uint64)
    base = rng.integers(0, 256, size=int(sizes.sum()) + 1, dtype=np.uint8)
    assert np.array_equal(snappy.sha1_batch(base, offsets, sizes), oracle.sha1_batch_np(base, offsets, sizes))
    os.environ.pop("SNAPPY_TEST_TRANSFER_DELAY_MS")
    os.environ.pop("SNAPPY_ARRIVAL_MAX_POLLS")
    assert np.array_equal(snappy.sha1_batch(base, offsets, sizes), oracle.sha1_batch_np(base, offsets, sizes))


def test_single_message_calls_on_every_path(snappy):
    """A batch of one message skips the sorting kernels: one GetHash call per length, on the
    planner's path and on every forced one."""
    import hashlib
    for path in ALL_PATHS + (None,):
        force_path(path)
        for n in (0, 1, 55, 56, 63, 64, 65, 119, 120, 1000, 4096, 100_003, 1_500_000):
            data = H.splitmix_bytes(n, 900 + n)
            assert snappy.get_hash(data) == hashlib.sha1(data).hexdigest().upper(), (path, n)
    force_path(None)


def test_planner_cost_table_is_measured_on_the_device(snappy, oracle):
    """The first hashing call measures the device's own chain time and scales the planner's table
    by it; the table must stay plausible (ordered, near what one MI355X gave) and the digests
    unaffected."""
    import ctypes

    L = snappy.native.lib()
    blob = H.random_bytes(300_000, 3)
    assert snappy.sha1(blob) == oracle.sha1(blob)
    chain = (ctypes.c_double * 4)()
    simd = (ctypes.c_double * 4)()
    measured = ctypes.c_int(0)
    snappy.native.check(L.snappy_sha1_cost_model(0, chain, simd, ctypes.byref(measured)), "cost model")
    chain, simd = list(chain), list(simd)
    # The measurement is dropped when the GPU was busy at that moment: either way the table holds.
    assert measured.value in (0, 1), (chain, simd)
    assert chain[0] < chain[1] < chain[2] < chain[3]
    assert simd[0] > simd[1] > simd[2] > simd[3]
    for got, table in zip(chain, (716.0, 795.0, 960.0, 1300.0)):
        assert 0.6 * table < got < 1.6 * table, chain
    for k, per_wave in enumerate((1, 4, 16, 64)):
        assert abs(simd[k] * per_wave - chain[k]) < 1e-6 * chain[k], (chain, simd)


def test_planner_prediction_on_hostile_size_mixes(snappy):
    """The planner's expected finishing time against the kernel's own time on size mixes chosen to
    hurt a list scheduler: one huge file beside 10^5 tiny ones, a few hundred equal large files
    (fewer than SIMDs), a few thousand (more than SIMDs), all-equal mid-size files, two size
    classes.  The kernel must finish within 1.10 x the expected time (= max(longest piece, average
    SIMD load + half a late piece) under the planner's cost table), the expectation must not be
    wildly pessimistic either, and sampled digests must be right."""
    import ctypes
    import hashlib

    import torch

    L = snappy.native.lib()
    L.snappy_profile_enable(1)
    mixes = {
        "one 64 MB + 10^5 x 8 KB": np.concatenate([[64 << 20], np.full(100_000, 8 << 10)]),
        "500 x 8 MB": np.full(500, 8 << 20),
        "3000 x 8 MB": np.full(3000, 8 << 20),
        "20000 x 1 MB": np.full(20000, 1 << 20),
        "2000 x 3 MB + 10^5 x 64 KB": np.concatenate([np.full(2000, 3 << 20), np.full(100_000, 64 << 10)]),
    }
    cap = int(max(int(m.sum()) for m in mixes.values()))
    gen = torch.Generator(device="cuda")
    gen.manual_seed(7)
    buf = torch.empty(cap, dtype=torch.uint8, device="cuda")
    step = 1 << 30
    for lo in range(0, cap, step):   # random bytes, a piece at a time (randint works in int64)
        hi = min(cap, lo + step)
        buf[lo:hi] = torch.randint(0, 256, (hi - lo,), dtype=torch.uint8, device="cuda", generator=gen)
    try:
        for label, lengths in mixes.items():
            lengths = lengths.astype(np.int64) - 9 + (np.arange(len(lengths)) % 7)   # not multiples of 64
            offsets = np.concatenate([[0], np.cumsum(lengths)[:-1]]).astype(np.int64)
            n = len(lengths)
            d_off = torch.from_numpy(offsets).cuda()
            d_len = torch.from_numpy(lengths).cuda()
            d_dig = torch.zeros(n * 20, dtype=torch.uint8, device="cuda")
            ms = []
            for _ in range(4):
                snappy.native.check(L.snappy_sha1_batch_resident(0, buf.data_ptr(), d_off.data_ptr(), d_len.data_ptr(),
                                                                 n, d_dig.data_ptr(), None), "resident")
                ms.append(L.snappy_profile_last_ms(0))
            ends = (ctypes.c_uint32 * 4)()
            planned = ctypes.c_double(0.0)
            snappy.native.check(L.snappy_sha1_last_plan(ends, ctypes.byref(planned)), "last plan")
            kernel = min(ms[1:])   # the first launch also pages the plan's buffers in
            assert kernel <= 1.10 * planned.value, (label, ms, planned.value, list(ends))
            assert planned.value <= 1.45 * kernel, (label, ms, planned.value, list(ends))
            got = d_dig.cpu().numpy().reshape(n, 20)
            for i in (0, 1, n // 2, n - 1):
                data = buf[int(offsets[i]): int(offsets[i] + lengths[i])].cpu().numpy().tobytes()
                assert got[i].tobytes() == hashlib.sha1(data).digest(), (label, i)
    finally:
        L.snappy_profile_enable(0)
