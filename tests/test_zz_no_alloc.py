"""The entry point that allocates nothing (snappy_sha1_batch_resident_noalloc) against the
allocating one it is a variant of (snappy_sha1_batch_resident), through the C ABI: two calls
in a row on one GPU, the second with a smaller input, bit-exact digests required.  The call
only enqueues: the tests synchronise the stream themselves, and one captures it into a graph."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _call(fn_name, device, dev, d_off, d_len, n, d_dig, stream=None):
    from snappy_amd import native

    fn = getattr(native.lib(), fn_name)
    native.check(fn(device, dev.data_ptr(), d_off.data_ptr(), d_len.data_ptr(), n, d_dig.data_ptr(),
                    stream), fn_name)


def test_noalloc_matches_allocating_entry_point(snappy, oracle):
    import torch

    from snappy_amd import native

    native.check(native.lib().snappy_init(0), "snappy_init")
    rng = np.random.default_rng(77)
    host = rng.integers(0, 256, size=48 << 20, dtype=np.uint8)
    dev = torch.from_numpy(host).to("cuda:0")
    # First a large ragged batch (every execution path of the kernel), then a smaller one:
    # the fixed planning area still holds the first call's header and task list.
    shapes = [(6000, 40_000, 3), (37, 3_000, 0)]
    for n, hi, n_big in shapes:
        lengths = rng.integers(0, hi, size=n).astype(np.int64)
        if n_big:
            lengths[:n_big] = [9 << 20, (2 << 20) + 5, 700_001]
        offsets = rng.integers(0, host.size - int(lengths.max()), size=n).astype(np.int64)
        d_off = torch.from_numpy(offsets).to("cuda:0")
        d_len = torch.from_numpy(lengths).to("cuda:0")
        want_dev = torch.zeros(n * 20, dtype=torch.uint8, device="cuda:0")
        got_dev = torch.full((n * 20,), 0xEE, dtype=torch.uint8, device="cuda:0")
        torch.cuda.synchronize()
        _call("snappy_sha1_batch_resident", 0, dev, d_off, d_len, n, want_dev)
        _call("snappy_sha1_batch_resident_noalloc", 0, dev, d_off, d_len, n, got_dev)
        torch.cuda.synchronize()   # the entry point enqueues and returns
        want = want_dev.cpu().numpy().reshape(n, 20)
        got = got_dev.cpu().numpy().reshape(n, 20)
        assert np.array_equal(got, want), (n, hi)
        # and the allocating entry point itself agrees with the CPU oracle
        ref = oracle.sha1_batch_np(host, offsets.astype(np.uint64), lengths.astype(np.uint64))
        assert np.array_equal(want, ref), (n, hi)

    # one file, on a caller's stream: nothing is there until the caller has synchronised it
    s = torch.cuda.Stream(device="cuda:0")
    one_off = torch.tensor([12345], dtype=torch.int64, device="cuda:0")
    one_len = torch.tensor([1_000_003], dtype=torch.int64, device="cuda:0")
    a = torch.zeros(20, dtype=torch.uint8, device="cuda:0")
    b = torch.zeros(20, dtype=torch.uint8, device="cuda:0")
    torch.cuda.synchronize()
    _call("snappy_sha1_batch_resident", 0, dev, one_off, one_len, 1, a)
    _call("snappy_sha1_batch_resident_noalloc", 0, dev, one_off, one_len, 1, b, s.cuda_stream)
    s.synchronize()
    assert bytes(a.cpu().numpy()) == bytes(b.cpu().numpy()) == oracle.sha1(host[12345:12345 + 1_000_003].tobytes())


def test_noalloc_captured_into_a_graph_and_replayed(snappy, oracle, tmp_path):
    """What the entry point is for: the call sits inside a stream capture (it must neither
    allocate nor synchronise), and the graph is replayed twice, the second time over other
    bytes in the same buffers.  The capture is done by a plain-HIP program
    (tests/c_driver/capture_probe.cpp, built by __graft_entry__.build()); each replay's digests
    are compared here with the CPU oracle's."""
    import os
    import subprocess

    import helpers as H

    exe = os.path.join(H.ROOT, "tests", "c_driver", "capture_probe")
    assert os.path.exists(exe), "tests/c_driver/capture_probe is not built: run __graft_entry__.build()"
    out_file = str(tmp_path / "digests.bin")
    run = subprocess.run([exe, out_file], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0, run.stdout + run.stderr
    assert "graph nodes: 5" in run.stdout            # one memset and four kernels, nothing else
    assert "replay 0 digests equal" in run.stdout and "replay 1 digests equal" in run.stdout

    n, size = 300, 4 << 20                           # the same figures as in capture_probe.cpp
    i = np.arange(size, dtype=np.uint64)
    host = ((i * np.uint64(2654435761)) >> np.uint64(13)).astype(np.uint8)
    k = np.arange(n, dtype=np.uint64)
    lengths = (k * np.uint64(7919)) % np.uint64(30000)
    lengths[0] = 1500001
    offsets = (k * np.uint64(104729)) % np.uint64(size - 1600000)
    got = np.fromfile(out_file, dtype=np.uint8).reshape(2, n, 20)
    for replay, data in enumerate((host, host ^ np.uint8(0x5A))):
        ref = oracle.sha1_batch_np(np.ascontiguousarray(data), offsets, lengths)
        assert np.array_equal(got[replay], ref), replay


def test_noalloc_rejects_what_it_cannot_hold(snappy):
    import torch

    from snappy_amd import native

    native.check(native.lib().snappy_init(0), "snappy_init")
    t = torch.zeros(64, dtype=torch.uint8, device="cuda:0")
    too_many = (1 << 20) + 1
    st = native.lib().snappy_sha1_batch_resident_noalloc(0, t.data_ptr(), t.data_ptr(), t.data_ptr(),
                                                         too_many, t.data_ptr(), None)
    assert st == native.E_CAPACITY
    st = native.lib().snappy_sha1_batch_resident_noalloc(99, t.data_ptr(), t.data_ptr(), t.data_ptr(),
                                                         1, t.data_ptr(), None)
    assert st == native.E_INVALID_ARG
