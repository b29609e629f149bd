"""The write side (archives): the GPU LZ4 compressor inside lz4net framing, and whole MCDF
archives made with it.  The compressor's blocks need not equal any other compressor's, so the
checks are: the CPU oracle's decoder (which enforces the end-of-block rules of the decoder
lz4net ports) accepts them and returns the input; so does the system liblz4 where present; so
does the GPU decoder; and an archive written here imports with the right digests."""
import ctypes
import ctypes.util
import hashlib

import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu


def _chunks(stream: bytes):
    """Walks the framing: yields (compressed, original_len, payload)."""
    pos = 0

    def varint():
        nonlocal pos
        v = shift = 0
        while True:
            b = stream[pos]
            pos += 1
            v |= (b & 0x7F) << shift
            shift += 7
            if b < 0x80:
                return v

    while pos < len(stream):
        flags = varint()
        n = varint()
        c = varint() if flags & 1 else n
        yield bool(flags & 1), n, stream[pos: pos + c]
        pos += c
    assert pos == len(stream)


@pytest.mark.parametrize("kind", ["asset", "compressible", "random", "zeros", "text"])
@pytest.mark.parametrize("size,block", [(0, 0), (1, 0), (12, 0), (13, 0), (64, 0), (77, 0), (4096, 0),
                                        (70_001, 65536), ((1 << 20) + 1, 0), (3_300_000, 0),
                                        (3_300_000, 262_144)])
def test_encode_roundtrip_through_oracle(snappy, oracle, kind, size, block):
    data = {"asset": H.asset_like, "compressible": H.compressible, "random": H.random_bytes,
            "zeros": lambda n, s: b"\0" * n,
            "text": lambda n, s: (b"the quick brown fox jumps over the lazy dog. " * (n // 45 + 1))[:n]}[kind](size, 5)
    stream = snappy.lz4net_encode(data, block)
    assert oracle.lz4net_decoded_size(stream) == (0, len(data))
    assert oracle.lz4net_decode(stream) == (0, data)
    bs = block or (1 << 20)
    parts = list(_chunks(stream))
    assert [n for _, n, _ in parts] == [min(bs, len(data) - k) for k in range(0, len(data), bs)]
    for compressed, n, payload in parts:
        assert len(payload) < n if compressed else len(payload) == n
    if size >= 70_001 and kind in ("asset", "compressible", "zeros", "text"):
        ratio = len(data) / len(stream)
        assert ratio > {"asset": 1.15, "compressible": 1.5, "zeros": 100, "text": 10}[kind], ratio
    if kind == "random" and size > 4096:
        assert not any(c for c, _, _ in parts)          # stored: random bytes do not shrink


def test_encode_lengths_around_the_steps(snappy, oracle):
    """The compressor looks at 64 or 256 positions per trip, extends a match 64 and then 256 bytes
    per step (a lane's dword may straddle the last byte a match may cover) and copies literals in
    passes of 256 and 1024 bytes: every input length around those figures, on content that is one
    long match (zeros, a 45-byte phrase), one long literal run (random) and a mix of both."""
    sizes = list(range(60, 84)) + list(range(250, 345)) + list(range(1016, 1050)) + list(range(1270, 1300)) + \
        [2303, 2304, 2305, 4607, 5121, 65535, 65536, 65537, 65800]
    phrase = b"the quick brown fox jumps over the lazy dog. "
    noise = H.random_bytes(70_000, 21)
    for n in sizes:
        half = n // 2
        for data in (b"\0" * n, (phrase * (n // 45 + 1))[:n], noise[:n],
                     noise[:half] + noise[:n - half],            # second half repeats the first
                     noise[:n // 3] + b"\0" * (n - n // 3)):
            stream = snappy.lz4net_encode(data)
            assert oracle.lz4net_decode(stream) == (0, data), n


def test_blocks_decode_with_liblz4(snappy):
    name = ctypes.util.find_library("lz4")
    if not name:
        pytest.skip("no system liblz4")
    lz4 = ctypes.CDLL(name)
    lz4.LZ4_decompress_safe.restype = ctypes.c_int
    lz4.LZ4_decompress_safe.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_int]
    data = H.asset_like(2_500_000, 9) + H.compressible(700_000, 3)
    pos = 0
    for compressed, n, payload in _chunks(snappy.lz4net_encode(data, 1 << 19)):
        if compressed:
            out = ctypes.create_string_buffer(n)
            assert lz4.LZ4_decompress_safe(bytes(payload), out, len(payload), n) == n
            assert out.raw == data[pos: pos + n]
        else:
            assert payload == data[pos: pos + n]
        pos += n
    assert pos == len(data)


def test_encode_then_gpu_decode_large(snappy):
    data = H.asset_like(48 << 20, 12)
    stream = snappy.lz4net_encode(data)
    assert len(stream) < len(data)
    assert snappy.lz4net_decode(stream) == data


def test_encode_capacity_and_arguments(snappy):
    L = snappy.native.lib()
    data = np.frombuffer(H.compressible(100_000, 1), dtype=np.uint8)
    n = ctypes.c_uint64(0)
    out = np.zeros(16, dtype=np.uint8)
    st = L.snappy_lz4net_encode(data.ctypes.data, data.size, 0, out.ctypes.data, out.size, ctypes.byref(n))
    assert st == snappy.native.E_CAPACITY and n.value > 16
    exact = np.zeros(n.value, dtype=np.uint8)
    m = ctypes.c_uint64(0)
    assert L.snappy_lz4net_encode(data.ctypes.data, data.size, 0, exact.ctypes.data, exact.size,
                                  ctypes.byref(m)) == 0 and m.value == n.value
    assert L.snappy_lz4net_encode(None, 5, 0, out.ctypes.data, out.size, ctypes.byref(n)) == snappy.native.E_INVALID_ARG
    assert L.snappy_lz4net_encode(data.ctypes.data, data.size, (1 << 30) + 1, out.ctypes.data, out.size,
                                  ctypes.byref(n)) == snappy.native.E_INVALID_ARG


def test_written_archive_imports(snappy, oracle, tmp_path):
    """write_mcdf -> the import path: header fields, digests and the snapshot folder."""
    rng = np.random.default_rng(17)
    bodies = [H.asset_like(int(n), 300 + i) for i, n in enumerate(rng.integers(1, 600_000, size=23))]
    bodies[5] = b""                                   # Length 0: skipped on import
    files = [[f"chara/human/c0101/obj/f{i}.tex", f"CHARA/alt/F{i}.tex"][: 1 + i % 2] for i in range(len(bodies))]
    stream = snappy.write_mcdf(files, bodies, description='Written "here"', glamourer_data="Z2xhbQ==",
                               manipulation_data="bWV0YQ==")
    got = snappy.mcdf_decode_and_hash(stream, want_payload=True)
    assert list(got["statuses"][: len(bodies)]) == [1 if not b else 0 for b in bodies]
    for i, b in enumerate(bodies):
        if b:
            assert got["digests"][i].tobytes() == hashlib.sha1(b).digest(), i
    _, hdr, _ = oracle.mcdf_parse_header(bytes(got["payload"]))
    assert hdr["Description"] == 'Written "here"' and hdr["GlamourerData"] == "Z2xhbQ=="
    assert hdr["Files"] == [(paths, len(b)) for paths, b in zip(files, bodies)]
    path = tmp_path / "made_here.mcdf"
    path.write_bytes(stream)
    snap = snappy.import_mcdf(str(path), str(tmp_path / "work"))
    stored = {p.name for p in (tmp_path / "work").rglob("*.dat")}
    assert stored == {hashlib.sha1(b).hexdigest().upper() + ".dat" for b in bodies if b}
    assert snap


@pytest.mark.parametrize("block", [1, 7, 12, 13, 64, 1000])
def test_tiny_block_sizes(snappy, oracle, block):
    data = H.compressible(5000, block) + b"\0" * 300 + H.random_bytes(77, block)
    stream = snappy.lz4net_encode(data, block)
    assert oracle.lz4net_decode(stream) == (0, data)
    assert snappy.lz4net_decode(stream) == data
    assert [n for _, n, _ in _chunks(stream)] == [min(block, len(data) - k) for k in range(0, len(data), block)]
