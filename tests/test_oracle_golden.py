"""Pins the CPU oracle: FIPS 180-4 answers, hashlib-generated digests, hand-assembled
LZ4 blocks, liblz4-produced blocks, lz4net framing cases.  No GPU, no product code."""
import hashlib

import numpy as np
import pytest

import helpers as H


def test_sha1_fips_vectors(oracle):
    g = H.load_golden("sha1_golden.json")
    for e in g["fips"]:
        data = e["ascii"].encode() if "ascii" in e else e["repeat"][0].encode() * e["repeat"][1]
        assert oracle.get_hash(data) == e["sha1"]
    for n, want in g["zeros"].items():
        assert oracle.get_hash(b"\0" * int(n)) == want


def test_sha1_golden_splitmix(oracle):
    g = H.load_golden("sha1_golden.json")
    for row in g["splitmix"]:
        data = H.splitmix_bytes(row["len"], H.file_seed(row["index"]))
        assert oracle.get_hash(data) == row["sha1"], row
        # the C generator used by the bench matches the pure-Python one
        assert oracle.fill_splitmix(min(row["len"], 4096), H.file_seed(row["index"])) == data[:4096]


def test_sha1_hex_is_uppercase_40(oracle):
    h = oracle.get_hash(b"snapshot")
    assert len(h) == 40 and h == h.upper() and h == hashlib.sha1(b"snapshot").hexdigest().upper()


def test_sha1_batch_matches_single(oracle):
    rng = np.random.default_rng(5)
    base = rng.integers(0, 256, size=50_000, dtype=np.uint8)
    lengths = rng.integers(0, 900, size=60).astype(np.uint64)
    offsets = rng.integers(0, 49_000, size=60).astype(np.uint64)
    d = oracle.sha1_batch_np(base, offsets, lengths)
    for i in range(60):
        want = hashlib.sha1(base[int(offsets[i]): int(offsets[i] + lengths[i])].tobytes()).digest()
        assert d[i].tobytes() == want


def test_lz4_hand_vectors_decode(oracle):
    for name, (blk, want) in H.good_block_vectors().items():
        st, got = oracle.lz4_block_decode(blk, len(want))
        assert st == oracle.OK, name
        assert got == want, name


def test_lz4_bad_vectors_rejected(oracle):
    for name, (blk, out_len) in H.bad_block_vectors().items():
        st, _ = oracle.lz4_block_decode(blk, out_len)
        assert st == oracle.E_LZ4_CORRUPT, name


def test_lz4_liblz4_golden_blocks(oracle):
    g = H.load_golden("lz4_golden.json")
    assert len(g["blocks"]) >= 20
    for row in g["blocks"]:
        st, got = oracle.lz4_block_decode(bytes.fromhex(row["block_hex"]), row["orig_len"])
        assert st == oracle.OK, row["gen"]
        assert hashlib.sha1(got).hexdigest().upper() == row["sha1"], row["gen"]
        kind, n, seed = row["gen"]
        assert got == H.compressible(n, seed)


def test_lz4_encoder_roundtrip(oracle):
    for n in (0, 1, 12, 13, 64, 5000, 70_000, 1_200_000):
        data = H.compressible(n, n + 1)
        blk = oracle.lz4_block_encode(data)
        st, got = oracle.lz4_block_decode(blk, n)
        if n == 0:
            continue
        assert st == oracle.OK and got == data, n


def test_lz4net_framing_cases(oracle):
    o = oracle
    data = H.compressible(5000, 1)
    blk = o.lz4_block_encode(data)
    stored = H.chunk(b"plain bytes", 11, False)
    comp = H.chunk(blk, len(data), True, flags_extra=2)
    empty = H.chunk(b"", 0, False)
    empty_comp = H.chunk(b"", 0, True)
    stream = stored + comp + empty + empty_comp + stored
    assert o.lz4net_decoded_size(stream) == (o.OK, 11 + 5000 + 11)
    st, out = o.lz4net_decode(stream)
    assert st == o.OK and out == b"plain bytes" + data + b"plain bytes"
    assert o.lz4net_decode(b"") == (o.OK, b"")
    # multi-byte varints: 5000 needs two bytes
    assert H.varint(5000) == bytes([0x88, 0x27])
    # errors
    assert o.lz4net_decode(stream[:-3])[0] == o.E_LZ4_TRUNCATED          # payload cut
    assert o.lz4net_decode(stored + b"\x81")[0] == o.E_LZ4_TRUNCATED      # varint cut
    assert o.lz4net_decode(stored + b"\x00")[0] == o.E_LZ4_TRUNCATED      # flags only
    assert o.lz4net_decode(H.varint(1) + H.varint(4) + H.varint(5) + b"abcde")[0] == o.E_LZ4_BAD_LENGTHS
    assert o.lz4net_decode(H.chunk(blk, len(data), True, flags_extra=4))[0] == o.E_LZ4_PASSES
    assert o.lz4net_decode(H.chunk(b"abc", 3, False, flags_extra=4)) == (o.OK, b"abc")  # stored ignores passes
    bad = H.chunk(blk[:-1], len(data), True)  # final literal run one byte short
    assert o.lz4net_decode(bad)[0] == o.E_LZ4_CORRUPT
    # errors surface in stream order: corrupt chunk 0 wins over truncated chunk 1
    assert o.lz4net_decode(bad + stored[:-2])[0] == o.E_LZ4_CORRUPT
    assert o.lz4net_decode(stored + comp, out_cap=100)[0] == o.E_CAPACITY
    # lengths are narrowed to int32 by the managed reader
    assert o.lz4net_decode(H.varint(0) + H.varint(1 << 31))[0] == o.E_LZ4_BAD_LENGTHS


def test_lz4net_encoder_chunks_and_roundtrip(oracle):
    data = H.asset_like(300_000, 3)
    stream = oracle.lz4net_encode(data, chunk_size=65536, hc_flag=True)
    st, n = oracle.lz4net_decoded_size(stream)
    assert (st, n) == (oracle.OK, len(data))
    st, out = oracle.lz4net_decode(stream)
    assert st == oracle.OK and out == data
    assert stream[0] in (2, 3)  # first chunk's flags: HighCompression bit set, Compressed or not


def test_mcdf_header_and_slicing(oracle):
    o = oracle
    bodies = [H.random_bytes(n, n) for n in (100, 0, 7, 3000)]
    files = [(["chara/a.tex", "CHARA/A.TEX"], bodies[0], 100), (["chara/empty.mdl"], None, 0),
             (["chara/b.mdl"], bodies[2], 7), (["chara/c.tex", "chara/a.tex"], bodies[3], 3000)]
    payload = o.build_mcdf_payload("desc", files)
    version, data, body = o.mcdf_parse_header(payload)
    assert version == 1 and data["Description"] == "desc" and len(data["Files"]) == 4
    mapping, statuses, blobs = o.mcdf_extract_and_hash(data["Files"], payload, body)
    assert statuses == [o.FILE_HASHED, o.FILE_SKIPPED_EMPTY, o.FILE_HASHED, o.FILE_HASHED]
    h = [hashlib.sha1(b).hexdigest().upper() for b in bodies]
    # OrdinalIgnoreCase: "CHARA/A.TEX" folds onto "chara/a.tex", last writer wins, first spelling kept
    assert mapping == {"chara/a.tex": h[3], "chara/b.mdl": h[2], "chara/c.tex": h[3]}
    assert set(blobs) == {h[0], h[2], h[3]}
    # short read: the reader consumes what is left, every later non-empty file comes up short
    cut = payload[: body + 100 + 3]
    mapping, statuses, _ = o.mcdf_extract_and_hash(data["Files"], cut, body)
    assert statuses == [o.FILE_HASHED, o.FILE_SKIPPED_EMPTY, o.FILE_SKIPPED_SHORT_READ,
                        o.FILE_SKIPPED_SHORT_READ]
    # header errors
    for bad, code in ((b"MCDX" + payload[4:], o.E_MCDF_MAGIC), (b"MC", o.E_MCDF_MAGIC),
                      (payload[:4] + b"\x02" + payload[5:], o.E_MCDF_VERSION),
                      (payload[:7], o.E_MCDF_TRUNCATED), (payload[: body - 5], o.E_MCDF_TRUNCATED),
                      (payload[:9] + b"{" * (body - 9) + payload[body:], o.E_MCDF_JSON)):
        with pytest.raises(o.McdfError) as e:
            o.mcdf_parse_header(bad)
        assert e.value.code == code


def test_key_folding_table_matches_the_oracle():
    """The library folds snapshot keys (StringComparer.OrdinalIgnoreCase) with a generated table
    (snappy_amd/csrc/unicode_upper_table.hpp); the oracle folds with simple_upper() at run time.
    Every code point must agree, so the table cannot go stale against its generator."""
    import os
    import re

    from oracle import oracle

    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(here, "snappy_amd", "csrc", "unicode_upper_table.hpp")).read()
    table = {int(a, 16): int(b, 16) for a, b in re.findall(r"\{0x([0-9A-F]+), 0x([0-9A-F]+)\}", text)}
    assert len(table) > 1000 and sorted(table) == list(table)
    for cp in range(0x80, 0x110000):
        if 0xD800 <= cp <= 0xDFFF:
            continue
        want = ord(oracle.simple_upper(chr(cp)))
        assert table.get(cp, cp) == want, hex(cp)
    assert oracle._ordinal_upper("chara/é/straße/\U00010428/ı") == "CHARA/É/STRAßE/\U00010400/ı"
