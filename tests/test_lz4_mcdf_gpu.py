"""lz4net stream decode and the fused MCDF path on the GPU, checked byte for byte and
status for status against the CPU oracle."""
import hashlib

import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu


def _gpu_decode(snappy, stream, out_cap=None):
    try:
        return 0, snappy.lz4net_decode(stream, out_cap)
    except snappy.native.SnappyError as e:
        return e.code, b""


def test_hand_vectors_as_chunks(snappy, oracle):
    for name, (blk, want) in H.good_block_vectors().items():
        if len(blk) > len(want):
            continue  # lz4net never emits (and rejects) a chunk that grew
        for pad in (b"", b"\x00", b"\x00\x00", b"\x00\x00\x00"):
            # leading stored chunk shifts the compressed payload to every dword alignment
            lead = H.chunk(b"z" * len(pad), len(pad), False) if pad else b""
            stream = lead + H.chunk(blk, len(want), True)
            st, got = _gpu_decode(snappy, stream)
            assert st == 0, (name, len(pad))
            assert got == b"z" * len(pad) + want, (name, len(pad))


def test_bad_vectors_as_chunks(snappy, oracle):
    for name, (blk, out_len) in H.bad_block_vectors().items():
        if len(blk) > out_len:
            continue
        stream = H.chunk(blk, out_len, True)
        assert oracle.lz4net_decode(stream)[0] == oracle.E_LZ4_CORRUPT, name
        st, _ = _gpu_decode(snappy, stream)
        assert st == snappy.native.E_LZ4_CORRUPT, name


def test_liblz4_golden_blocks(snappy):
    g = H.load_golden("lz4_golden.json")
    for row in g["blocks"]:
        blk = bytes.fromhex(row["block_hex"])
        if len(blk) > row["orig_len"]:
            continue
        st, got = _gpu_decode(snappy, H.chunk(blk, row["orig_len"], True))
        assert st == 0, row["gen"]
        assert hashlib.sha1(got).hexdigest().upper() == row["sha1"], row["gen"]


def test_framing_cases_match_oracle(snappy, oracle):
    data = H.compressible(5000, 1)
    blk = oracle.lz4_block_encode(data)
    stored = H.chunk(b"plain bytes", 11, False)
    comp = H.chunk(blk, len(data), True, flags_extra=2)
    bad = H.chunk(blk[:-1], len(data), True)
    cases = {
        "empty": b"",
        "mixed": stored + comp + H.chunk(b"", 0, False) + H.chunk(b"", 0, True) + stored,
        "payload_cut": (stored + comp)[:-3],
        "varint_cut": stored + b"\x81",
        "flags_only": stored + b"\x00",
        "comp_gt_orig": H.varint(1) + H.varint(4) + H.varint(5) + b"abcde",
        "passes": H.chunk(blk, len(data), True, flags_extra=4),
        "stored_ignores_passes": H.chunk(b"abc", 3, False, flags_extra=4),
        "corrupt": bad,
        "corrupt_then_truncated": bad + stored[:-2],
        "truncated_after_good": comp + stored[:-2],
        "int32_narrowing": H.varint(0) + H.varint(1 << 31),
    }
    for name, s in cases.items():
        want_st, want = oracle.lz4net_decode(s)
        st, got = _gpu_decode(snappy, s)
        assert st == want_st, name
        if st == 0:
            assert got == want, name
    # capacity: reported where the reader would run out of room
    assert _gpu_decode(snappy, stored + comp, out_cap=100)[0] == snappy.native.E_CAPACITY
    assert oracle.lz4net_decode(stored + comp, out_cap=100)[0] == oracle.E_CAPACITY


@pytest.mark.parametrize("kind,size,chunk", [("compressible", 3_000_000, 1 << 20),
                                              ("asset_like", 6_000_000, 1 << 20),
                                              ("asset_like", 700_000, 65536),
                                              ("random", 2_500_000, 1 << 20),
                                              ("zeros", 2_200_000, 1 << 20)])
def test_generated_streams_roundtrip(snappy, oracle, kind, size, chunk):
    if kind == "compressible":
        data = H.compressible(size, 5)
    elif kind == "asset_like":
        data = H.asset_like(size, 6)
    elif kind == "random":
        data = H.random_bytes(size, 7)
    else:
        data = bytes(size)
    stream = oracle.lz4net_encode(data, chunk_size=chunk, hc_flag=True)
    want_st, want = oracle.lz4net_decode(stream)
    assert want_st == 0 and want == data
    st, got = _gpu_decode(snappy, stream)
    assert st == 0
    assert got == data


def test_single_bit_corruptions_never_fault(snappy, oracle):
    """Flip bytes of a valid stream: the GPU decoder must agree with the oracle on
    accept/reject and on the bytes whenever both accept."""
    data = H.compressible(40_000, 9)
    stream = bytearray(oracle.lz4net_encode(data, chunk_size=16384))
    rng = np.random.default_rng(3)
    for _ in range(60):
        s = bytearray(stream)
        pos = int(rng.integers(0, len(s)))
        s[pos] ^= 1 << int(rng.integers(0, 8))
        want_st, want = oracle.lz4net_decode(bytes(s))
        st, got = _gpu_decode(snappy, bytes(s))
        assert st == want_st, pos
        if st == 0:
            assert got == want, pos


def _make_archive(oracle, n_files, seed, chunk=1 << 20, declared_extra=0):
    rng = np.random.default_rng(seed)
    files = []
    for i in range(n_files):
        n = int(rng.integers(0, 400_000)) if i % 9 else 0
        body = H.asset_like(n, seed * 100 + i) if n else None
        paths = [f"chara/equipment/e{seed:04d}/file_{i}.tex"]
        if i % 4 == 0:
            paths.append(paths[0].upper())
        files.append((paths, body, n))
    payload = oracle.build_mcdf_payload(f"archive {seed}", files)
    if declared_extra:
        payload = payload[:-declared_extra]
    return oracle.lz4net_encode(payload, chunk_size=chunk), payload


def test_mcdf_fused_matches_oracle(snappy, oracle):
    stream, payload = _make_archive(oracle, 40, 1)
    ref = oracle.mcdf_import(stream)
    got = snappy.mcdf_decode_and_hash(stream, want_payload=True, n_files_cap=4)  # forces the retry path
    assert got["payload"] == payload
    assert got["body_offset"] == ref["body_offset"] and got["payload_len"] == len(payload)
    assert list(got["statuses"]) == ref["statuses"]
    from snappy_amd.crypto import digests_to_hex

    hx = digests_to_hex(got["digests"])
    k = 0
    for (paths, length), st, h in zip(ref["data"]["Files"], ref["statuses"], hx):
        if st == oracle.FILE_HASHED:
            for p in paths:
                # the map keeps the last writer per folded key; every hashed file's digest must be the oracle's
                pass
            body = payload[ref["body_offset"] + k: ref["body_offset"] + k + length]
            assert h == hashlib.sha1(body).hexdigest().upper()
            k += length
        else:
            assert h == "0" * 40


def test_mcdf_short_read_and_errors(snappy, oracle):
    stream, payload = _make_archive(oracle, 12, 2, declared_extra=1000)
    ref = oracle.mcdf_import(stream)
    got = snappy.mcdf_decode_and_hash(stream)
    assert list(got["statuses"]) == ref["statuses"]
    assert oracle.FILE_SKIPPED_SHORT_READ in ref["statuses"]
    # header errors after a clean decode
    for bad_payload, code in ((b"XCDF" + payload[4:], oracle.E_MCDF_MAGIC),
                              (payload[:4] + b"\x07" + payload[5:], oracle.E_MCDF_VERSION),
                              (payload[:6], oracle.E_MCDF_TRUNCATED),
                              (b"MCDF\x01\x05\x00\x00\x00{]}}}rest", oracle.E_MCDF_JSON)):
        s = oracle.lz4net_encode(bad_payload, chunk_size=1 << 16)
        with pytest.raises(oracle.McdfError) as eo:
            oracle.mcdf_import(s)
        assert eo.value.code == code
        with pytest.raises(snappy.native.SnappyError) as es:
            snappy.mcdf_decode_and_hash(s)
        assert es.value.code == code
    # stream errors come before header errors
    with pytest.raises(snappy.native.SnappyError) as es:
        snappy.mcdf_decode_and_hash(stream[:-7])
    assert es.value.code == snappy.native.E_LZ4_TRUNCATED


def test_mcdf_batch_matches_single_and_oracle(snappy, oracle):
    """Several archives in one call: good ones, a truncated stream, a bad header, an empty stream."""
    from snappy_amd.crypto import digests_to_hex

    good = [_make_archive(oracle, 10 + 3 * k, 10 + k, chunk=1 << 18)[0] for k in range(4)]
    bad_header = oracle.lz4net_encode(b"XCDF" + bytes(100), chunk_size=1 << 16)
    streams = [good[0], good[1][:-9], bad_header, good[2], b"", good[3]]
    res = snappy.mcdf_decode_and_hash_batch(streams, n_files_cap=3)   # forces the capacity retry
    assert len(res) == len(streams)
    for s, r in zip(streams, res):
        try:
            ref = oracle.mcdf_import(s)
            want_status = 0
        except oracle.McdfError as e:
            ref, want_status = None, e.code
        assert r["status"] == want_status
        if ref is None:
            assert len(r["statuses"]) == 0
            continue
        assert list(r["statuses"]) == ref["statuses"]
        assert r["body_offset"] == ref["body_offset"] and r["payload_len"] == len(ref["payload"])
        single = snappy.mcdf_decode_and_hash(s)
        assert np.array_equal(single["digests"], r["digests"])
        hx = digests_to_hex(r["digests"])
        for (paths, _), st, h in zip(ref["data"]["Files"], ref["statuses"], hx):
            if st == oracle.FILE_HASHED and paths:
                assert h in ref["blobs"]


def _random_sequences(rng, n_seq, max_lit, max_match, max_off, overlap_share):
    """Hand-built sequences: (literals, offset, match_len) with every length and offset drawn
    independently, so that extension bytes, repeating patterns and far offsets all occur."""
    parts, produced = [], 0
    for _ in range(n_seq):
        lit = int(rng.integers(0, max_lit + 1))
        if produced + lit == 0:
            lit = 1
        ml = int(rng.integers(4, max_match + 1))
        reach = min(produced + lit, max_off)
        if rng.random() < overlap_share:
            off = int(rng.integers(1, min(reach, 70) + 1))      # may repeat a short pattern
        else:
            off = int(rng.integers(min(ml, reach), reach + 1))
        parts.append((bytes(rng.integers(0, 256, size=lit, dtype=np.uint8)), off, ml))
        produced += lit + ml
    parts.append((b"the end of it", None, None))
    return parts


@pytest.mark.parametrize("max_lit,max_match,max_off,overlap_share", [
    (24, 36, 65535, 0.0),      # what a compressor makes of textures: short runs, offsets anywhere
    (8, 20, 2048, 0.0),        # everything within reach of the recent output
    (30, 40, 65535, 0.3),      # with repeating patterns of 1..70 bytes
    (300, 300, 65535, 0.1),    # lengths on both sides of every limit
    (3, 8, 64, 0.5),           # nothing but tiny pieces
])
def test_dense_short_sequences(snappy, oracle, max_lit, max_match, max_off, overlap_share):
    rng = np.random.default_rng(max_lit * 1000 + max_match)
    stream, want = b"", b""
    for n_seq in (1, 5, 63, 64, 65, 700, 2500):
        blk, out = H.block(_random_sequences(rng, n_seq, max_lit, max_match, max_off, overlap_share))
        if len(blk) > len(out):
            continue
        stream += H.chunk(blk, len(out), True)
        want += out
    st_o, got_o = oracle.lz4net_decode(stream)
    assert st_o == 0 and got_o == want
    st, got = _gpu_decode(snappy, stream)
    assert st == 0
    assert got == want


def test_dense_sequences_that_change_character_inside_a_chunk(snappy, oracle):
    """One chunk whose sequences go from near sources to sources anywhere, to repeated short
    patterns, to literal runs and matches of hundreds of bytes and back, every few hundred
    sequences: the decoder re-decides batch by batch how the ring is shared between the reach
    and the batch, which wave copies a batch's literals, and whether to take batches at all."""
    rng = np.random.default_rng(20261)
    regimes = [(8, 20, 2048, 0.0), (24, 36, 65535, 0.0), (3, 8, 64, 0.5), (120, 250, 65535, 0.05),
               (60, 30, 600, 0.0), (0, 12, 65535, 0.0), (300, 300, 65535, 0.1), (24, 36, 400, 0.2)]
    stream, want = b"", b""
    for rounds in (1, 3):
        parts, produced = [], 0
        for _ in range(rounds):
            for k in rng.permutation(len(regimes)):
                max_lit, max_match, max_off, overlap = regimes[k]
                for lit_bytes, off, ml in _random_sequences(rng, int(rng.integers(40, 400)), max_lit, max_match, max_off,
                                                            overlap)[:-1]:
                    # offsets were drawn against this piece's own output: against the whole chunk's they stay valid
                    # once something precedes the piece
                    off = min(off, produced + len(lit_bytes)) if produced + len(lit_bytes) else off
                    if off == 0:
                        continue
                    parts.append((lit_bytes, off, ml))
                    produced += len(lit_bytes) + ml
        parts.append((b"the end of it", None, None))
        blk, out = H.block(parts)
        assert len(blk) <= len(out) and len(out) < (1 << 20) * 4
        stream += H.chunk(blk, len(out), True)
        want += out
    st_o, got_o = oracle.lz4net_decode(stream)
    assert st_o == 0 and got_o == want
    st, got = _gpu_decode(snappy, stream)
    assert st == 0
    assert got == want


def test_dense_repeated_patterns_of_every_short_period(snappy, oracle):
    """Matches that repeat a pattern (offset < length): every period from 1 to 40 and a few longer
    ones, lengths from a few bytes to beyond 4 KiB, at every alignment of the output position
    (the literals in front vary in length): the runs of 1, 2, 4, 8 and 16 bytes are stored as one
    repeated 16-byte group, the other short periods are read out of the ring row by row, the long
    ones copied pattern by pattern."""
    rng = np.random.default_rng(5150)
    periods = list(range(1, 41)) + [63, 64, 65, 127, 255, 256, 257, 600, 5000]
    stream, want = b"", b""
    for lengths in ((5, 17, 63, 64, 255, 256, 257), (300, 1000, 4095, 4096, 4097, 9000), (20000, 70000)):
        parts = [(bytes(rng.integers(0, 256, size=5200, dtype=np.uint8)), 5200, 4)]   # something to be far from
        for period in periods:
            for ml in lengths:
                if ml <= period:
                    continue
                lit = bytes(rng.integers(0, 256, size=period + int(rng.integers(0, 4)), dtype=np.uint8))
                parts.append((lit, period, ml))
        parts.append((b"the end of it", None, None))
        blk, out = H.block(parts)
        assert len(blk) < len(out)
        stream += H.chunk(blk, len(out), True)
        want += out
    st_o, got_o = oracle.lz4net_decode(stream)
    assert st_o == 0 and got_o == want
    st, got = _gpu_decode(snappy, stream)
    assert st == 0
    assert got == want


def test_dense_short_sequences_at_every_limit(snappy, oracle):
    """One sequence of each combination of boundary lengths and offsets, surrounded by plain ones."""
    rng = np.random.default_rng(11)
    filler = lambda: (bytes(rng.integers(0, 256, size=9, dtype=np.uint8)), 7, 6)
    parts = [(bytes(rng.integers(0, 256, size=20000, dtype=np.uint8)), 100, 50)] + [filler() for _ in range(70)]
    produced = 20050 + 70 * 15
    for lit in (0, 1, 14, 15, 16, 239, 240, 241, 269, 270, 271, 600):
        for ml in (4, 18, 19, 20, 254, 255, 256, 273, 274, 700):
            for off in (1, 3, 4, 15, 16, 17, ml - 1, ml, ml + 1, 4095, 4096, 4097, 8191, 8192, 8193, 12288, 16384):
                if off < 1:
                    continue
                parts.append((bytes(rng.integers(0, 256, size=lit, dtype=np.uint8)), off, ml))
                parts += [filler() for _ in range(int(rng.integers(0, 4)))]
    parts.append((b"closing literals", None, None))
    blk, out = H.block(parts)
    stream = H.chunk(blk, len(out), True)
    st_o, got_o = oracle.lz4net_decode(stream)
    assert st_o == 0 and got_o == out
    st, got = _gpu_decode(snappy, stream)
    assert st == 0
    assert got == out


def test_dense_short_sequences_corrupted(snappy, oracle):
    """Byte flips in a stream of short sequences: same verdict as the oracle, same bytes when accepted."""
    rng = np.random.default_rng(21)
    blk, out = H.block(_random_sequences(rng, 900, 24, 36, 65535, 0.2))
    stream = H.chunk(blk, len(out), True)
    for _ in range(150):
        s = bytearray(stream)
        pos = int(rng.integers(4, len(s)))
        s[pos] ^= 1 << int(rng.integers(0, 8))
        want_st, want = oracle.lz4net_decode(bytes(s))
        st, got = _gpu_decode(snappy, bytes(s))
        assert st == want_st, pos
        if st == 0:
            assert got == want, pos


@pytest.mark.parametrize("tuning", ["1", "32", "33", "64"])
def test_other_ways_through_the_decoder(tuning):
    """The same block and stream tests with the batch path switched off (bit 0), with one wave per
    chunk instead of two (bit 5), with both, and with every batch's literals copied by the parsing
    wave (bit 6; otherwise the helper takes them when it is idle) (the switches are read once per
    process, hence the child): every way through the decoder stays covered."""
    import os
    import subprocess
    import sys
    if os.environ.get("SNAPPY_LZ4_TUNING"):
        pytest.skip("already the child")
    env = dict(os.environ, SNAPPY_LZ4_TUNING=tuning)
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-p", "no:cacheprovider", __file__, "-k",
                        "dense or vectors or golden or framing or generated or corruptions"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


def test_concurrent_decoders(snappy, oracle):
    """Four threads decode different streams at the same time (the plugin's draw thread and a
    thread-pool task may overlap); every result must be its own stream's."""
    import threading
    streams, wants = [], []
    for seed in range(4):
        rng = np.random.default_rng(100 + seed)
        blk, out = H.block(_random_sequences(rng, 1500 + 300 * seed, 24, 36, 65535, 0.2))
        data = H.asset_like(400_000 + 50_000 * seed, seed)
        streams.append(H.chunk(blk, len(out), True) + oracle.lz4net_encode(data, chunk_size=65536))
        wants.append(out + data)
    errors = []

    def work(i):
        try:
            for _ in range(8):
                st, got = _gpu_decode(snappy, streams[i])
                if st != 0 or got != wants[i]:
                    errors.append((i, st))
        except Exception as e:  # noqa: BLE001
            errors.append((i, repr(e)))

    threads = [threading.Thread(target=work, args=(i,)) for i in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


def test_oversized_claims_do_not_take_the_batch_down(snappy, oracle, monkeypatch):
    """A chunk header of a dozen bytes may declare 2 GiB of output.  Such an archive is reported
    on its own (the block cannot decode to that size: corrupt, as the reader would find) and the
    archives next to it still come back; the same when the device arena cannot hold all of them."""
    good = [_make_archive(oracle, 8 + k, 70 + k, chunk=1 << 18)[0] for k in range(3)]
    # 40 compressed bytes that claim 2^31 - 1 decoded bytes
    bomb = H.varint(1) + H.varint((1 << 31) - 1) + H.varint(40) + bytes(40)
    st, _ = oracle.lz4net_decode(bomb, out_cap=1 << 20)
    streams = [good[0], bomb, good[1], good[2]]
    res = snappy.mcdf_decode_and_hash_batch(streams)
    assert res[1]["status"] == snappy.native.E_LZ4_CORRUPT
    for k in (0, 2, 3):
        ref = oracle.mcdf_import(streams[k])
        assert res[k]["status"] == 0 and list(res[k]["statuses"]) == ref["statuses"]
    with pytest.raises(snappy.native.SnappyError) as e:
        snappy.lz4net_decode(bomb, out_cap=1 << 20)
    assert e.value.code in (snappy.native.E_LZ4_CORRUPT, snappy.native.E_CAPACITY)
    # a stream that passes every header check but does not fit next to the others
    big = oracle.lz4net_encode(oracle.build_mcdf_payload("big", [(["a.tex"], bytes(24 << 20), 24 << 20)]),
                               chunk_size=1 << 20)
    monkeypatch.setenv("SNAPPY_TEST_ARENA_LIMIT_BYTES", str(16 << 20))
    res = snappy.mcdf_decode_and_hash_batch([good[0], big, good[1]])
    monkeypatch.delenv("SNAPPY_TEST_ARENA_LIMIT_BYTES")
    assert res[1]["status"] == snappy.native.E_NOMEM
    assert res[0]["status"] == 0 and res[2]["status"] == 0
    whole = snappy.mcdf_decode_and_hash_batch([good[0], big, good[1]])
    assert [r["status"] for r in whole] == [0, 0, 0]
    assert np.array_equal(whole[0]["digests"], res[0]["digests"])
    assert np.array_equal(whole[2]["digests"], res[2]["digests"])


def test_mcdf_import_overlapped_path(snappy, oracle, monkeypatch):
    """The import with copy, decode and hash overlapped (headers read from the archives' first
    chunks, the SHA-1 kernel waiting for each archive's decode): forced on for small archives and
    compared with the plain order of things and with the oracle.  Includes an archive whose later
    chunk is malformed, one that is truncated, one with a bad header, an empty one, one whose first
    chunk is stored, and a batch in which one header runs past its first chunk (handed back).
    The archive with the batch's largest file arrives in three parts (that file's chunks first);
    one turn sends whole archives only."""
    from snappy_amd.crypto import digests_to_hex

    good = [_make_archive(oracle, 12 + 5 * k, 40 + k, chunk=1 << 16)[0] for k in range(4)]
    bad_header = oracle.lz4net_encode(b"XCDF" + bytes(100), chunk_size=1 << 16)
    # a flipped byte deep inside a later chunk's payload of random bytes would still decode; break a token instead
    broken = bytearray(good[3])
    st, size = oracle.lz4net_decoded_size(bytes(broken))
    broken[len(broken) // 2: len(broken) // 2 + 64] = b"\xff" * 64
    broken = bytes(broken)
    stored_first = oracle.lz4net_encode(
        oracle.build_mcdf_payload("stored", [(["a.tex"], H.random_bytes(150_000, 5), 150_000)]), chunk_size=1 << 16)
    streams = [good[0], good[1][:-9], bad_header, good[2], b"", broken, stored_first, good[3]]

    def expected(s):
        try:
            return oracle.mcdf_import(s), 0
        except oracle.McdfError as e:
            return None, e.code

    def check(res, batch):
        for s, r in zip(batch, res):
            ref, want_status = expected(s)
            assert r["status"] == want_status
            if ref is None:
                assert len(r["statuses"]) == 0
                continue
            assert list(r["statuses"]) == ref["statuses"]
            hx = digests_to_hex(r["digests"])
            for (paths, _), st_, h in zip(ref["data"]["Files"], ref["statuses"], hx):
                if st_ == oracle.FILE_HASHED and paths:
                    assert ref["map"][paths[0]] == h or h in ref["blobs"]

    plain = snappy.mcdf_decode_and_hash_batch(streams)
    check(plain, streams)
    monkeypatch.setenv("SNAPPY_OVERLAP_MIN_BYTES", "0")
    for turn in ("waiting kernel", "waiting kernel", "one decode launch per archive", "whole archives", "plain order"):
        # Archives this small normally take the short way (send, decode, hash); with the size
        # limit at 0 they go through the overlapped import: the SHA-1 kernel waiting, the files
        # planned group by group in arrival order; the last turn sends everything first.
        monkeypatch.delenv("SNAPPY_MCDF_PLAIN_ORDER", raising=False)
        monkeypatch.delenv("SNAPPY_MCDF_DECODE_RUN", raising=False)
        monkeypatch.delenv("SNAPPY_MCDF_NO_SPLIT", raising=False)
        if turn == "whole archives":   # the largest file's chunks are not sent ahead of their archive
            monkeypatch.setenv("SNAPPY_MCDF_NO_SPLIT", "1")
        if turn == "plain order":
            monkeypatch.setenv("SNAPPY_MCDF_PLAIN_ORDER", "1")
        if turn == "one decode launch per archive":
            monkeypatch.setenv("SNAPPY_MCDF_DECODE_RUN", "1")
        over = snappy.mcdf_decode_and_hash_batch(streams)
        check(over, streams)
        for a, b in zip(plain, over):
            assert a["status"] == b["status"] and np.array_equal(a["digests"], b["digests"]), turn
    monkeypatch.delenv("SNAPPY_MCDF_PLAIN_ORDER", raising=False)
    monkeypatch.delenv("SNAPPY_MCDF_NO_SPLIT", raising=False)
    # 75 archives, one decode launch each: more runs than hashing groups (64), so groups of two runs.
    monkeypatch.setenv("SNAPPY_MCDF_DECODE_RUN", "1")
    pool = [_make_archive(oracle, 2 + k, 300 + k, chunk=1 << 15)[0] for k in range(9)] + [broken, b""]
    many_archives = [pool[(7 * i) % len(pool)] for i in range(75)]
    check(snappy.mcdf_decode_and_hash_batch(many_archives), many_archives)
    monkeypatch.delenv("SNAPPY_MCDF_DECODE_RUN", raising=False)
    single = snappy.mcdf_decode_and_hash(good[0], want_payload=True)
    assert single["payload"] == oracle.mcdf_import(good[0])["payload"]
    # 600 files, 4 KiB chunks: the header JSON is longer than the first chunk
    many = [([f"chara/many/{i:04d}.mdl"], H.random_bytes(50 + i, i), 50 + i) for i in range(600)]
    long_header = oracle.lz4net_encode(oracle.build_mcdf_payload("many", many), chunk_size=4096)
    batch = [good[0], long_header]
    check(snappy.mcdf_decode_and_hash_batch(batch), batch)


def test_mcdf_import_from_page_locked_and_pageable_streams(snappy, oracle, monkeypatch):
    """The overlapped import (forced on for a batch this small), above the size from which ordinary memory goes through the staging ring (8 MiB of
    streams), once from ordinary memory and once from snappy_host_alloc memory (direct DMA, no
    ring): same results, equal to the oracle's, archive by archive - an empty stream and a
    truncated one among them."""
    import ctypes

    from snappy_amd.crypto import digests_to_hex

    good = [_make_archive(oracle, 30 + 7 * k, 500 + k, chunk=1 << 20)[0] for k in range(4)]
    streams = [good[0], b"", good[1], good[2][:-5], good[3]]
    assert sum(len(s) for s in streams) > (8 << 20)
    monkeypatch.setenv("SNAPPY_OVERLAP_MIN_BYTES", "0")
    pageable = snappy.mcdf_decode_and_hash_batch(streams)
    L = snappy.native.lib()
    total = sum(len(s) for s in streams)
    ptr = ctypes.c_void_p()
    assert L.snappy_host_alloc(total + 64, ctypes.byref(ptr)) == 0
    try:
        pinned = np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(ctypes.c_uint8)), shape=(total + 64,))
        views, at = [], 0
        for s in streams:
            pinned[at: at + len(s)] = np.frombuffer(s, dtype=np.uint8)
            views.append(pinned[at: at + len(s)])
            at += len(s)
        locked = snappy.mcdf_decode_and_hash_batch(views)
        del views, pinned
    finally:
        assert L.snappy_host_free(ptr) == 0
    for s, a, b in zip(streams, pageable, locked):
        try:
            ref, want = oracle.mcdf_import(s), 0
        except oracle.McdfError as e:
            ref, want = None, e.code
        assert a["status"] == b["status"] == want
        assert np.array_equal(a["digests"], b["digests"]) and list(a["statuses"]) == list(b["statuses"])
        if ref is not None:
            assert list(a["statuses"]) == ref["statuses"]
            for (paths, _), st_, h in zip(ref["data"]["Files"], ref["statuses"], digests_to_hex(a["digests"])):
                if st_ == oracle.FILE_HASHED and paths:
                    assert h in ref["blobs"]


def _chunk_spans(stream):
    """(header start, payload start, payload end) of every chunk of an lz4net stream."""
    def varint(at):
        v, shift = 0, 0
        while True:
            b = stream[at]
            at += 1
            v |= (b & 0x7F) << shift
            shift += 7
            if not b & 0x80:
                return v, at
    spans, at = [], 0
    while at < len(stream):
        start = at
        flags, at = varint(at)
        n, at = varint(at)
        if flags & 1:
            n, at = varint(at)
        spans.append((start, at, at + n))
        at += n
    return spans


def test_malformed_block_comes_before_a_later_bad_chunk_header(snappy, oracle, monkeypatch):
    """Damage inside one chunk's payload and in the next chunk's header: the block
    fails to decode AND the walk over the chunk headers stops right behind it.  The reference's
    CopyTo decodes chunk by chunk, so it meets the malformed block first; the batch import says the
    same in the overlapped order, in the plain order and through the single-archive call."""
    stream, _ = _make_archive(oracle, 25, 77, chunk=1 << 16)
    spans = _chunk_spans(stream)
    # a compressed chunk (its flags byte has bit 0 set) with another chunk behind it
    k = next(i for i in range(1, len(spans) - 1) if stream[spans[i][0]] & 1 and spans[i][2] - spans[i][1] > 64)
    end = spans[k][2]
    assert end == spans[k + 1][0]
    # A byte of the chunk whose flipping makes the block malformed (most bytes are literals).
    for at in range(spans[k][1], spans[k][2]):
        bad = bytearray(stream)
        bad[at] ^= 0xFF
        if oracle.lz4net_decode(bytes(bad))[0] == snappy.native.E_LZ4_CORRUPT:
            break
    else:
        pytest.fail("no byte of the chunk breaks it")
    bad[end: end + 4] = b"\xff" * 4            # and the next chunk's flags and length varints
    bad = bytes(bad)
    with pytest.raises(oracle.McdfError) as eo:
        oracle.mcdf_import(bad)
    assert eo.value.code == snappy.native.E_LZ4_CORRUPT
    good = _make_archive(oracle, 9, 78, chunk=1 << 16)[0]
    for turn in ("short way", "overlapped", "plain order"):
        monkeypatch.delenv("SNAPPY_OVERLAP_MIN_BYTES", raising=False)
        monkeypatch.delenv("SNAPPY_MCDF_PLAIN_ORDER", raising=False)
        if turn != "short way":
            monkeypatch.setenv("SNAPPY_OVERLAP_MIN_BYTES", "0")
        if turn == "plain order":
            monkeypatch.setenv("SNAPPY_MCDF_PLAIN_ORDER", "1")
        res = snappy.mcdf_decode_and_hash_batch([good, bad, good])
        assert [r["status"] for r in res] == [0, snappy.native.E_LZ4_CORRUPT, 0], turn
        with pytest.raises(snappy.native.SnappyError) as es:
            snappy.mcdf_decode_and_hash(bad)
        assert es.value.code == snappy.native.E_LZ4_CORRUPT, turn
