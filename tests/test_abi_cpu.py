"""CPU-only checks of the C-ABI library: it loads, exports every declared symbol, the
host-only entry points behave, and compute calls fail loudly without a GPU."""
import ctypes
import os
import re

import numpy as np
import pytest

import helpers as H

HEADER = os.path.join(H.ROOT, "include", "snappyhash.h")


def _has_gpu():
    import torch

    return torch.cuda.is_available()


def test_library_exports_every_declared_symbol(snappy):
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    declared = set(re.findall(r"\b(snappy_[a-z0-9_]+)\s*\(", text))
    assert len(declared) >= 17
    lib = ctypes.CDLL(snappy.native.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    assert declared == {s[0] for s in snappy.native.SYMBOLS}


def test_strerror_and_hex_formatting(snappy):
    n = snappy.native
    assert n.strerror(0) == "ok"
    for code in (-1, -2, -3, -4, -10, -11, -12, -13, -14, -20, -21, -22, -23):
        assert n.strerror(code) not in ("ok", "unknown status")
    assert n.strerror(-999) == "unknown status"
    from snappy_amd.crypto import digests_to_hex

    d = np.arange(40, dtype=np.uint8).reshape(2, 20)
    d[1, 0] = 0xAB
    hx = digests_to_hex(d)
    assert hx[0] == "000102030405060708090A0B0C0D0E0F10111213"
    assert hx[1].startswith("AB15") and hx[1] == hx[1].upper() and len(hx[1]) == 40


def test_lz4net_decoded_size_matches_oracle(snappy, oracle):
    data = H.compressible(5000, 1)
    blk = oracle.lz4_block_encode(data)
    stored = H.chunk(b"plain bytes", 11, False)
    comp = H.chunk(blk, len(data), True, flags_extra=2)
    cases = [b"", stored, stored + comp + H.chunk(b"", 0, False), (stored + comp)[:-3],
             stored + b"\x81", H.varint(1) + H.varint(4) + H.varint(5) + b"abcde",
             H.chunk(blk, len(data), True, flags_extra=4), H.varint(0) + H.varint(1 << 31),
             oracle.lz4net_encode(H.asset_like(200_000, 4), 65536)]
    for s in cases:
        st, n = oracle.lz4net_decoded_size(s)
        if st == oracle.OK:
            assert snappy.lz4net_decoded_size(s) == n
        else:
            with pytest.raises(snappy.native.SnappyError) as e:
                snappy.lz4net_decoded_size(s)
            assert e.value.code == st


def test_mcdf_parse_header_matches_oracle(snappy, oracle):
    bodies = [H.random_bytes(n, n) for n in (100, 7, 3000)]
    files = [(["a/é.tex", "b\\\"q\".tex"], bodies[0], 100), (["e.mdl"], None, 0),
             (["b.mdl"], bodies[1], 7), ([], bodies[2], 3000)]
    payload = oracle.build_mcdf_payload("désc \U0001F600", files)
    _, data, body = oracle.mcdf_parse_header(payload)
    lengths, body2 = snappy.mcdf_parse_header(payload)
    assert body2 == body and list(lengths) == [f[1] for f in data["Files"]]
    variants = {
        "lower_case_names": b'{"files":[{"gamepaths":["x"],"length":5}]}',
        "length_as_string": b'{"Files":[{"GamePaths":["x"],"Length":"12"}]}',
        "no_files": b'{"Description":"x"}',
        "null_strings": b'{"Description":null,"Files":[]}',
        "missing_length": b'{"Files":[{"GamePaths":[]}]}',
        "big_length": b'{"Files":[{"GamePaths":[],"Length":4294967301}]}',
        "files_null": b'{"Files":null}',
        "paths_null": b'{"Files":[{"GamePaths":null,"Length":1}]}',
        "length_float": b'{"Files":[{"GamePaths":[],"Length":1.5}]}',
        "length_null": b'{"Files":[{"GamePaths":[],"Length":null}]}',
        "length_overflow": b'{"Files":[{"GamePaths":[],"Length":9223372036854775808}]}',
        "trailing_comma": b'{"Files":[],}',
        "not_object": b'[1,2]',
        "garbage": b'{"Files":[{]}',
        "escapes": b'{"Description":"a\\u00e9\\ud83d\\ude00\\n","Files":[{"GamePaths":["p\\\\q"],"Length":-0}]}',
    }
    for name, js in variants.items():
        p = b"MCDF\x01" + len(js).to_bytes(4, "little") + js + b"body"
        try:
            _, data, body = oracle.mcdf_parse_header(p)
            want = ("ok", [f[1] for f in data["Files"]], body)
        except oracle.McdfError as e:
            want = ("err", e.code)
        try:
            lengths, b2 = snappy.mcdf_parse_header(p)
            got = ("ok", list(lengths), b2)
        except snappy.native.SnappyError as e:
            got = ("err", e.code)
        assert got == want, name
    for bad in (b"MCDX" + payload[4:], b"MC", b"", payload[:4] + b"\x02" + payload[5:],
                payload[:7], payload[: body - 5]):
        with pytest.raises(oracle.McdfError) as eo:
            oracle.mcdf_parse_header(bad)
        with pytest.raises(snappy.native.SnappyError) as es:
            snappy.mcdf_parse_header(bad)
        assert es.value.code == eo.value.code


def test_compute_fails_loudly_without_gpu(snappy):
    if _has_gpu():
        pytest.skip("a GPU is present; the no-device behaviour is not reachable")
    n = snappy.native
    assert n.lib().snappy_init(0) == n.E_NO_DEVICE
    assert n.lib().snappy_device_count() == 0
    for call in (lambda: snappy.get_hash(b"abc"), lambda: snappy.sha1(b""),
                 lambda: snappy.sha1_batch(b"abcdef", [0, 2], [2, 4]),
                 lambda: snappy.sha1_files([b"a", b"bc"]),
                 lambda: snappy.lz4net_decode(H.chunk(b"abc", 3, False)),
                 lambda: snappy.mcdf_decode_and_hash(H.chunk(b"abc", 3, False))):
        with pytest.raises(n.SnappyError) as e:
            call()
        assert e.value.code == n.E_NO_DEVICE


def test_invalid_arguments(snappy):
    L = snappy.native.lib()
    out = (ctypes.c_uint8 * 20)()
    assert L.snappy_sha1(None, 5, out) == snappy.native.E_INVALID_ARG
    assert L.snappy_sha1(None, 0, None) == snappy.native.E_INVALID_ARG
    assert L.snappy_sha1_batch(None, None, None, 3, None) == snappy.native.E_INVALID_ARG
    assert L.snappy_lz4net_decoded_size(None, 4, None) == snappy.native.E_INVALID_ARG
    assert L.snappy_digests_to_hex(None, 0, None) == snappy.native.OK


def test_mcdf_write_header_is_compact_json_in_declaration_order(snappy, oracle):
    """McdfHeader.WriteToStream + JsonConvert.SerializeObject: host only, no GPU needed."""
    import json
    import struct

    files = [(["chara/a.tex", "chara/B.tex"], 12345), ([], 0), (['q"uote\\back\n.mdl'], 2 ** 40)]
    head = snappy.mcdf_write_header(files, description="d\u00e9sc \t tab", glamourer_data="g",
                                    customize_plus_data="", manipulation_data="m")
    assert head[:5] == b"MCDF\x01"
    (n,) = struct.unpack_from("<i", head, 5)
    assert len(head) == 9 + n
    text = head[9:].decode("utf-8")
    want = json.dumps({"Description": "d\u00e9sc \t tab", "GlamourerData": "g", "CustomizePlusData": "",
                       "ManipulationData": "m",
                       "Files": [{"GamePaths": p, "Length": ln} for p, ln in files]},
                      separators=(",", ":"), ensure_ascii=False)
    assert text == want
    # and the reader takes it back
    lengths, body = snappy.mcdf_parse_header(head + b"xyz")
    assert list(lengths) == [12345, 0, 2 ** 40] and body == len(head)
    version, parsed, body_offset = oracle.mcdf_parse_header(head)
    assert version == 1 and body_offset == len(head)
    assert parsed["Files"] == [(p, ln) for p, ln in files]
    empty = snappy.mcdf_write_header([])
    assert empty[9:] == b'{"Description":"","GlamourerData":"","CustomizePlusData":"","ManipulationData":"","Files":[]}'


def test_hostile_sizes_return_a_status(snappy):
    """Nothing may unwind through the C ABI: a header that would blow up the JSON reader and a
    file count no container can hold both come back as status codes."""
    import struct

    n = snappy.native
    js = b"[" + b"[]," * 3_000_000 + b"[]]"                 # ~9 MB of text, 3 million values
    payload = b"MCDF\x01" + struct.pack("<i", len(js)) + js
    with pytest.raises(n.SnappyError) as e:
        snappy.mcdf_parse_header(payload)
    assert e.value.code == n.E_MCDF_JSON
    deep = b"[" * 100_000
    with pytest.raises(n.SnappyError) as e:
        snappy.mcdf_parse_header(b"MCDF\x01" + struct.pack("<i", len(deep)) + deep)
    assert e.value.code == n.E_MCDF_JSON
    L = n.lib()
    one = np.zeros(1, dtype=np.uint64)
    out = np.zeros(20, dtype=np.uint8)
    for absurd in (1 << 62, (1 << 64) - 1):
        st = L.snappy_sha1_batch(out.ctypes.data, one.ctypes.data, one.ctypes.data, absurd, out.ctypes.data)
        assert st in (n.E_NOMEM, n.E_INVALID_ARG), st
        st = L.snappy_sha1_files(one.ctypes.data, one.ctypes.data, absurd, out.ctypes.data)
        assert st in (n.E_NOMEM, n.E_INVALID_ARG), st


def _lpt_restated(lengths, k):
    """snappy_plan_shards_lpt in plain Python."""
    blocks = [(int(v) + 8) // 64 + 1 for v in lengths]
    total = sum(blocks)
    average = -(-total // k)
    large_from = max(1, average // 64)
    load = [0] * k
    shard_of = [0] * len(blocks)
    large = sorted((i for i, b in enumerate(blocks) if b >= large_from), key=lambda i: -blocks[i])
    for i in large:
        d = load.index(min(load))
        shard_of[i] = d
        load[d] += blocks[i]
    d = 0
    for i, b in enumerate(blocks):
        if b >= large_from:
            continue
        while d + 1 < k and load[d] >= average:
            d += 1
        shard_of[i] = d
        load[d] += b
    return shard_of, load


def test_plan_shards_lpt_properties(snappy):
    from snappy_amd.crypto import plan_shards_lpt

    rng = np.random.default_rng(18)
    cases = [(0, 3), (1, 4), (5, 5), (37, 2), (1000, 8), (20000, 8)]
    for n, k in cases:
        if n == 20000:      # the small-file regime: nothing is "large", runs stay contiguous
            ln = rng.integers(4096, 32769, size=n).astype(np.uint64)
        else:
            ln = np.exp(rng.uniform(np.log(1), np.log(22 << 20), size=n)).astype(np.uint64)
        shard_of, blocks = plan_shards_lpt(ln, k)
        want_of, want_blocks = _lpt_restated(ln, k)
        assert list(shard_of) == want_of and list(blocks) == want_blocks
        per_file = (ln + 8) // 64 + 1
        assert int(blocks.sum()) == int(per_file.sum())
        if n >= 1000:
            # longest-first: no shard above the average by more than the largest file
            assert int(blocks.max()) <= int(per_file.sum()) // k + int(per_file.max())
        if n == 20000:
            assert np.all(np.diff(shard_of.astype(np.int64)) >= 0)      # contiguous runs in caller order
            assert int(blocks.max()) - int(blocks.min()) <= 2 * int(per_file.max()) + int(per_file.sum()) // (64 * k)
    # the full-size asset corpus: 100 GB, 70 % .tex-like bytes, 30 % .mdl-like, over 8 devices
    import sys

    sys.path.insert(0, H.ROOT)
    import bench

    ln = bench.make_sizes("corpus", np.random.default_rng(0x534E4150), 1.0)
    shard_of, blocks = plan_shards_lpt(ln, 8)
    assert blocks.max() / blocks.mean() < 1.001
    assert len(set(shard_of.tolist())) == 8


def test_integration_text_matches_the_library(snappy):
    """The P/Invoke declarations a maintainer would paste (the integration notes at the repository root) against the header and the
    ctypes table: same names, same number of arguments, pointers where the C side has pointers,
    integers of the same width."""
    text = open(os.path.join(H.ROOT, "INTEGRATION.md")).read()
    decls = re.findall(r"\[LibraryImport\([^\]]*\)\]\s*internal static partial\s+(\w+)\s+(snappy_\w+)\s*\(([^;]*?)\);",
                       text, flags=re.S)
    assert len(decls) >= 25
    header = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    protos = {m[1]: (m[0], m[2]) for m in re.findall(
        r"^\s*((?:const\s+)?[A-Za-z_0-9]+\s*\*?)\s*(snappy_\w+)\s*\(([^;]*?)\)\s*;", header, flags=re.S | re.M)}
    table = {name: (restype, argtypes) for name, restype, argtypes in snappy.native.SYMBOLS}

    scalars = {"uint64_t": "u64", "int64_t": "i64", "uint32_t": "u32", "int": "i32", "float": "f32",
               "double": "f64", "uint8_t": "u8", "char": "u8", "void": "void"}

    def c_kind(param):
        words = param.replace("const", " ").replace("*", " * ").split()
        stars = words.count("*") + (1 if "[" in param else 0)
        base = scalars[words[0]]
        if stars == 0:
            return base
        return "ptr:ptr" if stars > 1 else "ptr:" + base

    def cs_kind(param):
        t = param.strip().rsplit(None, 1)[0]
        if t in ("string", "IntPtr"):
            return "ptr:u8"
        if t == "string[]" or t.endswith("**"):
            return "ptr:ptr"
        base = {"ulong": "u64", "long": "i64", "uint": "u32", "int": "i32", "float": "f32", "double": "f64",
                "byte": "u8", "void": "void"}[t.rstrip("*")]
        return "ptr:" + base if t.endswith("*") else base

    def ct_kind(t):      # the ctypes table only tells pointers from integers
        if t in (ctypes.c_void_p, ctypes.c_char_p) or hasattr(t, "contents"):
            return "ptr"
        return {ctypes.c_uint64: "u64", ctypes.c_int64: "i64", ctypes.c_uint32: "u32", ctypes.c_int: "i32",
                ctypes.c_float: "f32"}[t]

    seen = set()
    for ret, name, params in decls:
        assert name in protos, f"{name} is bound in INTEGRATION.md but not declared in the header"
        c_params = [p for p in protos[name][1].split(",") if p.strip() and p.strip() != "void"]
        cs_params = [p for p in params.split(",") if p.strip()]
        assert len(cs_params) == len(c_params), name
        assert [cs_kind(p) for p in cs_params] == [c_kind(p) for p in c_params], name
        assert [ct_kind(t) for t in table[name][1]] == [c_kind(p).split(":")[0] for p in c_params], name
        c_ret = protos[name][0].strip()
        want_ret = {"int": "int", "void": "void", "float": "float", "const char *": "IntPtr"}[c_ret]
        assert ret == want_ret, name
        seen.add(name)
    # every entry point of the header has a binding in the text
    assert seen == set(protos), set(protos) ^ seen
