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
