"""Untrusted-input parsers on the host: the MCDF header reader under AddressSanitizer and UBSan
(its source file built on its own with g++), and the lz4net framing walk through the built
library with mutated streams (accept/reject must equal the CPU oracle's, and nothing may crash)."""
import os
import subprocess

import numpy as np

import helpers as H


def test_mcdf_header_reader_under_sanitizers(tmp_path):
    exe = str(tmp_path / "fuzz_mcdf")
    src = os.path.join(H.ROOT, "snappy_amd", "csrc")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-static-libasan",
           "-I", src, "-I", os.path.join(H.ROOT, "include"),
           os.path.join(H.ROOT, "tests", "fuzz", "fuzz_mcdf_header.cpp"), os.path.join(src, "mcdf_host.cpp"),
           "-o", exe]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-3000:]
    env = dict(os.environ)
    env["ASAN_OPTIONS"] = "detect_leaks=0:verify_asan_link_order=0"
    run = subprocess.run([exe, "60000"], capture_output=True, text=True, timeout=300, env=env)
    assert run.returncode == 0, run.stdout[-500:] + run.stderr[-3000:]
    assert "fuzz ok" in run.stdout


def test_lz4net_framing_walk_mutations(snappy, oracle):
    import ctypes

    L = snappy.native.lib()
    rng = np.random.default_rng(2024)
    base = oracle.lz4net_encode(H.compressible(300_000, 5), 1 << 16, True) + \
        oracle.lz4net_encode(H.random_bytes(70_000, 6), 1 << 15, False)
    for _ in range(3000):
        buf = bytearray(base)
        for _e in range(int(rng.integers(1, 4))):
            at = int(rng.integers(0, len(buf)))
            kind = int(rng.integers(0, 3))
            if kind == 0:
                buf[at] = int(rng.integers(0, 256))
            elif kind == 1:
                del buf[at:]
            else:
                buf[at:at] = bytes(rng.integers(0, 256, size=int(rng.integers(1, 6)), dtype=np.uint8))
        if not buf:
            continue
        arr = np.frombuffer(bytes(buf), dtype=np.uint8)
        v = ctypes.c_uint64(0)
        st = L.snappy_lz4net_decoded_size(arr.ctypes.data, arr.size, ctypes.byref(v))
        ost, osize = oracle.lz4net_decoded_size(bytes(buf))
        assert st == ost and (st != 0 or v.value == osize)
