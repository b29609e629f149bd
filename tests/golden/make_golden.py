"""Generates the committed golden vectors under tests/golden/.

Run in the build container only (python tests/golden/make_golden.py).  Nothing
here touches the reference tree: the reference holds no tests or vectors for
this path, so the vectors are pinned by independent implementations present in
the container:

  sha1_golden.json   digests from Python's hashlib (OpenSSL) over splitmix64
                     data generated in pure Python, plus the FIPS 180-4 answers
  lz4_golden.json    raw LZ4 blocks produced by the system liblz4 1.9.3
                     (LZ4_compress_default and LZ4_compress_HC) with the SHA-1 of
                     the bytes they must decode to
"""
from __future__ import annotations

import ctypes
import hashlib
import json
import os
import random

HERE = os.path.dirname(os.path.abspath(__file__))
MASK = (1 << 64) - 1


def splitmix_bytes(length: int, seed: int) -> bytes:
    x = seed & MASK
    out = bytearray()
    while len(out) < length:
        x = (x + 0x9E3779B97F4A7C15) & MASK
        z = x
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK
        z ^= z >> 31
        out += z.to_bytes(8, "little")
    return bytes(out[:length])


def file_seed(index: int) -> int:
    return (0x534E415000000000 ^ index) & MASK


def make_sha1():
    fips = [
        {"ascii": "", "sha1": "DA39A3EE5E6B4B0D3255BFEF95601890AFD80709"},
        {"ascii": "abc", "sha1": "A9993E364706816ABA3E25717850C26C9CD0D89D"},
        {"ascii": "abcdbcdecdefdefgefghfghighijhijkijkljklmklmnlmnomnopnopq",
         "sha1": "84983E441C3BD26EBAAE4AA1F95129E5E54670F1"},
        {"repeat": ["a", 1000000], "sha1": "34AA973CD4C4DAA4F61EEB2BDBAD27316534016F"},
    ]
    for e in fips:
        data = e["ascii"].encode() if "ascii" in e else e["repeat"][0].encode() * e["repeat"][1]
        assert hashlib.sha1(data).hexdigest().upper() == e["sha1"]
    zeros = {}
    for n in (55, 56, 63, 64, 65, 119, 120, 128):
        zeros[str(n)] = hashlib.sha1(b"\0" * n).hexdigest().upper()
    rng = random.Random(20250815)
    lengths = list(range(0, 130)) + [191, 192, 193, 255, 256, 257, 4095, 4096, 4097, 65535, 65536,
                                     65537, (1 << 20) - 1, 1 << 20, (1 << 20) + 1]
    lengths += [rng.randrange(0, (1 << 20) + 64) for _ in range(120)]
    rows = []
    for i, n in enumerate(lengths):
        data = splitmix_bytes(n, file_seed(i))
        rows.append({"index": i, "len": n, "sha1": hashlib.sha1(data).hexdigest().upper()})
    with open(os.path.join(HERE, "sha1_golden.json"), "w") as f:
        json.dump({"fips": fips, "zeros": zeros, "splitmix": rows}, f, indent=0)


def compressible(n: int, seed: int) -> bytes:
    r = random.Random(seed)
    out = bytearray()
    while len(out) < n:
        k = r.random()
        if k < 0.3:
            out += splitmix_bytes(r.randint(1, 300), r.getrandbits(64))
        elif k < 0.55:
            out += bytes([r.randint(0, 255)]) * r.randint(1, 700)
        else:
            out += splitmix_bytes(r.randint(1, 48), r.getrandbits(64)) * r.randint(1, 40)
    return bytes(out[:n])


def make_lz4():
    lz4 = ctypes.CDLL("liblz4.so.1")
    rows = []
    sizes = [1, 4, 5, 12, 13, 14, 20, 64, 100, 300, 1000, 4096, 5000, 20000, 70000]
    for i, n in enumerate(sizes):
        for hc in (0, 1):
            data = compressible(n, 1000 + i)
            cap = lz4.LZ4_compressBound(n)
            out = ctypes.create_string_buffer(cap)
            if hc:
                w = lz4.LZ4_compress_HC(data, out, n, cap, 9)
            else:
                w = lz4.LZ4_compress_default(data, out, n, cap)
            assert w > 0
            rows.append({"orig_len": n, "hc": hc, "block_hex": out.raw[:w].hex(),
                         "sha1": hashlib.sha1(data).hexdigest().upper(),
                         "gen": ["compressible", n, 1000 + i]})
    with open(os.path.join(HERE, "lz4_golden.json"), "w") as f:
        json.dump({"liblz4_version": lz4.LZ4_versionNumber(), "blocks": rows}, f, indent=0)


if __name__ == "__main__":
    make_sha1()
    make_lz4()
    print("wrote", sorted(os.listdir(HERE)))
