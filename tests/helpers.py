"""Shared test data: hand-assembled LZ4 blocks, lz4net streams, synthetic corpora."""
from __future__ import annotations

import json
import os
import random
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
MASK = (1 << 64) - 1


def load_golden(name: str):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def splitmix_bytes(length: int, seed: int) -> bytes:
    """Pure-Python splitmix64 stream (independent of the C oracle's generator)."""
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


def random_bytes(n: int, seed: int) -> bytes:
    return np.random.default_rng(seed).integers(0, 256, size=n, dtype=np.uint8).tobytes()


def compressible(n: int, seed: int) -> bytes:
    """Mix of random runs, byte runs and repeated short motifs (same recipe as make_golden.py)."""
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


def asset_like(n: int, seed: int) -> bytes:
    """Half random, half compressible in long stretches: motifs of 16-4096 bytes and zero runs."""
    r = random.Random(seed)
    rng = np.random.default_rng(seed)
    out = bytearray()
    while len(out) < n:
        k = r.random()
        if k < 0.5:
            out += rng.integers(0, 256, size=r.randint(256, 200_000), dtype=np.uint8).tobytes()
        elif k < 0.7:
            out += bytes(r.randint(64, 100_000))
        else:
            motif = rng.integers(0, 256, size=r.randint(16, 4096), dtype=np.uint8).tobytes()
            out += motif * r.randint(2, 60)
    return bytes(out[:n])


# ----------------------------------------------------------------- LZ4 block vectors

def _len_bytes(extra: int) -> bytes:
    out = bytearray()
    while extra >= 255:
        out.append(255)
        extra -= 255
    out.append(extra)
    return bytes(out)


def seq(literals: bytes, offset=None, match_len=None) -> bytes:
    """One LZ4 sequence; offset=None makes the final literals-only sequence."""
    lit = len(literals)
    tok_l = min(lit, 15)
    out = bytearray()
    if offset is None:
        out.append(tok_l << 4)
        if lit >= 15:
            out += _len_bytes(lit - 15)
        out += literals
        return bytes(out)
    code = match_len - 4
    tok_m = min(code, 15)
    out.append((tok_l << 4) | tok_m)
    if lit >= 15:
        out += _len_bytes(lit - 15)
    out += literals
    out += bytes([offset & 0xFF, offset >> 8])
    if code >= 15:
        out += _len_bytes(code - 15)
    return bytes(out)


def expand(parts) -> bytes:
    """Reference expansion of [(literals, offset, match_len), ...] done byte by byte."""
    out = bytearray()
    for literals, offset, mlen in parts:
        out += literals
        if offset is not None:
            for _ in range(mlen):
                out.append(out[-offset])
    return bytes(out)


def block(parts) -> tuple:
    """(block_bytes, expected_output) for a list of sequences."""
    return b"".join(seq(*p) for p in parts), expand(parts)


def good_block_vectors():
    r65535 = splitmix_bytes(65535, 77)
    vecs = {
        "literals_only": block([(b"hello", None, None)]),
        "single_byte": block([(b"x", None, None)]),
        "rle_offset1": block([(b"a", 1, 10), (b"bcdef", None, None)]),
        "pattern_offset3": block([(b"abc", 3, 9), (b"vwxyz", None, None)]),
        "offset_lt_len_long": block([(b"0123456", 7, 5000), (b"tail!", None, None)]),
        "offset2_len_70": block([(b"xy", 2, 70), (b"12345", None, None)]),
        "offset63": block([(splitmix_bytes(63, 1), 63, 300), (b"12345", None, None)]),
        "offset64": block([(splitmix_bytes(64, 2), 64, 300), (b"12345", None, None)]),
        "offset65": block([(splitmix_bytes(65, 3), 65, 300), (b"12345", None, None)]),
        "offset300_len_5000": block([(splitmix_bytes(300, 4), 300, 5000), (b"12345", None, None)]),
        "offset5000_len_20000": block([(splitmix_bytes(5000, 5), 5000, 20000), (b"12345", None, None)]),
        "lit_ext_273": block([(splitmix_bytes(273, 6), 100, 4), (b"12345", None, None)]),
        "lit_ext_exact_15": block([(splitmix_bytes(15, 7), 5, 4), (b"12345", None, None)]),
        "lit_ext_270": block([(splitmix_bytes(270, 8), 5, 4), (b"12345", None, None)]),
        "match_ext_chain": block([(b"abcdefgh", 8, 4 + 15 + 255 + 255 + 7), (b"12345", None, None)]),
        "match_ext_exact": block([(b"abcdefgh", 8, 4 + 15), (b"12345", None, None)]),
        "match_ext_1020": block([(b"abcdefgh", 3, 4 + 15 + 1020), (b"12345", None, None)]),
        "zero_literals_then_match": block([(b"abcd", 4, 8), (b"", 2, 6), (b"12345", None, None)]),
        "max_offset": block([(r65535, 65535, 8), (b"12345", None, None)]),
        "long_literals_5000": block([(splitmix_bytes(5000, 9), None, None)]),
        "many_sequences": block([(splitmix_bytes(3 + (i % 7), 100 + i), 1 + (i % 3), 4 + (i % 40))
                                 for i in range(400)] + [(b"12345", None, None)]),
        "chained_overlap": block([(b"ab", 2, 6), (b"", 1, 5), (b"c", 4, 30), (b"", 9, 100),
                                  (b"12345", None, None)]),
    }
    return vecs


def bad_block_vectors():
    """(block_bytes, claimed_output_len) pairs every conforming decoder must reject."""
    good, out = block([(b"a", 1, 10), (b"bcdef", None, None)])
    vecs = {
        "match_ends_at_end": (seq(b"a", 1, 10), 11),                       # r7x: "op == oend" after a match
        "match_ends_at_end_then_empty_literals": (seq(b"a", 1, 10) + seq(b""), 11),
        "match_past_end": (seq(b"a", 1, 10) + seq(b"bcde"), 10),
        "match_near_end_then_too_many_literals": (seq(b"a", 1, 10) + seq(b"bcdef"), 15),
        "literal_run_in_last8_not_at_end": (seq(b"0123456789", 4, 4) + seq(b"ab"), 16),
        "offset_zero": (seq(b"abcd", 0, 4) + seq(b"12345"), 13),
        "offset_before_start": (seq(b"abcd", 5, 4) + seq(b"12345"), 13),
        "truncated_literals": (good[:-2], len(out)),
        "trailing_garbage": (good + b"\x00", len(out)),
        "output_too_short": (good, len(out) - 1),
        "output_too_long": (good, len(out) + 1),
        "empty_input": (b"", 4),
        "token_only": (b"\x10", 1),
        "ext_runs_off_end": (b"\xf0\xff\xff", 600),
        "offset_cut": (seq(b"abcd", 4, 8)[:-1], 17),
        "huge_literal_claim": (b"\xf0" + b"\xff" * 8 + b"\x00", 64),
    }
    # A match that reaches into the last 5 bytes of the block (it ends 1-4 bytes before the
    # end, followed by that many final literals).  No compressor writes this; today's liblz4
    # refuses it and so does the LASTLITERALS test of the r7x decoder that lz4net ports (as
    # recalled: nothing here can check lz4net itself).
    for tail in (1, 2, 3, 4):
        for name, parts in ((f"match_into_last5_{tail}", [(b"a", 1, 10), (b"wxyz"[:tail], None, None)]),
                            (f"long_match_into_last5_{tail}", [(splitmix_bytes(40, 60 + tail), 33, 700),
                                                               (b"wxyz"[:tail], None, None)])):
            data, out = block(parts)
            vecs[name] = (data, len(out))
    return vecs


# ----------------------------------------------------------------- lz4net framing

def varint(v: int) -> bytes:
    out = bytearray()
    while v >= 0x80:
        out.append((v & 0x7F) | 0x80)
        v >>= 7
    out.append(v)
    return bytes(out)


def chunk(payload: bytes, original_len: int, compressed: bool, flags_extra: int = 0) -> bytes:
    flags = (1 if compressed else 0) | flags_extra
    out = varint(flags) + varint(original_len)
    if compressed:
        out += varint(len(payload))
    return out + payload


def mixed_sizes(rng: np.random.Generator, n: int, lo: int, hi: int) -> np.ndarray:
    """Log-uniform sizes in [lo, hi]."""
    return np.exp(rng.uniform(np.log(lo), np.log(hi), size=n)).astype(np.int64)
