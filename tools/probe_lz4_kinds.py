"""Decode-kernel time per kind of content (2 GiB each, 2048 chunks of 1 MiB): where does a chunk's
time go?  Streams are made with the library's compressor.  Usage: python tools/probe_lz4_kinds.py"""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def count_sequences(stream, max_chunks):
    """Walks the lz4net framing and the LZ4 tokens of the first chunks."""
    pos = seqs = decoded = 0

    def varint():
        nonlocal pos
        v = shift = 0
        while True:
            b = int(stream[pos])
            pos += 1
            v |= (b & 0x7F) << shift
            shift += 7
            if b < 0x80:
                return v

    for _ in range(max_chunks):
        if pos >= len(stream):
            break
        flags, n = varint(), varint()
        c = varint() if flags & 1 else n
        end = pos + c
        decoded += n
        if flags & 1:
            while pos < end:
                tok = int(stream[pos]); pos += 1
                lit = tok >> 4
                if lit == 15:
                    while True:
                        b = int(stream[pos]); pos += 1
                        lit += b
                        if b != 255:
                            break
                pos += lit
                seqs += 1
                if pos >= end:
                    break
                pos += 2
                if tok & 15 == 15:
                    while True:
                        b = int(stream[pos]); pos += 1
                        if b != 255:
                            break
        pos = end
    return seqs, decoded


def main():
    import bench
    from snappy_amd import native

    lib = native.lib()
    native.check(lib.snappy_init(0), "init")
    lib.snappy_profile_enable(1)
    rng = np.random.default_rng(2)
    n = 2 << 30

    def motif(lo, hi, reps_hi):
        parts, total = [], 0
        while total < n:
            m = rng.integers(0, 256, size=int(rng.integers(lo, hi)), dtype=np.uint8)
            seg = np.tile(m, int(rng.integers(2, reps_hi)))
            parts.append(seg)
            total += seg.size
        return np.concatenate(parts)[:n]

    kinds = {
        "random (stored chunks)": lambda: rng.integers(0, 256, size=n, dtype=np.uint8),
        "zeros (one match per chunk)": lambda: np.zeros(n, dtype=np.uint8),
        "motifs 16-64 B x <80": lambda: motif(16, 64, 80),
        "motifs 1-4 KiB x <80": lambda: motif(1024, 4096, 80),
        "motifs 16-64 B x <6": lambda: motif(16, 64, 6),
        "asset-like mix": lambda: bench.asset_like_bytes(n, rng),
    }
    out = np.zeros(n + 4096, dtype=np.uint8)
    only = os.environ.get("LZ4_KIND")          # substring filter, e.g. LZ4_KIND="x <6"
    for name, make in kinds.items():
        if only and only not in name:
            continue
        data = make()
        bound = ctypes.c_uint64(0)
        lib.snappy_lz4net_encode_bound(data.size, 0, ctypes.byref(bound))
        stream = np.zeros(bound.value, dtype=np.uint8)
        w = ctypes.c_uint64(0)
        native.check(lib.snappy_lz4net_encode(data.ctypes.data, data.size, 0, stream.ctypes.data, stream.size,
                                              ctypes.byref(w)), "encode")
        enc_ms = lib.snappy_profile_last_ms(1)
        ms = []
        v = ctypes.c_uint64(0)
        for _ in range(3):
            native.check(lib.snappy_lz4net_decode(stream.ctypes.data, w.value, out.ctypes.data, out.size,
                                                  ctypes.byref(v)), "decode")
            ms.append(lib.snappy_profile_last_ms(1))
        ok = v.value == n and np.array_equal(out[:n], data)
        if only:   # sequences in the first chunks: token count by walking the blocks on the host
            seqs, walked = count_sequences(stream[: w.value], 8)
            print(f"   first chunks: {seqs} sequences in {walked} decoded bytes = {walked / max(seqs, 1):.1f} B per sequence")
        k = min(ms)
        print(f"{name:30s} ratio {n / w.value:8.2f}  decode {k:6.3f} ms  {(n + w.value) / k / 1e6:8.1f} GB/s (in+out)  "
              f"encode {enc_ms:7.2f} ms  roundtrip {'ok' if ok else 'WRONG'}", flush=True)


if __name__ == "__main__":
    main()
