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


def batch_census(stream, max_chunks):
    """Which sequences of the first chunks would the decoder take in batches, and what ends a batch."""
    from collections import Counter
    pos = 0
    why = Counter()
    sizes = []

    def varint():
        nonlocal pos
        v = shift = 0
        while True:
            b = int(stream[pos]); pos += 1
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
        if flags & 1:
            op = 0
            start, count = 0, 0

            def close(reason):
                nonlocal start, count
                why[reason] += 1
                if count:
                    sizes.append(count)
                count = 0

            while pos < end:
                tok = int(stream[pos]); pos += 1
                lit, ext_l = tok >> 4, 0
                if lit == 15:
                    while True:
                        b = int(stream[pos]); pos += 1; ext_l += 1
                        lit += b
                        if b != 255:
                            break
                pos += lit
                if pos >= end:
                    close("end of block")
                    break
                off = int(stream[pos]) | int(stream[pos + 1]) << 8
                pos += 2
                ml, ext_m = tok & 15, 0
                if ml == 15:
                    while True:
                        b = int(stream[pos]); pos += 1; ext_m += 1
                        ml += b
                        if b != 255:
                            break
                ml += 4
                if count == 0:
                    start = op
                mo = op + lit
                reason = None
                if lit > 240 or ext_l > 1:
                    reason = "long literals"
                elif ml > 255 or ext_m > 1:
                    reason = "long match"
                elif off < ml:
                    reason = "overlapping match"
                elif pos + 4 > end or mo + ml + 16 > n:
                    reason = "near the end"
                elif mo + ml - start > 4096:
                    reason = "span"
                if reason is None:
                    s = mo - off
                    in_ring = s + 8192 >= start + 4096
                    if not in_ring and s + ml > start:
                        reason = "source straddles"
                if reason == "span":
                    close("full (span)")
                    start = op
                    count = 1
                elif reason:
                    close(reason)
                else:
                    count += 1
                    if count == 64:
                        close("full (64)")
                op = mo + ml
        pos = end
    total = sum(sizes)
    print(f"   batches: {len(sizes)}, sequences in batches {total}, mean size {total / max(len(sizes), 1):.1f}; "
          f"what ended them: {dict(why)}")


def main():
    import bench
    from snappy_amd import native

    lib = native.lib()
    native.check(lib.snappy_init(0), "init")
    lib.snappy_profile_enable(1)
    rng = np.random.default_rng(2)
    n = 2 << 30

    def motif(lo, hi, reps_hi):
        unit = 64 << 20   # generated once, then repeated: chunks are compressed independently
        parts, total = [], 0
        while total < unit:
            m = rng.integers(0, 256, size=int(rng.integers(lo, hi)), dtype=np.uint8)
            seg = np.tile(m, int(rng.integers(2, reps_hi)))
            parts.append(seg)
            total += seg.size
        return np.tile(np.concatenate(parts)[:unit], n // unit)

    def lz_like(max_lit, max_match, max_off):
        return bench.dense_bytes(n, rng, max_lit, max_match, max_off)

    def cpu_encoded(max_lit, max_match, max_off):
        """The same content through the CPU oracle's greedy encoder, which finds the copies the
        generator made (13 input bytes per sequence instead of 45): 16 MiB encoded once, repeated."""
        from oracle import oracle

        unit = bench.dense_bytes(16 << 20, rng, max_lit, max_match, max_off)
        one = np.frombuffer(oracle.lz4net_encode(unit.tobytes()), dtype=np.uint8)
        reps = n // unit.size
        return np.tile(unit, reps), np.tile(one, reps)

    kinds = {
        "lz-like 24/36, CPU-encoded": lambda: cpu_encoded(24, 36, 65535),
        "lz-like lit<=24 match<=36": lambda: lz_like(24, 36, 65535),
        "lz-like lit<=8 match<=20 near": lambda: lz_like(8, 20, 2048),
        "lz-like lit<=100 match<=200": lambda: lz_like(100, 200, 65535),
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
        w = ctypes.c_uint64(0)
        if isinstance(data, tuple):
            data, stream = data
            w.value = stream.size
            enc_ms = float("nan")
        else:
            bound = ctypes.c_uint64(0)
            lib.snappy_lz4net_encode_bound(data.size, 0, ctypes.byref(bound))
            stream = np.zeros(bound.value, dtype=np.uint8)
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
            batch_census(stream[: w.value], 8)
            print(f"   first chunks: {seqs} sequences in {walked} decoded bytes = {walked / max(seqs, 1):.1f} B per sequence")
        k = min(ms)
        print(f"{name:30s} ratio {n / w.value:8.2f}  decode {k:6.3f} ms  {(n + w.value) / k / 1e6:8.1f} GB/s (in+out)  "
              f"encode {enc_ms:7.2f} ms  roundtrip {'ok' if ok else 'WRONG'}", flush=True)


if __name__ == "__main__":
    main()
