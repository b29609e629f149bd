"""Randomised soak of the LZ4 paths: random content (motifs of random period and repeat count,
random bytes, zero runs, text) of random size and block size goes through the GPU compressor
and back through the GPU decoder; streams from the CPU oracle's encoder (another compressor)
go through the GPU decoder as well; a share of the streams gets a random byte flipped and must
then be either rejected or decoded exactly as the CPU oracle decodes it.
Usage: python tools/soak_lz4.py [seconds]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def content(rng, n):
    parts, total = [], 0
    while total < n:
        k = rng.random()
        if k < 0.25:
            seg = rng.integers(0, 256, size=int(rng.integers(1, 200_000)), dtype=np.uint8)
        elif k < 0.4:
            seg = np.zeros(int(rng.integers(1, 300_000)), dtype=np.uint8)
        elif k < 0.9:
            m = rng.integers(0, 256, size=int(np.exp(rng.uniform(0, np.log(5000)))), dtype=np.uint8)
            seg = np.tile(m, int(np.exp(rng.uniform(np.log(2), np.log(400)))))
        else:
            seg = np.frombuffer(b"lorem ipsum dolor sit amet, " * int(rng.integers(1, 3000)), dtype=np.uint8)
        parts.append(seg)
        total += seg.size
    return np.concatenate(parts)[:n]


def main():
    import snappy_amd as S
    from oracle import oracle

    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    rng = np.random.default_rng(7)
    t0 = time.time()
    rounds = flips = rejected = 0
    while time.time() - t0 < budget:
        n = int(np.exp(rng.uniform(np.log(1), np.log(24 << 20))))
        data = content(rng, n).tobytes()
        block = int(rng.choice([0, 4096, 65536, 1 << 18, 1 << 20, 4 << 20]))
        stream = S.lz4net_encode(data, block)
        assert S.lz4net_decode(stream) == data, ("gpu->gpu", n, block)
        if n <= 4 << 20:
            st, back = oracle.lz4net_decode(stream)
            assert st == 0 and back == data, ("gpu->oracle", n, block)
            other = oracle.lz4net_encode(data, block or (1 << 20), False)
            assert S.lz4net_decode(other) == data, ("oracle->gpu", n, block)
            if len(other) > 8:
                bad = bytearray(other)
                bad[int(rng.integers(0, len(bad)))] ^= 1 << int(rng.integers(0, 8))
                st, want = oracle.lz4net_decode(bytes(bad))
                flips += 1
                try:
                    got = S.lz4net_decode(bytes(bad))
                    assert st == 0 and got == want, ("flip accepted differently", n, st)
                except S.native.SnappyError as e:
                    assert st != 0 and e.code == st, ("flip rejected differently", n, st, e.code)
                    rejected += 1
        rounds += 1
        if rounds % 20 == 0:
            print(f"{rounds} rounds ok, {flips} flipped streams ({rejected} rejected), {time.time() - t0:.0f} s", flush=True)
    print("done:", rounds, "rounds,", flips, "flips,", rejected, "rejected")


if __name__ == "__main__":
    main()
