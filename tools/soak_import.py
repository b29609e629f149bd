"""Randomised soak of the MCDF batch import: batches of 1-14 random archives (random file counts and
sizes, asset-like and random bodies, zero-length files, declared lengths past the payload), some
truncated, corrupted, empty or with a bad header, go through snappy_mcdf_decode_and_hash_batch in
the overlapped order (hashing planned group by group in arrival order, one to eight archives per
decode launch) and in the plain order, from two threads at once; every archive's status, file
statuses and digests must equal the CPU oracle's import of the same stream.
Usage: python tools/soak_import.py [seconds]"""
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))


def make_stream(oracle, H, rng, k):
    n_files = int(rng.integers(1, 40))
    files = []
    for i in range(n_files):
        kind = rng.random()
        n = 0 if kind < 0.1 else int(np.exp(rng.uniform(0, np.log(1_500_000))))
        if n == 0:
            body = None
        elif kind < 0.6:
            body = H.asset_like(n, int(rng.integers(1 << 30)))
        else:
            body = H.random_bytes(n, int(rng.integers(1 << 30)))
        paths = [f"chara/soak/a{k}/f{i}.tex"] + ([f"CHARA/SOAK/A{k}/F{i}.TEX"] if i % 5 == 0 else [])
        files.append((paths, body, n))
    payload = oracle.build_mcdf_payload(f"soak {k}", files)
    cut = rng.random()
    if cut < 0.1:          # the last files' declared lengths run past the payload: short reads
        payload = payload[: len(payload) - int(rng.integers(1, min(5000, len(payload) // 2)))]
    stream = oracle.lz4net_encode(payload, chunk_size=int(rng.choice([1 << 12, 1 << 15, 1 << 16, 1 << 20])))
    flaw = rng.random()
    if flaw < 0.06:
        return stream[: int(rng.integers(1, len(stream)))]          # truncated stream
    if flaw < 0.12:
        b = bytearray(stream)
        at = int(rng.integers(0, len(b)))
        b[at: at + 16] = b"\xff" * min(16, len(b) - at)
        return bytes(b)                                              # damaged tokens or headers
    if flaw < 0.15:
        return b""
    if flaw < 0.18:
        return oracle.lz4net_encode(b"XCDF" + bytes(int(rng.integers(5, 300))), chunk_size=1 << 16)
    return stream


def main():
    import helpers as H
    import snappy_amd as S
    from oracle import oracle
    from snappy_amd.crypto import digests_to_hex

    oracle.build()
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    os.environ["SNAPPY_OVERLAP_MIN_BYTES"] = "0"
    t0 = time.time()
    counts = {"batches": 0, "archives": 0, "rejected": 0, "files": 0}
    errors = []
    lock = threading.Lock()

    def expected(s):
        try:
            return oracle.mcdf_import(s), 0
        except oracle.McdfError as e:
            return None, e.code

    def worker(seed, until):
        rng = np.random.default_rng(seed)
        k = 0
        while time.time() < until and not errors:
            batch = [make_stream(oracle, H, rng, seed * 100000 + k + j) for j in range(int(rng.integers(1, 15)))]
            k += len(batch)
            want = [expected(s) for s in batch]
            got = S.mcdf_decode_and_hash_batch(batch)
            for s, (ref, status), r in zip(batch, want, got):
                ok = r["status"] == status
                if ok and ref is not None:
                    ok = list(r["statuses"]) == ref["statuses"]
                    hx = digests_to_hex(r["digests"])
                    for (paths, _), st_, h in zip(ref["data"]["Files"], ref["statuses"], hx):
                        if st_ == oracle.FILE_HASHED and paths:
                            ok = ok and ref["map"][paths[0]] == h
                if not ok:
                    with lock:
                        errors.append((seed, len(s), status, r["status"]))
            with lock:
                counts["batches"] += 1
                counts["archives"] += len(batch)
                counts["rejected"] += sum(1 for _, st in want if st != 0)
                counts["files"] += sum(len(ref["statuses"]) for ref, _ in want if ref)

    # The order of things changes from one spell to the next (the environment is only touched while
    # no call is in flight).
    mode_rng = np.random.default_rng(99)
    spell = 0
    while time.time() - t0 < budget and not errors:
        if mode_rng.random() < 0.25:
            os.environ["SNAPPY_MCDF_PLAIN_ORDER"] = "1"
        else:
            os.environ.pop("SNAPPY_MCDF_PLAIN_ORDER", None)
        os.environ["SNAPPY_MCDF_DECODE_RUN"] = str(int(mode_rng.choice([1, 1, 2, 3, 8])))
        if mode_rng.random() < 0.3:   # whole archives only: the largest file's chunks are not sent ahead
            os.environ["SNAPPY_MCDF_NO_SPLIT"] = "1"
        else:
            os.environ.pop("SNAPPY_MCDF_NO_SPLIT", None)
        until = min(time.time() + 6.0, t0 + budget)
        threads = [threading.Thread(target=worker, args=(1000 * spell + s, until)) for s in (1, 2)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        spell += 1
    print(f"{counts['batches']} batches, {counts['archives']} archives ({counts['rejected']} rejected as the oracle "
          f"rejects them), {counts['files']} files in {time.time() - t0:.0f} s; mismatches: {errors[:5] if errors else 'none'}",
          flush=True)
    sys.exit(1 if errors else 0)


if __name__ == "__main__":
    main()
