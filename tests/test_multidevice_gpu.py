"""The in-process multi-device dispatcher: one physical GPU presented as three logical devices
(SNAPPY_LOGICAL_DEVICES), so that host batches are split into shards, each hashed or decoded
by its own host thread, slot and streams, and the results concatenated in input order."""
import os
import subprocess
import sys

import pytest

import helpers as H

pytestmark = pytest.mark.gpu

CHILD = r"""
import hashlib, sys
import numpy as np
sys.path.insert(0, {root!r})
sys.path.insert(0, {root!r} + "/tests")
import helpers as H
import snappy_amd as S
from snappy_amd.crypto import device_work, plan_shards_lpt
L = S.native.lib()
assert L.snappy_init(0) == 0
assert L.snappy_device_count() == 3, L.snappy_device_count()

def work():
    return [device_work(d) for d in range(3)]

def delta(before, after):
    return [(a[0] - b[0], a[1] - b[1]) for b, a in zip(before, after)]

def expected(sizes):
    k = min(3, len(sizes), max(1, int(sizes.sum()) // (8 << 20)))   # no shard below 8 MiB
    assert k == 3, "the test batches are meant to use every device"
    shard_of, _ = plan_shards_lpt(sizes, k)
    return [(int((shard_of == d).sum()), int(sizes[shard_of == d].sum())) for d in range(3)]

rng = np.random.default_rng(5)
sizes = np.concatenate([rng.integers(0, 200_000, size=400), rng.integers(1_000_000, 6_000_000, size=12)]).astype(np.uint64)
rng.shuffle(sizes)
offsets = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.uint64)
base = rng.integers(0, 256, size=int(sizes.sum()) + 1, dtype=np.uint8)
want = [hashlib.sha1(base[int(o): int(o + n)].tobytes()).digest() for o, n in zip(offsets, sizes)]
for rep in range(2):
    w0 = work()
    got = S.sha1_batch(base, offsets, sizes)
    assert [g.tobytes() for g in got] == want, "sha1_batch over 3 devices"
    # every device hashed exactly the files the longest-first plan gives it
    assert delta(w0, work()) == expected(sizes), (delta(w0, work()), expected(sizes))
blobs = [base[int(o): int(o + n)].tobytes() for o, n in zip(offsets[:300], sizes[:300])]
w0 = work()
got = S.sha1_files(blobs)
assert [g.tobytes() for g in got] == want[:300], "sha1_files over 3 devices"
assert delta(w0, work()) == expected(sizes[:300])

# The 100 GB asset corpus in small: 70 % of the bytes in .tex-like files (256 KB-22 MB, log-uniform),
# 30 % in .mdl-like ones (64 KB-6 MB), shuffled, sharded by file over the three devices.
target = 600e6
tex, mdl, t, m = [], [], 0.0, 0.0
while t < 0.7 * target:
    v = float(np.exp(rng.uniform(np.log(256 << 10), np.log(22 << 20)))); tex.append(v); t += v
while m < 0.3 * target:
    v = float(np.exp(rng.uniform(np.log(64 << 10), np.log(6 << 20)))); mdl.append(v); m += v
csz = np.array(tex + mdl); rng.shuffle(csz); csz = csz.astype(np.uint64)
coff = np.concatenate([[0], np.cumsum(csz)[:-1]]).astype(np.uint64)
corpus = np.frombuffer(H.random_bytes(int(csz.sum()), 4242), dtype=np.uint8)
cwant = [hashlib.sha1(corpus[int(o): int(o + n)].tobytes()).digest() for o, n in zip(coff, csz)]
w0 = work()
got = S.sha1_batch(corpus, coff, csz)
assert [g.tobytes() for g in got] == cwant, "corpus over 3 devices"
placed = delta(w0, work())
assert placed == expected(csz), (placed, expected(csz))
assert sum(p[0] for p in placed) == len(csz) and sum(p[1] for p in placed) == int(csz.sum())
blocks = (csz + 8) // 64 + 1
_, per_shard = plan_shards_lpt(csz, 3)
# longest-first by padded blocks: no device above the average by more than the largest file
assert int(per_shard.max()) <= int(blocks.sum()) // 3 + int(blocks.max())
assert min(p[0] for p in placed) > 0

# MCDF batch: five archives over three devices, sharded by archive (largest first, least loaded device)
archives, expect = [], []
for a in range(5):
    bodies = [H.asset_like(int(n), 100 * a + i) for i, n in enumerate(rng.integers(1, 300_000 * (a + 1), size=9))]
    archives.append(S.write_mcdf([[f"chara/a{{a}}/f{{i}}.tex"] for i in range(len(bodies))], bodies, description=f"a{{a}}"))
    expect.append([hashlib.sha1(b).digest() for b in bodies])
w0 = work()
res = S.mcdf_decode_and_hash_batch(archives)
for a, r in enumerate(res):
    assert r["status"] == 0
    assert [d.tobytes() for d in r["digests"][: len(expect[a])]] == expect[a], a
weights = [len(s) + S.lz4net_decoded_size(s) for s in archives]
load, count, size = [0, 0, 0], [0, 0, 0], [0, 0, 0]
for a in sorted(range(5), key=lambda a: -weights[a]):
    d = load.index(min(load))
    load[d] += weights[a]; count[d] += 1; size[d] += S.lz4net_decoded_size(archives[a])
assert delta(w0, work()) == list(zip(count, size)), (delta(w0, work()), count, size)
assert sum(count) == 5 and min(count) >= 1
print("MULTIDEVICE OK")
"""


def test_three_logical_devices(tmp_path):
    env = dict(os.environ)
    env["SNAPPY_LOGICAL_DEVICES"] = "3"
    out = subprocess.run([sys.executable, "-c", CHILD.format(root=H.ROOT)], capture_output=True, text=True,
                         timeout=300, env=env)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    assert "MULTIDEVICE OK" in out.stdout


LIFECYCLE = r"""
import sys
import numpy as np
sys.path.insert(0, {root!r})
import snappy_amd as S
L = S.native.lib()
assert L.snappy_device_count() == 0            # nothing bound before the first call
assert S.get_hash(b"abc") == "A9993E364706816ABA3E25717850C26C9CD0D89D"   # binds on first use
assert L.snappy_device_count() >= 1
big = np.random.default_rng(1).integers(0, 256, size=80 << 20, dtype=np.uint8)
lens = np.array([50 << 20, 30 << 20], dtype=np.uint64)
offs = np.array([0, 50 << 20], dtype=np.uint64)
first = S.sha1_batch(big, offs, lens)            # hash-while-copying path, staging ring allocated
L.snappy_shutdown()
assert L.snappy_device_count() == 0
again = S.sha1_batch(big, offs, lens)            # everything is rebuilt on demand
assert np.array_equal(first, again)
L.snappy_shutdown()
assert L.snappy_init(1) == 0 and L.snappy_device_count() == 1
assert S.get_hash(b"") == "DA39A3EE5E6B4B0D3255BFEF95601890AFD80709"
L.snappy_shutdown()
L.snappy_shutdown()                              # idempotent
print("LIFECYCLE OK")
"""


def test_shutdown_and_reinit():
    out = subprocess.run([sys.executable, "-c", LIFECYCLE.format(root=H.ROOT)], capture_output=True, text=True,
                         timeout=300)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    assert "LIFECYCLE OK" in out.stdout
