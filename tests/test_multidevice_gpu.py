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
L = S.native.lib()
assert L.snappy_init(0) == 0
assert L.snappy_device_count() == 3, L.snappy_device_count()
rng = np.random.default_rng(5)
sizes = np.concatenate([rng.integers(0, 200_000, size=400), rng.integers(1_000_000, 6_000_000, size=12)]).astype(np.uint64)
rng.shuffle(sizes)
offsets = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.uint64)
base = rng.integers(0, 256, size=int(sizes.sum()) + 1, dtype=np.uint8)
want = [hashlib.sha1(base[int(o): int(o + n)].tobytes()).digest() for o, n in zip(offsets, sizes)]
for rep in range(2):
    got = S.sha1_batch(base, offsets, sizes)
    assert [g.tobytes() for g in got] == want, "sha1_batch over 3 devices"
blobs = [base[int(o): int(o + n)].tobytes() for o, n in zip(offsets[:150], sizes[:150])]
got = S.sha1_files(blobs)
assert [g.tobytes() for g in got] == want[:150], "sha1_files over 3 devices"
# MCDF batch: five archives over three devices
archives, expect = [], []
for a in range(5):
    bodies = [H.asset_like(int(n), 100 * a + i) for i, n in enumerate(rng.integers(1, 900_000, size=9))]
    archives.append(S.write_mcdf([[f"chara/a{{a}}/f{{i}}.tex"] for i in range(len(bodies))], bodies, description=f"a{{a}}"))
    expect.append([hashlib.sha1(b).digest() for b in bodies])
res = S.mcdf_decode_and_hash_batch(archives)
for a, r in enumerate(res):
    assert r["status"] == 0
    assert [d.tobytes() for d in r["digests"][: len(expect[a])]] == expect[a], a
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
