"""The headless callers with their file-system side, against the oracle's restatement of the
reference loops: hash-and-store of a snapshot folder, and ImportMcdf into a snapshot directory."""
import hashlib
import json
import os

import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu


def test_store_files_dedups_by_digest(snappy, oracle, tmp_path):
    src = tmp_path / "mods"
    src.mkdir()
    blobs = [H.random_bytes(n, 40 + i) for i, n in enumerate((0, 17, 70_000, 1_300_000, 17))]
    blobs[4] = blobs[1]                       # duplicate content under another name
    paths = []
    for i, b in enumerate(blobs):
        p = src / f"file_{i}.tex"
        p.write_bytes(b)
        paths.append(str(p))
    paths.insert(2, str(src / "does_not_exist.mdl"))
    files_dir = tmp_path / "snap" / "_files"
    got = snappy.store_files(paths, str(files_dir))
    want = [oracle.get_hash(b) for b in blobs]
    want.insert(2, None)
    assert got == want
    stored = sorted(os.listdir(files_dir))
    assert stored == sorted({h + ".dat" for h in want if h})
    for name in stored:
        data = (files_dir / name).read_bytes()
        assert hashlib.sha1(data).hexdigest().upper() + ".dat" == name
    # an existing <HASH>.dat is left alone (File.Exists guard)
    victim = files_dir / (want[0] + ".dat")
    victim.write_bytes(b"keep me")
    snappy.store_files(paths, str(files_dir))
    assert victim.read_bytes() == b"keep me"
    # hashing only
    assert snappy.store_files(paths[:2]) == want[:2]


def test_store_files_many_files_parallel_reader(snappy, oracle, tmp_path, monkeypatch):
    """A folder's worth of files through the multi-threaded reader and writer, with the
    hash-while-copying transfer forced on; every stored file must carry its own SHA-1 as name."""
    monkeypatch.setenv("SNAPPY_OVERLAP_MIN_BYTES", "0")
    monkeypatch.setenv("SNAPPY_OVERLAP_SLICE_BYTES", "65536")
    rng = np.random.default_rng(77)
    src = tmp_path / "mods"
    src.mkdir()
    sizes = np.concatenate([rng.integers(0, 5000, size=200), rng.integers(50_000, 900_000, size=40),
                            np.array([3_000_001, 0, 64])])
    paths, want = [], []
    for i, n in enumerate(sizes):
        b = H.random_bytes(int(n), 9000 + i % 230)      # some contents repeat
        p = src / f"f{i:04d}.mdl"
        p.write_bytes(b)
        paths.append(str(p))
        want.append(hashlib.sha1(b).hexdigest().upper())
    files_dir = tmp_path / "_files"
    assert snappy.store_files(paths, str(files_dir)) == want
    stored = sorted(os.listdir(files_dir))
    assert stored == sorted({h + ".dat" for h in want})
    for name in stored[::7]:
        assert hashlib.sha1((files_dir / name).read_bytes()).hexdigest().upper() + ".dat" == name


def test_import_mcdf_writes_reference_layout(snappy, oracle, tmp_path):
    rng = np.random.default_rng(5)
    files = []
    for i in range(14):
        n = int(rng.integers(1, 200_000)) if i % 6 else 0
        body = H.asset_like(n, 300 + i) if n else None
        paths = [f"chara/human/c0101/obj/body/b0001/texture/f{i}.tex"]
        if i % 3 == 0:
            paths.append(paths[0].upper())
        if i == 7:
            paths.append("chara/shared.tex")
        if i == 9:
            paths.append("CHARA/SHARED.TEX")      # folds onto i == 7's key: last writer wins
        files.append((paths, body, n))
    payload = oracle.build_mcdf_payload("My Import", files,
                                        extra={"ManipulationData": "bWFuaXA=", "GlamourerData": "glam\"quoted\""})
    stream = oracle.lz4net_encode(payload, chunk_size=1 << 17)
    mcdf = tmp_path / "chara.mcdf"
    mcdf.write_bytes(stream)
    work = tmp_path / "snapshots"
    ref = oracle.mcdf_import(stream)

    # a stale folder of the same name is replaced (Directory.Delete(snapshotPath, true))
    (work / "My Import" / "_files").mkdir(parents=True)
    (work / "My Import" / "_files" / "stale.dat").write_bytes(b"x")
    out = snappy.import_mcdf(str(mcdf), str(work))
    assert out == str(work / "My Import")
    assert sorted(os.listdir(out)) == ["_files", "customize_history.json", "glamourer_history.json",
                                       "snapshot.json"]
    assert sorted(os.listdir(os.path.join(out, "_files"))) == sorted(h + ".dat" for h in ref["blobs"])
    for h, body in ref["blobs"].items():
        assert open(os.path.join(out, "_files", h + ".dat"), "rb").read() == body
    info = json.load(open(os.path.join(out, "snapshot.json")))
    assert list(info) == ["FormatVersion", "SourceActor", "LastUpdate", "FileReplacements",
                          "ManipulationString"]
    assert info["FormatVersion"] == 1 and info["SourceActor"] == "My Import"
    assert info["ManipulationString"] == "bWFuaXA="
    assert info["FileReplacements"] == ref["map"]
    assert list(info["FileReplacements"]) == list(ref["map"])      # Dictionary insertion order
    assert info["LastUpdate"].endswith("Z") and len(info["LastUpdate"]) == 28
    glam = json.load(open(os.path.join(out, "glamourer_history.json")))
    assert glam["Entries"][0]["GlamourerString"] == 'glam"quoted"'
    assert glam["Entries"][0]["Description"] == "Imported from MCDF"
    assert json.load(open(os.path.join(out, "customize_history.json"))) == {"Entries": []}
    text = open(os.path.join(out, "snapshot.json")).read()
    assert text.startswith('{\n  "FormatVersion": 1,\n  "SourceActor": "My Import",')   # Formatting.Indented


def test_import_mcdf_errors(snappy, oracle, tmp_path):
    bad = tmp_path / "bad.mcdf"
    bad.write_bytes(oracle.lz4net_encode(b"NOPE" + bytes(50), chunk_size=4096))
    with pytest.raises(snappy.native.SnappyError) as e:
        snappy.import_mcdf(str(bad), str(tmp_path / "w"))
    assert e.value.code == snappy.native.E_MCDF_MAGIC
    trunc = tmp_path / "trunc.mcdf"
    good = oracle.lz4net_encode(oracle.build_mcdf_payload("t", [(["a"], b"abc", 3)]), chunk_size=4096)
    trunc.write_bytes(good[:-2])
    with pytest.raises(snappy.native.SnappyError) as e:
        snappy.import_mcdf(str(trunc), str(tmp_path / "w"))
    assert e.value.code == snappy.native.E_LZ4_TRUNCATED
    with pytest.raises(snappy.native.SnappyError) as e:
        snappy.import_mcdf(str(tmp_path / "missing.mcdf"), str(tmp_path / "w"))
    assert e.value.code == snappy.native.E_INVALID_ARG


def test_store_files_reads_while_hashing(snappy, tmp_path):
    """A folder large enough for the hash-while-copying path, read by several threads while the
    transfer is under way; one source cannot be opened and is skipped like a missing one."""
    import hashlib
    import os
    rng = np.random.default_rng(8)
    paths, want = [], []
    for i in range(40):
        size = int(rng.integers(1, 6_000_000)) if i else 24_000_000
        data = rng.integers(0, 256, size=size, dtype=np.uint8).tobytes()
        p = tmp_path / f"src{i}.tex"
        p.write_bytes(data)
        paths.append(str(p))
        want.append(hashlib.sha1(data).hexdigest().upper())
    locked = 7
    os.chmod(paths[locked], 0)
    can_open = True
    try:
        open(paths[locked], "rb").close()
    except OSError:
        can_open = False
    files_dir = tmp_path / "_files"
    got = snappy.store_files(paths, str(files_dir))
    os.chmod(paths[locked], 0o644)
    for i, (g, w) in enumerate(zip(got, want)):
        if i == locked and not can_open:
            assert g is None
            assert not (files_dir / f"{w}.dat").exists()
        else:
            assert g == w, i
            assert (files_dir / f"{w}.dat").stat().st_size == os.path.getsize(paths[i])
