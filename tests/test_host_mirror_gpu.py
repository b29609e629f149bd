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
        if i == 10:                               # letters outside ASCII and outside the BMP fold too
            paths += ["chara/caf\u00e9.tex", "chara/\u0436\u0443\u043a.tex", "chara/\U00010428.tex", "chara/fu\u00df.tex"]
        if i == 11:
            paths += ["CHARA/CAF\u00c9.TEX", "CHARA/\u0416\u0423\u041a.TEX", "CHARA/\U00010400.TEX", "CHARA/FUSS.TEX"]
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
    assert "chara/caf\u00e9.tex" in info["FileReplacements"] and "CHARA/CAF\u00c9.TEX" not in info["FileReplacements"]
    assert "chara/\U00010428.tex" in info["FileReplacements"] and "CHARA/\U00010400.TEX" not in info["FileReplacements"]
    assert "chara/fu\u00df.tex" in info["FileReplacements"] and "CHARA/FUSS.TEX" in info["FileReplacements"]
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


@pytest.mark.parametrize("name", ["..", ".", "/tmp/snappy_escape_test", "a/b", "a\\b", "   "])
def test_import_mcdf_refuses_folder_names_that_leave_the_working_dir(snappy, oracle, tmp_path, name):
    """The Description names the folder that is deleted and recreated; it comes out of a file
    made by someone else.  Anything but a single path component is refused before the file
    system is touched (the reference would hand it to Directory.Delete as it is)."""
    work = tmp_path / "outer" / "snapshots"
    work.mkdir(parents=True)
    (tmp_path / "outer" / "neighbour.txt").write_text("keep")
    (work / "existing").mkdir()
    (work / "existing" / "f.txt").write_text("keep")
    payload = oracle.build_mcdf_payload(name, [(["chara/x.tex"], b"abcdef", 6)])
    mcdf = tmp_path / "evil.mcdf"
    mcdf.write_bytes(oracle.lz4net_encode(payload, chunk_size=4096))
    before = sorted(str(p.relative_to(tmp_path)) for p in tmp_path.rglob("*"))
    with pytest.raises(snappy.native.SnappyError) as e:
        snappy.import_mcdf(str(mcdf), str(work))
    assert e.value.code == snappy.native.E_INVALID_ARG
    assert sorted(str(p.relative_to(tmp_path)) for p in tmp_path.rglob("*")) == before
    assert not os.path.exists("/tmp/snappy_escape_test")


def test_import_mcdf_customize_data_and_escapes(snappy, oracle, tmp_path):
    """CustomizePlusData must be Base64 (Convert.FromBase64String throws after the files have been
    extracted, before anything is saved); JsonConvert escapes U+0085, U+2028 and U+2029."""
    import base64

    files = [(["chara/x.tex"], b"abcdef", 6)]
    work = tmp_path / "w"
    bad = oracle.build_mcdf_payload("bad64", files, extra={"CustomizePlusData": "not*base64"})
    p = tmp_path / "bad64.mcdf"
    p.write_bytes(oracle.lz4net_encode(bad, chunk_size=4096))
    with pytest.raises(snappy.native.SnappyError) as e:
        snappy.import_mcdf(str(p), str(work))
    assert e.value.code == snappy.native.E_MCDF_BASE64
    assert os.listdir(work / "bad64" / "_files") and not (work / "bad64" / "snapshot.json").exists()
    for wrong in ("QUJD=", "QUJ=Q", "QQ=", "Q==="):
        q = tmp_path / "w2.mcdf"
        q.write_bytes(oracle.lz4net_encode(oracle.build_mcdf_payload("w2", files, extra={"CustomizePlusData": wrong}),
                                           chunk_size=4096))
        with pytest.raises(snappy.native.SnappyError) as e:
            snappy.import_mcdf(str(q), str(work))
        assert e.value.code == snappy.native.E_MCDF_BASE64, wrong
    profile = base64.b64encode(b'{"Bones":{}}').decode()
    spaced = profile[:4] + " \r\n" + profile[4:]          # white space is ignored by the managed decoder
    manip = "line\u2028sep para\u2029graph nel\u0085."
    good = oracle.build_mcdf_payload("ok64", files, extra={"CustomizePlusData": spaced, "ManipulationData": manip})
    g = tmp_path / "ok64.mcdf"
    g.write_bytes(oracle.lz4net_encode(good, chunk_size=4096))
    out = snappy.import_mcdf(str(g), str(work))
    cust = json.load(open(os.path.join(out, "customize_history.json")))
    assert cust["Entries"][0]["CustomizeData"] == spaced and cust["Entries"][0]["CustomizeTemplate"] == ""
    assert list(cust["Entries"][0]) == ["Timestamp", "Description", "CustomizeData", "CustomizeTemplate"]
    text = open(os.path.join(out, "snapshot.json"), encoding="utf-8").read()
    assert "\\u2028" in text and "\\u2029" in text and "\\u0085" in text
    assert "\u2028" not in text and "\u2029" not in text and "\u0085" not in text
    assert json.loads(text)["ManipulationString"] == manip


def _folder_state(path):
    """Everything UpdateSnapshot leaves in a snapshot folder, with the time stamps blanked."""
    import re

    files = {name: open(os.path.join(path, "_files", name), "rb").read()
             for name in sorted(os.listdir(os.path.join(path, "_files")))}
    docs = {}
    for name in ("snapshot.json", "glamourer_history.json", "customize_history.json"):
        text = open(os.path.join(path, name), encoding="utf-8").read()
        text = re.sub(r"\d{4}-\d\d-\d\dT\d\d:\d\d:\d\d\.\d{7}Z", "<now>", text)
        text = re.sub(r"Update - \d{4}-\d\d-\d\d \d\d:\d\d:\d\d UTC", "Update - <now> UTC", text)
        docs[name] = text
    return sorted(os.listdir(path)), files, docs


def test_update_snapshot_keeps_json_nulls(snappy, oracle, tmp_path):
    """String properties that are null in the files of an existing snapshot (hand-edited, or written
    by another tool) are assigned as null by JsonConvert and written back as null; a null last
    GlamourerString differs from any new string, so an entry is added."""
    mods = tmp_path / "mods"
    mods.mkdir()
    (mods / "a.tex").write_bytes(H.random_bytes(5000, 1))
    for root in ("ours", "theirs"):
        d = tmp_path / root / "Hero"
        (d / "_files").mkdir(parents=True)
        (d / "snapshot.json").write_text(json.dumps({"FormatVersion": 1, "SourceActor": None, "LastUpdate": None,
                                                     "FileReplacements": {"chara/old.tex": "AB" * 20},
                                                     "ManipulationString": None}))
        (d / "glamourer_history.json").write_text(json.dumps({"Entries": [
            {"Timestamp": "t0", "Description": None, "GlamourerString": None}]}))
        (d / "customize_history.json").write_text(json.dumps({"Entries": [
            {"Timestamp": None, "Description": "d", "CustomizeData": None}]}))
    ours, theirs = tmp_path / "ours" / "Hero", tmp_path / "theirs" / "Hero"
    replacements = [(str(mods / "a.tex"), ["chara/a.tex"])]
    snappy.update_snapshot(str(ours), "Hero", replacements, "", '{"Bones":{}}', "m")
    oracle.update_snapshot(str(theirs), "Hero", replacements, "", '{"Bones":{}}', "m")
    a, b = _folder_state(str(ours)), _folder_state(str(theirs))
    assert a[0] == b[0] and a[1] == b[1]
    for name in a[2]:
        assert a[2][name] == b[2][name], name
    info = json.load(open(ours / "snapshot.json"))
    assert info["SourceActor"] is None and info["ManipulationString"] == "m"
    glam = json.load(open(ours / "glamourer_history.json"))["Entries"]
    assert len(glam) == 2 and glam[0]["GlamourerString"] is None and glam[0]["Description"] is None
    cust = json.load(open(ours / "customize_history.json"))["Entries"]
    assert len(cust) == 2 and cust[0]["CustomizeData"] is None and cust[0]["Timestamp"] is None
    assert cust[0]["CustomizeTemplate"] == ""          # absent in the file: the class's default


def test_update_snapshot_matches_the_reference_loop(snappy, oracle, tmp_path):
    """Save, then save again over the same folder: the folder the library leaves must equal,
    file for file and byte for byte (time stamps aside), the one the oracle's sequential
    restatement of BuildSnapshotFromLocalPlayer + UpdateSnapshot leaves."""
    rng = np.random.default_rng(33)
    mods = tmp_path / "mods"
    mods.mkdir()
    blobs = [H.random_bytes(int(n), 700 + i) for i, n in
             enumerate([0, 5, 70_000, 1_300_000, 2_000_003, 333, 64, 4_100_000])]
    blobs[5] = blobs[2]                                   # same content under two resolved paths
    paths = []
    for i, b in enumerate(blobs):
        p = mods / f"m{i}.tex"
        p.write_bytes(b)
        paths.append(str(p))
    first = [(paths[0], ["chara/empty.tex"]),
             (paths[1], ["chara/a.tex", "chara/A.TEX"]),           # two spellings, one key in snapshot.json
             (str(mods / "missing.mdl"), ["chara/missing.mdl"]),
             (paths[2], ["chara/b.tex"]),
             (paths[3], ["chara/c.tex", 'chara/quo"te.tex']),
             (paths[5], ["chara/b_copy.tex"]),                     # duplicate content: one .dat
             (paths[4], ["chara/b.tex"]),                          # same game path again: last hash wins
             # OrdinalIgnoreCase folds letters outside ASCII too (simple upper-case mapping per
             # character, supplementary planes included); sharp s has no single-letter upper case
             (paths[1], ["chara/\u00e9t\u00e9.tex", "chara/stra\u00dfe.tex", "chara/\u043f\u0440\u0438\u0432\u0435\u0442.tex",
                         "chara/\U00010428.tex", "chara/\u0131.tex", "chara/\u1f80.tex"])]
    second = [(paths[6], ["CHARA/A.tex"]),                         # folds onto chara/a.tex: spelling kept, hash replaced
              (paths[7], ["chara/new.mdl"]),
              (paths[2], ["chara/b.tex"]),
              (paths[6], ["CHARA/\u00c9T\u00c9.TEX",                   # one key with chara/été.tex
                          "CHARA/STRASSE.TEX",                      # NOT the key of straße: a new entry
                          "CHARA/\u041f\u0420\u0418\u0412\u0415\u0422.TEX",   # Cyrillic capitals: one key with привет
                          "CHARA/\U00010400.TEX",                  # Deseret capital long I: one key with U+10428
                          "CHARA/I.TEX",                           # dotless i does not fold onto I: a new entry
                          "CHARA/\u1f88.TEX"])]                    # Greek alpha psili prosgegrammeni: folds onto U+1F80
    ours, theirs = tmp_path / "ours" / "Hero", tmp_path / "theirs" / "Hero"
    for replacements, glam, cplus, manip in ((first, "glam-1", '{"Bones":{}}', "manip-1"),
                                            (second, "glam-1", '{"Bones":{"j_kao":{}}}', "manip\u2028two"),
                                            (second, "glam-3", '{"Bones":{"j_kao":{}}}', "")):
        got = snappy.update_snapshot(str(ours), "Hero", replacements, glam, cplus, manip)
        oracle.update_snapshot(str(theirs), "Hero", replacements, glam, cplus, manip)
        want = [hashlib.sha1(open(r, "rb").read()).hexdigest().upper() if os.path.isfile(r) else None
                for r, _ in replacements]
        assert got == want
        a, b = _folder_state(str(ours)), _folder_state(str(theirs))
        assert a[0] == b[0]
        assert a[1] == b[1]
        for name in a[2]:
            assert a[2][name] == b[2][name], name
    info = json.load(open(ours / "snapshot.json"))
    assert list(info["FileReplacements"])[:2] == ["chara/empty.tex", "chara/a.tex"]
    assert info["FileReplacements"]["chara/a.tex"] == hashlib.sha1(blobs[6]).hexdigest().upper()
    keys = list(info["FileReplacements"])
    for kept, absent in (("chara/\u00e9t\u00e9.tex", "CHARA/\u00c9T\u00c9.TEX"),
                         ("chara/\u043f\u0440\u0438\u0432\u0435\u0442.tex", "CHARA/\u041f\u0420\u0418\u0412\u0415\u0422.TEX"),
                         ("chara/\U00010428.tex", "CHARA/\U00010400.TEX"), ("chara/\u1f80.tex", "CHARA/\u1f88.TEX")):
        assert kept in keys and absent not in keys
        assert info["FileReplacements"][kept] == hashlib.sha1(blobs[6]).hexdigest().upper()
    assert "chara/stra\u00dfe.tex" in keys and "CHARA/STRASSE.TEX" in keys
    assert "chara/\u0131.tex" in keys and "CHARA/I.TEX" in keys
    assert len(json.load(open(ours / "glamourer_history.json"))["Entries"]) == 2
    assert len(json.load(open(ours / "customize_history.json"))["Entries"]) == 2
    # an existing <HASH>.dat is left alone, and a damaged snapshot.json stops the update
    victim = ours / "_files" / (hashlib.sha1(blobs[7]).hexdigest().upper() + ".dat")
    victim.write_bytes(b"keep me")
    snappy.update_snapshot(str(ours), "Hero", second, "glam-3")
    assert victim.read_bytes() == b"keep me"
    (ours / "snapshot.json").write_text("{ not json")
    with pytest.raises(snappy.native.SnappyError) as e:
        snappy.update_snapshot(str(ours), "Hero", second)
    assert e.value.code == snappy.native.E_SNAPSHOT_JSON
