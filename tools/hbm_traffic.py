#!/usr/bin/env python3
"""Turns the counter rows of a tools/collect_profiles.sh summary into HBM bytes per launch of the
dominant kernel of each workload, beside the algorithmic bytes the bench line states.

  python tools/hbm_traffic.py gpurun_out/<tag>/summary.txt <bench line>.json > profiles/hbm_traffic.json

FETCH_SIZE and WRITE_SIZE are reported in KiB; gfx950 tallies a 128-byte read request as 64
bytes, so reads are doubled (writes are not)."""
import json
import re
import sys

PASSES = {  # summary section -> (workload key, kernel)
    "corpus": ("corpus", "sha1_hop_kernel"),
    "blobs": ("blobs", "sha1_hop_kernel"),
    "small": ("small", "sha1_hop_kernel"),
    "mcdf": ("mcdf", "lz4_decode_pair_kernel"),
    "dense": ("mcdf_dense", "lz4_decode_pair_kernel"),
}


def counters(summary_path):
    found, section = {}, None
    for line in open(summary_path):
        m = re.match(r"== .*/pmc_([a-z]+)_(fetch|write|sq)/", line)
        if m:
            section = (m.group(1), m.group(2))
            continue
        if section is None:
            continue
        m = re.match(r"(?:void )?(\w+)(?:<[^>]*>)?\s+(FETCH_SIZE|WRITE_SIZE)\s+\d+\s+\d+\s+([\d.]+)\s*$", line)
        if m:   # kernel (template arguments dropped), counter, sum, dispatches, per dispatch
            found[(section[0], m.group(1), m.group(2))] = float(m.group(3))
    return found


def algorithmic(bench):
    other = bench.get("other_workloads", {})
    out = {"corpus": bench["roofline"].get("algorithmic_bytes_per_launch")}
    for key, name in (("blobs", "blobs"), ("small", "small_files"), ("mcdf", "mcdf_import"),
                      ("mcdf_dense", "mcdf_import_dense")):
        roof = other.get(name, {}).get("roofline", {})
        out[key] = roof.get("algorithmic_bytes_per_launch")
    return out


def main():
    summary, bench_path = sys.argv[1], sys.argv[2]
    bench = json.loads([ln for ln in open(bench_path).read().splitlines() if ln.startswith("{")][-1])
    rows, alg = counters(summary), algorithmic(bench)
    doc = {"_how": "rocprofv3 --pmc FETCH_SIZE (x2: gfx950 tallies 128-byte requests as 64 bytes) and --pmc "
                   "WRITE_SIZE, separate passes (tools/collect_profiles.sh), KiB x 1024, per launch of the dominant "
                   "kernel; built by tools/hbm_traffic.py from the rocprof summary of the same round.  The "
                   "corpus/blobs/small passes regenerate the bench's corpus with the same seed; the LZ4 passes run "
                   "the decode kernel alone, all chunks in one launch (bench.py --mcdf-decode-only)."}
    for section, (key, kernel) in PASSES.items():
        rd = rows.get((section, kernel, "FETCH_SIZE"))
        wr = rows.get((section, kernel, "WRITE_SIZE"))
        if rd is None or wr is None:
            continue
        read, write = rd * 1024 * 2, wr * 1024
        entry = {"kernel": kernel, "hbm_bytes_per_launch": read + write, "read": read, "write": write,
                 "algorithmic": alg.get(key)}
        if alg.get(key):
            entry["ratio"] = (read + write) / alg[key]
        doc[key] = entry
    json.dump(doc, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
