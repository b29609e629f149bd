#!/bin/bash
# Collects the rocprofv3 evidence for the bench numbers on the GPU box:
#   1. kernel trace of the default bench command (per-kernel durations),
#   2. counter passes (one rocprofv3 run per counter group) for the three kernels'
#      workloads: HBM read/write bytes, shader clock, VALU instruction counts.
# Results land under gpurun_out/<tag>/; tools/rocpd_summary.py turns the databases
# into the text that is committed under profiles/.
# Usage: tools/collect_profiles.sh <tag>
set -u
tag=${1:-prof}
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp

run() {  # name, seconds, rocprofv3 options..., --, program...
    local name=$1 limit=$2
    shift 2
    echo "== $name" >&2
    (cd "$root" && timeout -s KILL "$limit" rocprofv3 -d "$out/$name" -o "$name" "$@") \
        > "$out/$name.log" 2>&1 || { echo "FAILED: $name (see $name.log)" >&2; exit 1; }
}

quick="--no-extra --no-e2e --no-cpu-baseline --steps 2 --warmup 1"
# The planner's one-off calibration launches (four sha1_hop_kernel launches of 0.4-1.8 ms the
# first time a process hashes) would mix into the per-kernel averages: off for these runs.
export SNAPPY_SHA1_CALIBRATE=0

run trace_default 600 --kernel-trace --stats -- python3 bench.py
for w in corpus blobs small; do
    run pmc_${w}_fetch 300 --pmc FETCH_SIZE GRBM_GUI_ACTIVE -- python3 bench.py --workload $w $quick
    run pmc_${w}_write 300 --pmc WRITE_SIZE -- python3 bench.py --workload $w $quick
    run pmc_${w}_sq 300 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY -- python3 bench.py --workload $w $quick
done
# Counter passes serialise kernels, and the import's SHA-1 kernel waits for decode launches that
# would then never start: the counters are taken on the decode kernel alone, one launch over all chunks.
asset="--workload mcdf --archives 8 --archive-mib 256 --mcdf-decode-only"
run pmc_mcdf_fetch 300 --pmc FETCH_SIZE GRBM_GUI_ACTIVE -- python3 bench.py $asset
run pmc_mcdf_write 300 --pmc WRITE_SIZE -- python3 bench.py $asset
dense="--workload mcdf --mcdf-content dense --archives 8 --archive-mib 256 --mcdf-decode-only"
run pmc_dense_fetch 300 --pmc FETCH_SIZE GRBM_GUI_ACTIVE -- python3 bench.py $dense
run pmc_dense_write 300 --pmc WRITE_SIZE -- python3 bench.py $dense
run pmc_dense_sq 300 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY -- python3 bench.py $dense

cd "$root"
python3 tools/rocpd_summary.py $(find "$out" -name "*.db" | sort) > "$out/summary.txt"
# Only the summary and the logs travel back (the databases are hundreds of MB).
find "$out" -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
echo done >&2
