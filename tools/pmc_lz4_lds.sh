#!/bin/bash
# rocprofv3 counter passes over the LZ4 decode kernel alone (dense content), one run per
# counter group: LDS instruction counts, LDS-array cycles, unaligned-access stalls, and the
# waves' instruction mix and wait share.  Usage: tools/pmc_lz4_lds.sh <tag>
set -u
tag=${1:-lds}
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
dense="--workload mcdf --mcdf-content dense --archives 8 --archive-mib 256 --mcdf-decode-only"
pass() {  # name, counters...
    local name=$1
    shift
    (cd "$root" && timeout -s KILL 240 rocprofv3 -d "$out/$name" -o "$name" --pmc "$@" -- python3 bench.py $dense) \
        > "$out/$name.log" 2>&1 || { echo "FAILED: $name" >&2; tail -5 "$out/$name.log" >&2; exit 1; }
}
pass pmc_lds SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES
pass pmc_mix SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY
cd "$root"
python3 tools/rocpd_summary.py $(find "$out" -name "*.db" | sort) | grep -v "copyBuffer\|lz4_encode\|lz4_pack\|fillBuffer" > "$out/summary.txt"
find "$out" -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
cat "$out/summary.txt"
