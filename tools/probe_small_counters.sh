#!/bin/bash
# Counter passes over the small-files workload (one message per lane): what the memory side of
# sha1_hop_kernel does there.  Usage: tools/probe_small_counters.sh  (results: gpurun_out/probe/summary.txt)
set -u
root=$(pwd); out=$root/gpurun_out/probe; rm -rf $out; mkdir -p $out
export TMPDIR=/tmp SNAPPY_SHA1_CALIBRATE=0
q="--workload small --no-extra --no-e2e --no-cpu-baseline --steps 3 --warmup 1"
cd /tmp
pass() {  # name, counters...
    local name=$1; shift
    (cd $root && timeout -s KILL 120 rocprofv3 -d $out/$name -o $name --pmc "$@" -- python3 bench.py $q) > $out/$name.log 2>&1 \
        || echo "pass $name failed (see $name.log)"
}
pass fetch FETCH_SIZE GRBM_GUI_ACTIVE
pass tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
pass tlb TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_SERIALIZATION_STALL_sum
pass lat TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum
pass sq1 SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_ANY SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES
pass sq2 SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_WAIT_ANY SQ_CYCLES SQ_IFETCH SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_SALU
pass ta TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum
cd $root
python3 tools/rocpd_summary.py $(find $out -name "*.db" | sort) > $out/summary.txt
find $out -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
grep hop_kernel $out/summary.txt
