set -u
root=$(pwd); out=$root/gpurun_out/probe; rm -rf $out; mkdir -p $out
export TMPDIR=/tmp SNAPPY_SHA1_CALIBRATE=0
q="--workload small --no-extra --no-e2e --no-cpu-baseline --steps 3 --warmup 1"
for w in 1 2; do
  SNAPPY_SHA1_WAVES_PER_SIMD=$w timeout -k 10 100 python3 -u bench.py $q > $out/w$w.json || exit 1
  python3 -c "
import json; d=json.loads(open('$out/w$w.json').read().strip().splitlines()[-1]); print('waves',$w, d['value'], d['ms_per_step'])"
done
cd /tmp
(cd $root && timeout -s KILL 120 rocprofv3 -d $out/fetch -o fetch --pmc FETCH_SIZE GRBM_GUI_ACTIVE -- python3 bench.py $q) > $out/fetch.log 2>&1 || exit 1
(cd $root && timeout -s KILL 120 rocprofv3 -d $out/tcp -o tcp --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum -- python3 bench.py $q) > $out/tcp.log 2>&1 || echo "tcp pass failed"
(cd $root && timeout -s KILL 120 rocprofv3 -d $out/tcc -o tcc --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -- python3 bench.py $q) > $out/tcc.log 2>&1 || echo "tcc pass failed"
cd $root
python3 tools/rocpd_summary.py $(find $out -name "*.db" | sort) > $out/summary.txt
find $out -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
grep hop_kernel $out/summary.txt
