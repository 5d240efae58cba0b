#!/bin/bash
O=gpurun_out/s9; mkdir -p $O
R=/tmp/kmu_prof; rm -rf $R; mkdir -p $R
export TMPDIR=/tmp
run() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT: $*"; exit 1; fi; return 0; }
run 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -s -k "conv3x3" > $O/pytest_conv.log 2>&1
grep -E "\[conv|passed|failed" $O/pytest_conv.log | cut -c1-200 | tail -20
run 1000 python -m pytest tests -m gpu -q -s > $O/pytest_gpu.log 2>&1
grep -E "passed|failed|^FAILED|^ERROR|branch streams\]" $O/pytest_gpu.log | tail -12 | cut -c1-250
run 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
grep -o '"ms_per_step": [0-9.]*' $O/bench.json
run 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/stats -o r02 -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_prof.json 2> $O/bench_prof.err
ST=$(find $R/stats -name "*kernel_stats.csv" | head -1)
cp $ST $O/kernel_stats.csv
python3 tools/profile_summary.py $ST 24 40 > $O/summary.txt 2>&1; head -56 $O/summary.txt | cut -c1-170
