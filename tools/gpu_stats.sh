#!/bin/bash
# rocprofv3 kernel stats of the bench only (no PMC passes): gpu_stats.sh <tag> [ENV=VAL ...]
TAG=$1; shift
O=gpurun_out/stats_$TAG; mkdir -p $O; export TMPDIR=/tmp
R=/tmp/kmu_stats_$TAG; rm -rf $R
for kv in "$@"; do export "$kv"; done
timeout -k 10 400 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_plain.json 2> $O/bench_plain.err
rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo TIMEOUT; exit 1; fi
echo "plain: $(grep -o '"ms_per_step": [0-9.]*' $O/bench_plain.json)"
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R -o s -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_prof.json 2> $O/bench_prof.err
rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo TIMEOUT; exit 1; fi
ST=$(find $R -name "*kernel_stats.csv" | head -1)
cp $ST $O/kernel_stats.csv
python3 tools/profile_summary.py $ST 24 45 > $O/summary.txt 2>&1; head -58 $O/summary.txt | cut -c1-150
