#!/bin/bash
# rocprofv3 session: kernel stats of the bench (graph replay) + PMC passes on the isolated dominant kernels.
# Raw rocprofv3 output goes to /tmp on the box (it exceeds what gpurun copies back); only the summaries land in gpurun_out/.
TAG=${1:-r03}; O=gpurun_out/prof_$TAG; mkdir -p $O
R=/tmp/kmu_prof; rm -rf $R; mkdir -p $R
export TMPDIR=/tmp
run() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT: $*"; exit 1; fi; return 0; }
run 600 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_plain.json 2> $O/bench_plain.err
tail -c 400 $O/bench_plain.json
run 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/stats -o $TAG -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_prof.json 2> $O/bench_prof.err
tail -3 $O/bench_prof.err
ST=$(find $R/stats -name "*kernel_stats.csv" | head -1); echo "stats: $ST"
cp $ST $O/${TAG}_bench_n1_graph_kernel_stats.csv
python3 tools/profile_summary.py $ST 29 70 > $O/summary.txt 2>&1; head -75 $O/summary.txt
run 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/pmc_f -o f -- python3 tools/run_dominant_kernel.py 3 > $O/pmc_f.log 2>&1
run 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/pmc_w -o w -- python3 tools/run_dominant_kernel.py 3 > $O/pmc_w.log 2>&1
run 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $R/pmc_sq -o sq -- python3 tools/run_dominant_kernel.py 3 > $O/pmc_sq.log 2>&1
tail -2 $O/pmc_f.log $O/pmc_sq.log
F=$(find $R/pmc_f -name "*counter_collection.csv" | head -1); W=$(find $R/pmc_w -name "*counter_collection.csv" | head -1); S=$(find $R/pmc_sq -name "*counter_collection.csv" | head -1)
python3 tools/pmc_summary.py $O/pmc_traffic.json f=$F w=$W > $O/pmc_traffic.txt 2>&1
python3 tools/pmc_summary.py $O/pmc_sq.json sq=$S > $O/pmc_sq.txt 2>&1
cat $O/pmc_traffic.txt | cut -c1-200; cat $O/pmc_sq.txt | cut -c1-420
du -sh gpurun_out
