#!/bin/bash
O=gpurun_out/s7; mkdir -p $O
run() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT: $*"; exit 1; fi; return 0; }
KMU_BRANCH_STREAMS=1 run 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_streams.json 2> $O/bench_streams.err
tail -c 900 $O/bench_streams.json | head -c 500; echo
run 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_plain.json 2> $O/bench_plain.err
tail -c 900 $O/bench_plain.json | head -c 500; echo
