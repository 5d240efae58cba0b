#!/bin/bash
O=gpurun_out/s6; mkdir -p $O
run() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT: $*"; exit 1; fi; return 0; }
run 1000 python -m pytest tests -m gpu -q -s > $O/pytest_gpu.log 2>&1
grep -E "passed|failed|^FAILED|^ERROR" $O/pytest_gpu.log | tail -20 | cut -c1-250
run 600 python __graft_entry__.py smoke > $O/smoke.log 2>&1
grep -E "^smoke|Error|assert" $O/smoke.log | cut -c1-300 | tail -8
run 600 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
tail -c 2000 $O/bench.json; grep "^bench:" $O/bench.err | head -14
cp gpurun_out/bench_kernels.json $O/bench_kernels.json
