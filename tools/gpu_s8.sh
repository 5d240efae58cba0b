#!/bin/bash
O=gpurun_out/s8; mkdir -p $O
run() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT: $*"; exit 1; fi; return 0; }
run 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -s -k "k1 or conv3x3" > $O/pytest_k1.log 2>&1
grep -E "\[k1|\[conv3x3|passed|failed" $O/pytest_k1.log | cut -c1-200 | tail -40
run 1000 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1
grep -E "passed|failed|^FAILED|^ERROR" $O/pytest_gpu.log | tail -12 | cut -c1-250
run 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
grep -o '"ms_per_step": [0-9.]*' $O/bench.json; grep "^bench:" $O/bench.err | head -6
