#!/bin/bash
O=gpurun_out/s4; mkdir -p $O
run() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT: $*"; exit 1; fi; return 0; }
run 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -s -k "k2" > $O/pytest_k2.log 2>&1
grep -E "^\s+\[|passed|failed|Error|error" $O/pytest_k2.log | cut -c1-260 | tail -40
run 1000 python -m pytest tests -m gpu -q -s > $O/pytest_gpu.log 2>&1
grep -E "passed|failed|^FAILED|^ERROR" $O/pytest_gpu.log | tail -20 | cut -c1-250
run 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
tail -c 1200 $O/bench.json; grep "^bench:" $O/bench.err | head -14
