#!/bin/bash
O=gpurun_out/s16; mkdir -p $O
run() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT: $*"; exit 1; fi; return 0; }
run 600 python -m pytest tests/test_gpu_model.py -m gpu -q -s -x > $O/pytest_gpu.log 2>&1
grep -E "passed|failed|^FAILED|^ERROR|wgrad side|branch streams" $O/pytest_gpu.log | tail -12 | cut -c1-250
run 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
echo "default: $(grep -o '"ms_per_step": [0-9.]*' $O/bench.json)"
