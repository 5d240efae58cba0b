#!/bin/bash
# One GPU-box session: the whole GPU test suite, then the bench (no CPU baseline).  gpurun -- 'bash tools/gpu_check.sh'
O=gpurun_out/check; mkdir -p $O
run() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT: $*"; exit 1; fi; return 0; }
run 1000 python -m pytest tests -m gpu -q -s > $O/pytest_gpu.log 2>&1
grep -E "passed|failed|^FAILED|^ERROR" $O/pytest_gpu.log | tail -8 | cut -c1-200
run 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
grep -o '"ms_per_step": [0-9.]*' $O/bench.json
