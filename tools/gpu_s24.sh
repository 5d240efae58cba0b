#!/bin/bash
O=gpurun_out/s24; mkdir -p $O
run() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT: $*"; exit 1; fi; return 0; }
run 900 python -m pytest tests/test_gpu_model.py -m gpu -q -x -s > $O/pytest_m.log 2>&1
grep -E "passed|failed|^FAILED|^ERROR|Error|wgrad side" $O/pytest_m.log | tail -6 | cut -c1-220
run 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
echo "bench: $(grep -o '"ms_per_step": [0-9.]*' $O/bench.json)"
run 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench2.json 2> $O/bench2.err
echo "bench: $(grep -o '"ms_per_step": [0-9.]*' $O/bench2.json)"
