#!/bin/bash
O=gpurun_out/s22; mkdir -p $O
run() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT: $*"; exit 1; fi; return 0; }
run 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -s -x -k "dagem or k4" > $O/pytest_k.log 2>&1
grep -E "passed|failed|^FAILED|^ERROR|dagem|Error" $O/pytest_k.log | tail -12 | cut -c1-250
run 900 python -m pytest tests/test_gpu_model.py -m gpu -q -x > $O/pytest_m.log 2>&1
grep -E "passed|failed|^FAILED|^ERROR|Error" $O/pytest_m.log | tail -5 | cut -c1-200
run 300 python tools/time_block.py bridge 2>&1 | grep "fwd+bwd"
run 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
echo "bench: $(grep -o '"ms_per_step": [0-9.]*' $O/bench.json)"
