#!/bin/bash
O=gpurun_out/s17; mkdir -p $O
run() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT: $*"; exit 1; fi; return 0; }
run 300 python -m pytest tests/test_gpu_model.py -m gpu -q -s -x -k grouped > $O/pytest_grouped.log 2>&1
grep -E "passed|failed|^FAILED|^ERROR|grouped branches|Error" $O/pytest_grouped.log | tail -12 | cut -c1-300
