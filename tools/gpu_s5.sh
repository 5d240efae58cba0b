#!/bin/bash
O=gpurun_out/s5; mkdir -p $O
run() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT: $*"; exit 1; fi; return 0; }
run 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -s -k "k2 or evim" > $O/pytest_k2.log 2>&1
grep -E "^\.*F*\s+\[k2|passed|failed" $O/pytest_k2.log | cut -c1-230 | tail -40
run 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
tail -c 700 $O/bench.json; grep "^bench:" $O/bench.err | head -8
