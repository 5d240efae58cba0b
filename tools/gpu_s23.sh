#!/bin/bash
O=gpurun_out/s23; mkdir -p $O
for cfg in "KMU_WGRAD_USE_MAIN=0" "KMU_WGRAD_USE_MAIN=1" "KMU_WGRAD_USE_MAIN=0" "KMU_WGRAD_USE_MAIN=1" "KMU_WGRAD_USE_MAIN=1 KMU_WGRAD_STREAMS=2"; do
  env $cfg timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/b.json 2> $O/b.err
  rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT $cfg"; exit 1; fi
  echo "$cfg: $(grep -o '"ms_per_step": [0-9.]*' $O/b.json)"
done
timeout -k 10 300 python -m pytest tests/test_gpu_model.py -m gpu -q -x -k "wgrad" 2>&1 | tail -1
