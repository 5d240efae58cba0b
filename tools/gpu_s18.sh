#!/bin/bash
O=gpurun_out/s18; mkdir -p $O
for cfg in "KMU_GROUPED_BRANCHES=0" "KMU_GROUPED_MIN_C=64" "KMU_GROUPED_MIN_C=32" "KMU_GROUPED_MIN_C=0" "KMU_GROUPED_MIN_C=0 KMU_WGRAD_STREAMS=2" "KMU_GROUPED_MIN_C=32 KMU_WGRAD_BATCH=150"; do
  env $cfg timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/b.json 2> $O/b.err
  rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT $cfg"; exit 1; fi
  echo "$cfg: $(grep -o '"ms_per_step": [0-9.]*' $O/b.json)"
done
