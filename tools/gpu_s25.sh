#!/bin/bash
O=gpurun_out/s25; mkdir -p $O
for cfg in "KMU_WGRAD_BATCH=4096" "KMU_WGRAD_BATCH=250" "KMU_WGRAD_BATCH=170" "KMU_WGRAD_BATCH=120" "KMU_WGRAD_BATCH=4096"; do
  env $cfg timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/b.json 2> $O/b.err
  rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT $cfg"; exit 1; fi
  echo "$cfg: $(grep -o '"ms_per_step": [0-9.]*' $O/b.json)"
done
