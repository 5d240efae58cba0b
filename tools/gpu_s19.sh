#!/bin/bash
O=gpurun_out/s19; mkdir -p $O
for cfg in "GPU_MAX_HW_QUEUES=3" "GPU_MAX_HW_QUEUES=5" "GPU_MAX_HW_QUEUES=6" "KMU_WGRAD_STREAMS=1" "HIP_LAUNCH_BLOCKING=0 AMD_SERIALIZE_KERNEL=0"; do
  env $cfg timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/b.json 2> $O/b.err
  rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT $cfg"; exit 1; fi
  echo "$cfg: $(grep -o '"ms_per_step": [0-9.]*' $O/b.json)"
done
