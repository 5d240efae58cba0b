#!/bin/bash
# step time of the default bench under HIP runtime knobs that affect graph replay / multi-stream dispatch
O=gpurun_out/knobs; mkdir -p $O
run() { local t=$1; shift; timeout -k 10 $t "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT: $*"; exit 1; fi; return 0; }
i=0
for kv in "X=0" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=1" "GPU_MAX_HW_QUEUES=8" "GPU_MAX_HW_QUEUES=2" "HIP_FORCE_DEV_KERNARG=1" "DEBUG_HIP_GRAPH_DOT_PRINT=0 AMD_DIRECT_DISPATCH=1" "HSA_ENABLE_SDMA=0"; do
  i=$((i+1))
  env $kv timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/b$i.json 2> $O/b$i.err
  rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT $kv"; exit 1; fi
  echo "$kv: $(grep -o '"ms_per_step": [0-9.]*' $O/b$i.json) $(grep -o '"loss": [0-9.]*' $O/b$i.json)"
done
