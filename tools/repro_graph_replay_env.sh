#!/bin/bash
# failing case (full loss + DropPath, stream sync before replay 3) under different HIP runtime graph knobs
cd "$(dirname "$0")/.."
for e in "X=1" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0" "HIP_FORCE_DEV_KERNARG=0" "HIP_FORCE_DEV_KERNARG=1" "DEBUG_HIP_KERNARG_COPY_OPT=0" "DEBUG_CLR_KERNARG_HDP_FLUSH_WA=1" "DEBUG_HIP_GRAPH_BATCH_SIZE=1"; do
  echo "== $e"
  env $e timeout -k 10 120 python tools/repro_graph_replay_sync.py full 1 2>&1 | grep -E "^full|Error|error" | cut -c1-200 || true
done
