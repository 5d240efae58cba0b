#!/bin/bash
cd "$(dirname "$0")/.."
i=0
for e in "X=1" "MIOPEN_DEBUG_CONV_IMPLICIT_GEMM=0" "MIOPEN_DEBUG_CONV_DIRECT=0" "MIOPEN_DEBUG_CONV_GEMM=0" "MIOPEN_DEBUG_CONV_FFT=0" "MIOPEN_FIND_MODE=1" "MIOPEN_FIND_ENFORCE=3"; do
  i=$((i+1)); rm -rf /tmp/mdb_$i; mkdir -p /tmp/mdb_$i
  echo "--- $e"
  env $e MIOPEN_USER_DB_PATH=/tmp/mdb_$i MIOPEN_CUSTOM_CACHE_DIR=/tmp/mdb_$i/cache timeout -k 10 300 python tools/smoke_bisect.py 2>&1 | grep -E "^glue" || true
done
