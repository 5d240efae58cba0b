#!/bin/bash
cd "$(dirname "$0")/.."
i=0
for e in MIOPEN_DEBUG_CONV_IMPLICIT_GEMM_ASM_BWD_GTC_XDLOPS_NHWC=0 MIOPEN_DEBUG_CONV_IMPLICIT_GEMM_ASM_WRW_GTC_XDLOPS_NHWC=0 MIOPEN_DEBUG_CONV_IMPLICIT_GEMM_ASM_FWD_GTC_XDLOPS_NHWC=0; do
  i=$((i+1)); rm -rf /tmp/mdbk_$i; mkdir -p /tmp/mdbk_$i
  echo "--- $e"
  env $e MIOPEN_USER_DB_PATH=/tmp/mdbk_$i MIOPEN_CUSTOM_CACHE_DIR=/tmp/mdbk_$i/cache timeout -k 10 300 python bench.py --no-cpu-baseline 2>/dev/null | cut -c80-200
done
