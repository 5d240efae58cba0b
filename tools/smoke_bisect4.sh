#!/bin/bash
cd "$(dirname "$0")/.."
rm -rf /tmp/mdb_a; mkdir -p /tmp/mdb_a
MIOPEN_USER_DB_PATH=/tmp/mdb_a MIOPEN_CUSTOM_CACHE_DIR=/tmp/mdb_a/cache MIOPEN_ENABLE_LOGGING_CMD=0 timeout -k 10 300 python tools/smoke_bisect.py 2>&1 | grep -E "^glue|worst|params with"
for e in MIOPEN_DEBUG_CONV_IMPLICIT_GEMM_HIP_BWD_V1R1_XDLOPS=0 MIOPEN_DEBUG_CONV_IMPLICIT_GEMM_ASM_BWD_GTC_XDLOPS_NHWC=0 MIOPEN_DEBUG_CONV_IMPLICIT_GEMM_ASM_WRW_GTC_XDLOPS_NHWC=0 MIOPEN_DEBUG_CONV_IMPLICIT_GEMM_ASM_FWD_GTC_XDLOPS_NHWC=0; do
  rm -rf /tmp/mdb_b; mkdir -p /tmp/mdb_b
  echo "--- $e"
  env $e MIOPEN_USER_DB_PATH=/tmp/mdb_b MIOPEN_CUSTOM_CACHE_DIR=/tmp/mdb_b/cache timeout -k 10 300 python tools/smoke_bisect.py 2>&1 | grep -E "^glue|params with"
done
