#!/bin/bash
cd "$(dirname "$0")/.."
W=MIOPEN_DEBUG_CONV_WINOGRAD=0; G=MIOPEN_DEBUG_CONV_IMPLICIT_GEMM_ASM_BWD_GTC_XDLOPS_NHWC=0
for e in "X=1" "$W" "$G" "$W $G"; do
  rm -rf ~/.config/miopen ~/.cache/miopen
  env $e timeout -k 10 200 python tools/miopen_conv_accuracy.py ${1:-32} ${2:-2} 2>&1 | grep -vE "amdgpu.ids"
done
