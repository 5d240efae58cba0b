#!/bin/bash
cd "$(dirname "$0")/.."
echo "--- plain x2"; timeout -k 10 120 python tools/smoke_bisect.py 2>&1 | grep -E "^glue"; timeout -k 10 120 python tools/smoke_bisect.py 2>&1 | grep -E "^glue"
echo "--- no caching allocator"; PYTORCH_NO_CUDA_MEMORY_CACHING=1 timeout -k 10 200 python tools/smoke_bisect.py 2>&1 | grep -E "^glue"
echo "--- launch blocking"; HIP_LAUNCH_BLOCKING=1 AMD_SERIALIZE_KERNEL=3 timeout -k 10 200 python tools/smoke_bisect.py 2>&1 | grep -E "^glue"
echo "--- winograd on"; MIOPEN_DEBUG_CONV_WINOGRAD=1 timeout -k 10 200 python tools/smoke_bisect.py 2>&1 | grep -E "^glue"
echo "--- miopen immediate"; MIOPEN_FIND_MODE=1 timeout -k 10 200 python tools/smoke_bisect.py 2>&1 | grep -E "^glue"
echo "--- deform fused"; KMU_DEFORM_FUSED=1 timeout -k 10 200 python tools/smoke_bisect.py 2>&1 | grep -E "^glue"
