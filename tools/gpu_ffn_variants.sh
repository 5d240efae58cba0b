#!/bin/bash
# usage: gpu_ffn_variants.sh VARIANT...  -- rocprofv3 kernel durations of tools/bench_ffn.py per library variant ("base" = shipped)
RUNNER=tools/bench_ffn.py bash tools/gpu_variants.sh "ffn_" "$@"
