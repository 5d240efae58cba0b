#!/bin/bash
cd "$(dirname "$0")/.."
for g in "" gate_mlp iwp mix3 conv3tap pwconv layer_norm group_norm dwconv bn_blend qkv_gate; do
  KMU_GLUE_TORCH=$g timeout -k 10 120 python tools/smoke_bisect.py 2>&1 | grep -E "^glue|worst" || true
done
